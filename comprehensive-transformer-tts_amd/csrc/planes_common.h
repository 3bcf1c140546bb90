// The exact three-way bf16 split of an fp32 value and the plane-set layout (include/ctts.h ctts_split_planes) as device helpers, shared by
// the split kernel (gemm_pl.hip) and by the PRODUCERS that write the plane set of their output next to the fp32 tensor (round 6:
// LayerNorm forward, BatchNorm apply / backward, the producer-epilogue backward of the weight-stationary GEMM) - one arithmetic, so a set
// made by a producer is bit-identical to the set ctts_split_planes would make from the fp32 tensor.
#pragma once
#include "ctts_common.h"

namespace {

typedef unsigned int spl_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 spl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float spl_floatx2 __attribute__((ext_vector_type(2)));

// one float -> (hi bits, mid bits, lo bits) with the domain rules of include/ctts.h
__device__ __forceinline__ void spl_one(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const spl_floatx2 v0 = {x, 0.f};
  unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, spl_bf16x2)) & 0xFFFFu;
  const bool x_fin = fabsf(x) < __builtin_inff();
  if ((hb & 0x7F80u) == 0x7F80u) {               // hi is +-inf / NaN
    if (x_fin) hb = (hb & 0x8000u) | 0x7F7Fu;    // a finite x that would round to infinity: the largest bf16 (the remainder stays exact)
    else { hi = hb; mid = 0u; lo = 0u; return; }
  }
  const float r1 = x - __uint_as_float(hb << 16);
  const spl_floatx2 v1 = {r1, 0.f};
  const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, spl_bf16x2)) & 0xFFFFu;
  const float r2 = r1 - __uint_as_float(mb << 16);
  const spl_floatx2 v2 = {r2, 0.f};
  const unsigned lb = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, spl_bf16x2)) & 0xFFFFu;
  hi = hb; mid = mb; lo = lb;
}

// plane set dst [rows][ld / 32][3][32] bf16: the 8 consecutive k = c .. c + 7 (c % 8 == 0) of row r -> three 16-byte stores 64 bytes apart
// inside the K-block's 192 bytes
__device__ __forceinline__ void spl_store8(uint16_t* __restrict__ dst, long r, long ld, long c, const float (&xs)[8]) {
  unsigned hi[8], mid[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) spl_one(xs[e], hi[e], mid[e], lo[e]);
  spl_u32x4 ph, pm, pq;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ph[e] = hi[2 * e] | (hi[2 * e + 1] << 16);
    pm[e] = mid[2 * e] | (mid[2 * e + 1] << 16);
    pq[e] = lo[2 * e] | (lo[2 * e + 1] << 16);
  }
  uint16_t* o = dst + r * ld * 3 + (c >> 5) * 96 + (c & 31);
  *reinterpret_cast<spl_u32x4*>(o) = ph;
  *reinterpret_cast<spl_u32x4*>(o + 32) = pm;
  *reinterpret_cast<spl_u32x4*>(o + 64) = pq;
}

// the 4 consecutive k = c .. c + 3 (c % 4 == 0) of row r -> three 8-byte stores
__device__ __forceinline__ void spl_store4(uint16_t* __restrict__ dst, long r, long ld, long c, const float (&xs)[4]) {
  unsigned hi[4], mid[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) spl_one(xs[e], hi[e], mid[e], lo[e]);
  uint16_t* o = dst + r * ld * 3 + (c >> 5) * 96 + (c & 31);
  *reinterpret_cast<uint2*>(o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
  *reinterpret_cast<uint2*>(o + 32) = make_uint2(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16));
  *reinterpret_cast<uint2*>(o + 64) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
}

}  // namespace
