// Fused multi-head attention for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): no [T,T] score / probability tensor in HBM
// in the forward pass, backward by recomputation.
//
//   fs2   (transformer_fs2.py:385-394, F.multi_head_attention_forward): softmax(q k^T / sqrt(d_h), keys >= len masked) v, 2 heads x 128
//   rel   (conformer.py:347-431, RelativeMultiHeadAttention): softmax(((q+u) k^T + shift((q+v) p^T)) / sqrt(d_model)) with NO mask,
//         dropout on the probabilities, 8 heads x 32.  The reference pads PS = (q+v) p^T with a zero column and reinterprets the memory
//         (conformer.py:423-431); element (i, j) of the result is QVsel . Pt[j - i + T - 1] with the extended table
//         Pt = [pos rows | 0 | pos rows] and QVsel = (q+v)[i] below x = T, (q+v)[i+1] above - Toeplitz in (i, j).  Both kernels compute the
//         band a tile needs on the matrix cores and read it along the anti-diagonals: no score tensor, no shifted map, no probabilities
//         or dropped probabilities in HBM.  The backward kernel writes dS in the reference's `padded` layout ([B,H] slabs of T*(T+1)
//         floats, dS[i][j] = slab[(i+1)*T + j]), so the gradient of the unshifted scores is a VIEW of the same memory (rows of T+1, first
//         column skipped) for the two position GEMMs.
//
// Work decomposition: ONE WAVE PER WORKGROUP, 32 queries (forward) or 32 keys (backward) per wave, flash-style loop over the other
// axis in tiles of 32.  fp32 MFMA issues one 32x32x2 step per 64 cycles, so a wave needs only ~1 operand dword per 64 cycles: every
// operand is fetched straight from global / L2 in MFMA fragment order (A/B operand of lane (l31, h) = element [row l31][k-half h]),
// 64-byte runs per lane for the K-contiguous side and fully coalesced 128-byte rows for the transposed side.  No operand tiles in
// LDS and no workgroup barriers; LDS holds the two-block ring of position-score bands (rel forward) and, at d_head > 32, the staged
// Q / dO rows whose transposed reads feed dV^T / dK^T (backward).
//
// Forward computes S^T = K Q^T so that a lane owns one QUERY column: the online-softmax max / sum run over the lane's own 16
// accumulator registers (+ one exchange with lane^32), and P^T is already laid out as the B operand of O^T += V^T P^T.
// Backward (one wave = 32 keys, loop over query tiles) computes S = Q K^T and dP = dO V^T with the keys on the lanes, so that
// Pd and dS are the B operands of dV^T += dO^T Pd and dK^T += Q^T dS; dV / dK stay in registers for the whole loop (no atomics
// unless the query range is split, fs2 only), dS is written once for the remaining gradients dQ = dS K, dQV = dPS pos,
// dpos = dPS^T QV, which are plain GEMMs (rel: dPS is the padded view of the dS memory).
#include "ctts_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr float LOG2E = 1.44269504088896340736f;

struct AttnArgs {
  const float *q, *k, *v;           // head 0 of utterance 0; element (b, t, head, c) = p[b*s + t*ld + head*dh + c]
  long ldq, ldk, ldv, sq, sk, sv;
  float* out; long ldo, so;         // [B,T,H*dh]
  float* lse;                       // [B,H,T]  log2-domain log-sum-exp of the scaled scores
  const float* qv; long ldqv, sqv;  // rel: q + v_bias, same indexing as q (q itself is q + u_bias there)
  const float* pos; long ldpos;     // rel: projected sinusoid rows [T, H*dh] shared by the batch
  const int32_t* lens;              // fs2: valid length per utterance (keys >= len masked, query rows >= len are zero rows), else NULL
  int B, H, T;
  float scale, p_drop;
  const uint64_t* seed; uint32_t drop_offset;
  // backward
  const float* dout; long lddo, sdo;
  const float* D;                   // [B,H,T] rowsum(dO * O)
  float *dk, *dv; long lddk, lddv, sdk, sdv;
  long split_stride;                // q_split > 1: split s writes its partial dK / dV at dk/dv + s * split_stride (summed afterwards)
  float* dS;                        // d loss / d (q k^T + bias), scale included: [B,H,T,T], or (rel) [B,H] slabs of T*(T+1) floats
                                    // whose first T floats are padding: dS[i][j] = slab[(i+1)*T + j] (the reference's `padded` layout)
  int q_split;
};

__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // C/D row of accumulator register r

__device__ __forceinline__ void load16(const float* __restrict__ p, float (&f)[16]) {
  const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const float4 v = q[x];
    f[4 * x + 0] = v.x; f[4 * x + 1] = v.y; f[4 * x + 2] = v.z; f[4 * x + 3] = v.w;
  }
}

// Operand streaming.  A wave is alone on its SIMD in the fs2 shapes (fewer waves than SIMDs), so memory latency has to be hidden INSIDE
// the wave: every MFMA block of 16 steps is preceded by the loads of the NEXT block's operands (ping-pong register buffers indexed by
// compile-time chunk parity), fenced with sched_barrier so that hipcc keeps "issue loads -> 16 MFMAs (1,024 cycles) -> first use".
#define CTTS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }     // v_exp_f32; arguments here are <= 0 or -inf

// ctts_drop_scale(key, idx) with the index hash split into a lane-constant and a wave-uniform part: pre = idx * G + key arrives
// ready-made (one v_add per element instead of a multiply and an add), and the uniform compare u >= p is done on the integer:
// (h >> 8) * 2^-24 >= p  <=>  h >= ceil(p * 2^24) << 8  (both sides exact) - the same keep decisions as every other kernel's.
constexpr uint32_t DROP_G = 0x9E3779B1U;
__device__ __forceinline__ uint32_t drop_threshold(float p) { return ((uint32_t)ceilf(p * 16777216.0f)) << 8; }
__device__ __forceinline__ float drop_scale_pre(uint32_t pre, uint32_t thr, float inv_keep) {
  return ctts_mix32(pre) >= thr ? inv_keep : 0.0f;
}

// Addressing inside the tile loops goes through buffer descriptors: 32-bit byte offsets = (lane constant) + (wave-uniform term computed
// on the SALU); rows / keys outside a tensor are handled by the hardware range check (loads return 0, stores vanish) instead of
// clamps, selects and exec-mask branches around every access.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int ctts_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
constexpr float BIG = 1.0e30f;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bld(rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}
__device__ __forceinline__ void bld16(rsrc_t r, unsigned off, float (&f)[16]) {
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    ctts_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16u * x, 0, 0);
    const float4 v = *reinterpret_cast<float4*>(&u);
    f[4 * x + 0] = v.x; f[4 * x + 1] = v.y; f[4 * x + 2] = v.z; f[4 * x + 3] = v.w;
  }
}


// ---------------------------------------------------------------------------------------------------------------- forward
// occupancy targets (waves per SIMD): d_head 32 (conformer: thousands of waves) 3 forward / 2 backward (no spills), 64 -> 2, 128 -> 1
template <int DH, bool BIAS>
#ifndef CTTS_ATTN_FWD32B_MAXW
#define CTTS_ATTN_FWD32B_MAXW 3
#endif
__global__ __launch_bounds__(64, (DH <= 32 ? (BIAS ? 2 : 3) : (DH <= 64 ? (BIAS ? 1 : 2) : 1)))
__attribute__((amdgpu_waves_per_eu(1, (DH <= 32 && BIAS) ? CTTS_ATTN_FWD32B_MAXW : 8))) void attn_fwd_kernel(const AttnArgs d) {
  constexpr int NC = DH / 32;
  constexpr int GLD = 34;              // row stride of the band ring in LDS: conflict-free for the row writes and the skewed reads
  __shared__ float sg[BIAS ? 64 * GLD : 1];
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * 32, head = blockIdx.y, b = blockIdx.z;
  const int T = d.T, z = b * d.H + head;
  const int L = d.lens ? min(d.lens[b], T) : T;
  float* Ob = d.out + (long)b * d.so + head * DH;
  float* lse = d.lse + (long)z * T;
  const int i = i0 + l31;
  if (i0 >= L) {                       // a tile of padded query rows: defined as zero
    for (int e = lane; e < 32 * DH; e += 64) {
      const int r = e / DH, c = e - r * DH;
      if (i0 + r < T) Ob[(long)(i0 + r) * d.ldo + c] = 0.f;
    }
    if (h == 0 && i < T) lse[i] = 0.f;
    return;
  }
  const float* Qr = d.q + (long)b * d.sq + head * DH + (long)min(i, T - 1) * d.ldq;
  const unsigned ldk4 = 4u * (unsigned)d.ldk, ldv4 = 4u * (unsigned)d.ldv, T4 = 4u * (unsigned)T;
  const rsrc_t rK = make_rsrc(d.k + (long)b * d.sk + head * DH, (unsigned)(T - 1) * ldk4 + 4u * DH);     // rows >= T read as 0
  const rsrc_t rV = make_rsrc(d.v + (long)b * d.sv + head * DH, (unsigned)(T - 1) * ldv4 + 4u * DH);
  const unsigned ka_l = (unsigned)l31 * ldk4 + 64u * h;          // A operand of S^T: key row l31, k-half h
  const unsigned vt_l = 4u * h * ldv4 + 4u * l31;                // A operand of O^T: key row rowmap(st,h), column l31

  float Qf[DH / 2];                    // B operand of S^T = K Q^T: Q[i][32c + 16h + q]
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float t[16];
    load16(Qr + 32 * c + 16 * h, t);
#pragma unroll
    for (int q = 0; q < 16; ++q) Qf[16 * c + q] = t[q];
  }
  floatx16 Ot[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ot[c][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const float sl2 = d.scale * LOG2E;
  const bool do_drop = d.p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(d.seed, d.drop_offset); inv_keep = 1.f / (1.f - d.p_drop); }
  const uint32_t dthr = drop_threshold(d.p_drop);
  // dropout element index ((z*T + i)*T + j), j = j0 + rowmap(r,0) + 4h: lane part of the hash input idx * G + key
  const uint32_t xk_l = ((((uint32_t)z * (uint32_t)T + (uint32_t)i) * (uint32_t)T) + 4u * h) * DROP_G + dkey;
  const float b_one = h == 0 ? 1.f : 0.f;

  float kb[2][16], vb[2][16];          // ping-pong operand buffers (chunk parity)
  bld16(rK, ka_l, kb[0]);
  // ---- relative position scores, computed in the kernel (no [T,T] score tensor).  conformer.py:423-431 pads PS = QV pos^T with a zero
  // column and reinterprets the memory; element (i, j) of the result is
  //     x = j - i + T - 1:   x <= T-1: QV[i] . pos[x]      x == T: 0      x >= T+1: QV[i+1] . pos[x - T - 1]
  // i.e. QVsel . Pt[x] with the extended table Pt = [pos rows 0..T-1 | 0 | pos rows 0..T-2] - Toeplitz in (i, j).  A key tile needs
  // the band x0 .. x0+62, x0 = j0 - i0 + T - 32; bands move by 32 per key tile, so every tile adds ONE 32-row block
  // G^T[mm][i] = Pt[xb + mm] . QVsel[i] (16 MFMA steps per chunk; a block lies wholly on one side of x = T, so QVsel is uniform), kept in
  // a two-block LDS ring; lane i then reads its 16 scores of the tile along the anti-diagonal: mm = 31 - (i - i0) + (j - j0).
  float QVf[BIAS ? DH / 2 : 1], QVn[BIAS ? DH / 2 : 1], pa[BIAS ? DH / 2 : 1];
  const unsigned ldp4 = BIAS ? 4u * (unsigned)d.ldpos : 0u;
  const rsrc_t rP = make_rsrc(BIAS ? d.pos + head * DH : nullptr, BIAS ? (unsigned)(T - 1) * ldp4 + 4u * DH : 0u);
  const int xb0 = T - 32 - i0;         // first band row of block 0 (tile 0 reads blocks 0 and 1)
  auto load_band = [&](int bidx) {     // A operand: band row xb + l31, k-half h
    const int x = xb0 + 32 * bidx + l31;
    const int prow = x <= T - 1 ? x : x - T - 1;
    const unsigned off = (x < 0 || x == T) ? OOB : (unsigned)prow * ldp4 + 64u * h;      // rows past the table: out of range -> 0
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float t[16];
      bld16(rP, off + 128u * c, t);
#pragma unroll
      for (int q = 0; q < 16; ++q) pa[BIAS ? 16 * c + q : 0] = t[q];
    }
  };
  auto band_block = [&](int bidx) {    // G^T block from `pa` into ring slot bidx & 1
    const bool lower = xb0 + 32 * bidx + 31 <= T - 1;
    floatx16 G;
#pragma unroll
    for (int r = 0; r < 16; ++r) G[r] = 0.f;
#pragma unroll
    for (int q = 0; q < DH / 2; ++q)
      G = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[BIAS ? q : 0], lower ? QVf[BIAS ? q : 0] : QVn[BIAS ? q : 0], G, 0, 0, 0);
    float* slot = sg + (bidx & 1) * 32 * GLD;
#pragma unroll
    for (int r = 0; r < 16; ++r) slot[rowmap(r, h) * GLD + l31] = G[r];
  };
  if (BIAS) {
    const float* QVr = d.qv + (long)b * d.sqv + head * DH;
    const rsrc_t rQV = make_rsrc(QVr, (unsigned)(T - 1) * 4u * (unsigned)d.ldqv + 4u * DH);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float t[16];
      bld16(rQV, (unsigned)i * 4u * (unsigned)d.ldqv + 64u * h + 128u * c, t);             // rows >= T read as 0
#pragma unroll
      for (int q = 0; q < 16; ++q) QVf[BIAS ? 16 * c + q : 0] = t[q];
      bld16(rQV, (unsigned)(i + 1) * 4u * (unsigned)d.ldqv + 64u * h + 128u * c, t);
#pragma unroll
      for (int q = 0; q < 16; ++q) QVn[BIAS ? 16 * c + q : 0] = t[q];
    }
    load_band(0);
    band_block(0);
    load_band(1);
  }
  for (int j0 = 0; j0 < L; j0 += 32) {
    const unsigned s_k = (unsigned)j0 * ldk4, s_v = (unsigned)j0 * ldv4;
    // ---- S^T[j][i] = sum_k K[j][k] Q[i][k]
    floatx16 St;
#pragma unroll
    for (int r = 0; r < 16; ++r) St[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
        bld16(rK, ka_l + s_k + 128u * (c + 1), kb[(c + 1) & 1]);
      } else {                          // last score chunk: fetch the first V chunk of this tile underneath it
#pragma unroll
        for (int st = 0; st < 16; ++st) vb[0][st] = bld(rV, vt_l + s_v + (unsigned)rowmap(st, 0) * ldv4);
      }
      CTTS_SCHED_FENCE();
#pragma unroll
      for (int q = 0; q < 16; ++q) St = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[c & 1][q], Qf[16 * c + q], St, 0, 0, 0);
      CTTS_SCHED_FENCE();
    }
    if (j0 + 32 > L)                    // ragged last tile: S^T[j][i] -= BIG for the keys j >= L (one MFMA step instead of 16 selects)
      St = __builtin_amdgcn_mfma_f32_32x32x2f32((h == 0 && j0 + l31 >= L) ? -BIG : 0.f, b_one, St, 0, 0, 0);
    float s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = St[r];
    if (BIAS) {
      const int tb = j0 >> 5;
      band_block(tb + 1);              // this tile's second block (the first one is the previous tile's second)
      load_band(tb + 2);               // next tile's: a whole tile of cover
      // The workgroup is ONE wave and the LDS executes a wave's instructions in order, so the skewed read below sees the writes
      // above without a barrier.  __syncthreads() here would cost far more than its s_barrier: its fence waits vmcnt(0), i.e. it
      // drains every prefetch in flight.  wave_barrier only pins the program order.
      __builtin_amdgcn_wave_barrier();
      const int mm0 = 31 - l31 + 32 * (tb & 1);          // ring row of band element mm: (mm + 32 * (tb & 1)) & 63
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] += sg[((mm0 + rowmap(r, h)) & 63) * GLD + l31];
      __builtin_amdgcn_wave_barrier();
    }
    // ---- online softmax over the keys (log2 domain); a lane owns query column i, its partner lane^32 the other 16 key rows
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] *= sl2;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m, tmax);               // > -BIG*sl2: the tile holds at least one key < L
    const float alpha = fast_exp2(m - mnew);
    float psum = 0.f;
    floatx16 Pt;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = fast_exp2(s[r] - mnew);
      psum += p;
      if (do_drop) p *= drop_scale_pre(xk_l + (uint32_t)(j0 + rowmap(r, 0)) * DROP_G, dthr, inv_keep);
      Pt[r] = p;
    }
    lsum = lsum * alpha + psum;
    m = mnew;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ot[c][r] *= alpha;
    // ---- O^T[d][i] += sum_j V[j][d] P^T[j][i]: MFMA step st consumes key rows rowmap(st, 0) and rowmap(st, 1) = accumulator register st
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
#pragma unroll
        for (int st = 0; st < 16; ++st) vb[(c + 1) & 1][st] = bld(rV, vt_l + s_v + (unsigned)rowmap(st, 0) * ldv4 + 128u * (c + 1));
      } else {                          // last PV chunk: fetch the next tile's first K chunk underneath it (rows past the end read as 0)
        bld16(rK, ka_l + s_k + 32u * ldk4, kb[0]);
      }
      CTTS_SCHED_FENCE();
#pragma unroll
      for (int st = 0; st < 16; ++st) Ot[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[c & 1][st], Pt[st], Ot[c], 0, 0, 0);
      CTTS_SCHED_FENCE();
    }
  }
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = 1.f / lsum;
  if (i < T) {
    const bool live = i < L;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ob[(long)i * d.ldo + 32 * c + rowmap(r, h)] = live ? Ot[c][r] * inv : 0.f;
    if (h == 0) lse[i] = live ? m + log2f(lsum) : 0.f;
  }
}

// --------------------------------------------------------------------------------------------------------------- backward
// Row statistics reach the accumulator layout through the matrix core itself: one extra 32x32x2 step adds -lse_i/sl2 (k = 0) and
// -BIG * [key masked] (k = 1) to S, a second one broadcasts D_i - no cross-lane lookups, selects or branches in the element loop.
#ifndef CTTS_ATTN_BWD32_WAVES
#define CTTS_ATTN_BWD32_WAVES 2
#endif
template <int DH, bool BIAS>
__global__ __launch_bounds__(64, (DH <= 32 ? CTTS_ATTN_BWD32_WAVES : 1)) void attn_bwd_kernel(const AttnArgs d) {
  constexpr int NC = DH / 32;
  constexpr bool RES = DH <= 128;       // K / V fragments of the wave's 32 keys stay in registers
  // d_head > 32 (one wave per SIMD): a tile is L2-bound when Q and dO are fetched in both operand layouts (rows for S / dP, columns for
  // dV^T / dK^T).  The row fragments are written to LDS as they are consumed and the transposed operands are read back from there.
  constexpr bool STAGE = DH > 32;
  constexpr int SLD = DH + 4;           // row stride: conflict-free 16-byte row writes and column reads
  __shared__ __attribute__((aligned(16))) float t_q[STAGE ? 32 * SLD : 1], t_do[STAGE ? 32 * SLD : 1];
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  const int j0 = blockIdx.x * 32, split = blockIdx.y, z = blockIdx.z;
  const int b = z / d.H, head = z - b * d.H;
  const int T = d.T;
  const int L = d.lens ? min(d.lens[b], T) : T;
  const int j = j0 + l31;
  float* dKb = d.dk + (long)b * d.sdk + head * DH + (long)split * d.split_stride;
  float* dVb = d.dv + (long)b * d.sdv + head * DH + (long)split * d.split_stride;
  if (j0 >= L) {                        // masked keys receive no gradient (every split zeroes its own partial)
    if (j < T)
      for (int c = h; c < DH; c += 2) { dKb[(long)j * d.lddk + c] = 0.f; dVb[(long)j * d.lddv + c] = 0.f; }
    return;
  }
  const float* Qb = d.q + (long)b * d.sq + head * DH;
  const float* dOb = d.dout + (long)b * d.sdo + head * DH;
  const float* Kr = d.k + (long)b * d.sk + head * DH + (long)min(j, T - 1) * d.ldk + 16 * h;
  const float* Vr = d.v + (long)b * d.sv + head * DH + (long)min(j, T - 1) * d.ldv + 16 * h;
  const float* lse = d.lse + (long)z * T;
  const float* Dz = d.D + (long)z * T;
  const unsigned ldq4 = 4u * (unsigned)d.ldq, lddo4 = 4u * (unsigned)d.lddo, T4 = 4u * (unsigned)T;
  const unsigned slab = T4 * (unsigned)(BIAS ? T + 1 : T);                 // bytes of one [T,T] (rel: padded) slab (the C ABI checks < 2^31)
  const rsrc_t rQ = make_rsrc(Qb, (unsigned)(T - 1) * ldq4 + 4u * DH);      // rows >= T are out of range
  const rsrc_t rdO = make_rsrc(dOb, (unsigned)(T - 1) * lddo4 + 4u * DH);
  const rsrc_t rdS = make_rsrc(d.dS + (long)z * (slab / 4u), slab);
  // rel: position scores recomputed per tile (see the forward kernel): G[i][mm] = QVsel[i] . Pt[x0 + mm], x0 = j0 - i0 + T - 32
  const unsigned ldqv4 = BIAS ? 4u * (unsigned)d.ldqv : 0u, ldp4 = BIAS ? 4u * (unsigned)d.ldpos : 0u;
  const rsrc_t rQV = make_rsrc(BIAS ? d.qv + (long)b * d.sqv + head * DH : nullptr, BIAS ? (unsigned)(T - 1) * ldqv4 + 4u * DH : 0u);
  const rsrc_t rP = make_rsrc(BIAS ? d.pos + head * DH : nullptr, BIAS ? (unsigned)(T - 1) * ldp4 + 4u * DH : 0u);
  const unsigned qva_l = (unsigned)l31 * ldqv4 + 64u * h;
  // lane-constant byte offsets; the wave-uniform (query tile, register) terms are added per access
  const unsigned qa_l = (unsigned)l31 * ldq4 + 64u * h, da_l = (unsigned)l31 * lddo4 + 64u * h;     // A operands: row l31, k-half h
  const unsigned qt_l = 4u * h * ldq4 + 4u * l31, dt_l = 4u * h * lddo4 + 4u * l31;                  // transposed: row rowmap(st,h), column l31
  const unsigned x_l = 4u * h * (unsigned)T + (unsigned)j;                  // element (row 4h, key j) of a [T,T] map
  const unsigned e_l = j < T ? 4u * x_l + (BIAS ? T4 : 0u) : OOB;           // its byte offset in the dS slab; keys >= T: none

  float Kf[RES ? DH / 2 : 1], Vf[RES ? DH / 2 : 1];
  if (RES) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float t[16];
      load16(Kr + 32 * c, t);
#pragma unroll
      for (int q = 0; q < 16; ++q) Kf[RES ? 16 * c + q : 0] = t[q];
      load16(Vr + 32 * c, t);
#pragma unroll
      for (int q = 0; q < 16; ++q) Vf[RES ? 16 * c + q : 0] = t[q];
    }
  }
  floatx16 dKt[NC], dVt[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dKt[c][r] = 0.f; dVt[c][r] = 0.f; }
  const float sl2 = d.scale * LOG2E, inv_sl2 = 1.f / sl2;
  const bool do_drop = d.p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(d.seed, d.drop_offset); inv_keep = 1.f / (1.f - d.p_drop); }
  const uint32_t zTT = (uint32_t)z * (uint32_t)T * (uint32_t)T;
  const uint32_t dthr = drop_threshold(d.p_drop);
  const uint32_t xk_l = x_l * DROP_G + dkey;                                // lane part of hash input (z*T*T + ir*T + j) * G + key
  // B operands of the two statistic steps (lane <-> key j, k-half h)
  const float b_aug = h == 0 ? 1.f : (j < L ? 0.f : 1.f);
  const float b_one = h == 0 ? 1.f : 0.f;

  const int nq = (L + 31) / 32;
  const int per = (nq + d.q_split - 1) / max(d.q_split, 1);
  const int q_lo = split * per, q_hi = min(nq, q_lo + per);
  // ping-pong operand buffers (chunk parity): Q / dO rows as A operands of S and dP, K / V chunks as their B operands when they
  // are streamed, dO^T / Q^T rows as A operands of dV^T and dK^T
  float qa[2][16], da[2][16], kb[2][16], vb[2][16], dot[2][16], qtt[2][16];
  if (q_lo < q_hi) {
    bld16(rQ, qa_l + (unsigned)(q_lo * 32) * ldq4, qa[0]);
    if (!RES) load16(Kr, kb[0]);
  }
  // rel: the two band blocks of a tile (B operands: lane <-> band row, k-half h) and the QVsel rows (A operands) are fetched while the
  // previous tile's dV / dK MFMAs run and are dead after the G MFMAs (held longer they would push the kernel out of 2 waves per SIMD)
  float pb[2][BIAS ? DH / 2 : 1], qva[BIAS ? DH / 2 : 1];
  auto load_band = [&](int xb, float (&dst)[BIAS ? DH / 2 : 1]) {
    const int x = xb + l31;
    const int prow = x <= T - 1 ? x : x - T - 1;
    const unsigned off = (x < 0 || x == T) ? OOB : (unsigned)prow * ldp4 + 64u * h;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float t[16];
      bld16(rP, off + 128u * c, t);
#pragma unroll
      for (int q = 0; q < 16; ++q) dst[BIAS ? 16 * c + q : 0] = t[q];
    }
  };
  auto load_qv = [&](int row0) {        // A operand rows row0 + l31 (past the end: 0)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float t[16];
      bld16(rQV, qva_l + (unsigned)row0 * ldqv4 + 128u * c, t);
#pragma unroll
      for (int q = 0; q < 16; ++q) qva[BIAS ? 16 * c + q : 0] = t[q];
    }
  };
  auto band_upper = [&](int xb) -> int { return xb + 31 <= T - 1 ? 0 : 1; };     // a block lies wholly on one side of x = T
  auto load_rel_operands = [&](int it0) {
    const int x0 = j0 - it0 + T - 32;
    load_band(x0, pb[0]);
    load_band(x0 + 32, pb[1]);
    load_qv(it0 + band_upper(x0));
  };
  if (BIAS && q_lo < q_hi) load_rel_operands(q_lo * 32);
  float bias_n[16];
  const int skew_l = 31 - 4 * h + l31;                  // source lane of register r: (skew_l - rowmap(r,0)) & 31, same half
  for (int qt = q_lo; qt < q_hi; ++qt) {
    const int i0 = qt * 32;
    const unsigned s_q = (unsigned)i0 * ldq4, s_do = (unsigned)i0 * lddo4;
    // row statistics of this tile (lanes 0..31 and 32..63 again hold rows i0 .. i0+31): consumed after the S MFMAs
    const int ia = min(i0 + l31, T - 1);
    const bool rowok = i0 + l31 < L;
    const float lse_l = lse[ia], D_l = Dz[ia];
    // keep hipcc from hoisting the 16 per-register variants of each lane constant out of the loop (48 VGPRs -> spills): they are one
    // v_add with a scalar away
    unsigned e_i = e_l; uint32_t xk_i = xk_l; int skew_i = skew_l;
    asm volatile("" : "+v"(e_i), "+v"(xk_i), "+v"(skew_i));
    if (BIAS) {
      // ---- position scores of this tile: G_t[i][mm] = QVsel_t[i] . Pt[x0 + 32t + mm]; element (i, j) is G[i][31 - (i-i0) + (j-j0)]:
      // the source lane picks the block it has to supply, one rotation per register brings it to the key's lane
      const int x0 = j0 - i0 + T - 32;
      const int up0 = band_upper(x0), up1 = band_upper(x0 + 32);
      floatx16 G0, G1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { G0[r] = 0.f; G1[r] = 0.f; }
#pragma unroll
      for (int q = 0; q < DH / 2; ++q) G0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qva[BIAS ? q : 0], pb[0][BIAS ? q : 0], G0, 0, 0, 0);
      if (up1 != up0) load_qv(i0 + up1);                // the one tile per wave that straddles x = T (uniform branch)
#pragma unroll
      for (int q = 0; q < DH / 2; ++q) G1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qva[BIAS ? q : 0], pb[1][BIAS ? q : 0], G1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rm0 = rowmap(r, 0);
        const float v = (l31 + 4 * h >= 31 - rm0) ? G0[r] : G1[r];
        const int src = ((skew_i - rm0) & 31) | (h << 5);
        bias_n[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v)));
      }
    }
    // ---- S[i][j] = sum_k Q[i][k] K[j][k]
    floatx16 S, dP;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
        bld16(rQ, qa_l + s_q + 128u * (c + 1), qa[(c + 1) & 1]);
        if (!RES) load16(Kr + 32 * (c + 1), kb[(c + 1) & 1]);
      } else {
        bld16(rdO, da_l + s_do, da[0]);
        if (!RES) load16(Vr, vb[0]);
      }
      CTTS_SCHED_FENCE();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float bq;
        if constexpr (RES) bq = Kf[16 * c + q]; else bq = kb[c & 1][q];
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[c & 1][q], bq, S, 0, 0, 0);
      }
      if constexpr (STAGE) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          *reinterpret_cast<float4*>(t_q + l31 * SLD + 32 * c + 16 * h + 4 * x) =
              make_float4(qa[c & 1][4 * x], qa[c & 1][4 * x + 1], qa[c & 1][4 * x + 2], qa[c & 1][4 * x + 3]);
      }
      CTTS_SCHED_FENCE();
    }
    // S[i][j] += -lse_i / sl2  - BIG * [row i or key j masked]   ->   P = exp2(sl2 * (S + bias)) needs no selects
    S = __builtin_amdgcn_mfma_f32_32x32x2f32((h == 0 && rowok) ? -lse_l * inv_sl2 : -BIG, b_aug, S, 0, 0, 0);
    if (BIAS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] += bias_n[r];
    }
    // ---- dPd[i][j] = sum_k dO[i][k] V[j][k]
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
        bld16(rdO, da_l + s_do + 128u * (c + 1), da[(c + 1) & 1]);
        if (!RES) load16(Vr + 32 * (c + 1), vb[(c + 1) & 1]);
      } else {                          // operands of the first dV^T / dK^T chunk
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          const unsigned sr = (unsigned)(i0 + rowmap(st, 0));
          if constexpr (STAGE) {          // chunk 0 columns were staged by the first dP / S chunk of this tile
            dot[0][st] = t_do[rowmap(st, h) * SLD + l31];
            qtt[0][st] = t_q[rowmap(st, h) * SLD + l31];
          } else {
            dot[0][st] = bld(rdO, dt_l + sr * lddo4);
            qtt[0][st] = bld(rQ, qt_l + sr * ldq4);
          }
        }
      }
      CTTS_SCHED_FENCE();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float bq;
        if constexpr (RES) bq = Vf[16 * c + q]; else bq = vb[c & 1][q];
        dP = __builtin_amdgcn_mfma_f32_32x32x2f32(da[c & 1][q], bq, dP, 0, 0, 0);
      }
      if constexpr (STAGE) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          *reinterpret_cast<float4*>(t_do + l31 * SLD + 32 * c + 16 * h + 4 * x) =
              make_float4(da[c & 1][4 * x], da[c & 1][4 * x + 1], da[c & 1][4 * x + 2], da[c & 1][4 * x + 3]);
      }
      CTTS_SCHED_FENCE();
    }
    // ---- P = exp2(sl2 * (S + bias)),  Pd = P*keep/(1-p),  dS = scale * P * (dPd*keep/(1-p) - D);  Dbc[i][j] = D_i via one MFMA step
    floatx16 Pd, dSv, Dbc;
    {
      floatx16 zero;
#pragma unroll
      for (int r = 0; r < 16; ++r) zero[r] = 0.f;
      Dbc = __builtin_amdgcn_mfma_f32_32x32x2f32((h == 0 && rowok) ? D_l : 0.f, b_one, zero, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int sc = i0 + rowmap(r, 0);                 // uniform: the h = 0 half's query row of register r
      const float p = fast_exp2(S[r] * sl2);
      float ks = 1.f;
      if (do_drop) ks = drop_scale_pre(xk_i + (zTT + (uint32_t)sc * (uint32_t)T) * DROP_G, dthr, inv_keep);
      const float ds = p * (dP[r] * ks - Dbc[r]) * d.scale;
      Pd[r] = p * ks;
      dSv[r] = ds;
      bst(rdS, e_i + (unsigned)sc * T4, ds);             // masked rows / keys inside the slab receive their exact value 0
    }
    // ---- dV^T[d][j] += sum_i dO[i][d] Pd[i][j],  dK^T[d][j] += sum_i Q[i][d] dS[i][j]
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          const unsigned sr = (unsigned)(i0 + rowmap(st, 0));
          if constexpr (STAGE) {
            dot[(c + 1) & 1][st] = t_do[rowmap(st, h) * SLD + 32 * (c + 1) + l31];
            qtt[(c + 1) & 1][st] = t_q[rowmap(st, h) * SLD + 32 * (c + 1) + l31];
          } else {
            dot[(c + 1) & 1][st] = bld(rdO, dt_l + sr * lddo4 + 128u * (c + 1));
            qtt[(c + 1) & 1][st] = bld(rQ, qt_l + sr * ldq4 + 128u * (c + 1));
          }
        }
      } else {                          // next query tile's first S operands (rows past the end read as 0)
        bld16(rQ, qa_l + s_q + 32u * ldq4, qa[0]);
        if (!RES) load16(Kr, kb[0]);
        if (BIAS) load_rel_operands(i0 + 32);
      }
      CTTS_SCHED_FENCE();
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        dVt[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(dot[c & 1][st], Pd[st], dVt[c], 0, 0, 0);
        dKt[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(qtt[c & 1][st], dSv[st], dKt[c], 0, 0, 0);
      }
      CTTS_SCHED_FENCE();
    }
  }
  if (j < T) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dKb[(long)j * d.lddk + 32 * c + rowmap(r, h)] = dKt[c][r];
        dVb[(long)j * d.lddv + 32 * c + rowmap(r, h)] = dVt[c][r];
      }
  }
}

// dK | dV of the packed [B,T,3C] gradient = sum over the query-loop splits of the partials [S][B*T][2C] (fixed order: deterministic)
__global__ void attn_sum_splits_kernel(const float4* __restrict__ part, int n_split, long rows, int C2_4, float4* __restrict__ dqkv, int C3_4,
                                       int off4) {
  const long total = rows * C2_4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / C2_4; const int c = (int)(e - r * C2_4);
    float4 a = part[e];
    for (int s = 1; s < n_split; ++s) {
      const float4 v = part[(long)s * total + e];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    dqkv[r * C3_4 + off4 + c] = a;
  }
}

// padded score slabs ([T, T+1] = [T+1, T] floats): the zero column the shift inserts (forward), the T floats in front of dS (backward)
__global__ void pad_zero_kernel(float* p, long n, long outer_stride, int inner, int inner_stride) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long o = e / inner;
    p[o * outer_stride + (e - o * inner) * inner_stride] = 0.f;
  }
}

template <bool BIAS>
int launch_fwd(const AttnArgs& a, int dh, hipStream_t st) {
  dim3 grid((a.T + 31) / 32, a.H, a.B);
  switch (dh) {
    case 32: hipLaunchKernelGGL((attn_fwd_kernel<32, BIAS>), grid, dim3(64), 0, st, a); break;
    case 64: hipLaunchKernelGGL((attn_fwd_kernel<64, BIAS>), grid, dim3(64), 0, st, a); break;
    case 128: hipLaunchKernelGGL((attn_fwd_kernel<128, BIAS>), grid, dim3(64), 0, st, a); break;
    default: ctts_set_error("fused attention: head size %d is not instantiated (32, 64, 128)", dh); return -1;
  }
  CTTS_CHECK_LAUNCH("attn_fwd");
  return 0;
}

template <bool BIAS>
int launch_bwd(const AttnArgs& a, int dh, hipStream_t st) {
  dim3 grid((a.T + 31) / 32, max(a.q_split, 1), a.B * a.H);
  switch (dh) {
    case 32: hipLaunchKernelGGL((attn_bwd_kernel<32, BIAS>), grid, dim3(64), 0, st, a); break;
    case 64: hipLaunchKernelGGL((attn_bwd_kernel<64, BIAS>), grid, dim3(64), 0, st, a); break;
    case 128: hipLaunchKernelGGL((attn_bwd_kernel<128, BIAS>), grid, dim3(64), 0, st, a); break;
    default: ctts_set_error("fused attention: head size %d is not instantiated (32, 64, 128)", dh); return -1;
  }
  CTTS_CHECK_LAUNCH("attn_bwd");
  return 0;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// ------------------------------------------------------------------------------------------------------------------- C ABI
extern "C" int ctts_mha_supported(int C, int H) {
  if (H <= 0 || C % H) return 0;
  const int dh = C / H;
  return dh == 32 || dh == 64 || dh == 128;
}

extern "C" int ctts_mha_fwd(const float* qkv, const int32_t* lens, float* out, float* lse, int B, int T, int H, int C, float scale,
                            void* stream) {
  CTTS_REQUIRE(qkv && out && lse && B > 0 && T > 0, "ctts_mha_fwd: bad arguments");
  CTTS_REQUIRE(ctts_mha_supported(C, H) && al16(qkv) && al16(out), "ctts_mha_fwd: needs d_head in {32,64,128} and 16-byte aligned tensors");
  AttnArgs a = {};
  const long C3 = 3L * C;
  a.q = qkv; a.k = qkv + C; a.v = qkv + 2 * C;
  a.ldq = a.ldk = a.ldv = C3; a.sq = a.sk = a.sv = (long)T * C3;
  a.out = out; a.ldo = C; a.so = (long)T * C;
  a.lse = lse; a.lens = lens; a.B = B; a.H = H; a.T = T; a.scale = scale;
  return launch_fwd<false>(a, C / H, (hipStream_t)stream);
}

extern "C" int ctts_mha_bwd(const float* qkv, const int32_t* lens, const float* out, const float* dout, const float* lse, float* Dws,
                            float* dS, float* kv_part, float* dqkv, int B, int T, int H, int C, float scale, int q_split, void* ws,
                            void* stream) {
  CTTS_REQUIRE(qkv && out && dout && lse && Dws && dS && dqkv && B > 0 && T > 0, "ctts_mha_bwd: bad arguments");
  CTTS_REQUIRE(q_split <= 1 || kv_part, "ctts_mha_bwd: q_split > 1 needs the kv_part scratch [q_split, B, T, 2C]");
  CTTS_REQUIRE((long)T * T * 4 < 0x7FFFFFFFL && (long)T * C * 12 < 0x7FFFFFFFL, "ctts_mha_bwd: one utterance exceeds the 32-bit byte offsets of the kernel");
  CTTS_REQUIRE(ctts_mha_supported(C, H) && al16(qkv) && al16(dout) && al16(dqkv), "ctts_mha_bwd: needs d_head in {32,64,128} and 16-byte aligned tensors");
  hipStream_t st = (hipStream_t)stream;
  const int dh = C / H;
  const long C3 = 3L * C;
  if (ctts_zero_async(dqkv, sizeof(float) * (size_t)B * T * C3, st) != 0) return -2;     // dQ rows >= len, split accumulation targets
  int rc = ctts_rowdot_heads(dout, out, Dws, B, T, H, dh, stream);                       // D = rowsum(dO * O)
  if (rc) return rc;
  AttnArgs a = {};
  a.q = qkv; a.k = qkv + C; a.v = qkv + 2 * C;
  a.ldq = a.ldk = a.ldv = C3; a.sq = a.sk = a.sv = (long)T * C3;
  a.lse = const_cast<float*>(lse); a.lens = lens; a.B = B; a.H = H; a.T = T; a.scale = scale;
  a.dout = dout; a.lddo = C; a.sdo = (long)T * C; a.D = Dws;
  a.dS = dS; a.q_split = q_split < 1 ? 1 : q_split;
  if (a.q_split == 1) {
    a.dk = dqkv + C; a.dv = dqkv + 2 * C; a.lddk = a.lddv = C3; a.sdk = a.sdv = (long)T * C3;
  } else {                               // every split of a key tile's query loop writes its own partial; summed below, no atomics
    a.dk = kv_part; a.dv = kv_part + C; a.lddk = a.lddv = 2L * C; a.sdk = a.sdv = (long)T * 2 * C;
    a.split_stride = (long)B * T * 2 * C;
  }
  rc = launch_bwd<false>(a, dh, st);
  if (rc) return rc;
  if (a.q_split > 1) {
    const long rows = (long)B * T, total = rows * (2 * C / 4);
    hipLaunchKernelGGL(attn_sum_splits_kernel, dim3((unsigned)min((total + 255) / 256, 8192L)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(kv_part), a.q_split, rows, 2 * C / 4, reinterpret_cast<float4*>(dqkv), (int)(C3 / 4), C / 4);
    CTTS_CHECK_LAUNCH("ctts_mha_bwd(sum splits)");
  }
  // dQ[i,d] = sum_j dS[i,j] K[j,d]  (valid queries / keys only); the reduction over <= T keys is split so that the launch fills the chip
  ctts_gemm_desc g = {};
  g.A = dS; g.B = qkv + C; g.C = dqkv;
  g.M = T; g.N = dh; g.K = T; g.lda = T; g.ldb = C3; g.ldc = C3; g.a_kc = 1; g.b_kc = 0;
  g.nb0 = B; g.nb1 = H; g.sA0 = (long)H * T * T; g.sA1 = (long)T * T; g.sB0 = (long)T * C3; g.sB1 = dh; g.sC0 = (long)T * C3; g.sC1 = dh;
  g.lens = lens; g.lim_m = lens ? 1 : 0; g.lim_k = lens ? 1 : 0;
  g.alpha = 1.f;
  const long tiles = (long)((T + 63) / 64) * ((dh + 63) / 64) * B * H;
  // (the two pieces are summed in a fixed order through the caller's workspace; without one: no split)
  g.split_k = (ws && tiles < 1536 && T >= 512) ? 2 : 1;
  if (g.split_k > 1) { g.sk_ws = ws; g.sk_ws_bytes = (int64_t)CTTS_WS_BYTES; }
  return ctts_gemm(&g, stream);
}

extern "C" size_t ctts_relmha_workspace_floats(int B, int T, int H) { return (size_t)B * H * T * (T + 1); }

namespace {
int relmha_check_sizes(int B, int T, int H, int C, const char* who) {
  CTTS_REQUIRE((long)T * (T + 1) * 4 < 0x7FFFFFFFL && (long)T * C * 8 < 0x7FFFFFFFL && (long)B * H * T * T < 0xFFFFFFFFL,
               "%s: one utterance exceeds the 32-bit offsets of the kernel", who);
  return 0;
}
int pad_zero(float* p, long outer, long outer_stride, int inner, int inner_stride, hipStream_t st) {
  const long n = outer * inner;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pad_zero_kernel, dim3((unsigned)min((n + 255) / 256, 4096L)), dim3(256), 0, st, p, n, outer_stride, inner, inner_stride);
  CTTS_CHECK_LAUNCH("ctts_relmha(pad)");
  return 0;
}
}  // namespace

extern "C" int ctts_relmha_fwd(const float* qu, const float* qv, const float* kv, const float* pos, float* out, float* lse,
                               int B, int T, int H, int C, float scale, float p_drop, const uint64_t* seed, uint32_t drop_offset,
                               void* stream) {
  CTTS_REQUIRE(qu && qv && kv && pos && out && lse && B > 0 && T > 0, "ctts_relmha_fwd: bad arguments");
  CTTS_REQUIRE(ctts_mha_supported(C, H) && al16(qu) && al16(qv) && al16(kv) && al16(pos) && al16(out),
               "ctts_relmha_fwd: needs d_head in {32,64,128} and 16-byte aligned tensors");
  CTTS_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "ctts_relmha_fwd: p_drop out of range");
  if (relmha_check_sizes(B, T, H, C, "ctts_relmha_fwd")) return -1;
  AttnArgs a = {};
  a.q = qu; a.ldq = C; a.sq = (long)T * C;
  a.k = kv; a.v = kv + C; a.ldk = a.ldv = 2L * C; a.sk = a.sv = (long)T * 2 * C;
  a.out = out; a.ldo = C; a.so = (long)T * C; a.lse = lse;
  a.qv = qv; a.ldqv = C; a.sqv = (long)T * C; a.pos = pos; a.ldpos = C;
  a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.drop_offset = drop_offset;
  return launch_fwd<true>(a, C / H, (hipStream_t)stream);
}

extern "C" int ctts_relmha_bwd(const float* qu, const float* qv, const float* kv, const float* pos, const float* out,
                               const float* dout, const float* lse, float* Dws, float* dS, float* dqu, float* dqv, float* dkv,
                               float* dpos_b, int B, int T, int H, int C, float scale, float p_drop, const uint64_t* seed,
                               uint32_t drop_offset, void* stream) {
  CTTS_REQUIRE(qu && qv && kv && pos && out && dout && lse && Dws && dS && dqu && dqv && dkv && dpos_b && B > 0 && T > 0,
               "ctts_relmha_bwd: bad arguments");
  CTTS_REQUIRE(ctts_mha_supported(C, H) && al16(qu) && al16(qv) && al16(kv) && al16(pos) && al16(dout) && al16(dkv), "ctts_relmha_bwd: needs d_head in {32,64,128} and 16-byte aligned tensors");
  if (relmha_check_sizes(B, T, H, C, "ctts_relmha_bwd")) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int dh = C / H;
  const long slab = (long)T * (T + 1);
  int rc = ctts_rowdot_heads(dout, out, Dws, B, T, H, dh, stream);
  if (rc) return rc;
  rc = pad_zero(dS, (long)B * H, slab, T, 1, st);          // the T floats in front of every dS map: row 0 of the padded view reads them
  if (rc) return rc;
  AttnArgs a = {};
  a.q = qu; a.ldq = C; a.sq = (long)T * C;
  a.k = kv; a.v = kv + C; a.ldk = a.ldv = 2L * C; a.sk = a.sv = (long)T * 2 * C;
  a.lse = const_cast<float*>(lse);
  a.qv = qv; a.ldqv = C; a.sqv = (long)T * C; a.pos = pos; a.ldpos = C;
  a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.drop_offset = drop_offset;
  a.dout = dout; a.lddo = C; a.sdo = (long)T * C; a.D = Dws;
  a.dk = dkv; a.dv = dkv + C; a.lddk = a.lddv = 2L * C; a.sdk = a.sdv = (long)T * 2 * C;
  a.dS = dS; a.q_split = 1;
  rc = launch_bwd<true>(a, dh, st);
  if (rc) return rc;
  ctts_gemm_desc g = {};
  // dQU[i,d] = sum_j dS[i,j] K[j,d]                      (dS map = slab + T, rows of T)
  g.A = dS + T; g.B = kv; g.C = dqu; g.M = T; g.N = dh; g.K = T; g.lda = T; g.ldb = 2L * C; g.ldc = C; g.a_kc = 1; g.b_kc = 0;
  g.nb0 = B; g.nb1 = H; g.sA0 = (long)H * slab; g.sA1 = slab; g.sB0 = (long)T * 2 * C; g.sB1 = dh; g.sC0 = (long)T * C; g.sC1 = dh;
  g.alpha = 1.f; g.split_k = 1;
  rc = ctts_gemm(&g, stream);
  if (rc) return rc;
  // dQV[i,d] = sum_m dPS[i,m] pos[m,d]                   (dPS = the padded view of the same memory: slab + 1, rows of T+1)
  g.A = dS + 1; g.lda = T + 1; g.B = pos; g.C = dqv; g.ldb = C; g.sB0 = 0; g.sB1 = dh;
  rc = ctts_gemm(&g, stream);
  if (rc) return rc;
  // dpos_b[b][m,d] = sum_i dPS[b][i,m] QV[b][i,d]   (caller sums over b: pos is shared by the batch)
  g.B = qv; g.C = dpos_b; g.a_kc = 0; g.b_kc = 0; g.ldb = C; g.ldc = C;
  g.sB0 = (long)T * C; g.sB1 = dh; g.sC0 = (long)T * C; g.sC1 = dh;
  return ctts_gemm(&g, stream);
}
