// Pieces shared by the GEMM translation units (gemm.hip: tile-per-workgroup kernels, gemm_sk.hip: persistent stream-K kernel).
#pragma once
#include "ctts_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

// ---- epilogue shared by the kernels.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int MT, int NT>
__device__ __forceinline__ void gemm_epilogue(const ctts_gemm_desc& d, floatx16 (&acc)[MT][NT], float* Cb, int z, int row0, int col0,
                                              int wm0, int wn0, int l31, int h, int Mv, int Nv) {
  const float alpha = d.alpha;
  // split_k > 1 never reaches an epilogue: the kernels finish such launches through gemm_splitk_finish (ordered, no atomics)
  const bool do_drop = d.p_drop > 0.f;
  uint32_t dkey = 0;
  float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(d.seed, d.drop_offset); inv_keep = 1.f / (1.f - d.p_drop); }
  const uint32_t zoff = (uint32_t)z * (uint32_t)d.M * (uint32_t)d.N;
  if (d.E) {           // fused softmax backward: dS = P * (dP - D)
    const float* Eb = d.E + (Cb - d.C);
    const float* rs = d.rowsub + (long)z * d.M;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = col0 + wn0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < Mv && n < Nv) Cb[(long)m * d.ldc + n] = Eb[(long)m * d.ldc + n] * (alpha * acc[i][j][r] - rs[m]);
        }
      }
    return;
  }
  if (d.epi_bwd) {       // backward of the previous layer's epilogue: C = alpha * acc * drop_mask * act'(Z_prev)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = col0 + wn0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < Mv && n < Nv) {
            float v = alpha * acc[i][j][r];
            if (do_drop) v *= ctts_drop_scale(dkey, zoff + (uint32_t)m * (uint32_t)d.N + (uint32_t)n, d.p_drop, inv_keep);
            if (d.act) v *= ctts_act_grad(d.Z[(long)m * d.ldz + n], d.act);
            Cb[(long)m * d.ldc + n] = v;
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = col0 + wn0 + j * 32 + l31;
      const float bv = (d.bias && n < Nv) ? d.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < Mv && n < Nv) {
          float v = alpha * (acc[i][j][r] + bv);
          if (d.Z) d.Z[(long)m * d.ldz + n] = v;
          v = ctts_act(v, d.act);
          if (do_drop) v *= ctts_drop_scale(dkey, zoff + (uint32_t)m * (uint32_t)d.N + (uint32_t)n, d.p_drop, inv_keep);
          if (d.R) v += d.R[(long)m * d.ldr + n];
          if (d.rowscale) v *= d.rowscale[m];
          Cb[(long)m * d.ldc + n] = v;
        }
      }
    }
}


// ---- lean epilogue: the arithmetic of gemm_epilogue above, bit for bit, for the unbatched launches that carry the train step.
// Activation / dropout / direction / gathered operand are compile-time constants, and every memory operation is a raw buffer access
// whose descriptor ends at the last valid element: rows >= M fall out of range in hardware (loads return 0, stores are dropped), lanes
// with n >= N carry an out-of-range offset.  No exec masking, no 64-bit address arithmetic, straight-line code: hipcc batches the loads
// in front of the arithmetic and issues the stores back to back.  Measured on the persistent kernels: the generic epilogue (per element:
// exec-masked branch, 64-bit multiply-add address, every activation behind uniform branches) cost 4 - 8 % of the FFN convolutions of
// transformer_fs2 and twice the MFMA time of a tile in the K = 256 kernel.
//   RS: the row scale (non-pad mask) exists.  AUX: the gathered operand of the direction exists - the residual R (forward) or the stored pre-activation Z (backward, ACT != 0).
//   Forward with ACT != 0 stores the pre-activation to Z; without Z (inference) that store runs into an empty descriptor.
constexpr unsigned GEMM_OOB = 0x80000000u;

// one float -> the bit patterns of its three bf16 pieces: the arithmetic of planes_common.h spl_one / include/ctts.h ctts_split_planes,
// restated here so that gemm_common.h does not pull the plane headers into every GEMM translation unit
typedef __bf16 gemm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gemm_floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gemm_split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const gemm_floatx2 v0 = {x, 0.f};
  unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, gemm_bf16x2)) & 0xFFFFu;
  const bool x_fin = fabsf(x) < __builtin_inff();
  if ((hb & 0x7F80u) == 0x7F80u) {
    if (x_fin) hb = (hb & 0x8000u) | 0x7F7Fu;
    else { hi = hb; mid = 0u; lo = 0u; return; }
  }
  const float r1 = x - __uint_as_float(hb << 16);
  const gemm_floatx2 v1 = {r1, 0.f};
  const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, gemm_bf16x2)) & 0xFFFFu;
  const float r2 = r1 - __uint_as_float(mb << 16);
  const gemm_floatx2 v2 = {r2, 0.f};
  const unsigned lb = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, gemm_bf16x2)) & 0xFFFFu;
  hi = hb; mid = mb; lo = lb;
}
__device__ __host__ __forceinline__ bool gemm_fits32(const void* p, long M, long ld, long N);

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gemm_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// Batched / length-limited launches pass their batch base Cb, batch index z and limits Mv x Nv (only the combination without
// gathered operands is dispatched for them: R, Z and rowscale are not batched).
// PLANES (round 6; only the weight-stationary kernel's backward instantiations ask for it - the 64-VGPR tile kernels must not carry the
// code): when d.C_planes is set, the bf16 plane set of the stored values is written too.  A lane owns ONE column and 8 rows of a batch;
// a 32-bit plane store needs two neighbouring columns of one row: lanes l and l ^ 1 swap halves with one DPP move per piece - the even
// lane ends up with columns (n, n + 1) of the first row of a row pair, the odd lane with (n - 1, n) of the second.
template <int MT, int NT, int ACT, bool DROP, bool BWD, bool AUX, bool RS = false, bool PLANES = false>
__device__ __forceinline__ void gemm_epilogue_lean(const ctts_gemm_desc& d, const floatx16 (&acc)[MT][NT], float* Cb, int z, int row0,
                                                   int col0, int wm0, int wn0, int l31, int h, int Mv, int Nv) {
#pragma clang fp contract(off)
  const bool planes_on = PLANES && d.C_planes != nullptr;
  const __amdgpu_buffer_rsrc_t rp = gemm_rsrc(d.C_planes, planes_on ? (long)Mv * d.ldc * 6 : 0);
  const unsigned p_row = (unsigned)(d.ldc * 6);
  const unsigned p_sel = (l31 & 1) ? 0x03020706u : 0x05040100u;       // v_perm_b32 selectors: odd lane (nbr.hi16, own.hi16), even lane (own.lo16, nbr.lo16)
  const float alpha = d.alpha;
  uint32_t dkey = 0;
  float inv_keep = 1.f;
  if (DROP) { dkey = ctts_drop_key(d.seed, d.drop_offset); inv_keep = 1.f / (1.f - d.p_drop); }
  const uint32_t zoff = (uint32_t)z * (uint32_t)d.M * (uint32_t)d.N;
  // dropout element index zoff + m * N + n, hashed as idx * G + key: (lane constant) + (wave-uniform term of the row mu)
  const uint32_t dthr = DROP ? ctts_drop_threshold(d.p_drop) : 0u;
  const uint32_t drow = (uint32_t)d.N * CTTS_DROP_G;
  const long last = Mv - 1;
  const float* aux_p = BWD ? d.Z : d.R;
  const long aux_ld = BWD ? d.ldz : d.ldr;
  const bool has_z = !BWD && ACT && d.Z;
  const __amdgpu_buffer_rsrc_t rc = gemm_rsrc(Cb, (last * d.ldc + Nv) * 4);
  const __amdgpu_buffer_rsrc_t ra = gemm_rsrc(aux_p, AUX ? (last * aux_ld + Nv) * 4 : 0);
  const __amdgpu_buffer_rsrc_t rz = gemm_rsrc(d.Z, has_z ? (last * d.ldz + Nv) * 4 : 0);
  const __amdgpu_buffer_rsrc_t rr = gemm_rsrc(d.rowscale, RS ? (long)Mv * 4 : 0);
  const unsigned c_row = (unsigned)(d.ldc * 4), a_row = (unsigned)(aux_ld * 4), z_row = (unsigned)(d.ldz * 4);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = col0 + wn0 + j * 32 + l31;
    const bool n_ok = n < Nv;
    const unsigned c_lane = n_ok ? (unsigned)(4 * h) * c_row + (unsigned)n * 4u : GEMM_OOB;      // byte offset of (row 4h, column n)
    const unsigned a_lane = n_ok ? (unsigned)(4 * h) * a_row + (unsigned)n * 4u : GEMM_OOB;
    const unsigned z_lane = n_ok ? (unsigned)(4 * h) * z_row + (unsigned)n * 4u : GEMM_OOB;
    const float bv = (!BWD && d.bias && n_ok) ? d.bias[n] : 0.f;
    const uint32_t dlane = (zoff + (uint32_t)(4 * h) * (uint32_t)d.N + (uint32_t)n) * CTTS_DROP_G + dkey;
    // plane byte offset of (row 4h + (lane odd), column pair n & ~1): K-block n / 32 at + 192 bytes each, piece q at + 64 q
    const unsigned p_lane = n_ok ? (unsigned)(4 * h + (l31 & 1)) * p_row + (unsigned)(n >> 5) * 192u + (unsigned)(n & 30) * 2u : GEMM_OOB;
#pragma unroll
    for (int ib = 0; ib < 2 * MT; ++ib) {               // batches of 8 rows: 8 gathered values in registers at a time
      const int i = ib >> 1, rb = (ib & 1) * 8;
      float aux[8], rs[8];
      if (AUX) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = rb + q, mu = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2);      // the row of h = 0: wave-uniform
          aux[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, a_lane + (unsigned)mu * a_row, 0, 0));
        }
      }
      if (RS) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = rb + q, mu = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2);
          rs[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (unsigned)(mu + 4 * h) * 4u, 0, 0));
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = rb + q, mu = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2);
        const int m = mu + 4 * h;
        float v;
        if (BWD) {
          v = alpha * acc[i][j][r];
          if (DROP) v *= ctts_drop_scale_pre(dlane + (uint32_t)mu * drow, dthr, inv_keep);
          if (ACT) v *= ctts_act_grad(aux[q], ACT);
        } else {
          v = alpha * (acc[i][j][r] + bv);
          if (ACT) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rz, z_lane + (unsigned)mu * z_row, 0, 0);
          v = ctts_act(v, ACT);
          if (DROP) v *= ctts_drop_scale_pre(dlane + (uint32_t)mu * drow, dthr, inv_keep);
          if (AUX) v += aux[q];
          if (RS) v *= rs[q];
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, c_lane + (unsigned)mu * c_row, 0, 0);
        if (PLANES) aux[q] = v;            // the gathered operand is consumed: its registers carry the stored values to the split below
      }
      if (PLANES) {
        if (planes_on) {
#pragma unroll
          for (int q = 0; q < 8; q += 2) {         // rows mu, mu + 1 (q even: (r & 3) even, so r + 1 is the next row of the same group of four)
            const int r = rb + q, mu = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2);
            unsigned ha, ma, la, hb, mb, lb;
            gemm_split3(aux[q], ha, ma, la);
            gemm_split3(aux[q + 1], hb, mb, lb);
            const unsigned own[3] = {ha | (hb << 16), ma | (mb << 16), la | (lb << 16)};
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
              const unsigned nbr = (unsigned)__builtin_amdgcn_mov_dpp((int)own[pc], 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]: lane ^ 1
              const unsigned out = __builtin_amdgcn_perm(nbr, own[pc], p_sel);
              __builtin_amdgcn_raw_buffer_store_b32(out, rp, p_lane + (unsigned)mu * p_row + (unsigned)(pc * 64), 0, 0);
            }
          }
        }
      }
    }
  }
}

// C += alpha * acc with the same addressing (weight-gradient accumulation of the persistent kernel; last step of an ordered split-K sum).
// Cb / Mv / Nv: the batch base and limits of batched, length-limited launches.
template <int MT, int NT>
__device__ __forceinline__ void gemm_accumulate_lean(const ctts_gemm_desc& d, const floatx16 (&acc)[MT][NT], float* Cb, int row0, int col0,
                                                     int wm0, int wn0, int l31, int h, int Mv, int Nv) {
  const __amdgpu_buffer_rsrc_t rc = gemm_rsrc(Cb, ((long)(Mv - 1) * d.ldc + Nv) * 4);
  const unsigned c_row = (unsigned)(d.ldc * 4);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = col0 + wn0 + j * 32 + l31;
    const unsigned c_lane = n < Nv ? (unsigned)(4 * h) * c_row + (unsigned)n * 4u : GEMM_OOB;
#pragma unroll
    for (int ib = 0; ib < 2 * MT; ++ib) {
      const int i = ib >> 1, rb = (ib & 1) * 8;
      float old[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = rb + q, mu = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2);
        old[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, c_lane + (unsigned)mu * c_row, 0, 0));
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = rb + q, mu = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2);
        const float v = old[q] + d.alpha * acc[i][j][r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, c_lane + (unsigned)mu * c_row, 0, 0);
      }
    }
  }
}

// ---- split-K without atomics (round 4).  grid.y = split_k workgroups reduce disjoint K ranges of one output tile.  Each STORES its
// partial tile into the partial matrix P_s [M, ldp] of its split in the caller's workspace - ctts_gemm hands the tile kernels a rewritten
// descriptor whose C is that area, so this is their ordinary plain epilogue; a second launch (splitk_reduce_kernel, gemm.hip) adds the
// P_s in split order and accumulates alpha * sum into the caller's C.  The sum
// has a fixed order, so it is bit-reproducible from run to run - buffer_atomic_add_f32 in arrival order was not (profiles/
// r04_diag_determinism_*_before.txt) - and the GEMM kernels carry no tickets, flags or extra registers for it (a first version gathered
// the slabs inside the GEMM's epilogue: 65 - 69 VGPRs instead of <= 64 on the 64 x 64 kernels, i.e. 7 instead of 8 waves per SIMD for
// EVERY launch, and a 25 us serial gather at the tail of every 31-way split).
// P_s of (batch z, split s): the partial matrices live in the slab area of the workspace, [z][s][M][ldp], ldp = N rounded up to 4
__device__ __host__ __forceinline__ long gemm_partial_ld(int N) { return (long)((N + 3) / 4) * 4; }
__device__ __forceinline__ float* gemm_partial_base(const ctts_gemm_desc& d, int z, int split) {
  float* P = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(d.sk_ws) + CTTS_WS_SLABS);
  return P + ((long)z * d.split_k + split) * (long)d.M * gemm_partial_ld(d.N);
}

// fused softmax backward of the attention launches: C = E * (alpha * acc - rowsub[row]), E laid out like C
template <int MT, int NT>
__device__ __forceinline__ void gemm_softmax_bwd_lean(const ctts_gemm_desc& d, const floatx16 (&acc)[MT][NT], float* Cb, int z, int row0,
                                                      int col0, int wm0, int wn0, int l31, int h, int Mv, int Nv) {
  const long bytes = ((long)(Mv - 1) * d.ldc + Nv) * 4;
  const __amdgpu_buffer_rsrc_t rc = gemm_rsrc(Cb, bytes);
  const __amdgpu_buffer_rsrc_t re = gemm_rsrc(d.E + (Cb - d.C), bytes);
  const __amdgpu_buffer_rsrc_t rr = gemm_rsrc(d.rowsub + (long)z * d.M, (long)Mv * 4);
  const unsigned c_row = (unsigned)(d.ldc * 4);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      rs[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (unsigned)(row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 4u, 0, 0));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = col0 + wn0 + j * 32 + l31;
      const unsigned c_lane = n < Nv ? (unsigned)(4 * h) * c_row + (unsigned)n * 4u : GEMM_OOB;
#pragma unroll
      for (int rb = 0; rb < 16; rb += 8) {
        float e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = rb + q, mu = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2);
          e[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(re, c_lane + (unsigned)mu * c_row, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = rb + q, mu = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2);
          const float v = e[q] * (d.alpha * acc[i][j][r] - rs[r]);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, c_lane + (unsigned)mu * c_row, 0, 0);
        }
      }
    }
  }
}

// 32-bit byte offsets reach every element of an [M, ld] operand (with room for a tile of rows beyond M)
__device__ __host__ __forceinline__ bool gemm_fits32(const void* p, long M, long ld, long N) {
  return !p || ((M + 128) * ld * 4 < 0x7FFF0000L && ld >= N);
}

// The epilogue of an unbatched tile: the lean variant of the combinations the train steps launch (tools/profile_gemm_shapes.py prints the
// signature of every launch) - including the softmax-backward fusion and the plain
// batched / length-limited attention products - and the generic one for everything else (tanh, mixed combinations, 64-bit extents).
template <int MT, int NT>
__device__ __forceinline__ void gemm_epilogue_auto(const ctts_gemm_desc& d, floatx16 (&acc)[MT][NT], float* Cb, int z, int row0, int col0,
                                                   int wm0, int wn0, int l31, int h, int Mv, int Nv) {
  const bool flat = d.nb0 * d.nb1 == 1 && Mv == d.M && Nv == d.N;      // the gathered operands (R, Z, rowscale) are not batched
  const bool c_ok = gemm_fits32(Cb, Mv, d.ldc, Nv);
  if (d.E) {
    if (c_ok) return gemm_softmax_bwd_lean<MT, NT>(d, acc, Cb, z, row0, col0, wm0, wn0, l31, h, Mv, Nv);
  } else if (c_ok && (flat ? gemm_fits32(d.Z, d.M, d.ldz, d.N) && gemm_fits32(d.R, d.M, d.ldr, d.N) : !d.Z && !d.R && !d.rowscale)) {
#define CTTS_LEAN(ACT, DROP, BWD, AUX, RS) \
  return gemm_epilogue_lean<MT, NT, ACT, DROP, BWD, AUX, RS>(d, acc, Cb, z, row0, col0, wm0, wn0, l31, h, Mv, Nv)
    const bool drop = d.p_drop > 0.f, res = d.R != nullptr, rs = d.rowscale != nullptr;
    if (!d.epi_bwd) {
      if (d.act == 0 && !d.Z) {
        if (!res && !rs && !drop) CTTS_LEAN(0, false, false, false, false);      // -, b
        if (res && !rs) { if (drop) CTTS_LEAN(0, true, false, true, false); else CTTS_LEAN(0, false, false, true, false); }    // bdr, br
        if (res && rs) { if (drop) CTTS_LEAN(0, true, false, true, true); else CTTS_LEAN(0, false, false, true, true); }       // bdrs, brs
      } else if (!res && !rs && flat) {
        if (d.act == 1 && !drop) CTTS_LEAN(1, false, false, false, false);       // ba1z
        if (d.act == 2) { if (drop) CTTS_LEAN(2, true, false, false, false); else CTTS_LEAN(2, false, false, false, false); }  // ba2zd, ba2z
        if (d.act == 4) { if (drop) CTTS_LEAN(4, true, false, false, false); else CTTS_LEAN(4, false, false, false, false); }  // ba4zd, ba4z
      }
    } else if (d.Z && !rs && flat) {
      if (d.act == 1 && !drop) CTTS_LEAN(1, false, true, true, false);           // a1zB
      if (d.act == 2 && drop) CTTS_LEAN(2, true, true, true, false);             // a2zdB
      if (d.act == 4 && drop) CTTS_LEAN(4, true, true, true, false);             // a4zdB
    }
#undef CTTS_LEAN
  }
  gemm_epilogue<MT, NT>(d, acc, Cb, z, row0, col0, wm0, wn0, l31, h, Mv, Nv);
}

}  // namespace

// gemm_sk.hip: persistent stream-K kernel with direct-to-LDS operand loads.  Returns 1 when it took the launch, 0 when the descriptor is
// not eligible (the caller then uses the tile-per-workgroup kernels), < 0 on error.
int ctts_gemm_sk_try(const ctts_gemm_desc& d, hipStream_t st);

// gemm_ws.hip: weight-stationary kernel for K = 256 (the weight slice of a workgroup lives in registers, A tiles stream through LDS).
// Same return convention.
int ctts_gemm_ws_try(const ctts_gemm_desc& d, hipStream_t st);

// gemm_pl.hip: persistent stream-K kernel on pre-split bf16 planes (ctts_gemm_desc.A_planes / B_planes).  Same return convention.
int ctts_gemm_pl_try(const ctts_gemm_desc& d, hipStream_t st);

// gemm_plw.hip: the weight-gradient (TN) layout on the same plane sets.  _try: same return convention; _takes: 1 / 0 without launching.
int ctts_gemm_plw_try(const ctts_gemm_desc& d, hipStream_t st);
int ctts_gemm_plw_takes(const ctts_gemm_desc& d);
