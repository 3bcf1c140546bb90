// Fused multi-head attention for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): no [T,T] score / probability tensor in HBM
// in the forward pass, backward by recomputation.
//
//   fs2   (transformer_fs2.py:385-394, F.multi_head_attention_forward): softmax(q k^T / sqrt(d_h), keys >= len masked) v, 2 heads x 128
//   rel   (conformer.py:347-431, RelativeMultiHeadAttention): softmax(((q+u) k^T + shift((q+v) p^T)) / sqrt(d_model)) with NO mask,
//         dropout on the probabilities, 8 heads x 32; the position scores PS = (q+v) p^T come from ctts_gemm and are read through the
//         Transformer-XL shift as an index map (rel_index) - the shifted map, the probabilities and the dropped probabilities never exist
//
// Work decomposition: ONE WAVE PER WORKGROUP, 32 queries (forward) or 32 keys (backward) per wave, flash-style loop over the other
// axis in tiles of 32.  fp32 MFMA issues one 32x32x2 step per 64 cycles, so a wave needs only ~1 operand dword per 64 cycles: every
// operand is fetched straight from global / L2 in MFMA fragment order (A/B operand of lane (l31, h) = element [row l31][k-half h]),
// 64-byte runs per lane for the K-contiguous side and fully coalesced 128-byte rows for the transposed side.  No LDS tiles, no
// barriers (the only LDS use is a 32x33 transpose buffer for the position-score tile in the forward kernel).
//
// Forward computes S^T = K Q^T so that a lane owns one QUERY column: the online-softmax max / sum run over the lane's own 16
// accumulator registers (+ one exchange with lane^32), and P^T is already laid out as the B operand of O^T += V^T P^T.
// Backward (one wave = 32 keys, loop over query tiles) computes S = Q K^T and dP = dO V^T with the keys on the lanes, so that
// Pd and dS are the B operands of dV^T += dO^T Pd and dK^T += Q^T dS; dV / dK stay in registers for the whole loop (no atomics
// unless the query range is split, fs2 only), dS is written once (and, rel, a second time in the layout of the shift's adjoint)
// for the remaining gradients dQ = dS K, dQV = dPS pos, dpos = dPS^T QV, which are plain GEMMs.
#include "ctts_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr float LOG2E = 1.44269504088896340736f;

struct AttnArgs {
  const float *q, *k, *v;           // head 0 of utterance 0; element (b, t, head, c) = p[b*s + t*ld + head*dh + c]
  long ldq, ldk, ldv, sq, sk, sv;
  float* out; long ldo, so;         // [B,T,H*dh]
  float* lse;                       // [B,H,T]  log2-domain log-sum-exp of the scaled scores
  const float* bias;                // rel: PS [B,H,T,T] UNSHIFTED position scores, else NULL
  const int32_t* lens;              // fs2: valid length per utterance (keys >= len masked, query rows >= len are zero rows), else NULL
  int B, H, T;
  float scale, p_drop;
  const uint64_t* seed; uint32_t drop_offset;
  // backward
  const float* dout; long lddo, sdo;
  const float* D;                   // [B,H,T] rowsum(dO * O)
  float *dk, *dv; long lddk, lddv, sdk, sdv;
  long split_stride;                // q_split > 1: split s writes its partial dK / dV at dk/dv + s * split_stride (summed afterwards)
  float* dS;                        // [B,H,T,T]  d loss / d (q k^T + bias), scale included
  float* dPS;                       // rel: the same values scattered through the adjoint of the shift ([B,H,T,T]), or NULL
  int q_split;
};

__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // C/D row of accumulator register r

// conformer.py:423-431 `_relative_shift`: shifted[i,j] = padded.flat[i*T + j + T] with padded = [0 | PS] (rows of T+1).
// -> flat index into PS[T,T] of the element that lands at (i,j); -1 for the inserted zero (j == i+1).
// (32-bit: the C ABI checks T*T < 2^31)
__device__ __forceinline__ int rel_index(int i, int j, int T) {
  if (j == i + 1) return -1;
  return i * (T - 1) + (j <= i ? T - 1 : T - 2) + j;
}

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

// ---------------------------------------------------------------------------------------------------------------- forward
// occupancy targets (waves per SIMD): d_head 32 (conformer: thousands of waves) 3 forward / 2 backward (no spills), 64 -> 2, 128 -> 1
template <int DH, bool BIAS>
__global__ __launch_bounds__(64, (DH <= 32 ? (BIAS ? 2 : 3) : (DH <= 64 ? (BIAS ? 1 : 2) : 1))) void attn_fwd_kernel(const AttnArgs d) {
  constexpr int NC = DH / 32;
  __shared__ float sb[BIAS ? 32 * 33 : 1];
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
  const float* Kb = d.k + (long)b * d.sk + head * DH + 16 * h;
  const float* Vb = d.v + (long)b * d.sv + head * DH + l31;
  const float* PSz = BIAS ? d.bias + (long)z * T * T : nullptr;

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
  const uint32_t drow = ((uint32_t)z * (uint32_t)T + (uint32_t)i) * (uint32_t)T;     // dropout element index = ((z*T + i)*T + j)

  float kb[2][16], vb[2][16];          // ping-pong operand buffers (chunk parity)
  load16(Kb + (long)min(l31, T - 1) * d.ldk, kb[0]);
  // position-score tile (rows i0 .. i0+31, keys j0 .. j0+31) in load order: 16 x (2 queries x 32 consecutive keys) = 128-byte rows;
  // transposed through LDS later so that each lane gets the 16 values of ITS query column
  // The loads are UNCONDITIONAL (invalid elements read index 0 and are zeroed through `okm` when the tile is consumed): a load
  // under a per-element condition makes hipcc branch around it and wait vmcnt(0) at the join - 16 serialised memory round trips per tile.
  unsigned okm = 0;
  auto load_ps_tile = [&](int jt, float (&dst)[16]) {
    okm = 0;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int ii = i0 + 2 * it + h, jj = jt + l31;
      const bool ok = ii < T && jj < T && jj != ii + 1;
      const int idx = ii * (T - 1) + (jj <= ii ? T - 1 : T - 2) + jj;
      dst[it] = PSz[ok ? idx : 0];
      okm |= (ok ? 1u : 0u) << it;
    }
  };
  float psn[16];
  if (BIAS) load_ps_tile(0, psn);
  for (int j0 = 0; j0 < L; j0 += 32) {
    const float* Kr = Kb + (long)min(j0 + l31, T - 1) * d.ldk;
    // position scores: the tile needed NOW was fetched one tile ago (HBM latency under load is several microseconds - far more
    // than the 1,024 MFMA cycles of one score block)
    // (the registers `psn` hold this tile's scores; the NEXT tile's fetch is issued as soon as they have been handed to LDS below)
    // ---- S^T[j][i] = sum_k K[j][k] Q[i][k]
    floatx16 St;
#pragma unroll
    for (int r = 0; r < 16; ++r) St[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
        load16(Kr + 32 * (c + 1), kb[(c + 1) & 1]);
      } else {                          // last score chunk: fetch the first V chunk of this tile underneath it
#pragma unroll
        for (int st = 0; st < 16; ++st) vb[0][st] = Vb[(long)min(j0 + rowmap(st, h), T - 1) * d.ldv];
      }
      CTTS_SCHED_FENCE();
#pragma unroll
      for (int q = 0; q < 16; ++q) St = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[c & 1][q], Qf[16 * c + q], St, 0, 0, 0);
      CTTS_SCHED_FENCE();
    }
    float s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = St[r];
    if (BIAS) {
#pragma unroll
      for (int it = 0; it < 16; ++it) sb[(2 * it + h) * 33 + l31] = ((okm >> it) & 1u) ? psn[it] : 0.f;
      load_ps_tile(j0 + 32, psn);                        // next tile (unconditional: past the end every element is masked): a softmax,
                                                         // a PV block and a score block of cover
      // The workgroup is ONE wave and the LDS executes a wave's instructions in order, so the transposed read below sees the writes
      // above without a barrier.  __syncthreads() here would cost far more than its s_barrier: its fence waits vmcnt(0), i.e. it
      // drains every prefetch in flight (K, V and position-score tiles) twice per tile.  wave_barrier only pins the program order.
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] += sb[l31 * 33 + rowmap(r, h)];
      __builtin_amdgcn_wave_barrier();
    }
    // ---- online softmax over the keys (log2 domain); a lane owns query column i, its partner lane^32 the other 16 key rows
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + rowmap(r, h);
      const float v = j < L ? s[r] * sl2 : -INFINITY;
      s[r] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m, tmax);               // finite: the tile holds at least one key < L
    const float alpha = fast_exp2(m - mnew);
    float psum = 0.f;
    floatx16 Pt;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = fast_exp2(s[r] - mnew);
      psum += p;
      if (do_drop) p *= ctts_drop_scale(dkey, drow + (uint32_t)(j0 + rowmap(r, h)), d.p_drop, inv_keep);
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
        for (int st = 0; st < 16; ++st) vb[(c + 1) & 1][st] = Vb[(long)min(j0 + rowmap(st, h), T - 1) * d.ldv + 32 * (c + 1)];
      } else {                          // last PV chunk: fetch the next tile's first K chunk underneath it (unconditional, row clamped:
        load16(Kb + (long)min(j0 + 32 + l31, T - 1) * d.ldk, kb[0]);      // a load under a runtime condition costs a vmcnt(0) at the join)
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
#ifndef CTTS_ATTN_BWD32_WAVES
#define CTTS_ATTN_BWD32_WAVES 2
#endif
template <int DH, bool BIAS>
__global__ __launch_bounds__(64, (DH <= 32 ? CTTS_ATTN_BWD32_WAVES : 1)) void attn_bwd_kernel(const AttnArgs d) {
  constexpr int NC = DH / 32;
  constexpr bool RES = DH <= 64;        // K / V fragments of the wave's 32 keys stay in registers
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
  const float* PSz = BIAS ? d.bias + (long)z * T * T : nullptr;
  const float* lse = d.lse + (long)z * T;
  const float* Dz = d.D + (long)z * T;
  float* dSz = d.dS + (long)z * T * T;
  float* dPSz = (BIAS && d.dPS) ? d.dPS + (long)z * T * T : nullptr;

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
  const float sl2 = d.scale * LOG2E;
  const bool do_drop = d.p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(d.seed, d.drop_offset); inv_keep = 1.f / (1.f - d.p_drop); }

  const int nq = (L + 31) / 32;
  const int per = (nq + d.q_split - 1) / max(d.q_split, 1);
  const int q_lo = split * per, q_hi = min(nq, q_lo + per);
  // ping-pong operand buffers (chunk parity): Q / dO rows as A operands of S and dP, K / V chunks as their B operands when they
  // are streamed, dO^T / Q^T rows as A operands of dV^T and dK^T
  float qa[2][16], da[2][16], kb[2][16], vb[2][16], dot[2][16], qtt[2][16];
  if (q_lo < q_hi) {
    load16(Qb + (long)min(q_lo * 32 + l31, T - 1) * d.ldq + 16 * h, qa[0]);
    if (!RES) load16(Kr, kb[0]);
  }
  // position scores of a (query tile, this wave's keys) block: register r <-> query row it0 + rowmap(r, h), lane <-> key (coalesced rows)
  // (unconditional loads + validity mask, see the forward kernel)
  unsigned okb = 0;
  auto load_bias_tile = [&](int it0, float (&dst)[16]) {
    okb = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ir = it0 + rowmap(r, h);
      const bool ok = ir < T && j < T && j != ir + 1;
      const int bi = ir * (T - 1) + (j <= ir ? T - 1 : T - 2) + j;
      dst[r] = PSz[ok ? bi : 0];
      okb |= (ok ? 1u : 0u) << r;
    }
  };
  float bias_n[16];
  if (BIAS && q_lo < q_hi) load_bias_tile(q_lo * 32, bias_n);
  for (int qt = q_lo; qt < q_hi; ++qt) {
    const int i0 = qt * 32;
    const int ia = min(i0 + l31, T - 1);
    const float* Qr = Qb + (long)ia * d.ldq + 16 * h;
    const float* dOr = dOb + (long)ia * d.lddo + 16 * h;
    // ---- S[i][j] = sum_k Q[i][k] K[j][k]
    floatx16 S, dP;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
        load16(Qr + 32 * (c + 1), qa[(c + 1) & 1]);
        if (!RES) load16(Kr + 32 * (c + 1), kb[(c + 1) & 1]);
      } else {
        load16(dOr, da[0]);
        if (!RES) load16(Vr, vb[0]);
      }
      CTTS_SCHED_FENCE();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float bq;
        if constexpr (RES) bq = Kf[16 * c + q]; else bq = kb[c & 1][q];
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[c & 1][q], bq, S, 0, 0, 0);
      }
      CTTS_SCHED_FENCE();
    }
    // row statistics and position scores of this tile: issued before the dP MFMAs, consumed after them
    const float lse_l = lse[ia], D_l = Dz[ia];          // lanes 0..31 (and 32..63 again) hold rows i0 .. i0+31
    // (`bias_n` holds this tile's position scores, fetched one query tile ago)
    // ---- dPd[i][j] = sum_k dO[i][k] V[j][k]
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
        load16(dOr + 32 * (c + 1), da[(c + 1) & 1]);
        if (!RES) load16(Vr + 32 * (c + 1), vb[(c + 1) & 1]);
      } else {                          // operands of the first dV^T / dK^T chunk
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          const long ir = min(i0 + rowmap(st, h), T - 1);
          dot[0][st] = dOb[ir * d.lddo + l31];
          qtt[0][st] = Qb[ir * d.ldq + l31];
        }
      }
      CTTS_SCHED_FENCE();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float bq;
        if constexpr (RES) bq = Vf[16 * c + q]; else bq = vb[c & 1][q];
        dP = __builtin_amdgcn_mfma_f32_32x32x2f32(da[c & 1][q], bq, dP, 0, 0, 0);
      }
      CTTS_SCHED_FENCE();
    }
    // ---- P = exp2(scale*log2e*(S + bias) - lse),  Pd = P*keep/(1-p),  dS = scale * P * (dPd*keep/(1-p) - D)
    floatx16 Pd, dSv;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rm = rowmap(r, h), ir = i0 + rm;
      const float lse_i = __shfl(lse_l, rm, 64), Di = __shfl(D_l, rm, 64);
      float s = S[r];
      if (BIAS) s += ((okb >> r) & 1u) ? bias_n[r] : 0.f;
      const bool valid = ir < L && j < L;
      const float p = valid ? fast_exp2(s * sl2 - lse_i) : 0.f;
      float ks = 1.f;
      if (do_drop) ks = ctts_drop_scale(dkey, ((uint32_t)z * (uint32_t)T + (uint32_t)ir) * (uint32_t)T + (uint32_t)j, d.p_drop, inv_keep);
      const float ds = p * (dP[r] * ks - Di) * d.scale;
      Pd[r] = p * ks;
      dSv[r] = ds;
      if (valid) {
        dSz[(long)ir * T + j] = ds;
        if (BIAS) {                      // the same element through the adjoint of the shift (index recomputed: cheaper than 16 live registers)
          const int bi = rel_index(ir, j, T);
          if (dPSz && bi >= 0) dPSz[bi] = ds;
        }
      }
    }
    if (BIAS) load_bias_tile(i0 + 32, bias_n);          // next tile's scores (unconditional, masked past the end): three MFMA blocks of cover
    // ---- dV^T[d][j] += sum_i dO[i][d] Pd[i][j],  dK^T[d][j] += sum_i Q[i][d] dS[i][j]
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          const long ir = min(i0 + rowmap(st, h), T - 1);
          dot[(c + 1) & 1][st] = dOb[ir * d.lddo + 32 * (c + 1) + l31];
          qtt[(c + 1) & 1][st] = Qb[ir * d.ldq + 32 * (c + 1) + l31];
        }
      } else {                          // next query tile's first S operands (unconditional, row clamped)
        load16(Qb + (long)min(i0 + 32 + l31, T - 1) * d.ldq + 16 * h, qa[0]);
        if (!RES) load16(Kr, kb[0]);
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

// row 0 of the shift's adjoint layout is only partly covered by the scatter (columns 0..T-2 of row 0 are never read by the shift)
__global__ void dps_row0_zero_kernel(float* dPS, int nz, int T) {
  const long n = (long)nz * (T - 1);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long zz = e / (T - 1);
    dPS[zz * T * T + (e - zz * (T - 1))] = 0.f;
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
                            float* dS, float* kv_part, float* dqkv, int B, int T, int H, int C, float scale, int q_split, void* stream) {
  CTTS_REQUIRE(qkv && out && dout && lse && Dws && dS && dqkv && B > 0 && T > 0, "ctts_mha_bwd: bad arguments");
  CTTS_REQUIRE(q_split <= 1 || kv_part, "ctts_mha_bwd: q_split > 1 needs the kv_part scratch [q_split, B, T, 2C]");
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
  g.split_k = (tiles < 1536 && T >= 512) ? 2 : 1;
  return ctts_gemm(&g, stream);
}

extern "C" size_t ctts_relmha_workspace_floats(int B, int T, int H) { return (size_t)B * H * T * T; }

extern "C" int ctts_relmha_fwd(const float* qu, const float* qv, const float* kv, const float* pos, float* ps, float* out, float* lse,
                               int B, int T, int H, int C, float scale, float p_drop, const uint64_t* seed, uint32_t drop_offset,
                               void* stream) {
  CTTS_REQUIRE(qu && qv && kv && pos && ps && out && lse && B > 0 && T > 0, "ctts_relmha_fwd: bad arguments");
  CTTS_REQUIRE(ctts_mha_supported(C, H) && al16(qu) && al16(kv) && al16(out), "ctts_relmha_fwd: needs d_head in {32,64,128} and 16-byte aligned tensors");
  CTTS_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "ctts_relmha_fwd: p_drop out of range");
  CTTS_REQUIRE((long)B * H * T * T < 0xFFFFFFFFL && (long)T * T < 0x7FFFFFFFL, "ctts_relmha_fwd: [B,H,T,T] exceeds the 32-bit element index");
  const int dh = C / H;
  // PS[b,h] = (q + v_bias)[b,:,h] pos[:,h]^T   (conformer.py:405-407; unshifted, the shift is an index map inside the fused kernel)
  ctts_gemm_desc g = {};
  g.A = qv; g.B = pos; g.C = ps;
  g.M = T; g.N = T; g.K = dh; g.lda = C; g.ldb = C; g.ldc = T; g.a_kc = 1; g.b_kc = 1;
  g.nb0 = B; g.nb1 = H; g.sA0 = (long)T * C; g.sA1 = dh; g.sB0 = 0; g.sB1 = dh; g.sC0 = (long)H * T * T; g.sC1 = (long)T * T;
  g.alpha = 1.f; g.split_k = 1;
  int rc = ctts_gemm(&g, stream);
  if (rc) return rc;
  AttnArgs a = {};
  a.q = qu; a.ldq = C; a.sq = (long)T * C;
  a.k = kv; a.v = kv + C; a.ldk = a.ldv = 2L * C; a.sk = a.sv = (long)T * 2 * C;
  a.out = out; a.ldo = C; a.so = (long)T * C; a.lse = lse; a.bias = ps;
  a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.drop_offset = drop_offset;
  return launch_fwd<true>(a, dh, (hipStream_t)stream);
}

extern "C" int ctts_relmha_bwd(const float* qu, const float* qv, const float* kv, const float* pos, const float* ps, const float* out,
                               const float* dout, const float* lse, float* Dws, float* dS, float* dPS, float* dqu, float* dqv, float* dkv,
                               float* dpos_b, int B, int T, int H, int C, float scale, float p_drop, const uint64_t* seed,
                               uint32_t drop_offset, void* stream) {
  CTTS_REQUIRE(qu && qv && kv && pos && ps && out && dout && lse && Dws && dS && dPS && dqu && dqv && dkv && dpos_b && B > 0 && T > 0,
               "ctts_relmha_bwd: bad arguments");
  CTTS_REQUIRE(ctts_mha_supported(C, H) && al16(qu) && al16(kv) && al16(dout) && al16(dkv), "ctts_relmha_bwd: needs d_head in {32,64,128} and 16-byte aligned tensors");
  hipStream_t st = (hipStream_t)stream;
  const int dh = C / H;
  int rc = ctts_rowdot_heads(dout, out, Dws, B, T, H, dh, stream);
  if (rc) return rc;
  if (T > 1) {
    const long n = (long)B * H * (T - 1);
    hipLaunchKernelGGL(dps_row0_zero_kernel, dim3((unsigned)min((n + 255) / 256, 4096L)), dim3(256), 0, st, dPS, B * H, T);
    CTTS_CHECK_LAUNCH("ctts_relmha_bwd(dps row 0)");
  }
  AttnArgs a = {};
  a.q = qu; a.ldq = C; a.sq = (long)T * C;
  a.k = kv; a.v = kv + C; a.ldk = a.ldv = 2L * C; a.sk = a.sv = (long)T * 2 * C;
  a.lse = const_cast<float*>(lse); a.bias = ps;
  a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.drop_offset = drop_offset;
  a.dout = dout; a.lddo = C; a.sdo = (long)T * C; a.D = Dws;
  a.dk = dkv; a.dv = dkv + C; a.lddk = a.lddv = 2L * C; a.sdk = a.sdv = (long)T * 2 * C;
  a.dS = dS; a.dPS = dPS; a.q_split = 1;
  rc = launch_bwd<true>(a, dh, st);
  if (rc) return rc;
  const long sS0 = (long)H * T * T, sS1 = (long)T * T;
  ctts_gemm_desc g = {};
  // dQU[i,d] = sum_j dS[i,j] K[j,d]
  g.A = dS; g.B = kv; g.C = dqu; g.M = T; g.N = dh; g.K = T; g.lda = T; g.ldb = 2L * C; g.ldc = C; g.a_kc = 1; g.b_kc = 0;
  g.nb0 = B; g.nb1 = H; g.sA0 = sS0; g.sA1 = sS1; g.sB0 = (long)T * 2 * C; g.sB1 = dh; g.sC0 = (long)T * C; g.sC1 = dh;
  g.alpha = 1.f; g.split_k = 1;
  rc = ctts_gemm(&g, stream);
  if (rc) return rc;
  // dQV[i,d] = sum_m dPS[i,m] pos[m,d]
  g.A = dPS; g.B = pos; g.C = dqv; g.ldb = C; g.sB0 = 0; g.sB1 = dh;
  rc = ctts_gemm(&g, stream);
  if (rc) return rc;
  // dpos_b[b][m,d] = sum_i dPS[b][i,m] QV[b][i,d]   (caller sums over b: pos is shared by the batch)
  g.A = dPS; g.B = qv; g.C = dpos_b; g.a_kc = 0; g.b_kc = 0; g.lda = T; g.ldb = C; g.ldc = C;
  g.sB0 = (long)T * C; g.sB1 = dh; g.sC0 = (long)T * C; g.sC1 = dh;
  return ctts_gemm(&g, stream);
}
