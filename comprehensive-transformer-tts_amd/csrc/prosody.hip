// liu2021 implicit prosody modelling kernels (reference model/modules.py:332-648, model/coordconv.py).
//
//  * ctts_im2col_3x3s2 / ctts_col2im_3x3s2 : Conv2d 3x3, stride (1,2), padding (1,1) of ReferenceEncoder (modules.py:351-361) on
//      channel-last activations x[B,T,W,C]: the patch matrix col[(b,t,wo)][(kh,kw,c)] feeds ctts_gemm (forward, dgrad and wgrad
//      all on MFMA); col2im is the adjoint written as a GATHER (each input element sums the <= 6 patch entries that read
//      it) - deterministic, no atomics.  Both are pure streaming kernels (HBM bound).
//  * ctts_gru_fwd / ctts_gru_bwd : the sequential part of nn.GRU (modules.py:359-361,391 and :618-621,637).  The input projection
//      W_ih x + b_ih of ALL time steps is one GEMM done by the caller; here one workgroup per (utterance, direction) walks the T steps
//      with its W_hh slice held in REGISTERS (thread g owns row g: H floats), h_{t-1} broadcast from LDS, two barriers per step; the
//      per-step operands are prefetched in chunks of 8 steps (registers -> double-buffered LDS) so that no step waits on HBM latency.  Backward is BPTT with the same structure (thread (p,j) owns W_hh[pH..pH+H-1][j]); it emits
//      dgi (-> W_ih / input gradients by GEMM), dgh and h_{t-1} (-> dW_hh = dgh^T h_prev by one split-K GEMM, db_hh by a column sum).
//  * ctts_softmax_rect_fwd / _bwd : masked row softmax of a rectangular score matrix [nb,Tq,Tk] (PhonemeLevelProsodyEncoder
//      attention, modules.py:443-446: keys >= klens masked to -inf, query rows >= qlens zeroed; also the 32-token STL softmax).
#include "ctts_common.h"

namespace {

inline int grid_for(long n, int block = 256) { return (int)((n + block - 1) / block > 65535 * 16 ? 65535 * 16 : (n + block - 1) / block); }

__global__ void im2col_3x3s2_kernel(const float4* __restrict__ x, float4* __restrict__ col, int T, int W, int Wo, int C4, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % C4);
    long r = e / C4;
    const int tap = (int)(r % 9);
    r /= 9;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int t = (int)(r % T);
    const long b = r / T;
    const int ti = t + tap / 3 - 1, wi = 2 * wo - 1 + tap % 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ti >= 0 && ti < T && wi >= 0 && wi < W) v = x[((b * T + ti) * W + wi) * C4 + c4];
    col[e] = v;
  }
}

__global__ void col2im_3x3s2_kernel(const float4* __restrict__ dcol, float4* __restrict__ dx, int T, int W, int Wo, int C4, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % C4);
    long r = e / C4;
    const int w = (int)(r % W);
    r /= W;
    const int t = (int)(r % T);
    const long b = r / T;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int to = t - kh + 1;
      if (to < 0 || to >= T) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int num = w + 1 - kw;
        if (num < 0 || (num & 1)) continue;
        const int wo = num >> 1;
        if (wo >= Wo) continue;
        const float4 v = dcol[(((b * T + to) * Wo + wo) * 9 + kh * 3 + kw) * C4 + c4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    dx[e] = acc;
  }
}

// The gate non-linearities sit on the sequential critical path (one wave, 4 cycles per VALU instruction): hardware exp + fast
// reciprocal (~2 ulp) instead of the ~40-instruction libm expansions.  tanh(v) = 1 - 2 / (1 + e^{2v}) saturates correctly at +-inf.
__device__ __forceinline__ float sigmoidf_(float v) { return __frcp_rn(1.0f + __expf(-v)); }
__device__ __forceinline__ float tanhf_(float v) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * v)); }

constexpr int GRU_CH = 8;   // time steps per prefetch chunk: global loads are issued >= 8 steps (~2 us) before their first use

// gi [B,T,ndir,3H], whh [ndir,3H,H], bhh [ndir,3H], out [B,T,ndir,H], gates [B,T,ndir,4H] (r|z|n|W_hn h + b_hn) or NULL
template <int H>
__global__ __launch_bounds__((3 * H + 63) / 64 * 64) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ whh,
                                                                         const float* __restrict__ bhh, float* __restrict__ out,
                                                                         float* __restrict__ gates, int T, int ndir, int rev_mask) {
  __shared__ float s_h[H];
  __shared__ float s_gh[3 * H];
  __shared__ float s_gi[2][GRU_CH][3 * H];       // double-buffered chunk of input projections
  const int b = blockIdx.x, dir = blockIdx.y, g = threadIdx.x;
  const bool rev = (rev_mask >> dir) & 1;          // this sequence runs t = T-1 .. 0
  const bool act = g < 3 * H;
  float w[H];
  float bias = 0.f;
  if (act) {
    const float* wr = whh + ((long)dir * 3 * H + g) * H;
#pragma unroll
    for (int k = 0; k < H; ++k) w[k] = wr[k];
    bias = bhh[dir * 3 * H + g];
  }
  if (g < H) s_h[g] = 0.f;
  const long gs = (long)ndir * 3 * H;
  const float* gib = gi + (long)b * T * gs + (long)dir * 3 * H;
  float nx[GRU_CH];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) {
      const int s = c0 + i;
      nx[i] = (act && s < T) ? gib[(long)(rev ? T - 1 - s : s) * gs + g] : 0.f;
    }
  };
  load_chunk(0);
  if (act) {
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) s_gi[0][i][g] = nx[i];
  }
  __syncthreads();
  for (int c0 = 0; c0 < T; c0 += GRU_CH) {
    const int buf = (c0 / GRU_CH) & 1;
    if (c0 + GRU_CH < T) load_chunk(c0 + GRU_CH);
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) {
      const int s = c0 + i;
      if (s >= T) break;
      const int t = rev ? T - 1 - s : s;
      if (act) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < H; ++k) acc = fmaf(w[k], s_h[k], acc);
        s_gh[g] = acc;
      }
      __syncthreads();
      if (g < H) {
        const float r = sigmoidf_(s_gi[buf][i][g] + s_gh[g]);
        const float z = sigmoidf_(s_gi[buf][i][H + g] + s_gh[H + g]);
        const float ghn = s_gh[2 * H + g];
        const float n = tanhf_(s_gi[buf][i][2 * H + g] + r * ghn);
        const float h = (1.0f - z) * n + z * s_h[g];
        const long o = ((long)b * T + t) * ndir + dir;
        out[o * H + g] = h;
        if (gates) {
          float* gp = gates + o * 4 * H;
          gp[g] = r; gp[H + g] = z; gp[2 * H + g] = n; gp[3 * H + g] = ghn;
        }
        s_h[g] = h;
      }
      __syncthreads();
    }
    if (act && c0 + GRU_CH < T) {                 // the other buffer was last read one chunk ago: every thread has passed barriers since
#pragma unroll
      for (int i = 0; i < GRU_CH; ++i) s_gi[buf ^ 1][i][g] = nx[i];
    }
  }
}

// dout [B,T,ndir,H]; dgi, dgh [B,T,ndir,3H]; hprev [B,T,ndir,H]
template <int H>
__global__ __launch_bounds__((3 * H + 63) / 64 * 64) void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                                         const float* __restrict__ gates, const float* __restrict__ whh,
                                                                         float* __restrict__ dgi, float* __restrict__ dgh,
                                                                         float* __restrict__ hprev, int T, int ndir, int rev_mask) {
  __shared__ float s_dgh[3 * H];
  __shared__ float s_part[3 * H];
  __shared__ float s_in[2][GRU_CH][6 * H];       // per step: r|z|n|ghn (4H) | dout (H) | h_{prev} (H)
  const int b = blockIdx.x, dir = blockIdx.y, id = threadIdx.x;
  const bool rev = (rev_mask >> dir) & 1;
  const bool act = id < 3 * H;
  const int p = id / H, j = id - p * H;
  float wc[H];
  if (act) {
#pragma unroll
    for (int k = 0; k < H; ++k) wc[k] = whh[((long)dir * 3 * H + p * H + k) * H + j];
  }
  float nx[GRU_CH][2];
  auto load_chunk = [&](int c0) {                // chunk element i is processing step s = T-1-(c0+i)
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) {
      const int s = T - 1 - (c0 + i);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int v = id + q * 3 * H;
        float val = 0.f;
        if (act && s >= 0) {
          const int t = rev ? T - 1 - s : s;
          const long o = ((long)b * T + t) * ndir + dir;
          if (v < 4 * H) val = gates[o * 4 * H + v];
          else if (v < 5 * H) val = dout[o * H + (v - 4 * H)];
          else if (s > 0) val = out[(((long)b * T + (rev ? t + 1 : t - 1)) * ndir + dir) * H + (v - 5 * H)];
        }
        nx[i][q] = val;
      }
    }
  };
  load_chunk(0);
  if (act) {
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) { s_in[0][i][id] = nx[i][0]; s_in[0][i][id + 3 * H] = nx[i][1]; }
  }
  __syncthreads();
  float dh = 0.f;
  for (int c0 = 0; c0 < T; c0 += GRU_CH) {
    const int buf = (c0 / GRU_CH) & 1;
    if (c0 + GRU_CH < T) load_chunk(c0 + GRU_CH);
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) {
      const int s = T - 1 - (c0 + i);
      if (s < 0) break;
      const int t = rev ? T - 1 - s : s;
      float dcarry = 0.f;
      if (id < H) {
        const float* in = s_in[buf][i];
        const float r = in[id], z = in[H + id], n = in[2 * H + id], ghn = in[3 * H + id], hp = in[5 * H + id];
        const float d = dh + in[4 * H + id];
        const float dn = d * (1.0f - z);
        const float dz = d * (hp - n);
        dcarry = d * z;
        const float dnp = dn * (1.0f - n * n);
        const float dzp = dz * z * (1.0f - z);
        const float drp = dnp * ghn * r * (1.0f - r);
        const long o = ((long)b * T + t) * ndir + dir;
        float* a = dgi + o * 3 * H;
        a[id] = drp; a[H + id] = dzp; a[2 * H + id] = dnp;
        float* c = dgh + o * 3 * H;
        c[id] = drp; c[H + id] = dzp; c[2 * H + id] = dnp * r;
        s_dgh[id] = drp; s_dgh[H + id] = dzp; s_dgh[2 * H + id] = dnp * r;
        hprev[o * H + id] = hp;
      }
      __syncthreads();
      if (act) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) acc = fmaf(wc[k], s_dgh[p * H + k], acc);
        s_part[id] = acc;
      }
      __syncthreads();
      if (id < H) dh = dcarry + s_part[id] + s_part[H + id] + s_part[2 * H + id];
    }
    if (act && c0 + GRU_CH < T) {
#pragma unroll
      for (int i = 0; i < GRU_CH; ++i) { s_in[buf ^ 1][i][id] = nx[i][0]; s_in[buf ^ 1][i][id + 3 * H] = nx[i][1]; }
    }
    __syncthreads();                             // the next chunk's first phase reads s_in right away
  }
}

// ---- H = 32 in ONE wave (the Tm-step reference-encoder GRUs): no workgroup barriers at all.  Lane l < 32 owns unit j = l: gate rows
// r_j and n_j; lane 32 + j owns row z_j.  h is broadcast through LDS (in-order within a wave), z crosses the halves by one shuffle, each
// lane prefetches its own gi rows 8 steps ahead in registers.
__global__ __launch_bounds__(64) void gru_fwd_wave32_kernel(const float* __restrict__ gi, const float* __restrict__ whh,
                                                             const float* __restrict__ bhh, float* __restrict__ out,
                                                             float* __restrict__ gates, int T, int ndir, int rev_mask) {
  constexpr int H = 32;
  __shared__ __attribute__((aligned(16))) float s_h[H];
  const int b = blockIdx.x, dir = blockIdx.y, l = threadIdx.x, half = l >> 5, j = l & 31;
  const bool rev = (rev_mask >> dir) & 1;
  const int row0 = half ? H + j : j, row1 = 2 * H + j;          // lanes >= 32 compute row1 redundantly (never used)
  float w0[H], w1[H];
  {
    const float* a = whh + ((long)dir * 3 * H + row0) * H;
    const float* c = whh + ((long)dir * 3 * H + row1) * H;
#pragma unroll
    for (int k = 0; k < H; ++k) { w0[k] = a[k]; w1[k] = c[k]; }
  }
  const float b0 = bhh[dir * 3 * H + row0], b1 = bhh[dir * 3 * H + row1];
  const long gs = (long)ndir * 3 * H;
  const float* gib = gi + (long)b * T * gs + (long)dir * 3 * H;
  float n0[GRU_CH], n1[GRU_CH];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) {
      const int s = c0 + i;
      const float* p = gib + (long)(rev ? T - 1 - s : s) * gs;
      n0[i] = s < T ? p[row0] : 0.f;
      n1[i] = s < T ? p[row1] : 0.f;
    }
  };
  load_chunk(0);
  float h = 0.f;
  for (int c0 = 0; c0 < T; c0 += GRU_CH) {
    float g0[GRU_CH], g1[GRU_CH];
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) { g0[i] = n0[i]; g1[i] = n1[i]; }
    if (c0 + GRU_CH < T) load_chunk(c0 + GRU_CH);
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) {
      const int s = c0 + i;
      if (s >= T) break;
      const int t = rev ? T - 1 - s : s;
      if (half == 0) s_h[j] = h;
      __builtin_amdgcn_wave_barrier();                       // one wave: LDS executes in order, this only pins the compiler's order
      float a0 = b0, a1 = b1, c0_ = 0.f, c1_ = 0.f, d0_ = 0.f, d1_ = 0.f, e0_ = 0.f, e1_ = 0.f;   // 8 independent FMA chains of 8
#pragma unroll
      for (int k = 0; k < H; k += 4) {
        const float4 hv = *reinterpret_cast<const float4*>(s_h + k);          // same address in all lanes: LDS broadcast
        a0 = fmaf(w0[k], hv.x, a0); c0_ = fmaf(w0[k + 1], hv.y, c0_); d0_ = fmaf(w0[k + 2], hv.z, d0_); e0_ = fmaf(w0[k + 3], hv.w, e0_);
        a1 = fmaf(w1[k], hv.x, a1); c1_ = fmaf(w1[k + 1], hv.y, c1_); d1_ = fmaf(w1[k + 2], hv.z, d1_); e1_ = fmaf(w1[k + 3], hv.w, e1_);
      }
      a0 = (a0 + c0_) + (d0_ + e0_);
      a1 = (a1 + c1_) + (d1_ + e1_);
      const float sg = sigmoidf_(g0[i] + a0);            // r_j in lane j, z_j in lane 32 + j
      const float z = __shfl(sg, j + 32, 64);
      if (half == 0) {
        const float n = tanhf_(g1[i] + sg * a1);
        h = (1.0f - z) * n + z * h;
        const long o = ((long)b * T + t) * ndir + dir;
        out[o * H + j] = h;
        if (gates) {
          float* gp = gates + o * 4 * H;
          gp[j] = sg; gp[H + j] = z; gp[2 * H + j] = n; gp[3 * H + j] = a1;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// BPTT, one wave: lane j < 32 owns unit j (elementwise part) and the W_hh column-j segments of the r and n rows; lane 32 + j the z rows.
__global__ __launch_bounds__(64) void gru_bwd_wave32_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                             const float* __restrict__ gates, const float* __restrict__ whh,
                                                             float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ hprev,
                                                             int T, int ndir, int rev_mask) {
  constexpr int H = 32;
  __shared__ __attribute__((aligned(16))) float s_dgh[3 * H];
  const int b = blockIdx.x, dir = blockIdx.y, l = threadIdx.x, half = l >> 5, j = l & 31;
  const bool rev = (rev_mask >> dir) & 1;
  const int p0 = half ? 1 : 0;                               // gate block whose rows this lane contracts first (r | z), then n (lanes < 32)
  float wa[H], wn[H];
#pragma unroll
  for (int k = 0; k < H; ++k) {
    wa[k] = whh[((long)dir * 3 * H + p0 * H + k) * H + j];
    wn[k] = whh[((long)dir * 3 * H + 2 * H + k) * H + j];
  }
  float nx[GRU_CH][6];
  auto load_chunk = [&](int c0) {                            // chunk element i is processing step s = T-1-(c0+i); lanes < 32 only
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) {
      const int s = T - 1 - (c0 + i);
      const bool ok = half == 0 && s >= 0;
      const int t = rev ? T - 1 - s : s;
      const long o = ((long)b * T + (ok ? t : 0)) * ndir + dir;
      const float* gp = gates + o * 4 * H;
      nx[i][0] = ok ? gp[j] : 0.f; nx[i][1] = ok ? gp[H + j] : 0.f; nx[i][2] = ok ? gp[2 * H + j] : 0.f; nx[i][3] = ok ? gp[3 * H + j] : 0.f;
      nx[i][4] = ok ? dout[o * H + j] : 0.f;
      nx[i][5] = (ok && s > 0) ? out[(((long)b * T + (rev ? t + 1 : t - 1)) * ndir + dir) * H + j] : 0.f;
    }
  };
  load_chunk(0);
  float dh = 0.f;
  for (int c0 = 0; c0 < T; c0 += GRU_CH) {
    float cur[GRU_CH][6];
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i)
#pragma unroll
      for (int q = 0; q < 6; ++q) cur[i][q] = nx[i][q];
    if (c0 + GRU_CH < T) load_chunk(c0 + GRU_CH);
#pragma unroll
    for (int i = 0; i < GRU_CH; ++i) {
      const int s = T - 1 - (c0 + i);
      if (s < 0) break;
      const int t = rev ? T - 1 - s : s;
      float dcarry = 0.f;
      if (half == 0) {
        const float r = cur[i][0], z = cur[i][1], n = cur[i][2], ghn = cur[i][3], hp = cur[i][5];
        const float d = dh + cur[i][4];
        const float dn = d * (1.0f - z);
        const float dz = d * (hp - n);
        dcarry = d * z;
        const float dnp = dn * (1.0f - n * n);
        const float dzp = dz * z * (1.0f - z);
        const float drp = dnp * ghn * r * (1.0f - r);
        const long o = ((long)b * T + t) * ndir + dir;
        float* a = dgi + o * 3 * H;
        a[j] = drp; a[H + j] = dzp; a[2 * H + j] = dnp;
        float* c = dgh + o * 3 * H;
        c[j] = drp; c[H + j] = dzp; c[2 * H + j] = dnp * r;
        s_dgh[j] = drp; s_dgh[H + j] = dzp; s_dgh[2 * H + j] = dnp * r;
        hprev[o * H + j] = hp;
      }
      __builtin_amdgcn_wave_barrier();
      float pa = 0.f, pn = 0.f;
#pragma unroll
      for (int k = 0; k < H; k += 4) {
        const float4 va = *reinterpret_cast<const float4*>(s_dgh + p0 * H + k);       // r block (lanes < 32) / z block (lanes >= 32)
        const float4 vn = *reinterpret_cast<const float4*>(s_dgh + 2 * H + k);
        pa = fmaf(wa[k], va.x, pa); pa = fmaf(wa[k + 1], va.y, pa); pa = fmaf(wa[k + 2], va.z, pa); pa = fmaf(wa[k + 3], va.w, pa);
        pn = fmaf(wn[k], vn.x, pn); pn = fmaf(wn[k + 1], vn.y, pn); pn = fmaf(wn[k + 2], vn.z, pn); pn = fmaf(wn[k + 3], vn.w, pn);
      }
      const float pz = __shfl(pa, j + 32, 64);
      dh = dcarry + pa + pn + pz;                            // meaningful in lanes < 32
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void softmax_rect_kernel(float* __restrict__ S, const float* __restrict__ P,
                                                            const int32_t* __restrict__ klens, const int32_t* __restrict__ qlens,
                                                            int Tq, int Tk, long nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < nrows; row += (long)gridDim.x * 4) {
    const long zb = row / Tq;
    const int q = (int)(row - zb * Tq);
    const int L = klens ? min(klens[zb], Tk) : Tk;
    const int Lq = qlens ? min(qlens[zb], Tq) : Tq;
    float* s = S + row * Tk;
    if (q >= Lq || L <= 0) {
      for (int k = lane; k < Tk; k += 64) s[k] = 0.f;
      continue;
    }
    if (!BWD) {
      float mx = -INFINITY;
      for (int k = lane; k < L; k += 64) mx = fmaxf(mx, s[k]);
      mx = ctts_wave_max(mx);
      float sum = 0.f;
      for (int k = lane; k < L; k += 64) sum += expf(s[k] - mx);
      const float inv = 1.f / ctts_wave_sum(sum);
      for (int k = lane; k < Tk; k += 64) s[k] = k < L ? expf(s[k] - mx) * inv : 0.f;
    } else {
      const float* p = P + row * Tk;
      float dot = 0.f;
      for (int k = lane; k < L; k += 64) dot += s[k] * p[k];
      dot = ctts_wave_sum(dot);
      for (int k = lane; k < Tk; k += 64) s[k] = k < L ? p[k] * (s[k] - dot) : 0.f;
    }
  }
}

template <int H>
int launch_gru_fwd(const float* gi, const float* whh, const float* bhh, float* out, float* gates, int B, int T, int ndir, int rev_mask,
                   hipStream_t st) {
  hipLaunchKernelGGL((gru_fwd_kernel<H>), dim3(B, ndir), dim3((3 * H + 63) / 64 * 64), 0, st, gi, whh, bhh, out, gates, T, ndir, rev_mask);
  CTTS_CHECK_LAUNCH("ctts_gru_fwd");
  return 0;
}
template <int H>
int launch_gru_bwd(const float* dout, const float* out, const float* gates, const float* whh, float* dgi, float* dgh, float* hprev,
                   int B, int T, int ndir, int rev_mask, hipStream_t st) {
  hipLaunchKernelGGL((gru_bwd_kernel<H>), dim3(B, ndir), dim3((3 * H + 63) / 64 * 64), 0, st, dout, out, gates, whh, dgi, dgh, hprev, T,
                     ndir, rev_mask);
  CTTS_CHECK_LAUNCH("ctts_gru_bwd");
  return 0;
}

}  // namespace

extern "C" int ctts_im2col_3x3s2(const float* x, float* col, int B, int T, int W, int C, void* stream) {
  CTTS_REQUIRE(x && col && B >= 0 && T > 0 && W > 0 && C > 0 && (C % 4) == 0, "ctts_im2col_3x3s2: bad arguments (C %% 4 must be 0)");
  const int Wo = (W - 1) / 2 + 1;
  const long total = (long)B * T * Wo * 9 * (C / 4);
  if (total == 0) return 0;
  hipLaunchKernelGGL(im2col_3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)col, T, W,
                     Wo, C / 4, total);
  CTTS_CHECK_LAUNCH("ctts_im2col_3x3s2");
  return 0;
}

extern "C" int ctts_col2im_3x3s2(const float* dcol, float* dx, int B, int T, int W, int C, void* stream) {
  CTTS_REQUIRE(dcol && dx && B >= 0 && T > 0 && W > 0 && C > 0 && (C % 4) == 0, "ctts_col2im_3x3s2: bad arguments (C %% 4 must be 0)");
  const int Wo = (W - 1) / 2 + 1;
  const long total = (long)B * T * W * (C / 4);
  if (total == 0) return 0;
  hipLaunchKernelGGL(col2im_3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)dcol, (float4*)dx, T,
                     W, Wo, C / 4, total);
  CTTS_CHECK_LAUNCH("ctts_col2im_3x3s2");
  return 0;
}

extern "C" int ctts_gru_fwd(const float* gi, const float* whh, const float* bhh, float* out, float* gates, int B, int T, int H, int ndir,
                            int rev_mask, void* stream) {
  CTTS_REQUIRE(gi && whh && bhh && out && B >= 0 && T >= 0 && ndir >= 1 && ndir <= 8, "ctts_gru_fwd: bad arguments");
  if (B == 0 || T == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (H) {
    case 16: return launch_gru_fwd<16>(gi, whh, bhh, out, gates, B, T, ndir, rev_mask, st);
    case 32:
      if (getenv("CTTS_GRU_MULTIWAVE")) return launch_gru_fwd<32>(gi, whh, bhh, out, gates, B, T, ndir, rev_mask, st);
      hipLaunchKernelGGL(gru_fwd_wave32_kernel, dim3(B, ndir), dim3(64), 0, st, gi, whh, bhh, out, gates, T, ndir, rev_mask);
      CTTS_CHECK_LAUNCH("ctts_gru_fwd(wave32)");
      return 0;
    case 64: return launch_gru_fwd<64>(gi, whh, bhh, out, gates, B, T, ndir, rev_mask, st);
    case 128: return launch_gru_fwd<128>(gi, whh, bhh, out, gates, B, T, ndir, rev_mask, st);
    default:
      ctts_set_error("ctts_gru_fwd: hidden size %d is not instantiated (16, 32, 64, 128: W_hh rows live in registers)", H);
      return -1;
  }
}

extern "C" int ctts_gru_bwd(const float* dout, const float* out, const float* gates, const float* whh, float* dgi, float* dgh,
                            float* hprev, int B, int T, int H, int ndir, int rev_mask, void* stream) {
  CTTS_REQUIRE(dout && out && gates && whh && dgi && dgh && hprev && B >= 0 && T >= 0 && ndir >= 1 && ndir <= 8,
               "ctts_gru_bwd: bad arguments");
  if (B == 0 || T == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (H) {
    case 16: return launch_gru_bwd<16>(dout, out, gates, whh, dgi, dgh, hprev, B, T, ndir, rev_mask, st);
    case 32:
      if (getenv("CTTS_GRU_MULTIWAVE")) return launch_gru_bwd<32>(dout, out, gates, whh, dgi, dgh, hprev, B, T, ndir, rev_mask, st);
      hipLaunchKernelGGL(gru_bwd_wave32_kernel, dim3(B, ndir), dim3(64), 0, st, dout, out, gates, whh, dgi, dgh, hprev, T, ndir, rev_mask);
      CTTS_CHECK_LAUNCH("ctts_gru_bwd(wave32)");
      return 0;
    case 64: return launch_gru_bwd<64>(dout, out, gates, whh, dgi, dgh, hprev, B, T, ndir, rev_mask, st);
    case 128: return launch_gru_bwd<128>(dout, out, gates, whh, dgi, dgh, hprev, B, T, ndir, rev_mask, st);
    default:
      ctts_set_error("ctts_gru_bwd: hidden size %d is not instantiated (16, 32, 64, 128)", H);
      return -1;
  }
}

extern "C" int ctts_softmax_rect_fwd(float* S, const int32_t* klens, const int32_t* qlens, int nb, int Tq, int Tk, void* stream) {
  CTTS_REQUIRE(S && nb >= 0 && Tq > 0 && Tk > 0, "ctts_softmax_rect_fwd: bad arguments");
  const long nrows = (long)nb * Tq;
  if (nrows == 0) return 0;
  const int blocks = (int)((nrows + 3) / 4 > 8192 ? 8192 : (nrows + 3) / 4);
  hipLaunchKernelGGL((softmax_rect_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, (const float*)nullptr, klens,
                     qlens, Tq, Tk, nrows);
  CTTS_CHECK_LAUNCH("ctts_softmax_rect_fwd");
  return 0;
}

extern "C" int ctts_softmax_rect_bwd(const float* P, float* dP, const int32_t* klens, const int32_t* qlens, int nb, int Tq, int Tk,
                                     void* stream) {
  CTTS_REQUIRE(P && dP && nb >= 0 && Tq > 0 && Tk > 0, "ctts_softmax_rect_bwd: bad arguments");
  const long nrows = (long)nb * Tq;
  if (nrows == 0) return 0;
  const int blocks = (int)((nrows + 3) / 4 > 8192 ? 8192 : (nrows + 3) / 4);
  hipLaunchKernelGGL((softmax_rect_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dP, P, klens, qlens, Tq, Tk, nrows);
  CTTS_CHECK_LAUNCH("ctts_softmax_rect_bwd");
  return 0;
}
