// Variance / duration loss terms of CompTransTTSLoss (reference model/loss.py:123-243) as ONE kernel pair.
//
//   pdur  = lambda_ph   * sum nonpad (log_d - log(dur+1))^2 / sum nonpad                                    loss.py:140-152
//   wdur  = lambda_word * sum_w wn (log(wp+1) - log(wg+1))^2 / sum wn,  wp/wg = per-word sums of the linear
//           durations (words = runs between silence tokens, word id 0 = silences / leading tokens, dropped)   loss.py:154-161
//   sdur  = lambda_sent * mean_b (log(sum_t dur_lin + 1) - log(sum_t dur + 1))^2                             loss.py:162-166
//   C     = lambda_f0 * mean |cwt[..., :10] - cwt_spec|   (or squared, cwt_loss = l2; padded frames INCLUDED) loss.py:188-194,226-232
//   uv    = lambda_uv * sum nonpad BCEwithlogits(cwt[..., 10], uv) / sum nonpad                              loss.py:195-199
//   f0_mean, f0_std = lambda_f0 * mean_b |pred - target|                                                     loss.py:200-201
//   energy = sum nonpad |e_pred - e_tgt| / sum nonpad   (= l1_loss over masked_select)                       loss.py:234-243
//
// The reference spends ~60 tiny launches (and as many again in backward) on these [B,Ts] / [B,Tm]-sized tensors; worse, torch's
// multi-block reduction (the 164 k-element cwt mean) clears its semaphores with a memset node, which mis-replays inside a hipGraph on
// this ROCm stack - the replayed loss VALUE was intermittently garbage.  Here: one workgroup per utterance produces 13 partial sums
// (fixed-order tree reductions, no atomics, no memset), a one-block second stage folds them over the batch in order; the backward is
// one launch that writes all five gradients.
#include "ctts_common.h"

namespace {

constexpr int NP = 16;      // partials per utterance
enum { P_PD = 0, P_NP, P_SENT, P_WL, P_WN, P_C, P_UV, P_MNP, P_F0M, P_F0S, P_E, P_SP, P_SG };

// deterministic block sum (256 threads): wave butterflies, then waves combined in index order
__device__ __forceinline__ float block_sum(float v, float* s4) {
  v = ctts_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

struct VarLossArgs {
  const float* log_d; const void* dur; int dur_is_float; const int64_t* texts; const uint8_t* src_pad;
  const float* cwt; const float* cwt_spec; const float* uv; const uint8_t* mel_pad;
  const float *f0m_p, *f0m_t, *f0s_p, *f0s_t;
  const float *e_pred, *e_tgt;
  int B, Ts, Tm;
  float lam_ph, lam_word, lam_sent, lam_f0, lam_uv;
  int cwt_l2;
  int64_t sil0, sil1, sil2;
  float* partials;      // [B, NP]
  float* wsum;          // [B, 2, Ts+1]  word sums of predicted / target durations (kept for the backward)
  float* terms;         // [8]  pdur, wdur, sdur, C, uv, f0_mean, f0_std, energy   (lambda-weighted)
  float* denoms;        // [4]  sum src nonpad, sum wn, sum mel nonpad, spare
  // backward
  const float* g;       // [8] upstream gradient of every term
  float *d_log_d, *d_cwt, *d_f0m, *d_f0s, *d_e;
};

__device__ __forceinline__ float dur_at(const VarLossArgs& a, long i) {
  return a.dur_is_float ? static_cast<const float*>(a.dur)[i] : (float)static_cast<const int64_t*>(a.dur)[i];
}
__device__ __forceinline__ bool is_sil(const VarLossArgs& a, int64_t tok) { return tok == a.sil0 || tok == a.sil1 || tok == a.sil2; }

__global__ __launch_bounds__(256) void var_loss_fwd_kernel(const VarLossArgs a) {
  extern __shared__ float sw[];                 // wp[Ts+1] | wg[Ts+1]
  __shared__ float s4[4];
  const int b = blockIdx.x, tid = threadIdx.x, Ts = a.Ts, Tm = a.Tm;
  float* wp = sw; float* wg = sw + Ts + 1;
  for (int w = tid; w <= Ts; w += 256) { wp[w] = 0.f; wg[w] = 0.f; }
  float pd = 0.f, np = 0.f, sp = 0.f, sg = 0.f, en = 0.f;
  for (int t = tid; t < Ts; t += 256) {
    const long i = (long)b * Ts + t;
    const float nonpad = a.src_pad[i] ? 0.f : 1.f;
    const float dg = dur_at(a, i) * nonpad;
    const float ld = a.log_d[i];
    const float e = ld - logf(dg + 1.f);
    pd += e * e * nonpad; np += nonpad;
    sp += fmaxf(expf(ld) - 1.f, 0.f); sg += dg;
    en += fabsf(a.e_pred[i] - a.e_tgt[i]) * nonpad;
  }
  pd = block_sum(pd, s4); np = block_sum(np, s4); sp = block_sum(sp, s4); sg = block_sum(sg, s4); en = block_sum(en, s4);
  float wl = 0.f, wn = 0.f;
  if (a.lam_word > 0.f) {
    if (tid == 0) {                             // word segments: one sequential walk (Ts <= a few hundred tokens)
      int cs = 0;
      for (int t = 0; t < Ts; ++t) {
        const long i = (long)b * Ts + t;
        const bool sil = is_sil(a, a.texts[i]);
        cs += sil ? 1 : 0;
        const int wid = sil ? 0 : cs;
        const float nonpad = a.src_pad[i] ? 0.f : 1.f;
        wp[wid] += fmaxf(expf(a.log_d[i]) - 1.f, 0.f);
        wg[wid] += dur_at(a, i) * nonpad;
      }
    }
    __syncthreads();
    for (int w = 1 + tid; w <= Ts; w += 256) {
      const float e = logf(wp[w] + 1.f) - logf(wg[w] + 1.f);
      const float n = wg[w] > 0.f ? 1.f : 0.f;
      wl += e * e * n; wn += n;
    }
    for (int w = tid; w <= Ts; w += 256) {
      a.wsum[((long)b * 2 + 0) * (Ts + 1) + w] = wp[w];
      a.wsum[((long)b * 2 + 1) * (Ts + 1) + w] = wg[w];
    }
    wl = block_sum(wl, s4); wn = block_sum(wn, s4);
  }
  float cc = 0.f, uvs = 0.f, mnp = 0.f;
  for (int t = tid; t < Tm; t += 256) {
    const long r = (long)b * Tm + t;
    const float* cp = a.cwt + r * 11;
    const float* sp10 = a.cwt_spec + r * 10;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const float df = cp[c] - sp10[c];
      cc += a.cwt_l2 ? df * df : fabsf(df);
    }
    const float nonpad = a.mel_pad[r] ? 0.f : 1.f;
    const float x = cp[10], y = a.uv[r];
    uvs += (fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)))) * nonpad;
    mnp += nonpad;
  }
  cc = block_sum(cc, s4); uvs = block_sum(uvs, s4); mnp = block_sum(mnp, s4);
  if (tid == 0) {
    float* p = a.partials + (long)b * NP;
    const float se = logf(sp + 1.f) - logf(sg + 1.f);
    p[P_PD] = pd; p[P_NP] = np; p[P_SENT] = se * se; p[P_WL] = wl; p[P_WN] = wn; p[P_C] = cc; p[P_UV] = uvs; p[P_MNP] = mnp;
    p[P_F0M] = fabsf(a.f0m_p[b] - a.f0m_t[b]); p[P_F0S] = fabsf(a.f0s_p[b] - a.f0s_t[b]); p[P_E] = en; p[P_SP] = sp; p[P_SG] = sg;
  }
}

__global__ void var_loss_finalize_kernel(const VarLossArgs a) {
  __shared__ float tot[NP];
  const int k = threadIdx.x;
  if (k < NP) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += a.partials[(long)b * NP + k];     // fixed order over the batch
    tot[k] = s;
  }
  __syncthreads();
  if (k == 0) {
    const float Bf = (float)a.B;
    a.terms[0] = tot[P_PD] / tot[P_NP] * a.lam_ph;
    a.terms[1] = a.lam_word > 0.f ? tot[P_WL] / tot[P_WN] * a.lam_word : 0.f;     // 0/0 = NaN without a silence token, as the reference
    a.terms[2] = a.lam_sent > 0.f ? tot[P_SENT] / Bf * a.lam_sent : 0.f;
    a.terms[3] = tot[P_C] / (Bf * (float)a.Tm * 10.f) * a.lam_f0;
    a.terms[4] = tot[P_UV] / tot[P_MNP] * a.lam_uv;
    a.terms[5] = tot[P_F0M] / Bf * a.lam_f0;
    a.terms[6] = tot[P_F0S] / Bf * a.lam_f0;
    a.terms[7] = tot[P_E] / tot[P_NP];
    a.denoms[0] = tot[P_NP]; a.denoms[1] = tot[P_WN]; a.denoms[2] = tot[P_MNP]; a.denoms[3] = 0.f;
  }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void var_loss_bwd_kernel(const VarLossArgs a) {
  extern __shared__ float sw[];                 // word id per token (as float bits)
  int* wid = reinterpret_cast<int*>(sw);
  const int b = blockIdx.x, tid = threadIdx.x, Ts = a.Ts, Tm = a.Tm;
  const float Bf = (float)a.B;
  const float sum_np = a.denoms[0], sum_wn = a.denoms[1], sum_mnp = a.denoms[2];
  if (a.lam_word > 0.f) {
    if (tid == 0) {
      int cs = 0;
      for (int t = 0; t < Ts; ++t) {
        const bool sil = is_sil(a, a.texts[(long)b * Ts + t]);
        cs += sil ? 1 : 0;
        wid[t] = sil ? 0 : cs;
      }
    }
    __syncthreads();
  }
  const float* p = a.partials + (long)b * NP;
  const float sp = p[P_SP], sg = p[P_SG];
  const float g_pd = a.g[0] * a.lam_ph / sum_np, g_w = a.lam_word > 0.f ? a.g[1] * a.lam_word / sum_wn : 0.f;
  const float g_s = a.lam_sent > 0.f ? a.g[2] * a.lam_sent / Bf * 2.f * (logf(sp + 1.f) - logf(sg + 1.f)) / (sp + 1.f) : 0.f;
  const float g_e = a.g[7] / sum_np;
  for (int t = tid; t < Ts; t += 256) {
    const long i = (long)b * Ts + t;
    const float nonpad = a.src_pad[i] ? 0.f : 1.f;
    const float dg = dur_at(a, i) * nonpad;
    const float ld = a.log_d[i];
    const float ex = expf(ld);
    const float dlin = (ex - 1.f) > 0.f ? ex : 0.f;          // d clamp(exp(x) - 1, min 0) / dx
    float gd = g_pd * 2.f * (ld - logf(dg + 1.f)) * nonpad + g_s * dlin;
    if (a.lam_word > 0.f) {
      const int w = wid[t];
      if (w >= 1) {
        const float wpv = a.wsum[((long)b * 2 + 0) * (Ts + 1) + w], wgv = a.wsum[((long)b * 2 + 1) * (Ts + 1) + w];
        if (wgv > 0.f) gd += g_w * 2.f * (logf(wpv + 1.f) - logf(wgv + 1.f)) / (wpv + 1.f) * dlin;
      }
    }
    a.d_log_d[i] = gd;
    a.d_e[i] = g_e * sgn(a.e_pred[i] - a.e_tgt[i]) * nonpad;
  }
  const float g_c = a.g[3] * a.lam_f0 / (Bf * (float)Tm * 10.f), g_uv = a.g[4] * a.lam_uv / sum_mnp;
  for (int t = tid; t < Tm; t += 256) {
    const long r = (long)b * Tm + t;
    const float* cp = a.cwt + r * 11;
    const float* sp10 = a.cwt_spec + r * 10;
    float* dc = a.d_cwt + r * 11;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const float df = cp[c] - sp10[c];
      dc[c] = a.cwt_l2 ? g_c * 2.f * df : g_c * sgn(df);
    }
    const float nonpad = a.mel_pad[r] ? 0.f : 1.f;
    const float x = cp[10];
    dc[10] = g_uv * (1.f / (1.f + expf(-x)) - a.uv[r]) * nonpad;
  }
  if (tid == 0) {
    a.d_f0m[b] = a.g[5] * a.lam_f0 / Bf * sgn(a.f0m_p[b] - a.f0m_t[b]);
    a.d_f0s[b] = a.g[6] * a.lam_f0 / Bf * sgn(a.f0s_p[b] - a.f0s_t[b]);
  }
}

// ---- BinLoss (loss.py:380-386): -sum log(clamp(soft, 1e-12)) * hard / sum hard, two-stage deterministic reduction
constexpr int BIN_BLOCKS = 512;
__global__ __launch_bounds__(256) void bin_loss_partial_kernel(const float* __restrict__ soft, const float* __restrict__ hard, long n,
                                                                float* __restrict__ partials) {
  __shared__ float s4[4];
  float a = 0.f, h = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float hv = hard[i];
    a += logf(fmaxf(soft[i], 1e-12f)) * hv;
    h += hv;
  }
  a = block_sum(a, s4); h = block_sum(h, s4);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = h; }
}
__global__ __launch_bounds__(256) void bin_loss_finalize_kernel(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
  __shared__ float s4[4];
  float a = 0.f, h = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) { a += partials[2 * i]; h += partials[2 * i + 1]; }
  a = block_sum(a, s4); h = block_sum(h, s4);
  if (threadIdx.x == 0) { out[0] = -a / h; out[1] = h; }
}
__global__ void bin_loss_bwd_kernel(const float* __restrict__ soft, const float* __restrict__ hard, const float* __restrict__ out,
                                    const float* __restrict__ g, float* __restrict__ dsoft, long n) {
  const float k = -g[0] / out[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float s = soft[i];
    dsoft[i] = s > 1e-12f ? k * hard[i] / s : 0.f;
  }
}

// ---- masked mean loss: sum_i w_i * l(p_i, t_i) / sum_i w_i with l = |p-t| (kind 0), (p-t)^2 (1), BCE-with-logits (2).  The terms of the
// pitch_type "frame" / "ph" and frame-level energy configurations (loss.py:173-178,202-219,234-243) - same two-stage ordered reduction
__device__ __forceinline__ float masked_term(int kind, float p, float t) {
  if (kind == 0) return fabsf(p - t);
  if (kind == 1) return (p - t) * (p - t);
  return fmaxf(p, 0.f) - p * t + log1pf(expf(-fabsf(p)));
}
__global__ __launch_bounds__(256) void masked_loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                                   const float* __restrict__ w, long n, int kind, float* __restrict__ partials) {
  __shared__ float s4[4];
  float a = 0.f, h = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float wv = w[i];
    a += masked_term(kind, pred[i], tgt[i]) * wv;
    h += wv;
  }
  a = block_sum(a, s4); h = block_sum(h, s4);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = h; }
}
__global__ __launch_bounds__(256) void masked_loss_finalize_kernel(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
  __shared__ float s4[4];
  float a = 0.f, h = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) { a += partials[2 * i]; h += partials[2 * i + 1]; }
  a = block_sum(a, s4); h = block_sum(h, s4);
  if (threadIdx.x == 0) { out[0] = a / h; out[1] = h; }
}
__global__ void masked_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ w,
                                       const float* __restrict__ out, const float* __restrict__ g, float* __restrict__ dpred, long n, int kind) {
  const float k = g[0] / out[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float p = pred[i], t = tgt[i];
    float d;
    if (kind == 0) d = p > t ? 1.f : (p < t ? -1.f : 0.f);
    else if (kind == 1) d = 2.f * (p - t);
    else d = 1.f / (1.f + expf(-p)) - t;
    dpred[i] = k * w[i] * d;
  }
}

}  // namespace

extern "C" int ctts_var_loss_fwd(const float* log_d, const void* dur, int dur_is_float, const int64_t* texts, const uint8_t* src_pad,
                                 const float* cwt, const float* cwt_spec, const float* uv, const uint8_t* mel_pad, const float* f0m_p,
                                 const float* f0m_t, const float* f0s_p, const float* f0s_t, const float* e_pred, const float* e_tgt, int B,
                                 int Ts, int Tm, const float* lambdas5, int cwt_l2, const int64_t* sil_ids3, float* partials, float* wsum,
                                 float* terms, float* denoms, void* stream) {
  CTTS_REQUIRE(log_d && dur && texts && src_pad && cwt && cwt_spec && uv && mel_pad && f0m_p && f0m_t && f0s_p && f0s_t && e_pred && e_tgt &&
               lambdas5 && sil_ids3 && partials && wsum && terms && denoms && B > 0 && Ts > 0 && Tm > 0, "ctts_var_loss_fwd: bad arguments");
  CTTS_REQUIRE((size_t)(Ts + 1) * 8 <= 60 * 1024, "ctts_var_loss_fwd: Ts too large for the word-sum LDS buffer");
  VarLossArgs a = {};
  a.log_d = log_d; a.dur = dur; a.dur_is_float = dur_is_float; a.texts = texts; a.src_pad = src_pad;
  a.cwt = cwt; a.cwt_spec = cwt_spec; a.uv = uv; a.mel_pad = mel_pad;
  a.f0m_p = f0m_p; a.f0m_t = f0m_t; a.f0s_p = f0s_p; a.f0s_t = f0s_t; a.e_pred = e_pred; a.e_tgt = e_tgt;
  a.B = B; a.Ts = Ts; a.Tm = Tm;
  a.lam_ph = lambdas5[0]; a.lam_word = lambdas5[1]; a.lam_sent = lambdas5[2]; a.lam_f0 = lambdas5[3]; a.lam_uv = lambdas5[4];
  a.cwt_l2 = cwt_l2; a.sil0 = sil_ids3[0]; a.sil1 = sil_ids3[1]; a.sil2 = sil_ids3[2];
  a.partials = partials; a.wsum = wsum; a.terms = terms; a.denoms = denoms;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(var_loss_fwd_kernel, dim3(B), dim3(256), (size_t)(Ts + 1) * 8, st, a);
  CTTS_CHECK_LAUNCH("ctts_var_loss_fwd");
  hipLaunchKernelGGL(var_loss_finalize_kernel, dim3(1), dim3(64), 0, st, a);
  CTTS_CHECK_LAUNCH("ctts_var_loss_fwd(finalize)");
  return 0;
}

extern "C" int ctts_var_loss_bwd(const float* log_d, const void* dur, int dur_is_float, const int64_t* texts, const uint8_t* src_pad,
                                 const float* cwt, const float* cwt_spec, const float* uv, const uint8_t* mel_pad, const float* f0m_p,
                                 const float* f0m_t, const float* f0s_p, const float* f0s_t, const float* e_pred, const float* e_tgt, int B,
                                 int Ts, int Tm, const float* lambdas5, int cwt_l2, const int64_t* sil_ids3, const float* partials,
                                 const float* wsum, const float* denoms, const float* g8, float* d_log_d, float* d_cwt, float* d_f0m,
                                 float* d_f0s, float* d_e, void* stream) {
  CTTS_REQUIRE(log_d && dur && texts && src_pad && cwt && cwt_spec && uv && mel_pad && f0m_p && f0m_t && f0s_p && f0s_t && e_pred && e_tgt &&
               lambdas5 && sil_ids3 && partials && wsum && denoms && g8 && d_log_d && d_cwt && d_f0m && d_f0s && d_e && B > 0 && Ts > 0 && Tm > 0,
               "ctts_var_loss_bwd: bad arguments");
  VarLossArgs a = {};
  a.log_d = log_d; a.dur = dur; a.dur_is_float = dur_is_float; a.texts = texts; a.src_pad = src_pad;
  a.cwt = cwt; a.cwt_spec = cwt_spec; a.uv = uv; a.mel_pad = mel_pad;
  a.f0m_p = f0m_p; a.f0m_t = f0m_t; a.f0s_p = f0s_p; a.f0s_t = f0s_t; a.e_pred = e_pred; a.e_tgt = e_tgt;
  a.B = B; a.Ts = Ts; a.Tm = Tm;
  a.lam_ph = lambdas5[0]; a.lam_word = lambdas5[1]; a.lam_sent = lambdas5[2]; a.lam_f0 = lambdas5[3]; a.lam_uv = lambdas5[4];
  a.cwt_l2 = cwt_l2; a.sil0 = sil_ids3[0]; a.sil1 = sil_ids3[1]; a.sil2 = sil_ids3[2];
  a.partials = const_cast<float*>(partials); a.wsum = const_cast<float*>(wsum); a.denoms = const_cast<float*>(denoms);
  a.g = g8; a.d_log_d = d_log_d; a.d_cwt = d_cwt; a.d_f0m = d_f0m; a.d_f0s = d_f0s; a.d_e = d_e;
  hipLaunchKernelGGL(var_loss_bwd_kernel, dim3(B), dim3(256), (size_t)(Ts + 1) * 4, (hipStream_t)stream, a);
  CTTS_CHECK_LAUNCH("ctts_var_loss_bwd");
  return 0;
}

extern "C" int ctts_bin_loss_fwd(const float* soft, const float* hard, int64_t n, float* partials, float* out2, void* stream) {
  CTTS_REQUIRE(soft && hard && partials && out2 && n > 0, "ctts_bin_loss_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((n + 255) / 256 > BIN_BLOCKS ? BIN_BLOCKS : (n + 255) / 256);
  hipLaunchKernelGGL(bin_loss_partial_kernel, dim3(blocks), dim3(256), 0, st, soft, hard, (long)n, partials);
  CTTS_CHECK_LAUNCH("ctts_bin_loss_fwd");
  hipLaunchKernelGGL(bin_loss_finalize_kernel, dim3(1), dim3(256), 0, st, partials, blocks, out2);
  CTTS_CHECK_LAUNCH("ctts_bin_loss_fwd(finalize)");
  return 0;
}

extern "C" int ctts_bin_loss_bwd(const float* soft, const float* hard, const float* out2, const float* g, float* dsoft, int64_t n, void* stream) {
  CTTS_REQUIRE(soft && hard && out2 && g && dsoft && n > 0, "ctts_bin_loss_bwd: bad arguments");
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(bin_loss_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, soft, hard, out2, g, dsoft, (long)n);
  CTTS_CHECK_LAUNCH("ctts_bin_loss_bwd");
  return 0;
}

extern "C" int ctts_masked_loss_fwd(const float* pred, const float* target, const float* weight, int64_t n, int kind, float* partials,
                                    float* out2, void* stream) {
  CTTS_REQUIRE(pred && target && weight && partials && out2 && n > 0 && kind >= 0 && kind <= 2, "ctts_masked_loss_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((n + 255) / 256 > BIN_BLOCKS ? BIN_BLOCKS : (n + 255) / 256);
  hipLaunchKernelGGL(masked_loss_partial_kernel, dim3(blocks), dim3(256), 0, st, pred, target, weight, (long)n, kind, partials);
  CTTS_CHECK_LAUNCH("ctts_masked_loss_fwd");
  hipLaunchKernelGGL(masked_loss_finalize_kernel, dim3(1), dim3(256), 0, st, partials, blocks, out2);
  CTTS_CHECK_LAUNCH("ctts_masked_loss_fwd(finalize)");
  return 0;
}

extern "C" int ctts_masked_loss_bwd(const float* pred, const float* target, const float* weight, const float* out2, const float* g,
                                    float* dpred, int64_t n, int kind, void* stream) {
  CTTS_REQUIRE(pred && target && weight && out2 && g && dpred && n > 0 && kind >= 0 && kind <= 2, "ctts_masked_loss_bwd: bad arguments");
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(masked_loss_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, weight, out2, g, dpred, (long)n, kind);
  CTTS_CHECK_LAUNCH("ctts_masked_loss_bwd");
  return 0;
}
