// Fused global-norm clip + Adam over ONE flat fp32 parameter arena (reference train.py:118-125 =
// nn.utils.clip_grad_norm_(model.parameters(), 1.0) + ScheduledOptim/torch.optim.Adam step, model/optimizer.py:22-53).
//
// The reference walks ~170 parameter tensors three times (norm, scale, Adam).  Here parameters, gradients and both moments are flat
// arenas (views handed to torch), so the whole update is two streaming passes: (1) sum of squares of the gradient arena,
// (2) one kernel that applies the clip coefficient and the Adam update (reads g,p,m,v, writes p,m,v: 28 B per parameter - HBM bound).
// Step counter, learning rate and the norm accumulator live in device memory so the launches replay inside a hipGraph.
#include "ctts_common.h"

namespace {

// state[0] = sum of squares of the gradient (written by sqnorm_reduce_kernel), state[1] = step count (float), state[2] = last total
// gradient norm (output, for logging like clip_grad_norm_'s return value), state[3 .. 3 + CTTS_ADAM_PARTIALS) = per-block partial sums.
// The reduction is DETERMINISTIC (fixed block -> partial assignment, fixed-order tree over the partials, no atomics): data-parallel
// replicas that hold bit-identical gradients after the all-reduce must compute bit-identical clip coefficients, or they drift apart.
__global__ __launch_bounds__(256) void sqnorm_kernel(const float4* __restrict__ g, long n4, const float* __restrict__ gtail, int ntail,
                                                      float* __restrict__ state) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = g[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) acc += gtail[threadIdx.x] * gtail[threadIdx.x];
  acc = ctts_wave_sum(acc);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) state[3 + blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void sqnorm_reduce_kernel(float* __restrict__ state, int nparts) {
  __shared__ float s[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += state[3 + i];       // thread t: partials t, t+256, ... in order
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) state[0] = s[0];
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float coef, float wd, float b1, float b2, float eps,
                                         float step_size, float inv_sqrt_bc2) {
  g = g * coef + wd * p;
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, const float* __restrict__ lr, float b1, float b2,
                                                    float eps, float wd, float max_norm, const float* __restrict__ state) {
  const float total = sqrtf(state[0]);
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(1.f, max_norm / (total + 1e-6f));        // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
  const float step = state[1] + 1.f;
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  const float step_size = lr[0] / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  const long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, coef, wd, b1, b2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.y, gg.y, mm.y, vv.y, coef, wd, b1, b2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.z, gg.z, mm.z, vv.z, coef, wd, b1, b2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.w, gg.w, mm.w, vv.w, coef, wd, b1, b2, eps, step_size, inv_sqrt_bc2);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  if (blockIdx.x == 0) {
    const long i = (n4 << 2) + threadIdx.x;
    if (i < n) adam_one(p[i], g[i], m[i], v[i], coef, wd, b1, b2, eps, step_size, inv_sqrt_bc2);
  }
}

__global__ void adam_finalize_kernel(float* __restrict__ state) {
  state[2] = sqrtf(state[0]);
  state[1] += 1.f;
}

}  // namespace

extern "C" int ctts_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float beta1, float beta2,
                                   float eps, float weight_decay, float max_norm, float* state, void* stream) {
  CTTS_REQUIRE(p && g && m && v && lr && state && n >= 0, "ctts_adam_clip_step: bad arguments");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  CTTS_REQUIRE(al16(p) && al16(g) && al16(m) && al16(v), "ctts_adam_clip_step: arenas must be 16-byte aligned");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const long n4 = n >> 2;
  const int blocks = (int)((n4 + 255) / 256 > CTTS_ADAM_PARTIALS ? CTTS_ADAM_PARTIALS : ((n4 + 255) / 256 < 1 ? 1 : (n4 + 255) / 256));
  hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)g, n4, g + (n4 << 2), (int)(n - (n4 << 2)), state);
  CTTS_CHECK_LAUNCH("ctts_adam_clip_step(sqnorm)");
  hipLaunchKernelGGL(sqnorm_reduce_kernel, dim3(1), dim3(256), 0, st, state, blocks);
  CTTS_CHECK_LAUNCH("ctts_adam_clip_step(sqnorm reduce)");
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, max_norm, state);
  CTTS_CHECK_LAUNCH("ctts_adam_clip_step(adam)");
  hipLaunchKernelGGL(adam_finalize_kernel, dim3(1), dim3(1), 0, st, state);
  CTTS_CHECK_LAUNCH("ctts_adam_clip_step(finalize)");
  return 0;
}

// ---------------------------------------------------------------- masked L1 of the two mel predictions
// CompTransTTSLoss.get_mel_loss (model/loss.py:130-138, as l1_loss of transformer_fs2): with target rows zeroed at padded frames,
//   weights w = (sum_c |target[row,c]| != 0), loss = sum |pred - target| * w / (n_mel * sum w)   (per prediction; pads contribute 0).
// One pass over (mel, postnet_mel, target) produces both numerators and the shared denominator; the backward writes both gradients.
namespace {
__global__ __launch_bounds__(256) void mel_l1_fwd_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                          const float* __restrict__ tgt, const unsigned char* __restrict__ pad,
                                                          float* __restrict__ sums, float* __restrict__ roww, long rows, int C,
                                                          unsigned char* ws, int G) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a1 = 0.f, a2 = 0.f, aw = 0.f;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const bool is_pad = pad[r] != 0;
    float s_t = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float t = is_pad ? 0.f : tgt[r * C + c];
      const float x1 = is_pad ? 0.f : p1[r * C + c], x2 = is_pad ? 0.f : p2[r * C + c];
      s_t += fabsf(t); s1 += fabsf(x1 - t); s2 += fabsf(x2 - t);
    }
    s_t = ctts_wave_sum(s_t); s1 = ctts_wave_sum(s1); s2 = ctts_wave_sum(s2);
    const float w = s_t != 0.f ? 1.f : 0.f;
    if (lane == 0) roww[r] = w;
    a1 += s1 * w; a2 += s2 * w; aw += w;
  }
  __shared__ float s[3][4];
  if (lane == 0) { s[0][wave] = a1; s[1][wave] = a2; s[2][wave] = aw; }
  __syncthreads();
  // ordered cross-workgroup sum (ctts_common.h): lane 0 carries the three partials; the elected workgroup WRITES sums[0..2]
  float tot[3] = {0.f, 0.f, 0.f};
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) tot[k] = s[k][0] + s[k][1] + s[k][2] + s[k][3];
  }
  if (!ctts_ordered_colsum<float, 3>(tot, ws, 0, blockIdx.x, gridDim.x, G)) return;
  if (threadIdx.x == 0) { sums[0] = tot[0]; sums[1] = tot[1]; sums[2] = tot[2]; }
}

// d loss_k / d p_k = g_k * sign(p_k - t) * w[row] / (C * sum w)
__global__ void mel_l1_bwd_kernel(const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ tgt,
                                  const float* __restrict__ roww, const float* __restrict__ sums, const float* __restrict__ g,
                                  float* __restrict__ d1, float* __restrict__ d2, long total, int C) {
  const float inv = 1.f / ((float)C * sums[2]);
  const float g1 = g[0] * inv, g2 = g[1] * inv;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const float w = roww[e / C], t = tgt[e];
    const float x1 = p1[e] - t, x2 = p2[e] - t;
    d1[e] = w * g1 * (x1 > 0.f ? 1.f : (x1 < 0.f ? -1.f : 0.f));
    d2[e] = w * g2 * (x2 > 0.f ? 1.f : (x2 < 0.f ? -1.f : 0.f));
  }
}
}  // namespace

extern "C" int ctts_mel_l1_fwd(const float* p1, const float* p2, const float* tgt, const uint8_t* pad, float* sums, float* roww,
                               int64_t rows, int C, void* ws, void* stream) {
  CTTS_REQUIRE(p1 && p2 && tgt && pad && sums && roww && rows >= 0 && C > 0, "ctts_mel_l1_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) return ctts_zero_async(sums, 3 * sizeof(float), st) == 0 ? 0 : -2;
  const int blocks = !ws ? 1 : (int)((rows + 3) / 4 > 1024 ? 1024 : (rows + 3) / 4);       // sums[0..2] are WRITTEN (no zero fill needed)
  hipLaunchKernelGGL(mel_l1_fwd_kernel, dim3(blocks), dim3(256), 0, st, p1, p2, tgt, pad, sums, roww, (long)rows, C,
                     (unsigned char*)ws, ctts_red_group(blocks));
  CTTS_CHECK_LAUNCH("ctts_mel_l1_fwd");
  return 0;
}

extern "C" int ctts_mel_l1_bwd(const float* p1, const float* p2, const float* tgt, const float* roww, const float* sums, const float* g,
                               float* d1, float* d2, int64_t rows, int C, void* stream) {
  CTTS_REQUIRE(p1 && p2 && tgt && roww && sums && g && d1 && d2 && rows >= 0 && C > 0, "ctts_mel_l1_bwd: bad arguments");
  const long total = (long)rows * C;
  if (total == 0) return 0;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(mel_l1_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p1, p2, tgt, roww, sums, g, d1, d2, total, C);
  CTTS_CHECK_LAUNCH("ctts_mel_l1_bwd");
  return 0;
}
