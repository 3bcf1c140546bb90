// Target-side pitch chain of the cwt pitch branch as ONE launch: inverse CWT -> per-utterance normalisation -> f0 -> f0_denorm -> coarse bins.
// Replaces ~25 stock-torch launches per step (cwt2f0_norm / f0_to_coarse of model.py = utils/pitch_tools.py:27-36,258-294 and the glue of
// modules.py:1071-1091), among them torch's multi-block mean / std reductions - the kind of launch whose semaphore memset mis-replays inside
// a hipGraph on this stack.  One workgroup per utterance; the time reductions (mean, unbiased std over ALL T columns,
// exactly like the reference) are two-pass sums in double through LDS, in a fixed order: deterministic, no atomics, no memset.
// Element arithmetic follows torch's op sequence in fp32 (no contraction): the same libm entry points (expf / log2f / powf / logf) as torch's
// elementwise kernels on this ROCm, so the integer bins agree with the stock-torch chain except where a value sits within an ulp of a bin edge.
#include "ctts_common.h"

namespace {

constexpr int PITCH_THREADS = 256;

__device__ __forceinline__ double pitch_block_sum(double v, double* s_red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < PITCH_THREADS / 64; ++w) t += s_red[w];
  return t;
}

__global__ __launch_bounds__(PITCH_THREADS) void cwt_pitch_kernel(const float* __restrict__ spec, long ld_spec, int nscale, const float* __restrict__ f0_mean,
                                                                    const float* __restrict__ f0_std, float std_scale, const float* __restrict__ uv,
                                                                    int uv_chan, float eps, float mel_min, float mel_max, int f0_bin,
                                                                    float* __restrict__ f0_out, float* __restrict__ denorm_out,
                                                                    long long* __restrict__ ids_out, int T, int width) {
#pragma clang fp contract(off)
  extern __shared__ float s_rec[];          // [T]
  __shared__ double s_red[PITCH_THREADS / 64];
  const int b = blockIdx.x;
  const float* sp = spec + (long)b * T * ld_spec;
  // rec[t] = sum_j spec[t][j] * (j + 3.5)^-2.5   (inverse_cwt_torch: b = (arange + 1 + 2.5)^-2.5, summed left to right)
  double sum = 0.0;
  for (int t = threadIdx.x; t < T; t += PITCH_THREADS) {
    float r = 0.f;
    for (int j = 0; j < nscale; ++j) r += sp[(long)t * ld_spec + j] * powf((float)j + 3.5f, -2.5f);
    s_rec[t] = r;
    sum += (double)r;
  }
  const double mean = pitch_block_sum(sum, s_red) / (double)T;
  double sq = 0.0;
  for (int t = threadIdx.x; t < T; t += PITCH_THREADS) { const double dlt = (double)s_rec[t] - mean; sq += dlt * dlt; }
  const double var = pitch_block_sum(sq, s_red) / (double)(T - 1);            // unbiased (torch.std default); T = 1: 0 / 0 = NaN like torch
  const float meanf = (float)mean, stdf = (float)sqrt(var);
  const float m = f0_mean[b], sd = f0_std[b] * std_scale;
  const float mel_span = mel_max - mel_min;
  for (int t = threadIdx.x; t < width; t += PITCH_THREADS) {
    const int ts = t < T ? t : T - 1;                                          // cwt2f0_norm repeats the last column up to mel2ph's width
    const float recn = (s_rec[ts] - meanf) / stdf;
    const float f0lin = expf(recn * sd + m);
    const float f0 = log2f(f0lin + eps);                                       // norm_f0, pitch_norm == "log"
    const bool unvoiced = uv ? (uv[(long)b * width + t] > 0.f) : (sp[(long)ts * ld_spec + uv_chan] > 0.f);
    const float den = unvoiced ? 0.f : powf(2.0f, f0);                         // denorm_f0: 2 ** f0, zero where unvoiced
    // f0_to_coarse (utils/pitch_tools.py:27-36)
    float mel = 1127.f * logf(1.f + den / 700.f);
    if (mel > 0.f) mel = (mel - mel_min) * (float)(f0_bin - 2) / mel_span + 1.f;
    mel = fminf(fmaxf(mel, 1.0f), (float)(f0_bin - 1));
    f0_out[(long)b * width + t] = f0;
    denorm_out[(long)b * width + t] = den;
    ids_out[(long)b * width + t] = (long long)(mel + 0.5f);
  }
}

}  // namespace

extern "C" int ctts_cwt_pitch(const float* spec, int64_t ld_spec, int nscale, const float* f0_mean, const float* f0_std, float std_scale,
                              const float* uv, int uv_chan, float eps, float mel_min, float mel_max, int f0_bin, float* f0, float* f0_denorm,
                              int64_t* ids, int B, int T, int width, void* stream) {
  CTTS_REQUIRE(spec && f0_mean && f0_std && f0 && f0_denorm && ids && B >= 0 && T >= 1 && width >= T && nscale >= 1 && ld_spec >= nscale,
               "ctts_cwt_pitch: bad arguments");
  CTTS_REQUIRE(uv || (uv_chan >= 0 && uv_chan < ld_spec && width == T), "ctts_cwt_pitch: without uv the flag is read from channel uv_chan of spec (width == T)");
  CTTS_REQUIRE((size_t)T * sizeof(float) <= 60 * 1024, "ctts_cwt_pitch: T = %d frames exceed the LDS row buffer", T);
  if (B == 0) return 0;
  hipLaunchKernelGGL(cwt_pitch_kernel, dim3(B), dim3(PITCH_THREADS), (size_t)T * sizeof(float), (hipStream_t)stream, spec, (long)ld_spec, nscale,
                     f0_mean, f0_std, std_scale, uv, uv_chan, eps, mel_min, mel_max, f0_bin, f0, f0_denorm, (long long*)ids, T, width);
  CTTS_CHECK_LAUNCH("ctts_cwt_pitch");
  return 0;
}
