// Shared device/host helpers for libctts_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ctts.h"

#define CTTS_WAVE 64

void ctts_set_error(const char* fmt, ...);
// Zero-fill `bytes` (multiple of 4) on `st` with a KERNEL.  hipMemsetAsync must not be used in this library: memset nodes captured into
// a hipGraph re-execute incorrectly on this ROCm stack (from the second replay on, bytes 8..11 of the buffer keep stale data - measured
// with tools/check_graph_memset.py), which silently corrupts accumulators when a train step is replayed.
int ctts_zero_async(void* p, size_t bytes, hipStream_t st);

#define CTTS_CHECK_LAUNCH(name)                                                     \
  do {                                                                              \
    hipError_t _e = hipGetLastError();                                              \
    if (_e != hipSuccess) {                                                         \
      ctts_set_error("%s: launch failed: %s", name, hipGetErrorString(_e));         \
      return -2;                                                                    \
    }                                                                               \
  } while (0)

#define CTTS_REQUIRE(cond, ...)                                                     \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      ctts_set_error(__VA_ARGS__);                                                  \
      return -1;                                                                    \
    }                                                                               \
  } while (0)

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(idx) is a pure function of (seed, call-site offset, element index): the backward pass
// regenerates the forward mask instead of storing it.  seed lives in device memory so that a
// captured hipGraph sees a fresh value on every replay.
__device__ __forceinline__ uint32_t ctts_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t ctts_drop_key(const uint64_t* seed, uint32_t offset) {
  uint64_t s = seed ? *seed : 0x9E3779B97F4A7C15ULL;
  uint32_t k = ctts_mix32((uint32_t)s ^ (offset * 0x9E3779B1U));
  return ctts_mix32(k + (uint32_t)(s >> 32));
}
// returns the multiplicative factor: 0 or 1/(1-p)
__device__ __forceinline__ float ctts_drop_scale(uint32_t key, uint32_t idx, float p, float inv_keep) {
  uint32_t h = ctts_mix32(idx * 0x9E3779B1U + key);
  float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}
// The same decision with the index hash split by the caller: pre = idx * 0x9E3779B1 + key arrives ready-made (lane constant + wave-uniform
// SALU term: one v_add per element instead of two quarter-rate integer multiplies), and u >= p is tested on the integer:
// (h >> 8) * 2^-24 >= p  <=>  h >= ceil(p * 2^24) << 8  (both sides exact).  Used by the GEMM epilogues and the attention kernels.
constexpr uint32_t CTTS_DROP_G = 0x9E3779B1U;
__device__ __forceinline__ uint32_t ctts_drop_threshold(float p) { return ((uint32_t)ceilf(p * 16777216.0f)) << 8; }
__device__ __forceinline__ float ctts_drop_scale_pre(uint32_t pre, uint32_t thr, float inv_keep) {
  return ctts_mix32(pre) >= thr ? inv_keep : 0.0f;
}

// ---------------------------------------------------------------- activations
// erf for the GELU epilogues (F.gelu, transformer_fs2.py:228): Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 + fp32 rounding of 6 fused
// multiply-adds - 2 ulp of a result near 1, and 3e-7 * |v| / 2 in GELU(v): three orders of magnitude inside the 1e-3 mel tolerance and
// below the fp32 noise of the GEMM in front of it.  13 VALU instructions (one v_rcp, one v_exp) against ~35 with branches for the
// correctly rounded library erff: the epilogue of the dominant FFN convolution runs under the other workgroup's MFMAs and still costs
// 6 % of the launch.  CTTS_EXACT_ERF restores erff.
__device__ __forceinline__ float ctts_erf(float x) {
#ifdef CTTS_EXACT_ERF
  return erff(x);
#else
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
#endif
}
__device__ __forceinline__ float ctts_act(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return 0.5f * v * (1.0f + ctts_erf(v * 0.70710678118654752440f));
    case 3: return tanhf(v);
    case 4: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));     // swish = v * sigmoid(v); v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division
    default: return v;
  }
}
// derivative w.r.t. the pre-activation z
__device__ __forceinline__ float ctts_act_grad(float z, int act) {
  switch (act) {
    case 1: return z > 0.f ? 1.f : 0.f;
    case 2: {
      float cdf = 0.5f * (1.0f + ctts_erf(z * 0.70710678118654752440f));
      float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
      return cdf + z * pdf;
    }
    case 3: { float t = tanhf(z); return 1.f - t * t; }
    case 4: { float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z)); return sg * (1.f + z * (1.f - sg)); }
    default: return 1.f;
  }
}

__device__ __forceinline__ float ctts_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ctts_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
