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

// ---------------------------------------------------------------- bit-reproducible cross-workgroup reductions (round 4)
// No floating-point atomics anywhere on the gradient path: the order in which workgroups reach an atomicAdd changes from run to run, so
// two runs of the same train step from the same seed differed in the 7th digit and drifted apart (Adam amplifies: profiles/
// r04_diag_determinism_*_before.txt).  Instead every workgroup WRITES its partial into the caller's workspace and takes a ticket; the
// workgroup that draws the last ticket of its group sums the partials in INDEX order (never in arrival order) - two levels (groups of G
// partials, then the group sums) so that the serial part stays ~2 x G loads deep.  The partials travel by agent-scope relaxed atomic
// stores / loads (sc1: coherent across the 8 XCD L2s without a cache-wide write-back), the ticket is an agent-scope fetch-add after
// `s_waitcnt vmcnt(0)` + barrier - the hand-off of gemm_sk.hip.  Tickets return to zero: the workspace is zero-filled once, by the caller.
//
// Workspace layout (bytes from the base; include/ctts.h ctts_workspace_bytes()): launches that share a workspace must be stream-ordered.
constexpr size_t CTTS_WS_SK_FLAGS = 0;                              // stream-K flags + error word (gemm_sk.hip): 4096 words
constexpr size_t CTTS_WS_GEMM_TICKETS = 16384;                      // split-K tickets, one per (batch, tile): 65536 words
constexpr int    CTTS_WS_GEMM_TICKET_WORDS = 65536;
constexpr size_t CTTS_WS_RED_T1 = CTTS_WS_GEMM_TICKETS + 4 * (size_t)CTTS_WS_GEMM_TICKET_WORDS;       // level-1 tickets [1024 column blocks][64 groups]
constexpr size_t CTTS_WS_RED_T2 = CTTS_WS_RED_T1 + 4 * 65536;       // level-2 tickets [1024]
constexpr size_t CTTS_WS_RED_P1 = CTTS_WS_RED_T2 + 4 * 1024;        // level-1 partials: 4 MiB
constexpr size_t CTTS_WS_RED_P1_BYTES = 4u << 20;
constexpr size_t CTTS_WS_RED_P2 = CTTS_WS_RED_P1 + CTTS_WS_RED_P1_BYTES;   // group sums: 1 MiB
constexpr size_t CTTS_WS_RED_P2_BYTES = 1u << 20;
constexpr size_t CTTS_WS_SLABS = CTTS_WS_RED_P2 + CTTS_WS_RED_P2_BYTES;    // GEMM slabs (stream-K hand-off, split-K partial tiles)
constexpr size_t CTTS_WS_SLAB_FLOATS = (size_t)32 << 20;            // 128 MiB
constexpr size_t CTTS_WS_BYTES = CTTS_WS_SLABS + 4 * CTTS_WS_SLAB_FLOATS;
constexpr int CTTS_RED_MAX_COLBLOCKS = 1024, CTTS_RED_MAX_GROUPS = 64;

template <typename T> __device__ __forceinline__ void ctts_st_agent(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ T ctts_ld_agent(const T* p) {
  return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All threads of the workgroup call this after their partial stores (agent-scope / write-through).  True in the workgroup that draws
// ticket count-1 (it resets the ticket); that workgroup may then read every partial published before the others' tickets.
__device__ __forceinline__ bool ctts_arrive_last(unsigned* ticket, unsigned count) {
  __shared__ unsigned s_ctts_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // The ticket is RELAXED on purpose (ADVICE r04 asked for a release here).  What a release would add on gfx950 is `buffer_wbl2 sc1` +
    // `s_waitcnt vmcnt(0)`: a write-back of the XCD's L2.  The partials above do not live in L2 - they were stored with sc1 (write-through,
    // agent scope: ctts_st_agent) and every wave has drained them (`s_waitcnt vmcnt(0)`) before the barrier in front of this line, which
    // is exactly the state a release would establish; the last workgroup pairs it with an acquire fence (L1 invalidate) and sc1 loads.
    // Measured with __ATOMIC_RELEASE on this fetch_add (round 5, same box, driver-style bench): fs2 19.33 -> 19.67 ms, C5 30.2 -> 30.8 ms,
    // fp32-only fs2 23.19 -> 23.43 ms - ~1.7 us of L2 write-back per workgroup of ~150 reduction launches per step, for no change in
    // the bits (tests/test_determinism_gpu.py).  The bounded-spin / error-word hand-off of the GEMMs follows the same scheme.
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == count - 1;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ctts_last = last ? 1u : 0u;
  }
  __syncthreads();
  const bool last = s_ctts_last != 0u;
  if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return last;
}

// Host side: groups of the two-level scheme for n partials (G = 16 up to 1024 partials)
static inline int ctts_red_group(int n) { int g = 16; while ((n + g - 1) / g > CTTS_RED_MAX_GROUPS) g *= 2; return g; }

// Ordered sum over the `nstripes` workgroups (blockIdx.y) of one 64-column block (blockIdx.x = cb) of 256 threads: thread l < 64 brings NV
// values v[k] (its column's partial); returns true in exactly one workgroup per column block, where threads l < 64 then hold the totals.
// Summation order: members of a group ty, ty+4, ... (ty = 0..3) in index order, the four ty-sums left to right, the groups likewise.
template <typename T, int NV>
__device__ __forceinline__ bool ctts_ordered_colsum(T (&v)[NV], unsigned char* ws, int cb, int stripe, int nstripes, int G) {
  if (nstripes <= 1) return true;
  __shared__ T s_ctts_red[4][NV * 64];
  const int l = threadIdx.x & 63, ty = threadIdx.x >> 6;
  T* p1 = reinterpret_cast<T*>(ws + CTTS_WS_RED_P1);
  T* p2 = reinterpret_cast<T*>(ws + CTTS_WS_RED_P2);
  unsigned* t1 = reinterpret_cast<unsigned*>(ws + CTTS_WS_RED_T1);
  unsigned* t2 = reinterpret_cast<unsigned*>(ws + CTTS_WS_RED_T2);
  const int ngroups = (nstripes + G - 1) / G, g = stripe / G;
  const int g0 = g * G, gn = min(G, nstripes - g0);
  if (ty == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) ctts_st_agent(p1 + ((long)(cb * nstripes + stripe) * NV + k) * 64 + l, v[k]);
  }
  if (!ctts_arrive_last(t1 + cb * CTTS_RED_MAX_GROUPS + g, (unsigned)gn)) return false;
  auto gather = [&](const T* base, int first, int n) {      // base[(first + i) * NV * 64 + k * 64 + l], i < n
    T a[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) a[k] = (T)0;
    for (int i0 = ty; i0 < n; i0 += 16) {            // four members in flight per thread (the loads are independent), added in index order
      T x[4][NV];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 4 * u;
#pragma unroll
        for (int k = 0; k < NV; ++k) x[u][k] = i < n ? ctts_ld_agent(base + ((long)(first + i) * NV + k) * 64 + l) : (T)0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < NV; ++k) a[k] += x[u][k];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) s_ctts_red[ty][k * 64 + l] = a[k];
    __syncthreads();
    if (ty == 0) {
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] = ((s_ctts_red[0][k * 64 + l] + s_ctts_red[1][k * 64 + l]) + s_ctts_red[2][k * 64 + l]) + s_ctts_red[3][k * 64 + l];
    }
    __syncthreads();
  };
  gather(p1 + (long)cb * nstripes * NV * 64, g0, gn);
  if (ngroups == 1) return true;
  if (ty == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) ctts_st_agent(p2 + ((long)(cb * ngroups + g) * NV + k) * 64 + l, v[k]);
  }
  if (!ctts_arrive_last(t2 + cb, (unsigned)ngroups)) return false;
  gather(p2 + (long)cb * ngroups * NV * 64, 0, ngroups);
  return true;
}
