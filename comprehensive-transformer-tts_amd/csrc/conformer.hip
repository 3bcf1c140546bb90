// Conformer-specific kernels: GLU, depthwise Conv1d (k = 31), relative-position attention softmax with the
// Transformer-XL shift.  All HBM/L2-bound streaming kernels; the GEMM-shaped parts of the block go through ctts_gemm.
#include "ctts_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

__global__ void glu_fwd_kernel(const float4* __restrict__ a, float4* __restrict__ out, long rows, int C4) {
  const long total = rows * C4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / C4; const int c = (int)(e - r * C4);
    const float4 x = a[r * 2 * C4 + c], g = a[r * 2 * C4 + C4 + c];
    out[e] = make_float4(x.x * sigmoidf_(g.x), x.y * sigmoidf_(g.y), x.z * sigmoidf_(g.z), x.w * sigmoidf_(g.w));
  }
}

__global__ void glu_bwd_kernel(const float4* __restrict__ a, const float4* __restrict__ dout, float4* __restrict__ da, long rows,
                               int C4) {
  const long total = rows * C4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / C4; const int c = (int)(e - r * C4);
    const float4 x = a[r * 2 * C4 + c], g = a[r * 2 * C4 + C4 + c], d = dout[e];
    const float s0 = sigmoidf_(g.x), s1 = sigmoidf_(g.y), s2 = sigmoidf_(g.z), s3 = sigmoidf_(g.w);
    da[r * 2 * C4 + c] = make_float4(d.x * s0, d.y * s1, d.z * s2, d.w * s3);
    da[r * 2 * C4 + C4 + c] = make_float4(d.x * x.x * s0 * (1.f - s0), d.y * x.y * s1 * (1.f - s1), d.z * x.z * s2 * (1.f - s2),
                                          d.w * x.w * s3 * (1.f - s3));
  }
}

// depthwise conv: block = (C/4 lanes) x 4 time groups; each thread: 4 channels x DW_TT consecutive time steps, weights [K][C] in LDS.
// The DW_TT + K - 1 input rows a thread needs are fetched up front into registers (unconditional loads from clamped rows, masked
// afterwards: nothing in the tap loop waits on memory), then one LDS weight read per tap feeds DW_TT FMAs on statically indexed rows.
constexpr int DW_TT = 8;
constexpr int DW_MAXTAPS = 32;
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wT,
                                                          float* __restrict__ y, int T, int C, int K, int flip) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];   // [K][C]
  const int C4 = C >> 2;
  {
    const float4* w4 = reinterpret_cast<const float4*>(wT);
    float4* s4 = reinterpret_cast<float4*>(s_w);
    for (int e = threadIdx.x; e < K * C4; e += blockDim.x) {
      const int k = e / C4, c = e - k * C4;
      s4[e] = w4[(flip ? (K - 1 - k) : k) * C4 + c];
    }
  }
  __syncthreads();
  const int pad = (K - 1) / 2;
  const int lanes = blockDim.x;                 // threads: [tg][c4] with c4 fastest
  const int groups_per_block = lanes / C4;
  const int b = blockIdx.y;
  const int tg = threadIdx.x / C4, c4 = threadIdx.x - tg * C4;
  const int t0 = (blockIdx.x * groups_per_block + tg) * DW_TT;
  if (tg >= groups_per_block || t0 >= T) return;
  const float4* xb = reinterpret_cast<const float4*>(x + (long)b * T * C) + c4;
  float4 xr[DW_TT + DW_MAXTAPS - 1];            // input rows t0 - pad .. t0 + DW_TT - 1 + (K - 1 - pad)
#pragma unroll
  for (int i = 0; i < DW_TT + DW_MAXTAPS - 1; ++i) {
    const int u = t0 - pad + i;
    const bool ok = i < DW_TT + K - 1 && u >= 0 && u < T;
    const float4 v = xb[(long)min(max(u, 0), T - 1) * C4];
    xr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 acc[DW_TT];
#pragma unroll
  for (int j = 0; j < DW_TT; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < DW_MAXTAPS; ++k) {
    if (k < K) {                                  // uniform
      const float4 w = *reinterpret_cast<const float4*>(s_w + k * C + c4 * 4);
#pragma unroll
      for (int j = 0; j < DW_TT; ++j) {           // y[t0+j] += w[k] * x[t0+j + k - pad]
        acc[j].x += w.x * xr[j + k].x; acc[j].y += w.y * xr[j + k].y; acc[j].z += w.z * xr[j + k].z; acc[j].w += w.w * xr[j + k].w;
      }
    }
  }
  float4* yb = reinterpret_cast<float4*>(y + (long)b * T * C) + c4;
#pragma unroll
  for (int j = 0; j < DW_TT; ++j)
    if (t0 + j < T) yb[(long)(t0 + j) * C4] = acc[j];
}

// dw[c,k] = sum_{b,t} dy[b,t,c] x[b,t+k-pad,c].  Thread = channel; a workgroup walks a SLICE of (utterance, 32-row chunk) pairs and keeps
// its 31 tap sums in registers: per chunk the 32 dy values and the 32+K-1 x values it touches are loaded ONCE (statically indexed
// after unrolling), then 32 x K FMAs.  Slice sums go to partials[slice][c][32]; a second one-block pass adds the slices in a fixed
// order - deterministic, and no atomics (the previous version issued 31 atomics per thread per chunk: 4 M per call, 330 us).
constexpr int DW_MAXK = 32;
constexpr int DW_CH = 32;
constexpr int DW_SLICES = 128;
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ partials, int B, int T, int C, int K) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int pad = (K - 1) / 2;
  const int chunks_t = (T + DW_CH - 1) / DW_CH, n_chunks = B * chunks_t;
  const int per = (n_chunks + gridDim.x - 1) / gridDim.x;
  const int c_lo = blockIdx.x * per, c_hi = min(n_chunks, c_lo + per);
  float acc[DW_MAXK];
#pragma unroll
  for (int k = 0; k < DW_MAXK; ++k) acc[k] = 0.f;
  for (int ch = c_lo; ch < c_hi; ++ch) {
    const int b = ch / chunks_t, t0 = (ch - b * chunks_t) * DW_CH;
    const float* xb = x + (long)b * T * C + c;
    const float* db = dy + (long)b * T * C + c;
    float d[DW_CH], xs[DW_CH + DW_MAXK - 1];
#pragma unroll
    for (int i = 0; i < DW_CH; ++i) d[i] = (t0 + i < T) ? db[(long)(t0 + i) * C] : 0.f;
#pragma unroll
    for (int i = 0; i < DW_CH + DW_MAXK - 1; ++i) {
      const int u = t0 + i - pad;
      xs[i] = (i < DW_CH + K - 1 && u >= 0 && u < T) ? xb[(long)u * C] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < DW_MAXK; ++k) {
      float a = acc[k];
#pragma unroll
      for (int i = 0; i < DW_CH; ++i) a = fmaf(d[i], xs[i + k], a);
      acc[k] = a;
    }
  }
  float* p = partials + ((long)blockIdx.x * C + c) * DW_MAXK;
#pragma unroll
  for (int k = 0; k < DW_MAXK; ++k) p[k] = acc[k];
}

__global__ void dwconv_wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw, int C, int K, int n_slices, int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;          // e = c * 32 + k
  if (e >= C * DW_MAXK) return;
  const int c = e / DW_MAXK, k = e - c * DW_MAXK;
  if (k >= K) return;
  // fixed association (4 interleaved chains, then a fixed tree): deterministic, and 8 loads in flight instead of one
  const long st = (long)C * DW_MAXK;
  const float* p = partials + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int sl = 0;
  for (; sl + 8 <= n_slices; sl += 8) {
    const float a0 = p[(sl + 0) * st], a1 = p[(sl + 1) * st], a2 = p[(sl + 2) * st], a3 = p[(sl + 3) * st];
    const float a4 = p[(sl + 4) * st], a5 = p[(sl + 5) * st], a6 = p[(sl + 6) * st], a7 = p[(sl + 7) * st];
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    s0 += a4; s1 += a5; s2 += a6; s3 += a7;
  }
  for (; sl < n_slices; ++sl) s0 += p[sl * st];
  const float r = (s0 + s1) + (s2 + s3);
  if (accumulate) dw[(long)c * K + k] += r; else dw[(long)c * K + k] = r;
}

// ---- relative-position scores: shifted[i,j] = padded.flat[i*T + j + T], padded = [0 | PS] rows of T+1  (conformer.py:423-431)
__device__ __forceinline__ float rel_shifted(const float* __restrict__ PSz, int i, int j, int T) {
  const int q = j + T - i;                         // in [1, 2T-1]
  if (q <= T) return PSz[(long)i * T + (q - 1)];
  const int c = q - T - 1;                         // row i+1, padded column c (0 = the zero column)
  return c == 0 ? 0.f : PSz[(long)(i + 1) * T + (c - 1)];
}

__global__ __launch_bounds__(256) void relpos_softmax_fwd_kernel(float* __restrict__ S, const float* __restrict__ PS,
                                                                  float* __restrict__ Pd, long nrows, int T, float scale,
                                                                  float p_drop, const uint64_t* seed, uint32_t drop_offset) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  for (long row = (long)blockIdx.x * 4 + wave; row < nrows; row += (long)gridDim.x * 4) {
    const long z = row / T;
    const int i = (int)(row - z * T);
    float* s = S + row * T;
    const float* PSz = PS + z * T * T;
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) {
      const float v = (s[j] + rel_shifted(PSz, i, j, T)) * scale;
      s[j] = v;
      mx = fmaxf(mx, v);
    }
    mx = ctts_wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) { const float e = __expf(s[j] - mx); s[j] = e; sum += e; }
    const float inv = 1.f / ctts_wave_sum(sum);
    for (int j = lane; j < T; j += 64) {
      const float p = s[j] * inv;
      s[j] = p;
      if (Pd) Pd[row * T + j] = do_drop ? p * ctts_drop_scale(dkey, (uint32_t)(row * T + j), p_drop, inv_keep) : p;
    }
  }
}

__global__ __launch_bounds__(256) void relpos_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dPd, long nrows,
                                                                  int T, float scale, float p_drop, const uint64_t* seed,
                                                                  uint32_t drop_offset) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  for (long row = (long)blockIdx.x * 4 + wave; row < nrows; row += (long)gridDim.x * 4) {
    const float* p = P + row * T;
    float* d = dPd + row * T;
    float dot = 0.f;
    for (int j = lane; j < T; j += 64) {
      float g = d[j];
      if (do_drop) g *= ctts_drop_scale(dkey, (uint32_t)(row * T + j), p_drop, inv_keep);
      d[j] = g;
      dot += g * p[j];
    }
    dot = ctts_wave_sum(dot);
    for (int j = lane; j < T; j += 64) d[j] = p[j] * (d[j] - dot) * scale;
  }
}

// dPS[z,r,c'] = dS_z.flat[r*(T+1) + c' + 1 - T] when that index is >= 0, else 0
__global__ void relshift_bwd_kernel(const float* __restrict__ dS, float* __restrict__ dPS, long total, int T) {
  const long TT = (long)T * T;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long z = e / TT;
    const long rc = e - z * TT;
    const int r = (int)(rc / T), c = (int)(rc - (long)r * T);
    const long n = (long)r * (T + 1) + c + 1 - T;
    dPS[e] = n >= 0 ? dS[z * TT + n] : 0.f;
  }
}

inline int grid_for(long n, int cap = 8192) { return (int)min((n + 255) / 256, (long)cap); }

}  // namespace

extern "C" int ctts_glu_fwd(const float* a, float* out, int64_t rows, int C, void* stream) {
  CTTS_REQUIRE(a && out && (C % 4) == 0 && C > 0, "ctts_glu_fwd: bad arguments (C %% 4 must be 0)");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for((long)rows * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(a), reinterpret_cast<float4*>(out), (long)rows, C / 4);
  CTTS_CHECK_LAUNCH("ctts_glu_fwd");
  return 0;
}

extern "C" int ctts_glu_bwd(const float* a, const float* dout, float* da, int64_t rows, int C, void* stream) {
  CTTS_REQUIRE(a && dout && da && (C % 4) == 0 && C > 0, "ctts_glu_bwd: bad arguments (C %% 4 must be 0)");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for((long)rows * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(dout), reinterpret_cast<float4*>(da),
                     (long)rows, C / 4);
  CTTS_CHECK_LAUNCH("ctts_glu_bwd");
  return 0;
}

extern "C" int ctts_dwconv_fwd(const float* x, const float* wT, float* y, int B, int T, int C, int K, int flip, void* stream) {
  CTTS_REQUIRE(x && wT && y && (C % 4) == 0 && C >= 4 && C <= 1024 && (K & 1) && K >= 1 && K <= DW_MAXTAPS,
               "ctts_dwconv_fwd: need C %% 4 == 0, C <= 1024, odd K <= 32");
  CTTS_REQUIRE((size_t)K * C * 4 <= 64 * 1024, "ctts_dwconv_fwd: K*C weights do not fit the LDS staging buffer");
  if (B == 0 || T == 0) return 0;
  const int C4 = C / 4;
  const int groups = max(1, 256 / C4);
  const int threads = groups * C4;
  dim3 grid((T + groups * DW_TT - 1) / (groups * DW_TT), B);
  hipLaunchKernelGGL(dwconv_fwd_kernel, grid, dim3(threads), (size_t)K * C * sizeof(float), (hipStream_t)stream, x, wT, y, T, C, K,
                     flip);
  CTTS_CHECK_LAUNCH("ctts_dwconv_fwd");
  return 0;
}

extern "C" int ctts_dwconv_wgrad(const float* dy, const float* x, float* dw, float* partials, int B, int T, int C, int K, int accumulate, void* stream) {
  CTTS_REQUIRE(dy && x && dw && partials && K <= DW_MAXK && (K & 1), "ctts_dwconv_wgrad: need odd K <= 32 and a partials workspace");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || T == 0) {
    if (ctts_zero_async(dw, sizeof(float) * (size_t)C * K, st) != 0) { ctts_set_error("ctts_dwconv_wgrad: memset failed"); return -2; }
    return 0;
  }
  const int n_chunks = B * ((T + DW_CH - 1) / DW_CH);
  const int slices = n_chunks < DW_SLICES ? n_chunks : DW_SLICES;
  dim3 grid(slices, (C + 255) / 256);
  hipLaunchKernelGGL(dwconv_wgrad_kernel, grid, dim3(256), 0, st, dy, x, partials, B, T, C, K);
  CTTS_CHECK_LAUNCH("ctts_dwconv_wgrad");
  hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel, dim3((C * DW_MAXK + 63) / 64), dim3(64), 0, st, partials, dw, C, K, slices, accumulate);
  CTTS_CHECK_LAUNCH("ctts_dwconv_wgrad(reduce)");
  return 0;
}

namespace {
// RelativeMultiHeadAttention.forward's operand preparation (conformer.py:396-407) from the packed q | k | v projection [rows, 3C]:
// qu = q + u_bias, qv = q + v_bias, kv = k | v - one pass over the projection instead of two broadcast adds and a slice copy
__global__ void relattn_split_fwd_kernel(const float4* __restrict__ qkv, const float4* __restrict__ u, const float4* __restrict__ v,
                                         float4* __restrict__ qu, float4* __restrict__ qv, float4* __restrict__ kv, long rows, int C4) {
  const long total = rows * 3 * C4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / (3 * C4); const int c = (int)(e - r * 3 * C4);
    const float4 x = qkv[e];
    if (c < C4) {
      const float4 a = u[c], b = v[c];
      qu[r * C4 + c] = make_float4(x.x + a.x, x.y + a.y, x.z + a.z, x.w + a.w);
      qv[r * C4 + c] = make_float4(x.x + b.x, x.y + b.y, x.z + b.z, x.w + b.w);
    } else {
      kv[r * 2 * C4 + (c - C4)] = x;
    }
  }
}
// its adjoint: dqkv = (dqu + dqv) | dkv   (the bias gradients are column sums of dqu / dqv: ctts_colsum)
__global__ void relattn_split_bwd_kernel(const float4* __restrict__ dqu, const float4* __restrict__ dqv, const float4* __restrict__ dkv,
                                         float4* __restrict__ dqkv, long rows, int C4) {
  const long total = rows * 3 * C4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / (3 * C4); const int c = (int)(e - r * 3 * C4);
    if (c < C4) {
      const float4 a = dqu[r * C4 + c], b = dqv[r * C4 + c];
      dqkv[e] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    } else {
      dqkv[e] = dkv[r * 2 * C4 + (c - C4)];
    }
  }
}
}  // namespace

extern "C" int ctts_relattn_split_fwd(const float* qkv, const float* u_bias, const float* v_bias, float* qu, float* qv, float* kv,
                                      int64_t rows, int C, void* stream) {
  CTTS_REQUIRE(qkv && u_bias && v_bias && qu && qv && kv && C > 0 && (C % 4) == 0 && rows >= 0, "ctts_relattn_split_fwd: bad arguments");
  if (rows == 0) return 0;
  const long total = rows * 3 * (C / 4);
  hipLaunchKernelGGL(relattn_split_fwd_kernel, dim3((unsigned)min((total + 255) / 256, 8192L)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(qkv), reinterpret_cast<const float4*>(u_bias), reinterpret_cast<const float4*>(v_bias),
                     reinterpret_cast<float4*>(qu), reinterpret_cast<float4*>(qv), reinterpret_cast<float4*>(kv), (long)rows, C / 4);
  CTTS_CHECK_LAUNCH("ctts_relattn_split_fwd");
  return 0;
}

extern "C" int ctts_relattn_split_bwd(const float* dqu, const float* dqv, const float* dkv, float* dqkv, int64_t rows, int C, void* stream) {
  CTTS_REQUIRE(dqu && dqv && dkv && dqkv && C > 0 && (C % 4) == 0 && rows >= 0, "ctts_relattn_split_bwd: bad arguments");
  if (rows == 0) return 0;
  const long total = rows * 3 * (C / 4);
  hipLaunchKernelGGL(relattn_split_bwd_kernel, dim3((unsigned)min((total + 255) / 256, 8192L)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(dqu), reinterpret_cast<const float4*>(dqv), reinterpret_cast<const float4*>(dkv),
                     reinterpret_cast<float4*>(dqkv), (long)rows, C / 4);
  CTTS_CHECK_LAUNCH("ctts_relattn_split_bwd");
  return 0;
}

extern "C" int ctts_relpos_softmax_fwd(float* S, const float* PS, float* Pd, int nbatch, int T, float scale, float p_drop,
                                       const uint64_t* seed, uint32_t drop_offset, void* stream) {
  CTTS_REQUIRE(S && PS && nbatch > 0 && T > 0, "ctts_relpos_softmax_fwd: bad arguments");
  CTTS_REQUIRE((long)nbatch * T * T < 0xFFFFFFFFL, "ctts_relpos_softmax_fwd: tensor too large for the 32-bit dropout index");
  const long nrows = (long)nbatch * T;
  hipLaunchKernelGGL(relpos_softmax_fwd_kernel, dim3((int)min((nrows + 3) / 4, (long)16384)), dim3(256), 0, (hipStream_t)stream, S,
                     PS, Pd, nrows, T, scale, p_drop, seed, drop_offset);
  CTTS_CHECK_LAUNCH("ctts_relpos_softmax_fwd");
  return 0;
}

extern "C" int ctts_relpos_softmax_bwd(const float* P, float* dPd, int nbatch, int T, float scale, float p_drop,
                                       const uint64_t* seed, uint32_t drop_offset, void* stream) {
  CTTS_REQUIRE(P && dPd && nbatch > 0 && T > 0, "ctts_relpos_softmax_bwd: bad arguments");
  const long nrows = (long)nbatch * T;
  hipLaunchKernelGGL(relpos_softmax_bwd_kernel, dim3((int)min((nrows + 3) / 4, (long)16384)), dim3(256), 0, (hipStream_t)stream, P,
                     dPd, nrows, T, scale, p_drop, seed, drop_offset);
  CTTS_CHECK_LAUNCH("ctts_relpos_softmax_bwd");
  return 0;
}

extern "C" int ctts_relshift_bwd(const float* dS, float* dPS, int nbatch, int T, void* stream) {
  CTTS_REQUIRE(dS && dPS && nbatch > 0 && T > 0, "ctts_relshift_bwd: bad arguments");
  const long total = (long)nbatch * T * T;
  hipLaunchKernelGGL(relshift_bwd_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, dS, dPS, total, T);
  CTTS_CHECK_LAUNCH("ctts_relshift_bwd");
  return 0;
}
