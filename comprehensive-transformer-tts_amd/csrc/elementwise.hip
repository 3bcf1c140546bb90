// Backward-pass elementwise helpers, bias-gradient column sums, Conv1d weight repacks and the
// small kernels of the mel front end.  All HBM-bound streaming kernels (float4 where the shape allows).
#include "ctts_common.h"

namespace {
__global__ void zero_u32_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace

int ctts_zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  const size_t n = (bytes + 3) / 4;
  const unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(zero_u32_kernel, dim3(blocks), dim3(256), 0, st, (uint32_t*)p, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

namespace {

__global__ void act_dropout_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ z, float* __restrict__ dz,
                                       long total, int act, float p_drop, const uint64_t* seed, uint32_t drop_offset) {
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  if (total & 3) {  // unaligned shapes (N = 1, 2, 11 heads): scalar path
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      float g = dg[i];
      if (do_drop) g *= ctts_drop_scale(dkey, (uint32_t)i, p_drop, inv_keep);
      dz[i] = g * ctts_act_grad(z[i], act);
    }
    return;
  }
  const long n4 = total >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 g = reinterpret_cast<const float4*>(dg)[i];
    const float4 zz = reinterpret_cast<const float4*>(z)[i];
    float ga[4] = {g.x, g.y, g.z, g.w};
    const float za[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (do_drop) ga[e] *= ctts_drop_scale(dkey, (uint32_t)(i * 4 + e), p_drop, inv_keep);
      ga[e] *= ctts_act_grad(za[e], act);
    }
    reinterpret_cast<float4*>(dz)[i] = make_float4(ga[0], ga[1], ga[2], ga[3]);
  }
}

__global__ void rowscale_dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int C,
                                        const float* __restrict__ rowscale, float p_drop, const uint64_t* seed,
                                        uint32_t drop_offset) {
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  if (C & 3) {  // unaligned shapes: scalar path
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      float a = x[i];
      if (do_drop) a *= ctts_drop_scale(dkey, (uint32_t)i, p_drop, inv_keep);
      y[i] = a * (rowscale ? rowscale[i / C] : 1.f);
    }
    return;
  }
  const long n4 = total >> 2;
  const int C4 = C >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float a[4] = {v.x, v.y, v.z, v.w};
    const float sc = rowscale ? rowscale[i / C4] : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (do_drop) a[e] *= ctts_drop_scale(dkey, (uint32_t)(i * 4 + e), p_drop, inv_keep);
      a[e] *= sc;
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(a[0], a[1], a[2], a[3]);
  }
}

// Final write of an ordered column sum (the one workgroup ctts_ordered_colsum elects per column block): out[c] (+)= scale * total.  A folded
// view (C = k * creal <= 64, one column block) first adds its k copies of a channel in index order through LDS.
__device__ __forceinline__ void ctts_colsum_emit(float tot, float* s64, float* __restrict__ out, int c, int C, int creal, float scale,
                                                 int accumulate) {
  const int ty = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (C != creal) {                   // uniform
    __syncthreads();
    if (ty == 0) s64[l] = c < C ? tot : 0.f;
    __syncthreads();
    if (ty == 0 && l < creal) {
      float a = 0.f;
      for (int j = l; j < C; j += creal) a += s64[j];
      out[l] = accumulate ? out[l] + scale * a : scale * a;
    }
    return;
  }
  if (ty == 0 && c < C) out[c] = accumulate ? out[c] + scale * tot : scale * tot;
}

// C = row width of the (possibly folded) view; a dense narrow matrix [R, creal] (creal < 64) is read as [R/k, k*creal] so that all
// 64 lanes carry data; view column c accumulates into channel c % creal.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long rows, int C,
                                                      int creal, long ld, float scale, unsigned char* ws, int G, int accumulate,
                                                      float* __restrict__ parts) {
  __shared__ float s[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  const long stripe = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = blockIdx.y * stripe, r1 = min(rows, r0 + stripe);
  float a = 0.f;
  if (c < C) {
#pragma unroll 4
    for (long r = r0 + ty; r < r1; r += 4) a += x[r * ld + c];
  }
  s[ty][threadIdx.x & 63] = a;
  __syncthreads();
  float tot[1] = {0.f};
  if (ty == 0) { const int l = threadIdx.x; tot[0] = s[0][l] + s[1][l] + s[2][l] + s[3][l]; }
  if (parts) {                 // deferred: this stripe's partial row; ctts_partial_sums adds the stripes in order later
    if (ty == 0 && c < C) parts[(long)blockIdx.y * C + c] = tot[0];
    return;
  }
  if (!ctts_ordered_colsum<float, 1>(tot, ws, blockIdx.x, blockIdx.y, gridDim.y, G)) return;
  ctts_colsum_emit(tot[0], s[0], out, c, C, creal, scale, accumulate);
}

// Backward of the GEMM epilogue y = rowscale * (R + drop(act(alpha * (acc + bias)))) in ONE pass over dY:
//   gm = dY * rowscale (the residual's gradient, optional output), dZ = gm * drop * act'(Z), dbias (+)= bias_scale * colsum(dZ).
// Replaces the rowscale_dropout -> act_dropout_bwd -> colsum chain (three reads of a [rows, C] tensor, two intermediate writes).
__global__ __launch_bounds__(256) void epilogue_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ rowscale,
                                                            const float* __restrict__ z, float* __restrict__ dz, float* __restrict__ gm,
                                                            float* __restrict__ dbias, long rows, int C, int act, float p_drop,
                                                            const uint64_t* seed, uint32_t drop_offset, float bias_scale,
                                                            unsigned char* ws, int G, int accumulate, float* __restrict__ parts) {
  __shared__ float s[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  const long stripe = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = blockIdx.y * stripe, r1 = min(rows, r0 + stripe);
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  float a = 0.f;
  if (c < C) {
#pragma unroll 4
    for (long r = r0 + ty; r < r1; r += 4) {
      const long e = r * C + c;
      float g = dy[e];
      if (rowscale) g *= rowscale[r];
      if (gm) gm[e] = g;
      if (do_drop) g *= ctts_drop_scale(dkey, (uint32_t)e, p_drop, inv_keep);
      if (z) g *= ctts_act_grad(z[e], act);
      dz[e] = g;
      a += g;
    }
  }
  if (!dbias && !parts) return;
  s[ty][threadIdx.x & 63] = a;
  __syncthreads();
  float tot[1] = {0.f};
  if (ty == 0) { const int l = threadIdx.x; tot[0] = s[0][l] + s[1][l] + s[2][l] + s[3][l]; }
  if (parts) {
    if (ty == 0 && c < C) parts[(long)blockIdx.y * C + c] = tot[0];
    return;
  }
  if (!ctts_ordered_colsum<float, 1>(tot, ws, blockIdx.x, blockIdx.y, gridDim.y, G)) return;
  ctts_colsum_emit(tot[0], s[0], dbias, c, C, C, bias_scale, accumulate);
}

// w[Cout][Cin][K] <-> GEMM-friendly layouts
__global__ void conv_weight_repack_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int k,
                                          int mode, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    if (mode == 0) {         // dst[co][kk][ci] = src[co][ci][kk]
      const int ci = (int)(e % cin); const long t = e / cin; const int kk = (int)(t % k); const int co = (int)(t / k);
      dst[e] = src[((long)co * cin + ci) * k + kk];
    } else if (mode == 1) {  // dst[ci][kk][co] = src[co][ci][k-1-kk]
      const int co = (int)(e % cout); const long t = e / cout; const int kk = (int)(t % k); const int ci = (int)(t / k);
      dst[e] = src[((long)co * cin + ci) * k + (k - 1 - kk)];
    } else if (mode == 4) {  // dst[ci][kk][co] = src[co][k-1-kk][ci]      (src already in the GEMM-major forward layout)
      const int co = (int)(e % cout); const long t = e / cout; const int kk = (int)(t % k); const int ci = (int)(t / k);
      dst[e] = src[((long)co * k + (k - 1 - kk)) * cin + ci];
    } else {                 // dst[co][ci][kk] (+)= src[co][kk][ci]      (mode 3 accumulates)
      const int kk = (int)(e % k); const long t = e / k; const int ci = (int)(t % cin); const int co = (int)(t / cin);
      const float v = src[((long)co * k + kk) * cin + ci];
      dst[e] = mode == 3 ? dst[e] + v : v;
    }
  }
}

__global__ void reflect_pad_kernel(const float* __restrict__ y, float* __restrict__ ypad, int N, int pad, long ld_out,
                                   long total) {
  const int W = N + 2 * pad;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / ld_out;
    const int j = (int)(e - b * ld_out);
    float v = 0.f;
    if (j < W) {
      int i = j - pad;
      if (i < 0) i = -i;
      if (i >= N) i = 2 * (N - 1) - i;
      v = y[b * N + i];
    }
    ypad[e] = v;
  }
}

// reim [frames, 2*nbins] (cols [0,nbins) real, [nbins,2nbins) imag) -> mag [frames, ld_mag], energy[frames]
__global__ __launch_bounds__(256) void stft_magnitude_kernel(const float* __restrict__ reim, long ld_reim,
                                                              float* __restrict__ mag, long ld_mag,
                                                              float* __restrict__ energy, long frames, int nbins) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long f = (long)blockIdx.x * 4 + wave; f < frames; f += (long)gridDim.x * 4) {
    const float* r = reim + f * ld_reim;
    float acc = 0.f;
    for (int k = lane; k < ld_mag; k += 64) {
      float m = 0.f;
      if (k < nbins) {
        const float re = r[k], im = r[nbins + k];
        const float p = re * re + im * im;
        acc += p;
        m = sqrtf(p);
      }
      mag[f * ld_mag + k] = m;
    }
    acc = ctts_wave_sum(acc);
    if (lane == 0) energy[f] = sqrtf(acc);
  }
}

// mel_fm [B*F, n_mel] -> out [B, n_mel, F] = log(max(v, clip))
__global__ void log_clamp_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int F, int n_mel,
                                           float clip, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int f = (int)(e % F); const long t = e / F; const int m = (int)(t % n_mel); const long b = t / n_mel;
    out[e] = logf(fmaxf(in[(b * F + f) * n_mel + m], clip));
  }
}

inline int grid_for(long n, int cap = 4096) { return (int)min((n + 255) / 256, (long)cap); }

}  // namespace

extern "C" int ctts_act_dropout_bwd(const float* dg, const float* z, float* dz, int64_t rows, int C, int act,
                                    float alpha_unused, float p_drop, const uint64_t* seed, uint32_t drop_offset,
                                    void* stream) {
  (void)alpha_unused;
  CTTS_REQUIRE(dg && z && dz && C > 0, "ctts_act_dropout_bwd: bad arguments");
  const long total = (long)rows * C;
  if (total == 0) return 0;
  hipLaunchKernelGGL(act_dropout_bwd_kernel, dim3(grid_for((total & 3) ? total : (total >> 2))), dim3(256), 0, (hipStream_t)stream, dg, z, dz, total,
                     act, p_drop, seed, drop_offset);
  CTTS_CHECK_LAUNCH("ctts_act_dropout_bwd");
  return 0;
}

extern "C" int ctts_rowscale_dropout(const float* x, float* y, int64_t rows, int C, const float* rowscale, float p_drop,
                                     const uint64_t* seed, uint32_t drop_offset, void* stream) {
  CTTS_REQUIRE(x && y && C > 0, "ctts_rowscale_dropout: bad arguments");
  const long total = (long)rows * C;
  if (total == 0) return 0;
  hipLaunchKernelGGL(rowscale_dropout_kernel, dim3(grid_for((C & 3) ? total : (total >> 2))), dim3(256), 0, (hipStream_t)stream, x, y, total, C,
                     rowscale, p_drop, seed, drop_offset);
  CTTS_CHECK_LAUNCH("ctts_rowscale_dropout");
  return 0;
}

namespace {
// out[c] (+)= scale * sum_r w[r] * x[r,c]: weight gradient of a one-output Linear (N = 1 heads of the duration / energy predictors)
__global__ __launch_bounds__(256) void weighted_colsum_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ out, long rows, int C, float scale,
                                                               unsigned char* ws, int G, int accumulate, float* __restrict__ parts) {
  __shared__ float s[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  const long stripe = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = blockIdx.y * stripe, r1 = min(rows, r0 + stripe);
  float a = 0.f;
  if (c < C) {
#pragma unroll 4
    for (long r = r0 + ty; r < r1; r += 4) a = fmaf(w[r], x[r * C + c], a);
  }
  s[ty][threadIdx.x & 63] = a;
  __syncthreads();
  float tot[1] = {0.f};
  if (ty == 0) { const int l = threadIdx.x; tot[0] = s[0][l] + s[1][l] + s[2][l] + s[3][l]; }
  if (parts) {
    if (ty == 0 && c < C) parts[(long)blockIdx.y * C + c] = tot[0];
    return;
  }
  if (!ctts_ordered_colsum<float, 1>(tot, ws, blockIdx.x, blockIdx.y, gridDim.y, G)) return;
  ctts_colsum_emit(tot[0], s[0], out, c, C, C, scale, accumulate);
}
}  // namespace

// stripes (blockIdx.y) of a column-sum launch with gx column blocks: the workspace holds one 64-float partial per workgroup
static int colsum_stripes(int gx, long want, bool have_ws) {
  if (!have_ws) return 1;
  const long cap = (long)(CTTS_WS_RED_P1_BYTES / (64 * sizeof(float))) / gx;
  return (int)max((long)1, min(want, min(cap, (long)CTTS_RED_MAX_GROUPS * 64)));
}

int ctts_layernorm_bwd_blocks(int rows, int C);      // norm.hip

// Deferred mode of the column-sum entry points (`parts` != NULL): every stripe writes its partial row [C] to parts[stripe][C] and nothing
// else happens - no tickets, no tail; the caller adds the stripes in order later, many reductions in one launch (ctts_partial_sums).
// ctts_reduce_parts(kind, rows, C) = the number of partial rows the launch will write (0: this shape has no deferred mode).
//   kind 0: ctts_colsum   1: ctts_weighted_colsum   2: ctts_epilogue_bwd (bias gradient)   3: ctts_layernorm_bwd (rows of 2 C: dgamma | dbeta)
extern "C" int ctts_reduce_parts(int kind, int64_t rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  const int gx = (C + 63) / 64;
  if (gx > CTTS_RED_MAX_COLBLOCKS) return 0;
  switch (kind) {
    case 0: if (C * 2 <= 64 && rows % 2 == 0) return 0;           // narrow matrices are folded: immediate mode only
            return colsum_stripes(gx, min((long)max(1, 1024 / gx), (long)rows / 64), true);
    case 1: return colsum_stripes(gx, min((long)max(1, 1024 / gx), (long)rows / 64), true);
    case 2: return colsum_stripes(gx, min((long)max(1, 2048 / gx), (long)rows / 32), true);
    case 3: return ctts_layernorm_bwd_blocks((int)rows, C);
    default: return 0;
  }
}

extern "C" int ctts_weighted_colsum(const float* x, const float* w, float* out, int64_t rows, int C, float scale, int accumulate,
                                    void* ws, float* parts, void* stream) {
  CTTS_REQUIRE(x && w && (out || parts) && C > 0 && (C + 63) / 64 <= CTTS_RED_MAX_COLBLOCKS, "ctts_weighted_colsum: bad arguments");
  CTTS_REQUIRE(!parts || rows > 0, "ctts_weighted_colsum: deferred mode needs rows > 0");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (!accumulate && ctts_zero_async(out, sizeof(float) * C, st) != 0) { ctts_set_error("ctts_weighted_colsum: zero fill failed"); return -2; }
    return 0;
  }
  const int gx = (C + 63) / 64;
  const int gy = colsum_stripes(gx, min((long)max(1, 1024 / gx), (long)rows / 64), ws != nullptr || parts != nullptr);
  hipLaunchKernelGGL(weighted_colsum_kernel, dim3(gx, gy), dim3(256), 0, st, x, w, out, (long)rows, C, scale, (unsigned char*)ws,
                     ctts_red_group(gy), accumulate, parts);
  CTTS_CHECK_LAUNCH("ctts_weighted_colsum");
  return 0;
}

extern "C" int ctts_colsum(const float* x, float* out, int64_t rows, int C, int64_t ld, float scale, int accumulate, void* ws,
                           float* parts, void* stream) {
  CTTS_REQUIRE(x && (out || parts) && C > 0 && (C + 63) / 64 <= CTTS_RED_MAX_COLBLOCKS, "ctts_colsum: bad arguments");
  CTTS_REQUIRE(!parts || ctts_reduce_parts(0, rows, C) > 0, "ctts_colsum: this shape has no deferred mode (ctts_reduce_parts = 0)");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (!accumulate && ctts_zero_async(out, sizeof(float) * C, st) != 0) { ctts_set_error("ctts_colsum: zero fill failed"); return -2; }
    return 0;
  }
  int k = 1;                                   // fold narrow dense matrices so that a wave reads 64 useful floats per row
  if (ld == C && !parts)
    while (C * k * 2 <= 64 && rows % (k * 2) == 0) k *= 2;
  const int Cv = C * k, gx = (Cv + 63) / 64;
  const long Rv = rows / k;
  const int gy = colsum_stripes(gx, min((long)max(1, 1024 / gx), Rv / 64), ws != nullptr || parts != nullptr);
  hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, st, x, out, Rv, Cv, C, (long)ld * k, scale, (unsigned char*)ws,
                     ctts_red_group(gy), accumulate, parts);
  CTTS_CHECK_LAUNCH("ctts_colsum");
  return 0;
}

extern "C" int ctts_epilogue_bwd(const float* dy, const float* rowscale, const float* z, float* dz, float* gm, float* dbias,
                                 int64_t rows, int C, int act, float p_drop, const uint64_t* seed, uint32_t drop_offset, float bias_scale,
                                 int accumulate_bias, void* ws, float* parts, void* stream) {
  CTTS_REQUIRE(dy && dz && rows >= 0 && C > 0 && p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed), "ctts_epilogue_bwd: bad arguments");
  CTTS_REQUIRE(!dbias || (C + 63) / 64 <= CTTS_RED_MAX_COLBLOCKS, "ctts_epilogue_bwd: C too large for the bias-gradient reduction");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (dbias && !accumulate_bias && ctts_zero_async(dbias, sizeof(float) * C, st) != 0) { ctts_set_error("ctts_epilogue_bwd: zero fill failed"); return -2; }
    return 0;
  }
  const int gx = (C + 63) / 64;
  const long want = min((long)max(1, 2048 / gx), (long)rows / 32);
  CTTS_REQUIRE(!parts || rows > 0, "ctts_epilogue_bwd: deferred bias gradient needs rows > 0");
  const int gy = (dbias || parts) ? colsum_stripes(gx, want, ws != nullptr || parts != nullptr) : (int)max((long)1, want);
  hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(gx, gy), dim3(256), 0, st, dy, rowscale, act ? z : nullptr, dz, gm, dbias, (long)rows, C, act,
                     p_drop, seed, drop_offset, bias_scale, (unsigned char*)ws, ctts_red_group(gy), accumulate_bias, parts);
  CTTS_CHECK_LAUNCH("ctts_epilogue_bwd");
  return 0;
}

extern "C" int ctts_conv_weight_repack(const float* src, float* dst, int cout, int cin, int k, int mode, void* stream) {
  CTTS_REQUIRE(src && dst && mode >= 0 && mode <= 4, "ctts_conv_weight_repack: bad arguments");
  const long total = (long)cout * cin * k;
  if (total == 0) return 0;
  hipLaunchKernelGGL(conv_weight_repack_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, cout, cin, k,
                     mode, total);
  CTTS_CHECK_LAUNCH("ctts_conv_weight_repack");
  return 0;
}

extern "C" int ctts_reflect_pad(const float* y, float* ypad, int B, int N, int pad, int64_t ld_out, void* stream) {
  CTTS_REQUIRE(y && ypad && N > pad && ld_out >= N + 2 * pad, "ctts_reflect_pad: need N > pad and ld_out >= N + 2*pad");
  const long total = (long)B * ld_out;
  if (total == 0) return 0;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, ypad, N, pad, (long)ld_out, total);
  CTTS_CHECK_LAUNCH("ctts_reflect_pad");
  return 0;
}

extern "C" int ctts_stft_magnitude(const float* reim, int64_t ld_reim, float* mag, int64_t ld_mag, float* energy,
                                   int64_t frames, int nbins, void* stream) {
  CTTS_REQUIRE(reim && mag && energy && ld_mag >= nbins, "ctts_stft_magnitude: bad arguments");
  if (frames == 0) return 0;
  const int blocks = (int)min((frames + 3) / 4, (int64_t)4096);
  hipLaunchKernelGGL(stft_magnitude_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reim, (long)ld_reim, mag,
                     (long)ld_mag, energy, (long)frames, nbins);
  CTTS_CHECK_LAUNCH("ctts_stft_magnitude");
  return 0;
}

extern "C" int ctts_log_clamp_transpose(const float* mel_fm, float* out, int B, int F, int n_mel, float clip, void* stream) {
  CTTS_REQUIRE(mel_fm && out, "ctts_log_clamp_transpose: null pointer");
  const long total = (long)B * F * n_mel;
  if (total == 0) return 0;
  hipLaunchKernelGGL(log_clamp_transpose_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, mel_fm, out, F, n_mel,
                     clip, total);
  CTTS_CHECK_LAUNCH("ctts_log_clamp_transpose");
  return 0;
}


// ---------------------------------------------------------------- data-gradient weights of all Conv1d layers, one launch (round 4)
// The data gradient of a Conv1d is a convolution with the flipped, transposed taps: dst[ci][kk][co] = src[co][k-1-kk][ci] (src = the
// GEMM-major forward layout [Cout][K][Cin] of model._Conv).  Round 3 repacked each layer's weight inside its own backward - 21 launches
// of an uncoalesced gather (threads walked co with a stride of K*Cin floats: 10 us for 9.4 MB).  Now ONE launch at the start of the step
// transposes 32 x 32 (co, ci) tiles of every layer through LDS (both sides coalesced), tasks packed 32 per launch like ctts_partial_sums.
namespace {
constexpr int RPK_BATCH = 32;
struct RepackBatch {
  const float* src[RPK_BATCH];
  float* dst[RPK_BATCH];
  int cout[RPK_BATCH], cin[RPK_BATCH], k[RPK_BATCH];
  int first_block[RPK_BATCH + 1];
  int ntasks;
};

__global__ __launch_bounds__(256) void conv_dgrad_weights_kernel(const RepackBatch b) {
  __shared__ float tile[32][33];
  int t = 0;
  while (t + 1 < b.ntasks && (int)blockIdx.x >= b.first_block[t + 1]) ++t;
  const float* __restrict__ src = b.src[t];
  float* __restrict__ dst = b.dst[t];
  const int cout = b.cout[t], cin = b.cin[t], k = b.k[t];
  const int tco = (cout + 31) / 32, tci = (cin + 31) / 32;
  int blk = (int)blockIdx.x - b.first_block[t];
  const int kk = blk / (tco * tci);
  blk -= kk * tco * tci;
  const int co0 = (blk / tci) * 32, ci0 = (blk % tci) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < cout && ci < cin) ? src[((long)co * k + (k - 1 - kk)) * cin + ci] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < cin && co < cout) dst[((long)ci * k + kk) * cout + co] = tile[tx][r];
  }
}
}  // namespace

extern "C" int ctts_conv_dgrad_weights(const ctts_repack_task* tasks, int ntasks, void* stream) {
  CTTS_REQUIRE(ntasks >= 0 && (tasks || ntasks == 0), "ctts_conv_dgrad_weights: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < ntasks; t0 += RPK_BATCH) {
    RepackBatch b;
    b.ntasks = 0;
    int blocks = 0;
    for (int t = t0; t < ntasks && t < t0 + RPK_BATCH; ++t) {
      const ctts_repack_task& q = tasks[t];
      CTTS_REQUIRE(q.src && q.dst && q.cout > 0 && q.cin > 0 && q.k > 0, "ctts_conv_dgrad_weights: bad task %d", t);
      const int i = b.ntasks++;
      b.src[i] = q.src; b.dst[i] = q.dst; b.cout[i] = q.cout; b.cin[i] = q.cin; b.k[i] = q.k;
      b.first_block[i] = blocks;
      blocks += ((q.cout + 31) / 32) * ((q.cin + 31) / 32) * q.k;
    }
    if (b.ntasks == 0) continue;
    b.first_block[b.ntasks] = blocks;
    hipLaunchKernelGGL(conv_dgrad_weights_kernel, dim3(blocks), dim3(256), 0, st, b);
    CTTS_CHECK_LAUNCH("ctts_conv_dgrad_weights");
  }
  return 0;
}

// ---------------------------------------------------------------- positional embedding add (round 4)
// y = rowscale[row] * dropout(x + alpha * table[pos[row]])   -   `x + self.pos_embed_alpha * self.embed_positions(x)` followed by F.dropout
// and the non-pad mask multiply (transformer_fs2.py:41-52,113-119; PitchPredictor.forward modules.py:1349-1351) in ONE launch instead of
// int64 cast + gather + multiply + add + dropout / mask (five), and ONE in the backward (dx = mask / dropout of dy; dalpha = sum over all
// elements of dx * table[pos], ordered cross-workgroup sum) instead of six.  alpha: device scalar (the [1] parameter) or NULL (= 1).
namespace {
__global__ __launch_bounds__(256) void posembed_fwd_kernel(const float4* __restrict__ x, const int32_t* __restrict__ pos, const float4* __restrict__ table,
                                                            const float* __restrict__ alpha, const float* __restrict__ rowscale, float4* __restrict__ y,
                                                            long n4, int C4, float p_drop, const uint64_t* seed, uint32_t drop_offset) {
#pragma clang fp contract(off)          // x + alpha * pe with TWO roundings like torch's unfused sequence (HIP's __fmul_rn / __fadd_rn are plain operators)
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  const float a = alpha ? alpha[0] : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C4;
    const int c = (int)(i - row * C4);
    const float4 xv = x[i], tv = table[(long)pos[row] * C4 + c];
    float v[4] = {xv.x + a * tv.x, xv.y + a * tv.y, xv.z + a * tv.z, xv.w + a * tv.w};      // no FMA (pragma above): bit-identical to the unfused sequence
    const float sc = rowscale ? rowscale[row] : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (do_drop) v[e] *= ctts_drop_scale(dkey, (uint32_t)(i * 4 + e), p_drop, inv_keep);
      v[e] *= sc;
    }
    y[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__global__ __launch_bounds__(256) void posembed_bwd_kernel(const float4* __restrict__ dy, const int32_t* __restrict__ pos, const float4* __restrict__ table,
                                                            const float* __restrict__ rowscale, float4* __restrict__ dx, float* __restrict__ dalpha,
                                                            long n4, int C4, float p_drop, const uint64_t* seed, uint32_t drop_offset,
                                                            unsigned char* ws, int G, int accumulate) {
  __shared__ float s_w[4];
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  float dot = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C4;
    const int c = (int)(i - row * C4);
    const float4 g = dy[i];
    float v[4] = {g.x, g.y, g.z, g.w};
    const float sc = rowscale ? rowscale[row] : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] *= sc;
      if (do_drop) v[e] *= ctts_drop_scale(dkey, (uint32_t)(i * 4 + e), p_drop, inv_keep);
    }
    dx[i] = make_float4(v[0], v[1], v[2], v[3]);
    if (dalpha) {
      const float4 tv = table[(long)pos[row] * C4 + c];
      dot += v[0] * tv.x + v[1] * tv.y + v[2] * tv.z + v[3] * tv.w;
    }
  }
  if (!dalpha) return;
  dot = ctts_wave_sum(dot);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = dot;
  __syncthreads();
  float tot[1] = {0.f};
  if (threadIdx.x == 0) tot[0] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
  if (!ctts_ordered_colsum<float, 1>(tot, ws, 0, blockIdx.x, gridDim.x, G)) return;
  if (threadIdx.x == 0) dalpha[0] = accumulate ? dalpha[0] + tot[0] : tot[0];
}
}  // namespace

extern "C" int ctts_posembed_fwd(const float* x, const int32_t* pos, const float* table, const float* alpha, const float* rowscale, float* y,
                                 int64_t rows, int C, float p_drop, const uint64_t* seed, uint32_t drop_offset, void* stream) {
  CTTS_REQUIRE(x && pos && table && y && rows >= 0 && C > 0 && (C % 4) == 0 && p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed),
               "ctts_posembed_fwd: bad arguments (C %% 4 must be 0)");
  const long n4 = (long)rows * (C / 4);
  if (n4 == 0) return 0;
  hipLaunchKernelGGL(posembed_fwd_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, pos, (const float4*)table, alpha,
                     rowscale, (float4*)y, n4, C / 4, p_drop, seed, drop_offset);
  CTTS_CHECK_LAUNCH("ctts_posembed_fwd");
  return 0;
}

extern "C" int ctts_posembed_bwd(const float* dy, const int32_t* pos, const float* table, const float* rowscale, float* dx, float* dalpha,
                                 int64_t rows, int C, float p_drop, const uint64_t* seed, uint32_t drop_offset, int accumulate, void* ws,
                                 void* stream) {
  CTTS_REQUIRE(dy && dx && rows >= 0 && C > 0 && (C % 4) == 0 && p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed) &&
               (!dalpha || (pos && table)), "ctts_posembed_bwd: bad arguments (C %% 4 must be 0)");
  const long n4 = (long)rows * (C / 4);
  hipStream_t st = (hipStream_t)stream;
  if (n4 == 0) {
    if (dalpha && !accumulate && ctts_zero_async(dalpha, sizeof(float), st) != 0) return -2;
    return 0;
  }
  // the ordered sum of dalpha keeps one 64-float partial per workgroup in the workspace; without one (and dalpha wanted): one workgroup
  int grid = grid_for(n4, 1024);
  if (dalpha && !ws) grid = 1;
  hipLaunchKernelGGL(posembed_bwd_kernel, dim3(grid), dim3(256), 0, st, (const float4*)dy, pos, (const float4*)table, rowscale, (float4*)dx, dalpha,
                     n4, C / 4, p_drop, seed, drop_offset, (unsigned char*)ws, ctts_red_group(grid), accumulate);
  CTTS_CHECK_LAUNCH("ctts_posembed_bwd");
  return 0;
}

// ---------------------------------------------------------------- deferred ordered reductions, many per launch (round 4)
// dst[i] += alpha * (src[0*stride + i] + src[1*stride + i] + ... + src[(count-1)*stride + i]), partials added in index order: the second
// half of every split-K weight gradient, bias / LayerNorm column sum of a backward stage, finished by ONE launch per PSUM_BATCH tasks
// instead of one tail (or one reduce launch) per layer.  Workgroup b works on PSUM_CHUNKS x 1024 consecutive elements of the task whose
// block range contains b.
namespace {
constexpr int PSUM_BATCH = 24;
struct PsumBatch {
  const float* src[PSUM_BATCH];
  float* dst[PSUM_BATCH];
  long n[PSUM_BATCH];
  long stride[PSUM_BATCH];
  int count[PSUM_BATCH];
  float alpha[PSUM_BATCH];
  int first_block[PSUM_BATCH + 1];
  int ntasks;
};

#ifndef CTTS_PSUM_CHUNKS
#define CTTS_PSUM_CHUNKS 8
#endif
constexpr int PSUM_CHUNKS = CTTS_PSUM_CHUNKS;          // a workgroup walks PSUM_CHUNKS x 1024 consecutive elements: the partials of one output range lie `stride`
                                        // apart (a page each), so every page a workgroup opens is used for 32 KB instead of 4 KB (TLB reach; 1 / 4 / 8 chunks: fs2 23.63 / 23.58 / 23.56 ms, conformer 27.93 / 27.91 / 27.82 - within noise)
__global__ __launch_bounds__(256) void partial_sums_kernel(const PsumBatch b) {
  int t = 0;
  while (t + 1 < b.ntasks && (int)blockIdx.x >= b.first_block[t + 1]) ++t;        // uniform; <= 24 steps
  const float* src = b.src[t];
  float* dst = b.dst[t];
  const long n = b.n[t], stride = b.stride[t];
  const int count = b.count[t];
  const float alpha = b.alpha[t];
  const bool aligned = !(stride & 3) && !((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15);
  const long s4 = stride >> 2;
#pragma unroll 1
  for (int c = 0; c < PSUM_CHUNKS; ++c) {
    const long i0 = (((long)(blockIdx.x - b.first_block[t]) * PSUM_CHUNKS + c) * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (aligned && i0 + 4 <= n) {
      const float4* p = reinterpret_cast<const float4*>(src + i0);
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < count; s += 8) {       // eight partials in flight (HBM latency, not bandwidth, bounds this loop), added in index order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (s + u < count) ? p[(long)(s + u) * s4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
      }
      float4 cc = *reinterpret_cast<float4*>(dst + i0);
      cc.x += alpha * a.x; cc.y += alpha * a.y; cc.z += alpha * a.z; cc.w += alpha * a.w;
      *reinterpret_cast<float4*>(dst + i0) = cc;
    } else {
      for (int q = 0; q < 4 && i0 + q < n; ++q) {
        float a = 0.f;
        for (int s = 0; s < count; ++s) a += src[(long)s * stride + i0 + q];
        dst[i0 + q] += alpha * a;
      }
    }
  }
}
}  // namespace

extern "C" int ctts_partial_sums(const ctts_psum_task* tasks, int ntasks, void* stream) {
  CTTS_REQUIRE(ntasks >= 0 && (tasks || ntasks == 0), "ctts_partial_sums: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < ntasks; t0 += PSUM_BATCH) {
    PsumBatch b;
    b.ntasks = 0;
    int blocks = 0;
    for (int t = t0; t < ntasks && t < t0 + PSUM_BATCH; ++t) {
      const ctts_psum_task& k = tasks[t];
      CTTS_REQUIRE(k.src && k.dst && k.n >= 0 && k.count >= 0 && k.stride >= k.n, "ctts_partial_sums: bad task %d", t);
      if (k.n == 0 || k.count == 0) continue;
      const int i = b.ntasks++;
      b.src[i] = k.src; b.dst[i] = k.dst; b.n[i] = k.n; b.stride[i] = k.stride; b.count[i] = k.count; b.alpha[i] = k.alpha;
      b.first_block[i] = blocks;
      blocks += (int)((k.n + 1024 * PSUM_CHUNKS - 1) / (1024 * PSUM_CHUNKS));
    }
    if (b.ntasks == 0) continue;
    b.first_block[b.ntasks] = blocks;
    hipLaunchKernelGGL(partial_sums_kernel, dim3(blocks), dim3(256), 0, st, b);
    CTTS_CHECK_LAUNCH("ctts_partial_sums");
  }
  return 0;
}
