// LayerNorm / BatchNorm1d (channel-last) / masked softmax kernels.  All HBM-bound: one pass over
// the activation per direction, row statistics in registers (one wavefront per row).
#include "ctts_common.h"
#include "planes_common.h"
#include <stdlib.h>

namespace {

constexpr int LN_MAXV = 4;  // float4 per lane -> C <= 1024

// y = rowscale * drop(LN(x)); one wave per row, 4 rows in flight per block
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                             int rows, int C, float eps, float p_drop,
                                                             const uint64_t* seed, uint32_t drop_offset,
                                                             const float* __restrict__ rowscale, uint16_t* __restrict__ planes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 2;
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  const float invC = 1.f / (float)C;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * C);
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < nvec ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mu = ctts_wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      if (lane + 64 * i < nvec) {
        float a = v[i].x - mu, b = v[i].y - mu, c2 = v[i].z - mu, d2 = v[i].w - mu;
        q += a * a + b * b + c2 * c2 + d2 * d2;
      }
    }
    const float rs = rsqrtf(ctts_wave_sum(q) * invC + eps);
    if (lane == 0) { mean_o[row] = mu; rstd_o[row] = rs; }
    const float sc = rowscale ? rowscale[row] : 1.f;
    float4* yr = reinterpret_cast<float4*>(y + (long)row * C);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c];
        const float4 bb = reinterpret_cast<const float4*>(beta)[c];
        float o[4] = {(v[i].x - mu) * rs * g.x + bb.x, (v[i].y - mu) * rs * g.y + bb.y,
                      (v[i].z - mu) * rs * g.z + bb.z, (v[i].w - mu) * rs * g.w + bb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (do_drop) o[e] *= ctts_drop_scale(dkey, (uint32_t)row * (uint32_t)C + (uint32_t)(c * 4 + e), p_drop, inv_keep);
          o[e] *= sc;
        }
        yr[c] = make_float4(o[0], o[1], o[2], o[3]);
        // the bf16 plane set of y for the plane-kernel GEMM that consumes it (FFN conv): the values are in registers here - a separate
        // ctts_split_planes launch would read y back (round 6)
        if (planes) spl_store4(planes, row, C, c * 4, o);
      }
    }
  }
}

// V float4 per lane per row (C <= 256 V), R rows per wave and loop iteration: with one row in flight a wave is bound by the latency of
// its two row loads (16 iterations x ~1.3 us at [16384, 256]); R rows issue their loads back to back and reduce independently.
template <int V, int R>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ dx, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int rows, int C, float p_drop,
                                                             const uint64_t* seed, uint32_t drop_offset,
                                                             const float* __restrict__ rowscale, const float* __restrict__ dres,
                                                             unsigned char* ws, int G, int accumulate, float* __restrict__ parts) {
  __shared__ float s_red[2][4][V * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 2;
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  const float invC = 1.f / (float)C;
  float ag[V][4], ab[V][4];
  float4 gv[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; }
    const int c = lane + 64 * i;
    gv[i] = c < nvec ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row0 = (blockIdx.x * 4 + wave) * R; row0 < rows; row0 += gridDim.x * 4 * R) {
    float4 xv[R][V], dv[R][V];
    float mu[R], rs[R], sc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {          // every load of the R rows first (rows past the end: row clamped, contribution masked)
      const int row = min(row0 + r, rows - 1);
      const float4* xr = reinterpret_cast<const float4*>(x + (long)row * C);
      const float4* dr = reinterpret_cast<const float4*>(dy + (long)row * C);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = min(lane + 64 * i, nvec - 1);
        xv[r][i] = xr[c]; dv[r][i] = dr[c];
      }
      mu[r] = mean[row]; rs[r] = rstd[row];
      sc[r] = (row0 + r < rows) ? (rowscale ? rowscale[row] : 1.f) : 0.f;
    }
    float xh[R][V][4], g[R][V][4], s1[R], s2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s1[r] = 0.f; s2[r] = 0.f;
      const int row = row0 + r;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = lane + 64 * i;
        const bool live = c < nvec;
        const float xa[4] = {xv[r][i].x, xv[r][i].y, xv[r][i].z, xv[r][i].w}, da[4] = {dv[r][i].x, dv[r][i].y, dv[r][i].z, dv[r][i].w};
        const float ga[4] = {gv[i].x, gv[i].y, gv[i].z, gv[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float d = live ? da[e] * sc[r] : 0.f;
          if (do_drop) d *= ctts_drop_scale(dkey, (uint32_t)row * (uint32_t)C + (uint32_t)(c * 4 + e), p_drop, inv_keep);
          xh[r][i][e] = live ? (xa[e] - mu[r]) * rs[r] : 0.f;
          ag[i][e] += d * xh[r][i][e];
          ab[i][e] += d;
          g[r][i][e] = d * ga[e];
          s1[r] += g[r][i][e];
          s2[r] += g[r][i][e] * xh[r][i][e];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      const float c1 = ctts_wave_sum(s1[r]) * invC, c2 = ctts_wave_sum(s2[r]) * invC;
      if (row < rows) {
        float4* dxr = reinterpret_cast<float4*>(dx + (long)row * C);
        const float4* rr = dres ? reinterpret_cast<const float4*>(dres + (long)row * C) : nullptr;   // gradient of the residual branch
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const int c = lane + 64 * i;
          if (c < nvec) {
            float4 o = make_float4(rs[r] * (g[r][i][0] - c1 - xh[r][i][0] * c2), rs[r] * (g[r][i][1] - c1 - xh[r][i][1] * c2),
                                   rs[r] * (g[r][i][2] - c1 - xh[r][i][2] * c2), rs[r] * (g[r][i][3] - c1 - xh[r][i][3] * c2));
            if (rr) { const float4 a = rr[c]; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
            dxr[c] = o;
          }
        }
      }
    }
  }
  // block reduce of the per-wave dgamma / dbeta partials, then the ordered cross-workgroup sum (ctts_common.h): thread l < 64 carries
  // channels k * 64 + l of dgamma (k < 4 V) and of dbeta (k >= 4 V)
#pragma unroll
  for (int i = 0; i < V; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s_red[0][wave][(i * 64 + lane) * 4 + e] = ag[i][e];
      s_red[1][wave][(i * 64 + lane) * 4 + e] = ab[i][e];
    }
  __syncthreads();
  float tot[8 * V];
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < 4 * V; ++k) {
      const int idx = k * 64 + lane;
      tot[k] = s_red[0][0][idx] + s_red[0][1][idx] + s_red[0][2][idx] + s_red[0][3][idx];
      tot[4 * V + k] = s_red[1][0][idx] + s_red[1][1][idx] + s_red[1][2][idx] + s_red[1][3][idx];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8 * V; ++k) tot[k] = 0.f;
  }
  if (parts) {                 // deferred: this workgroup's partial row [dgamma (C) | dbeta (C)]; ctts_partial_sums adds the rows in order later
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < 4 * V; ++k) {
        const int ch = k * 64 + lane;
        if (ch < C) { parts[(long)blockIdx.x * 2 * C + ch] = tot[k]; parts[(long)blockIdx.x * 2 * C + C + ch] = tot[4 * V + k]; }
      }
    }
    return;
  }
  if (!ctts_ordered_colsum<float, 8 * V>(tot, ws, 0, blockIdx.x, gridDim.x, G)) return;
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < 4 * V; ++k) {
      const int ch = k * 64 + lane;
      if (ch < C) {
        dgamma[ch] = accumulate ? dgamma[ch] + tot[k] : tot[k];
        dbeta[ch] = accumulate ? dbeta[ch] + tot[4 * V + k] : tot[4 * V + k];
      }
    }
  }
}

// ---------------------------------------------------------------- BatchNorm1d (channel-last)
// column sums: block = 64 columns x a stripe of rows; thread (c = tid & 63, ty = tid >> 6)
template <int MODE>  // 0: sum x, sum x^2   1: BN-bwd sums (dt, dt*xhat)
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         double* __restrict__ sums, int rows, int C, int creal, int act,
                                                         float p_drop, const uint64_t* seed, uint32_t drop_offset,
                                                         unsigned char* ws, int G) {
  // C is the row width of the (possibly folded) view: a narrow matrix [R, creal] with creal < 64 is read as [R/k, k*creal] so that
  // all 64 lanes of a wave carry data; column c of the view is channel c % creal.
  __shared__ float s1[4][64], s2[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  const int ch = c % creal;
  const int stripe = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * stripe, r1 = min(rows, r0 + stripe);
  float a = 0.f, b = 0.f;
  if (c < C) {
    float mu = 0.f, rs = 1.f, g = 1.f, be = 0.f, inv_keep = 1.f;
    uint32_t dkey = 0;
    const bool do_drop = (MODE == 1) && p_drop > 0.f;
    if (MODE == 1) { mu = mean[ch]; rs = rstd[ch]; g = gamma[ch]; be = beta[ch]; }
    if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 4) {
      const float xv = x[(long)r * C + c];
      if (MODE == 0) { a += xv; b += xv * xv; }
      else {
        float d = dy[(long)r * C + c];
        if (do_drop) d *= ctts_drop_scale(dkey, (uint32_t)r * (uint32_t)C + (uint32_t)c, p_drop, inv_keep);
        const float xh = (xv - mu) * rs;
        d *= ctts_act_grad(xh * g + be, act);
        a += d; b += d * xh;
      }
    }
  }
  s1[ty][threadIdx.x & 63] = a; s2[ty][threadIdx.x & 63] = b;
  __syncthreads();
  double tot[2] = {0.0, 0.0};
  if (ty == 0) {
    const int l = threadIdx.x;
    tot[0] = (double)s1[0][l] + s1[1][l] + s1[2][l] + s1[3][l];
    tot[1] = (double)s2[0][l] + s2[1][l] + s2[2][l] + s2[3][l];
  }
  if (!ctts_ordered_colsum<double, 2>(tot, ws, blockIdx.x, blockIdx.y, gridDim.y, G)) return;
  // the elected workgroup of this column block WRITES the sums (no zero fill, no atomics); a folded view (C = k * creal <= 64, one column
  // block) first adds the k copies of a channel in index order
  if (C != creal) {
    __shared__ double sf[2][64];
    if (ty == 0) { sf[0][threadIdx.x] = c < C ? tot[0] : 0.0; sf[1][threadIdx.x] = c < C ? tot[1] : 0.0; }
    __syncthreads();
    if (ty == 0 && threadIdx.x < creal) {
      double a = 0.0, b = 0.0;
      for (int j = threadIdx.x; j < C; j += creal) { a += sf[0][j]; b += sf[1][j]; }
      sums[threadIdx.x] = a;
      sums[creal + threadIdx.x] = b;
    }
    return;
  }
  if (ty == 0 && c < C) { sums[c] = tot[0]; sums[creal + c] = tot[1]; }
}

__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                long total, int C, int act, float p_drop, const uint64_t* seed, uint32_t drop_offset) {
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    float v = ctts_act((x[e] - mean[c]) * rstd[c] * gamma[c] + beta[c], act);
    if (do_drop) v *= ctts_drop_scale(dkey, (uint32_t)e, p_drop, inv_keep);
    y[e] = v;
  }
}

__device__ __forceinline__ void bn_load8(const float* __restrict__ p, float (&o)[8]) {       // p 16-byte aligned (c % 8 == 0, aligned base)
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void bn_load8d(const double* __restrict__ p, float (&o)[8]) {     // (float) of 8 doubles, 16-byte aligned
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double2 v = *reinterpret_cast<const double2*>(p + 2 * i);
    o[2 * i] = (float)v.x; o[2 * i + 1] = (float)v.y;
  }
}

// the same element-wise map, 8 consecutive channels per thread (C % 8 == 0, 16-byte aligned), with the bf16 plane set of y written next
// to it: the PostNet convolutions that consume y run on the plane kernel (round 6: no separate ctts_split_planes pass over y)
__global__ __launch_bounds__(256) void bn_apply_planes_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              uint16_t* __restrict__ planes, long total8, int C, int act, float p_drop,
                                                              const uint64_t* seed, uint32_t drop_offset) {
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  const int c8n = C >> 3;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total8; g += (long)gridDim.x * blockDim.x) {
    const long r = g / c8n;
    const int c = (int)(g - r * c8n) * 8;
    const long e0 = r * C + c;
    const float4 x0 = *reinterpret_cast<const float4*>(x + e0), x1 = *reinterpret_cast<const float4*>(x + e0 + 4);
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    // per-channel vectors as 16-byte loads: with one channel per load instruction the 64 lanes of a wave touch 16 cache lines for 256
    // useful bytes, 32 times per thread - the first version of this kernel ran 2.8x slower than the scalar one on exactly that
    float pm[8], pr[8], pg[8], pb[8];
    bn_load8(mean + c, pm); bn_load8(rstd + c, pr); bn_load8(gamma + c, pg); bn_load8(beta + c, pb);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = ctts_act((xs[i] - pm[i]) * pr[i] * pg[i] + pb[i], act);
      if (do_drop) v *= ctts_drop_scale(dkey, (uint32_t)(e0 + i), p_drop, inv_keep);
      o[i] = v;
    }
    *reinterpret_cast<float4*>(y + e0) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(y + e0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
    if (planes) spl_store8(planes, r, C, c, o);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_planes_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const double* __restrict__ sums, float* __restrict__ dx,
                                                                  uint16_t* __restrict__ planes, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, long total8, int rows, int C, int act,
                                                                  float p_drop, const uint64_t* seed, uint32_t drop_offset, int batch_stats) {
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  const float invR = 1.f / (float)rows;
  const bool accumulate = (batch_stats & 2) != 0;
  batch_stats &= 1;
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      if (accumulate) { dbeta[c] += (float)sums[c]; dgamma[c] += (float)sums[C + c]; }
      else { dbeta[c] = (float)sums[c]; dgamma[c] = (float)sums[C + c]; }
    }
  const int c8n = C >> 3;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total8; g += (long)gridDim.x * blockDim.x) {
    const long r = g / c8n;
    const int c = (int)(g - r * c8n) * 8;
    const long e0 = r * C + c;
    const float4 d0 = *reinterpret_cast<const float4*>(dy + e0), d1 = *reinterpret_cast<const float4*>(dy + e0 + 4);
    const float4 x0 = *reinterpret_cast<const float4*>(x + e0), x1 = *reinterpret_cast<const float4*>(x + e0 + 4);
    const float ds[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    float pm[8], pr[8], pg[8], pb[8], s0[8], s1[8];
    bn_load8(mean + c, pm); bn_load8(rstd + c, pr); bn_load8(gamma + c, pg); bn_load8(beta + c, pb);
    if (batch_stats) { bn_load8d(sums + c, s0); bn_load8d(sums + C + c, s1); }
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float d = ds[i];
      if (do_drop) d *= ctts_drop_scale(dkey, (uint32_t)(e0 + i), p_drop, inv_keep);
      const float xh = (xs[i] - pm[i]) * pr[i];
      d *= ctts_act_grad(xh * pg[i] + pb[i], act);
      float v = d;
      if (batch_stats) v = d - s0[i] * invR - xh * s1[i] * invR;
      o[i] = pg[i] * pr[i] * v;
    }
    *reinterpret_cast<float4*>(dx + e0) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(dx + e0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
    if (planes) spl_store8(planes, r, C, c, o);
  }
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const double* __restrict__ sums, float* __restrict__ dx,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, long total, int rows, int C, int act,
                                    float p_drop, const uint64_t* seed, uint32_t drop_offset, int batch_stats) {
  const bool do_drop = p_drop > 0.f;
  uint32_t dkey = 0; float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(seed, drop_offset); inv_keep = 1.f / (1.f - p_drop); }
  const float invR = 1.f / (float)rows;
  const bool accumulate = (batch_stats & 2) != 0;      // bit 1: add to dgamma / dbeta (gradient-accumulation fusion into param.grad)
  batch_stats &= 1;
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      if (accumulate) { dbeta[c] += (float)sums[c]; dgamma[c] += (float)sums[C + c]; }
      else { dbeta[c] = (float)sums[c]; dgamma[c] = (float)sums[C + c]; }
    }
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    float d = dy[e];
    if (do_drop) d *= ctts_drop_scale(dkey, (uint32_t)e, p_drop, inv_keep);
    const float xh = (x[e] - mean[c]) * rstd[c];
    d *= ctts_act_grad(xh * gamma[c] + beta[c], act);
    float v = d;
    if (batch_stats) v = d - (float)sums[c] * invR - xh * (float)sums[C + c] * invR;
    dx[e] = gamma[c] * rstd[c] * v;
  }
}

// ---------------------------------------------------------------- masked softmax over keys
constexpr int SM_MAXV = 16;  // 64 * 16 = 1024 keys held in registers

template <bool BWD>
__global__ __launch_bounds__(256) void softmax_kernel(float* __restrict__ S, const float* __restrict__ P,
                                                       const int32_t* __restrict__ lens, int nb1, int T, long ld, long nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < nrows; row += (long)gridDim.x * 4) {
    const long z = row / T;
    const int q = (int)(row - z * T);
    const int L = lens ? min(lens[z / nb1], T) : T;
    if (q >= L) continue;
    float* s = S + (z * T + q) * ld;
    if (!BWD) {
      if (L <= 64 * SM_MAXV) {
        float v[SM_MAXV];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < SM_MAXV; ++i) {
          const int k = lane + 64 * i;
          v[i] = k < L ? s[k] : -INFINITY;
          mx = fmaxf(mx, v[i]);
        }
        mx = ctts_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < SM_MAXV; ++i) { v[i] = (lane + 64 * i) < L ? __expf(v[i] - mx) : 0.f; sum += v[i]; }
        const float inv = 1.f / ctts_wave_sum(sum);
#pragma unroll
        for (int i = 0; i < SM_MAXV; ++i) { const int k = lane + 64 * i; if (k < L) s[k] = v[i] * inv; }
      } else {
        float mx = -INFINITY;
        for (int k = lane; k < L; k += 64) mx = fmaxf(mx, s[k]);
        mx = ctts_wave_max(mx);
        float sum = 0.f;
        for (int k = lane; k < L; k += 64) sum += __expf(s[k] - mx);
        const float inv = 1.f / ctts_wave_sum(sum);
        for (int k = lane; k < L; k += 64) s[k] = __expf(s[k] - mx) * inv;
      }
    } else {
      const float* p = P + (z * T + q) * ld;
      float dot = 0.f;
      for (int k = lane; k < L; k += 64) dot += s[k] * p[k];
      dot = ctts_wave_sum(dot);
      for (int k = lane; k < L; k += 64) s[k] = p[k] * (s[k] - dot);
    }
  }
}

}  // namespace

extern "C" int ctts_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, int rows, int C, float eps, float p_drop, const uint64_t* seed,
                                  uint32_t drop_offset, const float* rowscale, uint16_t* planes, void* stream) {
  CTTS_REQUIRE(x && gamma && beta && y && mean && rstd, "ctts_layernorm_fwd: null pointer");
  CTTS_REQUIRE((C % 4) == 0 && C <= 1024 && C > 0, "ctts_layernorm_fwd: C=%d must be a multiple of 4 and <= 1024", C);
  CTTS_REQUIRE(!planes || ((C % 32) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15) == 0),
               "ctts_layernorm_fwd: a plane set needs C %% 32 == 0 (got %d) and a 16-byte aligned pointer", C);
  if (rows == 0) return 0;
  const int blocks = min((rows + 3) / 4, 2048);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd,
                     rows, C, eps, p_drop, seed, drop_offset, rowscale, planes);
  CTTS_CHECK_LAUNCH("ctts_layernorm_fwd");
  return 0;
}

// workgroups of a ctts_layernorm_bwd launch with a workspace (or in deferred mode) = partial rows it produces
static int ln_bwd_block_cap() {
  static const int ln_blocks_env = getenv("CTTS_LN_BWD_BLOCKS") ? atoi(getenv("CTTS_LN_BWD_BLOCKS")) : 256;     // tuning knob
  return max(1, min(ln_blocks_env, 512));       // the workspace keeps one partial (2 C floats) per workgroup: 512 x 8 KB = the 4 MiB partial area
}
int ctts_layernorm_bwd_blocks(int rows, int C) {
  if (rows <= 0) return 0;
  const int cap = ln_bwd_block_cap();
  return C <= 256 ? min((rows + 15) / 16, cap) : (C <= 512 ? min((rows + 7) / 8, cap) : min((rows + 3) / 4, cap));
}

extern "C" int ctts_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                  const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int C,
                                  float p_drop, const uint64_t* seed, uint32_t drop_offset, const float* rowscale,
                                  int accumulate, const float* dres, void* ws, float* parts, void* stream) {
  CTTS_REQUIRE(dy && x && gamma && mean && rstd && dx && ((dgamma && dbeta) || parts), "ctts_layernorm_bwd: null pointer");
  CTTS_REQUIRE((C % 4) == 0 && C <= 1024 && C > 0, "ctts_layernorm_bwd: C=%d must be a multiple of 4 and <= 1024", C);
  CTTS_REQUIRE(!parts || rows > 0, "ctts_layernorm_bwd: deferred mode needs rows > 0");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (!accumulate && (ctts_zero_async(dgamma, sizeof(float) * C, st) != 0 || ctts_zero_async(dbeta, sizeof(float) * C, st) != 0)) {
      ctts_set_error("ctts_layernorm_bwd: zero fill failed");
      return -2;
    }
    return 0;
  }
  // without a workspace (and not deferred): one workgroup - slow, still deterministic
  const int nb = (ws || parts) ? ctts_layernorm_bwd_blocks(rows, C) : 1;
  unsigned char* w8 = (unsigned char*)ws;
  if (C <= 256)
    hipLaunchKernelGGL((layernorm_bwd_kernel<1, 4>), dim3(nb), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx,
                       dgamma, dbeta, rows, C, p_drop, seed, drop_offset, rowscale, dres, w8, ctts_red_group(nb), accumulate, parts);
  else if (C <= 512)
    hipLaunchKernelGGL((layernorm_bwd_kernel<2, 2>), dim3(nb), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx,
                       dgamma, dbeta, rows, C, p_drop, seed, drop_offset, rowscale, dres, w8, ctts_red_group(nb), accumulate, parts);
  else
    hipLaunchKernelGGL((layernorm_bwd_kernel<4, 1>), dim3(nb), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows,
                       C, p_drop, seed, drop_offset, rowscale, dres, w8, ctts_red_group(nb), accumulate, parts);
  CTTS_CHECK_LAUNCH("ctts_layernorm_bwd");
  return 0;
}

// fold factor for narrow matrices and a grid of ~1024 workgroups (HBM-bound reduction: keep every CU streaming)
static int colreduce_fold(int rows, int C) {
  int k = 1;
  while (C * k * 2 <= 64 && rows % (k * 2) == 0) k *= 2;
  return k;
}
// stripes: the workspace holds one partial (2 x 64 doubles) per workgroup; without a workspace one stripe per column block
static int colreduce_grid_y(int rows, int gx, bool have_ws) {
  if (!have_ws) return 1;
  const int cap = (int)(CTTS_WS_RED_P1_BYTES / (2 * 64 * sizeof(double))) / gx;
  return max(1, min(min(max(1, 1024 / gx), rows / 64), min(cap, CTTS_RED_MAX_GROUPS * 64)));
}

extern "C" int ctts_colstats(const float* x, double* sums, int rows, int C, void* ws, void* stream) {
  CTTS_REQUIRE(x && sums && rows > 0 && C > 0 && (C + 63) / 64 <= CTTS_RED_MAX_COLBLOCKS, "ctts_colstats: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int k = colreduce_fold(rows, C), Cv = C * k, Rv = rows / k, gx = (Cv + 63) / 64;
  const int gy = colreduce_grid_y(Rv, gx, ws != nullptr);
  hipLaunchKernelGGL((colreduce_kernel<0>), dim3(gx, gy), dim3(256), 0, st, x, nullptr, nullptr,
                     nullptr, nullptr, nullptr, sums, Rv, Cv, C, 0, 0.f, nullptr, 0u, (unsigned char*)ws, ctts_red_group(gy));
  CTTS_CHECK_LAUNCH("ctts_colstats");
  return 0;
}

namespace {
// batch statistics from the double column sums + nn.BatchNorm running-stat update (momentum, unbiased variance), one launch
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int rows, int C, float eps, float momentum, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ num_batches) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const double mu = sums[c] / rows;
    double var = sums[C + c] / rows - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    const float m = (float)mu, v = (float)var;
    mean[c] = m;
    rstd[c] = rsqrtf(v + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (v * ((float)rows / (float)max(rows - 1, 1)));
  }
  if (c == 0 && num_batches) *num_batches += 1;
}
}  // namespace

extern "C" int ctts_bn_finalize(const double* sums, int rows, int C, float eps, float momentum, float* mean, float* rstd,
                                float* running_mean, float* running_var, int64_t* num_batches, void* stream) {
  CTTS_REQUIRE(sums && mean && rstd && rows > 0 && C > 0, "ctts_bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, rows, C, eps, momentum, mean, rstd,
                     running_mean, running_var, (long long*)num_batches);
  CTTS_CHECK_LAUNCH("ctts_bn_finalize");
  return 0;
}

extern "C" int ctts_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             float* y, int rows, int C, int act, float p_drop, const uint64_t* seed, uint32_t drop_offset,
                             uint16_t* planes, void* stream) {
  CTTS_REQUIRE(x && mean && rstd && gamma && beta && y, "ctts_bn_apply: null pointer");
  const long total = (long)rows * C;
  if (total == 0) return 0;
  const bool wide_ok = (C % 8) == 0 && total < (1L << 32) &&
                       ((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd) | reinterpret_cast<uintptr_t>(gamma) |
                         reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
  CTTS_REQUIRE(!planes || (wide_ok && (C % 32) == 0),
               "ctts_bn_apply: a plane set needs C %% 32 == 0 (got %d), 16-byte aligned pointers (x, y, planes and the per-channel vectors) and < 2^32 elements", C);
  static const bool wide_on = !(getenv("CTTS_BN_WIDE") && atoi(getenv("CTTS_BN_WIDE")) == 0);      // A/B switch: 0 = the scalar kernel unless planes are asked for
  if (planes || (wide_ok && wide_on)) {      // 8 channels per thread, 16-byte accesses (same arithmetic, element for element)
    const long total8 = total / 8;
    const int pb = (int)min((total8 + 255) / 256, (long)4096);
    hipLaunchKernelGGL(bn_apply_planes_kernel, dim3(pb), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, y, planes, total8, C,
                       act, p_drop, seed, drop_offset);
    CTTS_CHECK_LAUNCH("ctts_bn_apply");
    return 0;
  }
  const int blocks = (int)min((total + 255) / 256, (long)4096);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, y, total, C,
                     act, p_drop, seed, drop_offset);
  CTTS_CHECK_LAUNCH("ctts_bn_apply");
  return 0;
}

extern "C" int ctts_bn_bwd_reduce(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, double* sums, int rows, int C, int act, float p_drop,
                                  const uint64_t* seed, uint32_t drop_offset, void* ws, void* stream) {
  CTTS_REQUIRE(dy && x && mean && rstd && gamma && beta && sums && rows > 0 && (C + 63) / 64 <= CTTS_RED_MAX_COLBLOCKS,
               "ctts_bn_bwd_reduce: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int k = colreduce_fold(rows, C), Cv = C * k, Rv = rows / k, gx = (Cv + 63) / 64;
  const int gy = colreduce_grid_y(Rv, gx, ws != nullptr);
  hipLaunchKernelGGL((colreduce_kernel<1>), dim3(gx, gy), dim3(256), 0, st, x, dy, mean, rstd,
                     gamma, beta, sums, Rv, Cv, C, act, p_drop, seed, drop_offset, (unsigned char*)ws, ctts_red_group(gy));
  CTTS_CHECK_LAUNCH("ctts_bn_bwd_reduce");
  return 0;
}

extern "C" int ctts_bn_bwd_apply(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, const double* sums, float* dx, float* dgamma, float* dbeta, int rows,
                                 int C, int act, float p_drop, const uint64_t* seed, uint32_t drop_offset, int batch_stats,
                                 uint16_t* planes, void* stream) {
  CTTS_REQUIRE(dy && x && mean && rstd && gamma && beta && sums && dx && dgamma && dbeta, "ctts_bn_bwd_apply: null pointer");
  const long total = (long)rows * C;
  if (total == 0) return 0;
  const bool wide_ok = (C % 8) == 0 && total < (1L << 32) &&
                       ((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                         reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd) | reinterpret_cast<uintptr_t>(gamma) |
                         reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(sums)) & 15) == 0;
  CTTS_REQUIRE(!planes || (wide_ok && (C % 32) == 0),
               "ctts_bn_bwd_apply: a plane set needs C %% 32 == 0 (got %d), 16-byte aligned pointers (tensors, planes, per-channel vectors) and < 2^32 elements", C);
  static const bool wide_on = !(getenv("CTTS_BN_WIDE") && atoi(getenv("CTTS_BN_WIDE")) == 0);
  if (planes || (wide_ok && wide_on)) {
    const long total8 = total / 8;
    const int pb = (int)min((total8 + 255) / 256, (long)4096);
    hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3(pb), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, beta, sums, dx,
                       planes, dgamma, dbeta, total8, rows, C, act, p_drop, seed, drop_offset, batch_stats);
    CTTS_CHECK_LAUNCH("ctts_bn_bwd_apply");
    return 0;
  }
  const int blocks = (int)min((total + 255) / 256, (long)4096);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, beta, sums,
                     dx, dgamma, dbeta, total, rows, C, act, p_drop, seed, drop_offset, batch_stats);
  CTTS_CHECK_LAUNCH("ctts_bn_bwd_apply");
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void rowdot_heads_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ out, int T, int H, int dh, long nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < nrows; row += (long)gridDim.x * 4) {     // row = (b*H + h)*T + t
    const int t = (int)(row % T);
    const long bh = row / T;
    const int hh = (int)(bh % H);
    const long bb = bh / H;
    const long base = ((bb * T + t) * H + hh) * (long)dh;
    float s = 0.f;
    for (int d = lane; d < dh; d += 64) s += a[base + d] * b[base + d];
    s = ctts_wave_sum(s);
    if (lane == 0) out[row] = s;
  }
}
}  // namespace

extern "C" int ctts_rowdot_heads(const float* a, const float* b, float* out, int B, int T, int H, int dh, void* stream) {
  CTTS_REQUIRE(a && b && out && B >= 0 && T > 0 && H > 0 && dh > 0, "ctts_rowdot_heads: bad arguments");
  const long nrows = (long)B * H * T;
  if (nrows == 0) return 0;
  const int blocks = (int)min((nrows + 3) / 4, (long)8192);
  hipLaunchKernelGGL(rowdot_heads_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, T, H, dh, nrows);
  CTTS_CHECK_LAUNCH("ctts_rowdot_heads");
  return 0;
}

extern "C" int ctts_softmax_fwd(float* S, const int32_t* lens, int nb0, int nb1, int T, int64_t ld, void* stream) {
  CTTS_REQUIRE(S && nb0 > 0 && nb1 > 0 && T > 0, "ctts_softmax_fwd: bad arguments");
  const long nrows = (long)nb0 * nb1 * T;
  const int blocks = (int)min((nrows + 3) / 4, (long)8192);
  hipLaunchKernelGGL((softmax_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, (const float*)nullptr, lens,
                     nb1, T, (long)ld, nrows);
  CTTS_CHECK_LAUNCH("ctts_softmax_fwd");
  return 0;
}

extern "C" int ctts_softmax_bwd(const float* P, float* dP, const int32_t* lens, int nb0, int nb1, int T, int64_t ld,
                                void* stream) {
  CTTS_REQUIRE(P && dP && nb0 > 0 && nb1 > 0 && T > 0, "ctts_softmax_bwd: bad arguments");
  const long nrows = (long)nb0 * nb1 * T;
  const int blocks = (int)min((nrows + 3) / 4, (long)8192);
  hipLaunchKernelGGL((softmax_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dP, P, lens, nb1, T, (long)ld,
                     nrows);
  CTTS_CHECK_LAUNCH("ctts_softmax_bwd");
  return 0;
}
