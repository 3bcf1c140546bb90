// LengthRegulator / dur_to_mel2ph / make_positions: wavefront prefix scans + coalesced gathers.
// All index arithmetic is integer and bit-exact against the reference
// (model/modules.py:1216-1249, utils/tools.py:577-652).  HBM-bound: 4*C*(Ts+Tm) bytes per sample.
#include "ctts_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
void ctts_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ctts_last_error(void) { return g_err; }
extern "C" int ctts_version(void) { return 1; }

namespace {

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// one block per batch row.  Phase 1 (wave 0): scan durations -> cum (LDS + global).
// Phase 2 (all threads): upper-bound search of t in cum -> mel2ph.
__global__ __launch_bounds__(256) void lr_index_kernel(const void* dur, int dur_is_float, int round_mode,
                                                        const uint8_t* pad, int Ts, int Tm, int32_t* mel2ph,
                                                        int64_t* mel_len, int32_t* cum_out) {
  extern __shared__ int s_cum[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  __shared__ int s_total;
  if (tid < 64) {
    int carry = 0;
    for (int i0 = 0; i0 < Ts; i0 += 64) {
      const int i = i0 + lane;
      int dv = 0;
      if (i < Ts) {
        if (dur_is_float) {
          float f = reinterpret_cast<const float*>(dur)[(long)b * Ts + i];
          f = round_mode ? rintf(f) : truncf(f);
          dv = f > 0.f ? (f < 2.0e9f ? (int)f : 2000000000) : 0;
        } else {
          long long v = reinterpret_cast<const long long*>(dur)[(long)b * Ts + i];
          dv = v > 0 ? (v < 2000000000LL ? (int)v : 2000000000) : 0;
        }
        if (pad && pad[(long)b * Ts + i]) dv = 0;
      }
      int inc = wave_incl_scan(dv, lane) + carry;
      if (i < Ts) {
        s_cum[i] = inc;
        if (cum_out) cum_out[(long)b * Ts + i] = inc;
      }
      carry = __shfl(inc, 63, 64);
    }
    if (lane == 0) {
      s_total = carry;
      if (mel_len) mel_len[b] = carry;
    }
  }
  __syncthreads();
  if (!mel2ph) return;
  const int total = s_total;
  for (int t = tid; t < Tm; t += 256) {
    int v = 0;
    if (t < total) {
      int lo = 0, hi = Ts;  // first i with cum[i] > t
      while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (s_cum[mid] > t) hi = mid; else lo = mid + 1;
      }
      v = lo + 1;
    }
    mel2ph[(long)b * Tm + t] = v;
  }
}

__global__ void lr_gather_fwd_kernel(const float4* __restrict__ x, const int32_t* __restrict__ mel2ph,
                                     float4* __restrict__ out, int Ts, int Tm, int C4, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / C4;
    const int c = (int)(e - row * C4);
    const int b = (int)(row / Tm);
    const int ph = mel2ph[row];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ph > 0) v = x[((long)b * Ts + (ph - 1)) * C4 + c];
    out[e] = v;
  }
}

__global__ void lr_gather_bwd_kernel(const float4* __restrict__ dy, const int32_t* __restrict__ cum,
                                     float4* __restrict__ dx, int Ts, int Tm, int C4, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / C4;  // (b, i)
    const int c = (int)(e - row * C4);
    const int b = (int)(row / Ts), i = (int)(row - (long)b * Ts);
    const int start = i > 0 ? cum[row - 1] : 0;
    const int end = min(cum[row], Tm);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = start; t < end; ++t) {
      float4 v = dy[((long)b * Tm + t) * C4 + c];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dx[e] = s;
  }
}

__global__ __launch_bounds__(64) void positions_kernel(const void* src, int src_is_float, long stride, int T,
                                                        int32_t* pos) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int carry = 0;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    int nz = 0;
    if (t < T) {
      if (src_is_float) nz = reinterpret_cast<const float*>(src)[((long)b * T + t) * stride] != 0.f;
      else nz = reinterpret_cast<const long long*>(src)[((long)b * T + t) * stride] != 0;
    }
    int inc = wave_incl_scan(nz, lane) + carry;
    if (t < T) pos[(long)b * T + t] = nz ? inc : 0;
    carry = __shfl(inc, 63, 64);
  }
}

}  // namespace

extern "C" int ctts_lr_index(const void* dur, int dur_is_float, int round_mode, const uint8_t* pad, int B, int Ts,
                             int Tm, int32_t* mel2ph, int64_t* mel_len, int32_t* cum, void* stream) {
  CTTS_REQUIRE(dur && B >= 0 && Ts > 0 && Tm >= 0, "ctts_lr_index: bad arguments");
  CTTS_REQUIRE((size_t)Ts * 4 <= 60000, "ctts_lr_index: Ts=%d too large for the LDS scan buffer", Ts);
  if (B == 0) return 0;
  hipLaunchKernelGGL(lr_index_kernel, dim3(B), dim3(256), (size_t)Ts * sizeof(int), (hipStream_t)stream, dur,
                     dur_is_float, round_mode, pad, Ts, Tm, mel2ph, mel_len, cum);
  CTTS_CHECK_LAUNCH("ctts_lr_index");
  return 0;
}

extern "C" int ctts_lr_gather_fwd(const float* x, const int32_t* mel2ph, float* out, int B, int Ts, int Tm, int C,
                                  void* stream) {
  CTTS_REQUIRE(x && mel2ph && out && (C % 4) == 0, "ctts_lr_gather_fwd: bad arguments (C %% 4 must be 0)");
  const long total = (long)B * Tm * (C / 4);
  if (total == 0) return 0;
  const int blocks = (int)min((total + 255) / 256, (long)2048);
  hipLaunchKernelGGL(lr_gather_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), mel2ph, reinterpret_cast<float4*>(out), Ts, Tm, C / 4, total);
  CTTS_CHECK_LAUNCH("ctts_lr_gather_fwd");
  return 0;
}

extern "C" int ctts_lr_gather_bwd(const float* dy, const int32_t* cum, float* dx, int B, int Ts, int Tm, int C,
                                  void* stream) {
  CTTS_REQUIRE(dy && cum && dx && (C % 4) == 0, "ctts_lr_gather_bwd: bad arguments (C %% 4 must be 0)");
  const long total = (long)B * Ts * (C / 4);
  if (total == 0) return 0;
  const int blocks = (int)min((total + 255) / 256, (long)2048);
  hipLaunchKernelGGL(lr_gather_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(dy), cum, reinterpret_cast<float4*>(dx), Ts, Tm, C / 4, total);
  CTTS_CHECK_LAUNCH("ctts_lr_gather_bwd");
  return 0;
}

extern "C" int ctts_positions(const void* src, int src_is_float, int64_t stride, int B, int T, int32_t* pos,
                              void* stream) {
  CTTS_REQUIRE(src && pos && T > 0, "ctts_positions: bad arguments");
  if (B == 0) return 0;
  hipLaunchKernelGGL(positions_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, src, src_is_float, (long)stride, T,
                     pos);
  CTTS_CHECK_LAUNCH("ctts_positions");
  return 0;
}

// ---------------------------------------------------------------- embedding lookup (blocks.py:10-15 Embedding, modules.py:779-788,
// 947,958 pitch / energy embeddings): forward gather; backward without a sort and without atomics (below).
namespace {
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const long long* __restrict__ ids, const float4* __restrict__ w,
                                                             float4* __restrict__ out, int C4, int V, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / C4;
    const int c = (int)(e - r * C4);
    const long long id = ids[r];
    out[e] = (id >= 0 && id < V) ? w[id * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// Workgroup (v, g) of EMB_WAVES waves: vocabulary row v, group g of the 64-id chunks.  Wave w scans the chunks c0 + w, c0 + w + EMB_WAVES,
// ... of its group in order, ballots each chunk for its row and sums the selected gradient rows in registers (four rows in flight); the
// waves' sums are added through LDS in wave order, the groups' sums through the workspace in group order by the workgroup that takes the
// row's last ticket (ctts_common.h).  The order of every addition is fixed by (v, chunk, position), not by timing: bit-reproducible
// (round 3 issued one float atomic per (row, chunk, channel)).  A popular row (the unvoiced pitch bin, ~30 % of all frames) is spread
// over NG x 16 waves.
constexpr int EMB_WAVES = 16;
template <int NC>   // floats per lane: C <= 64 * NC
__global__ __launch_bounds__(64 * EMB_WAVES) void embedding_bwd_kernel(const long long* __restrict__ ids, const float* __restrict__ dy,
                                                                        float* __restrict__ dw, long n, int C, int V, int padding_idx,
                                                                        int accumulate, unsigned char* ws) {
  __shared__ float s_acc[EMB_WAVES][NC * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v = blockIdx.x, g = blockIdx.y, NG = gridDim.y;
  if (v == padding_idx) return;                   // its gradient row stays as it is (zero-filled by the launcher unless accumulating)
  const long chunks = (n + 63) / 64, cpg = (chunks + NG - 1) / NG;
  const long c_end = min(chunks, (long)(g + 1) * cpg);
  float acc[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) acc[i] = 0.f;
  for (long ch = (long)g * cpg + wave; ch < c_end; ch += EMB_WAVES) {
    const long base = ch * 64, p = base + lane;
    unsigned long long m = __ballot(p < n && ids[p] == (long long)v);
    while (m) {
      int bpos[4];
      float x[4][NC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                // up to four selected rows in flight, added in position order
        bpos[u] = m ? __builtin_ctzll(m) : -1;
        if (m) m &= m - 1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* row = dy + (base + (bpos[u] < 0 ? 0 : bpos[u])) * C;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const int c = lane + 64 * i;
          x[u][i] = (bpos[u] >= 0 && c < C) ? row[c] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NC; ++i) acc[i] += x[u][i];
    }
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) s_acc[wave][i * 64 + lane] = acc[i];
  __syncthreads();
  float a = 0.f;
  const int c = threadIdx.x;                      // C <= 512 < 1024 threads: thread c finishes channel c
  if (c < C) {
#pragma unroll
    for (int w = 0; w < EMB_WAVES; ++w) a += s_acc[w][c];
  }
  if (NG > 1) {
    float* p1 = reinterpret_cast<float*>(ws + CTTS_WS_RED_P1);
    unsigned* t1 = reinterpret_cast<unsigned*>(ws + CTTS_WS_RED_T1);
    if (c < C) ctts_st_agent(p1 + ((long)v * NG + g) * C + c, a);
    if (!ctts_arrive_last(t1 + v * CTTS_RED_MAX_GROUPS, (unsigned)NG)) return;
    a = 0.f;
    if (c < C)
      for (int gg = 0; gg < NG; ++gg) a += ctts_ld_agent(p1 + ((long)v * NG + gg) * C + c);
  }
  if (c < C) {
    float* o = dw + (long)v * C + c;
    *o = accumulate ? *o + a : a;
  }
}
}  // namespace

extern "C" int ctts_embedding_fwd(const int64_t* ids, const float* weight, float* out, int64_t n, int C, int V, void* stream) {
  CTTS_REQUIRE(ids && weight && out && n >= 0 && C > 0 && (C % 4) == 0 && V > 0, "ctts_embedding_fwd: bad arguments (C %% 4 must be 0)");
  const long total = (long)n * (C / 4);
  if (total == 0) return 0;
  const int blocks = (int)min((total + 255) / 256, (long)4096);
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, (const float4*)weight,
                     (float4*)out, C / 4, V, total);
  CTTS_CHECK_LAUNCH("ctts_embedding_fwd");
  return 0;
}

extern "C" int ctts_embedding_bwd(const int64_t* ids, const float* dy, float* dweight, int64_t n, int C, int V, int padding_idx,
                                  int accumulate, void* ws, void* stream) {
  CTTS_REQUIRE(ids && dy && dweight && n >= 0 && C > 0 && C <= 512 && V > 0, "ctts_embedding_bwd: bad arguments (C <= 512)");
  hipStream_t st = (hipStream_t)stream;
  // every non-padding row is WRITTEN by its workgroup; only the padding row (and everything when n == 0) needs the zero fill
  if (!accumulate) {
    const bool pad_row = padding_idx >= 0 && padding_idx < V;
    if (n == 0) { if (ctts_zero_async(dweight, sizeof(float) * (size_t)V * C, st) != 0) { ctts_set_error("ctts_embedding_bwd: zero fill failed"); return -2; } }
    else if (pad_row && ctts_zero_async(dweight + (size_t)padding_idx * C, sizeof(float) * (size_t)C, st) != 0) {
      ctts_set_error("ctts_embedding_bwd: zero fill failed");
      return -2;
    }
  }
  if (n == 0) return 0;
  // groups of chunks per vocabulary row: enough workgroups to fill the chip and to spread a popular row, within the workspace's partial
  // area (one C-float partial per workgroup) and ticket rows; without a workspace: one workgroup per row
  int NG = 1;
  if (ws && V <= CTTS_RED_MAX_COLBLOCKS) {
    const long chunks = (n + 63) / 64;
    const long cap = (long)(CTTS_WS_RED_P1_BYTES / sizeof(float)) / ((long)V * C);
    NG = (int)max(1L, min(min(8L, cap), chunks / (2 * EMB_WAVES)));
  }
  const dim3 grid(V, NG), block(64 * EMB_WAVES);
  const long long* idp = (const long long*)ids;
  unsigned char* w8 = (unsigned char*)ws;
  if (C <= 64) hipLaunchKernelGGL((embedding_bwd_kernel<1>), grid, block, 0, st, idp, dy, dweight, (long)n, C, V, padding_idx, accumulate, w8);
  else if (C <= 256) hipLaunchKernelGGL((embedding_bwd_kernel<4>), grid, block, 0, st, idp, dy, dweight, (long)n, C, V, padding_idx, accumulate, w8);
  else hipLaunchKernelGGL((embedding_bwd_kernel<8>), grid, block, 0, st, idp, dy, dweight, (long)n, C, V, padding_idx, accumulate, w8);
  CTTS_CHECK_LAUNCH("ctts_embedding_bwd");
  return 0;
}
