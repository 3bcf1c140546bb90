// Unsupervised-alignment kernels:
//  * ctts_neg_sqdist : the aligner's "Gaussian isotropic attention" scores  attn[b,t,s] = -temp * sum_c (q[b,t,c]-k[b,s,c])^2
//                      (model/modules.py:1199-1200) without the reference's [B,80,Tm,Ts] broadcast intermediate (671 MB at B=16).
//  * ctts_mas        : monotonic alignment search, width 1 (model/modules.py:36-75 mas_width1 / b_mas), on the device:
//                      one workgroup per utterance, threads over text positions, rows of the DP relaxed in sequence with the
//                      previous row in LDS; back-pointers as one byte per cell; the backtrack by a single lane.  Replaces the
//                      reference's device->host->device round trip through numba (modules.py:869-872).
#include "ctts_common.h"

namespace {

// q [B,Tq,C], k [B,Tk,C] (channel-last), out [B,Tq,Tk].  Block: 64 q-rows x all C of k tile in LDS.
__global__ __launch_bounds__(256) void neg_sqdist_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          float* __restrict__ out, int Tq, int Tk, int C, float temp) {
  extern __shared__ float s_k[];                  // [64][C+1] tile of keys
  const int b = blockIdx.z;
  const int s0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const int ld = C + 1;
  for (int e = threadIdx.x; e < 64 * C; e += 256) {
    const int s = e / C, c = e - s * C;
    s_k[s * ld + c] = (s0 + s < Tk) ? k[((long)b * Tk + s0 + s) * C + c] : 0.f;
  }
  __syncthreads();
  const int sl = threadIdx.x & 63, tg = threadIdx.x >> 6;     // lane -> key, wave -> 16 query rows
  for (int tt = tg; tt < 64; tt += 4) {
    const int t = t0 + tt;
    if (t >= Tq) break;
    const float* qr = q + ((long)b * Tq + t) * C;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = qr[c] - s_k[sl * ld + c];            // qr[c]: wave-uniform address (broadcast load)
      acc += d * d;
    }
    if (s0 + sl < Tk) out[((long)b * Tq + t) * Tk + s0 + sl] = -temp * acc;
  }
}

// attn [B, Tq, Tk] probabilities (soft attention); opt [B,Tq,Tk] (zero-filled here); dur [B,Tk] float.
__global__ __launch_bounds__(256) void mas_kernel(const float* __restrict__ attn, const int* __restrict__ in_lens,
                                                   const int* __restrict__ out_lens, float* __restrict__ opt,
                                                   float* __restrict__ dur, unsigned char* __restrict__ back, int Tq, int Tk) {
  extern __shared__ float s_row[];                // two rows of log_p: [2][Tk]
  const int b = blockIdx.x;
  const int T1 = min(out_lens[b], Tq), T2 = min(in_lens[b], Tk);
  const float* A = attn + (long)b * Tq * Tk;
  float* O = opt + (long)b * Tq * Tk;
  unsigned char* Bk = back + (long)b * Tq * Tk;
  for (long e = threadIdx.x; e < (long)Tq * Tk; e += 256) O[e] = 0.f;
  for (int j = threadIdx.x; j < Tk; j += 256) dur[(long)b * Tk + j] = 0.f;
  if (T1 <= 0 || T2 <= 0) return;
  // row 0: log(attn[0,0]), -inf elsewhere     (attn_map[0, 1:] = -inf)
  for (int j = threadIdx.x; j < T2; j += 256) s_row[j] = j == 0 ? (float)log((double)A[0]) : -INFINITY;
  __syncthreads();
  for (int i = 1; i < T1; ++i) {
    const float* prev = s_row + ((i - 1) & 1) * Tk;
    float* cur = s_row + (i & 1) * Tk;
    for (int j = threadIdx.x; j < T2; j += 256) {
      float pl = prev[j];
      unsigned char from_left = 0;
      if (j >= 1 && prev[j - 1] >= pl) { pl = prev[j - 1]; from_left = 1; }
      cur[j] = (float)log((double)A[(long)i * Tk + j]) + pl;
      Bk[(long)i * Tk + j] = from_left;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int curj = T2 - 1;
    for (int i = T1 - 1; i >= 0; --i) {
      O[(long)i * Tk + curj] = 1.f;
      if (i > 0 && Bk[(long)i * Tk + curj]) curj -= 1;       // prev_ind[0, :] = 0 in the reference: opt[0, 0] is set below
      else if (i == 0) curj = 0;
    }
    O[curj] = 1.f;                                            // opt[0, curr_text_idx] = 1 with curr_text_idx = prev_ind[0, .] = 0
  }
  __syncthreads();
  // durations = column sums of the hard alignment (attn_hard.sum(2), modules.py:1042)
  for (int j = threadIdx.x; j < T2; j += 256) {
    float s = 0.f;
    for (int i = 0; i < T1; ++i) s += O[(long)i * Tk + j];
    dur[(long)b * Tk + j] = s;
  }
}

// Fast path: back-pointers as BIT masks in LDS (one ballot per wave and row), attention rows prefetched one 8-row chunk ahead, the
// back-track walks LDS (not HBM) and the hard alignment / durations are written by all threads from the recorded path.
// Same arithmetic and tie rules as mas_kernel above.  NJ = columns per thread (Tk <= 256 * NJ).
constexpr int MAS_CH = 8;
// log of the soft attention for ALL utterances in parallel (correctly rounded float log via double), written into the `opt` output
// buffer, which the DP kernel reads as its input and overwrites with the hard alignment only after its last DP row.
__global__ void mas_log_kernel(const float* __restrict__ attn, float* __restrict__ out, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    out[e] = (float)log((double)attn[e]);
}

template <int NJ>
__global__ __launch_bounds__(256) void mas_lds_kernel(const int* __restrict__ in_lens, const int* __restrict__ out_lens,
                                                       float* __restrict__ opt, float* __restrict__ dur, int Tq, int Tk) {
  extern __shared__ __attribute__((aligned(8))) unsigned char mas_smem[];
  const int W64 = (Tk + 63) / 64;
  unsigned long long* s_bits = reinterpret_cast<unsigned long long*>(mas_smem);              // [Tq][W64]
  float* s_row = reinterpret_cast<float*>(s_bits + (size_t)Tq * W64);                         // [2][Tk]
  int* s_dur = reinterpret_cast<int*>(s_row + 2 * Tk);                                        // [Tk]
  short* s_path = reinterpret_cast<short*>(s_dur + Tk);                                       // [Tq]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T1 = min(out_lens[b], Tq), T2 = min(in_lens[b], Tk);
  float* O = opt + (long)b * Tq * Tk;
  const float* A = O;                       // log(attn), produced by mas_log_kernel
  for (int j = tid; j < Tk; j += 256) s_dur[j] = 0;
  if (T1 <= 0 || T2 <= 0) {
    for (long e = tid; e < (long)Tq * Tk; e += 256) O[e] = 0.f;
    for (int j = tid; j < Tk; j += 256) dur[(long)b * Tk + j] = 0.f;
    return;
  }
  for (int j = tid; j < T2; j += 256) s_row[j] = j == 0 ? A[0] : -INFINITY;
  float nx[MAS_CH][NJ];
  auto load_chunk = [&](int i0) {
#pragma unroll
    for (int r = 0; r < MAS_CH; ++r)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int i = i0 + r, j = tid + 256 * jj;
        nx[r][jj] = (i < T1 && j < T2) ? A[(long)i * Tk + j] : 0.f;
      }
  };
  load_chunk(1);
  __syncthreads();
  for (int i0 = 1; i0 < T1; i0 += MAS_CH) {
    float la[MAS_CH][NJ];
#pragma unroll
    for (int r = 0; r < MAS_CH; ++r)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) la[r][jj] = nx[r][jj];
    if (i0 + MAS_CH < T1) load_chunk(i0 + MAS_CH);              // in flight while this chunk's 8 DP rows run
#pragma unroll
    for (int r = 0; r < MAS_CH; ++r) {
      const int i = i0 + r;
      if (i >= T1) break;
      const float* prev = s_row + ((i - 1) & 1) * Tk;
      float* cur = s_row + (i & 1) * Tk;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = tid + 256 * jj;
        bool from_left = false;
        if (j < T2) {
          float pl = prev[j];
          if (j >= 1 && prev[j - 1] >= pl) { pl = prev[j - 1]; from_left = true; }
          cur[j] = la[r][jj] + pl;
        }
        const unsigned long long m = __ballot(from_left);
        const int word = wave + 4 * jj;
        if (lane == 0 && word < W64) s_bits[(size_t)i * W64 + word] = m;
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    int curj = T2 - 1;
    for (int i = T1 - 1; i >= 0; --i) {
      s_path[i] = (short)curj;
      if (i > 0 && ((s_bits[(size_t)i * W64 + (curj >> 6)] >> (curj & 63)) & 1ULL)) curj -= 1;
    }
  }
  __syncthreads();
  // hard alignment: one at (i, path[i]) for i < T1, plus the reference's extra opt[0, 0] = 1 (prev_ind[0, :] = 0)
  for (long e = tid; e < (long)Tq * Tk; e += 256) {
    const int i = (int)(e / Tk), j = (int)(e - (long)i * Tk);
    O[e] = (i < T1 && (j == s_path[i] || (i == 0 && j == 0))) ? 1.f : 0.f;
  }
  for (int i = tid; i < T1; i += 256) atomicAdd(&s_dur[s_path[i]], 1);
  __syncthreads();
  if (tid == 0 && s_path[0] != 0) s_dur[0] += 1;
  __syncthreads();
  for (int j = tid; j < Tk; j += 256) dur[(long)b * Tk + j] = (float)s_dur[j];
}

// ---------------------------------------------------------------- ForwardSumLoss (model/loss.py:350-377) on the device
// Per utterance: CTC negative log-likelihood of the target 1..K (every text token once, in order) under
// lp[t, c] = log_softmax_c([blank_logprob, a[t, 0..K-1]]) for the T valid frames.  Extended label sequence l' = [0,1,0,2,...,K,0],
// S = 2K+1 states; all labels are distinct, so the skip transition s-2 -> s is open for every odd s.  One workgroup per utterance,
// threads over the states, the T steps in sequence with the previous row in LDS; rows of `a` are prefetched 8 steps ahead.
// Same float32 log-space recursion as torch's ctc_loss (which the reference calls once per utterance from a Python loop).
constexpr int FS_CH = 8;
__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  return m == -INFINITY ? -INFINITY : m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  return m == -INFINITY ? -INFINITY : m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// lse[b,t] = log(exp(blank) + sum_{k<K} exp(a[t,k]))   (wave per row)
__device__ __forceinline__ void fs_row_lse(const float* __restrict__ A, float* __restrict__ s_lse, int T, int K, int Tk, float blank) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  for (int t = wave; t < T; t += nwave) {
    float mx = blank;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, A[(long)t * Tk + k]);
    mx = ctts_wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < K; k += 64) sum += expf(A[(long)t * Tk + k] - mx);
    sum = ctts_wave_sum(sum);
    if (lane == 0) s_lse[t] = mx + logf(sum + expf(blank - mx));
  }
}

template <int NS>   // states per thread: S = 2K+1 <= blockDim.x * NS; blockDim.x = min(1024, roundup(2 Tk + 1, 64))
__global__ __launch_bounds__(1024) void forward_sum_fwd_kernel(const float* __restrict__ attn, const int* __restrict__ in_lens,
                                                               const int* __restrict__ out_lens, float blank, float* __restrict__ lse,
                                                               float* __restrict__ alpha, float* __restrict__ nll, int Tq, int Tk) {
  extern __shared__ float fs_smem[];
  const int SM = 2 * Tk + 1;
  float* s_lse = fs_smem;                 // [Tq]
  float* s_row = fs_smem + Tq;            // [2][SM]
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int T = min(out_lens[b], Tq), K = min(in_lens[b], Tk), S = 2 * K + 1;
  const float* A = attn + (long)b * Tq * Tk;
  float* AL = alpha + (long)b * Tq * SM;
  if (T <= 0 || K <= 0) { if (tid == 0) nll[b] = INFINITY; return; }
  fs_row_lse(A, s_lse, T, K, Tk, blank);
  __syncthreads();
  for (int t = tid; t < T; t += nthr) lse[(long)b * Tq + t] = s_lse[t];
  float nx[FS_CH][NS];
  auto load_chunk = [&](int t0) {
#pragma unroll
    for (int r = 0; r < FS_CH; ++r)
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int t = t0 + r, st = tid + nthr * q;
        nx[r][q] = (t < T && st < S && (st & 1)) ? A[(long)t * Tk + (st >> 1)] : blank;     // raw logit of the state's class
      }
  };
  load_chunk(0);
  for (int t0 = 0; t0 < T; t0 += FS_CH) {
    float cur[FS_CH][NS];
#pragma unroll
    for (int r = 0; r < FS_CH; ++r)
#pragma unroll
      for (int q = 0; q < NS; ++q) cur[r][q] = nx[r][q];
    if (t0 + FS_CH < T) load_chunk(t0 + FS_CH);
#pragma unroll
    for (int r = 0; r < FS_CH; ++r) {
      const int t = t0 + r;
      if (t >= T) break;
      const float* prev = s_row + ((t - 1) & 1) * SM;
      float* now = s_row + (t & 1) * SM;
      const float norm = s_lse[t];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int st = tid + nthr * q;
        if (st < S) {
          float v;
          const float lp = cur[r][q] - norm;
          if (t == 0) v = st < 2 ? lp : -INFINITY;
          else {
            const float a0 = prev[st], a1 = st >= 1 ? prev[st - 1] : -INFINITY;
            const float a2 = ((st & 1) && st >= 3) ? prev[st - 2] : -INFINITY;
            v = lp + lse3(a0, a1, a2);
          }
          now[st] = v;
          AL[(long)t * SM + st] = v;
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    const float* last = s_row + ((T - 1) & 1) * SM;
    nll[b] = -lse2(last[S - 1], S >= 2 ? last[S - 2] : -INFINITY);
  }
}

// grad[b,t,k] = gscale[b] * (softmax prob of class k+1 at frame t - posterior occupancy of state 2k+1 at frame t); 0 outside (T, K)
template <int NS>
__global__ __launch_bounds__(1024) void forward_sum_bwd_kernel(const float* __restrict__ attn, const int* __restrict__ in_lens,
                                                               const int* __restrict__ out_lens, float blank,
                                                               const float* __restrict__ lse, const float* __restrict__ alpha,
                                                               const float* __restrict__ nll, const float* __restrict__ gscale,
                                                               float* __restrict__ grad, int Tq, int Tk) {
  extern __shared__ float fs_smem[];
  const int SM = 2 * Tk + 1;
  float* s_lse = fs_smem;
  float* s_row = fs_smem + Tq;
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int T = min(out_lens[b], Tq), K = min(in_lens[b], Tk), S = 2 * K + 1;
  const float* A = attn + (long)b * Tq * Tk;
  const float* AL = alpha + (long)b * Tq * SM;
  float* G = grad + (long)b * Tq * Tk;
  const float nl = (T > 0 && K > 0) ? nll[b] : INFINITY;
  const float gs = gscale[b];
  const bool dead = !(nl < INFINITY) || gs == 0.f;          // zero_infinity / nothing to propagate
  // rows >= T and classes >= K receive no gradient
  for (long e = tid; e < (long)Tq * Tk; e += nthr) {
    const int t = (int)(e / Tk), k = (int)(e - (long)t * Tk);
    if (dead || t >= T || k >= K) G[e] = 0.f;
  }
  if (dead) return;
  for (int t = tid; t < T; t += nthr) s_lse[t] = lse[(long)b * Tq + t];
  __syncthreads();
  float nxa[FS_CH][NS], nxl[FS_CH][NS];
  auto load_chunk = [&](int c0) {               // chunk element r is frame t = T-1-(c0+r)
#pragma unroll
    for (int r = 0; r < FS_CH; ++r)
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int t = T - 1 - (c0 + r), st = tid + nthr * q;
        const bool ok = t >= 0 && st < S;
        nxa[r][q] = (ok && (st & 1)) ? A[(long)t * Tk + (st >> 1)] : blank;
        nxl[r][q] = ok ? AL[(long)t * SM + st] : -INFINITY;
      }
  };
  load_chunk(0);
  for (int c0 = 0; c0 < T; c0 += FS_CH) {
    float ca[FS_CH][NS], cl[FS_CH][NS];
#pragma unroll
    for (int r = 0; r < FS_CH; ++r)
#pragma unroll
      for (int q = 0; q < NS; ++q) { ca[r][q] = nxa[r][q]; cl[r][q] = nxl[r][q]; }
    if (c0 + FS_CH < T) load_chunk(c0 + FS_CH);
#pragma unroll
    for (int r = 0; r < FS_CH; ++r) {
      const int t = T - 1 - (c0 + r);
      if (t < 0) break;
      const float* nextb = s_row + ((t + 1) & 1) * SM;
      float* now = s_row + (t & 1) * SM;
      const float norm = s_lse[t];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int st = tid + nthr * q;
        if (st < S) {
          const float lp = ca[r][q] - norm;
          float v;
          if (t == T - 1) v = st >= S - 2 ? lp : -INFINITY;
          else {
            const float b0 = nextb[st], b1 = st + 1 < S ? nextb[st + 1] : -INFINITY;
            const float b2 = ((st & 1) && st + 2 < S) ? nextb[st + 2] : -INFINITY;
            v = lp + lse3(b0, b1, b2);
          }
          now[st] = v;
          if (st & 1) {
            const float occ = __expf(cl[r][q] + v - lp + nl);
            G[(long)t * Tk + (st >> 1)] = gs * (__expf(lp) - occ);
          }
        }
      }
      __syncthreads();
    }
  }
}

}  // namespace

extern "C" int ctts_neg_sqdist(const float* q, const float* k, float* out, int B, int Tq, int Tk, int C, float temp, void* stream) {
  CTTS_REQUIRE(q && k && out && C > 0 && (size_t)64 * (C + 1) * 4 <= 64 * 1024, "ctts_neg_sqdist: bad arguments (C too large for the LDS tile)");
  if (B == 0 || Tq == 0 || Tk == 0) return 0;
  dim3 grid((Tq + 63) / 64, (Tk + 63) / 64, B);
  hipLaunchKernelGGL(neg_sqdist_kernel, grid, dim3(256), (size_t)64 * (C + 1) * sizeof(float), (hipStream_t)stream, q, k, out, Tq, Tk,
                     C, temp);
  CTTS_CHECK_LAUNCH("ctts_neg_sqdist");
  return 0;
}

extern "C" int ctts_mas(const float* attn, const int32_t* in_lens, const int32_t* out_lens, float* opt, float* dur,
                        uint8_t* back, int B, int Tq, int Tk, void* stream) {
  CTTS_REQUIRE(attn && in_lens && out_lens && opt && dur && back, "ctts_mas: null pointer");
  CTTS_REQUIRE((size_t)2 * Tk * 4 <= 64 * 1024, "ctts_mas: Tk=%d too large for the LDS row buffers", Tk);
  if (B == 0) return 0;
  const size_t lds = (size_t)Tq * ((Tk + 63) / 64) * 8 + (size_t)2 * Tk * 4 + (size_t)Tk * 4 + (size_t)Tq * 2;
  if (lds <= 60 * 1024 && Tk <= 512 && Tk < 32768) {       // back-pointer bits fit in LDS: fast path
    const long total = (long)B * Tq * Tk;
    hipLaunchKernelGGL(mas_log_kernel, dim3((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, attn, opt, total);
    if (Tk <= 256)
      hipLaunchKernelGGL((mas_lds_kernel<1>), dim3(B), dim3(256), lds, (hipStream_t)stream, in_lens, out_lens, opt, dur, Tq, Tk);
    else
      hipLaunchKernelGGL((mas_lds_kernel<2>), dim3(B), dim3(256), lds, (hipStream_t)stream, in_lens, out_lens, opt, dur, Tq, Tk);
  } else {
    hipLaunchKernelGGL(mas_kernel, dim3(B), dim3(256), (size_t)2 * Tk * sizeof(float), (hipStream_t)stream, attn, in_lens, out_lens,
                       opt, dur, back, Tq, Tk);
  }
  CTTS_CHECK_LAUNCH("ctts_mas");
  return 0;
}

static int fs_threads(int Tk) { const int S = 2 * Tk + 1; return S >= 1024 ? 1024 : (S + 63) / 64 * 64; }
static int fs_ns(int Tk) { return (2 * Tk + 1 + fs_threads(Tk) - 1) / fs_threads(Tk); }

extern "C" int ctts_forward_sum_fwd(const float* attn_logprob, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob,
                                    float* lse, float* alpha, float* nll, int B, int Tq, int Tk, void* stream) {
  CTTS_REQUIRE(attn_logprob && in_lens && out_lens && lse && alpha && nll && Tq > 0 && Tk > 0, "ctts_forward_sum_fwd: bad arguments");
  const size_t lds = ((size_t)Tq + 2 * (2 * (size_t)Tk + 1)) * sizeof(float);
  CTTS_REQUIRE(lds <= 60 * 1024 && fs_ns(Tk) <= 2, "ctts_forward_sum_fwd: Tq=%d / Tk=%d beyond the LDS row buffers (Tk <= 1023)", Tq, Tk);
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (fs_ns(Tk) == 1)
    hipLaunchKernelGGL((forward_sum_fwd_kernel<1>), dim3(B), dim3(fs_threads(Tk)), lds, st, attn_logprob, in_lens, out_lens,
                       blank_logprob, lse, alpha, nll, Tq, Tk);
  else
    hipLaunchKernelGGL((forward_sum_fwd_kernel<2>), dim3(B), dim3(fs_threads(Tk)), lds, st, attn_logprob, in_lens, out_lens,
                       blank_logprob, lse, alpha, nll, Tq, Tk);
  CTTS_CHECK_LAUNCH("ctts_forward_sum_fwd");
  return 0;
}

extern "C" int ctts_forward_sum_bwd(const float* attn_logprob, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob,
                                    const float* lse, const float* alpha, const float* nll, const float* gscale, float* grad, int B,
                                    int Tq, int Tk, void* stream) {
  CTTS_REQUIRE(attn_logprob && in_lens && out_lens && lse && alpha && nll && gscale && grad && Tq > 0 && Tk > 0,
               "ctts_forward_sum_bwd: bad arguments");
  const size_t lds = ((size_t)Tq + 2 * (2 * (size_t)Tk + 1)) * sizeof(float);
  CTTS_REQUIRE(lds <= 60 * 1024 && fs_ns(Tk) <= 2, "ctts_forward_sum_bwd: Tq=%d / Tk=%d beyond the LDS row buffers (Tk <= 1023)", Tq, Tk);
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (fs_ns(Tk) == 1)
    hipLaunchKernelGGL((forward_sum_bwd_kernel<1>), dim3(B), dim3(fs_threads(Tk)), lds, st, attn_logprob, in_lens, out_lens,
                       blank_logprob, lse, alpha, nll, gscale, grad, Tq, Tk);
  else
    hipLaunchKernelGGL((forward_sum_bwd_kernel<2>), dim3(B), dim3(fs_threads(Tk)), lds, st, attn_logprob, in_lens, out_lens,
                       blank_logprob, lse, alpha, nll, gscale, grad, Tq, Tk);
  CTTS_CHECK_LAUNCH("ctts_forward_sum_bwd");
  return 0;
}
