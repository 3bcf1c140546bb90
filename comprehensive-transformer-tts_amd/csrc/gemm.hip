// fp32 MFMA GEMM for gfx950 with implicit-im2col operand views and a fused epilogue.
//
// Tiling: 256 threads = 4 waves (2 x 2); block tile BM x BN x 32, wave tile (BM/2) x (BN/2) as
// (BM/64) x (BN/64) MFMA tiles of v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the
// fp32 roofline on CDNA4).  K is consumed 32 at a time; inside a K-block the MFMA's two
// k-lanes (lane>>5) take k = h*16 + j so that a K-contiguous operand is fetched from LDS as four
// ds_read_b128 per 32-row tile (row stride 36 floats -> conflict-free 16-lane groups).
// Global->LDS staging goes through registers (prefetch of K-block i+1 is issued before the
// MFMAs of block i), which lets the loader apply the conv time-boundary predicate and the
// batch/length limits for free.  blockIdx.x is remapped so that each XCD owns a contiguous
// run of tiles (tiles that share an A panel hit the same L2).
#include "ctts_common.h"
#include "gemm_common.h"
#include <stdlib.h>

namespace {

#ifndef CTTS_BK
#define CTTS_BK 32
#endif
#ifndef CTTS_GEMM_WAVES
#define CTTS_GEMM_WAVES 2
#endif
constexpr int BK = CTTS_BK;       // K elements staged per barrier pair (32 or 64); MFMA sub-blocks are always 32 deep
constexpr int KC_LD = BK + 4;     // row stride of K-contiguous operand tiles: 36 / 68 floats -> conflict-free b128 groups
constexpr int KCH = BK / 4;       // float4 chunks per tile row
static_assert(BK == 32 || BK == 64, "BK must be 32 or 64");

struct ConvView {
  int T, pad, cin;
};

// ---- K-contiguous operand tile: ROWS x BK, element (r,k) = P[r*ld + k]
template <int ROWS, bool CONV>
struct LoaderKC {
  static constexpr int NV = ROWS * BK / 4 / 256;
  const float* base;
  long ld;
  int row0, row_lim, kq;
  bool vec;
  ConvView cv;
  int trow[NV];
  __device__ void init(const float* p, long ld_, int row0_, int row_lim_, bool vec_, ConvView cv_) {
    base = p; ld = ld_; row0 = row0_; row_lim = row_lim_; vec = vec_; cv = cv_;
    kq = (threadIdx.x % KCH) << 2;
    if (CONV) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int gr = row0 + ((threadIdx.x + i * 256) / KCH);
        trow[i] = gr % cv.T;
      }
    }
  }
  __device__ __forceinline__ void load(int k0, int k_end, float4 (&r)[NV]) const {
    const int gk = k0 + kq;
    int tap = 0;
    if (CONV) tap = gk / cv.cin - cv.pad;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int gr = row0 + ((threadIdx.x + i * 256) / KCH);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < row_lim && gk < k_end) {
        const float* p = base + (long)gr * ld + gk;
        bool ok = true;
        if (CONV) { int tt = trow[i] + tap; ok = (tt >= 0) && (tt < cv.T); }
        if (vec && gk + 3 < k_end) {
          if (ok) v = *reinterpret_cast<const float4*>(p);
        } else {
          float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bool okq = (gk + q < k_end);
            if (CONV && okq) { int tt = trow[i] + (gk + q) / cv.cin - cv.pad; okq = (tt >= 0) && (tt < cv.T); }
            if (okq) e[q] = p[q];
          }
          v = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&r)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = (threadIdx.x + i * 256) / KCH;
      *reinterpret_cast<float4*>(s + row * KC_LD + kq) = r[i];
    }
  }
};

// ---- row-contiguous operand tile: BK x COLS, element (k,c) = P[k*ld + c]
template <int COLS, bool CONV>
struct LoaderRC {
  static constexpr int NV = COLS * BK / 4 / 256;
  static constexpr int LD = COLS + 4;
  const float* base;
  long ld;
  int col0, col_lim;
  bool vec;
  ConvView cv;
  int tap[1];
  __device__ void init(const float* p, long ld_, int col0_, int col_lim_, bool vec_, ConvView cv_) {
    base = p; ld = ld_; col0 = col0_; col_lim = col_lim_; vec = vec_; cv = cv_;
    tap[0] = 0;
  }
  __device__ __forceinline__ void load(int k0, int k_end, float4 (&r)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = threadIdx.x + i * 256;
      const int k = f / (COLS / 4);
      const int cq = (f % (COLS / 4)) << 2;
      const int gk = k0 + k, gc = col0 + cq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gk < k_end && gc < col_lim) {
        const float* p = base + (long)gk * ld + gc;
        int trow = 0;
        if (CONV) trow = gk % cv.T;
        if (vec && gc + 3 < col_lim) {
          bool ok = true;
          if (CONV) { int tt = trow + gc / cv.cin - cv.pad; ok = (tt >= 0) && (tt < cv.T); }
          if (ok) v = *reinterpret_cast<const float4*>(p);
        } else {
          float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bool okq = (gc + q < col_lim);
            if (CONV && okq) { int tt = trow + (gc + q) / cv.cin - cv.pad; okq = (tt >= 0) && (tt < cv.T); }
            if (okq) e[q] = p[q];
          }
          v = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&r)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = threadIdx.x + i * 256;
      const int k = f / (COLS / 4);
      const int cq = (f % (COLS / 4)) << 2;
      *reinterpret_cast<float4*>(s + k * LD + cq) = r[i];
    }
  }
};


// ---------------------------------------------------------------------------------------------
// Branch-free loaders for the aligned case (VEC): every 16-byte chunk is fetched UNCONDITIONALLY
// from an in-bounds address (invalid chunks read a safe dummy address) and the validity mask is
// applied only when the registers are written to LDS.  Nothing between the global_load and the
// ds_write consumes the loaded value, so hipcc keeps the loads in flight across the MFMA block
// (a select / branch join on the loaded value would force an s_waitcnt vmcnt(0) before the MFMAs).
// Requirements (checked on the host): leading dimensions, batch strides and base pointers are
// multiples of 4 floats and the contiguous extent (K, or M/N for row-contiguous operands) is a
// multiple of 4, so a chunk that starts inside the hard extent lies entirely inside it.
__device__ __forceinline__ float4 mask4(float4 v, int nv) {
  v.x = nv > 0 ? v.x : 0.f; v.y = nv > 1 ? v.y : 0.f; v.z = nv > 2 ? v.z : 0.f; v.w = nv > 3 ? v.w : 0.f;
  return v;
}

template <int ROWS, bool CONV>
struct VLoaderKC {
  static constexpr int NV = ROWS * BK / 4 / 256;
  const float* base; const float* safe;
  long ld;
  int row0, row_lim, kq;
  ConvView cv;
  int trow[NV];
  int nval[NV];
  __device__ void init(const float* p, const float* safe_, long ld_, int row0_, int row_lim_, ConvView cv_) {
    base = p; safe = safe_; ld = ld_; row0 = row0_; row_lim = row_lim_; cv = cv_;
    kq = (threadIdx.x % KCH) << 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int gr = row0 + ((threadIdx.x + i * 256) / KCH);
      trow[i] = CONV ? gr % cv.T : 0;
      nval[i] = 0;
    }
  }
  __device__ __forceinline__ void load(int k0, int k_end, float4 (&r)[NV]) {
    const int gk = k0 + kq;
    int tap = 0;
    if (CONV) tap = gk / cv.cin - cv.pad;
    const int nk = min(4, k_end - gk);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int gr = row0 + ((threadIdx.x + i * 256) / KCH);
      bool ok = (gr < row_lim) && (nk > 0);
      if (CONV) { const int tt = trow[i] + tap; ok = ok && (tt >= 0) && (tt < cv.T); }
      const float* p = ok ? base + (long)gr * ld + gk : safe;
      r[i] = *reinterpret_cast<const float4*>(p);
      nval[i] = ok ? nk : 0;
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&r)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = (threadIdx.x + i * 256) / KCH;
      *reinterpret_cast<float4*>(s + row * KC_LD + kq) = mask4(r[i], nval[i]);
    }
  }
};

template <int COLS, bool CONV>
struct VLoaderRC {
  static constexpr int NV = COLS * BK / 4 / 256;
  static constexpr int LD = COLS + 4;
  const float* base; const float* safe;
  long ld;
  int col0, col_lim;
  ConvView cv;
  int nval[NV];
  __device__ void init(const float* p, const float* safe_, long ld_, int col0_, int col_lim_, ConvView cv_) {
    base = p; safe = safe_; ld = ld_; col0 = col0_; col_lim = col_lim_; cv = cv_;
#pragma unroll
    for (int i = 0; i < NV; ++i) nval[i] = 0;
  }
  __device__ __forceinline__ void load(int k0, int k_end, float4 (&r)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = threadIdx.x + i * 256;
      const int k = f / (COLS / 4);
      const int cq = (f % (COLS / 4)) << 2;
      const int gk = k0 + k, gc = col0 + cq;
      const int nc = min(4, col_lim - gc);
      bool ok = (gk < k_end) && (nc > 0);
      if (CONV) { const int tt = gk % cv.T + gc / cv.cin - cv.pad; ok = ok && (tt >= 0) && (tt < cv.T); }
      const float* p = ok ? base + (long)gk * ld + gc : safe;
      r[i] = *reinterpret_cast<const float4*>(p);
      nval[i] = ok ? nc : 0;
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&r)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = threadIdx.x + i * 256;
      const int k = f / (COLS / 4);
      const int cq = (f % (COLS / 4)) << 2;
      *reinterpret_cast<float4*>(s + k * LD + cq) = mask4(r[i], nval[i]);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Buffer-descriptor loaders (VEC = 3) for the large GEMMs without per-batch length limits.
// The operand is addressed as  SRD(base, 2 GiB window) + 32-bit byte offset.  A masked-out chunk simply
// gets the offset 0x80000000, which the hardware bounds check answers with zeros: no zero page, no select
// on the loaded data, no 64-bit pointer arithmetic - per chunk and K-block the address work is one add and
// (conv / K-tail only) one compare + v_cndmask.  Host guarantees every valid offset is < 2^31 bytes.
typedef unsigned int ctts_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned CTTS_OOB = 0x80000000u;

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
  ctts_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
  return *reinterpret_cast<float4*>(&v);
}

template <int ROWS, bool CONV, bool PARTIAL>
struct BLoaderKC {
  static constexpr int NV = ROWS * BK / 4 / 256;
  unsigned boff[NV];      // byte offset of (row, kq) relative to the descriptor base, or CTTS_OOB
  int trow[NV];
  int kq, tid, T, cin, pad;
  int nk_cur;             // PARTIAL: valid elements of this thread's chunk in the staged K-block (same for all rows)
  __device__ void init(long ld, int row0, int row_lim, ConvView cv, int tid_) {
    tid = tid_; T = cv.T; cin = cv.cin; pad = cv.pad;
    kq = (tid % KCH) << 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int gr = row0 + ((tid + i * 256) / KCH);
      const long e = (long)(CONV ? gr - cv.pad : gr) * ld + kq;      // conv: im2col row starts pad rows earlier
      boff[i] = gr < row_lim ? (unsigned)(e * 4) : CTTS_OOB;
      trow[i] = CONV ? gr % cv.T : 0;
    }
    nk_cur = 4;
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int k0, int k_end, float4 (&r)[NV]) {
    const int gk = k0 + kq;
    const bool kok = gk < k_end;                       // K tail: chunk fully masked, or (PARTIAL) cut at k_end
    if (PARTIAL) nk_cur = min(4, k_end - gk);
    const unsigned koff = (unsigned)k0 * 4u;
    int tap = 0;
    if (CONV) tap = gk / cin - pad;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      bool ok = kok && (boff[i] != CTTS_OOB);
      if (CONV) ok = ok && ((unsigned)(trow[i] + tap) < (unsigned)T);
      const unsigned off = ok ? boff[i] + koff : CTTS_OOB;   // select LAST: a wrapped (negative) boff plus an offset must never look valid
      r[i] = buf_load16(rsrc, off);
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&r)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = (tid + i * 256) / KCH;
      *reinterpret_cast<float4*>(s + row * KC_LD + kq) = PARTIAL ? mask4(r[i], nk_cur) : r[i];
    }
  }
};

template <int COLS, bool CONV, bool PARTIAL>
struct BLoaderRC {
  static constexpr int NV = COLS * BK / 4 / 256;
  static constexpr int LD = COLS + 4;
  unsigned boff[NV];      // byte offset of (k_local row, column chunk) at k0 = 0, or CTTS_OOB
  int ctap[NV];           // CONV: gc / cin - pad (loop invariant)
  unsigned ldb4;          // row stride in bytes
  int tid, T;
  int ncol[NV];           // PARTIAL: valid elements of the chunk along the contiguous dim (loop invariant)
  __device__ void init(long ld, int col0, int col_lim, ConvView cv, int tid_) {
    tid = tid_; T = cv.T; ldb4 = (unsigned)(ld * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      const int k = f / (COLS / 4);
      const int gc = col0 + ((f % (COLS / 4)) << 2);
      const long e = (long)(CONV ? k - cv.pad : k) * ld + gc;
      boff[i] = gc < col_lim ? (unsigned)(e * 4) : CTTS_OOB;
      ctap[i] = CONV ? gc / cv.cin - cv.pad : 0;
      ncol[i] = PARTIAL ? min(4, col_lim - gc) : 4;
    }
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int k0, int k_end, float4 (&r)[NV]) const {
    const unsigned k0off = (unsigned)k0 * ldb4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int gk = k0 + (tid + i * 256) / (COLS / 4);
      unsigned off = boff[i] + k0off;
      bool ok = (boff[i] != CTTS_OOB) && (gk < k_end);
      if (CONV) ok = ok && ((unsigned)(gk % T + ctap[i]) < (unsigned)T);
      r[i] = buf_load16(rsrc, ok ? off : CTTS_OOB);
    }
  }
  __device__ __forceinline__ void store(float* s, const float4 (&r)[NV]) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      *reinterpret_cast<float4*>(s + (f / (COLS / 4)) * LD + ((f % (COLS / 4)) << 2)) = PARTIAL ? mask4(r[i], ncol[i]) : r[i];
    }
  }
};

template <bool KC, int EXT, bool CONV, bool PARTIAL>
struct BLoaderSel { using type = BLoaderKC<EXT, CONV, PARTIAL>; };
template <int EXT, bool CONV, bool PARTIAL>
struct BLoaderSel<false, EXT, CONV, PARTIAL> { using type = BLoaderRC<EXT, CONV, PARTIAL>; };

template <bool KC, int EXT, bool CONV, bool VEC>
struct LoaderSel { using type = LoaderKC<EXT, CONV>; };
template <int EXT, bool CONV>
struct LoaderSel<false, EXT, CONV, false> { using type = LoaderRC<EXT, CONV>; };
template <int EXT, bool CONV>
struct LoaderSel<true, EXT, CONV, true> { using type = VLoaderKC<EXT, CONV>; };
template <int EXT, bool CONV>
struct LoaderSel<false, EXT, CONV, true> { using type = VLoaderRC<EXT, CONV>; };

// fragment fetch for one 32-wide MFMA tile: 16 k-steps, lane (l31, h) gets element k = h*16 + j
template <bool KC, int LD>
__device__ __forceinline__ void fetch_frag(const float* s, int ext0, int l31, int h, int ksub, float (&f)[16]) {
  if (KC) {
    const float4* p = reinterpret_cast<const float4*>(s + (ext0 + l31) * LD + ksub + h * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = p[q];
      f[4 * q + 0] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
  } else {
    const float* p = s + (ksub + h * 16) * LD + ext0 + l31;
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = p[j * LD];
  }
}

template <int BM, int BN, bool A_KC, bool B_KC, bool CONV, bool VEC>
__global__ __launch_bounds__(256, CTTS_GEMM_WAVES) void gemm_kernel(const ctts_gemm_desc d) {
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int A_LD = A_KC ? KC_LD : BM + 4;
  constexpr int B_LD = B_KC ? KC_LD : BN + 4;
  constexpr int A_SZ = A_KC ? BM * KC_LD : BK * (BM + 4);
  constexpr int B_SZ = B_KC ? BN * KC_LD : BK * (BN + 4);
  __shared__ __attribute__((aligned(16))) float smem[A_SZ + B_SZ];
  float* sA = smem;
  float* sB = smem + A_SZ;

  // ---- batch / split decode
  const int z = blockIdx.z;
  const int split = blockIdx.y;                         // K split (grid.y = split_k, 1 when off); z = batch index
  const int z0 = z / d.nb1, z1 = z - z0 * d.nb1;
  int Mv = d.M, Nv = d.N, Kv = d.K;
  if (d.lens) {
    const int L = d.lens[z0];
    if (d.lim_m) Mv = min(Mv, L);
    if (d.lim_n) Nv = min(Nv, L);
    if (d.lim_k) Kv = min(Kv, L);
  }
  // ---- XCD-aware tile mapping (bijective for any grid size)
  const int tiles_n = (d.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  // m-tiles are visited in a scrambled order (odd-prime stride permutation): with padded-row skipping the
  // empty tiles of the short sequences would otherwise all land on the same XCD (static block->XCD map)
  const int tiles_m = nwg / tiles_n;
  int tm = wg / tiles_n;
  {
    const int P = (tiles_m % 37) ? 37 : ((tiles_m % 41) ? 41 : 43);
    tm = (int)(((long)tm * P) % tiles_m);
  }
  const int row0 = tm * BM, col0 = (wg % tiles_n) * BN;
  {
    // "write everything" (split_overwrite) on an unsplit launch with per-batch limits: the part of the batch's [M, N] block outside the
    // limits is written as zero by the tile that covers it (same rule as gemm_buf_kernel)
    const bool zero_outside = d.lens && d.split_overwrite && d.split_k <= 1;
    const bool beyond = row0 >= Mv || col0 >= Nv;
    if (zero_outside && row0 < d.M && col0 < d.N && (beyond || row0 + BM > Mv || col0 + BN > Nv)) {
      float* Cz = d.C + z0 * d.sC0 + z1 * d.sC1;
      const int nrows = min(BM, d.M - row0), ncols = min(BN, d.N - col0);
      for (int e = threadIdx.x; e < nrows * ncols; e += 256) {
        const int r = e / ncols, c = e - r * ncols;
        if (row0 + r >= Mv || col0 + c >= Nv) Cz[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
      }
    }
    if (beyond) return;
  }

  if (A_KC && d.row_lens) {  // whole tile of padded rows -> zeros, no operand traffic, no MFMA
    const int last = min(row0 + BM, Mv) - 1;
    const int b0 = row0 / d.row_T, b1 = last / d.row_T;
    if (b0 == b1 && (row0 - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo) {
      float* Cz = d.C + z0 * d.sC0 + z1 * d.sC1;
      const int ncols = min(BN, Nv - col0), nrows = last - row0 + 1;
      for (int e = threadIdx.x; e < nrows * ncols; e += 256) {
        const int r = e / ncols, c = e - r * ncols;
        Cz[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
        if (d.Z && !d.epi_bwd) d.Z[(long)(row0 + r) * d.ldz + col0 + c] = 0.f;
      }
      return;
    }
  }
  int k_begin = 0, k_end = Kv;
  if (d.split_k > 1) {
    int chunk = ((Kv + d.split_k - 1) / d.split_k + BK - 1) / BK * BK;
    k_begin = split * chunk;
    k_end = min(Kv, k_begin + chunk);
    if (k_begin >= k_end) return;
  }

  const float* Ab = d.A + z0 * d.sA0 + z1 * d.sA1;
  const float* Bb = d.B + z0 * d.sB0 + z1 * d.sB1;
  float* Cb = d.C + z0 * d.sC0 + z1 * d.sC1;
  if (d.split_k > 1) Cb += (long)split * d.M * d.ldc;      // split-K: C is the partial matrix P_split of the workspace (ctts_gemm rewrote the descriptor)
  const float* Asafe = Ab;   // in-bounds, 16-B aligned dummy addresses for masked-out chunks (VEC loaders)
  const float* Bsafe = Bb;
  ConvView cv{d.conv_T, d.conv_pad, d.conv_cin};
  ConvView nocv{1, 0, 1};
  constexpr bool CONV_A = CONV && A_KC;
  constexpr bool CONV_B = CONV && !A_KC && !B_KC;
  if (CONV_A) Ab -= (long)d.conv_pad * d.conv_cin;
  if (CONV_B) Bb -= (long)d.conv_pad * d.conv_cin;

  using LA = typename LoaderSel<A_KC, BM, CONV_A, VEC>::type;
  using LB = typename LoaderSel<B_KC, BN, CONV_B, VEC>::type;
  LA la; LB lb;
  if constexpr (VEC) {
    la.init(Ab, Asafe, d.lda, row0, Mv, CONV_A ? cv : nocv);
    lb.init(Bb, Bsafe, d.ldb, col0, Nv, CONV_B ? cv : nocv);
  } else {
    la.init(Ab, d.lda, row0, Mv, false, CONV_A ? cv : nocv);
    lb.init(Bb, d.ldb, col0, Nv, false, CONV_B ? cv : nocv);
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // weight-gradient reductions run over (b,t) rows: K-blocks that lie entirely in the padding of one
  // sequence contribute exactly zero (dZ is zero there) and are skipped (block-uniform decision)
  constexpr bool KSKIP = !A_KC && !B_KC;
  auto kblock_active = [&](int k0) -> bool {
    if (!KSKIP || !d.row_lens) return true;
    const int lastk = min(k0 + BK, k_end) - 1;
    const int b0 = k0 / d.row_T;
    return !(b0 == lastk / d.row_T && (k0 - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo);
  };
  float4 ra[LA::NV], rb[LB::NV];
  bool act_cur = kblock_active(k_begin);
  if (act_cur) {
    la.load(k_begin, k_end, ra);
    lb.load(k_begin, k_end, rb);
    la.store(sA, ra);
    lb.store(sB, rb);
  }
  __syncthreads();

  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    const bool has_next = (k0 + BK) < k_end;
    const bool act_next = has_next && kblock_active(k0 + BK);
    if (act_next) {
      la.load(k0 + BK, k_end, ra);
      lb.load(k0 + BK, k_end, rb);
    }
    if (act_cur) {
#pragma unroll
      for (int ksub = 0; ksub < BK; ksub += 32) {
        float fa[MT][16], fb[NT][16];
#pragma unroll
        for (int i = 0; i < MT; ++i) fetch_frag<A_KC, A_LD>(sA, wm0 + i * 32, l31, h, ksub, fa[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fetch_frag<B_KC, B_LD>(sB, wn0 + j * 32, l31, h, ksub, fb[j]);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
      }
    }
    if (KSKIP && !act_cur && !act_next) continue;   // nothing staged, nothing to publish: no barrier needed
    __syncthreads();
    if (act_next) {
      la.store(sA, ra);
      lb.store(sB, rb);
    }
    __syncthreads();
    act_cur = act_next;
  }

  gemm_epilogue_auto<MT, NT>(d, acc, Cb, z, row0, col0, wm0, wn0, l31, h, Mv, Nv);
}

template <int BM, int BN, bool A_KC, bool B_KC, bool CONV, bool VEC>
int launch(const ctts_gemm_desc& d, hipStream_t st) {
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  const int nz = d.nb0 * d.nb1, ny = d.split_k > 1 ? d.split_k : 1;
  dim3 grid(tiles, ny, nz);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, A_KC, B_KC, CONV, VEC>), grid, dim3(256), 0, st, d);
  CTTS_CHECK_LAUNCH("ctts_gemm");
  return 0;
}

// Same tiling / barrier structure as gemm_kernel, operands fetched through buffer descriptors.
template <int BM, int BN, bool A_KC, bool B_KC, bool CONV, bool PARTIAL>
__global__ __launch_bounds__(256, CTTS_GEMM_WAVES) void gemm_buf_kernel(const ctts_gemm_desc d) {
  // 4 waves as 2 x 2, or 4 x 1 for the narrow tile (BN = 32: outputs with N <= 32, e.g. the d_head = 32 attention gradients)
  constexpr int WAVES_N = BN >= 64 ? 2 : 1, WAVES_M = 4 / WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MT = WM / 32, NT = WN / 32;
  constexpr int A_LD = A_KC ? KC_LD : BM + 4;
  constexpr int B_LD = B_KC ? KC_LD : BN + 4;
  constexpr int A_SZ = A_KC ? BM * KC_LD : BK * (BM + 4);
  constexpr int B_SZ = B_KC ? BN * KC_LD : BK * (BN + 4);
  __shared__ __attribute__((aligned(16))) float smem[A_SZ + B_SZ];
  float* sA = smem;
  float* sB = smem + A_SZ;
  const int z = blockIdx.z;
  const int split = blockIdx.y;                         // K split (grid.y = split_k, 1 when off); z = batch index
  const int z0 = z / d.nb1, z1 = z - z0 * d.nb1;
  int Mv = d.M, Nv = d.N, Kv = d.K;
  if (PARTIAL && d.lens) {                // per-batch valid lengths (attention over non-padded tokens only)
    const int L = d.lens[z0];
    if (d.lim_m) Mv = min(Mv, L);
    if (d.lim_n) Nv = min(Nv, L);
    if (d.lim_k) Kv = min(Kv, L);
  }
  const int tiles_n = (d.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int wg, tm;
  bool scheduled_active = false;
  if (BM == 64 && A_KC && d.tile_map && d.tile_map != reinterpret_cast<const int32_t*>(1)) {
    // device-built schedule: active m-tiles first, natural workgroup order (round-robin over the XCDs)
    wg = blockIdx.x;
    int r = wg / tiles_n;
    if (d.tile_group_n > 0) {                     // XCD-grouped order (validated on the host): see ctts_gemm_desc.tile_group_n
      const int g = d.tile_group_n, ngroups = tiles_n / g, mways = 8 / ngroups;
      const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
      r = (i / g) * mways + x / ngroups;
      wg = r * tiles_n + (x % ngroups) * g + i % g;
    }
    tm = d.tile_map[1 + r];
    scheduled_active = r < d.tile_map[0];
  } else if (PARTIAL || d.tile_map == reinterpret_cast<const int32_t*>(1)) {
    // plain blockIdx order.  Always for per-batch length limits (attention): the valid tiles of a short utterance are its FIRST
    // rows / columns, and the XCD-contiguous remap below would put all of them on the first XCDs.  Also the CTTS_NATURAL_ORDER knob.
    wg = blockIdx.x;
    tm = wg / tiles_n;
  } else {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int tiles_m = nwg / tiles_n;
    tm = wg / tiles_n;
    const int P = (tiles_m % 37) ? 37 : ((tiles_m % 41) ? 41 : 43);
    tm = (int)(((long)tm * P) % tiles_m);
  }
  const int row0 = tm * BM, col0 = (wg % tiles_n) * BN;
  float* Cb = d.C + z0 * d.sC0 + z1 * d.sC1;
  // overwrite mode of an unsplit batched launch with per-batch limits: everything of the batch's [M, N] block outside the limits is
  // WRITTEN as zero by the tile that covers it (the caller passes an uninitialised C: no fill launch per attention product)
  const bool zero_outside = PARTIAL && d.split_overwrite && d.split_k <= 1;
  if (row0 >= Mv || col0 >= Nv) {
    if (zero_outside && row0 < d.M && col0 < d.N) {
      const int nrows = min(BM, d.M - row0), ncols = min(BN, d.N - col0);
      for (int e = threadIdx.x; e < nrows * ncols; e += 256) {
        const int r = e / ncols, c = e - r * ncols;
        Cb[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
      }
    }
    return;
  }
  if (zero_outside && (row0 + BM > Mv || col0 + BN > Nv)) {      // the part of a straddling tile beyond the limits (the epilogue never stores there)
    const int nrows = min(BM, d.M - row0), ncols = min(BN, d.N - col0);
    for (int e = threadIdx.x; e < nrows * ncols; e += 256) {
      const int r = e / ncols, c = e - r * ncols;
      if (row0 + r >= Mv || col0 + c >= Nv) Cb[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
    }
  }
  if (d.split_k > 1) Cb += (long)split * d.M * d.ldc;      // split-K: C is the partial matrix P_split of the workspace (ctts_gemm rewrote the descriptor)
  if (A_KC && d.row_lens && !scheduled_active) {
    const int last = min(row0 + BM, Mv) - 1;
    const int b0 = row0 / d.row_T, b1 = last / d.row_T;
    if (b0 == b1 && (row0 - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo) {
      const int ncols = min(BN, Nv - col0), nrows = last - row0 + 1;
      for (int e = threadIdx.x; e < nrows * ncols; e += 256) {
        const int r = e / ncols, c = e - r * ncols;
        Cb[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
        if (d.Z && !d.epi_bwd) d.Z[(long)(row0 + r) * d.ldz + col0 + c] = 0.f;
      }
      return;
    }
  }
  int k_begin = 0, k_end = Kv;
  if (d.split_k > 1) {
    int chunk = ((Kv + d.split_k - 1) / d.split_k + BK - 1) / BK * BK;
    k_begin = split * chunk;
    k_end = min(Kv, k_begin + chunk);
    if (k_begin >= k_end) return;
  }
  const float* Ab = d.A + z0 * d.sA0 + z1 * d.sA1;
  const float* Bb = d.B + z0 * d.sB0 + z1 * d.sB1;
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7FFFFFFE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 0x7FFFFFFE, 0x00020000);
  ConvView cv{d.conv_T, d.conv_pad, d.conv_cin};
  ConvView nocv{1, 0, 1};
  constexpr bool CONV_A = CONV && A_KC;
  constexpr bool CONV_B = CONV && !A_KC && !B_KC;
  using LA = typename BLoaderSel<A_KC, BM, CONV_A, PARTIAL>::type;
  using LB = typename BLoaderSel<B_KC, BN, CONV_B, PARTIAL>::type;
  LA la; LB lb;
  la.init(d.lda, row0, Mv, CONV_A ? cv : nocv, threadIdx.x);
  lb.init(d.ldb, col0, Nv, CONV_B ? cv : nocv, threadIdx.x);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr bool KSKIP = !A_KC && !B_KC;
  auto kblock_active = [&](int k0) -> bool {
    if (!KSKIP || !d.row_lens) return true;
    const int lastk = min(k0 + BK, k_end) - 1;
    const int b0 = k0 / d.row_T;
    return !(b0 == lastk / d.row_T && (k0 - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo);
  };
  float4 ra[LA::NV], rb[LB::NV];
  bool act_cur = kblock_active(k_begin);
  if (act_cur) {
    la.load(ra_src, k_begin, k_end, ra);
    lb.load(rb_src, k_begin, k_end, rb);
    la.store(sA, ra);
    lb.store(sB, rb);
  }
  __syncthreads();
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    const bool has_next = (k0 + BK) < k_end;
    const bool act_next = has_next && kblock_active(k0 + BK);
    if (act_next) {
      la.load(ra_src, k0 + BK, k_end, ra);
      lb.load(rb_src, k0 + BK, k_end, rb);
    }
    if (act_cur) {
#pragma unroll
      for (int ksub = 0; ksub < BK; ksub += 32) {
        float fa[MT][16], fb[NT][16];
#pragma unroll
        for (int i = 0; i < MT; ++i) fetch_frag<A_KC, A_LD>(sA, wm0 + i * 32, l31, h, ksub, fa[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fetch_frag<B_KC, B_LD>(sB, wn0 + j * 32, l31, h, ksub, fb[j]);
#ifdef CTTS_GEMM_SETPRIO
        __builtin_amdgcn_s_setprio(CTTS_GEMM_SETPRIO);
#endif
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
#ifdef CTTS_GEMM_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      }
    }
    if (KSKIP && !act_cur && !act_next) continue;
    __syncthreads();
    if (act_next) {
      la.store(sA, ra);
      lb.store(sB, rb);
    }
    __syncthreads();
    act_cur = act_next;
  }
  gemm_epilogue_auto<MT, NT>(d, acc, Cb, z, row0, col0, wm0, wn0, l31, h, Mv, Nv);
}

// ---- fp32 GEMM on the BF16 matrix pipe (NT layout: both operands K-contiguous, conv view on A allowed) --------------------------
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32 on gfx950.  Every fp32 operand is split EXACTLY into three
// bf16 pieces (x = hi + mid + lo, 8 significand bits each, each the round-to-nearest of what is left) once, on its way from the staging registers into LDS, and a
// product is the six cross terms  hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi  accumulated in fp32; the dropped terms are
// <= 2^-24 of the product in the worst case - ONE fp32 rounding - and 2^-29 in the median (|mid| <= 2^-8 |hi|, |lo| <= 2^-16 |hi|;
// tests/test_bf16_split_cpu.py; with truncated pieces the worst case was 2^-21),
// i.e. the products are fp32-class (measured: max error against float64 1.36e-5 where an fp32 FMA chain has 1.63e-5,
// tools/ubench/bf16x_split_gemm.hip; tests/test_kernels_gpu.py compares this kernel with float64 and with the fp32-MFMA kernels).
// Everything else is the tile kernel above: buffer loads with the im2col view, register staging, ONE LDS stage and two barriers per
// K-block (two workgroups per CU alternate), the same epilogues, the same zero-fill of wholly padded row tiles.  LDS: per operand three
// planes of [128 rows][32 bf16] = 64-byte rows with XOR-swizzled 16-byte chunks (conflict-free both ways) = 49,152 B per
// workgroup (LDS would take three per CU; the kernel's 188 VGPRs keep it at two waves per SIMD = two workgroups, which measured faster
// than three at 168 VGPRs with a spill).  The shader clock under this kernel is 1.8 - 1.9 GHz (2.39 under the fp32-MFMA kernels): DESIGN.md.
constexpr int X6_PROW = 64, X6_PLANE = 128 * X6_PROW;      // 64-byte rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3 (see x6_chunk)
// physical 16-byte chunk of logical chunk c (0..3) in `row`: staging writes (8 lanes per row, 4 aligned rows per 256-byte beat) and
// fragment reads (16 consecutive rows, one chunk) both touch every bank exactly once per beat; no padding
__device__ __forceinline__ int x6_chunk(int row, int c) { return c ^ ((row >> 2) & 3); }
typedef __bf16 x6_bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 x6_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x6_floatx2 __attribute__((ext_vector_type(2)));
// two floats -> their bf16 roundings (to nearest even: v_cvt_pk_bf16_f32), packed low | high, and the exact remainders in place
__device__ __forceinline__ unsigned x6_round_pair(float& a, float& b) {
  const x6_floatx2 v = {a, b};
  const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_bf16x2));
  a -= __uint_as_float(p << 16);                       // exact: a number minus its rounding to fewer bits
  b -= __uint_as_float(p & 0xFFFF0000u);
  return p;
}

__device__ __forceinline__ void x6_split_store(const float4 v, unsigned char* base) {      // 4 consecutive-K floats -> 8 bytes per plane
  float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
  uint2 hi, mid, lo;
  hi.x = x6_round_pair(x0, x1);  hi.y = x6_round_pair(x2, x3);          // x = hi + r1
  mid.x = x6_round_pair(x0, x1); mid.y = x6_round_pair(x2, x3);         // r1 = mid + r2
  lo.x = x6_round_pair(x0, x1);  lo.y = x6_round_pair(x2, x3);          // r2 = lo exactly (<= 8 significant bits are left)
  *reinterpret_cast<uint2*>(base) = hi;
  *reinterpret_cast<uint2*>(base + X6_PLANE) = mid;
  *reinterpret_cast<uint2*>(base + 2 * X6_PLANE) = lo;
}

__device__ __forceinline__ floatx16 x6_mma(const ctts_u32x4 a, const ctts_u32x4 b, const floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x6_bf16x8, a), __builtin_bit_cast(x6_bf16x8, b), c, 0, 0, 0);
}

template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(const ctts_gemm_desc d) {
  static_assert(BK == 32, "the bf16 planes hold 32-deep K-blocks");
  constexpr int BM = 128, BN = 128, MT = 2, NT = 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[6 * X6_PLANE];       // A hi | mid | lo, B hi | mid | lo
  const int Mv = d.M, Nv = d.N, Kv = d.K;
  const int tiles_n = d.N / BN, nwg = gridDim.x;
  int wg, tm;
  if (d.tile_map == reinterpret_cast<const int32_t*>(1)) {
    wg = blockIdx.x; tm = wg / tiles_n;
  } else {          // XCD-contiguous remap + scrambled m order, as the tile kernel without a device-built schedule
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int tiles_m = nwg / tiles_n;
    tm = wg / tiles_n;
    const int P = (tiles_m % 37) ? 37 : ((tiles_m % 41) ? 41 : 43);
    tm = (int)(((long)tm * P) % tiles_m);
  }
  const int row0 = tm * BM, col0 = (wg % tiles_n) * BN;
  float* Cb = d.C;
  if (d.row_lens) {                       // a tile that lies wholly in the padding of one utterance: defined as zero, no operand touched
    const int last = min(row0 + BM, Mv) - 1;
    const int b0 = row0 / d.row_T, b1 = last / d.row_T;
    if (b0 == b1 && (row0 - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo) {
      const int nrows = last - row0 + 1;
      for (int e = threadIdx.x; e < nrows * BN; e += 256) {
        const int r = e / BN, c = e - r * BN;
        Cb[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
        if (d.Z && !d.epi_bwd) d.Z[(long)(row0 + r) * d.ldz + col0 + c] = 0.f;
      }
      return;
    }
  }
  // the zero rule has 64-row granularity in the other kernels (and in the tile schedules built for them): when only the UPPER 64 rows of
  // this tile lie wholly in padding, the two waves that own them write zeros instead of their epilogue
  bool pad_hi = false;
  if (d.row_lens && row0 + 64 < Mv) {
    const int first = row0 + 64, last = min(row0 + BM, Mv) - 1;
    const int b0 = first / d.row_T;
    pad_hi = b0 == last / d.row_T && (first - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo;
  }
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, 0x7FFFFFFE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, 0x7FFFFFFE, 0x00020000);
  ConvView cv{d.conv_T, d.conv_pad, d.conv_cin};
  ConvView nocv{1, 0, 1};
  BLoaderKC<BM, CONV, false> la;
  BLoaderKC<BN, false, false> lb;
  la.init(d.lda, row0, Mv, CONV ? cv : nocv, threadIdx.x);
  lb.init(d.ldb, col0, Nv, nocv, threadIdx.x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[4], rb[4];
  // staging: float4 i of this thread is row (tid + 256 i) / 8 = tid / 8 + 32 i, K offset (tid % 8) * 4 (BLoaderKC's chunking)
  // (rows 32 i apart share (row >> 2) & 3, so one swizzled offset serves all four staging rows / both MFMA row tiles)
  const int srow = threadIdx.x >> 3, sc8 = threadIdx.x & 7;
  unsigned char* st_a = smem + srow * X6_PROW + x6_chunk(srow, sc8 >> 1) * 16 + (sc8 & 1) * 8;
  unsigned char* st_b = st_a + 3 * X6_PLANE;
  // fragment of a 32-row MFMA tile: row l31, the 8 consecutive K values from h * 8 of the 16-deep step ks = logical chunk ks * 2 + h
  const unsigned char* fr_a = smem + (wm0 + l31) * X6_PROW;
  const unsigned char* fr_b = smem + 3 * X6_PLANE + (wn0 + l31) * X6_PROW;
  const int fc0 = x6_chunk(l31, h) * 16, fc1 = x6_chunk(l31, 2 + h) * 16;      // byte offsets of the fragment chunk for ks = 0, 1
  la.load(ra_src, 0, Kv, ra);
  lb.load(rb_src, 0, Kv, rb);
#pragma unroll
  for (int i = 0; i < 4; ++i) { x6_split_store(ra[i], st_a + 32 * i * X6_PROW); x6_split_store(rb[i], st_b + 32 * i * X6_PROW); }
  __syncthreads();
  for (int k0 = 0; k0 < Kv; k0 += BK) {
    {                                      // unconditional (the last block is fetched again): keeps the staging registers in registers
      const int kn = k0 + BK < Kv ? k0 + BK : k0;
      la.load(ra_src, kn, Kv, ra);
      lb.load(rb_src, kn, Kv, rb);
    }
    // all fragments of the K-block first (one exposed LDS round trip per block instead of one per 16-deep step), then the 48 MFMAs
    ctts_u32x4 fa[2][MT][3], fb[2][NT][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[ks][i][p] = *reinterpret_cast<const ctts_u32x4*>(fr_a + p * X6_PLANE + 32 * i * X6_PROW + (ks ? fc1 : fc0));
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) fb[ks][j][p] = *reinterpret_cast<const ctts_u32x4*>(fr_b + p * X6_PLANE + 32 * j * X6_PROW + (ks ? fc1 : fc0));
    }
    // term-major order: consecutive MFMAs go to DIFFERENT accumulators (a chain of six on one accumulator waits for each result);
    // smallest terms first
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = x6_mma(fa[ks][i][PA[t]], fb[ks][j][PB[t]], acc[i][j]);
      }
    __syncthreads();                       // every wave has read the tile
#pragma unroll
    for (int i = 0; i < 4; ++i) { x6_split_store(ra[i], st_a + 32 * i * X6_PROW); x6_split_store(rb[i], st_b + 32 * i * X6_PROW); }
    __syncthreads();
  }
  if (pad_hi && wm0 == 64) {
    const int nrows = min(row0 + BM, Mv) - (row0 + 64);
    for (int e = lane; e < nrows * 64; e += 64) {
      const int r = e >> 6, c = e & 63;
      Cb[(long)(row0 + 64 + r) * d.ldc + col0 + wn0 + c] = 0.f;
      if (d.Z && !d.epi_bwd) d.Z[(long)(row0 + 64 + r) * d.ldz + col0 + wn0 + c] = 0.f;
    }
    return;
  }
  gemm_epilogue_auto<MT, NT>(d, acc, Cb, 0, row0, col0, wm0, wn0, l31, h, Mv, Nv);
}

// ---- the same arithmetic for WEIGHT GRADIENTS (TN layout: C[m][n] = sum_k A[k][m] B[k][n], both operands reduction-major, optional
// im2col view on B).  A bf16 MFMA fragment is 8 consecutive k of one row, which in memory lie a whole row apart: every thread owns ONE
// column (m or n) of the tile and reads 8 consecutive k of it with 8 dword loads (a wave still reads 256 contiguous bytes per k), splits
// the 8 values and stores 16 bytes per plane - the same LDS image as the NT kernel, so fragments, MFMA sequence and epilogue are shared.
// K-blocks that lie wholly in the padding of one utterance are skipped (their dZ rows are zero by construction); grid.y = split_k with
// ctts_gemm's ordered partial matrices.
template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_x6tn_kernel(const ctts_gemm_desc d) {
  static_assert(BK == 32, "the bf16 planes hold 32-deep K-blocks");
  constexpr int BM = 128, BN = 128, MT = 2, NT = 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[6 * X6_PLANE];
  const int Mv = d.M, Nv = d.N, Kv = d.K;
  const int tiles_n = d.N / BN, nwg = gridDim.x, split = blockIdx.y;
  int wg, tm;
  {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    tm = wg / tiles_n;
  }
  const int row0 = tm * BM, col0 = (wg % tiles_n) * BN;
  float* Cb = d.C;
  if (d.split_k > 1) Cb += (long)split * d.M * d.ldc;        // partial matrix P_split (ctts_gemm rewrote the descriptor)
  int k_begin = 0, k_end = Kv;
  if (d.split_k > 1) {
    const int chunk = ((Kv + d.split_k - 1) / d.split_k + BK - 1) / BK * BK;
    k_begin = split * chunk;
    k_end = min(Kv, k_begin + chunk);
    if (k_begin >= k_end) return;
  }
  auto kblock_active = [&](int k0) -> bool {
    if (!d.row_lens) return true;
    const int lastk = min(k0 + BK, k_end) - 1;
    const int b0 = k0 / d.row_T;
    return !(b0 == lastk / d.row_T && (k0 - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo);
  };
  auto next_active = [&](int k0) -> int { while (k0 < k_end && !kblock_active(k0)) k0 += BK; return k0; };
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, 0x7FFFFFFE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, 0x7FFFFFFE, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  // staging: thread -> column `scol` of both tiles, k octets og and og + 2 of the 32-deep block
  const int scol = threadIdx.x & 127, og = threadIdx.x >> 7;
  const unsigned a_col = (unsigned)(row0 + scol) * 4u, b_col = (unsigned)(col0 + scol) * 4u;
  const unsigned lda4 = (unsigned)(d.lda * 4), ldb4 = (unsigned)(d.ldb * 4);
  const int tapb = CONV ? (col0 + scol) / d.conv_cin - d.conv_pad : 0;      // im2col column = (tap, channel): the row shift of this column
  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float va[2][8], vb[2][8];
  auto gload = [&](int k0) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int kk0 = k0 + (og + 2 * o) * 8;
      const int t0 = CONV ? kk0 % d.conv_T + tapb : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = kk0 + j;
        const bool ok = k < k_end;
        const unsigned offa = ok ? (unsigned)k * lda4 + a_col : CTTS_OOB;
        bool okb = ok;
        if (CONV) okb = okb && ((unsigned)(t0 + j) < (unsigned)d.conv_T);
        const unsigned offb = okb ? (unsigned)(CONV ? k - d.conv_pad : k) * ldb4 + b_col : CTTS_OOB;
        va[o][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra_src, offa, 0, 0));
        vb[o][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_src, offb, 0, 0));
      }
    }
  };
  auto split_store8 = [&](const float (&xin)[8], unsigned char* base) {     // 8 consecutive k of one LDS row -> 16 bytes per plane
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = xin[i];
    ctts_u32x4 hi, mid, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) hi[q] = x6_round_pair(x[2 * q], x[2 * q + 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) mid[q] = x6_round_pair(x[2 * q], x[2 * q + 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) lo[q] = x6_round_pair(x[2 * q], x[2 * q + 1]);
    *reinterpret_cast<ctts_u32x4*>(base) = hi;
    *reinterpret_cast<ctts_u32x4*>(base + X6_PLANE) = mid;
    *reinterpret_cast<ctts_u32x4*>(base + 2 * X6_PLANE) = lo;
  };
  auto lstore = [&]() {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      unsigned char* pa = smem + scol * X6_PROW + x6_chunk(scol, og + 2 * o) * 16;
      split_store8(va[o], pa);
      split_store8(vb[o], pa + 3 * X6_PLANE);
    }
  };
  const unsigned char* fr_a = smem + (wm0 + l31) * X6_PROW;
  const unsigned char* fr_b = smem + 3 * X6_PLANE + (wn0 + l31) * X6_PROW;
  const int fc0 = x6_chunk(l31, h) * 16, fc1 = x6_chunk(l31, 2 + h) * 16;
  int k_cur = next_active(k_begin);
  if (k_cur < k_end) {
    gload(k_cur);
    lstore();
  }
  __syncthreads();
  while (k_cur < k_end) {
    const int k_nxt = next_active(k_cur + BK);
    gload(k_nxt < k_end ? k_nxt : k_cur);                      // unconditional: keeps the staging registers out of scratch
    ctts_u32x4 fa[2][MT][3], fb[2][NT][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[ks][i][p] = *reinterpret_cast<const ctts_u32x4*>(fr_a + p * X6_PLANE + 32 * i * X6_PROW + (ks ? fc1 : fc0));
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) fb[ks][j][p] = *reinterpret_cast<const ctts_u32x4*>(fr_b + p * X6_PLANE + 32 * j * X6_PROW + (ks ? fc1 : fc0));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = x6_mma(fa[ks][i][PA[t]], fb[ks][j][PB[t]], acc[i][j]);
      }
    __syncthreads();
    lstore();
    __syncthreads();
    k_cur = k_nxt;
  }
  gemm_epilogue_auto<MT, NT>(d, acc, Cb, 0, row0, col0, wm0, wn0, l31, h, Mv, Nv);
}

// Under-filled launches (few output tiles, long reduction: the 2,048-row phoneme-level layers give 128 tiles of 64 x 64 on 256 CUs, each
// a chain of K / 32 staged K-blocks on one workgroup per CU): 32 x 64 tiles - twice the workgroups - and the reduction split in two INSIDE
// the workgroup.  Waves 0,1 (32 columns each) take the even 32-deep K-blocks, waves 2,3 the odd ones: one load / barrier round per 64
// of K, half as many rounds per workgroup, every SIMD of the chip in use.  The halves are added through LDS in a fixed order and waves
// 0,1 run the ordinary epilogue: any epilogue works (no atomics, no zero fill).  Padded 32-row tiles are zero-filled and skipped.
template <bool A_KC, bool B_KC, bool CONV>
__global__ __launch_bounds__(256, CTTS_GEMM_WAVES) void gemm_buf_k2_kernel(const ctts_gemm_desc d) {
  constexpr int BM = 32, BN = 64;
  constexpr int A_LD = A_KC ? KC_LD : BM + 4;
  constexpr int B_LD = B_KC ? KC_LD : BN + 4;
  constexpr int A_SZ = A_KC ? BM * KC_LD : BK * (BM + 4);
  constexpr int B_SZ = B_KC ? BN * KC_LD : BK * (BN + 4);
  static_assert(BK == 32, "the two-group kernel stages two 32-deep K-blocks per round");
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];       // [group][A | B]; reused for the final reduction
  const int z = blockIdx.z, split = blockIdx.y;
  const int z0 = z / d.nb1, z1 = z - z0 * d.nb1;
  const int Mv = d.M, Nv = d.N, Kv = d.K;
  const int tiles_n = (d.N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n;
  const int row0 = tm * BM, col0 = (blockIdx.x - tm * tiles_n) * BN;
  if (row0 >= Mv || col0 >= Nv) return;
  float* Cb = d.C + z0 * d.sC0 + z1 * d.sC1;
  if (d.split_k > 1) Cb += (long)split * d.M * d.ldc;      // split-K: C is the partial matrix P_split of the workspace
  if (A_KC && d.row_lens) {
    const int last = min(row0 + BM, Mv) - 1;
    const int b0 = row0 / d.row_T, b1 = last / d.row_T;
    if (b0 == b1 && (row0 - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo) {
      const int ncols = min(BN, Nv - col0), nrows = last - row0 + 1;
      for (int e = threadIdx.x; e < nrows * ncols; e += 256) {
        const int r = e / ncols, c = e - r * ncols;
        Cb[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
        if (d.Z && !d.epi_bwd) d.Z[(long)(row0 + r) * d.ldz + col0 + c] = 0.f;
      }
      return;
    }
  }
  int k_begin = 0, k_end = Kv;
  if (d.split_k > 1) {
    int chunk = ((Kv + d.split_k - 1) / d.split_k + 2 * BK - 1) / (2 * BK) * (2 * BK);
    k_begin = split * chunk;
    k_end = min(Kv, k_begin + chunk);
    if (k_begin >= k_end) return;
  }
  const float* Ab = d.A + z0 * d.sA0 + z1 * d.sA1;
  const float* Bb = d.B + z0 * d.sB0 + z1 * d.sB1;
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7FFFFFFE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 0x7FFFFFFE, 0x00020000);
  ConvView cv{d.conv_T, d.conv_pad, d.conv_cin};
  ConvView nocv{1, 0, 1};
  constexpr bool CONV_A = CONV && A_KC;
  constexpr bool CONV_B = CONV && !A_KC && !B_KC;
  using LA = typename BLoaderSel<A_KC, BM, CONV_A, false>::type;
  using LB = typename BLoaderSel<B_KC, BN, CONV_B, false>::type;
  LA la; LB lb;
  la.init(d.lda, row0, Mv, CONV_A ? cv : nocv, threadIdx.x);
  lb.init(d.ldb, col0, Nv, CONV_B ? cv : nocv, threadIdx.x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int grp = wave >> 1, wn0 = (wave & 1) * 32;
  float* sA = smem + grp * (A_SZ + B_SZ);
  float* sB = sA + A_SZ;
  floatx16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  float4 ra[2][LA::NV], rb[2][LB::NV];
  auto load_round = [&](int k0) {                       // both 32-deep K-blocks of the round (the second may lie beyond k_end: zeros)
    la.load(ra_src, k0, k_end, ra[0]);      lb.load(rb_src, k0, k_end, rb[0]);
    la.load(ra_src, k0 + BK, k_end, ra[1]); lb.load(rb_src, k0 + BK, k_end, rb[1]);
  };
  auto store_round = [&]() {
    la.store(smem, ra[0]);                  lb.store(smem + A_SZ, rb[0]);
    la.store(smem + A_SZ + B_SZ, ra[1]);    lb.store(smem + 2 * A_SZ + B_SZ, rb[1]);
  };
  load_round(k_begin);
  store_round();
  __syncthreads();
  for (int k0 = k_begin; k0 < k_end; k0 += 2 * BK) {
    const bool has_next = (k0 + 2 * BK) < k_end;
    if (has_next) load_round(k0 + 2 * BK);
    {
      float fa[16], fb[16];
      fetch_frag<A_KC, A_LD>(sA, 0, l31, h, 0, fa);
      fetch_frag<B_KC, B_LD>(sB, wn0, l31, h, 0, fb);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc[0][0], 0, 0, 0);
    }
    __syncthreads();
    if (has_next) store_round();
    __syncthreads();
  }
  // waves 2,3 hand their half over through LDS (the operand tiles are dead: the loop's last barrier is behind every fragment read)
  float* red = smem + (wave & 1) * 1024 + lane;
  if (grp == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[r * 64] = acc[0][0][r];
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += red[r * 64];
  }
  if (grp == 0) gemm_epilogue_auto<1, 1>(d, acc, Cb, z, row0, col0, 0, wn0, l31, h, Mv, Nv);
}

template <bool A_KC, bool B_KC, bool CONV>
int launch_buf_k2(const ctts_gemm_desc& d, hipStream_t st) {
  const int tiles = ((d.M + 31) / 32) * ((d.N + 63) / 64);
  const int nz = d.nb0 * d.nb1, ny = d.split_k > 1 ? d.split_k : 1;
  hipLaunchKernelGGL((gemm_buf_k2_kernel<A_KC, B_KC, CONV>), dim3(tiles, ny, nz), dim3(256), 0, st, d);
  CTTS_CHECK_LAUNCH("ctts_gemm(buf,k2)");
  return 0;
}

int dispatch_buf_k2(const ctts_gemm_desc& d, hipStream_t st) {
  const bool conv = d.conv_T > 0;
  if (d.a_kc && d.b_kc) return conv ? launch_buf_k2<true, true, true>(d, st) : launch_buf_k2<true, true, false>(d, st);
  if (d.a_kc && !d.b_kc) return conv ? launch_buf_k2<true, false, true>(d, st) : launch_buf_k2<true, false, false>(d, st);
  if (!d.a_kc && !d.b_kc) return conv ? launch_buf_k2<false, false, true>(d, st) : launch_buf_k2<false, false, false>(d, st);
  ctts_set_error("ctts_gemm: layout a_kc=0,b_kc=1 is not instantiated");
  return -1;
}

template <int BM, int BN, bool A_KC, bool B_KC, bool CONV>
int launch_buf(const ctts_gemm_desc& d, hipStream_t st) {
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  const int nz = d.nb0 * d.nb1, ny = d.split_k > 1 ? d.split_k : 1;
  if (d.lens && (d.lim_m || d.lim_n || d.lim_k)) {
    if constexpr (!CONV) {
      hipLaunchKernelGGL((gemm_buf_kernel<BM, BN, A_KC, B_KC, false, true>), dim3(tiles, ny, nz), dim3(256), 0, st, d);
      CTTS_CHECK_LAUNCH("ctts_gemm(buf,partial)");
      return 0;
    } else {
      ctts_set_error("ctts_gemm: per-batch length limits together with a conv view are not supported");
      return -1;
    }
  }
  hipLaunchKernelGGL((gemm_buf_kernel<BM, BN, A_KC, B_KC, CONV, false>), dim3(tiles, ny, nz), dim3(256), 0, st, d);
  CTTS_CHECK_LAUNCH("ctts_gemm(buf)");
  return 0;
}

template <int BM, int BN>
int dispatch_buf(const ctts_gemm_desc& d, hipStream_t st) {
  const bool conv = d.conv_T > 0;
  if (d.a_kc && d.b_kc) return conv ? launch_buf<BM, BN, true, true, true>(d, st) : launch_buf<BM, BN, true, true, false>(d, st);
  if (d.a_kc && !d.b_kc) return conv ? launch_buf<BM, BN, true, false, true>(d, st) : launch_buf<BM, BN, true, false, false>(d, st);
  if (!d.a_kc && !d.b_kc) return conv ? launch_buf<BM, BN, false, false, true>(d, st) : launch_buf<BM, BN, false, false, false>(d, st);
  ctts_set_error("ctts_gemm: layout a_kc=0,b_kc=1 is not instantiated");
  return -1;
}

// 128 x 32 tiles (4 x 1 waves) for outputs at most 32 columns wide: with 64 x 64 tiles half of every workgroup's MFMAs would multiply
// zero columns.  The three products of the conformer's attention backward that consume dS ([T, 32] = [T, T] x [T, 32] per (b, h),
// ctts_relmha_bwd) are the users.  No conv views.
int dispatch_buf_narrow(const ctts_gemm_desc& d, hipStream_t st) {
  if (d.a_kc && d.b_kc) return launch_buf<128, 32, true, true, false>(d, st);
  if (d.a_kc && !d.b_kc) return launch_buf<128, 32, true, false, false>(d, st);
  if (!d.a_kc && !d.b_kc) return launch_buf<128, 32, false, false, false>(d, st);
  ctts_set_error("ctts_gemm: layout a_kc=0,b_kc=1 is not instantiated");
  return -1;
}

// buffer loaders need: no per-batch length limits and every operand element within 2 GiB of its (batch) base
bool buf_ok(const ctts_gemm_desc& d) {
  if (d.lens && (d.lim_m || d.lim_n || d.lim_k) && d.conv_T > 0) return false;
  const long a_ext = d.a_kc ? ((long)d.M * d.lda + d.K) : ((long)d.K * d.lda + d.M);
  const long b_ext = d.b_kc ? ((long)d.N * d.ldb + d.K) : ((long)d.K * d.ldb + d.N);
  return a_ext * 4 < 0x7FFF0000L && b_ext * 4 < 0x7FFF0000L;
}

template <int BM, int BN, bool VEC>
int dispatch_layout(const ctts_gemm_desc& d, hipStream_t st) {
  const bool conv = d.conv_T > 0;
  if (d.a_kc && d.b_kc) return conv ? launch<BM, BN, true, true, true, VEC>(d, st) : launch<BM, BN, true, true, false, VEC>(d, st);
  if (d.a_kc && !d.b_kc) return conv ? launch<BM, BN, true, false, true, VEC>(d, st) : launch<BM, BN, true, false, false, VEC>(d, st);
  if (!d.a_kc && !d.b_kc) return conv ? launch<BM, BN, false, false, true, VEC>(d, st) : launch<BM, BN, false, false, false, VEC>(d, st);
  ctts_set_error("ctts_gemm: layout a_kc=0,b_kc=1 is not instantiated");
  return -1;
}

// eligibility of the branch-free 16-byte loaders (see VLoaderKC)
bool chunks_ok(const ctts_gemm_desc& d) {          // every 4-float chunk along the contiguous dimension lies inside the operand or outside it
  const bool a_ext = d.a_kc ? (d.K % 4 == 0) : (d.M % 4 == 0);
  const bool b_ext = d.b_kc ? (d.K % 4 == 0) : (d.N % 4 == 0);
  return a_ext && b_ext;
}
bool vec_ok(const ctts_gemm_desc& d) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool strides = !((d.lda | d.ldb | d.sA0 | d.sA1 | d.sB0 | d.sB1) & 3);
  return strides && chunks_ok(d) && al16(d.A) && al16(d.B);
}

}  // namespace

namespace {
// one wave: stable partition of the m-tiles into active (first) and inactive, count in map[0]
__global__ __launch_bounds__(64) void row_tile_map_kernel(const int32_t* __restrict__ row_lens, int row_T, int row_halo, int M,
                                                            int32_t* __restrict__ map) {
  const int tiles = (M + 63) / 64, lane = threadIdx.x;
  auto inactive = [&](int tm) -> bool {
    const int row0 = tm * 64, last = min(row0 + 64, M) - 1;
    const int b0 = row0 / row_T, b1 = last / row_T;
    return b0 == b1 && (row0 - b0 * row_T) >= row_lens[b0] + row_halo;
  };
  int n_act = 0;
  for (int base = 0; base < tiles; base += 64) {
    const int tm = base + lane;
    const bool a = tm < tiles && !inactive(tm);
    const unsigned long long m = __ballot(a);
    if (a) map[1 + n_act + __popcll(m & ((1ULL << lane) - 1ULL))] = tm;
    n_act += __popcll(m);
  }
  int n_in = 0;
  for (int base = 0; base < tiles; base += 64) {
    const int tm = base + lane;
    const bool a = tm < tiles && inactive(tm);
    const unsigned long long m = __ballot(a);
    if (a) map[1 + n_act + n_in + __popcll(m & ((1ULL << lane) - 1ULL))] = tm;
    n_in += __popcll(m);
  }
  if (lane == 0) map[0] = n_act;
}
}  // namespace

namespace {
// Second launch of a split-K GEMM: C[m, n] += alpha * (P_0 + P_1 + ... )[m, n] in split order (fixed: bit-reproducible), for the rows /
// columns / batches the GEMM kernels computed - per-batch length limits, tiles of padded rows (zero-filled by the GEMM, skipped here) and
// splits whose K range was empty (never written) follow the kernels' own predicates.  One thread per 4 columns, all splits' loads in flight.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ctts_gemm_desc d, int BM, int kround) {
  const int z = blockIdx.z, z0 = z / d.nb1, z1 = z - z0 * d.nb1;
  int Mv = d.M, Nv = d.N, Kv = d.K;
  if (d.lens) {
    const int L = d.lens[z0];
    if (d.lim_m) Mv = min(Mv, L);
    if (d.lim_n) Nv = min(Nv, L);
    if (d.lim_k) Kv = min(Kv, L);
  }
  const bool ow = d.split_overwrite != 0;      // C = alpha * sum (and ZERO outside the limits / in padded tiles) instead of C += : no pre-zeroed C
  int nact = 0;
  if (Kv > 0) {
    const int chunk = ((Kv + d.split_k - 1) / d.split_k + kround - 1) / kround * kround;
    nact = (Kv + chunk - 1) / chunk;
  }
  if (nact == 0 && !ow) return;
  const long ldp = gemm_partial_ld(d.N);
  const int n4 = (int)(ldp >> 2);
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const int m = (int)(e / n4), n = (int)(e - (long)m * n4) * 4;
  if (m >= d.M || n >= d.N) return;
  bool live = m < Mv && n < Nv && nact > 0;
  if (live && d.a_kc && d.row_lens) {          // the GEMM wrote no partials for whole tiles of padded rows (their result is defined as zero)
    const int row0 = m / BM * BM, last = min(row0 + BM, Mv) - 1;
    const int b0 = row0 / d.row_T, b1 = last / d.row_T;
    if (b0 == b1 && (row0 - b0 * d.row_T) >= d.row_lens[b0] + d.row_halo) live = false;
  }
  if (!live && !ow) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float4* P = reinterpret_cast<const float4*>(gemm_partial_base(d, z, 0) + (long)m * ldp + n);
    const long sstride = (long)d.M * ldp / 4;          // P_s of a batch are consecutive [M, ldp] matrices
    for (int s = 0; s < nact; s += 8) {          // eight partials in flight, added in split order
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (s + u < nact) ? P[(long)(s + u) * sstride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
  }
  float* Cr = d.C + z0 * d.sC0 + z1 * d.sC1 + (long)m * d.ldc + n;
  const float al = d.alpha;
  const int ncols = ow ? d.N : Nv;              // overwrite mode also zeroes the columns between the batch's limit and N
  if (n + 4 <= Nv && ((reinterpret_cast<uintptr_t>(Cr) & 15) == 0)) {
    float4 c = ow ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(Cr);
    c.x += al * a.x; c.y += al * a.y; c.z += al * a.z; c.w += al * a.w;
    *reinterpret_cast<float4*>(Cr) = c;
  } else {
    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (n + q < ncols) Cr[q] = (ow ? 0.f : Cr[q]) + ((live && n + q < Nv) ? al * av[q] : 0.f);
  }
}

int splitk_reduce(const ctts_gemm_desc& d, int BM, int kround, hipStream_t st) {
  const long e = (long)d.M * (gemm_partial_ld(d.N) / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((e + 255) / 256), 1, d.nb0 * d.nb1), dim3(256), 0, st, d, BM, kround);
  CTTS_CHECK_LAUNCH("ctts_gemm(split-K reduce)");
  return 0;
}
}  // namespace

extern "C" int ctts_row_tile_map(const int32_t* row_lens, int row_T, int row_halo, int M, int32_t* tile_map, void* stream) {
  CTTS_REQUIRE(row_lens && tile_map && row_T > 0 && M >= 0, "ctts_row_tile_map: bad arguments");
  hipLaunchKernelGGL(row_tile_map_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, row_lens, row_T, row_halo, M, tile_map);
  CTTS_CHECK_LAUNCH("ctts_row_tile_map");
  return 0;
}

namespace {
struct GemmSplitPlan { int deferred_ok; int count; long stride; };
}
// plan != nullptr: no launch - only answer how a split-K launch of this descriptor would lay out its partial matrices
// fp32-on-bf16-pipe kernel (gemm_x6_kernel): large unbatched NT launches whose N is a multiple of the 128-column tile.  The arithmetic is
// the CALLER's choice, per descriptor (ctts_gemm_desc.bf16_split: 0 = fp32 MFMA only, 1 = allowed, 2 = allowed below the size thresholds).
static bool gemm_x6_takes(const ctts_gemm_desc& d) {
  const int on = d.bf16_split;
  static const long min_tiles = getenv("CTTS_X6_MIN_TILES") ? atol(getenv("CTTS_X6_MIN_TILES")) : 384;
  if (on < 1 || !d.a_kc || !d.b_kc || d.nb0 * d.nb1 != 1 || d.split_k > 1 || d.E || d.lens || d.conv_on_b) return false;
  if (d.K < 256 || d.K % BK || d.N % 128 || d.M < 1024) return false;
  if (d.conv_T > 0 && d.conv_cin % 4) return false;
  if (on != 2 && on != 4 && (long)((d.M + 127) / 128) * (d.N / 128) < min_tiles) return false;      // 2 / 4 = forced (tests): no size threshold
  return vec_ok(d) && buf_ok(d);
}

// the TN (weight-gradient) form: unbatched, both operands reduction-major, tile-aligned output, long reduction; CTTS_X6_TN=0 turns it off
static bool gemm_x6tn_takes(const ctts_gemm_desc& d) {
  static const int tn = getenv("CTTS_X6_TN") ? atoi(getenv("CTTS_X6_TN")) : 1;
  static const long min_wg = getenv("CTTS_X6_TN_MIN_WG") ? atol(getenv("CTTS_X6_TN_MIN_WG")) : 384;
  if (d.bf16_split < 1 || !tn || d.a_kc || d.b_kc || d.nb0 * d.nb1 != 1 || d.E || d.lens || d.epi_bwd) return false;
  if (d.M % 128 || d.N % 128 || d.K % BK || d.K < 2048) return false;
  // (a thread stages 8 consecutive reduction rows starting at a multiple of 8: they stay inside one utterance when conv_T % 8 == 0)
  if (d.conv_T > 0 && (!d.conv_on_b || d.conv_T % 8 || d.conv_cin % 4)) return false;
  if (d.bf16_split != 2 && d.bf16_split != 4 && (long)(d.M / 128) * (d.N / 128) * (d.split_k > 1 ? d.split_k : 1) < min_wg) return false;
  return vec_ok(d) && buf_ok(d);
}

static int gemm_x6tn_launch(const ctts_gemm_desc& d, hipStream_t st) {
  const dim3 grid((d.M / 128) * (d.N / 128), d.split_k > 1 ? d.split_k : 1, 1);
  if (d.conv_T > 0) hipLaunchKernelGGL(gemm_x6tn_kernel<true>, grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL(gemm_x6tn_kernel<false>, grid, dim3(256), 0, st, d);
  CTTS_CHECK_LAUNCH("ctts_gemm(x6tn)");
  return 0;
}

static int gemm_x6_launch(const ctts_gemm_desc& d, hipStream_t st) {
  const int tiles = ((d.M + 127) / 128) * (d.N / 128);
  if (d.conv_T > 0) hipLaunchKernelGGL(gemm_x6_kernel<true>, dim3(tiles), dim3(256), 0, st, d);
  else hipLaunchKernelGGL(gemm_x6_kernel<false>, dim3(tiles), dim3(256), 0, st, d);
  CTTS_CHECK_LAUNCH("ctts_gemm(x6)");
  return 0;
}

extern "C" int ctts_gemm_takes_bf16_split(const ctts_gemm_desc* dp) {
  if (!dp) return 0;
  ctts_gemm_desc d = *dp;
  if (d.nb0 < 1) d.nb0 = 1;
  if (d.nb1 < 1) d.nb1 = 1;
  if (ctts_gemm_takes_planes(&d)) return 0;                  // ctts_gemm asks the plane kernel first,
  if (ctts_gemm_takes_weight_stationary(&d)) return 0;       // then the weight-stationary kernel
  return (gemm_x6_takes(d) || gemm_x6tn_takes(d)) ? 1 : 0;
}

static int gemm_impl(const ctts_gemm_desc* dp, void* stream, GemmSplitPlan* plan) {
  CTTS_REQUIRE(dp != nullptr, "ctts_gemm: null descriptor");
  ctts_gemm_desc d = *dp;
  CTTS_REQUIRE(d.A && d.B && d.C, "ctts_gemm: null operand pointer");
  CTTS_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, "ctts_gemm: negative dimension");
  if (d.M == 0 || d.N == 0) return 0;
  if (d.nb0 < 1) d.nb0 = 1;
  if (d.nb1 < 1) d.nb1 = 1;
  if (d.conv_T > 0) {
    CTTS_REQUIRE(d.conv_cin > 0 && (d.conv_cin % 4) == 0, "ctts_gemm: conv view needs cin %% 4 == 0 (got %d)", d.conv_cin);
    CTTS_REQUIRE(d.conv_on_b ? (!d.a_kc && !d.b_kc) : (d.a_kc != 0), "ctts_gemm: conv view on an unsupported operand layout");
  }
  CTTS_REQUIRE(d.p_drop >= 0.f && d.p_drop < 1.f, "ctts_gemm: p_drop out of range");
  CTTS_REQUIRE(!d.epi_bwd || (d.split_k <= 1 && !d.bias && !d.R && !d.rowscale && !d.E && (!d.act || d.Z)),
               "ctts_gemm: epi_bwd excludes bias, residual, rowscale, E and split-K, and needs Z when an activation is given");
  CTTS_REQUIRE(!d.E || (d.rowsub && d.split_k <= 1 && !d.bias && !d.act && d.p_drop == 0.f && !d.R && !d.rowscale && !d.Z),
               "ctts_gemm: the E/rowsub epilogue excludes bias, activation, dropout, residual, rowscale and split-K");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool x6tn = gemm_x6tn_takes(d);            // weight gradient on the bf16-split kernel: a tile-kernel launch with ordered split-K partials
  if (plan) {
    plan->deferred_ok = 0;
    if (d.split_k <= 1 || ctts_gemm_plw_takes(d) || ctts_gemm_takes_weight_stationary(&d) || (!x6tn && ctts_gemm_takes_persistent(&d))) return 0;
  } else {
    const int pl = ctts_gemm_pl_try(d, st);      // pre-split bf16 planes given and eligible: persistent plane kernel (gemm_pl.hip)
    if (pl != 0) return pl > 0 ? 0 : pl;
    const int plw = ctts_gemm_plw_try(d, st);    // ... in the weight-gradient layout (gemm_plw.hip): adds into C itself, no partial matrices
    if (plw != 0) return plw > 0 ? 0 : plw;
    const int ws = ctts_gemm_ws_try(d, st);      // weight-stationary kernel (gemm_ws.hip) for K = 256 linears with many rows
    if (ws != 0) return ws > 0 ? 0 : ws;
    if (gemm_x6_takes(d)) return gemm_x6_launch(d, st);      // fp32 products on the bf16 matrix pipe (six-term split)
    if (!x6tn) {
      const int sk = ctts_gemm_sk_try(d, st);    // persistent stream-K kernel (gemm_sk.hip) when the descriptor is eligible
      if (sk != 0) return sk > 0 ? 0 : sk;
    }
  }
  const long tiles128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128) * (d.split_k > 1 ? d.split_k : 1) * d.nb0 * d.nb1;
  static const int force_tile = getenv("CTTS_FORCE_TILE") ? atoi(getenv("CTTS_FORCE_TILE")) : 0;   // tuning knob
  static const bool natural = getenv("CTTS_NATURAL_ORDER") != nullptr;
  // XCD-grouped order for scheduled launches: default 4 n-groups (XCD x: n-group x % 4, every second scheduled m-tile) - halves the
  // L2-miss traffic of the FFN conv (FETCH_SIZE 441 -> 195 MB per launch) at unchanged time; CTTS_TILE_GROUP=<g> overrides, 0 = off.
  static const int tile_group = getenv("CTTS_TILE_GROUP") ? atoi(getenv("CTTS_TILE_GROUP")) : -1;
  d.tile_group_n = 0;
  if (tile_group != 0 && d.tile_map && d.tile_map != reinterpret_cast<const int32_t*>(1)) {
    const int tn = (d.N + 63) / 64, tmn = (d.M + 63) / 64;
    const int g = tile_group > 0 ? tile_group : ((tn >= 8 && tn % 4 == 0) ? tn / 4 : 0);
    if (g > 0 && tn % g == 0) {
      const int ng = tn / g;
      if ((ng == 1 || ng == 2 || ng == 4 || ng == 8) && tmn % (8 / ng) == 0 && ((long)tmn * tn) % 8 == 0) d.tile_group_n = g;
    }
  }
  // CTTS_TN_NATURAL: plain blockIdx order for the weight-gradient (TN, split-K) launches only.  4-12 % faster in the isolated
  // micro-benchmark (96.6 -> 86 us FFN linear, 711 -> 683 us FFN conv) - in-step effect measured separately, off by default.
  static const bool tn_natural = getenv("CTTS_TN_NATURAL") != nullptr;
  if ((natural || (tn_natural && !d.a_kc && !d.b_kc)) && !d.tile_map) d.tile_map = reinterpret_cast<const int32_t*>(1);
  // 16-byte BUFFER loads only need dword-aligned addresses (the LDS side of the tile is aligned by construction), so the buffer kernels
  // also take operands whose base / leading dimension is not a multiple of 4 floats - the padded views (rows of T+1) of the relative
  // attention score slabs - as long as the chunking fits.  Pointer-based 16-byte loads (the non-buffer VEC path) need full alignment.
  const bool aligned = vec_ok(d);
#ifdef CTTS_NO_BUF
  const bool buf_unaligned = false;
#else
  const bool buf_unaligned = !aligned && chunks_ok(d) && d.conv_T <= 0 && !(d.lens && (d.lim_m || d.lim_n || d.lim_k));
#endif
  // ---- which kernel family / tile takes the launch (decided first: the ordered split-K sum needs the tile shape)
  enum { K_SCALAR64, K_BUF128, K_BUF_K2, K_BUF_NARROW, K_BUF64, K_VEC128, K_VEC64, K_X6TN } kind;
  int BMs = 64, BNs = 64;
  if (x6tn) { kind = K_X6TN; BMs = BNs = 128; }
  else if (!aligned && !(buf_unaligned && buf_ok(d))) kind = K_SCALAR64;
#ifndef CTTS_NO_BUF
  else if (buf_ok(d)) {
    // 64x64 tiles (64 VGPRs: 8 waves/SIMD) match the 128x128 kernel on every measured shape (109 / 119 / 108 / 107 TFLOP/s on FFN conv fwd,
    // 4096^3, conv dgrad, conv wgrad) and make padded-row skipping effective: 64-row granularity and many waves per CU instead of two
    // rounds of 128-row tiles (fs2 train step 33.3 -> 29.8 ms).  CTTS_FORCE_TILE=128 keeps the big-tile kernel reachable for A/B runs.
    // under-filled forward / data-gradient launches (A K-contiguous, unbatched): few 64 x 64 tiles - the phoneme-level layers (2,048
    // rows).  Routing by the step time (same box): tile limit 160 / 320 / 640 / 1024: fs2 23.36 / 23.27 / 23.17 / 23.24 ms from 23.48,
    // conformer 27.78 / 27.84 / 27.62 / 27.90 from 28.14; K >= 256 instead of 512: another -0.03 / -0.05 ms.
    static const int k2 = getenv("CTTS_K2_TILE") ? atoi(getenv("CTTS_K2_TILE")) : 640;          // largest 64 x 64 tile count routed here (0 = off)
    static const int k2_mink = getenv("CTTS_K2_MIN_K") ? atoi(getenv("CTTS_K2_MIN_K")) : 256;
    // the split-K weight gradients (TN) of those layers as well: conformer 27.82 -> 27.53 ms.  NOTE: the two-group kernel has no K-block
    // skipping (row_lens / tile_map are ignored for TN launches) - correct because the rows of dZ beyond a sequence's length are exactly
    // zero (every producer of a gradient masks or zero-fills its padded rows), it merely multiplies those zeros.
    static const int k2_tn = getenv("CTTS_K2_TN") ? atoi(getenv("CTTS_K2_TN")) : 1;
    static const bool narrow = getenv("CTTS_NARROW_TILE") ? atoi(getenv("CTTS_NARROW_TILE")) != 0 : true;
    if (force_tile == 128) { kind = K_BUF128; BMs = BNs = 128; }
    else if (k2 && (d.a_kc || (k2_tn && !d.b_kc)) && d.nb0 * d.nb1 == 1 && !d.lens && d.N >= 64 && d.M >= 256 && d.K >= k2_mink &&
             (long)((d.M + 63) / 64) * ((d.N + 63) / 64) * (d.split_k > 1 ? d.split_k : 1) <= k2) { kind = K_BUF_K2; BMs = 32; }
    else if (narrow && d.N <= 32 && d.M >= 256 && d.conv_T <= 0 && !d.tile_map && !d.row_lens) { kind = K_BUF_NARROW; BMs = 128; BNs = 32; }
    else kind = K_BUF64;
  }
#endif
  // weight-gradient (TN) reductions measured faster on 64x64 tiles (64 VGPRs: 8 waves/SIMD): 92.6 vs 84.6 TFLOP/s on the FFN conv wgrad
  else if (tiles128 >= 256 && d.N > 64 && (d.a_kc || d.b_kc)) { kind = K_VEC128; BMs = BNs = 128; }
  else kind = K_VEC64;

  if (d.split_k > 1) {
    // Ordered split-K (gemm_common.h; splitk_reduce_kernel above): one partial matrix [M, N] per (batch, split) in the
    // caller's workspace.  split_k is an upper bound: it is lowered until the partials fit (never below 2 - "C += alpha A B" is what
    // split_k > 1 means).
    CTTS_REQUIRE(plan || d.split_out || (d.sk_ws && d.sk_ws_bytes >= (int64_t)CTTS_WS_BYTES),
                 "ctts_gemm: split_k > 1 needs the workspace (ctts_workspace_bytes() bytes, zero-filled once) in sk_ws - partial sums "
                 "are added in a fixed order through it, the library has no floating-point atomics");
    const long per_split = (long)d.nb0 * d.nb1 * d.M * gemm_partial_ld(d.N);
    // split_out: the caller keeps the partial matrices and adds them itself later (ctts_partial_sums, many GEMMs per launch) - no reduce launch
    const long room = plan ? (1L << 60) : (d.split_out ? d.split_out_floats : (long)CTTS_WS_SLAB_FLOATS);
    const long cap = room / per_split;
    CTTS_REQUIRE(cap >= 2 && (long)d.M * gemm_partial_ld(d.N) * 4 < 0x7FFF0000L, "ctts_gemm: split-K output [%d, %d] x %d batches does not fit the %s",
                 d.M, d.N, d.nb0 * d.nb1, d.split_out ? "caller's split_out buffer" : "workspace");
    if (d.split_k > cap) d.split_k = (int)cap;
  }
  const bool split = d.split_k > 1;
  const int kround = kind == K_BUF_K2 ? 2 * BK : BK;          // the K granularity of the kernel's split (its `chunk`)
  if (plan) {
    if (split && d.nb0 * d.nb1 == 1 && !d.lens && d.K > 0 && !(d.a_kc && d.row_lens)) {
      const int chunk = ((d.K + d.split_k - 1) / d.split_k + kround - 1) / kround * kround;
      plan->deferred_ok = 1;
      plan->count = (d.K + chunk - 1) / chunk;
      plan->stride = (long)d.M * gemm_partial_ld(d.N);
    }
    return 0;
  }
  const ctts_gemm_desc d_user = d;                 // what the reduce launch accumulates into
  if (split) {
    // the tile kernels see the partial matrices as their output: C = P [z][split][M][ldp], plain stores (alpha = 1, no epilogue terms);
    // workgroup (z, split) adds split * M * ldc itself.  Whole tiles of padded rows are "zero-filled" into P_0 and skipped by the reduce.
    const long ldp = gemm_partial_ld(d.N);
    d.C = d.split_out ? d.split_out : reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(d.sk_ws) + CTTS_WS_SLABS);
    d.ldc = ldp;
    d.sC1 = (long)d.split_k * d.M * ldp;
    d.sC0 = (long)d.nb1 * d.sC1;
    d.alpha = 1.f; d.bias = nullptr; d.Z = nullptr; d.act = 0; d.p_drop = 0.f; d.R = nullptr; d.rowscale = nullptr;
  }
  int rc;
  switch (kind) {
    case K_SCALAR64: rc = dispatch_layout<64, 64, false>(d, st); break;
#ifndef CTTS_NO_BUF
    case K_BUF128: rc = dispatch_buf<128, 128>(d, st); break;
    case K_BUF_K2: rc = dispatch_buf_k2(d, st); break;
    case K_BUF_NARROW: rc = dispatch_buf_narrow(d, st); break;
    case K_BUF64: rc = dispatch_buf<64, 64>(d, st); break;
#endif
    case K_VEC128: rc = dispatch_layout<128, 128, true>(d, st); break;
    case K_X6TN: rc = gemm_x6tn_launch(d, st); break;
    default: rc = dispatch_layout<64, 64, true>(d, st); break;
  }
  if (rc != 0 || !split || d_user.split_out) return rc;
  ctts_gemm_desc dr = d_user;
  dr.split_k = d.split_k;                          // (possibly lowered above)
  return splitk_reduce(dr, BMs, kround, st);
}

extern "C" int ctts_gemm(const ctts_gemm_desc* dp, void* stream) { return gemm_impl(dp, stream, nullptr); }

// Deferred split-K (ctts_gemm_desc.split_out): would ctts_gemm run this descriptor as a split-K launch of the tile kernels whose partial
// matrices the caller may keep and add later?  Returns 1 and fills *count (partial matrices that will be written: P_0 .. P_count-1) and
// *stride (floats between them; each is [M, N rounded up to 4] row-major), else 0 (stream-K / weight-stationary take it, split_k <= 1,
// batched or length-limited launches).  Give ctts_gemm a split_out buffer of at least split_k * stride floats.
extern "C" int ctts_gemm_split_plan(const ctts_gemm_desc* dp, int32_t* count, int64_t* stride) {
  GemmSplitPlan pl = {0, 0, 0};
  ctts_gemm_desc d = *dp;
  d.split_out = nullptr;
  if (gemm_impl(&d, nullptr, &pl) != 0 || !pl.deferred_ok) return 0;
  if (count) *count = pl.count;
  if (stride) *stride = pl.stride;
  return 1;
}
