// Persistent stream-K fp32-MFMA GEMM for gfx950 with direct-to-LDS operand loads.
//
// Same arithmetic and the same fused epilogue as gemm.hip (v_mfma_f32_32x32x2_f32, 64x64x32 workgroup tiles, 2x2 waves), but
//   * the grid is PERSISTENT: 8*W workgroups (W per XCD, all co-resident) cut the (tile, K-block) unit space evenly (sk_plan.h), so
//     every CU carries the same amount of MFMA work whatever the tile count - no half-empty last round, no drain at 1 wave per SIMD;
//     a tile that is cut between workgroups is summed in a fixed order by its owner (slab + flag hand-off, agent scope), so results
//     are deterministic for a given grid;
//   * operands go global -> LDS by DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass; two LDS stages, ONE
//     barrier per K-block, and the load stream runs on across tile boundaries (the next tile's first K-block is in flight while the
//     epilogue of the current one runs);
//   * a K-contiguous operand tile (64 rows x 32 floats) is stored as plain 128-byte rows; the 16-byte chunks of row r are XOR-swizzled
//     with (r >> 1) & 7 on the SOURCE address (the DMA writes lane-linear), which makes the four ds_read_b128 of a fragment conflict
//     free without padding.  A row-contiguous operand tile (32 k-rows x 64 floats) needs no swizzle (ds_read_b32 along the row).
//   * masked chunks (rows / columns beyond the operand, conv time-boundary taps, K tail rows) use the buffer descriptor's range check:
//     offset 0x80000000 makes the DMA write zeros.
// Eligibility (ctts_gemm_sk_try): unbatched, no per-batch length limits, 16-byte aligned operands, K % 32 == 0 for K-contiguous
// operands, conv views with cin % 32 == 0.  Everything else stays on gemm.hip.
#include "ctts_common.h"
#include "gemm_common.h"
#include "sk_plan.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned int sk_gu32;
typedef __attribute__((address_space(3))) void sk_lds_void;

constexpr unsigned SK_OOB = 0x80000000u;
constexpr int SK_FLAG_WORDS = 4096;            // header of the workspace: flags[0..2047], error word at [2048]
constexpr int SK_SLAB_FLOATS_MAX = 2048 * 4096; // slab area: grid x (BM x BN) floats never exceeds this (host check)
constexpr int SK_MAX_WG = 2048;

struct SkArgs {
  int tiles_m, tiles_n;      // static tile grid (BM x BN tiles)
  int nkb;                   // K-blocks per tile when no K-block schedule is given
  int gw;                    // n-tiles per schedule group
  int whole_tiles;           // 1: never split a tile
  int accumulate;            // 1: C += alpha * acc (weight gradients), no other epilogue
  int conv_chan_major;       // conv view on A: walk K as (channel block, tap) instead of (tap, channel block) - see k0_of
  int debug;                 // CTTS_SK_DEBUG (tools only): 1 = no DMA after the first block, 2 = every workgroup loads tile (0,0), 4 = no epilogue, 16 = record shader cycles / wall ticks of workgroup 8 in the workspace header
  unsigned* ws;              // workspace: SK_FLAG_WORDS words, then one slab per workgroup
};

typedef int sk_i32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (2 GiB window, 32-bit raw-buffer format) as four SGPR words for the inline-asm DMA below
__device__ __forceinline__ sk_i32x4 sk_make_rsrc(const void* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  sk_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  r.z = 0x7FFFFFFE;
  r.w = 0x00020000;
  return r;
}

// One LDS-DMA instruction: 64 lanes x 16 bytes, global (descriptor + voff + soff) -> LDS [lds_addr + lane * 16).  Inline asm on purpose:
// hipcc's waitcnt pass knows nothing about it, so it does not put `s_waitcnt vmcnt(0)` in front of every ds_read that follows (it does
// for the builtin, because it cannot prove that the DMA target and the fragment reads are different LDS stages), and the loop's own
// `s_waitcnt vmcnt(N)` + `s_barrier` stay the only synchronisation between the DMA and its readers.
__device__ __forceinline__ void sk_dma16(sk_i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");      // m0 is not on the clobber list (hipcc rejects reserved registers there); nothing else in this
                                 // translation unit uses m0 - check the ISA (grep m0) when adding LDS-direct / GWS / movrel code
}

// r mod T for 0 <= r < 2^24 without the integer-division sequence
__device__ __forceinline__ int sk_mod(int r, int T, float rcpT) {
  int q = (int)((float)r * rcpT);
  int m = r - q * T;
  m = m < 0 ? m + T : m;
  m = m >= T ? m - T : m;
  return m;
}

// ---- K-contiguous operand tile: ROWS x 32 k, element (r, k) = P[(ext0 + r) * ld + k], stored as 128-byte rows whose 16-byte chunks are
//      XOR-swizzled with (r >> 1) & 7.  Each wave moves ROWS / 4 rows = NI DMA instructions of 8 rows.  Conv view: row r is shifted by
//      -pad rows and the chunk is zero unless 0 <= t(r) + tap - pad < T, tap = k / cin (cin % 32 == 0: one tap per K-block)
template <int ROWS, bool CONV>
struct SkLoadKC {
  static constexpr int NI = ROWS / 32;
  unsigned voff[NI];
  int trow[NI];
  __device__ __forceinline__ void set_piece(int ext0, int ext_lim, long ld, int wave, int lane, int T) {
    const int t0 = CONV ? ext0 % T : 0;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = (wave * NI + j) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (((j & 1) * 4) | (lane >> 4));     // = (lane & 7) ^ ((r >> 1) & 7)
      const int row = ext0 + r;
      voff[j] = row < ext_lim ? ((unsigned)row * (unsigned)ld + (unsigned)(c * 4)) * 4u : SK_OOB;     // < 2^31 (host check)
      if (CONV) {
        int t = t0 + r;                                               // T >= ROWS (host check)
        trow[j] = t >= T ? t - T : t;
      }
    }
  }
  // tap_m_pad = k0 / cin - pad (wave-uniform); lds = byte address of this wave's first row block
  __device__ __forceinline__ void issue(sk_i32x4 rsrc, unsigned lds, unsigned soff, int tap_m_pad, int T) const {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      unsigned v = voff[j];
      if (CONV) v = ((unsigned)(trow[j] + tap_m_pad) < (unsigned)T) ? v : SK_OOB;
      sk_dma16(rsrc, lds + j * 1024, v, soff);
    }
  }
};

// ---- row-contiguous operand tile: 32 k-rows x COLS columns, element (k, c) = P[k * ld + ext0 + c], stored as plain rows of COLS floats
//      (fragment reads go along the row: no swizzle).  One DMA instruction covers 256 / COLS k-rows; each wave issues NI of them.
//      Conv view (weight gradient): row k is shifted by -pad rows and the chunk is zero unless 0 <= t(k) + c / cin - pad < T
template <int COLS, bool CONV>
struct SkLoadRC {
  static constexpr int RPI = 256 / COLS;        // k-rows per instruction
  static constexpr int NI = 8 / RPI;            // instructions per wave (32 k-rows / 4 waves)
  unsigned voff[NI];
  int ctap;            // column tap - pad (piece invariant)
  int klocal[NI];
  __device__ __forceinline__ void set_piece(int ext0, int ext_lim, long ld, int wave, int lane, int cin, int pad) {
    const int col = ext0 + 4 * (lane % (COLS / 4));
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      klocal[j] = (wave * NI + j) * RPI + lane / (COLS / 4);
      voff[j] = col < ext_lim ? ((unsigned)klocal[j] * (unsigned)ld + (unsigned)col) * 4u : SK_OOB;
    }
    ctap = CONV ? col / cin - pad : 0;
  }
  __device__ __forceinline__ void issue(sk_i32x4 rsrc, unsigned lds, unsigned soff, int k0, int K, int T, float rcpT) const {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int krow = k0 + klocal[j];
      bool ok = krow < K;
      if (CONV) ok = ok && ((unsigned)(sk_mod(krow, T, rcpT) + ctap) < (unsigned)T);
      const unsigned v = ok ? voff[j] : SK_OOB;
      sk_dma16(rsrc, lds + j * 1024, v, soff);
    }
  }
};

// fragment of one 32-wide MFMA tile: 16 k-steps, lane (l31, h) gets element k = h * 16 + j
template <bool KC, int EXT>
__device__ __forceinline__ void sk_fetch(const float* s, int ext0, int l31, int h, float (&f)[16]) {
  if (KC) {
    const int sw = (l31 >> 1) & 7;
    const float* row = s + (ext0 + l31) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(row + (((h * 4 + q) ^ sw) << 2));
      f[4 * q + 0] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
  } else {
    const float* p = s + (h * 16) * EXT + ext0 + l31;
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = p[j * EXT];
  }
}

template <int BM, int BN>
__device__ __forceinline__ void sk_zero_tile(const ctts_gemm_desc& d, int row0, int col0) {
  const int nrows = min(BM, d.M - row0), ncols = min(BN, d.N - col0);
  for (int e = threadIdx.x; e < nrows * BN; e += 256) {
    const int r = e / BN, c = e - r * BN;
    if (c < ncols) {
      d.C[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
      if (d.Z && !d.epi_bwd) d.Z[(long)(row0 + r) * d.ldz + col0 + c] = 0.f;
    }
  }
}

// MT x NT MFMA tiles per wave: workgroup tile (64 MT) x (64 NT), 2 x 2 waves
template <bool A_KC, bool B_KC, bool CONV, int STAGES, int MT, int NT>
__global__ __launch_bounds__(256, (MT * NT >= 4 ? 2 : ((MT * NT >= 2 || STAGES >= 3) ? 3 : 4))) void gemm_sk_kernel(const ctts_gemm_desc d, const SkArgs p) {
  constexpr bool TN = !A_KC && !B_KC;
  constexpr bool CONV_A = CONV && A_KC;
  constexpr bool CONV_B = CONV && TN;
  constexpr int BM = 64 * MT, BN = 64 * NT, WM = 32 * MT, WN = 32 * NT;
  constexpr int A_FLOATS = BM * 32, STAGE_FLOATS = (BM + BN) * 32;
  constexpr int N_DMA = 2 * (MT + NT);                      // DMA instructions per K-block and wave
  constexpr int SLAB = BM * BN;
  __shared__ __attribute__((aligned(16))) float smem[STAGES * STAGE_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  if (p.debug & 96) {      // tools: de-phase the co-resident workgroups of a CU by half a K-block (bits 5 / 6 pick which ones wait)
    const int wjp = (int)(blockIdx.x >> 3), Wp = (int)(gridDim.x >> 3);
    const bool late = (p.debug & 32) ? (wjp & 1) : (wjp >= Wp / 2);
    if (late) for (int i = 0; i < ((p.debug >> 8) & 255); ++i) __builtin_amdgcn_s_sleep(127);      // 127 x 64 cycles per iteration
  }
  const unsigned long long dbg_c0 = (p.debug & 16) ? clock64() : 0ull, dbg_w0 = (p.debug & 16) ? wall_clock64() : 0ull;

  // ---- schedule inputs that live in device memory
  // (read through the constant address space = scalar loads: a vector load here would make hipcc wait vmcnt(0) inside the K loop,
  //  which drains the DMA stream; the maps were written by an earlier kernel, so the scalar cache is coherent for them)
  typedef const __attribute__((address_space(4))) int32_t* sk_cmap;
  const sk_cmap mmap = (A_KC && MT == 1) ? (sk_cmap)(uintptr_t)d.tile_map : (sk_cmap)0;   // 64-row m-tile schedule: [0] = active count, then tile ids
  const sk_cmap kmap = TN ? (sk_cmap)(uintptr_t)d.tile_map : (sk_cmap)0;                  // TN: the same map as a K-block schedule (64 rows per entry)
  const int n_mt = mmap ? mmap[0] : p.tiles_m;
  const int nkb = kmap ? 2 * kmap[0] : p.nkb;

  // ---- padded m-tiles are defined as zero: stores only, spread over the grid
  if (mmap) {
    const int n_zero = (p.tiles_m - n_mt) * p.tiles_n;
    for (int zt = blockIdx.x; zt < n_zero; zt += gridDim.x) {
      const int mi = zt / p.tiles_n;
      sk_zero_tile<BM, BN>(d, mmap[1 + n_mt + mi] * BM, (zt - mi * p.tiles_n) * BN);
    }
  }

  SkGeom g{n_mt * p.tiles_n, nkb, (int)(gridDim.x >> 3), p.whole_tiles};
  if (g.n_tiles <= 0 || nkb <= 0) return;
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  const SkRange rg = sk_range(g, xcd, wj);
  if (rg.hi <= rg.lo) return;

  const sk_i32x4 ra_src = sk_make_rsrc(d.A - (CONV_A ? (long)d.conv_pad * d.lda : 0));
  const sk_i32x4 rb_src = sk_make_rsrc(d.B - (CONV_B ? (long)d.conv_pad * d.ldb : 0));
  const unsigned smem_addr = (unsigned)reinterpret_cast<uintptr_t>(smem);      // low word of the flat address = LDS byte address
  unsigned* flags = p.ws;
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(p.ws) + CTTS_WS_SLABS);      // shared slab area of the workspace (ctts_common.h)
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, 0x7FFFFFFE, 0x00020000);
  const int T = d.conv_T > 0 ? d.conv_T : 1;
  const float rcpT = 1.0f / (float)T;
  const int cin = d.conv_cin > 0 ? d.conv_cin : 32;

  using LA = typename std::conditional<A_KC, SkLoadKC<BM, CONV_A>, SkLoadRC<BM, false>>::type;
  using LB = typename std::conditional<B_KC, SkLoadKC<BN, false>, SkLoadRC<BN, CONV_B>>::type;
  LA la; LB lb;

  auto decode = [&](const SkPiece& pc, int& row0, int& col0) {
    int mslot, nt;
    sk_tile_decode(rg.T0 + pc.t, n_mt, p.gw, mslot, nt);
    row0 = (mmap ? mmap[1 + mslot] : mslot) * BM;
    col0 = nt * BN;
  };
  // Conv view on A (k = tap * cin + channel): the K-blocks are walked channel block by channel block, all taps of a 32-channel block
  // in a row.  Tap t of output row r reads input row r + t - pad, so consecutive K-blocks then read the SAME 128-byte lines shifted by
  // one row: the activation tile comes from L2 / the vector cache instead of the fabric 9 times (tap-major, a line's next use is 8
  // K-blocks x ~100 co-resident workgroups away - beyond the XCD's 4 MiB L2).  The sum over K is the same set of products in another
  // order; the B operand's 128-byte segments follow the same k0.
  const int ntap = (CONV_A && p.conv_chan_major) ? d.K / cin : 0;
  auto k0_of = [&](int kb) -> int {
    if (TN && kmap) return (kmap[1 + (kb >> 1)] * 2 + (kb & 1)) * 32;
    if (CONV_A && ntap) { const int cb = kb / ntap; return (kb - cb * ntap) * cin + cb * 32; }
    return kb * 32;
  };

  // ---- loader cursor (runs STAGES - 1 K-blocks ahead of the MFMAs, across piece boundaries)
  int lu = rg.hi;
  SkPiece lp;
  bool have_l = sk_next_piece(lu, rg.lo, nkb, lp);
  int lkb = lp.kb_lo, lk0 = 0, ltap = 0, lkin = 0;
  auto loader_set_piece = [&]() {
    int row0, col0;
    decode(lp, row0, col0);
    if (p.debug & 2) row0 = col0 = 0;
    if constexpr (A_KC) la.set_piece(row0, d.M, d.lda, wave, lane, T);
    else la.set_piece(row0, d.M, d.lda, wave, lane, cin, 0);
    if constexpr (B_KC) lb.set_piece(col0, d.N, d.ldb, wave, lane, T);
    else lb.set_piece(col0, d.N, d.ldb, wave, lane, cin, d.conv_pad);
    lkb = lp.kb_lo;
    lk0 = k0_of(lkb);
    if (CONV_A) { ltap = lk0 / cin; lkin = lk0 - ltap * cin; }
  };
  auto loader_issue = [&](int stage) {
    const unsigned sA = smem_addr + (unsigned)(stage * STAGE_FLOATS + wave * (A_FLOATS / 4)) * 4u;
    const unsigned sB = smem_addr + (unsigned)(stage * STAGE_FLOATS + A_FLOATS + wave * (BN * 32 / 4)) * 4u;
    if constexpr (A_KC) la.issue(ra_src, sA, (unsigned)lk0 * 4u, ltap - d.conv_pad, T);
    else la.issue(ra_src, sA, (unsigned)lk0 * (unsigned)(d.lda * 4), lk0, d.K, 1, 1.f);
    if constexpr (B_KC) lb.issue(rb_src, sB, (unsigned)lk0 * 4u, 0, 1);
    else lb.issue(rb_src, sB, (unsigned)lk0 * (unsigned)(d.ldb * 4), lk0, d.K, T, rcpT);
  };
  auto loader_advance = [&]() {
    ++lkb;
    if (lkb == lp.kb_hi) {
      have_l = sk_next_piece(lu, rg.lo, nkb, lp);
      if (have_l) loader_set_piece();
    } else {
      if (CONV_A && ntap) {                 // next tap of the same channel block, or tap 0 of the next channel block
        ++ltap; lk0 += cin;
        if (ltap == ntap) { ltap = 0; lk0 += 32 - ntap * cin; }
      } else {
        lk0 = k0_of(lkb);
        if (CONV_A) { lkin += 32; if (lkin >= cin) { lkin -= cin; ++ltap; } }
      }
    }
  };

  // prologue: STAGES - 1 blocks in flight
  loader_set_piece();
  int inflight = 0;                     // blocks issued and not yet consumed
#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (have_l) {
      loader_issue(s0);
      loader_advance();
      ++inflight;
    }

  // ---- compute cursor
  int cu = rg.hi;
  SkPiece cp;
  sk_next_piece(cu, rg.lo, nkb, cp);
  int ckb = cp.kb_lo;
  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int stage = 0;
  __builtin_amdgcn_s_waitcnt(0);       // see the note at the end of the loop body
  while (true) {
    // The oldest block in flight must have landed; the younger one (N_DMA instructions per block and wave, the only vector-memory
    // operations in this loop) stays in flight.  Then the barrier: everybody's part of that block is in LDS, and everybody is done
    // reading the stage the next DMA overwrites.
    if (STAGES >= 3 && inflight == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N_DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (have_l) {
      int ls = stage + STAGES - 1;
      ls = ls >= STAGES ? ls - STAGES : ls;
      if (!(p.debug & 1)) loader_issue(ls);
      loader_advance();
    } else {
      --inflight;
    }
    {
      const float* sA = smem + stage * STAGE_FLOATS;
      const float* sB = sA + A_FLOATS;
      float fa[MT][16], fb[NT][16];
#pragma unroll
      for (int i = 0; i < MT; ++i) sk_fetch<A_KC, BM>(sA, wm0 + i * 32, l31, h, fa[i]);
#pragma unroll
      for (int j = 0; j < NT; ++j) sk_fetch<B_KC, BN>(sB, wn0 + j * 32, l31, h, fb[j]);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
#ifdef CTTS_SK_READS_FIRST
      // all fragment reads of the block first (they return in order: the first MFMA only waits for the first pair), then the MFMA chain
      __builtin_amdgcn_sched_group_barrier(0x100, MT * (A_KC ? 4 : 16) + NT * (B_KC ? 4 : 16), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 16 * MT * NT, 0);
#endif
    }
    ++ckb;
    stage = stage + 1 == STAGES ? 0 : stage + 1;
    if (ckb < cp.kb_hi) continue;

    // ---------------- the piece is complete
    int row0, col0;
    decode(cp, row0, col0);
    if (cp.kb_hi < nkb) {
      // contribution: slab (write-through stores) + flag
      const unsigned base = (unsigned)blockIdx.x * (SLAB * 4) + (unsigned)(wave * (SLAB / 4) + lane * 4) * 4u;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sk_u32x4 v;
            v.x = __float_as_uint(acc[i][j][4 * q + 0]); v.y = __float_as_uint(acc[i][j][4 * q + 1]);
            v.z = __float_as_uint(acc[i][j][4 * q + 2]); v.w = __float_as_uint(acc[i][j][4 * q + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_src, base + (unsigned)(((i * NT + j) * 4 + q) * 1024), 0, 16);      // aux 16 = sc1
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store((sk_gu32*)(flags + blockIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (cp.kb_lo > 0) {
        // owner of a cut tile: add the slabs of the workgroups below, nearest first, until the tile's unit 0 is covered
        const int tile_lo = cp.t * nkb;
        const int Ux = (rg.T1 - rg.T0) * nkb;
        int upper = rg.lo;                                   // start of the range that is already summed
        for (int jj = wj - 1; jj >= 0 && upper > tile_lo; --jj) {
          const int blo = sk_bound(g, Ux, jj);
          if (blo >= upper) continue;                        // empty range: that workgroup published nothing
          upper = blo;
          const int src = jj * 8 + xcd;
          if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load((sk_gu32*)(flags + src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins > (1u << 24)) {                    // bounded: report instead of hanging
                __hip_atomic_store((sk_gu32*)(flags + SK_MAX_WG), 1u + (unsigned)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
            __hip_atomic_store((sk_gu32*)(flags + src), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          const unsigned base = (unsigned)src * (SLAB * 4) + (unsigned)(wave * (SLAB / 4) + lane * 4) * 4u;
          // the four loads of an MFMA tile in flight, then their sums (fixed order): as "load, add" hipcc reused one register quad and
          // waited for every load - MT * NT * 4 dependent L2 round trips (~1 us each) per slab on the critical path of every cut tile
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              sk_u32x4 sv[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) sv[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, base + (unsigned)(((i * NT + j) * 4 + q) * 1024), 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                acc[i][j][4 * q + 0] += __uint_as_float(sv[q].x); acc[i][j][4 * q + 1] += __uint_as_float(sv[q].y);
                acc[i][j][4 * q + 2] += __uint_as_float(sv[q].z); acc[i][j][4 * q + 3] += __uint_as_float(sv[q].w);
              }
            }
        }
      }
      if (p.accumulate) {
        gemm_accumulate_lean<MT, NT>(d, acc, d.C, row0, col0, wm0, wn0, l31, h, d.M, d.N);
      } else if (!(p.debug & 4)) {
        gemm_epilogue_auto<MT, NT>(d, acc, d.C, 0, row0, col0, wm0, wn0, l31, h, d.M, d.N);
      }
    }
    // A wait hipcc can see: its scoreboard is empty when control returns to the K loop, so it never has a reason to put an
    // `s_waitcnt vmcnt(0)` of its own into the loop (it did, in front of a fragment read whose destination registers the epilogue's
    // loads had used - and that wait also drains the DMA of the next K-block).  It must sit in front of the loop exit as well: the exit
    // edge shares the latch block with the back edge.  Costs one drained prefetch per piece.  tools/check_sk_isa.py checks the ISA.
    if ((p.debug & 16) && blockIdx.x == 8 && tid == 0) {         // tools: shader cycles and 100 MHz wall ticks of one workgroup's life so far
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws + SK_MAX_WG + 2);
      o[0] = clock64() - dbg_c0;
      o[1] = wall_clock64() - dbg_w0;
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (!sk_next_piece(cu, rg.lo, nkb, cp)) break;
    ckb = cp.kb_lo;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
}

template <bool A_KC, bool B_KC, bool CONV>
int sk_launch(const ctts_gemm_desc& d, const SkArgs& p, int grid, int stages, int mt, int nt, hipStream_t st) {
#define SK_GO(S, MT_, NT_) hipLaunchKernelGGL((gemm_sk_kernel<A_KC, B_KC, CONV, S, MT_, NT_>), dim3(grid), dim3(256), 0, st, d, p)
  if (mt == 1 && nt == 1) { if (stages == 3) SK_GO(3, 1, 1); else SK_GO(2, 1, 1); }
  else if (mt == 1 && nt == 2) SK_GO(2, 1, 2);
  else if (mt == 2 && nt == 2) SK_GO(2, 2, 2);
  else if (mt == 1 && nt == 4) {
    if constexpr (A_KC && B_KC) SK_GO(2, 1, 4);       // the wide tile is only routed for NT launches (sk_try)
    else { ctts_set_error("ctts_gemm(stream-K): the 64x256 tile is instantiated for the NT layout only"); return -1; }
  }
  else { ctts_set_error("ctts_gemm(stream-K): tile %dx%d not instantiated", mt, nt); return -1; }
#undef SK_GO
  CTTS_CHECK_LAUNCH("ctts_gemm(stream-K)");
  return 1;
}

int sk_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

// One workspace per stream for everything that hands partial results between workgroups (layout: ctts_common.h): stream-K flags + slabs,
// split-K tickets + slabs, tickets + partials of the ordered column reductions.
extern "C" size_t ctts_gemm_workspace_bytes(void) { return CTTS_WS_BYTES; }
extern "C" size_t ctts_workspace_bytes(void) { return CTTS_WS_BYTES; }

// The persistent kernel's schedule and its slab hand-off assume the default (SPX) dispatch: workgroup b of a launch runs on XCD b % 8
// (sk_plan.h cuts the unit space per XCD; a cut tile's owner and its contributors share an XCD and therefore an L2).  This probe makes
// that checkable: workgroup b writes the hardware's XCC_ID into out[b]; the host compares with b % 8 (kernels.gemm_workspace does it
// once per device and refuses to enable the persistent kernel under any other partition mode).
namespace {
__global__ void xcd_probe_kernel(int32_t* out) {
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    out[blockIdx.x] = (int32_t)(id & 0xF);
  }
}
}  // namespace
extern "C" int ctts_xcd_probe(int32_t* out, int nblocks, void* stream) {
  CTTS_REQUIRE(out && nblocks > 0, "ctts_xcd_probe: bad arguments");
  hipLaunchKernelGGL(xcd_probe_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, out);
  CTTS_CHECK_LAUNCH("ctts_xcd_probe");
  return 0;
}
extern "C" const uint32_t* ctts_workspace_error_word(const void* ws) {
  return ws ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(ws) + CTTS_WS_SK_FLAGS) + SK_MAX_WG : nullptr;
}
static_assert((size_t)SK_FLAG_WORDS * 4 <= CTTS_WS_GEMM_TICKETS, "stream-K flag words overlap the split-K tickets");
static_assert((size_t)SK_SLAB_FLOATS_MAX <= CTTS_WS_SLAB_FLOATS, "stream-K slabs exceed the workspace slab area");

// launch == false: only answer whether the persistent kernel WOULD take this descriptor (ctts_gemm_takes_persistent)
static int sk_try(const ctts_gemm_desc& din, hipStream_t st, bool launch) {
  static const int enabled = sk_env("CTTS_SK", 1);
  static const int stages = sk_env("CTTS_SK_STAGES", 2) == 3 ? 3 : 2; // LDS stages
  static const int force_w = sk_env("CTTS_SK_W", 0);                  // workgroups per XCD (0 = by LDS footprint)
  static const int min_units = sk_env("CTTS_SK_MIN_UNITS", 4096);     // below this the launch is latency bound either way
  static const int split_from = sk_env("CTTS_SK_SPLIT_NKB", 24);      // tiles are cut only when K has at least this many blocks
  static const int force_gw = sk_env("CTTS_SK_GW", 0);
  static const int min_nkb = sk_env("CTTS_SK_MIN_NKB", 64);           // K >= 2048: with shorter reductions the tile-per-workgroup kernels win
  static const int tile_cfg = sk_env("CTTS_SK_TILE", 22);             // 11: 64x64, 12: 64x128, 22: 128x128 workgroup tiles
  static const int max_split = sk_env("CTTS_SK_MAX_SPLIT", 2);        // tiles * max_split >= grid: a tile is cut in two or three, never more (the owner gathers serially)
  static const int debug = sk_env("CTTS_SK_DEBUG", 0);
  static const int wg_units = sk_env("CTTS_SK_WG_UNITS", 16);         // a workgroup gets at least this many (tile, K-block) units
  const ctts_gemm_desc& d = din;
  if (!enabled || !d.sk_ws || d.sk_ws_bytes < (int64_t)ctts_gemm_workspace_bytes()) return 0;
  if (d.nb0 * d.nb1 != 1 || (d.lens && (d.lim_m || d.lim_n || d.lim_k)) || d.E) return 0;
  if (!d.a_kc && d.b_kc) return 0;
  const bool tn = !d.a_kc && !d.b_kc;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(d.A) || !al16(d.B) || ((d.lda | d.ldb) & 3)) return 0;
  if (d.a_kc ? (d.K % 32 != 0) : (d.M % 4 != 0)) return 0;
  if (d.b_kc ? (d.K % 32 != 0) : (d.N % 4 != 0)) return 0;
  const bool conv = d.conv_T > 0;
  if (conv) {
    if (d.conv_cin % 32 != 0 || d.conv_T < 128) return 0;
    if (d.conv_on_b ? !tn : !d.a_kc) return 0;
    if (tn && (long)d.K >= (1L << 24)) return 0;
  }
  const long a_ext = d.a_kc ? ((long)(d.M + 128) * d.lda + d.K) : ((long)(d.K + 64) * d.lda + d.M);
  const long b_ext = d.b_kc ? ((long)(d.N + 128) * d.ldb + d.K) : ((long)(d.K + 64) * d.ldb + d.N);
  if (a_ext * 4 >= 0x7FFF0000L || b_ext * 4 >= 0x7FFF0000L) return 0;
  if (d.row_lens && !d.tile_map) return 0;                           // padded-row skipping needs the device-built schedule here
  if (d.tile_map == reinterpret_cast<const int32_t*>(1)) return 0;

  int mt = tile_cfg / 10, nt = tile_cfg % 10;
  if (!((mt == 1 && nt == 1) || (mt == 1 && nt == 2) || (mt == 2 && nt == 2))) { mt = 1; nt = 1; }
  if (mt == 2 && d.a_kc && d.tile_map) mt = 1;                       // the m-tile schedule is built for 64-row tiles
  if (nt == 2 && d.N < 128) nt = 1;
  if (mt == 2 && d.M < 128) mt = 1;
  if (mt == 2 && nt == 1) nt = (d.N >= 128) ? 2 : 1;
  if (mt == 2 && nt == 1) mt = 1;
  // ragged rows keep 64-row tiles: make the tile 256 columns wide instead (1 x 4 MFMA tiles per wave: the same 64 MFMAs per K-block and
  // barrier as the 128 x 128 tile, 5 fragment reads per 64 MFMAs instead of 6; 80 KB of LDS: exactly two workgroups per CU).  FFN conv
  // forward 491 -> 467 us, its data gradient 478 -> 457 us, fs2 step -1.05 % (same box); CTTS_SK_WIDE=0 keeps 64 x 128.
  static const int wide = sk_env("CTTS_SK_WIDE", 1);
  if (wide && mt == 1 && nt == 2 && d.a_kc && d.b_kc && d.N % 256 == 0) nt = 4;
  const int BM = 64 * mt, BN = 64 * nt;
  SkArgs p;
  p.tiles_m = (d.M + BM - 1) / BM;
  p.tiles_n = (d.N + BN - 1) / BN;
  p.nkb = (d.K + 31) / 32;
  const long units = (long)p.tiles_m * p.tiles_n * p.nkb * mt * nt;  // in 64x64x32 equivalents
  if (units < min_units || p.nkb < min_nkb) return 0;
  p.whole_tiles = p.nkb < split_from ? 1 : 0;
  static const int chan_major = sk_env("CTTS_SK_CONV_ORDER", 1);      // 1: (channel block, tap) K order for conv views on A; 0: (tap, channel)
  p.conv_chan_major = (conv && d.a_kc && !d.conv_on_b && chan_major && d.K % d.conv_cin == 0) ? 1 : 0;
  // ABI: split_k > 1 means "add alpha * A B to C" (plain read-modify-write by the tile's owner) - unless split_overwrite asks for C = ...
  p.accumulate = (d.split_k > 1 && !d.split_overwrite) ? 1 : 0;
  p.debug = debug;
  // schedule groups: 4 groups of n-tiles when that divides (an XCD pair shares a group), else one group
  p.gw = (p.tiles_n % 4 == 0) ? p.tiles_n / 4 : p.tiles_n;
  if (force_gw > 0 && p.tiles_n % force_gw == 0) p.gw = force_gw;
  p.ws = reinterpret_cast<unsigned*>(d.sk_ws);
  // grid: W workgroups per XCD = what the LDS footprint keeps resident with a margin (160 KB per CU, 32 CUs per XCD), fewer when the
  // launch is small (>= 16 units each)
  const int lds = stages * (BM + BN) * 128;
  int per_cu = 160 * 1024 / lds;
  if (per_cu > 4) per_cu = 4;
  if (mt * nt >= 4 && per_cu > 2) per_cu = 2;
  if (mt * nt == 2 && per_cu > 3) per_cu = 3;
  long W = force_w > 0 ? force_w : per_cu * 32;
  const long Wu = (long)p.tiles_m * p.tiles_n * p.nkb / (8 * wg_units);
  if (W > Wu) W = Wu;
  if (W < 1) W = 1;
  const int grid = (int)W * 8;
  // a tile is cut in two (the owner gathers serially) - up to four pieces when every piece keeps a long reduction (>= 64 K-blocks: the
  // FFN weight gradient, 144 tiles x 512 K-blocks, measured 509 -> 494 us against the tile kernels with split-K atomics)
  const int cuts = (p.nkb >= 256 && max_split < 4) ? 4 : max_split;
  if ((long)p.tiles_m * p.tiles_n * cuts < grid) return 0;
  if (grid > SK_MAX_WG || (long)grid * BM * BN > SK_SLAB_FLOATS_MAX) return 0;
  if (!launch) return 1;
  if (d.a_kc && d.b_kc) return conv ? sk_launch<true, true, true>(d, p, grid, stages, mt, nt, st) : sk_launch<true, true, false>(d, p, grid, stages, mt, nt, st);
  if (d.a_kc && !d.b_kc) return conv ? sk_launch<true, false, true>(d, p, grid, stages, mt, nt, st) : sk_launch<true, false, false>(d, p, grid, stages, mt, nt, st);
  return conv ? sk_launch<false, false, true>(d, p, grid, stages, mt, nt, st) : sk_launch<false, false, false>(d, p, grid, stages, mt, nt, st);
}

int ctts_gemm_sk_try(const ctts_gemm_desc& d, hipStream_t st) { return sk_try(d, st, true); }

extern "C" int ctts_gemm_takes_persistent(const ctts_gemm_desc* d) {
  if (!d) return 0;
  ctts_gemm_desc c = *d;
  if (c.nb0 < 1) c.nb0 = 1;
  if (c.nb1 < 1) c.nb1 = 1;
  return sk_try(c, nullptr, false) > 0 ? 1 : 0;
}
