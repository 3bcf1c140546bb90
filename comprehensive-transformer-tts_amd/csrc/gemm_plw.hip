// fp32 WEIGHT GRADIENTS on the BF16 matrix pipe from pre-split operands: C[m][n] (+)= alpha * sum_k A[k][m] * B[k + shift(n)][n'] - the TN
// form of ctts_gemm (both operands reduction-major, optional im2col view on B: reference Conv1d weight gradients of
// transformer_fs2.py:220-239 / modules.py:140-148) as a persistent stream-K kernel whose main loop is LDS-DMA + ds_read_b64_tr_b16 + MFMA.
//
// gemm_x6tn_kernel (gemm.hip) reads both fp32 operands with 64 dword loads per thread and K-block (a bf16 MFMA fragment is 8 consecutive k
// of one column, which lie a row apart in memory), splits them into their three bf16 pieces in registers and stores the pieces into a
// single LDS stage between two barriers: 132 TFLOP/s on the valid rows of the FFN conv weight gradient, where the forward kernel on
// planes reaches 190.  This kernel needs NO transposed copy and no split of its own: it consumes the ROW-MAJOR plane sets
// ([rows][cols / 32][3][32] bf16, ctts_split_planes) the step has already made - dZ's for the data-gradient launch, x's for the forward
// launch - and lets the LDS do the transpose:
//   * LDS image of a 32-row K-block: per piece a [32 k][128 m] (A) / [32 k][256 n] (B) bf16 matrix, k-major exactly as in HBM, written by
//     `buffer_load_dwordx4 ... lds` (per lane a free gather address, lane-linear LDS side: one instruction = 4 A rows or 2 B rows);
//   * fragments with ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 m] block (lane i supplies the 8-byte address of row i >> 2,
//     columns 4 (i & 3) ..+3) and lane i receives column i of it - 4 consecutive k of one m, two reads per 8-deep MFMA operand.  A half
//     wave touches 4 k-rows x 64 bytes; the 64-byte windows of a row are XOR-ed with (k & 3) - on the DMA's SOURCE address - so that the
//     four rows fall into four different bank groups;
//   * the conv view is a ROW shift of B (tap - pad rows; cin % 256 == 0): it costs nothing but a per-lane validity test (rows shifted
//     across an utterance boundary read the hardware's out-of-range zero).  One-tap tiles (256 channels of one tap) when a K-block may
//     straddle two utterances (T % 32 != 0); otherwise the tiles are FOLDED over the taps - 8 / 4 / 2 taps x 32 / 64 / 128 channels
//     share one B image of 32 + taps - 1 rows (tile_kind below): a sixth of the B traffic at k = 9;
//   * ragged (b, t) rows: the K-blocks beyond an utterance's length are not part of the unit space at all (their dZ rows are zero by
//     construction - the rule gemm_x6tn_kernel relies on); every workgroup builds the prefix sums of the active K-blocks from row_lens;
//   * tile 128 x 256, 8 waves (2 x 4), six cross terms smallest first, two 72 KB stages, the mid-block barrier and the rotated
//     instruction order of the upper wave group, the even unit partition and the fixed-order slab hand-off: as in gemm_pl.hip;
//   * epilogue: C = alpha * acc or C += alpha * acc (what split_k > 1 without split_overwrite means in ctts_gemm) by the tile's owner -
//     the weight gradient is added to param.grad in place, in a fixed order, without partial matrices and without a reduce launch.
// Eligibility: plw_try below; everything else stays on gemm_x6tn_kernel / the fp32 kernels.
#include "gemm_pl_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef short plw_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) plw_s16x4 plw_lds_s16x4;

constexpr int PLW_A_PLANE = 32 * PL_BM * 2;          // one piece of the A image: [32 k][128 m] bf16 = 8 KB
constexpr int PLW_B_PLANE = 32 * PL_BN * 2;          // [32 k][256 n] bf16 = 16 KB
static_assert(3 * (PLW_A_PLANE + PLW_B_PLANE) == PL_STAGE, "same stage size as gemm_pl_kernel");

struct PlwArgs {
  int tiles_m, tiles_n;      // 128 x 256 tiles of the [M, N] output
  int nkb;                   // 32-row K-blocks of the reduction (dense; ragged: computed in the kernel from row_lens)
  int nutt, kbu;             // ragged rows: utterances and K-blocks per utterance (row_T / 32); nutt = 0: dense
  int fold_tt, fold_tiles;   // conv, conv_T % 32 == 0: the first `fold_tiles` n-tiles are FOLDED - 256 columns = fold_tt taps x 256 / fold_tt channels,
                             // served by ONE B image of 32 + fold_tt - 1 rows (fold_tt = 1: no folding); the remaining taps: one tap per tile
  int accumulate;            // 1: C += alpha * acc
  int debug;                 // CTTS_PL_DEBUG bits of gemm_pl.hip (tools builds)
  unsigned* ws;
};

// 24 x (one MFMA, one LDS read) in program order for the scheduling region that ends here (masks: 0x008 MFMA, 0x100 DS read)
#define PLW_SGB2() __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)
#define PLW_SGB8() PLW_SGB2(); PLW_SGB2(); PLW_SGB2(); PLW_SGB2()
#define PLW_INTERLEAVE_24() PLW_SGB8(); PLW_SGB8(); PLW_SGB8(); PLW_SGB8(); PLW_SGB8(); PLW_SGB8()

#define PLW_SGB12() __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0)
#define PLW_INTERLEAVE_4x2() PLW_SGB12(); PLW_SGB12(); PLW_SGB12(); PLW_SGB12()

struct PlwFrag { pl_u32x4 a[2][3], b[2][3]; };       // one 16-deep k-step: [MFMA row / column tile][piece]

// TERMS: 6 = the six cross terms; 1 = the "amp" arithmetic of gemm_pl_kernel (hi pieces only)
template <bool CONV, int TERMS>
__global__ __launch_bounds__(512, 2) void gemm_plw_kernel(const ctts_gemm_desc d, const PlwArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PL_STAGE];
  __shared__ int s_pref[PL_MAX_UTT + 1];          // active K-blocks of the utterances before b

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 64;
  const int nutt = p.nutt;
  constexpr int NQ = TERMS == 1 ? 1 : 3;          // pieces moved and read

  int nkb = p.nkb;
  if (nutt > 0) {
    // counts to LDS first (one global load per thread), then the prefix sums from LDS (see gemm_pl_kernel: t dependent global loads
    // per thread cost ~18 us at the head of the launch)
    for (int t = tid; t < nutt; t += 512) {
      const int L = d.row_lens[t] + d.row_halo;
      s_pref[t + 1] = L <= 0 ? 0 : min(p.kbu, (L + 31) >> 5);
    }
    if (tid == 0) s_pref[0] = 0;
    __syncthreads();
    int mine = 0;
    if (tid <= nutt)
      for (int b = 0; b <= tid; ++b) mine += s_pref[b];
    __syncthreads();
    if (tid <= nutt) s_pref[tid] = mine;
    __syncthreads();
    nkb = s_pref[nutt];
  }
  const int n_tiles = p.tiles_m * p.tiles_n;
  if (nkb <= 0) {                                  // empty reduction: C = 0 (or unchanged)
    if (!p.accumulate)
      for (long e = (long)blockIdx.x * 512 + tid; e < (long)d.M * d.N; e += (long)gridDim.x * 512) d.C[(e / d.N) * d.ldc + e % d.N] = 0.f;
    return;
  }
  SkGeom g{n_tiles, nkb, (int)(gridDim.x >> 3), 0};
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  const SkRange rg = sk_range(g, xcd, wj);
  if (rg.hi <= rg.lo) return;

  const int T = CONV ? d.conv_T : 0x3FFFFFFF;
  const int cin = CONV ? d.conv_cin : 1;
  const pl_i32x4 ra_src = pl_make_rsrc(d.A_planes);
  const pl_i32x4 rb_src = pl_make_rsrc(d.B_planes - (CONV ? (long)d.conv_pad * d.ldb * 3 : 0));      // row 0 of the resource = row -pad
  const unsigned lda2 = (unsigned)(d.lda * 6), ldb2 = (unsigned)(d.ldb * 6);
  const unsigned smem_addr = (unsigned)reinterpret_cast<uintptr_t>(smem);
  unsigned* flags = p.ws;
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(p.ws) + CTTS_WS_SLABS);
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, 0x7FFFFFFE, 0x00020000);

  auto decode = [&](const SkPiece& pc, int& row0, int& col0) {
    int mslot, nt;
    sk_tile_decode(rg.T0 + pc.t, p.tiles_m, p.tiles_n, mslot, nt);
    row0 = __builtin_amdgcn_readfirstlane(mslot * PL_BM);
    col0 = __builtin_amdgcn_readfirstlane(nt * PL_BN);
  };

  // ---- n-tile -> (first tap, taps in the tile, first channel, channels per tap).  Folded tiles (conv, T % 32 == 0): the taps of a layer
  //      read the SAME rows of x shifted by one row per tap, so a tile of 8 taps x 32 channels needs a B image of 39 rows x 64 bytes
  //      (7.5 KB for the three pieces) where 8 one-tap tiles of 256 channels move 8 x 48 KB: the B traffic of the launch falls to a
  //      sixth (k = 9: 8 folded + 1 one-tap tile per 256 channels), and every K-block comes from the MALL / HBM here.
  auto tile_kind = [&](int nt, int& tap0, int& tt, int& c0, int& cw) {
    if (CONV && nt < p.fold_tiles) {
      tt = p.fold_tt; cw = PL_BN / tt; tap0 = 0; c0 = nt * cw;
    } else if (CONV) {
      const int r = nt - p.fold_tiles, cpt = cin / PL_BN;
      tt = 1; cw = PL_BN; tap0 = (p.fold_tiles > 0 ? p.fold_tt : 0) + r / cpt; c0 = (r % cpt) * PL_BN;
    } else {
      tt = 1; cw = PL_BN; tap0 = 0; c0 = nt * PL_BN;
    }
  };
  // 64-byte windows of a B row are XOR-ed with bits of the row index so that the four rows of a half wave's transpose read fall into
  // different bank groups: rows of 512 / 256 bytes (all four on the same banks): row & 3; 128 bytes (rows 0, 2 collide): (row >> 1) & 1;
  // 64 bytes (four rows = 256 contiguous bytes): nothing
  auto b_swz = [](int row, int rs) { return rs >= 256 ? (row & 3) : (rs == 128 ? ((row >> 1) & 1) : 0); };

  // ---- loader: wave w moves k-rows 4 w .. 4 w + 3 of the K-block: the three A pieces (one instruction each: 4 rows x 256 bytes) and the
  //      three B pieces (two instructions each: 2 rows x 512 bytes).  LDS position (row, 16-byte chunk c') holds the logical chunk whose
  //      64-byte window index is (c' >> 2) ^ (row & 3).
  const int arow = 4 * wave + (lane >> 4);
  const int achunk = ((((lane & 15) >> 2) ^ (lane >> 4)) << 2) | (lane & 3);
  const unsigned voffA_lane = (unsigned)arow * lda2 + (unsigned)((achunk >> 2) * 192 + (achunk & 3) * 16);
  // B image: rows of `rs` bytes back to back, instruction n of a piece covers bytes [1024 n, 1024 n + 1024); wave w issues n = w, w + 8
  int brow[2];
  unsigned voffB_lane[2];
  bool bact[2];
  int l_ni = 16;                       // DMA instructions that hold rows of the image (regular: 16 per piece)
  auto loader_b_lanes = [&](int rs, int rows_needed) {
    const int rs_log = 31 - __builtin_clz(rs);
    l_ni = (rows_needed * rs + 1023) >> 10;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pos = (wave + 8 * j) * 1024 + lane * 16;
      const int row = pos >> rs_log, cp = (pos & (rs - 1)) >> 4;
      const int c = ((((cp >> 2) ^ b_swz(row, rs)) << 2) | (cp & 3));
      brow[j] = row;
      bact[j] = row < rows_needed;
      voffB_lane[j] = (unsigned)row * ldb2 + (unsigned)((c >> 2) * 192 + (c & 3) * 16);
    }
  };
  loader_b_lanes(PL_BN * 2, 32);
  int lu = rg.hi;
  SkPiece lp;
  bool have_l = sk_next_piece(lu, rg.lo, nkb, lp);
  int lkb = lp.kb_lo;
  int l_b = 0, l_kbu = 0;               // ragged cursor: utterance, K-block inside it
  int l_row32 = 0, l_tbase = 0;         // first row of the K-block; its time index inside the utterance (conv)
  unsigned l_soffA = 0, l_soffB = 0;
  int l_shift = 0;                      // conv: tap - pad of the piece's n-tile
  auto loader_set_piece = [&]() {
    int row0, col0;
    decode(lp, row0, col0);
    l_soffA = (unsigned)(row0 >> 5) * 192u;
    if (CONV) {
      int tap0, tt, c0, cw;
      tile_kind(col0 / PL_BN, tap0, tt, c0, cw);
      l_shift = tap0 - d.conv_pad;
      l_soffB = (unsigned)(c0 >> 5) * 192u + (unsigned)tap0 * ldb2;
      if (p.fold_tiles > 0) loader_b_lanes(cw * 2, 32 + tt - 1);
    } else {
      l_soffB = (unsigned)(col0 >> 5) * 192u;
    }
    lkb = lp.kb_lo;
    if (nutt > 0) {
      int lo = 0, hi = nutt;              // s_pref[lo] <= lkb < s_pref[hi]
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pref[mid] <= lkb) lo = mid; else hi = mid; }
      l_b = lo;
      l_kbu = lkb - s_pref[lo];
      l_row32 = l_b * d.row_T + 32 * l_kbu;
      l_tbase = 32 * l_kbu;
    } else {
      l_row32 = 32 * lkb;
      l_tbase = CONV ? l_row32 % T : 0;
    }
    l_b = __builtin_amdgcn_readfirstlane(l_b); l_kbu = __builtin_amdgcn_readfirstlane(l_kbu);
    l_row32 = __builtin_amdgcn_readfirstlane(l_row32); l_tbase = __builtin_amdgcn_readfirstlane(l_tbase);
  };
  auto loader_issue = [&](int stage) {
    const int rows_left = d.K - l_row32;
    const unsigned vA = arow < rows_left ? voffA_lane + (unsigned)l_row32 * lda2 : PL_OOB;
    unsigned vB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bool ok = bact[j] && (brow[j] >= 32 || brow[j] < rows_left);       // rows >= 32: the extra rows of a folded image (T % 32 == 0)
      if (CONV) {
        int t = l_tbase + brow[j];
        t = (brow[j] < 32 && t >= T) ? t - T : t;                           // (a K-block may straddle two utterances only unfolded)
        ok = ok && (unsigned)(t + l_shift) < (unsigned)T;
      }
      vB[j] = ok ? voffB_lane[j] + (unsigned)l_row32 * ldb2 : PL_OOB;
    }
    const unsigned sA = smem_addr + (unsigned)(stage * PL_STAGE + wave * 1024);
    const unsigned sB = smem_addr + (unsigned)(stage * PL_STAGE + 3 * PLW_A_PLANE + wave * 1024);       // instruction n = wave + 8 j
#pragma unroll
    for (int q = 0; q < NQ; ++q) pl_dma16(ra_src, sA + q * PLW_A_PLANE, vA, l_soffA + q * 64);
    if (!PL_DBG(64)) {                 // tools: 64 = no B DMA after the prologue (what would a smaller B image buy?)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (wave + 8 * j < l_ni) {         // wave-uniform: a folded image is 3 - 5 instructions per piece, not 16
#pragma unroll
        for (int q = 0; q < NQ; ++q) pl_dma16(rb_src, sB + q * PLW_B_PLANE + j * 8192, vB[j], l_soffB + q * 64);
      }
    }
  };
  auto loader_advance = [&]() {
    ++lkb;
    if (lkb == lp.kb_hi) {
      have_l = sk_next_piece(lu, rg.lo, nkb, lp);
      if (have_l) loader_set_piece();
    } else if (nutt > 0) {
      ++l_kbu;
      if (s_pref[l_b] + l_kbu == s_pref[l_b + 1]) {
        do { ++l_b; } while (l_b + 1 < nutt && s_pref[l_b + 1] == s_pref[l_b]);
        l_kbu = 0;
      }
      l_b = __builtin_amdgcn_readfirstlane(l_b);
      l_row32 = l_b * d.row_T + 32 * l_kbu;
      l_tbase = 32 * l_kbu;
    } else {
      l_row32 += 32;
      if (CONV) { l_tbase += 32; if (l_tbase >= T) l_tbase -= T; }
    }
  };

  // ---- fragments: lane = (h: k half of the 16-deep step, g1: 16-column half of the 32-wide MFMA tile, i16); the lane's transpose-read
  //      address is row (i16 >> 2) of a 4-row group, columns 16 g1 + 4 (i16 & 3) ..+3 of the tile's 64-byte window
  const int i16 = lane & 15, fsw = i16 >> 2;
  const int fcol2 = 2 * (16 * ((lane >> 4) & 1) + 4 * (i16 & 3));
  int fa_off[2], fb_off[2];
  int fb_rs = PL_BN * 2;             // row stride of the current piece's B image (bytes)
  int c_col[2];                      // first output column of the wave's two 32-column MFMA tiles
  // B fragment base of MFMA column tile j of this wave for an n-tile of kind (tap0, tt, c0, cw): tile jj = 2 (wave & 3) + j of the 8 holds
  // tap (32 jj) / cw, channels (32 jj) % cw ..+31; its rows sit `tap` rows further down the shared image
  auto set_b_frag = [&](int nt) {
    int tap0, tt, c0, cw;
    tile_kind(nt, tap0, tt, c0, cw);
    fb_rs = cw * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int jj = (wn0 >> 5) + j;
      const int tap_l = (32 * jj) >> (31 - __builtin_clz(cw)), ch_l = (32 * jj) & (cw - 1);
      const int row = fsw + tap_l;
      fb_off[j] = 3 * PLW_A_PLANE + (8 * h + row) * fb_rs + ((((ch_l >> 5)) ^ b_swz(row, fb_rs)) << 6) + fcol2;
      c_col[j] = (CONV ? (tap0 + tap_l) * cin + c0 + ch_l : c0 + 32 * jj);
    }
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) fa_off[i] = (8 * h + fsw) * (PL_BM * 2) + ((((wm0 >> 5) + i) ^ fsw) << 6) + fcol2;
  auto read_frag = [&](int stage, int ks, PlwFrag& f) {
    const unsigned char* base = smem + stage * PL_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const unsigned char* pa = base + fa_off[i] + q * PLW_A_PLANE + (16 * ks) * (PL_BM * 2);
        const plw_s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((plw_lds_s16x4*)(pa));
        const plw_s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((plw_lds_s16x4*)(pa + 4 * (PL_BM * 2)));
        const unsigned long long u0 = __builtin_bit_cast(unsigned long long, x0), u1 = __builtin_bit_cast(unsigned long long, x1);
        f.a[i][q] = pl_u32x4{(unsigned)u0, (unsigned)(u0 >> 32), (unsigned)u1, (unsigned)(u1 >> 32)};
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int rs = CONV ? fb_rs : PL_BN * 2;             // (compile-time without the conv view: immediates)
        const unsigned char* pb = base + fb_off[j] + q * PLW_B_PLANE + (16 * ks) * rs;
        const plw_s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((plw_lds_s16x4*)(pb));
        const plw_s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((plw_lds_s16x4*)(pb + 4 * rs));
        const unsigned long long u0 = __builtin_bit_cast(unsigned long long, x0), u1 = __builtin_bit_cast(unsigned long long, x1);
        f.b[j][q] = pl_u32x4{(unsigned)u0, (unsigned)(u0 >> 32), (unsigned)u1, (unsigned)(u1 >> 32)};
      }
  };
  floatx16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto mma_terms = [&](const PlwFrag& f, int t0, int t1) {
    if constexpr (TERMS == 1) {          // hi x hi only
      if (t0 <= 5 && 5 < t1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = pl_mma(f.a[i][0], f.b[j][0], acc[i][j]);
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      if (t < t0 || t >= t1) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = pl_mma(f.a[i][PA[t]], f.b[j][PB[t]], acc[i][j]);
    }
  };

  // ---- prologue: blocks 0 and 1 in flight
  loader_set_piece();
  loader_issue(0);
  loader_advance();
  if (have_l) {
    loader_issue(1);
    loader_advance();
  }
  int cu = rg.hi;
  SkPiece cp;
  sk_next_piece(cu, rg.lo, nkb, cp);
  int ckb = cp.kb_lo;
  { int r0_, c0_; decode(cp, r0_, c0_); set_b_frag(c0_ / PL_BN); }
  zero_acc();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // the K loop in two instruction orders (lower / upper wave group), as in gemm_pl_kernel
  auto run = [&](auto skew_c) {
    constexpr bool SKEW = decltype(skew_c)::value;
    PlwFrag f0, f1;
    read_frag(0, 0, f0);
    int stage = 0;
    __builtin_amdgcn_s_waitcnt(0);
    while (true) {
      const bool do_mma = PL_DBG(8) == 0;
      // One fragment read per MFMA (sched_group_barrier): hipcc otherwise emits the 24 reads of a half block as two bursts, during which
      // the wave issues no MFMA - the matrix pipe then depends on the SIMD's other wave being in an MFMA phase at that moment (measured
      // without DMA: 314 us with the bursts, 253 us with no reads at all, dense FFN shape).
      if (PL_DBG(256)) { if (do_mma) mma_terms(f0, 0, 6); }
      else {
        read_frag(stage, 1, f1);
        if (do_mma) mma_terms(f0, 0, 6);
        if constexpr (TERMS == 1) { PLW_INTERLEAVE_4x2(); } else { PLW_INTERLEAVE_24(); }
      }
      __builtin_amdgcn_sched_barrier(0);
      // lgkmcnt(0) as a wait hipcc can SEE (0xC07F = vmcnt 63, expcnt 7, lgkmcnt 0): inside the asm it left the compiler's scoreboard with the
      // fragment reads of k-step 1 still "pending", and - the counter being in order - every second-half MFMA on them then waited for the
      // NEWER reads of the next block's fragments as well (s_waitcnt lgkmcnt(5 .. 0) in front of the first six MFMAs after the barrier)
      __builtin_amdgcn_s_waitcnt(0xC07F);
      if (PL_DBG(128)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tools: 128 = no barrier in the loop (timing only)
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      const bool more = ckb + 1 < cp.kb_hi;
      // (read unconditionally: when this block ends the piece the values are dead - f0 is read again behind the epilogue - and the reads
      //  and the MFMAs stay in ONE scheduling region)
      if (!PL_DBG(256)) read_frag(stage ^ 1, 0, f0);
      if (do_mma) mma_terms(f1, 0, 6);
      if constexpr (TERMS == 1) { PLW_INTERLEAVE_4x2(); } else { PLW_INTERLEAVE_24(); }
      __builtin_amdgcn_sched_barrier(0);
      if (have_l && !PL_DBG(1)) loader_issue(stage);
      if (have_l) loader_advance();
      ++ckb;
      stage ^= 1;
      if (ckb < cp.kb_hi) continue;

      // ---------------- the piece is complete (see gemm_pl_kernel for the laundering of the lane constants and of the descriptor)
      int e_l31 = l31, e_h = h, e_wm0 = wm0, e_wn0 = wn0;
      unsigned long long kargs = (unsigned long long)(const void*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+v"(e_l31), "+v"(e_h), "+s"(e_wm0), "+s"(e_wn0), "+s"(kargs));
      const ctts_gemm_desc& dc = *(const ctts_gemm_desc*)(const __attribute__((address_space(4))) ctts_gemm_desc*)kargs;
      int row0, col0;
      decode(cp, row0, col0);
      if (cp.kb_hi < nkb) {
        const unsigned base = (unsigned)blockIdx.x * (PL_SLAB * 4) + (unsigned)(wave * (PL_SLAB / 8) + lane * 4) * 4u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              pl_u32x4 v;
              v.x = __float_as_uint(acc[i][j][4 * q + 0]); v.y = __float_as_uint(acc[i][j][4 * q + 1]);
              v.z = __float_as_uint(acc[i][j][4 * q + 2]); v.w = __float_as_uint(acc[i][j][4 * q + 3]);
              __builtin_amdgcn_raw_buffer_store_b128(v, rs_src, base + (unsigned)(((i * 2 + j) * 4 + q) * 1024), 0, 16);      // aux 16 = sc1
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((pl_gu32*)(flags + blockIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (cp.kb_lo > 0) {
          const int tile_lo = cp.t * nkb;
          const int Ux = (rg.T1 - rg.T0) * nkb;
          int upper = rg.lo;
          for (int jj = wj - 1; jj >= 0 && upper > tile_lo; --jj) {
            const int blo = sk_bound(g, Ux, jj);
            if (blo >= upper) continue;
            upper = blo;
            const int src = jj * 8 + xcd;
            if (tid == 0) {
              unsigned spins = 0;
              while (__hip_atomic_load((pl_gu32*)(flags + src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 24)) {
                  __hip_atomic_store((pl_gu32*)(flags + PL_MAX_WG), 1u + (unsigned)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  break;
                }
              }
              __hip_atomic_store((pl_gu32*)(flags + src), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned base = (unsigned)src * (PL_SLAB * 4) + (unsigned)(wave * (PL_SLAB / 8) + lane * 4) * 4u;
            // all 16 loads of the slab in flight, THEN the sums (in the fixed order): written as "load, add" hipcc reused one register quad and
            // waited for every load - 16 dependent L2 round trips, ~16 us per slab, on the critical path of every cut tile
            pl_u32x4 sv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) sv[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, base + (unsigned)(e * 1024), 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const pl_u32x4 v = sv[(i * 2 + j) * 4 + q];
                  acc[i][j][4 * q + 0] += __uint_as_float(v.x); acc[i][j][4 * q + 1] += __uint_as_float(v.y);
                  acc[i][j][4 * q + 2] += __uint_as_float(v.z); acc[i][j][4 * q + 3] += __uint_as_float(v.w);
                }
          }
        }
        if (!PL_DBG(4)) {
          // C[m][n] = (C[m][n] +) alpha * acc: column l31 of a 32-wide MFMA tile, rows (r & 3) + 8 (r >> 2) + 4 h
#pragma clang fp contract(off)
          const float alpha = dc.alpha;
          const long ldc = dc.ldc;
          const bool accum = p.accumulate != 0;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float* cj = dc.C + (long)(row0 + e_wm0 + 4 * e_h) * ldc + (c_col[j] + e_l31);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              float old[16];
              if (accum) {
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = cj[(long)(32 * i + (r & 3) + 8 * (r >> 2)) * ldc];
              }
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float v = alpha * acc[i][j][r];
                cj[(long)(32 * i + (r & 3) + 8 * (r >> 2)) * ldc] = accum ? old[r] + v : v;
              }
            }
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0);
      if (!sk_next_piece(cu, rg.lo, nkb, cp)) break;
      ckb = cp.kb_lo;
      zero_acc();
      { int r0_, c0_; decode(cp, r0_, c0_); set_b_frag(c0_ / PL_BN); }
      read_frag(stage, 0, f0);            // the next piece's first block landed before the last barrier
    }
  };
  if (wave >= 4 && !PL_DBG(32)) run(std::true_type{});
  else run(std::false_type{});
}

int plw_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

// launch == false: only answer whether this kernel WOULD take the descriptor
static int plw_try(const ctts_gemm_desc& d, hipStream_t st, bool launch) {
  static const int enabled = plw_env("CTTS_PLW", 1);
  static const int min_units = plw_env("CTTS_PLW_MIN_UNITS", 4096);
  static const int wg_units = plw_env("CTTS_PLW_WG_UNITS", 16);
  // few output tiles = every tile cut into many pieces, whose slabs the owner adds one after the other (25 pieces per tile: 198 us for the
  // [256 x 1280] gradient that the split-K kernel with its parallel reduce launch finishes in 92): such launches stay where they are
  static const int min_tiles = plw_env("CTTS_PLW_MIN_TILES", 32);
  static const int force_w = plw_env("CTTS_PLW_W", 0);
  static const int debug = plw_env("CTTS_PL_DEBUG", 0);
  if (!enabled || d.bf16_split < 1 || !d.A_planes || !d.B_planes) return 0;
  if (!d.sk_ws || d.sk_ws_bytes < (int64_t)CTTS_WS_BYTES) return 0;
  if (d.a_kc || d.b_kc || d.nb0 * d.nb1 != 1 || d.lens || d.E || d.epi_bwd || d.split_out) return 0;
  if (d.bias || d.act || d.Z || d.R || d.rowscale || d.p_drop > 0.f) return 0;          // a weight gradient has no epilogue terms
  if (d.M % PL_BM != 0 || d.N % PL_BN != 0 || d.K < 64) return 0;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(d.A_planes) || !al16(d.B_planes) || ((d.lda | d.ldb) & 31) || d.lda < d.M) return 0;
  const bool conv = d.conv_T > 0;
  if (conv) {
    if (!d.conv_on_b || d.conv_cin % PL_BN != 0 || d.N % d.conv_cin != 0 || d.ldb < d.conv_cin || d.conv_T < 32) return 0;
    if (d.conv_pad < 0 || d.conv_pad >= d.conv_T || d.N / d.conv_cin - 1 - d.conv_pad >= d.conv_T) return 0;
  } else if (d.ldb < d.N) {
    return 0;
  }
  // 32-bit buffer offsets over the plane sets (plus a K-block beyond the last row and the conv shift)
  const long a_ext = (long)(d.K + 64) * d.lda * 6, b_ext = (long)(d.K + 64 + (conv ? d.N / d.conv_cin : 0)) * d.ldb * 6;
  if (a_ext >= 0x7FFF0000L || b_ext >= 0x7FFF0000L) return 0;
  PlwArgs p;
  p.nutt = p.kbu = 0;
  p.nkb = (d.K + 31) / 32;
  if (d.row_lens) {
    if (d.row_T <= 0 || d.row_T % 32 != 0 || d.K % d.row_T != 0 || d.K / d.row_T > PL_MAX_UTT) return 0;
    if (conv && d.row_T != d.conv_T) return 0;
    p.nutt = d.K / d.row_T;
    p.kbu = d.row_T / 32;
  }
  p.tiles_m = d.M / PL_BM;
  p.tiles_n = d.N / PL_BN;
  p.accumulate = (d.split_k > 1 && !d.split_overwrite) ? 1 : 0;
  // tap folding (tile_kind in the kernel): conv view, every 32-row K-block inside one utterance, at least two taps
  static const int fold = plw_env("CTTS_PLW_FOLD", 1);
  p.fold_tt = 1; p.fold_tiles = 0;
  if (fold && conv && d.conv_T % 32 == 0 && d.K % d.conv_T == 0) {       // (whole utterances: the extra rows of a folded image are tested against [0, T) only)
    const int ntap = d.N / d.conv_cin;
    const int tt = ntap >= 8 ? 8 : (ntap >= 4 ? 4 : (ntap >= 2 ? 2 : 1));
    if (tt > 1) { p.fold_tt = tt; p.fold_tiles = d.conv_cin / (PL_BN / tt); }
  }
  p.debug = debug;
  p.ws = reinterpret_cast<unsigned*>(d.sk_ws);
  const long tiles = (long)p.tiles_m * p.tiles_n;
  const long units = tiles * p.nkb;
  const bool forced = d.bf16_split == 2 || d.bf16_split == 4;        // no size thresholds (parity tests of small launches)
  if (!forced && (units < min_units || tiles < min_tiles)) return 0;
  long W = force_w > 0 ? force_w : 32;
  const long Wu = units / (8L * wg_units);
  if (W > Wu) W = Wu;
  if (W < 1) W = 1;
  const int grid = (int)W * 8;
  if (grid > PL_MAX_WG || (long)grid * PL_SLAB > PL_SLAB_FLOATS_MAX) return 0;
  if (!launch) return 1;
  if (d.bf16_split >= 3) {
    if (conv) hipLaunchKernelGGL((gemm_plw_kernel<true, 1>), dim3(grid), dim3(512), 0, st, d, p);
    else hipLaunchKernelGGL((gemm_plw_kernel<false, 1>), dim3(grid), dim3(512), 0, st, d, p);
  } else {
    if (conv) hipLaunchKernelGGL((gemm_plw_kernel<true, 6>), dim3(grid), dim3(512), 0, st, d, p);
    else hipLaunchKernelGGL((gemm_plw_kernel<false, 6>), dim3(grid), dim3(512), 0, st, d, p);
  }
  CTTS_CHECK_LAUNCH("ctts_gemm(planes, weight gradient)");
  return 1;
}

int ctts_gemm_plw_try(const ctts_gemm_desc& d, hipStream_t st) { return plw_try(d, st, true); }
int ctts_gemm_plw_takes(const ctts_gemm_desc& d) { return plw_try(d, nullptr, false) > 0 ? 1 : 0; }
