// fp32 GEMM on the BF16 matrix pipe from PRE-SPLIT operands: persistent stream-K kernel whose main loop is LDS-DMA + ds_read + MFMA only.
//
// Round 4's gemm_x6_kernel (gemm.hip) splits every fp32 operand into its three bf16 pieces while it stages a tile into LDS: every
// workgroup that stages an element splits it again (an activation tile 8 times - once per n-tile -, a weight tile once per m-tile), the
// split's VALU work sits between the matrix pipe's K-blocks of an in-order wave, and the staging registers exist only for the split.
// Here the split has LEFT the GEMM: the operands arrive as three bf16 planes per matrix (ctts_split_planes, or a producer's epilogue),
//   * HBM layout of a plane set (ctts_split_planes): [rows][K / 32][3 pieces][32] bf16 - the hi | mid | lo pieces of one 32-deep K-block
//     of one row are 192 CONTIGUOUS bytes, consecutive K-blocks follow each other: the three 64-byte plane rows a K-block needs share
//     1.5 cache lines (two fetches) instead of three half-used lines ([3][rows][K] planes, the first version: the vector L1 moves whole
//     128-byte lines at 64 B/clk, so half-used lines halved the rate at which a tile arrives - the DMA alone took 1.4 us per K-block
//     next to 1.65 us of MFMAs, and with one block of prefetch the two did not overlap: 363 us for the dense FFN conv, 2.46 us per block);
//   * a K-block (32 deep) of a tile is 64 bytes per row and plane; `buffer_load_dwordx4 ... lds` moves 16 rows x 64 bytes per instruction
//     straight into LDS (the image of gemm_x6_kernel: 64-byte plane rows whose 16-byte chunks are XOR-swizzled by (row >> 2) & 3 - applied
//     to the SOURCE address, the DMA writes lane-linear - so that the ds_read_b128 fragment reads are conflict free);
//   * workgroup = 8 waves (2 x 4), tile 128 x 256 (a wave owns 64 x 64 = 2 x 2 MFMA tiles, 48 v_mfma_f32_32x32x16_bf16 per K-block: the six
//     cross terms hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi, smallest first, fp32 accumulate - the arithmetic of gemm_x6_kernel);
//     one workgroup per CU, two waves per SIMD; 2 LDS stages of 72 KB;
//   * ONE barrier per K-block, placed in the MIDDLE of the block's MFMAs: the fragments of the second half (k-step 1) are read before the
//     first half's MFMAs are issued; at the barrier every wave has finished reading this stage (so the DMA of block i+2 may overwrite it)
//     and block i+1 has landed (so its first-half fragments are read while the second half's MFMAs run).  Neither the DMA latency nor
//     the LDS read latency is exposed - only the barrier skew;
//   * persistent grid with the even (tile, K-block) partition and the fixed-order slab hand-off of gemm_sk.hip (sk_plan.h): deterministic;
//   * conv view on A (implicit im2col, K walked channel-block-major so that consecutive K-blocks re-read the same lines shifted by a row);
//   * ragged (b, t) rows: the schedule of the ACTIVE 128-row tiles is built by every workgroup itself from row_lens (prefix sums in LDS;
//     row_T % 128 == 0), wholly padded tiles are zero-filled, and - the zero rule has 64-row granularity in the other kernels - the waves
//     that own a wholly padded upper half of an active tile write zeros instead of their epilogue.
// Eligibility: pl_try below.  Everything else stays on gemm.hip / gemm_sk.hip / gemm_ws.hip.
#include "gemm_pl_common.h"
#include "planes_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct PlArgs {
  int tiles_m, tiles_n;      // static tile grid (128 x 256 tiles); tiles_m counts ALL m-tiles (the active count comes from row_lens)
  int nkb;                   // K-blocks per tile
  int gw;                    // n-tiles per schedule group
  int whole_tiles;           // 1: never split a tile
  int ntap;                  // conv view: taps (K / cin); K is walked (channel block, tap)
  int nutt, tpu;             // ragged rows: utterances and 128-row tiles per utterance (nutt = 0: dense)
  int debug;                 // CTTS_PL_DEBUG (tools): 1 = no DMA after the prologue, 4 = no epilogue, 8 = no MFMA, 32 = no rotated order in the upper wave group, 16 = record shader cycles / wall ticks of workgroup 8 in the workspace header
  unsigned* ws;
};

// The epilogues this kernel carries (pl_epilogue_ok is the host-side twin: descriptors with any other combination are not taken).  Only
// lean variants: the generic epilogue and the full dispatch of gemm_epilogue_auto (14 variants) cost this kernel 12 - 27 spilled VGPRs
// with reloads INSIDE the K loop.
//   forward:  bias, then act 0 / 1 / 2 / 4 (+ pre-activation store), dropout with act 2 / 4;   bias + dropout + residual + rowscale ("bdrs")
//   backward: the producer-epilogue backward of GELU / swish + dropout ("a2zdB", "a4zdB")
__host__ __device__ __forceinline__ int pl_epilogue_kind(const ctts_gemm_desc& d) {
  const bool drop = d.p_drop > 0.f, res = d.R != nullptr, rs = d.rowscale != nullptr;
  if (d.E || !gemm_fits32(d.C, d.M, d.ldc, d.N) || !gemm_fits32(d.Z, d.M, d.ldz, d.N) || !gemm_fits32(d.R, d.M, d.ldr, d.N)) return -1;
  if (d.epi_bwd) {
    if (d.Z && !rs && drop && d.act == 2) return 8;
    if (d.Z && !rs && drop && d.act == 4) return 9;
    return -1;
  }
  if (res || rs) return (d.act == 0 && !d.Z && res && rs && drop) ? 7 : -1;
  if (d.act == 0) return (!d.Z && !drop) ? 0 : -1;
  if (d.act == 1) return drop ? -1 : 1;
  if (d.act == 2) return drop ? 3 : 2;
  if (d.act == 4) return drop ? 5 : 4;
  return -1;
}
__device__ __forceinline__ void pl_epilogue(const ctts_gemm_desc& d, const floatx16 (&acc)[2][2], int row0, int col0, int wm0, int wn0,
                                            int l31, int h, int Mv, int Nv) {
#define PL_LEAN(ACT, DROP, BWD, AUX, RS) gemm_epilogue_lean<2, 2, ACT, DROP, BWD, AUX, RS>(d, acc, d.C, 0, row0, col0, wm0, wn0, l31, h, Mv, Nv)
  switch (pl_epilogue_kind(d)) {
    case 0: PL_LEAN(0, false, false, false, false); break;
    case 1: PL_LEAN(1, false, false, false, false); break;
    case 2: PL_LEAN(2, false, false, false, false); break;
    case 3: PL_LEAN(2, true, false, false, false); break;
    case 4: PL_LEAN(4, false, false, false, false); break;
    case 5: PL_LEAN(4, true, false, false, false); break;
    case 7: PL_LEAN(0, true, false, true, true); break;
    case 8: PL_LEAN(2, true, true, true, false); break;
    case 9: PL_LEAN(4, true, true, true, false); break;
    default: break;
  }
#undef PL_LEAN
}

// 12 x (two MFMAs, one LDS read) in program order for the scheduling region that ends here (masks: 0x008 MFMA, 0x100 DS read)
#define PL_SGB3() __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)
#define PL_INTERLEAVE_12() PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3(); PL_SGB3()

#define PL_SGB11() __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)
#define PL_INTERLEAVE_4x1() PL_SGB11(); PL_SGB11(); PL_SGB11(); PL_SGB11()

#ifndef PL_SLAB_BATCH
#define PL_SLAB_BATCH 8
#endif

struct PlFrag { pl_u32x4 a[2][3], b[2][3]; };      // one 16-deep k-step: [MFMA row / column tile][plane]

// TERMS = 6: fp32 products from the six cross terms of the three-way split (bf16_split 1 / 2).  TERMS = 1: the "amp" arithmetic
// (bf16_split 3 / 4; reference train.py:59,104 `amp.autocast`): operands ROUNDED to bf16 - only the hi pieces are moved and multiplied -
// fp32 accumulate: one MFMA term, a third of the operand traffic.  Never the default, reported separately.
template <bool CONV, int TERMS>
__global__ __launch_bounds__(512, 2) void gemm_pl_kernel(const ctts_gemm_desc d, const PlArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PL_STAGE];
  __shared__ int s_pref[PL_MAX_UTT + 1];          // active 128-row tiles of the utterances before b
  __shared__ int s_lenh[PL_MAX_UTT];              // row_lens[b] + row_halo

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 64;
  const int nutt = p.nutt, tpu = p.tpu;
  constexpr int NQ = TERMS == 1 ? 1 : 3;          // pieces moved and read
  const unsigned long long dbg_c0 = PL_DBG(16) ? clock64() : 0ull, dbg_w0 = PL_DBG(16) ? wall_clock64() : 0ull;

  // ---- schedule of the active m-tiles (ragged rows)
  int n_mt = p.tiles_m;
  if (nutt > 0) {
    // lengths to LDS first (ONE global load per thread), then the prefix sums from LDS: summing straight from row_lens made thread t wait
    // for t dependent global loads - ~18 us at the head of every ragged launch (16 utterances), measured on the encoder-sized shapes
    for (int t = tid; t < nutt; t += 512) s_lenh[t] = d.row_lens[t] + d.row_halo;
    __syncthreads();
    for (int t = tid; t <= nutt; t += 512) {
      int s = 0;
      for (int b = 0; b < t; ++b) {
        const int L = s_lenh[b];
        s += L <= 0 ? 0 : min(tpu, (L + PL_BM - 1) / PL_BM);
      }
      s_pref[t] = s;
    }
    __syncthreads();
    n_mt = s_pref[nutt];
    // wholly padded tiles are defined as zero: stores only, spread over the grid
    const int n_zero = (nutt * tpu - n_mt) * p.tiles_n;
    const bool v4 = ((d.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.C) & 15) == 0) &&
                    (!d.Z || d.epi_bwd || (((d.ldz & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.Z) & 15) == 0)));
    for (int zt = blockIdx.x; zt < n_zero; zt += gridDim.x) {
      const int mi = zt / p.tiles_n, nt = zt - mi * p.tiles_n;
      int lo = 0, hi = nutt;                      // inactive prefix ip(b) = b * tpu - s_pref[b]: ip(lo) <= mi < ip(hi)
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (mid * tpu - s_pref[mid] <= mi) lo = mid; else hi = mid; }
      const int j = (s_pref[lo + 1] - s_pref[lo]) + (mi - (lo * tpu - s_pref[lo]));
      const int row0 = lo * d.row_T + j * PL_BM, col0 = nt * PL_BN;
      const int ncols = min(PL_BN, d.N - col0);
      if (v4 && (ncols & 3) == 0) {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = tid; e < PL_BM * (PL_BN / 4); e += 512) {
          const int r = e / (PL_BN / 4), c = (e - r * (PL_BN / 4)) * 4;
          if (c < ncols) {
            *reinterpret_cast<float4*>(d.C + (long)(row0 + r) * d.ldc + col0 + c) = z4;
            if (d.Z && !d.epi_bwd) *reinterpret_cast<float4*>(d.Z + (long)(row0 + r) * d.ldz + col0 + c) = z4;
          }
        }
      } else {
        for (int e = tid; e < PL_BM * PL_BN; e += 512) {
          const int r = e / PL_BN, c = e - r * PL_BN;
          if (c < ncols) {
            d.C[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
            if (d.Z && !d.epi_bwd) d.Z[(long)(row0 + r) * d.ldz + col0 + c] = 0.f;
          }
        }
      }
    }
  }

  const int nkb = p.nkb;
  SkGeom g{n_mt * p.tiles_n, nkb, (int)(gridDim.x >> 3), p.whole_tiles};
  if (g.n_tiles <= 0 || nkb <= 0) return;
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  const SkRange rg = sk_range(g, xcd, wj);
  if (rg.hi <= rg.lo) return;

  const int cin = d.conv_cin > 0 ? d.conv_cin : 32;
  const int T = d.conv_T > 0 ? d.conv_T : 1;
  // plane set: row stride 3 * ld bf16 = 6 * ld bytes; K-block kb of a row at + kb * 192 bytes, piece q at + q * 64 bytes
  const pl_i32x4 ra_src = pl_make_rsrc(d.A_planes - (CONV ? (long)d.conv_pad * d.lda * 3 : 0));
  const pl_i32x4 rb_src = pl_make_rsrc(d.B_planes);
  const unsigned lda2 = (unsigned)(d.lda * 6), ldb2 = (unsigned)(d.ldb * 6);
  const unsigned smem_addr = (unsigned)reinterpret_cast<uintptr_t>(smem);
  unsigned* flags = p.ws;
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(p.ws) + CTTS_WS_SLABS);
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, 0x7FFFFFFE, 0x00020000);

  // ---- (schedule slot, n-tile) -> rows / columns; pad_hi: the upper 64 rows of the tile lie wholly in one utterance's padding
  auto decode = [&](const SkPiece& pc, int& row0, int& col0, bool& pad_hi) {
    int mslot, nt;
    sk_tile_decode(rg.T0 + pc.t, n_mt, p.gw, mslot, nt);
    pad_hi = false;
    if (nutt > 0) {
      int lo = 0, hi = nutt;                      // s_pref[lo] <= mslot < s_pref[hi]
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pref[mid] <= mslot) lo = mid; else hi = mid; }
      const int j = mslot - s_pref[lo];
      row0 = lo * d.row_T + j * PL_BM;
      pad_hi = j * PL_BM + 64 >= s_lenh[lo];
    } else {
      row0 = mslot * PL_BM;
    }
    row0 = __builtin_amdgcn_readfirstlane(row0);
    col0 = __builtin_amdgcn_readfirstlane(nt * PL_BN);
  };

  // ---- loader: wave w moves rows 16 w .. 16 w + 15 of the three A planes and rows 32 w .. 32 w + 31 of the three B planes of a K-block
  //      (9 DMA instructions); lane L -> row L >> 2 of its 16-row group, physical chunk L & 3 = logical chunk (L & 3) ^ ((L >> 4) & 3)
  const int lrow = lane >> 2, lchunk = (lane & 3) ^ ((lane >> 4) & 3);
  unsigned voffA = PL_OOB, voffB[2] = {PL_OOB, PL_OOB};
  int trowA = 0;
  int lu = rg.hi;
  SkPiece lp;
  bool have_l = sk_next_piece(lu, rg.lo, nkb, lp);
  int lkb = lp.kb_lo, ltap = 0, lcb = 0;
  auto loader_set_piece = [&]() {
    int row0, col0; bool ph;
    decode(lp, row0, col0, ph);
    const int rA = row0 + 16 * wave + lrow;
    voffA = rA < d.M ? (unsigned)rA * lda2 + (unsigned)(lchunk * 16) : PL_OOB;
    if (CONV) trowA = rA % T;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = col0 + 32 * wave + 16 * j + lrow;
      voffB[j] = n < d.N ? (unsigned)n * ldb2 + (unsigned)(lchunk * 16) : PL_OOB;
    }
    lkb = lp.kb_lo;
    if (CONV) { lcb = lkb / p.ntap; ltap = lkb - lcb * p.ntap; }
  };
  auto loader_issue = [&](int stage) {
    unsigned soffA, soffB, vA = voffA;
    if (CONV) {
      soffA = (unsigned)ltap * lda2 + (unsigned)(lcb * 192);
      soffB = (unsigned)(ltap * (cin >> 5) + lcb) * 192u;
      vA = ((unsigned)(trowA + ltap - d.conv_pad) < (unsigned)T) ? vA : PL_OOB;
    } else {
      soffA = soffB = (unsigned)lkb * 192u;
    }
    const unsigned sA = smem_addr + (unsigned)(stage * PL_STAGE + wave * 1024);
    const unsigned sB = smem_addr + (unsigned)(stage * PL_STAGE + 3 * PL_A_PLANE + wave * 2048);
    // the three pieces of a row group back to back: they share cache lines
#pragma unroll
    for (int q = 0; q < NQ; ++q) pl_dma16(ra_src, sA + q * PL_A_PLANE, vA, soffA + q * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < NQ; ++q) pl_dma16(rb_src, sB + q * PL_B_PLANE + j * 1024, voffB[j], soffB + q * 64);
  };
  auto loader_advance = [&]() {
    ++lkb;
    if (lkb == lp.kb_hi) {
      have_l = sk_next_piece(lu, rg.lo, nkb, lp);
      if (have_l) loader_set_piece();
    } else if (CONV) {
      ++ltap;
      if (ltap == p.ntap) { ltap = 0; ++lcb; }
    }
  };

  // ---- fragments: row l31 of a 32-row MFMA tile, the 8 consecutive k from h * 8 of the 16-deep k-step ks = logical chunk 2 ks + h
  const int fsw = (l31 >> 2) & 3;
  const int fcb[2] = {((0 + h) ^ fsw) * 16, ((2 + h) ^ fsw) * 16};
  const unsigned char* fr_a = smem + (wm0 + l31) * PL_ROW;
  const unsigned char* fr_b = smem + 3 * PL_A_PLANE + (wn0 + l31) * PL_ROW;
  auto read_frag = [&](int stage, int ks, PlFrag& f) {
    const unsigned char* pa = fr_a + stage * PL_STAGE + fcb[ks];
    const unsigned char* pb = fr_b + stage * PL_STAGE + fcb[ks];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < NQ; ++q) f.a[i][q] = *reinterpret_cast<const pl_u32x4*>(pa + q * PL_A_PLANE + i * 32 * PL_ROW);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < NQ; ++q) f.b[j][q] = *reinterpret_cast<const pl_u32x4*>(pb + q * PL_B_PLANE + j * 32 * PL_ROW);
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
  // terms [t0, t1) of the six-term product of one k-step; term-major: consecutive MFMAs hit different accumulators; smallest terms first
  auto mma_terms = [&](const PlFrag& f, int t0, int t1) {
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

  // ---- prologue: blocks 0 and 1 in flight, first-half fragments of block 0 in registers
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
  int remaining = rg.hi - rg.lo;              // K-blocks this workgroup still has to compute (the current one included)
  zero_acc();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // The K loop exists twice - once per wave group, chosen ONCE by a wave-uniform branch: the same work in two instruction orders (see the
  // comment in the loop).  Both copies execute the same sequence of barriers.
  auto run = [&](auto skew_c) {
  constexpr bool SKEW = decltype(skew_c)::value;
  PlFrag f0, f1;
  read_frag(0, 0, f0);
  int stage = 0;
  // ragged rows: when the UPPER 64 rows of the current tile lie wholly in padding, the upper wave group's products are discarded (its
  // waves write zeros instead of an epilogue) - it then issues neither fragment reads nor MFMAs for the piece, only its share of the DMA
  // and the barriers: ~4 % of the matrix work of a ragged launch, and on a power-limited pipe idle MFMAs are clock for the others
  bool idle = false;
  if constexpr (SKEW) { int r0_, c0_; decode(cp, r0_, c0_, idle); }
  __builtin_amdgcn_s_waitcnt(0);
  while (true) {
    // The two waves of a SIMD (w and w + 4) meet the same barrier, so their non-MFMA sections (fragment reads, DMA issue, cursor
    // arithmetic: ~450 issue cycles per block and wave) would coincide and the matrix pipe would idle through them.  The upper wave
    // group therefore runs the same work in a ROTATED order: its reads / DMA issue sit 8 MFMAs (one wave's 256 pipe cycles) later,
    // under the lower group's MFMAs and vice versa.  (p.debug & 32 switches the rotation off: A/B timing.)
    const bool do_mma = PL_DBG(8) == 0 && !(SKEW && idle);
    // first half: the fragments of k-step 0 are in registers (read during the previous block); k-step 1 is read under its MFMAs - ONE
    // ds_read_b128 behind every second MFMA (sched_group_barrier).  hipcc otherwise emits the 12 reads as a burst during which this wave
    // issues no MFMA, and the pipe then depends on the SIMD's other wave being in an MFMA phase right then (gemm_plw.hip measured the
    // same loop without DMA: 314 us with bursts, 253 us without any reads).  An idle upper wave (SKEW, half-padded tile) skips both.
    if (do_mma) {
      read_frag(stage, 1, f1);
      mma_terms(f0, 0, 6);
      if constexpr (TERMS == 1) { PL_INTERLEAVE_4x1(); } else { PL_INTERLEAVE_12(); }
    }
    __builtin_amdgcn_sched_barrier(0);
    // every wave is done reading this stage, and (mine of) block i + 1 has landed: after the barrier the whole block has
    // lgkmcnt(0) as a wait hipcc can SEE (0xC07F = vmcnt 63, expcnt 7, lgkmcnt 0): inside the asm it left the compiler's scoreboard with the
    // fragment reads of k-step 1 still "pending", and - the counter being in order - every second-half MFMA on them then waited for the
    // NEWER reads of the next block's fragments as well (s_waitcnt lgkmcnt(5 .. 0) in front of the first six MFMAs after the barrier)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // second half: the DMA of block i + 2 into this stage (lower wave group: in front of its MFMAs, upper group: behind them - the two
    // waves of a SIMD do not issue their ~100 loader instructions at the same time), the MFMAs of k-step 1 with the first-half fragment
    // reads of block i + 1 (it has landed) between them.  The reads are unconditional: when this block ends the piece the values are dead
    // (f0 is read again behind the epilogue) and reads and MFMAs stay in one scheduling region.
    if constexpr (!SKEW) {
      if (have_l && !PL_DBG(1)) loader_issue(stage);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (do_mma) {
      read_frag(stage ^ 1, 0, f0);
      mma_terms(f1, 0, 6);
      if constexpr (TERMS == 1) { PL_INTERLEAVE_4x1(); } else { PL_INTERLEAVE_12(); }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SKEW) {
      if (have_l && !PL_DBG(1)) loader_issue(stage);
    }
    if (have_l) loader_advance();
    --remaining;
    ++ckb;
    stage ^= 1;
    if (ckb < cp.kb_hi) continue;

    // ---------------- the piece is complete
    // Everything below runs once per piece and must not cost the K loop registers.  (1) Its lane / wave constants are laundered through an
    // empty asm so that loop-invariant code motion cannot hoist the epilogues' lane offsets in front of the K loop.  (2) The descriptor
    // fields only the epilogue needs (C, Z, bias, strides, dropout, ...) are read HERE from the kernel-argument segment (`d` is the first
    // kernel argument: offset 0) through a laundered pointer: s_load at the point of use instead of ~60 SGPRs that stay live across the
    // K loop - hipcc preloads a by-value struct argument and then spills it to VGPR lanes (560 SGPR spills, 27 VGPR spills and scratch
    // reloads in front of every DMA issue in the first build of this kernel).
    int e_l31 = l31, e_h = h, e_wm0 = wm0, e_wn0 = wn0;
    unsigned long long kargs = (unsigned long long)(const void*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+v"(e_l31), "+v"(e_h), "+s"(e_wm0), "+s"(e_wn0), "+s"(kargs));
    const ctts_gemm_desc& dc = *(const ctts_gemm_desc*)(const __attribute__((address_space(4))) ctts_gemm_desc*)kargs;
    const int Mv = dc.M, Nv = dc.N;
    int row0, col0; bool pad_hi;
    decode(cp, row0, col0, pad_hi);
    if (cp.kb_hi < nkb) {
      // contribution: slab (write-through stores) + flag
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
        // owner of a cut tile: add the slabs of the workgroups below, nearest first, until the tile's unit 0 is covered
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
          // The loads of a slab in flight in batches of PL_SLAB_BATCH quads, THEN their sums (in the fixed order): written as "load, add" hipcc
          // reused one register quad and waited for every load - 16 dependent L2 round trips, ~16 us per slab, on the critical path of every
          // cut tile.  (All 16 at once cost this kernel 59 - 72 spilled VGPRs with reloads inside the K loop; gemm_plw.hip affords it.)
#pragma unroll
          for (int g = 0; g < 16 / PL_SLAB_BATCH; ++g) {
            pl_u32x4 sv[PL_SLAB_BATCH];
#pragma unroll
            for (int e = 0; e < PL_SLAB_BATCH; ++e) sv[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, base + (unsigned)((g * PL_SLAB_BATCH + e) * 1024), 0, 0);
#pragma unroll
            for (int e = 0; e < PL_SLAB_BATCH; ++e) {
              const int n = g * PL_SLAB_BATCH + e, i = n >> 3, j = (n >> 2) & 1, q = n & 3;
              acc[i][j][4 * q + 0] += __uint_as_float(sv[e].x); acc[i][j][4 * q + 1] += __uint_as_float(sv[e].y);
              acc[i][j][4 * q + 2] += __uint_as_float(sv[e].z); acc[i][j][4 * q + 3] += __uint_as_float(sv[e].w);
            }
          }
        }
      }
      if (pad_hi && e_wm0 == 64) {
        for (int e = lane; e < 64 * 64; e += 64) {
          const int r = e >> 6, c = e & 63;
          const int n = col0 + e_wn0 + c;
          if (n < Nv) {
            dc.C[(long)(row0 + 64 + r) * dc.ldc + n] = 0.f;
            if (dc.Z && !dc.epi_bwd) dc.Z[(long)(row0 + 64 + r) * dc.ldz + n] = 0.f;
          }
        }
      } else if (!PL_DBG(4)) {
        pl_epilogue(dc, acc, row0, col0, e_wm0, e_wn0, e_l31, e_h, Mv, Nv);
      }
    }
    if (PL_DBG(16) && blockIdx.x == 8 && tid == 0) {         // tools: shader cycles and 100 MHz wall ticks of one workgroup's life so far
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws + PL_MAX_WG + 2);
      o[0] = clock64() - dbg_c0;
      o[1] = wall_clock64() - dbg_w0;
    }
    // a wait hipcc can see (gemm_sk.hip): its scoreboard is empty when control returns to the K loop
    __builtin_amdgcn_s_waitcnt(0);
    if (!sk_next_piece(cu, rg.lo, nkb, cp)) break;
    ckb = cp.kb_lo;
    zero_acc();
    if constexpr (SKEW) { int r0_, c0_; decode(cp, r0_, c0_, idle); }
    read_frag(stage, 0, f0);            // the next piece's first block landed before the last barrier
  }
  };
  if (wave >= 4 && !PL_DBG(32)) run(std::true_type{});
  else run(std::false_type{});
}

int pl_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// ---------------------------------------------------------------- exact three-way bf16 split of fp32 matrices (many per launch)
constexpr int SPL_BATCH = 32;          // all weight sets of an fs2 step (26 matrices) in ONE launch
struct SplitBatch {
  const float* src[SPL_BATCH];
  uint16_t* dst[SPL_BATCH];
  long rows[SPL_BATCH], cols8[SPL_BATCH], ld[SPL_BATCH];
  int first_block[SPL_BATCH + 1];
  int ntasks;
};
constexpr int SPL_PER_BLOCK = 256 * 4;           // 8-element groups per workgroup

// dst [rows][ld / 32][3][32] bf16: thread = 8 consecutive k of one row -> three 16-byte stores 64 bytes apart inside the K-block's 192 bytes
__global__ __launch_bounds__(256) void split_planes_kernel(const SplitBatch b) {
  int t = 0;
  while (t + 1 < b.ntasks && (int)blockIdx.x >= b.first_block[t + 1]) ++t;
  const float* __restrict__ src = b.src[t];
  uint16_t* __restrict__ dst = b.dst[t];
  const long c8 = b.cols8[t], ld = b.ld[t], n8 = b.rows[t] * c8;
  const long g0 = (long)((int)blockIdx.x - b.first_block[t]) * SPL_PER_BLOCK;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long gidx = g0 + u * 256 + threadIdx.x;
    if (gidx >= n8) continue;
    const long r = gidx / c8, c = (gidx - r * c8) * 8;
    const float4 x0 = *reinterpret_cast<const float4*>(src + r * ld + c), x1 = *reinterpret_cast<const float4*>(src + r * ld + c + 4);
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    spl_store8(dst, r, ld, c, xs);
  }
}

}  // namespace

extern "C" int ctts_split_planes(const ctts_split_task* tasks, int ntasks, void* stream) {
  CTTS_REQUIRE(ntasks >= 0 && (tasks || ntasks == 0), "ctts_split_planes: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < ntasks; t0 += SPL_BATCH) {
    SplitBatch b;
    b.ntasks = 0;
    long blocks = 0;
    for (int t = t0; t < ntasks && t < t0 + SPL_BATCH; ++t) {
      const ctts_split_task& q = tasks[t];
      CTTS_REQUIRE(q.src && q.dst && q.rows >= 0 && q.cols >= 0 && q.cols % 32 == 0 && q.ld >= q.cols && q.ld % 32 == 0 &&
                       (reinterpret_cast<uintptr_t>(q.src) & 15) == 0 && (reinterpret_cast<uintptr_t>(q.dst) & 15) == 0,
                   "ctts_split_planes: task %d needs 16-byte aligned pointers, cols %% 32 == 0 and ld %% 32 == 0", t);
      if (q.rows == 0 || q.cols == 0) continue;
      const int i = b.ntasks++;
      b.src[i] = q.src; b.dst[i] = q.dst; b.rows[i] = q.rows; b.cols8[i] = q.cols / 8; b.ld[i] = q.ld;
      b.first_block[i] = (int)blocks;
      blocks += (q.rows * (q.cols / 8) + SPL_PER_BLOCK - 1) / SPL_PER_BLOCK;
      CTTS_REQUIRE(blocks < (1L << 30), "ctts_split_planes: too many elements in one call");
    }
    if (b.ntasks == 0) continue;
    b.first_block[b.ntasks] = (int)blocks;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
    CTTS_CHECK_LAUNCH("ctts_split_planes");
  }
  return 0;
}

// launch == false: only answer whether the plane kernel WOULD take this descriptor (ctts_gemm_takes_planes)
static int pl_try(const ctts_gemm_desc& d, hipStream_t st, bool launch) {
  static const int enabled = pl_env("CTTS_PL", 1);
  static const int min_units = pl_env("CTTS_PL_MIN_UNITS", 4096);     // (tile, K-block) units; below this the launch is latency bound either way
  static const int split_from = pl_env("CTTS_PL_SPLIT_NKB", 24);
  // pieces a tile may be cut into when the tile count alone cannot fill the chip (2,048-row launches: 64 tiles).  2 until the slab
  // hand-off stopped costing ~16 us per slab (see the owner's loop); with ~4 us per slab: encoder FFN conv forward 93 -> 64 us at 4,
  // its data gradient (16 tiles x 288 K-blocks) 156 -> 92 us at 8
  static const int max_split = pl_env("CTTS_PL_MAX_SPLIT", 8);
  static const int wg_units = pl_env("CTTS_PL_WG_UNITS", 16);
  static const int force_w = pl_env("CTTS_PL_W", 0);
  static const int debug = pl_env("CTTS_PL_DEBUG", 0);
  if (!enabled || d.bf16_split < 1 || !d.A_planes || !d.B_planes) return 0;
  if (!d.sk_ws || d.sk_ws_bytes < (int64_t)CTTS_WS_BYTES) return 0;
  if (!d.a_kc || !d.b_kc || d.nb0 * d.nb1 != 1 || d.lens || d.E) return 0;
  if (d.split_k > 1 && !d.split_overwrite) return 0;                 // "C += alpha A B" is not built here
  if (d.K % 32 != 0 || d.K < 64 || d.N % 128 != 0 || d.N < 256 || d.M < 128) return 0;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(d.A_planes) || !al16(d.B_planes) || ((d.lda | d.ldb) & 31)) return 0;
  const bool conv = d.conv_T > 0;
  if (conv && (d.conv_on_b || d.conv_cin % 32 != 0 || d.K % d.conv_cin != 0 || d.conv_T < 16)) return 0;
  // 32-bit buffer offsets over the three planes (plus a tile of rows beyond M and the conv shift)
  const long a_ext = (long)(d.M + 256) * d.lda * 3 + 3L * d.K;
  const long b_ext = (long)(d.N + 256) * d.ldb * 3 + 3L * d.K;
  if (a_ext * 2 >= 0x7FFF0000L || b_ext * 2 >= 0x7FFF0000L) return 0;
  if (pl_epilogue_kind(d) < 0) return 0;          // only the lean epilogues the kernel carries
  PlArgs p;
  p.nutt = p.tpu = 0;
  p.tiles_m = (d.M + PL_BM - 1) / PL_BM;
  long act_tiles_m = p.tiles_m;
  if (d.row_lens) {
    if (d.row_T <= 0 || d.row_T % PL_BM != 0 || d.M % d.row_T != 0 || d.M / d.row_T > PL_MAX_UTT) return 0;
    p.nutt = d.M / d.row_T;
    p.tpu = d.row_T / PL_BM;
  }
  p.tiles_n = (d.N + PL_BN - 1) / PL_BN;
  p.nkb = d.K / 32;
  p.ntap = conv ? d.K / d.conv_cin : 1;
  p.whole_tiles = p.nkb < split_from ? 1 : 0;
  p.gw = (p.tiles_n % 4 == 0) ? p.tiles_n / 4 : p.tiles_n;
  p.debug = debug;
  p.ws = reinterpret_cast<unsigned*>(d.sk_ws);
  // grid from the STATIC tile count (the active count lives on the device): with ragged rows ~3/4 of the m-tiles are active - a grid that
  // is a little too large only makes the pieces shorter
  const long tiles = act_tiles_m * p.tiles_n;
  const long units = tiles * p.nkb;
  const bool forced = d.bf16_split == 2 || d.bf16_split == 4;        // no size thresholds (parity tests of small launches)
  if (!forced && units < min_units) return 0;
  const int cuts = p.whole_tiles ? 1 : ((p.nkb >= 256 && max_split < 4) ? 4 : max_split);
  long W = force_w > 0 ? force_w : 32;
  const long Wu = units / (8L * wg_units);
  if (W > Wu) W = Wu;
  // ragged rows: expect >= 3/4 of the tiles to be active when an utterance spans several tiles (decoder: T = 1024), all of them when it
  // spans one (encoder: T = 128 - a tile is inactive only for an empty utterance)
  const long Wt = (d.row_lens && p.tpu >= 4) ? (tiles * cuts * 3 / 4) / 8 : (tiles * cuts) / 8;
  if (W > Wt) W = Wt;
  if (W < 1) {
    if (!forced) return 0;
    W = 1;
  }
  const int grid = (int)W * 8;
  if (grid > PL_MAX_WG || (long)grid * PL_SLAB > PL_SLAB_FLOATS_MAX) return 0;
  if (!launch) return 1;
  if (d.bf16_split >= 3) {
    if (conv) hipLaunchKernelGGL((gemm_pl_kernel<true, 1>), dim3(grid), dim3(512), 0, st, d, p);
    else hipLaunchKernelGGL((gemm_pl_kernel<false, 1>), dim3(grid), dim3(512), 0, st, d, p);
  } else {
    if (conv) hipLaunchKernelGGL((gemm_pl_kernel<true, 6>), dim3(grid), dim3(512), 0, st, d, p);
    else hipLaunchKernelGGL((gemm_pl_kernel<false, 6>), dim3(grid), dim3(512), 0, st, d, p);
  }
  CTTS_CHECK_LAUNCH("ctts_gemm(planes)");
  return 1;
}

int ctts_gemm_pl_try(const ctts_gemm_desc& d, hipStream_t st) { return pl_try(d, st, true); }

extern "C" int ctts_gemm_takes_planes(const ctts_gemm_desc* d) {
  if (!d) return 0;
  ctts_gemm_desc c = *d;
  if (c.nb0 < 1) c.nb0 = 1;
  if (c.nb1 < 1) c.nb1 = 1;
  return (pl_try(c, nullptr, false) > 0 || ctts_gemm_plw_takes(c)) ? 1 : 0;
}
