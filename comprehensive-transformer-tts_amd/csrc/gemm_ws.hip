// Weight-stationary fp32-MFMA GEMM for SHORT reductions (K = 256) on gfx950:  C[M,N] = epi(alpha * A[M,K] B + bias), A K-contiguous.
//
// The linears of the conformer block (conformer.py: FF 256 -> 1024, q | k | v 256 -> 768, pos / out projections and pointwise
// convolutions 256 -> 256 / 512) and the attention projections of transformer_fs2 (transformer_fs2.py:385-394) have K = 256: with the
// K-loop kernels of gemm.hip a 64x64 tile is 8 K-blocks between a prologue (first operand round trip) and an epilogue, and they run at
// 40 - 100 TFLOP/s.  Here the whole K extent of the WEIGHT lives in registers:
//   * a workgroup of 4 waves owns 128 output columns; wave w keeps the B fragments of ITS 32 columns for all of K in 16 * K/32 = 128
//     VGPRs, loaded once per launch straight from global memory in MFMA fragment order;
//   * the workgroup is persistent over 64-row tiles of A.  A tile arrives as two K-halves (64 rows x 128 floats = 32 KB each) in two
//     LDS stages by DMA (buffer_load ... lds, issued from inline asm as in gemm_sk.hip): while the waves run the 128 MFMAs of one half,
//     the other half (of this tile or of the next one) is in flight - one barrier per 128 MFMAs of every wave;
//   * every wave alternates between two accumulators (rows 0-31 / 32-63 of the tile) - the regime in which the matrix pipe keeps its
//     full rate (tools/ubench/mfma_patterns.hip), fragment reads one group of 8 MFMAs ahead; 2 workgroups per CU (64 KB LDS,
//     <= 256 VGPRs each), so that the epilogue of one (and the drain of its stores) overlaps the MFMAs of the other;
//   * the fused epilogue (bias, activation, dropout, residual, pre-activation store, epi_bwd) is gemm_epilogue_lean of gemm_common.h,
//     specialised at compile time on activation / dropout / direction / gathered operand (the kernel's template parameters).
// A-tile LDS layout: per stage K/64 sub-tiles of 64 rows x 32 floats, each exactly the K-contiguous tile of gemm_sk.hip (128-byte rows,
// 16-byte chunks XOR-swizzled with (row >> 1) & 7 on the source address): conflict-free ds_read_b128 fragment reads.
// Placement: workgroup b runs on XCD b % 8; the n-blocks that walk the same m-tiles are put on ONE XCD when that costs no extra round,
// so an A tile is read from HBM / MALL once and from that XCD's L2 by the other n-blocks.
// Padded-row skipping: the device-built 64-row tile schedule (ctts_row_tile_map) is walked instead of all tiles; inactive tiles are
// zero-filled.  Eligibility: ctts_gemm_ws_try.
#include "ctts_common.h"
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef int ws_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ws_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned WS_OOB = GEMM_OOB;

__device__ __forceinline__ ws_i32x4 ws_make_rsrc(const void* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  ws_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  r.z = 0x7FFFFFFE;
  r.w = 0x00020000;
  return r;
}

// see gemm_sk.hip sk_dma16: inline asm keeps hipcc from draining the DMA in front of the fragment reads
__device__ __forceinline__ void ws_dma16(ws_i32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc) : "memory");
}

struct WsArgs {
  int tiles_m;         // ceil(M / 64)
  int n_blocks;        // ceil(N / 128)
  int wg_per_block;    // workgroups that share one 128-column block (grid = n_blocks * wg_per_block)
  int xcd_aligned;     // 1: the n-blocks of one m-tile sequence sit on one XCD (wg_per_block % 8 == 0)
  int debug;           // CTTS_WS_DEBUG: 1 = per-workgroup phase clocks into the tail of sk_ws, 2 = no epilogue, 4 = no in-loop DMA, 8 = no MFMA
};

// KB = K / 32 (compile time: the B fragments are a register array)
template <int KB, bool B_KC, int ACT, bool DROP, bool BWD, bool AUX>
__global__ __launch_bounds__(256, 2) void gemm_ws_kernel(const ctts_gemm_desc d, const WsArgs p) {
  constexpr int HB = KB / 2;                            // K-blocks per half
  constexpr int STAGE = 64 * HB * 32;                   // floats per stage (one K-half of a 64-row tile)
  extern __shared__ __attribute__((aligned(16))) float smem[];      // 2 stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  int nb, wj;
  if (p.xcd_aligned) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    nb = idx % p.n_blocks;
    wj = (idx / p.n_blocks) * 8 + xcd;
  } else {
    nb = blockIdx.x % p.n_blocks;
    wj = blockIdx.x / p.n_blocks;
  }
  const int col0 = nb * 128, wn0 = wave * 32;
  const int n = col0 + wn0 + l31;                       // this lane's output column

  // ---- tile schedule
  typedef const __attribute__((address_space(4))) int32_t* ws_cmap;
  const ws_cmap mmap = (ws_cmap)(uintptr_t)d.tile_map;
  const int n_mt = mmap ? mmap[0] : p.tiles_m;
  if (mmap) {                                           // padded tiles are defined as zero (C and the pre-activation store)
    const int n_zero = (p.tiles_m - n_mt) * p.n_blocks;
    for (int zt = blockIdx.x; zt < n_zero; zt += gridDim.x) {
      const int mi = zt / p.n_blocks, zb = zt - mi * p.n_blocks;
      const int row0 = mmap[1 + n_mt + mi] * 64, c0 = zb * 128;
      const int nrows = min(64, d.M - row0), ncols = min(128, d.N - c0);
      for (int e = tid; e < nrows * 128; e += 256) {
        const int r = e >> 7, c = e & 127;
        if (c < ncols) {
          d.C[(long)(row0 + r) * d.ldc + c0 + c] = 0.f;
          if (d.Z && !d.epi_bwd) d.Z[(long)(row0 + r) * d.ldz + c0 + c] = 0.f;
        }
      }
      if (BWD && d.C_planes) {          // the plane set of a zero tile: 128 columns = 4 K-blocks x 192 bytes per row, as 16-byte stores
        for (int e = tid; e < nrows * 48; e += 256) {
          const int r = e / 48, q = e - r * 48;
          if (q * 8 / 96 * 32 < ncols)
            *reinterpret_cast<uint4*>(d.C_planes + (long)(row0 + r) * d.ldc * 3 + (long)(c0 >> 5) * 96 + q * 8) = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
  }
  const ws_i32x4 ra_src = ws_make_rsrc(d.A);
  const unsigned smem_addr = (unsigned)reinterpret_cast<uintptr_t>(smem);
  // DMA mapping: wave w moves K-blocks w, w + 4, ... of the half; per K-block 8 instructions of 8 rows x 128 bytes
  const int r_in = lane >> 3;                           // row inside an 8-row group
  const int c_in = lane & 7;
  auto issue_half = [&](int slot, int half) {
    const int row0 = (mmap ? mmap[1 + slot] : slot) * 64;
    for (int kbi = wave; kbi < HB; kbi += 4) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int r = g * 8 + r_in;
        const int c = c_in ^ ((r >> 1) & 7);
        const int row = row0 + r;
        const unsigned v = row < d.M ? ((unsigned)row * (unsigned)d.lda + (unsigned)((half * HB + kbi) * 32 + c * 4)) * 4u : WS_OOB;
        ws_dma16(ra_src, smem_addr + (unsigned)(half * STAGE + kbi * 2048 + g * 256) * 4u, v);
      }
    }
  };
  if (wj < n_mt) issue_half(wj, 0);                    // the first A half is in flight while the weights load
  // ---- the weight slice of this wave (behind the first DMA: both round trips overlap): B fragment of K-block kb, k-step j  =  B[k = kb*32 + h*16 + j][n]
  float bf[KB][16];
  if (n < d.N) {
    if (B_KC) {                                         // B[n][k], K-contiguous: 64-byte runs per lane
      const float* bp = d.B + (long)n * d.ldb + h * 16;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(bp + kb * 32 + q * 4);
          bf[kb][4 * q + 0] = v.x; bf[kb][4 * q + 1] = v.y; bf[kb][4 * q + 2] = v.z; bf[kb][4 * q + 3] = v.w;
        }
    } else {                                            // B[k][n], row-contiguous: lanes 0..31 read 128 contiguous bytes per k
      const float* bp = d.B + (long)(h * 16) * d.ldb + n;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) bf[kb][j] = bp[(long)(kb * 32 + j) * d.ldb];
    }
  } else {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) bf[kb][j] = 0.f;
  }

  const int sw = (l31 >> 1) & 7;
  floatx16 acc[2][1];
  auto compute_half = [&](auto HALF) {                  // the half is a compile-time constant: bf[] stays in registers
    constexpr int half = decltype(HALF)::value;
    // fragment reads one group (4 k-steps of both row halves = 2 ds_read_b128) ahead of the MFMAs that use them
    const float* sA = smem + half * STAGE + l31 * 32;
    auto ld = [&](int g, int i) {
      return *reinterpret_cast<const float4*>(sA + (g >> 2) * 2048 + i * 1024 + (((h * 4 + (g & 3)) ^ sw) << 2));
    };
    float4 c0 = ld(0, 0), c1 = ld(0, 1);
#pragma unroll
    for (int g = 0; g < HB * 4; ++g) {
      float4 n0 = c0, n1 = c1;
      if (g + 1 < HB * 4) { n0 = ld(g + 1, 0); n1 = ld(g + 1, 1); }
      __builtin_amdgcn_sched_barrier(0);                // keep the reads of group g + 1 in front of the MFMAs of group g
      const int kb = half * HB + (g >> 2), j = (g & 3) * 4;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.x, bf[kb][j + 0], acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.x, bf[kb][j + 0], acc[1][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.y, bf[kb][j + 1], acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.y, bf[kb][j + 1], acc[1][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.z, bf[kb][j + 2], acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.z, bf[kb][j + 2], acc[1][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.w, bf[kb][j + 3], acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.w, bf[kb][j + 3], acc[1][0], 0, 0, 0);
      c0 = n0; c1 = n1;
    }
  };
  const bool wave_has_cols = col0 + wn0 < d.N;

  int slot = wj;
  long long t_start = 0, t_loop = 0, t_w1 = 0, t_w2 = 0, t_epi = 0, t_c0 = 0, t_c1 = 0;
  if (p.debug & 1) t_start = __builtin_readcyclecounter();
  int n_done = 0;
  for (; slot < n_mt; slot += p.wg_per_block) {
    const int row0 = (mmap ? mmap[1 + slot] : slot) * 64;
    // half 0 of this tile has landed, stage 1 is free, the stores of the previous epilogue are out (the other workgroup of the CU has
    // the matrix pipe meanwhile).  The second wait is free in hardware and tells hipcc that nothing is pending: loads of the epilogue
    // that it still tracks (results unused on some path) would otherwise cost a vmcnt(0) at the first fragment read below - behind
    // the DMA of the next half.  (Before the first tile it is the wait for the weight loads.)
    long long t0 = 0, t1 = 0;
    if (p.debug & 1) t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0);
    if (p.debug & 1) { t1 = __builtin_readcyclecounter(); if (n_done == 0) t_loop = t1; else t_w1 += t1 - t0; }
    if (!(p.debug & 4)) issue_half(slot, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    if (!(p.debug & 8)) compute_half(std::integral_constant<int, 0>{});
    if (p.debug & 1) { t0 = __builtin_readcyclecounter(); t_c0 += t0 - t1; }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");        // half 1 has landed; stage 0 is free
    if (p.debug & 1) { t1 = __builtin_readcyclecounter(); t_w2 += t1 - t0; }
    const int next = slot + p.wg_per_block;
    if (next < n_mt && !(p.debug & 4)) issue_half(next, 0);
    if (!(p.debug & 8)) compute_half(std::integral_constant<int, 1>{});
    if (p.debug & 1) { t0 = __builtin_readcyclecounter(); t_c1 += t0 - t1; }
    // BWD instantiations carry the plane-writing variant (d.C_planes: the dZ handed to the previous layer arrives with its operand planes)
    if (wave_has_cols && !(p.debug & 2)) gemm_epilogue_lean<2, 1, ACT, DROP, BWD, AUX, false, BWD>(d, acc, d.C, 0, row0, col0, 0, wn0, l31, h, d.M, d.N);
    if (p.debug & 1) { t1 = __builtin_readcyclecounter(); t_epi += t1 - t0; }
    ++n_done;
  }
  if ((p.debug & 1) && d.sk_ws && tid == 0 && blockIdx.x < 1024) {
    long long* o = reinterpret_cast<long long*>(reinterpret_cast<char*>(d.sk_ws) + d.sk_ws_bytes - 65536) + blockIdx.x * 8;
    const long long t_end = __builtin_readcyclecounter();
    o[0] = t_end - t_start; o[1] = t_loop - t_start; o[2] = t_w1; o[3] = t_w2; o[4] = t_c0; o[5] = t_c1; o[6] = t_epi; o[7] = n_done;
  }
}

template <int KB, bool B_KC, int ACT, bool DROP, bool BWD, bool AUX>
int ws_launch(const ctts_gemm_desc& d, const WsArgs& p, hipStream_t st) {
  const size_t lds = (size_t)64 * KB * 32 * sizeof(float);             // 2 stages of one K-half
  // every call: the attribute is per device, and a process may drive several (cheap next to the launch)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<KB, B_KC, ACT, DROP, BWD, AUX>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((gemm_ws_kernel<KB, B_KC, ACT, DROP, BWD, AUX>), dim3(p.n_blocks * p.wg_per_block), dim3(256), lds, st, d, p);
  CTTS_CHECK_LAUNCH("ctts_gemm(weight-stationary)");
  return 1;
}

template <int KB, bool B_KC, int ACT>
int ws_launch_act(const ctts_gemm_desc& d, const WsArgs& p, hipStream_t st) {
  const bool drop = d.p_drop > 0.f;
  if (d.epi_bwd)                                                       // the gathered operand is Z, read iff there is an activation
    return drop ? ws_launch<KB, B_KC, ACT, true, true, ACT != 0>(d, p, st) : ws_launch<KB, B_KC, ACT, false, true, ACT != 0>(d, p, st);
  if (d.R) return drop ? ws_launch<KB, B_KC, ACT, true, false, true>(d, p, st) : ws_launch<KB, B_KC, ACT, false, false, true>(d, p, st);
  return drop ? ws_launch<KB, B_KC, ACT, true, false, false>(d, p, st) : ws_launch<KB, B_KC, ACT, false, false, false>(d, p, st);
}

template <int KB, bool B_KC>
int ws_launch_layout(const ctts_gemm_desc& d, const WsArgs& p, hipStream_t st) {
  switch (d.act) {
    case 0: return ws_launch_act<KB, B_KC, 0>(d, p, st);
    case 1: return ws_launch_act<KB, B_KC, 1>(d, p, st);
    case 2: return ws_launch_act<KB, B_KC, 2>(d, p, st);
    case 4: return ws_launch_act<KB, B_KC, 4>(d, p, st);
    default: return 0;                                                 // tanh: no K = 256 linear uses it - the other kernels take it
  }
}

int ws_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

static int g_ws_enabled = -1;     // -1: take CTTS_WS (default on) at the first launch

extern "C" int ctts_gemm_ws_enable(int on) {
  const int prev = g_ws_enabled < 0 ? ws_env("CTTS_WS", 1) : g_ws_enabled;
  g_ws_enabled = on ? 1 : 0;
  return prev;
}

// eligibility + grid of the weight-stationary kernel
static bool ws_plan(const ctts_gemm_desc& d, WsArgs& p) {
  if (g_ws_enabled < 0) g_ws_enabled = ws_env("CTTS_WS", 1) ? 1 : 0;
  static const int min_rows = ws_env("CTTS_WS_MIN_ROWS", 4096);
  static const int slots = ws_env("CTTS_WS_SLOTS", 512);          // 256 CUs x 2 workgroups
  static const int align = ws_env("CTTS_WS_XCD_ALIGN", 1);
  static const int debug = ws_env("CTTS_WS_DEBUG", 0);
  if (!g_ws_enabled) return false;
  if (d.nb0 * d.nb1 != 1 || (d.lens && (d.lim_m || d.lim_n || d.lim_k)) || d.E || d.split_k > 1 || d.conv_T > 0) return false;
  if (!d.a_kc || d.K != 256) return false;
  if (d.M < min_rows || d.N < 64) return false;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(d.A) || !al16(d.B) || ((d.lda | d.ldb) & 3)) return false;
  if ((long)(d.M + 64) * d.lda * 4 >= 0x7FFF0000L) return false;
  if (!gemm_fits32(d.C, d.M, d.ldc, d.N) || !gemm_fits32(d.Z, d.M, d.ldz, d.N) || !gemm_fits32(d.R, d.M, d.ldr, d.N)) return false;
  if (d.row_lens && !d.tile_map) return false;           // padded-row zeroing needs the schedule
  // an output plane set (round 6) is written by the backward epilogue only, for whole 32-column K-blocks
  if (d.C_planes && (!d.epi_bwd || (d.ldc & 31) || (d.N & 31) || !al16(d.C_planes) || (long)(d.M + 64) * d.ldc * 6 >= 0x7FFF0000L)) return false;
  // combinations no K = 256 launch of the model uses are not compiled in: row scale, tanh, a pre-activation store without activation
  if (d.rowscale || d.act == 3 || (d.Z && !d.act && !d.epi_bwd)) return false;
  if (d.tile_map == reinterpret_cast<const int32_t*>(1)) return false;
  p.tiles_m = (d.M + 63) / 64;
  p.n_blocks = (d.N + 127) / 128;
  if (p.n_blocks > slots) return false;
  int per = slots / p.n_blocks;
  if (per > p.tiles_m) per = p.tiles_m;
  const int rounds = (p.tiles_m + per - 1) / per;
  const int per8 = 8 * ((slots / 8) / p.n_blocks);
  p.xcd_aligned = 0;
  if (align && per8 >= 8 && per8 <= p.tiles_m && (p.tiles_m + per8 - 1) / per8 <= rounds) { per = per8; p.xcd_aligned = 1; }
  p.wg_per_block = per;
  p.debug = debug;
  return true;
}

// 1 = launched, 0 = not eligible (the caller continues with the other kernels), < 0 error
int ctts_gemm_ws_try(const ctts_gemm_desc& d, hipStream_t st) {
  WsArgs p;
  if (!ws_plan(d, p)) return 0;
  return d.b_kc ? ws_launch_layout<8, true>(d, p, st) : ws_launch_layout<8, false>(d, p, st);
}

extern "C" int ctts_gemm_takes_weight_stationary(const ctts_gemm_desc* dp) {
  if (!dp) return 0;
  ctts_gemm_desc d = *dp;
  if (d.nb0 < 1) d.nb0 = 1;
  if (d.nb1 < 1) d.nb1 = 1;
  WsArgs p;
  return ws_plan(d, p) ? 1 : 0;
}
