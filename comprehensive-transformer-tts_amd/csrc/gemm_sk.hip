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

constexpr int SK_STAGE = 4096;                 // floats per LDS stage: A tile 2048 + B tile 2048 (16 KB)
constexpr unsigned SK_OOB = 0x80000000u;
constexpr int SK_FLAG_WORDS = 4096;            // header of the workspace: flags[0..2047], error word at [2048]
constexpr int SK_SLAB = 4096;                  // floats per contribution slab (64x64 accumulator tile)
constexpr int SK_MAX_WG = 2048;

struct SkArgs {
  int tiles_m, tiles_n;      // static tile grid
  int nkb;                   // K-blocks per tile when no K-block schedule is given
  int gw;                    // n-tiles per schedule group
  int whole_tiles;           // 1: never split a tile
  int accumulate;            // 1: C += alpha * acc (weight gradients), no other epilogue
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
// `s_waitcnt vmcnt(0)` + `s_barrier` stay the only synchronisation between the DMA and its readers.
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

// ---- K-contiguous operand tile: 64 rows x 32 k, element (r, k) = P[(ext0 + r) * ld + k]; conv view: row r is shifted by -pad rows and
//      the chunk is zero unless 0 <= t(r) + tap - pad < T, tap = k / cin (cin % 32 == 0: one tap per K-block)
template <bool CONV>
struct SkLoadKC {
  unsigned voff[2];
  int trow[2];
  __device__ __forceinline__ void set_piece(int ext0, int ext_lim, long ld, int wave, int lane, int T) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (j * 4) ^ (lane >> 4);          // = (lane & 7) ^ ((r >> 1) & 7)
      const int row = ext0 + r;
      voff[j] = row < ext_lim ? (unsigned)(((long)row * ld + c * 4) * 4) : SK_OOB;
      if (CONV) {
        int t = ext0 % T + r;                                       // T >= 64 (host check)
        trow[j] = t >= T ? t - T : t;
      }
    }
  }
  // tap_m_pad = k0 / cin - pad (wave-uniform)
  __device__ __forceinline__ void issue(sk_i32x4 rsrc, unsigned lds, unsigned soff, int tap_m_pad, int T) const {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned v = voff[j];
      if (CONV) v = ((unsigned)(trow[j] + tap_m_pad) < (unsigned)T) ? v : SK_OOB;
      sk_dma16(rsrc, lds + j * 1024, v, soff);
    }
  }
};

// ---- row-contiguous operand tile: 32 k-rows x 64 columns, element (k, c) = P[k * ld + ext0 + c]; conv view (weight gradient): row k is
//      shifted by -pad rows and the chunk is zero unless 0 <= t(k) + c / cin - pad < T
template <bool CONV>
struct SkLoadRC {
  unsigned voff[2];
  int ctap;            // column tap - pad (piece invariant)
  int klocal[2];
  __device__ __forceinline__ void set_piece(int ext0, int ext_lim, long ld, int wave, int lane, int cin, int pad) {
    const int col = ext0 + 4 * (lane & 15);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      klocal[j] = (wave * 2 + j) * 4 + (lane >> 4);
      voff[j] = col < ext_lim ? (unsigned)(((long)klocal[j] * ld + col) * 4) : SK_OOB;
    }
    ctap = CONV ? col / cin - pad : 0;
  }
  __device__ __forceinline__ void issue(sk_i32x4 rsrc, unsigned lds, unsigned soff, int k0, int K, int T, float rcpT) const {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int krow = k0 + klocal[j];
      bool ok = krow < K;
      if (CONV) ok = ok && ((unsigned)(sk_mod(krow, T, rcpT) + ctap) < (unsigned)T);
      const unsigned v = ok ? voff[j] : SK_OOB;
      sk_dma16(rsrc, lds + j * 1024, v, soff);
    }
  }
};

template <bool KC>
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
    const float* p = s + (h * 16) * 64 + ext0 + l31;
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = p[j * 64];
  }
}

__device__ __forceinline__ void sk_zero_tile(const ctts_gemm_desc& d, int row0, int col0) {
  const int nrows = min(64, d.M - row0), ncols = min(64, d.N - col0);
  for (int e = threadIdx.x; e < nrows * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (c < ncols) {
      d.C[(long)(row0 + r) * d.ldc + col0 + c] = 0.f;
      if (d.Z) d.Z[(long)(row0 + r) * d.ldz + col0 + c] = 0.f;
    }
  }
}

template <bool A_KC, bool B_KC, bool CONV>
__global__ __launch_bounds__(256, 4) void gemm_sk_kernel(const ctts_gemm_desc d, const SkArgs p) {
  constexpr bool TN = !A_KC && !B_KC;
  constexpr bool CONV_A = CONV && A_KC;
  constexpr bool CONV_B = CONV && TN;
  __shared__ __attribute__((aligned(16))) float smem[2 * SK_STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;

  // ---- schedule inputs that live in device memory
  // (read through the constant address space = scalar loads: a vector load here would make hipcc wait vmcnt(0) inside the K loop,
  //  which drains the DMA stream; the maps were written by an earlier kernel, so the scalar cache is coherent for them)
  typedef const __attribute__((address_space(4))) int32_t* sk_cmap;
  const sk_cmap mmap = A_KC ? (sk_cmap)(uintptr_t)d.tile_map : (sk_cmap)0;   // m-tile schedule: [0] = active count, then tile ids
  const sk_cmap kmap = TN ? (sk_cmap)(uintptr_t)d.tile_map : (sk_cmap)0;     // TN: the same map as a K-block schedule (64 rows per entry)
  const int n_mt = mmap ? mmap[0] : p.tiles_m;
  const int nkb = kmap ? 2 * kmap[0] : p.nkb;

  // ---- padded m-tiles are defined as zero: stores only, spread over the grid
  if (A_KC && mmap) {
    const int n_zero = (p.tiles_m - n_mt) * p.tiles_n;
    for (int zt = blockIdx.x; zt < n_zero; zt += gridDim.x) {
      const int mi = zt / p.tiles_n;
      sk_zero_tile(d, mmap[1 + n_mt + mi] * 64, (zt - mi * p.tiles_n) * 64);
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
  float* slabs = reinterpret_cast<float*>(p.ws + SK_FLAG_WORDS);
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, 0x7FFFFFFE, 0x00020000);
  const int T = d.conv_T > 0 ? d.conv_T : 1;
  const float rcpT = 1.0f / (float)T;
  const int cin = d.conv_cin > 0 ? d.conv_cin : 32;

  using LA = typename std::conditional<A_KC, SkLoadKC<CONV_A>, SkLoadRC<false>>::type;
  using LB = typename std::conditional<B_KC, SkLoadKC<false>, SkLoadRC<CONV_B>>::type;
  LA la; LB lb;

  auto decode = [&](const SkPiece& pc, int& row0, int& col0) {
    int mslot, nt;
    sk_tile_decode(rg.T0 + pc.t, n_mt, p.gw, mslot, nt);
    row0 = (mmap ? mmap[1 + mslot] : mslot) * 64;
    col0 = nt * 64;
  };
  auto k0_of = [&](int kb) -> int {
    if (TN && kmap) return (kmap[1 + (kb >> 1)] * 2 + (kb & 1)) * 32;
    return kb * 32;
  };

  // ---- loader cursor (runs one K-block ahead of the MFMAs, across piece boundaries)
  int lu = rg.hi;
  SkPiece lp;
  bool have_l = sk_next_piece(lu, rg.lo, nkb, lp);
  int lkb = lp.kb_lo, lk0 = 0, ltap = 0, lkin = 0;
  auto loader_set_piece = [&]() {
    int row0, col0;
    decode(lp, row0, col0);
    if constexpr (A_KC) la.set_piece(row0, d.M, d.lda, wave, lane, T);
    else la.set_piece(row0, d.M, d.lda, wave, lane, cin, 0);
    if constexpr (B_KC) lb.set_piece(col0, d.N, d.ldb, wave, lane, T);
    else lb.set_piece(col0, d.N, d.ldb, wave, lane, cin, d.conv_pad);
    lkb = lp.kb_lo;
    lk0 = k0_of(lkb);
    if (CONV_A) { ltap = lk0 / cin; lkin = lk0 - ltap * cin; }
  };
  auto loader_issue = [&](int stage) {
    const unsigned sA = smem_addr + (unsigned)(stage * SK_STAGE + wave * 512) * 4u;
    const unsigned sB = sA + 2048 * 4;
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
      lk0 = k0_of(lkb);
      if (CONV_A) { lkin += 32; if (lkin >= cin) { lkin -= cin; ++ltap; } }
    }
  };

  loader_set_piece();
  loader_issue(0);
  loader_advance();

  // ---- compute cursor
  int cu = rg.hi;
  SkPiece cp;
  sk_next_piece(cu, rg.lo, nkb, cp);
  int ckb = cp.kb_lo;
  floatx16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

  int stage = 0;
  while (true) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // block `ckb` has landed; the other stage is free
    if (have_l) {
      loader_issue(stage ^ 1);
      loader_advance();
    }
    {
      const float* sA = smem + stage * SK_STAGE;
      const float* sB = sA + 2048;
      float fa[16], fb[16];
      sk_fetch<A_KC>(sA, wm0, l31, h, fa);
      sk_fetch<B_KC>(sB, wn0, l31, h, fb);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc[0][0], 0, 0, 0);
    }
    ++ckb;
    stage ^= 1;
    if (ckb < cp.kb_hi) continue;

    // ---------------- the piece is complete
    int row0, col0;
    decode(cp, row0, col0);
    if (cp.kb_hi < nkb) {
      // contribution: slab (write-through stores) + flag
      const unsigned base = (unsigned)blockIdx.x * (SK_SLAB * 4) + (unsigned)(wave * 1024 + lane * 4) * 4u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sk_u32x4 v;
        v.x = __float_as_uint(acc[0][0][4 * q + 0]); v.y = __float_as_uint(acc[0][0][4 * q + 1]);
        v.z = __float_as_uint(acc[0][0][4 * q + 2]); v.w = __float_as_uint(acc[0][0][4 * q + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_src, base + q * 1024u, 0, 16);      // aux 16 = sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store((sk_gu32*)(flags + blockIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (cp.kb_lo > 0) {
        // owner of a cut tile: add the slabs of the workgroups below, nearest first, until the tile's unit 0 is covered
        const int tile_lo = cp.t * nkb;
        const int Ux = (rg.T1 - rg.T0) * nkb;
        int upper = rg.lo;                                  // start of the range that is already summed
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
          const unsigned base = (unsigned)src * (SK_SLAB * 4) + (unsigned)(wave * 1024 + lane * 4) * 4u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const sk_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_src, base + q * 1024u, 0, 0);
            acc[0][0][4 * q + 0] += __uint_as_float(v.x); acc[0][0][4 * q + 1] += __uint_as_float(v.y);
            acc[0][0][4 * q + 2] += __uint_as_float(v.z); acc[0][0][4 * q + 3] += __uint_as_float(v.w);
          }
        }
      }
      if (p.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * h, n = col0 + wn0 + l31;
          if (m < d.M && n < d.N) d.C[(long)m * d.ldc + n] += d.alpha * acc[0][0][r];
        }
      } else {
        gemm_epilogue<1, 1>(d, acc, d.C, 0, row0, col0, wm0, wn0, l31, h, d.M, d.N);
      }
    }
    if (!sk_next_piece(cu, rg.lo, nkb, cp)) break;
    ckb = cp.kb_lo;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  }
}

template <bool A_KC, bool B_KC, bool CONV>
int sk_launch(const ctts_gemm_desc& d, const SkArgs& p, int grid, hipStream_t st) {
  hipLaunchKernelGGL((gemm_sk_kernel<A_KC, B_KC, CONV>), dim3(grid), dim3(256), 0, st, d, p);
  CTTS_CHECK_LAUNCH("ctts_gemm(stream-K)");
  return 1;
}

int sk_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

extern "C" size_t ctts_gemm_workspace_bytes(void) { return (size_t)SK_FLAG_WORDS * 4 + (size_t)SK_MAX_WG * SK_SLAB * 4; }

int ctts_gemm_sk_try(const ctts_gemm_desc& din, hipStream_t st) {
  static const int enabled = sk_env("CTTS_SK", 1);
  static const int wg_per_xcd = sk_env("CTTS_SK_W", 128);            // 128 = 4 workgroups per CU (32 KB LDS each)
  static const int min_units = sk_env("CTTS_SK_MIN_UNITS", 4096);    // below this the launch is latency bound either way
  static const int split_from = sk_env("CTTS_SK_SPLIT_NKB", 24);     // tiles are cut only when K has at least this many blocks
  static const int force_gw = sk_env("CTTS_SK_GW", 0);
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
    if (d.conv_cin % 32 != 0 || d.conv_T < 64) return 0;
    if (d.conv_on_b ? !tn : !d.a_kc) return 0;
    if (tn && (long)d.K >= (1L << 24)) return 0;
  }
  const long a_ext = d.a_kc ? ((long)(d.M + 64) * d.lda + d.K) : ((long)(d.K + 64) * d.lda + d.M);
  const long b_ext = d.b_kc ? ((long)(d.N + 64) * d.ldb + d.K) : ((long)(d.K + 64) * d.ldb + d.N);
  if (a_ext * 4 >= 0x7FFF0000L || b_ext * 4 >= 0x7FFF0000L) return 0;
  if (d.row_lens && !d.tile_map) return 0;                           // padded-row skipping needs the device-built schedule here
  if (d.tile_map == reinterpret_cast<const int32_t*>(1)) return 0;
  if (tn && d.tile_map && (d.K + 63) / 64 < 1) return 0;

  SkArgs p;
  p.tiles_m = (d.M + 63) / 64;
  p.tiles_n = (d.N + 63) / 64;
  p.nkb = (d.K + 31) / 32;
  const long units = (long)p.tiles_m * p.tiles_n * p.nkb;
  if (units < min_units) return 0;
  p.whole_tiles = p.nkb < split_from ? 1 : 0;
  p.accumulate = tn && d.split_k > 1 ? 1 : 0;
  if (!tn && d.split_k > 1) return 0;
  // schedule groups: 4 groups of n-tiles when that divides (an XCD pair shares a group), else one group
  p.gw = (p.tiles_n % 4 == 0) ? p.tiles_n / 4 : p.tiles_n;
  if (force_gw > 0 && p.tiles_n % force_gw == 0) p.gw = force_gw;
  p.ws = reinterpret_cast<unsigned*>(d.sk_ws);
  // grid: W workgroups per XCD, fewer when the launch is small (>= 16 units each)
  long W = units / (8 * 16);
  if (W > wg_per_xcd) W = wg_per_xcd;
  if (W < 1) W = 1;
  const int grid = (int)W * 8;
  if (grid > SK_MAX_WG) return 0;
  if (d.a_kc && d.b_kc) return conv ? sk_launch<true, true, true>(d, p, grid, st) : sk_launch<true, true, false>(d, p, grid, st);
  if (d.a_kc && !d.b_kc) return conv ? sk_launch<true, false, true>(d, p, grid, st) : sk_launch<true, false, false>(d, p, grid, st);
  return conv ? sk_launch<false, false, true>(d, p, grid, st) : sk_launch<false, false, false>(d, p, grid, st);
}
