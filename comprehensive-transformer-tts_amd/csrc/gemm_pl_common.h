// Shared pieces of the plane kernels (gemm_pl.hip: forward / data-gradient layout; gemm_plw.hip: weight-gradient layout): vector types,
// the LDS-DMA instruction, the bf16 MFMA, sizes of the 128 x 256 tile and of the stream-K workspace.
#pragma once
#include "ctts_common.h"
#include "gemm_common.h"
#include "sk_plan.h"

namespace {

typedef unsigned int pl_u32x4 __attribute__((ext_vector_type(4)));
typedef int pl_i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pl_floatx2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned int pl_gu32;

// CTTS_PL_DEBUG bits (tools builds only: -DCTTS_PL_TOOLS; the product build compiles them out - their scalar branches and the clock
// stamps cost the conv instantiation three spilled VGPRs with reloads inside the K loop)
#ifdef CTTS_PL_TOOLS
#define PL_DBG(bit) (p.debug & (bit))
#else
#define PL_DBG(bit) 0
#endif

constexpr unsigned PL_OOB = 0x80000000u;
constexpr int PL_BM = 128, PL_BN = 256;
constexpr int PL_ROW = 64;                                    // bytes per plane row of a 32-deep K-block
constexpr int PL_A_PLANE = PL_BM * PL_ROW, PL_B_PLANE = PL_BN * PL_ROW;
constexpr int PL_STAGE = 3 * (PL_A_PLANE + PL_B_PLANE);       // 73,728 bytes
constexpr int PL_SLAB = PL_BM * PL_BN;                        // floats per workgroup slab
constexpr int PL_MAX_UTT = 256;
constexpr int PL_MAX_WG = 2048;                               // flags[0 .. 2047], error word at [2048]: the layout of gemm_sk.hip
constexpr int PL_SLAB_FLOATS_MAX = 2048 * 4096;


__device__ __forceinline__ pl_i32x4 pl_make_rsrc(const void* base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  pl_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  r.z = 0x7FFFFFFE;
  r.w = 0x00020000;
  return r;
}

// One LDS-DMA instruction (64 lanes x 16 bytes -> LDS [lds_addr + lane * 16)); inline asm for the reason given in gemm_sk.hip: hipcc's
// waitcnt pass must not know about it (it would drain the DMA in front of every fragment read).  m0 is used by nothing else here.
__device__ __forceinline__ void pl_dma16(pl_i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}

__device__ __forceinline__ floatx16 pl_mma(const pl_u32x4 a, const pl_u32x4 b, const floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pl_bf16x8, a), __builtin_bit_cast(pl_bf16x8, b), c, 0, 0, 0);
}

}  // namespace
