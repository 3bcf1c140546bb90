// Micro-benchmark: does the matrix pipe keep its rate when the SAME wave also issues what a GEMM K-block needs - LDS fragment reads,
// LDS-DMA loads, a barrier?  Body per iteration = one 128x128x32 K-block of a 4-wave workgroup: 64 MFMAs (2x2 register tile, 16 k-steps)
//   mode 0: MFMAs only (operands resident)
//   mode 1: + 16 ds_read_b128 whose results ARE the next operands (software-pipelined one quarter ahead, as hipcc schedules gemm_sk<2,2>)
//   mode 2: mode 1 + 8 LDS-DMA instructions (buffer_load ... lds, 1 KiB each, L2-resident source) + s_waitcnt vmcnt(0) at the top
//   mode 3: mode 2 + s_barrier at the top
//   mode 4: mode 3 with the DMA instructions spread over the MFMA stream (one per 8 MFMAs) instead of a burst at the top
//   mode 5: mode 3, but the first quarter's fragments of a block are read AFTER its barrier (as a real kernel must), not prefetched
//   mode 6: mode 5 + the DMA source walks a 256 MB footprint (L2 / MALL misses) instead of 4 KB
//   mode 7: mode 6 + per-DMA predicate arithmetic (add, compare, select) and ~40 scalar bookkeeping instructions per block
// 1 or 2 workgroups per CU (= waves per SIMD).  TFLOP/s over ~0.4 s per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(voff), "s"(rsrc) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const float* src) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 8192];          // 2 stages x 32 KB
  for (int i = threadIdx.x; i < 2 * 8192; i += 256) smem[i] = src[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const unsigned long long a64 = reinterpret_cast<unsigned long long>(src);
  i32x4 rsrc; rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a64); rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a64 >> 32)); rsrc.z = 0x7FFFFFFE; rsrc.w = 0x00020000;
  const unsigned smem_addr = (unsigned)reinterpret_cast<uintptr_t>(smem);
  floatx16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 fa[2][4], fb[2][4];
  const float* sA = smem + (wave >> 1) * 2048 + l31 * 32 + h * 16;
  const float* sB = smem + 4096 + (wave & 1) * 2048 + l31 * 32 + h * 16;
  for (int i = 0; i < 2; ++i) for (int q = 0; q < 4; ++q) { fa[i][q] = *reinterpret_cast<const float4*>(sA + i * 1024 + q * 4); fb[i][q] = *reinterpret_cast<const float4*>(sB + i * 1024 + q * 4); }
  int stage = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 2 || MODE == 3 || MODE == 5) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dma16(rsrc, smem_addr + (unsigned)((stage ^ 1) * 32768 + (wave * 8 + j) * 1024), (unsigned)(((it * 8 + j) & 3) * 1024 + lane * 16));
    }
    if (MODE >= 6) {
      // rows of 1 KB (conv activations), 8 rows per instruction, a different 128-byte column block every iteration, tile per workgroup
      const unsigned tile = (unsigned)(blockIdx.x * 977 + it) % 4000u;                  // 4000 tiles x 64 KB (+ 256 KB reach) < 256 MB
      unsigned book = 0;
      if (MODE == 7) {                                                                   // scalar bookkeeping stand-in
        unsigned x = (unsigned)it * 2654435761u + blockIdx.x;
#pragma unroll
        for (int r = 0; r < 20; ++r) { x = x * 1664525u + 1013904223u; book += x >> 31; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned v = tile * 65536u + (unsigned)((wave * 8 + j) * 8 + (lane >> 3)) * 1024u + (unsigned)((it & 7) * 128 + (lane & 7) * 16);
        if (MODE == 7) v = ((unsigned)((lane >> 3) + j + (it & 3) - 2 + (book & 1)) < 1000u) ? v : 0x80000000u;
        dma16(rsrc, smem_addr + (unsigned)((stage ^ 1) * 32768 + (wave * 8 + j) * 1024), v);
      }
    }
    const float* cA = sA + stage * 8192;
    const float* cB = sB + stage * 8192;
    if (MODE >= 5) {        // this block's first quarter, read after the barrier
#pragma unroll
      for (int i = 0; i < 2; ++i) { fa[i][0] = *reinterpret_cast<const float4*>(cA + i * 1024); fb[i][0] = *reinterpret_cast<const float4*>(cB + i * 1024); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 na[2], nb[2];
      if (MODE >= 1 && !(MODE >= 5 && q == 3)) {       // fragments of the NEXT quarter (modes 1-4: also the next block's first quarter)
#pragma unroll
        for (int i = 0; i < 2; ++i) { na[i] = *reinterpret_cast<const float4*>(cA + i * 1024 + ((q + 1) & 3) * 4); nb[i] = *reinterpret_cast<const float4*>(cB + i * 1024 + ((q + 1) & 3) * 4); }
      }
      if (MODE == 4) {
        dma16(rsrc, smem_addr + (unsigned)((stage ^ 1) * 32768 + (wave * 8 + 2 * q) * 1024), (unsigned)(((it * 8 + 2 * q) & 3) * 1024 + lane * 16));
        dma16(rsrc, smem_addr + (unsigned)((stage ^ 1) * 32768 + (wave * 8 + 2 * q + 1) * 1024), (unsigned)(((it * 8 + 2 * q + 1) & 3) * 1024 + lane * 16));
      }
      const float* pa0 = reinterpret_cast<const float*>(&fa[0][q]); const float* pa1 = reinterpret_cast<const float*>(&fa[1][q]);
      const float* pb0 = reinterpret_cast<const float*>(&fb[0][q]); const float* pb1 = reinterpret_cast<const float*>(&fb[1][q]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = MFMA(pa0[e], pb0[e], acc[0][0]); acc[0][1] = MFMA(pa0[e], pb1[e], acc[0][1]);
        acc[1][0] = MFMA(pa1[e], pb0[e], acc[1][0]); acc[1][1] = MFMA(pa1[e], pb1[e], acc[1][1]);
      }
      if (MODE >= 1 && !(MODE >= 5 && q == 3)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { fa[i][(q + 1) & 3] = na[i]; fb[i][(q + 1) & 3] = nb[i]; }
      }
    }
    if (MODE >= 2) stage ^= 1;
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(int grid, int iters, float* out, const float* src, double seconds) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)grid * 4 * iters * 64 * (2.0 * 32 * 32 * 2);
  std::vector<double> tf; double el = 0;
  while (el < seconds) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, src); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); tf.push_back(flop / (ms * 1e-3) / 1e12); el += ms * 1e-3;
  }
  double s = 0; size_t n = tf.size(), a = n / 2; for (size_t i = a; i < n; ++i) s += tf[i];
  return s / (n - a);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 0.4;
  float* out; hipMalloc(&out, (size_t)512 * 256 * 4);
  float* src; hipMalloc(&src, (size_t)256 << 20);
  { std::vector<float> hr(1 << 18); unsigned s = 12345u; for (auto& v : hr) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f); }
    for (size_t o = 0; o < ((size_t)256 << 20); o += (1 << 20)) hipMemcpy((char*)src + o, hr.data(), 1 << 20, hipMemcpyHostToDevice); }
  printf("mode | TFLOP/s at 1 and 2 workgroups per CU (1 / 2 waves per SIMD)\n");
#define ROW(M) { printf("  %d  | %7.1f %7.1f\n", M, run<M>(256, 512, out, src, seconds), run<M>(512, 256, out, src, seconds)); }
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7)
  return 0;
}
