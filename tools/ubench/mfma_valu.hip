// Micro-benchmark: does one wave's VALU work overlap with its own fp32 MFMAs on gfx950?  One wave per SIMD (grid = 1024 blocks of 64),
// cycles per iteration from s_memtime.  Cases: MFMA chain only, VALU only, both interleaved, two waves per SIMD each doing both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: 16 dependent MFMAs / iter   1: NV VALU ops / iter   2: interleaved   3: two independent MFMA chains   4: quarter-rate int mul
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters, int nv) {
  floatx16 acc, acc2;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
  uint32_t u0 = threadIdx.x, u1 = threadIdx.x * 3 + 1;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        if (MODE == 3) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc2, 0, 0, 0);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int x = 0; x < 8; ++x) { v0 = v0 * b + v1; v1 = v1 * b + v2; v2 = v2 * b + v3; v3 = v3 * b + v0; }
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int x = 0; x < 8; ++x) { v0 = v0 * b + v1; v1 = v1 * b + v2; v2 = v2 * b + v3; v3 = v3 * b + v0; }
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int x = 0; x < 8; ++x) { u0 = u0 * 0x9E3779B1u + u1; u1 = u1 * 0x7feb352du + u0; }
      }
    } else if (MODE == 5) {     // MFMA + quarter-rate int mul interleaved
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int x = 0; x < 8; ++x) { u0 = u0 * 0x9E3779B1u + u1; u1 = u1 * 0x7feb352du + u0; }
      }
    } else if (MODE == 6) {     // v_exp_f32 chain
#pragma unroll
      for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int x = 0; x < 8; ++x) { v0 = __builtin_amdgcn_exp2f(v0) ; v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3); }
      }
    }
  }
  long long t1 = clock64();
  float s = v0 + v1 + v2 + v3 + (float)(u0 ^ u1);
  for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 64 * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s blocks %5d  kernel %8.1f us   clock64 ticks/iter (block 0): %8.1f   us/iter-per-wave %7.3f\n", name, blocks, ms * 1e3,
         (double)h[0] / iters, ms * 1e3 / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int it = 2000;
  for (int blocks : {1024, 2048}) {   // 1 and 2 waves per SIMD
    run<0>("16 dependent MFMA 32x32x2 / iter", blocks, it);
    run<3>("2 x 16 MFMA (two chains) / iter", blocks, it);
    run<1>("512 dependent-ish v_fma / iter", blocks, it);
    run<2>("16 MFMA + 512 v_fma interleaved / iter", blocks, it);
    run<4>("256 v_mul_lo_u32+add / iter", blocks, it);
    run<5>("16 MFMA + 256 v_mul_lo interleaved / iter", blocks, it);
    run<6>("512 v_exp_f32 / iter", blocks, it);
  }
  return 0;
}
