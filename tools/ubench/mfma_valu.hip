// Micro-benchmark: does fp32 VALU work overlap with fp32 MFMAs on a gfx950 SIMD?
//   one wave per SIMD (256-thread workgroups, one per CU):  MFMA only / VALU only / both from the same wave (forced interleave)
//   two waves per SIMD (512-thread workgroups): waves 0-3 issue only MFMAs, waves 4-7 only VALU - cross-wave overlap
// MFMAs rotate over 4 independent accumulators and the VALU work over 8 independent chains, so neither side stalls on its own
// dependencies.  Cycles per iteration from clock64 (s_memtime), per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define MFMA_GROUP() __builtin_amdgcn_sched_group_barrier(0x008, 1, 0)
#define VALU_GROUP(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)

template <int MODE>   // 0: 16 MFMA   1: 256 VALU   2: 16 MFMA + 256 VALU interleaved 1:16   3: by wave: waves 0-3 MFMA, waves 4-7 VALU
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  floatx16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[8];
  for (int x = 0; x < 8; ++x) v[x] = a + x;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 3], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < 16; ++x) v[x & 7] = v[x & 7] * b + a;
        MFMA_GROUP(); VALU_GROUP(16);
      }
    } else if (do_mfma) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 3], 0, 0, 0);
    } else if (do_valu) {
#pragma unroll
      for (int x = 0; x < 256; ++x) v[x & 7] = v[x & 7] * b + a;
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int x = 0; x < 8; ++x) s += v[x];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int iters) {
  const int blocks = 256;
  float* out; long long* cyc;
  (void)hipMalloc(&out, blocks * 512 * sizeof(float)); (void)hipMalloc(&cyc, blocks * 8 * sizeof(long long));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-62s kernel %8.1f us   cycles/iter wave0 %7.1f  wave%d %7.1f\n", name, ms * 1e3, (double)h[0] / iters, threads / 64 - 1,
         (double)h[threads / 64 - 1] / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  const int it = 4000;
  run<0>("1 wave/SIMD: 16 MFMA 32x32x2 (4 accumulators) per iter", 256, it);
  run<1>("1 wave/SIMD: 256 v_fma (8 chains) per iter", 256, it);
  run<2>("1 wave/SIMD: 16 MFMA + 256 v_fma, forced 1:16 interleave", 256, it);
  run<0>("2 waves/SIMD: all 8 waves 16 MFMA per iter", 512, it);
  run<1>("2 waves/SIMD: all 8 waves 256 v_fma per iter", 512, it);
  run<3>("2 waves/SIMD: waves 0-3 MFMA only, waves 4-7 v_fma only", 512, it);
  return 0;
}
