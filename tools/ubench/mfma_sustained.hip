// Micro-benchmark: what fp32-MFMA rate does an MI355X SUSTAIN?  A kernel that does nothing but v_mfma_f32_32x32x2_f32 (random, non-zero
// operands; W waves per SIMD, 4 independent accumulators each) is launched back to back for about two seconds; every launch is timed
// with HIP events, and each wave also reports its shader-cycle count (s_memrealtime = 100 MHz wall clock vs clock64 = shader clock).
// Prints the TFLOP/s of the first launches (boost clock) and of the steady state, and the effective shader clock of both.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_sustained mfma_sustained.hip && ./mfma_sustained [waves_per_simd=4] [seconds=2] [random_operands=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <bool RANDOM>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamp, int iters, const float* rnd) {
  floatx16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const float a = 0.37f + threadIdx.x * 1.1e-3f, b = -1.0003f + blockIdx.x * 1e-5f;
  float ra[16], rb[16];      // RANDOM: 16 different N(0,1) operands per lane, as a GEMM's fragments are
  for (int q = 0; q < 16; ++q) { ra[q] = rnd[(threadIdx.x * 16 + q) & 4095]; rb[q] = rnd[(threadIdx.x * 16 + q + 2048 + blockIdx.x) & 4095]; }
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      acc[q & 3] = RANDOM ? __builtin_amdgcn_mfma_f32_32x32x2f32(ra[q], rb[q], acc[q & 3], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x2f32(a + q, b, acc[q & 3], 0, 0, 0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int a2 = 0; a2 < 4; ++a2) for (int r = 0; r < 16; ++r) s += acc[a2][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { stamp[0] = c1 - c0; stamp[1] = w1 - w0; }
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 4;
  const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
  const int random = argc > 3 ? atoi(argv[3]) : 0;     // 1: N(0,1) operands in 32 registers, 2: the same registers all holding 1.0
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, grid = cus * wps, iters = 2048;
  float* out; unsigned long long* stamp;
  hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&stamp, 16);
  float* rnd; hipMalloc(&rnd, 4096 * 4);
  { std::vector<float> hr(4096); unsigned s = 12345u; for (auto& v : hr) { float u = 0; for (int i = 0; i < 12; ++i) { s = s * 1664525u + 1013904223u; u += (s >> 8) * (1.0f / 16777216.0f); } v = u - 6.0f; }
    if (random == 2) for (auto& v : hr) v = 1.0f;
    hipMemcpy(rnd, hr.data(), 4096 * 4, hipMemcpyHostToDevice); }
  const double flop = (double)grid * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
  std::vector<double> tf, clk;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double elapsed = 0;
  while (elapsed < seconds) {
    hipEventRecord(e0); if (random) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, out, stamp, iters, rnd); else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, out, stamp, iters, rnd); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost);
    tf.push_back(flop / (ms * 1e-3) / 1e12);
    clk.push_back((double)h[0] / ((double)h[1] / 100e6) / 1e9);     // shader cycles per second of the 100 MHz wall clock
    elapsed += ms * 1e-3;
  }
  auto avg = [](const std::vector<double>& v, size_t a, size_t b) { double s = 0; for (size_t i = a; i < b; ++i) s += v[i]; return s / (b - a); };
  const size_t n = tf.size(), tail = n > 40 ? n / 4 : 1;
  printf("CUs %d, %d waves/SIMD, %s operands, %zu launches of %.2f ms\n", cus, wps, random == 1 ? "random N(0,1)" : (random == 2 ? "32 registers of 1.0" : "near-constant"), n, flop / tf[n - 1] / 1e9);
  printf("first launch      : %7.1f TFLOP/s   clock64/wall = %.3f GHz\n", tf[0], clk[0]);
  printf("launches 2-5      : %7.1f TFLOP/s   %.3f GHz\n", avg(tf, 1, n > 5 ? 5 : n), avg(clk, 1, n > 5 ? 5 : n));
  printf("steady (last 25%%) : %7.1f TFLOP/s   %.3f GHz\n", avg(tf, n - tail, n), avg(clk, n - tail, n));
  printf("peak at 2.4 GHz   : %7.1f TFLOP/s (64 FLOP/clk/SIMD)\n", cus * 4 * 64 * 2.4e9 / 1e12);
  return 0;
}
