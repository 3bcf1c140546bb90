// Micro-benchmark: fp32 MFMA rate as a function of the OPERAND REGISTER PATTERN and of the number of waves per SIMD (gfx950).
// mfma_sustained.hip showed 154 TFLOP/s with one wave per SIMD but only ~123 with 2-8 waves per SIMD as soon as every MFMA reads its A and B
// operands from a different register (as a GEMM's fragments do) - independent of the data values and at an unchanged 2.39 GHz shader clock.
// Patterns (16 MFMAs per iteration, 4 accumulators in rotation unless stated):
//   0  A = fresh VALU result, B = one register             (the "ideal" stream)
//   1  A[q], B[q]: 32 different registers                  (64x64 tile kernel: one MFMA per fragment pair)
//   2  A[q], B fixed                                        3  A fixed, B[q]
//   4  2x2 register tile: acc[i][j] += A[i][k] B[j][k], k = 0..3 (each operand register feeds two consecutive-ish MFMAs)
//   5  as 1, operands copied into fresh temporaries by v_mov right before each MFMA
//   6  as 1, but ONE accumulator (dependent chain, what gemm_sk<1,1> issues)
//   usage: mfma_patterns [seconds_per_case=0.5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int P>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* rnd) {
  floatx16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float ra[16], rb[16];
  for (int q = 0; q < 16; ++q) { ra[q] = rnd[(threadIdx.x * 16 + q) & 4095]; rb[q] = rnd[(threadIdx.x * 16 + q + 2048 + blockIdx.x) & 4095]; }
  const float a0 = ra[0], b0 = rb[0];
  for (int it = 0; it < iters; ++it) {
    if (P == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = MFMA(a0 + q, b0, acc[q & 3]);
    } else if (P == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = MFMA(ra[q], rb[q], acc[q & 3]);
    } else if (P == 2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = MFMA(ra[q], b0, acc[q & 3]);
    } else if (P == 3) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = MFMA(a0, rb[q], acc[q & 3]);
    } else if (P == 4) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = MFMA(ra[i * 4 + kk], rb[j * 4 + kk], acc[i * 2 + j]);
    } else if (P == 5) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float ta, tb;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(ta), "=v"(tb) : "v"(ra[q]), "v"(rb[q]));
        acc[q & 3] = MFMA(ta, tb, acc[q & 3]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[0] = MFMA(ra[q], rb[q], acc[0]);
    }
  }
  float s = 0.f;
  for (int a2 = 0; a2 < 4; ++a2) for (int r = 0; r < 16; ++r) s += acc[a2][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int P>
double run(int grid, int iters, float* out, const float* rnd, double seconds) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)grid * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
  std::vector<double> tf; double el = 0;
  while (el < seconds) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<P>, dim3(grid), dim3(256), 0, 0, out, iters, rnd); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); tf.push_back(flop / (ms * 1e-3) / 1e12); el += ms * 1e-3;
  }
  double s = 0; size_t n = tf.size(), a = n / 2; for (size_t i = a; i < n; ++i) s += tf[i];
  return s / (n - a);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 0.5;
  float* out; hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
  float* rnd; hipMalloc(&rnd, 4096 * 4);
  { std::vector<float> hr(4096); unsigned s = 12345u; for (auto& v : hr) { float u = 0; for (int i = 0; i < 12; ++i) { s = s * 1664525u + 1013904223u; u += (s >> 8) * (1.0f / 16777216.0f); } v = u - 6.0f; }
    hipMemcpy(rnd, hr.data(), 4096 * 4, hipMemcpyHostToDevice); }
  printf("pattern | TFLOP/s at 1, 2, 4, 8 waves per SIMD (peak 157.3 at 2.4 GHz)\n");
  const int wps[4] = {1, 2, 4, 8};
#define ROW(P) { printf("   %d    |", P); for (int w = 0; w < 4; ++w) printf(" %7.1f", run<P>(256 * wps[w], 2048 / wps[w], out, rnd, seconds)); printf("\n"); }
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6)
  return 0;
}
