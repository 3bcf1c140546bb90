// Micro-benchmark: an fp32 GEMM computed on the BF16 matrix pipe by splitting every fp32 operand into bf16 pieces in registers
// (x = hi + mid + lo, 8 significand bits each; THIS micro-benchmark truncates, the library kernels round to nearest - same instruction
// count, dropped terms <= 2^-24 instead of 2^-21 of a product: tests/test_bf16_split_cpu.py) and accumulating the cross products in fp32:
//   TERMS = 6: hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi   (dropped terms <= 2^-21 of a product worst case, < 2^-24 typically: fp32-class results)
//   TERMS = 3: hi*hi + hi*mid + mid*hi                              (2^-16 relative per product)
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32 on gfx950, so 6 of them per fp32-equivalent product are a
// 2.67x higher ceiling than the fp32 MFMA (157 TFLOP/s) - IF the operand split (VALU) and the fragment reads (LDS) hide under the MFMAs.
// C[M,N] = A[M,K] * B[N,K]^T, fp32 in HBM / LDS / out.  Workgroup 256 threads = 4 waves, tile 128 x 128, wave tile 64 x 64, K block 32,
// double-buffered LDS through registers.  Prints TFLOP/s (fp32-equivalent: 2 M N K) and the error against float64 next to the error of
// a plain fp32 FMA chain on the same data.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x_split_gemm bf16x_split_gemm.hip && ./bf16x_split_gemm [M=16384] [N=1024] [K=2304]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, LDW = 36;     // LDS row stride 36 floats: 16-byte fragment reads of 8 consecutive rows hit 8 distinct bank quads

struct Frag { uintx4 p[3]; };       // 8 consecutive-K values of one row as packed bf16: hi, mid, lo

template <int TERMS>
__device__ __forceinline__ Frag split8(const float4 x0, const float4 x1) {
  const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  float r1[8], r2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float h = __uint_as_float(__float_as_uint(x[i]) & 0xFFFF0000u);
    r1[i] = x[i] - h;                                     // exact
    if (TERMS == 6) { const float m = __uint_as_float(__float_as_uint(r1[i]) & 0xFFFF0000u); r2[i] = r1[i] - m; }
  }
  Frag f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {                           // upper halves of two floats -> one register (truncation = the masks above)
    f.p[0][j] = __builtin_amdgcn_perm(__float_as_uint(x[2 * j + 1]), __float_as_uint(x[2 * j]), 0x07060302u);
    f.p[1][j] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * j + 1]), __float_as_uint(r1[2 * j]), 0x07060302u);
    f.p[2][j] = TERMS == 6 ? __builtin_amdgcn_perm(__float_as_uint(r2[2 * j + 1]), __float_as_uint(r2[2 * j]), 0x07060302u) : 0u;
  }
  return f;
}

__device__ __forceinline__ floatx16 mma(const uintx4 a, const uintx4 b, const floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int TERMS>
__global__ __launch_bounds__(256) void gemm_split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][A 128 x LDW | B 128 x LDW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = N / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  floatx16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ga[4], gb[4];
  const int grow = tid >> 3, gkc = (tid & 7) * 4;          // float4 q = tid + 256 i: row = grow + 32 i, same K offset
  const float* gpa = A + (long)(m0 + grow) * K + gkc;
  const float* gpb = B + (long)(n0 + grow) * K + gkc;
  const int lofs = grow * LDW + gkc;
#define GLOAD(k0) _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
    ga[i] = *reinterpret_cast<const float4*>(gpa + (long)(32 * i) * K + (k0)); gb[i] = *reinterpret_cast<const float4*>(gpb + (long)(32 * i) * K + (k0)); }
#define LSTORE(buf) { float* la_ = lds + (buf) * (2 * 128 * LDW) + lofs; float* lb_ = la_ + 128 * LDW; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { *reinterpret_cast<float4*>(la_ + 32 * i * LDW) = ga[i]; *reinterpret_cast<float4*>(lb_ + 32 * i * LDW) = gb[i]; } }
  GLOAD(0)
  LSTORE(0)
  __syncthreads();
  const int nkb = K / BK;
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  for (int kb = 0; kb < nkb; ++kb) {
    const int buf = kb & 1;
    { const int kn = (kb + 1 < nkb ? kb + 1 : kb) * BK; GLOAD(kn) }      // unconditional (the last block is re-read): keeps the staging registers out of scratch
    const float* la = lds + buf * (2 * 128 * LDW);
    const float* lb = la + 128 * LDW;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float* pa = la + (wm + 32 * i + fr) * LDW + ks * 16 + fk;
        const float* pb = lb + (wn + 32 * i + fr) * LDW + ks * 16 + fk;
        fa[i] = split8<TERMS>(*reinterpret_cast<const float4*>(pa), *reinterpret_cast<const float4*>(pa + 4));
        fb[i] = split8<TERMS>(*reinterpret_cast<const float4*>(pb), *reinterpret_cast<const float4*>(pb + 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          floatx16 c = acc[i][j];
          if (TERMS == 6) { c = mma(fa[i].p[2], fb[j].p[0], c); c = mma(fa[i].p[0], fb[j].p[2], c); c = mma(fa[i].p[1], fb[j].p[1], c); }
          c = mma(fa[i].p[1], fb[j].p[0], c);
          c = mma(fa[i].p[0], fb[j].p[1], c);
          c = mma(fa[i].p[0], fb[j].p[0], c);
          acc[i][j] = c;
        }
    }
    LSTORE(buf ^ 1)
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3), col = n0 + wn + 32 * j + (lane & 31);
        C[(long)row * N + col] = acc[i][j][r];
      }
}

// ---- variant 2: every element is split ONCE, on its way from the staging registers into LDS (three bf16 planes per operand tile); the
// main loop then only reads 16-byte bf16 fragments and issues MFMAs.  LDS: plane row = 32 bf16 (64 B) + 16 B pad = 80 B -> the 16-byte
// reads of 8 consecutive rows hit 8 distinct bank quads.  2 stages x 2 operands x 3 planes x 128 rows x 80 B = 120 KB: one workgroup per CU.
constexpr int PROW = 80;                                   // bytes per plane row
constexpr int PLANE = 128 * PROW;                           // bytes per plane
constexpr int STAGE = 6 * PLANE;                            // A hi|mid|lo, B hi|mid|lo

template <int TERMS>
__device__ __forceinline__ void split4_store(const float4 v, char* base) {      // 4 consecutive-K floats -> 8 bytes in each plane
  const float x[4] = {v.x, v.y, v.z, v.w};
  float r1[4], r2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float h = __uint_as_float(__float_as_uint(x[i]) & 0xFFFF0000u);
    r1[i] = x[i] - h;
    const float m = __uint_as_float(__float_as_uint(r1[i]) & 0xFFFF0000u);
    r2[i] = r1[i] - m;
  }
  uint2 h2, m2, l2;
  h2.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
  h2.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
  m2.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
  m2.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
  *reinterpret_cast<uint2*>(base) = h2;
  *reinterpret_cast<uint2*>(base + PLANE) = m2;
  if (TERMS == 6) {
    l2.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
    l2.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
    *reinterpret_cast<uint2*>(base + 2 * PLANE) = l2;
  }
}

template <int TERMS, int STAGES>
__global__ __launch_bounds__(256) void gemm_split_lds(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = N / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  floatx16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ga[4], gb[4];
  const int grow = tid >> 3, gkc = (tid & 7) * 4;
  const float* gpa = A + (long)(m0 + grow) * K + gkc;
  const float* gpb = B + (long)(n0 + grow) * K + gkc;
  const int sofs = grow * PROW + gkc * 2;                  // byte offset of this thread's 4 values inside a plane
#define SSTORE(buf) { char* sa_ = smem + (buf) * STAGE + sofs; char* sb_ = sa_ + 3 * PLANE; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { split4_store<TERMS>(ga[i], sa_ + 32 * i * PROW); split4_store<TERMS>(gb[i], sb_ + 32 * i * PROW); } }
  GLOAD(0)
  SSTORE(0)
  __syncthreads();
  const int nkb = K / BK;
  const int fofs = (lane & 31) * PROW + (lane >> 5) * 16;   // fragment: row = lane & 31, 8 consecutive K from (lane >> 5) * 8
  for (int kb = 0; kb < nkb; ++kb) {
    const int buf = STAGES == 2 ? (kb & 1) : 0;
    { const int kn = (kb + 1 < nkb ? kb + 1 : kb) * BK; GLOAD(kn) }
    const char* sa = smem + buf * STAGE + fofs;
    const char* sb = sa + 3 * PLANE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uintx4 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < (TERMS == 6 ? 3 : 2); ++p) {
          fa[i][p] = *reinterpret_cast<const uintx4*>(sa + p * PLANE + (wm + 32 * i) * PROW + ks * 32);
          fb[i][p] = *reinterpret_cast<const uintx4*>(sb + p * PLANE + (wn + 32 * i) * PROW + ks * 32);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          floatx16 c = acc[i][j];
          if (TERMS == 6) { c = mma(fa[i][2], fb[j][0], c); c = mma(fa[i][0], fb[j][2], c); c = mma(fa[i][1], fb[j][1], c); }
          c = mma(fa[i][1], fb[j][0], c);
          c = mma(fa[i][0], fb[j][1], c);
          c = mma(fa[i][0], fb[j][0], c);
          acc[i][j] = c;
        }
    }
    if (STAGES == 1) __syncthreads();                      // one stage: everybody has read the tile before it is overwritten
    SSTORE(STAGES == 2 ? (buf ^ 1) : 0)
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3), col = n0 + wn + 32 * j + (lane & 31);
        C[(long)row * N + col] = acc[i][j][r];
      }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 2304;
  if (M % BM || N % BN || K % BK) { printf("M, N multiples of 128, K of 32\n"); return 1; }
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  unsigned s = 12345u;
  auto rnd = [&]() { float u = 0; for (int i = 0; i < 12; ++i) { s = s * 1664525u + 1013904223u; u += (s >> 8) * (1.0f / 16777216.0f); } return u - 6.0f; };
  for (auto& v : ha) v = rnd();
  for (auto& v : hb) v = rnd() * 0.05f;
  float *A, *B, *C;
  hipMalloc(&A, ha.size() * 4); hipMalloc(&B, hb.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  const size_t ldsb = 2 * 2 * 128 * LDW * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  const int grid = (M / BM) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> hc((size_t)M * N);
  const size_t ldsb2 = 2 * STAGE;
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_lds<6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_lds<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_lds<6, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_lds<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb2);
  for (int variant : {1, 2, 3}) for (int terms : {6, 3}) {
    auto launch = [&]() {
      if (variant == 1) { if (terms == 6) hipLaunchKernelGGL(gemm_split<6>, dim3(grid), dim3(256), ldsb, 0, A, B, C, M, N, K);
                          else hipLaunchKernelGGL(gemm_split<3>, dim3(grid), dim3(256), ldsb, 0, A, B, C, M, N, K); }
      else if (variant == 2) { if (terms == 6) hipLaunchKernelGGL((gemm_split_lds<6, 2>), dim3(grid), dim3(256), ldsb2, 0, A, B, C, M, N, K);
             else hipLaunchKernelGGL((gemm_split_lds<3, 2>), dim3(grid), dim3(256), ldsb2, 0, A, B, C, M, N, K); }
      else { if (terms == 6) hipLaunchKernelGGL((gemm_split_lds<6, 1>), dim3(grid), dim3(256), ldsb2 / 2, 0, A, B, C, M, N, K);
             else hipLaunchKernelGGL((gemm_split_lds<3, 1>), dim3(grid), dim3(256), ldsb2 / 2, 0, A, B, C, M, N, K); } };
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 30; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 30;
    hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
    // error on 4096 sampled outputs: against float64, next to a plain fp32 FMA chain (what v_mfma_f32_32x32x2_f32 computes, in some order)
    double emax = 0, e32max = 0, cmax = 0;
    unsigned t = 777u;
    for (int q = 0; q < 4096; ++q) {
      t = t * 1664525u + 1013904223u; const int m = (t >> 8) % M; t = t * 1664525u + 1013904223u; const int n = (t >> 8) % N;
      double ref = 0; float f32 = 0.f;
      for (int k = 0; k < K; ++k) { ref += (double)ha[(size_t)m * K + k] * hb[(size_t)n * K + k]; f32 = fmaf(ha[(size_t)m * K + k], hb[(size_t)n * K + k], f32); }
      emax = fmax(emax, fabs(hc[(size_t)m * N + n] - ref)); e32max = fmax(e32max, fabs((double)f32 - ref)); cmax = fmax(cmax, fabs(ref));
    }
    printf("%s bf16x%d: %8.1f us  %7.1f TFLOP/s (fp32-equivalent)   max |err| vs fp64 %.3e   (plain fp32 FMA chain: %.3e;  max |C| %.2f)\n",
           variant == 1 ? "split in the main loop (fp32 in LDS, 2 WG/CU)" : (variant == 2 ? "split while staging (bf16 planes, 2 stages, 1 WG/CU)" : "split while staging (bf16 planes, 1 stage, 2 WG/CU)"), terms, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, emax, e32max, cmax);
  }
  return 0;
}
