// Pieces shared by the GEMM translation units (gemm.hip: tile-per-workgroup kernels, gemm_sk.hip: persistent stream-K kernel).
#pragma once
#include "ctts_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

// ---- epilogue shared by the kernels.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int MT, int NT>
__device__ __forceinline__ void gemm_epilogue(const ctts_gemm_desc& d, floatx16 (&acc)[MT][NT], float* Cb, int z, int row0, int col0,
                                              int wm0, int wn0, int l31, int h, int Mv, int Nv) {
  const float alpha = d.alpha;
  if (d.split_k > 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int n = col0 + wn0 + j * 32 + l31;
          if (m < Mv && n < Nv) atomicAdd(Cb + (long)m * d.ldc + n, alpha * acc[i][j][r]);
        }
    return;
  }
  const bool do_drop = d.p_drop > 0.f;
  uint32_t dkey = 0;
  float inv_keep = 1.f;
  if (do_drop) { dkey = ctts_drop_key(d.seed, d.drop_offset); inv_keep = 1.f / (1.f - d.p_drop); }
  const uint32_t zoff = (uint32_t)z * (uint32_t)d.M * (uint32_t)d.N;
  if (d.E) {           // fused softmax backward: dS = P * (dP - D)
    const float* Eb = d.E + (Cb - d.C);
    const float* rs = d.rowsub + (long)z * d.M;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = col0 + wn0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < Mv && n < Nv) Cb[(long)m * d.ldc + n] = Eb[(long)m * d.ldc + n] * (alpha * acc[i][j][r] - rs[m]);
        }
      }
    return;
  }
  if (d.epi_bwd) {       // backward of the previous layer's epilogue: C = alpha * acc * drop_mask * act'(Z_prev)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = col0 + wn0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < Mv && n < Nv) {
            float v = alpha * acc[i][j][r];
            if (do_drop) v *= ctts_drop_scale(dkey, zoff + (uint32_t)m * (uint32_t)d.N + (uint32_t)n, d.p_drop, inv_keep);
            if (d.act) v *= ctts_act_grad(d.Z[(long)m * d.ldz + n], d.act);
            Cb[(long)m * d.ldc + n] = v;
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = col0 + wn0 + j * 32 + l31;
      const float bv = (d.bias && n < Nv) ? d.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < Mv && n < Nv) {
          float v = alpha * (acc[i][j][r] + bv);
          if (d.Z) d.Z[(long)m * d.ldz + n] = v;
          v = ctts_act(v, d.act);
          if (do_drop) v *= ctts_drop_scale(dkey, zoff + (uint32_t)m * (uint32_t)d.N + (uint32_t)n, d.p_drop, inv_keep);
          if (d.R) v += d.R[(long)m * d.ldr + n];
          if (d.rowscale) v *= d.rowscale[m];
          Cb[(long)m * d.ldc + n] = v;
        }
      }
    }
}

}  // namespace

// gemm_sk.hip: persistent stream-K kernel with direct-to-LDS operand loads.  Returns 1 when it took the launch, 0 when the descriptor is
// not eligible (the caller then uses the tile-per-workgroup kernels), < 0 on error.
int ctts_gemm_sk_try(const ctts_gemm_desc& d, hipStream_t st);

// gemm_ws.hip: weight-stationary kernel for K = 256 (the weight slice of a workgroup lives in registers, A tiles stream through LDS).
// Same return convention.
int ctts_gemm_ws_try(const ctts_gemm_desc& d, hipStream_t st);
