// Mel front end as a real FFT (reference audio/stft.py:59-88 STFT.transform, :166-185 TacotronSTFT.mel_spectrogram,
// audio/audio_processing.py:85-91 dynamic_range_compression) - ONE kernel: waveform -> (log-mel [B,80,F], energy [B,F]).
//
// The reference convolves with a [1026 x 1024] windowed DFT basis: 2.1 MFLOP per frame.  Here each frame is a 1024-point real FFT
// = one 512-point complex FFT (z[n] = x[2n] + i x[2n+1], three radix-8 Stockham passes, one wave per frame, 8 points per lane,
// exchange through LDS) + the even/odd split, 0.03 MFLOP per frame; hann window folded into the load, reflect padding as index
// arithmetic (no padded copy), |X| and the energy norm in registers, the mel filterbank as a banded GEMM on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32: 16 frames x 16 filters per tile, only the K range in which the 16 triangular filters are non-zero),
// log-clamp fused, output transposed through LDS so that both mel [B,n_mel,F] rows and the input reads are coalesced.
// HBM traffic = the algorithmic 1 KB in + 324 B out per frame (every sample is re-read by 4 overlapping frames from L1/L2).
#include "ctts_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NFFT = 1024, NC = 512, NBINS = 513;
constexpr int TILE_F = 16;                 // frames per workgroup tile (4 waves x 4 frames)
constexpr int MS_MAX = 517;                // magnitude-tile row stride when every bin may carry filter weight (odd: conflict-free column reads)
constexpr int SCR = 584;                   // per-wave FFT scratch: 512 complex + padding (pad32: index + index/32 + 8 * (index/64))
constexpr int LDS_HDR = 1540 + 112;        // twiddle tables in LDS: W512 | W1024 | the 7 x 8 twiddles of FFT pass 1, one row per r
// workspace layout (floats): [0, 1024) W512 (cos, sin) | [1024, 1024+516) W1024 k = 0..256 (cos, sin) | melT [516][96] | int32 kranges[12]
constexpr int WS_W512 = 0, WS_W1024 = 1024, WS_MELT = 1024 + 516, WS_KR = WS_MELT + 516 * 96, WS_FLOATS = WS_KR + 16;

struct float2_ { float x, y; };
__device__ __forceinline__ float2_ cmul(float2_ a, float2_ b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ void fft2(float2_& a, float2_& b) { const float2_ t = a; a = {t.x + b.x, t.y + b.y}; b = {t.x - b.x, t.y - b.y}; }
__device__ __forceinline__ float2_ mul_mi(float2_ a) { return {a.y, -a.x}; }       // * (-i)
__device__ __forceinline__ void fft4(float2_& v0, float2_& v1, float2_& v2, float2_& v3) {
  fft2(v0, v2); fft2(v1, v3); v3 = mul_mi(v3); fft2(v0, v1); fft2(v2, v3);         // -> X0 = v0, X2 = v1, X1 = v2, X3 = v3
}
// forward 8-point DFT in place; afterwards X[0..7] = v0, v4, v2, v6, v1, v5, v3, v7
__device__ __forceinline__ void fft8(float2_ (&v)[8]) {
  constexpr float h = 0.70710678118654752440f;
  fft2(v[0], v[4]); fft2(v[1], v[5]); fft2(v[2], v[6]); fft2(v[3], v[7]);
  v[5] = {(v[5].x + v[5].y) * h, (v[5].y - v[5].x) * h};        // * W8^1 = (1 - i)/sqrt2
  v[6] = mul_mi(v[6]);                                          // * W8^2 = -i
  v[7] = {(v[7].y - v[7].x) * h, -(v[7].x + v[7].y) * h};       // * W8^3 = (-1 - i)/sqrt2
  fft4(v[0], v[1], v[2], v[3]); fft4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ int brev3(int r) { return ((r & 1) << 2) | (r & 2) | (r >> 2); }
// scratch slot of complex element i.  Every exchange pattern of the three passes must spread a 16-lane group over all 64 banks
// (8-byte accesses): consecutive i (pass reads, pass-2 writes), i = 8 * lane + r (pass-0 writes: + i/32 shifts every 4th lane by a
// bank pair) and i = 64 * (lane / 8) + (lane & 7) + 8 r (pass-1 writes: + 8 * (i/64) moves the second 8-lane run 20 banks away from
// the first; with i + i/32 alone the two runs overlapped in 12 of 16 banks - 2-way conflicts on all 8 writes of every lane).
__device__ __forceinline__ int pad32(int i) { return i + (i >> 5) + ((i >> 6) << 3); }

__device__ __forceinline__ float sample_reflect(const float* __restrict__ yb, int N, int p) {
  // index into the reflect-padded waveform (F.pad(..., mode="reflect") by n_fft/2 on both sides, stft.py:66-71)
  int q = p - NFFT / 2;
  if (q < 0) q = -q;
  if (q >= N) q = 2 * (N - 1) - q;
  q = min(max(q, 0), N - 1);
  return yb[q];
}

// The FFT scratch S is private to a wave: its 64 lanes exchange data through LDS in program order (LDS operations of one wave complete in
// order), so the passes need no workgroup barrier - only a fence that keeps hipcc from moving LDS accesses across the exchange points.
// (Round 2 had __syncthreads() there: five workgroup barriers per frame that made the four independent FFTs of a workgroup march in step.)
#define CTTS_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier()

__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ y, const int32_t* __restrict__ lens, const float* __restrict__ window,
                                                   const float* __restrict__ ws,
                                                   float* __restrict__ mel, float* __restrict__ energy, float* __restrict__ mag_out,
                                                   long ld_mag, int B, int N, int F, int hop, int n_mel, float clip, int MS,
                                                   unsigned* __restrict__ range_flag) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // largest |sample| this thread loaded (v_max3_f32 with |.| modifiers: one instruction per pair; fmax drops NaNs, which instead turn the
  // frame's energy into NaN and are caught there): raises *range_flag when the waveform leaves [-1, 1] or holds a NaN (the reference's asserts)
  float amax = 0.f;
  bool saw_nan = false;
  float* tw512 = lds;                                   // 1024
  float* tw1024 = lds + 1024;                           // 516
  float* tw1 = lds + 1540;                              // 56 complex: W512^(8 * jm * r) at [(r - 1) * 8 + jm]
  float* magt = lds + LDS_HDR;                          // TILE_F x MS
  float* scr = magt + TILE_F * MS;                      // 4 waves x SCR complex (re | im interleaved)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 1540; e += 256) lds[e] = ws[e];
  // pass 1 needs only 8 distinct twiddles per r (jm = lane & 7); read from the W512 table they sit 64 r bytes apart - up to 8 lanes of
  // a group on one bank.  A row of 8 consecutive entries per r is conflict-free.
  if (tid < 56) { const int t = (tid & 7) * 8 * ((tid >> 3) + 1); tw1[2 * tid] = ws[2 * t]; tw1[2 * tid + 1] = ws[2 * t + 1]; }
  // only bins < MS-1 are kept in the tile (the filterbank is zero above fmax); columns >= 513 are K padding of the MFMA and stay zero
  for (int e = tid; e < TILE_F * 4; e += 256) { const int c = NBINS + (e & 3); if (c < MS) magt[(e >> 2) * MS + c] = 0.f; }
  const int* kr = reinterpret_cast<const int*>(ws + WS_KR);
  const float* melT = ws + WS_MELT;
  float win[16];
#pragma unroll
  for (int r = 0; r < 8; ++r) { win[2 * r] = window[2 * (lane + 64 * r)]; win[2 * r + 1] = window[2 * (lane + 64 * r) + 1]; }
  float* S = scr + wave * (2 * SCR);
  const bool keep_low = MS > 256;                        // every bin below 256 has a column in the magnitude tile (always, for real filterbanks)
  const int tiles_per_b = (F + TILE_F - 1) / TILE_F;
  const long n_tiles = (long)B * tiles_per_b;
  __syncthreads();
  int it = 0;
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, ++it) {
    const int b = (int)(tile / tiles_per_b), f0 = (int)(tile - (long)b * tiles_per_b) * TILE_F;
    const float* yb = y + (long)b * N;
    // ragged batches (preprocessing): utterance b has lens[b] samples; reflection happens at ITS end and it has Fb = 1 + lens[b]/hop
    // frames - later frames of the padded batch row repeat its last frame (the host slices them off)
    const int Nb = lens ? min(max(lens[b], NFFT / 2 + 1), N) : N;
    const int Fb = min(F, 1 + Nb / hop);
    // ------------------------------------------------------------------ FFT phase: wave w transforms frames f0 + 4w .. f0 + 4w + 3
    for (int ff = 0; ff < 4; ++ff) {
      const int fl = wave * 4 + ff, f = f0 + fl, fc = min(f, Fb - 1);
      float2_ v[8];
      const int base = fc * hop;
      // no reflection needed and the 8-byte pair loads are aligned: a property of the FRAME, so the two paths are a scalar branch (the
      // select-per-sample form cost ~100 VALU instructions per frame)
      const bool interior = base >= NFFT / 2 && base + NFFT - NFFT / 2 <= Nb && ((((long)b * N + base) | (reinterpret_cast<uintptr_t>(y) >> 2)) & 1) == 0;
      if (__builtin_amdgcn_readfirstlane((int)interior)) {
        const float2* p2 = reinterpret_cast<const float2*>(yb + (base - NFFT / 2)) + lane;      // 512-byte steps: immediate offsets
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float2 t = p2[64 * r];
          amax = fmaxf(amax, fmaxf(fabsf(t.x), fabsf(t.y)));
          v[r] = {t.x * win[2 * r], t.y * win[2 * r + 1]};
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int n = 2 * (lane + 64 * r);
          const float a = sample_reflect(yb, Nb, base + n), c = sample_reflect(yb, Nb, base + n + 1);
          amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(c)));
          v[r] = {a * win[2 * r], c * win[2 * r + 1]};
        }
      }
      // pass 0 (Ns = 1): no twiddles
      fft8(v);
#pragma unroll
      for (int r = 0; r < 8; ++r) { const int o = pad32(lane * 8 + r); S[2 * o] = v[brev3(r)].x; S[2 * o + 1] = v[brev3(r)].y; }
      CTTS_WAVE_SYNC();
      // pass 1 (Ns = 8) and pass 2 (Ns = 64)
#pragma unroll
      for (int pass = 1; pass < 3; ++pass) {
        const int Ns = pass == 1 ? 8 : 64;
        const int jm = lane & (Ns - 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int o = pad32(lane + 64 * r); v[r] = {S[2 * o], S[2 * o + 1]}; }
#pragma unroll
        for (int r = 1; r < 8; ++r) {                     // W512^(jm * r * 64 / Ns)
          const float* tw = pass == 1 ? tw1 + 2 * ((r - 1) * 8 + jm) : tw512 + 2 * (jm * r);
          v[r] = cmul(v[r], float2_{tw[0], tw[1]});
        }
        fft8(v);
        CTTS_WAVE_SYNC();                                 // every lane of THIS wave has read its inputs
        const int idx = (lane / Ns) * Ns * 8 + jm;
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int o = pad32(idx + r * Ns); S[2 * o] = v[brev3(r)].x; S[2 * o + 1] = v[brev3(r)].y; }
        CTTS_WAVE_SYNC();
      }
      // even / odd split: X[k] = E + W1024^k O, X[512-k] = conj(E - W1024^k O); magnitudes, energy
      float e2 = 0.f;
      float* mrow = magt + fl * MS;
      float* gmag = (mag_out && f < F) ? mag_out + ((long)b * F + f) * ld_mag : nullptr;
      // bins k = lane + 64 m and their mirrors 512 - k.  k < 256 < MS is always kept; a mirror at or above MS goes to an unused slot of
      // the wave's scratch (pad32 never produces slot 32) instead of under an exec mask: no saveexec / branch pairs in this loop
      float* dump = S + 64;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k = lane + 64 * m, km = NC - k;
        const int o1 = pad32(k), o2 = pad32(km & 511);
        const float a = S[2 * o1], bb = S[2 * o1 + 1], c = S[2 * o2], dd = S[2 * o2 + 1];
        const float er = 0.5f * (a + c), ei = 0.5f * (bb - dd), orr = 0.5f * (bb + dd), oi = -0.5f * (a - c);
        const float wr = tw1024[2 * k], wi = tw1024[2 * k + 1];
        const float tr = wr * orr - wi * oi, ti = wr * oi + wi * orr;
        // v_sqrt_f32 (1 ulp): the correctly rounded sqrtf expansion is ~15 VALU instructions, eight times per lane and frame
        const float m1 = __builtin_amdgcn_sqrtf((er + tr) * (er + tr) + (ei + ti) * (ei + ti));
        const float m2 = __builtin_amdgcn_sqrtf((er - tr) * (er - tr) + (ei - ti) * (ei - ti));
        if (keep_low) mrow[k] = m1; else if (k < MS) mrow[k] = m1;
        *(km < MS ? mrow + km : dump) = m2;
        e2 += m1 * m1 + m2 * m2;
        if (gmag) { gmag[k] = m1; gmag[km] = m2; }
      }
      {   // bin 256 pairs with itself: E = Re Z[256], O = Im Z[256], W1024^256 = -i  ->  |X[256]| = |Z[256]| (every lane reads the same slot)
        const int o = pad32(256);
        const float a = S[2 * o], bb = S[2 * o + 1];
        const float m1 = __builtin_amdgcn_sqrtf(a * a + bb * bb);
        if (lane == 0) { if (256 < MS) mrow[256] = m1; if (gmag) gmag[256] = m1; e2 += m1 * m1; }
      }
      e2 = ctts_wave_sum(e2);
      saw_nan |= !(e2 == e2);
      if (lane == 0 && f < F) energy[(long)b * F + f] = sqrtf(e2);
      CTTS_WAVE_SYNC();                                   // this wave's scratch is reused by its next frame
    }
    __syncthreads();                                      // the magnitude tile is complete (the mel phase reads every wave's rows)
    // ------------------------------------------------------------------ mel phase: [16 frames x K] x [K x 16 filters] per MFMA tile
    const int n_tiles16 = (n_mel + 15) / 16;
    float* T = scr;                                       // [n_mel_pad][17] staging for the transposed store
    // The 16-filter tiles differ 7x in K range (7 ... 46 steps of 4 bins for the 80-filter / 8 kHz bank): a fixed wave -> tile map
    // loads one SIMD of the CU with half of the phase while the others idle, for every co-resident workgroup alike; the map ROTATES with
    // the workgroup's tile counter instead, so that over four tiles every SIMD gets every share.
    const int role = (wave + it) & 3;
    for (int nt = role; nt < n_tiles16; nt += 4) {
      const int klo = kr[2 * nt], khi = kr[2 * nt + 1];  // multiples of 4
      floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
      const int fr = lane & 15, kq = lane >> 4;
      // A[frame][k] from the magnitude tile (LDS), B[k][filter] from the transposed filterbank (global, L2-resident; fr doubles as the
      // filter column).  One load per MFMA in program order made every step wait a full L2 round trip (~300 cycles for a 32-cycle MFMA):
      // groups of MU steps with all their loads in flight and two independent accumulation chains; the last, partial group is predicated
      // to weight 0 on a clamped (finite) magnitude.
      constexpr int MU = 8;
      const float* ap = magt + fr * MS + kq;
      const float* bp = melT + (long)kq * 96 + nt * 16 + fr;
      int k0 = klo;
      for (; k0 + 4 * MU <= khi; k0 += 4 * MU) {
        float bv[MU], av[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) bv[u] = bp[(long)(k0 + 4 * u) * 96];
#pragma unroll
        for (int u = 0; u < MU; ++u) av[u] = ap[k0 + 4 * u];
#pragma unroll
        for (int u = 0; u < MU; u += 2) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], acc2, 0, 0, 0);
        }
      }
      if (k0 < khi) {
        float bv[MU], av[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) { const int kk = min(k0 + 4 * u, khi - 4); bv[u] = k0 + 4 * u < khi ? bp[(long)kk * 96] : 0.f; av[u] = ap[kk]; }
#pragma unroll
        for (int u = 0; u < MU; u += 2) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], acc2, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
      // D: col = lane & 15 (filter), row = (lane >> 4) * 4 + reg (frame)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(nt * 16 + (lane & 15)) * 17 + (lane >> 4) * 4 + r] = logf(fmaxf(acc[r], clip));
    }
    __syncthreads();
    for (int e = tid; e < n_mel * TILE_F; e += 256) {
      const int n = e / TILE_F, fl = e - n * TILE_F;
      if (f0 + fl < F) mel[((long)b * n_mel + n) * F + f0 + fl] = T[n * 17 + fl];
    }
    __syncthreads();
  }
  // out-of-range or NaN input.  No traffic at all for valid input; the flag may live in pinned host memory (plain store: any offending
  // value will do, the host only tests for non-zero)
  if (range_flag && (amax > 1.0f || saw_nan)) *reinterpret_cast<volatile unsigned*>(range_flag) = saw_nan ? 0x7FC00000u : __float_as_uint(amax);
}

// workspace set-up (once per filterbank): twiddles in double precision, the transposed / zero-padded mel basis and, per tile of 16
// filters, the range of DFT bins in which any of them is non-zero
__global__ void mel_prepare_kernel(const float* __restrict__ mel_basis, int n_mel, int nbins, float* __restrict__ ws) {
  const int tid = threadIdx.x;
  for (int m = tid; m < 512; m += blockDim.x) {
    double s, c; sincos(-2.0 * 3.14159265358979323846 * m / 512.0, &s, &c);
    ws[WS_W512 + 2 * m] = (float)c; ws[WS_W512 + 2 * m + 1] = (float)s;
  }
  for (int m = tid; m < 258; m += blockDim.x) {
    double s, c; sincos(-2.0 * 3.14159265358979323846 * m / 1024.0, &s, &c);
    ws[WS_W1024 + 2 * m] = m <= 256 ? (float)c : 0.f; ws[WS_W1024 + 2 * m + 1] = m <= 256 ? (float)s : 0.f;
  }
  for (int e = tid; e < 516 * 96; e += blockDim.x) {
    const int k = e / 96, n = e - k * 96;
    ws[WS_MELT + e] = (k < nbins && n < n_mel) ? mel_basis[(long)n * nbins + k] : 0.f;
  }
  int* kr = reinterpret_cast<int*>(ws + WS_KR);
  if (tid < 6) {
    int lo = 516, hi = 0;
    for (int n = tid * 16; n < min(tid * 16 + 16, n_mel); ++n)
      for (int k = 0; k < nbins; ++k)
        if (mel_basis[(long)n * nbins + k] != 0.f) { lo = min(lo, k); hi = max(hi, k + 1); }
    if (hi <= lo) { lo = 0; hi = 0; }
    kr[2 * tid] = lo & ~3; kr[2 * tid + 1] = min((hi + 3) & ~3, 516);
  }
}

}  // namespace

extern "C" size_t ctts_mel_spectrogram_workspace_bytes(int n_fft, int n_mel) {
  (void)n_fft; (void)n_mel;
  return sizeof(float) * (size_t)WS_FLOATS;
}

extern "C" int ctts_mel_prepare(const float* mel_basis, int n_fft, int n_mel, float* workspace, void* stream) {
  CTTS_REQUIRE(mel_basis && workspace, "ctts_mel_prepare: null pointer");
  CTTS_REQUIRE(n_fft == NFFT && n_mel >= 1 && n_mel <= 96, "ctts_mel_prepare: built for n_fft = 1024 (reference filter_length) and n_mel <= 96; got %d / %d", n_fft, n_mel);
  hipLaunchKernelGGL(mel_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mel_basis, n_mel, NBINS, workspace);
  CTTS_CHECK_LAUNCH("ctts_mel_prepare");
  return 0;
}

extern "C" int ctts_mel_spectrogram(const float* y, const int32_t* lens, const float* window, const float* workspace, float* mel, float* energy,
                                    float* mag, int64_t ld_mag, int B, int N, int n_fft, int hop, int n_mel, float clip, int kmax, uint32_t* range_flag,
                                    void* stream) {
  CTTS_REQUIRE(y && window && workspace && mel && energy && B > 0 && N > 0, "ctts_mel_spectrogram: bad arguments");
  CTTS_REQUIRE(n_fft == NFFT && hop > 0 && n_mel >= 1 && n_mel <= 96, "ctts_mel_spectrogram: built for n_fft = 1024, n_mel <= 96");
  CTTS_REQUIRE(N > NFFT / 2, "ctts_mel_spectrogram: reflect padding needs more than n_fft/2 samples (got %d)", N);
  CTTS_REQUIRE(!mag || ld_mag >= NBINS, "ctts_mel_spectrogram: ld_mag must be >= 513");
  const int F = 1 + N / hop;
  const long tiles = (long)B * ((F + TILE_F - 1) / TILE_F);
  // kmax = 1 + the highest DFT bin with a non-zero filter weight (0 = unknown: keep all 513).  The magnitude tile holds bins < kmax only
  // (fmax 8 kHz of 11 kHz: 372 bins -> 48 KB of LDS per workgroup, 3 workgroups per CU instead of 2)
  CTTS_REQUIRE(kmax >= 0 && kmax <= NBINS, "ctts_mel_spectrogram: kmax out of range");
  const int MS = kmax > 0 ? (((kmax + 3) & ~3) | 1) : MS_MAX;
  const size_t lds_bytes = sizeof(float) * (size_t)(LDS_HDR + TILE_F * MS + 4 * 2 * SCR);
  // raise the kernel's dynamic-LDS limit to the LARGEST footprint (kmax = 0) on every call: the attribute is per device and per
  // process-wide function handle, a one-shot static flag would pin the first call's (possibly smaller) size and the first device
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * (size_t)(LDS_HDR + TILE_F * MS_MAX + 4 * 2 * SCR)));
  // persistent grid: the workgroups that fit the chip at once walk the tiles (twiddles / window / filter ranges are loaded once per
  // workgroup); env CTTS_MEL_GRID overrides (tools)
  static int resident = 0;
  if (!resident) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(mel_kernel), 256, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
    const char* ge = getenv("CTTS_MEL_GRID");
    resident = ge ? atoi(ge) : per_cu * prop.multiProcessorCount;
    if (resident < 1) resident = 768;
  }
  const int grid = (int)(tiles < resident ? tiles : resident);
  hipLaunchKernelGGL(mel_kernel, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, y, lens, window, workspace, mel, energy, mag, (long)ld_mag,
                     B, N, F, hop, n_mel, clip, MS, range_flag);
  CTTS_CHECK_LAUNCH("ctts_mel_spectrogram");
  return 0;
}
