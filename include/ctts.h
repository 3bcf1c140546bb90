/* ctts.h - C ABI of libctts_hip.so: the MI355X (gfx950) kernels behind the CompTransTTS hot path.
 *
 * Conventions:
 *   - every pointer is a raw DEVICE pointer (torch `tensor.data_ptr()`), shapes/strides are
 *     explicit, `stream` is a hipStream_t passed as void* (PyTorch's current stream);
 *   - functions return 0 on success, <0 on error (ctts_last_error() gives the text);
 *     no C++ exception crosses the ABI; the library never allocates caller-visible memory,
 *     all calls are asynchronous and stream-ordered (graph-capturable);
 *   - layouts are channel-last / row-major: activations are [B, T, C] (C contiguous).
 *
 * Each entry point names the reference code it replaces (paths relative to the reference
 * repository keonlee9420/Comprehensive-Transformer-TTS).  The reference has no FFI of its
 * own (it is pure Python on torch ops); INTEGRATION.md shows the ctypes binding.
 */
#ifndef CTTS_H
#define CTTS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ctts_last_error(void);
int ctts_version(void);

/* ---------------------------------------------------------------------------------------
 * General fp32 MFMA GEMM with implicit-im2col ("conv") operand views and fused epilogue.
 *   C[m,n] = epi( alpha * (sum_k opA[m,k] * opB[k,n] + bias[n]) )
 *   epi(v) = rowscale[m] * (R[m,n] + drop(act(v)))   (Z, if given, receives v)
 * opA[m,k] = A[m*lda + k] (a_kc=1) or A[k*lda + m] (a_kc=0); opB[k,n] = B[n*ldb + k] (b_kc=1)
 * or B[k*ldb + n] (b_kc=0).  Replaces torch.nn.functional.linear / conv1d / bmm call sites:
 *   - Conv1d(k=9) FFN            model/transformers/transformer_fs2.py:220-239
 *   - in/out projections, QK^T, PV  (F.multi_head_attention_forward)  transformer_fs2.py:385-394
 *   - predictor Conv1d(k=3,5)    model/modules.py:1299-1356
 *   - mel_linear                 model/CompTransTTS.py:133
 *   - PostNet Conv1d(k=5)        model/modules.py:140-148
 *   - STFT-as-conv1d + mel matmul  audio/stft.py:72-80,177
 * Conv view: rows of the operand are (b,t) pairs with t = row % conv_T; element (row, kk)
 * reads X[(row - conv_pad)*ld + kk] and is zero unless 0 <= t - conv_pad + kk/conv_cin < conv_T.
 */
typedef struct ctts_gemm_desc {
  const float* A; const float* B; float* C;
  int32_t M, N, K;
  int64_t lda, ldb, ldc;
  int32_t a_kc, b_kc;
  int32_t nb0, nb1;                       /* batch = nb0*nb1 (blockIdx.z = z0*nb1+z1); 1,1 = none   */
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;   /* element strides per batch level                          */
  const int32_t* lens;                    /* [nb0] valid length per z0, or NULL                       */
  int32_t lim_m, lim_n, lim_k;            /* clamp that dim to lens[z0]                               */
  int32_t conv_T, conv_pad, conv_cin, conv_on_b; /* conv_T=0: plain; conv_on_b=1 applies to B (b_kc=0) */
  int32_t split_k;                        /* >1: C += alpha * A B, K cut into <= split_k pieces summed in a FIXED order through sk_ws (no atomics, no epilogue) */
  float alpha;
  const float* bias;                      /* [N] or NULL                                              */
  float* Z; int64_t ldz;                  /* optional store of the pre-activation                     */
  int32_t act;                            /* 0 none, 1 relu, 2 gelu(erf), 3 tanh, 4 swish             */
  float p_drop; const uint64_t* seed; uint32_t drop_offset;   /* inverted dropout after act           */
  const float* R; int64_t ldr;            /* residual added after dropout                             */
  const float* rowscale;                  /* [M] multiplied last (non-pad mask), or NULL              */
  /* Padded-row skipping.  Rows (of C for a_kc=1, of the reduction dim for the TN layout) are (b,t) pairs,
   * t = row % row_T; rows with t >= row_lens[b] + row_halo are padding whose result is defined as ZERO:
   * whole output tiles of such rows are zero-filled without touching the operands (a_kc=1), and K-blocks
   * of such rows are skipped in weight-gradient reductions (a_kc=0, b_kc=0).  NULL = off.               */
  const int32_t* row_lens; int32_t row_T; int32_t row_halo;
  /* Optional m-tile schedule for row_lens launches (ctts_row_tile_map, 64-row tiles): tile_map[0] = number of ACTIVE m-tiles,
   * tile_map[1..] = m-tile indices, active ones first.  Workgroup b then works on m-tile tile_map[1 + b / tiles_n] and n-tile
   * b % tiles_n: the active tiles occupy the first workgroup ids, so the hardware's round-robin placement spreads them evenly over
   * the 8 XCDs (a fixed permutation leaves a +-10 % imbalance of active tiles per XCD), and each XCD only sees tiles_n / 8 weight
   * panels.  NULL = built-in scrambled order. */
  const int32_t* tile_map;
  /* with tile_map: n-tiles per XCD group (0 = plain order).  g > 0 (tiles_n % g == 0, tiles_n / g in {1,2,4,8}): XCD x works on the g
   * n-tiles of group x % (tiles_n/g) and on every (8*g/tiles_n)-th scheduled m-tile - trades weight-panel against activation-tile
   * footprint in the XCD's 4 MiB L2. */
  int32_t tile_group_n;
  /* Optional softmax-backward fusion (attention): C = E * (alpha * acc - rowsub[row]) with E laid out exactly like C (same ldc and
   * batch strides) and rowsub indexed [batch * M + row]:  dS = P * (dP - D), D = rowsum(dO * O), straight out of the dP = dO V^T GEMM -
   * no separate pass over the [T,T] maps.  NULL = off; not combinable with bias / act / dropout / R / rowscale / split_k. */
  const float* E; const float* rowsub;
  /* Workspace (ctts_workspace_bytes() bytes, zero-filled ONCE by the caller, private to the stream the launch goes to: launches that
   * share it must be stream-ordered; every kernel leaves its flag / ticket words at zero).  REQUIRED when split_k > 1: the partial tiles
   * of the K pieces are written to it and summed in split order by the workgroup that finishes last - bit-reproducible, unlike float
   * atomics in arrival order.  It also enables the persistent stream-K kernel (csrc/gemm_sk.hip): with it, large unbatched GEMMs
   * run on a persistent grid that cuts the (tile, K-block) space evenly over the CUs and sums cut tiles in a fixed order; split_k > 1
   * then only means "C += alpha * A B" (no atomics).  NULL = tile-per-workgroup kernels only. */
  void* sk_ws; int64_t sk_ws_bytes;
  /* epi_bwd != 0: the epilogue is the BACKWARD of a previous layer's forward epilogue - Z is then an INPUT (that layer's stored
   * pre-activation, laid out like C with ldz) and  C = drop_mask(seed, drop_offset, m*N+n)/(1-p) * act'(Z) * alpha * acc : the data-gradient
   * GEMM of layer k+1 hands layer k its dZ directly, no separate pass over the [M,N] gradient.  Excludes bias / R / rowscale / E / split_k. */
  int32_t epi_bwd;
  /* Deferred split-K (split_k > 1, tile kernels): the partial matrices P_0 .. P_count-1 ([M, N rounded up to 4] each, `stride` floats apart -
   * ctts_gemm_split_plan) are written HERE and left for the caller, who adds them in order later with ctts_partial_sums - the second
   * halves of all weight gradients of a backward stage in one launch instead of one reduce launch per GEMM.  NULL = the library adds them
   * itself (through sk_ws) before ctts_gemm returns control of the stream. */
  float* split_out; int64_t split_out_floats;
  /* "Write everything": every element of the [M, N] block of every batch is WRITTEN, zero where the per-batch limits (lens) or a whole
   * tile of padded rows (row_lens) leave nothing to compute - the caller needs no pre-zeroed C (one fill launch per attention product /
   * convolution data gradient saved).  With split_k > 1 (library-side sum only) this also turns C += alpha * A B into C = alpha * A B;
   * with split_k <= 1 it only adds the zeros outside the limits of a batched, length-limited launch. */
  int32_t split_overwrite;
  /* Arithmetic of the large launches (a DESCRIPTOR field since round 5; it used to be the process-wide ctts_gemm_bf16_split_enable
   * switch, which a captured graph ignored and two models in one process could not set differently):
   *   0 = fp32 MFMA only (v_mfma_f32_32x32x2_f32: every product an exact fp32 product) - what a zero-initialised descriptor gets;
   *   1 = launches that qualify by shape and size may run on the BF16 matrix pipe with the six-term operand split described at
   *       ctts_gemm_takes_bf16_split below (fp32-class results);
   *   2 = as 1 without the size thresholds (parity tests of small launches);
   *   3 = REDUCED precision, opt-in ("amp": reference train.py:59,104 amp.autocast): launches the plane kernels take (A_planes / B_planes
   *       given and eligible) use the hi pieces only - operands rounded to bf16 (round-to-nearest-even), ONE MFMA term, fp32 accumulate:
   *       the result equals the product of the bf16-rounded operands up to fp32 summation; every other launch is treated as 1;
   *   4 = as 3 without the size thresholds. */
  int32_t bf16_split;
  /* Optional PRE-SPLIT operands (ctts_split_planes): A_planes / B_planes hold the exact three-way bf16 split of the SAME fp32 matrices
   * A / B point to, in the K-block-interleaved layout [rows][ld / 32][3][32] (bf16 bit patterns): piece q (0 hi, 1 mid, 2 lo) of element
   * (row, k) at planes[row * 3 * ld + (k / 32) * 96 + q * 32 + k % 32] - the three pieces of one 32-deep K-block of a row are 192
   * contiguous bytes; same ld as the fp32 operand, ld % 32 == 0, 16-byte aligned.  With both given (bf16_split >= 1, a_kc = b_kc = 1,
   * unbatched, sk_ws given) the persistent plane kernel (csrc/gemm_pl.hip) moves the pieces to LDS by DMA and its main loop is
   * ds_read + MFMA only: no split arithmetic in the GEMM, every operand element split ONCE per tensor instead of once per tile that
   * stages it.  The SAME row-major plane sets serve the TN layout (a_kc = b_kc = 0: weight gradients C[m][n] (+)= alpha * sum_k
   * A[k][m] B[k + tap - pad][c], conv view on B): csrc/gemm_plw.hip DMAs 32-row K-blocks into LDS as they lie and transposes with
   * ds_read_b64_tr_b16 - pass the planes of dZ (made for the data gradient) as A_planes and of x (made for the forward) as B_planes;
   * split_k > 1 without split_overwrite adds into C in place (M % 128 == 0, N % 256 == 0, conv_cin % 256 == 0, no epilogue terms).
   * A / B must stay valid: descriptors the plane kernels do not take (ctts_gemm_takes_planes) run on the other kernels from A / B. */
  const uint16_t* A_planes;
  const uint16_t* B_planes;
  /* Optional plane set of the OUTPUT (round 6): with epi_bwd != 0 on the weight-stationary K = 256 kernel (ctts_gemm_takes_weight_stationary)
   * the epilogue writes the exact three-way bf16 split of every C element it stores into C_planes as well ([M][ldc / 32][3][32], ldc % 32
   * == 0, N % 32 == 0, 16-byte aligned; zeros in wholly padded tiles like C itself) - bit-identical to ctts_split_planes(C).  The dZ a
   * data-gradient GEMM hands the previous layer then arrives with the operand planes that layer's own data- and weight-gradient launches
   * read (transformer_fs2.py:220-239: ffn_2 -> ffn_1).  Launches on any other kernel IGNORE the field: ask first. */
  uint16_t* C_planes;
} ctts_gemm_desc;

int ctts_gemm(const ctts_gemm_desc* d, void* stream);
/* 1 (+ *count, *stride) when ctts_gemm would run `d` as a split-K launch whose partials the caller may keep (split_out); else 0. */
int ctts_gemm_split_plan(const ctts_gemm_desc* d, int32_t* count, int64_t* stride);
/* Data-gradient operands of many Conv1d layers in one launch: for every task dst[ci][kk][co] = src[co][k-1-kk][ci], src = the GEMM-major
 * forward weight [Cout][K][Cin] (what ctts_conv_weight_repack mode 4 does for one layer; 32 x 32 tiles through LDS, both sides
 * coalesced).  `tasks` is a HOST array.  The weights are constant during a step: trainer.TrainStep calls this once before the forward. */
typedef struct ctts_repack_task { const float* src; float* dst; int32_t cout, cin, k; } ctts_repack_task;
int ctts_conv_dgrad_weights(const ctts_repack_task* tasks, int ntasks, void* stream);
/* y = rowscale[row] * dropout(x + alpha * table[pos[row]]): positional-embedding add of the fs2 stacks and predictors
 * (`x + self.pos_embed_alpha * self.embed_positions(x)` + F.dropout + non-pad mask, transformer_fs2.py:41-52,113-119, modules.py:1349-1351)
 * as one launch each way; pos from ctts_positions, table [n_pos, C] row-major, alpha a device scalar or NULL (= 1), rowscale / dropout
 * optional.  Backward: dx = rowscale * dropmask * dy; dalpha (optional, (+)= by `accumulate`) = sum of dx * table[pos] (ordered sum, ws). */
int ctts_posembed_fwd(const float* x, const int32_t* pos, const float* table, const float* alpha, const float* rowscale, float* y,
                      int64_t rows, int C, float p_drop, const uint64_t* seed, uint32_t drop_offset, void* stream);
int ctts_posembed_bwd(const float* dy, const int32_t* pos, const float* table, const float* rowscale, float* dx, float* dalpha,
                      int64_t rows, int C, float p_drop, const uint64_t* seed, uint32_t drop_offset, int accumulate, void* ws, void* stream);
/* Deferred ordered reductions, many per launch: for every task  dst[i] += alpha * (src[i] + src[stride + i] + ... + src[(count-1)*stride + i]),
 * i < n, partials added in index order (bit-reproducible).  `tasks` is a HOST array (copied into kernel arguments, 24 tasks per launch).
 * Producers: ctts_gemm with split_out; ctts_colsum / ctts_weighted_colsum / ctts_epilogue_bwd / ctts_layernorm_bwd with `parts`
 * (ctts_reduce_parts(kind, rows, C) partial rows of C floats - 2 C for kind 3 - written, nothing else). */
typedef struct ctts_psum_task { const float* src; float* dst; int64_t n; int64_t stride; int32_t count; float alpha; } ctts_psum_task;
int ctts_partial_sums(const ctts_psum_task* tasks, int ntasks, void* stream);
int ctts_reduce_parts(int kind, int64_t rows, int C);     /* kind 0 colsum, 1 weighted_colsum, 2 epilogue_bwd bias, 3 layernorm_bwd; 0 = no deferred mode */
/* Size of the per-stream workspace shared by ctts_gemm (sk_ws) and by every entry point below that takes a `ws` argument (ordered
 * cross-workgroup reductions: column sums, LayerNorm / BatchNorm parameter sums, loss sums).  `ws` = NULL is legal for those: the
 * reduction then runs on ONE workgroup per column block (slow, still deterministic).  ctts_gemm_workspace_bytes is the older name. */
size_t ctts_workspace_bytes(void);
size_t ctts_gemm_workspace_bytes(void);
/* Error word of the stream-K hand-off in a workspace (0 = clean; n > 0: an owner gave up waiting for workgroup n - 1 and the launch's
 * result is invalid).  DEVICE pointer to one uint32 inside `ws`: copy it to the host (asynchronously, e.g. every N steps) and raise. */
const uint32_t* ctts_workspace_error_word(const void* ws);
/* out[b] = XCC_ID (the XCD) workgroup b of an nblocks-wide launch ran on.  The persistent stream-K kernel assumes b % 8 (the default SPX
 * dispatch of an MI355X): callers verify once per device before handing ctts_gemm a workspace. */
int ctts_xcd_probe(int32_t* out, int nblocks, void* stream);
/* 1 when ctts_gemm would run this descriptor on the persistent stream-K kernel (sk_ws given, shape / alignment eligible, enough tiles
 * for the grid), else 0.  Callers that otherwise split the reduction (split_k > 1 + zero fill) ask first: the persistent kernel balances
 * the reduction itself and wants split_k = 1. */
int ctts_gemm_takes_persistent(const ctts_gemm_desc* d);
/* Switch for the weight-stationary K = 256 kernel (csrc/gemm_ws.hip; default on, env CTTS_WS=0 turns it off): returns the previous
 * setting.  Process-wide, not thread-safe - for parity tests and A/B timing (same descriptor on both kernel families). */
int ctts_gemm_ws_enable(int on);
/* 1 when ctts_gemm would run this descriptor on the weight-stationary kernel (no launch). */
int ctts_gemm_takes_weight_stationary(const ctts_gemm_desc* d);
/* fp32 GEMM on the BF16 matrix pipe (ctts_gemm_desc.bf16_split >= 1; csrc/gemm.hip gemm_x6_kernel / gemm_x6tn_kernel, csrc/gemm_pl.hip
 * gemm_pl_kernel): large unbatched NT launches (both operands K-contiguous, conv view on A allowed, N a multiple of 128, split_k <= 1)
 * and large unbatched TN launches (weight gradients: both operands reduction-major, conv view on B allowed, M and N multiples of 128,
 * K >= 2048, any split_k) form every fp32 product from six v_mfma_f32_32x32x16_bf16 terms of the EXACT three-way bf16 split of both
 * operands (x = hi + mid + lo, each piece the round-to-nearest-even of what is left: hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi),
 * accumulated in fp32 - fp32-class results (exact wherever the fp32 result is exact; the three dropped cross terms are <= 2^-24 of a
 * product - one fp32 rounding - and 2^-29 in the median; error against float64 as the fp32-MFMA kernels' on the same launches) at up
 * to 16/6 of the fp32-MFMA rate.
 * DOMAIN (tests/test_kernels_gpu.py::test_bf16_split_domain_*):
 *   - finite operands with 2^-100 <= |x| < 3.396e38 (or x = 0): as stated above;
 *   - |x| < 2^-100: the lo (then mid) piece leaves the normal range of bf16 and is flushed by the matrix pipe: the product keeps >= 16
 *     (then >= 8) significant bits - an absolute error below 2^-116 |y| per product, far under the fp32 rounding of any sum that also
 *     holds a normal-sized term;
 *   - 3.396e38 <= |x| <= FLT_MAX (the top 0.2 % of the fp32 range, where hi would round to infinity): the pre-split path
 *     (ctts_split_planes) clamps hi to the largest bf16 and stays exact; the kernels that split inside the GEMM (x6 / x6tn) treat
 *     such an operand like an infinity;
 *   - an infinite or NaN operand makes every output element it contributes to NON-FINITE (NaN where fp32 arithmetic would give
 *     +-inf: inf * 0-piece = NaN), elements it does not contribute to are unaffected - the finite / non-finite pattern of the result
 *     equals the fp32-MFMA kernels', which is what overflow diagnostics (isfinite checks, GradScaler) look at.
 * ctts_gemm_takes_bf16_split: 1 when ctts_gemm would run this descriptor on the in-kernel-split kernels (x6 / x6tn; no launch);
 * ctts_gemm_takes_planes: 1 when it would run it on a plane kernel (gemm_pl.hip: NT; gemm_plw.hip: TN). */
int ctts_gemm_takes_bf16_split(const ctts_gemm_desc* d);
int ctts_gemm_takes_planes(const ctts_gemm_desc* d);
/* Exact three-way bf16 split of fp32 matrices, many per launch: for every task and element (r, c), c < cols (cols % 32 == 0, ld % 32 == 0,
 * src and dst 16-byte aligned): x = src[r * ld + c] -> dst[r * 3 * ld + (c / 32) * 96 + q * 32 + c % 32] = bf16 bits of piece q (the
 * layout of ctts_gemm_desc.A_planes), hi = RNE_bf16(x) (clamped to +-bf16 max when a finite x would round to infinity),
 * mid = RNE_bf16(x - hi), lo = x - hi - mid (exact); for x = +-inf / NaN: hi = x, mid = lo = 0.  dst holds rows * 3 * ld bf16.
 * `tasks` is a HOST array.  HBM-bound: 4 bytes read + 6 written per element. */
typedef struct ctts_split_task { const float* src; uint16_t* dst; int64_t rows, cols, ld; } ctts_split_task;
int ctts_split_planes(const ctts_split_task* tasks, int ntasks, void* stream);

/* out[b,h,t] = sum_d a[b,t,h*dh+d] * b[b,t,h*dh+d]   (a, b [B,T,H*dh] channel-last; the D vector of the fused softmax backward). */
int ctts_rowdot_heads(const float* a, const float* b, float* out, int B, int T, int H, int dh, void* stream);

/* out[c] (+)= scale * sum_r w[r] * x[r,c] (x [rows,C] dense): the weight gradient of a one-output Linear - the N = 1 heads of the
 * duration / energy predictors (modules.py:1296,1349) - as one streaming pass instead of a degenerate 1 x C GEMM. */
int ctts_weighted_colsum(const float* x, const float* w, float* out, int64_t rows, int C, float scale, int accumulate, void* ws, float* parts,
                         void* stream);

/* Backward of the ctts_gemm epilogue in one pass over dY [rows,C]:  gm = dY * rowscale[row] (optional output = gradient of the
 * residual R), dZ = gm * dropout_mask(seed, drop_offset, element) / (1-p) * act'(Z) (act as in ctts_gemm_desc; Z NULL or act 0: factor 1),
 * dbias[c] (+)= bias_scale * sum_rows dZ[.,c] (optional; bias_scale = the epilogue's alpha).  Any of rowscale, z, gm, dbias may be NULL. */
int ctts_epilogue_bwd(const float* dy, const float* rowscale, const float* z, float* dz, float* gm, float* dbias, int64_t rows, int C,
                      int act, float p_drop, const uint64_t* seed, uint32_t drop_offset, float bias_scale, int accumulate_bias,
                      void* ws, float* parts, void* stream);

/* m-tile schedule for padded-row skipping (see ctts_gemm_desc.tile_map): a 64-row tile is inactive when all its rows (b,t) belong to one
 * utterance b and t >= row_lens[b] + row_halo.  tile_map: 1 + ceil(M/64) int32. */
int ctts_row_tile_map(const int32_t* row_lens, int row_T, int row_halo, int M, int32_t* tile_map, void* stream);

/* Conv1d weight repack: w[Cout][Cin][K] (reference nn.Conv1d layout) ->
 *   mode 0: wf[Cout][K][Cin]                 (forward implicit-GEMM B operand)
 *   mode 1: wd[Cin][K][Cout], taps flipped   (data-gradient implicit-GEMM B operand)
 *   mode 2: inverse of mode 0 (wgrad result [Cout][K][Cin] -> [Cout][Cin][K])
 *   mode 3: as mode 2 but ADDED to dst (gradient accumulation straight into param.grad)
 *   mode 4: wd from a weight that is already stored GEMM-major, src = wf[Cout][K][Cin] (the layout this package keeps its own Conv1d
 *           parameters in - exposed with reference shape [Cout,Cin,K] through strides - so that forward and wgrad need no repack)  */
int ctts_conv_weight_repack(const float* src, float* dst, int cout, int cin, int k, int mode, void* stream);

/* ---------------------------------------------------------------------------------------
 * LengthRegulator / dur_to_mel2ph (model/modules.py:1216-1249, utils/tools.py:577-628).
 * ctts_lr_index: per batch row, prefix-sum of max(trunc(dur),0) by wavefront scan, then
 *   mel2ph[b,t] = 1 + #{i : cum[i] <= t} for t < min(total, Tm) else 0;  mel_len[b] = total
 *   (un-cropped).  dur_is_float selects float32 vs int64 input; round_mode 0 = trunc (LR),
 *   1 = round-half-even (dur_to_mel2ph); pad[b,i] != 0 zeroes that duration (dur_padding).
 * ctts_lr_gather_fwd: out[b,t,:] = x[b, mel2ph[b,t]-1, :] or 0.      (expand + pad/crop)
 * ctts_lr_gather_bwd: dx[b,i,:] = sum over the contiguous frame run of phoneme i of dy.   */
int ctts_lr_index(const void* dur, int dur_is_float, int round_mode, const uint8_t* pad, int B, int Ts, int Tm,
                  int32_t* mel2ph, int64_t* mel_len, int32_t* cum, void* stream);
int ctts_lr_gather_fwd(const float* x, const int32_t* mel2ph, float* out, int B, int Ts, int Tm, int C, void* stream);
int ctts_lr_gather_bwd(const float* dy, const int32_t* cum, float* dx, int B, int Ts, int Tm, int C, void* stream);

/* Target-side pitch chain of the cwt pitch branch in ONE launch (csrc/pitch.hip; utils/pitch_tools.py:27-36 f0_to_coarse, :258-294 inverse_cwt_torch /
 * cwt2f0 / cwt2f0_norm with pitch_norm "log", and the denormalisation of modules.py:1071-1091): per utterance b
 *   rec[t] = sum_{j < nscale} spec[b,t,j] * (j + 3.5)^-2.5;  rec = (rec - mean_t rec) / std_t rec  (unbiased, over all T columns);
 *   f0[b,t] = log2(exp(rec * f0_std[b] * std_scale + f0_mean[b]) + eps)  (columns T .. width-1 repeat column T-1);
 *   f0_denorm = unvoiced ? 0 : 2^f0;   ids = f0_to_coarse(f0_denorm) in [1, f0_bin - 1]  (mel_min / mel_max = 1127 ln(1 + f / 700) at 50 / 1100 Hz).
 * spec [B,T,ld_spec]; unvoiced = uv[b,t] > 0 (uv float [B,width]) or, with uv NULL, spec[b,t,uv_chan] > 0 (the predictor's uv logit).
 * Outputs f0, f0_denorm [B,width] float, ids [B,width] int64.  Deterministic (ordered two-pass sums in double), no atomics, no memset. */
int ctts_cwt_pitch(const float* spec, int64_t ld_spec, int nscale, const float* f0_mean, const float* f0_std, float std_scale,
                   const float* uv, int uv_chan, float eps, float mel_min, float mel_max, int f0_bin, float* f0, float* f0_denorm,
                   int64_t* ids, int B, int T, int width, void* stream);

/* make_positions (utils/tools.py:640-652): pos = cumsum(x != 0) * (x != 0) along T.
 * src is float32 (stride `stride` elements between time steps, e.g. channel 0 of [B,T,C]) or int64 tokens. */
int ctts_positions(const void* src, int src_is_float, int64_t stride, int B, int T, int32_t* pos, void* stream);

/* ---------------------------------------------------------------------------------------
 * LayerNorm over the last dim (blocks.py:137-156 eps 1e-12; nn.LayerNorm eps 1e-5), fused with
 * inverted dropout and the non-pad row mask:  y = rowscale * drop(LN(x)).
 * planes (optional, round 6): the bf16 plane set of y in the layout of ctts_split_planes ([rows][C / 32][3][32], C % 32 == 0),
 * written by the same launch from the registers that hold y - bit-identical to ctts_split_planes(y), for the plane-kernel GEMM that
 * consumes y (FFN Conv1d, transformer_fs2.py:220-239) without a second pass over y.  The same optional argument on ctts_bn_apply (the
 * PostNet convolutions' inputs, modules.py:140-148) and ctts_bn_bwd_apply (their output gradients dZ).                               */
int ctts_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int rows, int C, float eps, float p_drop, const uint64_t* seed, uint32_t drop_offset,
                       const float* rowscale, uint16_t* planes, void* stream);
int ctts_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx, float* dgamma, float* dbeta, int rows, int C, float p_drop, const uint64_t* seed,
                       uint32_t drop_offset, const float* rowscale, int accumulate, const float* dres, void* ws, float* parts,
                       void* stream);
/* accumulate != 0: dgamma / dbeta (and ctts_colsum's out) are ADDED to - gradient-accumulation fusion straight into param.grad.
 * dres (optional, [rows,C]): added to dx - the gradient arriving through the residual connection around the pre-LN sub-layer
 * (x -> LN -> f -> + x), so the autograd sum of the two paths costs no extra pass. */

/* BatchNorm1d over [rows, C] (channel-last view of nn.BatchNorm1d, modules.py:105,140-148),
 * fused with tanh (act=3) / none and inverted dropout.
 * stats: sums[0..C) = sum x, sums[C..2C) = sum x^2 (double, pre-zeroed by the call).
 * apply: y = drop(act((x-mean)*rstd*gamma+beta)).
 * bwd_reduce: sums[0..C) = sum dt, sums[C..2C) = sum dt*xhat with dt = dy*dropmask*act'(u).
 * bwd_apply: dx = gamma*rstd*(dt - sum_dt/rows - xhat*sum_dtxhat/rows); dgamma/dbeta from sums.
 *   batch_stats bit 0: train-mode statistics (the two correction terms above), bit 1: ADD to dgamma / dbeta instead of storing. */
int ctts_colstats(const float* x, double* sums, int rows, int C, void* ws, void* stream);
/* mean / rstd of the batch from ctts_colstats sums, plus the nn.BatchNorm train-mode bookkeeping (running_mean / running_var with
 * momentum and the unbiased variance, num_batches_tracked += 1; NULL pointers skip) - one launch instead of a dozen [C]-sized ops. */
int ctts_bn_finalize(const double* sums, int rows, int C, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                     float* running_var, int64_t* num_batches, void* stream);
int ctts_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  float* y, int rows, int C, int act, float p_drop, const uint64_t* seed, uint32_t drop_offset,
                  uint16_t* planes /* optional plane set of y, see ctts_layernorm_fwd */, void* stream);
int ctts_bn_bwd_reduce(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, double* sums, int rows, int C, int act, float p_drop, const uint64_t* seed,
                       uint32_t drop_offset, void* ws, void* stream);
int ctts_bn_bwd_apply(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const double* sums, float* dx, float* dgamma, float* dbeta, int rows, int C,
                      int act, float p_drop, const uint64_t* seed, uint32_t drop_offset, int batch_stats,
                      uint16_t* planes /* optional plane set of dx */, void* stream);

/* Masked row softmax for attention scores S[nbatch, T, T] (in place), keys >= lens[z/nb1] get 0
 * and query rows >= len are left untouched (F.multi_head_attention_forward key_padding_mask).
 * bwd: dS = P * (dP - sum_k dP*P), in place on dP.                                            */
int ctts_softmax_fwd(float* S, const int32_t* lens, int nb0, int nb1, int T, int64_t ld, void* stream);
int ctts_softmax_bwd(const float* P, float* dP, const int32_t* lens, int nb0, int nb1, int T, int64_t ld,
                     void* stream);

/* Elementwise helpers of the backward pass.
 * act_dropout_bwd: dz = dg * dropmask/(1-p) * act'(z)          (GELU/ReLU/tanh from saved pre-activation)
 * rowscale_dropout_bwd: dv = dy * rowscale[m] * dropmask/(1-p)  (output-side dropout + pad mask)
 * colsum: out[c] (+)= scale * sum_rows x[r, c]                  (bias gradients)               */
int ctts_act_dropout_bwd(const float* dg, const float* z, float* dz, int64_t rows, int C, int act, float alpha_unused,
                         float p_drop, const uint64_t* seed, uint32_t drop_offset, void* stream);
int ctts_rowscale_dropout(const float* x, float* y, int64_t rows, int C, const float* rowscale, float p_drop,
                          const uint64_t* seed, uint32_t drop_offset, void* stream);
int ctts_colsum(const float* x, float* out, int64_t rows, int C, int64_t ld, float scale, int accumulate, void* ws, float* parts, void* stream);

/* ---------------------------------------------------------------------------------------
 * Mel front end (audio/stft.py:59-88,166-185): reflect-pad, |DFT| from the [F, 2*nbins]
 * re/im GEMM result, energy = L2 norm over bins; the two GEMMs go through ctts_gemm.          */
int ctts_reflect_pad(const float* y, float* ypad, int B, int N, int pad, int64_t ld_out, void* stream);
int ctts_stft_magnitude(const float* reim, int64_t ld_reim, float* mag, int64_t ld_mag, float* energy, int64_t frames,
                        int nbins, void* stream);
int ctts_log_clamp_transpose(const float* mel_fm, float* out, int B, int F, int n_mel, float clip, void* stream);

/* ---------------------------------------------------------------------------------------
 * Conformer block pieces (model/transformers/conformer.py).
 * GLU over the channel dim (blocks.py GLU, conformer.py:458): a [rows, 2C] -> out[r,c] = a[r,c] * sigmoid(a[r,C+c]).
 * Depthwise Conv1d k (odd), 'same' padding, no bias, channel-last (conformer.py:522-560, DepthwiseConv1d):
 *   y[b,t,c] = sum_k wT[k,c] * x[b,t+k-pad,c];  flip=1 uses wT[K-1-k] (data gradient);
 *   wgrad: dw[c,k] += sum_{b,t} dy[b,t,c] * x[b,t+k-pad,c]   (dw [C,K], pre-zeroed by the call).
 * Relative-position attention scores (conformer.py:347-431, RelativeMultiHeadAttention):
 *   relpos_softmax_fwd: P = softmax_j( (S[i,j] + shift(PS)[i,j]) * scale ) over ALL keys (no mask, conformer.py:243 vs :326),
 *     shift = Transformer-XL _relative_shift (conformer.py:423-431); P overwrites S, Pd = inverted-dropout(P) (optional).
 *   relpos_softmax_bwd: dS = P * (dP - sum_j dP*P) * scale with dP = dropmask/(1-p) * dPd, in place on dPd;
 *   relshift_bwd: dPS = inverse of the shift applied to dS.                                               */
/* Operand preparation of RelativeMultiHeadAttention.forward (conformer.py:396-407) from the packed q | k | v projection [rows, 3C] in
 * one pass: qu = q + u_bias, qv = q + v_bias [rows, C], kv = k | v [rows, 2C]; and its adjoint dqkv = (dqu + dqv) | dkv (the bias
 * gradients are column sums of dqu / dqv: ctts_colsum). */
int ctts_relattn_split_fwd(const float* qkv, const float* u_bias, const float* v_bias, float* qu, float* qv, float* kv, int64_t rows,
                           int C, void* stream);
int ctts_relattn_split_bwd(const float* dqu, const float* dqv, const float* dkv, float* dqkv, int64_t rows, int C, void* stream);
int ctts_glu_fwd(const float* a, float* out, int64_t rows, int C, void* stream);
int ctts_glu_bwd(const float* a, const float* dout, float* da, int64_t rows, int C, void* stream);
int ctts_dwconv_fwd(const float* x, const float* wT, float* y, int B, int T, int C, int K, int flip, void* stream);
int ctts_dwconv_wgrad(const float* dy, const float* x, float* dw, float* partials, int B, int T, int C, int K, int accumulate,
                      void* stream);      /* accumulate != 0: dw += (gradient-accumulation fusion straight into param.grad) */
/* partials: scratch of 128 * C * 32 floats (slice sums, reduced in a fixed order - deterministic, no atomics) */
int ctts_relpos_softmax_fwd(float* S, const float* PS, float* Pd, int nbatch, int T, float scale, float p_drop,
                            const uint64_t* seed, uint32_t drop_offset, void* stream);
int ctts_relpos_softmax_bwd(const float* P, float* dPd, int nbatch, int T, float scale, float p_drop, const uint64_t* seed,
                            uint32_t drop_offset, void* stream);
int ctts_relshift_bwd(const float* dS, float* dPS, int nbatch, int T, void* stream);

/* Embedding lookup (model/transformers/blocks.py:10-15, model/modules.py:779-788,947,958).
 * fwd: out[r,:] = weight[ids[r],:] (ids int64 [n], weight [V,C], C % 4 == 0).
 * bwd: dweight[v,:] (+)= sum over r with ids[r] == v of dy[r,:], row padding_idx forced to zero (nn.Embedding(padding_idx=..));
 *      one wave per (vocabulary row, 64-id chunk), register partial sums + one atomicAdd per channel.  padding_idx < 0: none. */
int ctts_embedding_fwd(const int64_t* ids, const float* weight, float* out, int64_t n, int C, int V, void* stream);
int ctts_embedding_bwd(const int64_t* ids, const float* dy, float* dweight, int64_t n, int C, int V, int padding_idx, int accumulate,
                       void* ws, void* stream);

/* ---------------------------------------------------------------------------------------
 * Unsupervised duration modelling.
 * ctts_neg_sqdist: AlignmentEncoder scores out[b,t,s] = -temp * sum_c (q[b,t,c]-k[b,s,c])^2  (model/modules.py:1199-1200), channel-last.
 * ctts_mas: monotonic alignment search, width 1 (model/modules.py:36-75 mas_width1/b_mas; called from :863-872).
 *   attn [B,Tq,Tk] soft attention (probabilities), in_lens/out_lens [B] valid text / mel lengths;
 *   opt [B,Tq,Tk] <- hard 0/1 alignment, dur [B,Tk] <- frames per phoneme (attn_hard.sum(2)), back [B,Tq,Tk] scratch bytes. */
int ctts_neg_sqdist(const float* q, const float* k, float* out, int B, int Tq, int Tk, int C, float temp, void* stream);
int ctts_mas(const float* attn, const int32_t* in_lens, const int32_t* out_lens, float* opt, float* dur, uint8_t* back, int B,
             int Tq, int Tk, void* stream);
/* ForwardSumLoss (model/loss.py:350-377) for the whole batch in one launch: per utterance b the CTC negative
 * log-likelihood nll[b] of the target sequence 1..K_b under log_softmax([blank_logprob, attn_logprob[b,t,0..K_b-1]]) over the
 * T_b = out_lens[b] valid frames (K_b = in_lens[b]).  The reference's loss is mean_b(nll[b] / K_b) with infinities zeroed.
 *   fwd: attn_logprob [B,Tq,Tk] -> lse [B,Tq] (row normalisers), alpha [B,Tq,2*Tk+1] (log forward variables), nll [B]
 *   bwd: grad [B,Tq,Tk] = gscale[b] * d nll[b] / d attn_logprob (zero for frames >= T_b, tokens >= K_b, or nll[b] = inf)   */
int ctts_forward_sum_fwd(const float* attn_logprob, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob, float* lse,
                         float* alpha, float* nll, int B, int Tq, int Tk, void* stream);
int ctts_forward_sum_bwd(const float* attn_logprob, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob,
                         const float* lse, const float* alpha, const float* nll, const float* gscale, float* grad, int B, int Tq,
                         int Tk, void* stream);

/* ---------------------------------------------------------------------------------------
 * liu2021 implicit prosody modelling (model/modules.py:332-648, model/coordconv.py:140-159).
 * ctts_im2col_3x3s2: patch matrix of Conv2d(3x3, stride (1,2), padding (1,1)) (ReferenceEncoder convs, modules.py:351-361) on
 *   channel-last x[B,T,W,C]: col[(b,t,wo)][(kh*3+kw)*C + c] = x[b, t+kh-1, 2*wo-1+kw, c] (0 outside), Wo = (W-1)/2+1, C % 4 == 0.
 *   The convolution itself is ctts_gemm(col, w[Cout][kh][kw][Cin]).  ctts_col2im_3x3s2 is the adjoint (dx from dcol).
 * ctts_gru_fwd / ctts_gru_bwd: recurrent part of `ndir` independent one-layer nn.GRU sequences per utterance (1..8: the two directions
 *   of a bidirectional GRU, or several GRUs of equal H and T sharing one launch), batch_first, zero initial state; sequence d runs
 *   t = T-1..0 iff bit d of rev_mask is set
 *   (modules.py:359-361,391 ReferenceEncoder; :618-621,637 ParallelProsodyPredictor).  gate order r|z|n,
 *   n = tanh(gi_n + r * (W_hn h + b_hn)), h' = (1-z) n + z h.  The sequence runs over all T steps (the reference never packs).
 *   gi [B,T,ndir,3H] = W_ih x + b_ih (caller's GEMM); whh [ndir,3H,H]; bhh [ndir,3H]; out [B,T,ndir,H];
 *   gates [B,T,ndir,4H] <- r|z|n|(W_hn h + b_hn) saved for backward (NULL in inference).
 *   bwd: dout [B,T,ndir,H] -> dgi, dgh [B,T,ndir,3H] (gradients w.r.t. the input / hidden pre-activations) and hprev [B,T,ndir,H]
 *   (h_{t-1}); the caller forms dW_hh = dgh^T hprev and db_hh = colsum(dgh).  H in {16,32,64,128}.
 * ctts_softmax_rect_fwd/bwd: in-place masked row softmax of S[nb,Tq,Tk]; keys >= klens[b] -> 0, query rows >= qlens[b] -> 0
 *   (PhonemeLevelProsodyEncoder attention, modules.py:443-446; STL token attention :527-529).  NULL lens = no mask. */
int ctts_im2col_3x3s2(const float* x, float* col, int B, int T, int W, int C, void* stream);
int ctts_col2im_3x3s2(const float* dcol, float* dx, int B, int T, int W, int C, void* stream);
int ctts_gru_fwd(const float* gi, const float* whh, const float* bhh, float* out, float* gates, int B, int T, int H, int ndir,
                 int rev_mask, void* stream);
int ctts_gru_bwd(const float* dout, const float* out, const float* gates, const float* whh, float* dgi, float* dgh, float* hprev,
                 int B, int T, int H, int ndir, int rev_mask, void* stream);
int ctts_softmax_rect_fwd(float* S, const int32_t* klens, const int32_t* qlens, int nb, int Tq, int Tk, void* stream);
int ctts_softmax_rect_bwd(const float* P, float* dP, const int32_t* klens, const int32_t* qlens, int nb, int Tq, int Tk,
                          void* stream);

/* ---------------------------------------------------------------------------------------
 * Mel front end in ONE call (csrc/mel.hip): TacotronSTFT.mel_spectrogram (audio/stft.py:166-185) = STFT.transform (:59-88: reflect
 * padding by n_fft/2, hann-windowed DFT, hop) -> |X| -> mel filterbank -> log(clamp(., clip)) (audio_processing.py:85-91), and
 * energy = ||X||_2 per frame.  Computed as a 1024-point REAL FFT per frame (512-point complex radix-8 Stockham + even/odd split)
 * instead of the reference's 1026 x 1024 basis convolution, the filterbank as a banded GEMM on the fp32 MFMA.
 *   y [B,N] waveform in [-1,1]; window [n_fft] = analysis window zero-padded to n_fft (periodic hann); mel [B,n_mel,F],
 *   energy [B,F], F = 1 + N / hop; mag: optional [B*F, ld_mag >= 513] magnitudes (NULL = not stored).
 *   workspace: ctts_mel_spectrogram_workspace_bytes() bytes, filled once per filterbank by ctts_mel_prepare(mel_basis [n_mel,513]).
 *   Built for n_fft = 1024 (the reference's filter_length) and n_mel <= 96. */
size_t ctts_mel_spectrogram_workspace_bytes(int n_fft, int n_mel);
int ctts_mel_prepare(const float* mel_basis, int n_fft, int n_mel, float* workspace, void* stream);
int ctts_mel_spectrogram(const float* y, const int32_t* lens, const float* window, const float* workspace, float* mel, float* energy, float* mag,
                         int64_t ld_mag, int B, int N, int n_fft, int hop, int n_mel, float clip, int kmax, uint32_t* range_flag,
                         void* stream);
/* range_flag (optional, one uint32 in device OR pinned host memory, zero-initialised by the caller): the kernel stores the float bits of an
 * offending |sample| (> 1.0, or a NaN pattern) there when the waveform leaves [-1, 1] - the reference's two range asserts
 * (audio/stft.py:177-178) without a reduction pass and without a host synchronisation; valid input writes nothing. */
/* lens (optional, [B] int32): ragged batch for preprocessing (preprocessor.py:387,467 extracts one utterance at a time) - row b holds
 * lens[b] <= N samples (rest padding); reflection happens at the utterance's own end and its 1 + lens[b]/hop leading frames equal the
 * single-utterance result; the remaining frames of the row are don't-care. */
/* kmax: 1 + the highest DFT bin with a non-zero filter weight (372 for fmax 8 kHz at 22.05 kHz), or 0 = unknown (all 513 bins are kept
 * in the on-chip magnitude tile).  A smaller tile lets three workgroups share a CU instead of two; results do not depend on it as long as
 * no filter reaches beyond bin kmax-1. */

/* ---------------------------------------------------------------------------------------
 * Fused multi-head attention (csrc/attn.hip): exact-fp32 MFMA, flash-style - the [T,T] score / probability tensors never reach HBM in
 * the forward pass, the backward recomputes them.  d_head = C / H must be 32, 64 or 128 (ctts_mha_supported).
 *
 * ctts_mha_fwd / ctts_mha_bwd: the core of F.multi_head_attention_forward as transformer_fs2.py:385-394 calls it (q scaled by `scale` =
 *   d_head^-0.5, key-padding mask from valid lengths, no biases, no attention dropout) on the packed projection qkv [B,T,3C] (q | k | v).
 *   lens[b] (or NULL = T): keys >= lens[b] are masked, query rows >= lens[b] are ZERO rows of `out` [B,T,C].
 *   lse [B,H,T]: log2-domain log-sum-exp of the scaled scores (kept for the backward).
 *   bwd: out/dout [B,T,C]; Dws [B,H,T] and dS [B,H,T,T] are caller-provided scratch; dqkv [B,T,3C] receives all three gradients
 *   (zero at padded rows).  q_split >= 1 splits the query loop of a key tile over several waves - use 2 when B*H*T/32 waves do not fill
 *   the chip; each split writes its partial dK | dV into kv_part [q_split, B, T, 2C] (scratch, may be NULL for q_split 1) and a
 *   fixed-order sum produces the result: deterministic, no atomics.
 * ctts_relmha_fwd / ctts_relmha_bwd: the core of RelativeMultiHeadAttention (conformer.py:396-421) on channel-last projections:
 *   qu = q + u_bias, qv = q + v_bias [B,T,C]; kv [B,T,2C] (k | v); pos [T,C] = pos_proj(sinusoid rows) shared by the batch;
 *   score = (qu k^T + shift(qv pos^T)) * scale, softmax over ALL keys (the reference passes no mask, conformer.py:243),
 *   dropout(p_drop) on the probabilities (counter RNG: seed, drop_offset as in ctts_gemm), context = P v.  shift() is the
 *   Transformer-XL memory reinterpretation of conformer.py:423-431; the kernels evaluate it in closed form (no [B,H,T,T] score
 *   tensor exists, nothing but out and lse [B,H,T] has to be kept for the backward).
 *   bwd: Dws [B,H,T] and dS (ctts_relmha_workspace_floats(B,T,H) = B*H*T*(T+1) floats: d loss / d score in the layout of the
 *   reference's `padded` tensor, so that the gradient of the unshifted scores is a view of the same memory) are scratch; outputs
 *   dqu, dqv [B,T,C], dkv [B,T,2C] and dpos_b [B,T,C] = per-utterance gradient of pos (the caller sums it over B). */
int ctts_mha_supported(int C, int H);
int ctts_mha_fwd(const float* qkv, const int32_t* lens, float* out, float* lse, int B, int T, int H, int C, float scale, void* stream);
int ctts_mha_bwd(const float* qkv, const int32_t* lens, const float* out, const float* dout, const float* lse, float* Dws, float* dS,
                 float* kv_part, float* dqkv, int B, int T, int H, int C, float scale, int q_split, void* ws, void* stream);
/* ws: the per-stream workspace (ctts_workspace_bytes()) - the dQ = dS K reduction is split in two when the launch would not fill the
 * chip, and the pieces are summed through it in a fixed order; NULL = never split. */
size_t ctts_relmha_workspace_floats(int B, int T, int H);
int ctts_relmha_fwd(const float* qu, const float* qv, const float* kv, const float* pos, float* out, float* lse, int B, int T,
                    int H, int C, float scale, float p_drop, const uint64_t* seed, uint32_t drop_offset, void* stream);
int ctts_relmha_bwd(const float* qu, const float* qv, const float* kv, const float* pos, const float* out,
                    const float* dout, const float* lse, float* Dws, float* dS, float* dqu, float* dqv, float* dkv,
                    float* dpos_b, int B, int T, int H, int C, float scale, float p_drop, const uint64_t* seed, uint32_t drop_offset,
                    void* stream);

/* ---------------------------------------------------------------------------------------
 * Variance / duration terms of CompTransTTSLoss (model/loss.py:123-243) as one kernel pair (csrc/loss.hip):
 *   terms[8] = {pdur, wdur, sdur, C, uv, f0_mean, f0_std, energy}, each already multiplied by its lambda (lambdas5 = {lambda_ph_dur,
 *   lambda_word_dur, lambda_sent_dur, lambda_f0, lambda_uv}; lambda_word / lambda_sent <= 0 switch the term off as the reference does).
 *   log_d, e_pred, e_tgt [B,Ts] float; dur [B,Ts] int64 (dur_is_float 0) or float32 (1, the MAS durations of learn_alignment);
 *   texts [B,Ts] int64 and sil_ids3 = the three silence token ids that delimit words (loss.py:30-33); src_pad / mel_pad uint8, 1 = pad;
 *   cwt [B,Tm,11] (10 cwt bins + uv logit), cwt_spec [B,Tm,10], uv [B,Tm]; f0 statistics [B]; cwt_l2: 0 = l1, 1 = mse.
 *   Scratch kept for the backward: partials [B,16], wsum [B,2,Ts+1], denoms [4].  Deterministic (no atomics, no memset nodes:
 *   torch's multi-block reduction mis-replays its semaphore memset inside a hipGraph on this stack).
 *   bwd: g8 = upstream gradient of every term; writes d_log_d, d_e [B,Ts], d_cwt [B,Tm,11], d_f0m, d_f0s [B].
 * ctts_bin_loss_*: BinLoss (loss.py:380-386) = -sum(log(clamp(soft,1e-12)) * hard) / sum(hard) over n elements; partials [1024],
 *   out2 = {loss, sum hard}. */
int ctts_var_loss_fwd(const float* log_d, const void* dur, int dur_is_float, const int64_t* texts, const uint8_t* src_pad, const float* cwt,
                      const float* cwt_spec, const float* uv, const uint8_t* mel_pad, const float* f0m_p, const float* f0m_t,
                      const float* f0s_p, const float* f0s_t, const float* e_pred, const float* e_tgt, int B, int Ts, int Tm,
                      const float* lambdas5, int cwt_l2, const int64_t* sil_ids3, float* partials, float* wsum, float* terms, float* denoms,
                      void* stream);
int ctts_var_loss_bwd(const float* log_d, const void* dur, int dur_is_float, const int64_t* texts, const uint8_t* src_pad, const float* cwt,
                      const float* cwt_spec, const float* uv, const uint8_t* mel_pad, const float* f0m_p, const float* f0m_t,
                      const float* f0s_p, const float* f0s_t, const float* e_pred, const float* e_tgt, int B, int Ts, int Tm,
                      const float* lambdas5, int cwt_l2, const int64_t* sil_ids3, const float* partials, const float* wsum,
                      const float* denoms, const float* g8, float* d_log_d, float* d_cwt, float* d_f0m, float* d_f0s, float* d_e,
                      void* stream);
int ctts_bin_loss_fwd(const float* soft, const float* hard, int64_t n, float* partials, float* out2, void* stream);
int ctts_bin_loss_bwd(const float* soft, const float* hard, const float* out2, const float* g, float* dsoft, int64_t n, void* stream);
/* ctts_masked_loss_*: sum_i w_i l(pred_i, target_i) / sum_i w_i over n elements, l = |p-t| (kind 0), (p-t)^2 (1) or BCE-with-logits (2):
 *   the f0 / uv terms of pitch_type "frame" and "ph" (loss.py:173-178,206-219) and the frame-level energy term (loss.py:238-242).
 *   partials [1024], out2 = {loss, sum w}; bwd writes dpred [n] = g * w_i * l'(p_i, t_i) / sum w.  Ordered two-stage reduction. */
int ctts_masked_loss_fwd(const float* pred, const float* target, const float* weight, int64_t n, int kind, float* partials, float* out2,
                         void* stream);
int ctts_masked_loss_bwd(const float* pred, const float* target, const float* weight, const float* out2, const float* g, float* dpred,
                         int64_t n, int kind, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused gradient clipping + Adam over flat fp32 arenas (train.py:118-125, model/optimizer.py:22-53):
 *   total = ||g||_2 ; g <- g * min(1, max_norm / (total + 1e-6))  (nn.utils.clip_grad_norm_; max_norm <= 0: no clipping)
 *   g <- g + weight_decay * p ; m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2
 *   p <- p - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)           (torch.optim.Adam, amsgrad off)
 * p, g, m, v: n floats each, 16-byte aligned.  lr: device scalar.  state: CTTS_ADAM_STATE_FLOATS device floats {sum of squares of this
 * call (output), step count t-1 (incremented here), total norm of this call (output), CTTS_ADAM_PARTIALS per-block partial sums
 * (scratch)} - all device resident so that the launches replay inside a hipGraph.  The norm is reduced in a FIXED order (no atomics):
 * data-parallel replicas with bit-identical gradients stay bit-identical. */
#define CTTS_ADAM_PARTIALS 2048
#define CTTS_ADAM_STATE_FLOATS (3 + CTTS_ADAM_PARTIALS)
int ctts_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float beta1, float beta2, float eps,
                        float weight_decay, float max_norm, float* state, void* stream);

/* Masked L1 of the two mel predictions in one pass (CompTransTTSLoss mel / postnet-mel terms, model/loss.py:130-138,
 * 303-304): rows with pad[row] != 0 count as zeros, w[row] = (sum_c |target[row,c]| != 0),
 *   sums = { sum w |p1 - t|, sum w |p2 - t|, sum w }  ->  loss_k = sums[k] / (C * sums[2]);   roww [rows] <- w (kept for backward);
 *   sums must be ZERO on entry (the kernel accumulates into it)
 * bwd: d1, d2 = g[k] * sign(p_k - t) * w / (C * sums[2]) with g the two upstream scalars (device). */
int ctts_mel_l1_fwd(const float* p1, const float* p2, const float* tgt, const uint8_t* pad, float* sums, float* roww, int64_t rows,
                    int C, void* ws, void* stream);
int ctts_mel_l1_bwd(const float* p1, const float* p2, const float* tgt, const float* roww, const float* sums, const float* g, float* d1,
                    float* d2, int64_t rows, int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * Gradient all-reduce of the data-parallel step (SURVEY.md section 8(b) `ctts_allreduce_*`, 8(e); replaces what
 * `DistributedDataParallel(model, device_ids=[rank])` does after backward in the reference: train.py:29-35,58,112).
 *   ctts_comm_unique_id  rank 0 draws CTTS_COMM_ID_BYTES opaque bytes (ncclGetUniqueId) and hands them to every rank by any side
 *                        channel (the Python host: `torch.distributed.broadcast_object_list`; a C host: its own launcher)
 *   ctts_comm_create     one communicator per process, bound to the CURRENT HIP device (ncclCommInitRank; collective: every rank calls it)
 *   ctts_allreduce_mean  buf[i] <- mean over ranks of buf[i], in place, n floats, stream-ordered on `stream` and capturable into a
 *                        hipGraph (ncclAllReduce, ncclFloat32, ncclAvg: the division by the world size rides inside the collective).
 *                        The product calls it once per gradient BUCKET (4 contiguous ranges of the flat gradient arena, one per
 *                        backward stage, 8 - 60 MB each) on a side stream while the next stage runs.
 *   ctts_comm_destroy    ncclCommDestroy
 * RCCL (librccl.so.1) is bound at run time by dlopen - libctts_hip.so does not link it, single-GPU use never loads it; when it cannot be
 * loaded these calls fail with a text, nothing else is affected.  Which collective runs where: `dp.BucketedReducer` uses
 * `torch.distributed` (backend "nccl" = the same RCCL) by default because the rendezvous, the process group and its watchdog are
 * already there under a PyTorch host; `CTTS_ABI_COLLECTIVE=1` (or `BucketedReducer(..., abi_collective=True)`) routes the bucket
 * all-reduces through these entry points instead - same bytes, same stream, bit-identical result (tests/test_dp_gpu.py). */
#define CTTS_COMM_ID_BYTES 128
int ctts_comm_unique_id(void* id_out /* CTTS_COMM_ID_BYTES host bytes */);
int ctts_comm_create(void** comm_out, int32_t nranks, int32_t rank, const void* id /* CTTS_COMM_ID_BYTES host bytes */);
int ctts_comm_destroy(void* comm);
int ctts_allreduce_mean(float* buf, int64_t n, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif
