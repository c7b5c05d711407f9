/*
 * libmfp_hip.so -- C-ABI of the MI355X (gfx950) kernel library for the MFP hot path.
 *
 * The reference (CyberAgentAILab/flex-dm) is pure Python on TensorFlow/Keras and has no FFI
 * of its own; the arithmetic each entry point replaces lives in stock TF ops called from the
 * reference lines cited per function below (paths relative to /root/reference/src/mfp/mfp/).
 * SURVEY.md §8(b) fixes the convention:
 *   - extern "C", plain pointers + sizes, no C++/torch types in any signature;
 *   - every entry returns 0 or a negative MFP_E* code; mfp_last_error() gives the text;
 *   - never allocates, never synchronises, never owns: the caller (PyTorch-ROCm caching
 *     allocator) owns every buffer including workspaces; launches go on the caller's stream;
 *   - thread-safe for distinct streams; no global mutable state.
 * All pointers are DEVICE pointers unless a comment says "host".
 * "cdt" = compute dtype of the activations between kernels: MFP_F32 or MFP_BF16.
 */
#ifndef MFP_HIP_H
#define MFP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mfp_stream_t; /* hipStream_t */

enum { MFP_F32 = 0, MFP_BF16 = 1 };
#define MFP_RNG_STEP_STRIDE 4096ull

enum {
  MFP_OK = 0,
  MFP_EINVAL = -1,    /* bad argument / unsupported shape */
  MFP_ELAUNCH = -2,   /* hipGetLastError() after launch */
  MFP_EWORKSPACE = -3 /* workspace too small */
};

const char* mfp_last_error(void);
int mfp_version(void);

/* Compute units of the current device, and how many of them persistent launches (one workgroup per CU: grouped weight
 * gradients, weight-stationary products, single-pass attention backward) leave FREE: with N > 1 ranks RCCL's
 * workgroups cannot share a CU with a 160 KB / 8-wave workgroup, so a data-parallel run may reserve n CUs for the
 * all-reduce that overlaps the backward pass (new; the reference's train.py:25 strategy line is commented out).
 * 0 <= n <= #CUs - 8; process-wide; returns MFP_EINVAL otherwise. */
int mfp_cu_count(void);
int mfp_set_reserved_cus(int n);

/* ---------------------------------------------------------------------------------- GEMM
 * One MFMA tile kernel with epilogue flags.
 * C[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] ).  Operand storage:
 *   a_kmajor=1: A stored [M][lda] (k contiguous);  0: A stored [K][lda] (m contiguous)
 *   b_kmajor=1: B stored [N][ldb] (k contiguous);  0: B stored [K][ldb] (n contiguous)
 * Forward Dense (transformer.py:85-98,163-169; encoder.py:88-92; decoder.py:39-43):
 *   a_kmajor=1,b_kmajor=0 with B = Keras kernel (in,out).  dgrad: a_kmajor=1,b_kmajor=1.
 *   wgrad: a_kmajor=0,b_kmajor=0 (contraction over tokens), split-K via `splitk`.
 * The product stores every Dense kernel TRANSPOSED, W^T = [out][in] (so that the fused QKV and
 * the concatenated decoder heads stay one contiguous row block per Keras variable, which the
 * per-variable clipnorm needs).  With that storage: forward y = x W uses (a_kmajor=1,
 * b_kmajor=1, B = W^T); dgrad dx = dy W^T uses (1, 0, B = W^T); wgrad dW^T = dy^T x uses
 * (0, 0, A = dy, B = x).
 * Constraints: N%8==0, K%8==0 (kmajor operands), lda/ldb/ldc %8==0; M free.
 */
enum {
  MFP_GEMM_BIAS = 1,        /* + bias[n] (f32) */
  MFP_GEMM_RELU = 2,        /* max(.,0) after bias */
  MFP_GEMM_RESIDUAL = 4,    /* + residual[m][n] (f32, ld = ldc) after dropout */
  MFP_GEMM_DROPOUT = 8,     /* inverted dropout on (acc+bias) before the residual add */
  MFP_GEMM_ACCUM = 16,      /* C += result (C f32) */
  MFP_GEMM_ROWSKIP = 32,    /* rows with rowcode[m]!=0 contribute 0 (encoder.py:174-175) */
  MFP_GEMM_RELU_BWD = 64,   /* result *= (aux[m][n] > 0), aux in cdt, ld = ldc */
  MFP_GEMM_COLSUM_A = 128,  /* wgrad only: also colsum[m] = sum_k A[k][m] (bias grad of dY) */
  MFP_GEMM_ROWSKIP_A = 256  /* wgrad only: rows k of A with rowcode[k]!=0 count as zero */
};

typedef struct mfp_gemm_args {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const float* residual;
  const void* aux;
  const uint8_t* rowcode;
  float* colsum;        /* [M] f32, MFP_GEMM_COLSUM_A */
  void* workspace;      /* wgrad split-K partials: splitk*M*N (+ splitk*M) floats */
  size_t workspace_bytes;
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t a_kmajor, b_kmajor;
  int32_t in_dtype;     /* dtype of A and B (MFP_F32 -> exact f32 MFMA, MFP_BF16) */
  int32_t out_dtype;    /* dtype of C */
  int32_t flags;
  int32_t splitk;       /* >=1; >1 only with a_kmajor=0,b_kmajor=0 */
  float dropout_p;
  uint64_t seed;
  uint64_t offset;
  const int32_t* step_ptr; /* device; RNG offset += *step_ptr * MFP_RNG_STEP_STRIDE (graph replay) */
} mfp_gemm_args;

int mfp_gemm(const mfp_gemm_args* args /*host*/, mfp_stream_t stream);
size_t mfp_gemm_workspace_bytes(const mfp_gemm_args* args /*host*/);
/* Which kernel family mfp_gemm will launch for these arguments -- "gemm_ws_kernel" (weight-
 * stationary, warp-specialised), "gemm_wg_kernel" (streaming weight gradient + split-K reduce) or
 * "gemm_kernel" (LDS-tiled) -- so that measurements are booked under the kernel that ran. */
const char* mfp_gemm_kernel_family(const mfp_gemm_args* args /*host*/);

/* ---------------------------------------------------------------- grouped weight gradients
 * Up to MFP_MAX_WGRAD_JOBS products C_j[M_j][N_j] = A_j[K][M_j]^T B_j[K][N_j] over the SAME token
 * dimension K in ONE launch, with the split-K reduction inside the launch (last arriver per output
 * tile sums the partial slabs in a fixed order) -- the weight / bias gradients of the Dense layers
 * of one DeepSVG block (transformer.py:85-98,163-169), of the decoder heads (decoder.py:39-43) or
 * of the encoder (encoder.py:74-92,156-160: Dense kernels and, through the one-hot count matrix,
 * the embedding tables).  bf16 operands, f32 results.
 *   A bf16 [K][lda] (gradient side), B bf16 [K][ldb] (activation side), C f32 [M][ldc];
 *   colsum f32 [M] or NULL: column sums of A (bias gradient); rowcode u8 [K] or NULL: rows of A
 *   whose code is non-zero count as zero rows (encoder.py:174-175).
 *   M, N, lda, ldb % 8 == 0; ldc % 4 == 0; splitk 1, 2, 4 or a multiple of 8 (mfp_wgrad_group_splitk picks it);
 *   tickets: uint32 [>= mfp_wgrad_group_tiles()] owned by the caller, ALL ZERO before the first
 *   launch that uses them; the launch leaves them zero.  Launches that may run concurrently (other
 *   streams) need their own workspace and tickets. */
#define MFP_MAX_WGRAD_JOBS 16
typedef struct mfp_wgrad_job {
  const void* A;
  const void* B;
  float* C;
  float* colsum;
  const uint8_t* rowcode;
  int32_t M, N, lda, ldb, ldc;
  int32_t _pad;
  const float* n_affine;   /* NULL, or gamma[N] followed by beta[N] (device): B holds x-hat = (x - mean) rstd of a LayerNorm whose
                              output y = x-hat gamma + beta is the operand meant -- the reduction then writes
                              C[m][n] = gamma[n] (A^T x-hat)[m][n] + beta[n] colsum[m].  Deferred form only (mfp_wgrad_group_partial
                              + mfp_wgrad_reduce), colsum required.  The train step stashes x-hat instead of y (mfp_block_fwd
                              xhat_stash) so that the LayerNorm backward reads 0.5 KB per element instead of x's 1 KB. */
} mfp_wgrad_job;
int32_t mfp_wgrad_group_tiles(const mfp_wgrad_job* jobs /*host*/, int32_t njobs);
int32_t mfp_wgrad_group_splitk(const mfp_wgrad_job* jobs /*host*/, int32_t njobs, int32_t K, int32_t deferred /* the split for
                               mfp_wgrad_group_partial (1) or mfp_wgrad_group (0): they launch different tile units */);
size_t mfp_wgrad_group_workspace_bytes(const mfp_wgrad_job* jobs /*host*/, int32_t njobs, int32_t splitk);
int mfp_wgrad_group(const mfp_wgrad_job* jobs /*host*/, int32_t njobs, int32_t K, int32_t splitk,
                    void* workspace, size_t workspace_bytes, uint32_t* tickets, mfp_stream_t stream);
/* The same launch WITHOUT its split-K reduction: every workgroup leaves its partial tile in `workspace` and the
 * gradients are NOT written.  mfp_wgrad_reduce then sums the slabs of up to MFP_MAX_WGRAD_PENDING such launches (each with
 * its OWN workspace, untouched in between; same jobs / splitk as the launch that filled it) in one launch, in the fixed
 * order of the in-launch reduction (bit-identical results).  The train step defers the block / heads / encoder groups
 * of a backward pass (or of a data-parallel bucket) this way: ~10 us less per grouped launch. */
#define MFP_MAX_WGRAD_PENDING 8
typedef struct mfp_wgrad_pending {
  const mfp_wgrad_job* jobs;   /* host */
  int32_t njobs, splitk;
  const void* workspace;       /* device: what mfp_wgrad_group_partial filled */
} mfp_wgrad_pending;
int mfp_wgrad_group_partial(const mfp_wgrad_job* jobs /*host*/, int32_t njobs, int32_t K, int32_t splitk,
                            void* workspace, size_t workspace_bytes, mfp_stream_t stream);
int mfp_wgrad_reduce(const mfp_wgrad_pending* groups /*host*/, int32_t ngroups, mfp_stream_t stream);

/* ------------------------------------------------------------------------ fp8 forward Dense (MX block-scaled)
 * BASELINE config c5 ("fp8 MFMA"): the QKV / FFN1 products of a block (transformer.py:85-90,161-166) as OCP
 * Microscaling products: blocks of 32 consecutive k share one e8m0 scale (the smallest power of two that brings the
 * block's largest magnitude to <= 448: no saturation) and hold 32 e4m3 elements round-to-nearest-even(v / scale);
 * v_mfma_scale_f32_16x16x128_f8f6f4, f32 accumulation.
 *   mfp_quantize_mxfp8: w f32 [rows][K] -> out fp8 [rows][K], scales e8m0 [rows][K / 32] (weights, once per optimizer
 *                       step; K % 32 == 0).
 *   mfp_gemm_mxfp8:     C[M][N] (bf16) = relu?(sum over blocks of (scale_x scale_w) (Xq . Wq) + bias): X bf16 [M][lda]
 *                       quantised on the fly while it is staged (no amax pass), Wq / Ws from mfp_quantize_mxfp8.
 *                       K % 128 == 0, N % 8 == 0. */
int mfp_quantize_mxfp8(const float* w, int64_t rows, int64_t K, uint8_t* out, uint8_t* scales, mfp_stream_t stream);
int mfp_gemm_mxfp8(const void* X, const uint8_t* Wq, const uint8_t* Ws, const float* bias, void* C, int32_t M, int32_t N,
                   int32_t K, int32_t lda, int32_t ldc, int32_t relu, mfp_stream_t stream);

/* --------------------------------------------------------------------------- fused MLP half (forward)
 * x2 = x1 + Dropout(relu(LN(x1) W1^T + b1) W2^T + b2) in one launch (transformer.py:161-171,222-225),
 * d_model 256 / dim_feedforward 512 only.  x1, x2 f32 [T,256]; W1 bf16 [512][256], W2 bf16 [256][512]
 * (out, in); saved for the backward pass: y2 = LN(x1) bf16 [T,256], mean/rstd f32 [T], h bf16 [T,512].
 * Dropout stream identical to MFP_GEMM_DROPOUT of mfp_gemm (same seed / offset / step_ptr meaning).
 * x2_bf16 (may be NULL): a bf16 copy of x2, written by the same epilogue -- the last block hands the decoder
 * heads (decoder.py:95-111) their MFMA operand without a cast pass.
 */
int mfp_mlp_fused_fwd(const float* x1, const float* gamma, const float* beta, const void* W1,
                      const float* b1, const void* W2, const float* b2, void* y2, float* mean,
                      float* rstd, void* h, float* x2, void* x2_bf16, int32_t T, int32_t D, float eps, float dropout_p,
                      uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);

/* The attention half of a block, forward, in one launch (transformer.py:211-221,60-99):
 *   x1 = x + Dropout(MHSA(LN1(x)) Wo^T + bo), 8 heads of 32, key-padding mask nvalid[b] (keys >= nvalid[b] get -1e9),
 * for d_model 256 and documents of exactly S = 128 positions (a 128-row tile is a document: its attention is local to
 * the workgroup that owns the tile).  Saves what mfp_qkv_fused_fwd + mfp_attention_fwd + the output projection save
 * for the backward pass, with the same meaning and layout: y1 bf16 [T,256], mean / rstd f32 [T], qkv bf16 [T,768],
 * a bf16 [T,256] (attention output), lse f32 [B][8][128]; x1 f32 [T,256].  Wqkv bf16 [768][256], Wo bf16 [256][256]
 * (out, in).  Dropout stream = MFP_GEMM_DROPOUT's (seed, offset, step_ptr). */
int mfp_attn_block_fwd(const float* x, const float* gamma, const float* beta, const void* Wqkv, const float* bqkv,
                       const void* Wo, const float* bo, const int32_t* nvalid, void* y1, float* mean, float* rstd,
                       void* qkv, void* a, float* lse, float* x1, int32_t B, int32_t S, int32_t D, int32_t H,
                       float eps, float dropout_p, uint64_t seed, uint64_t offset, const int32_t* step_ptr,
                       mfp_stream_t stream);

/* A whole DeepSVG block, forward, in one launch (transformer.py:211-229): mfp_attn_block_fwd followed, on the same
 * 128-row tile, by mfp_mlp_fused_fwd (x2 = x1 + Dropout(relu(LN2(x1) W1^T + b1) W2^T + b2)); same saved tensors, same
 * dropout streams (offset_attn / offset_mlp), x2_bf16 optional as in mfp_mlp_fused_fwd.  S = 128, d_model 256. */
int mfp_block_fwd(const float* x, const float* gamma, const float* beta, const void* Wqkv, const float* bqkv,
                  const void* Wo, const float* bo, const int32_t* nvalid, void* y1, float* mean, float* rstd,
                  void* qkv, void* a, float* lse, float* x1, const float* gamma2, const float* beta2,
                  const void* W1, const float* b1, const void* W2, const float* b2, void* y2, float* mean2,
                  float* rstd2, void* h, float* x2, void* x2_bf16, int32_t B, int32_t S, int32_t D, int32_t H,
                  float eps, float dropout_p, uint64_t seed, uint64_t offset_attn, uint64_t offset_mlp,
                  const int32_t* step_ptr, mfp_stream_t stream);
/* The same launch leaving x-hat = (x - mean) rstd (bf16) in the place of y1 = LN1(x) / y2 = LN2(x1): what the x-hat forms of
 * mfp_attn_block_bwd_ln / mfp_mlp_bwd_ln read instead of the f32 rows (0.5 KB per element and LayerNorm less) and what
 * mfp_wgrad_job::n_affine turns back into the Q|K|V / FFN1 weight gradients.  Everything else as mfp_block_fwd. */
int mfp_block_fwd_xhat(const float* x, const float* gamma, const float* beta, const void* Wqkv, const float* bqkv,
                  const void* Wo, const float* bo, const int32_t* nvalid, void* xhat1, float* mean, float* rstd,
                  void* qkv, void* a, float* lse, float* x1, const float* gamma2, const float* beta2,
                  const void* W1, const float* b1, const void* W2, const float* b2, void* xhat2, float* mean2,
                  float* rstd2, void* h, float* x2, void* x2_bf16, int32_t B, int32_t S, int32_t D, int32_t H,
                  float eps, float dropout_p, uint64_t seed, uint64_t offset_attn, uint64_t offset_mlp,
                  const int32_t* step_ptr, mfp_stream_t stream);
/* mfp_block_fwd_xhat on HALF-document tiles: two four-wave workgroups per document (64 query rows each; the other half's K / V
 * are recomputed in the workgroup, nothing is exchanged), for batches with fewer documents than the device has CUs (BASELINE
 * config 4's per-GPU share: 128 documents).  S = 128, or S = 64 with `waves` = 8 (a half tile is then one document: the datasets'
 * shape at the reference's default batch of 256).  `waves`: 4 (a wave owns two 16-row tiles, one wave per SIMD) or 8 (one
 * row tile per wave, two waves per SIMD).  Otherwise the same arguments, same saved tensors, results bit-identical to
 * mfp_block_fwd_xhat. */
int mfp_block_fwd_xhat_half(const float* x, const float* gamma, const float* beta, const void* Wqkv, const float* bqkv,
                  const void* Wo, const float* bo, const int32_t* nvalid, void* xhat1, float* mean, float* rstd,
                  void* qkv, void* a, float* lse, float* x1, const float* gamma2, const float* beta2,
                  const void* W1, const float* b1, const void* W2, const float* b2, void* xhat2, float* mean2,
                  float* rstd2, void* h, float* x2, void* x2_bf16, int32_t B, int32_t S, int32_t D, int32_t H,
                  float eps, float dropout_p, uint64_t seed, uint64_t offset_attn, uint64_t offset_mlp,
                  const int32_t* step_ptr, int32_t waves, mfp_stream_t stream);

/* --------------------------------------------------------------------------- Dense layers of a block at d_model 512
 * (csrc/block_d512.hip; BASELINE config 5 = Crello Ours-EXP-FT: reference args.py:29-38 --latent_dim 512 --num_blocks 8;
 * the layers are transformer.py:85-90,161-171,216-225 and their autodiff).  All weights bf16 [out][in] (k-major).
 *
 * mfp_ln_dense_d512:       out bf16 [T,N] = (relu?)(LayerNorm(x) W^T + bias), x f32 [T,512], W [N][512]; saves y = LN(x) bf16
 *                          [T,512], mean / rstd f32 [T] (what mfp_layernorm_fwd + mfp_gemm leave): LN1 + Q|K|V, LN2 + FFN1.
 * mfp_dense_relumask_d512: out bf16 [T,N] = (A W^T) * [aux > 0], A bf16 [T,512], W [N][512], aux bf16 [T,N] (the saved ReLU
 *                          output): dh = (d_o2 W2) * [h > 0].
 * mfp_dense_n512_res:      out f32 [T,512] = residual + Dropout(A W^T + bias), A bf16 [T,K], W [512][K], K % 128 == 0;
 *                          out_bf16 (may be NULL): a bf16 copy; dropout stream = MFP_GEMM_DROPOUT's (seed, offset, step_ptr):
 *                          attention output projection (K = 512), FFN2 (K = 1024).
 * mfp_dense_n512:          out bf16 [T,512] = A W^T, A bf16 [T,K], W [512][K]: the input gradients da = d_o1 Wo (K = 512),
 *                          dy2 = dh W1 (K = 1024), dy1 = dqkv Wqkv (K = 1536) on the transposed shadows.
 * N % 128 == 0.  Never allocate, never synchronise. */
int mfp_ln_dense_d512(const float* x, const float* gamma, const float* beta, const void* W, const float* bias, void* y,
                      float* mean, float* rstd, void* out, int32_t T, int32_t N, int32_t relu, float eps, mfp_stream_t stream);
/* The same launch leaving x-hat = (x - mean) rstd (bf16) in the place of y (see mfp_block_fwd_xhat, mfp_layernorm_bwd_xhat,
 * mfp_wgrad_job::n_affine). */
int mfp_ln_dense_d512_xhat(const float* x, const float* gamma, const float* beta, const void* W, const float* bias, void* xhat,
                      float* mean, float* rstd, void* out, int32_t T, int32_t N, int32_t relu, float eps, mfp_stream_t stream);
int mfp_dense_relumask_d512(const void* A, const void* W, const void* aux, void* out, int32_t T, int32_t N, mfp_stream_t stream);
int mfp_dense_n512_res(const void* A, const void* W, const float* bias, const float* residual, float* out, void* out_bf16,
                       int32_t T, int32_t K, float dropout_p, uint64_t seed, uint64_t offset, const int32_t* step_ptr,
                       mfp_stream_t stream);
int mfp_dense_n512(const void* A, const void* W, void* out, int32_t T, int32_t K, mfp_stream_t stream);
/* mfp_dense_n512 for an A with row stride lda in (K - 128, K]: its columns lda .. K - 1 do not exist, W's columns >= lda must be
 * zero.  The decoder heads' input gradient at d_model 512 (A = d(logits) [T][U], W = the transposed heads zero-padded to K). */
int mfp_dense_n512_lda(const void* A, int32_t lda, const void* W, void* out, int32_t T, int32_t K, mfp_stream_t stream);
/* mfp_dense_n512 followed by mfp_layernorm_bwd_xhat on its result, in ONE launch (the Keras autodiff of
 * Dense(LayerNormalization(x)) down to x: transformer.py:216-217 / 222-223): A bf16 [T,K] is the gradient of the Dense's output
 * (dh, dqkv), W bf16 [512][K] its transposed kernel, so dy = A W^T -- which never reaches HBM:
 *   dx = dres + rstd (dy gamma - mean_c(dy gamma) - xhat mean_c(dy gamma xhat))   (bf16 residual-gradient stream),
 *   ddrop = Dropout-mask(dx) / keep for (dropout_p, seed, offset, *step_ptr), or NULL: none,
 *   part f32 [T/128][3][512] = per-128-row-tile sums of dy xhat | dy | ddrop (dgamma, dbeta, the consuming Dense's bias gradient):
 *   the caller reduces them (mfp_reduce_partials / _batch with P = T / 128, pstride = 1536).
 * A workgroup owns 128 rows x 256 columns; the two row sums are exchanged between the two workgroups of a row tile through
 * `exch` (f32 [T/128][2][128][2]) and `flags` (int32 [T/128][2]: ZERO on entry, zero again on exit).  T % 128 == 0. */
int mfp_dense_n512_lnb(const void* A, const void* W, const void* xhat, const float* gamma, const float* rstd, const void* dres,
                       void* dx, void* ddrop, float* part, float* exch, int32_t* flags, int32_t T, int32_t K, float dropout_p,
                       uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);

/* Inference form of mfp_block_fwd (what MFP.__call__(training=False), iterative_decode and eval.py run: reference
 * models/mfp.py:141-207, eval.py:35-118): the same single launch with nothing saved for a backward pass -- y1, qkv, a, lse,
 * y2 and h never reach memory (2 KB instead of 7.2 KB written per element); dropout off.  x1 f32 [T,256] is scratch (the
 * MLP half re-reads it as its residual), stats f32 [4 T] scratch for the LayerNorm statistics.  S = 128, or S = 64 with an
 * even number of documents (two per 128-row tile); d_model 256. */
int mfp_block_infer(const float* x, const float* gamma, const float* beta, const void* Wqkv, const float* bqkv,
                    const void* Wo, const float* bo, const int32_t* nvalid, const float* gamma2, const float* beta2,
                    const void* W1, const float* b1, const void* W2, const float* b2, float* x1, float* stats,
                    float* x2, int32_t B, int32_t S, int32_t D, int32_t H, float eps, mfp_stream_t stream);

/* LayerNormalization + the fused Q | K | V Dense of a block in one launch (transformer.py:216-217,85-90):
 * qkv bf16 [T,768] = LN(x) W^T + bias, with y1 = LN(x) (bf16 [T,256]), mean, rstd (f32 [T]) saved for the
 * backward pass.  x f32 [T,256]; W bf16 [768][256] (out, in); d_model 256 only. */
int mfp_qkv_fused_fwd(const float* x, const float* gamma, const float* beta, const void* W, const float* bias,
                      void* y1, float* mean, float* rstd, void* qkv, int32_t T, int32_t D, float eps,
                      mfp_stream_t stream);

/* Input gradients of the same half in one launch: dh = (d_o2 W2) * [h > 0] (bf16 [T,512]) and dy2 = dh W1
 * (bf16 [T,256]).  d_o2 = the dropout-masked output gradient, bf16 [T,256]; h as saved by the forward pass;
 * W2t bf16 [512][256] = W2 transposed, W1t bf16 [256][512] = W1 transposed (the k-major shadows). */
int mfp_mlp_fused_bwd(const void* d_o2, const void* h, const void* W2t, const void* W1t, void* dh, void* dy2,
                      int32_t T, int32_t D, mfp_stream_t stream);

/* The same launch with the backward of LN2 in its epilogue (round 5: dy2 never leaves the CU -- the tile's rows are complete
 * in LDS when the second product ends): dx bf16 [T,256] = dres + d(LN2)(dy2; x, gamma, mean, rstd), ddrop bf16 = its
 * dropout-masked copy (the stream of mfp_dropout_bwd for (seed, offset, *step_ptr); drop_p = 0: a plain copy), and one row of
 * partial sums per 128-row tile, part[T / 128][3][256] = dgamma | dbeta | column sums of ddrop, for mfp_reduce_partials(_batch)
 * (P = T / 128, pstride = 768).  The bf16 residual-gradient stream only (mfp_layernorm_bwd_res16's types); T % 128 == 0.
 * Replaces mfp_mlp_fused_bwd + mfp_layernorm_bwd_res16 (reference: Keras autodiff of transformer.py:161-171, 222-225). */
int mfp_mlp_bwd_ln(const void* d_o2, const void* h, const void* W2t, const void* W1t, void* dh, const float* x,
                   const void* xhat /* bf16 [T,256] = (x - mean) rstd as stashed by mfp_block_fwd(xhat_stash = 1), or NULL; not
                   NULL: x and mean are not read (0.5 KB per element less) */, const float* gamma, const float* mean, const float* rstd, const void* dres, void* dx, void* ddrop,
                   float* part, size_t part_bytes, int32_t T, int32_t D, float drop_p, uint64_t seed, uint64_t offset,
                   const int32_t* step_ptr, mfp_stream_t stream);

/* Input gradient of the fused Q | K | V Dense in one activation-stationary launch: dy bf16 [T,256] = dqkv Wqkv,
 * dqkv bf16 [T,768], Wt bf16 [256][768] = the kernel transposed (k-major shadow).  d_model 256 only. */
int mfp_dgrad_qkv(const void* dqkv, const void* Wt, void* dy, int32_t T, int32_t D, mfp_stream_t stream);
/* The same machine for a 256 -> 256 Dense (the attention output projection): dx bf16 [T,256] = dy W, Wt bf16 [256][256]. */
int mfp_dgrad_d256(const void* dy, const void* Wt, void* dx, int32_t T, int32_t D, mfp_stream_t stream);

/* The attention half of a block, backward input-gradient chain, in one launch (Keras autodiff of transformer.py:216-221,
 * 60-99) -- replaces mfp_dgrad_d256 + mfp_attention_bwd + mfp_dgrad_qkv for documents of exactly 128 positions:
 * da = d_o1 Wo (stays on chip), dqkv = MHSA'(qkv, a, lse; da) (bf16 [T,768], written for the weight-gradient launch),
 * dy1 = dqkv Wqkv (bf16 [T,256]).  d_o1 bf16 [T,256] = dropout-masked gradient of the projection output; Wot bf16
 * [256][256] / Wqkvt bf16 [256][768] = the transposed (k-major) shadows mfp_dgrad_d256 / mfp_dgrad_qkv take; qkv, a,
 * lse as saved by the forward pass; nvalid int32 [B].  S = 128, d_model 256, 8 heads. */
int mfp_attn_block_bwd(const void* d_o1, const void* Wot, const void* qkv, const void* a, const float* lse,
                       const int32_t* nvalid, const void* Wqkvt, void* dqkv, void* dy1, int32_t B, int32_t S,
                       int32_t D, int32_t H, mfp_stream_t stream);

/* mfp_dgrad_qkv (dy1 = dqkv Wqkv) on 64-row tiles with the x-hat backward of LN1 on its result in the same launch: what
 * mfp_dgrad_qkv + mfp_layernorm_bwd_xhat compute, dy1 never written -- the three-launch attention route of batches with fewer
 * 128-row tiles than CUs (BASELINE config 4).  ddrop may be NULL (block 0); part: T / 64 rows of [3][256].  T % 64 == 0. */
int mfp_dgrad_qkv_ln_half(const void* dqkv, const void* Wt, const void* xhat, const float* gamma, const float* rstd,
                          const void* dres, void* dx, void* ddrop, float* part, size_t part_bytes, int32_t T, int32_t D,
                          float drop_p, uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);
/* mfp_mlp_bwd_ln (x-hat form) on HALF tiles: two workgroups per 128-row tile, 64 rows each, one 16-row tile per wave -- for
 * batches with fewer 128-row tiles than the device has CUs (BASELINE config 4's per-GPU share).  `part`: T / 64 rows of [3][256].
 * dh, dx, ddrop bit-identical to mfp_mlp_bwd_ln; the partial rows sum to the same parameter gradients in another grouping. */
int mfp_mlp_bwd_ln_half(const void* d_o2, const void* h, const void* W2t, const void* W1t, void* dh, const void* xhat,
                        const float* gamma, const float* rstd, const void* dres, void* dx, void* ddrop, float* part,
                        size_t part_bytes, int32_t T, int32_t D, float drop_p, uint64_t seed, uint64_t offset,
                        const int32_t* step_ptr, mfp_stream_t stream);
/* The same launch with the backward of LN1 in its epilogue (round 5; the arguments of mfp_mlp_bwd_ln): dx = dres + d(LN1)(dy1),
 * ddrop = its dropout-masked copy or nullptr (block 0: nothing consumes one), part[T / 128][3][256].  Replaces
 * mfp_attn_block_bwd + mfp_layernorm_bwd_res16. */
int mfp_attn_block_bwd_ln(const void* d_o1, const void* Wot, const void* qkv, const void* a, const float* lse,
                          const int32_t* nvalid, const void* Wqkvt, void* dqkv, const float* x, const void* xhat, const float* gamma,
                          const float* mean, const float* rstd, const void* dres, void* dx, void* ddrop, float* part,
                          size_t part_bytes, int32_t B, int32_t S, int32_t D, int32_t H, float drop_p, uint64_t seed,
                          uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);

/* Encoder, both 512-wide numerical attributes in one launch (encoder.py:156-160,174-175,194-198):
 * h[t] += sum_j [code_j[t] == 0] (x_j[t] W_j^T + b_j), x_j bf16 [T,512], W_j bf16 [256][512], b_j f32 [256],
 * code_j u8 [T] (non-zero: the attribute is masked / absent at that position and contributes nothing here),
 * h f32 [T,256] accumulated in place.  d_model 256, K = 512 only. */
int mfp_encoder_dense2(const void* x0, const void* x1, const void* W0, const void* W1, const float* b0, const float* b1,
                       const uint8_t* code0, const uint8_t* code1, float* h, int32_t T, int32_t D, int32_t K,
                       mfp_stream_t stream);

/* Input gradient of a wide Dense in one activation-stationary launch: C f32 [T,256] = A[T][:K] Wt^T with A bf16
 * [T][lda] and Wt bf16 [256][ldw] = the kernel transposed, ZERO beyond column K (ldw a multiple of 128, >= K): the
 * decoder heads (decoder.py:39-43; K = 1384 at Crello).  d_model 256 only.
 * C_drop (may be NULL): bf16 [T,256] = C with the Dropout mask (dropout_p, seed, offset, step_ptr as in
 * MFP_GEMM_DROPOUT) and the 1 / keep scale applied -- the gradient entering the last block's second Dense
 * (transformer.py:225: x + dropout(mlp(...)) under Keras autodiff), i.e. mfp_dropout_bwd without its launch. */
int mfp_dgrad_rows(const void* A, int32_t lda, const void* Wt, int32_t ldw, float* C, int32_t T, int32_t D, int32_t K,
                   void* C_drop, float dropout_p, uint64_t seed, uint64_t offset, const int32_t* step_ptr,
                   mfp_stream_t stream);

/* --------------------------------------------------------------------------- LayerNorm
 * Keras LayerNormalization(), eps 1e-3 (transformer.py:172-173,216,222).
 * x f32 [T,D]; y cdt [T,D]; mean/rstd f32 [T].  D % 64 == 0, D <= 1024.
 */
int mfp_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y,
                      float* mean, float* rstd, int32_t T, int32_t D, float eps,
                      int32_t out_dtype, mfp_stream_t stream);
/* dx[t] = (dres ? dres[t] : 0) + LN'(dy)[t]; dgamma/dbeta f32 [D].
 * workspace: mfp_layernorm_bwd_workspace_bytes(T,D). dx may alias dres.
 * Optional fused consumer (ddrop != NULL): also writes ddrop = cdt(keep ? dx/(1-p) : 0) with the
 * dropout stream of MFP_GEMM_DROPOUT / mfp_dropout_bwd and drop_colsum[D] = column sums of ddrop
 * (the gradient of the Dense bias behind the Dropout, transformer.py:218-219,224-225). */
int mfp_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean,
                      const float* rstd, const float* dres, float* dx, float* dgamma,
                      float* dbeta, void* workspace, size_t workspace_bytes, int32_t T,
                      int32_t D, int32_t dy_dtype, void* ddrop, float* drop_colsum, float drop_p,
                      uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);
/* The same with the residual gradient stream (dres in, dx out) in bf16 [T,D] instead of f32: the bf16 train step carries
 * the gradient of the residual stream in the compute dtype (1 KB per element and LayerNorm less HBM traffic). */
int mfp_layernorm_bwd_res16(const void* dy, const float* x, const float* gamma, const float* mean,
                            const float* rstd, const void* dres, void* dx, float* dgamma,
                            float* dbeta, void* workspace, size_t workspace_bytes, int32_t T,
                            int32_t D, int32_t dy_dtype, void* ddrop, float* drop_colsum, float drop_p,
                            uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);
/* mfp_layernorm_bwd_res16 from the bf16 x-hat stash (xhat [T,D] = (x - mean) rstd as mfp_ln_dense_d512_xhat / mfp_block_fwd_xhat
 * leave it) instead of x and mean: 2 instead of 4 bytes per element read. */
int mfp_layernorm_bwd_xhat(const void* dy, const void* xhat, const float* gamma, const float* rstd, const void* dres, void* dx,
                           float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int32_t T, int32_t D,
                           int32_t dy_dtype, void* ddrop, float* drop_colsum, float drop_p, uint64_t seed, uint64_t offset,
                           const int32_t* step_ptr, mfp_stream_t stream);
size_t mfp_layernorm_bwd_workspace_bytes(int32_t T, int32_t D);
/* dgamma == dbeta == NULL: mfp_layernorm_bwd leaves the per-workgroup partials
 * [P = ceil(T/32)][3][D] (dgamma | dbeta | colsum(ddrop)) in `workspace` and the caller sums them
 * later, off the critical path, with
 *   out[c] = sum_{p<P} part[p*pstride + c], c < N;  c < split1 -> out0[c], c < split2 ->
 *   out1[c - split1], else out2[c - split2]. */
int32_t mfp_layernorm_bwd_partial_rows(int32_t T);
int mfp_reduce_partials(const float* part, float* out0, float* out1, float* out2, int64_t split1,
                        int64_t split2, int32_t P, int64_t N, int64_t pstride, mfp_stream_t stream);
/* The same reduction for up to MFP_MAX_REDUCE_JOBS independent partial buffers in one launch (the
 * partials of every LayerNorm layer of the step, summed once at the end of the backward pass;
 * same summation order per output as mfp_reduce_partials, N <= 8192 per job). */
#define MFP_MAX_REDUCE_JOBS 16
typedef struct mfp_reduce_job {
  const float* part;
  float* out0; float* out1; float* out2;
  int64_t split1, split2, N, pstride;
  int32_t P;
} mfp_reduce_job;
int mfp_reduce_partials_batch(const mfp_reduce_job* jobs /*host*/, int32_t njobs, mfp_stream_t stream);

/* --------------------------------------------------------------------------- attention
 * MultiHeadSelfAttention.attention (transformer.py:60-76) fused: softmax(QK^T/sqrt(hd) +
 * -1e9*(1-keymask)) V per (document, head); no (B,H,S,S) round trip.
 * qkv cdt [B*S, 3*D] rows = tokens, columns [q | k | v], head h = columns h*hd..;
 * nvalid int32 [B] = number of valid keys per document (= length+1, mask.py:29);
 * out cdt [B*S, D]; lse f32 [B,H,S].  hd in {16,32,64}; S <= 256.
 */
int mfp_attention_fwd(const void* qkv, const int32_t* nvalid, void* out, float* lse, int32_t B,
                      int32_t S, int32_t H, int32_t hd, int32_t dtype, mfp_stream_t stream);
int mfp_attention_bwd(const void* qkv, const int32_t* nvalid, const void* out, const void* dout,
                      const float* lse, void* dqkv, int32_t B, int32_t S, int32_t H, int32_t hd,
                      int32_t dtype, mfp_stream_t stream);

/* ------------------------------------------------------------- embedding gather + pooling
 * Encoder categorical path (encoder.py:156-160,194-199): out[t] = sum_c tables[rowoff[c] +
 * idx[t][c]] over NCOL index columns (idx < 0 -> column skipped: used for the <MASK>/<UNUSED>
 * special rows of numerical attributes, encoder.py:167-175).  idx int32 [T,NCOL];
 * rowoff int32 [NCOL]; tables f32 [ROWS,D]; out f32 [T,D].
 */
int mfp_embed_pool_fwd(const int32_t* idx, const int32_t* rowoff, const float* tables, float* out,
                       int32_t T, int32_t NCOL, int32_t ROWS, int32_t D, mfp_stream_t stream);
int mfp_embed_pool_bwd(const int32_t* idx, const int32_t* rowoff, const float* dout,
                       float* dtables, void* workspace, size_t workspace_bytes, int32_t T,
                       int32_t NCOL, int32_t ROWS, int32_t D, mfp_stream_t stream);
size_t mfp_embed_pool_bwd_workspace_bytes(int32_t T, int32_t NCOL, int32_t ROWS, int32_t D);
/* One-hot count matrix of the index columns: P bf16 [T][ROWSP] (ROWSP % 8 == 0, >= number of table
 * rows), P[t][rowoff[c] + idx[t][c]] += 1 for every column c with idx >= 0.  The table gradient of
 * Encoder.call's embedding sums (encoder.py:156-160,194-199) is then P^T * d(out) on mfp_gemm
 * (a_kmajor = b_kmajor = 0), which replaces mfp_embed_pool_bwd on the bf16 path. */
int mfp_embed_onehot(const int32_t* idx, const int32_t* rowoff, uint16_t* P, int32_t T, int32_t NCOL,
                     int32_t ROWSP, mfp_stream_t stream);

/* rowcode[t] = 1 if all(x[t]==10.0) (<MASK>), 2 if all(x[t]==0.0) (<UNUSED>, wins), else 0
 * (encoder.py:165-166; masking.py:8-9); special_idx[t*stride] = rowcode-1 (or -1). */
int mfp_row_flags(const float* x, uint8_t* rowcode, int32_t* special_idx, int32_t idx_stride,
                  int32_t T, int32_t K, mfp_stream_t stream);

/* ------------------------------------------------------------------------------ losses
 * LossLayer (metrics.py:213-299) fused per attribute: weight(t) = mfp_mask[t] &&
 * cond(type[t]) && s < nvalid[b]; loss/score/den summed over tokens; d(logits) written
 * scaled by 1/B (mean over B, metrics.py:277).  Categorical: compute_categorical_mfp_metric
 * (metrics.py:36-49) incl. Keras clip(p,1e-7,1-1e-7)->log->renormalise.  Numerical:
 * compute_continuous_mfp_metric (metrics.py:52-57), loss x512 (metrics.py:247-248).
 */
typedef struct mfp_loss_key {
  int32_t col_off;        /* first column of this head in the logits row */
  int32_t n_feat;         /* N (1, or 3 for color); numerical: 1 */
  int32_t n_class;        /* C; numerical: vector width (512) */
  int32_t is_numerical;
  const void* target;     /* int32 [T, n_feat] or f32 [T, n_class] */
  const uint8_t* mask;    /* mfp mask [T] */
  const int32_t* cond_idx;/* [T] values of the loss_condition key (stride cond_stride) or NULL */
  int32_t cond_stride;
  uint32_t cond_bits;     /* bit v set <=> condition mask[v] true */
} mfp_loss_key;

#define MFP_MAX_LOSS_KEYS 16
/* logits f32 [T, ld]; dlogits cdt [T, ld] (may be NULL: metrics only).  Heads must not overlap;
 * runs of < 8 columns between categorical heads (and up to the next multiple of 8 behind one) are
 * treated as padding: ignored in logits, written as 0 in dlogits;
 * sums f32 [nkeys][3] = {loss_sum (already / B), score_sum, den_sum}, zeroed by the call. */
int mfp_loss_fwd_bwd(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys /*host*/,
                     int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                     int32_t dl_dtype, mfp_stream_t stream);

/* RICO position-sorted loss (reference models/metrics.py:180-211, mfp.py:336-338): position t of
 * the loss reads logits row pred_row[t] (and writes that row of dlogits) and target / condition
 * row true_row[t]; the mfp mask and nvalid stay positional (metrics.py:251,263).  Either map may be
 * NULL (identity).  Each map must be a permutation of [0, T) so that every dlogits row is written. */
int mfp_loss_fwd_bwd_sorted(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys /*host*/,
                            int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                            int32_t dl_dtype, const int32_t* pred_row, const int32_t* true_row,
                            mfp_stream_t stream);

/* The same, ACCUMULATING: sums [nkeys][3] must be zero on entry (mfp_step_prologue) -- no zeroing launch.
 * pred_row / true_row may be NULL.  (A grand total is deliberately not accumulated here: one more same-address
 * atomic per workgroup cost the two loss kernels +33 us; the step's loss is the host-side sum of sums[:, 0].) */
int mfp_loss_fwd_bwd_acc(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys /*host*/,
                         int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                         int32_t dl_dtype, const int32_t* pred_row, const int32_t* true_row, mfp_stream_t stream);

/* Decoder heads + LossLayer + the heads' input gradient in ONE launch (decoder.py:95-111, metrics.py:213-299 and their
 * autodiff) -- replaces mfp_gemm (heads forward) + mfp_loss_fwd_bwd + mfp_dgrad_rows on the bf16 path, d_model 256:
 *   logits = x W^T + bias (x bf16 [B*S,256], W bf16 [U][256], bias f32 [U]; U % 8 == 0, U <= 1536), per key the losses
 *   of mfp_loss_fwd_bwd (same mfp_loss_key array; categorical items on 8-column boundaries and <= 64 classes, numerical
 *   widths % 8 == 0), dlogits bf16 [B*S,U] (what the heads' weight gradient multiplies), dx = dlogits W f32 [B*S,256]
 *   (dx and / or its plain bf16 copy dx_bf16: at least one) and, when dx_drop != NULL, its dropout-masked 1/keep-scaled
 *   bf16 copy (keying of mfp_dgrad_rows).
 * logits f32 [B*S,U] is written only when non-NULL.  The per-key sums leave as per-workgroup partials:
 * part f32 [mfp_heads_loss_partials(B*S)][48], entry [w][3 k + {0,1,2}] = {loss / B, score, count} of key k; reduce with
 * mfp_reduce_partials(part, sums, ..., P = partials, N = 3 * nkeys, pstride = 48).  No position-sorted variant. */
size_t mfp_heads_loss_partials(int32_t T);
int mfp_heads_loss_fwd_bwd(const void* x, const void* W, const float* bias, int32_t U, const mfp_loss_key* keys /*host*/,
                           int32_t nkeys, const int32_t* nvalid, float* part, void* dlogits, float* logits, float* dx,
                           void* dx_bf16, void* dx_drop, int32_t B, int32_t S, int32_t D, float dropout_p, uint64_t seed,
                           uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);
/* The same launch on 64-row tiles (one eight-wave workgroup per 64 rows) for batches with fewer 128-row tiles than CUs
 * (BASELINE config c4: 128 documents per GPU = 128 tiles on 256 CUs): part f32 [mfp_heads_loss_partials_half(B*S)][48];
 * logits, dlogits, dx and its copies are bit-identical to mfp_heads_loss_fwd_bwd's, the per-key sums the same terms in
 * twice as many partial rows. */
size_t mfp_heads_loss_partials_half(int32_t T);
int mfp_heads_loss_fwd_bwd_half(const void* x, const void* W, const float* bias, int32_t U, const mfp_loss_key* keys /*host*/,
                                int32_t nkeys, const int32_t* nvalid, float* part, void* dlogits, float* logits, float* dx,
                                void* dx_bf16, void* dx_drop, int32_t B, int32_t S, int32_t D, float dropout_p, uint64_t seed,
                                uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);

/* sort_inputs (reference models/tensor_utils.py:14-44) as a row map.  Per document b with flag[b]:
 *   priority(s) = sum_k v_k(s) * 100^(4-k) + [s >= nvalid[b]] * 100^5   (k over type, left, top,
 *   width, height), ascending stable order; row_map[b*S + r] = b*S + (position holding rank r).
 * Documents with flag[b] == 0 get the identity.  Two sources for v_k:
 *   labels mode (logits == NULL): labels[k] (host array of 5 device pointers) int32, element
 *     (b*S+s)*label_stride[k];
 *   logits mode (from_logits=True, :26-27): first-index argmax of logits[(b*S+s)*ld + col_off[k]
 *     .. + n_class[k]) (n_class < 100, :21). */
int mfp_sort_positions(const int32_t* const* labels /*host[5]*/, const int32_t* label_stride /*host[5]*/,
                       const float* logits, int32_t ld, const int32_t* col_off /*host[5]*/,
                       const int32_t* n_class /*host[5]*/, const int32_t* nvalid, const uint8_t* flag,
                       int32_t* row_map, int32_t B, int32_t S, mfp_stream_t stream);

/* --------------------------------------------------------------------------- optimizer
 * Keras Adam + per-variable clipnorm + L2 regularisers (train.py:71-77; utils.py:8-22), fused
 * over flat f32 buffers w,g,m,v that hold nseg variables back to back.  The buffers are cut
 * into <=4096-element chunks that never straddle a variable; the (static per model) chunk
 * table is built once on the host with mfp_adam_chunk_table() and uploaded by the caller.
 *   g_eff = g*grad_scale + 2*l2[seg]*w                 (L2 regulariser gradient)
 *   g_eff *= clipnorm / max(||g_eff||_seg, clipnorm)   (Keras clipnorm, per variable)
 *   Keras Adam with lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps outside the sqrt.
 * stats f32 [nseg][2] receives {||g_eff||^2, sum w^2} (pre-update; sum w^2 gives the L2 loss).
 * partial f32 [nchunks][2]: scratch for the per-chunk norm partials; they are combined per
 * variable in a FIXED order (no float atomics) so that data-parallel replicas stay bit-identical.
 * shadow (bf16 copy of the updated weights, same layout) may be NULL.
 * step_t: device int32 scalar, incremented by the call before use (1-based).
 */
int64_t mfp_adam_num_chunks(const int32_t* seg_off /*host [nseg+1]*/, int32_t nseg);
int mfp_adam_chunk_table(const int32_t* seg_off /*host*/, int32_t nseg, int32_t* chunk_seg /*host*/,
                         int64_t* chunk_beg /*host*/, int32_t* chunk_len /*host*/,
                         int32_t* seg_first /*host [nseg+1]*/);
int mfp_adam_keras(float* w, const float* g, float* m, float* v, uint16_t* shadow,
                   const int32_t* chunk_seg, const int64_t* chunk_beg, const int32_t* chunk_len,
                   const int32_t* seg_first, int64_t nchunks, const float* seg_l2, float* partial,
                   float* stats, int32_t nseg, int32_t* step_t, float lr, float beta1, float beta2,
                   float eps, float clipnorm, float grad_scale, mfp_stream_t stream);
int mfp_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, mfp_stream_t stream);

/* Transposed bf16 shadow of Dense kernels inside the flat parameter buffer: for each listed matrix
 * (flat offset, rows, cols; row-major [rows][cols] = Keras kernel stored [out][in]) writes
 * out[ooff + c*old + r] = bf16(w[off + r*cols + c]) (old >= rows: output row stride, so that the
 * fused Q|K|V kernels become one [D][3D] matrix).  seg_* are DEVICE arrays of nseg entries;
 * max_tiles = max over matrices of ceil(rows/32)*ceil(cols/32).  With it the input-gradient of a
 * Dense layer (reference architecture/transformer.py:85-98 backward) is the same k-major product as
 * its forward. */
int mfp_transpose_cast_bf16(const float* w, uint16_t* out, const int64_t* seg_off, const int32_t* seg_rows,
                            const int32_t* seg_cols, const int64_t* seg_ooff, const int32_t* seg_old,
                            int32_t nseg, int32_t max_tiles, mfp_stream_t stream);

/* dy = cdt( keep(seed,offset)[m][n] ? dx[m][n]/(1-p) : 0 ), colsum[n] = sum_m dy (bias grad).
 * Same dropout stream as MFP_GEMM_DROPOUT for equal (seed, offset); p == 0 -> plain cast.
 * workspace: mfp_colsum_workspace_bytes(M, N). */
size_t mfp_colsum_workspace_bytes(int32_t M, int32_t N);
int mfp_dropout_bwd(const float* dx, void* dy, float* colsum, void* workspace, size_t workspace_bytes,
                    int32_t M, int32_t N, float p, uint64_t seed, uint64_t offset,
                    const int32_t* step_ptr, int32_t out_dtype, mfp_stream_t stream);
/* The same with dx and dy both bf16: the train step that carries the gradient of the residual stream in the compute dtype
 * (mfp_layernorm_bwd_res16) -- d_model 512, where the gradient of the last block's output comes from a plain product. */
int mfp_dropout_bwd_res16(const void* dx, void* dy, float* colsum, void* workspace, size_t workspace_bytes,
                          int32_t M, int32_t N, float p, uint64_t seed, uint64_t offset,
                          const int32_t* step_ptr, mfp_stream_t stream);
/* colsum[n] = sum_m X[m][n] for a cdt matrix (bias gradients). */
int mfp_colsum(const void* X, float* colsum, void* workspace, size_t workspace_bytes, int32_t M,
               int32_t N, int32_t ld, int32_t dtype, mfp_stream_t stream);

/* ------------------------------------------------------------------------ input masking
 * preprocess_for_train fused (mfp.py:95-138; masking.py:24-53,68-155,227-269): per element and
 * attribute decide keep / <MASK> / <UNUSED> / random token from the document's task id
 * (0 random, 1 elem, 2+g attribute group g) and write the encoder's inputs directly:
 *   categorical: idx_all[t][idx_col + f] = modified index (C = <MASK>, C+1 = <UNUSED>)
 *   numerical  : x_out[t][0..W) (cdt) = row / 10.0 / 0.0 / N(0,0.1); rowcode[t] in {0,1,2};
 *                idx_all[t][idx_col] = 0 (<MASK>) / 1 (<UNUSED>) / -1
 *   mask_out[t] = MFP mask bit of the attribute (LossLayer weight).
 * Philox offset = offset + *step_ptr * MFP_RNG_STEP_STRIDE (step_ptr may be NULL).
 */
typedef struct mfp_mask_col {
  int32_t is_numerical;
  int32_t n_feat;          /* categorical: N; numerical: width W (W % 8 == 0) */
  int32_t input_dim;       /* categorical C */
  int32_t group;           /* attribute-group id (task = group + 2) */
  const void* src;         /* int32 [T][N] or f32 [T][W] (unmasked batch column) */
  const int32_t* cond_idx; /* loss_condition key values [T*cond_stride] or NULL */
  int32_t cond_stride;
  uint32_t cond_bits;
  int32_t idx_col;         /* first column of this attribute in idx_all */
  int32_t _pad;
  void* x_out;             /* numerical only */
  uint8_t* rowcode;        /* numerical only */
  uint8_t* mask_out;       /* [T] */
} mfp_mask_col;
#define MFP_MAX_MASK_COLS 16
int mfp_mask_tokens(const mfp_mask_col* cols /*host*/, int32_t ncols, int32_t* idx_all, int32_t NCOL,
                    const int32_t* nvalid, const int32_t* tasks, int32_t B, int32_t S, uint64_t seed,
                    uint64_t offset, const int32_t* step_ptr, int32_t x_dtype, mfp_stream_t stream);

/* tasks[b] ~ Categorical(probs / sum(probs)), b < B (reference mfp.py:34-43,301): one counter-based
 * uniform per document, offset as in mfp_mask_tokens.  probs: HOST array of n <= 16 weights. */
int mfp_sample_tasks(const float* probs /*host*/, int32_t n, int32_t* tasks, int32_t B, uint64_t seed,
                     uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream);

/* The head of a train step in one launch: mfp_sample_tasks (same stream of draws) + nvalid[b] = length[b] + 1
 * (reference architecture/mask.py get_seq_mask under mfp.py:101; nvalid may be NULL) + zero[0 .. nzero) = 0 (the
 * loss accumulators consumed by mfp_loss_fwd_bwd_acc; zero may be NULL with nzero = 0). */
int mfp_step_prologue(const float* probs /*host*/, int32_t n, int32_t* tasks, const int32_t* length, int32_t* nvalid,
                      int32_t B, uint64_t seed, uint64_t offset, const int32_t* step_ptr, float* zero, int32_t nzero,
                      mfp_stream_t stream);

/* Hardware probe (tests only): lane mapping of ds_read_b64_tr_b16.  byte_addr int32 [64]
 * (8-byte aligned offsets into a 4 KiB LDS image whose b16 element e holds e); out u16 [64][4]. */
int mfp_debug_tr_probe(const int32_t* byte_addr, uint16_t* out, mfp_stream_t stream);
/* One v_mfma_scale_f32_16x16x128_f8f6f4 (both operands e4m3): a, b uint8 [64 lanes][32], sa, sb int32 [64] (the scale
 * register of every lane, byte 0 used), out f32 [64 lanes][4]. */
int mfp_debug_mx_probe(const uint8_t* a, const uint8_t* b, const int32_t* sa, const int32_t* sb, float* out,
                       mfp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MFP_HIP_H */
