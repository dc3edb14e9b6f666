/*
 * mmvae_hip.h -- C-ABI of libmmvae_hip.so: the MI355X (gfx950) kernels of the MMVAE training step.
 *
 * This is the drop-in boundary for the hot path of zdebruine/MMVAE (`cmmvae`).  The reference owns no native code:
 * every entry point below replaces a chain of ATen dispatches made from the cited reference lines.  The Python
 * host (mmvae_amd/) binds these symbols with ctypes; a reference maintainer would bind them the same way
 * (INTEGRATION.md shows the stub).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes.  No torch / pybind types.
 *   - Every pointer is a DEVICE pointer to fp32 unless the name says otherwise (u8 masks, i64 labels).
 *   - Matrices are row-major with an explicit leading dimension (elements, not bytes).
 *   - `stream` is a hipStream_t passed as void*.  Calls only enqueue work on it: no allocation, no host sync,
 *     no hidden state -> every call is hipGraph-capturable and re-entrant per stream.
 *   - Caller owns all buffers, including `workspace`.  Required sizes come from the *_workspace_bytes helpers.
 *   - Return value: MMVAE_OK, or an MMVAE_ERR_* code.  No exceptions cross the ABI.  Shapes are validated on the
 *     host BEFORE anything is launched, so a bad call never reaches the GPU.
 *   - Arithmetic is fp32 end to end (reference: `precision: 32`, configs/trainer/config.yaml:5): fp32 operands,
 *     fp32 accumulation, fp32-GEMM accuracy.  Chip-filling GEMMs evaluate every fp32 product as six exact bf16 x bf16
 *     MFMA products of a three-way bf16 split of both operands (MMVAE_GEMM_PRECISION_BF16X3, the default; dropped
 *     terms < 2^-24 |ab|); small GEMMs and MMVAE_GEMM_PRECISION_F32 use v_mfma_f32_32x32x2_f32 (an fmaf chain).
 */
#ifndef MMVAE_HIP_H
#define MMVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mmvae_stream_t; /* hipStream_t */

#define MMVAE_OK 0
#define MMVAE_ERR_ARG 1       /* invalid shape / null pointer / unsupported flag */
#define MMVAE_ERR_LAUNCH 2    /* hipLaunch / hipMemsetAsync reported an error */
#define MMVAE_ERR_WORKSPACE 3 /* workspace too small */

/* ABI version: bumped whenever an entry point is added or a signature changes (mmvae_abi_version() returns the
 * value the library was built with; bindings compare it with the header they were written against).
 *   1  round-1 surface (first 20 entry points)      2  end of round 1 (50 entry points)      3+  round 2 */
#define MMVAE_ABI_VERSION 10
int mmvae_abi_version(void);
const char* mmvae_build_arch(void);

/* ------------------------------------------------------------------------------------------------------------
 * Dense Linear GEMMs  (replaces nn.Linear forward and its autograd: components.py:276, called from
 * FCBlock.forward components.py:292-314, Encoder.forward components.py:791-795, Adversarial.forward :666-674)
 *
 *   C[M,N] = alpha * A(M,K) . B(K,N)  (+ bias[N])  (relu)  (+ C if accumulate)
 *
 * layout selects how A and B sit in memory:
 *   MMVAE_GEMM_NT  A[M,K] (K contiguous), B[N,K] (K contiguous)   y  = x . W^T    Linear forward
 *   MMVAE_GEMM_NN  A[M,K] (K contiguous), B[K,N] (N contiguous)   dx = dy . W     input gradient
 *   MMVAE_GEMM_TN  A[K,M] (M contiguous), B[K,N] (N contiguous)   dW = dy^T . x   weight gradient
 *
 * splitk >= 1: number of K partitions.  With splitk > 1 the partial products go to `workspace`
 * ([splitk, M, N] fp32) and a second kernel of the same call reduces them (fixed order -> bitwise reproducible).
 * With MMVAE_GEMM_RAW_SLABS the reduce is skipped: C receives the raw slabs [splitk, M, ldc] (no alpha, bias,
 * relu), to be consumed by mmvae_fc_epilogue_fwd / _bwd which sum them on the fly.
 * splitk == 0 lets the library pick (mmvae_gemm_plan).
 * ------------------------------------------------------------------------------------------------------------ */
#define MMVAE_GEMM_NT 0
#define MMVAE_GEMM_NN 1
#define MMVAE_GEMM_TN 2

#define MMVAE_GEMM_RELU 1u        /* C = max(C, 0) after bias */
#define MMVAE_GEMM_ACCUMULATE 2u  /* C += result (beta = 1) */
#define MMVAE_GEMM_RAW_SLABS 4u   /* write exactly [splitk, M, ldc] partial slabs, no epilogue (bias must be NULL) */
#define MMVAE_GEMM_OPERAND_SLACK 8u /* the caller guarantees 16 readable bytes past the last element of A and of B: lets a
                                      rows-contiguous operand whose extent is not a multiple of 4 (60 530 / 52 437-gene
                                      matrices) use the pipelined 16-byte loader (its edge group reads past a row end) */

/* How the chip-filling GEMMs multiply (process-wide switch, host side; default BF16X3):
 *   MMVAE_GEMM_PRECISION_F32     v_mfma_f32_32x32x2_f32: exact f32 products (bitwise an fmaf chain), 157 TFLOP/s peak.
 *   MMVAE_GEMM_PRECISION_BF16X3  every f32 operand is split exactly into three bf16 pieces (3 x 8 = 24 significant
 *                                bits) while it is staged to LDS and a*b is evaluated as the six leading bf16 products
 *                                on v_mfma_f32_32x32x16_bf16 with f32 accumulation: f32-GEMM accuracy (dropped terms
 *                                < 2^-24 |a b|) at 2.67x the matrix-core throughput.  Inputs and outputs stay f32.
 * The small (64x64-tile) GEMMs always use the exact-f32 instruction.  mmvae_recon_tiles depends on the mode. */
#define MMVAE_GEMM_PRECISION_F32 0
#define MMVAE_GEMM_PRECISION_BF16X3 1
int mmvae_gemm_set_precision(int mode);
/* Host-side launch state (like the precision switch): max_workgroups > 0 caps the grid of the persistent
 * (wave-specialised) GEMM kernel for the launches that follow, 0 removes the cap.  A capped launch leaves the other
 * CUs to kernels of another stream: the engine runs a weight-gradient GEMM that way beside the latency-bound backward
 * chain of the core layers.  Results do not depend on the cap. */
int mmvae_gemm_set_workgroup_cap(int max_workgroups);
/* Kernel selection for the chip-filling bf16x3 GEMMs: 1 = the wave-specialised persistent kernel (one 512-thread
 * workgroup per CU), 0 = the 2 x 4-wave kernels, -1 = follow the environment (MMVAE_X3W, default on).  Process-wide
 * launch state like the precision switch; results do not depend on it.  mmvae_gemm_get_x3w: what a launch would take. */
int mmvae_gemm_set_x3w(int mode);
int mmvae_gemm_get_x3w(void);
int mmvae_gemm_get_precision(void);

/* Library heuristic: picks the block tile (128 or 64; the bf16x3 path may widen a 128 tile to 128x160 / 160x128 at
 * launch when that needs fewer rounds of resident workgroups) and split-K factor for a shape.  Pure host function. */
int mmvae_gemm_plan(int layout, int M, int N, int K, int* tile_out, int* splitk_out);
size_t mmvae_gemm_workspace_bytes(int layout, int M, int N, int K, int splitk);

int mmvae_gemm_f32(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda, const float* B,
                   int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags, int splitk,
                   float* workspace, size_t workspace_bytes, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused last decoder layer + reconstruction loss (k3)
 * replaces: expert decoder last Linear + ReLU (components.py:276,286 via cmmvae.py:111) followed by
 *           F.mse_loss(xhat, x, reduction="sum") and its autograd (vae.py:143)
 *
 *   P      = h[B,H] . W[G,H]^T + bias[G]
 *   xhat   = max(P, 0)                                   (optional store)
 *   dP     = 2 (xhat - x) * 1[P > 0]                     (optional store; unscaled d recon / d P)
 *   se_part[t, b] = sum over the genes of column tile t of (xhat - x)^2                 (t < mmvae_recon_tiles(G))
 *                   The column tile is 128 or 160 genes wide (mode and shape decide); all mmvae_recon_tiles(G) rows
 *                   are written by every call, rows the chosen tiling does not use as zeros.
 *
 * The per-cell squared error is reduced across the wavefront with shuffles inside the GEMM epilogue; the
 * [tiles, B] partials are summed in fixed order by mmvae_elbo_finalize (bitwise reproducible, no atomics).
 * ------------------------------------------------------------------------------------------------------------ */
int mmvae_recon_tiles(int G);
int mmvae_decoder_recon_f32(int B, int G, int H, const float* h, int64_t ldh, const float* W, int64_t ldw,
                            const float* bias, const float* x, int64_t ldx, float* xhat, int64_t ldxhat, float* dP,
                            int64_t lddp, float* se_part, mmvae_stream_t stream);
/* K-sample form: `rows` = K*B stacked decoder inputs, x has x_rows = B rows, output row r compares with x[r % B];
 * se_part is [tiles, rows]. */
int mmvae_decoder_recon_rows_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh, const float* W,
                                 int64_t ldw, const float* bias, const float* x, int64_t ldx, float* xhat,
                                 int64_t ldxhat, float* dP, int64_t lddp, float* se_part, mmvae_stream_t stream);
/* Launch state for the fused decoder / reconstruction entry points (host side, read when a launch is enqueued): on = the
 * caller vouches that h has ZERO columns from H up to the next multiple of 32 inside its leading dimension and that
 * every row of W may be read that far (the next row; 32 readable floats behind the last one).  A hidden width that is
 * not a multiple of the 32-wide k-tile (e.g. 1000) then takes the pipelined kernels over the padded K instead of the
 * guarded loop (2 x slower).  Results: the products beyond H are exact zeros. */
int mmvae_recon_set_h_kpad(int on);
/* The same launch, also leaving the column sums of dP (= the gradient of the layer's bias when the rows are not
 * re-weighted afterwards, K = 1) as per-row-tile partials: col_part [mmvae_recon_row_tiles(rows)][G], to be summed over
 * the row tiles in order (mmvae_sum_parts_batch).  col_part = NULL: exactly mmvae_decoder_recon_rows_f32.  Saves the
 * separate pass over the [rows, G] gradient (41 MB at C2).  Replaces: autograd of `bias` in components.py:276. */
int mmvae_recon_row_tiles(int rows);
int mmvae_decoder_recon_rows_colsum_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh, const float* W,
                                        int64_t ldw, const float* bias, const float* x, int64_t ldx, float* xhat,
                                        int64_t ldxhat, float* dP, int64_t lddp, float* se_part, float* col_part,
                                        mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * FCBlock layer epilogues ("column kernels")
 * replaces: BatchNorm1d(momentum=0.01, eps=0.001) train/eval forward (components.py:279), activation
 *           (components.py:282-286, ReLU only on the HIP path), Dropout (components.py:287-288) and their autograd.
 *
 * Forward:   z = bias + sum_s in[s]            (in: [S, B, ld] slabs; S = 1 for a plain GEMM output)
 *            BN (has_bn): training: batch mean / biased var over B, running stats updated with unbiased var;
 *                         eval: running stats.            y = gamma * (z - mean) * invstd + beta
 *            a = relu ? max(y, 0) : y                      (the tensor the reference collects as "hidden" after "af")
 *            d = mask ? a * mask * (1 / (1 - p)) : a        (mask: u8 0/1 keep mask, parity mode or Philox-filled)
 * Outputs: z_out (needed by BN backward; may be NULL when !has_bn), a_out (may be NULL when no dropout and
 *          d_out given), d_out, save_mean / save_invstd [N] (training BN only).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct mmvae_bn_params {
    const float* gamma;        /* [N] bn.weight */
    const float* beta;         /* [N] bn.bias */
    float* running_mean;       /* [N] updated in training */
    float* running_var;        /* [N] updated in training */
    int64_t* num_batches_tracked; /* scalar, incremented in training; may be NULL */
    float momentum;            /* 0.01 in the reference */
    float eps;                 /* 0.001 in the reference */
} mmvae_bn_params;

/* Scratch for the per-row-chunk column partials of the two-pass statistics (needed for training BN forward, and for
 * any backward that produces dbias / BN gradients). */
size_t mmvae_fc_workspace_bytes(int B, int N);

int mmvae_fc_epilogue_fwd(int B, int N, const float* in, int64_t ld_in, int n_slabs, const float* bias,
                          const mmvae_bn_params* bn /* NULL = no BN */, int training, int relu,
                          const uint8_t* keep_mask /* [B,N] or NULL */, float dropout_p, float* z_out,
                          float* a_out, float* d_out, int64_t ld_out, float* save_mean, float* save_invstd,
                          float* workspace, size_t workspace_bytes, mmvae_stream_t stream);

/* Backward of the same layer tail.
 *   dd   = row_scale[b] * (sum_s din[s] + addend)        (grad wrt d; addend/row_scale optional)
 *   da   = (mask ? dd * mask / (1 - p) : dd) + addend_a    (addend_a: optional gradient on the PRE-dropout activation
 *                                                           a -- the hidden representation FCBlock.forward returns,
 *                                                           components.py:308-314 -- which bypasses the keep mask)
 *   dy   = relu ? da * 1[a > 0] : da                      (a: forward a_out or d_out when no dropout)
 *   BN training backward: dbeta = sum_b dy, dgamma = sum_b dy*xhat, dz = gamma*invstd*(dy - dbeta/B - xhat*dgamma/B)
 *   no BN: dz = dy
 *   dbias = sum_b dz                                         (grad of the Linear bias; ahead of a BatchNorm this is
 *                                                             zero in exact arithmetic and is evaluated in closed form)
 * dz_out may be NULL (pure column sum).  dz_out may alias din when n_slabs == 1.
 * Without BN, a non-NULL workspace always receives the per-row-chunk column partials of dz as [ceil(B/32)][N] floats;
 * they are summed into dbias when dbias is non-NULL.  With dbias == NULL the caller finishes them itself, e.g.
 * together with other pending reductions in one mmvae_sum_parts_batch launch. */
int mmvae_fc_epilogue_bwd(int B, int N, const float* din, int64_t ld_in, int n_slabs, const float* addend,
                          const float* addend_a, const float* row_scale, const uint8_t* keep_mask, float dropout_p, int relu,
                          const float* a, const float* z, const float* gamma, const float* save_mean,
                          const float* save_invstd, int has_bn, float* dz_out, int64_t ld_out, float* dbias,
                          float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes,
                          mmvae_stream_t stream);

/* LayerNorm(elementwise_affine=False), eps 1e-5 (components.py:281) -- used by ConditionalLayer blocks ("next" row f2). */
int mmvae_layernorm_fwd(int B, int N, const float* x, int64_t ldx, float eps, float* y, int64_t ldy, float* save_mean,
                        float* save_invstd, mmvae_stream_t stream);
int mmvae_layernorm_bwd(int B, int N, const float* dy, int64_t lddy, const float* y, int64_t ldy,
                        const float* save_invstd, float* dx, int64_t lddx, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Reparameterisation + Gaussian KL (k8)
 * replaces: Encoder.forward tail (components.py:795-801: v = exp(a) + var_eps, Normal(m, sqrt v).rsample())
 *           and kl_divergence(qz, N(0,1)).sum(-1) (vae.py:136-137), plus the Mean/Variance log sums
 *           (cmmvae_model.py:170-171).
 *   v = exp(a) + var_eps ; s = sqrt(v) ; z[k,b,:] = mu + s * eps[k,b,:]      (k < K samples; K = 1 in the reference)
 *   kl_row[b] = sum_j 0.5 (s^2 + mu^2 - 1 - log(s^2))
 *   stat_row[0,b] = sum_j mu ; stat_row[1,b] = sum_j s^2                  (optional, [2,B])
 * One wavefront per cell; the latent axis is reduced with wavefront shuffles.
 * ------------------------------------------------------------------------------------------------------------ */
int mmvae_reparam_kl_fwd(int B, int Z, int K, const float* mu, const float* a_raw, const float* eps, float var_eps,
                         float* std_out, float* z_out, float* kl_row, float* stat_row, mmvae_stream_t stream);

/* Backward.  c_b = (dkl_row ? dkl_row[b] : 1) * (kl_scale_dev ? *kl_scale_dev : 1) * kl_scale_host
 *   dmu = sum_k dz[k] + dmu_extra + c_b * mu
 *   ds  = sum_k dz[k] * eps[k] + dstd_extra
 *   da  = (ds / (2 s) + c_b * 0.5 (1 - 1/v)) * (v - var_eps)
 * dz may be NULL (treated as 0); *_extra may be NULL. */
int mmvae_reparam_kl_bwd(int B, int Z, int K, const float* mu, const float* std, const float* eps, const float* dz,
                         const float* dmu_extra, const float* dstd_extra, const float* dkl_row,
                         const float* kl_scale_dev, float kl_scale_host, float var_eps, float* dmu, float* da_raw,
                         mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Stand-alone sum-of-squares reconstruction loss + gradient (k9)   replaces F.mse_loss(reduction="sum") vae.py:143
 *   se_row[b] = sum_g (xhat - x)^2 ; dxhat = gscale * 2 (xhat - x)   (dxhat optional; gscale_dev optional device scalar)
 * ------------------------------------------------------------------------------------------------------------ */
int mmvae_mse_sum_fwd_bwd(int B, int G, const float* xhat, int64_t ldxhat, const float* x, int64_t ldx, float* se_row,
                          float* dxhat, int64_t lddx, const float* gscale_dev, float gscale_host,
                          mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * ELBO finalisation (a6) and the K-sample extension (a7 / k10)
 * replaces: vae.py:136-152 (loss = recon + kl_weight * mean_b KL_b).  K > 1 is a build-defined extension
 * (SURVEY 0): recon = sum_b -logmeanexp_k(-SE[b,k]); K = 1 reduces exactly to the reference.
 *   se_part: [T, K*B] partial squared errors (T = tiles of mmvae_decoder_recon_f32, or 1 for mse_sum rows),
 *            sample k of cell b at column k*B + b.
 *   out[0] = loss, out[1] = recon, out[2] = kl (mean over cells), out[3] = kl_weight, out[4] = mean(mu),
 *   out[5] = mean(var)      (accumulated in fp64, stored fp32)
 *   w_out[k*B + b] = softmax_k(-SE[b,:])  : d recon / d SE[b,k]   (== 1 for K = 1).   K <= 64.
 * ------------------------------------------------------------------------------------------------------------ */
int mmvae_elbo_finalize(int B, int K, int T, const float* se_part, const float* kl_row, const float* stat_row, int Z,
                        const float* kl_weight_dev, float kl_weight_host, float* out6, float* w_out,
                        float* recon_row /* [B] scratch: per-cell reconstruction term */, mmvae_stream_t stream);

/* Opt-in "full IWAE" objective of the K-sample extension (the reference has no multi-sample ELBO): the
 * analytic KL is replaced by the sampled log-density ratio inside the log-mean-exp,
 *   r[k,b]  = log q(z_kb | x_b) - log p(z_kb) = sum_j ( -log s_bj - eps_kbj^2 / 2 + z_kbj^2 / 2 )     (_logratio)
 *   loss    = sum_b -logmeanexp_k( -SE[k,b] - (kl_weight / B) r[k,b] )                                   (_finalize_iwae)
 *   out[0] = loss, out[1] = sum_b sum_k w SE, out[2] = mean_b sum_k w r, out[3] = kl_weight, out[4..5] as above;
 *   w_out[k*B + b] = softmax_k of the log-weights = d loss / d SE[k,b];  rows3: [3, B] scratch.
 * Backward (_bwd_terms): dz[k,b,:] += (kl_weight / B) w[k,b] z[k,b,:] in place, dstd_extra[b,j] = -(kl_weight / B) / s[b,j];
 * then mmvae_reparam_kl_bwd with kl_scale_host = 0 and that dstd_extra.  kl_weight = *kl_weight_dev * kl_weight_host. */
int mmvae_iwae_logratio(int B, int Z, int K, const float* std, const float* eps, const float* z, float* out,
                        mmvae_stream_t stream);
int mmvae_elbo_finalize_iwae(int B, int K, int T, const float* se_part, const float* logratio, const float* stat_row,
                             int Z, const float* kl_weight_dev, float kl_weight_host, float* out6, float* w_out,
                             float* rows3, mmvae_stream_t stream);
int mmvae_iwae_bwd_terms(int B, int Z, int K, const float* kl_weight_dev, float kl_weight_host, const float* w,
                         const float* z, const float* std, float* dz, float* dstd_extra, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Adversarial heads: CrossEntropyLoss(reduction="sum") forward + gradient (k11)
 * replaces: cmmvae_model.py:54,85 (nn.CrossEntropyLoss(reduction="sum") on Adversarial head logits)
 *   loss_rows[b] = logsumexp(l_b) - l_b[y_b] ; dlogits = gscale * (softmax(l_b) - onehot(y_b)),
 *   gscale = gscale_host * (gscale_dev ? *gscale_dev : 1)
 * Gradient reversal (components.py:889-899) is a sign on the dx GEMM alpha, not a kernel.
 * ------------------------------------------------------------------------------------------------------------ */
int mmvae_cross_entropy_sum(int B, int C, const float* logits, int64_t ld, const int64_t* labels, float* loss_rows,
                            float* dlogits, int64_t ldd, const float* gscale_dev, float gscale_host,
                            mmvae_stream_t stream);
/* The heads of one adversary in ONE launch: head h reads columns [col[h], col[h] + classes[h]) of `logits` [B, ld]
 * (the heads' Linears packed into one matrix), its labels at labels + h*B, and writes its per-row losses to
 * loss_rows + h*B and its gradient into the same columns of `dlogits`.  col / classes: DEVICE int32 [H];
 * max_classes = the widest head (<= 8192).  Per head the arithmetic is that of the wide-head kernel of
 * mmvae_cross_entropy_sum. */
int mmvae_cross_entropy_heads(int B, int H, int max_classes, const int32_t* col_dev, const int32_t* classes_dev,
                              const float* logits, int64_t ld, const int64_t* labels, float* loss_rows, float* dlogits,
                              int64_t ldd, float gscale, mmvae_stream_t stream);
/* sums n floats in fixed order (fp64 accumulate) into out[0] (+= if accumulate). Used for loss_rows, se_row. */
int mmvae_sum_f32(int64_t n, const float* v, float* out, int accumulate, mmvae_stream_t stream);
/* H such sums in one launch: out_each[h] = sum of the n floats at v + h*ld (each reduced as mmvae_sum_f32 does),
 * out_total[0] = their float sum in row order (either output may be NULL).  The per-head losses of one adversarial
 * phase and their total (cmmvae_model.py:118-136: sum over heads of CrossEntropyLoss(sum)). */
int mmvae_sum_rows_f32(int H, int64_t n, const float* v, int64_t ld, float* out_each, float* out_total,
                       mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimiser: global-norm clip + Adam over a flat parameter arena (k12, k13)
 * replaces: clip_grad_norm_(params, 10) via Lightning clip_gradients (cmmvae_model.py:126-129,203-209) and
 *           torch.optim.Adam(lr=5e-3, weight_decay=1e-6).step() (cmmvae_model.py:309-318,130,212-213).
 * All parameters of one optimiser live contiguously in one arena (params/grads/exp_avg/exp_avg_sq each [n]).
 *
 * mmvae_grad_sqnorm:   partial[i] = sum of squares of chunk i (fixed chunking -> reproducible); needs
 *                      mmvae_sqnorm_partials(n) floats.
 * mmvae_adam_prepare:  single block.  flags & MMVAE_PREPARE_NORM: state[1] = total grad norm (pre-clip) from the
 *                      partials (else the stored norm is kept); flags & MMVAE_PREPARE_ADVANCE: state[0] (step) += 1.
 *                      Always: state[2] = clip coefficient min(1, max_norm/(norm+1e-6)) (1 if max_norm <= 0),
 *                      state[3] = 1 - beta1^step, state[4] = 1 - beta2^step.
 *                      grad_scale multiplies the gradients first (DDP averaging: 1/world_size).
 * mmvae_adam_step:     g = clip*grad_scale*grad + wd*p ; m += (1-b1)(g-m) ; v = b2 v + (1-b2) g^2 ;
 *                      p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)            (torch.optim.Adam, amsgrad=False)
 *                      state[5] = cv > 0 (written by the caller; mmvae_adam_prepare never touches it): clip BY VALUE --
 *                      GradientClipConfig(algorithm="value"), config.py:8, Lightning clip_gradients ->
 *                      clip_grad_value_ -- g = clamp(clip*grad_scale*grad, -cv, cv) + wd*p (pass max_norm = 0: clip = 1).
 * ------------------------------------------------------------------------------------------------------------ */
#define MMVAE_ADAM_STATE_FLOATS 8
#define MMVAE_PREPARE_NORM 1u
#define MMVAE_PREPARE_ADVANCE 2u
int64_t mmvae_sqnorm_partials(int64_t n);
int mmvae_grad_sqnorm(int64_t n, const float* grad, float* partials, mmvae_stream_t stream);
int mmvae_adam_prepare(int64_t n_partials, const float* partials, float max_norm, float grad_scale, float beta1,
                       float beta2, float* state, unsigned flags, mmvae_stream_t stream);
/* Host-side launch state (like mmvae_gemm_set_workgroup_cap): workgroups > 0 confines mmvae_adam_step /
 * mmvae_adam_step_copy to that many compute units (1024-thread workgroups, one per CU) for callers that run the update
 * beside a kernel of another stream whose grid is capped to the remaining units; 0 = the chip-filling grid.
 * Elementwise work: results do not depend on it.  Replaces nothing in the reference (torch.optim.Adam.step,
 * cmmvae_model.py:203-213, is stream-ordered behind backward). */
int mmvae_adam_set_workgroups(int workgroups);
int mmvae_adam_get_workgroups(void);
/* mmvae_adam_step with a rider: workgroup 0 also copies copy_n floats copy_src -> copy_dst (both must not overlap the
 * arenas) -- the step's logged scalars into a log buffer without a launch of their own.  copy_n = 0: mmvae_adam_step. */
int mmvae_adam_step_copy(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                         const float* state, float lr, float beta1, float beta2, float eps, float weight_decay,
                         float grad_scale, int copy_n, const float* copy_src, float* copy_dst, mmvae_stream_t stream);
/* The norm pass over 1..4 ranges of a gradient arena (e.g. what no fused GEMM epilogue covers) AND mmvae_adam_prepare
 * in one launch: range i = grads[i][0 .. lens[i]); its mmvae_sqnorm_partials(lens[i]) partials are written behind each
 * other at `partials`; the workgroup that finishes last (`ticket`: one zero-initialised word the kernel resets) sums the
 * n_partials_all partials at partials_all -- these and any written earlier, e.g. by mmvae_gemm_f32_sq -- in their fixed
 * order and fills state[] as mmvae_adam_prepare(flags | MMVAE_PREPARE_NORM) does.  Same numbers as the separate launches. */
int mmvae_grad_sqnorm_ranges_prepare(int n_ranges, const float* const* grads, const int64_t* lens, float* partials,
                                     unsigned* ticket, int64_t n_partials_all, const float* partials_all, float max_norm,
                                     float grad_scale, float beta1, float beta2, float* state, unsigned flags,
                                     mmvae_stream_t stream);
int mmvae_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* state,
                    float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                    mmvae_stream_t stream);

/* Adam over a list of arena segments in ONE launch (one workgroup per job): the update of a conditional-layer model
 * touches only the parameter tensors that took part in the step -- torch.optim.Adam skips parameters whose .grad is
 * None and keeps a step count per parameter (cmmvae_model.py:309-318 with ConditionalLayer parameters, components.py:
 * 346-351).  A job = up to MMVAE_ADAM_JOB_ELEMS consecutive elements of one tensor with that tensor's own bias
 * corrections bc1 = 1 - beta1^t, bc2 = 1 - beta2^t; `jobs_dev` is a DEVICE array; the clip coefficient is read from
 * `state` (mmvae_adam_prepare without MMVAE_PREPARE_ADVANCE). */
#define MMVAE_ADAM_JOB_ELEMS 16384
typedef struct {
    int64_t offset; /* first element (arena index) */
    int32_t len;    /* elements, <= MMVAE_ADAM_JOB_ELEMS */
    float bc1, bc2;
    int32_t reserved;
} mmvae_adam_job;
int mmvae_adam_step_jobs(int n_jobs, const mmvae_adam_job* jobs_dev, float* param, const float* grad, float* exp_avg,
                         float* exp_avg_sq, const float* state, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float grad_scale, mmvae_stream_t stream);
/* partials[j] = sum of squares of job j's segment of `grad` (0 for len == 0): the norm pass of such a step, over the
 * tensors that took part only.  A captured program launches a FIXED n_jobs and pads the table with empty jobs (both
 * job kernels return at once for them); mmvae_adam_prepare then sums the n_jobs partials. */
int mmvae_grad_sqnorm_jobs(int n_jobs, const mmvae_adam_job* jobs_dev, const float* grad, float* partials,
                           mmvae_stream_t stream);
/* grad[job.offset .. +job.len) = 0 for every job whose `reserved & 3` is 1 or 2 (the others return at once).  1: a
 * segment that takes part in this step only because another rank produced a gradient for it -- this rank contributes
 * zeros to the all-reduce that follows.  2: a "retired" segment (stepped last time, not now): zeroed so that a dense
 * all-reduce never sums stale values; mmvae_grad_sqnorm_jobs / mmvae_adam_step_jobs skip such jobs. */
int mmvae_grad_zero_flagged_jobs(int n_jobs, const mmvae_adam_job* jobs_dev, float* grad, mmvae_stream_t stream);
/* Gather the first n_jobs segments of a job table into a staging buffer and scatter them back: a gradient exchange over
 * the tensors that took part moves the staging buffer instead of the whole arena (conditional layers: hundreds of
 * blocks out of thousands).  Job j's place is (reserved >> 2) * 128 floats -- the host lays the segments out back to
 * back, each rounded up to 128 floats (padding zero-filled by the pack); the low 2 bits of `reserved` stay the flags of
 * mmvae_grad_zero_flagged_jobs. */
int mmvae_jobs_pack(int n_jobs, const mmvae_adam_job* jobs_dev, const float* arena, float* staging, mmvae_stream_t stream);
int mmvae_jobs_unpack(int n_jobs, const mmvae_adam_job* jobs_dev, float* arena, const float* staging, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Device RNG (k15): Philox4x32-10 streams for production mode (parity mode passes explicit masks / eps).
 * replaces: nn.Dropout mask draw (components.py:288) and Normal.rsample noise (components.py:801).
 * rng_state: device uint64[2] = {seed, offset}; offset is advanced by the kernel's consumption when
 * `advance` != 0 (so a captured hipGraph draws fresh numbers on every replay).
 * ------------------------------------------------------------------------------------------------------------ */
int mmvae_philox_keep_mask(int64_t n, float p_drop, uint8_t* mask, uint64_t* rng_state, uint64_t stream_id,
                           int advance, mmvae_stream_t stream);
int mmvae_philox_normal(int64_t n, float* out, uint64_t* rng_state, uint64_t stream_id, int advance,
                        mmvae_stream_t stream);
/* Several fills in one launch (the keep-masks and the noise of one step): job = {out, n, stream_id, p_drop, kind}, kind 0
 * = keep mask (uint8, as mmvae_philox_keep_mask), 1 = standard normal (float, as mmvae_philox_normal); no advance.  The
 * numbers are those of the single-fill entry points for the same (rng_state, stream_id).  `jobs_dev` is a DEVICE array;
 * max_n = the largest n among the jobs. */
typedef struct {
    void* out;
    int64_t n;
    uint64_t stream_id;
    float p_drop;
    int32_t kind;
} mmvae_philox_job;
int mmvae_philox_fill_jobs(int n_jobs, const mmvae_philox_job* jobs_dev, int64_t max_n, uint64_t* rng_state,
                           mmvae_stream_t stream);
/* The same fills, and rng_state[1] += advance_by by the workgroup that finishes last (`ticket`: one zero-initialised word
 * the kernel resets): no separate mmvae_philox_advance launch.  Same numbers. */
int mmvae_philox_fill_jobs_advance(int n_jobs, const mmvae_philox_job* jobs_dev, int64_t max_n, uint64_t* rng_state,
                                   uint64_t advance_by, unsigned* ticket, mmvae_stream_t stream);
/* rng_state[1] += by  (one call at the end of a step whose fills used distinct stream_ids with advance = 0). */
int mmvae_philox_advance(uint64_t* rng_state, uint64_t by, mmvae_stream_t stream);

/* Batched fixed-order reduction of partial results: for every job
 *     dst[r, c] (+)= alpha * sum_{p < n_parts} src[p * part_stride + r * ld_src + c]        (p ascending)
 * in ONE launch (grid.y = job).  Finishes split-K slabs of several weight-gradient GEMMs (MMVAE_GEMM_RAW_SLABS) and
 * the column partials of mmvae_fc_epilogue_bwd together, where one launch per reduction would cost more than the
 * reductions (a dependent launch is ~4.5 us inside a hipGraph).  `jobs` is a DEVICE array.  flags: MMVAE_GEMM_ACCUMULATE. */
typedef struct {
    const float* src;
    float* dst;
    int64_t part_stride, ld_src, ld_dst;
    int32_t n_parts, rows, cols;
    float alpha;
    uint32_t flags;
    uint32_t reserved;
} mmvae_sum_job;
int mmvae_sum_parts_batch(int n_jobs, const mmvae_sum_job* jobs, int64_t max_elems /* rows * cols of the largest job,
                          sizes the grid; 0 = unknown */, mmvae_stream_t stream);

/* Unsplit GEMM that also leaves the sum of squares of everything it stores, one partial per output tile, in
 * sq_partials[0 .. sq_capacity) (slots beyond the tiles it uses are zeroed): the weight-gradient GEMMs of the G-wide
 * layers write 82 MB each, and the gradient-norm pass of clip_grad_norm_ (cmmvae_model.py:126-129,203-209) would read
 * them all back.  The partials are summed with the other norm partials of the optimiser by mmvae_adam_prepare (fp64,
 * fixed order: bitwise reproducible).  mmvae_gemm_sq_partials: slots to reserve (0 = the shape would be split-K:
 * use mmvae_gemm_f32 and the norm pass); operands_regular != 0: A and B are 16-byte aligned with leading dimensions
 * that are multiples of 4 -- the count is then exact and the launch zero-fills nothing. */
int mmvae_gemm_sq_partials(int layout, int M, int N, int K, int operands_regular);
int mmvae_gemm_f32_sq(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, unsigned flags, float* sq_partials,
                      int64_t sq_capacity, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Pre-split operands of the bf16x3 GEMMs (r3)
 * replaces: the same nn.Linear forward / backward dispatches as mmvae_gemm_f32 (components.py:276 and its autograd)
 *
 * The bf16x3 GEMMs split every fp32 operand element exactly into three bf16 pieces a = p0 + p1 + p2 (p0 = the top 16
 * bits of a, p1 = the top 16 bits of a - p0, p2 = a - p0 - p1).  Done inside the GEMM, each element is split once per
 * tile that reads it (10 x on the G-wide layers of the step) by vector instructions that take the matrix cores' issue
 * slots.  An operand whose producer writes the three planes once -- mmvae_split_planes_f32, or a fused producer
 * epilogue -- is moved global -> LDS by LDS-DMA with no vector work at all.
 *
 * (Exact while the residuals are normal numbers, |a| >= 2^-110; smaller values lose what does not fit the pieces' bit
 * patterns, an absolute error below 2^-126 -- the GEMMs' in-kernel split is the same function.)
 * Plane layout: planes[p][row][col], bf16 bit patterns, row-major like the fp32 matrix (`ld` and `plane_stride` in
 * bf16 elements, both multiples of 8; base 16-byte aligned; 3 * plane_stride * 2 < 4 GiB).  Rows the consumer reads
 * beyond the matrix (the zero slack rows of a weight-gradient GEMM whose K is padded to 32) must hold zeros.
 * A column count that is not a multiple of 8 (r5: the reference's 60 530 / 52 437 genes, human_only.yaml:90): `ld` is
 * at least the count rounded up to 8 and the columns in between hold zeros -- mmvae_split_planes_f32 writes them --
 * because a rows-contiguous planes operand is fetched in whole 16-byte groups.
 *
 * mmvae_gemm_planes_f32 = mmvae_gemm_f32 (sq_partials == NULL) or mmvae_gemm_f32_sq (sq_partials != NULL) with
 * optional planes per operand.  Kernels exist for TN with A, B or both operands pre-split (late r5: A alone -- the
 * first layer's weight gradient dY^T x with the batch x read as fp32), for NT / NN with a pre-split A
 * and (r5) for NN with a pre-split B (the weights) against an fp32 A;
 * an operand whose planes cannot be used (other combinations, shapes off the wave-specialised kernel: K % 32 != 0,
 * rows-contiguous leading dimension below the extent rounded up to 8, small outputs, MMVAE_GEMM_PRECISION_F32) is read from its fp32 form when that
 * pointer is non-NULL, otherwise the call fails with MMVAE_ERR_ARG.  Results are bit-identical to mmvae_gemm_f32 on
 * the fp32 operands (same products, same order).  mmvae_gemm_planes_supported: 1 when a launch of that shape with
 * those operands pre-split would read them from planes.
 * ------------------------------------------------------------------------------------------------------------ */
int mmvae_split_planes_f32(int rows, int cols, const float* src, int64_t ld_src, uint16_t* planes,
                           int64_t ld, int64_t plane_stride, mmvae_stream_t stream);
int mmvae_gemm_planes_f32(int layout, int M, int N, int K, float alpha, const float* A, int64_t lda, const uint16_t* Ap,
                          int64_t ldap, int64_t a_plane_stride, const float* B, int64_t ldb, const uint16_t* Bp,
                          int64_t ldbp, int64_t b_plane_stride, float* C, int64_t ldc, const float* bias, unsigned flags,
                          int splitk, float* workspace, size_t workspace_bytes, float* sq_partials, int64_t sq_capacity,
                          mmvae_stream_t stream);
int mmvae_gemm_planes_supported(int layout, int M, int N, int K, int splitk, int a_planes, int b_planes);
/* mmvae_decoder_recon_rows_colsum_f32 with planes on either side of it:
 *   dP_planes != NULL  the epilogue also writes the three bf16 planes of dP (G % 8 == 0; dP itself may then be NULL):
 *                      the weight-gradient and input-gradient GEMMs that consume dP read it pre-split.
 *   hp != NULL         the decoder's hidden activations h pre-split (h may then be NULL): runs the wave-specialised
 *                      kernel (256-row tiles; col_part keeps its [mmvae_recon_row_tiles(rows)][G] layout, one partial
 *                      per 128-row half) when the shape fills the chip, otherwise falls back to the fp32 h if given.
 *                      (Measured slower than the two-workgroup kernel on fp32 h at C2: its epilogue is done by 4 of
 *                      the 8 waves; kept for shapes where the main loop dominates.)
 * Same products in the same order as the fp32 form. */
int mmvae_decoder_recon_planes_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh, const uint16_t* hp,
                                   int64_t ldhp, int64_t h_plane_stride, const float* W, int64_t ldw, const float* bias,
                                   const float* x, int64_t ldx, float* xhat, int64_t ldxhat, float* dP, int64_t lddp,
                                   uint16_t* dP_planes, int64_t lddpp, int64_t dp_plane_stride, float* se_part,
                                   float* col_part, mmvae_stream_t stream);
/* (r5) mmvae_decoder_recon_planes_f32 with the WEIGHTS pre-split as well (Wp [G][ldwp] planes of W; taken together with hp,
 * otherwise W is read).  A K-sample program reads every weight tile from rows / 256 row tiles (20 at BASELINE config 3)
 * and splits it there each time; with both operands pre-split the stagers only move planes (LDS-DMA, no vector work).
 * The caller re-splits W after every optimiser step (mmvae_split_planes_f32, or the piggy-backed split of a layer tail).
 * replaces: the decoder's last nn.Linear + ReLU + mse_loss(sum) of modules/vae.py:140-145 (as above).
 * Same products in the same order as the fp32 form: bit-identical outputs. */
int mmvae_decoder_recon_wplanes_f32(int rows, int x_rows, int G, int H, const float* h, int64_t ldh, const uint16_t* hp,
                                    int64_t ldhp, int64_t h_plane_stride, const float* W, int64_t ldw, const uint16_t* Wp,
                                    int64_t ldwp, int64_t w_plane_stride, const float* bias, const float* x, int64_t ldx,
                                    float* xhat, int64_t ldxhat, float* dP, int64_t lddp, float* se_part, float* col_part,
                                    mmvae_stream_t stream);
/* mmvae_fc_epilogue_fwd / _bwd that also write the three bf16 planes of their output (d_out / the final dz_out; N even):
 * the layer tails that produce an operand of a G-wide weight-gradient GEMM (the decoder's last hidden activations, the
 * gradient at the expert encoder's first layer) split it on the way out. */
int mmvae_fc_epilogue_fwd_planes(int B, int N, const float* in, int64_t ld_in, int n_slabs, const float* bias,
                                 const mmvae_bn_params* bn, int training, int relu, const uint8_t* keep_mask,
                                 float dropout_p, float* z_out, float* a_out, float* d_out, int64_t ld_out,
                                 float* save_mean, float* save_invstd, float* workspace, size_t workspace_bytes,
                                 uint16_t* d_planes, int64_t ldp, int64_t plane_stride, mmvae_stream_t stream);
/* mmvae_fc_epilogue_fwd with a piggy-backed pass: extra workgroups of its second launch split the unrelated fp32
 * matrix sp_src [sp_rows, sp_cols] into its bf16 planes (as mmvae_split_planes_f32).  The engine splits its input
 * batch this way beside the first layer's tail: as a launch of its own the pass sits on the critical path of the step. */
int mmvae_fc_epilogue_fwd_split(int B, int N, const float* in, int64_t ld_in, int n_slabs, const float* bias,
                                const mmvae_bn_params* bn, int training, int relu, const uint8_t* keep_mask,
                                float dropout_p, float* z_out, float* a_out, float* d_out, int64_t ld_out,
                                float* save_mean, float* save_invstd, float* workspace, size_t workspace_bytes,
                                int sp_rows, int sp_cols, const float* sp_src, int64_t sp_ld_src, uint16_t* sp_planes,
                                int64_t sp_ld, int64_t sp_plane_stride, mmvae_stream_t stream);
int mmvae_fc_epilogue_bwd_planes(int B, int N, const float* din, int64_t ld_in, int n_slabs, const float* addend,
                                 const float* addend_a, const float* row_scale, const uint8_t* keep_mask, float dropout_p,
                                 int relu, const float* a_act, const float* z, const float* gamma, const float* save_mean,
                                 const float* save_invstd, int has_bn, float* dz_out, int64_t ld_out, float* dbias,
                                 float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes,
                                 uint16_t* dz_planes, int64_t ldp, int64_t plane_stride, mmvae_stream_t stream);

/* Grouped launch of independent small GEMMs (same math and layouts as mmvae_gemm_f32, exact-f32 MFMA, 64x64 tiles,
 * unsplit, alpha / bias / relu / accumulate epilogue): ONE grid covers the tiles of every job.  Used for the
 * weight-gradient GEMMs of the core (<= 1024-wide) layers of a backward pass -- replaces the per-parameter
 * `addmm` backward dispatches of autograd (components.py:276) -- which are independent of each other and only feed
 * the optimiser.  Requirements per job (mmvae_gemm_batch_job_ok): 16-byte aligned A, B, C, bias; lda, ldb, ldc,
 * M, N, K multiples of 4.  mmvae_gemm_batch_prepare is a pure host function: it validates a HOST array of jobs and
 * fills first_block / n_blocks; the caller copies the array to the device once and passes that DEVICE array, the job
 * count and total_blocks to every mmvae_gemm_batch_f32 launch. */
typedef struct {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int64_t lda, ldb, ldc;
    int32_t layout, M, N, K;
    float alpha;
    uint32_t flags;      /* MMVAE_GEMM_RELU | MMVAE_GEMM_ACCUMULATE */
    int32_t first_block; /* filled by mmvae_gemm_batch_prepare */
    int32_t n_blocks;    /* filled by mmvae_gemm_batch_prepare */
} mmvae_gemm_job;
int mmvae_gemm_batch_job_ok(const mmvae_gemm_job* job);
int mmvae_gemm_batch_prepare(int n_jobs, mmvae_gemm_job* jobs, int* total_blocks);
int mmvae_gemm_batch_f32(int n_jobs, const mmvae_gemm_job* jobs_dev, int total_blocks, mmvae_stream_t stream);

/* CSR batch -> dense rows.  replaces: `x.to_dense()` on the `torch.sparse_csr` batches the datapipes
 * yield when `return_dense: false` (vae.py:140-141, cmmvae_model.py:166-167, data/local/cellxgene_datapipe.py:178-183).
 * nnz = stored elements (col_indices / values may be NULL when it is 0; row pointers are clamped to it).
 * crow_indices [B+1] and col_indices [nnz] are int64 (torch's CSR index dtype), values [nnz] fp32, out [B, ldo] fp32.
 * Column indices must be unique within a row (torch CSR invariant); indices outside [0, G) are dropped.
 * At ~10 % density the dense MFMA GEMM of the first layer beats a gather SpMM on this chip, so the CSR
 * path densifies straight into the step's input buffer: one pass, 4*B*G bytes written, 16*nnz bytes read. */
int mmvae_csr_to_dense_f32(int B, int G, int64_t nnz, const int64_t* crow_indices, const int64_t* col_indices,
                           const float* values, float* out, int64_t ldo, mmvae_stream_t stream);
/* Same for int32 index arrays -- what the reference's batches carry: torch.sparse_csr_tensor keeps the int32 indptr /
 * indices of the scipy slice it is built from (cellxgene_datapipe.py:178-183) -- a third less data on the wire. */
int mmvae_csr_to_dense_i32_f32(int B, int G, int64_t nnz, const int32_t* crow_indices, const int32_t* col_indices,
                               const float* values, float* out, int64_t ldo, mmvae_stream_t stream);

/* f1 measurement (NOT on the product path): the first layer's product straight from the CSR batch,
 *   y[B, N] = x_csr[B, G] . W^T (+ bias)      with Wt = W transposed, [G, ldwt >= N], N % 4 == 0, int32 indices.
 * replaces (would replace): `x.to_dense()` + nn.Linear of the expert encoder's first layer (vae.py:140-141,
 * components.py:276).  A gather of nnz rows of Wt (4 N bytes each); accumulation in stored order.  Measured against
 * the dense bf16x3 GEMM at the reference's widths and 5 / 10 % density in profiles/r2_sparse_input.txt: dense wins, so
 * the engine keeps densifying (mmvae_csr_to_dense_*); this entry point stays as the evidence and for wider matrices. */
int mmvae_csr_spmm_wt_i32_f32(int B, int N, int G, int64_t nnz, const int32_t* crow_indices, const int32_t* col_indices,
                              const float* values, const float* Wt, int64_t ldwt, const float* bias, float* y,
                              int64_t ldy, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Conditional layers: y[b] = W[c_b] x[b] + bias[c_b], the Linear of each cell's OWN condition.
 * replaces: ConditionalLayer.forward (components.py:365-413: per present condition index_select -> Linear ->
 * index_copy_) and its autograd, for conditional blocks that are one Linear (LayerNorm: mmvae_layernorm_fwd/bwd).
 * The condition blocks live in one parameter arena `params` (and `grads` of the same layout); w_off / b_off [C] are
 * DEVICE int64 element offsets of block c's weight [n_out, n_in] and bias [n_out]; cond [B] is the DEVICE int32
 * condition index of every cell.
 *   _fwd      one workgroup per cell; with `rows` (DEVICE int32 [B]: the cells sorted by condition, NULL = none) and
 *             n_in <= 256, one workgroup per 8 sorted cells, which reads a block shared by them once.  Same results.
 *   _bwd_dx   dx[b] = W[c_b]^T dy[b] (+= when `accumulate`); with `rows` (as for _fwd) 8 sorted cells that share their
 *             block are computed by one workgroup that reads the block once.
 *   _bwd_dw   dW[c] = sum_{b in c} dy[b] (x) x[b], db[c] = sum dy[b] for the conditions PRESENT in the batch.  `rows`
 *             (DEVICE int32 [B]) lists the cells sorted by condition, cells of a condition in batch order; the host
 *             cuts every present condition's range of `rows` into chunks of at most MMVAE_COND_DW_CHUNK cells:
 *             chunk i covers rows[chunk_beg[i] .. chunk_end[i]) and has chunk_dst[i] >= 0: the only chunk of condition
 *             chunk_dst[i], its gradient is written straight into `grads`; <= -2: one of several, its partial result
 *             goes to scratch slot -2 - chunk_dst[i] of `partials` (slot = n_in*n_out + n_out floats); -1: padding.
 *             Reduction r (n_red of them; red_cond[r] < 0: padding) sums slots red_slot[r] .. + red_n[r] - 1, in that
 *             order, into condition red_cond[r].  Fixed summation tree, no atomics: bitwise reproducible.  Absent
 *             conditions are not touched (their parameters have no gradient this step).  All index arrays are DEVICE
 *             int32; a captured program launches the fixed maxima (mmvae_amd/cond_tables.py) padded with -1.
 * ------------------------------------------------------------------------------------------------------------ */
#define MMVAE_COND_DW_CHUNK 32
int mmvae_cond_linear_fwd(int B, int n_in, int n_out, const float* x, int64_t ldx, const float* params,
                          const int64_t* w_off, const int64_t* b_off, const int32_t* cond, const int32_t* rows, float* y,
                          int64_t ldy, mmvae_stream_t stream);
int mmvae_cond_linear_bwd_dx(int B, int n_in, int n_out, const float* dy, int64_t lddy, const float* params,
                             const int64_t* w_off, const int32_t* cond, const int32_t* rows, float* dx, int64_t lddx,
                             int accumulate, mmvae_stream_t stream);
int mmvae_cond_linear_bwd_dw(int n_chunks, const int32_t* chunk_dst, const int32_t* chunk_beg, const int32_t* chunk_end,
                             const int32_t* rows, int n_in, int n_out, const float* dy, int64_t lddy, const float* x,
                             int64_t ldx, float* grads, const int64_t* w_off, const int64_t* b_off, int n_red,
                             const int32_t* red_cond, const int32_t* red_slot, const int32_t* red_n, float* partials,
                             mmvae_stream_t stream);
/* (ABI 9, r5) n_pos POSITIONS of a "parallel" selection order per launch -- ConditionalLayers.forward with
 * selection_order = ["parallel"] (components.py:586-631; configs/model/human_only.yaml:78-79): every conditional layer
 * reads the SAME input and the outputs are concatenated, so the layers are independent of each other.  Position j's index
 * tables start `tbl_stride` int32 elements behind position j - 1's (every pointer argument names position 0's array),
 * its output / output gradient is the column block `y_pos_stride` / `dy_pos_stride` floats further into rows of ldy /
 * lddy floats, its input `x_pos_stride` floats further (0: one input for all), its partial slots `part_pos_stride`
 * floats further.  _fwd_multi / _bwd_dw_multi: the same arithmetic per position as the single-position entry points
 * (same bits).  _bwd_dx_multi: dx[b] (+)= sum over the positions of W[c_j(b)]^T dy_j[b], one workgroup per cell, the
 * positions added in descending order (needs no sorted `rows`). */
int mmvae_cond_linear_fwd_multi(int n_pos, int B, int n_in, int n_out, const float* x, int64_t ldx, int64_t x_pos_stride,
                                const float* params, const int64_t* w_off, const int64_t* b_off, const int32_t* cond,
                                const int32_t* rows, int64_t tbl_stride, float* y, int64_t ldy, int64_t y_pos_stride,
                                mmvae_stream_t stream);
int mmvae_cond_linear_bwd_dx_multi(int n_pos, int B, int n_in, int n_out, const float* dy, int64_t lddy,
                                   int64_t dy_pos_stride, const float* params, const int64_t* w_off, const int32_t* cond,
                                   int64_t tbl_stride, float* dx, int64_t lddx, int accumulate, mmvae_stream_t stream);
int mmvae_cond_linear_bwd_dw_multi(int n_pos, int n_chunks, const int32_t* chunk_dst, const int32_t* chunk_beg,
                                   const int32_t* chunk_end, const int32_t* rows, int64_t tbl_stride, int n_in, int n_out,
                                   const float* dy, int64_t lddy, int64_t dy_pos_stride, const float* x, int64_t ldx,
                                   int64_t x_pos_stride, float* grads, const int64_t* w_off, const int64_t* b_off,
                                   int n_red, const int32_t* red_cond, const int32_t* red_slot, const int32_t* red_n,
                                   float* partials, int64_t part_pos_stride, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Row-owner adversary passes (ABI 6).  Replaces, per adversarial phase, the whole chain of
 * `CMMVAEModel.grf` (cmmvae_model.py:59-101): Adversarial.encoder (FCBlock of Linear -> ReLU -> Dropout layers WITHOUT
 * BatchNorm, components.py:638-674) -> every head -> CrossEntropyLoss(sum) (:54,85) -> its backward down to the hidden
 * representation, with the sign flip of GradientReversalFunction (components.py:879-899) -- two launches per phase for
 * ALL adversaries instead of 12-17 launches per adversary.
 *
 *   mmvae_adv_pass_f32   a workgroup owns 16 cells.  Forward launch: the encoder layers with the activations in LDS
 *       (exact-f32 MFMA, weights streamed from L2), then the stacked heads [Ct, width[L]] flash-style: per 16-class tile
 *       the logits (stored once to `logits`, -inf in a head's padding columns), a running max / sum of exponentials per
 *       (cell, head) and the running product softmax . W -- so that d(loss)/d(encoded) = gscale * (softmax . W - W[y])
 *       needs no second pass over the head weights.  The class tiles of a cell tile are split over `splits` workgroups
 *       (more parallelism than 16-cell tiles alone give); each leaves its partial (max, sum, product) in `partials`.
 *       Backward launch (the kernel boundary makes the partials visible: no ticket, no cache write-back): one workgroup
 *       per cell tile merges them in split order, writes lse / loss_rows, and runs the backward chain: dz[l] (gradient
 *       at layer l's pre-activation, for the weight-gradient launch) and, with `gx`, the reversed input gradient
 *       gx = -d(loss)/dx.  Bitwise reproducible.
 *   mmvae_adv_dw_f32     every weight / bias gradient of the pass in one launch: job = one Linear, gW[M,N] = dz^T . inp
 *       over the B cells, gb = column sums of dz (32 x 64 tiles, exact-f32 MFMA).  A heads job reads `logits` and
 *       forms dlogits = gscale * (exp(logit - lse) - onehot) while staging.  Every workgroup leaves the sum of squares
 *       of what it stored; the last one to finish (ticket) turns them into each optimiser's global gradient norm, clip
 *       coefficient, step count and bias corrections (mmvae_adam_prepare's arithmetic) -- no norm pass, no prepare
 *       launch.  The jobs of an optimiser must cover its whole gradient arena for that norm to be the arena's.  With
 *       `adv_jobs_dev` its first workgroup also sums the per-cell losses of every adversary (fp64, fixed order) into
 *       loss_each[h] / loss_total and stores the sum of total_scale * loss_total over the adversaries in *total_loss.
 *   mmvae_adam_step_multi   mmvae_adam_step over several optimisers' arenas in one launch.
 * Limits: <= 4 encoder layers, <= 8 heads, widths <= 1024 (LDS permitting: mmvae_adv_pass_plan), any B.
 * ------------------------------------------------------------------------------------------------------------ */
#define MMVAE_ADV_MAX_LAYERS 4
#define MMVAE_ADV_MAX_HEADS 8
typedef struct {
    const float* x;       /* [B, width[0]] the hidden representation the adversary reads */
    int64_t ldx;
    const float* W[MMVAE_ADV_MAX_LAYERS];      /* encoder layer l: [width[l+1], width[l]] row-major */
    const float* b[MMVAE_ADV_MAX_LAYERS];      /* [width[l+1]] or NULL */
    const uint8_t* mask[MMVAE_ADV_MAX_LAYERS]; /* dropout keep mask [B, width[l+1]] or NULL */
    float* act[MMVAE_ADV_MAX_LAYERS];          /* out [B, width[l+1]]: layer output (after dropout) */
    float* dz[MMVAE_ADV_MAX_LAYERS];           /* out [B, width[l+1]]: gradient at the layer's pre-activation */
    const float* Wh;      /* stacked heads [Ct, width[L]] */
    const float* bh;      /* [Ct] or NULL */
    const int64_t* labels; /* [H, B]; a label outside [0, classes[h]) contributes nothing (like ignore) */
    float* logits;        /* out [B, Ct] */
    float* lse;           /* out [H, B] log-sum-exp per cell and head */
    float* loss_rows;     /* out [H, B] */
    float* gx;            /* out [B, width[0]] = -d(loss)/dx, or NULL (discriminator phase: x is detached) */
    float* partials;      /* scratch, mmvae_adv_pass_plan's partial_floats */
    float* loss_each;     /* out [H] */
    float* loss_total;    /* out [1] */
    float* total_loss;    /* total_loss[0] = sum over the launch's adversaries of total_scale * loss_total[0], or NULL */
    float total_scale;
    float gscale;         /* scale of d(loss)/d(logits): 1 (discriminator) or adv_weight (generator) */
    float p_drop[MMVAE_ADV_MAX_LAYERS];
    int32_t relu[MMVAE_ADV_MAX_LAYERS];
    int32_t width[MMVAE_ADV_MAX_LAYERS + 1];
    int32_t n_layers, H, Ct, B;
    int32_t col[MMVAE_ADV_MAX_HEADS];     /* first column of head h in the stacked matrix (multiple of 4 when H > 1) */
    int32_t classes[MMVAE_ADV_MAX_HEADS];
    /* filled by mmvae_adv_pass_plan: class split s works on the 16-class tiles [seg_lo, seg_hi) of head h */
    int16_t seg_lo[8][MMVAE_ADV_MAX_HEADS], seg_hi[8][MMVAE_ADV_MAX_HEADS];
} mmvae_adv_job;
/* host-only: MMVAE_OK when the job's shape is supported.  *net in: 0 = choose (out: 1, 2, 4 or 8 >= ceil(width[L] / 16)),
 * or the tile count of the launch the job will share with wider jobs; out: the dynamic LDS bytes of a workgroup and the
 * floats of `partials` for `splits` (<= 8) class splits, both at that tile count; *fast: 1 when every width, Ct, ldx and
 * head start is a multiple of 4 and the pointers filled in so far are 16-byte aligned (the kernels' 16-byte loaders).
 * Also fills the job's seg_lo / seg_hi (which class tiles each split works on): plan every job before uploading it. */
int mmvae_adv_pass_plan(mmvae_adv_job* job_host, int splits, int* net, size_t* lds_bytes, int64_t* partial_floats,
                        int* fast);
/* jobs_dev: DEVICE array of n_jobs jobs with the same B; net / lds_bytes: the maxima of mmvae_adv_pass_plan over the
 * jobs; fast: 1 only if every job planned fast.  Two launches (forward, backward). */
int mmvae_adv_pass_f32(int n_jobs, const mmvae_adv_job* jobs_dev, int B, int splits, int net, int fast, size_t lds_bytes,
                       mmvae_stream_t stream);

typedef struct {
    const float* dz;      /* [B, M] gradient at the Linear's output -- or, with `lse`, the logits of mmvae_adv_pass_f32 */
    int64_t ld_dz;
    const float* inp;     /* [B, N] the Linear's input */
    int64_t ld_inp;
    float* gW;            /* out [M, N] */
    float* gb;            /* out [M] or NULL */
    const float* lse;     /* heads job: [H, B] (NULL: dz is a gradient) */
    const int64_t* labels; /* heads job: [H, B] */
    float gscale;
    int32_t M, N, B, H;
    int32_t opt;          /* index into the launch's optimiser table: which norm this gradient belongs to */
    int32_t col[MMVAE_ADV_MAX_HEADS], classes[MMVAE_ADV_MAX_HEADS];
    int32_t first_block, n_blocks; /* filled by mmvae_adv_dw_prepare */
} mmvae_adv_dw_job;
typedef struct {
    float* state;         /* the optimiser's state words (mmvae_adam_prepare) */
    float* norm_out;      /* the pre-clip norm is also stored here (a metrics word), or NULL */
    float max_norm, grad_scale, beta1, beta2;
    uint32_t flags;       /* MMVAE_PREPARE_*; 0: leave the state alone (norm_out is still written) */
    uint32_t reserved;
} mmvae_adv_opt;
/* host-only: fills first_block / n_blocks; *fast: 1 when every job's M, N and leading dimensions are multiples of 4 and its
 * operands 16-byte aligned (pass it on to the launch: the 16-byte loaders) */
int mmvae_adv_dw_prepare(int n_jobs, mmvae_adv_dw_job* jobs_host, int* total_blocks, int* fast);
/* partials: total_blocks floats; ticket: one zero word */
int mmvae_adv_dw_f32(int n_jobs, const mmvae_adv_dw_job* jobs_dev, int total_blocks, int n_opts,
                     const mmvae_adv_opt* opts_dev, float* partials, uint32_t* ticket, int n_adv,
                     const mmvae_adv_job* adv_jobs_dev, int fast, mmvae_stream_t stream);

typedef struct {
    float *p, *g, *m, *v;
    const float* state;
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay, grad_scale;
} mmvae_adam_arena;
int mmvae_adam_step_multi(int n_arenas, const mmvae_adam_arena* arenas_dev, int64_t max_n, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Sparse weight gradient of the first encoder layer (ABI 6).  Replaces the autograd of nn.Linear
 * (components.py:276) for the layer that reads the batch: dW[M, G] = dY[B, M]^T . x[B, G] with x ~5-20 % populated
 * (cellxgene_datapipe.py:169-193 yields CSR; the bench's synthetic batches store 19 %).
 *   mmvae_ell_from_dense_f32   the dense batch -> gene-major ELL in one pass: gene g owns slots [g * cap, g * cap + cnt[g])
 *                              of rows (64 * cell index, int32, ascending) / vals, zero-padded to a multiple of 8 entries;
 *                              cap >= B, a multiple of 8; rows / vals 16-byte aligned.
 *   mmvae_dw_sparse_ell_f32    a workgroup holds a 64-output slice of dY in LDS (B * 256 bytes <= 140 KB), a wave owns a
 *                              gene, lane = output: one LDS read + one FMA per stored entry; fp32 FMA chain in cell
 *                              order (bitwise reproducible).  ldw >= G.
 * ------------------------------------------------------------------------------------------------------------ */
int mmvae_ell_from_dense_f32(int B, int G, const float* x, int64_t ldx, int cap, int32_t* rows, float* vals, int32_t* cnt,
                             mmvae_stream_t stream);
int mmvae_dw_sparse_ell_f32(int B, int G, int M, const float* dY, int64_t ldy, const int32_t* rows, const float* vals,
                            const int32_t* cnt, int cap, float* dW, int64_t ldw, mmvae_stream_t stream);

/* Diagnostics: `workgroups` workgroups that each hold `lds_bytes` of LDS and spin for `micros` microseconds -- a stand-in
 * for a collective occupying workgroup slots beside the step (bench.py --sim-comm). */
int mmvae_debug_occupy(int workgroups, int lds_bytes, int micros, float* sink, mmvae_stream_t stream);
/* Diagnostics: a marker launch that writes the device's 100 MHz wall clock into buf[slot] -- milestones of a captured
 * program as it runs without a tracer attached (MMVAE_STAMPS=1: the engine places them; tools/stamps_timeline.py). */
int mmvae_debug_stamp(long long* buf, int slot, mmvae_stream_t stream);

/* (ABI 10) Per-step host tables into device memory by a kernel: dst[0..n_words) = host_src[0..n_words) (32-bit words),
 * host_src PAGE-LOCKED, device-mapped host memory (hipHostMalloc; MMVAE_ERR_ARG for anything hipHostGetDevicePointer
 * refuses), read in place over the host link.  The reference builds its per-condition masks on the host and lets torch
 * copy them (components.py:365-413); a hipMemcpyAsync behind a captured program makes the host wait for that program on
 * this runtime (SDMA path), a kernel launch does not.  The caller keeps host_src unchanged until the launch has run. */
int mmvae_upload_words(int64_t n_words, const void* host_src, void* dst, mmvae_stream_t stream);

/* Small utilities used by the step engine: y = alpha*x (+ y), fill. */
int mmvae_axpby(int64_t n, float alpha, const float* x, float beta, float* y, mmvae_stream_t stream);
int mmvae_scale_rows(int B, int N, const float* x, int64_t ldx, const float* row_scale, float* y, int64_t ldy,
                     mmvae_stream_t stream);
/* Row-weighted column sums of a wide matrix in one read-only pass (ABI 7): partials[chunk][c] = sum over the rows of the
 * chunk (mmvae_weighted_colsum_chunks(B) chunks of 256 rows) of row_weight[r] * x[r][c]; the caller sums the chunks
 * (mmvae_sum_parts_batch).  16-byte accesses when x is 16-byte aligned and ldx a multiple of 4.  The K-sample ELBO's decoder-bias gradient
 * w^T dP (the softmax weights of the bound times the reconstruction gradient; the reference has no K: SURVEY 8 a7)
 * without scaling the [K B, G] matrix in place. */
int mmvae_weighted_colsum_chunks(int B);
int mmvae_weighted_colsum_f32(int B, int N, const float* x, int64_t ldx, const float* row_weight, float* partials,
                              mmvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMVAE_HIP_H */
