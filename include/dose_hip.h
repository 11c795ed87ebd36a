/* dose_hip.h -- C ABI of libdose_hip.so, the MI355X (gfx950) kernel library behind the
 * DOSE-PYFER / OAR-TRANSEG nn.Module surface.
 *
 * The reference (GhTara/Dose_Prediction) has no FFI: its hot path is the chain of torch ops issued
 * by the nn.Modules in DosePrediction/Models/Networks/{dose_pyfer,c3d}.py,
 * OARSegmentation/Models/Nets/{base_blocks,blocks_MDUNet}.py and the MONAI blocks they import.
 * Every entry point below replaces one torch op (forward or a backward piece) on that path; the
 * "replaces:" line cites the reference call site.  dose_prediction_amd/ops.py binds these with
 * ctypes; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers are DEVICE pointers; `stream` is a hipStream_t passed as void*.
 *   - activations are NDHWC ("voxel rows"): element (n,d,h,w,c) at ((n*D+d)*H+h)*W+w)*ld + c, where
 *     `ld` >= C is the row pitch in elements (so a tensor may be a channel slice of a wider buffer).
 *   - dtype: DP_F32 (0), DP_BF16 (1) or DP_F16 (2) selects the storage type T of activations / packed weights.
 *     Accumulation and statistics are always fp32 (fp64 in the tiny finalize kernels).
 *     DP_F32 uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) -- the parity mode;
 *     DP_BF16 uses v_mfma_f32_16x16x32_bf16 -- the benchmark mode;
 *     DP_F16 uses v_mfma_f32_16x16x32_f16 (BASELINE.json configs[4] trains in fp16).
 *     DP_X3 (3), accepted by the tiled convolutions only: the "fp32x3" mode.  Activations live in HBM as fp32; dp_split_rows writes
 *     each operand as bf16 halves hi = bf16(v), lo = bf16(v - hi) side by side ([hi | lo] channel blocks), the packed weights hold
 *     [w_hi | w_hi | w_lo], and the convolution runs the bf16 MFMA kernels over the 3x wider "virtual" input [x_hi | x_lo | x_hi]:
 *     x w ~ x_hi w_hi + x_lo w_hi + x_hi w_lo (relative error of a product ~1e-5 instead of 4e-3 for bf16 operands), fp32
 *     accumulators written as fp32 OUTPUT.  x / wq are bf16, y / y2 are float.  In a DP_X3 launch `Cin` = 3 cp (cp a multiple of
 *     16) is the contraction axis of the packed weights, `x` is the [x_hi | x_lo] tensor of 2 cp channels (x2 must be NULL): the
 *     kernels stage every x_hi chunk once and sweep it against both of its weight blocks.
 *     DP_X1 (4), accepted by dp_conv3d_tiled / dp_conv3d_tiled2 only: bf16 operands, fp32 OUTPUT, ONE product -- the data gradient
 *     of an fp32x3 convolution from gy_hi and w_hi alone (config.set_x3_dgrad_terms(1): the forward pass keeps its three products,
 *     the backward pass has the accuracy of the bf16 mode).  x is typically the hi half of a [hi | lo] tensor (ldx = 2 cp), Cin the
 *     real channel count, wq the ordinary bf16 pack.
 *   - every function returns 0 on success, non-zero on error; dp_last_error() gives the message.
 *   - no function allocates, frees or synchronises: workspaces are caller-provided.
 */
#ifndef DOSE_HIP_H
#define DOSE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { DP_F32 = 0, DP_BF16 = 1, DP_F16 = 2, DP_X3 = 3, DP_X1 = 4 };
/* DP_ACT_MISH_FAST (fp32 storage only, used by the fp32x3 mode): Mish through the hardware exp2 / rcp approximations (~1e-7
 * relative, far below that mode's 4e-6 operator error) instead of the correctly rounded expf / divisions of DP_ACT_MISH. */
enum { DP_ACT_NONE = 0, DP_ACT_RELU = 1, DP_ACT_LRELU = 2, DP_ACT_MISH = 3, DP_ACT_GELU = 4, DP_ACT_MISH_FAST = 5 };

const char* dp_last_error(void);
int dp_version(void);
/* Deterministic mode (process-wide, default off): 1 = every reduction that normally meets in fp32 atomics takes a fixed-order path, so two
 * runs on the same inputs are BIT-IDENTICAL, as the reference's CPU path is (the parity gates run under it): split-kd convolutions and the
 * tiled / K-along-H weight-gradient kernels accumulate one scratch slab per (kd, chunk) share / voxel share, which the finish / unpack pass
 * adds in share order (dp_conv3d_tiled_ws_elems and dp_conv3d_wgrad_tiled_ws_elems grow accordingly), dp_gemm_nt / dp_gemm_tn run unsplit
 * (callers that want the parallelism over K pass the shares as a batch and add the slabs themselves), the generic weight gradient runs one
 * wave per result tile, dp_trilinear_up2_bwd gathers instead of scattering; LayerNorm's dgamma / dbeta need dp_add_layernorm_bwd_det.
 * Queries made before a switch (workspace sizes) do not carry over: ask again. */
int dp_set_deterministic(int on);
int dp_get_deterministic(void);

/* ---- layout / data movement ---------------------------------------------------------------- */
/* replaces: the implicit NCDHW layout of every torch op; entry `input_.to(device)` network_trainer.py:188.
 * src fp32 [N][C][V] -> dst T [N][V][ld] (channels >= C up to cpad zero-filled). */
int dp_ncdhw_to_ndhwc(const float* src, void* dst, int N, int C, int64_t V, int ld, int cpad, int dtype, void* stream);
/* dst fp32 [N][C][V] <- src T [N][V][ld] (first C channels). accumulate!=0: dst += . */
int dp_ndhwc_to_ncdhw(const void* src, float* dst, int N, int C, int64_t V, int ld, int accumulate, int dtype, void* stream);
/* replaces: torch.cat(dim=1) (c3d.py:103-113, dose_pyfer.py:357, base_blocks.py:139, blocks_MDUNet.py:154):
 * copy rows x C channels from src (pitch lds) into dst (pitch ldd). */
int dp_copy_rows(const void* src, int lds, void* dst, int ldd, int64_t rows, int C, int dtype, void* stream);
/* torch.cat((a, b), dim=1) in one pass: dst row = [a row (Ca channels, Ca % 8 == 0) | b row (Cb channels)]; every
 * destination row is written whole.  replaces: dose_pyfer.py:357 cat((out_net_A, x)), c3d.py:103-113. */
int dp_cat2_rows(const void* a, int lda, int Ca, const void* b, int ldb, int Cb, void* dst, int ldd, int64_t rows, int dtype, void* stream);
/* fp32 <-> T conversion of a flat array (weights, small vectors). */
int dp_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* replaces: einops Rearrange "b c (h p1)(w p2)(d p3) -> b (h w d)(p1 p2 p3 c)" in MONAI
 * PatchEmbeddingBlock (call site dose_pyfer.py:55-67).  x NDHWC pitch ld -> out [B][ntok][p^3*C]. */
int dp_patchify(const void* x, void* out, int B, int S0, int S1, int S2, int C, int ld, int p, int dtype, void* stream);
/* the same into token rows of pitch ldo >= p^3 C elements (16-bit storage, 16-byte aligned, p*ld <= 1024): used by the fp32x3 mode to
 * fill the [hi | lo | hi] column blocks of the patch-embedding operand directly from the split voxel tensor. */
int dp_patchify_ld(const void* x, void* out, int64_t ldo, int B, int S0, int S1, int S2, int C, int ld, int p, int dtype, void* stream);
int dp_unpatchify(const void* gout, void* gx, int B, int S0, int S1, int S2, int C, int ld, int p, int dtype, void* stream);
/* replaces: the scatter half of nn.ConvTranspose3d(k2,s2) (base_blocks.py:118-127): src [N*D*H*W][8*C]
 * (column = ((a*2+b)*2+c)*C + co) -> dst NDHWC at (2d+a,2h+b,2w+c), pitch ldd.  unshuffle = inverse gather. */
int dp_pixel_shuffle2(const void* src, void* dst, int N, int D, int H, int W, int C, int ldd, int dtype, void* stream);
int dp_pixel_unshuffle2(const void* src, int lds, void* dst, int N, int D, int H, int W, int C, int dtype, void* stream);
/* replaces: F.interpolate(scale_factor=2, mode='trilinear', align_corners=True) c3d.py:36 (+ its backward).
 * dtype DP_X3 (forward only, fp32x3 mode): x is fp32, y receives the bf16 [hi | lo] operand of the x3 convolution that reads it -- hi at
 * channel c, lo at channel c + ldy / 2 of rows of ldy = 2 cp bf16 elements -- instead of an fp32 tensor that a split pass would re-read. */
int dp_trilinear_up2_fwd(const void* x, int ldx, void* y, int ldy, int N, int D, int H, int W, int C, int dtype, void* stream);
int dp_trilinear_up2_bwd(const void* gy, int ldgy, float* gx_f32, int N, int D, int H, int W, int C, int dtype, void* stream);
/* batched 2-D transpose: dst[b][c][r] = src[b][r][c]; two batch levels with independent strides. */
int dp_transpose(const void* src, int64_t lds, int64_t sb0, int64_t sb1, void* dst, int64_t ldd, int64_t db0, int64_t db1,
                 int rows, int cols, int nb0, int nb1, int dtype, void* stream);

/* ---- elementwise ----------------------------------------------------------------------------- */
/* y = a + b (b broadcast with period `period` elements; period==n for plain add).
 * replaces: residual adds in MONAI TransformerBlock, `x + position_embeddings`. */
int dp_add(const void* a, const void* b, void* y, int64_t n, int64_t period, int dtype, void* stream);
/* out[i] (fp32) = sum over b < B of g[b * per + i]: gradient of `x + position_embeddings` (MONAI PatchEmbeddingBlock) w.r.t. the
 * [1, N, hidden] parameter that is broadcast over the batch. */
int dp_sum_batch(const void* g, float* out, int B, int64_t per, int dtype, void* stream);
/* replaces: nn.GELU in MONAI MLPBlock; bwd: gx = gy * gelu'(x). */
int dp_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int dp_gelu_bwd(const void* x, const void* gy, void* gx, int64_t n, int dtype, void* stream);
/* replaces: softmax(q k^T * scale) in MONAI SABlock (rows of length `cols`); in place allowed.
 * bwd: gs = scale * p * (gp - sum(gp*p)). */
int dp_softmax_fwd(const void* s, void* p, int64_t rows, int cols, float scale, int dtype, void* stream);
int dp_softmax_bwd(const void* p, const void* gp, void* gs, int64_t rows, int cols, float scale, int dtype, void* stream);
/* Fused multi-head self-attention, 16-bit storage (DP_BF16 / DP_F16), head dim d in {64, 128}, any token count N.
 * replaces: the whole of MONAI SABlock between its qkv Linear and out_proj (einops split "b h (qkv l d) -> qkv b l h d",
 * softmax(q k^T * scale) v, merge "b h l d -> b l (h d)"); call sites dose_pyfer.py:55-67,129, oar_transeg.py:79-91,172
 * (SURVEY.md §8b names dp_attention_{fwd,bwd}).  q, k, v point at head 0 of batch 0 inside a [B][N][ld] row-major tensor
 * (for the packed qkv Linear output: ld = 3*heads*d, k = q + heads*d, v = q + 2*heads*d); head h starts h*d elements
 * further.  o, go: [B][N][ldo] with the heads merged; dq/dk/dv: [B][N][ldg] laid out like q/k/v.  lse: float
 * [B*heads*Np], Np = N rounded up to 32, base-2 log-sum-exp of the scaled scores, written by fwd and read by bwd; delta:
 * float [B*heads*Np] scratch of bwd (rowsum(dO o O)); both 16-byte aligned.  The N x N scores are never stored.  fp32 storage: use dp_gemm_nt + dp_softmax_*,
 * or -- dp_attention_fwd only -- dtype DP_X3: q / k / v / o are FP32 tensors, every operand is split into bf16 halves in registers and every
 * product takes three MFMAs (the fp32x3 mode's forward pass: ~1e-5 relative on the output); lse is not written (may be NULL). */
int dp_attention_fwd(const void* q, const void* k, const void* v, int64_t ld, void* o, int64_t ldo, float* lse, int B, int heads, int N,
                     int d, float scale, int dtype, void* stream);
int dp_attention_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* o, const void* go, int64_t ldo,
                     const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t ldg, int B, int heads, int N, int d,
                     float scale, int dtype, void* stream);
/* fp32 helpers for gradient buffers */
int dp_fill_f32(float* p, float v, int64_t n, void* stream);

/* ---- normalisation --------------------------------------------------------------------------- */
/* replaces: the reductions of nn.InstanceNorm3d / nn.BatchNorm3d (c3d.py:17, blocks_MDUNet.py:69,103).
 * Pass 1: per-block partial sums of x and x^2 per channel.  part: float [N][nblk][2][C]; returns nblk via
 * dp_stats_nblk(V).  */
int dp_stats_nblk(int64_t V);
int dp_stats_partial(const void* x, int ld, int N, int64_t V, int C, float* part, int dtype, void* stream);
/* Pass 2 (fp64 combine): mean/rstd per (n,c) [instance: groups==N] or per c [batch: groups==1, written
 * to row 0].  If running_mean != NULL (batch, training): running = (1-m)*running + m*stat (unbiased var). */
int dp_stats_finalize(const float* part, int N, int nblk, int C, int64_t V, int batch_mode, float eps,
                      float* mean, float* rstd, float* running_mean, float* running_var, float momentum, void* stream);
/* y = act( (x-mean)*rstd*gamma + beta + res ).  stat_stride_n = C for instance stats, 0 for batch/eval stats.
 * gamma/beta/res may be NULL.  replaces: norm + activation (+ residual add of MONAI UnetResBlock). */
int dp_norm_act_fwd(const void* x, int ldx, const float* mean, const float* rstd, int stat_stride_n,
                    const float* gamma, const float* beta, const void* res, int ldr, int act,
                    void* y, int ldy, int N, int64_t V, int C, int dtype, void* stream);
/* cat((IN(xa) -> act, IN(xb) -> act), channels) written in ONE pass (blocks_MDUNet.conv_3_1, 141-147): non-affine instance
 * statistics per source (mean / rstd: [N][Ca], [N][Cb]), y rows of Ca + Cb channels written whole. */
int dp_norm_act_cat_fwd(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                        const float* mean_b, const float* rstd_b, int Cb, int act, void* y, int ldy, int N, int64_t V, int dtype,
                        void* stream);
/* Backward of dp_norm_act_cat_fwd in one pass over whole gy rows.  part: float [N][nblk][2][Ca+Cb]; combine it with
 * dp_norm_bwd_finalize(part, N, nblk, Ca+Cb, 0, s1, s2, NULL, NULL), then apply (inv_count = 1/V) -> gxa [.., Ca], gxb [.., Cb]. */
int dp_norm_act_cat_bwd_partial(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                                const float* mean_b, const float* rstd_b, int Cb, const void* gy, int ldgy, int act, int N, int64_t V,
                                float* part, int dtype, void* stream);
int dp_norm_act_cat_bwd_apply(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                              const float* mean_b, const float* rstd_b, int Cb, const void* gy, int ldgy, int act, const float* s1,
                              const float* s2, float inv_count, void* gxa, int ldgxa, void* gxb, int ldgxb, int N, int64_t V, int dtype,
                              void* stream);
/* backward pass 1: g = gy*act'(z); partials of sum(g) and sum(g*xhat): part float [N][nblk][2][C]. */
int dp_norm_act_bwd_partial(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd,
                            int stat_stride_n, const float* gamma, const float* beta, const void* res, int ldr, int act,
                            int N, int64_t V, int C, float* part, int dtype, void* stream);
/* combine partials: s1,s2 float [groups][C] (groups = N instance, 1 batch); dgamma/dbeta if non-NULL: the sums over ALL samples,
 * OVERWRITTEN in both modes (fixed-order fp64 combination of the partial rows: no atomics, no zero-fill needed). */
int dp_norm_bwd_finalize(const float* part, int N, int nblk, int C, int batch_mode, float* s1, float* s2,
                         float* dgamma, float* dbeta, void* stream);
/* The partial pass with its finalize folded in: ONE launch instead of two.  Every block writes its partial row with device-scope
 * (write-through) stores, waits until they are acknowledged (workgroup-scope release fence + s_waitcnt vmcnt(0): NO L2 write-back /
 * invalidate is issued -- see csrc/norm.hip ticket_is_last) and draws a ticket of its statistics group (device-scope atomicAdd on a
 * zeroed counter); the block that draws the group's last ticket runs that group's finalize itself, reading the rows with device-scope
 * loads -- the same code in the same order of additions, so the results are bit-identical to the two-call form (and deterministic).
 * Counters: eager launches rotate through an internal ring; launches recorded by a stream capture get counters of their own that are
 * never handed out again (a graph replays them).  Return value 3 = nothing was launched, make the two calls: the folded form
 * is switched off (env DP_NO_TICKET=1, dp_ticket_enabled() == 0), cannot serve the case (dgamma / dbeta of an affine INSTANCE
 * normalisation over N > 1 samples need every group's rows), or has no counter to give (first use inside a capture, capture pool
 * exhausted).  Arguments as in the calls they replace; `part` is still the scratch.
 * replaces: the same reference lines as dp_stats_partial / dp_stats_finalize / dp_norm_act_bwd_partial / dp_norm_bwd_finalize. */
int dp_ticket_enabled(void);
int dp_stats_partial_finalize(const void* x, int ld, int N, int64_t V, int C, float* part, int batch_mode, float eps, float* mean,
                              float* rstd, float* running_mean, float* running_var, float momentum, int dtype, void* stream);
int dp_norm_act_bwd_partial_finalize(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd,
                                     int stat_stride_n, const float* gamma, const float* beta, const void* res, int ldr, int act,
                                     int N, int64_t V, int C, float* part, int batch_mode, float* s1, float* s2, float* dgamma,
                                     float* dbeta, int dtype, void* stream);
int dp_norm_act_cat_bwd_partial_finalize(const void* xa, int lda, const float* mean_a, const float* rstd_a, int Ca, const void* xb, int ldb,
                                         const float* mean_b, const float* rstd_b, int Cb, const void* gy, int ldgy, int act, int N,
                                         int64_t V, float* part, float* s1, float* s2, int dtype, void* stream);
/* backward pass 2: gx = gamma*rstd*(g - s1/M - xhat*s2/M) (use_stats) or gamma*rstd*g (eval BN);
 * gres = g (if non-NULL). M = count per statistics group. */
int dp_norm_act_bwd_apply(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd,
                          int stat_stride_n, const float* gamma, const float* beta, const void* res, int ldr, int act,
                          const float* s1, const float* s2, float inv_count, int use_stats,
                          void* gx, int ldgx, void* gres, int ldgres, int N, int64_t V, int C, int dtype, void* stream);
/* replaces: nn.LayerNorm(hidden) in MONAI TransformerBlock / ViT.norm.  rows x C, one wave per row. */
int dp_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     int64_t rows, int C, float eps, int dtype, void* stream);
int dp_layernorm_bwd(const void* x, const void* gy, const float* gamma, const float* mean, const float* rstd,
                     void* gx, float* dgamma, float* dbeta, int64_t rows, int C, int dtype, void* stream);
/* replaces: the residual add of a pre-norm MONAI TransformerBlock together with the LayerNorm that follows it
 * (x = x + attn(...); norm2(x) / next block's norm1(x) / ViT.norm(x)): sum = a + b (stored, rounded to T), y = LayerNorm(sum).
 * Backward: gx = gsum + LayerNorm'(gy) (gsum = the gradient reaching `sum` through the residual path, may be NULL; C <= 1024). */
int dp_add_layernorm_fwd(const void* a, const void* b, void* sum, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int64_t rows, int C, float eps, int dtype, void* stream);
int dp_add_layernorm_bwd(const void* x, const void* gy, const void* gsum, const float* gamma, const float* mean, const float* rstd,
                         void* gx, float* dgamma, float* dbeta, int64_t rows, int C, int dtype, void* stream);
/* the deterministic form of both LayerNorm backward passes (dp_set_deterministic; gsum may be NULL): per-block dgamma / dbeta partial rows
 * in `part` (dp_layernorm_bwd_parts(rows, C) x 2 x C floats), combined in a fixed order; dgamma / dbeta are OVERWRITTEN.  C <= 1024. */
int dp_layernorm_bwd_parts(int64_t rows, int C);
int dp_add_layernorm_bwd_det(const void* x, const void* gy, const void* gsum, const float* gamma, const float* mean, const float* rstd,
                             void* gx, float* part, float* dgamma, float* dbeta, int64_t rows, int C, int dtype, void* stream);


/* ---- matrix products (MFMA) ------------------------------------------------------------------- */
/* C[b0][b1][m][n] = alpha * sum_k A[..][m][k] * B[..][n][k] (+ bias[n]) ; "NT" GEMM, both operands k-contiguous.
 * replaces: nn.Linear (MONAI ViT), 1x1x1 nn.Conv3d (blocks_MDUNet.py:146, dose_pyfer.py:292,353), the GEMM half of
 * ConvTranspose3d k2s2, einsum QK^T / PV (MONAI SABlock).
 * out_f32: C is float (else T).  splitk>1 requires out_f32 and atomically ACCUMULATES into C (caller zero-fills). */
int dp_gemm_nt(const void* A, int64_t lda, int64_t sa0, int64_t sa1, const void* B, int64_t ldb, int64_t sb0, int64_t sb1,
               void* C, int64_t ldc, int64_t sc0, int64_t sc1, const float* bias, int M, int N, int K, int nb0, int nb1,
               float alpha, int out_f32, int splitk, int dtype, void* stream);
/* replaces: MONAI MLPBlock's linear1 -> GELU (mode 1) and the gradient through that GELU inside linear2's data gradient (mode 2):
 * mode 1: aux = A B^T + bias (pre-activation, storage type), C = GELU(aux);  mode 2: C = (A B^T) * GELU'(aux). */
int dp_gemm_nt_gelu(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, void* aux,
                    int64_t ldaux, int M, int N, int K, int mode, int dtype, void* stream);
/* C[m*ldc + n] (fp32) = sum_k A[k*lda + m] * B[k*ldb + n]  -- both operands k-major in memory: the weight gradient of
 * nn.Linear / ConvTranspose3d (autograd of MONAI's ViT blocks, base_blocks.py:118-127) dW = gy^T x with k = token or voxel
 * rows, without transposing either operand.  splitk > 1: K is split over blockIdx.z and C (pre-zeroed) is accumulated
 * atomically; splitk == 1 overwrites C. */
int dp_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, int splitk,
               int dtype, void* stream);
/* Grouped form: one launch for the weight (and bias) gradients of many Linear layers (MONAI TransformerBlock x num_layers and the
 * patch embedding; autograd computes them one by one).  table: device array of rows {const T* A; const T* B; float* C;
 * float* colsum (or NULL); int64 lda, ldb, ldc, M, N, K, tile0, tiles_m}: C[m][n] = sum_k A[k][m] B[k][n] (A = gy [rows][out],
 * B = x [rows][in], C = dW [out][in]); colsum[m] = sum_k A[k][m] (the bias gradient).  Tiles of T x T numbered from tile0 per
 * problem (n-tile-major, m fastest, low 32 bits of tiles_m = ceil(M / T)); T = 64, or 128 when bit 32 of tiles_m is set (only for M, N
 * multiples of 128, lda / ldb multiples of 8, 16-byte aligned operands); total_tiles = sum over problems of ceil(M/T) * ceil(N/T). */
int dp_gemm_tn_grouped(const void* table, int nproblems, int64_t total_tiles, int dtype, void* stream);

/* skinny row GEMM over voxel rows: y[v][co] = sum_ci x[v][ci] w[co*ldw + ci] (+bias).  Two kernels:
 *   - the matrix-core row kernel (k_pointwise_mfma: weights as the MFMA A operand in registers, a lane's B fragment one 16-byte
 *     global load, no LDS, 16-byte stores) for the shapes dp_rows_mfma_ok() accepts -- 16-bit storage, 16 <= Cin <= 256, Cout > 8,
 *     16-byte aligned rows;
 *   - otherwise an HBM-bound row stream on the vector ALU, Cin <= 64 and Cout <= 32 (every dtype).
 * replaces: nn.Conv3d k1 at the 128^3 / 64^3 levels (blocks_MDUNet.py:146, dose_pyfer.py:292,353) forward and their data gradient
 * with a transposed weight matrix; the ConvTranspose3d k2 s2 data gradient (base_blocks.py:118-127). */
int dp_pointwise_rows(const void* x, int ldx, const void* w, int ldw, const float* bias, void* y, int ldy, int64_t rows, int Cin, int Cout,
                      int dtype, void* stream);
int dp_rows_mfma_ok(int ldx, int ldw, int ldy, int Cin, int Cout, int dtype);
/* nn.ConvTranspose3d(kernel 2, stride 2, bias=False) forward in ONE launch (base_blocks.py:118-127; MONAI UnetrPrUpBlock): the row
 * GEMM [voxels x Cin] x [Cin x 8 Cout] with the 2x2x2 pixel shuffle in its store.  w: [(abc, co)][ldw] (dp_pack_multi kind 1 / 2 of the
 * permuted weight), y: NDHWC of the (2D, 2H, 2W) volume, pitch ldy.  Returns 3 WITHOUT launching when the shape is outside the kernel
 * (dp_rows_mfma_ok(ldx, ldw, ldy, Cin, 8 Cout, dtype), Cout % 8 == 0, N D H W < 2^24, 16-byte aligned x / y): the caller then runs
 * dp_gemm_nt + dp_pixel_shuffle2. */
int dp_tconv2x_fwd(const void* x, int ldx, const void* w, int ldw, void* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int dtype,
                   void* stream);
/* Weight + bias gradient of that skinny pointwise conv (the deep-supervision heads, dose_pyfer.py:330-350; autograd of
 * nn.Conv3d k=1): dw[co*s_co + ci] = sum_v gy[v][co] x[v][ci] (fp32), db[co] = sum_v gy[v][co] (db may be NULL).
 * Cin in {<=8, 16, 32, 64}, Cout <= 16.  ws: fp32 workspace of dp_pointwise_wgrad_ws_elems() elements (0 = unsupported
 * shape); per-block partials are added in a fixed order, so the result is run-to-run deterministic. */
int64_t dp_pointwise_wgrad_ws_elems(int64_t rows, int Cin, int Cout);
int dp_pointwise_wgrad_rows(const void* x, int ldx, const void* gy, int ldgy, float* dw, int s_co, float* db, float* ws,
                            int64_t rows, int Cin, int Cout, int dtype, void* stream);

/* ---- convolution ------------------------------------------------------------------------------ */
/* weight packing: torch fp32 [Cout][Cin][k^3] -> T [Cout][k^3][CinP] (mode 0, forward),
 * -> T [Cin][k^3][CoutP] (mode 1: transposed, for data-gradient "gather" form),
 * -> T [Cin][k^3 flipped][CoutP] (mode 2: transposed + spatially flipped: stride-1 data gradient as a forward conv).
 * CinP/CoutP = channel count rounded up to 8 (zero filled). */
int dp_pack_conv_weight(const float* w, void* dst, int Cout, int Cin, int taps, int mode, int dtype, void* stream);
/* replaces: nn.Conv3d k in {3,7} (any k), stride, padding, dilation (c3d.py:16, blocks_MDUNet.py:68,102,163,178).
 * Implicit GEMM on MFMA.  mode 0: y[v] = sum_tap,ci x[v*stride - pad + tap*dil][ci] * wp[co][tap][ci] (+bias)
 * mode 1 (data gradient of a strided conv): y[v] = sum over taps with (v + pad - tap*dil) % stride == 0 of
 *         x[(v + pad - tap*dil)/stride][ci] * wp[co][tap][ci].
 * x: [N][Di][Hi][Wi][ldx] (Cin channels), y: [N][Do][Ho][Wo][ldy] (Cout channels). CinP = roundup8(Cin). */
int dp_conv3d(const void* x, int ldx, const void* wp, const float* bias, void* y, int ldy,
              int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int Cin, int Cout,
              int k, int stride, int pad, int dil, int mode, int dtype, void* stream);
/* LDS-tiled fast path for the hot layers (stride 1, dilation 1, "same" padding, k in {3,7}; forward, and data gradient with
 * transposed_flipped weights).  dp_conv3d_tiled_weight_elems returns the packed-weight element count for a supported
 * shape and 0 when the shape must take dp_conv3d (NOT an error code).  Packed layout: see conv_tiled.hip. */
int dp_conv3d_tiled_weight_elems(int Cin, int Cout, int k, int stride, int pad, int dil, int W);
int dp_pack_conv_weight_tiled(const float* w, void* dst, int Cout, int Cin, int k, int transposed_flipped, int dtype, void* stream);
/* Which LDS-tiled kernel (and packed-weight layout) a shape takes: 0 none (generic dp_conv3d), 1 the 32x32x16 kernel
 * (dp_pack_conv_weight_tiled), 2 the 16x16x32 kernel for Cout <= 16 at W >= 96 (dp_pack_conv_weight_cc16; layout
 * [kd][ci chunk][kw pair][kh][co 16][32 = two horizontal taps x 16 ci]).  dp_conv3d_tiled_weight_elems covers both. */
int dp_conv3d_tiled_layout(int Cin, int Cout, int k, int stride, int pad, int dil, int W);
int dp_pack_conv_weight_cc16(const float* w, void* dst, int Cout, int Cin, int k, int transposed_flipped, int dtype, void* stream);
/* small volumes split the kd loop over blocks and accumulate in an fp32 scratch: dp_conv3d_tiled_ws_elems gives its size (0 = not needed). */
int dp_conv3d_tiled_ws_elems(int N, int D, int H, int W, int Cin, int Cout, int k);
int dp_conv3d_tiled(const void* x, int ldx, const void* wq, const float* bias, void* y, int ldy, float* ws, int N, int D, int H, int W,
                    int Cin, int Cout, int k, int dtype, void* stream);
/* "virtual concat" forms: torch.cat((a, b), dim=1) feeding (or, for a data gradient, fed by) a convolution is never
 * materialised.  Input channels >= csplit come from x2 (pitch ldx2); output channels >= osplit go to y2 (pitch ldy2); splits are
 * multiples of 8; NULL x2 / y2 = single operand.  replaces: torch.cat + nn.Conv3d (base_blocks.py:139, c3d.py:103-113). */
int dp_conv3d_tiled2(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias,
                     void* y, int ldy, void* y2, int ldy2, int osplit, float* ws, int N, int D, int H, int W,
                     int Cin, int Cout, int k, int dtype, void* stream);
/* The same convolution (one output tensor) that ALSO leaves the normalisation statistics of its output, taken from the fp32
 * accumulators: stat_part[((n * nblk + b) * 2 + {0: sum, 1: sum of squares}) * Cout + c] for the nblk =
 * dp_conv3d_tiled_stat_blocks(...) blocks of sample n; feed it to dp_stats_finalize(part, N, nblk, Cout, D*H*W, ...) in place of
 * dp_stats_partial's output.  replaces: the statistics pass of the nn.InstanceNorm3d / nn.BatchNorm3d that follows every
 * convolution of the path (c3d.py:15-19, blocks_MDUNet.py:67-74,101-108).  dp_conv3d_tiled_stat_blocks returns 0 when the shape
 * cannot produce statistics (split-kd volumes, unaligned output rows). */
int dp_conv3d_tiled_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int k, int ldy, int dtype);
int dp_conv3d_tiled_stats(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* wq, const float* bias,
                          void* y, int ldy, float* ws, float* stat_part, int N, int D, int H, int W, int Cin, int Cout, int k,
                          int dtype, void* stream);
int dp_conv3d_wgrad_tiled2(const void* x, int ldx, const void* x2, int ldx2, int csplit, const void* gy, int ldgy, float* dw, float* ws,
                           int N, int D, int H, int W, int Cin, int Cout, int k, int64_t s_co, int64_t s_ci, int64_t s_tap, int dtype, void* stream);
/* weight gradient (fp32, ACCUMULATES): for every tap t, co, ci:
 *   dw[co*s_co + ci*s_ci + t*s_tap] += sum_v gy[v][co + t*gy_tap_choff] * x[shift ? v*stride - pad + t*dil : v][ci]
 * replaces: autograd's conv/linear/conv-transpose weight gradients.  (v ranges over the gy voxels.) */
int dp_conv3d_wgrad(const void* x, int ldx, const void* gy, int ldgy, float* dw,
                    int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int Cin, int Cout,
                    int k, int stride, int pad, int dil, int shift, int gy_tap_choff,
                    int64_t s_co, int64_t s_ci, int64_t s_tap, int dtype, void* stream);

/* LDS-tiled weight gradient for the hot layers (stride 1, dilation 1, "same" padding, k in {3,7}).
 * dp_conv3d_wgrad_tiled_ws_elems: fp32 scratch elements needed, or 0 when the shape must take dp_conv3d_wgrad (NOT an error
 * code).  dp_conv3d_wgrad_tiled OVERWRITES dw[co*s_co + ci*s_ci + tap*s_tap] (every element of the
 * [Cout][Cin][taps] index space is written); neither dw nor ws need be initialised.  (k = 3: the scratch also holds the per-block
 * partial sums of the depth-marching kernel, 512 x 27 x 256 floats, so that no atomics are needed.) */
int dp_conv3d_wgrad_tiled_ws_elems(int Cin, int Cout, int k, int stride, int pad, int dil, int shift, int W);
/* 1 when dp_conv3d_wgrad_tiled(k = 1) takes the streaming row kernel for these rows (>= 32768 rows, 8 <= Cin <= 64, Cout in {8 .. 32}).
 * dtype DP_X1 (accepted by dp_conv3d_wgrad_tiled for exactly these shapes): x and gy are FP32 rows and dW = x_hi gy_hi -- both operands
 * rounded to bf16 between the global load and the LDS image: the fp32x3 mode's one-product weight gradient of the 1x1x1 mixers
 * (blocks_MDUNet.py:146) without a cast pass and without the exact-fp32 tiled kernel. */
int dp_conv3d_wgrad_rows_ok(int ldx, int ldgy, int64_t rows, int Cin, int Cout, int dtype);
int dp_conv3d_wgrad_tiled(const void* x, int ldx, const void* gy, int ldgy, float* dw, float* ws, int N, int D, int H, int W,
                          int Cin, int Cout, int k, int64_t s_co, int64_t s_ci, int64_t s_tap, int dtype, void* stream);

/* ---- optimizer ------------------------------------------------------------ */
/* replaces: optim.Adam(..., amsgrad=True).step() as built by NetworkTrainer.set_optimizer (network_trainer.py:120-125): one
 * launch over all parameter tensors.  table: device array of {float* p; const float* g; float* m; float* v; float* vmax;
 * int64_t n; void* cdst; int64_t ckind, K, cpp} -- cdst: a kernel-layout copy of the parameter written by the same pass (ckind 0
 * none, 1 / 2 plain bf16 / fp16 cast, 3 the fp32x3 Linear operand bf16 [n / K][3 cp] with cpp = cp | pattern << 32; the vector
 * path needs K % 4 == 0) --; chunk_t/chunk_i map each block to (tensor, chunk of dp_adam_chunk() elements); step = count after this update.
 * inv_grad_scale multiplies every gradient first (1 / loss scale when the backward pass ran on a scaled loss: fp16 storage).
 * An element whose gradient is not finite is left untouched (parameter and moments) and *found_inf (device int32, may be NULL;
 * the caller zeroes it) is set to 1: an fp16 overflow cannot poison the AMSGrad state. */
int dp_adam_chunk(void);
int dp_adam_multi(const void* table, const int32_t* chunk_t, const int32_t* chunk_i, int nchunks, double lr, double beta1, double beta2,
                  double eps, double weight_decay, double inv_grad_scale, int step, int amsgrad, int32_t* found_inf, void* stream);

/* Capturable variant: *step_dev (device int32, the number of updates done so far) is incremented by the call and the bias
 * corrections are computed from it on the device, so a captured HIP graph replays a correct Adam step. */
int dp_adam_multi_dev(const void* table, const int32_t* chunk_t, const int32_t* chunk_i, int nchunks, double lr, double beta1, double beta2,
                      double eps, double weight_decay, double inv_grad_scale, int32_t* step_dev, int amsgrad, int32_t* found_inf, void* stream);

/* ---- packed-weight refresh after optimizer.step() (network_trainer.py:213) ----------------------------------------------
 * The kernels read kernel-layout copies of the fp32 nn.Parameters; ONE launch rebuilds all copies of the parameters a step
 * changed.  table: device array of {const float* src; void* dst; int64_t kind, a, b, c, d, e}:
 *   kind 0 cast: dst[i] = src[i], i < a;                       kind 1 matrix: dst[r][c] = src[r*b + c] (a rows, pitch c, zero pad);
 *   kind 2 transposed matrix: dst[r][c] = src[c*a + r] (a rows, b valid columns, pitch c);
 *   kind 3 = dp_pack_conv_weight(Cout=a, Cin=b, taps=c, mode=d);
 *   kind 4 = dp_pack_conv_weight_tiled(Cout=a, Cin=b, k=c, NPAIR=d (dp_conv3d_tiled_npair), transposed_flipped=e);
 *   kind 6 = dp_pack_conv_weight_cc16(Cout=a, Cin=b, k=c, transposed_flipped=e);
 *   kind 5 ConvTranspose3d(k2,s2) weight [Cin=a][Cout=b][8] -> d == 0: [(abc,co)][pitch c over ci], d != 0: [ci][pitch c over (abc,co)].
 * chunk_t/chunk_i map each block to (table row, chunk); a chunk is dp_pack_chunk() destination elements, for kind 2 one
 * 64-row x 128-column destination tile (row-major tile index).  All destinations share the storage type `dtype`.
 * fp32x3 packs (dtype DP_BF16 destinations for DP_X3 launches): kind = base kind (1, 2, 4 or 6) | pattern << 8 | cp << 16 with cp > 0;
 * the contraction axis of the copy (input channels of kinds 4 / 6 -- Cin = b stays the REAL channel count --, the columns of kinds
 * 1 / 2, pitch c = 3 cp) is three blocks of cp virtual channels over the same zero-padded real ones; block p holds bf16(w) if
 * pattern bit p is 0, bf16(w - bf16(w)) if it is 1. */
int dp_pack_chunk(void);
/* number of chunks (blocks of dp_pack_multi) of one table row: kind 2: 64 x 128 transpose tiles; tiled convolution layouts (kinds 4, 6)
 * with 16-bit destinations: (4 destination columns) x (16-channel chunks), all taps each; everything else: dp_pack_chunk() elements. */
int64_t dp_pack_chunks(int64_t kind, int64_t a, int64_t b, int64_t c, int64_t d, int64_t e, int64_t dst_elems);
int dp_conv3d_tiled_npair(int Cout);
int dp_pack_multi(const void* table, const int32_t* chunk_t, const int32_t* chunk_i, int nchunks, int dtype, void* stream);

/* ---- fp32x3 mode: operand splitting (x3.hip) ---------------------------------------------------------------------------
 * replaces: nothing in the reference (which computes in fp32 throughout); these two feed the bf16 matrix-core
 * kernels with fp32-class operands.  dp_split_rows: dst bf16 [rows][parts * cp]; block p (p < parts <= 3) of a row holds, for the
 * fp32 source row cat(a[row][0:ca], b[row][0:cb]) zero-padded to cp channels (cp % 8 == 0; b may be NULL with cb == 0), hi = bf16(v)
 * when pattern bit p is 0 and lo = bf16(v - hi) when it is 1.  dp_x3_wgrad_combine: dw[co][ci][tap] = sum_{p < nblk}
 * S[co][p*cp + ci][tap] (the partial weight gradients x_hi gy_hi, x_lo gy_hi, x_hi gy_lo of the bf16 weight-gradient kernels). */
int dp_split_rows(const float* a, int lda, int ca, const float* b, int ldb, int cb, void* dst, int cp, int parts, int pattern,
                  int64_t rows, void* stream);
int dp_x3_wgrad_combine(const float* S, float* dw, int cout, int cin, int cp, int taps, int nblk, void* stream);
/* dp_norm_act_fwd / dp_norm_act_bwd_apply on fp32 tensors with the RESULT (y, resp. gx) written directly as the bf16 [hi | lo]
 * operand (row pitch 2 cp, C <= cp, C % 8 == 0) of the x3 convolution that consumes it: no fp32 copy, no dp_split_rows pass. */
int dp_norm_act_fwd_x3(const void* x, int ldx, const float* mean, const float* rstd, int ssn, const float* gamma, const float* beta,
                       const void* res, int ldr, int act, void* ys, int cp, int N, int64_t V, int C, void* stream);
int dp_norm_act_bwd_apply_x3(const void* x, int ldx, const void* gy, int ldgy, const float* mean, const float* rstd, int ssn,
                             const float* gamma, const float* beta, const void* res, int ldr, int act, const float* s1, const float* s2,
                             float inv_count, int use_stats, void* gxs, int cp, void* gres, int ldgres, int N, int64_t V, int C, void* stream);

/* ---- cascade glue ----------------------------------------------------------------------------- */
/* replaces: AsDiscrete(argmax=True, to_onehot=True) + channel concat (train_light_linked_model.py:157-167):
 * logits NDHWC [rows][ld] (C classes) -> one-hot of the arg-max (first max wins, as torch.argmax) for classes
 * 1..C-1 written to out[rows][ldo] channels [choff, choff+C-1); also writes the label (int32) if labels!=NULL. */
int dp_argmax_onehot(const void* logits, int ld, void* out, int ldo, int choff, int32_t* labels, int64_t rows, int C,
                     int dtype, void* stream);
/* the same with the one-hot destination in another storage type than the logits (fp32 logits of an fp32x3 segmentation pass into
 * the bf16 / fp16 staging buffer of the dose network). */
int dp_argmax_onehot2(const void* logits, int ld, int logits_dtype, void* out, int ldo, int out_dtype, int choff, int32_t* labels,
                      int64_t rows, int C, void* stream);

/* Gather for small-volume convolutions (c3d.py:16 at the 8^3 / 16^3 stages): col[row][tap*CinP + c] with CinP = Cin rounded
 * up to 8, row = output voxel (n, od, oh, ow), zero padding.  conv = dp_gemm_nt(col, weights packed [Cout][tap][CinP]). */
int dp_im2col3d(const void* x, int ldx, void* col, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int Cin, int k, int stride,
                int pad, int dil, int dtype, void* stream);

/* Scratch contract for the fp32 workspaces `ws` of dp_conv3d_tiled* (split-kd accumulation) and dp_conv3d_wgrad_tiled*:
 * 0 (default) = ws may hold anything, each call clears it with a memset launch; 1 = the caller passes an all-zero ws and
 * gets it back all zero (the finish / unpack kernels re-zero what they read), so a persistent scratch never needs a memset.
 * Process-wide; the Python host side keeps one persistent zeroed buffer per device and selects 1. */
int dp_scratch_contract(int zeroed);

/* ---- sliding-window stitching (MONAI sliding_window_inference, constant blending; train_light_linked_model.py:152-153) --
 * acc[n][z0+z][y0+y][x0+x][c] += win[z][y][x][c] (fp32, C channels dense), cnt[voxel] += 1 for one rz x ry x rx window of
 * image n; then out[row][c] = acc / cnt.  acc and cnt are zero-initialised by the caller. */
int dp_window_accumulate(const void* win, int ldw, float* acc, float* cnt, int n, int D, int H, int W, int rz, int ry, int rx,
                         int z0, int y0, int x0, int C, int dtype, void* stream);
int dp_window_normalize(const float* acc, const float* cnt, void* out, int ldo, int64_t rows, int C, int dtype, void* stream);

/* ---- loss and dose metrics on device ------------------------------------------------------
 * Masked L1 of DosePrediction/Train/loss.py:13-28,69-107 (Loss / GenLoss: mean |pred - gt| over possible_dose_mask > 0) and
 * the validation metric of train_light_pyfer.py:166-172 + Evaluate/evaluate_openKBP.py:42-48.  fp32 tensors of n elements.
 * out3 = { sum |p - g| over the mask, number of masked elements, their ratio (0 for an empty mask) }; ws holds
 * dp_masked_l1_ws_elems(n) floats.  postprocess != 0 first applies p = (mask < 1 || p < 0) ? 0 : p (dose score; multiply
 * the ratio by 70 for Gy).  Backward: gpred = gup[0] * sign(p - g) * (mask > 0) / max(out3[1], 1), gup on the device. */
int64_t dp_masked_l1_ws_elems(int64_t n);
int dp_masked_l1_fwd(const float* pred, const float* gt, const float* mask, int64_t n, float* ws, float* out3, int postprocess,
                     void* stream);
int dp_masked_l1_bwd(const float* pred, const float* gt, const float* mask, const float* out3, const float* gup, float* gpred,
                     int64_t n, void* stream);
/* GenLoss(huber=True) (Train/loss.py:53,100-103,112-115): nn.HuberLoss(reduction='mean', delta) over mask > 0: element
 * 0.5 d^2 for |d| < delta, delta (|d| - delta / 2) otherwise; same out3 / ws / backward conventions as dp_masked_l1_*. */
int dp_masked_huber_fwd(const float* pred, const float* gt, const float* mask, int64_t n, float delta, float* ws, float* out3, void* stream);
int dp_masked_huber_bwd(const float* pred, const float* gt, const float* mask, const float* out3, const float* gup, float* gpred,
                        int64_t n, float delta, void* stream);
/* Ground-truth pyramid of GenLoss (Train/loss.py:56-66, 88-97): dose resampled like F.interpolate(mode="trilinear",
 * align_corners=True), mask like mode="nearest-exact"; fp32 single-channel volumes [N][D][H][W]. */
int dp_resample_gt(const float* dose, const float* mask, float* out_dose, float* out_mask, int N, int Di, int Hi, int Wi, int Do, int Ho,
                   int Wo, void* stream);
/* replaces: monai.losses.DiceCELoss(to_onehot_y=True, softmax=True) -- OAR-TRANSEG's training / validation loss
 * (OARSegmentation/train_light_transeg.py:148,196,212): logits fp32 [B][C][V] (NCDHW, V = D H W), labels [B][V] class indices stored as
 * float32 (label_kind 0, what the reference's loader delivers), int64 (1), int32 (2) or uint8 (3); 2 <= C <= 16.
 *   loss = lambda_dice * mean_{b,c}(1 - (2 I + smooth_nr) / (G + P + smooth_dr)) + lambda_ce * mean_{b,v}(-log softmax[label])
 * with I = sum_v onehot p, G = sum_v onehot, P = sum_v p (MONAI 0.7.0 defaults: smooth 1e-5, lambdas 1, include_background, no squared /
 * jaccard / batch).  ws: dp_dice_ce_ws_elems floats; stats: dp_dice_ce_stats_elems floats = {loss, dice, ce, ...coefficients the backward
 * pass reads}.  Backward: glogits fp32 [B][C][V] = gup[0] * d loss / d logits (softmax recomputed; gup on the device). */
int64_t dp_dice_ce_ws_elems(int B, int C, int64_t V);
int64_t dp_dice_ce_stats_elems(int B, int C);
int dp_dice_ce_fwd(const float* logits, const void* labels, int label_kind, int B, int C, int64_t V, float smooth_nr, float smooth_dr,
                   float lambda_dice, float lambda_ce, float* ws, float* stats, void* stream);
int dp_dice_ce_bwd(const float* logits, const void* labels, int label_kind, int B, int C, int64_t V, const float* stats, const float* gup,
                   float* glogits, void* stream);
/* out = (mask < 1 || pred < 0) ? 0 : scale * pred   (train_light_pyfer.py:166-172, scale = 70 Gy) */
int dp_dose_postprocess(const float* pred, const float* mask, float* out, int64_t n, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
