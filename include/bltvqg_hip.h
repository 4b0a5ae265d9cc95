/* libbltvqg_hip.so — C ABI of the MI355X-native BLT-VQG training hot path.
 *
 * The reference (nihirv/blt-vqg) is pure Python on PyTorch and has no FFI of its own; the operator boundary this
 * library replaces is the set of torch operator call sites on the train step:
 *   models/iq.py:82-114            IQ.forward            -> bltvqg_engine_forward
 *   train_iq.py:81-103             calculate_losses      -> bltvqg_engine_loss_backward (losses fused with backward)
 *   train_iq.py:105-132 (+ Lightning backward / clip 5 / Adam, train_iq.py:260,372)
 *                                                        -> bltvqg_engine_loss_backward + bltvqg_engine_optimizer_step
 * and, one level down, the individual operators, exported for unit parity tests and for callers that want one op:
 *   nn.Linear / F.conv2d           (transformer_layers.py:453-456,400-408; encoder_cnn.py:20,33)  -> bltvqg_gemm / bltvqg_conv2d
 *   nn.LayerNorm                   (transformer_layers.py:134,202,256-257,320-322)                -> bltvqg_layernorm_fwd/bwd
 *   MultiHeadAttention core        (transformer_layers.py:494-526)                                -> bltvqg_attn_fwd/bwd
 *   nn.CrossEntropyLoss(ignore 0)  (train_iq.py:54-55,82-83,92-94)                                -> bltvqg_ce_fwd_bwd / bltvqg_bow_ce_fwd_bwd
 *   Latent reparam + gaussian_kld  (transformer_layers.py:41-59,536-540)                          -> bltvqg_latent_fwd/bwd
 *   BatchNorm2d(train) statistics  (encoder_cnn.py:33)                                            -> bltvqg_bn_finalize / bltvqg_bn_apply
 *   Adam + clip_grad_norm_         (train_iq.py:260,372)                                          -> bltvqg_adam_step
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the library borrows it for the duration of the
 *     enqueue, allocates nothing persistent on the device and frees nothing;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no hidden synchronisation;
 *   - return value 0 = success, negative = error; bltvqg_last_error_string() describes the last error of the thread;
 *   - dtype: 0 = fp32 (exact-fp32 MFMA; parity mode), 1 = bf16 storage with fp32 accumulation (performance mode).
 */
#ifndef BLTVQG_HIP_H
#define BLTVQG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLTVQG_F32 0
#define BLTVQG_BF16 1

int bltvqg_version(void);
const char* bltvqg_last_error_string(void);
/* tuning switches for A/B benchmarks: key 0 = disable the LDS-DMA GEMM ring (value 1), key 1 = force a GEMM tile (64/128/12864) */
void bltvqg_debug_set(int key, int value);
int bltvqg_debug_get(int key);      /* keys 0..31; bench.py echoes every non-zero key in its JSON line.  Round 4: 23 dead-work A/B, 24 LayerNorm-backward
                                      * forms, 25 = 1 LayerNorm fold off (2: off for head-padded widths only), 27 = 1 per-pixel img_pack, 28 = 1 decode: cross K/V per step, 29 = 1 decode: one
                                      * full decoder pass per step (the round-2 form) */
/* 1 only in the ablation build (make -C blt-vqg_amd/csrc ablate -> libbltvqg_hip_ablate.so, -DBLT_ABLATE): there debug keys 14 (skip the
 * grouped weight-gradient launches) and 15 (skip the conv stack) exist as TIMING ablations with wrong results.  The shipped library has
 * no switch that skips work. */
int bltvqg_build_has_ablations(void);

/* ---------------- operator-level entry points ---------------- */

/* C[M,N] = epilogue(sum_k A^[m,k] * B^[n,k]).  transA/transB: 0 = k-contiguous storage, 1 = m/n-contiguous storage.
 * Optional epilogue terms (NULL / 0 to disable), applied in this order: +bias[n] (fp32), ReLU, dropout(p, seed, stream_id),
 * *(maskY != 0)*mask_scale, +R[m,n], +old C (accumulate).  out_f32: store C as fp32 even when dtype is bf16.
 * force_tile: 0 = heuristic, 64 / 128 / 12864 (= 128x64).  split_k > 0 allows up to that many K-slices for the weight-gradient
 * form (transA = transB = 1, fp32 output, no epilogue terms): partial tiles are ADDED to C with fp32 atomics. */
int bltvqg_gemm(int dtype, const void* A, int lda, int transA, const void* B, int ldb, int transB, void* C, int ldc,
                int M, int N, int K, const float* bias, int relu, float drop_p, uint64_t seed, uint32_t stream_id,
                const void* maskY, int ldm, float mask_scale, const void* R, int ldr, int accumulate, int out_f32,
                int force_tile, int split_k, void* stream);

/* bf16 Linear forward / input gradient C = epilogue(A[M,K] B[N,K]^T) on the planned-tile kernel (gemm2.hip) with EVERY epilogue term
 * exposed: v = acc + bias[n] + rowtab[rowidx[m]][n]; relu; dropout(seed, stream_id); (maskY[m,n] != 0) * mask_scale; -> C2 (optional
 * copy); + R[m,n]; (+ old C); -> C.  tile_m x tile_n selects one of the compiled tile shapes (0, 0 = the planner's choice for this
 * problem; tile_m < 0 = the round-1 64x64 ring kernel, for A/B comparisons). */
int bltvqg_gemm_ex(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, const float* rowtab,
                   const int32_t* rowidx, int ldt, int relu, float drop_p, uint64_t seed, uint32_t stream_id, const void* maskY, int ldm, float mask_scale,
                   void* C2, int ldc2, const void* R, int ldr, int accumulate, int tile_m, int tile_n, void* stream);
/* ---- LayerNorm folded into the Linear that consumes it (round 4).  Every pre-LayerNorm inside the reference's Encoder / Decoder layers
 * feeds exactly one Linear (transformer_layers.py:260-262 -> q|k|v, :271-273 -> FFN layer 0, :326-328, :340-342 -> cross-attention query,
 * :356-358 -> FFN layer 0), so  LN(x) W^T + b = rstd_m (x W'^T - mean_m s_n) + c_n  with W' = W diag(gamma), s_n = sum_k W'[n,k],
 * c_n = sum_k beta[k] W[n,k] + b_n: the normalisation moves into the GEMM epilogue, the row sums come from the epilogue of the GEMM that
 * PRODUCED x, and the layernorm launch between the two disappears.
 * bltvqg_gemm_rowstat: bltvqg_gemm_ex's Linear (bias, relu, dropout, second output C2, residual R) that also STORES, per result row m and
 *   group g of 64 columns, the {sum, sum of squares} of those columns AS STORED (bf16-rounded) into out_stat[m][g][0..1] (stat_slots slots
 *   per row; plain stores, no atomics; one slot per 64 columns WHATEVER the tile shape: the consumer adds the slots in order, so the folded
 *   LayerNorm is bit-reproducible and does not depend on tile shapes or on how many rows a launch covers — an incremental pass over some rows
 *   reproduces the full pass).  bltvqg_gemm_rowstat_parts(M, N, K, tile_n) = how many slots that launch writes = ceil(N / 64).
 * bltvqg_ln_fold_prepare: W'[N,K] (bf16), s[N] (of the ROUNDED W'), c[N] (fp32 W; bias may be NULL) from the fp32 parameters.
 * bltvqg_linear_ln_folded: Y = [dropout][relu](rstd_m (X Wf^T - mean_m s_n) + c_n) with mean_m / rstd_m from row_stat[m] = {sum, sum of
 *   squares} of row m of X over its K features, given as stat_parts partial sums in the first slots of row_stat[m][stat_slots][2]
 *   (biased variance, eps inside the square root, as nn.LayerNorm); mean / rstd (both or neither) receive the statistics (the
 *   LayerNorm's backward reads them). */
int bltvqg_gemm_rowstat(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, int relu, float drop_p,
                        uint64_t seed, uint32_t stream_id, void* C2, int ldc2, const void* R, int ldr, float* out_stat, int stat_slots, int tile_m, int tile_n,
                        void* stream);
int bltvqg_gemm_rowstat_parts(int M, int N, int K, int tile_n);
int bltvqg_ln_fold_prepare(const float* W, int N, int K, const float* gamma, const float* beta, const float* bias, void* Wf_bf16, float* fold_s,
                           float* fold_c, void* stream);
int bltvqg_linear_ln_folded(const void* X, int ldx, const void* Wf, int ldw, void* Y, int ldy, int M, int N, int K, const float* fold_s, const float* fold_c,
                            const float* row_stat, int stat_slots, int stat_parts, float* mean, float* rstd, float eps, int relu, float drop_p, uint64_t seed,
                            uint32_t stream_id, int tile_m, int tile_n, void* stream);
/* Weight gradients of n Linear layers in ONE launch (bf16 operands, fp32 results): dW_i[N_i, K_i] = dY_i[rows_i, N_i]^T X_i[rows_i, K_i]
 * and dbias_i[N_i] = column sums of dY_i (dbias_i may be NULL).  Results are STORED unless the launch is short of tiles and slices
 * the contraction (then they are atomically added: dW / dbias must be zero on entry, as the engine's gradient buffer is).  The
 * pointer / size arrays are HOST arrays; table_dev is caller-owned device scratch for the problem table (128 + 80 * n bytes). */
int bltvqg_linear_wgrad_group(int n, const void* const* dY, const int32_t* ldy, const void* const* X, const int32_t* ldx, float* const* dW,
                              const int32_t* ldw, float* const* dbias, const int32_t* rows, const int32_t* N, const int32_t* K, void* table_dev,
                              int64_t table_bytes, void* stream);
/* Backward of y = x W^T + b w.r.t. the parameters (autograd of nn.Linear, transformer_layers.py:453-456,400-408):
 * dW[N,K] += dY[rows,N]^T X[rows,K] and, when dbias is non-null, dbias[N] += column sums of dY — both fp32, in ONE launch (the bias
 * gradient falls out of the staging registers of the dY operand).  split_k > 0: up to that many slices of `rows`, fp32 atomics. */
int bltvqg_linear_wgrad(int dtype, const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw, float* dbias, int rows,
                        int N, int K, int split_k, void* stream);

/* NHWC implicit-GEMM convolution y[N,Ho,Wo,Cout] = conv(x[N,Hi,Wi,Cin], w[Cout,KH,KW,Cin]); Cin a power of two >= 8 (bf16)
 * / 4 (fp32).  stat_sum/stat_sq (optional): per-half-tile column partial sums, bltvqg_conv2d_stat_rows() rows of Cout. */
int bltvqg_conv2d(int dtype, const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW,
                  int stride, int pad, float* stat_sum, float* stat_sq, void* stream);
int bltvqg_conv2d_stat_rows(int N, int Hi, int Wi, int Cout, int KH, int KW, int stride, int pad);
/* ---- padded-pitch (PP) activations: the layout of the ResNet-18 stack between its 3x3 convolutions (encoder_cnn.py:17,33) ----
 * [N][H+1][W+1][C]: one zero pixel after every image row, one zero row after every image, so that in linear pixel order all four
 * neighbours of a border pixel are zeros and a 3x3 stride-1 window of a run of pixels is a run of pixels.  A PP buffer holds
 * bltvqg_pp_guard_front() zero pixels, then bltvqg_pp_pixels(N,H,W) positions, then bltvqg_pp_guard_tail() zero pixels; the functions
 * below take the address of position 0.  The guards and pad positions must be zero on input; bn_apply_pp / bn_relu_maxpool_pp keep
 * them zero on output, conv outputs leave arbitrary values at pad positions (excluded from the statistics). */
int64_t bltvqg_pp_pixels(int N, int H, int W);
int bltvqg_pp_guard_front(void);
int bltvqg_pp_guard_tail(void);
/* bf16 3x3 stride-1 pad-1 convolution, PP in -> PP out, Cin and Cout multiples of 64, W <= 62: the input patch of 128 output
 * positions is staged in LDS once per 64-channel slice and serves all nine taps (csrc/conv_pp.hip).  w [Cout,3,3,Cin]. */
int bltvqg_conv3x3_pp(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, float* stat_sum, float* stat_sq,
                      void* stream);
/* The same convolution reading the RAW output x_raw of the previous convolution (PP, Cin channels): that convolution's train-mode
 * BatchNorm + ReLU, max(x * in_scale[c] + in_shift[c], 0) with zeros at pad positions (encoder_cnn.py:17, torchvision BasicBlock
 * bn1 + relu), is applied to the input patch in LDS — y == bltvqg_conv3x3_pp(bltvqg_bn_apply_pp(x_raw, relu)), without that pass. */
int bltvqg_conv3x3_pp_bn_relu_in(const void* x_raw, const float* in_scale, const float* in_shift, const void* w, void* y, int N, int H, int W, int Cin,
                                 int Cout, float* stat_sum, float* stat_sq, void* stream);
int bltvqg_conv3x3_pp_stat_rows(int N, int H, int W);
/* general implicit-GEMM convolution (as bltvqg_conv2d) with a PP input and / or PP output (the stride-2 and 1x1 convolutions of the
 * stack, and every convolution in fp32 mode); PP output rows at pad positions are written as zeros */
int bltvqg_conv2d_pp(int dtype, const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW,
                     int stride, int pad, int in_pp, int out_pp, float* stat_sum, float* stat_sq, void* stream);
int bltvqg_conv2d_pp_stat_rows(int dtype, int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int out_pp);
int bltvqg_bn_apply_pp(int dtype, const void* x, const float* scale, const float* shift, const void* res, void* y, int N, int H, int W,
                       int C, int relu, void* stream);
int bltvqg_bn_relu_maxpool_pp(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi, int C,
                              void* stream);
int bltvqg_avgpool_pp(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream);

/* NCHW fp32 -> NHWC [N,Hp,Wp,Cpad] with the image at (pad_top, pad_left) and zeros elsewhere */
int bltvqg_img_pack(int dtype, const float* nchw, void* nhwc, int N, int C, int H, int W, int Cpad, int pad_top, int pad_left, int Hp,
                    int Wp, void* stream);
/* [Cout,Cin,KH,KW] fp32 -> [Cout,KH,KWpad,Cpad] */
int bltvqg_conv_pack_w(int dtype, const float* w, void* out, int Cout, int Cin, int KH, int KW, int Cpad, int KWpad, void* stream);
/* ResNet stem: 7x7 stride-2 pad-3 conv of a zero-bordered NHWC4 image [N,Hp,Wp,4] (image at (3,3), Wp even, Wp >= W+7) with weights
 * packed [Cout,7,8,4]; y [N,Ho,Wo,Cout].  bf16 with Cout = 64, Ho % 8 == 0, Wo % 16 == 0 runs the LDS-patch kernel (the 21x38-pixel
 * input patch of an 8x16 output tile and the whole filter are staged once, csrc/conv_pp.hip), everything else the implicit GEMM.
 * stat_sum / stat_sq: bltvqg_conv_stem_stat_rows() rows of Cout, an upper bound for either kernel — zero them before the call, the
 * rows are summed. */
int bltvqg_conv_stem(int dtype, const void* x_padded, const void* w, void* y, int N, int H, int W, int Hp, int Wp, int Cout,
                     float* stat_sum, float* stat_sq, void* stream);
int bltvqg_conv_stem_stat_rows(int N, int H, int W, int Cout);
/* Stem + MaxPool2d(3, 2, 1) in ONE launch (torchvision resnet conv1 -> bn1 -> relu -> maxpool, encoder_cnn.py:17,33).  relu(bn(x)) is monotone
 * in x with the sign of the BatchNorm scale = the sign of gamma, so the kernel writes the pooling-window EXTREMUM of the raw convolution
 * output (max where gamma[c] >= 0, min where gamma[c] < 0) into the pooled padded-pitch tensor y_pool_pp [N][Ho/2+1][Wo/2+1][64] (real pixels
 * only) and the BatchNorm partial sums of ALL convolution outputs (bltvqg_conv_stem_pool_stat_rows() rows of 64); bltvqg_bn_apply_pp(relu)
 * on y_pool_pp with the finalised scale / shift then equals bn -> relu -> maxpool of the stored pre-pool tensor bit for bit, without
 * writing and re-reading that tensor.  bf16, Cout = 64, Ho % 8 == 0 and Wo % 14 == 0 (bltvqg_conv_stem_pool_ok). */
int bltvqg_conv_stem_pool(const void* x_padded, const void* w, const float* gamma, void* y_pool_pp, int N, int H, int W, int Hp, int Wp,
                          float* stat_sum, float* stat_sq, void* stream);
int bltvqg_conv_stem_pool_ok(int dtype, int H, int W, int Hp, int Wp, int Cout);
int bltvqg_conv_stem_pool_stat_rows(int N, int H, int W);

int bltvqg_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int64_t rows, int cols, float eps, void* stream);
/* dgamma / dbeta are accumulated (+=). dres (optional) is added to dx. */
int bltvqg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                         const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int cols, void* stream);

/* Per-channel batch statistics from the convolution's partial sums -> scale / shift (+ running statistics), one launch: the slice sums
 * meet in fp64 accumulators through device-scope atomics and the last workgroup (ticket counter) finishes.  `scratch` holds
 * bltvqg_bn_scratch_doubles(C) doubles; the owner zeroes it ONCE (hipMemset) before the first call, every call leaves it zeroed. */
int bltvqg_bn_scratch_doubles(int C);
int bltvqg_bn_finalize(const float* psum, const float* psq, int nparts, int C, int64_t count, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                       float* shift, double* scratch, void* stream);
int bltvqg_bn_apply(int dtype, const void* x, const float* scale, const float* shift, const void* res, void* y, int64_t rows,
                    int C, int relu, void* stream);
int bltvqg_bn_relu_maxpool(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi,
                           int C, void* stream);
int bltvqg_avgpool(int dtype, const void* x, void* y, int N, int HW, int C, void* stream);
/* BatchNorm1d over the batch dimension (encoder_cnn.py:21,34), B <= 512 rows: a thread keeps its rows of a column in registers */
int bltvqg_bn1d_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                    float* running_mean, float* running_var, int B, int C, float eps, float momentum, void* stream);
int bltvqg_bn1d_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                    void* dx, float* dgamma, float* dbeta, int B, int C, void* stream);

/* Attention core on packed projections.  Q [B*Tq, ldq], K/V [B*Tk, ld*]; head h uses columns [h*d, (h+1)*d).
 * key_ids [B,Tk] int32: id 0 => key masked (value -1e18 like the reference); causal: key j > query i masked. */
int bltvqg_attn_fwd(int dtype, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                    const int32_t* key_ids, int B, int heads, int Tq, int Tk, int d, int causal, float scale, float drop_p,
                    uint64_t seed, uint32_t stream_id, void* stream);
/* The same forward on ROW SUBSETS of wider tensors (round 4, incremental greedy decoding: models/iq.py:134-141 re-decodes the whole prefix
 * for every new token; here the projections of earlier steps are the key / value cache).  Q / O hold q_rows rows per batch element of which
 * the first Tq (from the given pointer) are queries; K / V / key_ids hold k_rows rows per batch element of which the first Tk are keys
 * (0 = Tq / Tk, i.e. bltvqg_attn_fwd).  No dropout (inference).  Query i against keys 0..Tk-1 gives the values bltvqg_attn_fwd gives for
 * that query over the same keys. */
int bltvqg_attn_fwd_rows(int dtype, const void* Q, int ldq, int q_rows, const void* K, int ldk, const void* V, int ldv, int k_rows, void* O, int ldo,
                         const int32_t* key_ids, int B, int heads, int Tq, int Tk, int d, int causal, float scale, void* stream);
int bltvqg_attn_bwd(int dtype, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* dO, int lddo,
                    void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, const int32_t* key_ids, int B, int heads, int Tq,
                    int Tk, int d, int causal, float scale, float drop_p, uint64_t seed, uint32_t stream_id, void* stream);

int bltvqg_embed_gather(int dtype, const float* table, const int32_t* ids, void* out, int64_t rows, int E, int ld, void* stream);
int bltvqg_embed_scatter(int dtype, const void* d, int ld, const int32_t* ids, float* dtable, int64_t rows, int E, int pad_id,
                         void* stream);

/* Token cross-entropy, ignore_index = 0, mean over count[0] targets.  loss_out += loss.  If write_grad, the logits buffer is
 * overwritten with gscale * dloss/dlogits (pad columns [V, ld) zeroed). */
int bltvqg_ce_fwd_bwd(int dtype, void* logits, int ld, const int32_t* target, int64_t M, int V, const float* count, float gscale,
                      float* loss_out, int write_grad, void* stream);
int bltvqg_bow_ce_fwd_bwd(int dtype, const void* zlogit, int ld, const int32_t* target, int B, int T, int V, const float* count,
                          float gscale, float* loss_out, void* dz, void* stream);
int bltvqg_mse_fwd_bwd(int dtype, const void* a, const void* b, int64_t n, float gscale, float* loss_out, void* da, void* db,
                       void* stream);
int bltvqg_latent_fwd(int dtype, const void* mlv_prior, const void* mlv_post, const float* eps, void* z, float* kld_out, int B,
                      int Z, int ld, void* stream);
int bltvqg_latent_bwd(int dtype, const void* mlv_prior, const void* mlv_post, const float* eps, const void* dz, float kld_gscale,
                      void* dmlv_prior, void* dmlv_post, int B, int Z, int ld, void* stream);

int bltvqg_sumsq(const float* x, int64_t n, float* out, void* stream);
int bltvqg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm_sq, float max_norm, float lr,
                     float beta1, float beta2, float eps, int step, void* stream);
/* keep-mask (1 = keep) the kernels use for dropout site `stream_id`: element (r, c) has index r*ld_index + c */
int bltvqg_dropout_mask(uint64_t seed, uint32_t stream_id, int64_t rows, int cols, int ld_index, float p, uint8_t* out,
                        void* stream);
int bltvqg_cast(int dtype_src, const void* src, int ld_src, int dtype_dst, void* dst, int ld_dst, int64_t rows, int cols,
                void* stream);

/* ---------------- batch producer ----------------
 * Replaces, for a store that lives in HBM, what the reference does per sample on host workers: IQDataset.__getitem__
 * (utils/data_loader.py:45-129), collate_fn's stacking (utils/data_loader.py:150-163) and the image transform
 * (train_iq.py:264-272).  All pointers are device pointers unless noted. */
/* ToTensor -> ToPILImage on a stored float HWC image (train_iq.py:265-266): out[i] = byte(255 * images[i]), fp32 product, truncation,
 * wrap modulo 256 (|255 x| < 2^31).  images 16-B aligned, out 4-B aligned. */
int bltvqg_image_store_u8(const float* images, uint8_t* out, int64_t count, void* stream);
/* Token rows of the samples index[0..B) (data_loader.py:59-86,115-116), int64 like collate_fn's .long():
 * questions [B,q_len], posteriors [B,q_len+1], answers [B,a_len+1], answer_types [B] (category WORD ids),
 * answer_types_for_input [B,3].  Stored tables are int32: questions [n_rows,q_len], answers [n_rows,a_len], answer_types [n_rows]
 * (index into cat_word_ids[n_cat], the vocabulary ids of the sorted category names, data_loader.py:42,78-79).  An index outside
 * [0,n_rows) yields all-zero rows. */
int bltvqg_batch_rows(const int32_t* questions, const int32_t* answers, const int32_t* answer_types, const int32_t* cat_word_ids, int n_cat,
                      int64_t n_rows, const int64_t* index, int B, int q_len, int a_len, int64_t* out_questions, int64_t* out_posteriors,
                      int64_t* out_answers, int64_t* out_answer_types, int64_t* out_types_for_input, void* stream);
/* Images of the samples index[0..B): gather table[image_indices[index[b]]] (uint8 HWC [n_images,S,S,3]), crop boxes[b] =
 * (top,left,h,w), resize to osz x osz as Pillow's resize(BILINEAR) does for 8-bit images (horizontal pass rounded to 8 bits, then
 * vertical, 22-bit fixed-point weights), /255, (x-mean)/std -> out fp32 [B,3,osz,osz]; out_u8 (optional) receives the resized bytes
 * [B,osz,osz,3].  coeffs int32 [B,2,osz,2+KS] = per output column (then row): first source pixel inside the crop, tap count, KS
 * weights — from blt-vqg_amd/batch.py::resample_coeffs; NULL / KS 0 when every box is osz x osz (copy).  mean_std: 6 HOST floats
 * (mean rgb, std rgb).  Out-of-range indices/boxes produce the transform of a black pixel; no out-of-bounds access is possible. */
int bltvqg_batch_images(const uint8_t* table, int64_t n_images, int S, const int32_t* image_indices, int64_t n_rows, const int64_t* index,
                        const int32_t* boxes, const int32_t* coeffs, int KS, int B, int osz, const float* mean_std, float* out,
                        uint8_t* out_u8, void* stream);

/* bltvqg_batch_images written straight into the train-step engine's stem input (bltvqg_engine_image_input gives out / dtype / Hp / Wp;
 * pad_top = pad_left = 3): zero-bordered NHWC4 [B,Hp,Wp,4], what bltvqg_img_pack makes of the fp32 NCHW tensor — bit-identical to
 * batch_images + img_pack, without the fp32 tensor and the extra launch.  Follow with bltvqg_engine_forward(images = NULL). */
int bltvqg_batch_images_packed(const uint8_t* table, int64_t n_images, int S, const int32_t* image_indices, int64_t n_rows,
                               const int64_t* index, const int32_t* boxes, const int32_t* coeffs, int KS, int B, int osz, const float* mean_std,
                               int dtype, void* out, int Hp, int Wp, int pad_top, int pad_left, void* stream);

/* ---------------- train-step engine ---------------- */

typedef struct bltvqg_config {
    int32_t batch;          /* per-GPU batch B */
    int32_t hidden_dim, pwffn_dim, latent_dim, emb_dim, num_layers, num_heads, vocab_size;
    int32_t len_context;    /* S_a: 5 ("ans") or 3 ("cat") */
    int32_t len_posterior;  /* S_p: 21 */
    int32_t len_target;     /* T: 20 */
    int32_t image_h, image_w;
    int32_t dtype;          /* BLTVQG_F32 or BLTVQG_BF16 */
    float attention_dropout, relu_dropout;   /* 0.1 / 0.1 in the reference (transformer_layers.py:97,164) */
    float kl_ceiling, aux_ceiling, image_recon_lambda;
    /* Bottom-up feature mode (BASELINE.json configs[4]; SURVEY A2': the reference has no implementation): num_regions > 0 makes the
     * `images` argument of forward / decode_greedy a fp32 [B, num_regions, region_dim] tensor of precomputed region features and
     * replaces the ResNet by mean_r(Linear(region_dim -> H)(x_r)) -> the same BatchNorm1d (parameters encoder_cnn.region_proj.{weight,
     * bias}, encoder_cnn.bn.*); image_h / image_w are ignored.  0 = image mode.  region_dim % 8 == 0. */
    int32_t num_regions, region_dim;
    /* Region pooling of the bottom-up mode (the reference names the model "Bottom-Up" but has no region code, README.md:2):
     * 0 = mean over the regions (above); 1 = region-attention pooling: p_r = Linear(region_dim -> H)(x_r), s_r = w_a . tanh(p_r),
     * alpha = softmax_r(s), feature = sum_r alpha_r p_r -> the same BatchNorm1d.  w_a is the extra parameter
     * encoder_cnn.region_attn.weight [1, H].  Defined by this build, parity against the oracle's restatement only. */
    int32_t region_pool;
    /* Padded model widths.  The kernels move activations in 16-byte vectors, so hidden_dim / latent_dim / pwffn_dim are multiples of 8
     * and the head width hidden_dim / num_heads is what the attention kernels tile; the reference's CLI defaults (train_iq.py:315-325:
     * hidden 300 = 4 heads x 75, latent 300, FFN 600) are not.  The host mirror (models.IQ) creates the engine with PADDED widths —
     * hidden_dim = num_heads * round8(75) = 320, latent_dim 304 (pwffn_dim 600 needs none) — keeps every pad weight at zero and scatters the
     * reference-shaped parameters into the padded layout; head_dim_true (75; 0 = not padded) tells the engine the real head width for
     * the places where the width itself enters the arithmetic: LayerNorm mean / variance over the real features only (pad columns stay
     * zero and get zero gradient), the 1/sqrt(d_head) attention scale, the MSE mean over B x true width, the sinusoid timing signal. */
    int32_t head_dim_true;
} bltvqg_config;

typedef struct bltvqg_engine bltvqg_engine;

/* which: 0 = trainable parameters (flat fp32 buffer, gradients and Adam state share the layout), 1 = frozen backbone
 * parameters and BatchNorm running statistics (flat fp32 buffer). */
bltvqg_engine* bltvqg_engine_create(const bltvqg_config* cfg);
void bltvqg_engine_destroy(bltvqg_engine* e);
int bltvqg_engine_num_params(const bltvqg_engine* e, int which);
/* name_host: caller buffer of name_cap bytes; shape as (rows, cols) with cols = 0 for 1-D and rows = cols = 0 for 4-D conv
 * weights whose dims come back in dims4_host[4]; offset in floats into the flat buffer; late = 1 for parameters that only
 * receive gradients after the pre-training phase. */
int bltvqg_engine_param_info(const bltvqg_engine* e, int which, int index, char* name_host, int name_cap, int64_t* offset,
                             int64_t* numel, int32_t* dims4_host, int32_t* ndim, int32_t* late);
int64_t bltvqg_engine_flat_size(const bltvqg_engine* e, int which);      /* floats */
int64_t bltvqg_engine_late_offset(const bltvqg_engine* e);               /* first float of the phase-2-only region */
int64_t bltvqg_engine_workspace_bytes(const bltvqg_engine* e);
/* Binds caller-owned device memory.  workspace must be 256-byte aligned and is zeroed by bind (synchronously). */
int bltvqg_engine_bind(bltvqg_engine* e, float* train, float* grad, float* adam_m, float* adam_v, float* frozen,
                       void* workspace, int64_t workspace_bytes);
/* Parameters were written from outside the engine (load_state_dict, a broadcast): the frozen backbone is repacked and the bf16
 * weight shadows of EVERY engine sharing these parameters are rebuilt on their next forward. */
void bltvqg_engine_invalidate_frozen(bltvqg_engine* e);
/* bf16 engines keep a bf16 mirror ("shadow") of the weight matrices.  By default every forward rebuilds it from the fp32 parameters
 * (anyone may have written them: a torch optimiser on the autograd path).  A caller that updates the parameters ONLY through
 * bltvqg_engine_optimizer_step[_async] (and reports every other write with bltvqg_engine_invalidate_frozen) sets on = 1: the update
 * then writes the shadow in the same pass and forward only derives the transposed copies (saves a 0.5 GB pass per step). */
int bltvqg_engine_trust_shadows(bltvqg_engine* e, int on);
/* The TRAINABLE parameters may have been written from outside the engine (a torch optimiser stepping the nn.Parameter views, an in-place
 * edit): no bf16 shadow of any engine sharing them is current any more.  Cheaper than bltvqg_engine_invalidate_frozen (the frozen conv
 * weights are not repacked); the autograd path (models.IQ.forward) calls it before every forward. */
void bltvqg_engine_invalidate_params(bltvqg_engine* e);

/* The engine's stem input: zero-bordered NHWC4 [batch, *Hp, *Wp, 4] of *dtype inside the bound workspace, image at (3,3).  A caller
 * that fills it itself (bltvqg_batch_images_packed) passes images = NULL to bltvqg_engine_forward / decode_greedy, which then skip
 * bltvqg_img_pack.  Image mode only. */
int bltvqg_engine_image_input(bltvqg_engine* e, void** ptr, int* Hp, int* Wp, int* dtype);

/* The frozen conv stack one batch ahead.  models/encoder_cnn.py:18-19 freezes the ResNet-18 backbone, so the 20 convolutions (+ train-mode
 * BatchNorm2d, average pool) of batch i+1 depend on nothing step i updates: this call enqueues them on the engine's own conv stream, ordered
 * behind everything enqueued on `stream` so far (the producer of `images`), and the NEXT bltvqg_engine_forward — called with images = NULL —
 * waits for the pooled feature and starts at the trainable head (fc + BatchNorm1d).  images fp32 NCHW [B,3,h,w], or NULL when the caller
 * filled the stem input itself (bltvqg_engine_image_input).  The conv stack's results (pooled feature, BatchNorm2d running statistics) are
 * bit-identical to the inline forward's: same kernels, same order of running-statistics updates (one in-order stream); a whole training STEP is
 * bit-identical to the inline step in its first step only — from the second step on the parameters have been through float-atomic
 * gradient sums, whose order differs between any two runs (tests/test_prefetch_gpu.py states the bars).  The statistics in the frozen buffer
 * run one batch AHEAD of the step while a stack is pending: readers order themselves with bltvqg_engine_conv_stream_wait.  At most two batches may be pending; the second one may only be enqueued
 * after the backward of the step that consumed the previous batch (the pooled feature is double-buffered).  Image mode, train-mode
 * BatchNorm only.  bltvqg_engine_prefetch_pending: number of prefetched batches not yet consumed by a forward. */
int bltvqg_engine_prefetch_images(bltvqg_engine* e, const float* images, void* stream);
int bltvqg_engine_prefetch_pending(const bltvqg_engine* e);
/* How much of the conv stack a prefetch runs: the stack is 10 stages — [0] image pack + 7x7/2 stem + BatchNorm + ReLU + max-pool (HBM-bound),
 * [1..8] the eight BasicBlocks, [9] the average pool; `stages` = 1..10 leading stages go to the conv stream, the forward that consumes the
 * batch runs the rest inline.  10 (default) = the whole stack.  With stages < 10 only ONE batch may be pending and prefetch_images must be
 * called AFTER the forward of the current step (the partial result lives in the single-buffered conv workspace). */
int bltvqg_engine_set_prefetch_split(bltvqg_engine* e, int stages);
/* CU partition (MI355X: 256 CUs in 8 XCDs).  A prefetched conv stack and the dependent chain of the step want different things from a CU
 * (convolution workgroups hold 2 x 80 KB of LDS for ~25 us each, the chain's launches are short and queue behind them when they share
 * CUs), so the engine can create its streams under COMPLEMENTARY CU masks (hipExtStreamCreateWithCUMask; words x 32 bits, HOST arrays,
 * NULL = unmasked): chain_mask for side stream 0 and the stream bltvqg_engine_chain_stream hands out (run forward / loss_backward on
 * it), side_mask for side stream 1 (deferred weight gradients, asynchronous optimiser), conv_mask for the prefetch stream.  chain_cus =
 * number of CUs the chain mask leaves (the launch planners size one round of workgroups for it; 0 = the whole chip).  Call while
 * nothing of this engine is in flight (synchronises the device); the planner setting is process-wide (one process per GPU). */
int bltvqg_engine_set_cu_masks(bltvqg_engine* e, const uint32_t* chain_mask_host, const uint32_t* side_mask_host, const uint32_t* conv_mask_host,
                               int words, int chain_cus);
int bltvqg_engine_chain_stream(bltvqg_engine* e, void** stream);
/* the engine's prefetch (conv) stream, for diagnostics: which CUs its work lands on (bltvqg_hw_id_probe) */
int bltvqg_engine_conv_stream(bltvqg_engine* e, void** stream);
/* Orders `stream` behind the last conv stack enqueued on the look-ahead stream (no-op when none ran ahead): a prefetched stack advances
 * the BatchNorm2d running statistics in the frozen buffer one batch ahead, so readers of that buffer (state_dict, checkpoints) wait here
 * and see the statistics INCLUDING the look-ahead batch, whole. */
int bltvqg_engine_conv_stream_wait(bltvqg_engine* e, void* stream);
/* Diagnostic: the engine's two side streams (0: posterior encoder / branch work, 1: context encoder / deferred weight gradients / optimiser),
 * for stream-ordering tests that delay one of them. */
int bltvqg_engine_side_stream(bltvqg_engine* e, int which, void** stream);
/* Run the prefetched conv stacks on a stream the CALLER owns (and outlives the engine with) instead of the engine's own: a feeder that
 * copies the next batch over PCIe enqueues copy and bltvqg_engine_prefetch_images on that one stream — one pipeline underneath the current
 * step, no extra stream.  Not combined with conv_mask_host (the caller's stream has the caller's CU mask). */
int bltvqg_engine_adopt_conv_stream(bltvqg_engine* e, void* stream);

/* IQ.forward (iq.py:82-114).  images fp32 NCHW [B,3,h,w]; token tensors int64 like the reference batch; eps fp32 [B,Z]
 * (may be NULL in phase 1).  train_bn: BatchNorm in train mode (batch statistics + running-stat update), as the reference. */
int bltvqg_engine_forward(bltvqg_engine* e, const float* images, const int64_t* context, const int64_t* posterior,
                          const int64_t* target, const float* eps, int phase2, uint64_t seed, void* stream);
/* IQ.decode_greedy (models/iq.py:117-152) on an engine created with len_target = max_decode_length + 1 and dropout 0:
 * tokens [B,T] (argmax per step), top_idx / top_val [B,T,6] (top-6 softmax probabilities, torch.topk order).
 * train_bn = 0: BatchNorm layers use running statistics (module.eval(), as under Lightning validation); eps [B,Z] is the latent
 * noise (required when phase2). */
int bltvqg_engine_decode_greedy(bltvqg_engine* e, const float* images, const int64_t* context, const float* eps, int phase2, int train_bn,
                                int32_t* tokens, int32_t* top_idx, float* top_val, void* stream);
/* BatchNorm mode of bltvqg_engine_forward: 1 = batch statistics + running-stat update (the reference's training behaviour,
 * default), 0 = running statistics (module.eval()). */
int bltvqg_engine_set_bn_train(bltvqg_engine* e, int train);
/* calculate_losses (train_iq.py:81-103) fused with the whole backward pass.  Gradients of every trainable parameter are
 * written to the bound flat gradient buffer (zeroed first).  kl_weight = min(tanh(6*kliter/full_kl_step-3)+1, 1). */
int bltvqg_engine_loss_backward(bltvqg_engine* e, float kl_weight, void* stream);
/* Backward from caller-supplied output gradients (fp32, any may be NULL = zero): autograd integration of IQ.forward. */
int bltvqg_engine_backward_external(bltvqg_engine* e, const float* d_output, const float* d_zlogit, float d_kld,
                                    const float* d_feats, const float* d_recon, void* stream);
/* clip_grad_norm_(max_norm) + Adam over the regions that received gradients this step. */
int bltvqg_engine_optimizer_step(bltvqg_engine* e, float lr, float max_norm, float beta1, float beta2, float eps,
                                 void* stream);
/* The same update enqueued on the engine's own optimiser stream, ordered behind everything enqueued on `stream` so far (e.g. the
 * gradient all-reduce that `stream` was made to wait for).  `stream` itself does NOT wait for it: the next bltvqg_engine_forward
 * lets the frozen CNN start at once and makes only the consumers of trainable parameters wait, which hides the update and the tail
 * of the all-reduce behind ~45 % of the next step; every other engine call orders itself behind the update first.  A caller that
 * reads the parameter / moment buffers itself must call bltvqg_engine_optimizer_wait(stream) before doing so on `stream`. */
int bltvqg_engine_optimizer_step_async(bltvqg_engine* e, float lr, float max_norm, float beta1, float beta2, float eps,
                                       void* stream);
int bltvqg_engine_optimizer_wait(bltvqg_engine* e, void* stream);
/* Adam step counters (bias correction) of the always-trained and the latent-phase-only parameter regions: checkpoint / resume */
int bltvqg_engine_adam_steps(const bltvqg_engine* e, int32_t* steps_main_host, int32_t* steps_late_host);
int bltvqg_engine_set_adam_steps(bltvqg_engine* e, int32_t steps_main, int32_t steps_late);
/* Engines of ONE model with different batch shapes share the parameter / gradient / Adam-moment buffers (bind the same pointers);
 * this call makes `e` share `primary`'s Adam step counters as well, so that a step taken by either engine (the ragged last batch
 * of an epoch, utils/data_loader.py: no drop_last) advances the one bias correction that belongs to those moments. */
int bltvqg_engine_share_optimizer_state(bltvqg_engine* e, bltvqg_engine* primary);
/* outputs, converted to contiguous fp32: what = 0 output [B,T,V], 1 z_logit [B,V], 2 image_features [B,H],
 * 3 reconstructed [B,H], 4 stats float[8] = {loss_rec, loss_img, kld, loss_aux, grad_norm_sq, n_targets, bad_token_ids, 0}
 * (bad_token_ids = number of input token ids outside [0, vocab_size) in the last forward: they were replaced by <pad>; the host
 * mirror raises on a non-zero count, as the reference's embedding lookup raises a device-side index error),
 * 5 encoder_outputs [B,S_a,H], 6 decoder_outputs [B,T,H] */
int bltvqg_engine_read(bltvqg_engine* e, int what, float* dst, void* stream);
uint32_t bltvqg_engine_dropout_stream_id(int stack, int layer, int site);
/* In-stream timing of the two dominant kernel families: while enabled, every launch of the family is bracketed by a hipEvent pair
 * on the stream it is launched on.  mask bit 0: the convolution launches of the frozen ResNet-18 stack (class 0, ALGORITHMIC flops
 * 2*M*Cout*KH*KW*Cin, unpadded); bit 1: every Linear-layer GEMM of the transformer stacks, embedding, vocabulary projection and
 * latent nets - forward, input gradient and weight gradient (class 1, flops 2*M*N*K).  profile_read_class synchronises on the
 * class's events and returns the summed kernel time, the number of launches and their flops since the last read.  enable(0) only
 * pauses recording (an event record costs the stream a ~6 us bubble, so callers sample a subset of their steps); the recorded
 * launches accumulate until they are read.  profile_read = profile_read_class(0).
 * mask bits 8..14 (optional): n > 1 brackets only every n-th plain Linear GEMM launch of class 1 and counts its time, flops and
 * launch n times (grouped weight-gradient launches and convolutions are always bracketed): 4x fewer bubbles in the profiled step. */
int bltvqg_engine_profile_enable(bltvqg_engine* e, int mask);
int bltvqg_engine_profile_read(bltvqg_engine* e, double* total_ms_host, int32_t* launches_host, double* flops_host);
int bltvqg_engine_profile_read_class(bltvqg_engine* e, int cls, double* total_ms_host, int32_t* launches_host, double* flops_host);
/* same, plus the bracketed time split by the stream the launches sat on (per_stream_ms_host4: [0] the caller's stream, [1] / [2] the engine's
 * two side streams, [3] the conv look-ahead stream; may be NULL): the step's streams run side by side, so the SUM over streams can exceed the
 * step's wall time while every single stream's share fits inside it */
int bltvqg_engine_profile_read_streams(bltvqg_engine* e, int cls, double* total_ms_host, int32_t* launches_host, double* flops_host,
                                       double* per_stream_ms_host4);
/* Diagnostic (bltvqg_debug_set(12, 1)): milliseconds from the start of the last forward to the phase boundaries of the step on the
 * caller's stream — [1] CNN + encoders joined, [2] decoder starts, [3] decoder done, [4] end of forward, [5] losses, [6] decoder backward
 * starts, [7] decoder backward done, [8] encoder backward starts, [9] done, [10] end of backward, [11] image feature done on the CNN's
 * stream (before the encoder streams are joined: [11] ~ [1] means the CNN chain is the long pole of the first phase); -1 = not recorded. */
int bltvqg_engine_phase_stamps(bltvqg_engine* e, float* ms_host12);
/* Gradient buckets for the data-parallel exchange (reference: pl.Trainer(gpus=N), train_iq.py:372-373 = DDP gradient mean): contiguous
 * float ranges of the flat gradient buffer, LISTED IN THE ORDER BACKWARD COMPLETES THEM (not in address order; together they tile the
 * buffer once): decoder layer groups, the latent-phase heads, the two encoder stacks' layer groups, the tail (shared embedding + CNN head).
 * A bucket is a group of whole layers of >= ~32 MB of parameters.  bltvqg_engine_bucket_wait makes `stream` wait until bucket i of the last
 * backward is final.  bltvqg_engine_set_bucket_flush(1): the collected weight gradients of a stack are flushed to the weight-gradient
 * stream at every bucket boundary, so a bucket's event fires (and its all-reduce can start) while backward is still running; 0 (default,
 * single GPU): one flush per stack, the buckets of a stack become final together. */
int bltvqg_engine_set_bucket_flush(bltvqg_engine* e, int on);
int bltvqg_engine_num_buckets(const bltvqg_engine* e);
int bltvqg_engine_bucket_info(const bltvqg_engine* e, int i, int64_t* offset, int64_t* numel, int32_t* late);
int bltvqg_engine_bucket_wait(bltvqg_engine* e, int i, void* stream);

#ifdef __cplusplus
}
#endif
#endif
