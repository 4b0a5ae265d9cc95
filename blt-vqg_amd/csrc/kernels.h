// Internal C++ launch API of libbltvqg_hip.so.  Every launcher enqueues asynchronously on `stream`,
// allocates nothing and returns BLT_OK or a negative error (message via bltvqg_last_error_string()).
// `dtype` is BLT_F32 or BLT_BF16 and names the storage type T of activations / (shadow) weights.
#pragma once
#include <vector>
#include "common.h"

// "Padded-pitch" (PP) activation layout [N][H+1][W+1][C]: every image row is followed by ONE zero pixel and every image by ONE
// zero row, so that in linear pixel order the left/right/top/bottom neighbours of a border pixel are zeros (the pad pixel after row
// h is the right pad of row h and the left pad of row h+1) — a 3x3 stride-1 window of a run of consecutive pixels is then itself a
// run of consecutive pixels, which is what conv3x3_pp (conv_pp.hip) stages through LDS once instead of nine times.  A PP buffer
// carries BLT_PP_GUARD_FRONT zero pixels before pixel 0 and BLT_PP_GUARD_TAIL after the last one.
#define BLT_PP_GUARD_FRONT 64
#define BLT_PP_GUARD_TAIL 384
struct ConvGeom {
    int Hi, Wi, Cin, cin_log2, Ho, Wo, KH, KW, stride, pad;
    // generalisation for PP operands (0 = plain NHWC): rows per image / pixels per row of the INPUT buffer, and the number of valid
    // rows / columns of the output grid Ho x Wo (PP output: Ho = valid + 1, Wo = valid + 1; rows of the pad positions come out 0)
    int in_rows = 0, in_pitch = 0, Hov = 0, Wov = 0;
};

// C[M,N] = epilogue( alpha * sum_k A^[m,k] * B^[n,k] )
//   transA = 0: A stored [M, lda] (k contiguous)      transA = 1: A stored [K, lda] (m contiguous)
//   transB = 0: B stored [N, ldb] (k contiguous)      transB = 1: B stored [K, ldb] (n contiguous)
// epilogue order: *alpha, +bias[n], +rowtab[rowidx[m]][n], relu, dropout, *(maskY!=0)*mask_scale,
//                 ->C2 (optional copy), +R[m,n], +old C (accumulate), ->C
struct GemmArgs {
    const void* A = nullptr;
    const void* B = nullptr;
    void* C = nullptr;
    int M = 0, N = 0, K = 0;
    int lda = 0, ldb = 0, ldc = 0;
    int transA = 0, transB = 0;
    int out_f32 = 0;
    float alpha = 1.f;
    const float* bias = nullptr;
    const float* rowtab = nullptr;
    const int* rowidx = nullptr;
    int ldt = 0;
    int relu = 0;
    float drop_p = 0.f;
    uint64_t seed = 0;
    uint32_t stream_id = 0;
    const void* maskY = nullptr;
    int ldm = 0;
    float mask_scale = 1.f;
    void* C2 = nullptr;
    int ldc2 = 0;
    const void* R = nullptr;
    int ldr = 0;
    int accumulate = 0;
    float* stat_sum = nullptr;   // [2*tiles_m, N] per-half-tile column sums of the raw accumulators
    float* stat_sq = nullptr;
    int is_conv = 0;             // 1 = NHWC implicit-GEMM gather, 2 = 7x7/2 stem on a zero-bordered NHWC4 image (cg.Hi/Wi = padded dims)
    int no_dma = 0;              // 1 = force the register-staged kernel (A/B testing of the LDS-DMA ring)
    int split_k = 0;             // max K-splits (fp32 atomic accumulation) for the weight-gradient form; 0 = off
    ConvGeom cg = {};

    int force_tile = 0;          // 0 = heuristic, 64 or 128
    float* a_rowsum = nullptr;   // transA only: a_rowsum[m] += sum_k op(A)[m,k]  (bias gradient of the weight-gradient form, float atomics)
    // Fused LayerNorm of the result rows (bf16, k-contiguous operands, N <= 256 so that one workgroup owns whole rows): after the
    // epilogue above has produced C (rounded to bf16, as a separate LayerNorm launch would read it), ln_out = LN(C) * ln_gamma + ln_beta
    // and the row statistics are written as well (transformer_layers.py:134,202,256-257,320-322: every LayerNorm of the stacks reads
    // the output of a Linear + residual)
    // LayerNorm applied to the A operand inside the GEMM (bf16, k-contiguous operands, K <= 256 so that a workgroup's 64 rows of A sit in
    // LDS whole): A := LN(A) * lnA_gamma + lnA_beta (rounded to bf16, as a separate LayerNorm launch would store it) before the MFMA
    // loop; the workgroups of the first column tile also write that normalised A (lnA_out, ld = K) and the row statistics, which the
    // backward needs.  Every LayerNorm of the stacks feeds exactly one Linear (QKV / query / first FFN layer).
    const float* lnA_gamma = nullptr;
    const float* lnA_beta = nullptr;
    void* lnA_out = nullptr;
    float* lnA_mean = nullptr;
    float* lnA_rstd = nullptr;
    float lnA_eps = 1e-5f;
    const float* ln_gamma = nullptr;
    const float* ln_beta = nullptr;
    void* ln_out = nullptr;
    float* ln_mean = nullptr;
    float* ln_rstd = nullptr;
    float ln_eps = 1e-5f;
    int nt2_bm = 0, nt2_bn = 0;      // force a tile shape of the planned-tile kernel (bltvqg_gemm_ex)
    // ---- LayerNorm folded into the Linear that consumes it (gemm2.hip, bf16): LN(x) W^T + b = rstd_m (x W'^T - mean_m s_n) + c_n with
    // W' = W diag(gamma) (the bf16 operand B), s_n = sum_k W'[n,k], c_n = sum_k beta[k] W[n,k] + b_n.  A holds the RAW rows x; their
    // {sum, sum of squares} come from fold_stat, left there by the launch that produced x (out_stat of a GEMM, rows_add's statistics
    // form).  No bias / alpha (folded into fold_c), no residual / mask / second output / row table / accumulate with it.
    const float* fold_s = nullptr;      // [N]
    const float* fold_c = nullptr;      // [N]
    const float* fold_stat = nullptr;   // [M][stat_slots][2]: the first fold_np slots of a row are partial {sum, sum of squares}, added in slot order
    int fold_np = 1;
    float* fold_mean = nullptr;         // [M], written by the first column tile (the LayerNorm's backward reads them); may be null
    float* fold_rstd = nullptr;
    int fold_sstride = 0;               // slots between consecutive rows of fold_stat (0 = stat_slots): strided row views of a wider statistics buffer
    float fold_eps = 1e-5f;
    float fold_n = 0.f;                 // features per row the statistics cover (the LayerNorm's width)
    // out_stat[m][g][0..1] = {sum, sum of squares} of columns 64 g .. 64 g + 63 of result row m AS STORED (bf16-rounded): one slot per 64
    // columns whatever the tile shape (plain stores; the consumer adds the cdiv(N, 64) partials in slot order, so the folded LayerNorm is
    // bit-reproducible AND independent of tile shapes / row counts).  stat_slots = slots per row (both).
    float* out_stat = nullptr;
    int stat_slots = 1;
};

// The second problem of a paired planned-tile launch (gemm2.hip: gemm_nt2_pair_kernel): same N, K, tile shape and epilogue terms as the
// GemmArgs it rides with; its own operands, leading dimensions, row count and dropout stream.
struct GemmPair {
    int tiles1 = 0;      // row tiles of problem 1 (filled in by the launcher)
    int M = 0;
    const void* A = nullptr; int lda = 0;
    const void* B = nullptr; int ldb = 0;
    void* C = nullptr; int ldc = 0;
    const float* bias = nullptr;
    const void* maskY = nullptr; int ldm = 0;
    void* C2 = nullptr; int ldc2 = 0;
    const void* R = nullptr; int ldr = 0;
    uint32_t stream_id = 0;
    const float* fold_s = nullptr; const float* fold_c = nullptr; const float* fold_stat = nullptr;
    float* fold_mean = nullptr; float* fold_rstd = nullptr;
    float* out_stat = nullptr;
};
int blt_gemm(int dtype, const GemmArgs& a, hipStream_t stream);
int blt_gemm_validate(int dtype, const GemmArgs& a);      // the operand checks of blt_gemm alone
int blt_gemm_stat_rows(const GemmArgs& a, int dtype);   // number of partial rows written to stat_sum/stat_sq (2*tiles_m)
int blt_gemm_tile(const GemmArgs& a, int dtype);
void blt_debug_set(int key, int value);
int blt_debug_get(int key);      // keys 4..6: A/B switches of conv_pp.hip (4 = force BN 64/128, 5 = XCD mapping 1 chunked / 2 round-robin, 6 = ring 1 deep / 2 shallow)
int blt_gemm_splits(const GemmArgs& a, int dtype);
// gemm2.hip: one-round-per-chip NT GEMM (bf16, k-contiguous operands, bf16 output): tile shape planned per problem
bool blt_gemm_nt2_ok(int dtype, const GemmArgs& a, bool any_rows = false);
int blt_gemm_nt2(const GemmArgs& a, hipStream_t s, int force_bm = 0, int force_bn = 0);
void blt_gemm_nt2_tile(int M, int N, int K, int* bm, int* bn, bool row_stat = false, int M2 = 0);      // row_stat: a launch with out_stat; M2: rows of a paired launch's second problem
bool blt_gemm_nt2_pair_ok(int dtype, const GemmArgs& a, const GemmArgs& b);
int blt_gemm_nt2_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t s);
// CUs the launch planners size a "round" for: 256 (the chip) unless the dependent chain runs on a CU partition (engine_set_cu_masks)
int blt_hw_id_probe(int* out, int n_wg, int spin_ticks, hipStream_t s);      // misc.hip (experiments build)
void blt_set_plan_cus(int n);      // 0 = the whole chip
int blt_plan_cus();
// gemm2.hip: grouped weight gradients (one launch for a table of dW = dY^T X problems; GemmArgs in the transA/transB weight-gradient
// form: A = dY [rows, lda], B = X [rows, ldb], C = dW fp32 [M = out features, ldc], K = rows, a_rowsum = bias gradient or null)
struct blt_wg_problem {
    const void* A; const void* B; float* C; float* bias;
    int Nw, Kw, Mtok, lda, ldb, ldc, tiles_n, splits, accumulate, pad;
};
bool blt_wgrad_group_ok(int dtype, const GemmArgs& a);
int blt_wgrad_group_plan(const std::vector<GemmArgs>& g, std::vector<blt_wg_problem>& probs, std::vector<int>& wg0, int* tile_rows);
int blt_wgrad_group_launch(const blt_wg_problem* probs_dev, const int* wg0_dev, int nprob, int nwg, int tile_rows, hipStream_t s);

// ---- padded-pitch 3x3 stride-1 convolution (conv_pp.hip), bf16 only ----------------
long blt_pp_pixels(int N, int H, int W);                 // N*(H+1)*(W+1) positions (without the guards)
int blt_conv3x3_pp_stat_rows(int N, int H, int W);       // upper bound of the partial rows written to stat_sum / stat_sq (any tile plan)
int blt_conv3x3_pp_stat_rows_for(int N, int H, int W, int Cin, int Cout);      // exactly the rows the launch for this layer writes
// in_scale / in_shift (both or neither, [Cin] floats): x is the previous convolution's RAW output, its BatchNorm + ReLU is applied to
// the staged input patch in LDS (pad positions become the zeros the taps expect) — saves the bn_apply_pp pass between two convolutions
int blt_conv3x3_pp(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, float* stat_sum, float* stat_sq,
                   hipStream_t s, const float* in_scale = nullptr, const float* in_shift = nullptr);
// 7x7/2 stem on the zero-bordered NHWC4 image with the patch + filter staged in LDS (bf16, Cout = 64, Ho % 8 == 0, Wo % 16 == 0)
// stem + 3x3/2 max-pool in one launch: writes the pooling-window EXTREMUM (max where gamma >= 0, min where gamma < 0) of the raw
// convolution output into the pooled PP tensor + the BatchNorm partial sums of all outputs; bn_apply_pp(relu) finishes it
bool blt_conv_stem_pool_ok(int dtype, int H, int W, int Hp, int Wp, int Cout);
int blt_conv_stem_pool_stat_rows(int N, int H, int W);
int blt_conv_stem_pool(const void* x_padded, const void* w, const float* gamma, void* y_pool_pp, int N, int H, int W, int Hp, int Wp, float* stat_sum,
                       float* stat_sq, hipStream_t s);
bool blt_conv_stem_direct_ok(int dtype, int H, int W, int Hp, int Wp, int Cout);
int blt_conv_stem_direct_stat_rows(int N, int H, int W);
int blt_conv_stem_direct(const void* x_padded, const void* w, void* y, int N, int H, int W, int Hp, int Wp, float* stat_sum, float* stat_sq,
                         hipStream_t s);

// ---- normalisation -----------------------------------------------------------------
// pad_period / pad_valid (0 / 0 = none): column c is a real feature iff c % pad_period < pad_valid; the other columns are zero pads that
// carry gamma = beta = 0, do not count in mean / variance and get a zero gradient (the reference's default widths: 4 heads of 75 stored
// as 4 x 80, models.IQ pads them)
int blt_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      long rows, int cols, float eps, hipStream_t s, int pad_period = 0, int pad_valid = 0, long ld = 0);
// dx = LNbwd(dy) (+ dres if non-null); dgamma/dbeta are ACCUMULATED (+=) with float atomics
// optional second output out2 = (maskY != 0) ? dx * mask_scale : 0 (the ReLU/dropout backward that consumes dx, fused)
// partials (optional, blt_layernorm_bwd_grid(rows, cols) * 2 * cols floats): the workgroups' dgamma / dbeta sums are stored there
// instead of being added to dgamma / dbeta; blt_ln_param_reduce adds them later (one launch for many LayerNorms, off the chain)
int blt_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const void* dres, void* dx, float* dgamma, float* dbeta, long rows, int cols, hipStream_t s,
                      const void* maskY = nullptr, float mask_scale = 1.f, void* out2 = nullptr, float* partials = nullptr, int pad_period = 0,
                      int pad_valid = 0, long ld = 0,      // ld: row stride (elements) of every row-indexed tensor, 0 = cols
                      const float* beta = nullptr, void* xn_out = nullptr);      // xn_out: also store the LayerNorm's forward output (needs beta)
int blt_layernorm_bwd_grid(long rows, int cols);
#define BLT_LN_RED_MAX 24
struct LnRed { const float* part; float* dgamma; float* dbeta; int nblocks; int cols; };
struct LnRedArgs { LnRed e[BLT_LN_RED_MAX]; int n; };
int blt_ln_param_reduce(const LnRedArgs& a, hipStream_t s);

// BatchNorm2d (train mode) on NHWC: finalize partial sums -> scale/shift (+ running stat update)
int blt_bn_finalize(const float* psum, const float* psq, int nparts, int C, long count, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                    float* scale, float* shift, float* save_mean, float* save_var, double* scratch, hipStream_t s);
int blt_bn_scratch_doubles(int C);   // doubles of scratch blt_bn_finalize needs
// y = [relu]( x*scale[c] + shift[c] (+ res) ), in place allowed
int blt_bn_apply(int dtype, const void* x, const float* scale, const float* shift, const void* res, void* y, long rows,
                 int C, int relu, hipStream_t s);
// y[n,ho,wo,c] = max 3x3/2 pad1 of relu(x*scale+shift)
int blt_bn_relu_maxpool(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi,
                        int C, hipStream_t s);
int blt_avgpool(int dtype, const void* x, void* y, int N, int HW, int C, int out_f32, hipStream_t s);
// padded-pitch (PP) variants: activations [N][H+1][W+1][C] with zero pad pixels (see ConvGeom)
// res (optional) is added before the ReLU; with res_scale / res_shift it is a raw convolution output normalised on the fly
int blt_bn_apply_pp(int dtype, const void* x, const float* scale, const float* shift, const void* res, const float* res_scale,
                    const float* res_shift, void* y, int N, int H, int W, int C, int relu, hipStream_t s);
int blt_bn_relu_maxpool_pp(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi, int C,
                           hipStream_t s);
int blt_avgpool_pp(int dtype, const void* x, void* y, int N, int H, int W, int C, int out_f32, hipStream_t s);
// BatchNorm1d over the batch (train mode); saves mean / rstd; updates running stats (unbiased var)
int blt_bn1d_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                 float* running_mean, float* running_var, int B, int C, float eps, float momentum, hipStream_t s);
int blt_bn1d_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                 void* dx, float* dgamma, float* dbeta, int B, int C, hipStream_t s);

// ---- attention -----------------------------------------------------------------------
struct AttnArgs {
    const void* Q = nullptr; const void* K = nullptr; const void* V = nullptr;   // row strides in elements
    int ldq = 0, ldk = 0, ldv = 0;
    void* O = nullptr; int ldo = 0;
    const int* key_ids = nullptr;   // [B, Tk] token ids; id == 0 -> key masked
    int B = 0, heads = 0, Tq = 0, Tk = 0, d = 0;
    // forward only: rows per batch element of the Q / O tensors and of the K / V / key_ids tensors when only the first Tq / Tk of them take
    // part (0 = Tq / Tk).  Incremental decoding: Q points at the newest row of a [B, T, ld] projection (Tq = 1, q_rows = T), K / V at row 0
    // of the rows written so far (Tk = t + 1, k_rows = T) — the earlier steps' projections are the key / value cache.
    int q_rows = 0, k_rows = 0;
    int causal = 0;
    float scale = 1.f;
    float drop_p = 0.f; uint64_t seed = 0; uint32_t stream_id = 0;
    // backward only
    const void* dO = nullptr; int lddo = 0;
    void* dQ = nullptr; void* dK = nullptr; void* dV = nullptr; int lddq = 0, lddk = 0, lddv = 0;
    int accumulate_dkv = 0;
    // fused forward (blt_attn_out_fwd): Y[B*Tq, H] = O Wo^T + R, with O (the attention context, H = heads * d) also written as usual
    const void* Wo = nullptr; int ldwo = 0;      // [H, ldwo] bf16, k-contiguous (the Linear's weight)
    const void* R = nullptr; int ldr = 0;        // residual rows [B*Tq, ldr] bf16
    void* Y = nullptr; int ldy = 0;
};
int blt_attn_fwd(int dtype, const AttnArgs& a, hipStream_t s);
// MultiHeadAttention core + its output Linear + the residual add of the surrounding sub-layer in ONE launch (transformer_layers.py:494-532
// + the `+ x` of :139,208,267,274): one workgroup per batch element, one wave per head; bf16, d = 64, heads <= 8, Tq, Tk <= 32.  The
// context O is still written (backward's weight gradient reads it).  Bit-identical to blt_attn_fwd followed by the planned-tile GEMM.
// Measured SLOWER in the step than the two launches (every workgroup streams the whole weight): an operator for callers with few rows, not the engine's path.
bool blt_attn_out_fwd_ok(int dtype, const AttnArgs& a);
int blt_attn_out_fwd(int dtype, const AttnArgs& a, hipStream_t s);
int blt_attn_bwd(int dtype, const AttnArgs& a, hipStream_t s);

// ---- embedding / token plumbing --------------------------------------------------------
// out[m, 0:E] = table[ids[m], :] ; out[m, E:ld] = 0
int blt_embed_gather(int dtype, const float* table, const int* ids, void* out, long rows, int E, int ld, hipStream_t s);
// dtable[ids[m], :] += d[m, 0:E]  for ids[m] != pad
int blt_embed_scatter(int dtype, const void* d, int ld, const int* ids, float* dtable, long rows, int E, int pad_id,
                      hipStream_t s);
// builds int32 token streams from the int64 batch tensors (see engine.hip)
int blt_prep_tokens(const long long* ctx, const long long* post, const long long* tgt, int B, int Sa, int Sp, int T,
                    int* ids_all, int* pos_all, int* tgt_shift, int* tgt32, int* ctx32, int* post32, float* counters, int V,
                    float* bad_ids /* += number of ids outside [0, V); they are replaced by <pad> */, hipStream_t s);

// ---- elementwise ---------------------------------------------------------------------------
// y[b*ystride + j] (+)= a[b*astride + j] (+ c[b*cstride + j]) for j < n  (row-0 injections and their gradients)
int blt_rows_add(int dtype, void* y, long ystride, const void* a, long astride, const void* c, long cstride, int B, int n,
                 int accumulate, hipStream_t s);
// same, one wave per row, and stat[b * stat_stride + 0..1] = {sum, sum of squares} of result row b as stored (GemmArgs::fold_stat)
int blt_rows_add_stat(int dtype, void* y, long ystride, const void* a, long astride, const void* c, long cstride, int B, int n, int accumulate,
                      float* stat, long stat_stride, int np, hipStream_t s);      // stat[b * stat_stride + 0..1] = the sums, slots 1 .. np-1 of the row zeroed
// LayerNorm folded into its consumer Linear: W' = bf16(W diag(gamma)) at the weight's offset in wfold_bf16, fold_s / fold_c rows at
// the entry's srow (misc.hip::FoldEnt table on the device, `row0` = prefix sum of rows)
struct BltFoldEnt { long w_off, g_off, b_off, bias_off; int rows, K, srow, row0; };
int blt_ln_fold_prepare(const float* train, void* wfold_bf16, float* fold_s, float* fold_c, const void* table_dev, int nent, int total_rows, hipStream_t s);
int blt_ln_fold_prepare_one(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf_bf16, float* fold_s, float* fold_c, int N,
                            int K, hipStream_t s);
// y = dy * (ymask != 0) * scale
// d_feats += dx0 + g_zc; d_zproj = dx0 + g_rin + g_zc (if non-null); d_enc[:,0] += g_rin   (misc.hip: the row-0 injections' backward)
int blt_row0_sums(int dtype, const void* dx0, long sdx, const void* g_rin, const void* g_zc, void* d_feats, void* d_zproj, void* d_enc, long senc,
                  int B, int n, hipStream_t s);
int blt_mask_scale(int dtype, const void* dy, const void* ymask, void* y, long n, float scale, hipStream_t s);
// out[n] (+)= sum_m x[m, n]
int blt_colsum(int dtype, const void* x, int ld, long M, int N, float* out, int accumulate, hipStream_t s);
int blt_cast_pad(const float* src, int rows, int cols, void* dst, int ld, int dtype, hipStream_t s);
int blt_cast_rows(int dtype_src, const void* src, int lds_, int dtype_dst, void* dst, int ldd, long rows, int cols,
                  hipStream_t s);
// NCHW fp32 -> NHWC [N, Hp, Wp, Cpad] with the image at (pad_top, pad_left) and zeros elsewhere
int blt_img_pack(int dtype, const float* nchw, void* nhwc, int N, int C, int H, int W, int Cpad, int pad_top, int pad_left, int Hp, int Wp,
                 hipStream_t s);
// [Cout, Cin, KH, KW] fp32 -> [Cout, KH, KWpad, Cpad] (zeros in the padding)
int blt_conv_pack_w(int dtype, const float* w, void* out, int Cout, int Cin, int KH, int KW, int Cpad, int KWpad, hipStream_t s);
int blt_copy2d(int dtype, const void* src, int lds_, void* dst, int ldd, long rows, int cols, hipStream_t s);
// table_dev: int4 {element offset, rows, cols, first 64x64 tile} per matrix; dst_bf16 may be null (transposed copy only)
int blt_shadow_transpose(const float* src, void* dst_bf16, void* dstT_bf16, const void* table_dev, int nent, int total_tiles, hipStream_t s, int tile_base = 0);
// transposed shadows only, from the plain bf16 shadow (kept current by blt_adam_step's shadow output)
int blt_shadow_transpose_bf16(const void* src_bf16, void* dstT_bf16, const void* table_dev, int nent, int total_tiles, hipStream_t s, int tile_base = 0);

// ---- losses ------------------------------------------------------------------------------------
// token CE with ignore_index=0, mean over non-pad targets (count from counters[0]); writes d(logits) IN PLACE scaled by
// gscale/count; loss_out += sum(-logp)/count.  logits [M, ld] with pad columns zeroed in the gradient.
int blt_ce_fwd_bwd(int dtype, void* logits, int ld, const int* target, long M, int V, const float* count, float gscale,
                   float* loss_out, int write_grad, hipStream_t s);
// bag-of-words CE: one logit row per sample against T targets (train_iq.py:92-94)
int blt_bow_ce_fwd_bwd(int dtype, const void* zlogit, int ld, const int* target, int B, int T, int V, const float* count,
                       float gscale, float* loss_out, void* dz, hipStream_t s);
// mse = mean((a-b)^2); da = gscale*2(a-b)/n ; db = -da   (train_iq.py:84: gradient flows to both arguments)
// n_div (0 = n): the divisor of the mean when the n elements include zero pads that nn.MSELoss would not have seen
int blt_mse_fwd_bwd(int dtype, const void* a, const void* b, long n, float gscale, float* loss_out, void* da, void* db,
                    hipStream_t s, long n_div = 0);
// reparameterisation + KL (transformer_layers.py:41-59, 536-540)
int blt_latent_fwd(int dtype, const void* mlv_p, const void* mlv_q, const float* eps, void* z, float* kld_out, int B,
                   int Z, int ld, hipStream_t s);
int blt_latent_bwd(int dtype, const void* mlv_p, const void* mlv_q, const float* eps, const void* dz, float kld_gscale,
                   void* dmlv_p, void* dmlv_q, int B, int Z, int ld, hipStream_t s);

// ---- region-attention pooling (bottom-up mode) ---------------------------------------------------------
// P [B, R, H] projected regions; w [H]; out [B, H] fp32; alpha [B, R] fp32 (saved for backward); dP [B, R, H]; dw [H] += (atomics)
int blt_region_attn_fwd(int dtype, const void* P, const float* w, float* out, float* alpha, int B, int R, int H, hipStream_t s);
int blt_region_attn_bwd(int dtype, const void* P, const float* w, const float* alpha, const float* dout, void* dP, float* dw, int B, int R, int H,
                        hipStream_t s);

// ---- greedy decoding ----------------------------------------------------------------------------------
int blt_prep_decode(const long long* ctx, int B, int Sa, int T, int* ids_all, int* pos_all, int* ctx32, int V, float* bad_ids, hipStream_t s);
int blt_argmax_top6(int dtype, const void* logits, int ld, int B, int V, int t, int T, int* ys, int* tokens, int* top_idx, float* top_val,
                    hipStream_t s);
int blt_bn_eval_scale(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, float* scale, float* shift, int C,
                      hipStream_t s);

// ---- optimiser ---------------------------------------------------------------------------------
int blt_sumsq(const float* x, long n, float* out /* += */, hipStream_t s);
// clip_grad_norm_(max_norm) + Adam (torch defaults) over a flat fp32 buffer; gnorm_sq is a device scalar
int blt_adam_step(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq, float max_norm, float lr,
                  float beta1, float beta2, float eps, int step, hipStream_t s, void* shadow_bf16 = nullptr /* bf16 mirror of p, written in the same pass */);
int blt_dropout_mask(uint64_t seed, uint32_t stream_id, long rows, int cols, int ld_index, float p, unsigned char* out,
                     hipStream_t s);

// batch producer (batch.hip)
int blt_image_store_u8(const float* images, uint8_t* out, long count, hipStream_t s);
int blt_batch_rows(const int* questions, const int* answers, const int* answer_types, const int* cat_word_ids, int n_cat, long n_rows,
                   const long* index, int B, int q_len, int a_len, long* oq, long* op, long* oa, long* ot, long* oti, hipStream_t s);
int blt_batch_images_packed(const uint8_t* table, long n_images, int S, const int* image_indices, long n_rows, const long* index,
                            const int* boxes, const int* coeffs, int KS, int B, int osz, const float* mean_std, int dtype, void* out, int Hp,
                            int Wp, int pad_top, int pad_left, hipStream_t s);
int blt_batch_images(const uint8_t* table, long n_images, int S, const int* image_indices, long n_rows, const long* index, const int* boxes,
                     const int* coeffs, int KS, int B, int osz, const float* mean_std, float* out, uint8_t* out_u8, hipStream_t s);
