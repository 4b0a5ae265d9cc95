// extern "C" operator-level entry points (see include/bltvqg_hip.h): thin argument marshalling over kernels.h.
#include <vector>
#include "kernels.h"
#include "../../include/bltvqg_hip.h"
#ifdef BLT_EXPERIMENTS
#include "../../include/bltvqg_hip_experiments.h"
#endif

static inline int ilog2i(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

extern "C" {

void bltvqg_debug_set(int key, int value) { blt_debug_set(key, value); }
int bltvqg_debug_get(int key) { return blt_debug_get(key); }
int bltvqg_build_has_ablations(void) {
#ifdef BLT_ABLATE
    return 1;
#else
    return 0;
#endif
}

int bltvqg_gemm(int dtype, const void* A, int lda, int transA, const void* B, int ldb, int transB, void* C, int ldc, int M, int N, int K,
                const float* bias, int relu, float drop_p, uint64_t seed, uint32_t stream_id, const void* maskY, int ldm, float mask_scale,
                const void* R, int ldr, int accumulate, int out_f32, int force_tile, int split_k, void* stream) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.transA = transA; g.B = B; g.ldb = ldb; g.transB = transB; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.relu = relu; g.drop_p = drop_p; g.seed = seed; g.stream_id = stream_id; g.maskY = maskY; g.ldm = ldm;
    g.mask_scale = mask_scale; g.R = R; g.ldr = ldr; g.accumulate = accumulate; g.out_f32 = out_f32; g.force_tile = force_tile; g.split_k = split_k;
    return blt_gemm(dtype, g, (hipStream_t)stream);
}

int bltvqg_gemm_ex(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, const float* rowtab,
                   const int32_t* rowidx, int ldt, int relu, float drop_p, uint64_t seed, uint32_t stream_id, const void* maskY, int ldm, float mask_scale,
                   void* C2, int ldc2, const void* R, int ldr, int accumulate, int tile_m, int tile_n, void* stream) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.rowtab = rowtab; g.rowidx = rowidx; g.ldt = ldt; g.relu = relu; g.drop_p = drop_p; g.seed = seed; g.stream_id = stream_id;
    g.maskY = maskY; g.ldm = ldm; g.mask_scale = mask_scale; g.C2 = C2; g.ldc2 = ldc2; g.R = R; g.ldr = ldr; g.accumulate = accumulate;
    if (tile_m < 0) { g.no_dma = 0; g.force_tile = 64; return blt_gemm(BLT_BF16, g, (hipStream_t)stream); }      // round-1 kernel (64x64 ring)
    BLT_REQUIRE((tile_m == 0 && tile_n == 0) || (tile_m > 0 && tile_n > 0), "gemm_ex: tile_m / tile_n must both be 0 or both be a compiled tile shape");
    g.nt2_bm = tile_m; g.nt2_bn = tile_n;
    BLT_REQUIRE(blt_gemm_nt2_ok(BLT_BF16, g),
                "gemm_ex: operands do not fit the planned-tile kernel (bf16 NT, lda/ldb %% 8 == 0, M >= 256 unless a tile shape is given)");
    // through blt_gemm: the same operand validation as every other entry point (pitches >= round8(K), 16-byte aligned epilogue operands,
    // rowtab needs rowidx, dropout range); a bad call is BLT_ERR_ARG, not an out-of-bounds LDS-DMA read on the device
    g.nt2_bm = tile_m; g.nt2_bn = tile_n;
    return blt_gemm(BLT_BF16, g, (hipStream_t)stream);
}

int bltvqg_gemm_rowstat(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, int relu, float drop_p,
                        uint64_t seed, uint32_t stream_id, void* C2, int ldc2, const void* R, int ldr, float* out_stat, int stat_slots, int tile_m, int tile_n,
                        void* stream) {
    BLT_REQUIRE(out_stat != nullptr && stat_slots >= 1, "gemm_rowstat: out_stat is null / stat_slots < 1");
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.relu = relu; g.drop_p = drop_p; g.seed = seed; g.stream_id = stream_id; g.C2 = C2; g.ldc2 = ldc2; g.R = R; g.ldr = ldr;
    g.out_stat = out_stat; g.stat_slots = stat_slots;
    BLT_REQUIRE(blt_gemm_nt2_ok(BLT_BF16, g), "gemm_rowstat: operands do not fit the planned-tile kernel (bf16 NT, lda/ldb %% 8 == 0)");
    BLT_REQUIRE((tile_m == 0 && tile_n == 0) || (tile_m > 0 && tile_n > 0), "gemm_rowstat: tile_m / tile_n must both be 0 or both be a compiled tile shape");
    g.nt2_bm = tile_m; g.nt2_bn = tile_n;
    return blt_gemm(BLT_BF16, g, (hipStream_t)stream);
}

int bltvqg_gemm_rowstat_parts(int M, int N, int K, int tile_n) {
    (void)M; (void)K; (void)tile_n;      // one slot per 64 columns of the row, whatever the tile
    return N > 0 ? cdiv(N, 64) : 0;
}

int bltvqg_ln_fold_prepare(const float* W, int N, int K, const float* gamma, const float* beta, const float* bias, void* Wf_bf16, float* fold_s,
                           float* fold_c, void* stream) {
    return blt_ln_fold_prepare_one(W, gamma, beta, bias, Wf_bf16, fold_s, fold_c, N, K, (hipStream_t)stream);
}

int bltvqg_linear_ln_folded(const void* X, int ldx, const void* Wf, int ldw, void* Y, int ldy, int M, int N, int K, const float* fold_s, const float* fold_c,
                            const float* row_stat, int stat_slots, int stat_parts, float* mean, float* rstd, float eps, int relu, float drop_p, uint64_t seed,
                            uint32_t stream_id, int tile_m, int tile_n, void* stream) {
    BLT_REQUIRE(X && Wf && Y && fold_s && fold_c && row_stat && M > 0 && N > 0 && K > 0, "linear_ln_folded: null pointer / bad shape");
    BLT_REQUIRE(stat_parts >= 1 && stat_parts <= stat_slots, "linear_ln_folded: stat_parts must be in [1, stat_slots]");
    BLT_REQUIRE((mean == nullptr) == (rstd == nullptr), "linear_ln_folded: mean and rstd go together");
    GemmArgs g;
    g.A = X; g.lda = ldx; g.B = Wf; g.ldb = ldw; g.C = Y; g.ldc = ldy; g.M = M; g.N = N; g.K = K;
    g.relu = relu; g.drop_p = drop_p; g.seed = seed; g.stream_id = stream_id;
    g.stat_slots = stat_slots; g.fold_np = stat_parts;
    g.fold_s = fold_s; g.fold_c = fold_c; g.fold_stat = row_stat; g.fold_mean = mean; g.fold_rstd = rstd; g.fold_eps = eps; g.fold_n = (float)K;
    BLT_REQUIRE(blt_gemm_nt2_ok(BLT_BF16, g), "linear_ln_folded: operands do not fit the planned-tile kernel (bf16 NT, ldx/ldw %% 8 == 0)");
    BLT_REQUIRE((tile_m == 0 && tile_n == 0) || (tile_m > 0 && tile_n > 0), "linear_ln_folded: tile_m / tile_n must both be 0 or both be a compiled tile shape");
    g.nt2_bm = tile_m; g.nt2_bn = tile_n;
    return blt_gemm(BLT_BF16, g, (hipStream_t)stream);
}

#ifdef BLT_EXPERIMENTS
int bltvqg_linear_pair(const bltvqg_linear_desc* p1, const bltvqg_linear_desc* p2, int N, int K, int relu, float drop_p, uint64_t seed, float mask_scale,
                       int stat_slots, int stat_parts, float eps, int tile_m, int tile_n, void* stream) {
    BLT_REQUIRE(p1 && p2 && N > 0 && K > 0, "linear_pair: null descriptor / bad shape");
    BLT_REQUIRE((tile_m == 0 && tile_n == 0) || (tile_m > 0 && tile_n > 0), "linear_pair: tile_m / tile_n must both be 0 or both be a compiled tile shape");
    GemmArgs g[2];
    const bltvqg_linear_desc* d[2] = {p1, p2};
    for (int i = 0; i < 2; ++i) {
        GemmArgs& a = g[i];
        const bltvqg_linear_desc& q = *d[i];
        BLT_REQUIRE(q.A && q.W && q.C && q.M > 0, "linear_pair: problem %d: null operand / no rows", i + 1);
        BLT_REQUIRE((q.mean == nullptr) == (q.rstd == nullptr), "linear_pair: problem %d: mean and rstd go together", i + 1);
        a.A = q.A; a.lda = q.lda; a.B = q.W; a.ldb = q.ldw; a.C = q.C; a.ldc = q.ldc; a.M = q.M; a.N = N; a.K = K;
        a.bias = q.bias; a.maskY = q.maskY; a.ldm = q.ldm; a.mask_scale = mask_scale; a.C2 = q.C2; a.ldc2 = q.ldc2; a.R = q.R; a.ldr = q.ldr;
        a.relu = relu; a.drop_p = drop_p; a.seed = seed; a.stream_id = q.stream_id;
        a.fold_s = q.fold_s; a.fold_c = q.fold_c; a.fold_stat = q.row_stat; a.fold_mean = q.mean; a.fold_rstd = q.rstd; a.fold_eps = eps; a.fold_n = (float)K;
        a.fold_np = stat_parts > 0 ? stat_parts : 1; a.out_stat = q.out_stat; a.stat_slots = stat_slots > 0 ? stat_slots : 1;
        a.nt2_bm = tile_m; a.nt2_bn = tile_n;
        const int rc = blt_gemm_validate(BLT_BF16, a);
        if (rc != BLT_OK) return rc;
    }
    BLT_REQUIRE(blt_gemm_nt2_pair_ok(BLT_BF16, g[0], g[1]),
                "linear_pair: the two problems must both fit the planned-tile kernel (bf16 NT, ld %% 8 == 0) and carry the same epilogue terms");
    return blt_gemm_nt2_pair(g[0], g[1], (hipStream_t)stream);
}
#endif

int bltvqg_linear_wgrad_group(int n, const void* const* dY, const int32_t* ldy, const void* const* X, const int32_t* ldx, float* const* dW,
                              const int32_t* ldw, float* const* dbias, const int32_t* rows, const int32_t* N, const int32_t* K, void* table_dev,
                              int64_t table_bytes, void* stream) {
    BLT_REQUIRE(n > 0 && n < 250 && dY && ldy && X && ldx && dW && ldw && dbias && rows && N && K && table_dev, "linear_wgrad_group: bad args");
    std::vector<GemmArgs> gs((size_t)n);
    for (int i = 0; i < n; ++i) {
        GemmArgs& g = gs[i];
        g.A = dY[i]; g.lda = ldy[i]; g.transA = 1; g.B = X[i]; g.ldb = ldx[i]; g.transB = 1; g.C = dW[i]; g.ldc = ldw[i];
        g.M = N[i]; g.N = K[i]; g.K = rows[i]; g.out_f32 = 1; g.a_rowsum = dbias[i];
        BLT_REQUIRE(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K > 0 && g.ldc >= g.N, "linear_wgrad_group: bad problem %d", i);
        BLT_REQUIRE(blt_wgrad_group_ok(BLT_BF16, g), "linear_wgrad_group: problem %d does not fit the grouped kernel (bf16, ld %% 8 == 0, 16-byte aligned)", i);
    }
    std::vector<blt_wg_problem> probs;
    std::vector<int> wg0;
    int bm = 128;
    const int nwg = blt_wgrad_group_plan(gs, probs, wg0, &bm);
    const size_t pb = probs.size() * sizeof(blt_wg_problem), pb_al = (pb + 63) / 64 * 64;
    BLT_REQUIRE((int64_t)(pb_al + wg0.size() * 4) <= table_bytes, "linear_wgrad_group: table_dev too small (%lld bytes needed)", (long long)(pb_al + wg0.size() * 4));
    // synchronous copies: the host vectors die at return
    if (hipMemcpy(table_dev, probs.data(), pb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy((char*)table_dev + pb_al, wg0.data(), wg0.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        blt_set_error("linear_wgrad_group: table upload failed");
        return BLT_ERR_HIP;
    }
    return blt_wgrad_group_launch((const blt_wg_problem*)table_dev, (const int*)((char*)table_dev + pb_al), n, nwg, bm, (hipStream_t)stream);
}

#ifdef BLT_EXPERIMENTS
int bltvqg_gemm_repeat(int dtype, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, int relu,
                       const void* R, int ldr, int reps, void* stream) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.relu = relu; g.R = R; g.ldr = ldr;
    for (int i = 0; i < reps; ++i) {
        const int rc = blt_gemm(dtype, g, (hipStream_t)stream);
        if (rc) return rc;
    }
    return BLT_OK;
}
#endif

#ifdef BLT_EXPERIMENTS
int bltvqg_gemm_rotate(int dtype, const void* A, int lda, int a_copies, int64_t a_stride_bytes, const void* B, int ldb, int b_copies,
                       int64_t b_stride_bytes, void* C, int ldc, int c_copies, int64_t c_stride_bytes, int M, int N, int K, int chain, int reps,
                       void* stream) {
    BLT_REQUIRE(a_copies >= 1 && b_copies >= 1 && c_copies >= 1 && reps > 0 && (!chain || (N == K && c_copies >= 2)), "gemm_rotate: bad args");
    GemmArgs g;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    for (int i = 0; i < reps; ++i) {
        g.B = (const char*)B + (int64_t)(i % b_copies) * b_stride_bytes;
        if (chain) {      // launch i reads what launch i-1 wrote: the dependent-chain case (operands freshly written by another kernel)
            g.A = i == 0 ? A : (const void*)((char*)C + (int64_t)((i - 1) % c_copies) * c_stride_bytes);
            g.lda = i == 0 ? lda : ldc;
        } else {
            g.A = (const char*)A + (int64_t)(i % a_copies) * a_stride_bytes;
        }
        g.C = (char*)C + (int64_t)(i % c_copies) * c_stride_bytes;
        const int rc = blt_gemm(dtype, g, (hipStream_t)stream);
        if (rc) return rc;
    }
    return BLT_OK;
}
#endif

int bltvqg_linear_wgrad(int dtype, const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw, float* dbias, int rows, int N, int K,
                        int split_k, void* stream) {
    BLT_REQUIRE(dY && X && dW && rows > 0 && N > 0 && K > 0, "linear_wgrad: bad args");
    GemmArgs g;
    g.A = dY; g.lda = ldy; g.transA = 1; g.B = X; g.ldb = ldx; g.transB = 1; g.C = dW; g.ldc = ldw; g.M = N; g.N = K; g.K = rows;
    g.out_f32 = 1; g.accumulate = 1; g.split_k = split_k; g.a_rowsum = dbias;
    return blt_gemm(dtype, g, (hipStream_t)stream);
}

#ifdef BLT_EXPERIMENTS
int bltvqg_linear_layernorm(const void* X, int ldx, const void* W, int ldw, const float* bias, int relu, float drop_p, uint64_t seed,
                            uint32_t stream_id, void* C2, int ldc2, const void* R, int ldr, void* C, int ldc, const float* ln_gamma,
                            const float* ln_beta, float ln_eps, void* ln_out, float* ln_mean, float* ln_rstd, int M, int N, int K, void* stream) {
    BLT_REQUIRE(X && W && C && ln_out, "linear_layernorm: null pointer");
    GemmArgs g;
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.relu = relu; g.drop_p = drop_p; g.seed = seed; g.stream_id = stream_id; g.C2 = C2; g.ldc2 = ldc2; g.R = R; g.ldr = ldr;
    g.ln_gamma = ln_gamma; g.ln_beta = ln_beta; g.ln_eps = ln_eps; g.ln_out = ln_out; g.ln_mean = ln_mean; g.ln_rstd = ln_rstd;
    return blt_gemm(BLT_BF16, g, (hipStream_t)stream);
}
#endif

#ifdef BLT_EXPERIMENTS
int bltvqg_layernorm_linear(const void* X, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, void* Xn, float* ln_mean,
                            float* ln_rstd, const void* W, int ldw, const float* bias, int relu, float drop_p, uint64_t seed, uint32_t stream_id,
                            const void* R, int ldr, void* C, int ldc, int M, int N, int K, void* stream) {
    BLT_REQUIRE(X && W && C && Xn, "layernorm_linear: null pointer");
    GemmArgs g;
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.relu = relu; g.drop_p = drop_p; g.seed = seed; g.stream_id = stream_id; g.R = R; g.ldr = ldr;
    g.lnA_gamma = ln_gamma; g.lnA_beta = ln_beta; g.lnA_eps = ln_eps; g.lnA_out = Xn; g.lnA_mean = ln_mean; g.lnA_rstd = ln_rstd;
    return blt_gemm(BLT_BF16, g, (hipStream_t)stream);
}
#endif

static GemmArgs conv_args(const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    GemmArgs g;
    const int Ho = (Hi + 2 * pad - KH) / stride + 1, Wo = (Wi + 2 * pad - KW) / stride + 1;
    g.A = x; g.B = w; g.C = y;
    g.M = N * Ho * Wo; g.N = Cout; g.K = KH * KW * Cin;
    g.lda = Cin; g.ldb = g.K; g.ldc = Cout; g.is_conv = 1;
    g.cg.Hi = Hi; g.cg.Wi = Wi; g.cg.Cin = Cin; g.cg.cin_log2 = ilog2i(Cin); g.cg.Ho = Ho; g.cg.Wo = Wo; g.cg.KH = KH; g.cg.KW = KW;
    g.cg.stride = stride; g.cg.pad = pad;
    return g;
}

int bltvqg_conv2d(int dtype, const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride,
                  int pad, float* stat_sum, float* stat_sq, void* stream) {
    BLT_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "conv2d: bad sizes");
    BLT_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "conv2d: stat_sum and stat_sq go together");
    GemmArgs g = conv_args(x, w, y, N, Hi, Wi, Cin, Cout, KH, KW, stride, pad);
    g.stat_sum = stat_sum; g.stat_sq = stat_sq;
    return blt_gemm(dtype, g, (hipStream_t)stream);
}

/* ---- padded-pitch (PP) activations ---- */
int64_t bltvqg_pp_pixels(int N, int H, int W) { return (int64_t)blt_pp_pixels(N, H, W); }
int bltvqg_pp_guard_front(void) { return BLT_PP_GUARD_FRONT; }
int bltvqg_pp_guard_tail(void) { return BLT_PP_GUARD_TAIL; }

int bltvqg_conv3x3_pp(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, float* stat_sum, float* stat_sq,
                      void* stream) {
    return blt_conv3x3_pp(x, w, y, N, H, W, Cin, Cout, stat_sum, stat_sq, (hipStream_t)stream);
}
int bltvqg_conv3x3_pp_bn_relu_in(const void* x_raw, const float* in_scale, const float* in_shift, const void* w, void* y, int N, int H, int W, int Cin,
                                 int Cout, float* stat_sum, float* stat_sq, void* stream) {
    BLT_REQUIRE(in_scale && in_shift, "conv3x3_pp_bn_relu_in: null scale / shift");
    return blt_conv3x3_pp(x_raw, w, y, N, H, W, Cin, Cout, stat_sum, stat_sq, (hipStream_t)stream, in_scale, in_shift);
}
int bltvqg_conv3x3_pp_stat_rows(int N, int H, int W) { return blt_conv3x3_pp_stat_rows(N, H, W); }

static GemmArgs conv_args_pp(const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad,
                             int in_pp, int out_pp) {
    GemmArgs g = conv_args(x, w, y, N, Hi, Wi, Cin, Cout, KH, KW, stride, pad);
    const int Ho = g.cg.Ho, Wo = g.cg.Wo;
    g.cg.in_rows = Hi + (in_pp ? 1 : 0); g.cg.in_pitch = Wi + (in_pp ? 1 : 0);
    g.cg.Hov = Ho; g.cg.Wov = Wo;
    g.cg.Ho = Ho + (out_pp ? 1 : 0); g.cg.Wo = Wo + (out_pp ? 1 : 0);
    g.M = N * g.cg.Ho * g.cg.Wo;
    return g;
}

int bltvqg_conv2d_pp(int dtype, const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride,
                     int pad, int in_pp, int out_pp, float* stat_sum, float* stat_sq, void* stream) {
    BLT_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "conv2d_pp: bad sizes");
    BLT_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "conv2d_pp: stat_sum and stat_sq go together");
    GemmArgs g = conv_args_pp(x, w, y, N, Hi, Wi, Cin, Cout, KH, KW, stride, pad, in_pp, out_pp);
    g.stat_sum = stat_sum; g.stat_sq = stat_sq;
    return blt_gemm(dtype, g, (hipStream_t)stream);
}
int bltvqg_conv2d_pp_stat_rows(int dtype, int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int out_pp) {
    GemmArgs g = conv_args_pp((const void*)16, (const void*)16, (void*)16, N, Hi, Wi, Cin, Cout, KH, KW, stride, pad, 0, out_pp);
    return blt_gemm_stat_rows(g, dtype);
}
int bltvqg_bn_apply_pp(int dtype, const void* x, const float* scale, const float* shift, const void* res, void* y, int N, int H, int W, int C,
                       int relu, void* stream) {
    return blt_bn_apply_pp(dtype, x, scale, shift, res, nullptr, nullptr, y, N, H, W, C, relu, (hipStream_t)stream);
}
int bltvqg_bn_relu_maxpool_pp(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi, int C,
                              void* stream) {
    return blt_bn_relu_maxpool_pp(dtype, x, scale, shift, y, N, Hi, Wi, C, (hipStream_t)stream);
}
int bltvqg_avgpool_pp(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream) {
    return blt_avgpool_pp(dtype, x, y, N, H, W, C, 0, (hipStream_t)stream);
}

int bltvqg_conv2d_stat_rows(int N, int Hi, int Wi, int Cout, int KH, int KW, int stride, int pad) {
    GemmArgs g = conv_args(nullptr, nullptr, nullptr, N, Hi, Wi, 8, Cout, KH, KW, stride, pad);
    return blt_gemm_stat_rows(g, BLT_BF16);
}

int bltvqg_img_pack(int dtype, const float* nchw, void* nhwc, int N, int C, int H, int W, int Cpad, int pad_top, int pad_left, int Hp, int Wp,
                    void* stream) {
    return blt_img_pack(dtype, nchw, nhwc, N, C, H, W, Cpad, pad_top, pad_left, Hp, Wp, (hipStream_t)stream);
}
int bltvqg_conv_pack_w(int dtype, const float* w, void* out, int Cout, int Cin, int KH, int KW, int Cpad, int KWpad, void* stream) {
    return blt_conv_pack_w(dtype, w, out, Cout, Cin, KH, KW, Cpad, KWpad, (hipStream_t)stream);
}

// 7x7 stride-2 pad-3 stem convolution on a zero-bordered NHWC4 image [N, Hp, Wp, 4] (Hp >= H+6, Wp >= W+7 and even) with
// weights packed [Cout, 7, 8, 4]
int bltvqg_conv_stem(int dtype, const void* x_padded, const void* w, void* y, int N, int H, int W, int Hp, int Wp, int Cout, float* stat_sum,
                     float* stat_sq, void* stream) {
    BLT_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0, "conv_stem: bad sizes");
    if (blt_conv_stem_direct_ok(dtype, H, W, Hp, Wp, Cout))
        return blt_conv_stem_direct(x_padded, w, y, N, H, W, Hp, Wp, stat_sum, stat_sq, (hipStream_t)stream);
    GemmArgs g;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    g.A = x_padded; g.B = w; g.C = y;
    g.M = N * Ho * Wo; g.N = Cout; g.K = 224;
    g.lda = 4; g.ldb = 224; g.ldc = Cout; g.is_conv = 2;
    g.cg.Hi = Hp; g.cg.Wi = Wp; g.cg.Cin = 4; g.cg.cin_log2 = 2; g.cg.Ho = Ho; g.cg.Wo = Wo; g.cg.KH = 7; g.cg.KW = 8; g.cg.stride = 2; g.cg.pad = 0;
    g.stat_sum = stat_sum; g.stat_sq = stat_sq;
    return blt_gemm(dtype, g, (hipStream_t)stream);
}
// stem + 3x3/2 max-pool in one launch (bf16, Cout 64, Ho % 8 == 0, Wo % 14 == 0): pooling-window extrema of the raw convolution output
// (max where gamma >= 0, min where gamma < 0) into the pooled PP tensor, BatchNorm partial sums of every output
int bltvqg_conv_stem_pool(const void* x_padded, const void* w, const float* gamma, void* y_pool_pp, int N, int H, int W, int Hp, int Wp,
                          float* stat_sum, float* stat_sq, void* stream) {
    return blt_conv_stem_pool(x_padded, w, gamma, y_pool_pp, N, H, W, Hp, Wp, stat_sum, stat_sq, (hipStream_t)stream);
}
int bltvqg_conv_stem_pool_ok(int dtype, int H, int W, int Hp, int Wp, int Cout) { return blt_conv_stem_pool_ok(dtype, H, W, Hp, Wp, Cout) ? 1 : 0; }
int bltvqg_conv_stem_pool_stat_rows(int N, int H, int W) { return blt_conv_stem_pool_stat_rows(N, H, W); }
int bltvqg_conv_stem_stat_rows(int N, int H, int W, int Cout) {
    // an upper bound valid for both kernels (the bf16 LDS-patch kernel writes 2 rows per 8x16 tile, the implicit GEMM 2 per M tile);
    // rows that are not written must be zero (they are summed)
    GemmArgs g;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    g.M = N * Ho * Wo; g.N = Cout; g.K = 224;
    int rows = blt_gemm_stat_rows(g, BLT_BF16);
    const int rows32 = blt_gemm_stat_rows(g, BLT_F32);
    if (rows32 > rows) rows = rows32;
    if (Cout == 64 && Ho % 8 == 0 && Wo % 16 == 0 && blt_conv_stem_direct_stat_rows(N, H, W) > rows) rows = blt_conv_stem_direct_stat_rows(N, H, W);
    return rows;
}

int bltvqg_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                         int cols, float eps, void* stream) {
    return blt_layernorm_fwd(dtype, x, gamma, beta, y, mean, rstd, (long)rows, cols, eps, (hipStream_t)stream);
}
int bltvqg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                         void* dx, float* dgamma, float* dbeta, int64_t rows, int cols, void* stream) {
    return blt_layernorm_bwd(dtype, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, (long)rows, cols, (hipStream_t)stream);
}

int bltvqg_bn_scratch_doubles(int C) { return blt_bn_scratch_doubles(C); }
int bltvqg_bn_finalize(const float* psum, const float* psq, int nparts, int C, int64_t count, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, float* scale, float* shift, double* scratch, void* stream) {
    return blt_bn_finalize(psum, psq, nparts, C, (long)count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, nullptr,
                           nullptr, scratch, (hipStream_t)stream);
}
int bltvqg_bn_apply(int dtype, const void* x, const float* scale, const float* shift, const void* res, void* y, int64_t rows, int C, int relu,
                    void* stream) {
    return blt_bn_apply(dtype, x, scale, shift, res, y, (long)rows, C, relu, (hipStream_t)stream);
}
int bltvqg_bn_relu_maxpool(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi, int C, void* stream) {
    return blt_bn_relu_maxpool(dtype, x, scale, shift, y, N, Hi, Wi, C, (hipStream_t)stream);
}
int bltvqg_avgpool(int dtype, const void* x, void* y, int N, int HW, int C, void* stream) {
    return blt_avgpool(dtype, x, y, N, HW, C, 0, (hipStream_t)stream);
}
int bltvqg_bn1d_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, float* running_mean,
                    float* running_var, int B, int C, float eps, float momentum, void* stream) {
    return blt_bn1d_fwd(dtype, x, gamma, beta, y, mean, rstd, running_mean, running_var, B, C, eps, momentum, (hipStream_t)stream);
}
int bltvqg_bn1d_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                    float* dbeta, int B, int C, void* stream) {
    return blt_bn1d_bwd(dtype, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, B, C, (hipStream_t)stream);
}

int bltvqg_attn_fwd(int dtype, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, const int32_t* key_ids,
                    int B, int heads, int Tq, int Tk, int d, int causal, float scale, float drop_p, uint64_t seed, uint32_t stream_id, void* stream) {
    AttnArgs a;
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.key_ids = key_ids; a.B = B; a.heads = heads;
    a.Tq = Tq; a.Tk = Tk; a.d = d; a.causal = causal; a.scale = scale; a.drop_p = drop_p; a.seed = seed; a.stream_id = stream_id;
    return blt_attn_fwd(dtype, a, (hipStream_t)stream);
}
int bltvqg_attn_fwd_rows(int dtype, const void* Q, int ldq, int q_rows, const void* K, int ldk, const void* V, int ldv, int k_rows, void* O, int ldo,
                         const int32_t* key_ids, int B, int heads, int Tq, int Tk, int d, int causal, float scale, void* stream) {
    AttnArgs a;
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.key_ids = key_ids; a.B = B; a.heads = heads;
    a.Tq = Tq; a.Tk = Tk; a.d = d; a.causal = causal; a.scale = scale; a.q_rows = q_rows; a.k_rows = k_rows;
    return blt_attn_fwd(dtype, a, (hipStream_t)stream);
}
#ifdef BLT_EXPERIMENTS
int bltvqg_attn_out_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, const void* Wo, int ldwo,
                        const void* R, int ldr, void* Y, int ldy, const int32_t* key_ids, int B, int heads, int Tq, int Tk, int d, int causal,
                        float scale, float drop_p, uint64_t seed, uint32_t stream_id, void* stream) {
    AttnArgs a;
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.key_ids = key_ids; a.B = B; a.heads = heads;
    a.Tq = Tq; a.Tk = Tk; a.d = d; a.causal = causal; a.scale = scale; a.drop_p = drop_p; a.seed = seed; a.stream_id = stream_id;
    a.Wo = Wo; a.ldwo = ldwo; a.R = R; a.ldr = ldr; a.Y = Y; a.ldy = ldy;
    BLT_REQUIRE(Y && Wo && ldy >= heads * d && (!R || ldr >= heads * d) && ldwo >= heads * d, "attn_out_fwd: bad output / weight operands");
    return blt_attn_out_fwd(BLT_BF16, a, (hipStream_t)stream);
}
#endif
int bltvqg_attn_bwd(int dtype, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* dO, int lddo, void* dQ, int lddq,
                    void* dK, int lddk, void* dV, int lddv, const int32_t* key_ids, int B, int heads, int Tq, int Tk, int d, int causal, float scale,
                    float drop_p, uint64_t seed, uint32_t stream_id, void* stream) {
    AttnArgs a;
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.key_ids = key_ids; a.B = B; a.heads = heads;
    a.Tq = Tq; a.Tk = Tk; a.d = d; a.causal = causal; a.scale = scale; a.drop_p = drop_p; a.seed = seed; a.stream_id = stream_id;
    a.dO = dO; a.lddo = lddo; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
    return blt_attn_bwd(dtype, a, (hipStream_t)stream);
}

int bltvqg_embed_gather(int dtype, const float* table, const int32_t* ids, void* out, int64_t rows, int E, int ld, void* stream) {
    return blt_embed_gather(dtype, table, ids, out, (long)rows, E, ld, (hipStream_t)stream);
}
int bltvqg_embed_scatter(int dtype, const void* d, int ld, const int32_t* ids, float* dtable, int64_t rows, int E, int pad_id, void* stream) {
    return blt_embed_scatter(dtype, d, ld, ids, dtable, (long)rows, E, pad_id, (hipStream_t)stream);
}
int bltvqg_ce_fwd_bwd(int dtype, void* logits, int ld, const int32_t* target, int64_t M, int V, const float* count, float gscale, float* loss_out,
                      int write_grad, void* stream) {
    return blt_ce_fwd_bwd(dtype, logits, ld, target, (long)M, V, count, gscale, loss_out, write_grad, (hipStream_t)stream);
}
int bltvqg_bow_ce_fwd_bwd(int dtype, const void* zlogit, int ld, const int32_t* target, int B, int T, int V, const float* count, float gscale,
                          float* loss_out, void* dz, void* stream) {
    return blt_bow_ce_fwd_bwd(dtype, zlogit, ld, target, B, T, V, count, gscale, loss_out, dz, (hipStream_t)stream);
}
int bltvqg_mse_fwd_bwd(int dtype, const void* a, const void* b, int64_t n, float gscale, float* loss_out, void* da, void* db, void* stream) {
    return blt_mse_fwd_bwd(dtype, a, b, (long)n, gscale, loss_out, da, db, (hipStream_t)stream);
}
int bltvqg_latent_fwd(int dtype, const void* mlv_prior, const void* mlv_post, const float* eps, void* z, float* kld_out, int B, int Z, int ld,
                      void* stream) {
    return blt_latent_fwd(dtype, mlv_prior, mlv_post, eps, z, kld_out, B, Z, ld, (hipStream_t)stream);
}
int bltvqg_latent_bwd(int dtype, const void* mlv_prior, const void* mlv_post, const float* eps, const void* dz, float kld_gscale, void* dmlv_prior,
                      void* dmlv_post, int B, int Z, int ld, void* stream) {
    return blt_latent_bwd(dtype, mlv_prior, mlv_post, eps, dz, kld_gscale, dmlv_prior, dmlv_post, B, Z, ld, (hipStream_t)stream);
}
int bltvqg_sumsq(const float* x, int64_t n, float* out, void* stream) { return blt_sumsq(x, (long)n, out, (hipStream_t)stream); }
int bltvqg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm_sq, float max_norm, float lr, float beta1,
                     float beta2, float eps, int step, void* stream) {
    return blt_adam_step(p, g, m, v, (long)n, gnorm_sq, max_norm, lr, beta1, beta2, eps, step, (hipStream_t)stream);
}
int bltvqg_dropout_mask(uint64_t seed, uint32_t stream_id, int64_t rows, int cols, int ld_index, float p, uint8_t* out, void* stream) {
    return blt_dropout_mask(seed, stream_id, (long)rows, cols, ld_index, p, out, (hipStream_t)stream);
}
#ifdef BLT_EXPERIMENTS
int bltvqg_hw_id_probe(int32_t* out, int n_workgroups, int spin_ticks, void* stream) {
    return blt_hw_id_probe((int*)out, n_workgroups, spin_ticks, (hipStream_t)stream);
}
#endif
int bltvqg_cast(int dtype_src, const void* src, int ld_src, int dtype_dst, void* dst, int ld_dst, int64_t rows, int cols, void* stream) {
    return blt_cast_rows(dtype_src, src, ld_src, dtype_dst, dst, ld_dst, (long)rows, cols, (hipStream_t)stream);
}

int bltvqg_image_store_u8(const float* images, uint8_t* out, int64_t count, void* stream) {
    return blt_image_store_u8(images, out, (long)count, (hipStream_t)stream);
}
int bltvqg_batch_rows(const int32_t* questions, const int32_t* answers, const int32_t* answer_types, const int32_t* cat_word_ids, int n_cat,
                      int64_t n_rows, const int64_t* index, int B, int q_len, int a_len, int64_t* out_questions, int64_t* out_posteriors,
                      int64_t* out_answers, int64_t* out_answer_types, int64_t* out_types_for_input, void* stream) {
    BLT_REQUIRE(q_len >= 2 && a_len >= 2, "batch_rows: rows shorter than 2 tokens");
    return blt_batch_rows(questions, answers, answer_types, cat_word_ids, n_cat, (long)n_rows, (const long*)index, B, q_len, a_len,
                          (long*)out_questions, (long*)out_posteriors, (long*)out_answers, (long*)out_answer_types, (long*)out_types_for_input,
                          (hipStream_t)stream);
}
int bltvqg_batch_images(const uint8_t* table, int64_t n_images, int S, const int32_t* image_indices, int64_t n_rows, const int64_t* index,
                        const int32_t* boxes, const int32_t* coeffs, int KS, int B, int osz, const float* mean_std, float* out,
                        uint8_t* out_u8, void* stream) {
    return blt_batch_images(table, (long)n_images, S, image_indices, (long)n_rows, (const long*)index, boxes, coeffs, KS, B, osz, mean_std, out,
                            out_u8, (hipStream_t)stream);
}
int bltvqg_batch_images_packed(const uint8_t* table, int64_t n_images, int S, const int32_t* image_indices, int64_t n_rows,
                               const int64_t* index, const int32_t* boxes, const int32_t* coeffs, int KS, int B, int osz, const float* mean_std,
                               int dtype, void* out, int Hp, int Wp, int pad_top, int pad_left, void* stream) {
    return blt_batch_images_packed(table, (long)n_images, S, image_indices, (long)n_rows, (const long*)index, boxes, coeffs, KS, B, osz, mean_std,
                                   dtype, out, Hp, Wp, pad_top, pad_left, (hipStream_t)stream);
}

}  // extern "C"
