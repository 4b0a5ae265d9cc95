/* Experiment surface of libbltvqg_hip — NOT part of the product ABI.
 *
 * These entry points exist only in the experiments build (make -C blt-vqg_amd/csrc experiments -> libbltvqg_hip_exp.so, compiled with
 * -DBLT_EXPERIMENTS): timing aids, a hardware-id probe, and fused operators that were built, tested bit-identical, measured in the train
 * step and NOT adopted (DESIGN.md sections 5, 5c).  The shipped library (libbltvqg_hip.so, include/bltvqg_hip.h) exports none of them;
 * tests that cover them load the experiments build next to the product library and skip when it has not been built.
 */
#ifndef BLTVQG_HIP_EXPERIMENTS_H
#define BLTVQG_HIP_EXPERIMENTS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Timing aid: `reps` back-to-back launches of the Linear forward C = [relu](A B^T + bias) (+ R) from inside the library, so that a
 * host-side event pair measures device time per launch (a Python call per launch costs more than these kernels run). */
int bltvqg_gemm_repeat(int dtype, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias, int relu,
                       const void* R, int ldr, int reps, void* stream);

/* Timing aid like bltvqg_gemm_repeat, with COLD operands: launch i takes A / B / C from copy i % copies (copies are *_stride_bytes apart), so
 * that the operands of a launch were not touched by the launches just before it (weights of another layer); chain = 1 (N == K): launch i
 * reads as A what launch i-1 wrote as C — the dependent chain of the transformer stacks, operands freshly written by the previous kernel. */
int bltvqg_gemm_rotate(int dtype, const void* A, int lda, int a_copies, int64_t a_stride_bytes, const void* B, int ldb, int b_copies,
                       int64_t b_stride_bytes, void* C, int ldc, int c_copies, int64_t c_stride_bytes, int M, int N, int K, int chain, int reps,
                       void* stream);

/* bf16 Linear with the FOLLOWING LayerNorm in its epilogue (every LayerNorm of the transformer stacks reads the output of a Linear +
 * residual, transformer_layers.py:134,202,256-257,320-322): C = [dropout(relu(]X W^T + bias[))] (copied to C2 if non-null) + R, rounded
 * to bf16, then ln_out = LayerNorm(C) * ln_gamma + ln_beta with the row statistics in ln_mean / ln_rstd.  One workgroup owns whole rows:
 * N <= 256, N % 8 == 0; X [M,ldx], W [N,ldw] k-contiguous. */
int bltvqg_linear_layernorm(const void* X, int ldx, const void* W, int ldw, const float* bias, int relu, float drop_p, uint64_t seed,
                            uint32_t stream_id, void* C2, int ldc2, const void* R, int ldr, void* C, int ldc, const float* ln_gamma,
                            const float* ln_beta, float ln_eps, void* ln_out, float* ln_mean, float* ln_rstd, int M, int N, int K,
                            void* stream);

/* bf16 LayerNorm folded into the Linear that consumes it (every LayerNorm inside the transformer stacks feeds exactly one Linear:
 * q|k|v, the cross-attention query, the first FFN layer): Xn = LayerNorm(X) * ln_gamma + ln_beta (rounded to bf16, written [M,K] with the
 * row statistics, which backward needs) is formed on the A tile in LDS, C = [dropout(relu(]Xn W^T + bias[))] + R.  K <= 256, K % 8 == 0. */
int bltvqg_layernorm_linear(const void* X, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, void* Xn, float* ln_mean,
                            float* ln_rstd, const void* W, int ldw, const float* bias, int relu, float drop_p, uint64_t seed,
                            uint32_t stream_id, const void* R, int ldr, void* C, int ldc, int M, int N, int K, void* stream);

/* The attention core, its output Linear and the residual add of the surrounding sub-layer in ONE launch (transformer_layers.py:494-532: the
 * heads' context -> output_linear; + the `x +` of :139,208,267,274): O as bltvqg_attn_fwd writes it (backward's weight gradient reads it) AND
 * Y[B*Tq, heads*d] = O Wo^T + R (R may be NULL).  One workgroup per batch element, one wave per head, the head's 64 rows of Wo [heads*d, ldwo]
 * (bf16, k-contiguous) go straight from global memory into MFMA fragments.  bf16, d = 64, heads <= 8, Tq, Tk <= 32; bit-identical to
 * bltvqg_attn_fwd followed by bltvqg_gemm (same MFMA, same ascending k order).  The train-step engine does NOT use it: at B = 256 each of the
 * 256 workgroups streams the whole weight (512 KB) through its CU's ~70 GB/s intake, which costs more than the launch it saves. */
int bltvqg_attn_out_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, const void* Wo, int ldwo,
                        const void* R, int ldr, void* Y, int ldy, const int32_t* key_ids, int B, int heads, int Tq, int Tk, int d, int causal,
                        float scale, float drop_p, uint64_t seed, uint32_t stream_id, void* stream);

/* Two Linear problems in ONE launch of the planned-tile kernel (round 4).  The encoder stack and the posterior encoder stack of the reference
 * (models/iq.py:31-34: two Encoder instances of the same shape, run one after the other in IQ.forward, iq.py:66-78) execute the same Linear
 * positions on different rows with different weights; paired, a launch fills the chip with both problems' tiles instead of two launches
 * time-slicing it.  The two problems share N, K, the tile shape and the epilogue terms (relu, dropout p / seed, mask_scale, and WHICH
 * optional operands are present: bias, maskY, C2, R, fold_*, mean / rstd, out_stat must be NULL in both or in neither); operands, leading
 * dimensions, row counts and dropout streams are their own.  Results are bit-identical to the two separate launches at the same tile shape.
 * fold_s != NULL selects bltvqg_linear_ln_folded's form (row_stat / stat_parts / mean / rstd / eps), out_stat != NULL bltvqg_gemm_rowstat's.
 * The train-step engine does NOT use it: run in lockstep on one stream the two stacks lose the overlap two streams give them — the launch
 * gaps of one chain are the other chain's run time (forward: 6.79 against 6.79 ms / step, backward: 7.12 against 6.79, profiles/r04_pair_ab_*;
 * the engine side of the experiment is profiles/r04_pair_engine.patch). */
typedef struct bltvqg_linear_desc {
    const void* A; int32_t lda;          /* [M, lda] bf16 rows */
    const void* W; int32_t ldw;          /* [N, ldw] bf16 weight (the gamma-scaled shadow for the folded form) */
    void* C; int32_t ldc;                /* [M, ldc] bf16 result */
    int32_t M;
    const float* bias;
    const void* maskY; int32_t ldm;
    void* C2; int32_t ldc2;
    const void* R; int32_t ldr;
    uint32_t stream_id;
    const float* fold_s; const float* fold_c; const float* row_stat; float* mean; float* rstd;
    float* out_stat;
} bltvqg_linear_desc;
int bltvqg_linear_pair(const bltvqg_linear_desc* p1, const bltvqg_linear_desc* p2, int N, int K, int relu, float drop_p, uint64_t seed, float mask_scale,
                       int stat_slots, int stat_parts, float eps, int tile_m, int tile_n, void* stream);
/* Diagnostic: n_workgroups one-wave workgroups each write {HW_REG_HW_ID, XCC id} to out[2 * workgroup] after spinning spin_ticks of the
 * 100 MHz wall clock (so that they spread over every CU the stream may use): which CUs a (CU-masked) stream's work lands on. */
int bltvqg_hw_id_probe(int32_t* out, int n_workgroups, int spin_ticks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
