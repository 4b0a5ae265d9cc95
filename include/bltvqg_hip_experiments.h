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

/* Diagnostic: n_workgroups one-wave workgroups each write {HW_REG_HW_ID, XCC id} to out[2 * workgroup] after spinning spin_ticks of the
 * 100 MHz wall clock (so that they spread over every CU the stream may use): which CUs a (CU-masked) stream's work lands on. */
int bltvqg_hw_id_probe(int32_t* out, int n_workgroups, int spin_ticks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
