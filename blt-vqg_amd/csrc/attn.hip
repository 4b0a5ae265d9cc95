// Fused small-sequence multi-head attention core (forward + backward) for sequences of <= 64 tokens.
// One wavefront per (batch, head): Q/K/V head slices live in LDS as fp32, the whole Tq x Tk score tile stays on
// chip (never written to HBM), the pad / causal mask is derived from the key token ids, softmax statistics are
// per-lane rows, dropout masks come from the Philox counter RNG so that backward regenerates them exactly.
// The QKV / output projections are MFMA GEMMs (gemm.hip); this kernel is the part in between.
//
// Replaces models/transformer_layers.py:494-526 (split heads, scale, QK^T, masked_fill(-1e18), softmax, dropout,
// weights @ V, merge heads) and its autograd backward.  Sequences are 3..21 tokens in the reference
// (utils/data_loader.py:81,84,115), so a 32x32 MFMA tile would be > 55 % padding: the core runs on the VALU.
#include "kernels.h"

namespace {

template <typename T>
__device__ __forceinline__ void load_head(const T* __restrict__ src, int ld, int b, int Tn, int h, int d, float* dst,
                                          int lane) {
    const int dp = d + 1;
    for (int idx = lane; idx < Tn * d; idx += 64) {
        const int i = idx / d, c = idx - i * d;
        dst[i * dp + c] = to_f32(src[(size_t)(b * Tn + i) * ld + h * d + c]);
    }
}

// scores -> normalised probabilities Pn (in LDS), returns nothing.  Masked logits are REPLACED by -1e18
// (transformer_layers.py:504-506), so a fully masked row becomes uniform, exactly like the reference.
__device__ __forceinline__ void scores_softmax(const AttnArgs& a, int b, const float* Qs, const float* Ks, float* Pn,
                                               int lane) {
    const int dp = a.d + 1, tp = a.Tk + 1;
    for (int idx = lane; idx < a.Tq * a.Tk; idx += 64) {
        const int i = idx / a.Tk, j = idx - i * a.Tk;
        float acc = 0.f;
        for (int c = 0; c < a.d; ++c) acc += Qs[i * dp + c] * Ks[j * dp + c];
        const bool masked = (a.key_ids != nullptr && a.key_ids[b * a.Tk + j] == 0) || (a.causal && j > i);
        Pn[i * tp + j] = masked ? -1e18f : acc * a.scale;
    }
    __syncthreads();
    if (lane < a.Tq) {
        float m = -INFINITY;
        for (int j = 0; j < a.Tk; ++j) m = fmaxf(m, Pn[lane * tp + j]);
        float s = 0.f;
        for (int j = 0; j < a.Tk; ++j) {
            const float e = __expf(Pn[lane * tp + j] - m);
            Pn[lane * tp + j] = e;
            s += e;
        }
        const float inv = 1.f / s;
        for (int j = 0; j < a.Tk; ++j) Pn[lane * tp + j] *= inv;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int dp = a.d + 1, tp = a.Tk + 1;
    float* Qs = sm;
    float* Ks = Qs + a.Tq * dp;
    float* Vs = Ks + a.Tk * dp;
    float* Pn = Vs + a.Tk * dp;
    load_head((const T*)a.Q, a.ldq, b, a.Tq, h, a.d, Qs, lane);
    load_head((const T*)a.K, a.ldk, b, a.Tk, h, a.d, Ks, lane);
    load_head((const T*)a.V, a.ldv, b, a.Tk, h, a.d, Vs, lane);
    __syncthreads();
    scores_softmax(a, b, Qs, Ks, Pn, lane);
    if (a.drop_p > 0.f) {
        const uint32_t thresh = dropout_threshold(a.drop_p);
        const float ks = 1.f / (1.f - a.drop_p);
        for (int idx = lane; idx < a.Tq * a.Tk; idx += 64) {
            const int i = idx / a.Tk, j = idx - i * a.Tk;
            const uint64_t e = ((uint64_t)blockIdx.x * a.Tq + i) * a.Tk + j;
            Pn[i * tp + j] = dropout_keep(a.seed, a.stream_id, e, thresh) ? Pn[i * tp + j] * ks : 0.f;
        }
        __syncthreads();
    }
    T* O = (T*)a.O;
    for (int idx = lane; idx < a.Tq * a.d; idx += 64) {
        const int i = idx / a.d, c = idx - i * a.d;
        float acc = 0.f;
        for (int j = 0; j < a.Tk; ++j) acc += Pn[i * tp + j] * Vs[j * dp + c];
        O[(size_t)(b * a.Tq + i) * a.ldo + h * a.d + c] = from_f32<T>(acc);
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int dp = a.d + 1, tp = a.Tk + 1;
    float* Qs = sm;
    float* Ks = Qs + a.Tq * dp;
    float* Vs = Ks + a.Tk * dp;
    float* dOs = Vs + a.Tk * dp;
    float* Pn = dOs + a.Tq * dp;      // normalised probabilities (pre-dropout)
    float* Pd = Pn + a.Tq * tp;       // dropped/rescaled probabilities (what multiplied V in forward)
    float* dS = Pd + a.Tq * tp;       // d(logits) * scale
    load_head((const T*)a.Q, a.ldq, b, a.Tq, h, a.d, Qs, lane);
    load_head((const T*)a.K, a.ldk, b, a.Tk, h, a.d, Ks, lane);
    load_head((const T*)a.V, a.ldv, b, a.Tk, h, a.d, Vs, lane);
    load_head((const T*)a.dO, a.lddo, b, a.Tq, h, a.d, dOs, lane);
    __syncthreads();
    scores_softmax(a, b, Qs, Ks, Pn, lane);
    const uint32_t thresh = dropout_threshold(a.drop_p);
    const float ks = (a.drop_p > 0.f) ? 1.f / (1.f - a.drop_p) : 1.f;
    for (int idx = lane; idx < a.Tq * a.Tk; idx += 64) {
        const int i = idx / a.Tk, j = idx - i * a.Tk;
        float g = 0.f;
        for (int c = 0; c < a.d; ++c) g += dOs[i * dp + c] * Vs[j * dp + c];
        bool keep = true;
        if (a.drop_p > 0.f) keep = dropout_keep(a.seed, a.stream_id, ((uint64_t)blockIdx.x * a.Tq + i) * a.Tk + j, thresh);
        Pd[i * tp + j] = keep ? Pn[i * tp + j] * ks : 0.f;
        dS[i * tp + j] = keep ? g * ks : 0.f;      // d(Pn)
    }
    __syncthreads();
    if (lane < a.Tq) {
        float delta = 0.f;
        for (int j = 0; j < a.Tk; ++j) delta += dS[lane * tp + j] * Pn[lane * tp + j];
        for (int j = 0; j < a.Tk; ++j) {
            // masked_fill REPLACES the logit, so no gradient reaches a masked position — this matters for a fully
            // masked row, whose probabilities are uniform (non-zero) rather than 0
            const bool masked = (a.key_ids != nullptr && a.key_ids[b * a.Tk + j] == 0) || (a.causal && j > lane);
            dS[lane * tp + j] = masked ? 0.f : Pn[lane * tp + j] * (dS[lane * tp + j] - delta) * a.scale;
        }
    }
    __syncthreads();
    T* dQ = (T*)a.dQ;
    T* dK = (T*)a.dK;
    T* dV = (T*)a.dV;
    for (int idx = lane; idx < a.Tq * a.d; idx += 64) {
        const int i = idx / a.d, c = idx - i * a.d;
        float acc = 0.f;
        for (int j = 0; j < a.Tk; ++j) acc += dS[i * tp + j] * Ks[j * dp + c];
        dQ[(size_t)(b * a.Tq + i) * a.lddq + h * a.d + c] = from_f32<T>(acc);
    }
    for (int idx = lane; idx < a.Tk * a.d; idx += 64) {
        const int j = idx / a.d, c = idx - j * a.d;
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < a.Tq; ++i) {
            ak += dS[i * tp + j] * Qs[i * dp + c];
            av += Pd[i * tp + j] * dOs[i * dp + c];
        }
        dK[(size_t)(b * a.Tk + j) * a.lddk + h * a.d + c] = from_f32<T>(ak);
        dV[(size_t)(b * a.Tk + j) * a.lddv + h * a.d + c] = from_f32<T>(av);
    }
}

int check(const AttnArgs& a, bool bwd) {
    BLT_REQUIRE(a.Q && a.K && a.V, "attn: null Q/K/V");
    BLT_REQUIRE(a.B > 0 && a.heads > 0 && a.d > 0, "attn: bad sizes");
    BLT_REQUIRE(a.Tq > 0 && a.Tq <= 64 && a.Tk > 0 && a.Tk <= 64, "attn: Tq=%d Tk=%d must be in 1..64", a.Tq, a.Tk);
    BLT_REQUIRE(a.drop_p >= 0.f && a.drop_p < 1.f, "attn: bad dropout p");
    if (!bwd) BLT_REQUIRE(a.O != nullptr, "attn_fwd: null O");
    else BLT_REQUIRE(a.dO && a.dQ && a.dK && a.dV, "attn_bwd: null gradient pointer");
    return BLT_OK;
}

size_t lds_bytes(const AttnArgs& a, bool bwd) {
    const size_t dp = a.d + 1, tp = a.Tk + 1;
    size_t f = (size_t)a.Tq * dp + 2 * (size_t)a.Tk * dp + (size_t)a.Tq * tp;
    if (bwd) f += (size_t)a.Tq * dp + 2 * (size_t)a.Tq * tp;
    return f * sizeof(float);
}

template <typename K>
int set_lds(K kern, size_t bytes, const char* what) {
    if (bytes > 160 * 1024) {
        blt_set_error("%s: needs %zu bytes of LDS (> 160 KiB)", what, bytes);
        return BLT_ERR_ARG;
    }
    if (bytes > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            blt_set_error("%s: hipFuncSetAttribute failed", what);
            return BLT_ERR_HIP;
        }
    }
    return BLT_OK;
}

}  // namespace

int blt_attn_fwd(int dtype, const AttnArgs& a, hipStream_t s) {
    BLT_REQUIRE(dtype == BLT_F32 || dtype == BLT_BF16, "attn_fwd: bad dtype");
    int rc = check(a, false);
    if (rc) return rc;
    const size_t lds = lds_bytes(a, false);
    if (dtype == BLT_F32) {
        if ((rc = set_lds(attn_fwd_kernel<float>, lds, "attn_fwd"))) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(a.B * a.heads), dim3(64), lds, s, a);
    } else {
        if ((rc = set_lds(attn_fwd_kernel<bf16>, lds, "attn_fwd"))) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<bf16>, dim3(a.B * a.heads), dim3(64), lds, s, a);
    }
    return blt_check_launch("attn_fwd");
}

int blt_attn_bwd(int dtype, const AttnArgs& a, hipStream_t s) {
    BLT_REQUIRE(dtype == BLT_F32 || dtype == BLT_BF16, "attn_bwd: bad dtype");
    int rc = check(a, true);
    if (rc) return rc;
    const size_t lds = lds_bytes(a, true);
    if (dtype == BLT_F32) {
        if ((rc = set_lds(attn_bwd_kernel<float>, lds, "attn_bwd"))) return rc;
        hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(a.B * a.heads), dim3(64), lds, s, a);
    } else {
        if ((rc = set_lds(attn_bwd_kernel<bf16>, lds, "attn_bwd"))) return rc;
        hipLaunchKernelGGL(attn_bwd_kernel<bf16>, dim3(a.B * a.heads), dim3(64), lds, s, a);
    }
    return blt_check_launch("attn_bwd");
}
