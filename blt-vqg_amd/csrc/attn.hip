// Fused small-sequence multi-head attention core (forward + backward) for sequences of <= 64 tokens.
// One workgroup (4 wavefronts) per (batch, head): Q/K/V head slices live in LDS as fp32, the whole Tq x Tk score tile stays on
// chip (never written to HBM), the pad / causal mask is derived from the key token ids, softmax statistics are
// per-lane rows, dropout masks come from the Philox counter RNG so that backward regenerates them exactly.
// The QKV / output projections are MFMA GEMMs (gemm.hip); this kernel is the part in between.
//
// Replaces models/transformer_layers.py:494-526 (split heads, scale, QK^T, masked_fill(-1e18), softmax, dropout,
// weights @ V, merge heads) and its autograd backward.  Sequences are 3..21 tokens in the reference
// (utils/data_loader.py:81,84,115), so a 32x32 MFMA tile would be > 55 % padding: the core runs on the VALU.
#include "kernels.h"

namespace {

// four waves per (batch, head): the tile is tiny, so the kernel is bound by LDS / global latency, not throughput — more waves
// per workgroup (sharing one LDS image) hide it
constexpr int ATT_THREADS = 256;
// LDS row stride of a head slice (floats): d + 4 keeps rows 16-byte aligned, so the dot products and the P.V / dS.K products read
// float4s (4x fewer LDS instructions than the d + 1 layout, which these latency-bound kernels feel directly); a stride of d + 4
// floats still walks the banks from row to row
__host__ __device__ __forceinline__ int att_dp(int d) { return ((d + 3) & ~3) + 4; }
__device__ __forceinline__ float dot_rows(const float* a, const float* b, int d) {
    float acc = 0.f;
    if ((d & 3) == 0) {
        for (int c = 0; c < d; c += 4) {
            const float4 x = *reinterpret_cast<const float4*>(a + c), y = *reinterpret_cast<const float4*>(b + c);
            acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
    } else {
        for (int c = 0; c < d; ++c) acc += a[c] * b[c];
    }
    return acc;
}

// head slice [Tn, d] of a packed projection -> fp32 LDS rows of d+1 floats; 16-byte global loads when the layout allows
template <typename T>
__device__ __forceinline__ void load_head(const T* __restrict__ src, int ld, int b, int Tn, int h, int d, float* dst, int lane) {
    const int dp = att_dp(d);
    if ((d & 7) == 0 && (ld & 7) == 0 && (((uintptr_t)src) & 15) == 0) {
        const int d8 = d >> 3;
        for (int idx = lane; idx < Tn * d8; idx += ATT_THREADS) {
            const int i = idx / d8, c = (idx - i * d8) * 8;
            float v[8];
            Vec8<T>::load(src + (size_t)(b * Tn + i) * ld + h * d + c, v);
            Vec8<float>::store(dst + i * dp + c, v);          // rows are 16-byte aligned (dp % 4 == 0)
        }
        return;
    }
    for (int idx = lane; idx < Tn * d; idx += ATT_THREADS) {
        const int i = idx / d, c = idx - i * d;
        dst[i * dp + c] = to_f32(src[(size_t)(b * Tn + i) * ld + h * d + c]);
    }
}

// out[i, c] = sum_j W[i or j][...] * X[j][c] written as 8 consecutive columns per lane (16-byte stores when possible).
// TRANS = false: out[i,c] = sum_j Wt[i*tp + j] * X[j*dp + c]   (rows of W);  TRANS = true: out[j,c] = sum_i Wt[i*tp + j] * X[i*dp + c]
template <typename T, bool TRANS>
__device__ __forceinline__ void matmul_store(const float* Wt, int tp, int nI, int nJ, const float* X, int dp, int d, float scale,
                                             T* __restrict__ out, int ld, int b, int h, int lane) {
    const int nOut = TRANS ? nJ : nI, nRed = TRANS ? nI : nJ;
    const bool vec = (d & 7) == 0 && (ld & 7) == 0 && (((uintptr_t)out) & 15) == 0;
    if (vec) {
        const int d8 = d >> 3;
        for (int idx = lane; idx < nOut * d8; idx += ATT_THREADS) {
            const int o = idx / d8, c = (idx - o * d8) * 8;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int r = 0; r < nRed; ++r) {
                const float w = TRANS ? Wt[r * tp + o] : Wt[o * tp + r];
                const float4 x0 = *reinterpret_cast<const float4*>(X + r * dp + c), x1 = *reinterpret_cast<const float4*>(X + r * dp + c + 4);
                acc[0] += w * x0.x; acc[1] += w * x0.y; acc[2] += w * x0.z; acc[3] += w * x0.w;
                acc[4] += w * x1.x; acc[5] += w * x1.y; acc[6] += w * x1.z; acc[7] += w * x1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] *= scale;
            Vec8<T>::store(out + (size_t)(b * nOut + o) * ld + h * d + c, acc);
        }
        return;
    }
    for (int idx = lane; idx < nOut * d; idx += ATT_THREADS) {
        const int o = idx / d, c = idx - o * d;
        float acc = 0.f;
        for (int r = 0; r < nRed; ++r) acc += (TRANS ? Wt[r * tp + o] : Wt[o * tp + r]) * X[r * dp + c];
        out[(size_t)(b * nOut + o) * ld + h * d + c] = from_f32<T>(acc * scale);
    }
}

// scores -> normalised probabilities Pn (in LDS), returns nothing.  Masked logits are REPLACED by -1e18
// (transformer_layers.py:504-506), so a fully masked row becomes uniform, exactly like the reference.
__device__ __forceinline__ void scores_softmax(const AttnArgs& a, int b, const float* Qs, const float* Ks, float* Pn,
                                               int lane) {
    const int dp = att_dp(a.d), tp = a.Tk + 1;
    for (int idx = lane; idx < a.Tq * a.Tk; idx += ATT_THREADS) {
        const int i = idx / a.Tk, j = idx - i * a.Tk;
        const float acc = dot_rows(Qs + i * dp, Ks + j * dp, a.d);
        const bool masked = (a.key_ids != nullptr && a.key_ids[b * a.Tk + j] == 0) || (a.causal == 1 && j > i);
        // causal == 2: a future key does not exist for this query (greedy decoding re-runs the decoder on the PREFIX, iq.py:134-141),
        // so it is excluded from the softmax instead of being filled with -1e18 (matters only for fully pad-masked rows)
        Pn[i * tp + j] = (a.causal == 2 && j > i) ? -INFINITY : (masked ? -1e18f : acc * a.scale);
    }
    __syncthreads();
    // softmax: 8 lanes per row (rows of <= 64 keys: up to 8 keys per lane), 32 rows per pass
    for (int row0 = 0; row0 < a.Tq; row0 += ATT_THREADS / 8) {
        const int row = row0 + (lane >> 3), sub = lane & 7;
        const bool rok = row < a.Tq;
        float v[8];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = sub + 8 * k;
            v[k] = (rok && j < a.Tk) ? Pn[row * tp + j] : -INFINITY;
            m = fmaxf(m, v[k]);
        }
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = (rok && sub + 8 * k < a.Tk) ? __expf(v[k] - m) : 0.f;
            s += v[k];
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float inv = 1.f / s;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = sub + 8 * k;
            if (rok && j < a.Tk) Pn[row * tp + j] = v[k] * inv;
        }
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int dp = att_dp(a.d), tp = a.Tk + 1;
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
        for (int idx = lane; idx < a.Tq * a.Tk; idx += ATT_THREADS) {
            const int i = idx / a.Tk, j = idx - i * a.Tk;
            const uint64_t e = ((uint64_t)blockIdx.x * a.Tq + i) * a.Tk + j;
            Pn[i * tp + j] = dropout_keep(a.seed, a.stream_id, e, thresh) ? Pn[i * tp + j] * ks : 0.f;
        }
        __syncthreads();
    }
    matmul_store<T, false>(Pn, tp, a.Tq, a.Tk, Vs, dp, a.d, 1.f, (T*)a.O, a.ldo, b, h, lane);
}

template <typename T>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int dp = att_dp(a.d), tp = a.Tk + 1;
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
    for (int idx = lane; idx < a.Tq * a.Tk; idx += ATT_THREADS) {
        const int i = idx / a.Tk, j = idx - i * a.Tk;
        const float g = dot_rows(dOs + i * dp, Vs + j * dp, a.d);
        bool keep = true;
        if (a.drop_p > 0.f) keep = dropout_keep(a.seed, a.stream_id, ((uint64_t)blockIdx.x * a.Tq + i) * a.Tk + j, thresh);
        Pd[i * tp + j] = keep ? Pn[i * tp + j] * ks : 0.f;
        dS[i * tp + j] = keep ? g * ks : 0.f;      // d(Pn)
    }
    __syncthreads();
    // softmax backward, 8 lanes per row like the forward: delta = sum_j dP * P, dS = P * (dP - delta) * scale
    for (int row0 = 0; row0 < a.Tq; row0 += ATT_THREADS / 8) {
        const int row = row0 + (lane >> 3), sub = lane & 7;
        const bool rok = row < a.Tq;
        float pv[8], gv[8];
        float delta = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = sub + 8 * k;
            const bool in = rok && j < a.Tk;
            pv[k] = in ? Pn[row * tp + j] : 0.f;
            gv[k] = in ? dS[row * tp + j] : 0.f;
            delta += pv[k] * gv[k];
        }
        delta += __shfl_xor(delta, 1, 64); delta += __shfl_xor(delta, 2, 64); delta += __shfl_xor(delta, 4, 64);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = sub + 8 * k;
            if (rok && j < a.Tk) {
                // masked_fill REPLACES the logit, so no gradient reaches a masked position — this matters for a fully
                // masked row, whose probabilities are uniform (non-zero) rather than 0
                const bool masked = (a.key_ids != nullptr && a.key_ids[b * a.Tk + j] == 0) || (a.causal != 0 && j > row);
                dS[row * tp + j] = masked ? 0.f : pv[k] * (gv[k] - delta) * a.scale;
            }
        }
    }
    __syncthreads();
    matmul_store<T, false>(dS, tp, a.Tq, a.Tk, Ks, dp, a.d, 1.f, (T*)a.dQ, a.lddq, b, h, lane);      // dQ = dS K
    matmul_store<T, true>(dS, tp, a.Tq, a.Tk, Qs, dp, a.d, 1.f, (T*)a.dK, a.lddk, b, h, lane);       // dK = dS^T Q
    matmul_store<T, true>(Pd, tp, a.Tq, a.Tk, dOs, dp, a.d, 1.f, (T*)a.dV, a.lddv, b, h, lane);      // dV = Pd^T dO
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
    const size_t dp = att_dp(a.d), tp = a.Tk + 1;
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
        hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(a.B * a.heads), dim3(ATT_THREADS), lds, s, a);
    } else {
        if ((rc = set_lds(attn_fwd_kernel<bf16>, lds, "attn_fwd"))) return rc;
        hipLaunchKernelGGL(attn_fwd_kernel<bf16>, dim3(a.B * a.heads), dim3(ATT_THREADS), lds, s, a);
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
        hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(a.B * a.heads), dim3(ATT_THREADS), lds, s, a);
    } else {
        if ((rc = set_lds(attn_bwd_kernel<bf16>, lds, "attn_bwd"))) return rc;
        hipLaunchKernelGGL(attn_bwd_kernel<bf16>, dim3(a.B * a.heads), dim3(ATT_THREADS), lds, s, a);
    }
    return blt_check_launch("attn_bwd");
}
