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
__device__ __forceinline__ void load_head(const T* __restrict__ src, int ld, int b, int Tn, int h, int d, float* dst, int lane, int rows = 0) {
    const int dp = att_dp(d);
    if (rows == 0) rows = Tn;      // rows per batch element of the tensor (>= Tn: only the first Tn are loaded)
    if ((d & 7) == 0 && (ld & 7) == 0 && (((uintptr_t)src) & 15) == 0) {
        const int d8 = d >> 3;
        for (int idx = lane; idx < Tn * d8; idx += ATT_THREADS) {
            const int i = idx / d8, c = (idx - i * d8) * 8;
            float v[8];
            Vec8<T>::load(src + (size_t)(b * rows + i) * ld + h * d + c, v);
            Vec8<float>::store(dst + i * dp + c, v);          // rows are 16-byte aligned (dp % 4 == 0)
        }
        return;
    }
    for (int idx = lane; idx < Tn * d; idx += ATT_THREADS) {
        const int i = idx / d, c = idx - i * d;
        dst[i * dp + c] = to_f32(src[(size_t)(b * rows + i) * ld + h * d + c]);
    }
}

// out[i, c] = sum_j W[i or j][...] * X[j][c] written as 8 consecutive columns per lane (16-byte stores when possible).
// TRANS = false: out[i,c] = sum_j Wt[i*tp + j] * X[j*dp + c]   (rows of W);  TRANS = true: out[j,c] = sum_i Wt[i*tp + j] * X[i*dp + c]
template <typename T, bool TRANS>
__device__ __forceinline__ void matmul_store(const float* Wt, int tp, int nI, int nJ, const float* X, int dp, int d, float scale,
                                             T* __restrict__ out, int ld, int b, int h, int lane, int orows = 0) {
    const int nOut = TRANS ? nJ : nI, nRed = TRANS ? nI : nJ;
    if (orows == 0) orows = nOut;      // rows per batch element of `out`
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
            Vec8<T>::store(out + (size_t)(b * orows + o) * ld + h * d + c, acc);
        }
        return;
    }
    for (int idx = lane; idx < nOut * d; idx += ATT_THREADS) {
        const int o = idx / d, c = idx - o * d;
        float acc = 0.f;
        for (int r = 0; r < nRed; ++r) acc += (TRANS ? Wt[r * tp + o] : Wt[o * tp + r]) * X[r * dp + c];
        out[(size_t)(b * orows + o) * ld + h * d + c] = from_f32<T>(acc * scale);
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
        const bool masked = (a.key_ids != nullptr && a.key_ids[b * (a.k_rows ? a.k_rows : a.Tk) + j] == 0) || (a.causal == 1 && j > i);
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
    load_head((const T*)a.Q, a.ldq, b, a.Tq, h, a.d, Qs, lane, a.q_rows);
    load_head((const T*)a.K, a.ldk, b, a.Tk, h, a.d, Ks, lane, a.k_rows);
    load_head((const T*)a.V, a.ldv, b, a.Tk, h, a.d, Vs, lane, a.k_rows);
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
    matmul_store<T, false>(Pn, tp, a.Tq, a.Tk, Vs, dp, a.d, 1.f, (T*)a.O, a.ldo, b, h, lane, a.q_rows);
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

// ---------------------------------------------------------------------------------------------------------------
// MFMA form (bf16, head dim 32/64/128, <= 32 queries and keys): ONE WAVE per (batch, head), no workgroup barrier anywhere.
//
// The VALU kernels above keep the head slices in LDS as fp32 and every product re-reads them: 2048 (batch, head) pairs x 5 products
// x ~200 KB of LDS reads is ~27 us of LDS time per CU for the big configuration's backward — what they measured (32 us).  Here the
// five products are v_mfma_f32_16x16x32_bf16 tiles on operands that sit in registers:
//   * S^T = K.Q^T (not Q.K^T): the result has the QUERY on the lane (n = lane & 15) and 4 consecutive KEYS in the registers of lane
//     group g, so (a) the softmax reduction over keys is 8 in-lane values + two cross-group shuffles, (b) the 4 keys share one Philox
//     block of the dropout stream, and (c) P^T / dS^T are already the B operand of the products that sum over keys (O^T = V^T.P^T,
//     dQ^T = K^T.dS^T) — with the key order inside the 32-wide k-step permuted (slot e of group g = key 4g+e for e < 4, 16+4g+e-4
//     otherwise), which the other operand follows.
//   * operands with the token on the k axis (V^T, K^T, Q^T, dO^T) come from a wave-private row-major LDS image of the head slice
//     through ds_read_b64_tr_b16 (the transposing read), products that sum over QUERIES (dK, dV) take P / dS from a 32x32 LDS image
//     the same way.  One wave writes and reads its own images: LDS operations of one wave complete in order, no barrier.
// Rows / keys past Tq / Tk are zero operands; padded queries produce P = dS = 0 so that the query-summed products stay exact.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ s16x4 tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (bf16)lo[e]; v[4 + e] = (bf16)hi[e]; }
    return v;
}
// keep decisions of the 4 consecutive dropout elements e0 .. e0+3 (bit r = element e0 + r), same stream as dropout_keep()
__device__ __forceinline__ uint32_t keep4(uint64_t seed, uint32_t sid, uint64_t e0, uint32_t thresh) {
    uint32_t w[4];
    dropout_words(seed, sid, e0 >> 3, w);
    uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32), hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    const int o = (int)(e0 & 7);
    uint64_t lo2 = 0;
    if (o > 4) {      // the 4 elements straddle two Philox blocks (only when Tk is not a multiple of 4)
        dropout_words(seed, sid, (e0 >> 3) + 1, w);
        lo2 = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    }
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = o + r;
        const uint64_t x = k < 4 ? lo : (k < 8 ? hi : lo2);
        const uint32_t v = (uint32_t)(x >> ((k & 3) * 16)) & 0xFFFFu;
        bits |= (v >= thresh ? 1u : 0u) << r;
    }
    return bits;
}

struct MfmaHead {
    int b, h, hb, n16, g;
};

// S^T tile registers -> normalised probabilities (same layout).  st[jt][it][r]: key j = 16 jt + 4 g + r, query i = 16 it + n16.
__device__ __forceinline__ void softmax_t(const AttnArgs& a, const MfmaHead& m, f32x4 (&st)[2][2], uint32_t (&masked)[2]) {
    // masked[it] bit (4 jt + r): the logit was REPLACED (pad key / causal), no gradient flows through it
    int kid[2][4];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * m.g + r;
            kid[jt][r] = (a.key_ids != nullptr && j < a.Tk) ? a.key_ids[m.b * (a.k_rows ? a.k_rows : a.Tk) + j] : 1;
        }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = 16 * it + m.n16;
        float mx = -INFINITY;
        masked[it] = 0;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * m.g + r;
                const bool msk = kid[jt][r] == 0 || (a.causal != 0 && j > i);
                float v = st[jt][it][r] * a.scale;
                if (msk) v = -1e18f;
                if (j >= a.Tk || (a.causal == 2 && j > i)) v = -INFINITY;
                if (msk) masked[it] |= 1u << (4 * jt + r);
                st[jt][it][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = max_across_rows(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(st[jt][it][r] - mx);      // -inf (key past Tk) -> 0
                st[jt][it][r] = e;
                sum += e;
            }
        sum = sum_across_rows(sum);
        const float inv = (i < a.Tq) ? 1.f / sum : 0.f;       // padded queries: P = 0
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[jt][it][r] *= inv;
    }
}

template <int D>
__device__ __forceinline__ void load_slice(const bf16* __restrict__ X, int ld, int T, const MfmaHead& m, bf16x8 (&f)[2][D / 32], char* img) {
    constexpr int PITCH = D * 2 + 16;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 16 * t + m.n16;
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (row < T) v = *reinterpret_cast<const bf16x8*>(X + (size_t)row * ld + ks * 32 + m.g * 8);
            f[t][ks] = v;
            if (img != nullptr) *reinterpret_cast<bf16x8*>(img + row * PITCH + (ks * 32 + m.g * 8) * 2) = v;
        }
    }
}

// X^T operand of c-tile ct from a row-major [token][D] image: PERM = keys in the permuted slot order of P^T / dS^T, else tokens 8g..8g+7
template <int D, bool PERM>
__device__ __forceinline__ bf16x8 tr_operand(const char* img, int ct, const MfmaHead& m) {
    constexpr int PITCH = D * 2 + 16;
    const int q = m.n16 >> 2, p = m.n16 & 3;
    const int r0 = PERM ? 4 * m.g : 8 * m.g, r1 = PERM ? 16 + 4 * m.g : 8 * m.g + 4;
    const s16x4 lo = tr16(img + (r0 + q) * PITCH + (ct * 16 + 4 * p) * 2);
    const s16x4 hi = tr16(img + (r1 + q) * PITCH + (ct * 16 + 4 * p) * 2);
    return join8(lo, hi);
}

constexpr int ATT_PP = 72;      // row pitch (bytes) of the 32 x 32 bf16 P / dS images

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const AttnArgs a) {
    constexpr int KS = D / 32, CT = D / 16, PITCH = D * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smc[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    MfmaHead m;
    m.hb = blockIdx.x * 4 + w;
    if (m.hb >= a.B * a.heads) return;          // whole wave leaves: the transposing reads below need a full EXEC mask
    m.b = m.hb / a.heads; m.h = m.hb % a.heads; m.n16 = lane & 15; m.g = lane >> 4;
    char* vimg = smc + w * (32 * PITCH);
    bf16x8 kf[2][KS], qf[2][KS], vf[2][KS];
    const size_t qr = a.q_rows ? a.q_rows : a.Tq, kr = a.k_rows ? a.k_rows : a.Tk;      // rows per batch element of the tensors
    load_slice<D>((const bf16*)a.K + (size_t)m.b * kr * a.ldk + m.h * D, a.ldk, a.Tk, m, kf, nullptr);
    load_slice<D>((const bf16*)a.Q + (size_t)m.b * qr * a.ldq + m.h * D, a.ldq, a.Tq, m, qf, nullptr);
    load_slice<D>((const bf16*)a.V + (size_t)m.b * kr * a.ldv + m.h * D, a.ldv, a.Tk, m, vf, vimg);
    f32x4 st[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            st[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) st[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][ks], qf[it][ks], st[jt][it], 0, 0, 0);
        }
    uint32_t masked[2];
    softmax_t(a, m, st, masked);
    if (a.drop_p > 0.f) {
        const uint32_t thresh = dropout_threshold(a.drop_p);
        const float ks_ = 1.f / (1.f - a.drop_p);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = 16 * it + m.n16;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int j0 = 16 * jt + 4 * m.g;
                if (i < a.Tq && j0 < a.Tk) {
                    const uint32_t keep = keep4(a.seed, a.stream_id, ((uint64_t)m.hb * a.Tq + i) * a.Tk + j0, thresh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[jt][it][r] = ((keep >> r) & 1u) ? st[jt][it][r] * ks_ : 0.f;
                }
            }
        }
    }
    bf16x8 pb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) pb[it] = pack8(st[0][it], st[1][it]);
    bf16* O = (bf16*)a.O + (size_t)m.b * qr * a.ldo + m.h * D;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const bf16x8 vt = tr_operand<D, true>(vimg, ct, m);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pb[it], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const int i = 16 * it + m.n16;
            if (i < a.Tq) {
                s16x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(short, (bf16)o[e]);
                *reinterpret_cast<s16x4*>(O + (size_t)i * a.ldo + ct * 16 + 4 * m.g) = r;
            }
        }
    }
}

#ifdef BLT_EXPERIMENTS      // (built, tested bit-identical, measured +0.27 ms per step, not adopted: DESIGN.md 5c.9)
// ---------------------------------------------------------------------------------------------------------------
// Attention + output projection + residual in one launch.  One workgroup per BATCH ELEMENT, wave h = head h: the attention part is the
// one-wave kernel above; every wave then owns 64 output columns of Y = O Wo^T + R.  The A operand (the [Tq, H] context of this batch
// element, all heads) is exchanged through LDS; the B operand needs no sharing at all — wave w's 64 rows of Wo go straight from
// global memory into its MFMA fragments (16 bytes per lane = one fragment row), and the first quarter of them is requested BEFORE the
// attention arithmetic, so the weight round trip hides under it.  Same MFMA, same ascending k order as gemm_nt2_kernel: the result is
// bit-identical to the two-launch path (tests/test_ops_gpu.py).  Saves the output projection's launch (~10 us of skeleton) per attention.
// ---------------------------------------------------------------------------------------------------------------
template <int D, int MAXT>      // MAXT: thread budget (heads * 64 <= MAXT): 512 threads leave every wave 256 VGPRs (192 used); a 1024-thread form for 9..16 heads would spill
__global__ __launch_bounds__(MAXT) void attn_out_fwd_kernel(const AttnArgs a) {
    constexpr int KS = D / 32, CT = D / 16, PITCH = D * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smc[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;      // w = head = 64-column slice of the output
    const int H = a.heads * D;
    const int CP = H * 2 + 16;                                     // row pitch (bytes) of the [32, H] context tile
    MfmaHead m;
    m.b = blockIdx.x; m.h = w; m.hb = m.b * a.heads + w; m.n16 = lane & 15; m.g = lane >> 4;
    char* vimg = smc + w * (32 * PITCH);
    char* ctile = smc + a.heads * (32 * PITCH);
    // ---- weights of this wave's 64 output columns: k-steps 0..3 of 16 requested now ----
    const bf16* Wrow = (const bf16*)a.Wo + (size_t)(w * 64 + m.n16) * a.ldwo + m.g * 8;      // + jt * 16 rows, + ks * 32 columns
    constexpr int NKS = 4;                                          // k-steps (of 32) per batch of weight fragments
    bf16x8 wf[NKS][4];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) wf[ks][jt] = *reinterpret_cast<const bf16x8*>(Wrow + (size_t)jt * 16 * a.ldwo + ks * 32);
    // ---- attention of head w (as attn_fwd_mfma_kernel) ----
    bf16x8 kf[2][KS], qf[2][KS], vf[2][KS];
    load_slice<D>((const bf16*)a.K + (size_t)m.b * a.Tk * a.ldk + m.h * D, a.ldk, a.Tk, m, kf, nullptr);
    load_slice<D>((const bf16*)a.Q + (size_t)m.b * a.Tq * a.ldq + m.h * D, a.ldq, a.Tq, m, qf, nullptr);
    load_slice<D>((const bf16*)a.V + (size_t)m.b * a.Tk * a.ldv + m.h * D, a.ldv, a.Tk, m, vf, vimg);
    f32x4 st[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            st[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) st[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][ks], qf[it][ks], st[jt][it], 0, 0, 0);
        }
    uint32_t masked[2];
    softmax_t(a, m, st, masked);
    if (a.drop_p > 0.f) {
        const uint32_t thresh = dropout_threshold(a.drop_p);
        const float ks_ = 1.f / (1.f - a.drop_p);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = 16 * it + m.n16;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int j0 = 16 * jt + 4 * m.g;
                if (i < a.Tq && j0 < a.Tk) {
                    const uint32_t keep = keep4(a.seed, a.stream_id, ((uint64_t)m.hb * a.Tq + i) * a.Tk + j0, thresh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[jt][it][r] = ((keep >> r) & 1u) ? st[jt][it][r] * ks_ : 0.f;
                }
            }
        }
    }
    bf16x8 pb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) pb[it] = pack8(st[0][it], st[1][it]);
    bf16* O = (bf16*)a.O + (size_t)m.b * a.Tq * a.ldo + m.h * D;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const bf16x8 vt = tr_operand<D, true>(vimg, ct, m);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pb[it], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const int i = 16 * it + m.n16;
            s16x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(short, (bf16)((i < a.Tq) ? o[e] : 0.f));
            if (i < a.Tq) *reinterpret_cast<s16x4*>(O + (size_t)i * a.ldo + ct * 16 + 4 * m.g) = r;
            *reinterpret_cast<s16x4*>(ctile + i * CP + (m.h * D + ct * 16 + 4 * m.g) * 2) = r;      // rows >= Tq: zeros
        }
    }
    __syncthreads();      // every head's context of this batch element is in the tile
    // ---- Y[:, 64 w .. 64 w + 64) = ctx Wo^T + R ----
    f32x4 acc[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = H / 32;
    for (int k0 = 0; k0 < nks; k0 += NKS) {
        bf16x8 wn[NKS][4];
        const bool more = k0 + NKS < nks;
        if (more) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) wn[ks][jt] = *reinterpret_cast<const bf16x8*>(Wrow + (size_t)jt * 16 * a.ldwo + (k0 + NKS + ks) * 32);
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            bf16x8 af[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) af[it] = *reinterpret_cast<const bf16x8*>(ctile + (16 * it + m.n16) * CP + ((k0 + ks) * 32 + m.g * 8) * 2);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], wf[ks][jt], acc[it][jt], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) wf[ks][jt] = wn[ks][jt];
        }
    }
    // D[m = 4 g + r][n = n16] of tile (it, jt): row 16 it + 4 g + r, column 64 w + 16 jt + n16
    const bf16* R = (const bf16*)a.R;
    bf16* Y = (bf16*)a.Y;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * it + 4 * m.g + r;
            if (i >= a.Tq) continue;
            const size_t row = (size_t)m.b * a.Tq + i;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int n = w * 64 + 16 * jt + m.n16;
                float v = acc[it][jt][r];
                if (R != nullptr) v += (float)R[row * a.ldr + n];
                Y[row * a.ldy + n] = (bf16)v;
            }
        }
}
#endif      // BLT_EXPERIMENTS

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const AttnArgs a) {
    constexpr int KS = D / 32, CT = D / 16, PITCH = D * 2 + 16;
    constexpr int WAVE_LDS = 3 * 32 * PITCH + 2 * 32 * ATT_PP;
    extern __shared__ __attribute__((aligned(16))) char smc[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    MfmaHead m;
    m.hb = blockIdx.x * 4 + w;
    if (m.hb >= a.B * a.heads) return;
    m.b = m.hb / a.heads; m.h = m.hb % a.heads; m.n16 = lane & 15; m.g = lane >> 4;
    char* kimg = smc + w * WAVE_LDS;
    char* qimg = kimg + 32 * PITCH;
    char* doimg = qimg + 32 * PITCH;
    char* pdimg = doimg + 32 * PITCH;       // [query][key] dropped / rescaled probabilities
    char* dsimg = pdimg + 32 * ATT_PP;      // [query][key] d(logits)
    bf16x8 kf[2][KS], qf[2][KS], vf[2][KS], dof[2][KS];
    load_slice<D>((const bf16*)a.K + (size_t)m.b * a.Tk * a.ldk + m.h * D, a.ldk, a.Tk, m, kf, kimg);
    load_slice<D>((const bf16*)a.Q + (size_t)m.b * a.Tq * a.ldq + m.h * D, a.ldq, a.Tq, m, qf, qimg);
    load_slice<D>((const bf16*)a.V + (size_t)m.b * a.Tk * a.ldv + m.h * D, a.ldv, a.Tk, m, vf, nullptr);
    load_slice<D>((const bf16*)a.dO + (size_t)m.b * a.Tq * a.lddo + m.h * D, a.lddo, a.Tq, m, dof, doimg);
    f32x4 st[2][2], dp[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            st[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                st[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][ks], qf[it][ks], st[jt][it], 0, 0, 0);
                dp[jt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[jt][ks], dof[it][ks], dp[jt][it], 0, 0, 0);     // dP^T = V.dO^T
            }
        }
    uint32_t masked[2];
    softmax_t(a, m, st, masked);
    const uint32_t thresh = dropout_threshold(a.drop_p);
    const float ks_ = (a.drop_p > 0.f) ? 1.f / (1.f - a.drop_p) : 1.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = 16 * it + m.n16;
        float delta = 0.f;
        f32x4 pd[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int j0 = 16 * jt + 4 * m.g;
            uint32_t keep = 0xFu;
            if (a.drop_p > 0.f && i < a.Tq && j0 < a.Tk) keep = keep4(a.seed, a.stream_id, ((uint64_t)m.hb * a.Tq + i) * a.Tk + j0, thresh);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool kp = (keep >> r) & 1u;
                pd[jt][r] = kp ? st[jt][it][r] * ks_ : 0.f;
                dp[jt][it][r] = kp ? dp[jt][it][r] * ks_ : 0.f;          // d(normalised P)
                delta += st[jt][it][r] * dp[jt][it][r];
            }
        }
        delta = sum_across_rows(delta);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // masked_fill REPLACES the logit: no gradient reaches a masked position (a fully masked row has uniform, non-zero P)
                const bool msk = (masked[it] >> (4 * jt + r)) & 1u;
                dp[jt][it][r] = msk ? 0.f : st[jt][it][r] * (dp[jt][it][r] - delta) * a.scale;      // now dS^T
            }
            // [query][key] images for the products that sum over queries: 4 consecutive keys of query i, 8 bytes
            s16x4 pk, dk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pk[r] = __builtin_bit_cast(short, (bf16)pd[jt][r]);
                dk[r] = __builtin_bit_cast(short, (bf16)dp[jt][it][r]);
            }
            *reinterpret_cast<s16x4*>(pdimg + i * ATT_PP + (16 * jt + 4 * m.g) * 2) = pk;
            *reinterpret_cast<s16x4*>(dsimg + i * ATT_PP + (16 * jt + 4 * m.g) * 2) = dk;
        }
    }
    // dQ^T[c][i] = sum_j K^T[c][j] dS^T[j][i]
    {
        bf16x8 dsb[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) dsb[it] = pack8(dp[0][it], dp[1][it]);
        bf16* dQ = (bf16*)a.dQ + (size_t)m.b * a.Tq * a.lddq + m.h * D;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const bf16x8 kt = tr_operand<D, true>(kimg, ct, m);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsb[it], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const int i = 16 * it + m.n16;
                if (i < a.Tq) {
                    s16x4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(short, (bf16)o[e]);
                    *reinterpret_cast<s16x4*>(dQ + (size_t)i * a.lddq + ct * 16 + 4 * m.g) = r;
                }
            }
        }
    }
    // dK^T[c][j] = sum_i Q^T[c][i] dS[i][j],  dV^T[c][j] = sum_i dO^T[c][i] Pd[i][j]: B operand = key column j of 8 queries
    {
        const int q = m.n16 >> 2, p = m.n16 & 3;
        bf16x8 dsq[2], pdq[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int off = (8 * m.g + q) * ATT_PP + (16 * jt + 4 * p) * 2;
            dsq[jt] = join8(tr16(dsimg + off), tr16(dsimg + off + 4 * ATT_PP));
            pdq[jt] = join8(tr16(pdimg + off), tr16(pdimg + off + 4 * ATT_PP));
        }
        bf16* dK = (bf16*)a.dK + (size_t)m.b * a.Tk * a.lddk + m.h * D;
        bf16* dV = (bf16*)a.dV + (size_t)m.b * a.Tk * a.lddv + m.h * D;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const bf16x8 qt = tr_operand<D, false>(qimg, ct, m);
            const bf16x8 dot = tr_operand<D, false>(doimg, ct, m);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const f32x4 ok = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsq[jt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4 ov = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pdq[jt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const int j = 16 * jt + m.n16;
                if (j < a.Tk) {
                    s16x4 rk, rv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rk[e] = __builtin_bit_cast(short, (bf16)ok[e]); rv[e] = __builtin_bit_cast(short, (bf16)ov[e]); }
                    *reinterpret_cast<s16x4*>(dK + (size_t)j * a.lddk + ct * 16 + 4 * m.g) = rk;
                    *reinterpret_cast<s16x4*>(dV + (size_t)j * a.lddv + ct * 16 + 4 * m.g) = rv;
                }
            }
        }
    }
}

// bf16, head dim 32 / 64 / 128, <= 32 queries and keys, 16-byte-aligned row slices
bool mfma_ok(int dtype, const AttnArgs& a, bool bwd) {
    if (dtype != BLT_BF16 || blt_debug_get(16) == 1) return false;      // debug key 16 = 1: the VALU kernels (A/B)
    if (!(a.d == 32 || a.d == 64 || a.d == 128) || a.Tq > 32 || a.Tk > 32) return false;
    // a row-subset view takes the kernel its whole tensor would take, so that step t of an incremental pass reproduces row t of the full
    // pass bit for bit (the two kernels round P differently: bf16 MFMA operand here, fp32 on the VALU)
    if (a.q_rows > 32 || a.k_rows > 32) return false;
    auto al = [](const void* p, int ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 7) == 0; };
    if (!al(a.Q, a.ldq) || !al(a.K, a.ldk) || !al(a.V, a.ldv)) return false;
    if (!bwd) return (((uintptr_t)a.O) & 7) == 0 && (a.ldo & 3) == 0;
    return al(a.dO, a.lddo) && (((uintptr_t)a.dQ) & 7) == 0 && (((uintptr_t)a.dK) & 7) == 0 && (((uintptr_t)a.dV) & 7) == 0 && (a.lddq & 3) == 0 &&
           (a.lddk & 3) == 0 && (a.lddv & 3) == 0;
}

template <int D>
int launch_mfma(const AttnArgs& a, bool bwd, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B * a.heads + 3) / 4);
    constexpr int PITCH = D * 2 + 16;
    if (!bwd) {
        hipLaunchKernelGGL(attn_fwd_mfma_kernel<D>, dim3(grid), dim3(256), 4 * 32 * PITCH, s, a);
        return blt_check_launch("attn_fwd");
    }
    const size_t lds = 4 * (3 * 32 * PITCH + 2 * 32 * ATT_PP);
    if (lds > 64 * 1024) {
        static BltDevFlag set;
        if (!set.get()) {
            if (hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                blt_set_error("attn_bwd: hipFuncSetAttribute failed");
                return BLT_ERR_HIP;
            }
            set.set();
        }
    }
    hipLaunchKernelGGL(attn_bwd_mfma_kernel<D>, dim3(grid), dim3(256), lds, s, a);
    return blt_check_launch("attn_bwd");
}
int launch_mfma_d(const AttnArgs& a, bool bwd, hipStream_t s) {
    if (a.d == 32) return launch_mfma<32>(a, bwd, s);
    if (a.d == 64) return launch_mfma<64>(a, bwd, s);
    return launch_mfma<128>(a, bwd, s);
}

int check(const AttnArgs& a, bool bwd) {
    BLT_REQUIRE(a.Q && a.K && a.V, "attn: null Q/K/V");
    BLT_REQUIRE(a.B > 0 && a.heads > 0 && a.d > 0, "attn: bad sizes");
    BLT_REQUIRE(a.Tq > 0 && a.Tq <= 64 && a.Tk > 0 && a.Tk <= 64, "attn: Tq=%d Tk=%d must be in 1..64", a.Tq, a.Tk);
    BLT_REQUIRE(a.drop_p >= 0.f && a.drop_p < 1.f, "attn: bad dropout p");
    BLT_REQUIRE((a.q_rows == 0 || a.q_rows >= a.Tq) && (a.k_rows == 0 || a.k_rows >= a.Tk), "attn: q_rows=%d / k_rows=%d must be 0 or >= Tq / Tk", a.q_rows, a.k_rows);
    if (!bwd) BLT_REQUIRE(a.O != nullptr, "attn_fwd: null O");
    else BLT_REQUIRE(a.dO && a.dQ && a.dK && a.dV && a.q_rows == 0 && a.k_rows == 0, "attn_bwd: null gradient pointer (or a row-subset view: forward only)");
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

#ifdef BLT_EXPERIMENTS
bool blt_attn_out_fwd_ok(int dtype, const AttnArgs& a) {
    return dtype == BLT_BF16 && a.d == 64 && a.heads >= 1 && a.heads <= 8 && a.Tq <= 32 && a.Tk <= 32 && a.q_rows == 0 && a.k_rows == 0 && mfma_ok(dtype, a, false) && a.Wo && a.Y &&
           a.ldwo % 8 == 0 && (((uintptr_t)a.Wo) & 15) == 0 && a.ldo >= a.heads * 64;
}
int blt_attn_out_fwd(int dtype, const AttnArgs& a, hipStream_t s) {
    const int rc = check(a, false);
    if (rc) return rc;
    BLT_REQUIRE(blt_attn_out_fwd_ok(dtype, a), "attn_out_fwd: needs bf16, d = 64, heads <= 8, Tq, Tk <= 32, 16-byte aligned operands");
    constexpr int PITCH = 64 * 2 + 16;
    const size_t lds = (size_t)a.heads * 32 * PITCH + 32 * ((size_t)a.heads * 64 * 2 + 16);
    static BltDevFlag set;
    if (lds > 64 * 1024 && !set.get()) {
        if (hipFuncSetAttribute((const void*)attn_out_fwd_kernel<64, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            blt_set_error("attn_out_fwd: hipFuncSetAttribute failed");
            return BLT_ERR_HIP;
        }
        set.set();
    }
    hipLaunchKernelGGL((attn_out_fwd_kernel<64, 512>), dim3((unsigned)a.B), dim3((unsigned)a.heads * 64), lds, s, a);
    return blt_check_launch("attn_out_fwd");
}
#endif

int blt_attn_fwd(int dtype, const AttnArgs& a, hipStream_t s) {
    BLT_REQUIRE(dtype == BLT_F32 || dtype == BLT_BF16, "attn_fwd: bad dtype");
    int rc = check(a, false);
    if (rc) return rc;
    if (mfma_ok(dtype, a, false)) return launch_mfma_d(a, false, s);
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
    if (mfma_ok(dtype, a, true)) return launch_mfma_d(a, true, s);
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
