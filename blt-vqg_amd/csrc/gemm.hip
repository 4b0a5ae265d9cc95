// MFMA GEMM for gfx950: linear layers (fwd / dgrad / wgrad) and the implicit-GEMM convolution of the frozen
// ResNet-18 stack share one LDS-tiled kernel.
//
//   * T = bf16: v_mfma_f32_16x16x32_bf16 (8 bf16 per lane per operand), fp32 accumulate.
//   * T = f32 : v_mfma_f32_16x16x4_f32 (exact fp32 fma chain) — the parity mode.
//   * 256 threads = 4 waves (2x2); block tile BMxBN in {128x128, 128x64, 64x64}; K-tile = 128 bytes of K per row
//     (64 bf16 / 32 f32); double-buffered LDS, register-staged prefetch (global->VGPR issued before the
//     MFMAs of the current tile, VGPR->LDS after them), one barrier per K-tile.
//   * An operand keeps its GLOBAL orientation in LDS: k-contiguous operands are read with ds_read_b128
//     (bf16) / ds_read_b32 (f32); m/n-contiguous ("transposed") operands are read with ds_read_b64_tr_b16
//     (bf16, hardware transpose) / ds_read_b32 (f32).  No operand is ever transposed in HBM.
//   * A-operand loaders: plain matrix, NHWC implicit-GEMM gather (3x3 / 1x1 convs), and the 7x7/2 stem conv on a
//     zero-bordered NHWC4 image (no bounds checks, K = 7 rows x 8 pixels x 4 channels).
//   * split-K (gridDim.y slices of the K loop, fp32 atomic accumulation) for weight-gradient GEMMs whose output has too
//     few tiles to fill 256 CUs.
//   * The accumulators go through LDS once so that the epilogue (bias, position table, ReLU, dropout,
//     ReLU/dropout-backward mask, residual, accumulate, BN column statistics) runs on 8 consecutive columns per
//     thread and stores 16 bytes per lane.
//
// Replaces: torch.nn.Linear / F.conv2d call sites of the reference hot path
// (models/transformer_layers.py:453-456,489-491,530,400-408; models/encoder_cnn.py:20,33; models/iq.py:39,72-78).
#include <unordered_map>
#include <vector>
#include "kernels.h"

namespace {

enum { LD_PLAIN = 0, LD_CONV = 1, LD_STEM = 2 };

template <typename T> struct Frag;
template <> struct Frag<bf16> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef float type; };

__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

template <typename T, int BM_, int BN_, bool TA_, bool TB_, int LOADER_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, LOADER = LOADER_;
    static constexpr bool TA = TA_, TB = TB_;
    static constexpr int ES = sizeof(T);
    static constexpr int CE = 16 / ES;           // elements per 16-byte chunk
    static constexpr int BK = 128 / ES;          // K elements per tile (128 bytes per row)
    static constexpr int KSTEP = (ES == 2) ? 32 : 4;
    static constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    static constexpr int SA = 144;               // row stride (bytes) of a k-contiguous LDS image: 128 + 16 pad
    static constexpr int SAT = BM * ES + 16;     // row stride of an m-contiguous image [BK][BM]
    static constexpr int SBT = BN * ES + 16;
    static constexpr int A_BYTES = TA ? BK * SAT : BM * SA;
    static constexpr int B_BYTES = TB ? BK * SBT : BN * SA;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int CH_A = BM * 8 / 256, CH_B = BN * 8 / 256;
    static constexpr int CS = BN + 4;            // fp32 C-tile row stride (floats)
    static constexpr int RS_OFF = BM * CS * 4;   // [4 waves][BM] fp32 row sums of op(A) (transA form), behind the C tile
    static constexpr int EPI_BYTES = BM * CS * 4 + (TA ? 4 * BM * 4 : 0);
    static constexpr int LDS_BYTES = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
};

// per-thread geometry of the rows a thread stages for the implicit-GEMM loaders
template <int CH>
struct ConvRows {
    int base[CH], h0[CH], w0[CH];
};

template <typename C, typename T>
__device__ __forceinline__ void load_a(const GemmArgs& p, const T* __restrict__ Ag, const ConvRows<C::CH_A>& cr, int m0, int k0, int tid,
                                       uint4 (&ra)[C::CH_A]) {
#pragma unroll
    for (int i = 0; i < C::CH_A; ++i) {
        const int c = tid + i * 256;
        if constexpr (C::LOADER == LD_CONV) {
            const int k = k0 + (c & 7) * C::CE;
            const int q = k >> p.cg.cin_log2, ci = k & (p.cg.Cin - 1);
            const int r = q / p.cg.KW, s = q - r * p.cg.KW;
            const int hi = cr.h0[i] + r, wi = cr.w0[i] + s;
            const bool ok = (cr.base[i] >= 0) && (k < p.K) && ((unsigned)hi < (unsigned)p.cg.Hi) && ((unsigned)wi < (unsigned)p.cg.Wi);
            const T* src = Ag + (((size_t)(cr.base[i] + hi * p.cg.in_pitch + wi)) << p.cg.cin_log2) + ci;
            { uint4 v_ = make_uint4(0u, 0u, 0u, 0u); if (ok) v_ = *reinterpret_cast<const uint4*>(src); ra[i] = v_; }
        } else if constexpr (C::LOADER == LD_STEM) {
            // zero-bordered NHWC4 image: k = r*32 + pixel*4 + c ; every 16-byte chunk is in bounds and aligned
            const int k = k0 + (c & 7) * C::CE;
            const bool ok = (cr.base[i] >= 0) && (k < p.K);
            const T* src = Ag + ((size_t)(cr.base[i] + (k >> 5) * p.cg.Wi + ((k & 31) >> 2)) << 2) + (k & 3);
            { uint4 v_ = make_uint4(0u, 0u, 0u, 0u); if (ok) v_ = *reinterpret_cast<const uint4*>(src); ra[i] = v_; }
        } else if constexpr (!C::TA) {
            const int gm = m0 + (c >> 3), gk = k0 + (c & 7) * C::CE;
            const bool ok = (gm < p.M) && (gk < p.K);
            { uint4 v_ = make_uint4(0u, 0u, 0u, 0u); if (ok) v_ = *reinterpret_cast<const uint4*>(Ag + (size_t)gm * p.lda + gk); ra[i] = v_; }
        } else {
            constexpr int CPR = C::BM * C::ES / 16;
            const int gk = k0 + c / CPR, gm = m0 + (c % CPR) * C::CE;
            const bool ok = (gk < p.K) && (gm < p.M);
            { uint4 v_ = make_uint4(0u, 0u, 0u, 0u); if (ok) v_ = *reinterpret_cast<const uint4*>(Ag + (size_t)gk * p.lda + gm); ra[i] = v_; }
        }
    }
}

template <typename C, typename T>
__device__ __forceinline__ void load_b(const GemmArgs& p, const T* __restrict__ Bg, int n0, int k0, int tid, uint4 (&rb)[C::CH_B]) {
#pragma unroll
    for (int i = 0; i < C::CH_B; ++i) {
        const int c = tid + i * 256;
        if constexpr (!C::TB) {
            const int gn = n0 + (c >> 3), gk = k0 + (c & 7) * C::CE;
            const bool ok = (gn < p.N) && (gk < p.K);
            { uint4 v_ = make_uint4(0u, 0u, 0u, 0u); if (ok) v_ = *reinterpret_cast<const uint4*>(Bg + (size_t)gn * p.ldb + gk); rb[i] = v_; }
        } else {
            constexpr int CPR = C::BN * C::ES / 16;
            const int gk = k0 + c / CPR, gn = n0 + (c % CPR) * C::CE;
            const bool ok = (gk < p.K) && (gn < p.N);
            { uint4 v_ = make_uint4(0u, 0u, 0u, 0u); if (ok) v_ = *reinterpret_cast<const uint4*>(Bg + (size_t)gk * p.ldb + gn); rb[i] = v_; }
        }
    }
}

template <typename C>
__device__ __forceinline__ void store_ab(char* a_buf, char* b_buf, int tid, const uint4 (&ra)[C::CH_A], const uint4 (&rb)[C::CH_B]) {
#pragma unroll
    for (int i = 0; i < C::CH_A; ++i) {
        const int c = tid + i * 256;
        if constexpr (!C::TA) {
            *reinterpret_cast<uint4*>(a_buf + (c >> 3) * C::SA + (c & 7) * 16) = ra[i];
        } else {
            constexpr int CPR = C::BM * C::ES / 16;
            *reinterpret_cast<uint4*>(a_buf + (c / CPR) * C::SAT + (c % CPR) * 16) = ra[i];
        }
    }
#pragma unroll
    for (int i = 0; i < C::CH_B; ++i) {
        const int c = tid + i * 256;
        if constexpr (!C::TB) {
            *reinterpret_cast<uint4*>(b_buf + (c >> 3) * C::SA + (c & 7) * 16) = rb[i];
        } else {
            constexpr int CPR = C::BN * C::ES / 16;
            *reinterpret_cast<uint4*>(b_buf + (c / CPR) * C::SBT + (c % CPR) * 16) = rb[i];
        }
    }
}

typedef __attribute__((ext_vector_type(8))) short s16x8;

template <typename C, typename T>
__device__ __forceinline__ void compute_tile(const char* a_buf, const char* b_buf, int wm, int wn, int l15, int lg, f32x4 (&acc)[C::TM][C::TN]) {
#pragma unroll
    for (int ks = 0; ks < C::BK / C::KSTEP; ++ks) {
        typename Frag<T>::type af[C::TM], bfr[C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
            const int r0 = wm * C::WM + i * 16;
            if constexpr (C::ES == 2) {
                if constexpr (!C::TA) {
                    af[i] = *reinterpret_cast<const bf16x8*>(a_buf + (r0 + l15) * C::SA + ks * 64 + lg * 16);
                } else {
                    const char* q = a_buf + (ks * 32 + lg * 8 + (l15 >> 2)) * C::SAT + (r0 + (l15 & 3) * 4) * 2;
                    const s16x4 lo = lds_tr16(q), hi = lds_tr16(q + 4 * C::SAT);
                    af[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            } else {
                if constexpr (!C::TA) af[i] = *reinterpret_cast<const float*>(a_buf + (r0 + l15) * C::SA + (ks * 4 + lg) * 4);
                else af[i] = *reinterpret_cast<const float*>(a_buf + (ks * 4 + lg) * C::SAT + (r0 + l15) * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
            const int c0 = wn * C::WN + j * 16;
            if constexpr (C::ES == 2) {
                if constexpr (!C::TB) {
                    bfr[j] = *reinterpret_cast<const bf16x8*>(b_buf + (c0 + l15) * C::SA + ks * 64 + lg * 16);
                } else {
                    const char* q = b_buf + (ks * 32 + lg * 8 + (l15 >> 2)) * C::SBT + (c0 + (l15 & 3) * 4) * 2;
                    const s16x4 lo = lds_tr16(q), hi = lds_tr16(q + 4 * C::SBT);
                    bfr[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            } else {
                if constexpr (!C::TB) bfr[j] = *reinterpret_cast<const float*>(b_buf + (c0 + l15) * C::SA + (ks * 4 + lg) * 4);
                else bfr[j] = *reinterpret_cast<const float*>(b_buf + (ks * 4 + lg) * C::SBT + (c0 + l15) * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
                if constexpr (C::ES == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
    }
}

// 8 consecutive elements of type T kept packed (2 x 16 B for fp32, 1 x 16 B for bf16) so that many rows can be in flight
template <typename T> struct Raw8;
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    __device__ __forceinline__ void get(float* v) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};
template <> struct Raw8<bf16> {
    bf16x8 a;
    __device__ __forceinline__ void load(const bf16* p) { a = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ void get(float* v) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
};

// Epilogue shared by the GEMM kernels: Cs is the block's fp32 result tile in LDS (row stride BN + 4 floats).
// A thread owns ONE group of 8 columns (so bias is loaded once) and BM*BN/2048 rows of it; the global loads of up to four rows
// (residual, ReLU/dropout mask, position-table row, old C) are issued together before any arithmetic, otherwise the epilogue is a
// chain of dependent L2 round trips.
template <typename T, int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, const float* Cs, int tile_m, int m0, int n0, int tid) {
    constexpr int ES = sizeof(T), CE = 16 / ES, CS = BN + 4;
    constexpr int GPR = BN / 8;                // column groups per tile row
    constexpr int RPT = BM * GPR / 256;        // rows per thread
    constexpr int RSTEP = 256 / GPR;           // distance between a thread's rows

    // ---- per-column statistics of the raw result (BatchNorm2d batch statistics, encoder_cnn.py:33) --------
    if (p.stat_sum != nullptr) {
        for (int c = tid; c < BN * 2; c += 256) {
            const int col = c % BN, half = c / BN;
            float s = 0.f, s2 = 0.f;
            for (int r = half * (BM / 2); r < (half + 1) * (BM / 2); ++r) {
                const float v = Cs[r * CS + col];
                s += v;
                s2 += v * v;
            }
            if (n0 + col < p.N) {
                p.stat_sum[(size_t)(tile_m * 2 + half) * p.N + n0 + col] = s;
                p.stat_sq[(size_t)(tile_m * 2 + half) * p.N + n0 + col] = s2;
            }
        }
    }
    if (gridDim.y > 1) {
        // split-K partial: fp32 atomic accumulation into a zero-initialised (or accumulating) C; no other epilogue terms.
        // consecutive lanes add consecutive floats: every atomic wave-instruction covers one contiguous 256-byte segment
        float* Cg = (float*)p.C;
        for (int idx = tid; idx < BM * BN; idx += 256) {
            const int row = idx / BN, col = idx % BN;
            const int m = m0 + row, n = n0 + col;
            if (m < p.M && n < p.N) atomicAdd(Cg + (size_t)m * p.ldc + n, Cs[row * CS + col] * p.alpha);
        }
        return;
    }

    const int cgp = tid % GPR, row0 = tid / GPR;
    const int n = n0 + cgp * 8;
    if (n >= p.N) return;
    const int nv = (p.N - n < 8) ? (p.N - n) : 8;
    const bool out32 = p.out_f32 || ES == 4;
    const uint32_t thresh = dropout_threshold(p.drop_p);
    const float keep_scale = (p.drop_p > 0.f) ? 1.f / (1.f - p.drop_p) : 1.f;
    const int drop_ld = (p.N + 7) & ~7;
    const bool fast = (nv == 8) && (out32 ? (p.ldc % 4) == 0 : (p.ldc % CE) == 0) && (!p.R || (p.ldr % CE) == 0) &&
                      (!p.maskY || (p.ldm % CE) == 0) && (!p.C2 || (p.ldc2 % CE) == 0) && (!p.rowtab || (p.ldt % 4) == 0) &&
                      (!p.bias || (((uintptr_t)(p.bias + n)) & 15) == 0);
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
    if (p.bias != nullptr) {
        if (fast) Vec8<float>::load(p.bias + n, bias);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) bias[e] = p.bias[n + e];
        }
    }

    if (fast) {
        constexpr int NB = (RPT < 4) ? RPT : 4;
#pragma unroll
        for (int i0 = 0; i0 < RPT; i0 += NB) {
            Raw8<T> rR[NB], rM[NB], rC[NB];
            Raw8<float> rC32[NB], rT[NB];
            bool ok[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int m = m0 + row0 + (i0 + i) * RSTEP;
                ok[i] = m < p.M;
                if (!ok[i]) continue;
                if (p.R) rR[i].load((const T*)p.R + (size_t)m * p.ldr + n);
                if (p.maskY) rM[i].load((const T*)p.maskY + (size_t)m * p.ldm + n);
                if (p.rowtab) rT[i].load(p.rowtab + (size_t)p.rowidx[m] * p.ldt + n);
                if (p.accumulate) {
                    if (out32) rC32[i].load((const float*)p.C + (size_t)m * p.ldc + n);
                    else rC[i].load((const T*)p.C + (size_t)m * p.ldc + n);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (!ok[i]) continue;
                const int row = row0 + (i0 + i) * RSTEP, m = m0 + row;
                float v[8], t[8];
                {
                    const float4 x0 = *reinterpret_cast<const float4*>(&Cs[row * CS + cgp * 8]);
                    const float4 x1 = *reinterpret_cast<const float4*>(&Cs[row * CS + cgp * 8 + 4]);
                    v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias[e];
                if (p.rowtab) {
                    rT[i].get(t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.drop_p > 0.f) {
                    uint32_t w[4];
                    dropout_words(p.seed, p.stream_id, ((uint64_t)m * (uint64_t)drop_ld + (uint64_t)n) >> 3, w);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (dropout_lane(w, e) >= thresh) ? v[e] * keep_scale : 0.f;
                }
                if (p.maskY) {
                    rM[i].get(t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (t[e] != 0.f) ? v[e] * p.mask_scale : 0.f;
                }
                if (p.C2) Vec8<T>::store((T*)p.C2 + (size_t)m * p.ldc2 + n, v);
                if (p.R) {
                    rR[i].get(t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t[e];
                }
                if (out32) {
                    if (p.accumulate) {
                        rC32[i].get(t);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += t[e];
                    }
                    Vec8<float>::store((float*)p.C + (size_t)m * p.ldc + n, v);
                } else {
                    if (p.accumulate) {
                        rC[i].get(t);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += t[e];
                    }
                    Vec8<T>::store((T*)p.C + (size_t)m * p.ldc + n, v);
                }
            }
        }
        return;
    }

    // ---- general path: ragged last column group / unaligned leading dimensions (element-wise accesses) ----
    for (int i = 0; i < RPT; ++i) {
        const int row = row0 + i * RSTEP, m = m0 + row;
        if (m >= p.M) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Cs[row * CS + cgp * 8 + e] * p.alpha + bias[e];
        if (p.rowtab != nullptr) {
            const float* tr = p.rowtab + (size_t)p.rowidx[m] * p.ldt + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) v[e] += tr[e];
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.drop_p > 0.f) {
            uint32_t w[4];
            dropout_words(p.seed, p.stream_id, ((uint64_t)m * (uint64_t)drop_ld + (uint64_t)n) >> 3, w);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (dropout_lane(w, e) >= thresh) ? v[e] * keep_scale : 0.f;
        }
        if (p.maskY != nullptr) {
            const T* mp = (const T*)p.maskY + (size_t)m * p.ldm + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) v[e] = (to_f32(mp[e]) != 0.f) ? v[e] * p.mask_scale : 0.f;
        }
        if (p.C2 != nullptr) {
            T* cp = (T*)p.C2 + (size_t)m * p.ldc2 + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) cp[e] = from_f32<T>(v[e]);
        }
        if (p.R != nullptr) {
            const T* rp = (const T*)p.R + (size_t)m * p.ldr + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) v[e] += to_f32(rp[e]);
        }
        if (out32) {
            float* cp = (float*)p.C + (size_t)m * p.ldc + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) cp[e] = p.accumulate ? cp[e] + v[e] : v[e];
        } else {
            T* cp = (T*)p.C + (size_t)m * p.ldc + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) cp[e] = from_f32<T>(p.accumulate ? to_f32(cp[e]) + v[e] : v[e]);
        }
    }
}

template <typename T, int BM, int BN, bool TA, bool TB, int LOADER>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    typedef Cfg<T, BM, BN, TA, TB, LOADER> C;
    constexpr int ES = C::ES, CE = C::CE, BK = C::BK, CS = C::CS;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ Ag = (const T*)p.A;
    const T* __restrict__ Bg = (const T*)p.B;

    ConvRows<C::CH_A> cr;
    if constexpr (LOADER != LD_PLAIN) {
#pragma unroll
        for (int i = 0; i < C::CH_A; ++i) {
            const int gm = m0 + ((tid + i * 256) >> 3);
            const int hw = p.cg.Ho * p.cg.Wo;
            const int n = gm / hw, rem = gm - n * hw;
            const int ho = rem / p.cg.Wo, wo = rem - ho * p.cg.Wo;
            if constexpr (LOADER == LD_CONV) {
                cr.base[i] = (gm < p.M && ho < p.cg.Hov && wo < p.cg.Wov) ? n * p.cg.in_rows * p.cg.in_pitch : -1;
                cr.h0[i] = ho * p.cg.stride - p.cg.pad;
                cr.w0[i] = wo * p.cg.stride - p.cg.pad;
            } else {   // stem: Hi/Wi are the PADDED image dims; the border already holds the conv padding
                cr.base[i] = (gm < p.M) ? (n * p.cg.Hi + ho * p.cg.stride) * p.cg.Wi + wo * p.cg.stride : -1;
                cr.h0[i] = 0; cr.w0[i] = 0;
            }
        }
    }

    // split-K: gridDim.y slices of the K-tile range
    const int nk_total = (p.K + BK - 1) / BK;
    const int per = (nk_total + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kt0 = (int)blockIdx.y * per;
    const int kt1 = (kt0 + per < nk_total) ? kt0 + per : nk_total;

    f32x4 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lg = lane >> 4;

    // Row sums of op(A) (the bias gradient of the weight-gradient form) fall out of the A staging registers: with an m-contiguous
    // A every chunk a thread stages covers the SAME CE columns m0 + (tid % CPR)*CE.., so it keeps CE running sums; only the
    // workgroups of the first column tile do it, once per K-slice.
    constexpr int CPR_A = BM * ES / 16;
    const bool do_rs = TA && p.a_rowsum != nullptr && tile_n == 0;
    float rs[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) rs[e] = 0.f;
    auto add_rs = [&](const uint4 (&ra)[C::CH_A]) {
#pragma unroll
        for (int i = 0; i < C::CH_A; ++i) {
            const T* v = reinterpret_cast<const T*>(&ra[i]);
#pragma unroll
            for (int e = 0; e < CE; ++e) rs[e] += to_f32(v[e]);
        }
    };

    if (kt0 < kt1) {
        uint4 ra[C::CH_A], rb[C::CH_B];
        load_a<C, T>(p, Ag, cr, m0, kt0 * BK, tid, ra);
        load_b<C, T>(p, Bg, n0, kt0 * BK, tid, rb);
        store_ab<C>(smem, smem + C::A_BYTES, tid, ra, rb);
        if constexpr (TA) { if (do_rs) add_rs(ra); }
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            char* a_cur = smem + cur * C::STAGE;
            char* a_nxt = smem + (cur ^ 1) * C::STAGE;
            const bool more = kt + 1 < kt1;
            if (more) {
                load_a<C, T>(p, Ag, cr, m0, (kt + 1) * BK, tid, ra);
                load_b<C, T>(p, Bg, n0, (kt + 1) * BK, tid, rb);
            }
            compute_tile<C, T>(a_cur, a_cur + C::A_BYTES, wm, wn, l15, lg, acc);
            if (more) {
                store_ab<C>(a_nxt, a_nxt + C::A_BYTES, tid, ra, rb);
                if constexpr (TA) { if (do_rs) add_rs(ra); }
            }
            __syncthreads();
        }
    }
    if constexpr (TA) {
        if (do_rs) {
            // lanes with equal (lane % CPR) hold the same columns: butterfly over the remaining lane bits, park the wave's sums behind
            // the C tile; after the barrier below one thread per column adds the four waves and issues ONE atomic (device-scope
            // float atomics on a single address serialise, so the count per address is what matters)
#pragma unroll
            for (int o = 32; o >= CPR_A; o >>= 1)
#pragma unroll
                for (int e = 0; e < CE; ++e) rs[e] += __shfl_xor(rs[e], o, 64);
            float* Rs = reinterpret_cast<float*>(smem + C::RS_OFF);
            if (lane < CPR_A) {
#pragma unroll
                for (int e = 0; e < CE; ++e) Rs[wave * BM + lane * CE + e] = rs[e];
            }
        }
    }

    // ---- accumulators -> LDS (fp32 C tile) ---------------------------------------------------------
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm * C::WM + i * 16 + lg * 4 + r) * CS + wn * C::WN + j * 16 + l15] = acc[i][j][r];
    __syncthreads();

    if constexpr (TA) {
        if (do_rs && tid < BM && m0 + tid < p.M) {
            const float* Rs = reinterpret_cast<const float*>(smem + C::RS_OFF);
            atomicAdd(p.a_rowsum + m0 + tid, (Rs[tid] + Rs[BM + tid] + Rs[2 * BM + tid] + Rs[3 * BM + tid]) * p.alpha);
        }
    }

    gemm_epilogue<T, BM, BN>(p, Cs, tile_m, m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16, both operands k-contiguous (Linear forward, implicit-GEMM conv, stem): LDS-DMA ring.
//   * global_load_lds_dwordx4 writes each operand tile straight into LDS (no VGPR staging, no ds_write); the LDS image is
//     lane-linear (unpadded 128-byte rows), bank conflicts are removed by an XOR swizzle applied on the SOURCE chunk index
//     (slot s of row r holds global chunk s ^ (r & 7)) and again on the ds_read_b128 address;
//   * NST = 3 stage ring, two K-tiles in flight across the barrier: counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier
//     per K-tile; out-of-range chunks (M/N/K tails, conv padding) are fetched from a 16-byte zero page.
// ---------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint4 g_zero_page[1];

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

template <int BM, int BN, int LOADER, int NST_>
struct DmaCfg {
    // ring depth.  128x128 tile: 2 stages (one K-tile in flight per block) so that TWO blocks fit a CU (68 KB each: the second
    // block's MFMAs cover this block's DMA wait) — unless the grid has no second block to give a CU (<= 256 tiles, e.g. the
    // 7x7x512 convs: 196 tiles), where a lone block is bound by the DMA landing latency (~0.9 us per K-tile); it then takes 4
    // stages (three tiles in flight, 128 KB).  Smaller tiles: 3 stages (two tiles in flight).
    static constexpr int NST = NST_;
    static constexpr int BK = 64;
    static constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int CH_A = BM * 8 / 256, CH_B = BN * 8 / 256;      // DMA instructions per wave per tile
    static constexpr int CS = BN + 4;
    static constexpr int LDS_BYTES = (NST * STAGE > BM * CS * 4) ? NST * STAGE : BM * CS * 4;
};

#ifdef BLT_STAMP
__device__ unsigned long long g_stamps[1 << 16];      // [block][8] phase time stamps of the DMA kernel (experiment builds only)
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int BM, int BN, int LOADER, int NST>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const GemmArgs p) {
    typedef DmaCfg<BM, BN, LOADER, NST> C;
    STAMP(0);
    constexpr int BK = C::BK, CS = C::CS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: it addresses the LDS-DMA destination
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bf16* __restrict__ Ag = (const bf16*)p.A;
    const bf16* __restrict__ Bg = (const bf16*)p.B;
    const void* zero = (const void*)g_zero_page;

    // DMA instruction j of this wave covers LDS chunks c = (j*4 + wave)*64 + lane: row = c >> 3, slot = c & 7, and fetches
    // global chunk gc = slot ^ (row & 7) of that row.  Everything that does not depend on the K-tile is hoisted here, so that
    // one DMA costs a 64-bit add, the bounds test and a select per K-tile (the per-tile (r, s, c0) of a conv tap is scalar).
    long a_off[C::CH_A];          // element offset of the lane's chunk at k0 = 0 (tap (0,0) for convs); < 0 = row out of range
    int a_gk[C::CH_A];            // k offset of the chunk inside a K-tile (gc * 8)
    int a_h0[C::CH_A], a_w0[C::CH_A];
#pragma unroll
    for (int j = 0; j < C::CH_A; ++j) {
        const int c = (j * 4 + wave) * 64 + lane;
        const int row = c >> 3, gc = (c & 7) ^ (row & 7);
        const int gm = m0 + row;
        a_gk[j] = gc * 8;
        a_h0[j] = 0; a_w0[j] = 0;
        if constexpr (LOADER == LD_PLAIN) {
            a_off[j] = (gm < p.M) ? (long)gm * p.lda + gc * 8 : -1;
        } else {
            const int hw = p.cg.Ho * p.cg.Wo;
            const int n = gm / hw, rem = gm - n * hw;
            const int ho = rem / p.cg.Wo, wo = rem - ho * p.cg.Wo;
            if constexpr (LOADER == LD_CONV) {
                a_h0[j] = ho * p.cg.stride - p.cg.pad;
                a_w0[j] = wo * p.cg.stride - p.cg.pad;
                a_off[j] = (gm < p.M && ho < p.cg.Hov && wo < p.cg.Wov)
                               ? (((long)n * p.cg.in_rows + a_h0[j]) * p.cg.in_pitch + a_w0[j]) * p.cg.Cin + gc * 8 : -(1L << 40);
            } else {   // stem: k = r*32 + pixel*4 + c on the zero-bordered NHWC4 image; a K-tile spans two filter rows
                a_off[j] = (gm < p.M) ? ((((long)n * p.cg.Hi + ho * 2 + (gc >> 2)) * p.cg.Wi + wo * 2 + (gc & 3) * 2) << 2) : -1;
            }
        }
    }
    long b_off[C::CH_B];
    int b_gk[C::CH_B];
#pragma unroll
    for (int j = 0; j < C::CH_B; ++j) {
        const int c = (j * 4 + wave) * 64 + lane;
        const int row = c >> 3, gc = (c & 7) ^ (row & 7);
        const int gn = n0 + row;
        b_gk[j] = gc * 8;
        b_off[j] = (gn < p.N) ? (long)gn * p.ldb + gc * 8 : -1;
    }

    auto issue = [&](int kt, int stage) {
        char* a_st = smem + stage * C::STAGE;
        char* b_st = a_st + C::A_BYTES;
        const int k0 = kt * BK;
        // scalar (wave-uniform) part of the A address for this K-tile
        long a_koff = k0;
        int tap_r = 0, tap_s = 0;
        if constexpr (LOADER == LD_CONV) {       // Cin % 64 == 0: the whole K-tile lies inside one filter tap (r, s)
            const int q = k0 >> p.cg.cin_log2, c0 = k0 & (p.cg.Cin - 1);
            tap_r = q / p.cg.KW;
            tap_s = q - tap_r * p.cg.KW;
            a_koff = ((long)tap_r * p.cg.in_pitch + tap_s) * p.cg.Cin + c0;
        } else if constexpr (LOADER == LD_STEM) {
            a_koff = ((long)(2 * kt) * p.cg.Wi) << 2;
        }
#pragma unroll
        for (int j = 0; j < C::CH_A; ++j) {
            bool ok = (k0 + a_gk[j] < p.K);
            if constexpr (LOADER == LD_CONV) {
                ok = ok && (a_off[j] > -(1L << 39)) && ((unsigned)(a_h0[j] + tap_r) < (unsigned)p.cg.Hi) &&
                     ((unsigned)(a_w0[j] + tap_s) < (unsigned)p.cg.Wi);
            } else {
                ok = ok && (a_off[j] >= 0);
            }
            const void* src = ok ? (const void*)(Ag + a_off[j] + a_koff) : zero;
            dma16(src, a_st + (j * 4 + wave) * 1024);
        }
#pragma unroll
        for (int j = 0; j < C::CH_B; ++j) {
            const bool ok = (b_off[j] >= 0) && (k0 + b_gk[j] < p.K);
            const void* src = ok ? (const void*)(Bg + b_off[j] + k0) : zero;
            dma16(src, b_st + (j * 4 + wave) * 1024);
        }
    };

    f32x4 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lg = lane >> 4;

    const int nk = (p.K + BK - 1) / BK;
    constexpr int AHEAD = C::NST - 1;                          // K-tiles kept in flight
    constexpr int PER_TILE = C::CH_A + C::CH_B;                // DMA instructions per wave per K-tile (vmcnt units)
    static_assert((AHEAD - 1) * PER_TILE <= 63, "vmcnt is a 6-bit counter");
    STAMP(1);
    issue(0, 0);
    if (AHEAD > 1 && nk > 1) issue(1, 1);
    if (AHEAD > 2 && nk > 2) issue(2, 2);
    if (AHEAD > 3 && nk > 3) issue(3, 3);
    static_assert(AHEAD <= 4, "prologue issues at most four K-tiles");
    for (int kt = 0; kt < nk; ++kt) {
        if (kt == 1) STAMP(2);
        // this wave's part of tile kt has landed once at most min(AHEAD-1, tiles issued beyond kt) tiles are still in flight
        const int beyond = (nk - 1 - kt < AHEAD - 1) ? nk - 1 - kt : AHEAD - 1;
        if (AHEAD > 3 && beyond == 3) wait_vmcnt<3 * PER_TILE>();
        else if (AHEAD > 2 && beyond == 2) wait_vmcnt<2 * PER_TILE>();
        else if (AHEAD > 1 && beyond >= 1) wait_vmcnt<PER_TILE>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                          // every wave's part of tile kt landed; stage (kt+AHEAD)%NST is free
        if (kt + AHEAD < nk) issue(kt + AHEAD, (kt + AHEAD) % C::NST);
        const char* a_st = smem + (kt % C::NST) * C::STAGE;
        const char* b_st = a_st + C::A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[C::TM], bfr[C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i) {
                const int r = wm * C::WM + i * 16 + l15;
                af[i] = *reinterpret_cast<const bf16x8*>(a_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
                const int r = wn * C::WN + j * 16 + l15;
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    STAMP(3);
    __syncthreads();          // all MFMA reads of the ring are done before the C tile overwrites it

    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm * C::WM + i * 16 + lg * 4 + r) * CS + wn * C::WN + j * 16 + l15] = acc[i][j][r];
    __syncthreads();
    STAMP(4);
    gemm_epilogue<bf16, BM, BN>(p, Cs, tile_m, m0, n0, tid);
    STAMP(5);
}
#ifdef BLT_STAMP
extern "C" int bltvqg_debug_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

#ifdef BLT_EXPERIMENTS      // (operators measured in the step and not adopted: include/bltvqg_hip_experiments.h)
// ---------------------------------------------------------------------------------------------------------------
// Linear (+ bias / ReLU / dropout / residual) with the FOLLOWING LayerNorm fused in: tile 32 rows x 256 columns, so that a workgroup
// owns complete rows (N <= 256); the four waves sit side by side (each 32 x 64), same LDS-DMA ring as above (3 stages of 4 KB A +
// 32 KB B).  Every workgroup streams the whole weight matrix, so the row tile is kept small: 84 workgroups at 2.6 k rows (with 64-row
// tiles only 42 CUs pulled 160-320 KB each and the fused launch was slower than GEMM + LayerNorm).  The stacks' LayerNorms all read the output of such a Linear; as separate launches they were 14 of the ~250 dependent
// launches of a step, each with its ~5 us launch-to-launch floor.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_dma_ln_kernel(const GemmArgs p) {
    constexpr int BM = 32, BN = 256, BK = 64, NST = 3, AHEAD = 2;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int CH_A = BM * 8 / 256, CH_B = BN * 8 / 256, PER_TILE = CH_A + CH_B;
    constexpr int TM = BM / 16, TN = 4, CS = BN + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    const bf16* __restrict__ Ag = (const bf16*)p.A;
    const bf16* __restrict__ Bg = (const bf16*)p.B;
    const void* zero = (const void*)g_zero_page;

    long a_off[CH_A], b_off[CH_B];
    int a_gk[CH_A], b_gk[CH_B];
#pragma unroll
    for (int j = 0; j < CH_A; ++j) {
        const int c = (j * 4 + wave) * 64 + lane;
        const int row = c >> 3, gc = (c & 7) ^ (row & 7);
        a_gk[j] = gc * 8;
        a_off[j] = (m0 + row < p.M) ? (long)(m0 + row) * p.lda + gc * 8 : -1;
    }
#pragma unroll
    for (int j = 0; j < CH_B; ++j) {
        const int c = (j * 4 + wave) * 64 + lane;
        const int row = c >> 3, gc = (c & 7) ^ (row & 7);
        b_gk[j] = gc * 8;
        b_off[j] = (row < p.N) ? (long)row * p.ldb + gc * 8 : -1;
    }
    auto issue = [&](int kt, int stage) {
        char* a_st = smem + stage * STAGE;
        char* b_st = a_st + A_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < CH_A; ++j) {
            const bool ok = (a_off[j] >= 0) && (k0 + a_gk[j] < p.K);
            dma16(ok ? (const void*)(Ag + a_off[j] + k0) : zero, a_st + (j * 4 + wave) * 1024);
        }
#pragma unroll
        for (int j = 0; j < CH_B; ++j) {
            const bool ok = (b_off[j] >= 0) && (k0 + b_gk[j] < p.K);
            dma16(ok ? (const void*)(Bg + b_off[j] + k0) : zero, b_st + (j * 4 + wave) * 1024);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lg = lane >> 4;
    const int nk = (p.K + BK - 1) / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<PER_TILE>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + AHEAD < nk) issue(kt + AHEAD, (kt + AHEAD) % NST);
        const char* a_st = smem + (kt % NST) * STAGE;
        const char* b_st = a_st + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = i * 16 + l15;
                af[i] = *reinterpret_cast<const bf16x8*>(a_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wave * 64 + j * 16 + l15;
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(i * 16 + lg * 4 + r) * CS + wave * 64 + j * 16 + l15] = acc[i][j][r];
    __syncthreads();

    // ---- epilogue: 32 column groups of 8 per row -> half a wave per row, 8 rows per pass, 8 passes; the row statistics are 32-lane
    // butterflies.  Same term order as gemm_epilogue (bias, ReLU, dropout, C2 copy, residual), then LayerNorm of the ROUNDED result.
    const int cgp = tid & 31, row0 = tid >> 5;
    const int n = cgp * 8;
    const bool colok = n < p.N;              // N is a multiple of 8 (checked on the host)
    static_assert(BM % 8 == 0, "8 rows per epilogue pass");
    const uint32_t thresh = dropout_threshold(p.drop_p);
    const float keep_scale = (p.drop_p > 0.f) ? 1.f / (1.f - p.drop_p) : 1.f;
    const int drop_ld = (p.N + 7) & ~7;
    float bias[8], gam[8], bet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias[e] = 0.f; gam[e] = 0.f; bet[e] = 0.f; }
    if (colok) {
        if (p.bias) Vec8<float>::load(p.bias + n, bias);
        Vec8<float>::load(p.ln_gamma + n, gam);
        Vec8<float>::load(p.ln_beta + n, bet);
    }
    const float inv_n = 1.f / (float)p.N;
#pragma unroll
    for (int i = 0; i < BM / 8; ++i) {
        const int row = row0 + i * 8, m = m0 + row;
        const bool ok = colok && m < p.M;
        float v[8], t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (ok) {
            const float4 x0 = *reinterpret_cast<const float4*>(&Cs[row * CS + n]);
            const float4 x1 = *reinterpret_cast<const float4*>(&Cs[row * CS + n + 4]);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias[e];
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (p.drop_p > 0.f) {
                uint32_t w[4];
                dropout_words(p.seed, p.stream_id, ((uint64_t)m * (uint64_t)drop_ld + (uint64_t)n) >> 3, w);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (dropout_lane(w, e) >= thresh) ? v[e] * keep_scale : 0.f;
            }
            if (p.C2) Vec8<bf16>::store((bf16*)p.C2 + (size_t)m * p.ldc2 + n, v);
            if (p.R) {
                Vec8<bf16>::load((const bf16*)p.R + (size_t)m * p.ldr + n, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += t[e];
            }
            Vec8<bf16>::store((bf16*)p.C + (size_t)m * p.ldc + n, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)(bf16)v[e];       // what a LayerNorm launch would read back
        }
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += v[e];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mu = s1 * inv_n;
        float q = 0.f;
        if (colok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - mu; q += d * d; }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rs = rsqrtf(q * inv_n + p.ln_eps);
        if (ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (v[e] - mu) * rs * gam[e] + bet[e];
            Vec8<bf16>::store((bf16*)p.ln_out + (size_t)m * p.N + n, t);
            if (cgp == 0) { p.ln_mean[m] = mu; p.ln_rstd[m] = rs; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Linear whose input is a LayerNorm: the LayerNorm runs on the A tile in LDS (GemmArgs::lnA_*).  64 x 64 tile, K <= 256: all (<= 4)
// K-tiles of A and B are DMA'd at once into their own stages (64 KB), then 4 threads per row normalise the row in place — each its
// 64-column K-tile, the swizzled slots in any order since only sums are needed — and the MFMA loop runs without further loads.
// Every column tile of a row block repeats the (cheap) normalisation; the first one writes LN(A) and the statistics for backward.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_dma_lnA_kernel(const GemmArgs p) {
    constexpr int BM = 64, BN = 64, BK = 64, MAXKT = 4, CS = BN + 4;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bf16* __restrict__ Ag = (const bf16*)p.A;
    const bf16* __restrict__ Bg = (const bf16*)p.B;
    const void* zero = (const void*)g_zero_page;
    const int nk = (p.K + BK - 1) / BK;
    float* gb = reinterpret_cast<float*>(smem + MAXKT * STAGE);       // gamma[K] | beta[K] (K <= 256)

    // ---- all K-tiles in flight: DMA instruction j of a tile covers chunks (j*4 + wave)*64 + lane, j < 2 for A and for B ----
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
        if (kt < nk) {
            char* a_st = smem + kt * STAGE;
            char* b_st = a_st + A_BYTES;
            const int k0 = kt * BK;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = (j * 4 + wave) * 64 + lane;
                const int row = c >> 3, gc = (c & 7) ^ (row & 7);
                const bool oka = (m0 + row < p.M) && (k0 + gc * 8 < p.K);
                dma16(oka ? (const void*)(Ag + (size_t)(m0 + row) * p.lda + k0 + gc * 8) : zero, a_st + (j * 4 + wave) * 1024);
                const bool okb = (n0 + row < p.N) && (k0 + gc * 8 < p.K);
                dma16(okb ? (const void*)(Bg + (size_t)(n0 + row) * p.ldb + k0 + gc * 8) : zero, b_st + (j * 4 + wave) * 1024);
            }
        }
    }
    for (int i = tid; i < p.K; i += 256) { gb[i] = p.lnA_gamma[i]; gb[256 + i] = p.lnA_beta[i]; }
    wait_vmcnt<0>();
    __syncthreads();

    // ---- LayerNorm of the 64 rows in place.  Wave w owns rows 16w .. 16w+15 in two passes of 8 rows; in a pass lane l reads slot
    // (l & 7) of row (l >> 3) of every K-tile: one wave instruction covers 8 whole 128-byte rows = 1 KB contiguous (conflict-free; a
    // thread-per-row-quarter mapping put 32 lanes on one bank group).  The swizzled slot order is irrelevant for the sums.
    {
        const int slot = lane & 7;
        float v[2][MAXKT][8];
        float mu2[2], rs2[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = wave * 16 + ps * 8 + (lane >> 3);
            float s1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < MAXKT; ++kt) {
                if (kt < nk) {
                    const bf16x8 x = *reinterpret_cast<const bf16x8*>(smem + kt * STAGE + r * 128 + slot * 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[ps][kt][e] = (float)x[e]; s1 += v[ps][kt][e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[ps][kt][e] = 0.f;
                }
            }
            s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
            const float mu = s1 / (float)p.K;
            float qq = 0.f;
#pragma unroll
            for (int kt = 0; kt < MAXKT; ++kt) {
                const int kc = kt * 64 + ((slot ^ (r & 7)) << 3);          // first column of the chunk that sits in this slot
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (kc + e < p.K) { const float d = v[ps][kt][e] - mu; qq += d * d; }
            }
            qq += __shfl_xor(qq, 1, 64); qq += __shfl_xor(qq, 2, 64); qq += __shfl_xor(qq, 4, 64);
            mu2[ps] = mu;
            rs2[ps] = rsqrtf(qq / (float)p.K + p.lnA_eps);
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = wave * 16 + ps * 8 + (lane >> 3);
            const int m = m0 + r;
#pragma unroll
            for (int kt = 0; kt < MAXKT; ++kt) {
                if (kt < nk) {
                    const int kc = kt * 64 + ((slot ^ (r & 7)) << 3);
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float y = (kc + e < p.K) ? (v[ps][kt][e] - mu2[ps]) * rs2[ps] * gb[kc + e] + gb[256 + kc + e] : 0.f;
                        o[e] = (bf16)y;
                    }
                    *reinterpret_cast<bf16x8*>(smem + kt * STAGE + r * 128 + slot * 16) = o;
                    if (tile_n == 0 && m < p.M && kc < p.K) *reinterpret_cast<bf16x8*>((bf16*)p.lnA_out + (size_t)m * p.K + kc) = o;
                }
            }
            if (tile_n == 0 && slot == 0 && m < p.M) { p.lnA_mean[m] = mu2[ps]; p.lnA_rstd[m] = rs2[ps]; }
        }
    }
    __syncthreads();

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const char* a_st = smem + kt * STAGE;
        const char* b_st = a_st + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 32 + i * 16 + l15;
                af[i] = *reinterpret_cast<const bf16x8*>(a_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wn * 32 + j * 16 + l15;
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wm * 32 + i * 16 + lg * 4 + r) * CS + wn * 32 + j * 16 + l15] = acc[i][j][r];
    __syncthreads();
    gemm_epilogue<bf16, BM, BN>(p, Cs, tile_m, m0, n0, tid);
}

int launch_dma_lnA(const GemmArgs& a, hipStream_t stream) {
    constexpr int LDS = 4 * (64 * 128 + 64 * 128) + 2 * 256 * 4;
    static BltDevFlag attr_set;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)gemm_dma_lnA_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            blt_set_error("gemm: hipFuncSetAttribute(%d) failed", LDS);
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    const long tiles = (long)cdiv(a.M, 64) * cdiv(a.N, 64);
    hipLaunchKernelGGL(gemm_dma_lnA_kernel, dim3((unsigned)tiles), dim3(256), LDS, stream, a);
    return blt_check_launch("gemm_dma_lnA");
}

int launch_dma_ln(const GemmArgs& a, hipStream_t stream) {
    constexpr int LDS = 3 * (32 * 128 + 256 * 128);
    static BltDevFlag attr_set;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)gemm_dma_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            blt_set_error("gemm: hipFuncSetAttribute(%d) failed", LDS);
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    hipLaunchKernelGGL(gemm_dma_ln_kernel, dim3(cdiv(a.M, 32)), dim3(256), LDS, stream, a);
    return blt_check_launch("gemm_dma_ln");
}

#endif      // BLT_EXPERIMENTS

template <int BM, int BN, int LOADER, int NST>
int launch_dma(const GemmArgs& a, hipStream_t stream) {
    typedef DmaCfg<BM, BN, LOADER, NST> C;
    static BltDevFlag attr_set;
    auto kern = gemm_dma_kernel<BM, BN, LOADER, NST>;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
            blt_set_error("gemm: hipFuncSetAttribute(%d) failed", C::LDS_BYTES);
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    const long tiles = (long)cdiv(a.M, BM) * cdiv(a.N, BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, 1), dim3(256), C::LDS_BYTES, stream, a);
    return blt_check_launch("gemm_dma");
}

template <int BM, int BN, int NST>
int dispatch_dma(const GemmArgs& a, hipStream_t s) {
    if (a.is_conv == 2) return launch_dma<BM, BN, LD_STEM, NST>(a, s);
    if (a.is_conv) return launch_dma<BM, BN, LD_CONV, NST>(a, s);
    return launch_dma<BM, BN, LD_PLAIN, NST>(a, s);
}

template <typename T, int BM, int BN, bool TA, bool TB, int LOADER>
int launch(const GemmArgs& a, int splits, hipStream_t stream) {
    typedef Cfg<T, BM, BN, TA, TB, LOADER> C;
    static BltDevFlag attr_set;
    auto kern = gemm_kernel<T, BM, BN, TA, TB, LOADER>;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
            blt_set_error("gemm: hipFuncSetAttribute(%d) failed", C::LDS_BYTES);
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    const long tiles = (long)cdiv(a.M, BM) * cdiv(a.N, BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(256), C::LDS_BYTES, stream, a);
    return blt_check_launch("gemm");
}

template <typename T, int BM, int BN>
int dispatch_layout(const GemmArgs& a, int splits, hipStream_t s) {
    if (a.is_conv == 2) return launch<T, BM, BN, false, false, LD_STEM>(a, splits, s);
    if (a.is_conv) return launch<T, BM, BN, false, false, LD_CONV>(a, splits, s);
    if (!a.transA && !a.transB) return launch<T, BM, BN, false, false, LD_PLAIN>(a, splits, s);
    if (!a.transA && a.transB) return launch<T, BM, BN, false, true, LD_PLAIN>(a, splits, s);
    if (a.transA && a.transB) return launch<T, BM, BN, true, true, LD_PLAIN>(a, splits, s);
    return launch<T, BM, BN, true, false, LD_PLAIN>(a, splits, s);
}

template <typename T>
int dispatch_tile(const GemmArgs& a, int bm, int bn, int splits, hipStream_t s) {
    if (bm == 128 && bn == 128) return dispatch_layout<T, 128, 128>(a, splits, s);
    if (bm == 128 && bn == 64) return dispatch_layout<T, 128, 64>(a, splits, s);
    return dispatch_layout<T, 64, 64>(a, splits, s);
}

}  // namespace

// process-wide tuning switches (bltvqg_debug_set): [0] = disable the LDS-DMA ring, [1] = force a tile, [2] = autotune mode,
// [3] = ring depth of the DMA kernels (0 = by grid size; 128x128: 1 = always 4 stages, 2 = always 2; 64x64: 3 = always 5, 4 = always 3)
// [8] = 1: round-1 kernels for the Linear GEMMs instead of gemm2.hip's planned-tile kernel (A/B); [9] / [10] = force its BM / BN
// [4]..[6] conv_pp.hip (force BN, XCD mapping, ring depth); [7] BatchNorm / LayerNorm variants (norm.hip, engine.hip)
// [11] weight-gradient flush mode (1 = ungrouped launches, n >= 2 = flush every n-1 layers); [12] = 1 phase stamps; [13] weight-gradient
// tile rows (128 / 256); [14] / [15] = 1 TIMING ABLATIONS ONLY (results wrong): skip the grouped weight-gradient launches / the conv stack
// [16] = 1 VALU attention kernels instead of the MFMA form; [17] = 1 bn1 as a separate pass (not fused into conv2's patch staging);
// [18] = 1 stem and max-pool as two launches; [19] conv3x3_pp tile form (1 = 128 positions everywhere, 2 = 256 positions also for
// Cout % 128 == 0).  Every key defaults to 0 = the shipped path; the A/B keys exist so that tests and measurements can compare forms.
static int g_debug[32] = {0};
void blt_debug_set(int key, int value) { if (key >= 0 && key < 32) g_debug[key] = value; }
int blt_debug_get(int key) { return (key >= 0 && key < 32) ? g_debug[key] : 0; }

struct Choice { int bm, bn, no_dma; };
static std::unordered_map<uint64_t, Choice> g_tuned;     // filled by autotune mode: measured best kernel per GEMM descriptor

static uint64_t tune_key(const GemmArgs& a, int dtype) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix((uint64_t)dtype); mix((uint64_t)a.M); mix((uint64_t)a.N); mix((uint64_t)a.K); mix((uint64_t)(a.transA * 2 + a.transB));
    mix((uint64_t)a.is_conv); mix((uint64_t)(a.cg.KH * 16 + a.cg.stride)); mix((uint64_t)(a.split_k > 0)); mix((uint64_t)a.out_f32);
    return h;
}

// heuristic: 128-wide tiles once they fill the chip; a 64-column tile when N <= 64 (Cout = 64 convolutions)
static Choice heuristic(const GemmArgs& a) {
    if (a.N <= 64) {
        if (cdiv(a.M, 128) >= 192) return {128, 64, 0};
        return {64, 64, 0};
    }
    const long t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    if (t128 >= 192) return {128, 128, 0};
    // weight-gradient form with split-K: the K-slices provide the parallelism, so keep the (more efficient) large tile
    if (a.split_k > 0 && a.transA && a.transB && t128 >= 32) return {128, 128, 0};
    return {64, 64, 0};
}

static Choice choose(const GemmArgs& a, int dtype) {
    Choice c = heuristic(a);
    auto it = g_tuned.find(tune_key(a, dtype));
    if (it != g_tuned.end()) c = it->second;
    const int ft = a.force_tile ? a.force_tile : g_debug[1];
    if (ft == 64) { c.bm = 64; c.bn = 64; }
    else if (ft == 128) { c.bm = 128; c.bn = 128; }
    else if (ft == 12864) { c.bm = 128; c.bn = 64; }
    if (a.no_dma || g_debug[0]) c.no_dma = 1;
    return c;
}

int blt_gemm_tile(const GemmArgs& a, int dtype) { return choose(a, dtype).bm; }
int blt_gemm_stat_rows(const GemmArgs& a, int dtype) { return 2 * cdiv(a.M, choose(a, dtype).bm); }

// split-K for fp32-accumulating outputs without epilogue terms (weight gradients; input gradients with a vocabulary-sized K)
int blt_gemm_splits(const GemmArgs& a, int dtype) {
    if (a.split_k <= 0) return 1;
    if (!(a.out_f32 || dtype == BLT_F32)) return 1;
    if (a.bias || a.relu || a.drop_p > 0.f || a.maskY || a.C2 || a.R || a.rowtab || a.stat_sum) return 1;
    const Choice c = choose(a, dtype);
    const long tiles = (long)cdiv(a.M, c.bm) * cdiv(a.N, c.bn);
    const int bk = (dtype == BLT_BF16) ? 64 : 32;
    const int nk = cdiv(a.K, bk);
    long s = (tiles >= 256) ? 1 : (384 / tiles);
    if (s > nk / 2) s = nk / 2;
    if (s > a.split_k) s = a.split_k;
    if (s < 1) s = 1;
    return (int)s;
}

static int run_choice(int dtype, const GemmArgs& a, const Choice& c, int splits, hipStream_t stream) {
    const bool dma_ok = (a.is_conv != 1) || (a.cg.Cin % 64 == 0);      // the DMA conv loader wants one filter tap per K-tile
    if (dtype == BLT_BF16 && !a.transA && !a.transB && splits == 1 && !c.no_dma && dma_ok) {
        if (c.bm == 128 && c.bn == 128) {
            // one block per CU at most -> deep ring (see DmaCfg); debug key 3: 1 = always deep, 2 = never
            const long tiles = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
            const bool deep = g_debug[3] == 1 || (g_debug[3] != 2 && tiles <= 256);
            return deep ? dispatch_dma<128, 128, 4>(a, stream) : dispatch_dma<128, 128, 2>(a, stream);
        }
        if (c.bm == 128 && c.bn == 64) return dispatch_dma<128, 64, 3>(a, stream);
        // small grids of 64x64 tiles (the transformer's Linear layers: 40-700 workgroups, K = 256..1024) are bound by the DMA landing
        // latency times the number of ring refills, so they keep four K-tiles in flight (80 KB ring)
        const long t64 = (long)cdiv(a.M, 64) * cdiv(a.N, 64);
        const bool deep64 = g_debug[3] == 3 || (g_debug[3] != 4 && t64 <= 1024);
        return deep64 ? dispatch_dma<64, 64, 5>(a, stream) : dispatch_dma<64, 64, 3>(a, stream);
    }
    if (dtype == BLT_BF16) return dispatch_tile<bf16>(a, c.bm, c.bn, splits, stream);
    return dispatch_tile<float>(a, c.bm, c.bn, splits, stream);
}

// autotune: time every candidate kernel for this descriptor on the real operands (idempotent launches only) and remember the best
static int autotune(int dtype, const GemmArgs& a, hipStream_t stream) {
    std::vector<Choice> cands;
    const bool nt = !a.transA && !a.transB;
    const int tiles[3][2] = {{64, 64}, {128, 64}, {128, 128}};
    for (int t = 0; t < 3; ++t) {
        if (dtype == BLT_BF16 && nt) cands.push_back({tiles[t][0], tiles[t][1], 0});
        cands.push_back({tiles[t][0], tiles[t][1], 1});
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return BLT_ERR_HIP;
    float best = 1e30f;
    Choice bc = heuristic(a);
    for (const Choice& c : cands) {
        int rc = run_choice(dtype, a, c, 1, stream);      // warm-up (also sets function attributes)
        if (rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
        (void)hipEventRecord(e0, stream);
        for (int r = 0; r < 3; ++r) run_choice(dtype, a, c, 1, stream);
        (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return BLT_ERR_HIP; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; bc = c; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    g_tuned[tune_key(a, dtype)] = bc;
    return BLT_OK;
}

// operand validation shared by every launch path: a bad call is BLT_ERR_ARG, not an out-of-bounds LDS-DMA read on the device
int blt_gemm_validate(int dtype, const GemmArgs& a) {
    BLT_REQUIRE(dtype == BLT_F32 || dtype == BLT_BF16, "gemm: bad dtype %d", dtype);
    BLT_REQUIRE(a.A && a.B && a.C, "gemm: null operand");
    BLT_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    const int ce = (dtype == BLT_BF16) ? 8 : 4;
    BLT_REQUIRE(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.B % 16) == 0 && ((uintptr_t)a.C % 16) == 0,
                "gemm: operands must be 16-byte aligned");
    BLT_REQUIRE((a.is_conv == 2 || a.lda % ce == 0) && a.ldb % ce == 0, "gemm: lda=%d / ldb=%d must be multiples of %d elements", a.lda, a.ldb, ce);
    if (a.is_conv == 1) {
        BLT_REQUIRE(!a.transA && !a.transB, "gemm: conv loader is NT only");
        BLT_REQUIRE((1 << a.cg.cin_log2) == a.cg.Cin && a.cg.Cin % ce == 0, "gemm: conv Cin=%d must be a power of two >= %d", a.cg.Cin, ce);
        BLT_REQUIRE(a.K == a.cg.KH * a.cg.KW * a.cg.Cin, "gemm: conv K mismatch");
        BLT_REQUIRE(a.M % (a.cg.Ho * a.cg.Wo) == 0, "gemm: conv M must be N*Ho*Wo");
    } else if (a.is_conv == 2) {
        BLT_REQUIRE(!a.transA && !a.transB, "gemm: stem loader is NT only");
        BLT_REQUIRE(a.K == 224 && a.cg.Cin == 4 && a.cg.stride == 2 && (a.cg.Wi % 2) == 0, "gemm: stem loader needs K=7*8*4, stride 2, even padded width");
        BLT_REQUIRE(a.cg.Hi >= 2 * (a.cg.Ho - 1) + 7 && a.cg.Wi >= 2 * (a.cg.Wo - 1) + 8, "gemm: stem image border too small");
        BLT_REQUIRE(a.M % (a.cg.Ho * a.cg.Wo) == 0, "gemm: conv M must be N*Ho*Wo");
    } else {
        if (!a.transA) BLT_REQUIRE(a.lda >= ((a.K + ce - 1) / ce) * ce, "gemm: lda=%d too small for K=%d", a.lda, a.K);
        else BLT_REQUIRE(a.lda >= a.M, "gemm: lda=%d too small for M=%d (transA)", a.lda, a.M);
    }
    if (!a.transB) BLT_REQUIRE(a.ldb >= ((a.K + ce - 1) / ce) * ce, "gemm: ldb=%d too small for K=%d", a.ldb, a.K);
    else BLT_REQUIRE(a.ldb >= a.N, "gemm: ldb=%d too small for N=%d (transB)", a.ldb, a.N);
    BLT_REQUIRE(a.ldc >= a.N, "gemm: ldc=%d < N=%d", a.ldc, a.N);
    BLT_REQUIRE(a.drop_p >= 0.f && a.drop_p < 1.f, "gemm: bad dropout p");
    BLT_REQUIRE((!a.R || ((uintptr_t)a.R % 16) == 0) && (!a.maskY || ((uintptr_t)a.maskY % 16) == 0) && (!a.C2 || ((uintptr_t)a.C2 % 16) == 0) &&
                (!a.rowtab || ((uintptr_t)a.rowtab % 16) == 0), "gemm: epilogue operands must be 16-byte aligned");
    BLT_REQUIRE(!(a.rowtab && !a.rowidx), "gemm: rowtab without rowidx");
    BLT_REQUIRE(!a.a_rowsum || (a.transA && !a.is_conv), "gemm: a_rowsum needs the transA (weight-gradient) form");
    return BLT_OK;
}

int blt_gemm(int dtype, const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    if (a.is_conv == 1) {      // plain NHWC defaults of the PP generalisation
        if (a.cg.in_rows == 0) a.cg.in_rows = a.cg.Hi;
        if (a.cg.in_pitch == 0) a.cg.in_pitch = a.cg.Wi;
        if (a.cg.Hov == 0) a.cg.Hov = a.cg.Ho;
        if (a.cg.Wov == 0) a.cg.Wov = a.cg.Wo;
    }
    { const int rc = blt_gemm_validate(dtype, a); if (rc != BLT_OK) return rc; }
#ifdef BLT_EXPERIMENTS
    if (a.lnA_out != nullptr) {
        BLT_REQUIRE(dtype == BLT_BF16 && !a.transA && !a.transB && !a.is_conv, "gemm: LayerNorm on A needs bf16 k-contiguous operands");
        BLT_REQUIRE(a.K <= 256 && a.K % 8 == 0 && a.lnA_gamma && a.lnA_beta && a.lnA_mean && a.lnA_rstd, "gemm: LayerNorm on A needs K <= 256, K %% 8 == 0 (K=%d)", a.K);
        BLT_REQUIRE(a.split_k == 0 && !a.stat_sum && !a.ln_out && ((uintptr_t)a.lnA_out % 16) == 0, "gemm: unsupported combination with LayerNorm on A");
        return launch_dma_lnA(a, stream);
    }
    if (a.ln_out != nullptr) {
        BLT_REQUIRE(dtype == BLT_BF16 && !a.transA && !a.transB && !a.is_conv, "gemm: the fused LayerNorm needs bf16 k-contiguous operands");
        BLT_REQUIRE(a.N <= 256 && a.N % 8 == 0 && a.ln_gamma && a.ln_beta && a.ln_mean && a.ln_rstd, "gemm: fused LayerNorm needs N <= 256, N %% 8 == 0 (N=%d)", a.N);
        BLT_REQUIRE(!a.out_f32 && !a.accumulate && !a.maskY && !a.rowtab && !a.stat_sum && a.split_k == 0, "gemm: epilogue term not supported with the fused LayerNorm");
        BLT_REQUIRE(a.ldc % 8 == 0 && (!a.R || a.ldr % 8 == 0) && (!a.C2 || a.ldc2 % 8 == 0) && (!a.bias || ((uintptr_t)a.bias % 16) == 0) &&
                    ((uintptr_t)a.ln_gamma % 16) == 0 && ((uintptr_t)a.ln_beta % 16) == 0 && ((uintptr_t)a.ln_out % 16) == 0, "gemm: fused LayerNorm operands must be 16-byte aligned");
        return launch_dma_ln(a, stream);
    }
#else
    BLT_REQUIRE(a.lnA_out == nullptr && a.ln_out == nullptr, "gemm: the LayerNorm-in-GEMM experiments are not part of this build (make experiments)");
#endif
    if ((!g_debug[8] || a.fold_s || a.out_stat) && !a.force_tile && !a.no_dma && blt_gemm_nt2_ok(dtype, a))
        return blt_gemm_nt2(a, stream, a.nt2_bm ? a.nt2_bm : g_debug[9], a.nt2_bm ? a.nt2_bn : g_debug[10]);
    BLT_REQUIRE(!a.fold_s && !a.out_stat, "gemm: the LayerNorm-fold / row-statistics epilogue needs the planned-tile bf16 NT kernel (bf16, k-contiguous "
                                          "operands with ld %% 8 == 0, no residual / mask / second output with fold)");
    const int splits = blt_gemm_splits(a, dtype);
    if (g_debug[2] && splits == 1 && !a.accumulate && !a.force_tile && g_tuned.find(tune_key(a, dtype)) == g_tuned.end()) {
        int rc = autotune(dtype, a, stream);
        if (rc) return rc;
    }
    return run_choice(dtype, a, choose(a, dtype), splits, stream);
}
