// MFMA GEMM for gfx950: linear layers (fwd / dgrad / wgrad) and the implicit-GEMM convolution of the frozen
// ResNet-18 stack share one LDS-tiled kernel.
//
//   * T = bf16: v_mfma_f32_16x16x32_bf16 (8 bf16 per lane per operand), fp32 accumulate.
//   * T = f32 : v_mfma_f32_16x16x4_f32 (exact fp32 fma chain) — the parity mode.
//   * 256 threads = 4 waves (2x2); block tile BMxBN in {128x128, 64x64}; K-tile = 128 bytes of K per row
//     (64 bf16 / 32 f32); double-buffered LDS, register-staged prefetch (global->VGPR issued before the
//     MFMAs of the current tile, VGPR->LDS after them), one barrier per K-tile.
//   * An operand keeps its GLOBAL orientation in LDS: k-contiguous operands are read with ds_read_b128
//     (bf16) / ds_read_b32 (f32); m/n-contiguous ("transposed") operands are read with ds_read_b64_tr_b16
//     (bf16, hardware transpose) / ds_read_b32 (f32).  No operand is ever transposed in HBM.
//   * The accumulators go through LDS once so that the epilogue (bias, position table, ReLU, dropout,
//     ReLU/dropout-backward mask, residual, accumulate, BN column statistics) runs on 8 consecutive columns per
//     thread and stores 16 bytes per lane.
//
// Replaces: torch.nn.Linear / F.conv2d call sites of the reference hot path
// (models/transformer_layers.py:453-456,489-491,530,400-408; models/encoder_cnn.py:20,33; models/iq.py:39,72-78).
#include "kernels.h"

namespace {

template <typename T> struct Frag;
template <> struct Frag<bf16> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef float type; };

__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

template <typename T, int BM, int BN, bool TA, bool TB, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int ES = sizeof(T);
    constexpr int CE = 16 / ES;          // elements per 16-byte chunk
    constexpr int BK = 128 / ES;         // K elements per tile (128 bytes per row)
    constexpr int KSTEP = (ES == 2) ? 32 : 4;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int SA = 144;              // row stride (bytes) of a k-contiguous LDS image: 128 + 16 pad
    constexpr int SAT = BM * ES + 16;    // row stride of an m-contiguous image [BK][BM]
    constexpr int SBT = BN * ES + 16;
    constexpr int A_BYTES = TA ? BK * SAT : BM * SA;
    constexpr int B_BYTES = TB ? BK * SBT : BN * SA;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int CH_A = BM * 8 / 256, CH_B = BN * 8 / 256;
    constexpr int CS = BN + 4;           // fp32 C-tile row stride (floats)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ Ag = (const T*)p.A;
    const T* __restrict__ Bg = (const T*)p.B;

    // ---- per-thread staging geometry ------------------------------------------------------------
    int cv_base[CH_A], cv_h0[CH_A], cv_w0[CH_A];
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < CH_A; ++i) {
            const int row = (tid + i * 256) >> 3;
            const int gm = m0 + row;
            const int hw = p.cg.Ho * p.cg.Wo;
            const int n = gm / hw, rem = gm - n * hw;
            const int ho = rem / p.cg.Wo, wo = rem - ho * p.cg.Wo;
            cv_base[i] = (gm < p.M) ? n * p.cg.Hi * p.cg.Wi : -1;
            cv_h0[i] = ho * p.cg.stride - p.cg.pad;
            cv_w0[i] = wo * p.cg.stride - p.cg.pad;
        }
    }
    uint4 ra[CH_A], rb[CH_B];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < CH_A; ++i) {
            const int c = tid + i * 256;
            if constexpr (CONV) {
                const int cc = c & 7;
                const int k = k0 + cc * CE;
                const int q = k >> p.cg.cin_log2, ci = k & (p.cg.Cin - 1);
                const int r = q / p.cg.KW, s = q - r * p.cg.KW;
                const int hi = cv_h0[i] + r, wi = cv_w0[i] + s;
                const bool ok = (cv_base[i] >= 0) && (k < p.K) && ((unsigned)hi < (unsigned)p.cg.Hi) &&
                                ((unsigned)wi < (unsigned)p.cg.Wi);
                const T* src = Ag + (((size_t)(cv_base[i] + hi * p.cg.Wi + wi)) << p.cg.cin_log2) + ci;
                ra[i] = ok ? *reinterpret_cast<const uint4*>(src) : zero4;
            } else if constexpr (!TA) {
                const int row = c >> 3, cc = c & 7;
                const int gm = m0 + row, gk = k0 + cc * CE;
                const bool ok = (gm < p.M) && (gk < p.K);
                ra[i] = ok ? *reinterpret_cast<const uint4*>(Ag + (size_t)gm * p.lda + gk) : zero4;
            } else {
                constexpr int CPR = BM * ES / 16;
                const int kr = c / CPR, cc = c % CPR;
                const int gk = k0 + kr, gm = m0 + cc * CE;
                const bool ok = (gk < p.K) && (gm < p.M);
                ra[i] = ok ? *reinterpret_cast<const uint4*>(Ag + (size_t)gk * p.lda + gm) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < CH_B; ++i) {
            const int c = tid + i * 256;
            if constexpr (!TB) {
                const int row = c >> 3, cc = c & 7;
                const int gn = n0 + row, gk = k0 + cc * CE;
                const bool ok = (gn < p.N) && (gk < p.K);
                rb[i] = ok ? *reinterpret_cast<const uint4*>(Bg + (size_t)gn * p.ldb + gk) : zero4;
            } else {
                constexpr int CPR = BN * ES / 16;
                const int kr = c / CPR, cc = c % CPR;
                const int gk = k0 + kr, gn = n0 + cc * CE;
                const bool ok = (gk < p.K) && (gn < p.N);
                rb[i] = ok ? *reinterpret_cast<const uint4*>(Bg + (size_t)gk * p.ldb + gn) : zero4;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        char* a_buf = smem + buf * STAGE;
        char* b_buf = a_buf + A_BYTES;
#pragma unroll
        for (int i = 0; i < CH_A; ++i) {
            const int c = tid + i * 256;
            if constexpr (!TA) {
                *reinterpret_cast<uint4*>(a_buf + (c >> 3) * SA + (c & 7) * 16) = ra[i];
            } else {
                constexpr int CPR = BM * ES / 16;
                *reinterpret_cast<uint4*>(a_buf + (c / CPR) * SAT + (c % CPR) * 16) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < CH_B; ++i) {
            const int c = tid + i * 256;
            if constexpr (!TB) {
                *reinterpret_cast<uint4*>(b_buf + (c >> 3) * SA + (c & 7) * 16) = rb[i];
            } else {
                constexpr int CPR = BN * ES / 16;
                *reinterpret_cast<uint4*>(b_buf + (c / CPR) * SBT + (c % CPR) * 16) = rb[i];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, lg = lane >> 4;
    auto compute = [&](int buf) {
        const char* a_buf = smem + buf * STAGE;
        const char* b_buf = a_buf + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / KSTEP; ++ks) {
            typename Frag<T>::type af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r0 = wm * WM + i * 16;
                if constexpr (ES == 2) {
                    if constexpr (!TA) {
                        af[i] = *reinterpret_cast<const bf16x8*>(a_buf + (r0 + l15) * SA + ks * 64 + lg * 16);
                    } else {
                        const char* q = a_buf + (ks * 32 + lg * 8 + (l15 >> 2)) * SAT + (r0 + (l15 & 3) * 4) * 2;
                        s16x4 lo = lds_tr16(q), hi = lds_tr16(q + 4 * SAT);
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        af[i] = __builtin_bit_cast(bf16x8, v);
                    }
                } else {
                    if constexpr (!TA) af[i] = *reinterpret_cast<const float*>(a_buf + (r0 + l15) * SA + (ks * 4 + lg) * 4);
                    else af[i] = *reinterpret_cast<const float*>(a_buf + (ks * 4 + lg) * SAT + (r0 + l15) * 4);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c0 = wn * WN + j * 16;
                if constexpr (ES == 2) {
                    if constexpr (!TB) {
                        bfr[j] = *reinterpret_cast<const bf16x8*>(b_buf + (c0 + l15) * SA + ks * 64 + lg * 16);
                    } else {
                        const char* q = b_buf + (ks * 32 + lg * 8 + (l15 >> 2)) * SBT + (c0 + (l15 & 3) * 4) * 2;
                        s16x4 lo = lds_tr16(q), hi = lds_tr16(q + 4 * SBT);
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        bfr[j] = __builtin_bit_cast(bf16x8, v);
                    }
                } else {
                    if constexpr (!TB) bfr[j] = *reinterpret_cast<const float*>(b_buf + (c0 + l15) * SA + (ks * 4 + lg) * 4);
                    else bfr[j] = *reinterpret_cast<const float*>(b_buf + (ks * 4 + lg) * SBT + (c0 + l15) * 4);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (ES == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // ---- main loop ------------------------------------------------------------------------------
    const int nk = (p.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        compute(kt & 1);
        if (kt + 1 < nk) store_tiles((kt + 1) & 1);
        __syncthreads();
    }

    // ---- accumulators -> LDS (fp32 C tile) ---------------------------------------------------------
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm * WM + i * 16 + lg * 4 + r) * CS + wn * WN + j * 16 + l15] = acc[i][j][r];
    __syncthreads();

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

    // ---- epilogue: 8 consecutive columns per thread --------------------------------------------------
    const uint32_t thresh = dropout_threshold(p.drop_p);
    const float keep_scale = (p.drop_p > 0.f) ? 1.f / (1.f - p.drop_p) : 1.f;
    const int drop_ld = (p.N + 7) & ~7;
    constexpr int GPR = BN / 8;
    for (int g = tid; g < BM * GPR; g += 256) {
        const int row = g / GPR, cgp = g % GPR;
        const int m = m0 + row, n = n0 + cgp * 8;
        if (m >= p.M || n >= p.N) continue;
        const int nv = (p.N - n < 8) ? (p.N - n) : 8;
        float v[8];
        {
            const float4 x0 = *reinterpret_cast<const float4*>(&Cs[row * CS + cgp * 8]);
            const float4 x1 = *reinterpret_cast<const float4*>(&Cs[row * CS + cgp * 8 + 4]);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        }
        if (p.alpha != 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        }
        if (p.bias != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) v[e] += p.bias[n + e];
        }
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
            const uint64_t e0 = (uint64_t)m * (uint64_t)drop_ld + (uint64_t)n;
            uint32_t w[8];
            dropout_words(p.seed, p.stream_id, e0 >> 2, w);
            dropout_words(p.seed, p.stream_id, (e0 >> 2) + 1, w + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (w[e] >= thresh) ? v[e] * keep_scale : 0.f;
        }
        const bool full = (nv == 8);
        if (p.maskY != nullptr) {
            const T* mp = (const T*)p.maskY + (size_t)m * p.ldm + n;
            float mv[8];
            if (full && (p.ldm % CE) == 0) Vec8<T>::load(mp, mv);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) mv[e] = (e < nv) ? to_f32(mp[e]) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (mv[e] != 0.f) ? v[e] * p.mask_scale : 0.f;
        }
        if (p.C2 != nullptr) {
            T* cp = (T*)p.C2 + (size_t)m * p.ldc2 + n;
            if (full && (p.ldc2 % CE) == 0) Vec8<T>::store(cp, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < nv) cp[e] = from_f32<T>(v[e]);
            }
        }
        if (p.R != nullptr) {
            const T* rp = (const T*)p.R + (size_t)m * p.ldr + n;
            float rv[8];
            if (full && (p.ldr % CE) == 0) Vec8<T>::load(rp, rv);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = (e < nv) ? to_f32(rp[e]) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (p.out_f32 || ES == 4) {
            float* cp = (float*)p.C + (size_t)m * p.ldc + n;
            const bool vec = full && (p.ldc % 4) == 0;
            if (p.accumulate) {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < nv) v[e] += cp[e];
            }
            if (vec) Vec8<float>::store(cp, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < nv) cp[e] = v[e];
            }
        } else {
            T* cp = (T*)p.C + (size_t)m * p.ldc + n;
            const bool vec = full && (p.ldc % CE) == 0;
            if (p.accumulate) {
                float ov[8];
                if (vec) Vec8<T>::load(cp, ov);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (e < nv) ? to_f32(cp[e]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += ov[e];
            }
            if (vec) Vec8<T>::store(cp, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < nv) cp[e] = from_f32<T>(v[e]);
            }
        }
    }
}

template <typename T, int BM, int BN, bool TA, bool TB, bool CONV>
int launch(const GemmArgs& a, hipStream_t stream) {
    constexpr int ES = sizeof(T);
    constexpr int BK = 128 / ES;
    constexpr int A_BYTES = TA ? BK * (BM * ES + 16) : BM * 144;
    constexpr int B_BYTES = TB ? BK * (BN * ES + 16) : BN * 144;
    constexpr int STAGE2 = 2 * (A_BYTES + B_BYTES);
    constexpr int CBYTES = BM * (BN + 4) * 4;
    constexpr int LDS = STAGE2 > CBYTES ? STAGE2 : CBYTES;
    static bool attr_set = false;
    auto kern = gemm_kernel<T, BM, BN, TA, TB, CONV>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            blt_set_error("gemm: hipFuncSetAttribute(%d) failed", LDS);
            return BLT_ERR_HIP;
        }
        attr_set = true;
    }
    const long tiles = (long)cdiv(a.M, BM) * cdiv(a.N, BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), LDS, stream, a);
    return blt_check_launch("gemm");
}

template <typename T, int BT>
int dispatch_layout(const GemmArgs& a, hipStream_t s) {
    if (a.is_conv) return launch<T, BT, BT, false, false, true>(a, s);
    if (!a.transA && !a.transB) return launch<T, BT, BT, false, false, false>(a, s);
    if (!a.transA && a.transB) return launch<T, BT, BT, false, true, false>(a, s);
    if (a.transA && a.transB) return launch<T, BT, BT, true, true, false>(a, s);
    return launch<T, BT, BT, true, false, false>(a, s);
}

}  // namespace

int blt_gemm_tile(const GemmArgs& a) {
    if (a.force_tile == 64 || a.force_tile == 128) return a.force_tile;
    const long t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    return (t128 >= 192) ? 128 : 64;
}

int blt_gemm_stat_rows(const GemmArgs& a) { return 2 * cdiv(a.M, blt_gemm_tile(a)); }

int blt_gemm(int dtype, const GemmArgs& a, hipStream_t stream) {
    BLT_REQUIRE(dtype == BLT_F32 || dtype == BLT_BF16, "gemm: bad dtype %d", dtype);
    BLT_REQUIRE(a.A && a.B && a.C, "gemm: null operand");
    BLT_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    const int ce = (dtype == BLT_BF16) ? 8 : 4;
    BLT_REQUIRE(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.B % 16) == 0 && ((uintptr_t)a.C % 16) == 0,
                "gemm: operands must be 16-byte aligned");
    BLT_REQUIRE(a.lda % ce == 0 && a.ldb % ce == 0, "gemm: lda=%d / ldb=%d must be multiples of %d elements", a.lda, a.ldb, ce);
    if (a.is_conv) {
        BLT_REQUIRE(!a.transA && !a.transB, "gemm: conv loader is NT only");
        BLT_REQUIRE((1 << a.cg.cin_log2) == a.cg.Cin && a.cg.Cin % ce == 0, "gemm: conv Cin=%d must be a power of two >= %d", a.cg.Cin, ce);
        BLT_REQUIRE(a.K == a.cg.KH * a.cg.KW * a.cg.Cin, "gemm: conv K mismatch");
        BLT_REQUIRE(a.M % (a.cg.Ho * a.cg.Wo) == 0, "gemm: conv M must be N*Ho*Wo");
    } else {
        if (!a.transA) BLT_REQUIRE(a.lda >= ((a.K + ce - 1) / ce) * ce, "gemm: lda=%d too small for K=%d", a.lda, a.K);
        else BLT_REQUIRE(a.lda >= a.M, "gemm: lda=%d too small for M=%d (transA)", a.lda, a.M);
    }
    if (!a.transB) BLT_REQUIRE(a.ldb >= ((a.K + ce - 1) / ce) * ce, "gemm: ldb=%d too small for K=%d", a.ldb, a.K);
    else BLT_REQUIRE(a.ldb >= a.N, "gemm: ldb=%d too small for N=%d (transB)", a.ldb, a.N);
    BLT_REQUIRE(a.ldc >= a.N, "gemm: ldc=%d < N=%d", a.ldc, a.N);
    BLT_REQUIRE(a.drop_p >= 0.f && a.drop_p < 1.f, "gemm: bad dropout p");
    BLT_REQUIRE(!(a.rowtab && !a.rowidx), "gemm: rowtab without rowidx");
    const int bt = blt_gemm_tile(a);
    if (dtype == BLT_BF16) return bt == 128 ? dispatch_layout<bf16, 128>(a, stream) : dispatch_layout<bf16, 64>(a, stream);
    return bt == 128 ? dispatch_layout<float, 128>(a, stream) : dispatch_layout<float, 64>(a, stream);
}
