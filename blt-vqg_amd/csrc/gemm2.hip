// Second-generation bf16 GEMMs for the transformer half of the step, built around what bounds these shapes on MI355X.
//
// The Linear layers of the stacks are [5120 | 5376 | 1280 rows] x [512 .. 2048] x [K = 512 .. 2048] problems: 2.7 - 10.7 GFLOP, i.e.
// 1 - 4 us at the MFMA peak.  What bounds them is not the matrix pipe but what one CU can take in from L2 into LDS (~70 GB/s per CU
// through the LDS-DMA path, MI355X_MICROARCH "ring-gemm"): a 128x128 tile needs 32 KB per 64-deep K-step for 2.1 MFLOP, so its CU is
// intake-bound at <= 45 % of its MFMA rate, and a grid of 640 such tiles is 1.25 rounds of 512 resident workgroups.  Round 1's
// kernels (gemm.hip) lost another 2-3x to that quantisation, to one K-tile in flight per workgroup and to a serial epilogue.  Here:
//
//   * gemm_nt2_kernel<BM, BN, ...>: C = epilogue(A[M,K] B[N,K]^T), both operands k-contiguous (Linear forward; input gradients through
//     the transposed weight shadow).  ONE workgroup of 8 waves per CU, and the tile shape is chosen PER PROBLEM so that the grid is
//     one round of <= 256 workgroups with the smallest (BM + BN) per flop (5120 x 2048 -> 160 x 256 tiles, 5120 x 512 -> 160 x 64,
//     5376 x 1536 -> 192 x 192, ...: blt_gemm_nt2_plan).  LDS-DMA ring of 3-4 stages with two or three K-steps (>= 80 KB) in flight,
//     counted vmcnt + one raw barrier per K-step; the epilogue's residual / mask rows are fetched BEFORE the K loop ends, the result
//     goes through LDS in 16-row slabs and leaves as 16-byte stores.
//   * wgrad_group_kernel: dW[N,K] = dY[M,N]^T X[M,K] for MANY Linear layers in one launch (a device-side problem table): the
//     weight gradients of a whole stack are ~1500 tiles of 128x128 with an M = 5120-deep contraction each — six full rounds of the
//     chip — where round 1 launched them one by one as 32-64 workgroups with split-K float atomics (16 MB of atomics per 512x512
//     gradient).  Token-major operands are DMA'd as they are and read with ds_read_b64_tr_b16; no split-K unless a launch is short
//     of tiles, results are stored (not added), bias gradients come from one extra MFMA against a ones fragment.
//
// Replaces torch.nn.Linear forward / backward call sites of the reference hot path (models/transformer_layers.py:453-456,489-491,530,
// 400-408; models/iq.py:39,72-78; models/decoder_transformer.py:19-20,40).
#include <type_traits>
#include <vector>
#include "kernels.h"

// LDS budget of one planned-tile workgroup: the whole CU (one workgroup per CU, as deep a ring as fits) unless an experiment build says
// otherwise (make coexist: 79 KB, so that a Linear workgroup fits beside one 80 KB convolution workgroup)
#ifndef NT2_LDS_CAP
#define NT2_LDS_CAP (160 * 1024)
#endif

namespace {

__device__ __attribute__((aligned(16))) uint4 g2_zero_page[1];

template <int N>
__device__ __forceinline__ void wait_vmcnt2() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dma16b(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}
// ds_read_b64_tr_b16 as inline asm: behind the builtin hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every group of
// transposed reads while LDS-DMA is in flight (it treats the DMA as a pending store the read may alias), which drains the whole ring
// every K-step.  The asm form is invisible to that pass: the data dependency is carried by lds_tr_wait() (every destination "+v",
// cdna_hip_programming.md 5.7 form (ii)) and the MFMAs are fenced below it with sched_barrier (rule 18).
__device__ __forceinline__ s16x4 lds_tr16b(const char* p) {
    s16x4 r;
    const unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(a));
    return r;
}
typedef __attribute__((ext_vector_type(8))) short s16x8b;

// ---------------------------------------------------------------------------------------------------------------
// NT GEMM
// ---------------------------------------------------------------------------------------------------------------
constexpr int cmin(int a, int b) { return a < b ? a : b; }
// CAP: LDS budget of one workgroup (the whole CU by default: one workgroup per CU with as deep a ring as fits; a half-CU budget lets TWO
// 4-wave workgroups share a CU, so that one's prologue / epilogue latency runs under the other's K loop)
template <int BM_, int BN_, int NWM_, int NWN_, int CAP_ = NT2_LDS_CAP>
struct Nt2 {
    static constexpr int BM = BM_, BN = BN_, NWM = NWM_, NWN = NWN_, CAP = CAP_;
    static constexpr int NW = NWM * NWN, NT = NW * 64, BK = 64;
    static constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 16, TN = WN / 16;
    static_assert(BM % (16 * NWM) == 0 && BN % (16 * NWN) == 0 && BM % 8 == 0 && BN % 8 == 0, "tile / wave layout mismatch");
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    // DMA wave-instructions per wave per stage: instruction j of wave w covers the 64 chunks (j * NW + w) * 64 ..; a wave whose 64
    // chunks lie beyond the tile still issues (vmcnt stays uniform) — from the zero page into its 1 KB slot of the dump area
    static constexpr int CH_A = (BM * 8 + NT - 1) / NT, CH_B = (BN * 8 + NT - 1) / NT, PER_STAGE = CH_A + CH_B;
    static constexpr int DUMP = 1024;          // ONE slot for every wave's beyond-the-tile DMA (zeros over zeros)
    // ring depth: as many stages as LDS holds (<= 8), so that NST - 1 K-steps (>= ~96 KB for the big tiles) are in flight per CU;
    // (AHEAD - 1) * PER_STAGE must fit the 6-bit vmcnt
    static constexpr int NST = cmin(cmin(8, (CAP - DUMP) / STAGE), 63 / PER_STAGE + 2);
    static constexpr int RING = NST * STAGE;
    static constexpr int AHEAD = NST - 1;
    static_assert(NST >= 2 && (AHEAD - 1) * PER_STAGE <= 63, "vmcnt is a 6-bit counter");
    // epilogue: the result leaves in TM passes of one 16-row block per wave row: slab = NWM*16 rows x BN fp32, double-buffered
    static constexpr int SLAB_ROWS = NWM * 16, CS = BN + 4, SLAB_BYTES = SLAB_ROWS * CS * 4;
    // a thread owns ONE group of 8 columns (tid % GPR; bias is loaded once) and every RPS-th slab row from tid / GPR on: IPP rows per pass
    static constexpr int GPR = BN / 8, RPS = NT / GPR, IPP = (SLAB_ROWS + RPS - 1) / RPS;
    // (+ BM x {mean, rstd} behind the slabs: the folded-LayerNorm variant keeps its row statistics there during the epilogue)
    static constexpr int ROWSTAT_OFF = 2 * SLAB_BYTES;
    static constexpr int LDS = (RING + DUMP > ROWSTAT_OFF + BM * 8) ? RING + DUMP : ROWSTAT_OFF + BM * 8;
    static_assert(LDS <= CAP, "LDS budget");
};

// sum over GPR (8, 16 or 32) consecutive, GPR-aligned lanes (all lanes of the wave take part)
template <int GPR>
__device__ __forceinline__ float rowgroup_sum(float v) {
    static_assert(GPR == 8 || GPR == 16 || GPR == 32, "lanes per group");
    v += dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                       // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);                      // row_half_mirror: the other quad of the 8
    if (GPR >= 16) v += dpp_mov<0x140>(v);       // row_mirror: the other half of the 16
    if (GPR >= 32) { float a, b; xor16_pair(v, a, b); v = a + b; }
    return v;
}

// FOLD: the consumer of a LayerNorm (GemmArgs::fold_*): A holds the RAW rows, B the gamma-scaled weight shadow W'; the epilogue finishes
// rstd_m * (acc - mean_m * s_n) + c_n.  Those launches (q|k|v, cross-attention query, first FFN layer) carry no bias / residual / mask /
// second output / row table, so the variant reuses their registers.
template <typename C, bool FOLD>
__device__ __forceinline__ void nt2_body(const GemmArgs& p, const int tile_m, const int tile_n) {
    constexpr int BM = C::BM, BN = C::BN, BK = C::BK, NW = C::NW, TM = C::TM, TN = C::TN, CS = C::CS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::NWN, wn = wave % C::NWN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bf16* __restrict__ Ag = (const bf16*)p.A;
    const bf16* __restrict__ Bg = (const bf16*)p.B;
    const void* zero = (const void*)g2_zero_page;
    char* dump = smem + C::RING;

    long a_off[C::CH_A], b_off[C::CH_B];
    int a_gk[C::CH_A], b_gk[C::CH_B];
    bool a_in[C::CH_A], b_in[C::CH_B];
#pragma unroll
    for (int j = 0; j < C::CH_A; ++j) {
        const int c = (j * NW + wave) * 64 + lane;
        const int row = c >> 3, gc = (c & 7) ^ (row & 7);
        a_in[j] = (j * NW + wave) * 64 < BM * 8;                  // wave-uniform
        a_gk[j] = gc * 8;
        a_off[j] = (a_in[j] && m0 + row < p.M) ? (long)(m0 + row) * p.lda + gc * 8 : -1;
    }
#pragma unroll
    for (int j = 0; j < C::CH_B; ++j) {
        const int c = (j * NW + wave) * 64 + lane;
        const int row = c >> 3, gc = (c & 7) ^ (row & 7);
        b_in[j] = (j * NW + wave) * 64 < BN * 8;
        b_gk[j] = gc * 8;
        b_off[j] = (b_in[j] && n0 + row < p.N) ? (long)(n0 + row) * p.ldb + gc * 8 : -1;
    }
    // DMA wave-instruction d (0 .. PER_STAGE-1: the A ones first) of K-step kt into ring stage `stage`; K-steps beyond the last one
    // are still "issued" — every lane from the zero page — so that the vmcnt arithmetic of the loop is the same in every iteration
    auto issue_one = [&](int d, int kt, int stage) {
        char* a_st = smem + stage * C::STAGE;
        char* b_st = a_st + C::A_BYTES;
        const int k0 = kt * BK;
        if (d < C::CH_A) {
            const int j = d;
            const bool ok = (a_off[j] >= 0) && (k0 + a_gk[j] < p.K);
            dma16b(ok ? (const void*)(Ag + a_off[j] + k0) : zero, a_in[j] ? a_st + (j * NW + wave) * 1024 : dump);
        } else {
            const int j = d - C::CH_A;
            const bool ok = (b_off[j] >= 0) && (k0 + b_gk[j] < p.K);
            dma16b(ok ? (const void*)(Bg + b_off[j] + k0) : zero, b_in[j] ? b_st + (j * NW + wave) * 1024 : dump);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lg = lane >> 4;
    const int nk = (p.K + BK - 1) / BK;
    constexpr int AHEAD = C::AHEAD, PS = C::PER_STAGE;

    // ---- epilogue operands that do not depend on the result (bias, residual rows, ReLU/dropout-backward mask rows) are fetched
    // before the K loop: item (pass pp, q) of this thread is slab row sr = tid / GPR + q * RPS (wave row sr / 16, row sr % 16 of its
    // pass-pp block) of its column group cg = tid % GPR ----
    constexpr int GPR = C::GPR, IPP = C::IPP, RPS = C::RPS;
    const int cg = tid % GPR, sr0 = tid / GPR;
    const int n = n0 + cg * 8;
    const bool t_on = sr0 < RPS && n < p.N;              // threads beyond RPS * GPR (BN = 192) and beyond the N tail sit the epilogue out
    const int nv = (p.N - n < 8) ? (p.N - n) : 8;
    uint4 rP[TM][IPP];                                   // residual rows, or (no residual) the mask rows; if both, the mask rows are read late
    const bool pre_ok = (p.ldr % 8 == 0) && (p.ldm % 8 == 0);
    const bool fast = t_on && (nv == 8) && pre_ok && (p.ldc % 8 == 0) && (!p.C2 || p.ldc2 % 8 == 0) && (!p.rowtab || p.ldt % 4 == 0) &&
                      (!p.bias || (((uintptr_t)(p.bias + n)) & 15) == 0);
    float bias[8], fsn[FOLD ? 8 : 1];
    float row_mu = 0.f, row_rs = 0.f;      // FOLD: thread t < BM owns the LayerNorm statistics of tile row t
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
    {
        const float* bp = FOLD ? p.fold_c : p.bias;      // (folded: c_n = sum_k beta_k W[n,k] + b_n takes the bias' place)
        if (bp && t_on) {
            if (fast) Vec8<float>::load(bp + n, bias);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < nv) bias[e] = bp[n + e];
            }
        }
    }
    if constexpr (FOLD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) fsn[e] = 0.f;
        if (t_on) {
            if (fast && (((uintptr_t)(p.fold_s + n)) & 15) == 0) Vec8<float>::load(p.fold_s + n, fsn);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < nv) fsn[e] = p.fold_s[n + e];
            }
        }
        // LayerNorm statistics of tile row `tid` (threads 0 .. BM-1) from the partial sums the producer of A left behind, added in slot
        // order (reproducible): mean = S1 / n, var = S2 / n - mean^2 (fp32; biased, as nn.LayerNorm).  They go to LDS behind the K loop;
        // the first column tile also stores them for the LayerNorm's backward.
        if (tid < BM && m0 + tid < p.M) {
            const int m = m0 + tid;
            const float2* sp = reinterpret_cast<const float2*>(p.fold_stat) + (size_t)m * (p.fold_sstride ? p.fold_sstride : p.stat_slots);
            float s1 = 0.f, s2 = 0.f;
            for (int k0 = 0; k0 < p.fold_np; k0 += 8) {      // eight loads in flight, then added in slot order
                float2 st[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) st[k] = (k0 + k < p.fold_np) ? sp[k0 + k] : make_float2(0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s1 += st[k].x; s2 += st[k].y; }
            }
            row_mu = s1 / p.fold_n;
            row_rs = rsqrtf(fmaxf(s2 / p.fold_n - row_mu * row_mu, 0.f) + p.fold_eps);
            if (tile_n == 0 && p.fold_mean != nullptr) { p.fold_mean[m] = row_mu; p.fold_rstd[m] = row_rs; }
        }
    }
    if (!FOLD && fast) {
#pragma unroll
        for (int pp = 0; pp < TM; ++pp)
#pragma unroll
            for (int q = 0; q < IPP; ++q) {
                const int sr = sr0 + q * RPS;
                const int m = m0 + (sr >> 4) * C::WM + pp * 16 + (sr & 15);
                const bool ok = sr < C::SLAB_ROWS && m < p.M;
                if (p.R && ok) rP[pp][q] = *reinterpret_cast<const uint4*>((const bf16*)p.R + (size_t)m * p.ldr + n);
                else if (p.maskY && ok) rP[pp][q] = *reinterpret_cast<const uint4*>((const bf16*)p.maskY + (size_t)m * p.ldm + n);
            }
    }

    // (the operand fetches above are the OLDEST entries of the in-order vmcnt queue: the counted waits below cover them)
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
#pragma unroll
        for (int d = 0; d < PS; ++d) issue_one(d, t, t);

    int st_cur = 0, st_nxt = AHEAD % C::NST;
    for (int kt = 0; kt < nk; ++kt) {
        // AHEAD - 1 K-steps are issued beyond kt (null ones past the end): this wave's part of K-step kt has landed
        wait_vmcnt2<(AHEAD - 1) * PS>();
        __builtin_amdgcn_s_barrier();                          // every wave's part of K-step kt landed; stage (kt + AHEAD) % NST is free
        const char* a_st = smem + st_cur * C::STAGE;
        const char* b_st = a_st + C::A_BYTES;
        // K-step kt + AHEAD is issued BETWEEN the MFMA groups of K-step kt (one DMA wave-instruction costs its wave ~100 issue cycles;
        // in front of the MFMAs they would be 0.3 us of every K-step during which the matrix pipe idles): 2 * TM slots per K-step
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bfr[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * C::WN + j * 16 + l15;
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
            bf16x8 a_cur, a_nxt;
            {
                const int r = wm * C::WM + l15;
                a_cur = *reinterpret_cast<const bf16x8*>(a_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (i + 1 < TM) {      // the next A fragment is in flight while this one's MFMAs run
                    const int r = wm * C::WM + (i + 1) * 16 + l15;
                    a_nxt = *reinterpret_cast<const bf16x8*>(a_st + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, bfr[j], acc[i][j], 0, 0, 0);
                constexpr int SLOTS = 2 * TM;
                const int slot = ks * TM + i;
#pragma unroll
                for (int d = 0; d < PS; ++d)
                    if (d * SLOTS / PS == slot) issue_one(d, kt + AHEAD, st_nxt);
                a_cur = a_nxt;
            }
        }
        st_cur = (st_cur + 1 == C::NST) ? 0 : st_cur + 1;
        st_nxt = (st_nxt + 1 == C::NST) ? 0 : st_nxt + 1;
    }
    __syncthreads();          // all MFMA reads of the ring are done (and every DMA has landed) before the slabs overwrite it
    float* s_rowstat = reinterpret_cast<float*>(smem + C::ROWSTAT_OFF);
    if constexpr (FOLD) {
        if (tid < BM) { s_rowstat[2 * tid] = row_mu; s_rowstat[2 * tid + 1] = row_rs; }      // (visible behind the first slab barrier below)
    }

    // ---- epilogue: same term order as gemm.hip::gemm_epilogue ----
    const uint32_t thresh = dropout_threshold(p.drop_p);
    const float keep_scale = (p.drop_p > 0.f) ? 1.f / (1.f - p.drop_p) : 1.f;
    const int drop_ld = (p.N + 7) & ~7;
#pragma unroll
    for (int pp = 0; pp < TM; ++pp) {
        float* Cs = reinterpret_cast<float*>(smem + (pp & 1) * C::SLAB_BYTES);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wm * 16 + lg * 4 + r) * CS + wn * C::WN + j * 16 + l15] = acc[pp][j][r];
        __syncthreads();      // slab pp complete; slab pp-1's readers are behind this barrier too, so pp+1 may overwrite it next pass
#pragma unroll
        for (int q = 0; q < IPP; ++q) {
            const int sr = sr0 + q * RPS;
            const int m = m0 + (sr >> 4) * C::WM + pp * 16 + (sr & 15);
            const bool act = t_on && sr < C::SLAB_ROWS && m < p.M;
            float st1 = 0.f, st2 = 0.f;
            if (act) {
            float v[8], t[8];
            {
                const float4 x0 = *reinterpret_cast<const float4*>(&Cs[sr * CS + cg * 8]);
                const float4 x1 = *reinterpret_cast<const float4*>(&Cs[sr * CS + cg * 8 + 4]);
                v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            }
            if constexpr (FOLD) {
                const float2 mr = *reinterpret_cast<const float2*>(&s_rowstat[2 * (m - m0)]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = mr.y * (v[e] - mr.x * fsn[e]) + bias[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias[e];
            }
            if (!FOLD && p.rowtab) {
                const float* tr = p.rowtab + (size_t)p.rowidx[m] * p.ldt + n;
                if (fast) {
                    Vec8<float>::load(tr, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e < nv) v[e] += tr[e];
                }
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
            if constexpr (!FOLD) {
            if (p.maskY) {
                if (fast) {
                    uint4 mraw = rP[pp][q];
                    if (p.R) mraw = *reinterpret_cast<const uint4*>((const bf16*)p.maskY + (size_t)m * p.ldm + n);
                    const bf16x8 mv = __builtin_bit_cast(bf16x8, mraw);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((float)mv[e] != 0.f) ? v[e] * p.mask_scale : 0.f;
                } else {
                    const bf16* mp = (const bf16*)p.maskY + (size_t)m * p.ldm + n;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e < nv) v[e] = ((float)mp[e] != 0.f) ? v[e] * p.mask_scale : 0.f;
                }
            }
            if (p.C2) {
                if (fast) Vec8<bf16>::store((bf16*)p.C2 + (size_t)m * p.ldc2 + n, v);
                else {
                    bf16* cp = (bf16*)p.C2 + (size_t)m * p.ldc2 + n;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e < nv) cp[e] = (bf16)v[e];
                }
            }
            if (p.R) {
                if (fast) {
                    const bf16x8 rv = __builtin_bit_cast(bf16x8, rP[pp][q]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                } else {
                    const bf16* rp = (const bf16*)p.R + (size_t)m * p.ldr + n;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e < nv) v[e] += (float)rp[e];
                }
            }
            }
            bf16* cp = (bf16*)p.C + (size_t)m * p.ldc + n;
            bf16x8 ov;
            if (fast) {
                if (!FOLD && p.accumulate) {
                    Vec8<bf16>::load(cp, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (bf16)v[e];
                *reinterpret_cast<bf16x8*>(cp) = ov;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ov[e] = (bf16)0.f;
                    if (e < nv) { ov[e] = (bf16)((!FOLD && p.accumulate) ? (float)cp[e] + v[e] : v[e]); cp[e] = ov[e]; }
                }
            }
            if (p.out_stat) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float x = (float)ov[e]; st1 += x; st2 += x * x; }      // (of the row AS STORED)
            }
            }
            // row statistics for the LayerNorm that reads this result and is folded into ITS consumer, one slot per 64 COLUMNS of the row
            // whatever the tile width: the 8 lanes that hold 64 consecutive columns add up (every lane of the wave takes part, inactive ones
            // with zeros; a fixed butterfly), the first of them (active whenever any of the 8 is) stores slot (first column / 64).  The
            // partition — and with it the consumer's in-order sum — does not depend on the tile shape or the row count of the launch:
            // incremental decoding (B rows per launch) reproduces the full pass (B * T rows) bit for bit
            if (p.out_stat) {
                st1 = rowgroup_sum<8>(st1); st2 = rowgroup_sum<8>(st2);
                if (act && (cg & 7) == 0)
                    *reinterpret_cast<float2*>(p.out_stat + 2 * ((size_t)m * p.stat_slots + tile_n * (BN / 64) + (cg >> 3))) = make_float2(st1, st2);
            }
        }
    }
}

template <typename C, bool FOLD>
__global__ __launch_bounds__(C::NT) void gemm_nt2_kernel(const GemmArgs p) {
    // consecutive workgroups walk down a column of tiles: they share the B (weight) panel, and A panels are re-read tiles_n times from L2
    const int tiles_m = (p.M + C::BM - 1) / C::BM;
    nt2_body<C, FOLD>(p, blockIdx.x % tiles_m, blockIdx.x / tiles_m);
}

#ifdef BLT_EXPERIMENTS
// Two problems in one launch (GemmPair: the encoder stack's and the posterior encoder stack's Linear of the same layer position — same
// N, K, tile shape and epilogue terms, their own operands, weights and row counts): a column of the launch holds problem 1's row tiles,
// then problem 2's.  One launch fills the CUs with both problems' tiles instead of two launches sharing them by time-slicing, and the
// smaller problem's partial last round disappears into the bigger one's.
template <typename C, bool FOLD>
__global__ __launch_bounds__(C::NT) void gemm_nt2_pair_kernel(const GemmArgs p1, const GemmPair d) {
    const int tiles_m = d.tiles1 + (d.M + C::BM - 1) / C::BM;
    int tile_m = blockIdx.x % tiles_m;
    const int tile_n = blockIdx.x / tiles_m;
    GemmArgs p = p1;
    if (tile_m >= d.tiles1) {      // (workgroup-uniform: scalar selects)
        tile_m -= d.tiles1;
        p.M = d.M; p.A = d.A; p.lda = d.lda; p.B = d.B; p.ldb = d.ldb; p.C = d.C; p.ldc = d.ldc;
        p.bias = d.bias; p.maskY = d.maskY; p.ldm = d.ldm; p.C2 = d.C2; p.ldc2 = d.ldc2; p.R = d.R; p.ldr = d.ldr;
        p.stream_id = d.stream_id; p.fold_s = d.fold_s; p.fold_c = d.fold_c; p.fold_stat = d.fold_stat; p.fold_mean = d.fold_mean;
        p.fold_rstd = d.fold_rstd; p.out_stat = d.out_stat;
    }
    nt2_body<C, FOLD>(p, tile_m, tile_n);
}

template <typename C, bool FOLD>
int launch_nt2_pair_(const GemmArgs& a, const GemmPair& d, hipStream_t s) {
    static BltDevFlag attr_set;
    auto kern = gemm_nt2_pair_kernel<C, FOLD>;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) {
            blt_set_error("gemm_nt2_pair: hipFuncSetAttribute(%d) failed", C::LDS);
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    GemmPair dd = d;
    dd.tiles1 = cdiv(a.M, C::BM);
    const long tiles = (long)(dd.tiles1 + cdiv(d.M, C::BM)) * cdiv(a.N, C::BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(C::NT), C::LDS, s, a, dd);
    return blt_check_launch("gemm_nt2_pair");
}
template <typename C>
int launch_nt2_pair(const GemmArgs& a, const GemmPair& d, hipStream_t s) {
    return a.fold_s ? launch_nt2_pair_<C, true>(a, d, s) : launch_nt2_pair_<C, false>(a, d, s);
}
#endif

template <typename C, bool FOLD>
int launch_nt2_(const GemmArgs& a, hipStream_t s) {
    static BltDevFlag attr_set;
    auto kern = gemm_nt2_kernel<C, FOLD>;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) {
            blt_set_error("gemm_nt2: hipFuncSetAttribute(%d) failed", C::LDS);
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    const long tiles = (long)cdiv(a.M, C::BM) * cdiv(a.N, C::BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(C::NT), C::LDS, s, a);
    return blt_check_launch("gemm_nt2");
}
template <typename C>
int launch_nt2(const GemmArgs& a, hipStream_t s) {
    return a.fold_s ? launch_nt2_<C, true>(a, s) : launch_nt2_<C, false>(a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Grouped weight gradients: dW_p[Nw, Kw] = dY_p[Mtok, Nw]^T X_p[Mtok, Kw] (+ db_p[Nw] = column sums of dY_p) for a table of problems.
// 128 x 128 output tiles, 64 tokens per K-step, 8 waves as 2 x 4 (wave tile 64 x 32).  Both operands are token-major, i.e. the
// contraction index is the ROW index in memory: the tiles are DMA'd as they are ([64 tokens][128 columns] bf16 = 256-byte rows, the
// 16-byte chunks of a row XOR-swizzled by f(row) on the SOURCE side) and the MFMA operands come out of LDS through
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10, image (b): conflict-free for the 16x16x32 operand).
// ---------------------------------------------------------------------------------------------------------------
// 16 waves (4 x 4, wave tile 32 x 32): four waves per SIMD.  The per-wave chain of a K-step (transposed reads -> lgkmcnt wait -> MFMAs,
// twice) is latency-bound, so it is the number of waves interleaving on a SIMD that keeps the matrix pipe and the DMA queue fed: with 8
// waves (2 per SIMD, wave tile 64 x 32) a K-step took 1.27 us against 0.47 us of LDS-DMA intake (PMC: 40 % of wave cycles in waits).
// Tile BM x 128 with BM = 256 (wave tile 64 x 32; 48 KB per K-step for twice the MFMA work of a 128 x 128 tile: 25 % less intake per flop
// and the fixed cost of a K-step — barrier, waits — spread over twice the flops) or 128 (narrow gradients).
template <int BM_, int BN_ = 128, int NWM_ = 4, int NWN_ = 4, int BK_ = 64>
struct WgCfgT {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, NWM = NWM_, NWN = NWN_, NW = NWM * NWN, NT = NW * 64;
    static constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 16, TN = WN / 16;
    static constexpr int A_ROW = BM * 2, B_ROW = BN * 2, A_BYTES = BK * A_ROW, B_BYTES = BK * B_ROW, STAGE = A_BYTES + B_BYTES;      // 32 / 48 / 64 KB
    static constexpr int CH_A = BK * (BM / 8) / NT, CH_B = BK * (BN / 8) / NT, PER_STAGE = CH_A + CH_B;      // DMA wave-instructions per wave per stage
    static constexpr int NST = (BM == 256 && BN == 256) ? (BK == 32 ? 4 : 2) : (BM == 256) ? 3 : 4, AHEAD = NST - 1;
    static constexpr int LDS = NST * STAGE + 1024;       // + the table's workgroup offsets
    static_assert(CH_A >= 1 && CH_B >= 1 && ((TM == 2 && TN == 2) || (TM == 4 && TN == 2) || (TM == 4 && TN == 8) || (TM == 4 && TN == 4)) && LDS <= 160 * 1024, "wave layout");
};
// The 256 x 256 form (8 waves as 4 x 2, wave tile 64 x 128, four 32 KB stages of 32 tokens): the 256 x 128 form measured 363-405 us for the
// decoder's set with 331 us of it explained by L2->LDS intake alone (48 KB per K-step per CU at ~70 GB/s; ablation: without the DMA
// -106 us, without the transposed reads -127 us, without the MFMAs -88 us, empty skeleton 113 us).  A square tile stages 64 KB per 64
// tokens for TWICE the flops (intake per flop -33 %) and reads 0.75 instead of 1.5 fragments per MFMA — but its 128 accumulator registers
// per lane allow only 8 waves per CU, and with two waves per SIMD the read -> wait -> MFMA phases of a wave are no longer covered by its
// neighbours: 361 us, the same.  Kept as a tested option (tile form 512), not selected.
typedef WgCfgT<256, 256, 4, 2, 32> WgCfgSq;      // 32-token K-steps: four 32 KB stages, three in flight
// the square tile with SIXTEEN waves (4 x 4, wave tile 64 x 64: 64 accumulator registers per lane, 0.5 fragments per MFMA) — round 4: what
// the 8-wave form lacked was waves to cover its read -> wait -> MFMA chain; tile form 1024 (debug key 13)
typedef WgCfgT<256, 256, 4, 4, 32> WgCfgSq16;
typedef WgCfgT<128> WgCfg;

__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <typename C>
__global__ __launch_bounds__(C::NT) void wgrad_group_kernel(const blt_wg_problem* __restrict__ probs, const int* __restrict__ wg0, int nprob) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::NWN, wn = wave % C::NWN;
    // ---- which problem / tile / K-slice is this workgroup?  wg0[p] = first workgroup of problem p (wg0[nprob] = grid size) ----
    int* s_wg0 = reinterpret_cast<int*>(smem + C::NST * C::STAGE);
    if (tid <= nprob) s_wg0[tid] = wg0[tid];
    __syncthreads();
    // XCD-aware order (cdna_hip_programming.md T1, bijective form): workgroups are dealt round-robin over the 8 XCDs, each with its own
    // L2; the remap gives every XCD a CONTIGUOUS run of the launch's work list, so that the tiles that stream the same dY / X panel (1.3 MB
    // each at 5 120 tokens) run side by side on one L2 instead of every tile pulling its own copy through the Infinity Cache
    int vid;
    {
        const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int pi = 0;
    {
        int lo = 0, hi = nprob;            // largest p with wg0[p] <= vid
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_wg0[mid] <= vid) lo = mid; else hi = mid; }
        pi = __builtin_amdgcn_readfirstlane(lo);
    }
    const blt_wg_problem P = probs[pi];
    const int local = vid - s_wg0[pi];
    const int split = local % P.splits, tile = local / P.splits;
    const int tile_m = tile / P.tiles_n, tile_n = tile % P.tiles_n;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    const int nk_all = (P.Mtok + C::BK - 1) / C::BK;
    const int per = (nk_all + P.splits - 1) / P.splits;
    const int kt0 = split * per;
    const int nk = (kt0 + per < nk_all ? kt0 + per : nk_all) - kt0;
    if (nk <= 0) return;                   // (uniform) an empty K-slice: nothing to add
    const bf16* __restrict__ Ag = (const bf16*)P.A;
    const bf16* __restrict__ Bg = (const bf16*)P.B;
    const void* zero = (const void*)g2_zero_page;
    const int a_cols = (P.Nw + 7) & ~7, b_cols = (P.Kw + 7) & ~7;      // loadable columns (lda / ldb cover the rounded width)

    // DMA instruction j of this wave covers chunks c = (j*8 + wave)*64 + lane of a [64][16-chunk] tile: token row c >> 4, LDS slot
    // c & 15 <- source chunk (c & 15) ^ f(row)
    // (A rows are BM * 2 bytes: BM / 8 chunks, swizzled inside each 256-byte half)
    int a_row[C::CH_A], b_row[C::CH_B];
    int a_off[C::CH_A], b_off[C::CH_B];          // element offsets inside one 64-token K-step (< 64 * ld + width: far below 2^31)
    constexpr int ACH = C::BM / 8;          // chunks per A row (16 or 32)
#pragma unroll
    for (int j = 0; j < C::CH_A; ++j) {
        const int c = (j * C::NW + wave) * 64 + lane;
        const int row = c / ACH, slot = c % ACH, gc = (slot & ~15) | ((slot & 15) ^ wg_swz(row));
        a_row[j] = row;
        a_off[j] = (m0 + gc * 8 < a_cols) ? row * P.lda + m0 + gc * 8 : -1;
    }
    constexpr int BCH = C::BN / 8;          // chunks per B row (16 or 32)
#pragma unroll
    for (int j = 0; j < C::CH_B; ++j) {
        const int c = (j * C::NW + wave) * 64 + lane;
        const int row = c / BCH, slot = c % BCH, gc = (slot & ~15) | ((slot & 15) ^ wg_swz(row));
        b_row[j] = row;
        b_off[j] = (n0 + gc * 8 < b_cols) ? row * P.ldb + n0 + gc * 8 : -1;
    }
    auto issue_one = [&](int d, int kt, int stage) {      // kt relative to kt0; K-steps beyond the slice come from the zero page
        char* a_st = smem + stage * C::STAGE;
        char* b_st = a_st + C::A_BYTES;
        const long t0 = (long)(kt0 + kt) * C::BK;
        if (d < C::CH_A) {
            const int j = d;
            const bool ok = kt < nk && a_off[j] >= 0 && t0 + a_row[j] < P.Mtok;
            dma16b(ok ? (const void*)(Ag + t0 * P.lda + a_off[j]) : zero, a_st + (j * C::NW + wave) * 1024);
        } else {
            const int j = d - C::CH_A;
            const bool ok = kt < nk && b_off[j] >= 0 && t0 + b_row[j] < P.Mtok;
            dma16b(ok ? (const void*)(Bg + t0 * P.ldb + b_off[j]) : zero, b_st + (j * C::NW + wave) * 1024);
        }
    };

    f32x4 acc[C::TM][C::TN], accb[C::TM];
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = P.bias != nullptr && tile_n == 0 && wn == 0;          // wave-uniform
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    const int l15 = lane & 15, lg = lane >> 4;
    // transposed-read addressing: lane 4q+p of a 16-lane group reads token row (8*lg + q [+4]) and the 4 columns 4p.. of its block
    const int tq = l15 >> 2, tp = l15 & 3;

#pragma unroll
    for (int t = 0; t < C::AHEAD; ++t)
#pragma unroll
        for (int d = 0; d < C::PER_STAGE; ++d) issue_one(d, t, t);

    // the K loop exists twice — with and without the bias-gradient MFMAs — selected by a wave-uniform branch OUTSIDE the loop: as a
    // condition inside it, every MFMA group carried a set of accumulator copies (phi nodes) and the loop was VALU-bound on v_mov
    auto k_loop = [&](auto with_bias) {
    int st_cur = 0, st_nxt = C::AHEAD % C::NST;
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt2<(C::AHEAD - 1) * C::PER_STAGE>();
            __builtin_amdgcn_s_barrier();
            const char* a_st = smem + st_cur * C::STAGE;
            const char* b_st = a_st + C::A_BYTES;
#pragma unroll
            for (int ks = 0; ks < C::BK / 32; ++ks) {
                const int r0 = ks * 32 + lg * 8 + tq;            // token row of the first transposed read (the second: + 4)
                const int f0 = wg_swz(r0), f1 = wg_swz(r0 + 4);
                // B fragments in groups of JG (all of them for TN = 2; two groups of 4 for the square tile, whose 8 + 4 fragments and
                // 128 accumulator registers would not fit otherwise); the A fragments are read once with the first group
                // (the 16-wave square tile, TM = TN = 4 under a 128-register budget: B fragments two at a time)
                constexpr int JG = (C::TN == 4 && C::TM == 4) ? 2 : (C::TN > 4 ? 4 : C::TN), NJG = C::TN / JG;
                bf16x8 af[C::TM];
                s16x4 alo[C::TM], ahi[C::TM];
#pragma unroll
                for (int jg = 0; jg < NJG; ++jg) {
                    bf16x8 bfr[JG];
                    s16x4 blo[JG], bhi[JG];
#pragma unroll
                    for (int j = 0; j < JG; ++j) {
                        const int cb = wn * C::WN + (jg * JG + j) * 16 + tp * 4, hb = (cb >> 7) * 256, cl = cb & 127;      // first of the lane's 4 columns
                        blo[j] = lds_tr16b(b_st + r0 * C::B_ROW + hb + (((cl >> 3) ^ f0) << 4) + (cl & 4) * 2);
                        bhi[j] = lds_tr16b(b_st + (r0 + 4) * C::B_ROW + hb + (((cl >> 3) ^ f1) << 4) + (cl & 4) * 2);
                    }
                    if (jg == 0) {
#pragma unroll
                        for (int i = 0; i < C::TM; ++i) {
                            const int cb = wm * C::WM + i * 16 + tp * 4, hb = (cb >> 7) * 256, cl = cb & 127;      // 256-byte half of the row, column inside it
                            alo[i] = lds_tr16b(a_st + r0 * C::A_ROW + hb + (((cl >> 3) ^ f0) << 4) + (cl & 4) * 2);
                            ahi[i] = lds_tr16b(a_st + (r0 + 4) * C::A_ROW + hb + (((cl >> 3) ^ f1) << 4) + (cl & 4) * 2);
                        }
                    }
                    if constexpr (JG == 4) {
                        if (jg == 0)
                            asm volatile("s_waitcnt lgkmcnt(0)"
                                         : "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]), "+v"(blo[3]), "+v"(bhi[3]),
                                           "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3])
                                         :: "memory");
                        else
                            asm volatile("s_waitcnt lgkmcnt(0)"
                                         : "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]), "+v"(blo[3]), "+v"(bhi[3])
                                         :: "memory");
                    } else if constexpr (C::TM == 4) {
                        if (jg == 0)
                            asm volatile("s_waitcnt lgkmcnt(0)"
                                         : "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]),
                                           "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3])
                                         :: "memory");
                        else
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]) :: "memory");
                    } else {
                        static_assert(C::TM == 2 || C::TM == 4, "lds_tr wait lists");
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1])
                                     :: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < JG; ++j) bfr[j] = __builtin_bit_cast(bf16x8, (s16x8b)__builtin_shufflevector(blo[j], bhi[j], 0, 1, 2, 3, 4, 5, 6, 7));
                    if (jg == 0) {
#pragma unroll
                        for (int i = 0; i < C::TM; ++i) af[i] = __builtin_bit_cast(bf16x8, (s16x8b)__builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7));
                    }
#pragma unroll
                    for (int i = 0; i < C::TM; ++i) {
#pragma unroll
                        for (int j = 0; j < JG; ++j) acc[i][jg * JG + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][jg * JG + j], 0, 0, 0);
                        if constexpr (decltype(with_bias)::value) { if (jg == 0) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0); }
                        // K-step kt + AHEAD goes out between the MFMA groups: PER_STAGE DMA wave-instructions over the 2 * TM * NJG slots of a K-step
                        constexpr int SLOTS = (C::BK / 32) * C::TM * NJG;
                        const int slot = (ks * NJG + jg) * C::TM + i;
#pragma unroll
                        for (int d = 0; d < C::PER_STAGE; ++d)
                            if (d * SLOTS / C::PER_STAGE == slot) issue_one(d, kt + C::AHEAD, st_nxt);
                    }
                }
            }
            st_cur = (st_cur + 1 == C::NST) ? 0 : st_cur + 1;
            st_nxt = (st_nxt + 1 == C::NST) ? 0 : st_nxt + 1;
        }
    };
    if (do_bias) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    wait_vmcnt2<0>();          // the null K-steps issued past the end must have landed before the workgroup retires its LDS

    // ---- results: fp32, stored (one K-slice) or added (split-K: atomics into the zeroed gradient buffer) ----
    float* Cg = P.C;
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * C::WM + i * 16 + lg * 4 + r, n = n0 + wn * C::WN + j * 16 + l15;
                if (m < P.Nw && n < P.Kw) {
                    float* dst = Cg + (size_t)m * P.ldc + n;
                    if (P.splits > 1 || P.accumulate) atomicAdd(dst, acc[i][j][r]);
                    else *dst = acc[i][j][r];
                }
            }
    if (do_bias && l15 == 0) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * C::WM + i * 16 + lg * 4 + r;
                if (m < P.Nw) {
                    if (P.splits > 1 || P.accumulate) atomicAdd(P.bias + m, accb[i][r]);
                    else P.bias[m] = accb[i][r];
                }
            }
    }
}

// tile shapes compiled in: BM x BN, 8 waves as NWM x NWN, ring depth
#ifdef BLT_EXPERIMENTS
struct TileOpt { int bm, bn; int (*launch)(const GemmArgs&, hipStream_t); int (*launch_pair)(const GemmArgs&, const GemmPair&, hipStream_t); };
#define NT2(BM, BN, NWM, NWN) {BM, BN, launch_nt2<Nt2<BM, BN, NWM, NWN>>, launch_nt2_pair<Nt2<BM, BN, NWM, NWN>>}
#else
struct TileOpt { int bm, bn; int (*launch)(const GemmArgs&, hipStream_t); };
#define NT2(BM, BN, NWM, NWN) {BM, BN, launch_nt2<Nt2<BM, BN, NWM, NWN>>}
#endif
const TileOpt kTiles[] = {
    NT2(64, 64, 2, 4),   NT2(64, 128, 2, 4),  NT2(128, 64, 4, 2),  NT2(128, 128, 2, 4),
    NT2(160, 64, 2, 4),  NT2(160, 128, 2, 4), NT2(192, 64, 4, 2),  NT2(96, 64, 2, 4),   NT2(32, 64, 2, 4),
#if NT2_LDS_CAP >= 160 * 1024      // (the co-residency experiment build caps a workgroup at 79 KB: two ring stages of <= 39 KB)
    NT2(160, 192, 2, 4), NT2(160, 256, 2, 4), NT2(192, 128, 2, 4), NT2(192, 192, 2, 4), NT2(192, 256, 2, 4),
    NT2(128, 256, 2, 4), NT2(256, 128, 4, 2), NT2(256, 64, 4, 2), NT2(224, 256, 2, 4),
#endif
    // (round 4: half-CU forms — 4 waves, <= 79 KB of LDS, two workgroups per CU so that one's prologue / epilogue runs under the other's
    // K loop: NT2H(80, 64, 1, 4) 10.3 us against 10.1 for (160, 64) on [5120 x 512] x 512, 24.2 against 20.7 at K = 2048, NT2H(80, 128)
    // 22.3 against 16.9 for (128, 256) on N = 1536 — measured, not kept)
};
#undef NT2
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

}  // namespace

// Cost model of one launch (microseconds, relative): rounds of <= 256 one-per-CU workgroups x (K-steps x max(intake, MFMA) + a fixed
// prologue / epilogue); intake = (BM + BN) * 128 B per K-step at ~70 GB/s per CU, MFMA = BM * BN * 64 * 2 flop at 9.8 TFLOP/s per CU.
static int g_plan_cus = 256;
void blt_set_plan_cus(int n) { g_plan_cus = (n > 0 && n <= 256) ? n : 256; }
int blt_plan_cus() { return g_plan_cus; }

int blt_gemm_nt2_plan(int M, int N, int K, int force_bm, int force_bn, bool row_stat, int M2 = 0) {
    int best = -1;
    double best_cost = 1e30;
    const int nk = cdiv(K, 64);
    const int cus = g_plan_cus;
    for (int i = 0; i < kNumTiles; ++i) {
        const TileOpt& t = kTiles[i];
        if (force_bm && (t.bm != force_bm || t.bn != force_bn)) continue;
        const long tiles = (long)(cdiv(M, t.bm) + (M2 > 0 ? cdiv(M2, t.bm) : 0)) * cdiv(N, t.bn);      // (M2: the second problem of a paired launch)
        const long rounds = (tiles + cus - 1) / cus;
        const double intake = (t.bm + t.bn) * 128.0 / 70e3;                   // us per K-step
        const double mfma = (double)t.bm * t.bn * 128.0 / 9.8e6;               // us per K-step
        const double epi = 1.0 + (double)t.bm * t.bn / 40960.0 * 1.0;         // slab passes
        // a partly filled last round still costs a full round; waste in padded tiles is priced through the tile count
        const double cost = rounds * (nk * (intake > mfma ? intake : mfma) + 2.0 + epi);
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    // measured corrections of the model (scratch/mb_rep.py, un-profiled back-to-back launches on MI355X): at N ~ 1536 the 128 x 256
    // tile (240-252 workgroups, 64 x 64 per wave) beats the 160/192-row tiles the model prefers by 10-15 %
    if (!force_bm && !row_stat && cus == 256 && M + M2 >= 4096 && N > 1024 && N < 2048)
        for (int i = 0; i < kNumTiles; ++i)
            if (kTiles[i].bm == 128 && kTiles[i].bn == 256) return i;
    return best;
}

bool blt_gemm_nt2_ok(int dtype, const GemmArgs& a, bool any_rows) {
    // (the LayerNorm-fold / row-statistics forms exist only here: they take this kernel at any row count; so do a forced tile and a pair)
    const bool fold_ok = !a.fold_s || (a.fold_c && a.fold_stat && a.fold_n > 0.f && !a.bias && a.alpha == 1.f && !a.R && !a.maskY && !a.C2 && !a.rowtab &&
                                       !a.accumulate);
    return dtype == BLT_BF16 && !a.transA && !a.transB && !a.is_conv && !a.out_f32 && a.split_k == 0 && !a.stat_sum && !a.ln_out && !a.lnA_out &&
           !a.a_rowsum && a.lda % 8 == 0 && a.ldb % 8 == 0 && fold_ok && (a.M >= 256 || a.fold_s || a.out_stat || a.nt2_bm > 0 || any_rows);
}

int blt_gemm_nt2(const GemmArgs& a, hipStream_t s, int force_bm, int force_bn) {
    const int i = blt_gemm_nt2_plan(a.M, a.N, a.K, force_bm, force_bn, a.out_stat != nullptr);
    BLT_REQUIRE(i >= 0, "gemm_nt2: no tile %dx%d compiled in", force_bm, force_bn);
    BLT_REQUIRE(!a.out_stat || cdiv(a.N, 64) <= a.stat_slots, "gemm_nt2: %d groups of 64 columns but only %d statistics slots per row", cdiv(a.N, 64),
                a.stat_slots);
    BLT_REQUIRE(!a.fold_s || (a.fold_np >= 1 && a.fold_np <= a.stat_slots), "gemm_nt2: fold_np %d outside [1, stat_slots %d]", a.fold_np, a.stat_slots);
    return kTiles[i].launch(a, s);
}
void blt_gemm_nt2_tile(int M, int N, int K, int* bm, int* bn, bool row_stat, int M2) {
    // the tile a launch WITHOUT a per-call shape takes: debug keys 9 / 10 (A/B: one tile shape for every planned-tile launch) apply here
    // as they do in blt_gemm's dispatch, so that a consumer's count of statistics slots follows the producer's actual column tiles
    const int i = blt_gemm_nt2_plan(M, N, K, blt_debug_get(9), blt_debug_get(9) ? blt_debug_get(10) : 0, row_stat, M2);
    if (i < 0) { *bm = 0; *bn = 0; return; }      // (no such tile: the launch itself fails loudly)
    *bm = kTiles[i].bm; *bn = kTiles[i].bn;
}

#ifdef BLT_EXPERIMENTS
// (experiments build: measured in the train step and not adopted — DESIGN.md 9; profiles/r04_pair_*)
// Paired launch: a and b are the same Linear position of two stacks — equal N, K and epilogue terms (checked), their own operands.
bool blt_gemm_nt2_pair_ok(int dtype, const GemmArgs& a, const GemmArgs& b) {
    auto same_null = [](const void* x, const void* y) { return (x == nullptr) == (y == nullptr); };
    return blt_gemm_nt2_ok(dtype, a, true) && blt_gemm_nt2_ok(dtype, b, true) && a.N == b.N && a.K == b.K && a.alpha == b.alpha && a.relu == b.relu &&
           a.drop_p == b.drop_p && a.seed == b.seed && a.mask_scale == b.mask_scale && a.accumulate == b.accumulate && !a.rowtab && !b.rowtab &&
           same_null(a.bias, b.bias) && same_null(a.maskY, b.maskY) && same_null(a.C2, b.C2) && same_null(a.R, b.R) && same_null(a.fold_s, b.fold_s) &&
           same_null(a.fold_mean, b.fold_mean) && same_null(a.out_stat, b.out_stat) && a.stat_slots == b.stat_slots && a.fold_np == b.fold_np &&
           a.fold_n == b.fold_n && a.fold_eps == b.fold_eps && a.fold_sstride == b.fold_sstride && a.nt2_bm == b.nt2_bm && a.nt2_bn == b.nt2_bn;
}
int blt_gemm_nt2_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t s) {
    const int i = blt_gemm_nt2_plan(a.M, a.N, a.K, a.nt2_bm, a.nt2_bn, a.out_stat != nullptr, b.M);
    BLT_REQUIRE(i >= 0, "gemm_nt2_pair: no tile %dx%d compiled in", a.nt2_bm, a.nt2_bn);
    BLT_REQUIRE(!a.out_stat || cdiv(a.N, 64) <= a.stat_slots, "gemm_nt2_pair: %d groups of 64 columns but only %d statistics slots per row",
                cdiv(a.N, 64), a.stat_slots);
    BLT_REQUIRE(!a.fold_s || (a.fold_np >= 1 && a.fold_np <= a.stat_slots), "gemm_nt2_pair: fold_np %d outside [1, stat_slots %d]", a.fold_np, a.stat_slots);
    GemmPair d;
    d.M = b.M; d.A = b.A; d.lda = b.lda; d.B = b.B; d.ldb = b.ldb; d.C = b.C; d.ldc = b.ldc; d.bias = b.bias; d.maskY = b.maskY; d.ldm = b.ldm;
    d.C2 = b.C2; d.ldc2 = b.ldc2; d.R = b.R; d.ldr = b.ldr; d.stream_id = b.stream_id; d.fold_s = b.fold_s; d.fold_c = b.fold_c;
    d.fold_stat = b.fold_stat; d.fold_mean = b.fold_mean; d.fold_rstd = b.fold_rstd; d.out_stat = b.out_stat;
    return kTiles[i].launch_pair(a, d, s);
}
#endif

// ---- grouped weight gradients ----------------------------------------------------------------------------------------------
bool blt_wgrad_group_ok(int dtype, const GemmArgs& a) {
    return dtype == BLT_BF16 && a.transA && a.transB && a.out_f32 && !a.is_conv && !a.bias && !a.relu && a.drop_p == 0.f && !a.maskY && !a.C2 &&
           !a.R && !a.rowtab && !a.stat_sum && a.alpha == 1.f && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.lda >= ((a.M + 7) & ~7) &&
           a.ldb >= ((a.N + 7) & ~7) && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.B % 16) == 0;
}
// host table -> (problems, wg0, tile rows); returns the number of workgroups.  256-row tiles when they still fill the chip; split-K only
// when the whole launch is short of tiles.
int blt_wgrad_group_plan(const std::vector<GemmArgs>& g, std::vector<blt_wg_problem>& probs, std::vector<int>& wg0, int* bm_out) {
    // tile form (returned through *bm_out): 128 = 128 x 128, 256 = 256 x 128, 512 = 256 x 256 (square) — the largest that still gives every
    // CU a tile; debug key 13 forces one
    long t256 = 0, tsq = 0;
    for (const GemmArgs& a : g) { t256 += (long)cdiv(a.M, 256) * cdiv(a.N, 128); tsq += (long)cdiv(a.M, 256) * cdiv(a.N, 256); }
    const int forced = blt_debug_get(13);
    // The square 256 x 256 form (8 waves) wherever it still gives every CU a tile.  Alone it measures the same as 256 x 128 (361 vs 363 us
    // for the decoder's set, round 2: its 8 waves do not cover their own read -> wait -> MFMA chain) — but in the STEP the grouped launches
    // run beside the encoder chains, and there half the workgroups and a third less L2 -> LDS intake per flop is what counts: 6.84 ms
    // against 6.93 per step (round 4, A/B by debug key 13 on one box).  (A 16-wave square form — 64 accumulator + fragment registers
    // over the 128-register budget, 21 spilled — 7.57 ms: tile form 1024, kept selectable for the record.)
    const int bm = (forced == 128 || forced == 256 || forced == 512 || forced == 1024) ? forced : (tsq >= 256 ? 512 : (t256 >= 256 ? 256 : 128));
    const int tm = bm == 128 ? 128 : 256, tn = bm >= 512 ? 256 : 128;
    long tiles = 0;
    for (const GemmArgs& a : g) tiles += (long)cdiv(a.M, tm) * cdiv(a.N, tn);
    probs.clear(); wg0.clear();
    int wg = 0;
    for (const GemmArgs& a : g) {
        blt_wg_problem p;
        p.A = a.A; p.B = a.B; p.C = (float*)a.C; p.bias = a.a_rowsum;
        p.Nw = a.M; p.Kw = a.N; p.Mtok = a.K; p.lda = a.lda; p.ldb = a.ldb; p.ldc = a.ldc;
        p.tiles_n = cdiv(a.N, tn);
        const int t = cdiv(a.M, tm) * p.tiles_n;
        const int nk = cdiv(a.K, 64);
        int s = 1;
        if (tiles < 192) {                 // fewer tiles than CUs in the whole launch: slice K (>= 4 K-steps per slice)
            s = (int)(256 / (tiles > 0 ? tiles : 1));
            if (s > nk / 4) s = nk / 4;
            if (s < 1) s = 1;
            if (s > 32) s = 32;
        }
        p.splits = s;
        p.accumulate = a.accumulate ? 1 : 0;
        p.pad = 0;
        probs.push_back(p);
        wg0.push_back(wg);
        wg += t * s;
    }
    wg0.push_back(wg);
    *bm_out = bm;
    return wg;
}
template <typename C>
static int launch_wg(const blt_wg_problem* probs_dev, const int* wg0_dev, int nprob, int nwg, hipStream_t s) {
    static BltDevFlag attr_set;
    auto kern = wgrad_group_kernel<C>;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) {
            blt_set_error("wgrad_group: hipFuncSetAttribute(%d) failed", C::LDS);
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(C::NT), C::LDS, s, probs_dev, wg0_dev, nprob);
    return blt_check_launch("wgrad_group");
}
int blt_wgrad_group_launch(const blt_wg_problem* probs_dev, const int* wg0_dev, int nprob, int nwg, int bm, hipStream_t s) {
    BLT_REQUIRE(probs_dev && wg0_dev && nprob > 0 && nprob < 250 && nwg > 0 && (bm == 128 || bm == 256 || bm == 512 || bm == 1024), "wgrad_group: bad table");
    if (bm == 512) return launch_wg<WgCfgSq>(probs_dev, wg0_dev, nprob, nwg, s);
    if (bm == 1024) return launch_wg<WgCfgSq16>(probs_dev, wg0_dev, nprob, nwg, s);
    return bm == 256 ? launch_wg<WgCfgT<256>>(probs_dev, wg0_dev, nprob, nwg, s) : launch_wg<WgCfgT<128>>(probs_dev, wg0_dev, nprob, nwg, s);
}
