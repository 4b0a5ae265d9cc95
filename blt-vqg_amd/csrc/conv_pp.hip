// 3x3 stride-1 pad-1 convolution on padded-pitch (PP, kernels.h) bf16 activations — the workhorse of the frozen ResNet-18 stack
// (models/encoder_cnn.py:17,33: torchvision BasicBlock conv1/conv2, 13 of the 20 convolutions, 85 % of the CNN's flops).
//
// Why not the implicit GEMM of gemm.hip: there every K-tile re-fetches its [128 pixels x 64 channels] A operand from L2 — nine
// times per input pixel, once per filter tap — and a CU takes in only ~70 GB/s through its L1 (measured: that, not the MFMA pipe,
// bounds those kernels at 0.4-0.65 PFLOP/s).  In the PP layout the nine tap windows of a run of 128 consecutive output pixels are
// nine SHIFTED VIEWS of one run of 128 + 2*(W+1) + 2 consecutive input pixels.  So per 64-channel slice a workgroup stages that
// patch in LDS ONCE (LDS-DMA, XOR-swizzled 128-byte pixel rows) and runs the nine taps' MFMAs off it, moving only the row origin of
// the A-fragment reads; only the weights (B) stream per tap through a 2-stage ring.  L2->LDS traffic per output tile drops from
// 9 x 16 KB to one 19-31 KB patch per slice (A) + unchanged B.
//
//   * output tile: 128 consecutive PP positions x BN channels (BN = 128, or 64 for Cout = 64); 4 waves as 2 x 2, MFMA 16x16x32 bf16
//   * the next slice's patch is fetched in pieces, one 1-KB DMA per wave per tap step, under the current slice's MFMAs
//   * one K-step (tap) in flight across the barrier; two workgroups per CU (<= 80 KB LDS each) cover each other's waits
//   * no bounds logic in the loader: the PP zero pixels are the padding, guards in front of / behind the buffer cover the tile
//     overhang; no divisions either (the validity of the 128 output rows is one table, used only to mask the BN statistics)
//   * epilogue: train-mode BatchNorm partial sums (per half tile = per wave row, straight from the fp32 accumulators), bf16 tile
//     staged through LDS for 16-byte coalesced stores.  Rows at PP pad positions receive meaningless values: they are masked out of
//     the statistics here and overwritten with zeros by bn_apply_pp (norm.hip), the only consumer.
#include "kernels.h"

namespace {

__device__ __attribute__((aligned(16))) uint4 g_zero16[1];

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

struct ConvPPArgs {
    const bf16* X;      // PP input, pixel 0 (guards around it)
    const bf16* Wt;     // weights [Cout][3][3][Cin]
    bf16* Y;            // PP output, pixel 0
    long Mq;            // N * (H+1) * (W+1)
    int H, W, Cin, Cout;
    float *stat_sum, *stat_sq;
    int pw;             // patch DMA instructions per wave (patch = pw * 4 KB)
    int nblocks, tiles_n;
    int chunked;        // 1: consecutive tiles go to the same XCD (patch halos and tile rows shared in its L2)
    int valid_off;      // byte offset of the 128-float row-validity table (epilogue only: behind the staged output tile)
    // FUSE_IN: X is the RAW output of the previous convolution; its train-mode BatchNorm + ReLU (y = max(x*scale + shift, 0), zeros at
    // PP pad positions) is applied to each 64-channel patch slice in LDS after it has landed, instead of in a bn_apply_pp pass of its own
    const float *in_scale, *in_shift;
};

// NSTB = stages of the weight ring: 2 (one tap in flight; two workgroups per CU cover each other's waits) or 4 (three taps in flight, for
// grids that leave a workgroup alone on its CU, e.g. the 7x7x512 layer: 256 tiles)
// NWM = wave rows: 2 -> 128 positions per workgroup, 4 waves (two or three workgroups per CU); 4 -> 256 positions, 8 waves, ONE workgroup per
// CU with a 4-stage ring: with a single tap in flight every tap of the 128-position form waits a full L2->LDS round trip (1.1-1.4 us per tap
// measured against 0.2 us of MFMAs per wave), and LDS has no room for a deeper ring beside a second workgroup; the 256-position form keeps
// three taps (3 x 0.43 us of MFMA time per SIMD) in flight and halves the weight bytes per output position
template <int BN, int NSTB, bool FUSE_IN, int NWM>
__global__ __launch_bounds__(NWM * 128) void conv3x3_pp_kernel(const ConvPPArgs p) {
    constexpr int NW = NWM * 2, NT = NW * 64, BM = NWM * 64;
    constexpr int WN = BN / 2, TN = WN / 16, TM = 4, CH_B = BN * 8 / NT, B_STAGE = BN * 128, CSB = BN + 8, AHEAD = NSTB - 1;
    static_assert(BN * 8 % NT == 0, "weight tile chunks per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lg = lane >> 4;

    int b = blockIdx.x;
    if (p.chunked) b = (b & 7) * ((int)gridDim.x >> 3) + (b >> 3);     // hardware deals workgroup ids round-robin over the 8 XCDs
    if (b >= p.nblocks) return;
    const int tile_m = b / p.tiles_n, tile_n = b - tile_m * p.tiles_n;
    const long q0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;
    const int pitch = p.W + 1;
    const int NS = p.Cin >> 6, nsteps = 9 * NS;
    const int patch_bytes = p.pw * NW * 1024;
    char* const patch0 = smem;
    char* const patch1 = smem + (NS > 1 ? patch_bytes : 0);
    char* const ring = smem + (NS > 1 ? 2 : 1) * patch_bytes;
    float* const valid = reinterpret_cast<float*>(smem + p.valid_off);

    // patch row pp <-> input pixel q0 - pitch - 1 + pp; DMA instruction ii = j*NW + wave covers rows ii*8 .. ii*8+7, LDS slot
    // (lane & 7) of row pp holds global chunk (lane & 7) ^ (pp & 7)
    const bf16* const xb = p.X + (q0 - pitch - 1) * (long)p.Cin;
    auto issue_patch = [&](int sl, int j) {
        const int pp = (j * NW + wave) * 8 + (lane >> 3);
        const int off = pp * p.Cin + ((((lane & 7) ^ (pp & 7))) << 3) + (sl << 6);
        dma16(xb + off, ((sl & 1) ? patch1 : patch0) + (j * NW + wave) * 1024);
    };
    long b_off[CH_B];
#pragma unroll
    for (int j = 0; j < CH_B; ++j) {
        const int r = (j * NW + wave) * 8 + (lane >> 3);
        b_off[j] = (n0 + r < p.Cout) ? (long)(n0 + r) * (9 * p.Cin) + ((((lane & 7) ^ (r & 7))) << 3) : -1;
    }
    auto issue_b = [&](int s, int sl, int t) {
        const int k0 = t * p.Cin + (sl << 6);
        char* st = ring + (s % NSTB) * B_STAGE;
#pragma unroll
        for (int j = 0; j < CH_B; ++j) {
            const void* src = (b_off[j] >= 0) ? (const void*)(p.Wt + b_off[j] + k0) : (const void*)g_zero16;
            dma16(src, st + (j * NW + wave) * 1024);
        }
    };

    // FUSE_IN: this thread transforms chunk (j*NT + tid) of a patch slice, j < pw: patch row j*NW*8 + (tid >> 3), always the same 8
    // channels of the slice.  Which of its rows are real pixels (not PP pads, guards or past the end) is one bit each, computed once.
    unsigned vbits = 0;
    if constexpr (FUSE_IN) {
        for (int j = 0; j < p.pw; ++j) {
            const long q = q0 - pitch - 1 + j * (NW * 8) + (tid >> 3);
            if (q >= 0 && q < p.Mq) {
                const unsigned row = (unsigned)q / (unsigned)pitch;
                const int w = (int)((unsigned)q - row * (unsigned)pitch), h = (int)(row % (unsigned)(p.H + 1));
                if (w != p.W && h != p.H) vbits |= 1u << j;
            }
        }
    }
    // FUSE_IN staging: the piece goes global -> registers -> (BatchNorm + ReLU, pad mask) -> LDS, into the slot the DMA would have
    // filled; an in-place LDS pass after a DMA was measured at +26 us per launch (LDS bandwidth is what bounds this kernel)
    auto load_piece = [&](int sl_, int j) -> bf16x8 {
        const int pp = (j * NW + wave) * 8 + (lane >> 3);
        return *reinterpret_cast<const bf16x8*>(xb + (long)pp * p.Cin + ((((lane & 7) ^ (pp & 7))) << 3) + (sl_ << 6));
    };
    auto load_bn = [&](int sl_, float (&sc)[8], float (&sh)[8]) {
        const int cc = (tid & 7) ^ ((tid >> 3) & 7);
        Vec8<float>::load(p.in_scale + (sl_ << 6) + cc * 8, sc);
        Vec8<float>::load(p.in_shift + (sl_ << 6) + cc * 8, sh);
    };
    auto store_piece = [&](int sl_, int j, bf16x8 v, const float (&sc)[8], const float (&sh)[8]) {
        const bool ok = (vbits >> j) & 1u;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ok ? (bf16)fmaxf((float)v[e] * sc[e] + sh[e], 0.f) : (bf16)0.f;
        *reinterpret_cast<bf16x8*>(((sl_ & 1) ? patch1 : patch0) + (j * NT + tid) * 16) = v;
    };
    float nsc[8], nsh[8];      // FUSE_IN: scale / shift of this thread's 8 channels of the slice being staged
    bf16x8 staged;

    bf16x8 first[8];                                         // FUSE_IN: slice 0's pieces, pw <= 8 (P <= 256 patch pixels)
    if constexpr (FUSE_IN) {
        load_bn(0, nsc, nsh);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < p.pw) first[j] = load_piece(0, j);
    } else {
        for (int j = 0; j < p.pw; ++j) issue_patch(0, j);
    }
    // the tap steps run s = sl*9 + t; (sa, ta) walks AHEAD steps in front of (sl, t) for the weight ring
    int sa = 0, ta = 0;
    for (int s = 0; s < AHEAD && s < nsteps; ++s) {
        issue_b(s, sa, ta);
        if (++ta == 9) { ta = 0; ++sa; }
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (FUSE_IN) {
        // behind the weight ring's first DMAs in program order: waiting for the pieces leaves those in flight
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < p.pw) store_piece(0, j, first[j], nsc, nsh);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible to the other waves at the first barrier of the tap loop
    }

    int sl = 0, t = 0;
    for (int s = 0; s < nsteps; ++s) {
        // this wave's share of B(s) and of every older DMA has landed once only the younger taps' weight DMAs are outstanding (the
        // patch piece of an iteration is issued BEFORE its weight tile, so it is never younger than the tile that is waited for)
        const int younger = (nsteps - 1 - s < AHEAD - 1) ? nsteps - 1 - s : AHEAD - 1;
        if (AHEAD > 2 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CH_B) : "memory");
        else if (AHEAD > 1 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH_B) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // ... everyone's has; ring stage (s-1) % NSTB and the idle patch buffer are free
        if constexpr (FUSE_IN) {
            if (sl + 1 < NS) {
                if (t >= 1 && t <= p.pw) store_piece(sl + 1, t - 1, staged, nsc, nsh);      // the piece loaded one tap ago
                if (t == 0) load_bn(sl + 1, nsc, nsh);
                if (t < p.pw) staged = load_piece(sl + 1, t);
            }
        } else {
            if (sl + 1 < NS && t < p.pw) issue_patch(sl + 1, t);
        }
        if (s + AHEAD < nsteps) {
            issue_b(s + AHEAD, sa, ta);
            if (++ta == 9) { ta = 0; ++sa; }
        }
        const char* pa = (sl & 1) ? patch1 : patch0;
        const char* bs = ring + (s % NSTB) * B_STAGE;
        // (reading tap t+1's A fragments under tap t's MFMAs was measured: no change — with two workgroups per CU the partner's MFMAs
        // already cover these LDS reads)
        const int tr = (t >= 6) ? 2 : (t >= 3) ? 1 : 0;
        const int toff = tr * pitch + (t - 3 * tr);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pr = wm * 64 + i * 16 + l15 + toff;
                af[i] = *reinterpret_cast<const bf16x8*>(pa + pr * 128 + (((ks * 4 + lg) ^ (pr & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WN + j * 16 + l15;
                bfr[j] = *reinterpret_cast<const bf16x8*>(bs + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (++t == 9) { t = 0; ++sl; }
    }
    __syncthreads();          // every wave is done with the patch and the ring: the bf16 tile is staged over them

    // ---- BatchNorm partial statistics of the raw result: rows of wave row wm = half tile wm (encoder_cnn.py:33) ----
    if (p.stat_sum != nullptr) {
        // which of the 128 output positions are real pixels (not PP pads, not beyond the end): a table behind the staging area
        if (tid < BM) {
            const unsigned q = (unsigned)(q0 + tid);                  // < 2^31 positions (checked on the host)
            const unsigned row = q / (unsigned)pitch;
            const int w = (int)(q - row * (unsigned)pitch), h = (int)(row % (unsigned)(p.H + 1));
            valid[tid] = ((long)q < p.Mq && w != p.W && h != p.H) ? 1.f : 0.f;
        }
        __syncthreads();
        f32x4 vm[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) vm[i] = *reinterpret_cast<const f32x4*>(valid + wm * 64 + i * 16 + lg * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (vm[i][r] != 0.f) ? acc[i][j][r] : 0.f;
                    s1 += v;
                    s2 += v * v;
                }
            s1 = sum_across_rows(s1); s2 = sum_across_rows(s2);
            if (lg == 0) {
                const size_t o = (size_t)(tile_m * NWM + wm) * p.Cout + n0 + wn * WN + j * 16 + l15;
                p.stat_sum[o] = s1;
                p.stat_sq[o] = s2;
            }
        }
    }
    // ---- bf16 tile -> LDS -> 16-byte coalesced stores ----
    bf16* Cs = reinterpret_cast<bf16*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wm * 64 + i * 16 + lg * 4 + r) * CSB + wn * WN + j * 16 + l15] = (bf16)acc[i][j][r];
    __syncthreads();
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int c = tid; c < BM * CPR; c += NT) {
        const int row = c / CPR, ch = c - row * CPR;
        const long q = q0 + row;
        if (q < p.Mq)
            *reinterpret_cast<uint4*>(p.Y + q * p.Cout + n0 + ch * 8) = *reinterpret_cast<const uint4*>(Cs + row * CSB + ch * 8);
    }
}

template <int BN, int NSTB, bool FUSE_IN, int NWM = 2>
int launch(const ConvPPArgs& a, int grid, size_t lds, hipStream_t s) {
    static BltDevFlag attr_set;
    auto kern = conv3x3_pp_kernel<BN, NSTB, FUSE_IN, NWM>;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            blt_set_error("conv3x3_pp: hipFuncSetAttribute failed");
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWM * 128), lds, s, a);
    return blt_check_launch("conv3x3_pp");
}

}  // namespace

long blt_pp_pixels(int N, int H, int W) { return (long)N * (H + 1) * (W + 1); }
// Tile plan shared by the launch and by the count of statistics rows: `wide` = the 256-position / 8-wave / 4-stage form — ON for
// Cout = 64 (see below), OFF by default for Cout % 128 == 0 (debug key 19 = 2 turns it on: measured 90-94 us against 82-88 us for the
// 128-position form at B = 256) — else 128 positions with BN = 128 or 64.
struct ConvPPPlan { bool wide; int BN; int pw; };
static ConvPPPlan conv_pp_plan(int N, int H, int W, int Cin, int Cout) {
    ConvPPPlan pl;
    const int NS = Cin / 64;
    const int Pw = 256 + 2 * (W + 1) + 2;
    const int pww = cdiv(cdiv(Pw, 8), 8);
    const size_t ldsw = (size_t)(NS > 1 ? 2 : 1) * pww * 8192 + 4 * (size_t)128 * 128;
    pl.wide = Cout % 128 == 0 && pww <= 8 && ldsw <= 160 * 1024 && pww * 64 - (W + 2) <= BLT_PP_GUARD_TAIL && blt_debug_get(19) == 2;
    if (pl.wide) { pl.BN = 128; pl.pw = pww; return pl; }
    // ... ON for the 64-channel layers (256 positions x 64 channels, 80 KB: two 8-wave workgroups per CU instead of three 4-wave ones):
    // 6 498 tiles of 128 positions are 8.5 rounds of 5.5 us of fill each — half the tiles, half of that; measured 94 -> 82 us
    // (plain) and 123 -> 103 us (bn1 fused in) at B = 256.  debug key 19 = 1: the 128-position form everywhere.
    if (Cout == 64 && pww <= 8 && pww * 64 - (W + 2) <= BLT_PP_GUARD_TAIL && blt_debug_get(19) != 1) {
        pl.wide = true; pl.BN = 64; pl.pw = pww; return pl;
    }
    pl.pw = cdiv(cdiv(128 + 2 * (W + 1) + 2, 8), 4);
    pl.BN = (Cout % 128 == 0) ? 128 : 64;
    // a grid that would leave most CUs with a single workgroup (7x7x512: 256 tiles of 128 channels) runs twice as many half-width
    // tiles instead: two workgroups per CU cover each other's DMA waits (measured 50 -> 42 us; a deeper ring did not help)
    if (pl.BN == 128 && (long)cdiv(blt_pp_pixels(N, H, W), 128) * (Cout / 128) <= 320) pl.BN = 64;
    if (blt_debug_get(4) == 64) pl.BN = 64;
    if (blt_debug_get(4) == 128 && Cout % 128 == 0) pl.BN = 128;
    return pl;
}
// upper bound for any plan (rows that a launch does not write must be zero: they are summed)
int blt_conv3x3_pp_stat_rows(int N, int H, int W) { return 4 * cdiv(blt_pp_pixels(N, H, W), 256); }
// the rows the launch for this layer writes
int blt_conv3x3_pp_stat_rows_for(int N, int H, int W, int Cin, int Cout) {
    const ConvPPPlan pl = conv_pp_plan(N, H, W, Cin, Cout);
    return pl.wide ? 4 * cdiv(blt_pp_pixels(N, H, W), 256) : 2 * cdiv(blt_pp_pixels(N, H, W), 128);
}

int blt_conv3x3_pp(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, float* stat_sum, float* stat_sq,
                   hipStream_t s, const float* in_scale, const float* in_shift) {
    BLT_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0, "conv3x3_pp: bad args");
    BLT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3x3_pp: in_scale and in_shift go together");
    BLT_REQUIRE(((uintptr_t)in_scale % 16) == 0 && ((uintptr_t)in_shift % 16) == 0, "conv3x3_pp: in_scale / in_shift must be 16-byte aligned");
    BLT_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "conv3x3_pp: Cin=%d / Cout=%d must be multiples of 64", Cin, Cout);
    BLT_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "conv3x3_pp: stat_sum and stat_sq go together");
    BLT_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, "conv3x3_pp: operands must be 16-byte aligned");
    const int P = 128 + 2 * (W + 1) + 2;                     // patch pixels of the 128-position form
    BLT_REQUIRE(W + 2 <= BLT_PP_GUARD_FRONT && P <= 256, "conv3x3_pp: W=%d too wide for the patch / guards", W);
    const ConvPPPlan pl = conv_pp_plan(N, H, W, Cin, Cout);
    ConvPPArgs a;
    a.X = (const bf16*)x; a.Wt = (const bf16*)w; a.Y = (bf16*)y;
    a.Mq = blt_pp_pixels(N, H, W); a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stat_sum = stat_sum; a.stat_sq = stat_sq;
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.pw = pl.pw;
    const int BN = pl.BN, BM = pl.wide ? 256 : 128;
    a.tiles_n = Cout / BN;
    a.nblocks = cdiv(a.Mq, BM) * a.tiles_n;
    BLT_REQUIRE(a.Mq < (1L << 31), "conv3x3_pp: too many positions");
    // weights that fit an XCD's L2 beside the patches: keep neighbouring tiles on one XCD; otherwise the round-robin deal, which
    // gives each XCD every 8th tile and hence only (tiles_n | 8) of the weight slices
    a.chunked = ((long)Cout * 9 * Cin * 2 <= (2L << 20)) ? 1 : 0;
    if (blt_debug_get(5)) a.chunked = blt_debug_get(5) == 1;
    const int grid = a.chunked ? 8 * cdiv(a.nblocks, 8) : a.nblocks;
    const int NS = Cin / 64;
    const size_t stage = ((size_t)BM * (BN + 8) * 2 + 15) / 16 * 16;      // the staged bf16 output tile; the validity table sits behind it
    a.valid_off = (int)stage;
    if (pl.wide) {
        size_t lds = (size_t)(NS > 1 ? 2 : 1) * a.pw * 8192 + 4 * (size_t)BN * 128;
        if (lds < stage + 1024) lds = stage + 1024;
        if (BN == 64) return in_scale ? launch<64, 4, true, 4>(a, grid, lds, s) : launch<64, 4, false, 4>(a, grid, lds, s);
        return in_scale ? launch<128, 4, true, 4>(a, grid, lds, s) : launch<128, 4, false, 4>(a, grid, lds, s);
    }
    // two workgroups per CU need <= 80 KB each: the validity table (512 B) shares the epilogue's space behind the staged tile
    // a grid that gives most CUs a single workgroup: deep weight ring instead of a partner workgroup
    bool deep = a.nblocks <= 320;
    // (four stages for the 64-channel layers, which would still fit two workgroups per CU, measured SLOWER: 101 vs 85 us — those layers run
    // three 48 KB workgroups per CU and lose the third)
    if (blt_debug_get(6)) deep = blt_debug_get(6) == 1;
    size_t lds = (size_t)(NS > 1 ? 2 : 1) * a.pw * 4096 + (deep ? 4 : 2) * (size_t)BN * 128;
    if (lds < stage + 512) lds = stage + 512;
    // the tile overhang of the last workgroup's patch must stay inside the tail guard
    BLT_REQUIRE(a.pw * 32 - (W + 2) <= BLT_PP_GUARD_TAIL, "conv3x3_pp: tail guard too small");
    if (in_scale != nullptr) {
        if (BN == 128) return deep ? launch<128, 4, true>(a, grid, lds, s) : launch<128, 2, true>(a, grid, lds, s);
        return deep ? launch<64, 4, true>(a, grid, lds, s) : launch<64, 2, true>(a, grid, lds, s);
    }
    if (BN == 128) return deep ? launch<128, 4, false>(a, grid, lds, s) : launch<128, 2, false>(a, grid, lds, s);
    return deep ? launch<64, 4, false>(a, grid, lds, s) : launch<64, 2, false>(a, grid, lds, s);
}

// ---------------------------------------------------------------------------------------------------------------
// ResNet stem: 7x7 stride-2 pad-3 convolution of the 3-channel image, 64 output channels (encoder_cnn.py:17, torchvision conv1).
// Input: zero-bordered NHWC4 bf16 image [N][Hp][Wp][4] (image at (3,3)); weights packed [64][7][8][4] (K = 224: tap column 7 and
// channel 3 are zero).  As an implicit GEMM each output pixel fetched its own 7 x 64 B window — 448 B per pixel, 75 % of it shared
// with its neighbours.  Here a workgroup owns an 8 x 16 tile of output pixels of one image, stages the 21 x 38-pixel input patch
// (6.4 KB instead of 57 KB) and the whole filter (28 KB, seven [64 couts][64 B] planes, one per filter row) in LDS, and runs
// seven K = 32 MFMA steps: for filter row r the A fragment of output (orow, ocol) is the 16 bytes at patch pixel
// (2*orow + r, 2*ocol + 2*lg) — consecutive ocol are 16 B apart, so a fragment read is one contiguous 1 KB of LDS.
// 35 KB of LDS per workgroup: four workgroups per CU overlap their load / MFMA / store phases (a tile is only 56 MFMAs per wave).
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct StemArgs {
    const bf16* X;
    const bf16* Wt;
    bf16* Y;
    int N, Hp, Wp, Ho, Wo;
    float *stat_sum, *stat_sq;
};

constexpr int STEM_PITCH = 304;                 // 38 pixels x 8 B per patch row
constexpr int STEM_PATCH_BYTES = 7 * 1024;      // 21 rows x 19 chunks = 399 chunks -> 7 DMA instructions
constexpr int STEM_W_BYTES = 28 * 1024;         // 7 planes x 64 couts x 64 B

__global__ __launch_bounds__(256) void conv_stem_direct_kernel(const StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lg = lane >> 4;
    const int tiles_w = p.Wo >> 4, tiles_h = p.Ho >> 3;
    const int b = blockIdx.x;
    const int n = b / (tiles_w * tiles_h), rem = b - n * (tiles_w * tiles_h);
    const int th = rem / tiles_w, tw = rem - th * tiles_w;
    const int oh0 = th * 8, ow0 = tw * 16;
    char* const patch = smem;
    char* const wl = smem + STEM_PATCH_BYTES;

    // ---- LDS-DMA: patch (instructions 0..6), filter planes (28 instructions); instruction ii = j*4 + wave ----
    const bf16* xin = p.X + (((size_t)n * p.Hp + 2 * oh0) * p.Wp + 2 * ow0) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ii = j * 4 + wave;
        if (ii < 7) {
            const int c = ii * 64 + lane;
            const int row = c / 19, cc = c - row * 19;
            const void* src = (c < 21 * 19) ? (const void*)(xin + ((size_t)row * p.Wp) * 4 + cc * 8) : (const void*)g_zero16;
            dma16(src, patch + ii * 1024);
        }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int ii = j * 4 + wave;
        const int c = ii * 64 + lane;                    // chunk (r, cout, q): c = (r*64 + cout)*4 + q
        const int q = c & 3, co = (c >> 2) & 63, r = c >> 8;
        dma16(p.Wt + co * 224 + r * 32 + q * 8, wl + ii * 1024);
    }

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        bf16x8 af[4], bfr[2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            af[i] = *reinterpret_cast<const bf16x8*>(patch + (2 * (wm * 4 + i) + r) * STEM_PITCH + (l15 + lg) * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bfr[j] = *reinterpret_cast<const bf16x8*>(wl + ((r * 64 + wn * 32 + j * 16 + l15) * 4 + lg) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();

    // ---- BatchNorm partial statistics (every position of the tile is a real pixel): half tile = wave row ----
    if (p.stat_sum != nullptr) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[i][j][r];
                    s1 += v;
                    s2 += v * v;
                }
            s1 = sum_across_rows(s1); s2 = sum_across_rows(s2);
            if (lg == 0) {
                const size_t o = (size_t)(b * 2 + wm) * 64 + wn * 32 + j * 16 + l15;
                p.stat_sum[o] = s1;
                p.stat_sq[o] = s2;
            }
        }
    }
    // ---- bf16 tile -> LDS -> 16-byte stores.  MFMA result row (lg*4 + r) of fragment i is output column ocol = lg*4 + r of row orow = wm*4 + i
    constexpr int CSB = 72;
    bf16* Cs = reinterpret_cast<bf16*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[((wm * 4 + i) * 16 + lg * 4 + r) * CSB + wn * 32 + j * 16 + l15] = (bf16)acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int c = tid; c < 128 * 8; c += 256) {
        const int m = c >> 3, ch = c & 7;
        const int orow = m >> 4, ocol = m & 15;
        bf16* dst = p.Y + (((size_t)n * p.Ho + oh0 + orow) * p.Wo + ow0 + ocol) * 64 + ch * 8;
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(Cs + m * CSB + ch * 8);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stem + max-pool in one launch ("pooled extrema").  relu(bn(x)) is monotone in x with the sign of the BatchNorm scale, and that sign
// is the sign of gamma (a frozen parameter, known before the batch statistics are): so max_pool(relu(bn(x))) == relu(bn(ext(x))) with
// ext = max over the 3x3/2 window where gamma >= 0 and min where gamma < 0 — bit for bit, because the affine map is applied to
// the same bf16-rounded convolution outputs either way.  The kernel therefore writes ONE extremum per pooled pixel and channel
// (103 MB at B = 256) instead of the 411 MB pre-pool tensor that bn_relu_maxpool read back, plus the BatchNorm partial sums of ALL
// convolution outputs; bn_apply_pp (+ReLU) on the pooled tensor finishes it once the statistics are known.
//
// A workgroup owns 4 x 7 pooled pixels = convolution rows 8th-1 .. 8th+7 and columns 14tw-1 .. 14tw+13; it computes a 10 x 16 tile
// (rows 8th-1 .. 8th+8, columns 14tw-1 .. 14tw+14: 1.43x the MFMA work of the 8 x 16 tiles, the stem is write-bound) from a 25 x 38
// pixel patch.  Row / column -1 are the pool's padding (never selected); a convolution output enters the statistics in exactly one
// workgroup (tile rows 1..8, columns 1..14).  Patch chunks that would start outside the padded image read a zero page.
// ---------------------------------------------------------------------------------------------------------------
struct StemPoolArgs {
    const bf16* X;
    const bf16* Wt;
    const float* gamma;      // [64] BatchNorm weight: its sign selects max / min
    bf16* Y;                 // pooled PP output [N][Ho/2 + 1][Wo/2 + 1][64]: real pixels only
    int N, Hp, Wp, Ho, Wo;
    float *stat_sum, *stat_sq;
};
constexpr int STEMP_PATCH_BYTES = 8 * 1024;     // 25 rows x 19 chunks = 475 chunks -> 8 DMA instructions
constexpr int STEMP_LDS = STEM_W_BYTES + 2 * STEMP_PATCH_BYTES;      // 28 + 16 KB: three workgroups per CU

__device__ __forceinline__ float dpp_row_shl(float x, int n) {      // lane i of a 16-lane row receives lane i + n (n = 1, 2)
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, n == 1 ? __builtin_amdgcn_update_dpp(v, v, 0x101, 0xF, 0xF, false) : __builtin_amdgcn_update_dpp(v, v, 0x102, 0xF, 0xF, false));
}

// PERSISTENT, register-resident pooling.  History (B = 256, standalone): one workgroup per tile, tile staged in LDS, pooled from LDS:
// each workgroup lived 7.9 us for 0.5 us of MFMAs; persistent with the staged tile: 267 us (two 67 KB workgroups per CU, every phase
// serialised).  Now a workgroup keeps the filter in LDS, walks tiles b, b + grid, ... with the NEXT tile's patch in flight, and pools
// straight from the accumulators:
//   * the filter fragment is the MFMA's A operand, so a lane holds tile COLUMN l15 and 4 consecutive channels (lg*4 + r) of 5 tile rows;
//   * wave row 0 computes tile rows 0..4, wave row 1 rows 4..8 (row 9 was never needed; row 4 is computed twice), so both pooled rows of a
//     wave — windows over rows {0,1,2} and {2,3,4} of its five — are in-lane maxima, and the three columns of a window are two DPP row
//     shifts: no LDS staging, no second barrier, 44 KB of LDS;
//   * max / min by the sign of gamma as max(s*x) with s = +-1 (exact), on the bf16-ROUNDED outputs (what the two-pass form stored);
//   * BatchNorm partial sums stay in registers for the whole walk: two statistics rows per workgroup (1.5 K rows instead of 57 K).
__global__ __launch_bounds__(256) void conv_stem_pool_kernel(const StemPoolArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lg = lane >> 4;
    const int tiles_w = p.Wo / 14, tiles_h = p.Ho >> 3, per_img = tiles_w * tiles_h;
    const int ntiles = p.N * per_img;
    char* const wl = smem;
    char* const patch0 = smem + STEM_W_BYTES;

    // patch of tile `tile` -> buffer `buf` (two DMA instructions per wave); chunks that would start outside the padded image read zeros
    auto issue_patch = [&](int tile, int buf) {
        const int n = tile / per_img, rem = tile - n * per_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const int or0 = 8 * th - 1, oc0 = 14 * tw - 1;
        const bf16* ximg = p.X + (size_t)n * p.Hp * p.Wp * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ii = j * 4 + wave;
            const int c = ii * 64 + lane;
            const int row = c / 19, cc = c - row * 19;
            int ir = 2 * or0 + row;                              // padded-image row of patch row `row`
            ir = ir < 0 ? 0 : (ir > p.Hp - 1 ? p.Hp - 1 : ir);   // clamped rows feed only tile positions outside the image (never used)
            const int ic = 2 * oc0 + 2 * cc;                     // first of the chunk's two pixels
            const void* src = (c < 25 * 19 && ic >= 0 && ic + 1 < p.Wp) ? (const void*)(ximg + ((size_t)ir * p.Wp + ic) * 4) : (const void*)g_zero16;
            dma16(src, patch0 + buf * STEMP_PATCH_BYTES + ii * 1024);
        }
    };

    // ---- filter planes (28 DMA instructions, once) + the first patch ----
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int ii = j * 4 + wave;
        const int c = ii * 64 + lane;                    // LDS chunk (r, cout, q'): c = (r*64 + cout)*4 + q'
        // slot (r, cout, q') holds the filter's k-chunk q = (q' - 2*(cout >> 2)) & 3: without the rotation the 16 lanes of a
        // ds_read_b128 group hit each bank twice (couts 4 apart share a bank)
        const int qs = c & 3, co = (c >> 2) & 63, r = c >> 8;
        const int q = (qs - 2 * (co >> 2)) & 3;
        dma16(p.Wt + co * 224 + r * 32 + q * 8, wl + ii * 1024);
    }
    // XCD-aware walk: the hardware deals workgroup ids round-robin over the 8 XCDs, each with its own L2.  XCD x takes the CONTIGUOUS
    // tile range [x * chunk, (x+1) * chunk) and its workgroups walk it side by side, so that the patch halo two neighbouring tiles share
    // (the 25 x 38-pixel patch covers 16 x 28 owned pixels) is in that XCD's L2 instead of being fetched once per XCD (measured: 188 MB
    // read per launch against 105 MB of image).  Grids that are not a multiple of 8 keep the plain walk.
    int tile, tile_end, tile_step;
    if ((gridDim.x & 7) == 0) {
        const int chunk = (ntiles + 7) >> 3, xcd = (int)blockIdx.x & 7;
        tile = xcd * chunk + ((int)blockIdx.x >> 3);
        tile_end = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
        tile_step = (int)gridDim.x >> 3;
    } else {
        tile = blockIdx.x; tile_end = ntiles; tile_step = gridDim.x;
    }
    if (tile < tile_end) issue_patch(tile, 0);
    float st1[2][4], st2[2][4];      // this lane's share (tile column l15) of the statistics of channels wn*32 + j*16 + lg*4 + r
    float sgn[2][4];                 // +1: the pooled extremum of that channel is the max, -1: the min
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st1[j][r] = 0.f; st2[j][r] = 0.f;
            sgn[j][r] = (p.gamma[wn * 32 + j * 16 + lg * 4 + r] >= 0.f) ? 1.f : -1.f;
        }
    const int Hq = (p.Ho >> 1) + 1, Wq = (p.Wo >> 1) + 1;             // PP pitch of the pooled tensor
    const bool own_col = l15 >= 1 && l15 <= 14;
    const bool writer = (l15 & 1) == 0 && l15 <= 12;                  // holds the window over tile columns l15 .. l15 + 2
    int buf = 0;
    bf16* pend_dst = nullptr;
    s16x4 pend[2][2];
    for (; tile < tile_end; tile += tile_step, buf ^= 1) {
        const int n = tile / per_img, rem = tile - n * per_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const char* patch = patch0 + buf * STEMP_PATCH_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // this tile's patch (and the filter) have landed for every wave; everyone is done reading the other buffer
        if (tile + tile_step < tile_end) issue_patch(tile + tile_step, buf ^ 1);      // lands under the MFMAs / pooling below
        // the previous tile's pooled pixels leave HERE, one iteration late: the vmcnt(0) above would otherwise wait for stores that were
        // issued just before it (loads and stores do not retire in order with each other, so a counted wait cannot skip them)
        if (pend_dst != nullptr) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int lp = 0; lp < 2; ++lp) *reinterpret_cast<s16x4*>(pend_dst + (size_t)lp * Wq * 64 + j * 16) = pend[j][lp];
            pend_dst = nullptr;
        }
        f32x4 acc[5][2];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            bf16x8 af[5], bfr[2];
#pragma unroll
            for (int i = 0; i < 5; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(patch + (2 * (wm * 4 + i) + r) * STEM_PITCH + (l15 + lg) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = wn * 32 + j * 16 + l15;
                bfr[j] = *reinterpret_cast<const bf16x8*>(wl + ((r * 64 + co) * 4 + ((lg + 2 * (co >> 2)) & 3)) * 16);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        // ---- statistics over the positions this wave OWNS: its rows 1..4 (tile rows 1..4 / 5..8); the column mask (tile columns 1..14)
        //      is a per-lane constant and is applied once, after the walk.  This kernel is VALU-bound (K = 147: 70 MFMAs feed 40 results
        //      per lane that each need statistics + rounding + pooling), so every select here counts ----
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 1; i < 5; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[i][j][r];
                    st1[j][r] += v;
                    st2[j][r] += v * v;
                }
        // ---- 3x3 / 2 pooling: rows in-lane, columns by DPP; the pool's padding is convolution row / column -1 (tile row 0 of the
        //      first tile row, tile column 0 of the first tile column) ----
        const bool row0_pad = th == 0 && wm == 0;
        const bool col_pad = tw == 0 && l15 == 0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) v[i] = sgn[j][r] * (float)(bf16)acc[i][j][r];
                if (row0_pad) v[0] = -INFINITY;
                float m[2] = {fmaxf(fmaxf(v[0], v[1]), v[2]), fmaxf(fmaxf(v[2], v[3]), v[4])};
#pragma unroll
                for (int lp = 0; lp < 2; ++lp) {
                    if (col_pad) m[lp] = -INFINITY;
                    const float h = fmaxf(fmaxf(m[lp], dpp_row_shl(m[lp], 1)), dpp_row_shl(m[lp], 2));
                    pend[j][lp][r] = __builtin_bit_cast(short, (bf16)(sgn[j][r] * h));
                }
            }
        if (writer) pend_dst = p.Y + (((size_t)n * Hq + 4 * th + 2 * wm) * Wq + 7 * tw + (l15 >> 1)) * 64 + wn * 32 + lg * 4;
    }
    if (pend_dst != nullptr) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int lp = 0; lp < 2; ++lp) *reinterpret_cast<s16x4*>(pend_dst + (size_t)lp * Wq * 64 + j * 16) = pend[j][lp];
    }
    if (p.stat_sum != nullptr) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = own_col ? st1[j][r] : 0.f, s2 = own_col ? st2[j][r] : 0.f;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (l15 == 0) {
                    const size_t o = (size_t)(blockIdx.x * 2 + wm) * 64 + wn * 32 + j * 16 + lg * 4 + r;
                    p.stat_sum[o] = s1;
                    p.stat_sq[o] = s2;
                }
            }
    }
}

}  // namespace

bool blt_conv_stem_pool_ok(int dtype, int H, int W, int Hp, int Wp, int Cout) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    return dtype == BLT_BF16 && Cout == 64 && Ho % 8 == 0 && Wo % 14 == 0 && Hp >= H + 6 && Wp >= W + 6 && Wp % 2 == 0;
}
static int stem_pool_grid(int N, int H, int W) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const long tiles = (long)N * (Ho / 8) * (Wo / 14);
    return (int)(tiles < 768 ? tiles : 768);      // three 44 KB workgroups per CU
}
int blt_conv_stem_pool_stat_rows(int N, int H, int W) { return 2 * stem_pool_grid(N, H, W); }
int blt_conv_stem_pool(const void* x_padded, const void* w, const float* gamma, void* y_pool_pp, int N, int H, int W, int Hp, int Wp, float* stat_sum,
                       float* stat_sq, hipStream_t s) {
    BLT_REQUIRE(x_padded && w && gamma && y_pool_pp && N > 0 && blt_conv_stem_pool_ok(BLT_BF16, H, W, Hp, Wp, 64), "conv_stem_pool: unsupported geometry");
    BLT_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "conv_stem_pool: stat_sum and stat_sq go together");
    BLT_REQUIRE(((uintptr_t)x_padded % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y_pool_pp % 16) == 0 && ((uintptr_t)gamma % 16) == 0,
                "conv_stem_pool: operands must be 16-byte aligned");
    StemPoolArgs a;
    a.X = (const bf16*)x_padded; a.Wt = (const bf16*)w; a.gamma = gamma; a.Y = (bf16*)y_pool_pp; a.N = N; a.Hp = Hp; a.Wp = Wp;
    a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1; a.stat_sum = stat_sum; a.stat_sq = stat_sq;
    BLT_REQUIRE((long)N * (a.Ho / 8) * (a.Wo / 14) < (1L << 31), "conv_stem_pool: too many tiles");
    static BltDevFlag attr_set;
    if (!attr_set.get()) {
        if (hipFuncSetAttribute((const void*)conv_stem_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEMP_LDS) != hipSuccess) {
            blt_set_error("conv_stem_pool: hipFuncSetAttribute failed");
            return BLT_ERR_HIP;
        }
        attr_set.set();
    }
    hipLaunchKernelGGL(conv_stem_pool_kernel, dim3((unsigned)stem_pool_grid(N, H, W)), dim3(256), STEMP_LDS, s, a);
    return blt_check_launch("conv_stem_pool");
}

bool blt_conv_stem_direct_ok(int dtype, int H, int W, int Hp, int Wp, int Cout) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    return dtype == BLT_BF16 && Cout == 64 && Ho % 8 == 0 && Wo % 16 == 0 && Hp >= 2 * (Ho - 1) + 7 && Wp >= 2 * (Wo - 1) + 8 && Wp % 2 == 0;
}
int blt_conv_stem_direct_stat_rows(int N, int H, int W) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    return 2 * N * (Ho / 8) * (Wo / 16);
}
int blt_conv_stem_direct(const void* x_padded, const void* w, void* y, int N, int H, int W, int Hp, int Wp, float* stat_sum, float* stat_sq,
                         hipStream_t s) {
    BLT_REQUIRE(x_padded && w && y && N > 0 && blt_conv_stem_direct_ok(BLT_BF16, H, W, Hp, Wp, 64), "conv_stem_direct: unsupported geometry");
    BLT_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "conv_stem_direct: stat_sum and stat_sq go together");
    BLT_REQUIRE(((uintptr_t)x_padded % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, "conv_stem_direct: operands must be 16-byte aligned");
    StemArgs a;
    a.X = (const bf16*)x_padded; a.Wt = (const bf16*)w; a.Y = (bf16*)y; a.N = N; a.Hp = Hp; a.Wp = Wp;
    a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1; a.stat_sum = stat_sum; a.stat_sq = stat_sq;
    const int grid = N * (a.Ho / 8) * (a.Wo / 16);
    hipLaunchKernelGGL(conv_stem_direct_kernel, dim3(grid), dim3(256), STEM_PATCH_BYTES + STEM_W_BYTES, s, a);
    return blt_check_launch("conv_stem_direct");
}
