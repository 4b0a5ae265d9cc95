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
};

template <int BN>
__global__ __launch_bounds__(256) void conv3x3_pp_kernel(const ConvPPArgs p) {
    constexpr int WN = BN / 2, TN = WN / 16, TM = 4, CH_B = BN * 8 / 256, B_STAGE = BN * 128, CSB = BN + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lg = lane >> 4;

    int b = blockIdx.x;
    if (p.chunked) b = (b & 7) * ((int)gridDim.x >> 3) + (b >> 3);     // hardware deals workgroup ids round-robin over the 8 XCDs
    if (b >= p.nblocks) return;
    const int tile_m = b / p.tiles_n, tile_n = b - tile_m * p.tiles_n;
    const long q0 = (long)tile_m * 128;
    const int n0 = tile_n * BN;
    const int pitch = p.W + 1;
    const int NS = p.Cin >> 6, nsteps = 9 * NS;
    const int patch_bytes = p.pw * 4096;
    char* const patch0 = smem;
    char* const patch1 = smem + (NS > 1 ? patch_bytes : 0);
    char* const ring = smem + (NS > 1 ? 2 : 1) * patch_bytes;
    float* const valid = reinterpret_cast<float*>(smem + p.valid_off);

    // patch row pp <-> input pixel q0 - pitch - 1 + pp; DMA instruction ii = j*4 + wave covers rows ii*8 .. ii*8+7, LDS slot
    // (lane & 7) of row pp holds global chunk (lane & 7) ^ (pp & 7)
    const bf16* const xb = p.X + (q0 - pitch - 1) * (long)p.Cin;
    auto issue_patch = [&](int sl, int j) {
        const int pp = (j * 4 + wave) * 8 + (lane >> 3);
        const int off = pp * p.Cin + ((((lane & 7) ^ (pp & 7))) << 3) + (sl << 6);
        dma16(xb + off, ((sl & 1) ? patch1 : patch0) + (j * 4 + wave) * 1024);
    };
    long b_off[CH_B];
#pragma unroll
    for (int j = 0; j < CH_B; ++j) {
        const int r = (j * 4 + wave) * 8 + (lane >> 3);
        b_off[j] = (n0 + r < p.Cout) ? (long)(n0 + r) * (9 * p.Cin) + ((((lane & 7) ^ (r & 7))) << 3) : -1;
    }
    auto issue_b = [&](int s, int sl, int t) {
        const int k0 = t * p.Cin + (sl << 6);
        char* st = ring + (s & 1) * B_STAGE;
#pragma unroll
        for (int j = 0; j < CH_B; ++j) {
            const void* src = (b_off[j] >= 0) ? (const void*)(p.Wt + b_off[j] + k0) : (const void*)g_zero16;
            dma16(src, st + (j * 4 + wave) * 1024);
        }
    };

    for (int j = 0; j < p.pw; ++j) issue_patch(0, j);
    issue_b(0, 0, 0);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int sl = 0, t = 0;
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of B(s) (and of every older DMA) has landed
        __builtin_amdgcn_s_barrier();                        // ... everyone's has; the other ring stage and the idle patch buffer are free
        {
            int t1 = t + 1, sl1 = sl;
            if (t1 == 9) { t1 = 0; ++sl1; }
            if (s + 1 < nsteps) issue_b(s + 1, sl1, t1);
        }
        if (sl + 1 < NS && t < p.pw) issue_patch(sl + 1, t);
        const char* pa = (sl & 1) ? patch1 : patch0;
        const char* bs = ring + (s & 1) * B_STAGE;
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
        if (tid < 128) {
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
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lg == 0) {
                const size_t o = (size_t)(tile_m * 2 + wm) * p.Cout + n0 + wn * WN + j * 16 + l15;
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
    for (int c = tid; c < 128 * CPR; c += 256) {
        const int row = c / CPR, ch = c - row * CPR;
        const long q = q0 + row;
        if (q < p.Mq)
            *reinterpret_cast<uint4*>(p.Y + q * p.Cout + n0 + ch * 8) = *reinterpret_cast<const uint4*>(Cs + row * CSB + ch * 8);
    }
}

template <int BN>
int launch(const ConvPPArgs& a, int grid, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv3x3_pp_kernel<BN>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            blt_set_error("conv3x3_pp: hipFuncSetAttribute failed");
            return BLT_ERR_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    return blt_check_launch("conv3x3_pp");
}

}  // namespace

long blt_pp_pixels(int N, int H, int W) { return (long)N * (H + 1) * (W + 1); }
int blt_conv3x3_pp_stat_rows(int N, int H, int W) { return 2 * cdiv(blt_pp_pixels(N, H, W), 128); }

int blt_conv3x3_pp(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, float* stat_sum, float* stat_sq,
                   hipStream_t s) {
    BLT_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0, "conv3x3_pp: bad args");
    BLT_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "conv3x3_pp: Cin=%d / Cout=%d must be multiples of 64", Cin, Cout);
    BLT_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "conv3x3_pp: stat_sum and stat_sq go together");
    BLT_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, "conv3x3_pp: operands must be 16-byte aligned");
    const int P = 128 + 2 * (W + 1) + 2;                     // patch pixels
    BLT_REQUIRE(W + 2 <= BLT_PP_GUARD_FRONT && P <= 256, "conv3x3_pp: W=%d too wide for the patch / guards", W);
    ConvPPArgs a;
    a.X = (const bf16*)x; a.Wt = (const bf16*)w; a.Y = (bf16*)y;
    a.Mq = blt_pp_pixels(N, H, W); a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stat_sum = stat_sum; a.stat_sq = stat_sq;
    a.pw = cdiv(cdiv(P, 8), 4);
    const int BN = (Cout % 128 == 0) ? 128 : 64;
    a.tiles_n = Cout / BN;
    a.nblocks = cdiv(a.Mq, 128) * a.tiles_n;
    // weights that fit an XCD's L2 beside the patches: keep neighbouring tiles on one XCD; otherwise the round-robin deal, which
    // gives each XCD every 8th tile and hence only (tiles_n | 8) of the weight slices
    a.chunked = ((long)Cout * 9 * Cin * 2 <= (2L << 20)) ? 1 : 0;
    const int grid = a.chunked ? 8 * cdiv(a.nblocks, 8) : a.nblocks;
    const int NS = Cin / 64;
    // two workgroups per CU need <= 80 KB each: the validity table (512 B) shares the epilogue's space behind the staged tile
    size_t lds = (size_t)(NS > 1 ? 2 : 1) * a.pw * 4096 + 2 * (size_t)BN * 128;
    const size_t stage = ((size_t)128 * (BN + 8) * 2 + 15) / 16 * 16;
    if (lds < stage + 512) lds = stage + 512;
    a.valid_off = (int)stage;
    BLT_REQUIRE(a.Mq < (1L << 31), "conv3x3_pp: too many positions");
    // the tile overhang of the last workgroup's patch must stay inside the tail guard
    BLT_REQUIRE(a.pw * 32 - (W + 2) <= BLT_PP_GUARD_TAIL, "conv3x3_pp: tail guard too small");
    return BN == 128 ? launch<128>(a, grid, lds, s) : launch<64>(a, grid, lds, s);
}
