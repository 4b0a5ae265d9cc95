// Shared device/host helpers for libbltvqg_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define BLT_OK 0
#define BLT_ERR_ARG (-1)
#define BLT_ERR_HIP (-2)
#define BLT_ERR_STATE (-3)

enum { BLT_F32 = 0, BLT_BF16 = 1 };

void blt_set_error(const char* fmt, ...);
int blt_check_launch(const char* what);

#define BLT_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            blt_set_error(__VA_ARGS__);   \
            return BLT_ERR_ARG;           \
        }                                 \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------
// Cross-row exchange without LDS: gfx950's v_permlane16_swap / v_permlane32_swap swap the odd 16- (32-) lane rows of the first
// operand with the even rows of the second; with both operands = x the two results are x of "my row pair's even row" and of its
// odd row in every lane, i.e. lane l gets x[l] and x[l ^ 16] (x[l ^ 32]) in some order.  One VALU instruction instead of a
// ds_bpermute round trip through the LDS pipe (__shfl_xor) — these sit at the tail of latency-bound kernels.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void xor16_pair(float x, float& a, float& b) {
    const unsigned v = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]); b = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ void xor32_pair(float x, float& a, float& b) {
    const unsigned v = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]); b = __builtin_bit_cast(float, (unsigned)r[1]);
}
// sum / max over the four lanes l, l^16, l^32, l^48 (the same value in all four afterwards)
__device__ __forceinline__ float sum_across_rows(float x) {
    float a, b;
    xor16_pair(x, a, b); x = a + b;
    xor32_pair(x, a, b); return a + b;
}
__device__ __forceinline__ float max_across_rows(float x) {
    float a, b;
    xor16_pair(x, a, b); x = fmaxf(a, b);
    xor32_pair(x, a, b); return fmaxf(a, b);
}

// ---------------------------------------------------------------------------------
// element conversion
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// 8 consecutive elements of type T <-> 8 floats (16-B vector access for bf16, 2x16 B for f32)
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        float4 a = *reinterpret_cast<const float4*>(p);
        float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float* v) {
        bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void store(bf16* p, const float* v) {
        bf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
        *reinterpret_cast<bf16x8*>(p) = a;
    }
};

// ---------------------------------------------------------------------------------
// wave (64 lanes) and block reductions
// ---------------------------------------------------------------------------------
// (__shfl_xor is a ds_bpermute: six dependent LDS round trips per reduction.  Within a 16-lane row the reduction is four DPP row
// rotations — plain VALU operands — and across rows the two permlane swaps above.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum(float v) {      // all 16 lanes of a row get the row's sum
    v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); v += dpp_mov<0x121>(v);      // row_ror:8, 4, 2, 1
    return v;
}
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dpp_mov<0x128>(v)); v = fmaxf(v, dpp_mov<0x124>(v)); v = fmaxf(v, dpp_mov<0x122>(v)); v = fmaxf(v, dpp_mov<0x121>(v));
    return v;
}
// sum over a group of `lanes` (16, 32 or 64) consecutive lanes starting at a multiple of `lanes`
__device__ __forceinline__ float group_sum(float v, int lanes) {
    v = row_sum(v);
    if (lanes >= 32) { float a, b; xor16_pair(v, a, b); v = a + b; }
    if (lanes >= 64) { float a, b; xor32_pair(v, a, b); v = a + b; }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return sum_across_rows(row_sum(v)); }
__device__ __forceinline__ float wave_max(float v) { return max_across_rows(row_max(v)); }
// all threads get the block total; `red` is >= 16 floats of LDS; blockDim.x multiple of 64
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (explicit (seed, stream, element) addressing so that forward and
// backward regenerate identical dropout masks; SURVEY §7 "Stochastic parity")
// ---------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t blt_mulhi(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}
__host__ __device__ __forceinline__ void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                                    uint32_t c3, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = blt_mulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = blt_mulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// Dropout addressing: one Philox call serves the 8 elements [8*q, 8*q+7] of dropout site `stream`; element e uses the 16-bit
// lane (e & 7) of the 128 random bits and is kept when that lane >= round(p * 65536).
__host__ __device__ __forceinline__ void dropout_words(uint64_t seed, uint32_t stream, uint64_t q, uint32_t* out) {
    philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)q, (uint32_t)(q >> 32), stream, 0x6b657970u, out);
}
__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) {
    double t = (double)p * 65536.0 + 0.5;
    if (t < 0) t = 0;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}
__host__ __device__ __forceinline__ uint32_t dropout_lane(const uint32_t* w, int lane8) {
    return (w[lane8 >> 1] >> ((lane8 & 1) * 16)) & 0xFFFFu;
}
// keep decision for a single element index e
__host__ __device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t stream, uint64_t e, uint32_t thresh) {
    uint32_t w[4];
    dropout_words(seed, stream, e >> 3, w);
    return dropout_lane(w, (int)(e & 7)) >= thresh;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: one flag per (kernel instantiation, device), so that a process
// that drives several GPUs sets it on each of them (one process per GPU is the deployed form; tests and tools may not be)
struct BltDevFlag {
    bool done[32] = {};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 31; }
    bool get() const { return done[dev()]; }
    void set() { done[dev()] = true; }
};
