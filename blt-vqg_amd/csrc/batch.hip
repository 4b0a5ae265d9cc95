// Batch producer kernels (reference utils/data_loader.py:45-129 and the transforms.Compose at train_iq.py:264-272).
//
// MI355X-first: at the train step's rate (tens of thousands of pairs/s) a host pipeline of float HWC images has to move 29 GB/s
// (602 KB per stored image; a finished pinned batch over PCIe alone costs the step 9 %), and 288 GB of HBM hold the whole VQA image table once the
// reference's ToTensor -> ToPILImage round trip has been applied to it (that round trip is deterministic per image and produces
// bytes: 150 KB per image).  So the store lives in HBM as uint8 HWC, and one batch is an index gather + crop + Pillow-exact
// antialiased bilinear resample + /255 + Normalize, written as the fp32 NCHW tensor IQ.forward takes.  Integer work throughout up
// to the final normalisation: HBM/L2-bound gathers, nothing here is GEMM-shaped.
#include "kernels.h"

// byte(255 * x) as `pic.mul(255).byte()` computes it for a float tensor: fp32 product, truncation, wrap modulo 256
__global__ void image_store_u8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        uchar4 o;
        o.x = (uint8_t)((int)(v.x * 255.0f) & 255);
        o.y = (uint8_t)((int)(v.y * 255.0f) & 255);
        o.z = (uint8_t)((int)(v.z * 255.0f) & 255);
        o.w = (uint8_t)((int)(v.w * 255.0f) & 255);
        *reinterpret_cast<uchar4*>(dst + i) = o;
    } else {
        for (long j = i; j < n; ++j) dst[j] = (uint8_t)((int)(src[j] * 255.0f) & 255);
    }
}

// One thread per sample: the token rows of data_loader.py:59-86,115-116 as int64 (collate_fn's .long()).
//   posterior = question with [0] = <pos>, FIRST <end> removed (+ <pad> appended), category inserted at 1      -> q_len + 1
//   answer    = stored answer with FIRST <end> removed (+ <pad> appended), category inserted at 1              -> a_len + 1
// A row without <end> (truncated by utils/vocab.py:33-34) keeps all its tokens.
__global__ void batch_rows_kernel(const int* __restrict__ questions, const int* __restrict__ answers, const int* __restrict__ answer_types,
                                  const int* __restrict__ cat_word_ids, int n_cat, long n_rows, const long* __restrict__ index, int B,
                                  int q_len, int a_len, long* __restrict__ oq, long* __restrict__ op, long* __restrict__ oa,
                                  long* __restrict__ ot, long* __restrict__ oti) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    long r = index[b];
    const bool ok = r >= 0 && r < n_rows;
    if (!ok) r = 0;
    const int* q = questions + r * q_len;
    const int* a = answers + r * a_len;
    int ty = answer_types[r];
    ty = ty < 0 ? 0 : (ty >= n_cat ? n_cat - 1 : ty);
    const long cat = ok ? (long)cat_word_ids[ty] : 0;
    long* pq = oq + (long)b * q_len;
    long* pp = op + (long)b * (q_len + 1);
    long* pa = oa + (long)b * (a_len + 1);
    // posterior: walk the question once, skipping the first <end>
    int w = 0;
    bool dropped = false;
    for (int t = 0; t < q_len; ++t) {
        const int tok = ok ? q[t] : 0;
        pq[t] = tok;
        const int v = (t == 0) ? 5 : tok;            // <pos> replaces the first token BEFORE the <end> search
        if (!dropped && v == 3) { dropped = true; continue; }
        pp[w == 0 ? 0 : w + 1] = v;                  // leave slot 1 for the category
        ++w;
    }
    if (dropped) pp[w + 1] = 0;                       // the appended <pad> (w == q_len - 1 here)
    pp[1] = cat;
    w = 0;
    dropped = false;
    for (int t = 0; t < a_len; ++t) {
        const int v = ok ? a[t] : 0;
        if (!dropped && v == 3) { dropped = true; continue; }
        pa[w == 0 ? 0 : w + 1] = v;
        ++w;
    }
    if (dropped) pa[w + 1] = 0;
    pa[1] = cat;
    ot[b] = cat;
    oti[(long)b * 3 + 0] = 1;
    oti[(long)b * 3 + 1] = cat;
    oti[(long)b * 3 + 2] = 3;
}

struct BatchImgArgs {
    const uint8_t* table;      // [n_images, S, S, 3]
    const int* image_indices;  // [n_rows]
    const long* index;         // [B]
    const int* boxes;          // [B, 4] top, left, h, w
    const int* coeffs;         // [B, 2, out, 2 + KS]: horizontal then vertical; {first source pixel, taps, weights (22-bit fixed point)}
    float* out;                // [B, 3, out, out]
    uint8_t* out_u8;           // optional [B, out, out, 3]
    long n_images, n_rows;
    int S, B, osz, KS;
    float mean[3], stdv[3];
};

#define BLT_RESAMPLE_BITS 22      // Pillow: PRECISION_BITS = 32 - 8 - 2 for 8-bit channels

__device__ __forceinline__ int clip8(int v) {
    v >>= BLT_RESAMPLE_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One thread per output pixel (3 channels).  Identity crops (h == w == out: the only case the reference's scale=(1.0,1.2) leaves for
// a stored 224x224 image) copy; everything else runs Pillow's two passes — horizontal with rounding to 8 bits, then vertical — with
// the host-computed fixed-point weights, recomputing the <= KS horizontal results a pixel needs (the source rows sit in L2).
__device__ __forceinline__ void batch_sample_pixel(const BatchImgArgs& a, int b, int oy, int ox, int* px) {
    long r = a.index[b];
    px[0] = px[1] = px[2] = 0;
    const int top = a.boxes[b * 4 + 0], left = a.boxes[b * 4 + 1], h = a.boxes[b * 4 + 2], w = a.boxes[b * 4 + 3];
    bool ok = r >= 0 && r < a.n_rows && top >= 0 && left >= 0 && h > 0 && w > 0 && top + h <= a.S && left + w <= a.S;
    long img = ok ? (long)a.image_indices[r] : 0;
    ok = ok && img >= 0 && img < a.n_images;
    if (ok) {
        const uint8_t* src = a.table + img * a.S * a.S * 3;
        if (h == a.osz && w == a.osz) {
            const uint8_t* p = src + ((long)(top + oy) * a.S + left + ox) * 3;
            px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
        } else if (a.coeffs) {
            const int stride = 2 + a.KS;
            const int* ch = a.coeffs + (((long)b * 2 + 0) * a.osz + ox) * stride;
            const int* cv = a.coeffs + (((long)b * 2 + 1) * a.osz + oy) * stride;
            const int xmin = ch[0], ymin = cv[0];
            const int nx = min(ch[1], a.KS), ny = min(cv[1], a.KS);
            int acc[3] = {1 << (BLT_RESAMPLE_BITS - 1), 1 << (BLT_RESAMPLE_BITS - 1), 1 << (BLT_RESAMPLE_BITS - 1)};
            for (int y = 0; y < ny; ++y) {
                const int row = min(top + ymin + y, a.S - 1);
                int hs[3] = {1 << (BLT_RESAMPLE_BITS - 1), 1 << (BLT_RESAMPLE_BITS - 1), 1 << (BLT_RESAMPLE_BITS - 1)};
                for (int x = 0; x < nx; ++x) {
                    const int col = min(left + xmin + x, a.S - 1);
                    const uint8_t* p = src + ((long)row * a.S + col) * 3;
                    const int k = ch[2 + x];
                    hs[0] += (int)p[0] * k; hs[1] += (int)p[1] * k; hs[2] += (int)p[2] * k;
                }
                const int k = cv[2 + y];
                acc[0] += clip8(hs[0]) * k; acc[1] += clip8(hs[1]) * k; acc[2] += clip8(hs[2]) * k;
            }
            px[0] = clip8(acc[0]); px[1] = clip8(acc[1]); px[2] = clip8(acc[2]);
        }
    }
}

__global__ void batch_images_kernel(BatchImgArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)a.osz * a.osz;
    if (i >= per * a.B) return;
    const int b = (int)(i / per);
    const int oy = (int)((i - (long)b * per) / a.osz);
    const int ox = (int)(i - (long)b * per - (long)oy * a.osz);
    int px[3];
    batch_sample_pixel(a, b, oy, ox, px);
    float* o = a.out + ((long)b * 3 * a.osz + oy) * a.osz + ox;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // ToTensor: byte / 255 ; Normalize: (x - mean) / std — fp32, correctly rounded divisions like the reference's CPU ops
        const float x = __fdiv_rn((float)px[c], 255.0f);
        o[(long)c * per] = __fdiv_rn(__fsub_rn(x, a.mean[c]), a.stdv[c]);
    }
    if (a.out_u8) {
        uint8_t* u = a.out_u8 + i * 3;
        u[0] = (uint8_t)px[0]; u[1] = (uint8_t)px[1]; u[2] = (uint8_t)px[2];
    }
}

// Same transform written straight into the train-step engine's stem input (bltvqg_engine_image_input): zero-bordered NHWC4
// [B, Hp, Wp, 4] in the engine's dtype, image at (pad_top, pad_left), channel 3 = 0 — what bltvqg_img_pack makes of the fp32 NCHW
// tensor, without that tensor (77 MB written + read per batch of 128) and without the img_pack launch.  One thread per packed pixel.
template <typename T>
__global__ void batch_images_packed_kernel(BatchImgArgs a, T* __restrict__ out, int Hp, int Wp, int pt, int pl) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)Hp * Wp;
    if (i >= per * a.B) return;
    const int b = (int)(i / per);
    const int y = (int)((i - (long)b * per) / Wp) - pt;
    const int x = (int)((i - (long)b * per) % Wp) - pl;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)a.osz && (unsigned)x < (unsigned)a.osz) {
        int px[3];
        batch_sample_pixel(a, b, y, x, px);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[c], 255.0f), a.mean[c]), a.stdv[c]);
    }
    T* o = out + i * 4;
    if constexpr (sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 w;
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = (__bf16)v[c];
        *reinterpret_cast<bf16x4*>(o) = w;
    } else {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

int blt_batch_images_packed(const uint8_t* table, long n_images, int S, const int* image_indices, long n_rows, const long* index,
                            const int* boxes, const int* coeffs, int KS, int B, int osz, const float* mean_std, int dtype, void* out, int Hp,
                            int Wp, int pad_top, int pad_left, hipStream_t s) {
    BLT_REQUIRE(table && image_indices && index && boxes && mean_std && out, "batch_images_packed: null pointer");
    BLT_REQUIRE(dtype == BLT_F32 || dtype == BLT_BF16, "batch_images_packed: bad dtype");
    BLT_REQUIRE(n_images > 0 && n_rows > 0 && S > 0 && B > 0 && osz > 0 && KS >= 0 && KS <= 64, "batch_images_packed: bad sizes");
    BLT_REQUIRE(pad_top >= 0 && pad_left >= 0 && Hp >= osz + pad_top && Wp >= osz + pad_left, "batch_images_packed: image does not fit the packed buffer");
    BLT_REQUIRE(coeffs || KS == 0, "batch_images_packed: KS > 0 needs a coefficient table");
    BLT_REQUIRE(((uintptr_t)out % 16) == 0, "batch_images_packed: output must be 16-byte aligned");
    BatchImgArgs a;
    a.table = table; a.image_indices = image_indices; a.index = index; a.boxes = boxes; a.coeffs = coeffs; a.out = nullptr; a.out_u8 = nullptr;
    a.n_images = n_images; a.n_rows = n_rows; a.S = S; a.B = B; a.osz = osz; a.KS = KS;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean_std[c]; a.stdv[c] = mean_std[3 + c]; }
    const long n = (long)B * Hp * Wp;
    if (dtype == BLT_BF16) hipLaunchKernelGGL(batch_images_packed_kernel<bf16>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, a, (bf16*)out, Hp, Wp, pad_top, pad_left);
    else hipLaunchKernelGGL(batch_images_packed_kernel<float>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, a, (float*)out, Hp, Wp, pad_top, pad_left);
    return blt_check_launch("batch_images_packed");
}

int blt_image_store_u8(const float* images, uint8_t* out, long count, hipStream_t s) {
    BLT_REQUIRE(images && out && count > 0, "image_store_u8: bad args");
    BLT_REQUIRE(((uintptr_t)images % 16) == 0 && ((uintptr_t)out % 4) == 0, "image_store_u8: pointers must be 16 B / 4 B aligned");
    hipLaunchKernelGGL(image_store_u8_kernel, dim3((unsigned)cdiv(cdiv(count, 4), 256)), dim3(256), 0, s, images, out, count);
    return blt_check_launch("image_store_u8");
}

int blt_batch_rows(const int* questions, const int* answers, const int* answer_types, const int* cat_word_ids, int n_cat, long n_rows,
                   const long* index, int B, int q_len, int a_len, long* oq, long* op, long* oa, long* ot, long* oti, hipStream_t s) {
    BLT_REQUIRE(questions && answers && answer_types && cat_word_ids && index && oq && op && oa && ot && oti, "batch_rows: null pointer");
    BLT_REQUIRE(n_cat > 0 && n_rows > 0 && B > 0 && q_len > 0 && a_len > 0, "batch_rows: bad sizes");
    hipLaunchKernelGGL(batch_rows_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, s, questions, answers, answer_types, cat_word_ids, n_cat,
                       n_rows, index, B, q_len, a_len, oq, op, oa, ot, oti);
    return blt_check_launch("batch_rows");
}

int blt_batch_images(const uint8_t* table, long n_images, int S, const int* image_indices, long n_rows, const long* index, const int* boxes,
                     const int* coeffs, int KS, int B, int osz, const float* mean_std, float* out, uint8_t* out_u8, hipStream_t s) {
    BLT_REQUIRE(table && image_indices && index && boxes && mean_std && out, "batch_images: null pointer");
    BLT_REQUIRE(n_images > 0 && n_rows > 0 && S > 0 && B > 0 && osz > 0 && KS >= 0 && KS <= 64, "batch_images: bad sizes");
    BLT_REQUIRE(coeffs || KS == 0, "batch_images: KS > 0 needs a coefficient table");
    BatchImgArgs a;
    a.table = table; a.image_indices = image_indices; a.index = index; a.boxes = boxes; a.coeffs = coeffs; a.out = out; a.out_u8 = out_u8;
    a.n_images = n_images; a.n_rows = n_rows; a.S = S; a.B = B; a.osz = osz; a.KS = KS;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean_std[c]; a.stdv[c] = mean_std[3 + c]; }
    const long n = (long)B * osz * osz;
    hipLaunchKernelGGL(batch_images_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, a);
    return blt_check_launch("batch_images");
}
