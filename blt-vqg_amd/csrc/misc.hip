// HBM-bound plumbing kernels of the train step: token preparation, embedding gather / scatter-add, row-0 injections,
// ReLU/dropout backward mask, bias-gradient column sums, dtype casts / layout packs, the loss kernels (token CE with
// fused backward, bag-of-words CE, MSE, reparameterisation + KL), gradient norm and fused clip + Adam.
//
// Reference call sites: models/iq.py:57-79 (embedding), models/decoder_transformer.py:24-35 (shift / injection),
// models/transformer_layers.py:41-59,536-540 (Latent, gaussian_kld), train_iq.py:81-103 (losses),
// train_iq.py:259-261,372 (Adam, clip_grad_norm 5).
#include "kernels.h"

namespace {

inline int ew_grid(long n, int per_block = 256) {
    long g = (n + per_block - 1) / per_block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------------------------
// tokens
// ---------------------------------------------------------------------------------------------------------------
// ids_all = [ctx (B*Sa) | shifted target (B*T) | post (B*Sp)], pos_all = position inside the sequence,
// tgt_shift = [<start>, target[:, :-1]] (decoder_transformer.py:24-27), counters[0] = #non-pad targets,
// counters[1 + b] = #non-pad targets of sample b.
__global__ void prep_tokens_kernel(const long long* __restrict__ ctx, const long long* __restrict__ post,
                                   const long long* __restrict__ tgt, int B, int Sa, int Sp, int T, int* __restrict__ ids_all,
                                   int* __restrict__ pos_all, int* __restrict__ tgt_shift, int* __restrict__ tgt32,
                                   int* __restrict__ ctx32, int* __restrict__ post32, float* __restrict__ counters, int V,
                                   float* __restrict__ bad_ids) {
    const int na = B * Sa, np = B * Sp, nt = B * T;
    const int total = na + np + nt;
    // Every id that reaches the embedding gather / scatter, the attention masks and the cross-entropy kernels passes through here: an
    // id outside [0, V) (vocabulary / dataset mismatch; the reference raises a device-side index error) is counted in *bad_ids, which
    // the host surfaces as an error (engine_read item 4), and replaced by <pad> so that nothing indexes out of bounds meanwhile.
    int nbad = 0;
    auto chk = [&](long long v) -> int {
        if (v < 0 || v >= (long long)V) { ++nbad; return 0; }
        return (int)v;
    };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < na) {
            const int v = chk(ctx[i]);
            ids_all[i] = v; ctx32[i] = v; pos_all[i] = i % Sa;
        } else if (i < na + nt) {
            const int j = i - na;
            const int t = j % T;
            const int v = (t == 0) ? 1 : chk(tgt[j - 1]);
            ids_all[i] = v; tgt_shift[j] = v; pos_all[i] = t;
            tgt32[j] = chk(tgt[j]);
        } else {
            const int j = i - na - nt;
            const int v = chk(post[j]);
            ids_all[i] = v; post32[j] = v; pos_all[i] = j % Sp;
        }
    }
    if (nbad && bad_ids) atomicAdd(bad_ids, (float)nbad);
    if (blockIdx.x == 0) {
        // per-sample and total non-pad target counts (exact small integers in fp32; block reduction, deterministic)
        __shared__ float red[16];
        float mine = 0.f;
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            int c = 0;
            for (int t = 0; t < T; ++t) { const long long v = tgt[b * T + t]; c += (v > 0 && v < (long long)V); }
            counters[1 + b] = (float)c;
            mine += (float)c;
        }
        const float total_cnt = block_sum(mine, red);
        if (threadIdx.x == 0) counters[0] = total_cnt;
    }
}

template <typename T>
__global__ void embed_gather_kernel(const float* __restrict__ table, const int* __restrict__ ids, T* __restrict__ out, long rows,
                                    int E, int ld) {
    const long total = rows * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / ld;
        const int e = (int)(i - m * ld);
        out[i] = from_f32<T>(e < E ? table[(long)ids[m] * E + e] : 0.f);
    }
}

template <typename T>
__global__ void embed_scatter_kernel(const T* __restrict__ d, int ld, const int* __restrict__ ids, float* __restrict__ dtable,
                                     long rows, int E, int pad_id) {
    const long total = rows * E;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / E;
        const int e = (int)(i - m * E);
        const int id = ids[m];
        if (id != pad_id) atomicAdd(dtable + (long)id * E + e, to_f32(d[m * ld + e]));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// elementwise
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rows_add_kernel(T* __restrict__ y, long ys, const T* __restrict__ a, long as, const T* __restrict__ c, long cs,
                                int B, int n, int accumulate) {
    const long total = (long)B * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / n;
        const int j = (int)(i - b * n);
        float v = to_f32(a[b * as + j]);
        if (c != nullptr) v += to_f32(c[b * cs + j]);
        if (accumulate) v += to_f32(y[b * ys + j]);
        y[b * ys + j] = from_f32<T>(v);
    }
}

// The gradient bookkeeping of the "row 0" injections (iq.py:95-105: image feature and z are ADDED to token 0 of the decoder input, z +
// encoder row 0 feed the reconstructor, z + feature feed z_classifier) in one launch instead of six rows_add launches:
//   d_feats += dx0 + g_zc ;  d_zproj = dx0 + g_rin + g_zc (latent phase) ;  d_enc[:,0] += g_rin
// dx0 = d(decoder input)[:,0], g_rin = d(reconstructor input), g_zc = d(z_classifier input); null pointers drop their terms.
template <typename T>
__global__ void row0_sums_kernel(const T* __restrict__ dx0, long sdx, const T* __restrict__ g_rin, const T* __restrict__ g_zc, T* __restrict__ d_feats,
                                 T* __restrict__ d_zproj, T* __restrict__ d_enc, long senc, int B, int n) {
    const long total = (long)B * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / n;
        const int j = (int)(i - b * n);
        const float x = to_f32(dx0[b * sdx + j]);
        const float r = g_rin ? to_f32(g_rin[i]) : 0.f;
        const float z = g_zc ? to_f32(g_zc[i]) : 0.f;
        d_feats[i] = from_f32<T>(to_f32(d_feats[i]) + x + z);
        if (d_zproj) d_zproj[i] = from_f32<T>(x + r + z);
        if (g_rin) d_enc[b * senc + j] = from_f32<T>(to_f32(d_enc[b * senc + j]) + r);
    }
}

template <typename T>
__global__ void mask_scale_kernel(const T* __restrict__ dy, const T* __restrict__ ym, T* __restrict__ y, long nchunks, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
        float d[8], m[8];
        Vec8<T>::load(dy + i * 8, d);
        Vec8<T>::load(ym + i * 8, m);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = (m[e] != 0.f) ? d[e] * scale : 0.f;
        Vec8<T>::store(y + i * 8, d);
    }
}

// column sums: block = 64 columns x 4 row-lanes; grid.y splits rows; float atomics combine the row splits
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ld, long M, int N, float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (n < N)
        for (long m = (long)blockIdx.y * 4 + ry; m < M; m += (long)gridDim.y * 4) s += to_f32(x[m * ld + n]);
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && n < N) atomicAdd(out + n, sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx]);
}

template <typename TS, typename TD>
__global__ void cast_rows_kernel(const TS* __restrict__ src, int lds_, TD* __restrict__ dst, int ldd, long rows, int cols, int width) {
    // width = ldd: columns [cols, ldd) are zero-filled (padded operand); width = cols: plain strided 2-D copy
    const long total = rows * width;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / width;
        const int c = (int)(i - r * width);
        dst[r * ldd + c] = from_f32<TD>(c < cols ? to_f32(src[r * lds_ + c]) : 0.f);
    }
}

// bf16 shadows of a list of fp32 matrices that live in one flat buffer: every table entry {off, rows, cols, first tile} names a
// row-major [rows, cols] matrix at element offset `off`; its 64x64 tiles are cast into `dst` (same offsets, same layout) and into
// `dstT` as the TRANSPOSE [cols, rows] (leading dimension rows).  The transposed copy turns the input-gradient GEMM
// dX = dY W into the k-contiguous (NT) form that the LDS-DMA kernel takes.  One launch for the whole parameter buffer.
__global__ __launch_bounds__(256) void shadow_transpose_kernel(const float* __restrict__ src, bf16* __restrict__ dst, bf16* __restrict__ dstT,
                                                              const int4* __restrict__ table, int nent, int tile_base) {
    __shared__ float tile[64][65];
    const int bid = (int)blockIdx.x + tile_base;      // (a launch may cover a sub-range of the table: `table` points at its first entry)
    int e = 0;
    while (e + 1 < nent && bid >= table[e + 1].w) ++e;          // uniform scan, <= a few dozen entries
    const int4 t = table[e];
    const int off = t.x, rows = t.y, cols = t.z < 0 ? -t.z : t.z;
    const bool transposed = t.z > 0;          // cols < 0: plain shadow only (rows not a multiple of 8)
    const int tiles_c = (cols + 63) >> 6;
    const int lt = bid - t.w;
    const int r0 = (lt / tiles_c) << 6, c0 = (lt % tiles_c) << 6;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = src[(size_t)off + (size_t)r * cols + c];
            if (dst) dst[(size_t)off + (size_t)r * cols + c] = (bf16)v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (!transposed) return;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dstT[(size_t)off + (size_t)c * rows + r] = (bf16)tile[tx][i];
    }
}

// NCHW fp32 image -> NHWC [N, Hp, Wp, Cpad]: image placed at (pt, pl), zeros in the border and in the padded channels.
// One thread per output PIXEL: the C plane reads are coalesced along w, the Cpad channel values are written contiguously.
template <typename T>
__global__ void img_pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int H, int W, int Cpad, int pt, int pl,
                                int Hp, int Wp) {
    const long total = (long)N * Hp * Wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int w = (int)(t % Wp) - pl; t /= Wp;
        const int h = (int)(t % Hp) - pt;
        const long n = t / Hp;
        const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        T* o = dst + i * Cpad;
        for (int c = 0; c < Cpad; ++c) o[c] = from_f32<T>((in && c < C) ? src[((n * C + c) * H + h) * W + w] : 0.f);
    }
}

// The stem's case (3 planes -> 4 channels): one workgroup per padded output ROW, one thread per pixel, so no per-thread div/mod; the three
// plane reads are coalesced along w and the pixel's 4 channels leave as ONE 8-byte (bf16) / 16-byte (fp32) store.
template <typename T>
__global__ __launch_bounds__(256) void img_pack_c3_kernel(const float* __restrict__ src, T* __restrict__ dst, int H, int W, int pt, int pl, int Hp, int Wp) {
    const int hp = blockIdx.x % Hp;
    const long n = blockIdx.x / Hp;
    const int h = hp - pt;
    const bool row_in = (unsigned)h < (unsigned)H;
    const float* p = src + (n * 3 * H + (row_in ? h : 0)) * W;
    const long plane = (long)H * W;
    T* o = dst + ((n * Hp + hp) * (long)Wp) * 4;
    for (int wp = threadIdx.x; wp < Wp; wp += 256) {
        const int w = wp - pl;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (row_in && (unsigned)w < (unsigned)W) { v0 = p[w]; v1 = p[plane + w]; v2 = p[2 * plane + w]; }
        if constexpr (sizeof(T) == 2) {
            s16x4 r;
            r[0] = __builtin_bit_cast(short, from_f32<T>(v0)); r[1] = __builtin_bit_cast(short, from_f32<T>(v1));
            r[2] = __builtin_bit_cast(short, from_f32<T>(v2)); r[3] = 0;
            *reinterpret_cast<s16x4*>(o + (long)wp * 4) = r;
        } else {
            *reinterpret_cast<float4*>(o + (long)wp * 4) = make_float4(v0, v1, v2, 0.f);
        }
    }
}

// bf16, W % 4 == 0, Wp even, Wp <= 256: one WAVE per padded output row (four rows per workgroup).  A lane reads FOUR pixels of each plane as
// one 16-byte load (coalesced along w), the row is assembled in LDS ([Wp] x 8 bytes, pad pixels zero) and leaves as 16-byte stores that are
// consecutive across the wave — the per-pixel form above moves 4-byte loads and 8-byte stores (217 us for 263 MB at B = 256: 1.2 TB/s).
__global__ __launch_bounds__(256) void img_pack_c3v_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int N, int H, int W, int pt, int pl, int Hp,
                                                           int Wp) {
    __shared__ __attribute__((aligned(16))) s16x4 row[4][256];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + wv;          // padded row index over the batch
    const bool live = r < (long)N * Hp;
    const int hp = live ? (int)(r % Hp) : 0;
    const long n = live ? r / Hp : 0;
    const int h = hp - pt;
    const bool row_in = live && (unsigned)h < (unsigned)H;
    const long plane = (long)H * W;
    const float* p = src + (n * 3 * H + (row_in ? h : 0)) * W;
    const s16x4 z = {0, 0, 0, 0};
    if (row_in) {
        for (int q = lane; q < W / 4; q += 64) {
            const float4 a = *reinterpret_cast<const float4*>(p + 4 * q);
            const float4 b = *reinterpret_cast<const float4*>(p + plane + 4 * q);
            const float4 c = *reinterpret_cast<const float4*>(p + 2 * plane + 4 * q);
            const float va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {b.x, b.y, b.z, b.w}, vc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s16x4 o;
                o[0] = __builtin_bit_cast(short, from_f32<bf16>(va[i])); o[1] = __builtin_bit_cast(short, from_f32<bf16>(vb[i]));
                o[2] = __builtin_bit_cast(short, from_f32<bf16>(vc[i])); o[3] = 0;
                row[wv][pl + 4 * q + i] = o;
            }
        }
        for (int j = lane; j < Wp - W; j += 64) row[wv][j < pl ? j : W + j] = z;      // left / right border pixels
    } else {
        for (int wp = lane; wp < Wp; wp += 64) row[wv][wp] = z;                       // top / bottom border rows
    }
    __syncthreads();
    if (!live) return;
    uint4* o = reinterpret_cast<uint4*>(dst + r * (long)Wp * 4);
    const uint4* rs = reinterpret_cast<const uint4*>(&row[wv][0]);
    for (int i = lane; i < Wp / 2; i += 64) o[i] = rs[i];
}

// conv weight [Cout, Cin, KH, KW] fp32 -> [Cout, KH, KWpad, Cpad] T (k index = (r*KWpad + s)*Cpad + c)
template <typename T>
__global__ void conv_pack_w_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int KH, int KW, int Cpad, int KWpad) {
    const long total = (long)Cout * KH * KWpad * Cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        long t = i / Cpad;
        const int s = (int)(t % KWpad); t /= KWpad;
        const int r = (int)(t % KH);
        const long o = t / KH;
        out[i] = from_f32<T>((c < Cin && s < KW) ? w[((o * Cin + c) * KH + r) * KW + s] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// losses
// ---------------------------------------------------------------------------------------------------------------
// One block per logits row: three passes (max, sum-exp, gradient) over a row that stays in L2/L1.
template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(T* __restrict__ logits, int ld, const int* __restrict__ target, int V,
                                                const float* __restrict__ count, float gscale, float* __restrict__ loss_out,
                                                int write_grad) {
    __shared__ float red[16];
    const long row = blockIdx.x;
    T* x = logits + row * ld;
    const int tgt = target[row];
    const float inv_count = 1.f / fmaxf(count[0], 1.f);
    if (tgt == 0) {   // ignore_index: contributes nothing, gradient row is zero
        if (write_grad)
            for (int v = threadIdx.x; v < ld; v += blockDim.x) x[v] = from_f32<T>(0.f);
        return;
    }
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += blockDim.x) m = fmaxf(m, to_f32(x[v]));
    m = block_max(m, red);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) s += __expf(to_f32(x[v]) - m);
    s = block_sum(s, red);
    const float lse = m + __logf(s);
    // The target logit must be consumed BEFORE the barrier: the gradient pass below overwrites the row in place, and a plain
    // load that only thread 0 needs may legally be sunk past __syncthreads() into the `if`, where it races with the thread that
    // writes x[tgt] (seen as an occasional wrong loss for tgt = 1, same wave as thread 0; gradients were never affected).
    if (threadIdx.x == 0) atomicAdd(loss_out, (lse - to_f32(x[tgt])) * inv_count);
    __syncthreads();
    if (write_grad) {
        const float g = gscale * inv_count;
        for (int v = threadIdx.x; v < ld; v += blockDim.x) {
            float d = 0.f;
            if (v < V) d = (__expf(to_f32(x[v]) - lse) - (v == tgt ? 1.f : 0.f)) * g;
            x[v] = from_f32<T>(d);
        }
    }
}

// Register-resident variants (rows of up to 256 * 8 * NCH columns): every thread keeps its NCH chunks of 8 columns in registers, so a
// row is read ONCE with 16-byte loads and its gradient written once — the three passes above are three trips through L2 with 2-byte
// accesses.  Thread t owns chunks t, t+256, ... (coalesced).
constexpr int CE_ROWS_PER_BLOCK = 4;
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ce_rows_kernel(T* __restrict__ logits, int ld, const int* __restrict__ target, long nrows, int V,
                                                     const float* __restrict__ count, float gscale, float* __restrict__ loss_out,
                                                     int write_grad) {
    __shared__ float red[16];
    const float inv_count = 1.f / fmaxf(count[0], 1.f);
    const int nch = ld >> 3;
    float loss_acc = 0.f;       // of the thread that holds the target column, over this block's rows
    // CE_ROWS_PER_BLOCK rows per block: the loss of every row ends in a float atomic on ONE address, and those serialise
    for (long row = (long)blockIdx.x * CE_ROWS_PER_BLOCK; row < (long)(blockIdx.x + 1) * CE_ROWS_PER_BLOCK && row < nrows; ++row) {
    T* x = logits + row * ld;
    const int tgt = target[row];
    if (tgt == 0) {   // ignore_index: contributes nothing, gradient row is zero
        if (write_grad) {
            float z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = 0.f;
            for (int c = threadIdx.x; c < nch; c += 256) Vec8<T>::store(x + c * 8, z);
        }
        continue;
    }
    float v[NCH][8];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nch) Vec8<T>::load(x + c * 8, v[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (c >= nch || c * 8 + e >= V) v[j][e] = -INFINITY;
            m = fmaxf(m, v[j][e]);
        }
    }
    m = block_max(m, red);
    float s = 0.f, xt = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s += __expf(v[j][e] - m);
            if ((threadIdx.x + 256 * j) * 8 + e == tgt) xt = v[j][e];
        }
    s = block_sum(s, red);
    const float lse = m + __logf(s);
    if (((tgt >> 3) & 255) == (int)threadIdx.x) loss_acc += (lse - xt) * inv_count;     // the thread that holds the target column
    if (write_grad) {
        const float g = gscale * inv_count;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < nch) {
                float d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int col = c * 8 + e;
                    d[e] = (col < V) ? (__expf(v[j][e] - lse) - (col == tgt ? 1.f : 0.f)) * g : 0.f;
                }
                Vec8<T>::store(x + c * 8, d);
            }
        }
    }
    }
    loss_acc = block_sum(loss_acc, red);
    if (threadIdx.x == 0 && loss_acc != 0.f) atomicAdd(loss_out, loss_acc);
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void bow_ce_rows_kernel(const T* __restrict__ z, int ld, const int* __restrict__ target, int Tn, int V,
                                                         const float* __restrict__ count, float gscale, float* __restrict__ loss_out,
                                                         T* __restrict__ dz) {
    __shared__ float red[16];
    __shared__ int tg[64];
    const int b = blockIdx.x;
    const T* x = z + (long)b * ld;
    if (threadIdx.x < Tn) tg[threadIdx.x] = target[b * Tn + threadIdx.x];
    __syncthreads();
    int nb = 0;
    for (int t = 0; t < Tn; ++t) nb += (tg[t] != 0);
    const float inv_count = 1.f / fmaxf(count[0], 1.f);
    const int nch = ld >> 3;
    float v[NCH][8];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nch) Vec8<T>::load(x + c * 8, v[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (c >= nch || c * 8 + e >= V) v[j][e] = -INFINITY;
            m = fmaxf(m, v[j][e]);
        }
    }
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __expf(v[j][e] - m);
    s = block_sum(s, red);
    const float lse = m + __logf(s);
    const float g = gscale * inv_count;
    float l = 0.f;      // this thread's share of sum_t (lse - x[target_t]): its columns, weighted by how often they are a target
    T* d = dz ? dz + (long)b * ld : nullptr;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int col = c * 8 + e;
                int hits = 0;
                for (int t = 0; t < Tn; ++t) hits += (tg[t] == col);
                if (col == 0) hits = 0;                                  // ignore_index
                if (hits) l += (float)hits * (lse - v[j][e]);
                o[e] = (col < V) ? ((float)nb * __expf(v[j][e] - lse) - (float)hits) * g : 0.f;
            }
            if (d) Vec8<T>::store(d + c * 8, o);
        }
    }
    l = block_sum(l, red);
    if (threadIdx.x == 0) atomicAdd(loss_out, l * inv_count);
}

// bag-of-words CE (train_iq.py:92-94 without materialising the (B,T,V) repeat): one block per sample.
template <typename T>
__global__ __launch_bounds__(256) void bow_ce_kernel(const T* __restrict__ z, int ld, const int* __restrict__ target, int Tn, int V,
                                                    const float* __restrict__ count, float gscale, float* __restrict__ loss_out,
                                                    T* __restrict__ dz) {
    __shared__ float red[16];
    __shared__ int tg[64];
    const int b = blockIdx.x;
    const T* x = z + (long)b * ld;
    if (threadIdx.x < Tn) tg[threadIdx.x] = target[b * Tn + threadIdx.x];
    __syncthreads();
    int nb = 0;
    for (int t = 0; t < Tn; ++t) nb += (tg[t] != 0);
    const float inv_count = 1.f / fmaxf(count[0], 1.f);
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += blockDim.x) m = fmaxf(m, to_f32(x[v]));
    m = block_max(m, red);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) s += __expf(to_f32(x[v]) - m);
    s = block_sum(s, red);
    const float lse = m + __logf(s);
    if (threadIdx.x == 0) {
        float l = 0.f;
        for (int t = 0; t < Tn; ++t)
            if (tg[t] != 0) l += lse - to_f32(x[tg[t]]);
        atomicAdd(loss_out, l * inv_count);
    }
    if (dz != nullptr) {
        const float g = gscale * inv_count;
        T* d = dz + (long)b * ld;
        for (int v = threadIdx.x; v < ld; v += blockDim.x) {
            float o = 0.f;
            if (v < V) {
                int hits = 0;
                for (int t = 0; t < Tn; ++t) hits += (tg[t] == v && v != 0);
                o = ((float)nb * __expf(to_f32(x[v]) - lse) - (float)hits) * g;
            }
            d[v] = from_f32<T>(o);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mse_kernel(const T* __restrict__ a, const T* __restrict__ b, long n, float gscale,
                                                 float* __restrict__ loss_out, T* __restrict__ da, T* __restrict__ db, long n_div) {
    __shared__ float red[16];
    float s = 0.f;
    const float inv = 1.f / (float)n_div;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = to_f32(a[i]) - to_f32(b[i]);
        s += d * d;
        const float g = 2.f * d * inv * gscale;
        if (da) da[i] = from_f32<T>(g);
        if (db) db[i] = from_f32<T>(-g);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(loss_out, s * inv);
}

// z = eps*exp(0.5*logvar_q) + mu_q ; kld = mean_b( -0.5 sum(1 + lq - lp - (mp-mq)^2/e^lp - e^lq/e^lp) )
template <typename T>
__global__ __launch_bounds__(256) void latent_fwd_kernel(const T* __restrict__ mlvp, const T* __restrict__ mlvq,
                                                        const float* __restrict__ eps, T* __restrict__ z, float* __restrict__ kld,
                                                        int B, int Z, int ld) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < Z; j += blockDim.x) {
        const float mp = to_f32(mlvp[(long)b * ld + j]), lp = to_f32(mlvp[(long)b * ld + Z + j]);
        const float mq = to_f32(mlvq[(long)b * ld + j]), lq = to_f32(mlvq[(long)b * ld + Z + j]);
        z[(long)b * Z + j] = from_f32<T>(eps[(long)b * Z + j] * __expf(0.5f * lq) + mq);
        const float ip = __expf(-lp);
        s += 1.f + (lq - lp) - (mp - mq) * (mp - mq) * ip - __expf(lq) * ip;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(kld, -0.5f * s / (float)B);
}

template <typename T>
__global__ void latent_bwd_kernel(const T* __restrict__ mlvp, const T* __restrict__ mlvq, const float* __restrict__ eps,
                                  const T* __restrict__ dz, float G, T* __restrict__ dmlvp, T* __restrict__ dmlvq, int B, int Z, int ld) {
    const long total = (long)B * Z;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / Z;
        const int j = (int)(i - b * Z);
        const float mp = to_f32(mlvp[b * ld + j]), lp = to_f32(mlvp[b * ld + Z + j]);
        const float mq = to_f32(mlvq[b * ld + j]), lq = to_f32(mlvq[b * ld + Z + j]);
        const float g = to_f32(dz[i]);
        const float ip = __expf(-lp), r = __expf(lq) * ip, dm = mp - mq;
        dmlvq[b * ld + j] = from_f32<T>(g - G * dm * ip);
        dmlvq[b * ld + Z + j] = from_f32<T>(g * 0.5f * eps[i] * __expf(0.5f * lq) - 0.5f * G * (1.f - r));
        dmlvp[b * ld + j] = from_f32<T>(G * dm * ip);
        dmlvp[b * ld + Z + j] = from_f32<T>(-0.5f * G * (-1.f + dm * dm * ip + r));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// greedy decoding helpers (reference models/iq.py:117-152)
// ---------------------------------------------------------------------------------------------------------------
__global__ void prep_decode_kernel(const long long* __restrict__ ctx, int B, int Sa, int T, int* __restrict__ ids_all, int* __restrict__ pos_all,
                                   int* __restrict__ ctx32, int V, float* __restrict__ bad_ids) {
    const int na = B * Sa, nt = B * T;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na + nt; i += gridDim.x * blockDim.x) {
        if (i < na) {
            int v = (int)ctx[i];
            if (ctx[i] < 0 || ctx[i] >= (long long)V) { v = 0; if (bad_ids) atomicAdd(bad_ids, 1.f); }      // see prep_tokens_kernel
            ids_all[i] = v; ctx32[i] = v; pos_all[i] = i % Sa;
        } else {
            ids_all[i] = 0;                 // ys starts as <pad> everywhere (iq.py:129)
            pos_all[i] = (i - na) % T;
        }
    }
}

// One block per sample: argmax of the logits row (first maximum, like torch.max) and the 6 largest softmax probabilities
// (torch.topk order).  Writes the next input token ys[b, t+1].
template <typename T>
__global__ __launch_bounds__(256) void argmax_top6_kernel(const T* __restrict__ logits, int ld, int V, int t, int Tn, int* __restrict__ ys,
                                                         int* __restrict__ tokens, int* __restrict__ top_idx, float* __restrict__ top_val) {
    __shared__ float red[16];
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int picked[6];
    const int b = blockIdx.x;
    const T* x = logits + (long)b * ld;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += blockDim.x) m = fmaxf(m, to_f32(x[v]));
    m = block_max(m, red);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) s += __expf(to_f32(x[v]) - m);
    s = block_sum(s, red);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = 0; k < 6; ++k) {
        float best = -INFINITY;
        int bidx = 0x7fffffff;
        for (int v = threadIdx.x; v < V; v += blockDim.x) {
            bool used = false;
            for (int q = 0; q < k; ++q) used = used || (picked[q] == v);
            const float xv = to_f32(x[v]);
            if (!used && (xv > best || (xv == best && v < bidx))) { best = xv; bidx = v; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bidx, o, 64);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        __syncthreads();
        if (lane == 0) { bv[w] = best; bi[w] = bidx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int q = 1; q < 4; ++q)
                if (bv[q] > bv[0] || (bv[q] == bv[0] && bi[q] < bi[0])) { bv[0] = bv[q]; bi[0] = bi[q]; }
            picked[k] = bi[0];
            top_idx[((long)b * Tn + t) * 6 + k] = bi[0];
            top_val[((long)b * Tn + t) * 6 + k] = __expf(bv[0] - m) / s;
            if (k == 0) {
                tokens[(long)b * Tn + t] = bi[0];
                if (t + 1 < Tn) ys[(long)b * Tn + t + 1] = bi[0];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// optimiser
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    const long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per thread: the loop is bound by load latency, not by the adds
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        s += (a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) + (b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w) +
             (c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) + (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
    }
    for (; i < n4; i += stride) {
        const float4 v = x4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) { const float v = x[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// torch.nn.utils.clip_grad_norm_(max_norm) followed by torch.optim.Adam.step (defaults: amsgrad off, wd 0).  HBM-bound: 28 bytes per
// parameter (p, g, m, v read; p, m, v written) + 2 when `shadow` (the bf16 mirror the next forward's GEMMs read) is written in the same
// pass; 16-byte accesses (n4 = number of float4 groups, the flat buffers are 16-byte aligned and padded to a multiple of 8 floats).
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                  float4* __restrict__ v, long n4, const float* __restrict__ gnorm_sq, float max_norm,
                                                  float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt, s16x4* __restrict__ shadow) {
    float clip = 1.f;
    if (max_norm > 0.f) {
        const float norm = sqrtf(gnorm_sq[0]);
        clip = fminf(max_norm / (norm + 1e-6f), 1.f);
    }
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 g4 = g[i], m4 = m[i], v4 = v[i];
        float4 p4 = p[i];
        float gg[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = gg[e] * clip;
            mm[e] = beta1 * mm[e] + (1.f - beta1) * gi;
            vv[e] = beta2 * vv[e] + (1.f - beta2) * gi * gi;
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pp[e] -= step_size * (mm[e] / denom);
        }
        m[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        v[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        p[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        if (shadow) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = (bf16)pp[e];
            shadow[i] = __builtin_bit_cast(s16x4, b);
        }
    }
}

// scalar form for ragged / unaligned callers of the operator (the engine's flat buffers always take the vector form)
__global__ __launch_bounds__(256) void adam_scalar_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                                         const float* __restrict__ gnorm_sq, float max_norm, float lr, float beta1, float beta2, float eps,
                                                         float bc1, float bc2_sqrt, bf16* __restrict__ shadow) {
    float clip = 1.f;
    if (max_norm > 0.f) clip = fminf(max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f), 1.f);
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float pn = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        p[i] = pn;
        if (shadow) shadow[i] = (bf16)pn;
    }
}

// transposed bf16 shadows from the PLAIN bf16 shadow (which the optimiser pass above keeps current): same table as
// shadow_transpose_kernel, half the bytes (no fp32 read, no plain write)
__global__ __launch_bounds__(256) void shadow_transpose_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dstT, const int4* __restrict__ table,
                                                                   int nent, int tile_base) {
    // 64 x 64 tile through LDS; both sides move 4 bytes (2 bf16) per lane — a 2-byte-per-lane version ran at 1.5 TB/s
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][66];
    const int bid = (int)blockIdx.x + tile_base;      // (a launch may cover a sub-range of the table: `table` points at its first entry)
    int e = 0;
    while (e + 1 < nent && bid >= table[e + 1].w) ++e;
    const int4 t = table[e];
    if (t.z <= 0) return;                      // plain shadow only
    const int off = t.x, rows = t.y, cols = t.z;
    const int tiles_c = (cols + 63) >> 6;
    const int lt = bid - t.w;
    const int r0 = (lt / tiles_c) << 6, c0 = (lt % tiles_c) << 6;
    const unsigned short* s16 = reinterpret_cast<const unsigned short*>(src) + (size_t)off;
    unsigned short* d16 = reinterpret_cast<unsigned short*>(dstT) + (size_t)off;
    if (((rows | cols) & 7) == 0 && (((uintptr_t)s16 | (uintptr_t)d16) & 15) == 0) {
        // 16 bytes per lane on both sides (round 4; the 4-byte form below ran at 1.65 TB/s): a lane loads 8 columns of one source row, the
        // 64 x 64 tile sits in LDS with its 16-byte chunks XOR-swizzled by the 8-row group (the transposed read — 8 two-byte reads down a
        // column — is conflict-free: the 8 lanes of a destination line hit 8 different chunks), a lane stores 8 consecutive source rows of
        // one column = 16 contiguous bytes of the transposed row, 8 lanes = one 128-byte line
        unsigned short* tl = &tile[0][0];      // used as [64][64], no padding
        for (int q = threadIdx.x; q < 512; q += 256) {
            const int row = q >> 3, ch = q & 7;
            const int r = r0 + row, c = c0 + ch * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (r < rows && c < cols) v = *reinterpret_cast<const uint4*>(s16 + (size_t)r * cols + c);
            *reinterpret_cast<uint4*>(tl + row * 64 + ((ch ^ ((row >> 3) & 7)) << 3)) = v;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < 512; q += 256) {
            const int rch = q & 7, oc = q >> 3;
            const int c = c0 + oc, r = r0 + rch * 8;
            if (c >= cols || r >= rows) continue;
            const unsigned short* col = tl + (rch * 8) * 64 + ((((oc >> 3) ^ rch) & 7) << 3) + (oc & 7);
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (unsigned)col[(2 * k) * 64] | ((unsigned)col[(2 * k + 1) * 64] << 16);
            *reinterpret_cast<uint4*>(d16 + (size_t)c * rows + r) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 column pairs x 8 rows per pass
    const bool pair_in = ((cols & 1) == 0);                           // rows of the source start 4-byte aligned
    for (int i = ty; i < 64; i += 8) {
        const int r = r0 + i, c = c0 + 2 * tx;
        unsigned short a = 0, b = 0;
        if (r < rows) {
            if (pair_in && c + 1 < cols) {
                const unsigned v = *reinterpret_cast<const unsigned*>(s16 + (size_t)r * cols + c);
                a = (unsigned short)(v & 0xFFFFu); b = (unsigned short)(v >> 16);
            } else {
                if (c < cols) a = s16[(size_t)r * cols + c];
                if (c + 1 < cols) b = s16[(size_t)r * cols + c + 1];
            }
        }
        tile[i][2 * tx] = a; tile[i][2 * tx + 1] = b;
    }
    __syncthreads();
    const bool pair_out = ((rows & 1) == 0);
    for (int i = ty; i < 64; i += 8) {
        const int c = c0 + i, r = r0 + 2 * tx;
        if (c >= cols) continue;
        const unsigned short a = tile[2 * tx][i], b = tile[2 * tx + 1][i];
        if (pair_out && r + 1 < rows) *reinterpret_cast<unsigned*>(d16 + (size_t)c * rows + r) = (unsigned)a | ((unsigned)b << 16);
        else {
            if (r < rows) d16[(size_t)c * rows + r] = a;
            if (r + 1 < rows) d16[(size_t)c * rows + r + 1] = b;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Region-attention pooling of the bottom-up mode (bltvqg_config::region_pool = 1): one workgroup per sample.
//   s_r = w . tanh(p_r);  alpha = softmax_r(s);  out[h] = sum_r alpha_r p[r][h]        (p = projected regions [B, R, H])
// backward: dalpha_r = dout . p_r; ds = alpha * (dalpha - sum alpha dalpha); dp[r][h] = alpha_r dout[h] + ds_r w[h] (1 - tanh^2 p[r][h]);
//           dw[h] += sum_r ds_r tanh(p[r][h])
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void region_attn_fwd_kernel(const T* __restrict__ P, const float* __restrict__ w, float* __restrict__ out,
                                                             float* __restrict__ alpha, int R, int H) {
    __shared__ float sc[64];
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* p = P + (size_t)b * R * H;
    for (int r = wave; r < R; r += 4) {
        float acc = 0.f;
        for (int h = lane; h < H; h += 64) acc += w[h] * tanhf(to_f32(p[(size_t)r * H + h]));
        acc = wave_sum(acc);
        if (lane == 0) sc[r] = acc;
    }
    __syncthreads();
    float v = (tid < R) ? sc[tid] : -INFINITY;
    const float m = block_max(v, red);
    const float e = (tid < R) ? __expf(v - m) : 0.f;
    const float tot = block_sum(e, red);
    __syncthreads();
    if (tid < R) { sc[tid] = e / tot; alpha[(size_t)b * R + tid] = e / tot; }
    __syncthreads();
    for (int h = tid; h < H; h += 256) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += sc[r] * to_f32(p[(size_t)r * H + h]);
        out[(size_t)b * H + h] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void region_attn_bwd_kernel(const T* __restrict__ P, const float* __restrict__ w, const float* __restrict__ alpha,
                                                             const float* __restrict__ dout, T* __restrict__ dP, float* __restrict__ dw, int R, int H) {
    __shared__ float da[64], ds[64];
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* p = P + (size_t)b * R * H;
    const float* g = dout + (size_t)b * H;
    for (int r = wave; r < R; r += 4) {
        float acc = 0.f;
        for (int h = lane; h < H; h += 64) acc += g[h] * to_f32(p[(size_t)r * H + h]);
        acc = wave_sum(acc);
        if (lane == 0) da[r] = acc;
    }
    __syncthreads();
    const float a = (tid < R) ? alpha[(size_t)b * R + tid] : 0.f;
    const float dot = block_sum((tid < R) ? a * da[tid] : 0.f, red);
    __syncthreads();
    if (tid < R) { ds[tid] = a * (da[tid] - dot); da[tid] = a; }          // da now holds alpha
    __syncthreads();
    for (int h = tid; h < H; h += 256) {
        const float gh = g[h], wh = w[h];
        float dwh = 0.f;
        for (int r = 0; r < R; ++r) {
            const float t = tanhf(to_f32(p[(size_t)r * H + h]));
            dP[((size_t)b * R + r) * H + h] = from_f32<T>(da[r] * gh + ds[r] * wh * (1.f - t * t));
            dwh += ds[r] * t;
        }
        atomicAdd(dw + h, dwh);
    }
}

__global__ void dropout_mask_kernel(uint64_t seed, uint32_t stream_id, long rows, int cols, int ld_index, uint32_t thresh,
                                    unsigned char* __restrict__ out) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        out[i] = dropout_keep(seed, stream_id, (uint64_t)r * (uint64_t)ld_index + (uint64_t)c, thresh) ? 1 : 0;   // ld_index % 8 == 0 for GEMM sites
    }
}


// rows_add with the row statistics of the RESULT (one wave per row): stat[b * stat_stride + 0..1] = {sum, sum of squares} of row b as
// stored — the row-0 injections change rows whose LayerNorm is folded into the Linear that consumes it (GemmArgs::fold_stat), so the
// sums the embedding GEMM left for those rows are replaced here
template <typename T>
__global__ __launch_bounds__(256) void rows_add_stat_kernel(T* __restrict__ y, long ys, const T* __restrict__ a, long as, const T* __restrict__ c, long cs,
                                                            int B, int n, int accumulate, float* __restrict__ stat, long stat_stride, int np) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < n; j += 64) {
        float v = to_f32(a[(long)b * as + j]);
        if (c != nullptr) v += to_f32(c[(long)b * cs + j]);
        if (accumulate) v += to_f32(y[(long)b * ys + j]);
        const T o = from_f32<T>(v);
        y[(long)b * ys + j] = o;
        const float x = to_f32(o);
        s1 += x; s2 += x * x;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    // slot 0 of the row's partial-sum slots holds the whole row, the other np - 1 slots (the producer GEMM's column tiles) are zeroed
    if (lane < 2 * np) stat[(long)b * stat_stride + lane] = lane == 0 ? s1 : (lane == 1 ? s2 : 0.f);
}

// LayerNorm folded into its consumer Linear (gemm2.hip, GemmArgs::fold_*): per output row n of the weight W [rows, K] (fp32 master)
//   W'[n, k] = bf16(W[n, k] * gamma[k])        the GEMM's B operand
//   s[n]     = sum_k float(W'[n, k])           (of the ROUNDED operand: what the MFMA multiplies the mean with)
//   c[n]     = sum_k beta[k] * W[n, k] + bias[n]
// One wave per row, all entries of a table in one launch.
struct FoldEnt { long w_off, g_off, b_off, bias_off; int rows, K, srow, row0; };
__device__ __forceinline__ void fold_row(const float* __restrict__ w, const float* __restrict__ g, const float* __restrict__ bt, bf16* __restrict__ o,
                                         int K, int lane, float& s_out, float& c_out) {
    float s = 0.f, c = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        float wv[8], gv[8], bv[8], ov[8];
        Vec8<float>::load(w + k, wv);
        Vec8<float>::load(g + k, gv);
        Vec8<float>::load(bt + k, bv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16 q = (bf16)(wv[i] * gv[i]);
            ov[i] = (float)q;
            s += ov[i];
            c += bv[i] * wv[i];
        }
        Vec8<bf16>::store(o + k, ov);
    }
    s_out = wave_sum(s); c_out = wave_sum(c);
}
__global__ __launch_bounds__(256) void ln_fold_prepare_kernel(const float* __restrict__ train, bf16* __restrict__ wfold, float* __restrict__ fs,
                                                              float* __restrict__ fc, const FoldEnt* __restrict__ tab, int nent, int total_rows) {
    const int lane = threadIdx.x & 63;
    const int R = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (R >= total_rows) return;
    int e = 0;
    while (e + 1 < nent && tab[e + 1].row0 <= R) ++e;
    const FoldEnt t = tab[e];
    const int r = R - t.row0;
    float s, c;
    fold_row(train + t.w_off + (long)r * t.K, train + t.g_off, train + t.b_off, wfold + t.w_off + (long)r * t.K, t.K, lane, s, c);
    if (lane == 0) {
        fs[t.srow + r] = s;
        fc[t.srow + r] = c + (t.bias_off >= 0 ? train[t.bias_off + r] : 0.f);
    }
}
// one weight, explicit pointers (the C-ABI operator bltvqg_ln_fold_prepare)
__global__ __launch_bounds__(256) void ln_fold_prepare_one_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ bias, bf16* __restrict__ Wf, float* __restrict__ fs,
                                                                  float* __restrict__ fc, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    float s, c;
    fold_row(W + (long)r * K, gamma, beta, Wf + (long)r * K, K, lane, s, c);
    if (lane == 0) { fs[r] = s; fc[r] = c + (bias ? bias[r] : 0.f); }
}

}  // namespace

#define T_SWITCH(dtype, NAME, GRID, BLOCK, LDS, STREAM, ...)                                        \
    do {                                                                                           \
        if ((dtype) == BLT_F32) hipLaunchKernelGGL(NAME<float>, GRID, BLOCK, LDS, STREAM, __VA_ARGS__); \
        else hipLaunchKernelGGL(NAME<bf16>, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                 \
    } while (0)

#define CHECK_DTYPE(dtype, what) BLT_REQUIRE((dtype) == BLT_F32 || (dtype) == BLT_BF16, what ": bad dtype %d", (int)(dtype))

int blt_prep_tokens(const long long* ctx, const long long* post, const long long* tgt, int B, int Sa, int Sp, int T,
                    int* ids_all, int* pos_all, int* tgt_shift, int* tgt32, int* ctx32, int* post32, float* counters, int V,
                    float* bad_ids, hipStream_t s) {
    BLT_REQUIRE(ctx && post && tgt && ids_all && pos_all && tgt_shift && tgt32 && ctx32 && post32 && counters, "prep_tokens: null pointer");
    BLT_REQUIRE(B > 0 && Sa > 0 && Sp > 0 && T > 0 && T <= 64 && V > 0, "prep_tokens: bad sizes");
    hipLaunchKernelGGL(prep_tokens_kernel, dim3(ew_grid((long)B * (Sa + Sp + T))), dim3(256), 0, s, ctx, post, tgt, B, Sa, Sp, T,
                       ids_all, pos_all, tgt_shift, tgt32, ctx32, post32, counters, V, bad_ids);
    return blt_check_launch("prep_tokens");
}

int blt_embed_gather(int dtype, const float* table, const int* ids, void* out, long rows, int E, int ld, hipStream_t s) {
    CHECK_DTYPE(dtype, "embed_gather");
    BLT_REQUIRE(table && ids && out && rows > 0 && E > 0 && ld >= E, "embed_gather: bad args");
    if (dtype == BLT_F32) hipLaunchKernelGGL(embed_gather_kernel<float>, dim3(ew_grid(rows * ld)), dim3(256), 0, s, table, ids, (float*)out, rows, E, ld);
    else hipLaunchKernelGGL(embed_gather_kernel<bf16>, dim3(ew_grid(rows * ld)), dim3(256), 0, s, table, ids, (bf16*)out, rows, E, ld);
    return blt_check_launch("embed_gather");
}

int blt_embed_scatter(int dtype, const void* d, int ld, const int* ids, float* dtable, long rows, int E, int pad_id, hipStream_t s) {
    CHECK_DTYPE(dtype, "embed_scatter");
    BLT_REQUIRE(d && ids && dtable && rows > 0 && E > 0 && ld >= E, "embed_scatter: bad args");
    if (dtype == BLT_F32) hipLaunchKernelGGL(embed_scatter_kernel<float>, dim3(ew_grid(rows * E)), dim3(256), 0, s, (const float*)d, ld, ids, dtable, rows, E, pad_id);
    else hipLaunchKernelGGL(embed_scatter_kernel<bf16>, dim3(ew_grid(rows * E)), dim3(256), 0, s, (const bf16*)d, ld, ids, dtable, rows, E, pad_id);
    return blt_check_launch("embed_scatter");
}

int blt_rows_add(int dtype, void* y, long ys, const void* a, long as, const void* c, long cs, int B, int n, int accumulate, hipStream_t s) {
    CHECK_DTYPE(dtype, "rows_add");
    BLT_REQUIRE(y && a && B > 0 && n > 0, "rows_add: bad args");
    if (dtype == BLT_F32) hipLaunchKernelGGL(rows_add_kernel<float>, dim3(ew_grid((long)B * n)), dim3(256), 0, s, (float*)y, ys, (const float*)a, as, (const float*)c, cs, B, n, accumulate);
    else hipLaunchKernelGGL(rows_add_kernel<bf16>, dim3(ew_grid((long)B * n)), dim3(256), 0, s, (bf16*)y, ys, (const bf16*)a, as, (const bf16*)c, cs, B, n, accumulate);
    return blt_check_launch("rows_add");
}

int blt_rows_add_stat(int dtype, void* y, long ys, const void* a, long as, const void* c, long cs, int B, int n, int accumulate, float* stat,
                      long stat_stride, int np, hipStream_t s) {
    CHECK_DTYPE(dtype, "rows_add_stat");
    BLT_REQUIRE(y && a && stat && B > 0 && n > 0 && np >= 1 && np <= 32 && stat_stride >= 2 * np, "rows_add_stat: bad args");
    if (dtype == BLT_F32) hipLaunchKernelGGL(rows_add_stat_kernel<float>, dim3(cdiv(B, 4)), dim3(256), 0, s, (float*)y, ys, (const float*)a, as, (const float*)c, cs, B, n, accumulate, stat, stat_stride, np);
    else hipLaunchKernelGGL(rows_add_stat_kernel<bf16>, dim3(cdiv(B, 4)), dim3(256), 0, s, (bf16*)y, ys, (const bf16*)a, as, (const bf16*)c, cs, B, n, accumulate, stat, stat_stride, np);
    return blt_check_launch("rows_add_stat");
}

int blt_ln_fold_prepare(const float* train, void* wfold_bf16, float* fold_s, float* fold_c, const void* table_dev, int nent, int total_rows, hipStream_t s) {
    BLT_REQUIRE(train && wfold_bf16 && fold_s && fold_c && table_dev && nent > 0 && total_rows > 0, "ln_fold_prepare: bad args");
    hipLaunchKernelGGL(ln_fold_prepare_kernel, dim3(cdiv(total_rows, 4)), dim3(256), 0, s, train, (bf16*)wfold_bf16, fold_s, fold_c, (const FoldEnt*)table_dev, nent, total_rows);
    return blt_check_launch("ln_fold_prepare");
}

int blt_ln_fold_prepare_one(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf_bf16, float* fold_s, float* fold_c, int N,
                            int K, hipStream_t s) {
    BLT_REQUIRE(W && gamma && beta && Wf_bf16 && fold_s && fold_c && N > 0 && K > 0 && K % 8 == 0, "ln_fold_prepare: bad args (K %% 8 == 0)");
    BLT_REQUIRE(((uintptr_t)W % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 && ((uintptr_t)Wf_bf16 % 16) == 0,
                "ln_fold_prepare: operands must be 16-byte aligned");
    hipLaunchKernelGGL(ln_fold_prepare_one_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, W, gamma, beta, bias, (bf16*)Wf_bf16, fold_s, fold_c, N, K);
    return blt_check_launch("ln_fold_prepare");
}

int blt_row0_sums(int dtype, const void* dx0, long sdx, const void* g_rin, const void* g_zc, void* d_feats, void* d_zproj, void* d_enc, long senc,
                  int B, int n, hipStream_t s) {
    CHECK_DTYPE(dtype, "row0_sums");
    BLT_REQUIRE(dx0 && d_feats && (g_rin == nullptr || d_enc != nullptr) && B > 0 && n > 0, "row0_sums: bad args");
    if (dtype == BLT_F32)
        hipLaunchKernelGGL(row0_sums_kernel<float>, dim3(ew_grid((long)B * n)), dim3(256), 0, s, (const float*)dx0, sdx, (const float*)g_rin,
                           (const float*)g_zc, (float*)d_feats, (float*)d_zproj, (float*)d_enc, senc, B, n);
    else
        hipLaunchKernelGGL(row0_sums_kernel<bf16>, dim3(ew_grid((long)B * n)), dim3(256), 0, s, (const bf16*)dx0, sdx, (const bf16*)g_rin,
                           (const bf16*)g_zc, (bf16*)d_feats, (bf16*)d_zproj, (bf16*)d_enc, senc, B, n);
    return blt_check_launch("row0_sums");
}

int blt_mask_scale(int dtype, const void* dy, const void* ym, void* y, long n, float scale, hipStream_t s) {
    CHECK_DTYPE(dtype, "mask_scale");
    BLT_REQUIRE(dy && ym && y && n > 0 && n % 8 == 0, "mask_scale: n=%ld must be a positive multiple of 8", n);
    if (dtype == BLT_F32) hipLaunchKernelGGL(mask_scale_kernel<float>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (const float*)dy, (const float*)ym, (float*)y, n / 8, scale);
    else hipLaunchKernelGGL(mask_scale_kernel<bf16>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (const bf16*)dy, (const bf16*)ym, (bf16*)y, n / 8, scale);
    return blt_check_launch("mask_scale");
}

int blt_colsum(int dtype, const void* x, int ld, long M, int N, float* out, int accumulate, hipStream_t s) {
    CHECK_DTYPE(dtype, "colsum");
    BLT_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, "colsum: bad args");
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s) != hipSuccess) { blt_set_error("colsum: memset failed"); return BLT_ERR_HIP; }
    }
    int gy = (int)((M + 31) / 32);      // ~8 rows per thread: the loop is latency-bound, not bandwidth-bound
    if (gy > 256) gy = 256;
    if (gy < 1) gy = 1;
    if (dtype == BLT_F32) hipLaunchKernelGGL(colsum_kernel<float>, dim3(cdiv(N, 64), gy), dim3(256), 0, s, (const float*)x, ld, M, N, out);
    else hipLaunchKernelGGL(colsum_kernel<bf16>, dim3(cdiv(N, 64), gy), dim3(256), 0, s, (const bf16*)x, ld, M, N, out);
    return blt_check_launch("colsum");
}

static int cast_rows_impl(int dtype_src, const void* src, int lds_, int dtype_dst, void* dst, int ldd, long rows, int cols, int width, hipStream_t s) {
    CHECK_DTYPE(dtype_src, "cast_rows");
    CHECK_DTYPE(dtype_dst, "cast_rows");
    BLT_REQUIRE(src && dst && rows > 0 && cols > 0 && lds_ >= cols && ldd >= cols, "cast_rows: bad args");
    const dim3 g(ew_grid(rows * width)), b(256);
    if (dtype_src == BLT_F32 && dtype_dst == BLT_F32) hipLaunchKernelGGL((cast_rows_kernel<float, float>), g, b, 0, s, (const float*)src, lds_, (float*)dst, ldd, rows, cols, width);
    else if (dtype_src == BLT_F32) hipLaunchKernelGGL((cast_rows_kernel<float, bf16>), g, b, 0, s, (const float*)src, lds_, (bf16*)dst, ldd, rows, cols, width);
    else if (dtype_dst == BLT_F32) hipLaunchKernelGGL((cast_rows_kernel<bf16, float>), g, b, 0, s, (const bf16*)src, lds_, (float*)dst, ldd, rows, cols, width);
    else hipLaunchKernelGGL((cast_rows_kernel<bf16, bf16>), g, b, 0, s, (const bf16*)src, lds_, (bf16*)dst, ldd, rows, cols, width);
    return blt_check_launch("cast_rows");
}

// table_dev points at the FIRST entry of the range to process, nent entries, whose tiles are [tile_base, tile_base + total_tiles)
int blt_shadow_transpose(const float* src, void* dst_bf16, void* dstT_bf16, const void* table_dev, int nent, int total_tiles, hipStream_t s, int tile_base) {
    BLT_REQUIRE(src && dstT_bf16 && table_dev && nent > 0 && total_tiles > 0 && tile_base >= 0, "shadow_transpose: bad args");
    hipLaunchKernelGGL(shadow_transpose_kernel, dim3(total_tiles), dim3(256), 0, s, src, (bf16*)dst_bf16, (bf16*)dstT_bf16, (const int4*)table_dev, nent, tile_base);
    return blt_check_launch("shadow_transpose");
}

// dst[r, 0:cols] = cast(src[r, 0:cols]); dst[r, cols:ldd] = 0
int blt_cast_rows(int dtype_src, const void* src, int lds_, int dtype_dst, void* dst, int ldd, long rows, int cols, hipStream_t s) {
    return cast_rows_impl(dtype_src, src, lds_, dtype_dst, dst, ldd, rows, cols, ldd, s);
}

int blt_cast_pad(const float* src, int rows, int cols, void* dst, int ld, int dtype, hipStream_t s) {
    return blt_cast_rows(BLT_F32, src, cols, dtype, dst, ld, rows, cols, s);
}

// strided 2-D copy: only columns [0, cols) of each destination row are written
int blt_copy2d(int dtype, const void* src, int lds_, void* dst, int ldd, long rows, int cols, hipStream_t s) {
    return cast_rows_impl(dtype, src, lds_, dtype, dst, ldd, rows, cols, cols, s);
}

int blt_img_pack(int dtype, const float* nchw, void* nhwc, int N, int C, int H, int W, int Cpad, int pt, int pl, int Hp, int Wp, hipStream_t s) {
    CHECK_DTYPE(dtype, "img_pack");
    BLT_REQUIRE(nchw && nhwc && N > 0 && C > 0 && C <= Cpad && H > 0 && W > 0 && pt >= 0 && pl >= 0 && Hp >= H + pt && Wp >= W + pl, "img_pack: bad args");
    const long n = (long)N * Hp * Wp;
    if (C == 3 && Cpad == 4 && (long)N * Hp < (1l << 31) && ((uintptr_t)nhwc % 16) == 0) {
        if (dtype == BLT_BF16 && W % 4 == 0 && Wp % 2 == 0 && Wp <= 256 && ((uintptr_t)nchw % 16) == 0 && blt_debug_get(27) != 1) {      // (key 27 = 1: A/B, the per-pixel form)
            hipLaunchKernelGGL(img_pack_c3v_kernel, dim3((unsigned)(((long)N * Hp + 3) / 4)), dim3(256), 0, s, nchw, (bf16*)nhwc, N, H, W, pt, pl, Hp, Wp);
            return blt_check_launch("img_pack");
        }
        if (dtype == BLT_F32) hipLaunchKernelGGL(img_pack_c3_kernel<float>, dim3((unsigned)(N * Hp)), dim3(256), 0, s, nchw, (float*)nhwc, H, W, pt, pl, Hp, Wp);
        else hipLaunchKernelGGL(img_pack_c3_kernel<bf16>, dim3((unsigned)(N * Hp)), dim3(256), 0, s, nchw, (bf16*)nhwc, H, W, pt, pl, Hp, Wp);
        return blt_check_launch("img_pack");
    }
    if (dtype == BLT_F32) hipLaunchKernelGGL(img_pack_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, s, nchw, (float*)nhwc, N, C, H, W, Cpad, pt, pl, Hp, Wp);
    else hipLaunchKernelGGL(img_pack_kernel<bf16>, dim3(ew_grid(n)), dim3(256), 0, s, nchw, (bf16*)nhwc, N, C, H, W, Cpad, pt, pl, Hp, Wp);
    return blt_check_launch("img_pack");
}

int blt_conv_pack_w(int dtype, const float* w, void* out, int Cout, int Cin, int KH, int KW, int Cpad, int KWpad, hipStream_t s) {
    CHECK_DTYPE(dtype, "conv_pack_w");
    BLT_REQUIRE(w && out && Cout > 0 && Cin > 0 && Cin <= Cpad && KWpad >= KW, "conv_pack_w: bad args");
    const long n = (long)Cout * KH * KWpad * Cpad;
    if (dtype == BLT_F32) hipLaunchKernelGGL(conv_pack_w_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, s, w, (float*)out, Cout, Cin, KH, KW, Cpad, KWpad);
    else hipLaunchKernelGGL(conv_pack_w_kernel<bf16>, dim3(ew_grid(n)), dim3(256), 0, s, w, (bf16*)out, Cout, Cin, KH, KW, Cpad, KWpad);
    return blt_check_launch("conv_pack_w");
}

#define ROWS_DISPATCH(KERN, T_, nch_, grid_, ...)                                                                      \
    do {                                                                                                              \
        if (nch_ <= 1) hipLaunchKernelGGL((KERN<T_, 1>), dim3(grid_), dim3(256), 0, s, __VA_ARGS__);                   \
        else if (nch_ <= 2) hipLaunchKernelGGL((KERN<T_, 2>), dim3(grid_), dim3(256), 0, s, __VA_ARGS__);              \
        else if (nch_ <= 4) hipLaunchKernelGGL((KERN<T_, 4>), dim3(grid_), dim3(256), 0, s, __VA_ARGS__);              \
        else if (nch_ <= 8) hipLaunchKernelGGL((KERN<T_, 8>), dim3(grid_), dim3(256), 0, s, __VA_ARGS__);              \
        else hipLaunchKernelGGL((KERN<T_, 16>), dim3(grid_), dim3(256), 0, s, __VA_ARGS__);                            \
    } while (0)

int blt_ce_fwd_bwd(int dtype, void* logits, int ld, const int* target, long M, int V, const float* count, float gscale,
                   float* loss_out, int write_grad, hipStream_t s) {
    CHECK_DTYPE(dtype, "ce");
    BLT_REQUIRE(logits && target && count && loss_out && M > 0 && V > 0 && ld >= V, "ce: bad args");
    const int nch = cdiv(ld / 8, 256);       // chunks of 8 columns per thread
    if (ld % 8 == 0 && nch <= 16 && ((uintptr_t)logits % 16) == 0) {
        const unsigned grid = (unsigned)cdiv(M, CE_ROWS_PER_BLOCK);
        if (dtype == BLT_F32) ROWS_DISPATCH(ce_rows_kernel, float, nch, grid, (float*)logits, ld, target, M, V, count, gscale, loss_out, write_grad);
        else ROWS_DISPATCH(ce_rows_kernel, bf16, nch, grid, (bf16*)logits, ld, target, M, V, count, gscale, loss_out, write_grad);
        return blt_check_launch("ce");
    }
    if (dtype == BLT_F32) hipLaunchKernelGGL(ce_kernel<float>, dim3((unsigned)M), dim3(256), 0, s, (float*)logits, ld, target, V, count, gscale, loss_out, write_grad);
    else hipLaunchKernelGGL(ce_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, s, (bf16*)logits, ld, target, V, count, gscale, loss_out, write_grad);
    return blt_check_launch("ce");
}

int blt_bow_ce_fwd_bwd(int dtype, const void* z, int ld, const int* target, int B, int T, int V, const float* count, float gscale,
                       float* loss_out, void* dz, hipStream_t s) {
    CHECK_DTYPE(dtype, "bow_ce");
    BLT_REQUIRE(z && target && count && loss_out && B > 0 && T > 0 && T <= 64 && V > 0 && ld >= V, "bow_ce: bad args");
    const int nch = cdiv(ld / 8, 256);
    if (ld % 8 == 0 && nch <= 16 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)dz % 16) == 0) {
        if (dtype == BLT_F32) ROWS_DISPATCH(bow_ce_rows_kernel, float, nch, B, (const float*)z, ld, target, T, V, count, gscale, loss_out, (float*)dz);
        else ROWS_DISPATCH(bow_ce_rows_kernel, bf16, nch, B, (const bf16*)z, ld, target, T, V, count, gscale, loss_out, (bf16*)dz);
        return blt_check_launch("bow_ce");
    }
    if (dtype == BLT_F32) hipLaunchKernelGGL(bow_ce_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)z, ld, target, T, V, count, gscale, loss_out, (float*)dz);
    else hipLaunchKernelGGL(bow_ce_kernel<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)z, ld, target, T, V, count, gscale, loss_out, (bf16*)dz);
    return blt_check_launch("bow_ce");
}

int blt_mse_fwd_bwd(int dtype, const void* a, const void* b, long n, float gscale, float* loss_out, void* da, void* db, hipStream_t s, long n_div) {
    if (n_div <= 0) n_div = n;
    CHECK_DTYPE(dtype, "mse");
    BLT_REQUIRE(a && b && loss_out && n > 0, "mse: bad args");
    if (dtype == BLT_F32) hipLaunchKernelGGL(mse_kernel<float>, dim3(ew_grid(n, 1024)), dim3(256), 0, s, (const float*)a, (const float*)b, n, gscale, loss_out, (float*)da, (float*)db, n_div);
    else hipLaunchKernelGGL(mse_kernel<bf16>, dim3(ew_grid(n, 1024)), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, n, gscale, loss_out, (bf16*)da, (bf16*)db, n_div);
    return blt_check_launch("mse");
}

int blt_latent_fwd(int dtype, const void* mlv_p, const void* mlv_q, const float* eps, void* z, float* kld_out, int B, int Z, int ld, hipStream_t s) {
    CHECK_DTYPE(dtype, "latent_fwd");
    BLT_REQUIRE(mlv_p && mlv_q && eps && z && kld_out && B > 0 && Z > 0 && ld >= 2 * Z, "latent_fwd: bad args");
    if (dtype == BLT_F32) hipLaunchKernelGGL(latent_fwd_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)mlv_p, (const float*)mlv_q, eps, (float*)z, kld_out, B, Z, ld);
    else hipLaunchKernelGGL(latent_fwd_kernel<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)mlv_p, (const bf16*)mlv_q, eps, (bf16*)z, kld_out, B, Z, ld);
    return blt_check_launch("latent_fwd");
}

int blt_latent_bwd(int dtype, const void* mlv_p, const void* mlv_q, const float* eps, const void* dz, float kld_gscale,
                   void* dmlv_p, void* dmlv_q, int B, int Z, int ld, hipStream_t s) {
    CHECK_DTYPE(dtype, "latent_bwd");
    BLT_REQUIRE(mlv_p && mlv_q && eps && dz && dmlv_p && dmlv_q && B > 0 && Z > 0 && ld >= 2 * Z, "latent_bwd: bad args");
    const float G = kld_gscale / (float)B;
    if (dtype == BLT_F32) hipLaunchKernelGGL(latent_bwd_kernel<float>, dim3(ew_grid((long)B * Z)), dim3(256), 0, s, (const float*)mlv_p, (const float*)mlv_q, eps, (const float*)dz, G, (float*)dmlv_p, (float*)dmlv_q, B, Z, ld);
    else hipLaunchKernelGGL(latent_bwd_kernel<bf16>, dim3(ew_grid((long)B * Z)), dim3(256), 0, s, (const bf16*)mlv_p, (const bf16*)mlv_q, eps, (const bf16*)dz, G, (bf16*)dmlv_p, (bf16*)dmlv_q, B, Z, ld);
    return blt_check_launch("latent_bwd");
}

int blt_prep_decode(const long long* ctx, int B, int Sa, int T, int* ids_all, int* pos_all, int* ctx32, int V, float* bad_ids, hipStream_t s) {
    BLT_REQUIRE(ctx && ids_all && pos_all && ctx32 && B > 0 && Sa > 0 && T > 0 && V > 0, "prep_decode: bad args");
    hipLaunchKernelGGL(prep_decode_kernel, dim3(ew_grid((long)B * (Sa + T))), dim3(256), 0, s, ctx, B, Sa, T, ids_all, pos_all, ctx32, V, bad_ids);
    return blt_check_launch("prep_decode");
}

int blt_argmax_top6(int dtype, const void* logits, int ld, int B, int V, int t, int T, int* ys, int* tokens, int* top_idx, float* top_val,
                    hipStream_t s) {
    CHECK_DTYPE(dtype, "argmax_top6");
    BLT_REQUIRE(logits && ys && tokens && top_idx && top_val && B > 0 && V >= 6 && ld >= V && t >= 0 && t < T, "argmax_top6: bad args");
    if (dtype == BLT_F32) hipLaunchKernelGGL(argmax_top6_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)logits, ld, V, t, T, ys, tokens, top_idx, top_val);
    else hipLaunchKernelGGL(argmax_top6_kernel<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)logits, ld, V, t, T, ys, tokens, top_idx, top_val);
    return blt_check_launch("argmax_top6");
}

int blt_sumsq(const float* x, long n, float* out, hipStream_t s) {
    BLT_REQUIRE(x && out && n > 0 && ((uintptr_t)x % 16) == 0, "sumsq: bad args");
    // every block ends with one float atomic on the same address and those serialise (~15 ns each): 512 blocks keep that tail under
    // the time the loads take, and still have 8 MB of 16-byte loads in flight
    int grid = ew_grid(n / 4 + 1, 1024);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, x, n, out);
    return blt_check_launch("sumsq");
}

int blt_adam_step(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq, float max_norm, float lr,
                  float beta1, float beta2, float eps, int step, hipStream_t s, void* shadow_bf16) {
    BLT_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam: bad args");
    BLT_REQUIRE(max_norm <= 0.f || gnorm_sq != nullptr, "adam: clipping needs gnorm_sq");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) == 0 && ((uintptr_t)shadow_bf16 % 8) == 0;
    const long nvec = vec ? n / 4 * 4 : 0;
    if (nvec > 0)
        hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(nvec / 4, 2048)), dim3(256), 0, s, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, nvec / 4,
                           gnorm_sq, max_norm, lr, beta1, beta2, eps, bc1, bc2_sqrt, (s16x4*)shadow_bf16);
    if (nvec < n)
        hipLaunchKernelGGL(adam_scalar_kernel, dim3(ew_grid(n - nvec, 1024)), dim3(256), 0, s, p + nvec, g + nvec, m + nvec, v + nvec, n - nvec, gnorm_sq,
                           max_norm, lr, beta1, beta2, eps, bc1, bc2_sqrt, shadow_bf16 ? (bf16*)shadow_bf16 + nvec : nullptr);
    return blt_check_launch("adam");
}

int blt_shadow_transpose_bf16(const void* src_bf16, void* dstT_bf16, const void* table_dev, int nent, int total_tiles, hipStream_t s, int tile_base) {
    BLT_REQUIRE(src_bf16 && dstT_bf16 && table_dev && nent > 0 && total_tiles > 0 && tile_base >= 0, "shadow_transpose_bf16: bad args");
    hipLaunchKernelGGL(shadow_transpose_bf16_kernel, dim3(total_tiles), dim3(256), 0, s, (const bf16*)src_bf16, (bf16*)dstT_bf16, (const int4*)table_dev, nent, tile_base);
    return blt_check_launch("shadow_transpose_bf16");
}

int blt_dropout_mask(uint64_t seed, uint32_t stream_id, long rows, int cols, int ld_index, float p, unsigned char* out, hipStream_t s) {
    BLT_REQUIRE(out && rows > 0 && cols > 0 && ld_index >= cols && p >= 0.f && p < 1.f, "dropout_mask: bad args");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, s, seed, stream_id, rows, cols, ld_index, dropout_threshold(p), out);
    return blt_check_launch("dropout_mask");
}

int blt_region_attn_fwd(int dtype, const void* P, const float* w, float* out, float* alpha, int B, int R, int H, hipStream_t s) {
    CHECK_DTYPE(dtype, "region_attn_fwd");
    BLT_REQUIRE(P && w && out && alpha && B > 0 && R > 0 && R <= 64 && H > 0, "region_attn_fwd: bad args (regions <= 64)");
    if (dtype == BLT_F32) hipLaunchKernelGGL(region_attn_fwd_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)P, w, out, alpha, R, H);
    else hipLaunchKernelGGL(region_attn_fwd_kernel<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)P, w, out, alpha, R, H);
    return blt_check_launch("region_attn_fwd");
}
int blt_region_attn_bwd(int dtype, const void* P, const float* w, const float* alpha, const float* dout, void* dP, float* dw, int B, int R, int H,
                        hipStream_t s) {
    CHECK_DTYPE(dtype, "region_attn_bwd");
    BLT_REQUIRE(P && w && alpha && dout && dP && dw && B > 0 && R > 0 && R <= 64 && H > 0, "region_attn_bwd: bad args");
    if (dtype == BLT_F32) hipLaunchKernelGGL(region_attn_bwd_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)P, w, alpha, dout, (float*)dP, dw, R, H);
    else hipLaunchKernelGGL(region_attn_bwd_kernel<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)P, w, alpha, dout, (bf16*)dP, dw, R, H);
    return blt_check_launch("region_attn_bwd");
}

// ---- hardware-id probe: which XCD / CU a stream's workgroups land on (tests of the CU partition, bltvqg_engine_set_cu_masks) ----
#ifdef BLT_EXPERIMENTS
namespace {
__global__ void __launch_bounds__(64) hw_id_probe_kernel(int* __restrict__ out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that a launch of many workgroups spreads over every CU the stream may use
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (int)hw;
        out[2 * blockIdx.x + 1] = (int)(xcc & 0xF);
    }
}
}  // namespace
int blt_hw_id_probe(int* out, int n_wg, int spin_ticks, hipStream_t s) {
    BLT_REQUIRE(out && n_wg > 0 && n_wg <= 65536 && spin_ticks >= 0 && spin_ticks <= 10000000, "hw_id_probe: bad args");
    hipLaunchKernelGGL(hw_id_probe_kernel, dim3((unsigned)n_wg), dim3(64), 0, s, out, spin_ticks);
    return blt_check_launch("hw_id_probe");
}
#endif      // BLT_EXPERIMENTS
