// LayerNorm (fwd/bwd), BatchNorm2d train-mode statistics/apply for the frozen ResNet stack, BatchNorm1d (fwd/bwd),
// max-pool and global average pool.  All are HBM-bound: one wave per row (LayerNorm) or 8 channels per lane (NHWC
// elementwise), 16-byte vector accesses, wave-shuffle reductions.
//
// Replaces: torch LayerNorm (models/transformer_layers.py:134,202,256-257,320-322), torchvision BatchNorm2d in train
// mode + MaxPool2d + AdaptiveAvgPool2d (models/encoder_cnn.py:17,33), BatchNorm1d(momentum=0.01) (encoder_cnn.py:21,34).
#include "kernels.h"

namespace {

constexpr int LN_MAX_CHUNKS = 4;   // per lane: 4 chunks x 8 elements x 64 lanes = 2048 columns

template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y,
                                                           float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                           int cols, float eps, int pad_period, int pad_valid, long ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = cols >> 3;
    // Padded rows (the reference's default widths, e.g. 4 heads of 75 stored as 4 x 80): column c is a real feature iff
    // c % pad_period < pad_valid; pad columns hold zeros, carry gamma = beta = 0 and count neither in the mean nor in the variance
    const float n_true = pad_period ? (float)(cols / pad_period * pad_valid) : (float)cols;
    unsigned vm[LN_MAX_CHUNKS];
#pragma unroll
    for (int j = 0; j < LN_MAX_CHUNKS; ++j) {
        vm[j] = 0xFFu;
        if (pad_period) {
            vm[j] = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) vm[j] |= ((((lane + 64 * j) * 8 + e) % pad_period) < pad_valid ? 1u : 0u) << e;
        }
    }
    float v[LN_MAX_CHUNKS][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_CHUNKS; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) {
            Vec8<T>::load(x + row * ld + c * 8, v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        }
    }
    const float mu = wave_sum(s) / n_true;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_CHUNKS; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = ((vm[j] >> e) & 1u) ? v[j][e] - mu : 0.f; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / n_true + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < LN_MAX_CHUNKS; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) {
            float g[8], b[8], o[8];
            Vec8<float>::load(gamma + c * 8, g);
            Vec8<float>::load(beta + c * 8, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mu) * rs * g[e] + b[e];
            Vec8<T>::store(y + row * ld + c * 8, o);
        }
    }
}

// Each wave walks rows with a grid stride, keeps its slice of dgamma/dbeta in registers and adds it to the
// global fp32 gradient once at the end (one float atomic per column per block).  Rows of <= 256 columns need only 32 lanes
// (8 columns each), so a wave then carries TWO rows at once (`lpr` lanes per row) — the kernel is latency-bound, rows in flight
// are what count.
// NCH = chunk groups per lane (cols <= 512 NCH), R = rows a wave keeps in flight per iteration: every load of the R rows is issued
// before the first reduction, so a wave pays the memory round trip once per R rows (with R = 1 and ~2.5 serial iterations per wave the
// [5120 x 512] launch took 13.4 us against 3 us of traffic).
template <typename T, int NCH, int R, int NW = 4>
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const T* dres,
                                                           T* dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, long rows, int cols,
                                                           const T* __restrict__ maskY, float mask_scale, T* __restrict__ out2,
                                                           float* __restrict__ partials, int pad_period, int pad_valid, long ld,
                                                           const float* __restrict__ beta, T* __restrict__ xn_out) {
    const int lane = threadIdx.x & 63;
    const int nch = cols >> 3;
    const int lpr = (nch <= 32) ? 32 : 64, rpw = 64 / lpr;
    const int l = lane & (lpr - 1), sub = lane / lpr;
    const float n_true = pad_period ? (float)(cols / pad_period * pad_valid) : (float)cols;      // see layernorm_fwd_kernel
    float ag[NCH][8], ab[NCH][8], g[NCH][8];
    unsigned vm[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = l + lpr * j;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; g[j][e] = 0.f; }
        if (c < nch) Vec8<float>::load(gamma + c * 8, g[j]);
        vm[j] = 0xFFu;
        if (pad_period) {
            vm[j] = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) vm[j] |= (((c * 8 + e) % pad_period) < pad_valid ? 1u : 0u) << e;
        }
    }
    const long rstride = (long)gridDim.x * NW * rpw;
    for (long row0 = ((long)blockIdx.x * NW + (threadIdx.x >> 6)) * rpw + sub; row0 < rows; row0 += rstride * R) {
        float d[R][NCH][8], xh[R][NCH][8], rr[R][NCH][8];
        float mu[R], rs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = row0 + r * rstride;
            const bool rok = row < rows;
            mu[r] = rok ? mean[row] : 0.f;
            rs[r] = rok ? rstd[row] : 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = l + lpr * j;
                if (rok && c < nch) {
                    Vec8<T>::load(dy + row * ld + c * 8, d[r][j]);
                    Vec8<T>::load(x + row * ld + c * 8, xh[r][j]);
                    if (dres != nullptr) Vec8<T>::load(dres + row * ld + c * 8, rr[r][j]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { d[r][j][e] = 0.f; xh[r][j][e] = 0.f; }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = row0 + r * rstride;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xn = (xh[r][j][e] - mu[r]) * rs[r];
                    const float dg = d[r][j][e] * g[j][e];
                    s1 += dg;
                    s2 += dg * xn;
                    ag[j][e] += d[r][j][e] * xn;
                    ab[j][e] += d[r][j][e];
                    xh[r][j][e] = xn;
                    d[r][j][e] = dg;
                }
            s1 = group_sum(s1, lpr); s2 = group_sum(s2, lpr);
            const float c1 = s1 / n_true, c2 = s2 / n_true;
            if (row < rows) {
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int c = l + lpr * j;
                    if (c < nch) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = ((vm[j] >> e) & 1u) ? rs[r] * (d[r][j][e] - c1 - xh[r][j][e] * c2) : 0.f;
                        if (dres != nullptr) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] += rr[r][j][e];
                        }
                        Vec8<T>::store(dx + row * ld + c * 8, o);
                        if (xn_out != nullptr) {
                            // the LayerNorm's OUTPUT, as its forward launch would have stored it ((x - mean) * rstd * gamma + beta, rounded to
                            // T): with the LayerNorm folded into the Linear that consumes it nothing wrote it in forward, and the weight
                            // gradient of that Linear (deferred, behind this launch) reads it
                            float bt[8], xo[8];
                            Vec8<float>::load(beta + c * 8, bt);
#pragma unroll
                            for (int e = 0; e < 8; ++e) xo[e] = ((vm[j] >> e) & 1u) ? xh[r][j][e] * g[j][e] + bt[e] : 0.f;      // (pad columns: zeros, as the forward launch stores them)
                            Vec8<T>::store(xn_out + row * ld + c * 8, xo);
                        }
                        if (out2 != nullptr) {
                            // the consumer of dx is a ReLU + dropout backward (the FFN output of the next layer down): emit its masked,
                            // rescaled gradient here instead of in a launch of its own
                            float mk[8];
                            Vec8<T>::load(maskY + row * ld + c * 8, mk);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (mk[e] != 0.f) ? o[e] * mask_scale : 0.f;
                            Vec8<T>::store(out2 + row * ld + c * 8, o);
                        }
                    }
                }
            }
        }
    }
    if (lpr == 32) {
        // both half-waves hold partial sums of the same 256 columns (chunk group 0); after the exchange lanes >= 32 carry
        // duplicates that land beyond `cols` in the pass below and are dropped there
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[0][e] += __shfl_xor(ag[0][e], 32, 64); ab[0][e] += __shfl_xor(ab[0][e], 32, 64); }
    }
    // combine the 4 waves of the block through LDS, then one partial row (or one atomic) per column per block
    __shared__ float red[2][NW][512];   // [gamma|beta][wave][512 columns of one chunk group]
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (64 * 8 * j >= cols) break;
        if (j > 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[0][w][lane * 8 + e] = ag[j][e]; red[1][w][lane * 8 + e] = ab[j][e]; }
        __syncthreads();
        // 512 columns of chunk-group j: column = (lane' + 64 j) * 8 + e  -> index lane'*8+e in red
        for (int i = threadIdx.x; i < 1024; i += NW * 64) {
            const int pass = i >> 9, ii = i & 511;
            const int col = 64 * 8 * j + ii;
            if (col < cols) {
                float t = 0.f;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) t += red[pass][wv][ii];
                // partials: one row of dgamma and one of dbeta per workgroup, summed later by ln_param_reduce_kernel (off the
                // dependent chain) instead of gridDim.x same-address atomics per column at the tail of this launch
                if (partials != nullptr) partials[((size_t)blockIdx.x * 2 + pass) * cols + col] = t;
                else atomicAdd((pass == 0 ? dgamma : dbeta) + col, t);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm2d, train mode.  Stage 1 reduces the GEMM's per-half-tile partial sums (fp32) into `S` slices in fp64;
// stage 2 finishes per channel and applies the reference's running-statistics update (momentum 0.1, unbiased var).
// (A single-launch variant with a device-scope ticket counter was measured 2.4x SLOWER: the release fence writes back the whole
// XCD L2, which is full of freshly written conv output at that point.)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ psum, const float* __restrict__ psq,
                                                       int nparts, int C, double* __restrict__ tmp /* [S][2][C] */) {
    __shared__ double sh[2][4][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int S = gridDim.y, sl = blockIdx.y;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int part = sl * 4 + py; part < nparts; part += S * 4) {
            a += (double)psum[(size_t)part * C + c];
            b += (double)psq[(size_t)part * C + c];
        }
    sh[0][py][cx] = a;
    sh[1][py][cx] = b;
    __syncthreads();
    if (py == 0 && c < C) {
        tmp[((size_t)sl * 2 + 0) * C + c] = sh[0][0][cx] + sh[0][1][cx] + sh[0][2][cx] + sh[0][3][cx];
        tmp[((size_t)sl * 2 + 1) * C + c] = sh[1][0][cx] + sh[1][1][cx] + sh[1][2][cx] + sh[1][3][cx];
    }
}

__global__ __launch_bounds__(256) void bn_finish_kernel(const double* __restrict__ tmp, int S, int C, double count, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float momentum, float* __restrict__ rmean,
                                                        float* __restrict__ rvar, float* __restrict__ scale, float* __restrict__ shift,
                                                        float* __restrict__ save_mean, float* __restrict__ save_var) {
    // block = 8 channels x 32 slice-lanes: the S (= 32) partial sums of a channel are loaded in parallel and shuffled together
    const int sl = threadIdx.x & 31;
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int s = sl; s < S; s += 32) { a += tmp[((size_t)s * 2) * C + c]; b += tmp[((size_t)s * 2 + 1) * C + c]; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 32); b += __shfl_xor(b, o, 32); }
    if (c >= C || sl != 0) return;
    const double mu = a / count;
    double var = b / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
    scale[c] = sc;
    shift[c] = beta[c] - (float)mu * sc;
    if (save_mean) save_mean[c] = (float)mu;
    if (save_var) save_var[c] = (float)var;
    if (rmean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// Single-launch variant: the slice sums meet in fp64 accumulators through device-scope atomics (no fence: every word that crosses
// workgroups is only ever touched by atomics, so nothing depends on L2 write-back), a ticket counter finds the last workgroup of a
// channel block, which swaps the totals out (leaving zeros for the next call) and finishes.  The fp64 additions commute up to the
// last bit of a double, i.e. far below the fp32 scale / shift that leave this kernel.
__global__ __launch_bounds__(256) void bn_stats_ticket_kernel(const float* __restrict__ psum, const float* __restrict__ psq, int nparts, int C,
                                                             double* acc /* [2][C], zero between calls */, int* counters, double count,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                             float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                                             float* __restrict__ scale, float* __restrict__ shift,
                                                             float* __restrict__ save_mean, float* __restrict__ save_var) {
    __shared__ double sh[2][4][64];
    __shared__ int is_last;
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int S = gridDim.y, sl = blockIdx.y;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int part = sl * 4 + py; part < nparts; part += S * 4) {
            a += (double)psum[(size_t)part * C + c];
            b += (double)psq[(size_t)part * C + c];
        }
    sh[0][py][cx] = a;
    sh[1][py][cx] = b;
    __syncthreads();
    if (py == 0) {       // wave 0: one pair of atomics per channel, then (same wave, after they are acknowledged) the ticket
        if (c < C) {
            __hip_atomic_fetch_add(acc + c, sh[0][0][cx] + sh[0][1][cx] + sh[0][2][cx] + sh[0][3][cx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(acc + C + c, sh[1][0][cx] + sh[1][1][cx] + sh[1][2][cx] + sh[1][3][cx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (cx == 0) is_last = (__hip_atomic_fetch_add(counters + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1);
    }
    __syncthreads();
    if (!is_last || py != 0) return;
    if (cx == 0) __hip_atomic_store(counters + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c >= C) return;
    a = __hip_atomic_exchange(acc + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b = __hip_atomic_exchange(acc + C + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double mu = a / count;
    double var = b / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
    scale[c] = sc;
    shift[c] = beta[c] - (float)mu * sc;
    if (save_mean) save_mean[c] = (float)mu;
    if (save_var) save_var[c] = (float)var;
    if (rmean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

__global__ void bn_eval_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rmean,
                                     const float* __restrict__ rvar, float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const T* __restrict__ res,
                                                      T* __restrict__ y, long nchunks, int cpr /* chunks per row */, int relu,
                                                      int ppH, int ppW, const float* __restrict__ rscale, const float* __restrict__ rshift) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cpr;
        const int c = (int)(i - row * cpr) * 8;
        float v[8], sc[8], sh[8];
        if (ppW > 0) {
            // padded-pitch layout (kernels.h): pad positions hold whatever the convolution computed there -> write the zeros that the
            // next convolution reads as its padding
            const long r2 = row / (ppW + 1);
            if ((int)(row - r2 * (ppW + 1)) == ppW || (int)(r2 % (ppH + 1)) == ppH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
                Vec8<T>::store(y + i * 8, v);
                continue;
            }
        }
        Vec8<T>::load(x + i * 8, v);
        Vec8<float>::load(scale + c, sc);
        Vec8<float>::load(shift + c, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (res != nullptr) {
            float r[8];
            Vec8<T>::load(res + i * 8, r);
            if (rscale != nullptr) {      // the residual is itself a raw convolution output (downsample branch): its BatchNorm on the fly
                float rs[8], rh[8];
                Vec8<float>::load(rscale + c, rs);
                Vec8<float>::load(rshift + c, rh);
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = r[e] * rs[e] + rh[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        Vec8<T>::store(y + i * 8, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ y, int N,
                                                             int Hi, int Wi, int C, int Ho, int Wo, int pp /* 1: padded-pitch output */) {
    const int cpr = C >> 3;
    const long total = (long)N * Ho * Wo * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cpr);
        long t = i / cpr;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float sc[8], sh[8], m[8];
        Vec8<float>::load(scale + cc * 8, sc);
        Vec8<float>::load(shift + cc * 8, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = 0.f;   // relu output >= 0 and every 3x3/2 pad-1 window holds >= 1 valid pixel
        for (int r = 0; r < 3; ++r) {
            const int hi = ho * 2 - 1 + r;
            if ((unsigned)hi >= (unsigned)Hi) continue;
            for (int s = 0; s < 3; ++s) {
                const int wi = wo * 2 - 1 + s;
                if ((unsigned)wi >= (unsigned)Wi) continue;
                float v[8];
                Vec8<T>::load(x + (((long)n * Hi + hi) * Wi + wi) * C + cc * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e] * sc[e] + sh[e]);
            }
        }
        Vec8<T>::store(y + ((((long)n * (Ho + pp) + ho) * (Wo + pp) + wo) * cpr + cc) * 8, m);
    }
}

// mean over `count` of the HW positions of each image (padded-pitch input: HW = (H+1)(W+1) positions of which the H*W real ones are
// non-zero)
template <typename T, typename TO>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, TO* __restrict__ y, int N, int HW, int C, int count) {
    // four adjacent lanes share one (image, 8-channel chunk) and split its positions: 4x the loads in flight of a thread-per-chunk walk
    // (this kernel is pure load latency: 49-64 positions per output)
    const int cpr = C >> 3;
    const long total = (long)N * cpr;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i = gid >> 2;
    const int part = (int)(gid & 3);
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    const bool ok = i < total;
    const int cc = ok ? (int)(i % cpr) : 0;
    const long n = ok ? i / cpr : 0;
    if (ok) {
#pragma unroll 4
        for (int p = part; p < HW; p += 4) {
            float v[8];
            Vec8<T>::load(x + (n * HW + p) * C + cc * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] += __shfl_xor(a[e], 1, 64); a[e] += __shfl_xor(a[e], 2, 64); }
    if (ok && part == 0) {
        const float inv = 1.f / (float)count;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= inv;
        Vec8<TO>::store(y + n * C + cc * 8, a);
    }
}

// BatchNorm1d over the batch dimension (B <= 16 * BN1D_RPT rows): block = 16 columns x 16 row lanes, every thread keeps its rows of a
// column in registers, so the tensor is read once with all loads in flight (the kernels are pure latency: [128, 256] tensors) and the
// batch reductions go through LDS.
constexpr int BN1D_RPT = 32;
template <typename T>
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                       float* __restrict__ rmean, float* __restrict__ rvar, int B, int C, float eps, float momentum) {
    __shared__ float sh[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const bool ok = c < C;
    float v[BN1D_RPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BN1D_RPT; ++i) {
        const int b = ry + 16 * i;
        v[i] = (ok && b < B) ? to_f32(x[(long)b * C + c]) : 0.f;
        s += v[i];
    }
    sh[ry][cx] = s;
    __syncthreads();
    float mu = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) mu += sh[r][cx];
    mu /= (float)B;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < BN1D_RPT; ++i)
        if (ry + 16 * i < B) { const float d = v[i] - mu; q += d * d; }
    sh[ry][cx] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) var += sh[r][cx];
    var /= (float)B;
    const float rs = rsqrtf(var + eps);
    if (!ok) return;
    const float g = gamma[c], be = beta[c];
#pragma unroll
    for (int i = 0; i < BN1D_RPT; ++i) {
        const int b = ry + 16 * i;
        if (b < B) y[(long)b * C + c] = from_f32<T>((v[i] - mu) * rs * g + be);
    }
    if (ry == 0) {
        mean[c] = mu;
        rstd[c] = rs;
        if (rmean) {
            const float unb = B > 1 ? var * (float)B / (float)(B - 1) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C) {
    __shared__ float sh[2][16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const bool ok = c < C;
    const float mu = ok ? mean[c] : 0.f, rs = ok ? rstd[c] : 0.f, g = ok ? gamma[c] : 0.f;
    float d[BN1D_RPT], xh[BN1D_RPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < BN1D_RPT; ++i) {
        const int b = ry + 16 * i;
        const bool in = ok && b < B;
        d[i] = in ? to_f32(dy[(long)b * C + c]) : 0.f;
        xh[i] = in ? (to_f32(x[(long)b * C + c]) - mu) * rs : 0.f;
        s1 += d[i];
        s2 += d[i] * xh[i];
    }
    sh[0][ry][cx] = s1;
    sh[1][ry][cx] = s2;
    __syncthreads();
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1 += sh[0][r][cx]; s2 += sh[1][r][cx]; }
    if (!ok) return;
    if (ry == 0) { dgamma[c] = s2; dbeta[c] = s1; }
    const float inv = 1.f / (float)B;
#pragma unroll
    for (int i = 0; i < BN1D_RPT; ++i) {
        const int b = ry + 16 * i;
        if (b < B) dx[(long)b * C + c] = from_f32<T>(g * rs * (d[i] - s1 * inv - xh[i] * s2 * inv));
    }
}

inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define DISPATCH_T(dtype, expr_f32, expr_bf16) \
    do {                                       \
        if ((dtype) == BLT_F32) { expr_f32; }  \
        else { expr_bf16; }                    \
    } while (0)

int blt_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      long rows, int cols, float eps, hipStream_t s, int pad_period, int pad_valid, long ld) {
    if (ld == 0) ld = cols;
    BLT_REQUIRE(ld >= cols && ld % 8 == 0, "layernorm_fwd: row stride %ld must be a multiple of 8 and >= cols", ld);
    BLT_REQUIRE((pad_period == 0 && pad_valid == 0) || (pad_period > 0 && pad_valid > 0 && pad_valid <= pad_period && cols % pad_period == 0),
                "layernorm_fwd: bad pad pattern %d / %d for %d columns", pad_valid, pad_period, cols);
    BLT_REQUIRE(dtype == BLT_F32 || dtype == BLT_BF16, "layernorm_fwd: bad dtype");
    BLT_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= LN_MAX_CHUNKS * 512, "layernorm_fwd: cols=%d must be a multiple of 8 and <= %d", cols, LN_MAX_CHUNKS * 512);
    BLT_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    const int grid = cdiv(rows, 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, rows, cols, eps, pad_period, pad_valid, ld),
               hipLaunchKernelGGL(layernorm_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, rows, cols, eps, pad_period, pad_valid, ld));
    return blt_check_launch("layernorm_fwd");
}

static inline int ln_bwd_rows_in_flight(int cols) { return cols <= 512 ? 4 : (cols <= 1024 ? 2 : 1); }
// workgroup form for cols <= 512 (debug key 24, A/B): 0 / 1 = 4 waves x 4 rows in flight, 2 = 8 waves x 2 rows, 3 = 16 waves x 1 row,
// 4 = 8 waves x 4 rows (half the workgroups), 5 = 4 waves x 2 rows (twice the workgroups)
static inline void ln_bwd_form(int cols, int* nw, int* r) {
    *nw = 4; *r = ln_bwd_rows_in_flight(cols);
    if (cols > 512) return;
    switch (blt_debug_get(24)) {
        case 2: *nw = 8; *r = 2; break;
        case 3: *nw = 16; *r = 1; break;
        case 4: *nw = 8; *r = 4; break;
        case 5: *nw = 4; *r = 2; break;
        default: break;
    }
}

int blt_layernorm_bwd_grid(long rows, int cols) {
    const int rpw = (cols <= 256) ? 2 : 1;
    int nw, r;
    ln_bwd_form(cols, &nw, &r);
    int grid = cdiv(rows, nw * rpw * r);      // one iteration per wave where that fits in the cap
    if (grid > 1024) grid = 1024;
    return grid < 1 ? 1 : grid;
}

// dgamma[c] += sum_b part[(b*2+0)*cols + c], dbeta[c] += sum_b part[(b*2+1)*cols + c] for up to BLT_LN_RED_MAX LayerNorms per launch
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const LnRedArgs a) {
    const LnRed e = a.e[blockIdx.x];
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= 2 * e.cols) return;
    const int pass = j / e.cols, col = j - pass * e.cols;
    const float* p = e.part + (size_t)pass * e.cols + col;
    const size_t stride = (size_t)2 * e.cols;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < e.nblocks; b += 4) {
        s0 += p[(size_t)b * stride]; s1 += p[(size_t)(b + 1) * stride]; s2 += p[(size_t)(b + 2) * stride]; s3 += p[(size_t)(b + 3) * stride];
    }
    for (; b < e.nblocks; ++b) s0 += p[(size_t)b * stride];
    float* dst = (pass == 0 ? e.dgamma : e.dbeta) + col;
    *dst += (s0 + s1) + (s2 + s3);
}

int blt_ln_param_reduce(const LnRedArgs& a, hipStream_t s) {
    BLT_REQUIRE(a.n > 0 && a.n <= BLT_LN_RED_MAX, "ln_param_reduce: bad count");
    int maxc = 0;
    for (int i = 0; i < a.n; ++i) {
        BLT_REQUIRE(a.e[i].part && a.e[i].dgamma && a.e[i].dbeta && a.e[i].nblocks > 0 && a.e[i].cols > 0, "ln_param_reduce: bad entry");
        if (a.e[i].cols > maxc) maxc = a.e[i].cols;
    }
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(a.n, cdiv(2 * maxc, 256)), dim3(256), 0, s, a);
    return blt_check_launch("ln_param_reduce");
}

int blt_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const void* dres, void* dx, float* dgamma, float* dbeta, long rows, int cols, hipStream_t s,
                      const void* maskY, float mask_scale, void* out2, float* partials, int pad_period, int pad_valid, long ld,
                      const float* beta, void* xn_out) {
    if (ld == 0) ld = cols;
    BLT_REQUIRE(xn_out == nullptr || beta != nullptr, "layernorm_bwd: xn_out needs beta");
    BLT_REQUIRE(ld >= cols && ld % 8 == 0, "layernorm_bwd: row stride %ld must be a multiple of 8 and >= cols", ld);
    BLT_REQUIRE((maskY == nullptr) == (out2 == nullptr), "layernorm_bwd: maskY and out2 go together");
    BLT_REQUIRE((pad_period == 0 && pad_valid == 0) || (pad_period > 0 && pad_valid > 0 && pad_valid <= pad_period && cols % pad_period == 0),
                "layernorm_bwd: bad pad pattern %d / %d for %d columns", pad_valid, pad_period, cols);
    BLT_REQUIRE(dtype == BLT_F32 || dtype == BLT_BF16, "layernorm_bwd: bad dtype");
    BLT_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= LN_MAX_CHUNKS * 512, "layernorm_bwd: bad cols=%d", cols);
    BLT_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
    int grid = blt_layernorm_bwd_grid(rows, cols);
    if (grid < 1) grid = 1;
    int nw_, r_;
    ln_bwd_form(cols, &nw_, &r_);
#define LN_BWD_LAUNCH(T_, NCH_, R_, NW_)                                                                                                         \
    hipLaunchKernelGGL((layernorm_bwd_kernel<T_, NCH_, R_, NW_>), dim3(grid), dim3(NW_ * 64), 0, s, (const T_*)dy, (const T_*)x, gamma, mean, rstd, \
                       (const T_*)dres, (T_*)dx, dgamma, dbeta, rows, cols, (const T_*)maskY, mask_scale, (T_*)out2, partials, pad_period, pad_valid, ld, beta, (T_*)xn_out)
#define LN_BWD_BY_COLS(T_)                                                                                                                 \
    do {                                                                                                                                    \
        if (cols <= 512) {                                                                                                                  \
            if (nw_ == 8 && r_ == 2) LN_BWD_LAUNCH(T_, 1, 2, 8);                                                                            \
            else if (nw_ == 16) LN_BWD_LAUNCH(T_, 1, 1, 16);                                                                                \
            else if (nw_ == 8) LN_BWD_LAUNCH(T_, 1, 4, 8);                                                                                  \
            else if (r_ == 2) LN_BWD_LAUNCH(T_, 1, 2, 4);                                                                                   \
            else LN_BWD_LAUNCH(T_, 1, 4, 4);                                                                                                \
        } else if (cols <= 1024) LN_BWD_LAUNCH(T_, 2, 2, 4);                                                                                \
        else LN_BWD_LAUNCH(T_, 4, 1, 4);                                                                                                    \
    } while (0)
    if (dtype == BLT_F32) LN_BWD_BY_COLS(float);
    else LN_BWD_BY_COLS(bf16);
#undef LN_BWD_BY_COLS
#undef LN_BWD_LAUNCH
    return blt_check_launch("layernorm_bwd");
}

static constexpr int BN_MAX_SLICES = 512;
int blt_bn_scratch_doubles(int C) { return BN_MAX_SLICES * 2 * C; }

int blt_bn_finalize(const float* psum, const float* psq, int nparts, int C, long count, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                    float* shift, float* save_mean, float* save_var, double* scratch, hipStream_t s) {
    BLT_REQUIRE(psum && psq && gamma && beta && scale && shift && scratch, "bn_finalize: null pointer");
    BLT_REQUIRE(nparts > 0 && C > 0 && count > 0, "bn_finalize: bad sizes");
    BLT_REQUIRE(((uintptr_t)scratch % 8) == 0, "bn_finalize: scratch must be 8-byte aligned");
    // one slice per ~64 partial rows (32..512 slices, a multiple of 32): enough workgroups for the 25 k partials of the stem conv
    int slices = (nparts / 64 + 31) / 32 * 32;
    if (slices < 32) slices = 32;
    if (slices > BN_MAX_SLICES) slices = BN_MAX_SLICES;
    if (blt_debug_get(7) != 2) {      // default: single launch (atomics + ticket), measured -40 us per step; debug key 7 = 2 selects the two-launch form
        if (slices > 128) slices = 128;
        int* counters = (int*)(scratch + 2 * (size_t)C);
        hipLaunchKernelGGL(bn_stats_ticket_kernel, dim3(cdiv(C, 64), slices), dim3(256), 0, s, psum, psq, nparts, C, scratch, counters, (double)count,
                           gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_var);
        return blt_check_launch("bn_finalize");
    }
    hipLaunchKernelGGL(bn_reduce_kernel, dim3(cdiv(C, 64), slices), dim3(256), 0, s, psum, psq, nparts, C, scratch);
    hipLaunchKernelGGL(bn_finish_kernel, dim3(cdiv(C, 8)), dim3(256), 0, s, (const double*)scratch, slices, C, (double)count, gamma, beta, eps,
                       momentum, running_mean, running_var, scale, shift, save_mean, save_var);
    return blt_check_launch("bn_finalize");
}

// eval-mode BatchNorm: y = x*scale + shift with scale = gamma/sqrt(running_var + eps), shift = beta - running_mean*scale
int blt_bn_eval_scale(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, float* scale, float* shift, int C,
                      hipStream_t s) {
    BLT_REQUIRE(gamma && beta && rmean && rvar && scale && shift && C > 0, "bn_eval_scale: bad args");
    hipLaunchKernelGGL(bn_eval_scale_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, gamma, beta, rmean, rvar, eps, scale, shift, C);
    return blt_check_launch("bn_eval_scale");
}

int blt_bn_apply(int dtype, const void* x, const float* scale, const float* shift, const void* res, void* y, long rows,
                 int C, int relu, hipStream_t s) {
    BLT_REQUIRE(x && scale && shift && y && rows > 0 && C % 8 == 0, "bn_apply: bad args (C=%d)", C);
    const long n = rows * (C / 8);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, s, (const float*)x, scale, shift, (const float*)res, (float*)y, n, C / 8, relu, 0, 0, nullptr, nullptr),
               hipLaunchKernelGGL(bn_apply_kernel<bf16>, dim3(ew_grid(n)), dim3(256), 0, s, (const bf16*)x, scale, shift, (const bf16*)res, (bf16*)y, n, C / 8, relu, 0, 0, nullptr, nullptr));
    return blt_check_launch("bn_apply");
}

// x, res, y in the padded-pitch layout [N][H+1][W+1][C]; pad positions of y are written as zeros
int blt_bn_apply_pp(int dtype, const void* x, const float* scale, const float* shift, const void* res, const float* res_scale,
                    const float* res_shift, void* y, int N, int H, int W, int C, int relu, hipStream_t s) {
    BLT_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (res != nullptr || res_scale == nullptr), "bn_apply_pp: bad residual args");
    BLT_REQUIRE(x && scale && shift && y && N > 0 && H > 0 && W > 0 && C % 8 == 0, "bn_apply_pp: bad args (C=%d)", C);
    const long n = (long)N * (H + 1) * (W + 1) * (C / 8);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, s, (const float*)x, scale, shift, (const float*)res, (float*)y, n, C / 8, relu, H, W, res_scale, res_shift),
               hipLaunchKernelGGL(bn_apply_kernel<bf16>, dim3(ew_grid(n)), dim3(256), 0, s, (const bf16*)x, scale, shift, (const bf16*)res, (bf16*)y, n, C / 8, relu, H, W, res_scale, res_shift));
    return blt_check_launch("bn_apply_pp");
}

static int bn_relu_maxpool_impl(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi,
                                int C, int pp, hipStream_t s) {
    BLT_REQUIRE(x && scale && shift && y && C % 8 == 0 && N > 0, "bn_relu_maxpool: bad args");
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const long n = (long)N * Ho * Wo * (C / 8);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, s, (const float*)x, scale, shift, (float*)y, N, Hi, Wi, C, Ho, Wo, pp),
               hipLaunchKernelGGL(bn_relu_maxpool_kernel<bf16>, dim3(ew_grid(n)), dim3(256), 0, s, (const bf16*)x, scale, shift, (bf16*)y, N, Hi, Wi, C, Ho, Wo, pp));
    return blt_check_launch("bn_relu_maxpool");
}
int blt_bn_relu_maxpool(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi,
                        int C, hipStream_t s) {
    return bn_relu_maxpool_impl(dtype, x, scale, shift, y, N, Hi, Wi, C, 0, s);
}
// y in the padded-pitch layout [N][Ho+1][Wo+1][C]: only the real pixels are written (the pads keep their zeros)
int blt_bn_relu_maxpool_pp(int dtype, const void* x, const float* scale, const float* shift, void* y, int N, int Hi, int Wi,
                           int C, hipStream_t s) {
    return bn_relu_maxpool_impl(dtype, x, scale, shift, y, N, Hi, Wi, C, 1, s);
}

static int avgpool_impl(int dtype, const void* x, void* y, int N, int HW, int C, int count, int out_f32, hipStream_t s) {
    BLT_REQUIRE(x && y && C % 8 == 0 && N > 0 && HW > 0 && count > 0, "avgpool: bad args");
    const long n = (long)N * (C / 8) * 4;           // four lanes per output chunk
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (dtype == BLT_F32) hipLaunchKernelGGL((avgpool_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, N, HW, C, count);
    else if (out_f32) hipLaunchKernelGGL((avgpool_kernel<bf16, float>), dim3(grid), dim3(256), 0, s, (const bf16*)x, (float*)y, N, HW, C, count);
    else hipLaunchKernelGGL((avgpool_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, N, HW, C, count);
    return blt_check_launch("avgpool");
}
int blt_avgpool(int dtype, const void* x, void* y, int N, int HW, int C, int out_f32, hipStream_t s) {
    return avgpool_impl(dtype, x, y, N, HW, C, HW, out_f32, s);
}
// x in the padded-pitch layout [N][H+1][W+1][C] with zero pads
int blt_avgpool_pp(int dtype, const void* x, void* y, int N, int H, int W, int C, int out_f32, hipStream_t s) {
    return avgpool_impl(dtype, x, y, N, (H + 1) * (W + 1), C, H * W, out_f32, s);
}

int blt_bn1d_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                 float* running_mean, float* running_var, int B, int C, float eps, float momentum, hipStream_t s) {
    BLT_REQUIRE(x && gamma && beta && y && mean && rstd && B > 0 && C > 0, "bn1d_fwd: bad args");
    BLT_REQUIRE(B <= 16 * BN1D_RPT, "bn1d_fwd: batch %d > %d", B, 16 * BN1D_RPT);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn1d_fwd_kernel<float>, dim3(cdiv(C, 16)), dim3(256), 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, running_mean, running_var, B, C, eps, momentum),
               hipLaunchKernelGGL(bn1d_fwd_kernel<bf16>, dim3(cdiv(C, 16)), dim3(256), 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, running_mean, running_var, B, C, eps, momentum));
    return blt_check_launch("bn1d_fwd");
}

int blt_bn1d_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                 void* dx, float* dgamma, float* dbeta, int B, int C, hipStream_t s) {
    BLT_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && B > 0 && C > 0, "bn1d_bwd: bad args");
    BLT_REQUIRE(B <= 16 * BN1D_RPT, "bn1d_bwd: batch %d > %d", B, 16 * BN1D_RPT);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn1d_bwd_kernel<float>, dim3(cdiv(C, 16)), dim3(256), 0, s, (const float*)dy, (const float*)x, gamma, mean, rstd, (float*)dx, dgamma, dbeta, B, C),
               hipLaunchKernelGGL(bn1d_bwd_kernel<bf16>, dim3(cdiv(C, 16)), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (bf16*)dx, dgamma, dbeta, B, C));
    return blt_check_launch("bn1d_bwd");
}
