// Training-side normalisation / pooling / resampling kernels (fp32, channels-last):
//   BatchNorm2d in training mode (batch statistics, running-stat update) forward + backward
//     - torchvision ResNet-50 and utils/layers.py:76-100 Bottleneck_Conv under graph.train()
//   GroupNorm backward            - timm ResNetV2 (GroupNormAct) inside DPT-hybrid
//   3x3/s2 max-pool backward, global-mean backward, x2 bilinear (align_corners=True) backward
//   NHWC -> NCHW with a per-pixel mask (adjoint of zs_nchw_to_nhwc)
// Reductions are two-stage with double accumulators and a fixed order: deterministic.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace {

inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }
inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int BLOCK>
__device__ __forceinline__ float block_sum(float v, float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = lds[0];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; w++) r += lds[w];
    return r;
}

// ---------------- BatchNorm (training) ----------------
// column partials over row chunks: MODE 0: (sum x, sum x^2); MODE 1: (sum g, sum g*xhat) with
// g = dy (masked by y > 0 when y is given), xhat = (x - mean) * rstd
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 64 channels per workgroup as 16 channel quads (float4 loads) x 16 row lanes
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                         const float *__restrict__ y, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, double *__restrict__ partial,
                                                         int rows, int C, int rows_per_chunk) {
    __shared__ double lds[2][16][65];
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4, c = blockIdx.x * 64 + 4 * cq;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (c < C) {                                   // C % 4 == 0: a quad is in range as a whole
        f32x4 mu = {0, 0, 0, 0}, rs = {0, 0, 0, 0};
        if (MODE) {
            mu = *reinterpret_cast<const f32x4 *>(mean + c);
            rs = *reinterpret_cast<const f32x4 *>(rstd + c);
        }
        for (int r = r0 + ry; r < r1; r += 16) {
            const size_t o = (size_t)r * C + c;
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + o);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; e++) { a[e] += xv[e]; b[e] += (double)xv[e] * xv[e]; }
            } else {
                f32x4 g = *reinterpret_cast<const f32x4 *>(dy + o);
                if (y) {
                    const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + o);
#pragma unroll
                    for (int e = 0; e < 4; e++) g[e] = yv[e] > 0.f ? g[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; e++) { a[e] += g[e]; b[e] += (double)g[e] * ((xv[e] - mu[e]) * rs[e]); }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) { lds[0][ry][4 * cq + e] = a[e]; lds[1][ry][4 * cq + e] = b[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, col = threadIdx.x & 63, cc = blockIdx.x * 64 + col;
        if (cc < C) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++) t += lds[which][k][col];
            partial[((size_t)blockIdx.y * 2 + which) * C + cc] = t;
        }
    }
}

// sum of the per-chunk double partials of 64 columns: the chunks strided over 4 lanes, fixed order
__device__ __forceinline__ void bn_sum_partials(const double *__restrict__ partial, int chunks, int C, int c, int ry,
                                                double (*lds)[4][64], double &s, double &q) {
    const int cx = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int k = ry; k < chunks; k += 4) {
            a += partial[((size_t)k * 2 + 0) * C + c];
            b += partial[((size_t)k * 2 + 1) * C + c];
        }
    lds[0][ry][cx] = a;
    lds[1][ry][cx] = b;
    __syncthreads();
    s = (lds[0][0][cx] + lds[0][1][cx]) + (lds[0][2][cx] + lds[0][3][cx]);
    q = (lds[1][0][cx] + lds[1][1][cx]) + (lds[1][2][cx] + lds[1][3][cx]);
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const double *__restrict__ partial, int chunks, int C, int rows,
                                                       float eps, float momentum, float *__restrict__ running_mean,
                                                       float *__restrict__ running_var, float *__restrict__ save_mean,
                                                       float *__restrict__ save_rstd) {
    __shared__ double lds[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ry = threadIdx.x >> 6;
    double s, q;
    bn_sum_partials(partial, chunks, C, c, ry, lds, s, q);
    if (ry != 0 || c >= C) return;
    const double mean = s / rows;
    double var = q / rows - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    save_mean[c] = (float)mean;
    save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {      // torch: running = (1 - m) * running + m * batch, unbiased variance
        const double unbiased = rows > 1 ? var * rows / (rows - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const float *__restrict__ res,
                                                       const float *__restrict__ mean, const float *__restrict__ rstd,
                                                       float *__restrict__ y, size_t total, int C, int relu) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;          // C % 4 == 0
    if (i >= total) return;
    const int c = i % C;
    const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + i), mu = *reinterpret_cast<const f32x4 *>(mean + c),
                rs = *reinterpret_cast<const f32x4 *>(rstd + c), ga = *reinterpret_cast<const f32x4 *>(gamma + c),
                be = *reinterpret_cast<const f32x4 *>(beta + c);
    f32x4 v = (xv - mu) * rs * ga + be;
    if (res) v += *reinterpret_cast<const f32x4 *>(res + i);
    if (relu)
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
    *reinterpret_cast<f32x4 *>(y + i) = v;
}

__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const double *__restrict__ partial, int chunks, int C,
                                                            float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ double lds[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ry = threadIdx.x >> 6;
    double s, q;
    bn_sum_partials(partial, chunks, C, c, ry, lds, s, q);
    if (ry != 0 || c >= C) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

// dx = gamma * rstd * (g - dbeta/N - xhat * dgamma/N); dres = g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                           const float *__restrict__ y, const float *__restrict__ gamma,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd,
                                                           const float *__restrict__ dgamma,
                                                           const float *__restrict__ dbeta, float *__restrict__ dx,
                                                           float *__restrict__ dres, size_t total, int C, float inv_n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;          // C % 4 == 0
    if (i >= total) return;
    const int c = i % C;
    f32x4 g = *reinterpret_cast<const f32x4 *>(dy + i);
    if (y) {
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + i);
#pragma unroll
        for (int e = 0; e < 4; e++) g[e] = yv[e] > 0.f ? g[e] : 0.f;
    }
    const f32x4 rs = *reinterpret_cast<const f32x4 *>(rstd + c);
    const f32x4 xh = (*reinterpret_cast<const f32x4 *>(x + i) - *reinterpret_cast<const f32x4 *>(mean + c)) * rs;
    *reinterpret_cast<f32x4 *>(dx + i) = *reinterpret_cast<const f32x4 *>(gamma + c) * rs *
        (g - *reinterpret_cast<const f32x4 *>(dbeta + c) * inv_n - xh * *reinterpret_cast<const f32x4 *>(dgamma + c) * inv_n);
    if (dres) *reinterpret_cast<f32x4 *>(dres + i) = g;
}

// ---------------- BatchNorm of a short tensor in ONE launch ----------------
// At 4 images per GPU most ResNet-50 BatchNorms see 196 - 3136 rows: three launches of 4 - 11 us each (partials,
// statistics, apply) for a few hundred KB.  A channel never needs another channel's rows, so a workgroup that owns
// 16 channels (a 64-byte column strip: 4 channel quads x 64 row lanes) can do the whole layer on its strip: sum it
// (double, fixed order), finish the statistics, and walk it again (the strip is a few hundred KB, the second walk
// hits the cache) to write the output.  Used for rows <= bn_fused_rows(); longer tensors keep the chunked path, whose
// row chunks spread over more workgroups than C / 16.
static int bn_fused_rows() {           // A/B switch: ZS_BN_FUSED_ROWS=0 is always the three-launch path
    static const int v = getenv("ZS_BN_FUSED_ROWS") ? atoi(getenv("ZS_BN_FUSED_ROWS")) : 1024;
    return v;
}
// sum over the 64 row lanes of per-thread a[4], b[4] -> tot[2][16] (threads 0..31 hold one total each, in LDS)
__device__ __forceinline__ void bn_strip_totals(const double *a, const double *b, double (*lds)[64][17], double (*tot)[16]) {
    const int cq = threadIdx.x & 3, ry = threadIdx.x >> 2;
#pragma unroll
    for (int e = 0; e < 4; e++) { lds[0][ry][4 * cq + e] = a[e]; lds[1][ry][4 * cq + e] = b[e]; }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int which = threadIdx.x >> 4, col = threadIdx.x & 15;
        double t = 0.0;
        for (int k = 0; k < 64; k++) t += lds[which][k][col];
        tot[which][col] = t;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void bn_fused_train_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ res, float *__restrict__ y, float *__restrict__ running_mean,
    float *__restrict__ running_var, float *__restrict__ save_mean, float *__restrict__ save_rstd, int rows, int C,
    float eps, float momentum, int relu) {
    __shared__ double lds[2][64][17];
    __shared__ double tot[2][16];
    __shared__ __attribute__((aligned(16))) float stat[2][16];
    const int cq = threadIdx.x & 3, ry = threadIdx.x >> 2, c = blockIdx.x * 16 + 4 * cq;
    const bool live = c < C;                                   // C % 4 == 0: a quad is in range as a whole
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (live)
        for (int r = ry; r < rows; r += 64) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + (size_t)r * C + c);
#pragma unroll
            for (int e = 0; e < 4; e++) { a[e] += xv[e]; b[e] += (double)xv[e] * xv[e]; }
        }
    bn_strip_totals(a, b, lds, tot);
    if (threadIdx.x < 16) {
        const int cc = blockIdx.x * 16 + threadIdx.x;
        const double mean = tot[0][threadIdx.x] / rows;
        double var = tot[1][threadIdx.x] / rows - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
        stat[0][threadIdx.x] = mf;
        stat[1][threadIdx.x] = rf;
        if (cc < C) {
            save_mean[cc] = mf;
            save_rstd[cc] = rf;
            if (running_mean) {      // torch: running = (1 - m) * running + m * batch, unbiased variance
                const double unbiased = rows > 1 ? var * rows / (rows - 1.0) : var;
                running_mean[cc] = (float)((1.0 - momentum) * running_mean[cc] + momentum * mean);
                running_var[cc] = (float)((1.0 - momentum) * running_var[cc] + momentum * unbiased);
            }
        }
    }
    __syncthreads();
    if (!live) return;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(&stat[0][4 * cq]), rs = *reinterpret_cast<const f32x4 *>(&stat[1][4 * cq]),
                ga = *reinterpret_cast<const f32x4 *>(gamma + c), be = *reinterpret_cast<const f32x4 *>(beta + c);
    for (int r = ry; r < rows; r += 64) {
        const size_t o = (size_t)r * C + c;
        f32x4 v = (*reinterpret_cast<const f32x4 *>(x + o) - mu) * rs * ga + be;
        if (res) v += *reinterpret_cast<const f32x4 *>(res + o);
        if (relu)
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
        *reinterpret_cast<f32x4 *>(y + o) = v;
    }
}

__global__ __launch_bounds__(256) void bn_fused_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ y,
    const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ rstd,
    float *__restrict__ dx, float *__restrict__ dres, float *__restrict__ dgamma, float *__restrict__ dbeta, int rows,
    int C) {
    __shared__ double lds[2][64][17];
    __shared__ double tot[2][16];
    __shared__ __attribute__((aligned(16))) float sums[2][16];
    const int cq = threadIdx.x & 3, ry = threadIdx.x >> 2, c = blockIdx.x * 16 + 4 * cq;
    const bool live = c < C;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    f32x4 mu = {0, 0, 0, 0}, rs = {0, 0, 0, 0};
    if (live) {
        mu = *reinterpret_cast<const f32x4 *>(mean + c);
        rs = *reinterpret_cast<const f32x4 *>(rstd + c);
        for (int r = ry; r < rows; r += 64) {
            const size_t o = (size_t)r * C + c;
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + o);
            f32x4 g = *reinterpret_cast<const f32x4 *>(dy + o);
            if (y) {
                const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + o);
#pragma unroll
                for (int e = 0; e < 4; e++) g[e] = yv[e] > 0.f ? g[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) { a[e] += g[e]; b[e] += (double)g[e] * ((xv[e] - mu[e]) * rs[e]); }
        }
    }
    bn_strip_totals(a, b, lds, tot);
    if (threadIdx.x < 16) {
        const int cc = blockIdx.x * 16 + threadIdx.x;
        const float sb = (float)tot[0][threadIdx.x], sg = (float)tot[1][threadIdx.x];
        sums[0][threadIdx.x] = sb;
        sums[1][threadIdx.x] = sg;
        if (cc < C) { dbeta[cc] = sb; dgamma[cc] = sg; }
    }
    __syncthreads();
    if (!live) return;
    const float inv_n = 1.0f / rows;
    const f32x4 db = *reinterpret_cast<const f32x4 *>(&sums[0][4 * cq]), dg = *reinterpret_cast<const f32x4 *>(&sums[1][4 * cq]),
                ga = *reinterpret_cast<const f32x4 *>(gamma + c);
    for (int r = ry; r < rows; r += 64) {
        const size_t o = (size_t)r * C + c;
        f32x4 g = *reinterpret_cast<const f32x4 *>(dy + o);
        if (y) {
            const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + o);
#pragma unroll
            for (int e = 0; e < 4; e++) g[e] = yv[e] > 0.f ? g[e] : 0.f;
        }
        const f32x4 xh = (*reinterpret_cast<const f32x4 *>(x + o) - mu) * rs;
        *reinterpret_cast<f32x4 *>(dx + o) = ga * rs * (g - db * inv_n - xh * dg * inv_n);
        if (dres) *reinterpret_cast<f32x4 *>(dres + o) = g;
    }
}

// ---------------- GroupNorm backward: (sample, group) x pixel slices, two launches ----------------
// A (sample, group) holds HW * cg values behind one stride-C gather; on one workgroup it runs at one CU's bandwidth
// with half the chip idle (28 us per layer at batch 4), so its pixels are cut into `slices` slices.  Launch 1 sums
// every slice at once (x, x^2 and per channel dy, dy x with dy masked by the fused ReLU) in double into
// stat[bg][slice][2 + 2 cg]; launch 2 adds the slices in slice order - statistics and the projections follow
// algebraically (sum g xhat = rstd (sum g x - mean sum g), g = gamma dy) - and writes dx (and the masked dy for the
// residual) of its slice.  VEC = min(cg, 4) channels per thread (one 4 / 8 / 16-byte load per tensor and pixel); the
// threads of a pixel are adjacent lanes, so a wave reads whole group rows.
constexpr int GNB_BLOCK = 256;
template <int VEC>
__device__ __forceinline__ void gn_load(const float *ptr, size_t o, float *v) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    if constexpr (VEC == 1) v[0] = ptr[o];
    else {
        const vec_t t = *reinterpret_cast<const vec_t *>(ptr + o);
#pragma unroll
        for (int e = 0; e < VEC; e++) v[e] = t[e];
    }
}
template <int VEC>
__global__ __launch_bounds__(GNB_BLOCK) void group_norm_bwd_stats_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ y, double *__restrict__ stat,
    int HW, int C, int groups, int per_slice) {
    __shared__ double red[GNB_BLOCK / 64][2];
    __shared__ double col[GNB_BLOCK / 64][2][64];              // [wave][dy | dy x][channel of the group]
    const int cg = C / groups, qn = cg / VEC, b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cq = threadIdx.x % qn, p0 = threadIdx.x / qn, pstep = GNB_BLOCK / qn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p_begin = blockIdx.y * per_slice, p_end = min(HW, p_begin + per_slice);
    const size_t base = (size_t)b * HW * C + (size_t)g * cg + cq * VEC;
    double sx = 0, sxx = 0, sr[VEC], srx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) sr[e] = srx[e] = 0;
    auto accumulate = [&](const float *xv, const float *gv, const float *yv) {
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const float gr = (y && !(yv[e] > 0.f)) ? 0.f : gv[e];
            sx += xv[e];
            sxx += (double)xv[e] * xv[e];
            sr[e] += gr;
            srx[e] += (double)gr * xv[e];
        }
    };
    int p = p_begin + p0;
    for (; p + pstep < p_end; p += 2 * pstep) {                // two pixels in flight
        const size_t o0 = base + (size_t)p * C, o1 = o0 + (size_t)pstep * C;
        float x0[VEC], g0[VEC], y0[VEC], x1[VEC], g1[VEC], y1[VEC];
        gn_load<VEC>(x, o0, x0); gn_load<VEC>(dy, o0, g0); gn_load<VEC>(x, o1, x1); gn_load<VEC>(dy, o1, g1);
        if (y) { gn_load<VEC>(y, o0, y0); gn_load<VEC>(y, o1, y1); }
        accumulate(x0, g0, y0);
        accumulate(x1, g1, y1);
    }
    if (p < p_end) {
        const size_t o0 = base + (size_t)p * C;
        float x0[VEC], g0[VEC], y0[VEC];
        gn_load<VEC>(x, o0, x0); gn_load<VEC>(dy, o0, g0);
        if (y) gn_load<VEC>(y, o0, y0);
        accumulate(x0, g0, y0);
    }
    // fixed-order sums: butterflies inside a wave (all lanes for x, the lanes of the same channels for dy), then the waves
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off, 64); sxx += __shfl_xor(sxx, off, 64); }
#pragma unroll
    for (int e = 0; e < VEC; e++)
        for (int off = 32; off >= qn; off >>= 1) {
            sr[e] += __shfl_xor(sr[e], off, 64);
            srx[e] += __shfl_xor(srx[e], off, 64);
        }
    if (lane == 0) { red[wave][0] = sx; red[wave][1] = sxx; }
    if (lane < qn)
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            col[wave][0][lane * VEC + e] = sr[e];
            col[wave][1][lane * VEC + e] = srx[e];
        }
    __syncthreads();
    double *out = stat + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (2 + 2 * cg);
    const int t = threadIdx.x;
    if (t < 2) out[t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    else if (t < 2 + 2 * cg) {
        const int which = (t - 2) / cg, c = (t - 2) % cg;
        out[t] = (col[0][which][c] + col[1][which][c]) + (col[2][which][c] + col[3][which][c]);
    }
}

template <int VEC>
__global__ __launch_bounds__(GNB_BLOCK) void group_norm_bwd_apply_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ y,
    const float *__restrict__ gamma, const double *__restrict__ stat, float *__restrict__ dx, float *__restrict__ dres,
    float *__restrict__ dgamma_part, float *__restrict__ dbeta_part, int HW, int C, int groups, int per_slice, float eps) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ double tot[2 + 2 * 64];
    __shared__ float bc[4];                                    // mean, rstd, mean of g, mean of g xhat
    const int cg = C / groups, qn = cg / VEC, b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cq = threadIdx.x % qn, p0 = threadIdx.x / qn, pstep = GNB_BLOCK / qn, n_stat = 2 + 2 * cg;
    const int slices = gridDim.y;
    if (threadIdx.x < n_stat) {
        const double *src = stat + (size_t)blockIdx.x * slices * n_stat + threadIdx.x;
        double t = 0;
        for (int sl = 0; sl < slices; sl++) t += src[(size_t)sl * n_stat];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const double n = (double)HW * cg, mean = tot[0] / n;
        double var = tot[1] / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        double sg = 0, sgx = 0;
        if ((int)threadIdx.x < cg) {
            const double gm = gamma[g * cg + threadIdx.x], tr = tot[2 + threadIdx.x], trx = tot[2 + cg + threadIdx.x];
            sg = tr * gm;
            sgx = trx * gm;
            if (blockIdx.y == 0) {                             // rows of [dgamma | dbeta], one per sample
                dgamma_part[(size_t)b * 2 * C + g * cg + threadIdx.x] = (float)((trx - mean * tr) * rstd);
                dbeta_part[(size_t)b * 2 * C + g * cg + threadIdx.x] = (float)tr;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sg += __shfl_xor(sg, off, 64); sgx += __shfl_xor(sgx, off, 64); }
        if (threadIdx.x == 0) {
            bc[0] = (float)mean;
            bc[1] = (float)rstd;
            bc[2] = (float)(sg / n);
            bc[3] = (float)((sgx - mean * sg) * rstd / n);
        }
    }
    __syncthreads();
    const float meanf = bc[0], rstdf = bc[1], mg = bc[2], mgx = bc[3];
    float gam[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) gam[e] = gamma[g * cg + cq * VEC + e];
    const int p_begin = blockIdx.y * per_slice, p_end = min(HW, p_begin + per_slice);
    const size_t base = (size_t)b * HW * C + (size_t)g * cg + cq * VEC;
    for (int p = p_begin + p0; p < p_end; p += pstep) {
        const size_t o = base + (size_t)p * C;
        float xv[VEC], gv[VEC], yv[VEC];
        gn_load<VEC>(x, o, xv); gn_load<VEC>(dy, o, gv);
        if (y) gn_load<VEC>(y, o, yv);
        vec_t out, res;
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const float gr = (y && !(yv[e] > 0.f)) ? 0.f : gv[e], xh = (xv[e] - meanf) * rstdf;
            out[e] = rstdf * (gr * gam[e] - mg - xh * mgx);
            res[e] = gr;
        }
        *reinterpret_cast<vec_t *>(dx + o) = out;
        if (dres) *reinterpret_cast<vec_t *>(dres + o) = res;
    }
}

// out[c] = sum_r part[r][c], c < C; columns C .. 2C-1 (the second vector of the same rows) go to out2
__global__ __launch_bounds__(256) void sum_rows_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                       float *__restrict__ out2, int rows, int C, int C2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C2) return;
    float s = 0.f;
    for (int r = 0; r < rows; r++) s += part[(size_t)r * C2 + c];
    if (c < C) out[c] = s;
    else out2[c - C] = s;
}

// ---------------- pooling / resampling backward (gather form, no atomics) ----------------
// max pool: an input pixel receives dy of every window whose FIRST maximum (row-major scan, as
// torch's max_pool2d picks it) it is
__global__ __launch_bounds__(256) void max_pool_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                           float *__restrict__ dx, int B, int Hin, int Win, int C,
                                                           int Hout, int Wout, int k, int stride, int pad_t, int pad_l) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * Hin * Win * C;
    if (i >= total) return;
    const int c = i % C, ix = (i / C) % Win, iy = (i / C / Win) % Hin, b = i / C / Win / Hin;
    const float *X = x + (size_t)b * Hin * Win * C + c;
    float acc = 0.f;
    // windows (oy, ox) with oy*stride - pad_t <= iy < oy*stride - pad_t + k
    const int oy_lo = max(0, (iy + pad_t - k + stride) / stride), oy_hi = min(Hout - 1, (iy + pad_t) / stride);
    const int ox_lo = max(0, (ix + pad_l - k + stride) / stride), ox_hi = min(Wout - 1, (ix + pad_l) / stride);
    for (int oy = oy_lo; oy <= oy_hi; oy++)
        for (int ox = ox_lo; ox <= ox_hi; ox++) {
            float m = -INFINITY;
            int my = -1, mx = -1;
            for (int ky = 0; ky < k; ky++)
                for (int kx = 0; kx < k; kx++) {
                    const int yy = oy * stride - pad_t + ky, xx = ox * stride - pad_l + kx;
                    if (yy < 0 || yy >= Hin || xx < 0 || xx >= Win) continue;
                    const float v = X[((size_t)yy * Win + xx) * C];
                    if (v > m || my < 0) { m = v; my = yy; mx = xx; }
                }
            if (my == iy && mx == ix) acc += dy[(((size_t)b * Hout + oy) * Wout + ox) * C + c];
        }
    dx[i] = acc;
}

__global__ __launch_bounds__(256) void global_mean_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx,
                                                              int HW, int C, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = i % C;
    const size_t b = i / C / HW;
    dx[i] = dy[b * C + c] / HW;
}

// x2 bilinear, align_corners=True: adjoint by gathering from the (few) outputs that read this input; a thread owns
// VEC channels of one input pixel (VEC = 4 when C % 4 == 0: the tap weights are computed once per four channels)
template <int VEC>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int B,
                                                             int Hin, int Win, int C) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int Hout = 2 * Hin, Wout = 2 * Win, CV = C / VEC;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * Hin * Win * CV;
    if (i >= total) return;
    const int c = (int)(i % CV) * VEC, ix = (i / CV) % Win, iy = (i / CV / Win) % Hin, b = i / CV / Win / Hin;
    const float sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f,
                sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    // forward taps of output o: y0 = (int)(sy*o), y1 = y0 + (y0 < Hin-1), weights (1-l, l), l = sy*o - y0
    auto weight = [](int o, int in, float s, int n) -> float {
        const float f = s * o;
        const int i0 = (int)f, i1 = i0 + (i0 < n - 1 ? 1 : 0);
        const float l = f - i0;
        float w = 0.f;
        if (i0 == in) w += 1.f - l;
        if (i1 == in) w += l;
        return w;
    };
    const int oy_lo = max(0, 2 * iy - 3), oy_hi = min(Hout - 1, 2 * iy + 4);
    const int ox_lo = max(0, 2 * ix - 3), ox_hi = min(Wout - 1, 2 * ix + 4);
    const float *G = dy + (size_t)b * Hout * Wout * C + c;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[e] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; oy++) {
        const float wy = weight(oy, iy, sy, Hin);
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ox++) {
            const float wx = weight(ox, ix, sx, Win);
            if (wx == 0.f) continue;
            const float *g = G + ((size_t)oy * Wout + ox) * C;
            if constexpr (VEC == 1) acc[0] += wy * wx * g[0];
            else {
                const vec_t v = *reinterpret_cast<const vec_t *>(g);
#pragma unroll
                for (int e = 0; e < VEC; e++) acc[e] += wy * wx * v[e];
            }
        }
    }
    float *o = dx + (((size_t)b * Hin + iy) * Win + ix) * C + c;
#pragma unroll
    for (int e = 0; e < VEC; e++) o[e] = acc[e];
}

// NHWC [B][HW][Cpad] -> NCHW [B][C][HW] (first C channels), optionally times mask [B][HW]
__global__ __launch_bounds__(256) void nhwc_to_nchw_masked_kernel(const float *__restrict__ x,
                                                                  const float *__restrict__ mask, float *__restrict__ y,
                                                                  int B, int C, int HW, int Cpad) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * HW * C;
    if (i >= total) return;
    const int p = i % HW, c = (i / HW) % C, b = i / HW / C;
    float v = x[((size_t)b * HW + p) * Cpad + c];
    if (mask) v *= mask[(size_t)b * HW + p];
    y[i] = v;
}

// ---- bilinear resize, align_corners=False, channels-last [Hi][Wi][C] -> [Ho][Wo][C] ----
// (vit.py:103-120 _resize_pos_embed: the ViT position-embedding grid is re-sampled on every forward)
struct RTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ RTap resize_tap(int dst, float scale, int in_size) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    RTap t;
    t.i0 = min((int)src, in_size - 1);
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.0f - t.l1;
    return t;
}
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ x, float *__restrict__ y, int Hi,
                                                              int Wi, int Ho, int Wo, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)Ho * Wo * C;
    if (i >= total) return;
    const int c = i % C, ox = (i / C) % Wo, oy = i / C / Wo;
    const RTap ty = resize_tap(oy, (float)Hi / Ho, Hi), tx = resize_tap(ox, (float)Wi / Wo, Wi);
    const float *X = x + c;
    y[i] = ty.l0 * (tx.l0 * X[((size_t)ty.i0 * Wi + tx.i0) * C] + tx.l1 * X[((size_t)ty.i0 * Wi + tx.i1) * C]) +
           ty.l1 * (tx.l0 * X[((size_t)ty.i1 * Wi + tx.i0) * C] + tx.l1 * X[((size_t)ty.i1 * Wi + tx.i1) * C]);
}
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx,
                                                                  int Hi, int Wi, int Ho, int Wo, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)Hi * Wi * C;
    if (i >= total) return;
    const int c = i % C, ix = (i / C) % Wi, iy = i / C / Wi;
    float acc = 0.f;
    for (int oy = 0; oy < Ho; oy++) {
        const RTap ty = resize_tap(oy, (float)Hi / Ho, Hi);
        const float wy = (ty.i0 == iy ? ty.l0 : 0.f) + (ty.i1 == iy ? ty.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = 0; ox < Wo; ox++) {
            const RTap tx = resize_tap(ox, (float)Wi / Wo, Wi);
            const float wx = (tx.i0 == ix ? tx.l0 : 0.f) + (tx.i1 == ix ? tx.l1 : 0.f);
            if (wx != 0.f) acc += wy * wx * dy[((size_t)oy * Wo + ox) * C + c];
        }
    }
    dx[i] = acc;
}

// adjoint of zs_readout_concat: dtok[b][1+i][c] = dout[b][i][c]; dtok[b][0][c] = sum_i dout[b][i][C+c]
__global__ __launch_bounds__(256) void readout_concat_bwd_kernel(const float *__restrict__ dout,
                                                                 float *__restrict__ dtok, int B, int n, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * (n + 1) * C;
    if (i >= total) return;
    const int c = i % C, t = (i / C) % (n + 1), b = i / C / (n + 1);
    const float *G = dout + (size_t)b * n * 2 * C;
    float v;
    if (t > 0) v = G[(size_t)(t - 1) * 2 * C + c];
    else {
        v = 0.f;
        for (int k = 0; k < n; k++) v += G[(size_t)k * 2 * C + C + c];
    }
    dtok[i] = v;
}

int bn_chunks(int rows) {
    int chunks = (rows + 255) / 256;
    if (chunks > 256) chunks = 256;
    return chunks < 1 ? 1 : chunks;
}

}  // namespace

#define ZS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            zs::set_err(__VA_ARGS__);    \
            return 0;                    \
        }                                \
    } while (0)

extern "C" size_t zs_batch_norm_workspace_bytes(int rows, int C) {
    return (size_t)bn_chunks(rows) * 2 * C * sizeof(double);
}

extern "C" int zs_batch_norm_train(const float *x, const float *gamma, const float *beta, const float *residual,
                                   float *y, float *running_mean, float *running_var, float *save_mean,
                                   float *save_rstd, int rows, int C, float eps, float momentum, int relu,
                                   void *workspace, void *stream) {
    ZS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0, "zs_batch_norm_train: bad size (rows=%d C=%d; C %% 4 == 0)", rows, C);
    ZS_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && workspace, "zs_batch_norm_train: null pointer");
    ZS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "zs_batch_norm_train: running stats come in pairs");
    if (rows <= bn_fused_rows()) {
        hipLaunchKernelGGL(bn_fused_train_kernel, dim3((C + 15) / 16), dim3(256), 0, S(stream), x, gamma, beta, residual, y,
                           running_mean, running_var, save_mean, save_rstd, rows, C, eps, momentum, relu);
        return zs::check_launch("zs_batch_norm_train") ? 1 : 0;
    }
    const int chunks = bn_chunks(rows), per = (rows + chunks - 1) / chunks;
    double *partial = static_cast<double *>(workspace);
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3((C + 63) / 64, chunks), dim3(256), 0, S(stream), x, nullptr, nullptr,
                       nullptr, nullptr, partial, rows, C, per);
    hipLaunchKernelGGL(bn_stats_kernel, dim3((C + 63) / 64), dim3(256), 0, S(stream), partial, chunks, C, rows, eps,
                       momentum, running_mean, running_var, save_mean, save_rstd);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks_for((size_t)rows * C / 4)), dim3(256), 0, S(stream), x, gamma, beta,
                       residual, save_mean, save_rstd, y, (size_t)rows * C, C, relu);
    return zs::check_launch("zs_batch_norm_train") ? 1 : 0;
}

extern "C" int zs_batch_norm_bwd(const float *x, const float *dy, const float *y_relu, const float *gamma,
                                 const float *save_mean, const float *save_rstd, float *dx, float *dresidual,
                                 float *dgamma, float *dbeta, int rows, int C, void *workspace, void *stream) {
    ZS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0, "zs_batch_norm_bwd: bad size (rows=%d C=%d; C %% 4 == 0)", rows, C);
    ZS_REQUIRE(x && dy && gamma && save_mean && save_rstd && dx && dgamma && dbeta && workspace,
               "zs_batch_norm_bwd: null pointer");
    if (rows <= bn_fused_rows()) {
        hipLaunchKernelGGL(bn_fused_bwd_kernel, dim3((C + 15) / 16), dim3(256), 0, S(stream), x, dy, y_relu, gamma, save_mean,
                           save_rstd, dx, dresidual, dgamma, dbeta, rows, C);
        return zs::check_launch("zs_batch_norm_bwd") ? 1 : 0;
    }
    const int chunks = bn_chunks(rows), per = (rows + chunks - 1) / chunks;
    double *partial = static_cast<double *>(workspace);
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3((C + 63) / 64, chunks), dim3(256), 0, S(stream), x, dy, y_relu,
                       save_mean, save_rstd, partial, rows, C, per);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 63) / 64), dim3(256), 0, S(stream), partial, chunks, C, dgamma,
                       dbeta);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for((size_t)rows * C / 4)), dim3(256), 0, S(stream), x, dy, y_relu,
                       gamma, save_mean, save_rstd, dgamma, dbeta, dx, dresidual, (size_t)rows * C, C, 1.0f / rows);
    return zs::check_launch("zs_batch_norm_bwd") ? 1 : 0;
}

constexpr int GNB_MAX_SLICES = 16;
// pixel slices per (sample, group): ~1024 workgroups in all, at least 64 pixels each
static int gn_bwd_slices(int batch, int HW, int groups) {
    int s = (1024 + batch * groups - 1) / (batch * groups);
    if (s > HW / 64) s = HW / 64;
    if (s > GNB_MAX_SLICES) s = GNB_MAX_SLICES;
    return s < 1 ? 1 : s;
}

extern "C" int zs_group_norm_bwd(const float *x, const float *dy, const float *y_relu, const float *gamma, float *dx,
                                 float *dresidual, float *dgamma, float *dbeta, int batch, int HW, int C, int groups,
                                 float eps, void *workspace, void *stream) {
    ZS_REQUIRE(batch > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C / groups <= 64 &&
                   ((C / groups) & (C / groups - 1)) == 0,
               "zs_group_norm_bwd: bad size (B=%d HW=%d C=%d groups=%d; channels per group: a power of two <= 64)", batch,
               HW, C, groups);
    ZS_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && workspace, "zs_group_norm_bwd: null pointer");
    const int cg = C / groups, slices = gn_bwd_slices(batch, HW, groups), per_slice = (HW + slices - 1) / slices;
    double *stat = static_cast<double *>(workspace);
    float *pg = reinterpret_cast<float *>(stat + (size_t)batch * groups * slices * (2 + 2 * cg)), *pb = pg + C;
    const dim3 grid(batch * groups, slices);
#define ZS_GNB(V)                                                                                                          \
    do {                                                                                                                   \
        hipLaunchKernelGGL((group_norm_bwd_stats_kernel<V>), grid, dim3(GNB_BLOCK), 0, S(stream), x, dy, y_relu, stat, HW,  \
                           C, groups, per_slice);                                                                          \
        hipLaunchKernelGGL((group_norm_bwd_apply_kernel<V>), grid, dim3(GNB_BLOCK), 0, S(stream), x, dy, y_relu, gamma,     \
                           stat, dx, dresidual, pg, pb, HW, C, groups, per_slice, eps);                                    \
    } while (0)
    if (cg >= 4) ZS_GNB(4);
    else if (cg == 2) ZS_GNB(2);
    else ZS_GNB(1);
#undef ZS_GNB
    hipLaunchKernelGGL(sum_rows_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, S(stream), pg, dgamma, dbeta, batch, C,
                       2 * C);
    return zs::check_launch("zs_group_norm_bwd") ? 1 : 0;
}

extern "C" size_t zs_group_norm_bwd_workspace_bytes(int batch, int C) {
    // slice statistics (at most GNB_MAX_SLICES slices of 2 + 2 cg doubles per (sample, group): <= 4 C + 2 C doubles
    // per sample and slice whatever the group count) + the per-sample [dgamma | dbeta] rows
    return (size_t)batch * GNB_MAX_SLICES * 6 * C * sizeof(double) + (size_t)2 * batch * C * sizeof(float);
}

extern "C" int zs_max_pool_bwd_nhwc(const float *x, const float *dy, float *dx, int batch, int Hin, int Win, int C,
                                    int Hout, int Wout, int k, int stride, int pad_t, int pad_l, void *stream) {
    ZS_REQUIRE(batch >= 0 && Hin > 0 && Win > 0 && C > 0 && Hout > 0 && Wout > 0 && k > 0 && stride > 0,
               "zs_max_pool_bwd_nhwc: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(x && dy && dx, "zs_max_pool_bwd_nhwc: null pointer");
    hipLaunchKernelGGL(max_pool_bwd_kernel, dim3(blocks_for((size_t)batch * Hin * Win * C)), dim3(256), 0, S(stream), x,
                       dy, dx, batch, Hin, Win, C, Hout, Wout, k, stride, pad_t, pad_l);
    return zs::check_launch("zs_max_pool_bwd_nhwc") ? 1 : 0;
}

extern "C" int zs_global_mean_bwd_nhwc(const float *dy, float *dx, int batch, int HW, int C, void *stream) {
    ZS_REQUIRE(batch >= 0 && HW > 0 && C > 0, "zs_global_mean_bwd_nhwc: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(dy && dx, "zs_global_mean_bwd_nhwc: null pointer");
    const size_t total = (size_t)batch * HW * C;
    hipLaunchKernelGGL(global_mean_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, S(stream), dy, dx, HW, C, total);
    return zs::check_launch("zs_global_mean_bwd_nhwc") ? 1 : 0;
}

extern "C" int zs_upsample2x_bwd_nhwc(const float *dy, float *dx, int batch, int Hin, int Win, int C, void *stream) {
    ZS_REQUIRE(batch >= 0 && Hin > 0 && Win > 0 && C > 0, "zs_upsample2x_bwd_nhwc: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(dy && dx, "zs_upsample2x_bwd_nhwc: null pointer");
    if ((C & 3) == 0 && ((reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(dx)) & 15) == 0)
        hipLaunchKernelGGL(upsample2x_bwd_kernel<4>, dim3(blocks_for((size_t)batch * Hin * Win * C / 4)), dim3(256), 0,
                           S(stream), dy, dx, batch, Hin, Win, C);
    else
        hipLaunchKernelGGL(upsample2x_bwd_kernel<1>, dim3(blocks_for((size_t)batch * Hin * Win * C)), dim3(256), 0,
                           S(stream), dy, dx, batch, Hin, Win, C);
    return zs::check_launch("zs_upsample2x_bwd_nhwc") ? 1 : 0;
}

extern "C" int zs_nhwc_to_nchw_masked(const float *x, const float *mask, float *y, int batch, int C, int HW, int Cpad,
                                      void *stream) {
    ZS_REQUIRE(batch >= 0 && C > 0 && HW > 0 && Cpad >= C, "zs_nhwc_to_nchw_masked: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(x && y, "zs_nhwc_to_nchw_masked: null pointer");
    hipLaunchKernelGGL(nhwc_to_nchw_masked_kernel, dim3(blocks_for((size_t)batch * HW * C)), dim3(256), 0, S(stream), x,
                       mask, y, batch, C, HW, Cpad);
    return zs::check_launch("zs_nhwc_to_nchw_masked") ? 1 : 0;
}

extern "C" int zs_resize_bilinear_nhwc(const float *x, float *y, int Hi, int Wi, int Ho, int Wo, int C, int backward,
                                       void *stream) {
    ZS_REQUIRE(Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "zs_resize_bilinear_nhwc: bad size");
    ZS_REQUIRE(x && y, "zs_resize_bilinear_nhwc: null pointer");
    if (!backward)
        hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks_for((size_t)Ho * Wo * C)), dim3(256), 0, S(stream), x, y,
                           Hi, Wi, Ho, Wo, C);
    else        // x = gradient w.r.t. the [Ho][Wo][C] output, y = gradient w.r.t. the [Hi][Wi][C] input
        hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(blocks_for((size_t)Hi * Wi * C)), dim3(256), 0, S(stream), x,
                           y, Hi, Wi, Ho, Wo, C);
    return zs::check_launch("zs_resize_bilinear_nhwc") ? 1 : 0;
}

extern "C" int zs_readout_concat_bwd(const float *dout, float *dtokens, int batch, int n, int C, void *stream) {
    ZS_REQUIRE(batch >= 0 && n > 0 && C > 0, "zs_readout_concat_bwd: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(dout && dtokens, "zs_readout_concat_bwd: null pointer");
    hipLaunchKernelGGL(readout_concat_bwd_kernel, dim3(blocks_for((size_t)batch * (n + 1) * C)), dim3(256), 0, S(stream),
                       dout, dtokens, batch, n, C);
    return zs::check_launch("zs_readout_concat_bwd") ? 1 : 0;
}
