// The non-GEMM layers of the image encoders, channels-last
// fp32, each an HBM-bound single pass (or a per-group / per-row reduction):
//   GroupNorm(+residual)(+ReLU)    timm ResNetV2 GroupNormAct inside DPT-hybrid's backbone
//   LayerNorm                      ViT blocks
//   multi-head attention           ViT blocks (197 tokens; softmax(q k^T / sqrt(d)) v)
//   3x3/s2 max pool, global mean   ResNet stems / heads
//   x2 bilinear (align_corners)    DPT fusion blocks and head (model/depth/blocks.py:330-336)
//   NCHW <-> NHWC                  boundary conversions (the reference's tensors are NCHW)
//   token assembly / readout cat   model/depth/vit.py:127-148 / :31-43
#include "zs_common.h"
#include "zs_split16.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdlib.h>
#include <stdint.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = lds[0];
    for (int w = 1; w < nw; w++) r += lds[w];
    return r;
}

__device__ __forceinline__ double block_sum_d(double v, double *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = lds[0];
    for (int w = 1; w < nw; w++) r += lds[w];
    return r;
}

// ---- GroupNorm over (HW x C/groups) of one sample, then affine, optional residual, optional ReLU ----
#ifndef ZS_GN_BLOCK
#define ZS_GN_BLOCK 512      /* 1024 / 512 / 256 lanes per (sample, group) slice: encoder batch 28 15.93 / 15.76 / 15.78 ms, batch 1 3.64 / 3.60 / 3.70 */
#endif
constexpr int GN_BLOCK = ZS_GN_BLOCK;
constexpr int GN_CACHE_BYTES = 112 * 1024;     // a (sample, group) slice up to this size is read from HBM once
// generic form: any group width, two passes over global memory
__global__ __launch_bounds__(GN_BLOCK) void group_norm_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta,
                                                         const float *__restrict__ res, float *__restrict__ y,
                                                         int HW, int C, int groups, float eps, int relu) {
    __shared__ double ldsd[GN_BLOCK / 64];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups, cg = C / groups, n = HW * cg;
    const size_t base = (size_t)b * HW * C + (size_t)g * cg;
    auto at = [&](int e) -> size_t { return base + (size_t)(e / cg) * C + (e % cg); };
    // one statistics pass: sum and sum of squares in double (no cancellation issue), then one output pass
    double s = 0.0, q = 0.0;
    for (int e = threadIdx.x; e < n; e += GN_BLOCK) {
        const float v = x[at(e)];
        s += v;
        q += (double)v * v;
    }
    const double S = block_sum_d(s, ldsd), Q = block_sum_d(q, ldsd);
    const double mean_d = S / n;
    double var = Q / n - mean_d * mean_d;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int e = threadIdx.x; e < n; e += GN_BLOCK) {
        const size_t o = at(e);
        const int c = g * cg + e % cg;
        float v = (x[o] - mean) * rstd * gamma[c] + beta[c];
        if (res) v += res[o];
        y[o] = relu ? fmaxf(v, 0.f) : v;
    }
}

// Group width a power of two (every layer of the encoders: 2, 8, 16, 32, 64 channels per group): VEC-wide loads
// (VEC = min(4, cg)), pixel / channel split by shift and mask instead of a division per element, and - CACHE - the
// slice kept in LDS between the statistics pass and the output pass (<= 112 KiB: all of the encoders' layers up to
// batch-independent 56 x 56 x 8-channel slices), so x is read from HBM once.  Same formulae, double sums.
template <int VEC, bool CACHE>
__global__ __launch_bounds__(GN_BLOCK) void group_norm_pow2_kernel(const float *__restrict__ x,
                                                                   const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta,
                                                                   const float *__restrict__ res, float *__restrict__ y,
                                                                   int HW, int C, int groups, int cgv_shift, float eps,
                                                                   int relu) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) float gn_cache[];
    __shared__ double ldsd[GN_BLOCK / 64];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups, cg = C / groups;
    const int nv = (HW * cg) / VEC, cgv_mask = (1 << cgv_shift) - 1;          // cg / VEC = 1 << cgv_shift vectors per pixel
    const size_t base = (size_t)b * HW * C + (size_t)g * cg;
    auto at = [&](int ev) -> size_t { return base + (size_t)(ev >> cgv_shift) * C + (size_t)(ev & cgv_mask) * VEC; };
    double s = 0.0, q = 0.0;
    for (int ev = threadIdx.x; ev < nv; ev += GN_BLOCK) {
        const vec_t v = *reinterpret_cast<const vec_t *>(x + at(ev));
        if (CACHE) reinterpret_cast<vec_t *>(gn_cache)[ev] = v;
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            const float f = VEC == 1 ? ((const float *)&v)[0] : v[i];
            s += f;
            q += (double)f * f;
        }
    }
    const double S = block_sum_d(s, ldsd), Q = block_sum_d(q, ldsd);
    const double n = (double)HW * cg, mean_d = S / n;
    double var = Q / n - mean_d * mean_d;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int ev = threadIdx.x; ev < nv; ev += GN_BLOCK) {
        const size_t o = at(ev);
        const int c = g * cg + (ev & cgv_mask) * VEC;
        const vec_t xv = CACHE ? reinterpret_cast<const vec_t *>(gn_cache)[ev] : *reinterpret_cast<const vec_t *>(x + o);
        const vec_t ga = *reinterpret_cast<const vec_t *>(gamma + c), be = *reinterpret_cast<const vec_t *>(beta + c);
        vec_t v = (xv - mean) * rstd * ga + be;
        if (res) v += *reinterpret_cast<const vec_t *>(res + o);
        if (relu)
#pragma unroll
            for (int i = 0; i < VEC; i++) {
                if (VEC == 1) ((float *)&v)[0] = fmaxf(((float *)&v)[0], 0.f);
                else v[i] = fmaxf(v[i], 0.f);
            }
        *reinterpret_cast<vec_t *>(y + o) = v;
    }
}

// ---- large tensors (round 3): two coalesced launches instead of one workgroup per (sample, group) -------------- //
// The kernels above give a (sample, group) slice to one workgroup.  In NHWC a slice is cg channels (32 bytes at
// cg = 8) of every pixel, C * 4 bytes apart: a wave instruction touches 32 cache lines and uses a quarter of each, a
// slice up to 112 KiB fills the LDS of a CU with ONE workgroup, and the big layers of the batch-28 encoder ran at
// 1.4-2.1 TB/s (2.1 ms of 21 per forward).  Here a workgroup owns a run of PIXELS with all their channels: every load
// is a full contiguous line, many workgroups per CU.  Launch 1 (gn_partial_kernel): per (sample, pixel chunk) the
// sum and sum of squares of every group, in double, combined in a fixed order.  Launch 2 (gn_apply_kernel): every
// workgroup adds up the chunks' partials of its sample (fixed order: deterministic), then normalises its own chunk.
// x is read twice (the second time mostly from the Infinity Cache) and written once.  Same formulae as above.
constexpr int GN2_THREADS = 256;
__host__ __device__ inline int gn2_chunk_pixels(int C) {          // ~64 KiB of input per workgroup
    const int p = 16384 / C;
    return p < 16 ? 16 : p;
}
// thread t owns channel quad (t % quads) of the pixels po, po + step, ...; quads = C / 4 <= 256; a quad lies inside one
// group (cg >= 4) or covers two (cg = 2: halves xy | zw)
__global__ __launch_bounds__(GN2_THREADS) void gn_partial_kernel(const float *__restrict__ x, int HW, int C, int groups,
                                                                 double *__restrict__ partial, int chunks) {
    __shared__ double ss[2 * GN2_THREADS], sq[2 * GN2_THREADS];       // [half][thread]
    const int b = blockIdx.y, chunk = blockIdx.x, quads = C >> 2, step = GN2_THREADS / quads;
    const int cq = threadIdx.x % quads, po = threadIdx.x / quads;
    const int cp = gn2_chunk_pixels(C), p0 = chunk * cp, p1 = min(HW, p0 + cp);
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + (size_t)b * HW * C);
    double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
    if (po < step)
        for (int p = p0 + po; p < p1; p += step) {
            const f32x4 v = xr[(size_t)p * quads + cq];
            s0 += (double)v.x + (double)v.y;
            q0 += (double)v.x * v.x + (double)v.y * v.y;
            s1 += (double)v.z + (double)v.w;
            q1 += (double)v.z * v.z + (double)v.w * v.w;
        }
    ss[threadIdx.x] = s0;
    sq[threadIdx.x] = q0;
    ss[GN2_THREADS + threadIdx.x] = s1;
    sq[GN2_THREADS + threadIdx.x] = q1;
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int cg = C / groups;
        double S = 0.0, Q = 0.0;
        if (cg >= 4) {
            const int qg = cg >> 2;                           // quads per group, both halves of each
            for (int r = 0; r < step; r++)
                for (int k = 0; k < qg; k++) {
                    const int t = r * quads + threadIdx.x * qg + k;
                    S += ss[t] + ss[GN2_THREADS + t];
                    Q += sq[t] + sq[GN2_THREADS + t];
                }
        } else {                                              // cg = 2: group g = half (g & 1) of quad g / 2
            const int h = (threadIdx.x & 1) * GN2_THREADS;
            for (int r = 0; r < step; r++) {
                S += ss[h + r * quads + (threadIdx.x >> 1)];
                Q += sq[h + r * quads + (threadIdx.x >> 1)];
            }
        }
        double *o = partial + (((size_t)b * chunks + chunk) * groups + threadIdx.x) * 2;
        o[0] = S;
        o[1] = Q;
    }
}
__global__ __launch_bounds__(GN2_THREADS) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, const float *__restrict__ res,
                                                               float *__restrict__ y, int HW, int C, int groups, float eps,
                                                               int relu, const double *__restrict__ partial, int chunks) {
    __shared__ float s_mean[GN2_THREADS], s_rstd[GN2_THREADS];
    const int b = blockIdx.y, chunk = blockIdx.x, quads = C >> 2, step = GN2_THREADS / quads;
    if ((int)threadIdx.x < groups) {
        double S = 0.0, Q = 0.0;
        for (int c = 0; c < chunks; c++) {
            const double *o = partial + (((size_t)b * chunks + c) * groups + threadIdx.x) * 2;
            S += o[0];
            Q += o[1];
        }
        const double n = (double)HW * (C / groups), mean_d = S / n;
        double var = Q / n - mean_d * mean_d;
        var = var < 0.0 ? 0.0 : var;
        s_mean[threadIdx.x] = (float)mean_d;
        s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cq = threadIdx.x % quads, po = threadIdx.x / quads;
    if (po >= step) return;
    const int cg = C / groups, g0 = (cq << 2) / cg, g1 = ((cq << 2) + 2) / cg;      // groups of the quad's halves
    const f32x4 mean = {s_mean[g0], s_mean[g0], s_mean[g1], s_mean[g1]}, rstd = {s_rstd[g0], s_rstd[g0], s_rstd[g1], s_rstd[g1]};
    const f32x4 ga = reinterpret_cast<const f32x4 *>(gamma)[cq], be = reinterpret_cast<const f32x4 *>(beta)[cq];
    const int cp = gn2_chunk_pixels(C), p0 = chunk * cp, p1 = min(HW, p0 + cp);
    const size_t base = (size_t)b * HW * quads;
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x) + base;
    const f32x4 *rr = res ? reinterpret_cast<const f32x4 *>(res) + base : nullptr;
    f32x4 *yr = reinterpret_cast<f32x4 *>(y) + base;
    for (int p = p0 + po; p < p1; p += step) {
        const size_t o = (size_t)p * quads + cq;
        f32x4 v = (xr[o] - mean) * rstd * ga + be;
        if (rr) v += rr[o];
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        yr[o] = v;
    }
}

// ---- LayerNorm over the last dimension, one wave per row ----
__global__ __launch_bounds__(256) void layer_norm_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float *__restrict__ y,
                                                         int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / C + eps);
    for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}
// C % 4 == 0 and C <= 256 NV: the row stays in registers (NV float4 per lane) between the three sweeps - one read of x
template <int NV>
__global__ __launch_bounds__(256) void layer_norm_reg_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float *__restrict__ y,
                                                             int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, nq = C >> 2;
    if (row >= rows) return;
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + (size_t)row * C);
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int idx = lane + 64 * i;
        v[i] = idx < nq ? xr[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
        if (lane + 64 * i < nq) {
            const f32x4 d = v[i] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / C + eps);
    f32x4 *yr = reinterpret_cast<f32x4 *>(y + (size_t)row * C);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int idx = lane + 64 * i;
        if (idx < nq)
            yr[idx] = (v[i] - mean) * rstd * reinterpret_cast<const f32x4 *>(gamma)[idx] + reinterpret_cast<const f32x4 *>(beta)[idx];
    }
}

// ---- attention on the MFMA pipe: one wave per (sample, head, 32-query tile) ----
// Transposed formulation (the accumulator layout of one product is the operand layout of the
// next, no shuffles):  S^T[key][query] = K Q^T  (A = K tile, B = Q^T),  softmax over keys =
// over a lane's 16 registers + its partner half + key tiles (online),  O^T[d][query] = V^T P^T
// (A = V^T gathered in the accumulator's key order, B = P^T = the probabilities as they sit).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D>
__global__ __launch_bounds__(64) void attention_kernel(const float *__restrict__ qkv, float *__restrict__ out, int L,
                                                       int heads, float scale) {
    constexpr int DQ = D / 8;                 // float4 operand quads per lane along d
    constexpr int DT = D / 32;                // 32-row tiles of O^T
    const int lane = threadIdx.x, l32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, C = heads * D, q0 = blockIdx.y * 32;
    const float *base = qkv + (size_t)b * L * 3 * C + h * D;
    const int qrow = min(q0 + l32, L - 1);
    f32x4 qf[DQ];
#pragma unroll
    for (int t = 0; t < DQ; t++)
        qf[t] = *reinterpret_cast<const f32x4 *>(base + (size_t)qrow * 3 * C + 4 * (2 * t + half)) * scale;
    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] = 0.f;
    float mx = -INFINITY, den = 0.f;
    for (int k0 = 0; k0 < L; k0 += 32) {
        // S^T tile: rows = keys, columns = queries
        const int krow = min(k0 + l32, L - 1);
        f32x16 sT;
#pragma unroll
        for (int r = 0; r < 16; r++) sT[r] = 0.f;
#pragma unroll
        for (int t = 0; t < DQ; t++) {
            const f32x4 kf = *reinterpret_cast<const f32x4 *>(base + (size_t)krow * 3 * C + C + 4 * (2 * t + half));
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s4], qf[t][s4], sT, 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
            sT[r] = key < L ? sT[r] : -INFINITY;
            tmax = fmaxf(tmax, sT[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float nm = fmaxf(mx, tmax), corr = __expf(mx - nm);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sT[r] = __expf(sT[r] - nm);
            psum += sT[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        den = den * corr + psum;
        mx = nm;
        // O^T += V^T P^T: step r pairs key(r, half) on both operands
#pragma unroll
        for (int i = 0; i < DT; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) o[i][r] *= corr;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = min(k0 + 8 * (r >> 2) + 4 * half + (r & 3), L - 1);
                const float vf = base[(size_t)key * 3 * C + 2 * C + 32 * i + l32];
                o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sT[r], o[i], 0, 0, 0);
            }
        }
    }
    if (q0 + l32 < L) {
        const float inv = 1.0f / den;
        float *dst = out + ((size_t)b * L + q0 + l32) * C + h * D;
#pragma unroll
        for (int i = 0; i < DT; i++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 v = {o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
                *reinterpret_cast<f32x4 *>(dst + 32 * i + 8 * g + 4 * half) = v;
            }
    }
}

// The same attention in split-fp16 arithmetic (csrc/zs_split16.h: every product as three 16-bit MFMAs on hi / lo
// operand halves, ~2^-21 relative): 24 K = 16 MFMAs of 32 cycles per key tile instead of 64 K = 2 ones of 64.  The
// transposed formulation survives the wider contraction step because a lane's eight k values of one step may be any
// eight, as long as both operands use the same: for S^T they are two of the d quads it loads anyway, for O^T the keys
// of accumulator registers 8 s .. 8 s + 7 - the probabilities never leave the lane that computed them.
template <int D>
__global__ __launch_bounds__(64) void attention_split_kernel(const float *__restrict__ qkv, float *__restrict__ out, int L,
                                                             int heads, float scale) {
    using zs::s16::mfma3;
    using zs::s16::split8;
    constexpr int DS = D / 16;                // K = 16 steps along d
    constexpr int DT = D / 32;                // 32-row tiles of O^T
    const int lane = threadIdx.x, l32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, C = heads * D, q0 = blockIdx.y * 32;
    const float *base = qkv + (size_t)b * L * 3 * C + h * D;
    const int qrow = min(q0 + l32, L - 1);
    u32x4 qh[DS], ql[DS];                     // step s: d quads 4 s + half and 4 s + 2 + half of the query row, scaled
#pragma unroll
    for (int t = 0; t < DS; t++) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(base + (size_t)qrow * 3 * C + 4 * (4 * t + half)) * scale;
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(base + (size_t)qrow * 3 * C + 4 * (4 * t + 2 + half)) * scale;
        split8(a0, a1, qh[t], ql[t]);
    }
    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] = 0.f;
    float mx = -INFINITY, den = 0.f;
    for (int k0 = 0; k0 < L; k0 += 32) {
        const int krow = min(k0 + l32, L - 1);
        f32x16 sT;
#pragma unroll
        for (int r = 0; r < 16; r++) sT[r] = 0.f;
#pragma unroll
        for (int t = 0; t < DS; t++) {
            const f32x4 k0q = *reinterpret_cast<const f32x4 *>(base + (size_t)krow * 3 * C + C + 4 * (4 * t + half));
            const f32x4 k1q = *reinterpret_cast<const f32x4 *>(base + (size_t)krow * 3 * C + C + 4 * (4 * t + 2 + half));
            u32x4 kh, kl;
            split8(k0q, k1q, kh, kl);
            mfma3(sT, kh, kl, qh[t], ql[t]);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
            sT[r] = key < L ? sT[r] : -INFINITY;
            tmax = fmaxf(tmax, sT[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float nm = fmaxf(mx, tmax), corr = __expf(mx - nm);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sT[r] = __expf(sT[r] - nm);
            psum += sT[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        den = den * corr + psum;
        mx = nm;
        // O^T += V^T P^T: step s contracts the keys of accumulator registers 8 s .. 8 s + 7
        u32x4 ph[2], pl[2];
#pragma unroll
        for (int sidx = 0; sidx < 2; sidx++)
            split8(f32x4{sT[8 * sidx], sT[8 * sidx + 1], sT[8 * sidx + 2], sT[8 * sidx + 3]},
                   f32x4{sT[8 * sidx + 4], sT[8 * sidx + 5], sT[8 * sidx + 6], sT[8 * sidx + 7]}, ph[sidx], pl[sidx]);
#pragma unroll
        for (int i = 0; i < DT; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) o[i][r] *= corr;
#pragma unroll
            for (int sidx = 0; sidx < 2; sidx++) {
                float vf[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int r = 8 * sidx + e;
                    const int key = min(k0 + 8 * (r >> 2) + 4 * half + (r & 3), L - 1);
                    vf[e] = base[(size_t)key * 3 * C + 2 * C + 32 * i + l32];
                }
                u32x4 vh, vl;
                split8(f32x4{vf[0], vf[1], vf[2], vf[3]}, f32x4{vf[4], vf[5], vf[6], vf[7]}, vh, vl);
                mfma3(o[i], vh, vl, ph[sidx], pl[sidx]);
            }
        }
    }
    if (q0 + l32 < L) {
        const float inv = 1.0f / den;
        float *dst = out + ((size_t)b * L + q0 + l32) * C + h * D;
#pragma unroll
        for (int i = 0; i < DT; i++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 v = {o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
                *reinterpret_cast<f32x4 *>(dst + 32 * i + 8 * g + 4 * half) = v;
            }
    }
}

// ---- many sequences (batch 28; the window stage's 43,904 windows x heads): K and V staged ONCE per (sample, head) ----
// attention_split_kernel gives every 32-query tile a workgroup of one wave that loads - and splits - the sequence's whole K
// and V straight from global memory: at L = 197 seven waves fetch and split the same 100 KB (2,352 waves x 100 KB = 235 MB
// per ViT block at batch 28, 52 us).  Here one workgroup owns a (sample, head): its waves first stage K and V into LDS,
// ALREADY SPLIT into the fp16 halves and in the order the MFMA fragments want them (one conflict-free ds_read_b128 each),
// then take the query tiles qt = wave, wave + NW, ...  Per (query tile, key tile) the same MFMAs on the same operand bits in
// the same order as attention_split_kernel: bit-identical results.
//   K: row = key, 4 DS slots of 16 B: slot ((t * 2 + half) * 2 + plane) ^ swizzle(key) holds the hi (plane 0) / lo halves of the
//      eight d values {4 (4 t + half) .. + 3, 4 (4 t + 2 + half) .. + 3} - what lane (key, half) contracts in d-step t;
//   V: [key group g of 16][half][plane][d][16 B]: the eight keys 16 g + 4 half + {0..3, 8..11} of column d - what lane
//      (d, half) contracts in step g & 1 of key tile g / 2.
template <int D, int LP>
__global__ __launch_bounds__(64 * (LP / 32 < 8 ? LP / 32 : 8)) void attention_lds_kernel(const float *__restrict__ qkv, float *__restrict__ out,
                                                                                      int L, int heads, float scale) {
    using zs::s16::mfma3;
    using zs::s16::split8;
    constexpr int DS = D / 16, DT = D / 32, NW = LP / 32 < 8 ? LP / 32 : 8, SLOTS = 4 * DS;
    __shared__ u32x4 kl[LP * SLOTS];                      // LP x D x 4 bytes
    __shared__ u32x4 vl[(LP / 16) * 2 * 2 * D];           // the same size
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, C = heads * D;
    const float *base = qkv + (size_t)b * L * 3 * C + h * D;
    const int QT = (L + 31) / 32;
    auto swz = [](int key) { return D == 64 ? (key & 15) : ((key >> 1) & 7); };
    // ---- stage K: item = (key, t, half).  Compile-time trip counts (the padded image, keys beyond L clamp to L - 1): every
    // global load of the staging phase is in flight before the first split - one latency, not one per item (35.1 -> 33.0 us per ViT block at batch 28)
    constexpr int KIT = LP * DS * 2 / (64 * NW), VIT = (LP / 16) * 2 * D / (64 * NW);
    static_assert(KIT * 64 * NW == LP * DS * 2 && VIT * 64 * NW == (LP / 16) * 2 * D, "staging items divide evenly");
    f32x4 kq[KIT][2];
    float vf[VIT][8];
#pragma unroll
    for (int u = 0; u < KIT; u++) {
        const int it = tid + u * 64 * NW;
        const int hf = it & 1, t = (it >> 1) % DS, key = it / (2 * DS), kr = min(key, L - 1);
        kq[u][0] = *reinterpret_cast<const f32x4 *>(base + (size_t)kr * 3 * C + C + 4 * (4 * t + hf));
        kq[u][1] = *reinterpret_cast<const f32x4 *>(base + (size_t)kr * 3 * C + C + 4 * (4 * t + 2 + hf));
    }
#pragma unroll
    for (int u = 0; u < VIT; u++) {
        const int it = tid + u * 64 * NW;
        const int d = it % D, hf = (it / D) & 1, g = it / (2 * D);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int key = min(16 * g + 4 * hf + (e & 3) + 8 * (e >> 2), L - 1);
            vf[u][e] = base[(size_t)key * 3 * C + 2 * C + d];
        }
    }
#pragma unroll
    for (int u = 0; u < KIT; u++) {
        const int it = tid + u * 64 * NW;
        const int hf = it & 1, t = (it >> 1) % DS, key = it / (2 * DS);
        u32x4 hi, lo;
        split8(kq[u][0], kq[u][1], hi, lo);
        const int s0 = (t * 2 + hf) * 2;
        kl[key * SLOTS + (s0 ^ swz(key))] = hi;
        kl[key * SLOTS + ((s0 + 1) ^ swz(key))] = lo;
    }
#pragma unroll
    for (int u = 0; u < VIT; u++) {
        const int it = tid + u * 64 * NW;
        const int d = it % D, hf = (it / D) & 1, g = it / (2 * D);
        u32x4 hi, lo;
        split8(f32x4{vf[u][0], vf[u][1], vf[u][2], vf[u][3]}, f32x4{vf[u][4], vf[u][5], vf[u][6], vf[u][7]}, hi, lo);
        vl[((g * 2 + hf) * 2 + 0) * D + d] = hi;
        vl[((g * 2 + hf) * 2 + 1) * D + d] = lo;
    }
    __syncthreads();
    for (int qt = wave; qt < QT; qt += NW) {
        const int q0 = qt * 32, qrow = min(q0 + l32, L - 1);
        u32x4 qh[DS], ql[DS];
#pragma unroll
        for (int t = 0; t < DS; t++) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(base + (size_t)qrow * 3 * C + 4 * (4 * t + half)) * scale;
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(base + (size_t)qrow * 3 * C + 4 * (4 * t + 2 + half)) * scale;
            split8(a0, a1, qh[t], ql[t]);
        }
        f32x16 o[DT];
#pragma unroll
        for (int i = 0; i < DT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[i][r] = 0.f;
        float mx = -INFINITY, den = 0.f;
        for (int kt = 0; kt < QT; kt++) {
            const int k0 = kt * 32, key = k0 + l32;
            f32x16 sT;
#pragma unroll
            for (int r = 0; r < 16; r++) sT[r] = 0.f;
#pragma unroll
            for (int t = 0; t < DS; t++) {
                const int s0 = (t * 2 + half) * 2;
                const u32x4 kh = kl[key * SLOTS + (s0 ^ swz(key))], klo = kl[key * SLOTS + ((s0 + 1) ^ swz(key))];
                mfma3(sT, kh, klo, qh[t], ql[t]);
            }
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int kk = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
                sT[r] = kk < L ? sT[r] : -INFINITY;
                tmax = fmaxf(tmax, sT[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float nm = fmaxf(mx, tmax), corr = __expf(mx - nm);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                sT[r] = __expf(sT[r] - nm);
                psum += sT[r];
            }
            psum += __shfl_xor(psum, 32, 64);
            den = den * corr + psum;
            mx = nm;
            u32x4 ph[2], pl[2];
#pragma unroll
            for (int sidx = 0; sidx < 2; sidx++)
                split8(f32x4{sT[8 * sidx], sT[8 * sidx + 1], sT[8 * sidx + 2], sT[8 * sidx + 3]},
                       f32x4{sT[8 * sidx + 4], sT[8 * sidx + 5], sT[8 * sidx + 6], sT[8 * sidx + 7]}, ph[sidx], pl[sidx]);
#pragma unroll
            for (int i = 0; i < DT; i++) {
#pragma unroll
                for (int r = 0; r < 16; r++) o[i][r] *= corr;
#pragma unroll
                for (int sidx = 0; sidx < 2; sidx++) {
                    const int g = kt * 2 + sidx, d = 32 * i + l32;
                    const u32x4 vh = vl[((g * 2 + half) * 2 + 0) * D + d], vlo = vl[((g * 2 + half) * 2 + 1) * D + d];
                    mfma3(o[i], vh, vlo, ph[sidx], pl[sidx]);
                }
            }
        }
        if (q0 + l32 < L) {
            const float inv = 1.0f / den;
            float *dst = out + ((size_t)b * L + q0 + l32) * C + h * D;
#pragma unroll
            for (int i = 0; i < DT; i++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const f32x4 v = {o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
                    *reinterpret_cast<f32x4 *>(dst + 32 * i + 8 * g + 4 * half) = v;
                }
        }
    }
}

// ---- few sequences (batch 1): the key tiles of one (sample, head, 32-query tile) spread over the waves of a workgroup ----
// attention_split_kernel walks its key tiles one after the other, each a dependent pair of load round trips (K, then V):
// 7 tiles at L = 197 = 21 us for 84 waves on 256 CUs.  Here every wave takes the key tiles kt = wave, wave + KW, ... (one
// each at L <= 256), so all K / V loads of the workgroup are in flight at once; the partial (max, sum, O) triples are merged
// through LDS in wave order (the flash-decoding merge: the same sums in another association, ~1e-7 relative).
constexpr int ATT_KW = 8;                      // waves per workgroup
template <int D>
__global__ __launch_bounds__(64 * ATT_KW) void attention_split_kw_kernel(const float *__restrict__ qkv, float *__restrict__ out, int L,
                                                                         int heads, float scale) {
    using zs::s16::mfma3;
    using zs::s16::split8;
    constexpr int DS = D / 16, DT = D / 32, PADD = D + 4;
    __shared__ __attribute__((aligned(16))) float opart[ATT_KW][32][PADD];
    __shared__ float mpart[ATT_KW][32], dpart[ATT_KW][32];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, C = heads * D, q0 = blockIdx.y * 32;
    const float *base = qkv + (size_t)b * L * 3 * C + h * D;
    const int qrow = min(q0 + l32, L - 1);
    u32x4 qh[DS], ql[DS];
#pragma unroll
    for (int t = 0; t < DS; t++) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(base + (size_t)qrow * 3 * C + 4 * (4 * t + half)) * scale;
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(base + (size_t)qrow * 3 * C + 4 * (4 * t + 2 + half)) * scale;
        split8(a0, a1, qh[t], ql[t]);
    }
    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] = 0.f;
    float mx = -INFINITY, den = 0.f;
    for (int k0 = 32 * wave; k0 < L; k0 += 32 * ATT_KW) {
        const int krow = min(k0 + l32, L - 1);
        // all of this tile's loads first (K rows as quads, V columns as the accumulator's keys), then the arithmetic
        f32x4 kq[DS][2];
#pragma unroll
        for (int t = 0; t < DS; t++) {
            kq[t][0] = *reinterpret_cast<const f32x4 *>(base + (size_t)krow * 3 * C + C + 4 * (4 * t + half));
            kq[t][1] = *reinterpret_cast<const f32x4 *>(base + (size_t)krow * 3 * C + C + 4 * (4 * t + 2 + half));
        }
        float vf[DT][16];
#pragma unroll
        for (int i = 0; i < DT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = min(k0 + 8 * (r >> 2) + 4 * half + (r & 3), L - 1);
                vf[i][r] = base[(size_t)key * 3 * C + 2 * C + 32 * i + l32];
            }
        f32x16 sT;
#pragma unroll
        for (int r = 0; r < 16; r++) sT[r] = 0.f;
#pragma unroll
        for (int t = 0; t < DS; t++) {
            u32x4 kh, kl;
            split8(kq[t][0], kq[t][1], kh, kl);
            mfma3(sT, kh, kl, qh[t], ql[t]);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
            sT[r] = key < L ? sT[r] : -INFINITY;
            tmax = fmaxf(tmax, sT[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float nm = fmaxf(mx, tmax), corr = __expf(mx - nm);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sT[r] = __expf(sT[r] - nm);
            psum += sT[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        den = den * corr + psum;
        mx = nm;
        u32x4 ph[2], pl[2];
#pragma unroll
        for (int sidx = 0; sidx < 2; sidx++)
            split8(f32x4{sT[8 * sidx], sT[8 * sidx + 1], sT[8 * sidx + 2], sT[8 * sidx + 3]},
                   f32x4{sT[8 * sidx + 4], sT[8 * sidx + 5], sT[8 * sidx + 6], sT[8 * sidx + 7]}, ph[sidx], pl[sidx]);
#pragma unroll
        for (int i = 0; i < DT; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) o[i][r] *= corr;
#pragma unroll
            for (int sidx = 0; sidx < 2; sidx++) {
                u32x4 vh, vl;
                split8(f32x4{vf[i][8 * sidx], vf[i][8 * sidx + 1], vf[i][8 * sidx + 2], vf[i][8 * sidx + 3]},
                       f32x4{vf[i][8 * sidx + 4], vf[i][8 * sidx + 5], vf[i][8 * sidx + 6], vf[i][8 * sidx + 7]}, vh, vl);
                mfma3(o[i], vh, vl, ph[sidx], pl[sidx]);
            }
        }
    }
    // partial (max, sum, O^T) of this wave's keys -> LDS: register 4 g + e of lane (query l32, half) = d 32 i + 8 g + 4 half + e
#pragma unroll
    for (int i = 0; i < DT; i++)
#pragma unroll
        for (int g = 0; g < 4; g++)
            *reinterpret_cast<f32x4 *>(&opart[wave][l32][32 * i + 8 * g + 4 * half]) =
                f32x4{o[i][4 * g], o[i][4 * g + 1], o[i][4 * g + 2], o[i][4 * g + 3]};
    if (half == 0) {
        mpart[wave][l32] = mx;
        dpart[wave][l32] = den;
    }
    __syncthreads();
    for (int e = tid; e < 32 * (D / 4); e += 64 * ATT_KW) {
        const int q = e / (D / 4), dq = e % (D / 4);
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < ATT_KW; w++) M = fmaxf(M, mpart[w][q]);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float dsum = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_KW; w++) {
            const float f = mpart[w][q] == -INFINITY ? 0.f : __expf(mpart[w][q] - M);
            dsum += f * dpart[w][q];
            acc += f * *reinterpret_cast<const f32x4 *>(&opart[w][q][4 * dq]);
        }
        if (q0 + q < L) *reinterpret_cast<f32x4 *>(out + ((size_t)b * L + q0 + q) * C + h * D + 4 * dq) = acc * (1.0f / dsum);
    }
}

// ---- pooling / resampling / layout ----
__global__ __launch_bounds__(256) void max_pool_kernel(const float *__restrict__ x, float *__restrict__ y, int B,
                                                       int Hin, int Win, int C, int Hout, int Wout, int k, int stride,
                                                       int pad_t, int pad_l) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * Hout * Wout * C;
    if (i >= total) return;
    const int c = i % C, ox = (i / C) % Wout, oy = (i / C / Wout) % Hout, b = i / C / Wout / Hout;
    float m = -INFINITY;
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) {
            const int iy = oy * stride - pad_t + ky, ix = ox * stride - pad_l + kx;
            if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win)
                m = fmaxf(m, x[(((size_t)b * Hin + iy) * Win + ix) * C + c]);
        }
    y[i] = m;
}

// the same, four channels per lane and 32-bit index arithmetic (C % 4 == 0, fewer than 2^31 output quads): fmaxf per component in
// the same tap order - the same values
__global__ __launch_bounds__(256) void max_pool_quad_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, unsigned total,
                                                            int Hin, int Win, int C4, int Hout, int Wout, int k, int stride,
                                                            int pad_t, int pad_l) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned c = i % C4, p = i / C4, ox = p % Wout, q = p / Wout, oy = q % Hout, b = q / Hout;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) {
            const int iy = (int)oy * stride - pad_t + ky, ix = (int)ox * stride - pad_l + kx;
            if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
                const f32x4 v = x[(((size_t)b * Hin + iy) * Win + ix) * C4 + c];
#pragma unroll
                for (int e = 0; e < 4; e++) m[e] = fmaxf(m[e], v[e]);
            }
        }
    y[i] = m;
}

// 64 channels x 4 pixel slices per workgroup (a lane per channel walking all pixels alone was 13 us for a 7 x 7 map: a chain of
// dependent adds on 8 workgroups); the four slice sums are added in slice order
__global__ __launch_bounds__(256) void global_mean_kernel(const float *__restrict__ x, float *__restrict__ y, int HW,
                                                          int C) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, slice = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (int p = slice; p < HW; p += 4) s += x[((size_t)b * HW + p) * C + c];
    part[slice][cl] = s;
    __syncthreads();
    if (slice == 0 && c < C) y[(size_t)b * C + c] = ((part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl])) / HW;
}

// ---- GroupNorm(32) + ReLU riding on the max pool behind it (timm ResNetV2 stem: conv -> GroupNormAct -> MaxPool2dSame) ----
// The stem convolution's epilogue wrote per-tile group sums (zs_conv_fuse.out_mode 1); gn_table_kernel turns them into one
// (scale, shift) per channel (one small workgroup per sample: 392 tiles at 112 x 112 would be too much for every pooling
// workgroup to re-reduce), max_pool_affine_kernel pools relu(x * scale + shift): the normalised map is never written.
__global__ __launch_bounds__(256) void gn_table_kernel(const float *__restrict__ stats, int tiles, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ table, int HW, int C,
                                                       int gshift, float eps) {
    __shared__ double red[8][32][2];
    __shared__ float stat[32][2];
    const int tid = threadIdx.x, sample = blockIdx.x, g = tid & 31, slice = tid >> 5;
    double S = 0.0, Q = 0.0;
    for (int tl = slice; tl < tiles; tl += 8) {
        const float2 e = *reinterpret_cast<const float2 *>(stats + ((size_t)(sample * tiles + tl) * 32 + g) * 2);
        S += (double)e.x;
        Q += (double)e.y;
    }
    red[slice][g][0] = S;
    red[slice][g][1] = Q;
    __syncthreads();
    if (tid < 32) {
        double s2 = 0.0, q2 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) { s2 += red[k][tid][0]; q2 += red[k][tid][1]; }
        const double cnt = (double)HW * (double)(1 << gshift), mean = s2 / cnt, var = fmax(q2 / cnt - mean * mean, 0.0);
        stat[tid][0] = (float)mean;
        stat[tid][1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float sc = gamma[c] * stat[c >> gshift][1];
        table[((size_t)sample * C + c) * 2] = sc;
        table[((size_t)sample * C + c) * 2 + 1] = beta[c] - stat[c >> gshift][0] * sc;
    }
}

__global__ __launch_bounds__(256) void max_pool_affine_kernel(const float *__restrict__ x, const float *__restrict__ table,
                                                              float *__restrict__ y, int B, int Hin, int Win, int C, int Hout,
                                                              int Wout, int k, int stride, int pad_t, int pad_l) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * Hout * Wout * C;
    if (i >= total) return;
    const int c = i % C, ox = (i / C) % Wout, oy = (i / C / Wout) % Hout, b = i / C / Wout / Hout;
    const float sc = table[((size_t)b * C + c) * 2], sh = table[((size_t)b * C + c) * 2 + 1];
    float m = -INFINITY;
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) {
            const int iy = oy * stride - pad_t + ky, ix = ox * stride - pad_l + kx;
            if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win)
                m = fmaxf(m, fmaxf(x[(((size_t)b * Hin + iy) * Win + ix) * C + c] * sc + sh, 0.f));
        }
    y[i] = m;
}

__global__ __launch_bounds__(256) void upsample2x_kernel(const float *__restrict__ x, float *__restrict__ y, int B,
                                                         int Hin, int Win, int C) {
    const int Hout = 2 * Hin, Wout = 2 * Win;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * Hout * Wout * C;
    if (i >= total) return;
    const int c = i % C, ox = (i / C) % Wout, oy = (i / C / Wout) % Hout, b = i / C / Wout / Hout;
    // torch upsample_bilinear2d, align_corners=True: src = dst * (in - 1) / (out - 1)
    const float sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f, sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    const float fy = sy * oy, fx = sx * ox;
    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float *X = x + (size_t)b * Hin * Win * C + c;
    const float v00 = X[((size_t)y0 * Win + x0) * C], v01 = X[((size_t)y0 * Win + x1) * C],
                v10 = X[((size_t)y1 * Win + x0) * C], v11 = X[((size_t)y1 * Win + x1) * C];
    y[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// the same, four channels per lane and no 64-bit divisions: grid (ceil(Wout * C / 4 / 256), Hout, B)
__global__ __launch_bounds__(256) void upsample2x_vec_kernel(const float *__restrict__ x, float *__restrict__ y, int Hin,
                                                             int Win, int C) {
    const int Hout = 2 * Hin, Wout = 2 * Win, cq = C >> 2;
    const int j = blockIdx.x * 256 + threadIdx.x;                  // (ox, channel quad)
    if (j >= Wout * cq) return;
    const int ox = j / cq, c = (j - ox * cq) * 4, oy = blockIdx.y, b = blockIdx.z;
    const float sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f, sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    const float fy = sy * oy, fx = sx * ox;
    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float *X = x + (size_t)b * Hin * Win * C + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4 *>(X + ((size_t)y0 * Win + x0) * C),
                v01 = *reinterpret_cast<const f32x4 *>(X + ((size_t)y0 * Win + x1) * C),
                v10 = *reinterpret_cast<const f32x4 *>(X + ((size_t)y1 * Win + x0) * C),
                v11 = *reinterpret_cast<const f32x4 *>(X + ((size_t)y1 * Win + x1) * C);
    *reinterpret_cast<f32x4 *>(y + (((size_t)b * Hout + oy) * Wout + ox) * C + c) =
        (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// NCHW [B][C][H][W] -> NHWC [B][H][W][Cpad] (extra channels zero), optionally times mask [B][HW]
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ x,
                                                           const float *__restrict__ mask, float *__restrict__ y,
                                                           int B, int C, int HW, int Cpad) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * HW * Cpad;
    if (i >= total) return;
    const int c = i % Cpad, p = (i / Cpad) % HW, b = i / Cpad / HW;
    float v = c < C ? x[((size_t)b * C + c) * HW + p] : 0.f;
    if (mask) v *= mask[(size_t)b * HW + p];
    y[i] = v;
}

// CoordEmb's token preparation (seen_coord_enc.py:50-71): emb [B][H][W][C] with invalid pixels
// replaced by a learned token, cut into win x win windows, + window-local position embedding,
// cls token (+ its embedding) prepended: out [B*(H/win)*(W/win)][win*win+1][C]
__global__ __launch_bounds__(256) void window_tokens_kernel(const float *__restrict__ emb,
                                                            const uint8_t *__restrict__ mask,
                                                            const float *__restrict__ invalid,
                                                            const float *__restrict__ cls,
                                                            const float *__restrict__ pos, float *__restrict__ out,
                                                            int B, int H, int W, int C, int win) {
    const int T = win * win + 1, nwx = W / win, nwy = H / win;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * nwy * nwx * T * C;
    if (i >= total) return;
    const int c = i % C, t = (i / C) % T;
    const size_t wdx = i / C / T;
    const int wx = wdx % nwx, wy = (wdx / nwx) % nwy, b = wdx / nwx / nwy;
    float v;
    if (t == 0) v = cls[c];
    else {
        const int py = wy * win + (t - 1) / win, px = wx * win + (t - 1) % win;
        const size_t pix = ((size_t)b * H + py) * W + px;
        v = mask[pix] ? emb[pix * C + c] : invalid[c];
    }
    out[i] = v + pos[(size_t)t * C + c];
}
// The same, four channels per lane and 32-bit index arithmetic (C % 4 == 0, 256 % (C / 4) == 0, fewer than 2^31 quads): the
// scalar form above spends its time in 64-bit divisions - 390 us for the 356,720 x 256 window tokens of batch 28, against
// ~180 us of HBM traffic (round 5)
__global__ __launch_bounds__(256) void window_tokens_quad_kernel(const f32x4 *__restrict__ emb, const uint8_t *__restrict__ mask,
                                                                 const f32x4 *__restrict__ invalid, const f32x4 *__restrict__ cls,
                                                                 const f32x4 *__restrict__ pos, f32x4 *__restrict__ out,
                                                                 unsigned tokens, int H, int W, int C4, int win) {
    const unsigned T = win * win + 1, nwx = W / win, nwy = H / win, per = 256 / C4;
    const unsigned tok = blockIdx.x * per + threadIdx.x / C4, c = threadIdx.x % C4;
    if (tok >= tokens) return;
    const unsigned t = tok % T, wdx = tok / T, wx = wdx % nwx, wy = (wdx / nwx) % nwy, b = wdx / nwx / nwy;
    f32x4 v;
    if (t == 0) v = cls[c];
    else {
        const unsigned py = wy * win + (t - 1) / win, px = wx * win + (t - 1) % win;
        const size_t pix = ((size_t)b * H + py) * W + px;
        v = mask[pix] ? emb[pix * C4 + c] : invalid[c];
    }
    out[(size_t)tok * C4 + c] = v + pos[(size_t)t * C4 + c];
}
// NHWC [B][H][W][C] -> NCHW [B][C][H][W]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ x, float *__restrict__ y, int B,
                                                           int C, int HW) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * HW * C;
    if (i >= total) return;
    const int p = i % HW, c = (i / HW) % C, b = i / HW / C;
    y[i] = x[((size_t)b * HW + p) * C + c];
}

// tokens[b][0] = cls + pos[0]; tokens[b][1+i] = feat[b][i] + pos[1+i]   (vit.py:139-147)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float *__restrict__ feat,
                                                              const float *__restrict__ cls,
                                                              const float *__restrict__ pos, float *__restrict__ tok,
                                                              int B, int n, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * (n + 1) * C;
    if (i >= total) return;
    const int c = i % C, t = (i / C) % (n + 1), b = i / C / (n + 1);
    tok[i] = (t == 0 ? cls[c] : feat[((size_t)b * n + t - 1) * C + c]) + pos[(size_t)t * C + c];
}

// ProjectReadout's concatenation (vit.py:39-41): out[b][i] = [tok[b][1+i] | tok[b][0]]
__global__ __launch_bounds__(256) void readout_concat_kernel(const float *__restrict__ tok, float *__restrict__ out,
                                                             int B, int n, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * n * 2 * C;
    if (i >= total) return;
    const int c = i % (2 * C), t = (i / (2 * C)) % n, b = i / (2 * C) / n;
    out[i] = c < C ? tok[((size_t)b * (n + 1) + 1 + t) * C + c] : tok[(size_t)b * (n + 1) * C + (c - C)];
}

inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }
inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }

}  // namespace

// ---- GroupNorm from statistics a convolution's epilogue already wrote (zs_conv_fuse.out_mode 1): ONE pass ----
// Every workgroup turns the per-tile group sums of its sample into (mean, rstd) itself (a few KB, the same fixed-order
// sum everywhere), then streams its share of the tensor: y = [relu](x * sc_c + sh_c + r [* ru_c]).
__global__ __launch_bounds__(256) void gn_apply_stats_kernel(const float *__restrict__ x, const float *__restrict__ stats, int tiles,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             const float *__restrict__ res, const float *__restrict__ rstats, int rtiles,
                                                             const float *__restrict__ rgamma, const float *__restrict__ rbeta,
                                                             float *__restrict__ y, int HW, int C, int gshift, float eps, int relu,
                                                             int blocks_per_sample) {
    __shared__ float red[8][32][2], stat[2][32][2];
    const int tid = threadIdx.x, sample = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
    const double cnt = (double)HW * (double)(1 << gshift);
    for (int which = 0; which < (rstats ? 2 : 1); which++) {
        const float *st = which ? rstats : stats;
        const int nt = which ? rtiles : tiles, g = tid & 31, slice = tid >> 5;
        float S = 0.f, Q = 0.f;
        for (int tl = slice; tl < nt; tl += 8) {
            const float2 e = *reinterpret_cast<const float2 *>(st + ((size_t)(sample * nt + tl) * 32 + g) * 2);
            S += e.x;
            Q += e.y;
        }
        red[slice][g][0] = S;
        red[slice][g][1] = Q;
        __syncthreads();
        if (tid < 32) {
            double s2 = 0.0, q2 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) { s2 += (double)red[k][tid][0]; q2 += (double)red[k][tid][1]; }
            const double mean = s2 / cnt, var = fmax(q2 / cnt - mean * mean, 0.0);
            stat[which][tid][0] = (float)mean;
            stat[which][tid][1] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
    }
    const int cq = C >> 2;                                      // channel quads per pixel
    const size_t total = (size_t)HW * cq, per = (total + blocks_per_sample - 1) / blocks_per_sample;
    const size_t lo = (size_t)blk * per, hi = lo + per < total ? lo + per : total;
    const f32x4 *xs = reinterpret_cast<const f32x4 *>(x) + (size_t)sample * total;
    const f32x4 *rs = res ? reinterpret_cast<const f32x4 *>(res) + (size_t)sample * total : nullptr;
    f32x4 *ys = reinterpret_cast<f32x4 *>(y) + (size_t)sample * total;
    for (size_t i = lo + tid; i < hi; i += 256) {
        const int c = (int)(i % cq) * 4, g = c >> gshift;            // a quad lies inside one group (>= 4 channels per group) ...
        f32x4 v = xs[i];
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + c), be = *reinterpret_cast<const f32x4 *>(beta + c);
        f32x4 r = {0.f, 0.f, 0.f, 0.f}, rg = {1.f, 1.f, 1.f, 1.f}, rb = {0.f, 0.f, 0.f, 0.f};
        if (rs) r = rs[i];
        if (rstats) { rg = *reinterpret_cast<const f32x4 *>(rgamma + c); rb = *reinterpret_cast<const f32x4 *>(rbeta + c); }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int ge = gshift >= 2 ? g : (c + e) >> gshift;      // ... or two groups per quad (2 channels per group)
            const float sc = ga[e] * stat[0][ge][1];
            float o = v[e] * sc + (be[e] - stat[0][ge][0] * sc);
            if (rstats) {
                const float rsc = rg[e] * stat[1][ge][1];
                o += r[e] * rsc + (rb[e] - stat[1][ge][0] * rsc);
            } else {
                o += r[e];
            }
            v[e] = relu ? fmaxf(o, 0.f) : o;
        }
        ys[i] = v;
    }
}

#define ZS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            zs::set_err(__VA_ARGS__);    \
            return 0;                    \
        }                                \
    } while (0)

extern "C" size_t zs_group_norm_workspace_bytes(int batch, int HW, int C, int groups) {
    if (batch <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
    const int chunks = (HW + gn2_chunk_pixels(C) - 1) / gn2_chunk_pixels(C);
    return (size_t)batch * chunks * groups * 2 * sizeof(double);
}

// workspace (may be NULL: the one-launch kernels; else >= zs_group_norm_workspace_bytes): enables the coalesced two-launch
// form for tensors of >= 8 MiB with 2, 4 or 8 channels per group (narrower slices than a cache line per pixel: where the
// one-launch kernels waste most of every line; from 16 channels per group on they are as fast - measured, tools/bench_gn.py)
extern "C" int zs_group_norm_nhwc_ws(const float *x, const float *gamma, const float *beta, const float *residual,
                                     float *y, int batch, int HW, int C, int groups, float eps, int relu, void *workspace,
                                     void *stream) {
    ZS_REQUIRE(batch >= 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0,
               "zs_group_norm_nhwc: bad size (B=%d HW=%d C=%d groups=%d)", batch, HW, C, groups);
    if (batch == 0) return 1;
    ZS_REQUIRE(x && gamma && beta && y, "zs_group_norm_nhwc: null pointer");
    static const bool no_two = getenv("ZS_GN_ONE_LAUNCH") != nullptr;          // A/B switch for measurements
    const int cg0 = C / groups;
    if (workspace && !no_two && (C & 3) == 0 && C <= 4 * GN2_THREADS && (cg0 == 2 || cg0 == 4 || cg0 == 8) &&
        groups <= GN2_THREADS && GN2_THREADS % (C >> 2) == 0 && batch <= 65535 &&
        (size_t)batch * HW * C * sizeof(float) >= ((size_t)8 << 20)) {
        const int chunks = (HW + gn2_chunk_pixels(C) - 1) / gn2_chunk_pixels(C);
        double *partial = static_cast<double *>(workspace);
        hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, batch), dim3(GN2_THREADS), 0, S(stream), x, HW, C, groups, partial,
                           chunks);
        hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, batch), dim3(GN2_THREADS), 0, S(stream), x, gamma, beta, residual, y,
                           HW, C, groups, eps, relu, static_cast<const double *>(partial), chunks);
        return zs::check_launch("zs_group_norm_nhwc") ? 1 : 0;
    }
    return zs_group_norm_nhwc(x, gamma, beta, residual, y, batch, HW, C, groups, eps, relu, stream);
}

extern "C" int zs_group_norm_nhwc(const float *x, const float *gamma, const float *beta, const float *residual,
                                  float *y, int batch, int HW, int C, int groups, float eps, int relu, void *stream) {
    ZS_REQUIRE(batch >= 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0,
               "zs_group_norm_nhwc: bad size (B=%d HW=%d C=%d groups=%d)", batch, HW, C, groups);
    if (batch == 0) return 1;
    ZS_REQUIRE(x && gamma && beta && y, "zs_group_norm_nhwc: null pointer");
    const int cg = C / groups;
    const bool pow2 = (cg & (cg - 1)) == 0 && (long long)HW * cg < (1LL << 30);
    static const bool gn_generic = getenv("ZS_GN_GENERIC") != nullptr;         // A/B switch for measurements
    if (!pow2 || gn_generic) {
        hipLaunchKernelGGL(group_norm_kernel, dim3(batch * groups), dim3(GN_BLOCK), 0, S(stream), x, gamma, beta, residual,
                           y, HW, C, groups, eps, relu);
        return zs::check_launch("zs_group_norm_nhwc") ? 1 : 0;
    }
    const int vec = cg >= 4 ? 4 : cg;
    int shift = 0;
    while ((vec << shift) < cg) shift++;
    const size_t bytes = (size_t)HW * cg * sizeof(float);
    const bool cache = bytes <= (size_t)GN_CACHE_BYTES;
#define ZS_GN_LAUNCH(V, CA)                                                                                          \
    do {                                                                                                             \
        if (CA) {                                                                                                    \
            static bool once = false;                                                                                \
            if (!once) {                                                                                             \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(group_norm_pow2_kernel<V, CA>),            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, GN_CACHE_BYTES);               \
                once = true;                                                                                         \
            }                                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL((group_norm_pow2_kernel<V, CA>), dim3(batch * groups), dim3(GN_BLOCK), (CA) ? bytes : 0,  \
                           S(stream), x, gamma, beta, residual, y, HW, C, groups, shift, eps, relu);                 \
    } while (0)
    if (vec == 4) { if (cache) ZS_GN_LAUNCH(4, true); else ZS_GN_LAUNCH(4, false); }
    else if (vec == 2) { if (cache) ZS_GN_LAUNCH(2, true); else ZS_GN_LAUNCH(2, false); }
    else { if (cache) ZS_GN_LAUNCH(1, true); else ZS_GN_LAUNCH(1, false); }
#undef ZS_GN_LAUNCH
    return zs::check_launch("zs_group_norm_nhwc") ? 1 : 0;
}

extern "C" int zs_group_norm_apply_stats(const float *x, const float *stats, int tiles, const float *gamma, const float *beta,
                                         const float *residual, const float *res_stats, int res_tiles, const float *res_gamma,
                                         const float *res_beta, float *y, int batch, int HW, int C, float eps, int relu,
                                         void *stream) {
    int gshift = 0;
    while ((32 << gshift) < C) gshift++;
    ZS_REQUIRE(batch >= 0 && HW > 0 && C >= 64 && (32 << gshift) == C && tiles > 0,
               "zs_group_norm_apply_stats: bad size (B=%d HW=%d C=%d tiles=%d; C = 32 * 2^k >= 64)", batch, HW, C, tiles);
    if (batch == 0) return 1;
    ZS_REQUIRE(x && stats && gamma && beta && y, "zs_group_norm_apply_stats: null pointer");
    ZS_REQUIRE(!res_stats || (residual && res_gamma && res_beta && res_tiles > 0), "zs_group_norm_apply_stats: residual statistics need residual, gamma, beta, tiles");
    const size_t quads = (size_t)HW * (C >> 2);
    int bps = (int)((quads + 1023) / 1024);                     // ~4 quads per thread
    if (bps > 1024) bps = 1024;
    hipLaunchKernelGGL(gn_apply_stats_kernel, dim3((unsigned)(batch * bps)), dim3(256), 0, S(stream), x, stats, tiles, gamma, beta,
                       residual, res_stats, res_tiles, res_gamma, res_beta, y, HW, C, gshift, eps, relu, bps);
    return zs::check_launch("zs_group_norm_apply_stats") ? 1 : 0;
}

extern "C" int zs_gn_relu_max_pool_nhwc(const float *x, const float *stats, int tiles, const float *gamma, const float *beta,
                                        float *table, float *y, int batch, int Hin, int Win, int C, int Hout, int Wout, int k,
                                        int stride, int pad_t, int pad_l, float eps, void *stream) {
    int gshift = 0;
    while ((32 << gshift) < C) gshift++;
    ZS_REQUIRE(batch >= 0 && Hin > 0 && Win > 0 && C >= 32 && (32 << gshift) == C && tiles > 0 && Hout > 0 && Wout > 0 && k > 0 && stride > 0,
               "zs_gn_relu_max_pool_nhwc: bad size (B=%d %dx%dx%d tiles=%d; C = 32 * 2^k)", batch, Hin, Win, C, tiles);
    if (batch == 0) return 1;
    ZS_REQUIRE(x && stats && gamma && beta && table && y, "zs_gn_relu_max_pool_nhwc: null pointer");
    hipLaunchKernelGGL(gn_table_kernel, dim3(batch), dim3(256), 0, S(stream), stats, tiles, gamma, beta, table, Hin * Win, C, gshift, eps);
    hipLaunchKernelGGL(max_pool_affine_kernel, dim3(blocks_for((size_t)batch * Hout * Wout * C)), dim3(256), 0, S(stream), x, table, y,
                       batch, Hin, Win, C, Hout, Wout, k, stride, pad_t, pad_l);
    return zs::check_launch("zs_gn_relu_max_pool_nhwc") ? 1 : 0;
}

extern "C" int zs_layer_norm(const float *x, const float *gamma, const float *beta, float *y, int rows, int C,
                             float eps, void *stream) {
    ZS_REQUIRE(rows >= 0 && C > 0, "zs_layer_norm: bad size (rows=%d C=%d)", rows, C);
    if (rows == 0) return 1;
    ZS_REQUIRE(x && gamma && beta && y, "zs_layer_norm: null pointer");
    const dim3 grid((rows + 3) / 4);
    if ((C & 3) == 0 && C <= 256)
        hipLaunchKernelGGL(layer_norm_reg_kernel<1>, grid, dim3(256), 0, S(stream), x, gamma, beta, y, rows, C, eps);
    else if ((C & 3) == 0 && C <= 512)
        hipLaunchKernelGGL(layer_norm_reg_kernel<2>, grid, dim3(256), 0, S(stream), x, gamma, beta, y, rows, C, eps);
    else if ((C & 3) == 0 && C <= 1024)
        hipLaunchKernelGGL(layer_norm_reg_kernel<4>, grid, dim3(256), 0, S(stream), x, gamma, beta, y, rows, C, eps);
    else
        hipLaunchKernelGGL(layer_norm_kernel, grid, dim3(256), 0, S(stream), x, gamma, beta, y, rows, C, eps);
    return zs::check_launch("zs_layer_norm") ? 1 : 0;
}

extern "C" int zs_attention_split(const float *qkv, float *out, int batch, int L, int heads, int head_dim, void *stream) {
    ZS_REQUIRE(batch >= 0 && L > 0 && heads > 0 && (head_dim == 32 || head_dim == 64),
               "zs_attention_split: bad size (B=%d L=%d heads=%d head_dim=%d; head_dim must be 32 or 64)", batch, L, heads,
               head_dim);
    if (batch == 0) return 1;
    ZS_REQUIRE(qkv && out, "zs_attention_split: null pointer");
    const float scale = 1.0f / sqrtf((float)head_dim);
    ZS_REQUIRE(L <= 32 * 65535, "zs_attention_split: L = %d too long", L);
    const dim3 grid(batch * heads, (L + 31) / 32);
    // few (sample, head, query tile) triples and several key tiles: the key tiles across the waves of a workgroup
    static const long long kw_below = getenv("ZS_ATT_KW_BELOW") ? atoll(getenv("ZS_ATT_KW_BELOW")) : 512;
    if ((long long)grid.x * grid.y < kw_below && L > 64) {
        if (head_dim == 64)
            hipLaunchKernelGGL(attention_split_kw_kernel<64>, grid, dim3(64 * ATT_KW), 0, S(stream), qkv, out, L, heads, scale);
        else
            hipLaunchKernelGGL(attention_split_kw_kernel<32>, grid, dim3(64 * ATT_KW), 0, S(stream), qkv, out, L, heads, scale);
        return zs::check_launch("zs_attention_split") ? 1 : 0;
    }
    // many (sample, head) pairs of several query tiles: K / V staged once per pair in LDS (attention_lds_kernel); the staged
    // image holds the keys padded to whole tiles: L <= 96 (window stage: 65) or <= 224 (ViT: 197)
    static const bool no_lds = getenv("ZS_ATT_NO_LDS") != nullptr;           // A/B switch
    static const long long lds_min_pairs = getenv("ZS_ATT_LDS_MIN_PAIRS") ? atoll(getenv("ZS_ATT_LDS_MIN_PAIRS")) : 128;
    if (!no_lds && (long long)batch * heads >= lds_min_pairs && L > 32 && L <= 224) {
        const dim3 g1(batch * heads);
#define ZS_ATT_LDS(D_, LP_) hipLaunchKernelGGL((attention_lds_kernel<D_, LP_>), g1, dim3(64 * (LP_ / 32 < 8 ? LP_ / 32 : 8)), 0, S(stream), qkv, out, L, heads, scale)
        if (head_dim == 64) { if (L <= 96) ZS_ATT_LDS(64, 96); else ZS_ATT_LDS(64, 224); }
        else { if (L <= 96) ZS_ATT_LDS(32, 96); else ZS_ATT_LDS(32, 224); }
#undef ZS_ATT_LDS
        return zs::check_launch("zs_attention_split") ? 1 : 0;
    }
    if (head_dim == 64)
        hipLaunchKernelGGL(attention_split_kernel<64>, grid, dim3(64), 0, S(stream), qkv, out, L, heads, scale);
    else
        hipLaunchKernelGGL(attention_split_kernel<32>, grid, dim3(64), 0, S(stream), qkv, out, L, heads, scale);
    return zs::check_launch("zs_attention_split") ? 1 : 0;
}

extern "C" int zs_attention(const float *qkv, float *out, int batch, int L, int heads, int head_dim, void *stream) {
    ZS_REQUIRE(batch >= 0 && L > 0 && heads > 0 && (head_dim == 32 || head_dim == 64),
               "zs_attention: bad size (B=%d L=%d heads=%d head_dim=%d; head_dim must be 32 or 64)", batch, L, heads,
               head_dim);
    if (batch == 0) return 1;
    ZS_REQUIRE(qkv && out, "zs_attention: null pointer");
    const float scale = 1.0f / sqrtf((float)head_dim);
    ZS_REQUIRE(L <= 32 * 65535, "zs_attention: L = %d too long", L);
    const dim3 grid(batch * heads, (L + 31) / 32);
    if (head_dim == 64)
        hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(64), 0, S(stream), qkv, out, L, heads, scale);
    else
        hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(64), 0, S(stream), qkv, out, L, heads, scale);
    return zs::check_launch("zs_attention") ? 1 : 0;
}

extern "C" int zs_max_pool_nhwc(const float *x, float *y, int batch, int Hin, int Win, int C, int Hout, int Wout,
                                int k, int stride, int pad_t, int pad_l, void *stream) {
    ZS_REQUIRE(batch >= 0 && Hin > 0 && Win > 0 && C > 0 && Hout > 0 && Wout > 0 && k > 0 && stride > 0,
               "zs_max_pool_nhwc: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(x && y, "zs_max_pool_nhwc: null pointer");
    const size_t quads = (size_t)batch * Hout * Wout * (C / 4);
    if ((C & 3) == 0 && quads < (1u << 31)) {
        hipLaunchKernelGGL(max_pool_quad_kernel, dim3(blocks_for(quads)), dim3(256), 0, S(stream), reinterpret_cast<const f32x4 *>(x),
                           reinterpret_cast<f32x4 *>(y), (unsigned)quads, Hin, Win, C / 4, Hout, Wout, k, stride, pad_t, pad_l);
        return zs::check_launch("zs_max_pool_nhwc") ? 1 : 0;
    }
    hipLaunchKernelGGL(max_pool_kernel, dim3(blocks_for((size_t)batch * Hout * Wout * C)), dim3(256), 0, S(stream), x,
                       y, batch, Hin, Win, C, Hout, Wout, k, stride, pad_t, pad_l);
    return zs::check_launch("zs_max_pool_nhwc") ? 1 : 0;
}

extern "C" int zs_global_mean_nhwc(const float *x, float *y, int batch, int HW, int C, void *stream) {
    ZS_REQUIRE(batch >= 0 && batch <= 65535 && HW > 0 && C > 0, "zs_global_mean_nhwc: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(x && y, "zs_global_mean_nhwc: null pointer");
    hipLaunchKernelGGL(global_mean_kernel, dim3((C + 63) / 64, batch), dim3(256), 0, S(stream), x, y, HW, C);
    return zs::check_launch("zs_global_mean_nhwc") ? 1 : 0;
}

extern "C" int zs_upsample2x_nhwc(const float *x, float *y, int batch, int Hin, int Win, int C, void *stream) {
    ZS_REQUIRE(batch >= 0 && Hin > 0 && Win > 0 && C > 0, "zs_upsample2x_nhwc: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(x && y, "zs_upsample2x_nhwc: null pointer");
    if ((C & 3) == 0 && 2 * Hin <= 65535 && batch <= 65535 && (long long)2 * Win * (C / 4) < (1LL << 30))
        hipLaunchKernelGGL(upsample2x_vec_kernel, dim3((unsigned)((2 * Win * (C / 4) + 255) / 256), 2 * Hin, batch), dim3(256),
                           0, S(stream), x, y, Hin, Win, C);
    else
        hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks_for((size_t)batch * 4 * Hin * Win * C)), dim3(256), 0, S(stream),
                           x, y, batch, Hin, Win, C);
    return zs::check_launch("zs_upsample2x_nhwc") ? 1 : 0;
}

extern "C" int zs_nchw_to_nhwc(const float *x, const float *mask, float *y, int batch, int C, int HW, int Cpad,
                               void *stream) {
    ZS_REQUIRE(batch >= 0 && C > 0 && HW > 0 && Cpad >= C, "zs_nchw_to_nhwc: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(x && y, "zs_nchw_to_nhwc: null pointer");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks_for((size_t)batch * HW * Cpad)), dim3(256), 0, S(stream), x,
                       mask, y, batch, C, HW, Cpad);
    return zs::check_launch("zs_nchw_to_nhwc") ? 1 : 0;
}

extern "C" int zs_nhwc_to_nchw(const float *x, float *y, int batch, int C, int HW, void *stream) {
    ZS_REQUIRE(batch >= 0 && C > 0 && HW > 0, "zs_nhwc_to_nchw: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(x && y, "zs_nhwc_to_nchw: null pointer");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks_for((size_t)batch * HW * C)), dim3(256), 0, S(stream), x, y,
                       batch, C, HW);
    return zs::check_launch("zs_nhwc_to_nchw") ? 1 : 0;
}

extern "C" int zs_assemble_tokens(const float *feat, const float *cls, const float *pos, float *tokens, int batch,
                                  int n, int C, void *stream) {
    ZS_REQUIRE(batch >= 0 && n > 0 && C > 0, "zs_assemble_tokens: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(feat && cls && pos && tokens, "zs_assemble_tokens: null pointer");
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(blocks_for((size_t)batch * (n + 1) * C)), dim3(256), 0, S(stream),
                       feat, cls, pos, tokens, batch, n, C);
    return zs::check_launch("zs_assemble_tokens") ? 1 : 0;
}

extern "C" int zs_readout_concat(const float *tokens, float *out, int batch, int n, int C, void *stream) {
    ZS_REQUIRE(batch >= 0 && n > 0 && C > 0, "zs_readout_concat: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(tokens && out, "zs_readout_concat: null pointer");
    hipLaunchKernelGGL(readout_concat_kernel, dim3(blocks_for((size_t)batch * n * 2 * C)), dim3(256), 0, S(stream),
                       tokens, out, batch, n, C);
    return zs::check_launch("zs_readout_concat") ? 1 : 0;
}

extern "C" int zs_window_tokens(const float *emb, const uint8_t *mask, const float *invalid_token, const float *cls,
                                const float *pos, float *out, int batch, int H, int W, int C, int win, void *stream) {
    ZS_REQUIRE(batch >= 0 && H > 0 && W > 0 && C > 0 && win > 0 && H % win == 0 && W % win == 0,
               "zs_window_tokens: bad size (B=%d H=%d W=%d C=%d win=%d)", batch, H, W, C, win);
    if (batch == 0) return 1;
    ZS_REQUIRE(emb && mask && invalid_token && cls && pos && out, "zs_window_tokens: null pointer");
    const size_t total = (size_t)batch * (H / win) * (W / win) * (win * win + 1) * C;
    const size_t tokens = total / C;
    if ((C & 3) == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 && tokens < (1u << 31)) {
        const unsigned per = 256 / (C / 4);
        hipLaunchKernelGGL(window_tokens_quad_kernel, dim3((unsigned)((tokens + per - 1) / per)), dim3(256), 0, S(stream),
                           reinterpret_cast<const f32x4 *>(emb), mask, reinterpret_cast<const f32x4 *>(invalid_token),
                           reinterpret_cast<const f32x4 *>(cls), reinterpret_cast<const f32x4 *>(pos),
                           reinterpret_cast<f32x4 *>(out), (unsigned)tokens, H, W, C / 4, win);
        return zs::check_launch("zs_window_tokens") ? 1 : 0;
    }
    hipLaunchKernelGGL(window_tokens_kernel, dim3(blocks_for(total)), dim3(256), 0, S(stream), emb, mask, invalid_token,
                       cls, pos, out, batch, H, W, C, win);
    return zs::check_launch("zs_window_tokens") ? 1 : 0;
}
