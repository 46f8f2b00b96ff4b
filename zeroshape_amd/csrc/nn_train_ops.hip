// Training-side non-GEMM kernels (fp32, channels-last): activation forward/backward, column sums
// (bias gradients), LayerNorm backward, per-sample row scaling (stochastic depth), and the
// backward of the token attention of csrc/nn_ops.hip.  All reductions run in a fixed order
// (two-stage partial sums, no atomics): bit-reproducible from run to run.
// Reference call sites: timm Block / Mlp / DropPath as used by model/shape/implicit.py:8,83-109
// and model/depth/vit.py; utils/loss.py.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdlib.h>
#include <stdint.h>

namespace {

inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }
inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- activations ----
// torch GELU(approximate='none'), Softplus(beta, threshold=20), ReLU, ReLU then clamp(max=1)
__device__ __forceinline__ float act_fwd(float x, int act, float beta) {
    if (act == ZS_ACT_RELU) return fmaxf(x, 0.f);
    if (act == ZS_ACT_GELU) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    if (act == ZS_ACT_RELU_CLAMP1) return fminf(fmaxf(x, 0.f), 1.f);
    if (act == ZS_ACT_SOFTPLUS) return x * beta > 20.f ? x : log1pf(expf(x * beta)) / beta;
    return x;
}
// ref: the pre-activation input for GELU / softplus, the OUTPUT for ReLU / ReLU+clamp
__device__ __forceinline__ float act_bwd(float dy, float ref, int act, float beta) {
    if (act == ZS_ACT_RELU) return ref > 0.f ? dy : 0.f;
    if (act == ZS_ACT_RELU_CLAMP1) return (ref > 0.f && ref < 1.f) ? dy : 0.f;
    if (act == ZS_ACT_GELU) {
        const float cdf = 0.5f * (1.0f + erff(ref * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * ref * ref);
        return dy * (cdf + ref * pdf);
    }
    if (act == ZS_ACT_SOFTPLUS) {
        const float z = ref * beta;
        return z > 20.f ? dy : dy / (1.0f + expf(-z));
    }
    return dy;
}

// four elements per thread (one 16-byte access per tensor); the tail and unaligned tensors go element by element
typedef float act_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                      int act, float beta, int vec) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * (vec ? 4 : 1);
    if (vec && i + 3 < n) {
        const act_f32x4 v = *reinterpret_cast<const act_f32x4 *>(x + i);
        act_f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = act_fwd(v[e], act, beta);
        *reinterpret_cast<act_f32x4 *>(y + i) = o;
        return;
    }
    for (size_t j = i; j < n && j < i + (vec ? 4 : 1); j++) y[j] = act_fwd(x[j], act, beta);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ ref,
                                                      float *__restrict__ dx, size_t n, int act, float beta, int vec) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * (vec ? 4 : 1);
    if (vec && i + 3 < n) {
        const act_f32x4 g = *reinterpret_cast<const act_f32x4 *>(dy + i), r = *reinterpret_cast<const act_f32x4 *>(ref + i);
        act_f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = act_bwd(g[e], r[e], act, beta);
        *reinterpret_cast<act_f32x4 *>(dx + i) = o;
        return;
    }
    for (size_t j = i; j < n && j < i + (vec ? 4 : 1); j++) dx[j] = act_bwd(dy[j], ref[j], act, beta);
}

// ---- y = x + scale[b] * branch   /   y = scale[b] * x   (per-sample stochastic depth) ----
__global__ __launch_bounds__(256) void add_scaled_rows_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ branch,
                                                              const float *__restrict__ scale, float *__restrict__ y,
                                                              size_t per_sample, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float s = scale[i / per_sample];
    y[i] = x ? x[i] + s * branch[i] : s * branch[i];
}

// ---- column sums: partial[chunk][C] over row chunks, then a fixed-order sum of the chunks ----
__global__ __launch_bounds__(256) void column_partial_kernel(const float *__restrict__ x, float *__restrict__ partial,
                                                             int rows, int C, int rows_per_chunk) {
    __shared__ float lds[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6, c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s = 0.f;
    if (c < C)
        for (int r = r0 + ry; r < r1; r += 4) s += x[(size_t)r * C + c];
    lds[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = (lds[0][cx] + lds[1][cx]) + (lds[2][cx] + lds[3][cx]);
}
// out[c] = scale * sum_k partial[k * stride + c], c < C  (columns C .. C2-1 go to out2 - a second vector in the same
// partial rows, e.g. dbeta beside dgamma): 16 columns per workgroup, the chunks strided over 16 row-lanes and
// combined in a fixed order (a pass over a few thousand chunks must not run on four workgroups)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                                              float *__restrict__ out2, int chunks, int C, int C2,
                                                              int stride, float scale) {
    __shared__ float lds[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4, c = blockIdx.x * 16 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C2) {
        int k = ry;
        for (; k + 48 < chunks; k += 64) {
            s0 += partial[(size_t)k * stride + c];
            s1 += partial[(size_t)(k + 16) * stride + c];
            s2 += partial[(size_t)(k + 32) * stride + c];
            s3 += partial[(size_t)(k + 48) * stride + c];
        }
        for (; k < chunks; k += 16) s0 += partial[(size_t)k * stride + c];
    }
    lds[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < C2) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) t += lds[r][cx];
        if (c < C) out[c] = t * scale;
        else out2[c - C] = t * scale;
    }
}

// ---- LayerNorm backward: wave per row for dx, per-workgroup partial dgamma / dbeta ----
constexpr int LN_ROWS = 16;      // rows per workgroup (4 per wave): many small workgroups, the rows are latency bound
constexpr int LN_ROWS_FEW = 4;   // ... one per wave while that still leaves <= 1,024 workgroups (batch 4: 788 token rows on
                                 // 197 instead of 50 workgroups, 13.9 -> x us per launch)
static int ln_rows_per_wg(int rows) { return rows <= 1024 * LN_ROWS_FEW ? LN_ROWS_FEW : LN_ROWS; }
constexpr int LN_MAXQ = 16;      // C <= 64 * LN_MAXQ
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                             const float *__restrict__ gamma, float *__restrict__ dx,
                                                             float *__restrict__ partial, int rows, int C, float eps,
                                                             int per_wg, const float *__restrict__ add) {
    extern __shared__ float lds[];                      // [4][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[LN_MAXQ], db[LN_MAXQ];
#pragma unroll
    for (int q = 0; q < LN_MAXQ; q++) dg[q] = db[q] = 0.f;
    const int r0 = blockIdx.x * per_wg;
    float gam[LN_MAXQ];
#pragma unroll
    for (int q = 0; q < LN_MAXQ; q++) gam[q] = lane + 64 * q < C ? gamma[lane + 64 * q] : 0.f;
    for (int rr = wave; rr < per_wg; rr += 4) {
        const int row = r0 + rr;
        if (row >= rows) break;
        // the row lives in registers (C <= 64 * LN_MAXQ): x and dy are read once
        const float *xr = x + (size_t)row * C, *gr = dy + (size_t)row * C;
        float xv[LN_MAXQ], gv[LN_MAXQ];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < LN_MAXQ; q++) {
            const int c = lane + 64 * q;
            xv[q] = c < C ? xr[c] : 0.f;
            gv[q] = c < C ? gr[c] : 0.f;
            s += xv[q];
        }
        const float mean = wave_sum(s) / C;
        float q2 = 0.f;
#pragma unroll
        for (int q = 0; q < LN_MAXQ; q++) {
            const float d = lane + 64 * q < C ? xv[q] - mean : 0.f;
            q2 += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q2) / C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int q = 0; q < LN_MAXQ; q++) {
            if (lane + 64 * q < C) {
                xv[q] = (xv[q] - mean) * rstd;                 // xhat from here on
                const float g = gv[q] * gam[q];
                sg += g;
                sgx += g * xv[q];
                dg[q] += gv[q] * xv[q];
                db[q] += gv[q];
            }
        }
        const float mg = wave_sum(sg) / C, mgx = wave_sum(sgx) / C;
#pragma unroll
        for (int q = 0; q < LN_MAXQ; q++) {
            const int c = lane + 64 * q;
            if (c < C) {
                const float v = rstd * (gv[q] * gam[q] - mg - xv[q] * mgx);
                dx[(size_t)row * C + c] = add ? v + add[(size_t)row * C + c] : v;      // (+ the gradient of x's other consumer)
            }
        }
    }
#pragma unroll
    for (int q = 0; q < LN_MAXQ; q++) {
        const int c = lane + 64 * q;
        if (c < C) {
            lds[(wave * 2 + 0) * C + c] = dg[q];
            lds[(wave * 2 + 1) * C + c] = db[q];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        const int which = e / C, c = e % C;
        partial[((size_t)blockIdx.x * 2 + which) * C + c] =
            (lds[(0 * 2 + which) * C + c] + lds[(1 * 2 + which) * C + c]) +
            (lds[(2 * 2 + which) * C + c] + lds[(3 * 2 + which) * C + c]);
    }
}

// ---- attention backward (softmax(q k^T scale) v per head; qkv [B][L][3][H][D]) ----
// pass 1, workgroup = (b, h, tile of rows): K and V of the head staged in LDS (rows padded to D+1
// floats: lane j reading row j is conflict-free); a wave per query row recomputes the probability
// row, dP = dO V^T, dS = P * (dP - sum_j P dP), and stores the P and dS rows.
// pass 2, one wave per 32x32 output tile on the MFMA pipe:
//   dQ = scale * dS K,   dK = scale * dS^T Q,   dV = P^T dO      (contractions over L)
constexpr int ATT_MAXJ = 8;      // L <= 512
constexpr int ATT_ROWS = 32;     // query rows per workgroup in pass 1
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D>
__global__ __launch_bounds__(256) void attention_bwd_rows_kernel(const float *__restrict__ qkv,
                                                                 const float *__restrict__ dout,
                                                                 float *__restrict__ Pbuf, float *__restrict__ dSbuf,
                                                                 int L, int LS, int heads, float scale) {
    extern __shared__ float lds[];
    constexpr int KS = D + 1;
    float *Ks = lds, *Vs = lds + (size_t)L * KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x, h = bh % heads, b = bh / heads, C = heads * D;
    const float *base = qkv + (size_t)b * L * 3 * C + h * D;
    for (int e = threadIdx.x; e < L * D; e += 256) {
        const int j = e / D, d = e % D;
        Ks[j * KS + d] = base[(size_t)j * 3 * C + C + d];
        Vs[j * KS + d] = base[(size_t)j * 3 * C + 2 * C + d];
    }
    __syncthreads();
    const int i_end = min(L, (int)(blockIdx.y + 1) * ATT_ROWS);
    for (int i = blockIdx.y * ATT_ROWS + wave; i < i_end; i += 4) {
        const float *qi = base + (size_t)i * 3 * C, *doi = dout + ((size_t)b * L + i) * C + h * D;
        float q[D], go[D];
#pragma unroll
        for (int d = 0; d < D; d++) { q[d] = qi[d]; go[d] = doi[d]; }
        float s[ATT_MAXJ], dp[ATT_MAXJ];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < ATT_MAXJ; t++) {
            const int j = lane + 64 * t;
            s[t] = -INFINITY;
            dp[t] = 0.f;
            if (j < L) {
                float a = 0.f, c2 = 0.f;
#pragma unroll
                for (int d = 0; d < D; d++) { a += q[d] * Ks[j * KS + d]; c2 += go[d] * Vs[j * KS + d]; }
                s[t] = a * scale;
                dp[t] = c2;
                mx = fmaxf(mx, s[t]);
            }
        }
        mx = wave_max(mx);
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < ATT_MAXJ; t++) { s[t] = (lane + 64 * t < L) ? expf(s[t] - mx) : 0.f; den += s[t]; }
        den = wave_sum(den);
        float delta = 0.f;
#pragma unroll
        for (int t = 0; t < ATT_MAXJ; t++) { s[t] /= den; delta += s[t] * dp[t]; }
        delta = wave_sum(delta);
        float *Prow = Pbuf + ((size_t)bh * L + i) * LS, *dSrow = dSbuf + ((size_t)bh * L + i) * LS;
#pragma unroll
        for (int t = 0; t < ATT_MAXJ; t++) {
            const int j = lane + 64 * t;
            if (j < L) {
                Prow[j] = s[t];
                dSrow[j] = s[t] * (dp[t] - delta);
            }
        }
    }
}

// pass 1 on the MFMA pipe (L <= 256): one wave per (b, h, 32-query tile), transposed like the forward kernel
// (csrc/nn_ops.hip): S^T[key][query] = K Q^T and dP^T[key][query] = V dO^T for all key tiles stay in the accumulators
// (2 x 8 tiles x 16 registers), the softmax statistics and delta = sum_j P dP are sums over a lane's registers + its
// partner half; P and dS = P (dP - delta) leave as float4 row segments (rows of LS = L rounded up to 4 floats, so that
// the segments are aligned; the padding receives zeros).  Replaces attention_bwd_rows_kernel, whose dot
// products run on the vector ALU with one LDS read per multiply-add (103 us per ViT block at batch 4).
constexpr int ATT_MFMA_TILES = 8;
template <int D>
__global__ __launch_bounds__(64) void attention_bwd_probs_kernel(const float *__restrict__ qkv,
                                                                 const float *__restrict__ dout,
                                                                 float *__restrict__ Pbuf, float *__restrict__ dSbuf,
                                                                 int L, int LS, int heads, float scale) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int DQ = D / 8;
    const int lane = threadIdx.x, l32 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.x, h = bh % heads, b = bh / heads, C = heads * D, q0 = blockIdx.y * 32;
    const int LT = (L + 31) / 32;
    const float *base = qkv + (size_t)b * L * 3 * C + h * D;
    const int qrow = min(q0 + l32, L - 1);
    f32x4 qf[DQ], gf[DQ];
#pragma unroll
    for (int t = 0; t < DQ; t++) {
        qf[t] = *reinterpret_cast<const f32x4 *>(base + (size_t)qrow * 3 * C + 4 * (2 * t + half)) * scale;
        gf[t] = *reinterpret_cast<const f32x4 *>(dout + ((size_t)b * L + qrow) * C + h * D + 4 * (2 * t + half));
    }
    f32x16 sT[ATT_MFMA_TILES], dT[ATT_MFMA_TILES];
    float mx = -INFINITY;
    // the K / V rows of key tile kt + 1 are requested in front of the 64 MFMAs of tile kt (a lone wave per workgroup)
    f32x4 kf[DQ], vf[DQ], kn[DQ] = {}, vn[DQ] = {};
    auto fetch = [&](int kt, f32x4 (&k_)[DQ], f32x4 (&v_)[DQ]) {
        const float *row = base + (size_t)min(kt * 32 + l32, L - 1) * 3 * C + 4 * half;
#pragma unroll
        for (int t = 0; t < DQ; t++) {
            k_[t] = *reinterpret_cast<const f32x4 *>(row + C + 8 * t);
            v_[t] = *reinterpret_cast<const f32x4 *>(row + 2 * C + 8 * t);
        }
    };
    fetch(0, kf, vf);
#pragma unroll
    for (int kt = 0; kt < ATT_MFMA_TILES; kt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) { sT[kt][r] = -INFINITY; dT[kt][r] = 0.f; }
        if (kt < LT) {
            if (kt + 1 < LT) fetch(kt + 1, kn, vn);
            f32x16 a, c;
#pragma unroll
            for (int r = 0; r < 16; r++) { a[r] = 0.f; c[r] = 0.f; }
#pragma unroll
            for (int t = 0; t < DQ; t++) {
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t][s4], qf[t][s4], a, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[t][s4], gf[t][s4], c, 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < DQ; t++) { kf[t] = kn[t]; vf[t] = vn[t]; }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = kt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                sT[kt][r] = key < L ? a[r] : -INFINITY;
                dT[kt][r] = c[r];
                mx = fmaxf(mx, sT[kt][r]);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float den = 0.f;
#pragma unroll
    for (int kt = 0; kt < ATT_MFMA_TILES; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sT[kt][r] = expf(sT[kt][r] - mx);          // exp(-inf) = 0 for padded keys / unused tiles
            den += sT[kt][r];
        }
    den += __shfl_xor(den, 32, 64);
    const float inv = 1.0f / den;
    float delta = 0.f;
#pragma unroll
    for (int kt = 0; kt < ATT_MFMA_TILES; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sT[kt][r] *= inv;
            delta += sT[kt][r] * dT[kt][r];
        }
    delta += __shfl_xor(delta, 32, 64);
    if (q0 + l32 >= L) return;
    float *Prow = Pbuf + ((size_t)bh * L + q0 + l32) * LS, *dSrow = dSbuf + ((size_t)bh * L + q0 + l32) * LS;
#pragma unroll
    for (int kt = 0; kt < ATT_MFMA_TILES; kt++)
        if (kt < LT)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int key0 = kt * 32 + 8 * g + 4 * half;
                if (key0 < LS) {          // keys L .. LS - 1: P = exp(-inf) = 0 and dS = 0 * (finite) = 0
                    const f32x4 pv = {sT[kt][4 * g], sT[kt][4 * g + 1], sT[kt][4 * g + 2], sT[kt][4 * g + 3]};
                    const f32x4 dv = {dT[kt][4 * g] - delta, dT[kt][4 * g + 1] - delta, dT[kt][4 * g + 2] - delta,
                                      dT[kt][4 * g + 3] - delta};
                    *reinterpret_cast<f32x4 *>(Prow + key0) = pv;
                    *reinterpret_cast<f32x4 *>(dSrow + key0) = pv * dv;
                }
            }
}

template <int D>
__global__ __launch_bounds__(64) void attention_bwd_mfma_kernel(const float *__restrict__ qkv,
                                                                const float *__restrict__ dout, float *__restrict__ dqkv,
                                                                const float *__restrict__ Pbuf,
                                                                const float *__restrict__ dSbuf, int L, int LS, int heads,
                                                                float scale) {
    constexpr int DT = D / 32;
    const int lane = threadIdx.x, l32 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.x, h = bh % heads, b = bh / heads, C = heads * D;
    const int LT = (L + 31) / 32;
    const int t = blockIdx.y, which = t / (LT * DT), r0 = ((t / DT) % LT) * 32, d0 = (t % DT) * 32;
    const float *base = qkv + (size_t)b * L * 3 * C + h * D;
    const float *M = (which == 2 ? Pbuf : dSbuf) + (size_t)bh * L * LS;
    // A[m][k]: which 0 (dQ): dS[r0+m][k];  which 1, 2 (dK, dV): dS / P [k][r0+m]   (k = contraction index)
    // B[k][n]: which 0: K[k][d0+n];  which 1: Q[k][d0+n];  which 2: dO[k][d0+n]
    const int m = r0 + l32;
    const bool m_ok = m < L;
    const size_t a_row = which == 0 ? (size_t)min(m, L - 1) * LS : (size_t)min(m, L - 1), a_step = which == 0 ? 1 : LS;
    const float *Bp = which == 0 ? base + C + d0 + l32 : (which == 1 ? base + d0 + l32 : dout + (size_t)b * L * C + h * D + d0 + l32);
    const size_t b_step = which == 2 ? C : 3 * C;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    // the 16 operand values of the NEXT eight K = 2 steps are requested in front of the MFMAs of the current ones (a lone
    // wave per workgroup: nothing else hides the loads; 34 -> 2x us per ViT block at batch 4)
    constexpr int U = 8;
    float av[U], bv[U], an[U] = {}, bn[U] = {};
    auto fetch = [&](int k0, float (&a_)[U], float (&b_)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kc = min(k0 + 2 * u + half, L - 1);
            a_[u] = M[a_row + (size_t)kc * a_step];
            b_[u] = Bp[(size_t)kc * b_step];
        }
    };
    fetch(0, av, bv);
    for (int k0 = 0; k0 < L; k0 += 2 * U) {
        if (k0 + 2 * U < L) fetch(k0 + 2 * U, an, bn);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool ok = k0 + 2 * u + half < L;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((ok && m_ok) ? av[u] : 0.f, ok ? bv[u] : 0.f, acc, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) { av[u] = an[u]; bv[u] = bn[u]; }
    }
    const float mul = which == 2 ? 1.0f : scale;
    const int col = which == 0 ? 0 : (which == 1 ? C : 2 * C);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = r0 + 8 * (r >> 2) + 4 * half + (r & 3);
        if (row < L) dqkv[((size_t)b * L + row) * 3 * C + col + h * D + d0 + l32] = acc[r] * mul;
    }
}


// ---- CoordEmb token preparation, backward (adjoint of window_tokens_kernel, csrc/nn_ops.hip) ----
// d_out [B*nwy*nwx][win*win+1][C] -> per pixel (each pixel belongs to exactly one window token: a gather,
// no atomics): d_emb [B][H][W][C] = the token's gradient where the pixel is valid, else 0; d_inv_rows
// [B][H][W][C] = the complement (rows that fed the learned invalid-coordinate token); d_cls_rows
// [B*nwy*nwx][C] = the gradient of every window's class token.  Their column sums (zs_column_sum) are the
// gradients of invalid_coord_token and cls_token (seen_coord_enc.py:52-66).
__global__ __launch_bounds__(256) void window_tokens_bwd_kernel(const float *__restrict__ d_out,
                                                                const uint8_t *__restrict__ mask, float *__restrict__ d_emb,
                                                                float *__restrict__ d_inv_rows,
                                                                float *__restrict__ d_cls_rows, int B, int H, int W, int C,
                                                                int win) {
    const int T = win * win + 1, nwx = W / win, nwy = H / win;
    const size_t n_pix = (size_t)B * H * W * C, n_cls = (size_t)B * nwy * nwx * C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_pix) {
        const int c = i % C;
        const size_t pix = i / C;
        const int px = pix % W, py = (pix / W) % H, b = pix / W / H;
        const size_t wdx = ((size_t)b * nwy + py / win) * nwx + px / win;
        const int t = 1 + (py % win) * win + (px % win);
        const float g = d_out[(wdx * T + t) * C + c];
        const bool ok = mask[pix] != 0;
        d_emb[i] = ok ? g : 0.f;
        d_inv_rows[i] = ok ? 0.f : g;
    } else if (i < n_pix + n_cls) {
        const size_t j = i - n_pix;
        d_cls_rows[j] = d_out[(j / C) * T * C + (j % C)];
    }
}

// Adjoint of interpolate_coordmap at half size (utils/util.py:336-345; bilinear, align_corners=False, exact
// factor 2 = the 2x2 mean): coord_dsp[o] = keep[o] * mean4(seen * m) / (mean4(m) + 1e-6), keep = mean4(m) > 0.5.
// Expressed as the SAME-SIZE coordinate-map gradient zs_seen_surface_bwd expects at valid pixels (it multiplies by
// 1 / (1 + 1e-6) itself): d_full[y][x] = 0.25 d_dsp[o] keep[o] (1 + 1e-6) / (mean4(m) + 1e-6); 0 at invalid pixels.
__global__ __launch_bounds__(256) void coord_dsp2_bwd_kernel(const float *__restrict__ d_dsp, const float *__restrict__ mask,
                                                             const float *__restrict__ mask_dsp, float *__restrict__ d_full,
                                                             int B, int H, int W) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * 3 * H * W;
    if (i >= total) return;
    const int x = i % W, y = (i / W) % H, c = (i / W / H) % 3, b = i / W / H / 3;
    const int Ho = H / 2, Wo = W / 2, yo = y / 2, xo = x / 2;
    const float *M = mask + (size_t)b * H * W;
    if (yo >= Ho || xo >= Wo || !(M[(size_t)y * W + x] > 0.5f)) {
        d_full[i] = 0.f;
        return;
    }
    const float *m0 = M + (size_t)(2 * yo) * W + 2 * xo;
    const float den = 0.25f * ((m0[0] > 0.5f ? 1.f : 0.f) + (m0[1] > 0.5f ? 1.f : 0.f) + (m0[W] > 0.5f ? 1.f : 0.f) +
                               (m0[W + 1] > 0.5f ? 1.f : 0.f));
    const size_t o = ((size_t)b * Ho + yo) * Wo + xo;
    const float keep = mask_dsp[o] > 0.5f ? 1.f : 0.f;
    d_full[i] = 0.25f * d_dsp[((size_t)b * 3 + c) * Ho * Wo + (size_t)yo * Wo + xo] * keep * (1.0f + 1.e-6f) / (den + 1.e-6f);
}

}  // namespace

#define ZS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            zs::set_err(__VA_ARGS__);    \
            return 0;                    \
        }                                \
    } while (0)

extern "C" int zs_act_forward(const float *x, float *y, size_t n, int act, float beta, void *stream) {
    ZS_REQUIRE(act >= 0 && act <= ZS_ACT_SOFTPLUS, "zs_act_forward: unknown activation %d", act);
    if (n == 0) return 1;
    ZS_REQUIRE(x && y, "zs_act_forward: null pointer");
    const int vec = ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) == 0 ? 1 : 0;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(vec ? (n + 3) / 4 : n)), dim3(256), 0, S(stream), x, y, n, act, beta,
                       vec);
    return zs::check_launch("zs_act_forward") ? 1 : 0;
}

extern "C" int zs_act_backward(const float *dy, const float *ref, float *dx, size_t n, int act, float beta,
                               void *stream) {
    ZS_REQUIRE(act >= 0 && act <= ZS_ACT_SOFTPLUS, "zs_act_backward: unknown activation %d", act);
    if (n == 0) return 1;
    ZS_REQUIRE(dy && ref && dx, "zs_act_backward: null pointer");
    const int vec = ((reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(ref) | reinterpret_cast<size_t>(dx)) & 15) == 0;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(vec ? (n + 3) / 4 : n)), dim3(256), 0, S(stream), dy, ref, dx, n, act,
                       beta, vec);
    return zs::check_launch("zs_act_backward") ? 1 : 0;
}

// NeRF positional encoding of 3D points (utils/layers.py:8-53 of the reference, get_embedder(L, 3): include_input,
// log-sampled frequencies 2^0 .. 2^(L-1), [sin, cos] per frequency): out[i] = [x | sin(x 1) | cos(x 1) | sin(x 2) | cos(x 2)
// | ...] (3 + 6 L values, x*freq rounded to fp32 first like torch), zero padded to `stride` floats.
__global__ __launch_bounds__(256) void posenc3d_kernel(const float *__restrict__ pts, size_t n, int L, float *__restrict__ out,
                                                       int stride) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    float *o = out + i * stride;
    for (int d = 0; d < 3; d++) o[d] = x[d];
    float freq = 1.0f;
    for (int k = 0; k < L; k++, freq *= 2.0f)
        for (int d = 0; d < 3; d++) {
            const float a = x[d] * freq;
            o[3 + 6 * k + d] = sinf(a);
            o[3 + 6 * k + 3 + d] = cosf(a);
        }
    for (int c = 3 + 6 * L; c < stride; c++) o[c] = 0.f;
}

extern "C" int zs_posenc3d(const float *points, size_t n, int L, float *out, int stride, void *stream) {
    ZS_REQUIRE(L >= 0 && L <= 16 && stride >= 3 + 6 * L, "zs_posenc3d: bad sizes (L=%d stride=%d)", L, stride);
    if (n == 0) return 1;
    ZS_REQUIRE(points && out, "zs_posenc3d: null pointer");
    hipLaunchKernelGGL(posenc3d_kernel, dim3(blocks_for(n)), dim3(256), 0, S(stream), points, n, L, out, stride);
    return zs::check_launch("zs_posenc3d") ? 1 : 0;
}

extern "C" int zs_add_scaled_rows(const float *x, const float *branch, const float *scale, float *y, int batch,
                                  size_t per_sample, void *stream) {
    ZS_REQUIRE(batch >= 0 && per_sample > 0, "zs_add_scaled_rows: bad size");
    if (batch == 0) return 1;
    ZS_REQUIRE(branch && scale && y, "zs_add_scaled_rows: null pointer");
    const size_t total = (size_t)batch * per_sample;
    hipLaunchKernelGGL(add_scaled_rows_kernel, dim3(blocks_for(total)), dim3(256), 0, S(stream), x, branch, scale, y,
                       per_sample, total);
    return zs::check_launch("zs_add_scaled_rows") ? 1 : 0;
}

static int column_chunks(int rows) {
    int chunks = (rows + 255) / 256;
    if (chunks > 512) chunks = 512;
    return chunks < 1 ? 1 : chunks;
}

extern "C" size_t zs_column_sum_workspace_bytes(int rows, int C) {
    return (size_t)column_chunks(rows) * C * sizeof(float);
}

extern "C" int zs_column_sum(const float *x, float *out, int rows, int C, float scale, void *workspace,
                             void *stream) {
    ZS_REQUIRE(rows > 0 && C > 0, "zs_column_sum: bad size (rows=%d C=%d)", rows, C);
    ZS_REQUIRE(x && out && workspace, "zs_column_sum: null pointer");
    const int chunks = column_chunks(rows), per = (rows + chunks - 1) / chunks;
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(column_partial_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, S(stream), x, partial, rows, C,
                       per);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 15) / 16), dim3(256), 0, S(stream), partial, out,
                       static_cast<float *>(nullptr), chunks, C, C, C, scale);
    return zs::check_launch("zs_column_sum") ? 1 : 0;
}

extern "C" size_t zs_layer_norm_bwd_workspace_bytes(int rows, int C) {
    const int per = ln_rows_per_wg(rows);
    return (size_t)((rows + per - 1) / per) * 2 * C * sizeof(float);
}

extern "C" int zs_layer_norm_bwd(const float *dy, const float *x, const float *gamma, float *dx, float *dgamma,
                                 float *dbeta, int rows, int C, float eps, void *workspace, void *stream) {
    return zs_layer_norm_bwd_add(dy, x, gamma, nullptr, dx, dgamma, dbeta, rows, C, eps, workspace, stream);
}

extern "C" int zs_layer_norm_bwd_add(const float *dy, const float *x, const float *gamma, const float *add, float *dx,
                                     float *dgamma, float *dbeta, int rows, int C, float eps, void *workspace, void *stream) {
    ZS_REQUIRE(rows > 0 && C > 0 && C <= 64 * LN_MAXQ, "zs_layer_norm_bwd: bad size (rows=%d C=%d, C <= %d)", rows, C,
               64 * LN_MAXQ);
    ZS_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && workspace, "zs_layer_norm_bwd: null pointer");
    const int per = ln_rows_per_wg(rows), wgs = (rows + per - 1) / per;
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3(wgs), dim3(256), 8 * C * sizeof(float), S(stream), dy, x, gamma, dx,
                       partial, rows, C, eps, per, add);
    // partial is [wg][2][C]: rows of stride 2C, dgamma in the first half, dbeta in the second
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C + 15) / 16), dim3(256), 0, S(stream), partial, dgamma, dbeta,
                       wgs, C, 2 * C, 2 * C, 1.0f);
    return zs::check_launch("zs_layer_norm_bwd") ? 1 : 0;
}

extern "C" size_t zs_attention_bwd_workspace_bytes(int batch, int L, int heads) {
    return (size_t)2 * batch * heads * L * ((L + 3) & ~3) * sizeof(float);       // P and dS, rows padded to 16 bytes
}

extern "C" int zs_attention_bwd(const float *qkv, const float *dout, float *dqkv, void *workspace, int batch, int L,
                                int heads, int head_dim, void *stream) {
    ZS_REQUIRE(batch >= 0 && L > 0 && L <= 64 * ATT_MAXJ && heads > 0 && (head_dim == 32 || head_dim == 64),
               "zs_attention_bwd: bad size (B=%d L=%d heads=%d head_dim=%d; L <= %d, head_dim 32 or 64)", batch, L,
               heads, head_dim, 64 * ATT_MAXJ);
    if (batch == 0) return 1;
    ZS_REQUIRE(qkv && dout && dqkv && workspace, "zs_attention_bwd: null pointer");
    const float scale = 1.0f / sqrtf((float)head_dim);
    const int BH = batch * heads;
    const int LS = (L + 3) & ~3;
    ZS_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "zs_attention_bwd: workspace must be 16-byte aligned");
    float *P = static_cast<float *>(workspace), *dS = P + (size_t)BH * L * LS;
    const dim3 grid1(BH, (L + ATT_ROWS - 1) / ATT_ROWS), grid2(BH, 3 * ((L + 31) / 32) * (head_dim / 32));
    const size_t lds = (size_t)2 * L * (head_dim + 1) * sizeof(float);
    ZS_REQUIRE(lds <= 160 * 1024, "zs_attention_bwd: L = %d needs %zu bytes of LDS", L, lds);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(attention_bwd_rows_kernel<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(attention_bwd_rows_kernel<32>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    static const bool rows_valu = getenv("ZS_ATTN_BWD_VALU") != nullptr;     // A/B switch: the vector-ALU pass 1
    if (L <= 32 * ATT_MFMA_TILES && !rows_valu) {
        const dim3 gridp(BH, (L + 31) / 32);
        if (head_dim == 64)
            hipLaunchKernelGGL(attention_bwd_probs_kernel<64>, gridp, dim3(64), 0, S(stream), qkv, dout, P, dS, L, LS, heads, scale);
        else
            hipLaunchKernelGGL(attention_bwd_probs_kernel<32>, gridp, dim3(64), 0, S(stream), qkv, dout, P, dS, L, LS, heads, scale);
        if (head_dim == 64)
            hipLaunchKernelGGL(attention_bwd_mfma_kernel<64>, grid2, dim3(64), 0, S(stream), qkv, dout, dqkv, P, dS, L, LS, heads,
                               scale);
        else
            hipLaunchKernelGGL(attention_bwd_mfma_kernel<32>, grid2, dim3(64), 0, S(stream), qkv, dout, dqkv, P, dS, L, LS, heads,
                               scale);
        return zs::check_launch("zs_attention_bwd") ? 1 : 0;
    }
    if (head_dim == 64) {
        hipLaunchKernelGGL(attention_bwd_rows_kernel<64>, grid1, dim3(256), lds, S(stream), qkv, dout, P, dS, L, LS, heads, scale);
        hipLaunchKernelGGL(attention_bwd_mfma_kernel<64>, grid2, dim3(64), 0, S(stream), qkv, dout, dqkv, P, dS, L, LS, heads,
                           scale);
    } else {
        hipLaunchKernelGGL(attention_bwd_rows_kernel<32>, grid1, dim3(256), lds, S(stream), qkv, dout, P, dS, L, LS, heads, scale);
        hipLaunchKernelGGL(attention_bwd_mfma_kernel<32>, grid2, dim3(64), 0, S(stream), qkv, dout, dqkv, P, dS, L, LS, heads,
                           scale);
    }
    return zs::check_launch("zs_attention_bwd") ? 1 : 0;
}

extern "C" int zs_window_tokens_bwd(const float *d_out, const uint8_t *mask, float *d_emb, float *d_inv_rows,
                                    float *d_cls_rows, int batch, int H, int W, int C, int win, void *stream) {
    if (batch < 0 || H <= 0 || W <= 0 || C <= 0 || win <= 0 || H % win || W % win) {
        zs::set_err("zs_window_tokens_bwd: bad geometry (batch=%d %dx%dx%d win %d)", batch, H, W, C, win);
        return 0;
    }
    if (batch == 0) return 1;
    if (!d_out || !mask || !d_emb || !d_inv_rows || !d_cls_rows) {
        zs::set_err("zs_window_tokens_bwd: null pointer");
        return 0;
    }
    const size_t total = (size_t)batch * H * W * C + (size_t)batch * (H / win) * (W / win) * C;
    hipLaunchKernelGGL(window_tokens_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, S(stream), d_out, mask, d_emb,
                       d_inv_rows, d_cls_rows, batch, H, W, C, win);
    return zs::check_launch("zs_window_tokens_bwd") ? 1 : 0;
}

extern "C" int zs_coord_dsp2_bwd(const float *d_dsp, const float *mask, const float *mask_dsp, float *d_full, int batch,
                                 int H, int W, void *stream) {
    if (batch < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) {
        zs::set_err("zs_coord_dsp2_bwd: bad size (batch=%d H=%d W=%d; even sizes)", batch, H, W);
        return 0;
    }
    if (batch == 0) return 1;
    if (!d_dsp || !mask || !mask_dsp || !d_full) {
        zs::set_err("zs_coord_dsp2_bwd: null pointer");
        return 0;
    }
    hipLaunchKernelGGL(coord_dsp2_bwd_kernel, dim3(blocks_for((size_t)batch * 3 * H * W)), dim3(256), 0, S(stream), d_dsp,
                       mask, mask_dsp, d_full, batch, H, W);
    return zs::check_launch("zs_coord_dsp2_bwd") ? 1 : 0;
}
