// Training-side non-GEMM kernels (fp32, channels-last): activation forward/backward, column sums
// (bias gradients), LayerNorm backward, per-sample row scaling (stochastic depth), and the
// backward of the token attention of csrc/nn_ops.hip.  All reductions run in a fixed order
// (two-stage partial sums, no atomics): bit-reproducible from run to run.
// Reference call sites: timm Block / Mlp / DropPath as used by model/shape/implicit.py:8,83-109
// and model/depth/vit.py; utils/loss.py.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
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

__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                      int act, float beta) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = act_fwd(x[i], act, beta);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ ref,
                                                      float *__restrict__ dx, size_t n, int act, float beta) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = act_bwd(dy[i], ref[i], act, beta);
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
// out[c] = scale * sum_k partial[k * stride + c]: 64 columns per workgroup, the chunks strided over 4
// row-lanes and combined in a fixed order
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                                              int chunks, int C, int stride, float scale) {
    __shared__ float lds[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6, c = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (c < C)
        for (int k = ry; k < chunks; k += 4) s += partial[(size_t)k * stride + c];
    lds[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) out[c] = ((lds[0][cx] + lds[1][cx]) + (lds[2][cx] + lds[3][cx])) * scale;
}

// ---- LayerNorm backward: wave per row for dx, per-workgroup partial dgamma / dbeta ----
constexpr int LN_ROWS = 16;      // rows per workgroup (4 per wave): many small workgroups, the rows are latency bound
constexpr int LN_MAXQ = 16;      // C <= 64 * LN_MAXQ
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                             const float *__restrict__ gamma, float *__restrict__ dx,
                                                             float *__restrict__ partial, int rows, int C, float eps) {
    extern __shared__ float lds[];                      // [4][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[LN_MAXQ], db[LN_MAXQ];
#pragma unroll
    for (int q = 0; q < LN_MAXQ; q++) dg[q] = db[q] = 0.f;
    const int r0 = blockIdx.x * LN_ROWS;
    for (int rr = wave; rr < LN_ROWS; rr += 4) {
        const int row = r0 + rr;
        if (row >= rows) break;
        const float *xr = x + (size_t)row * C, *gr = dy + (size_t)row * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        const float mean = wave_sum(s) / C;
        float q2 = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q2 += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q2) / C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int q = 0; q < LN_MAXQ; q++) {
            const int c = lane + 64 * q;
            if (c < C) {
                const float xh = (xr[c] - mean) * rstd, g = gr[c] * gamma[c];
                sg += g;
                sgx += g * xh;
                dg[q] += gr[c] * xh;
                db[q] += gr[c];
            }
        }
        const float mg = wave_sum(sg) / C, mgx = wave_sum(sgx) / C;
        for (int c = lane; c < C; c += 64) {
            const float xh = (xr[c] - mean) * rstd;
            dx[(size_t)row * C + c] = rstd * (gr[c] * gamma[c] - mg - xh * mgx);
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
// pass 1, wave per (b, h, query i): recompute the probability row, dP = dO V^T,
// dS = P * (dP - sum_j P dP); store P and dS rows; dQ_i = scale * sum_j dS_ij K_j
constexpr int ATT_MAXJ = 8;      // L <= 512
template <int D>
__global__ __launch_bounds__(256) void attention_bwd_rows_kernel(const float *__restrict__ qkv,
                                                                 const float *__restrict__ dout,
                                                                 float *__restrict__ dqkv, float *__restrict__ Pbuf,
                                                                 float *__restrict__ dSbuf, int L, int heads,
                                                                 float scale, int total_rows) {
    extern __shared__ float lds[];                      // per wave: ds[L]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gr_raw = blockIdx.x * 4 + wave;           // (b*heads + h) * L + i
    const bool active = gr_raw < total_rows;
    const int gr = active ? gr_raw : total_rows - 1;    // idle waves redo the last row, write nothing
    const int i = gr % L, bh = gr / L, h = bh % heads, b = bh / heads, C = heads * D;
    const float *base = qkv + (size_t)b * L * 3 * C + h * D;
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
            const float *kj = base + (size_t)j * 3 * C + C, *vj = kj + C;
            float a = 0.f, c2 = 0.f;
#pragma unroll
            for (int d = 0; d < D; d++) { a += q[d] * kj[d]; c2 += go[d] * vj[d]; }
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
    float *dsl = lds + wave * L;
    float *Prow = Pbuf + (size_t)gr * L, *dSrow = dSbuf + (size_t)gr * L;
#pragma unroll
    for (int t = 0; t < ATT_MAXJ; t++) {
        const int j = lane + 64 * t;
        if (j < L) {
            const float ds = s[t] * (dp[t] - delta);
            if (active) {
                Prow[j] = s[t];
                dSrow[j] = ds;
            }
            dsl[j] = ds;
        }
    }
    __syncthreads();
    // dQ: lanes own d; two halves of the key range for D = 32
    constexpr int PARTS = 64 / D;
    const int d = lane % D, part = lane / D;
    float acc = 0.f;
    for (int j = part; j < L; j += PARTS) acc += dsl[j] * base[(size_t)j * 3 * C + C + d];
    if (PARTS == 2) acc += __shfl_xor(acc, 32, 64);
    if (part == 0 && active) dqkv[((size_t)b * L + i) * 3 * C + h * D + d] = acc * scale;
}
// pass 2, wave per (b, h, key j): dV_j = sum_i P_ij dO_i, dK_j = scale * sum_i dS_ij Q_i
template <int D>
__global__ __launch_bounds__(256) void attention_bwd_cols_kernel(const float *__restrict__ qkv,
                                                                 const float *__restrict__ dout,
                                                                 float *__restrict__ dqkv,
                                                                 const float *__restrict__ Pbuf,
                                                                 const float *__restrict__ dSbuf, int L, int heads,
                                                                 float scale, int total_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gr = blockIdx.x * 4 + wave;
    if (gr >= total_rows) return;
    const int j = gr % L, bh = gr / L, h = bh % heads, b = bh / heads, C = heads * D;
    constexpr int PARTS = 64 / D;
    const int d = lane % D, part = lane / D;
    const float *qb = qkv + (size_t)b * L * 3 * C + h * D + d, *dob = dout + (size_t)b * L * C + h * D + d;
    const float *Pc = Pbuf + (size_t)bh * L * L + j, *dSc = dSbuf + (size_t)bh * L * L + j;
    float dv = 0.f, dk = 0.f;
    for (int i = part; i < L; i += PARTS) {
        dv += Pc[(size_t)i * L] * dob[(size_t)i * C];
        dk += dSc[(size_t)i * L] * qb[(size_t)i * 3 * C];
    }
    if (PARTS == 2) { dv += __shfl_xor(dv, 32, 64); dk += __shfl_xor(dk, 32, 64); }
    if (part == 0) {
        float *o = dqkv + ((size_t)b * L + j) * 3 * C + h * D + d;
        o[C] = dk * scale;
        o[2 * C] = dv;
    }
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
    hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, S(stream), x, y, n, act, beta);
    return zs::check_launch("zs_act_forward") ? 1 : 0;
}

extern "C" int zs_act_backward(const float *dy, const float *ref, float *dx, size_t n, int act, float beta,
                               void *stream) {
    ZS_REQUIRE(act >= 0 && act <= ZS_ACT_SOFTPLUS, "zs_act_backward: unknown activation %d", act);
    if (n == 0) return 1;
    ZS_REQUIRE(dy && ref && dx, "zs_act_backward: null pointer");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, S(stream), dy, ref, dx, n, act, beta);
    return zs::check_launch("zs_act_backward") ? 1 : 0;
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
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 63) / 64), dim3(256), 0, S(stream), partial, out, chunks, C,
                       C, scale);
    return zs::check_launch("zs_column_sum") ? 1 : 0;
}

extern "C" size_t zs_layer_norm_bwd_workspace_bytes(int rows, int C) {
    return (size_t)((rows + LN_ROWS - 1) / LN_ROWS) * 2 * C * sizeof(float);
}

extern "C" int zs_layer_norm_bwd(const float *dy, const float *x, const float *gamma, float *dx, float *dgamma,
                                 float *dbeta, int rows, int C, float eps, void *workspace, void *stream) {
    ZS_REQUIRE(rows > 0 && C > 0 && C <= 64 * LN_MAXQ, "zs_layer_norm_bwd: bad size (rows=%d C=%d, C <= %d)", rows, C,
               64 * LN_MAXQ);
    ZS_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && workspace, "zs_layer_norm_bwd: null pointer");
    const int wgs = (rows + LN_ROWS - 1) / LN_ROWS;
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3(wgs), dim3(256), 8 * C * sizeof(float), S(stream), dy, x, gamma, dx,
                       partial, rows, C, eps);
    // partial is [wg][2][C]: rows of stride 2C, dgamma in the first half, dbeta in the second
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 63) / 64), dim3(256), 0, S(stream), partial, dgamma, wgs, C,
                       2 * C, 1.0f);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 63) / 64), dim3(256), 0, S(stream), partial + C, dbeta, wgs,
                       C, 2 * C, 1.0f);
    return zs::check_launch("zs_layer_norm_bwd") ? 1 : 0;
}

extern "C" size_t zs_attention_bwd_workspace_bytes(int batch, int L, int heads) {
    return (size_t)2 * batch * heads * L * L * sizeof(float);
}

extern "C" int zs_attention_bwd(const float *qkv, const float *dout, float *dqkv, void *workspace, int batch, int L,
                                int heads, int head_dim, void *stream) {
    ZS_REQUIRE(batch >= 0 && L > 0 && L <= 64 * ATT_MAXJ && heads > 0 && (head_dim == 32 || head_dim == 64),
               "zs_attention_bwd: bad size (B=%d L=%d heads=%d head_dim=%d; L <= %d, head_dim 32 or 64)", batch, L,
               heads, head_dim, 64 * ATT_MAXJ);
    if (batch == 0) return 1;
    ZS_REQUIRE(qkv && dout && dqkv && workspace, "zs_attention_bwd: null pointer");
    const float scale = 1.0f / sqrtf((float)head_dim);
    const int rows = batch * heads * L;
    float *P = static_cast<float *>(workspace), *dS = P + (size_t)rows * L;
    const dim3 grid((rows + 3) / 4);
    const size_t lds = 4 * (size_t)L * sizeof(float);
    if (head_dim == 64) {
        hipLaunchKernelGGL(attention_bwd_rows_kernel<64>, grid, dim3(256), lds, S(stream), qkv, dout, dqkv, P, dS, L,
                           heads, scale, rows);
        hipLaunchKernelGGL(attention_bwd_cols_kernel<64>, grid, dim3(256), 0, S(stream), qkv, dout, dqkv, P, dS, L,
                           heads, scale, rows);
    } else {
        hipLaunchKernelGGL(attention_bwd_rows_kernel<32>, grid, dim3(256), lds, S(stream), qkv, dout, dqkv, P, dS, L,
                           heads, scale, rows);
        hipLaunchKernelGGL(attention_bwd_cols_kernel<32>, grid, dim3(256), 0, S(stream), qkv, dout, dqkv, P, dS, L,
                           heads, scale, rows);
    }
    return zs::check_launch("zs_attention_bwd") ? 1 : 0;
}
