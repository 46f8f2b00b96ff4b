// Training path of the implicit decoder and the optimiser (fp32):
//   zs_point_attention / _bwd   ImplFuncAttention (model/shape/implicit.py:25-79): every query
//                               point attends to the latent tokens and to itself (softmax over
//                               Ll + 1), forward and backward
//   zs_bce_logits / _bwd        Loss.shape_loss (utils/loss.py:18-28)
//   zs_adamw_multi              torch.optim.AdamW step over a table of tensors (one launch)
//   zs_gather_flat / zs_sumsq_multi   gradient bucketing for the RCCL all-reduce, gradient norm
// The inference path of the same decoder is the fused kernel of csrc/sdf_decoder.hip; training
// runs layer by layer (GEMMs in csrc/nn_conv.hip / nn_train_gemm.hip) because the backward
// pass needs the intermediate activations.  Reductions run in a fixed order (no atomics).
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

constexpr int D = 32;            // head dimension of the decoder (options/shape.yaml:21-22: 256 / 8)
constexpr int KS = D + 1;        // padded LDS row: lane j reading row j is conflict-free
constexpr int PA_MAXJ = 4;       // Ll <= 256 latent tokens
constexpr int PT = 256;          // points per workgroup

// qkv_p [B][M][3][H][D], qkv_l [B][Ll][3][H][D]; workgroup = (b, h, tile of PT points), 4 waves,
// a wave handles one point at a time: lanes own the latent keys for the logits, then the head
// dimensions for the value sum.
__device__ __forceinline__ void stage_kv(const float *qkv_l, float *Ks, float *Vs, int b, int h, int Ll, int C) {
    for (int e = threadIdx.x; e < Ll * D; e += 256) {
        const int j = e / D, d = e % D;
        const float *r = qkv_l + ((size_t)b * Ll + j) * 3 * C + h * D + d;
        Ks[j * KS + d] = r[C];
        Vs[j * KS + d] = r[2 * C];
    }
}

struct Row {                     // softmax row of one point: probabilities of this lane's keys + self
    float p[PA_MAXJ], p_self;
};
__device__ __forceinline__ Row softmax_row(const float *q, const float *kself, const float *Ks, int Ll, int lane,
                                           float scale) {
    Row r;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < PA_MAXJ; t++) {
        const int j = lane + 64 * t;
        float a = -INFINITY;
        if (j < Ll) {
            a = 0.f;
#pragma unroll
            for (int d = 0; d < D; d++) a += q[d] * Ks[j * KS + d];
            a *= scale;
        }
        r.p[t] = a;
        mx = fmaxf(mx, a);
    }
    float ss = 0.f;
#pragma unroll
    for (int d = 0; d < D; d++) ss += q[d] * kself[d];
    ss *= scale;
    mx = fmaxf(wave_max(mx), ss);
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < PA_MAXJ; t++) {
        r.p[t] = (lane + 64 * t < Ll) ? expf(r.p[t] - mx) : 0.f;
        den += r.p[t];
    }
    r.p_self = expf(ss - mx);
    den = wave_sum(den) + r.p_self;
    const float inv = 1.0f / den;
#pragma unroll
    for (int t = 0; t < PA_MAXJ; t++) r.p[t] *= inv;
    r.p_self *= inv;
    return r;
}

__global__ __launch_bounds__(256) void point_attention_kernel(const float *__restrict__ qkv_p,
                                                              const float *__restrict__ qkv_l, float *__restrict__ out,
                                                              int M, int Ll, int heads, float scale) {
    extern __shared__ float lds[];
    float *Ks = lds, *Vs = Ks + Ll * KS, *prow = Vs + Ll * KS;          // prow: [4][Ll]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, C = heads * D;
    stage_kv(qkv_l, Ks, Vs, b, h, Ll, C);
    __syncthreads();
    const int i_end = min(M, (int)(blockIdx.y + 1) * PT);
    float *pw = prow + wave * Ll;
    for (int i = blockIdx.y * PT + wave; i < i_end; i += 4) {
        const float *src = qkv_p + ((size_t)b * M + i) * 3 * C + h * D;
        float q[D], ks[D];
#pragma unroll
        for (int d = 0; d < D; d++) { q[d] = src[d]; ks[d] = src[C + d]; }
        const Row r = softmax_row(q, ks, Ks, Ll, lane, scale);
#pragma unroll
        for (int t = 0; t < PA_MAXJ; t++)
            if (lane + 64 * t < Ll) pw[lane + 64 * t] = r.p[t];
        const int d = lane & 31, part = lane >> 5;
        float acc = 0.f;
        for (int j = part; j < Ll; j += 2) acc += pw[j] * Vs[j * KS + d];
        acc += __shfl_xor(acc, 32, 64);
        if (part == 0) out[((size_t)b * M + i) * C + h * D + d] = acc + r.p_self * src[2 * C + d];
    }
}

// The attention MAP of the decoder call (implicit.py:60-66,277: per block the softmax probabilities of a point over the latent
// tokens - the point's own column dropped after the softmax - averaged over the heads; the blocks' maps averaged by the
// caller through `weight` / `accumulate`).  For decoder geometries the fused kernels (csrc/sdf_decoder.hip) are not specialised
// for; a visualisation path: one wave per point, the heads in a loop, K_h restaged per head.
constexpr int PP = 32;           // points per workgroup: 8 per wave, their probability sums stay in registers over the heads
__global__ __launch_bounds__(256) void point_attention_probs_kernel(const float *__restrict__ qkv_p,
                                                                    const float *__restrict__ qkv_l, float *__restrict__ attn,
                                                                    int M, int Ll, int heads, float scale, float weight,
                                                                    int accumulate) {
    extern __shared__ float lds[];
    float *Ks = lds, *Vs = Ks + Ll * KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, C = heads * D;
    float acc[PP / 4][PA_MAXJ];
#pragma unroll
    for (int s = 0; s < PP / 4; s++)
#pragma unroll
        for (int t = 0; t < PA_MAXJ; t++) acc[s][t] = 0.f;
    for (int h = 0; h < heads; h++) {
        __syncthreads();
        stage_kv(qkv_l, Ks, Vs, b, h, Ll, C);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < PP / 4; s++) {
            const int i = blockIdx.y * PP + wave + 4 * s;
            if (i >= M) continue;                      // (wave-uniform)
            const float *src = qkv_p + ((size_t)b * M + i) * 3 * C + h * D;
            float q[D], ks[D];
#pragma unroll
            for (int d = 0; d < D; d++) { q[d] = src[d]; ks[d] = src[C + d]; }
            const Row r = softmax_row(q, ks, Ks, Ll, lane, scale);
#pragma unroll
            for (int t = 0; t < PA_MAXJ; t++) acc[s][t] += r.p[t];
        }
    }
    const float w = weight / (float)heads;
#pragma unroll
    for (int s = 0; s < PP / 4; s++) {
        const int i = blockIdx.y * PP + wave + 4 * s;
        if (i >= M) continue;
#pragma unroll
        for (int t = 0; t < PA_MAXJ; t++) {
            const int j = lane + 64 * t;
            if (j < Ll) {
                float *o = attn + ((size_t)b * M + i) * Ll + j;
                *o = (accumulate ? *o : 0.f) + w * acc[s][t];
            }
        }
    }
}

// The same on the MFMA pipe: one wave per (b, h, 32 points), transposed like the encoder's attention kernel
// (csrc/nn_ops.hip): S^T[latent][point] = K_l Q^T per 32-latent tile, online softmax over a lane's registers + its
// partner half, O^T[d][point] += V_l^T P^T with the probabilities staying in the lane that computed them; the point's
// own (k, v) pair joins the softmax at the end (one more logit per point, its value row added to the accumulators).
// fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products and sums).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void point_attention_mfma_kernel(const float *__restrict__ qkv_p,
                                                                  const float *__restrict__ qkv_l, float *__restrict__ out,
                                                                  int M, int Ll, int heads, float scale) {
    constexpr int DQ = D / 8;                 // float4 operand quads per lane along d
    const int lane = threadIdx.x, l32 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, C = heads * D, p0 = blockIdx.y * 32;
    const float *lbase = qkv_l + (size_t)b * Ll * 3 * C + h * D;
    const int prow = min(p0 + l32, M - 1);
    const float *src = qkv_p + ((size_t)b * M + prow) * 3 * C + h * D;
    f32x4v qf[DQ];
    float s_self = 0.f;
#pragma unroll
    for (int t = 0; t < DQ; t++) {
        qf[t] = *reinterpret_cast<const f32x4v *>(src + 4 * (2 * t + half)) * scale;
        const f32x4v ks = *reinterpret_cast<const f32x4v *>(src + C + 4 * (2 * t + half));
        s_self += (qf[t].x * ks.x + qf[t].y * ks.y) + (qf[t].z * ks.z + qf[t].w * ks.w);
    }
    s_self += __shfl_xor(s_self, 32, 64);
    f32x16v o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = 0.f;
    float mx = -INFINITY, den = 0.f;
    for (int k0 = 0; k0 < Ll; k0 += 32) {
        const int krow = min(k0 + l32, Ll - 1);
        f32x16v sT;
#pragma unroll
        for (int r = 0; r < 16; r++) sT[r] = 0.f;
#pragma unroll
        for (int t = 0; t < DQ; t++) {
            const f32x4v kf = *reinterpret_cast<const f32x4v *>(lbase + (size_t)krow * 3 * C + C + 4 * (2 * t + half));
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s4], qf[t][s4], sT, 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
            sT[r] = key < Ll ? sT[r] : -INFINITY;
            tmax = fmaxf(tmax, sT[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float nm = fmaxf(mx, tmax), corr = expf(mx - nm);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sT[r] = expf(sT[r] - nm);
            psum += sT[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        den = den * corr + psum;
        mx = nm;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] *= corr;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = min(k0 + 8 * (r >> 2) + 4 * half + (r & 3), Ll - 1);
            const float vf = lbase[(size_t)key * 3 * C + 2 * C + l32];
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sT[r], o, 0, 0, 0);
        }
    }
    // the point itself: logit s_self, value row v_self
    const float nm = fmaxf(mx, s_self), corr = expf(mx - nm), p_self = expf(s_self - nm);
    den = den * corr + p_self;
    const float inv = 1.0f / den;
    if (p0 + l32 < M) {
        float *dst = out + ((size_t)b * M + p0 + l32) * C + h * D;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4v vs = *reinterpret_cast<const f32x4v *>(src + 2 * C + 8 * g + 4 * half);
            const f32x4v v = {(o[4 * g] * corr + p_self * vs.x) * inv, (o[4 * g + 1] * corr + p_self * vs.y) * inv,
                              (o[4 * g + 2] * corr + p_self * vs.z) * inv, (o[4 * g + 3] * corr + p_self * vs.w) * inv};
            *reinterpret_cast<f32x4v *>(dst + 8 * g + 4 * half) = v;
        }
    }
}

// Backward.  Per point: dP = dO V^T (and dO v_self), dS = P (dP - sum P dP), dq, dk_self, dv_self
// written per point; dK_l / dV_l accumulated over the tile's points in registers (thread owns the
// (j, d) pairs tid + 256 r; d = tid % 32 is the same for all of them) and written as one partial
// per workgroup, reduced over tiles by point_attention_reduce_kernel.
constexpr int ACC_R = (256 * D + 255) / 256;      // 32 accumulator pairs cover Ll <= 256
__global__ __launch_bounds__(256) void point_attention_bwd_kernel(const float *__restrict__ qkv_p,
                                                                  const float *__restrict__ qkv_l,
                                                                  const float *__restrict__ dout,
                                                                  float *__restrict__ dqkv_p,
                                                                  float *__restrict__ partial, int M, int Ll, int heads,
                                                                  float scale) {
    extern __shared__ float lds[];
    float *Ks = lds, *Vs = Ks + Ll * KS, *prow = Vs + Ll * KS, *dsrow = prow + 4 * Ll;   // [4][Ll] each
    float *qv = dsrow + 4 * Ll, *gv = qv + 4 * D;                                     // [4][D] each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, C = heads * D;
    stage_kv(qkv_l, Ks, Vs, b, h, Ll, C);
    float aK[ACC_R], aV[ACC_R];
#pragma unroll
    for (int r = 0; r < ACC_R; r++) aK[r] = aV[r] = 0.f;
    __syncthreads();
    const int i0 = blockIdx.y * PT, i_end = min(M, i0 + PT);
    float *pw = prow + wave * Ll, *dw = dsrow + wave * Ll;
    const int dmine = threadIdx.x & 31, jbase = threadIdx.x >> 5;
    for (int ib = i0; ib < i_end; ib += 4) {
        const int i = ib + wave;
        const bool live = i < i_end;
        if (live) {
            const float *src = qkv_p + ((size_t)b * M + i) * 3 * C + h * D;
            const float *gsrc = dout + ((size_t)b * M + i) * C + h * D;
            float q[D], ks[D], go[D];
#pragma unroll
            for (int d = 0; d < D; d++) { q[d] = src[d]; ks[d] = src[C + d]; go[d] = gsrc[d]; }
            const Row r = softmax_row(q, ks, Ks, Ll, lane, scale);
            float dp[PA_MAXJ], delta = 0.f;
#pragma unroll
            for (int t = 0; t < PA_MAXJ; t++) {
                const int j = lane + 64 * t;
                dp[t] = 0.f;
                if (j < Ll) {
#pragma unroll
                    for (int d = 0; d < D; d++) dp[t] += go[d] * Vs[j * KS + d];
                }
                delta += r.p[t] * dp[t];
            }
            float dp_self = 0.f;
#pragma unroll
            for (int d = 0; d < D; d++) dp_self += go[d] * src[2 * C + d];
            delta = wave_sum(delta) + r.p_self * dp_self;
            const float ds_self = r.p_self * (dp_self - delta);
#pragma unroll
            for (int t = 0; t < PA_MAXJ; t++) {
                const int j = lane + 64 * t;
                if (j < Ll) {
                    pw[j] = r.p[t];
                    dw[j] = r.p[t] * (dp[t] - delta);
                }
            }
            // q / dO vectors for the accumulation phase
            if (lane < D) { qv[wave * D + lane] = src[lane]; gv[wave * D + lane] = gsrc[lane]; }
            // dq (lanes own d, two halves of the keys), dk_self, dv_self
            const int d = lane & 31, part = lane >> 5;
            float acc = 0.f;
            for (int j = part; j < Ll; j += 2) acc += dw[j] * Ks[j * KS + d];
            acc += __shfl_xor(acc, 32, 64);
            if (part == 0) {
                float *o = dqkv_p + ((size_t)b * M + i) * 3 * C + h * D + d;
                const float qd = qv[wave * D + d], gd = gv[wave * D + d];
                o[0] = scale * (acc + ds_self * src[C + d]);
                o[C] = scale * ds_self * qd;
                o[2 * C] = r.p_self * gd;
            }
        } else {
            for (int j = lane; j < Ll; j += 64) { pw[j] = 0.f; dw[j] = 0.f; }
            if (lane < D) { qv[wave * D + lane] = 0.f; gv[wave * D + lane] = 0.f; }
        }
        __syncthreads();
        float qw[4], gw[4];
#pragma unroll
        for (int w = 0; w < 4; w++) { qw[w] = qv[w * D + dmine]; gw[w] = gv[w * D + dmine]; }
#pragma unroll
        for (int r = 0; r < ACC_R; r++) {
            const int j = jbase + 8 * r;
            if (j < Ll) {
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    aK[r] += dsrow[w * Ll + j] * qw[w];
                    aV[r] += prow[w * Ll + j] * gw[w];
                }
            }
        }
        __syncthreads();
    }
    // partial [tile][bh][2][Ll][D]
    float *dst = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * Ll * D;
#pragma unroll
    for (int r = 0; r < ACC_R; r++) {
        const int j = jbase + 8 * r;
        if (j < Ll) {
            dst[j * D + dmine] = aK[r] * scale;
            dst[Ll * D + j * D + dmine] = aV[r];
        }
    }
}

// ---- the backward on the MFMA pipe --------------------------------------------------------------------------- //
// pass 1, one wave per (b, h, 32 points), transposed like the forward kernel: S^T = K_l Q^T and dP^T = V_l dO^T of
// all latent tiles stay in the accumulators; with the point's own logit and dO . v_self they give the probabilities,
// delta = sum P dP and dS = P (dP - delta).  dS^T in the accumulators IS the B operand of dq^T = K_l^T dS^T (the forward
// kernel's O^T = V^T P^T again: step r pairs key(r, half) on both operands), so dq = scale (dS K_l + ds_self k_self) is
// finished here - no dS^T copy in memory (103 MB per call at 4 x 4,096 points, written dword-wise and read back), no
// read-modify-write of dq in pass 2.  Written: dS and P as rows [BH][M][LS] and the point gradients (dq, dk_self = scale
// ds_self q, dv_self = p_self dO).
// pass 2, one wave per 32 x 32 output tile: dK_l = scale dS^T Q and dV_l = P^T dO over chunks of PA2_CHUNK points ->
// partial tiles, summed in chunk order by point_attention_reduce_kernel.
constexpr int PA2_TILES = 8;          // Ll <= 256
constexpr int PA2_CHUNK = 512;        // points per dK_l / dV_l partial
__global__ __launch_bounds__(64) void point_attention_bwd_probs_kernel(
    const float *__restrict__ qkv_p, const float *__restrict__ qkv_l, const float *__restrict__ dout,
    float *__restrict__ dqkv_p, float *__restrict__ dSbuf, float *__restrict__ Pbuf, int M, int Ll, int LS, int heads,
    float scale) {
    static_assert(D == 32, "dq^T is one 32-row MFMA tile");
    constexpr int DQ = D / 8;
    const int lane = threadIdx.x, l32 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, C = heads * D, p0 = blockIdx.y * 32;
    const int LT = (Ll + 31) / 32;
    const float *lbase = qkv_l + (size_t)b * Ll * 3 * C + h * D;
    const int prow = min(p0 + l32, M - 1);
    const float *src = qkv_p + ((size_t)b * M + prow) * 3 * C + h * D;
    const float *gsrc = dout + ((size_t)b * M + prow) * C + h * D;
    f32x4v qf[DQ], gf[DQ], qr[DQ], ksf[DQ];
    float s_self = 0.f, dp_self = 0.f;
#pragma unroll
    for (int t = 0; t < DQ; t++) {
        qr[t] = *reinterpret_cast<const f32x4v *>(src + 4 * (2 * t + half));
        qf[t] = qr[t] * scale;
        gf[t] = *reinterpret_cast<const f32x4v *>(gsrc + 4 * (2 * t + half));
        ksf[t] = *reinterpret_cast<const f32x4v *>(src + C + 4 * (2 * t + half));
        const f32x4v vs = *reinterpret_cast<const f32x4v *>(src + 2 * C + 4 * (2 * t + half));
        s_self += (qf[t].x * ksf[t].x + qf[t].y * ksf[t].y) + (qf[t].z * ksf[t].z + qf[t].w * ksf[t].w);
        dp_self += (gf[t].x * vs.x + gf[t].y * vs.y) + (gf[t].z * vs.z + gf[t].w * vs.w);
    }
    s_self += __shfl_xor(s_self, 32, 64);
    dp_self += __shfl_xor(dp_self, 32, 64);
    f32x16v sT[PA2_TILES], dT[PA2_TILES];
    float mx = s_self;
    // a lone wave per SIMD: the K_l / V_l rows of latent tile kt + 1 are requested in front of the MFMAs of tile kt
    f32x4v kf[DQ], vf[DQ], kn[DQ] = {}, vn[DQ] = {};
    auto fetch = [&](int kt, f32x4v (&k_)[DQ], f32x4v (&v_)[DQ]) {
        const float *row = lbase + (size_t)min(kt * 32 + l32, Ll - 1) * 3 * C + 4 * half;
#pragma unroll
        for (int t = 0; t < DQ; t++) {
            k_[t] = *reinterpret_cast<const f32x4v *>(row + C + 8 * t);
            v_[t] = *reinterpret_cast<const f32x4v *>(row + 2 * C + 8 * t);
        }
    };
    fetch(0, kf, vf);
#pragma unroll
    for (int kt = 0; kt < PA2_TILES; kt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) { sT[kt][r] = -INFINITY; dT[kt][r] = 0.f; }
        if (kt < LT) {
            if (kt + 1 < LT) fetch(kt + 1, kn, vn);
            f32x16v a, c;
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
                sT[kt][r] = key < Ll ? a[r] : -INFINITY;
                dT[kt][r] = c[r];
                mx = fmaxf(mx, sT[kt][r]);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float den = 0.f;
#pragma unroll
    for (int kt = 0; kt < PA2_TILES; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sT[kt][r] = expf(sT[kt][r] - mx);
            den += sT[kt][r];
        }
    den += __shfl_xor(den, 32, 64);
    float p_self = expf(s_self - mx);
    den += p_self;
    const float inv = 1.0f / den;
    p_self *= inv;
    float delta = 0.f;
#pragma unroll
    for (int kt = 0; kt < PA2_TILES; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sT[kt][r] *= inv;
            delta += sT[kt][r] * dT[kt][r];
        }
    delta += __shfl_xor(delta, 32, 64);
    delta += p_self * dp_self;
    const float ds_self = p_self * (dp_self - delta);
    const bool live = p0 + l32 < M;
    // dS^T = P (dP - delta) replaces dP^T in its registers; dq^T[d][point] += K_l[key][d] dS^T[key][point]
    f32x16v dq;
#pragma unroll
    for (int r = 0; r < 16; r++) dq[r] = 0.f;
    float kv[16], kvn[16] = {};
    auto fetch_k = [&](int kt, float (&k_)[16]) {
#pragma unroll
        for (int r = 0; r < 16; r++)           // padded keys re-read the last row: their dS is 0
            k_[r] = lbase[(size_t)min(kt * 32 + 8 * (r >> 2) + 4 * half + (r & 3), Ll - 1) * 3 * C + C + l32];
    };
    fetch_k(0, kv);
#pragma unroll
    for (int kt = 0; kt < PA2_TILES; kt++)
        if (kt < LT) {
            if (kt + 1 < LT) fetch_k(kt + 1, kvn);
#pragma unroll
            for (int r = 0; r < 16; r++) dT[kt][r] = sT[kt][r] * (dT[kt][r] - delta);
#pragma unroll
            for (int r = 0; r < 16; r++) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[r], dT[kt][r], dq, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) kv[r] = kvn[r];
        }
    if (live) {
        float *o = dqkv_p + ((size_t)b * M + p0 + l32) * 3 * C + h * D;
#pragma unroll
        for (int t = 0; t < DQ; t++) {
            const f32x4v dqt = {dq[4 * t], dq[4 * t + 1], dq[4 * t + 2], dq[4 * t + 3]};    // d = 8 t + 4 half + e
            *reinterpret_cast<f32x4v *>(o + 4 * (2 * t + half)) = (ksf[t] * ds_self + dqt) * scale;
            *reinterpret_cast<f32x4v *>(o + C + 4 * (2 * t + half)) = qr[t] * (scale * ds_self);
            *reinterpret_cast<f32x4v *>(o + 2 * C + 4 * (2 * t + half)) = gf[t] * p_self;
        }
    }
    // rows of LS = Ll rounded up to 4 floats: a lane's four consecutive keys leave as ONE aligned 16-byte store (the padding
    // receives P = exp(-inf) = 0 and dS = 0).  Per-key dword stores to 64 different rows per instruction ran this kernel at
    // 1.3 TB/s (240 us per call at 4 x 4,096 points; tools/ubench/store_rate.hip: dword stores are 4x slower)
    float *dSrow = dSbuf + ((size_t)bh * M + min(p0 + l32, M - 1)) * LS;
    float *Prow = Pbuf + ((size_t)bh * M + min(p0 + l32, M - 1)) * LS;
#pragma unroll
    for (int kt = 0; kt < PA2_TILES; kt++)
        if (kt < LT)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int key0 = kt * 32 + 8 * g + 4 * half;
                const f32x4v pv = {sT[kt][4 * g], sT[kt][4 * g + 1], sT[kt][4 * g + 2], sT[kt][4 * g + 3]};
                const f32x4v ds = {dT[kt][4 * g], dT[kt][4 * g + 1], dT[kt][4 * g + 2], dT[kt][4 * g + 3]};
                if (live && key0 < LS) {
                    *reinterpret_cast<f32x4v *>(dSrow + key0) = ds;
                    *reinterpret_cast<f32x4v *>(Prow + key0) = pv;
                }
            }
}

// which 1 / 2: a dK_l / dV_l partial (32 latents x one chunk of points)
__global__ __launch_bounds__(64) void point_attention_bwd_gemm_kernel(
    const float *__restrict__ qkv_p, const float *__restrict__ dout, float *__restrict__ partial,
    const float *__restrict__ dSbuf, const float *__restrict__ Pbuf, int M, int Ll, int LS, int heads, float scale, int LT,
    int MS) {
    const int lane = threadIdx.x, l32 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, C = heads * D;
    int t = blockIdx.y;
    const int which = 1 + t / (LT * MS);
    t %= LT * MS;
    const int r0 = (t / MS) * 32, ms = t % MS;
    f32x16v acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int m = r0 + l32;                                    // this lane's A row
    constexpr int U = 8;
    // the operands of the next eight K = 2 steps are requested in front of the MFMAs of the current ones
    const bool m_ok = m < Ll;
    const float *A = (which == 1 ? dSbuf : Pbuf) + (size_t)bh * M * LS + min(m, Ll - 1);   // A[m = latent][k = point]
    const float *Bp = which == 1 ? qkv_p + (size_t)b * M * 3 * C + h * D + l32 : dout + (size_t)b * M * C + h * D + l32;
    const size_t b_step = which == 1 ? 3 * C : C;
    const int k_end = min(M, (ms + 1) * PA2_CHUNK);
    float av[U], bv[U], an[U] = {}, bn[U] = {};
    auto fetch = [&](int k0, float (&a_)[U], float (&b_)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kc = min(k0 + 2 * u + half, k_end - 1);
            a_[u] = A[(size_t)kc * LS];
            b_[u] = Bp[(size_t)kc * b_step];
        }
    };
    fetch(ms * PA2_CHUNK, av, bv);
    for (int k0 = ms * PA2_CHUNK; k0 < k_end; k0 += 2 * U) {
        if (k0 + 2 * U < k_end) fetch(k0 + 2 * U, an, bn);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool ok = k0 + 2 * u + half < k_end;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((ok && m_ok) ? av[u] : 0.f, ok ? bv[u] : 0.f, acc, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) { av[u] = an[u]; bv[u] = bn[u]; }
    }
    float *dst = partial + (((size_t)ms * gridDim.x + bh) * 2 + (which - 1)) * Ll * D;
    const float mul = which == 1 ? scale : 1.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = r0 + 8 * (r >> 2) + 4 * half + (r & 3);
        if (row < Ll) dst[(size_t)row * D + l32] = acc[r] * mul;
    }
}

// dqkv_l[b][j][{1,2}][h][d] (+)= sum over tiles; the q columns are zeroed when not accumulating
__global__ __launch_bounds__(256) void point_attention_reduce_kernel(const float *__restrict__ partial,
                                                                     float *__restrict__ dqkv_l, int tiles, int BH,
                                                                     int Ll, int heads, int accumulate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)BH * 3 * Ll * D;
    if (i >= total) return;
    const int d = i % D, j = (i / D) % Ll, which = (i / D / Ll) % 3, bh = i / D / Ll / 3;
    const int b = bh / heads, h = bh % heads, C = heads * D;
    float *o = dqkv_l + ((size_t)b * Ll + j) * 3 * C + which * C + h * D + d;
    if (which == 0) {
        if (!accumulate) *o = 0.f;
        return;
    }
    float s = 0.f;
    for (int t = 0; t < tiles; t++) s += partial[(((size_t)t * BH + bh) * 2 + (which - 1)) * Ll * D + j * D + d];
    *o = accumulate ? *o + s : s;
}

// ---- Loss.shape_loss (utils/loss.py:18-28) ----
__device__ __forceinline__ float bce_weight(float sdf, float thres, float weight) {
    return fabsf(sdf) < thres ? weight : 1.0f;
}
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float *__restrict__ x, const float *__restrict__ sdf,
                                                      size_t n, float thres, float weight, float *__restrict__ partial) {
    __shared__ float lds[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i], t = sdf[i] < 0.f ? 1.f : 0.f;
        // torch binary_cross_entropy_with_logits: (1 - t) x + max(-x, 0) + log(exp(-max) + exp(-x - max))
        const float l = fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
        s += l * bce_weight(sdf[i], thres, weight);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}
__global__ __launch_bounds__(256) void bce_finish_kernel(const float *__restrict__ partial, int blocks, float inv_n,
                                                         float *__restrict__ loss) {
    __shared__ float lds[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < blocks; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = ((lds[0] + lds[1]) + (lds[2] + lds[3])) * inv_n;
}
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float *__restrict__ x, const float *__restrict__ sdf,
                                                      size_t n, float thres, float weight,
                                                      const float *__restrict__ grad_loss, float *__restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i], t = sdf[i] < 0.f ? 1.f : 0.f;
    const float sig = 1.0f / (1.0f + expf(-v));
    dx[i] = (sig - t) * bce_weight(sdf[i], thres, weight) * (*grad_loss) / (float)n;
}

// ---- multi-tensor kernels over a table of tensors ----
struct TensorEntry {             // mirrors include/zeroshape_hip.h zs_tensor_entry
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    unsigned long long n;
    float lr, weight_decay;
};
constexpr int MT_CHUNK = 16384;  // elements per workgroup

// chunk table: chunk c covers elements [start[c], start[c] + MT_CHUNK) of tensor tid[c]
__global__ __launch_bounds__(256) void adamw_multi_kernel(const TensorEntry *__restrict__ tab,
                                                          const int *__restrict__ chunk_tensor,
                                                          const unsigned long long *__restrict__ chunk_start,
                                                          float beta1, float beta2, float eps, int step,
                                                          const float *__restrict__ grad_scale,
                                                          const int *__restrict__ skipped_steps) {
    const TensorEntry t = tab[chunk_tensor[blockIdx.x]];
    const unsigned long long s0 = chunk_start[blockIdx.x], s1 = min(t.n, s0 + (unsigned long long)MT_CHUNK);
    const float gs = grad_scale ? *grad_scale : 1.0f;
    if (!(gs > 0.0f || gs < 0.0f)) return;      // 0 or nan: a loss scaler found an overflowed gradient - skip the step
    // a skipped step is no step: torch's GradScaler never calls optimizer.step() on an overflow, so the bias
    // correction counts the steps that were APPLIED (host count minus the device-side count of skips)
    const int applied = max(1, step - (skipped_steps ? *skipped_steps : 0));
    const float bias1 = (float)(1.0 - pow((double)beta1, (double)applied));
    const float bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)applied));
    for (unsigned long long i = s0 + threadIdx.x; i < s1; i += 256) {
        // torch.optim.AdamW (single-tensor path): decoupled decay, then Adam with bias correction
        const float g = t.grad[i] * gs;
        float p = t.param[i] * (1.0f - t.lr * t.weight_decay);
        const float m = t.exp_avg[i] + (g - t.exp_avg[i]) * (1.0f - beta1);          // lerp, as torch does
        const float v = beta2 * t.exp_avg_sq[i] + (1.0f - beta2) * g * g;
        const float denom = sqrtf(v) / bias2_sqrt + eps;
        p -= (t.lr / bias1) * (m / denom);
        t.param[i] = p;
        t.exp_avg[i] = m;
        t.exp_avg_sq[i] = v;
    }
}

// gather: dst (entry.param) <- src (entry.grad) * scale, used to pack gradients into a flat bucket
__global__ __launch_bounds__(256) void copy_multi_kernel(const TensorEntry *__restrict__ tab,
                                                         const int *__restrict__ chunk_tensor,
                                                         const unsigned long long *__restrict__ chunk_start, float scale) {
    const TensorEntry t = tab[chunk_tensor[blockIdx.x]];
    const unsigned long long s0 = chunk_start[blockIdx.x], s1 = min(t.n, s0 + (unsigned long long)MT_CHUNK);
    for (unsigned long long i = s0 + threadIdx.x; i < s1; i += 256) t.param[i] = t.grad[i] * scale;
}

// per-chunk sum of squares of entry.grad -> partial[chunk]; finish with bce_finish_kernel (inv_n = 1)
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const TensorEntry *__restrict__ tab,
                                                          const int *__restrict__ chunk_tensor,
                                                          const unsigned long long *__restrict__ chunk_start,
                                                          float *__restrict__ partial) {
    __shared__ float lds[4];
    const TensorEntry t = tab[chunk_tensor[blockIdx.x]];
    const unsigned long long s0 = chunk_start[blockIdx.x], s1 = min(t.n, s0 + (unsigned long long)MT_CHUNK);
    float s = 0.f;
    for (unsigned long long i = s0 + threadIdx.x; i < s1; i += 256) s += t.grad[i] * t.grad[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

}  // namespace

#define ZS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            zs::set_err(__VA_ARGS__);    \
            return 0;                    \
        }                                \
    } while (0)

static size_t pa_lds_bytes(int Ll, bool bwd) {
    return ((size_t)2 * Ll * KS + (bwd ? 8 : 4) * (size_t)Ll + (bwd ? 8 * D : 0)) * sizeof(float);
}

extern "C" int zs_point_attention(const float *qkv_points, const float *qkv_latent, float *out, int batch, int M,
                                  int Ll, int heads, int head_dim, void *stream) {
    ZS_REQUIRE(batch >= 0 && M > 0 && Ll > 0 && Ll <= 64 * PA_MAXJ && heads > 0 && head_dim == D,
               "zs_point_attention: bad size (B=%d M=%d Ll=%d heads=%d head_dim=%d; Ll <= %d, head_dim %d)", batch, M, Ll,
               heads, head_dim, 64 * PA_MAXJ, D);
    if (batch == 0) return 1;
    ZS_REQUIRE(qkv_points && qkv_latent && out, "zs_point_attention: null pointer");
    static const bool valu = getenv("ZS_POINT_ATTN_VALU") != nullptr;        // A/B switch: the vector-ALU kernel
    if (!valu && (M + 31) / 32 <= 65535) {
        hipLaunchKernelGGL(point_attention_mfma_kernel, dim3(batch * heads, (M + 31) / 32), dim3(64), 0, S(stream),
                           qkv_points, qkv_latent, out, M, Ll, heads, 1.0f / sqrtf((float)head_dim));
        return zs::check_launch("zs_point_attention") ? 1 : 0;
    }
    const dim3 grid(batch * heads, (M + PT - 1) / PT);
    hipLaunchKernelGGL(point_attention_kernel, grid, dim3(256), pa_lds_bytes(Ll, false), S(stream), qkv_points,
                       qkv_latent, out, M, Ll, heads, 1.0f / sqrtf((float)head_dim));
    return zs::check_launch("zs_point_attention") ? 1 : 0;
}

extern "C" int zs_point_attention_probs(const float *qkv_points, const float *qkv_latent, float *attn, int batch, int M,
                                        int Ll, int heads, int head_dim, float weight, int accumulate, void *stream) {
    ZS_REQUIRE(batch >= 0 && batch <= 65535 && M > 0 && Ll > 0 && Ll <= 64 * PA_MAXJ && heads > 0 && head_dim == D,
               "zs_point_attention_probs: bad size (B=%d M=%d Ll=%d heads=%d head_dim=%d; Ll <= %d, head_dim %d)", batch, M, Ll,
               heads, head_dim, 64 * PA_MAXJ, D);
    if (batch == 0) return 1;
    ZS_REQUIRE(qkv_points && qkv_latent && attn, "zs_point_attention_probs: null pointer");
    ZS_REQUIRE((M + PP - 1) / PP <= 65535, "zs_point_attention_probs: too many points per call (M=%d)", M);
    hipLaunchKernelGGL(point_attention_probs_kernel, dim3(batch, (M + PP - 1) / PP), dim3(256), pa_lds_bytes(Ll, false), S(stream),
                       qkv_points, qkv_latent, attn, M, Ll, heads, 1.0f / sqrtf((float)head_dim), weight, accumulate ? 1 : 0);
    return zs::check_launch("zs_point_attention_probs") ? 1 : 0;
}

static size_t pa_partial_floats(int batch, int M, int Ll, int heads) {
    const int tiles = (M + PT - 1) / PT, chunks = (M + PA2_CHUNK - 1) / PA2_CHUNK;
    return (size_t)(tiles > chunks ? tiles : chunks) * batch * heads * 2 * Ll * D;
}
extern "C" size_t zs_point_attention_bwd_workspace_bytes(int batch, int M, int Ll, int heads) {
    // [partial tiles of dK_l / dV_l][dS rows][P rows]; the rows are padded to 16 bytes
    return (pa_partial_floats(batch, M, Ll, heads) + (size_t)batch * heads * M * 2 * ((Ll + 3) & ~3)) * sizeof(float);
}

extern "C" int zs_point_attention_bwd(const float *qkv_points, const float *qkv_latent, const float *dout,
                                      float *dqkv_points, float *dqkv_latent, int accumulate_latent, void *workspace,
                                      int batch, int M, int Ll, int heads, int head_dim, void *stream) {
    ZS_REQUIRE(batch >= 0 && M > 0 && Ll > 0 && Ll <= 64 * PA_MAXJ && heads > 0 && head_dim == D,
               "zs_point_attention_bwd: bad size (B=%d M=%d Ll=%d heads=%d head_dim=%d)", batch, M, Ll, heads, head_dim);
    if (batch == 0) return 1;
    ZS_REQUIRE(qkv_points && qkv_latent && dout && dqkv_points && dqkv_latent && workspace,
               "zs_point_attention_bwd: null pointer");
    const int BH = batch * heads;
    float *partial = static_cast<float *>(workspace);
    static const bool bwd_valu = getenv("ZS_POINT_ATTN_VALU") != nullptr;    // A/B switch: the vector-ALU kernel
    const int MT = (M + 31) / 32, LT = (Ll + 31) / 32, MS = (M + PA2_CHUNK - 1) / PA2_CHUNK;
    if (!bwd_valu && LT <= PA2_TILES && MT <= 65535 && 2 * LT * MS <= 65535) {
        const float scale = 1.0f / sqrtf((float)head_dim);
        const int LS = (Ll + 3) & ~3;
        ZS_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "zs_point_attention_bwd: workspace must be 16-byte aligned");
        float *dS = partial + pa_partial_floats(batch, M, Ll, heads), *PTb = dS + (size_t)BH * M * LS;
        hipLaunchKernelGGL(point_attention_bwd_probs_kernel, dim3(BH, MT), dim3(64), 0, S(stream), qkv_points, qkv_latent,
                           dout, dqkv_points, dS, PTb, M, Ll, LS, heads, scale);
        hipLaunchKernelGGL(point_attention_bwd_gemm_kernel, dim3(BH, 2 * LT * MS), dim3(64), 0, S(stream), qkv_points, dout,
                           partial, dS, PTb, M, Ll, LS, heads, scale, LT, MS);
        if (!zs::check_launch("zs_point_attention_bwd")) return 0;
        hipLaunchKernelGGL(point_attention_reduce_kernel, dim3(blocks_for((size_t)BH * 3 * Ll * D)), dim3(256), 0, S(stream),
                           partial, dqkv_latent, MS, BH, Ll, heads, accumulate_latent ? 1 : 0);
        return zs::check_launch("zs_point_attention_bwd(reduce)") ? 1 : 0;
    }
    const int tiles = (M + PT - 1) / PT;
    const dim3 grid(BH, tiles);
    hipLaunchKernelGGL(point_attention_bwd_kernel, grid, dim3(256), pa_lds_bytes(Ll, true), S(stream), qkv_points,
                       qkv_latent, dout, dqkv_points, partial, M, Ll, heads, 1.0f / sqrtf((float)head_dim));
    if (!zs::check_launch("zs_point_attention_bwd")) return 0;
    hipLaunchKernelGGL(point_attention_reduce_kernel, dim3(blocks_for((size_t)BH * 3 * Ll * D)), dim3(256), 0, S(stream),
                       partial, dqkv_latent, tiles, BH, Ll, heads, accumulate_latent ? 1 : 0);
    return zs::check_launch("zs_point_attention_bwd(reduce)") ? 1 : 0;
}

static int bce_blocks(size_t n) {
    size_t b = (n + 1023) / 1024;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" size_t zs_bce_logits_workspace_bytes(size_t n) { return (size_t)bce_blocks(n) * sizeof(float); }

extern "C" int zs_bce_logits(const float *logits, const float *sdf, size_t n, float impt_thres, float impt_weight,
                             float *loss, void *workspace, void *stream) {
    ZS_REQUIRE(n > 0, "zs_bce_logits: empty input");
    ZS_REQUIRE(logits && sdf && loss && workspace, "zs_bce_logits: null pointer");
    const int blocks = bce_blocks(n);
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(blocks), dim3(256), 0, S(stream), logits, sdf, n, impt_thres, impt_weight,
                       partial);
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(256), 0, S(stream), partial, blocks, 1.0f / (float)n, loss);
    return zs::check_launch("zs_bce_logits") ? 1 : 0;
}

extern "C" int zs_bce_logits_bwd(const float *logits, const float *sdf, size_t n, float impt_thres, float impt_weight,
                                 const float *grad_loss, float *dlogits, void *stream) {
    ZS_REQUIRE(n > 0, "zs_bce_logits_bwd: empty input");
    ZS_REQUIRE(logits && sdf && grad_loss && dlogits, "zs_bce_logits_bwd: null pointer");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, S(stream), logits, sdf, n, impt_thres,
                       impt_weight, grad_loss, dlogits);
    return zs::check_launch("zs_bce_logits_bwd") ? 1 : 0;
}

static_assert(sizeof(TensorEntry) == sizeof(zs_tensor_entry), "TensorEntry must mirror zs_tensor_entry");

extern "C" int zs_multi_tensor_chunk_elems(void) { return MT_CHUNK; }

extern "C" int zs_adamw_multi(const zs_tensor_entry *table, const int *chunk_tensor,
                              const unsigned long long *chunk_start, int n_chunks, float beta1, float beta2, float eps,
                              int step, const float *grad_scale, const int *skipped_steps, void *stream) {
    ZS_REQUIRE(n_chunks >= 0 && step >= 1, "zs_adamw_multi: bad arguments (chunks=%d step=%d)", n_chunks, step);
    if (n_chunks == 0) return 1;
    ZS_REQUIRE(table && chunk_tensor && chunk_start, "zs_adamw_multi: null pointer");
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(n_chunks), dim3(256), 0, S(stream),
                       reinterpret_cast<const TensorEntry *>(table), chunk_tensor, chunk_start, beta1, beta2, eps, step,
                       grad_scale, skipped_steps);
    return zs::check_launch("zs_adamw_multi") ? 1 : 0;
}

extern "C" int zs_copy_multi(const zs_tensor_entry *table, const int *chunk_tensor, const unsigned long long *chunk_start,
                             int n_chunks, float scale, void *stream) {
    ZS_REQUIRE(n_chunks >= 0, "zs_copy_multi: bad arguments");
    if (n_chunks == 0) return 1;
    ZS_REQUIRE(table && chunk_tensor && chunk_start, "zs_copy_multi: null pointer");
    hipLaunchKernelGGL(copy_multi_kernel, dim3(n_chunks), dim3(256), 0, S(stream),
                       reinterpret_cast<const TensorEntry *>(table), chunk_tensor, chunk_start, scale);
    return zs::check_launch("zs_copy_multi") ? 1 : 0;
}

extern "C" int zs_sumsq_multi(const zs_tensor_entry *table, const int *chunk_tensor, const unsigned long long *chunk_start,
                              int n_chunks, float *partial, float *sumsq, void *stream) {
    ZS_REQUIRE(n_chunks > 0, "zs_sumsq_multi: bad arguments");
    ZS_REQUIRE(table && chunk_tensor && chunk_start && partial && sumsq, "zs_sumsq_multi: null pointer");
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(n_chunks), dim3(256), 0, S(stream),
                       reinterpret_cast<const TensorEntry *>(table), chunk_tensor, chunk_start, partial);
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(256), 0, S(stream), partial, n_chunks, 1.0f, sumsq);
    return zs::check_launch("zs_sumsq_multi") ? 1 : 0;
}
