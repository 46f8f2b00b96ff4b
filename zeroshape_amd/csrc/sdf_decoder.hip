// Fused implicit-occupancy decoder for MI355X (gfx950): one launch evaluates
// Implicit.forward (model/shape/implicit.py:251-288) for every query point, with the
// point-independent latent half hoisted into the prologue (csrc/sdf_prologue.hip).
//
// Mapping to the hardware (zeroshape_amd/program.py):
//  * a wave owns 32 query points for the whole network; a workgroup is 4 waves
//    (one per SIMD, 1 workgroup per CU: the kernel uses the full 512-register file);
//  * every activation lives in registers in the v_mfma_f32_32x32x2_f32 accumulator
//    layout and each layer is computed transposed (Y^T = W X^T), so a layer's output
//    registers ARE the next layer's B operands: no LDS traffic, no transposes;
//  * weights arrive as one linear stream of pre-packed A operands ("records",
//    4 per 16-byte load), prefetched 8 loads (32 MFMAs) ahead through a register ring;
//  * biases / LayerNorm affine / xyz columns sit in LDS (54 KiB, loaded once);
//  * exact-fp32 MFMA (bitwise an fmaf chain) - parity mode; 39,424 MFMAs per wave tile
//    = 5.05 MFLOP per point including the 224-vs-197 latent padding.
//
// Per-point arithmetic follows the reference op for op (LayerNorm eps 1e-6, softmax
// over 197 latent logits + 1 self logit, exact-erf GELU, softplus(beta=100,
// threshold=20), cat(..)/sqrt(2) skips); only summation order differs.
#include "zs_common.h"
#include "sdf_layout.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>

namespace {

using namespace zs::lay;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 4;
constexpr int PTS_PER_WAVE = 32;
constexpr int PTS_PER_BLOCK = WAVES * PTS_PER_WAVE;

#define DEV __device__ __forceinline__
// Bounds the machine scheduler's window: without it hipcc interleaves neighbouring output
// tiles / layers of the fully unrolled code and the live ranges overflow the register file.
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// ---- weight stream: register ring, 8 x 16-byte loads in flight per lane ------------- //
struct AStream {
    const f32x4 *__restrict__ base;  // wave-uniform: group (consumed + RING)
    int lane;
    f32x4 ring[RING];

    DEV void init(const f32x4 *b, int ln) {
        lane = ln;
#pragma unroll
        for (int i = 0; i < RING; i++) ring[i] = b[i * 64 + ln];
        base = b + RING * 64;
    }
    DEV f32x4 next(int slot) {  // slot is a compile-time constant after unrolling
        f32x4 a = ring[slot];
        ring[slot] = base[lane];
        base += 64;
        return a;
    }
};

DEV f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc += W_tile * X.  X = KT activation tiles as 16*KT scalars (register r of tile kt is
// X[16*kt + r]); consumes KT*4 groups starting at ring slot `phase` (kt-major, then
// register) - the order program.py packs them in.  `phase` (0 or 4) must be a
// compile-time constant at every call site after unrolling.
// Activations are deliberately plain scalars, not f32x16 tuples: only accumulators need
// 16-register tuples, and scalars let the allocator place each B operand in either half
// of the unified register file.
template <int KT>
DEV void gemm_tile(AStream &s, const float *X, f32x16 &acc, int phase) {
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 a = s.next((phase + kt * 4 + g) & (RING - 1));
            acc = mfma(a.x, X[kt * 16 + 4 * g + 0], acc);
            acc = mfma(a.y, X[kt * 16 + 4 * g + 1], acc);
            acc = mfma(a.z, X[kt * 16 + 4 * g + 2], acc);
            acc = mfma(a.w, X[kt * 16 + 4 * g + 3], acc);
        }
    }
}

// one-tile variant whose B operand is an accumulator tuple (q, P, o, hidden)
DEV void gemm_tile_v(AStream &s, const f32x16 &X, f32x16 &acc, int phase) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const f32x4 a = s.next((phase + g) & (RING - 1));
        acc = mfma(a.x, X[4 * g + 0], acc);
        acc = mfma(a.y, X[4 * g + 1], acc);
        acc = mfma(a.z, X[4 * g + 2], acc);
        acc = mfma(a.w, X[4 * g + 3], acc);
    }
}

// same as gemm_tile<NT>, with the B operands (feat / sqrt(2)) read back from the wave's
// LDS slab ([kt][g][lane] float4: lane-contiguous -> conflict-free ds_read_b128)
DEV void gemm_tile_lds(AStream &s, const f32x4 *fl, f32x16 &acc, int phase) {
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 a = s.next((phase + kt * 4 + g) & (RING - 1));
            const f32x4 b = fl[(kt * 4 + g) * 64];
            acc = mfma(a.x, b.x, acc);
            acc = mfma(a.y, b.y, acc);
            acc = mfma(a.z, b.z, acc);
            acc = mfma(a.w, b.w, acc);
        }
    }
}

DEV float xhalf(float v) { return __shfl_xor(v, 32, 64); }  // value of lane l ^ 32

// row-param read from LDS: 16 floats for (tile, lane half)
DEV void rp(const float *prm, int off, int tile, int hi, float *v) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(prm + off + tile * 32 + hi * 16);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f32x4 a = q[i];
        v[4 * i + 0] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
    }
}
DEV f32x16 rp16(const float *prm, int off, int tile, int hi) {
    float t[16];
    rp(prm, off, tile, hi, t);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = t[r];
    return v;
}

// w.w + w.x*x + w.y*y + w.z*z for the 16 registers of (tile, hi): [tile][hi][r][4] table
DEV f32x16 xyz_affine(const float *prm, int off, int tile, int hi, float x, float y, float z) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(prm + off + tile * 128 + hi * 64);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const f32x4 w = q[r];
        v[r] = fmaf(w.z, z, fmaf(w.y, y, fmaf(w.x, x, w.w)));
    }
    return v;
}

// LayerNorm statistics over the 256 features of each point: 8 accumulator tuples
// (16 regs each) x 2 lane halves
DEV void ln_stats(const f32x16 *x, float &mean, float &rstd) {
    float s = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += x[kt][r];
    s += xhalf(s);
    mean = s * (1.0f / 256.0f);
    float v = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float d = x[kt][r] - mean;
            v = fmaf(d, d, v);
        }
    v += xhalf(v);
    rstd = 1.0f / sqrtf(v * (1.0f / 256.0f) + 1e-6f);
}
DEV void layer_norm(const f32x16 *x, float *h, const float *prm, int g_off, int b_off, int hi) {
    float mean, rstd;
    ln_stats(x, mean, rstd);
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        float g[16], b[16];
        rp(prm, g_off, kt, hi, g);
        rp(prm, b_off, kt, hi, b);
#pragma unroll
        for (int r = 0; r < 16; r++) h[kt * 16 + r] = fmaf((x[kt][r] - mean) * rstd, g[r], b[r]);
    }
}

// Branch-free erff (two polynomial regimes, both evaluated, then selected); max error
// 0.99 ulp (5.8e-8 abs) against erf() in fp64 - checked on the host in
// tests/test_device_math.py with the same coefficients.
DEV float erf_nb(float a) {
    const float t = fabsf(a), s2 = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s2, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s2, 4.99119423e-3f);
    q = fmaf(q, s2, -2.67681349e-2f);
    q = fmaf(q, s2, 1.12819925e-1f);
    q = fmaf(q, s2, -3.76125336e-1f);
    q = fmaf(q, s2, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? big : q;
}

// exact-erf GELU (nn.GELU default; timm Mlp)
DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_nb(x * 0.70710678118654752440f)); }

// torch.nn.Softplus(beta=100, threshold=20): z > 20 ? x : log1p(exp(z)) / 100, z = 100 x.
// Evaluated branch-free in the overflow-safe form max(x,0) + log1p(exp(-|z|)) / 100 (same
// function).  log1p(t) = log(w) + (t - (w - 1)) / w with w = fl(1 + t): the second term is
// the rounding error of w, so no special case is needed when w == 1.  7e-9 max abs error
// vs fp64 (the reference's own fp32 formula: 1.6e-8); host check in tests/test_device_math.py.
DEV float softplus100(float x) {
    const float z = x * 100.0f;
    const float t = __expf(-fabsf(z));  // (0, 1]
    const float w = 1.0f + t;
    const float c = t - (w - 1.0f);
    const float l = fmaf(c, __frcp_rn(w), __logf(w));
    const float r = fmaf(l, 0.01f, fmaxf(x, 0.0f));
    return z > 20.0f ? x : r;
}

// One wave: 32 points (lane & 31; both lane halves carry the same point).
// `prm`: LDS params region (phase A: program params [0, P_PHASE_B); phase B: the rest);
// `fl`: this wave's LDS slab for feat / sqrt(2), already offset by lane.
DEV float decode_tile(const f32x4 *__restrict__ recs, const float *__restrict__ prog_params,
                      float *prm, f32x4 *fl, float px, float py, float pz, int lane) {
    const int hi = lane >> 5;
    AStream s;
    s.init(recs, lane);

    // point_proj (implicit.py:128-131); y is the residual stream, kept as accumulators
    f32x16 y[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) y[kt] = xyz_affine(prm, P_PP, kt, hi, px, py, pz);

    const float scale = 0.17677669529663688110f;  // 32 ** -0.5
    float h[NT * 16];

#pragma unroll 1
    for (int blk = 0; blk < BLOCKS; blk++) {
        const int pb = P_BLK0 + blk * P_BLK_STRIDE;
        layer_norm(y, h, prm, pb + PB_LN1G, pb + PB_LN1B, hi);
        // y = x + proj_bias + sum_heads Wproj_h o_h
#pragma unroll
        for (int nt = 0; nt < NT; nt++) y[nt] += rp16(prm, pb + PB_BPROJ, nt, hi);

#pragma unroll 1
        for (int hd = 0; hd < HEADS; hd++) {
            f32x16 q = rp16(prm, pb + PB_BQKV, hd * 3 + 0, hi);
            gemm_tile<NT>(s, h, q, 0);
            f32x16 k = rp16(prm, pb + PB_BQKV, hd * 3 + 1, hi);
            gemm_tile<NT>(s, h, k, 0);
            f32x16 v = rp16(prm, pb + PB_BQKV, hd * 3 + 2, hi);
            gemm_tile<NT>(s, h, v, 0);

            // self logit (implicit.py:44)
            float s_self = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) s_self = fmaf(q[r], k[r], s_self);
            s_self = (s_self + xhalf(s_self)) * scale;

            // online softmax over 7 latent tiles (+ self), o = sum P V
            float m_run = -INFINITY, z_run = 0.f;
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; r++) o[r] = 0.f;
#pragma unroll 1
            for (int lt = 0; lt < LT; lt++) {
                f32x16 S;
#pragma unroll
                for (int r = 0; r < 16; r++) S[r] = 0.f;
                gemm_tile_v(s, q, S, 0);
                const int lim = (lt == LT - 1) ? (L - 32 * (LT - 1)) : 64;  // valid rows in tile
                float mt = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rw = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float sv = rw < lim ? S[r] * scale : -INFINITY;
                    S[r] = sv;
                    mt = fmaxf(mt, sv);
                }
                mt = fmaxf(mt, xhalf(mt));
                const float m_new = fmaxf(m_run, mt);
                const float alpha = __expf(m_run - m_new);
                float zs_ = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float p = __expf(S[r] - m_new);
                    S[r] = p;
                    zs_ += p;
                }
                z_run = fmaf(z_run, alpha, zs_);
#pragma unroll
                for (int r = 0; r < 16; r++) o[r] *= alpha;
                gemm_tile_v(s, S, o, 4);
                m_run = m_new;
            }
            {
                const float m_new = fmaxf(m_run, s_self);
                const float alpha = __expf(m_run - m_new);
                const float p_self = __expf(s_self - m_new);
                const float z = fmaf(z_run + xhalf(z_run), alpha, p_self);
                const float inv = 1.0f / z;
#pragma unroll
                for (int r = 0; r < 16; r++) o[r] = fmaf(p_self, v[r], o[r] * alpha) * inv;
            }
            // y += Wproj[:, head] o_h
#pragma unroll
            for (int nt = 0; nt < NT; nt++) gemm_tile_v(s, o, y[nt], (nt & 1) * 4);
        }

        // MLP (timm Mlp): y += b2 + W2 gelu(W1 LN2(y) + b1), one hidden tile at a time
        layer_norm(y, h, prm, pb + PB_LN2G, pb + PB_LN2B, hi);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) y[nt] += rp16(prm, pb + PB_B2, nt, hi);
#pragma unroll 1
        for (int ht = 0; ht < HT; ht++) {
            f32x16 hid = rp16(prm, pb + PB_B1, ht, hi);
            gemm_tile<NT>(s, h, hid, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) hid[r] = gelu_erf(hid[r]);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) gemm_tile_v(s, hid, y[nt], (nt & 1) * 4);
        }
    }

    // final norm (implicit.py:275) -> h
    layer_norm(y, h, prm, P_LNFG, P_LNFB, hi);

    // phase B params (impl_mlp) replace the phase A ones in LDS
    __syncthreads();
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(prog_params + P_PHASE_B);
        f32x4 *dst = reinterpret_cast<f32x4 *>(prm);
        for (int i = threadIdx.x; i < (P_USED - P_PHASE_B + 3) / 4; i += WAVES * 64) dst[i] = src[i];
    }
    __syncthreads();

    // impl_mlp (implicit.py:168-184): inputs = cat[xyz, feat]; feat = h
    float cur[NT * 16];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        f32x16 acc = xyz_affine(prm, P_IMPL0 - P_PHASE_B, nt, hi, px, py, pz);
        gemm_tile<NT>(s, h, acc, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) cur[nt * 16 + r] = softplus100(acc[r]);
    }
    // the skip layers consume inputs / sqrt(2); park feat / sqrt(2) in LDS (frees 128 registers)
    const float sqrt2 = 1.41421356237309504880f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            f32x4 t;
            t.x = h[kt * 16 + 4 * g + 0] / sqrt2;
            t.y = h[kt * 16 + 4 * g + 1] / sqrt2;
            t.z = h[kt * 16 + 4 * g + 2] / sqrt2;
            t.w = h[kt * 16 + 4 * g + 3] / sqrt2;
            fl[(kt * 4 + g) * 64] = t;
        }
    const float sx = px / sqrt2, sy = py / sqrt2, sz = pz / sqrt2;

    // layer 1 (plain): cur -> h, pre-divided by sqrt(2) because layer 2 is a skip layer
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        f32x16 acc = rp16(prm, P_IMPL1 - P_PHASE_B, nt, hi);
        gemm_tile<NT>(s, cur, acc, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) h[nt * 16 + r] = softplus100(acc[r]) / sqrt2;
    }
#pragma unroll 1
    for (int i = 0; i < 3; i++) {
        const int pp = P_IMPL_PAIR - P_PHASE_B + i * P_IMPL_PAIR_STRIDE;
        // skip layer 2+2i: cat[x, xyz, feat] / sqrt(2) (h already holds x / sqrt(2)) -> cur
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            f32x16 acc = xyz_affine(prm, pp, nt, hi, sx, sy, sz);
            gemm_tile<NT>(s, h, acc, 0);
            gemm_tile_lds(s, fl, acc, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) cur[nt * 16 + r] = softplus100(acc[r]);
        }
        // plain layer 3+2i: cur -> h (/ sqrt(2) when the next layer is a skip layer)
        const float post = i < 2 ? sqrt2 : 1.0f;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            f32x16 acc = rp16(prm, pp + 1024, nt, hi);
            gemm_tile<NT>(s, cur, acc, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) h[nt * 16 + r] = softplus100(acc[r]) / post;
        }
    }
    // layer 8: 256 -> 1
    float out = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        float w[16];
        rp(prm, P_W8 - P_PHASE_B, kt, hi, w);
#pragma unroll
        for (int r = 0; r < 16; r++) out = fmaf(h[kt * 16 + r], w[r], out);
    }
    out += xhalf(out);
    return out + prm[P_B8 - P_PHASE_B];
}

template <bool GRID>
__global__ __launch_bounds__(WAVES * 64, 1) void sdf_decode_kernel(
    const float *__restrict__ programs, size_t program_stride_floats,
    const float *__restrict__ points,  // !GRID: [batch][m][3]
    const float *__restrict__ axis,    //  GRID: [G]
    int G, long long first_point,      //  GRID: linear index of the first grid point
    int m,                             // points per image handled by this launch
    float *__restrict__ out, int apply_sigmoid) {
    // LDS: [params 32 KiB][4 x 32 KiB feat slabs] = 160 KiB, one workgroup per CU
    __shared__ __attribute__((aligned(16))) float lds[P_PHASE_B + WAVES * NT * 16 * 64];
    float *prm = lds;

    const int img = blockIdx.y;
    const float *prog = programs + (size_t)img * program_stride_floats;
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(prog + REC_FLOATS);
        f32x4 *dst = reinterpret_cast<f32x4 *>(prm);
        for (int i = threadIdx.x; i < P_PHASE_B / 4; i += WAVES * 64) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int p = blockIdx.x * PTS_PER_BLOCK + wave * PTS_PER_WAVE + (lane & 31);
    const int pc = p < m ? p : m - 1;  // clamp: tail lanes recompute the last point
    float px, py, pz;
    if (GRID) {
        const long long gp = first_point + pc;
        const long long gg = (long long)G * G;
        const int ix = (int)(gp / gg);
        const int rem = (int)(gp - (long long)ix * gg);
        const int iy = rem / G;
        const int iz = rem - iy * G;
        px = axis[ix];
        py = axis[iy];
        pz = axis[iz];
    } else {
        const float *q = points + ((size_t)img * m + pc) * 3;
        px = q[0];
        py = q[1];
        pz = q[2];
    }
    f32x4 *fl = reinterpret_cast<f32x4 *>(lds + P_PHASE_B + wave * (NT * 16 * 64)) + lane;
    float logit = decode_tile(reinterpret_cast<const f32x4 *>(prog), prog + REC_FLOATS, prm, fl, px, py,
                              pz, lane);
    if (apply_sigmoid) logit = 1.0f / (1.0f + expf(-logit));
    if (lane < 32 && p < m) out[(size_t)img * m + p] = logit;
}

}  // namespace

extern "C" size_t zs_sdf_program_bytes(void) { return (size_t)PROGRAM_FLOATS * sizeof(float); }

extern "C" int zs_sdf_query_points(const void *programs, size_t program_stride_bytes, int batch,
                                   const float *points, int m, float *logits, float *attn,
                                   void *stream) {
    if (batch < 0 || m < 0) {
        zs::set_err("zs_sdf_query_points: negative size (batch=%d m=%d)", batch, m);
        return 0;
    }
    if (batch == 0 || m == 0) return 1;
    if (!programs || !points || !logits) {
        zs::set_err("zs_sdf_query_points: null pointer");
        return 0;
    }
    if (attn) {
        zs::set_err("zs_sdf_query_points: attention output not implemented in this build");
        return 0;
    }
    if (program_stride_bytes % 16 != 0 || program_stride_bytes < zs_sdf_program_bytes()) {
        zs::set_err("zs_sdf_query_points: bad program stride %zu", program_stride_bytes);
        return 0;
    }
    dim3 grid((m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK, batch);
    hipLaunchKernelGGL(sdf_decode_kernel<false>, grid, dim3(WAVES * 64), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float *>(programs),
                       program_stride_bytes / sizeof(float), points, nullptr, 0, 0LL, m, logits, 0);
    return zs::check_launch("zs_sdf_query_points") ? 1 : 0;
}

extern "C" int zs_sdf_query_grid(const void *programs, size_t program_stride_bytes, int batch,
                                 const float *axis, int G, int slice_begin, int slice_end,
                                 int apply_sigmoid, float *out, void *stream) {
    if (batch < 0 || G <= 0 || slice_begin < 0 || slice_end > G || slice_begin > slice_end) {
        zs::set_err("zs_sdf_query_grid: bad range (batch=%d G=%d slices=[%d,%d))", batch, G,
                    slice_begin, slice_end);
        return 0;
    }
    const long long mm = (long long)(slice_end - slice_begin) * G * G;
    if (batch == 0 || mm == 0) return 1;
    if (!programs || !axis || !out) {
        zs::set_err("zs_sdf_query_grid: null pointer");
        return 0;
    }
    if (mm > 0x7fffffffLL - PTS_PER_BLOCK) {
        zs::set_err("zs_sdf_query_grid: %lld points per launch exceed 2^31; split the slab", mm);
        return 0;
    }
    if (program_stride_bytes % 16 != 0 || program_stride_bytes < zs_sdf_program_bytes()) {
        zs::set_err("zs_sdf_query_grid: bad program stride %zu", program_stride_bytes);
        return 0;
    }
    const int m = (int)mm;
    dim3 grid((m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK, batch);
    hipLaunchKernelGGL(sdf_decode_kernel<true>, grid, dim3(WAVES * 64), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float *>(programs),
                       program_stride_bytes / sizeof(float), nullptr, axis, G,
                       (long long)slice_begin * G * G, m, out, apply_sigmoid);
    return zs::check_launch("zs_sdf_query_grid") ? 1 : 0;
}
