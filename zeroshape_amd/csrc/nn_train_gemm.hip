// Training-side GEMM work of the convolution / linear layers (fp32 MFMA, v_mfma_f32_32x32x2_f32):
//   zs_pack_conv_weight  torch-layout weights -> the [K16/4][CoutPad][4] operand of zs_conv2d_nhwc,
//                        either for the forward product or for the data gradient (taps flipped,
//                        Cin <-> Cout swapped), on the GPU so it can run every optimiser step
//   zs_conv2d_wgrad      dW[cout][k] = sum_pixels dY[pixel][cout] * A[pixel][k]  (A = the same
//                        implicit im2col operand as the forward pass, incl. its input transform)
//   zs_standardize_weight / _bwd   timm StdConv2d weight standardisation and its adjoint
// The data gradient itself is zs_conv2d_nhwc on dY with the flipped pack (ZS_CONV_IN_DILATE2 for
// stride 2).  Used by: Implicit (model/shape/implicit.py) in training, DPT / ResNet encoders
// (training half).
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }
inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

// ---------------------------------------------------------------------------------------------
// weight packing.  Source: w[cout * ld + (cin0 + c) * kh*kw + tap], c < Cin (a channel sub-range
// of a torch [Cout][CinTot][kh][kw] tensor, ld = CinTot*kh*kw).
//   forward pack : GEMM K = taps * CinP  (CinP = Cin rounded up to 4, zero filled), N = Cout
//                  k = tap * CinP + c           -> w[n][c][tap]
//   dgrad pack   : GEMM K = taps * CoutP (CoutP = Cout rounded up to 4), N = Cin
//                  k = tap' * CoutP + co        -> w[co][n][taps-1-tap']   (both axes flipped)
// dst[(k/4) * NPad * 4 + n * 4 + k%4], NPad = N rounded up to 128, K16 = K rounded up to 16.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ dst,
                                                          int Cout, int Cin, int cin0, int ld, int taps, int dgrad,
                                                          int K16, int NPad) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)K16 * NPad;
    if (i >= total) return;
    const int e = i & 3, n = (i >> 2) % NPad, k = (int)((i >> 2) / NPad) * 4 + e;
    float v = 0.f;
    if (!dgrad) {
        const int CinP = (Cin + 3) & ~3, tap = k / CinP, c = k - tap * CinP;
        if (tap < taps && c < Cin && n < Cout) v = w[(size_t)n * ld + (size_t)(cin0 + c) * taps + tap];
    } else {
        const int CoutP = (Cout + 3) & ~3, tap = k / CoutP, co = k - tap * CoutP;
        if (tap < taps && co < Cout && n < Cin) v = w[(size_t)co * ld + (size_t)(cin0 + n) * taps + (taps - 1 - tap)];
    }
    dst[i] = v;
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  Workgroup tile: 128 couts x 128 k over a range of pixels; the pixel range is
// split over blockIdx.z and the partial sums are reduced in a fixed order by wgrad_reduce_kernel
// (deterministic).  LDS holds the two operand tiles in their natural [pixel][column] layout; an
// MFMA step contracts two pixels: lane (l32, half) supplies dY[p+half][cout l32] and
// A[p+half][k l32] - plain ds_read_b32, rows padded to 160 floats so the two halves hit
// disjoint banks.
// ---------------------------------------------------------------------------------------------
constexpr int WM = 128, WN = 128, WP = 16, WLD = 160;

struct WgradArgs {
    const float *in, *dy;
    float *partial;                 // [splits][CoutP][K]
    float *bias_partial;            // [splits][CoutP] column sums of dY (the bias gradient), or NULL
    int B, Hin, Win, Cin, Hout, Wout, CoutP, kh, kw, stride, pad_t, pad_l, K, M;
    int in_relu;
    float in_scale, in_shift;
    int pix_per_split;
};

__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds_y[2][WP][WLD];
    __shared__ __attribute__((aligned(16))) float lds_a[2][WP][WLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int c0 = blockIdx.x * WM, k0 = blockIdx.y * WN;
    const int p_begin = blockIdx.z * a.pix_per_split, p_end = min(a.M, p_begin + a.pix_per_split);

    // loader role: rows (pixels) tid/32 and tid/32 + 8 of the step, column quad tid%32
    const int prow = tid >> 5, quad = tid & 31;
    // dY column quad
    const int yc = c0 + 4 * quad;
    const bool yc_ok = yc < a.CoutP;
    // A column quad: k fixed for the whole loop
    const int kk = k0 + 4 * quad;
    const bool k_ok = kk < a.K;
    int kc = 0, ky = 0, kx = 0;
    if (k_ok) {
        const int tap = kk / a.Cin;
        kc = kk - tap * a.Cin;
        ky = tap / a.kw;
        kx = tap - ky * a.kw;
    }
    const float relu_floor = a.in_relu ? 0.f : -INFINITY;
    const int HW = a.Hout * a.Wout;

    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    struct Frag { f32x4 y, x; bool ok; };
    auto load = [&](int p) -> Frag {
        Frag f;
        const bool p_ok = p < p_end;
        const int pc = p_ok ? p : p_begin;
        f.y = yc_ok ? *reinterpret_cast<const f32x4 *>(a.dy + (size_t)pc * a.CoutP + (yc_ok ? yc : 0)) : f32x4{0, 0, 0, 0};
        if (!p_ok) f.y = f32x4{0, 0, 0, 0};
        const int pb = pc / HW, rem = pc - pb * HW, py = rem / a.Wout, px = rem - py * a.Wout;
        const int iy = py * a.stride - a.pad_t + ky, ix = px * a.stride - a.pad_l + kx;
        f.ok = p_ok && k_ok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const size_t off = f.ok ? (((size_t)pb * a.Hin + iy) * a.Win + ix) * a.Cin + kc : 0;
        f.x = *reinterpret_cast<const f32x4 *>(a.in + off);
        return f;
    };
    auto store = [&](int buf, int row, const Frag &f) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = f.ok ? fmaxf(f.x[e], relu_floor) * a.in_scale + a.in_shift : 0.f;
        *reinterpret_cast<f32x4 *>(&lds_y[buf][row][4 * quad]) = f.y;
        *reinterpret_cast<f32x4 *>(&lds_a[buf][row][4 * quad]) = v;
        bsum += f.y;                 // every dY row of the tile passes through exactly one thread per column quad
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const int steps = (p_end - p_begin + WP - 1) / WP;
    if (steps > 0) {
        Frag f0 = load(p_begin + prow), f1 = load(p_begin + prow + 8);
        store(0, prow, f0);
        store(0, prow + 8, f1);
        __syncthreads();
        for (int s = 0; s < steps; s++) {
            const int cur = s & 1;
            const bool more = s + 1 < steps;
            if (more) {
                f0 = load(p_begin + (s + 1) * WP + prow);
                f1 = load(p_begin + (s + 1) * WP + prow + 8);
            }
#pragma unroll
            for (int t = 0; t < WP / 2; t++) {
                const float ya0 = lds_y[cur][2 * t + half][wm + l32], ya1 = lds_y[cur][2 * t + half][wm + 32 + l32];
                const float xb0 = lds_a[cur][2 * t + half][wn + l32], xb1 = lds_a[cur][2 * t + half][wn + 32 + l32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya0, xb0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya0, xb1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya1, xb0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya1, xb1, acc[1][1], 0, 0, 0);
            }
            if (more) {
                store(cur ^ 1, prow, f0);
                store(cur ^ 1, prow + 8, f1);
            }
            __syncthreads();
        }
    }
    if (a.bias_partial && blockIdx.y == 0) {          // bias gradient: column sums of this tile's dY rows
        __syncthreads();
        *reinterpret_cast<f32x4 *>(&lds_y[0][prow][4 * quad]) = bsum;
        __syncthreads();
        if (prow == 0 && yc_ok) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 8; r++) t += *reinterpret_cast<const f32x4 *>(&lds_y[0][r][4 * quad]);
            *reinterpret_cast<f32x4 *>(a.bias_partial + (size_t)blockIdx.z * a.CoutP + yc) = t;
        }
    }
    float *dst = a.partial + (size_t)blockIdx.z * a.CoutP * a.K;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int k = k0 + wn + 32 * j + l32;
        if (k >= a.K) continue;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = c0 + wm + 32 * i + 8 * (r >> 2) + 4 * half + (r & 3);
                if (co < a.CoutP) dst[(size_t)co * a.K + k] = acc[i][j][r];
            }
    }
}

// partial [splits][CoutP][K] (k = tap*CinP + c) -> dw[cout*ld + (cin0+c)*taps + tap], cout < Cout, c < Cin
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw,
                                                           int splits, int CoutP, int K, int Cout, int Cin, int CinP,
                                                           int cin0, int ld, int taps, int accumulate,
                                                           const float *__restrict__ bias_partial,
                                                           float *__restrict__ db) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)Cout * K;
    if (db && i < (size_t)Cout) {
        float t = 0.f;
        for (int z = 0; z < splits; z++) t += bias_partial[(size_t)z * CoutP + i];
        db[i] = t;
    }
    if (i >= total) return;
    const int k = i % K, co = i / K, tap = k / CinP, c = k - tap * CinP;
    if (c >= Cin) return;
    float s = 0.f;
    for (int z = 0; z < splits; z++) s += partial[((size_t)z * CoutP + co) * K + k];
    float *o = dw + (size_t)co * ld + (size_t)(cin0 + c) * taps + tap;
    *o = accumulate ? *o + s : s;
}

// ---- StdConv2d weight standardisation: per output channel (w - mean) / sqrt(biased var + eps) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum256(float v, float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__global__ __launch_bounds__(256) void std_weight_kernel(const float *__restrict__ w, float *__restrict__ out, int n,
                                                         float eps) {
    __shared__ float lds[4];
    const float *r = w + (size_t)blockIdx.x * n;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) s += r[e];
    const float mean = block_sum256(s, lds) / n;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) { const float d = r[e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum256(q, lds) / n + eps);
    for (int e = threadIdx.x; e < n; e += 256) out[(size_t)blockIdx.x * n + e] = (r[e] - mean) * rstd;
}

// dw = rstd * (g - mean(g) - what * mean(g * what))
__global__ __launch_bounds__(256) void std_weight_bwd_kernel(const float *__restrict__ w, const float *__restrict__ g,
                                                             float *__restrict__ dw, int n, float eps) {
    __shared__ float lds[4];
    const float *r = w + (size_t)blockIdx.x * n, *gr = g + (size_t)blockIdx.x * n;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) s += r[e];
    const float mean = block_sum256(s, lds) / n;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) { const float d = r[e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum256(q, lds) / n + eps);
    float sg = 0.f, sgx = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) { sg += gr[e]; sgx += gr[e] * (r[e] - mean) * rstd; }
    const float mg = block_sum256(sg, lds) / n, mgx = block_sum256(sgx, lds) / n;
    for (int e = threadIdx.x; e < n; e += 256)
        dw[(size_t)blockIdx.x * n + e] = rstd * (gr[e] - mg - (r[e] - mean) * rstd * mgx);
}

// ---- data gradient of a convolution with <= 4 input channels (the network stems) ----
// As a GEMM this has N = 3 columns in a 128-wide tile, and at stride 2 three quarters of the
// zero-stuffed taps are padding: 2 % useful work.  Direct form instead: one thread per input pixel
// gathers the (at most ceil(k/s)^2) output pixels that read it; weights sit in LDS as
// [tap][cout][4 channels] (every lane of a wave that needs an entry reads the same address).
__global__ __launch_bounds__(256) void dgrad_small_cin_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                              float *__restrict__ dx, int B, int H, int W, int CinP,
                                                              int Ho, int Wo, int Cout, int kh, int kw, int stride,
                                                              int pad_t, int pad_l, int Cin, int cin0, int ld,
                                                              float scale) {
    extern __shared__ f32x4 wl[];                    // [taps][Cout]
    const int taps = kh * kw;
    for (int e = threadIdx.x; e < taps * Cout; e += 256) {
        const int tap = e / Cout, co = e - tap * Cout;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < Cin; c++) v[c] = w[(size_t)co * ld + (size_t)(cin0 + c) * taps + tap];
        wl[e] = v;
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * H * W;
    if (i >= total) return;
    const int ix = i % W, iy = (i / W) % H, b = i / W / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < kh; ky++) {
        const int vy = iy + pad_t - ky;
        if (vy < 0 || vy % stride) continue;
        const int oy = vy / stride;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < kw; kx++) {
            const int vx = ix + pad_l - kx;
            if (vx < 0 || vx % stride) continue;
            const int ox = vx / stride;
            if (ox >= Wo) continue;
            const f32x4 *g = reinterpret_cast<const f32x4 *>(dy + (((size_t)b * Ho + oy) * Wo + ox) * Cout);
            const f32x4 *wt = wl + (ky * kw + kx) * Cout;
            for (int q = 0; q < Cout / 4; q++) {
                const f32x4 gv = g[q];
#pragma unroll
                for (int e = 0; e < 4; e++) acc += wt[4 * q + e] * gv[e];
            }
        }
    }
    f32x4 out = acc * scale;
    for (int c = Cin; c < 4; c++) out[c] = 0.f;
    float *o = dx + i * CinP;
    for (int c = 0; c < CinP; c++) o[c] = out[c];
}

}  // namespace

#define ZS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            zs::set_err(__VA_ARGS__);    \
            return 0;                    \
        }                                \
    } while (0)

extern "C" int zs_pack_conv_weight(const float *w, float *packed, int Cout, int Cin, int cin0, int CinTot, int kh,
                                   int kw, int dgrad, void *stream) {
    ZS_REQUIRE(Cout > 0 && Cin > 0 && cin0 >= 0 && cin0 + Cin <= CinTot && kh > 0 && kw > 0,
               "zs_pack_conv_weight: bad geometry (Cout=%d Cin=%d cin0=%d CinTot=%d k %dx%d)", Cout, Cin, cin0, CinTot,
               kh, kw);
    ZS_REQUIRE(w && packed, "zs_pack_conv_weight: null pointer");
    const int taps = kh * kw;
    const int Kc = dgrad ? (Cout + 3) / 4 * 4 : (Cin + 3) / 4 * 4, N = dgrad ? Cin : Cout;
    const int K16 = (taps * Kc + 15) / 16 * 16, NPad = (N + 127) / 128 * 128;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks_for((size_t)K16 * NPad)), dim3(256), 0, S(stream), w, packed,
                       Cout, Cin, cin0, CinTot * taps, taps, dgrad ? 1 : 0, K16, NPad);
    return zs::check_launch("zs_pack_conv_weight") ? 1 : 0;
}

static int wgrad_splits(long long M, int CoutP, int K) {
    const long long tiles = (long long)((CoutP + WM - 1) / WM) * ((K + WN - 1) / WN);
    static const long long target = getenv("ZS_WGRAD_TARGET") ? atoll(getenv("ZS_WGRAD_TARGET")) : 768;
    long long splits = (target + tiles - 1) / tiles;              // aim at ~3 workgroups per CU
    const long long max_by_pixels = (M + 127) / 128;              // at least 128 pixels per split
    if (splits > max_by_pixels) splits = max_by_pixels;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    return (int)splits;
}

extern "C" size_t zs_conv2d_wgrad_workspace_bytes(int batch, int Hout, int Wout, int Cin, int Cout, int kh, int kw) {
    const int CoutP = (Cout + 3) / 4 * 4, K = kh * kw * ((Cin + 3) / 4 * 4);
    const long long M = (long long)batch * Hout * Wout;
    return (size_t)wgrad_splits(M, CoutP, K) * CoutP * (K + 1) * sizeof(float);
}

extern "C" int zs_conv2d_wgrad(const float *in, const float *dy, float *dw, float *db, void *workspace, int batch, int Hin,
                               int Win, int CinP, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t,
                               int pad_l, int flags, float in_scale, float in_shift, int Cin, int cin0, int CinTot,
                               int accumulate, void *stream) {
    ZS_REQUIRE(batch >= 0 && Hin > 0 && Win > 0 && CinP > 0 && (CinP & 3) == 0 && Cin > 0 && Cin <= CinP &&
                   Hout > 0 && Wout > 0 && Cout > 0 && kh > 0 && kw > 0 && stride > 0 && cin0 >= 0 &&
                   cin0 + Cin <= CinTot,
               "zs_conv2d_wgrad: bad geometry (B=%d in %dx%dx%d out %dx%dx%d k %dx%d s %d; Cin %d at %d of %d)", batch,
               Hin, Win, CinP, Hout, Wout, Cout, kh, kw, stride, Cin, cin0, CinTot);
    if (batch == 0) return 1;
    ZS_REQUIRE(in && dy && dw && workspace, "zs_conv2d_wgrad: null pointer");
    const long long M = (long long)batch * Hout * Wout;
    ZS_REQUIRE(M <= (1LL << 30), "zs_conv2d_wgrad: %lld output pixels", M);
    WgradArgs a;
    a.in = in; a.dy = dy; a.partial = static_cast<float *>(workspace);
    a.B = batch; a.Hin = Hin; a.Win = Win; a.Cin = CinP; a.Hout = Hout; a.Wout = Wout;
    a.CoutP = (Cout + 3) / 4 * 4;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
    a.K = kh * kw * CinP; a.M = (int)M;
    a.in_relu = (flags & ZS_CONV_IN_RELU) ? 1 : 0;
    a.in_scale = in_scale; a.in_shift = in_shift;
    const int splits = wgrad_splits(M, a.CoutP, a.K);
    a.bias_partial = db ? a.partial + (size_t)splits * a.CoutP * a.K : nullptr;
    a.pix_per_split = (int)((M + splits - 1) / splits);
    a.pix_per_split = (a.pix_per_split + WP - 1) / WP * WP;
    const dim3 grid((a.CoutP + WM - 1) / WM, (a.K + WN - 1) / WN, splits);
    hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, S(stream), a);
    if (!zs::check_launch("zs_conv2d_wgrad")) return 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks_for((size_t)Cout * a.K)), dim3(256), 0, S(stream), a.partial, dw,
                       splits, a.CoutP, a.K, Cout, Cin, CinP, cin0, CinTot * kh * kw, kh * kw, accumulate ? 1 : 0, a.bias_partial,
                       db);
    return zs::check_launch("zs_conv2d_wgrad(reduce)") ? 1 : 0;
}

extern "C" int zs_standardize_weight(const float *w, float *out, int Cout, int n, float eps, void *stream) {
    ZS_REQUIRE(Cout > 0 && n > 0 && w && out, "zs_standardize_weight: bad arguments");
    hipLaunchKernelGGL(std_weight_kernel, dim3(Cout), dim3(256), 0, S(stream), w, out, n, eps);
    return zs::check_launch("zs_standardize_weight") ? 1 : 0;
}

extern "C" int zs_standardize_weight_bwd(const float *w, const float *grad_out, float *dw, int Cout, int n, float eps,
                                         void *stream) {
    ZS_REQUIRE(Cout > 0 && n > 0 && w && grad_out && dw, "zs_standardize_weight_bwd: bad arguments");
    hipLaunchKernelGGL(std_weight_bwd_kernel, dim3(Cout), dim3(256), 0, S(stream), w, grad_out, dw, n, eps);
    return zs::check_launch("zs_standardize_weight_bwd") ? 1 : 0;
}

extern "C" int zs_conv2d_dgrad_small_cin(const float *dy, const float *w, float *dx, int batch, int H, int W, int CinP,
                                         int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t, int pad_l,
                                         int Cin, int cin0, int CinTot, float scale, void *stream) {
    ZS_REQUIRE(batch >= 0 && H > 0 && W > 0 && CinP > 0 && CinP <= 4 && Cin > 0 && Cin <= CinP && Hout > 0 && Wout > 0 &&
                   Cout > 0 && (Cout & 3) == 0 && kh > 0 && kw > 0 && stride > 0 && cin0 >= 0 && cin0 + Cin <= CinTot &&
                   (size_t)kh * kw * Cout * 16 <= 160 * 1024,
               "zs_conv2d_dgrad_small_cin: bad geometry (B=%d in %dx%dx%d out %dx%dx%d k %dx%d s %d; Cin <= 4, "
               "Cout %% 4 == 0, taps*Cout*16 B of LDS)", batch, H, W, CinP, Hout, Wout, Cout, kh, kw, stride);
    if (batch == 0) return 1;
    ZS_REQUIRE(dy && w && dx, "zs_conv2d_dgrad_small_cin: null pointer");
    const size_t lds = (size_t)kh * kw * Cout * sizeof(f32x4);
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(dgrad_small_cin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(dgrad_small_cin_kernel, dim3(blocks_for((size_t)batch * H * W)), dim3(256), lds, S(stream), dy, w, dx,
                       batch, H, W, CinP, Hout, Wout, Cout, kh, kw, stride, pad_t, pad_l, Cin, cin0, CinTot * kh * kw, scale);
    return zs::check_launch("zs_conv2d_dgrad_small_cin") ? 1 : 0;
}
