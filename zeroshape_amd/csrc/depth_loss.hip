// The depth task's training losses (options/depth.yaml), forward and backward, fp32:
//   zs_midas_loss / _bwd  MidasLoss (model/depth/midas_loss.py:142-185, shrink_mask False):
//       scale-and-shift-invariant MAE  - per image, both maps are aligned by their masked median
//       and mean absolute deviation (:33-62, SSIMAE :112-119), L1 over the batch's valid pixels -
//       plus alpha * the multi-scale gradient-matching term (:122-139, :90-108, image-based
//       reduction :76-86) on the least-squares aligned (inverse) depth (:11-30).
//   zs_intr_loss / _bwd   Loss.intr_loss (utils/loss.py:36-43)
// One 1024-lane workgroup per image; medians by an 8-bit radix select over order-preserving keys
// (LDS integer histograms: exact and deterministic); every float reduction in a fixed order.
// The backward differentiates what torch.autograd differentiates in the reference: through the
// median (its gradient goes to the median element), the deviations, and the 2x2 least-squares solve.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int BLOCK = 1024;
constexpr int MAX_SCALES = 4;
// per-image record shared by the forward, finish and backward kernels
enum { S_N = 0, S_TP, S_SP, S_MP, S_TG, S_SG, S_A, S_SCALE, S_SHIFT, S_DETOK, S_R, S_A00, S_A01, S_A11, S_B0, S_B1,
       S_M0, S_M1, S_M2, S_M3, S_STRIDE = 24 };

__device__ __forceinline__ float block_sum(float v, float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = lds[0];
    for (int w = 1; w < BLOCK / 64; w++) r += lds[w];
    return r;
}
__device__ __forceinline__ double block_sum_d(double v, double *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = lds[0];
    for (int w = 1; w < BLOCK / 64; w++) r += lds[w];
    return r;
}
__device__ __forceinline__ int block_min_i(int v, int *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    int r = lds[0];
    for (int w = 1; w < BLOCK / 64; w++) r = min(r, lds[w]);
    return r;
}

__device__ __forceinline__ uint32_t order_key(float x) {     // monotone float -> uint
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// k-th smallest (0-based) of the valid elements of x: value, and the lowest index holding it
__device__ void radix_select(const float *__restrict__ x, const float *__restrict__ mask, int n, int k, int *hist,
                             int *ilds, float *value, int *index) {
    uint32_t prefix = 0;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        for (int e = threadIdx.x; e < 256; e += BLOCK) hist[e] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += BLOCK)
            if (mask[i] > 0.5f) {
                const uint32_t u = order_key(x[i]);
                if (pass == 0 || (u >> (shift + 8)) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1);
            }
        __syncthreads();
        if (threadIdx.x == 0) {
            int cum = 0, bin = 0;
            for (; bin < 256; bin++) {
                if (cum + hist[bin] > k) break;
                cum += hist[bin];
            }
            hist[256] = bin;
            hist[257] = k - cum;
        }
        __syncthreads();
        prefix = (prefix << 8) | (uint32_t)hist[256];
        k = hist[257];
        __syncthreads();
    }
    int first = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += BLOCK)
        if (mask[i] > 0.5f && order_key(x[i]) == prefix) first = min(first, i);
    *index = block_min_i(first, ilds);
    *value = key_value(prefix);
}

struct Align { float t, s; int m; };
__device__ Align align_stats(const float *x, const float *mask, int n, float count, int *hist, int *ilds, float *lds) {
    Align a = {0.f, 0.f, -1};
    if (count > 0.f) radix_select(x, mask, n, ((int)count - 1) / 2, hist, ilds, &a.t, &a.m);   // lower median
    float dev = 0.f;
    for (int i = threadIdx.x; i < n; i += BLOCK)
        if (mask[i] > 0.5f) dev += fabsf(x[i] - a.t);
    a.s = block_sum(dev, lds) / (count + 1.f);
    return a;
}

__global__ __launch_bounds__(BLOCK) void midas_fwd_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                          const float *__restrict__ mask, int H, int W, float alpha,
                                                          int scales, int inverse, float *__restrict__ stats) {
    __shared__ float lds[BLOCK / 64];
    __shared__ double ldsd[BLOCK / 64];
    __shared__ int ilds[BLOCK / 64];
    __shared__ int hist[258];
    const int b = blockIdx.x, n = H * W;
    const float *x = pred + (size_t)b * n, *g = target + (size_t)b * n, *m = mask + (size_t)b * n;
    float *st = stats + (size_t)b * S_STRIDE;
    float cnt = 0.f;
    for (int i = threadIdx.x; i < n; i += BLOCK) cnt += m[i] > 0.5f ? 1.f : 0.f;
    cnt = block_sum(cnt, lds);
    const Align ap = align_stats(x, m, n, cnt, hist, ilds, lds), ag = align_stats(g, m, n, cnt, hist, ilds, lds);
    float A = 0.f;
    for (int i = threadIdx.x; i < n; i += BLOCK)
        if (m[i] > 0.5f) A += fabsf((x[i] - ap.t) / (ap.s + 1e-6f) - (g[i] - ag.t) / (ag.s + 1e-6f));
    A = block_sum(A, lds);
    // least-squares scale / shift of the (inverse) prediction onto the (inverse) target
    double a00 = 0, a01 = 0, b0 = 0, b1 = 0;
    for (int i = threadIdx.x; i < n; i += BLOCK)
        if (m[i] > 0.5f) {
            const float p = inverse ? 1.f / (x[i] + 1e-6f) : x[i], t = inverse ? 1.f / (g[i] + 1e-6f) : g[i];
            a00 += (double)p * p; a01 += p; b0 += (double)p * t; b1 += t;
        }
    const float fa00 = (float)block_sum_d(a00, ldsd), fa01 = (float)block_sum_d(a01, ldsd), fa11 = cnt,
                fb0 = (float)block_sum_d(b0, ldsd), fb1 = (float)block_sum_d(b1, ldsd);
    const float det = fa00 * fa11 - fa01 * fa01;
    const bool det_ok = det != 0.f;
    const float sc = det_ok ? (fa11 * fb0 - fa01 * fb1) / (det + 1e-6f) : 0.f;
    const float sh = det_ok ? (-fa01 * fb0 + fa00 * fb1) / (det + 1e-6f) : 0.f;
    float R = 0.f, Ms[MAX_SCALES] = {0, 0, 0, 0};
    if (alpha > 0.f)
        for (int k = 0; k < scales; k++) {
            const int s = 1 << k, Hs = (H + s - 1) / s, Ws = (W + s - 1) / s;
            auto dval = [&](int r, int c, float &mv) -> float {
                const int i = (r * s) * W + c * s;
                mv = m[i] > 0.5f ? 1.f : 0.f;
                const float p = inverse ? 1.f / (x[i] + 1e-6f) : x[i], t = inverse ? 1.f / (g[i] + 1e-6f) : g[i];
                return mv * (sc * p + sh - t);
            };
            float img = 0.f, M = 0.f;
            for (int e = threadIdx.x; e < Hs * Ws; e += BLOCK) {
                const int r = e / Ws, c = e - r * Ws;
                float m0, m1;
                const float d0 = dval(r, c, m0);
                M += m0;
                if (c + 1 < Ws) { const float d1 = dval(r, c + 1, m1); img += fabsf(d1 - d0) * m0 * m1; }
                if (r + 1 < Hs) { const float d1 = dval(r + 1, c, m1); img += fabsf(d1 - d0) * m0 * m1; }
            }
            img = block_sum(img, lds);
            M = block_sum(M, lds);
            Ms[k] = M;
            R += M != 0.f ? img / M : img;
        }
    if (threadIdx.x == 0) {
        st[S_N] = cnt; st[S_TP] = ap.t; st[S_SP] = ap.s; st[S_MP] = __int_as_float(ap.m); st[S_TG] = ag.t; st[S_SG] = ag.s;
        st[S_A] = A; st[S_SCALE] = sc; st[S_SHIFT] = sh; st[S_DETOK] = det_ok ? 1.f : 0.f; st[S_R] = R;
        st[S_A00] = fa00; st[S_A01] = fa01; st[S_A11] = fa11; st[S_B0] = fb0; st[S_B1] = fb1;
        st[S_M0] = Ms[0]; st[S_M1] = Ms[1]; st[S_M2] = Ms[2]; st[S_M3] = Ms[3];
    }
}

// loss = sum_b A_b / (sum_b N_b + 1e-6) + alpha * mean_b R_b ; also leaves N_total behind the records
__global__ void midas_finish_kernel(float *__restrict__ stats, int B, float alpha, float *__restrict__ loss) {
    if (threadIdx.x != 0) return;
    float A = 0.f, N = 0.f, R = 0.f;
    for (int b = 0; b < B; b++) { A += stats[b * S_STRIDE + S_A]; N += stats[b * S_STRIDE + S_N]; R += stats[b * S_STRIDE + S_R]; }
    stats[(size_t)B * S_STRIDE] = N;
    *loss = A / (N + 1e-6f) + (alpha > 0.f ? alpha * R / B : 0.f);
}

__global__ __launch_bounds__(BLOCK) void midas_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                          const float *__restrict__ mask, int B, int H, int W, float alpha,
                                                          int scales, int inverse, const float *__restrict__ stats,
                                                          const float *__restrict__ grad_loss, float *__restrict__ dx) {
    __shared__ float lds[BLOCK / 64];
    const int b = blockIdx.x, n = H * W;
    const float *x = pred + (size_t)b * n, *g = target + (size_t)b * n, *m = mask + (size_t)b * n;
    const float *st = stats + (size_t)b * S_STRIDE;
    float *out = dx + (size_t)b * n;
    const float G = *grad_loss, Ntot = stats[(size_t)B * S_STRIDE];
    const float N = st[S_N], tp = st[S_TP], sp = st[S_SP], tg = st[S_TG], sg = st[S_SG];
    const int mp = __float_as_int(st[S_MP]);
    // ---- scale-and-shift-invariant MAE ----
    const float cw = G / (Ntot + 1e-6f), isp = 1.f / (sp + 1e-6f), isg = 1.f / (sg + 1e-6f);
    float S1 = 0.f, S2 = 0.f, S3 = 0.f;
    for (int i = threadIdx.x; i < n; i += BLOCK)
        if (m[i] > 0.5f) {
            const float c = cw * sgn((x[i] - tp) * isp - (g[i] - tg) * isg);
            S1 += c; S2 += c * (x[i] - tp); S3 += sgn(x[i] - tp);
        }
    S1 = block_sum(S1, lds); S2 = block_sum(S2, lds); S3 = block_sum(S3, lds);
    const float dLds = -S2 * isp * isp, dLdt = -S1 * isp - dLds * S3 / (N + 1.f);
    // ---- gradient-matching term: dL/dq per pixel (q = scale * p + shift), gathered over the scales ----
    const float sc = st[S_SCALE], sh = st[S_SHIFT];
    const bool det_ok = st[S_DETOK] > 0.5f;
    float Ga = 0.f, Gc = 0.f;
    auto pval = [&](int i) -> float { return inverse ? 1.f / (x[i] + 1e-6f) : x[i]; };
    auto tval = [&](int i) -> float { return inverse ? 1.f / (g[i] + 1e-6f) : g[i]; };
    auto dval = [&](int i, float &mv) -> float { mv = m[i] > 0.5f ? 1.f : 0.f; return mv * (sc * pval(i) + sh - tval(i)); };
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        float a = 0.f, q = 0.f;
        if (m[i] > 0.5f) {
            a = cw * sgn((x[i] - tp) * isp - (g[i] - tg) * isg) * isp + dLds * sgn(x[i] - tp) / (N + 1.f);
            if (i == mp) a += dLdt;
            if (alpha > 0.f) {
                const int r = i / W, c = i - r * W;
                float m0;
                const float d0 = dval(i, m0);
                for (int k = 0; k < scales; k++) {
                    const int s = 1 << k;
                    if ((r & (s - 1)) || (c & (s - 1))) break;          // not on the coarser grids either
                    const float Mk = st[S_M0 + k], wk = alpha * G / (B * (Mk != 0.f ? Mk : 1.f));
                    float m1, acc = 0.f;
                    if (c + s < W) { const float d1 = dval(i + s, m1); acc -= sgn(d1 - d0) * m1; }
                    if (c - s >= 0) { const float d1 = dval(i - s, m1); acc += sgn(d0 - d1) * m1; }
                    if (r + s < H) { const float d1 = dval(i + s * W, m1); acc -= sgn(d1 - d0) * m1; }
                    if (r - s >= 0) { const float d1 = dval(i - s * W, m1); acc += sgn(d0 - d1) * m1; }
                    q += wk * acc;
                }
                Ga += q * pval(i);
                Gc += q;
            }
        }
        out[i] = a;                      // part A now; part B is added below once Ga / Gc are known
    }
    if (alpha <= 0.f) return;
    Ga = block_sum(Ga, lds); Gc = block_sum(Gc, lds);
    // derivatives of the 2x2 solve (scale, shift) w.r.t. a00, a01, b0 (a11 and b1 do not depend on the prediction)
    float k00 = 0.f, k01 = 0.f, kb0 = 0.f;
    if (det_ok) {
        const float a00 = st[S_A00], a01 = st[S_A01], a11 = st[S_A11], b0 = st[S_B0], b1 = st[S_B1];
        const float D = a00 * a11 - a01 * a01 + 1e-6f;
        k00 = Ga * (-sc * a11 / D) + Gc * ((b1 - sh * a11) / D);
        k01 = Ga * ((-b1 + 2.f * sc * a01) / D) + Gc * ((-b0 + 2.f * sh * a01) / D);
        kb0 = Ga * (a11 / D) + Gc * (-a01 / D);
    }
    for (int i = threadIdx.x; i < n; i += BLOCK)
        if (m[i] > 0.5f) {
            // recompute q (cheap) instead of storing it
            const int r = i / W, c = i - r * W;
            float m0, q = 0.f;
            const float d0 = dval(i, m0);
            for (int k = 0; k < scales; k++) {
                const int s = 1 << k;
                if ((r & (s - 1)) || (c & (s - 1))) break;
                const float Mk = st[S_M0 + k], wk = alpha * G / (B * (Mk != 0.f ? Mk : 1.f));
                float m1, acc = 0.f;
                if (c + s < W) { const float d1 = dval(i + s, m1); acc -= sgn(d1 - d0) * m1; }
                if (c - s >= 0) { const float d1 = dval(i - s, m1); acc += sgn(d0 - d1) * m1; }
                if (r + s < H) { const float d1 = dval(i + s * W, m1); acc -= sgn(d1 - d0) * m1; }
                if (r - s >= 0) { const float d1 = dval(i - s * W, m1); acc += sgn(d0 - d1) * m1; }
                q += wk * acc;
            }
            const float p = pval(i), t = tval(i);
            const float dLdp = sc * q + 2.f * p * k00 + k01 + t * kb0;
            out[i] += inverse ? -p * p * dLdp : dLdp;
        }
}

// ---- Loss.intr_loss: sum(mask * |seen_pred - seen_gt|^2) / (sum(mask) + 1e-8) over the batch ----
__global__ __launch_bounds__(BLOCK) void intr_loss_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ mask, size_t n,
                                                          float *__restrict__ out /* [2]: loss, mask sum */) {
    __shared__ double ldsd[BLOCK / 64];
    double s = 0, ms = 0;
    for (size_t i = threadIdx.x; i < n; i += BLOCK) {
        const float dx = a[3 * i] - b[3 * i], dy = a[3 * i + 1] - b[3 * i + 1], dz = a[3 * i + 2] - b[3 * i + 2];
        s += (double)((dx * dx + dy * dy + dz * dz) * mask[i]);
        ms += mask[i];
    }
    s = block_sum_d(s, ldsd);
    ms = block_sum_d(ms, ldsd);
    if (threadIdx.x == 0) { out[0] = (float)(s / (ms + 1.e-8)); out[1] = (float)ms; }
}
__global__ __launch_bounds__(256) void intr_loss_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                            const float *__restrict__ mask, size_t n,
                                                            const float *__restrict__ fwd, const float *__restrict__ grad_loss,
                                                            float *__restrict__ da) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float w = 2.f * mask[i] * (*grad_loss) / (fwd[1] + 1.e-8f);
#pragma unroll
    for (int c = 0; c < 3; c++) da[3 * i + c] = w * (a[3 * i + c] - b[3 * i + c]);
}

}  // namespace

#define ZS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            zs::set_err(__VA_ARGS__);    \
            return 0;                    \
        }                                \
    } while (0)

// MidasLoss.erode_mask (model/depth/midas_loss.py:153-162): a pixel stays valid iff every mask value
// of its pool x pool block (max_pool2d, stride = kernel, floor) equals 1; pixels map to blocks the
// way F.interpolate(mode='nearest') maps them back: src = min(floor(dst * (float)(in / out)), in - 1).
__global__ __launch_bounds__(256) void erode_mask_kernel(const float *__restrict__ mask, int batch, int H, int W,
                                                         int pool, float *__restrict__ out) {
    const int Hp = H / pool, Wp = W / pool;
    const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
    const size_t n = (size_t)batch * H * W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int x = (int)(e % W), y = (int)((e / W) % H);
        const size_t b = e / ((size_t)W * H);
        const int by = min((int)floorf(y * sy), Hp - 1), bx = min((int)floorf(x * sx), Wp - 1);
        const float *M = mask + b * H * W + (size_t)by * pool * W + bx * pool;
        bool ok = true;
        for (int i = 0; i < pool; i++)
            for (int j = 0; j < pool; j++) ok = ok && (1.0f - M[(size_t)i * W + j] == 0.0f);
        out[e] = ok ? 1.0f : 0.0f;
    }
}

extern "C" int zs_erode_mask(const float *mask, int batch, int H, int W, int pool, float *out, void *stream) {
    ZS_REQUIRE(batch >= 0 && H > 0 && W > 0 && pool > 0 && H >= pool && W >= pool,
               "zs_erode_mask: bad arguments (B=%d H=%d W=%d pool=%d; the map must hold one pool window)", batch, H, W,
               pool);
    if (batch == 0) return 1;
    ZS_REQUIRE(mask && out, "zs_erode_mask: null pointer");
    const size_t n = (size_t)batch * H * W;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(erode_mask_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), mask, batch, H,
                       W, pool, out);
    return zs::check_launch("zs_erode_mask") ? 1 : 0;
}

extern "C" size_t zs_midas_loss_workspace_bytes(int batch) { return ((size_t)batch * S_STRIDE + 4) * sizeof(float); }

extern "C" int zs_midas_loss(const float *prediction, const float *target, const float *mask, int batch, int H, int W,
                             float alpha, int scales, int inverse_depth, float *loss, void *workspace, void *stream) {
    ZS_REQUIRE(batch > 0 && H > 0 && W > 0 && scales >= 0 && scales <= MAX_SCALES,
               "zs_midas_loss: bad arguments (B=%d H=%d W=%d scales=%d <= %d)", batch, H, W, scales, MAX_SCALES);
    ZS_REQUIRE(prediction && target && mask && loss && workspace, "zs_midas_loss: null pointer");
    float *stats = static_cast<float *>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(midas_fwd_kernel, dim3(batch), dim3(BLOCK), 0, st, prediction, target, mask, H, W, alpha, scales,
                       inverse_depth ? 1 : 0, stats);
    hipLaunchKernelGGL(midas_finish_kernel, dim3(1), dim3(64), 0, st, stats, batch, alpha, loss);
    return zs::check_launch("zs_midas_loss") ? 1 : 0;
}

extern "C" int zs_midas_loss_bwd(const float *prediction, const float *target, const float *mask, int batch, int H, int W,
                                 float alpha, int scales, int inverse_depth, const void *workspace, const float *grad_loss,
                                 float *dprediction, void *stream) {
    ZS_REQUIRE(batch > 0 && H > 0 && W > 0 && scales >= 0 && scales <= MAX_SCALES, "zs_midas_loss_bwd: bad arguments");
    ZS_REQUIRE(prediction && target && mask && workspace && grad_loss && dprediction, "zs_midas_loss_bwd: null pointer");
    hipLaunchKernelGGL(midas_bwd_kernel, dim3(batch), dim3(BLOCK), 0, static_cast<hipStream_t>(stream), prediction, target,
                       mask, batch, H, W, alpha, scales, inverse_depth ? 1 : 0, static_cast<const float *>(workspace),
                       grad_loss, dprediction);
    return zs::check_launch("zs_midas_loss_bwd") ? 1 : 0;
}

extern "C" int zs_intr_loss(const float *seen_pred, const float *seen_gt, const float *mask, size_t n, float *out2,
                            void *stream) {
    ZS_REQUIRE(n > 0 && seen_pred && seen_gt && mask && out2, "zs_intr_loss: bad arguments");
    hipLaunchKernelGGL(intr_loss_kernel, dim3(1), dim3(BLOCK), 0, static_cast<hipStream_t>(stream), seen_pred, seen_gt, mask,
                       n, out2);
    return zs::check_launch("zs_intr_loss") ? 1 : 0;
}

extern "C" int zs_intr_loss_bwd(const float *seen_pred, const float *seen_gt, const float *mask, size_t n,
                                const float *out2, const float *grad_loss, float *dseen_pred, void *stream) {
    ZS_REQUIRE(n > 0 && seen_pred && seen_gt && mask && out2 && grad_loss && dseen_pred, "zs_intr_loss_bwd: bad arguments");
    hipLaunchKernelGGL(intr_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), seen_pred, seen_gt, mask, n, out2, grad_loss, dseen_pred);
    return zs::check_launch("zs_intr_loss_bwd") ? 1 : 0;
}
