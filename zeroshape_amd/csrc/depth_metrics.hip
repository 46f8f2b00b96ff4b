// Scale/shift-aligned depth metrics (utils/eval_depth.py:5-116, DepthMetric.compute_metrics).
//
// The reference spends ~60 masked-index launches per call; this is one launch, one 1024-lane
// workgroup per image, two passes over 12 B/pixel: (1) the five sums of the 2x2 least-squares
// system (:12-34), (2) align, optional cap, invert, accumulate the error sums, write the aligned
// depth.  Sums are accumulated in double (the reference sums in fp32; the tests bound the
// difference).
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int BLOCK = 1024;
constexpr int MAX_THR = 8;
struct Thresholds { float v[MAX_THR]; int n; };

__device__ __forceinline__ double block_sum(double v, double *lds) {
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

__global__ __launch_bounds__(BLOCK) void depth_metrics_kernel(
    const float *__restrict__ prediction, const float *__restrict__ target, const float *__restrict__ mask, int n,
    int flags, float depth_cap, Thresholds thr, float *__restrict__ metrics,
    float *__restrict__ prediction_depth, float *__restrict__ scale_shift) {
    __shared__ double lds[BLOCK / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *P = prediction + (size_t)b * n, *T = target + (size_t)b * n, *M = mask + (size_t)b * n;

    // disparities of the valid pixels (:64-75)
    const bool is_disparity = flags & ZS_DEPTH_PRED_IS_DISPARITY, solve_only = flags & ZS_DEPTH_SOLVE_ONLY;
    auto pred_disp = [&](int i) { return (is_disparity || solve_only) ? P[i] : 1.0f / (P[i] + 1.e-6f); };
    double a00 = 0, a01 = 0, a11 = 0, b0 = 0, b1 = 0;
    for (int i = tid; i < n; i += BLOCK)
        if (M[i] > 0.5f) {
            const float p = pred_disp(i), t = solve_only ? T[i] : 1.0f / T[i];
            a00 += (double)(p * p); a01 += (double)p; a11 += 1.0;
            b0 += (double)(p * t); b1 += (double)t;
        }
    // the reference's sums are fp32 tensors: round each once before the 2x2 solve (:27-34)
    const float A00 = (float)block_sum(a00, lds), A01 = (float)block_sum(a01, lds), A11 = (float)block_sum(a11, lds),
                B0 = (float)block_sum(b0, lds), B1 = (float)block_sum(b1, lds);
    const float det = A00 * A11 - A01 * A01;
    float scale = 0.f, shift = 0.f;
    if (det > 0.f) {
        scale = (A11 * B0 - A01 * B1) / det;
        shift = (-A01 * B0 + A00 * B1) / det;
    }
    if (solve_only) {          // compute_scale_and_shift (:11-34) on its own
        if (tid == 0 && scale_shift) { scale_shift[b * 2] = scale; scale_shift[b * 2 + 1] = shift; }
        return;
    }
    const float disparity_cap = depth_cap > 0.f ? 1.0f / depth_cap : 0.f;

    double se = 0, ae = 0, re = 0, cnt[MAX_THR] = {0};
    float *O = prediction_depth ? prediction_depth + (size_t)b * n : nullptr;
    for (int i = tid; i < n; i += BLOCK) {
        const bool valid = M[i] > 0.5f;
        float aligned = scale * (valid ? pred_disp(i) : 0.f) + shift;          // :77
        if (depth_cap > 0.f && aligned < disparity_cap) aligned = disparity_cap;   // :79-81
        const float d = 1.0f / aligned;                                           // :83
        if (O) O[i] = d;
        if (valid) {
            const float t = T[i], diff = d - t;
            const float ratio = fmaxf(d / t, t / d);                              // :88-91
#pragma unroll
            for (int k = 0; k < MAX_THR; k++)
                if (k < thr.n && ratio > thr.v[k]) cnt[k] += 1.0;
            se += (double)(diff * diff);
            ae += (double)fabsf(diff);
            re += (double)(fabsf(diff) / t);
        }
    }
    float *out = metrics + (size_t)b * (thr.n + 3);
    const float N = A11;
    for (int k = 0; k < thr.n; k++) {
        const float c = (float)block_sum(cnt[k], lds);
        if (tid == 0) out[k] = c / N;
    }
    const float SE = (float)block_sum(se, lds), AE = (float)block_sum(ae, lds), RE = (float)block_sum(re, lds);
    if (tid == 0) {
        out[thr.n] = sqrtf(SE / N);
        out[thr.n + 1] = AE / N;
        out[thr.n + 2] = RE / N;
        if (scale_shift) { scale_shift[b * 2] = scale; scale_shift[b * 2 + 1] = shift; }
    }
}

}  // namespace

extern "C" int zs_depth_metrics(const float *prediction, const float *target, const float *mask, int batch, int n,
                                int flags, float depth_cap, const float *thresholds,
                                int n_thresholds, float *metrics, float *prediction_depth, float *scale_shift,
                                void *stream) {
    if (batch < 0 || n <= 0 || n_thresholds < 0 || n_thresholds > MAX_THR) {
        zs::set_err("zs_depth_metrics: bad size (batch=%d n=%d n_thresholds=%d, max %d)", batch, n, n_thresholds,
                    MAX_THR);
        return 0;
    }
    if (batch == 0) return 1;
    const bool solve_only = flags & ZS_DEPTH_SOLVE_ONLY;
    if (!prediction || !target || !mask || (!solve_only && !metrics) || (solve_only && !scale_shift) ||
        (n_thresholds > 0 && !thresholds)) {
        zs::set_err("zs_depth_metrics: null pointer");
        return 0;
    }
    Thresholds thr;
    thr.n = n_thresholds;
    for (int k = 0; k < MAX_THR; k++) thr.v[k] = k < n_thresholds ? thresholds[k] : 0.f;
    hipLaunchKernelGGL(depth_metrics_kernel, dim3(batch), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       prediction, target, mask, n, flags, depth_cap, thr, metrics,
                       prediction_depth, scale_shift);
    return zs::check_launch("zs_depth_metrics") ? 1 : 0;
}
