// Backward of the seen-surface geometry front-end (csrc/seen_surface.hip), i.e. of
//   unproj_depth -> valid_norm_fac -> (p - mean) / scale, invalid := 0 -> interpolate_coordmap
// (model/compute_graph/graph_shape.py:131-144; utils/camera.py:52-108; utils/util.py:336-345) as
// torch.autograd differentiates it in the reference when the depth model is trained
// (options/shape.yaml:93 fix_dpt false): gradients reach the depth map AND the intrinsics, through
// the masked mean and through the max-radius scale (sub-gradient at the arg-max pixel).
//   zs_seen_surface_bwd   : d_seen [B][HW][3] and/or d_coord [B][3][H][W] (same-size resample,
//                           the ResNet coordinate encoder's dsp = 1) -> d_depth [B][HW], d_intr [B][9]
//   zs_intr_param2mtx_bwd : d_intr [B][9] -> d_params [B][3]     (graph_shape.py:89-113)
// One 1024-lane workgroup per image, fixed-order reductions.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int BLOCK = 1024;

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

__device__ __forceinline__ void inverse3x3(const float *__restrict__ K, float *Ki) {
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double inv = 1.0 / (a * A + b * B + c * C);
    Ki[0] = (float)(A * inv); Ki[1] = (float)(-(b * i - c * h) * inv); Ki[2] = (float)((b * f - c * e) * inv);
    Ki[3] = (float)(B * inv); Ki[4] = (float)((a * i - c * g) * inv);  Ki[5] = (float)(-(a * f - c * d) * inv);
    Ki[6] = (float)(C * inv); Ki[7] = (float)(-(a * h - b * g) * inv); Ki[8] = (float)((a * e - b * d) * inv);
}

__global__ __launch_bounds__(BLOCK) void seen_surface_bwd_kernel(
    const float *__restrict__ depth, const float *__restrict__ intr, const float *__restrict__ mask,
    const float *__restrict__ mean, const float *__restrict__ scale, const float *__restrict__ d_seen,
    const float *__restrict__ d_coord, int H, int W, float *__restrict__ d_depth, float *__restrict__ d_intr) {
    __shared__ float lds[BLOCK / 64];
    __shared__ float best_v[BLOCK / 64];
    __shared__ int best_i[BLOCK / 64];
    const int b = blockIdx.x, tid = threadIdx.x, n = H * W;
    const float *D = depth + (size_t)b * n, *M = mask + (size_t)b * n;
    const float *GS = d_seen ? d_seen + (size_t)b * n * 3 : nullptr, *GC = d_coord ? d_coord + (size_t)b * 3 * n : nullptr;
    float Ki[9];
    inverse3x3(intr + (size_t)b * 9, Ki);
    const float mx = mean[b * 3], my = mean[b * 3 + 1], mz = mean[b * 3 + 2], sc = scale[b];
    const float inv_eps = 1.0f / (1.0f + 1.e-6f);          // interpolate_coordmap at equal size: v * 1 / (1 + 1e-6)
    auto ray = [&](int i, float &rx, float &ry, float &rz) {
        const float x = (float)(i % W), y = (float)(i / W);
        rx = Ki[0] * x + Ki[1] * y + Ki[2];
        ry = Ki[3] * x + Ki[4] * y + Ki[5];
        rz = Ki[6] * x + Ki[7] * y + Ki[8];
    };
    auto grad_q = [&](int i, float &gx, float &gy, float &gz) {      // dL/d(normalised point) of a valid pixel
        gx = gy = gz = 0.f;
        if (GS) { gx += GS[i * 3]; gy += GS[i * 3 + 1]; gz += GS[i * 3 + 2]; }
        if (GC) { gx += GC[i] * inv_eps; gy += GC[n + i] * inv_eps; gz += GC[2 * n + i] * inv_eps; }
    };
    // pass 1: A = sum g.q', G = sum g, count, arg-max radius
    float A = 0.f, Gx = 0.f, Gy = 0.f, Gz = 0.f, cnt = 0.f, rbest = -INFINITY;
    int ibest = 0x7fffffff;
    for (int i = tid; i < n; i += BLOCK)
        if (M[i] > 0.5f) {
            float rx, ry, rz, gx, gy, gz;
            ray(i, rx, ry, rz);
            grad_q(i, gx, gy, gz);
            const float qx = rx * D[i] - mx, qy = ry * D[i] - my, qz = rz * D[i] - mz;
            A += gx * qx + gy * qy + gz * qz;
            Gx += gx; Gy += gy; Gz += gz;
            cnt += 1.f;
            const float r = sqrtf(qx * qx + qy * qy + qz * qz);
            if (r > rbest) { rbest = r; ibest = i; }
        }
    A = block_sum(A, lds);
    Gx = block_sum(Gx, lds); Gy = block_sum(Gy, lds); Gz = block_sum(Gz, lds);
    cnt = block_sum(cnt, lds);
    // arg-max (largest radius, lowest index among equals)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(rbest, o, 64);
        const int oi = __shfl_xor(ibest, o, 64);
        if (ov > rbest || (ov == rbest && oi < ibest)) { rbest = ov; ibest = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { best_v[tid >> 6] = rbest; best_i[tid >> 6] = ibest; }
    __syncthreads();
    rbest = best_v[0]; ibest = best_i[0];
    for (int w = 1; w < BLOCK / 64; w++)
        if (best_v[w] > rbest || (best_v[w] == rbest && best_i[w] < ibest)) { rbest = best_v[w]; ibest = best_i[w]; }
    // dL/dscale = -sum g.q' / scale^2 ; it enters only through the arg-max pixel's q' / |q'|
    const float dscale = -A / (sc * sc);
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (cnt > 0.f && rbest > 0.f) {
        float rx, ry, rz;
        ray(ibest, rx, ry, rz);
        ux = (rx * D[ibest] - mx) / rbest; uy = (ry * D[ibest] - my) / rbest; uz = (rz * D[ibest] - mz) / rbest;
    }
    // T = sum_j dL/dq'_j (what the mean subtraction sends back to every valid pixel, / count)
    const float Tx = (Gx / sc + dscale * ux) / cnt, Ty = (Gy / sc + dscale * uy) / cnt, Tz = (Gz / sc + dscale * uz) / cnt;
    // pass 2: per-pixel dL/dp -> d_depth, and the 3x3 gradient of K^-1
    float g9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += BLOCK) {
        float dd = 0.f;
        if (M[i] > 0.5f) {
            float rx, ry, rz, gx, gy, gz;
            ray(i, rx, ry, rz);
            grad_q(i, gx, gy, gz);
            float px = gx / sc - Tx, py = gy / sc - Ty, pz = gz / sc - Tz;
            if (i == ibest) { px += dscale * ux; py += dscale * uy; pz += dscale * uz; }
            dd = px * rx + py * ry + pz * rz;
            const float x = (float)(i % W), y = (float)(i / W), d = D[i];
            g9[0] += px * d * x; g9[1] += px * d * y; g9[2] += px * d;
            g9[3] += py * d * x; g9[4] += py * d * y; g9[5] += py * d;
            g9[6] += pz * d * x; g9[7] += pz * d * y; g9[8] += pz * d;
        }
        d_depth[(size_t)b * n + i] = dd;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) g9[k] = block_sum(g9[k], lds);
    if (tid == 0) {
        // d(K^-1) = -K^-1 dK K^-1  =>  dL/dK = -K^-T G K^-T
        float t[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                float s = 0.f;
                for (int k = 0; k < 3; k++) s += Ki[k * 3 + r] * g9[k * 3 + c];      // K^-T G
                t[r * 3 + c] = s;
            }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                float s = 0.f;
                for (int k = 0; k < 3; k++) s += t[r * 3 + k] * Ki[c * 3 + k];      // (.) K^-T
                d_intr[(size_t)b * 9 + r * 3 + c] = -s;
            }
    }
}

__global__ __launch_bounds__(256) void intr_param2mtx_bwd_kernel(const float *__restrict__ params,
                                                                 const float *__restrict__ d_intr, int batch, float H,
                                                                 float W, float *__restrict__ d_params) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const float f = 1.3875f, ln4 = 1.38629436111989061883f;
    const float t0 = tanhf(params[b * 3]), t1 = tanhf(params[b * 3 + 1]), t2 = tanhf(params[b * 3 + 2]);
    const float *g = d_intr + (size_t)b * 9;
    const float s = powf(4.0f, t0);
    d_params[b * 3] = (g[0] * f * W + g[4] * f * H) * s * ln4 * (1.f - t0 * t0);
    d_params[b * 3 + 1] = g[2] * (W / 2) * (1.f - t1 * t1);
    d_params[b * 3 + 2] = g[5] * (H / 2) * (1.f - t2 * t2);
}

// graph_shape.py:163-173: GT query points -> camera frame -> the seen surface's normalised frame
__global__ __launch_bounds__(256) void transform_points_kernel(const float *__restrict__ pts, const float *__restrict__ pose,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ scale, float *__restrict__ out,
                                                               int n) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *P = pose + (size_t)b * 12, *p = pts + ((size_t)b * n + i) * 3;
    float *o = out + ((size_t)b * n + i) * 3;
    const float s = scale[b];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float cam = P[r * 4] * p[0] + P[r * 4 + 1] * p[1] + P[r * 4 + 2] * p[2] + P[r * 4 + 3];
        o[r] = (cam - mean[b * 3 + r]) / s;
    }
}

}  // namespace

extern "C" int zs_transform_points(const float *points, const float *pose, const float *mean, const float *scale,
                                   float *out, int batch, int n, void *stream) {
    if (batch < 0 || batch > 65535 || n <= 0) { zs::set_err("zs_transform_points: bad size (batch=%d n=%d)", batch, n); return 0; }
    if (batch == 0) return 1;
    if (!points || !pose || !mean || !scale || !out) { zs::set_err("zs_transform_points: null pointer"); return 0; }
    hipLaunchKernelGGL(transform_points_kernel, dim3((n + 255) / 256, batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), points, pose, mean, scale, out, n);
    return zs::check_launch("zs_transform_points") ? 1 : 0;
}

extern "C" int zs_seen_surface_bwd(const float *depth, const float *intr, const float *mask, const float *mean,
                                   const float *scale, const float *d_seen_points, const float *d_coord_dsp, int batch,
                                   int H, int W, float *d_depth, float *d_intr, void *stream) {
    if (batch < 0 || H <= 0 || W <= 0 || (long long)H * W > (1 << 28)) {
        zs::set_err("zs_seen_surface_bwd: bad size (batch=%d H=%d W=%d)", batch, H, W);
        return 0;
    }
    if (batch == 0) return 1;
    if (!depth || !intr || !mask || !mean || !scale || !d_depth || !d_intr || (!d_seen_points && !d_coord_dsp)) {
        zs::set_err("zs_seen_surface_bwd: null pointer");
        return 0;
    }
    hipLaunchKernelGGL(seen_surface_bwd_kernel, dim3(batch), dim3(BLOCK), 0, static_cast<hipStream_t>(stream), depth,
                       intr, mask, mean, scale, d_seen_points, d_coord_dsp, H, W, d_depth, d_intr);
    return zs::check_launch("zs_seen_surface_bwd") ? 1 : 0;
}

extern "C" int zs_intr_param2mtx_bwd(const float *params, const float *d_intr, int batch, int H, int W, float *d_params,
                                     void *stream) {
    if (batch < 0 || H <= 0 || W <= 0) { zs::set_err("zs_intr_param2mtx_bwd: bad size"); return 0; }
    if (batch == 0) return 1;
    if (!params || !d_intr || !d_params) { zs::set_err("zs_intr_param2mtx_bwd: null pointer"); return 0; }
    hipLaunchKernelGGL(intr_param2mtx_bwd_kernel, dim3((batch + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), params, d_intr, batch, (float)H, (float)W, d_params);
    return zs::check_launch("zs_intr_param2mtx_bwd") ? 1 : 0;
}
