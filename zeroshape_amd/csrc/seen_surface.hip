// Seen-surface geometry front-end: depth map + intrinsics -> normalised view-centric point map
// -> (optionally down-sampled) coordinate map for the coordinate encoder.
//
// Replaces, for one forward of the shape graph (model/compute_graph/graph_shape.py:131-144):
//   unproj_depth          utils/camera.py:88-108   K^-1 [x,y,1]^T * depth
//   valid_norm_fac        utils/camera.py:52-78    per-sample masked mean and max radius
//                                                  (a Python loop over B with boolean gathers)
//   (p - mean) / scale, invalid pixels := 0        graph_shape.py:139-141
//   interpolate_coordmap  utils/util.py:336-345    masked bilinear resample (align_corners=False)
//   intr_param2mtx        graph_shape.py:89-113    3 raw parameters -> 3x3 intrinsics
//
// HBM-bound and tiny (20 B per pixel, 1 MB per 224^2 image): what the reference pays for here is
// ~40 small launches and B device syncs, so the fused entry point does the whole chain in ONE
// launch, one 1024-lane workgroup per image, recomputing the unprojection in each of its three
// passes instead of round-tripping it through memory.  The stand-alone entry points mirror the
// reference's individual functions for callers that use them separately.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int BLOCK = 1024;

__device__ __forceinline__ float block_reduce(float v, float *lds, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float y = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, y) : v + y;
    }
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = lds[0];
    for (int w = 1; w < nw; w++) r = is_max ? fmaxf(r, lds[w]) : r + lds[w];
    return r;
}

// inverse of a 3x3 matrix by the adjugate, in double, rounded once (torch.linalg.inv(...).float(),
// utils/camera.py:98, is an fp32 LU: both are within a few ulp of the exact inverse)
__device__ __forceinline__ void inverse3x3(const float *__restrict__ K, float *Ki) {
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double inv = 1.0 / (a * A + b * B + c * C);
    Ki[0] = (float)(A * inv); Ki[1] = (float)(-(b * i - c * h) * inv); Ki[2] = (float)((b * f - c * e) * inv);
    Ki[3] = (float)(B * inv); Ki[4] = (float)((a * i - c * g) * inv);  Ki[5] = (float)(-(a * f - c * d) * inv);
    Ki[6] = (float)(C * inv); Ki[7] = (float)(-(a * h - b * g) * inv); Ki[8] = (float)((a * e - b * d) * inv);
}

struct Point { float x, y, z; };

// one pixel of unproj_depth: ray = K^-1 [x, y, 1]^T (a K=3 matmul row, utils/camera.py:104), * depth
__device__ __forceinline__ Point unproject(const float *Ki, int px, int py, float depth) {
    const float x = (float)px, y = (float)py;
    Point p;
    p.x = (Ki[0] * x + Ki[1] * y + Ki[2]) * depth;
    p.y = (Ki[3] * x + Ki[4] * y + Ki[5]) * depth;
    p.z = (Ki[6] * x + Ki[7] * y + Ki[8]) * depth;
    return p;
}

// ---- torch upsample_bilinear2d, align_corners=False (utils/util.py:340-341) ----
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap bilinear_tap(int dst, float scale, int in_size) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Tap t;
    t.i0 = min((int)src, in_size - 1);
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.0f - t.l1;
    return t;
}
__device__ __forceinline__ float bilinear_mix(const Tap &ty, const Tap &tx, float v00, float v01, float v10,
                                              float v11) {
    return ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
}
// interpolate_coordmap's epilogue (utils/util.py:342-344)
__device__ __forceinline__ float masked_out(float v, float m, float bg) {
    const float mb = m > 0.5f ? 1.0f : 0.0f;
    return (v / (m + 1.e-6f)) * mb + bg * (1.0f - mb);
}

__global__ __launch_bounds__(256) void intr_param2mtx_kernel(const float *__restrict__ params, int batch, float H,
                                                             float W, float *__restrict__ intr) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const float f = 1.3875f;
    const float scale_f = powf(4.0f, tanhf(params[b * 3]));          // [1/4, 4]
    const float shift_cx = tanhf(params[b * 3 + 1]) * W / 2, shift_cy = tanhf(params[b * 3 + 2]) * H / 2;
    float *K = intr + (size_t)b * 9;
    K[0] = f * W * scale_f; K[1] = 0.f;             K[2] = W / 2 + shift_cx;
    K[3] = 0.f;             K[4] = f * H * scale_f; K[5] = H / 2 + shift_cy;
    K[6] = 0.f;             K[7] = 0.f;             K[8] = 1.f;
}

__global__ __launch_bounds__(256) void unproj_kernel(const float *__restrict__ depth, const float *__restrict__ intr,
                                                     int H, int W, float *__restrict__ points) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    float Ki[9];
    inverse3x3(intr + (size_t)b * 9, Ki);
    if (i >= H * W) return;
    const Point p = unproject(Ki, i % W, i / W, depth[(size_t)b * H * W + i]);
    float *o = points + ((size_t)b * H * W + i) * 3;
    o[0] = p.x; o[1] = p.y; o[2] = p.z;
}

// valid_norm_fac on arbitrary points [B][n][3] with a byte mask [B][n]
__global__ __launch_bounds__(BLOCK) void norm_fac_kernel(const float *__restrict__ points,
                                                         const uint8_t *__restrict__ mask, int n,
                                                         float *__restrict__ mean, float *__restrict__ max_dist) {
    __shared__ float lds[BLOCK / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *P = points + (size_t)b * n * 3;
    const uint8_t *M = mask + (size_t)b * n;
    float sx = 0.f, sy = 0.f, sz = 0.f, cnt = 0.f;
    for (int i = tid; i < n; i += BLOCK)
        if (M[i]) { sx += P[i * 3]; sy += P[i * 3 + 1]; sz += P[i * 3 + 2]; cnt += 1.f; }
    cnt = block_reduce(cnt, lds, false);
    const float mx = block_reduce(sx, lds, false) / cnt, my = block_reduce(sy, lds, false) / cnt,
                mz = block_reduce(sz, lds, false) / cnt;
    float r = -INFINITY;
    for (int i = tid; i < n; i += BLOCK)
        if (M[i]) {
            const float dx = P[i * 3] - mx, dy = P[i * 3 + 1] - my, dz = P[i * 3 + 2] - mz;
            r = fmaxf(r, sqrtf(dx * dx + dy * dy + dz * dz));
        }
    r = block_reduce(r, lds, true);
    if (tid == 0) {
        mean[b * 3] = mx; mean[b * 3 + 1] = my; mean[b * 3 + 2] = mz;
        max_dist[b] = cnt > 0.f ? r : NAN;     // the reference raises on an empty selection
    }
}

// interpolate_coordmap / interpolate_depth on an arbitrary [B][C][H][W] map
__global__ __launch_bounds__(256) void masked_resample_kernel(const float *__restrict__ map,
                                                              const float *__restrict__ mask, int C, int H, int W,
                                                              int Ho, int Wo, float bg, float *__restrict__ out,
                                                              float *__restrict__ mask_out) {
    const int b = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
    if (o >= Ho * Wo) return;
    const Tap ty = bilinear_tap(o / Wo, (float)H / (float)Ho, H), tx = bilinear_tap(o % Wo, (float)W / (float)Wo, W);
    const float *M = mask + (size_t)b * H * W;
    const int a00 = ty.i0 * W + tx.i0, a01 = ty.i0 * W + tx.i1, a10 = ty.i1 * W + tx.i0, a11 = ty.i1 * W + tx.i1;
    const float m00 = M[a00] > 0.5f ? 1.f : 0.f, m01 = M[a01] > 0.5f ? 1.f : 0.f, m10 = M[a10] > 0.5f ? 1.f : 0.f,
                m11 = M[a11] > 0.5f ? 1.f : 0.f;
    const float m = bilinear_mix(ty, tx, m00, m01, m10, m11);
    for (int c = 0; c < C; c++) {
        const float *V = map + ((size_t)b * C + c) * H * W;
        const float v = bilinear_mix(ty, tx, V[a00] * m00, V[a01] * m01, V[a10] * m10, V[a11] * m11);
        out[((size_t)b * C + c) * Ho * Wo + o] = masked_out(v, m, bg);
    }
    mask_out[(size_t)b * Ho * Wo + o] = m > 0.5f ? 1.f : 0.f;
}

// the fused chain, one workgroup per image
__global__ __launch_bounds__(BLOCK) void seen_surface_kernel(
    const float *__restrict__ depth, const float *__restrict__ intr, const float *__restrict__ mask, int H, int W,
    int Ho, int Wo, float *__restrict__ seen, float *__restrict__ mean, float *__restrict__ scale,
    float *__restrict__ coord_dsp, float *__restrict__ mask_dsp) {
    __shared__ float lds[BLOCK / 64];
    const int b = blockIdx.x, tid = threadIdx.x, n = H * W;
    const float *D = depth + (size_t)b * n, *M = mask + (size_t)b * n;
    float Ki[9];
    inverse3x3(intr + (size_t)b * 9, Ki);

    // pixel i = tid + k BLOCK as (x, y), advanced without a division per pixel
    const int step_x = BLOCK % W, step_y = BLOCK / W, x0 = tid % W, y0 = tid / W;
    auto advance = [&](int &x, int &y) {
        x += step_x;
        y += step_y;
        if (x >= W) { x -= W; y++; }
    };
    float sx = 0.f, sy = 0.f, sz = 0.f, cnt = 0.f;
    for (int i = tid, x = x0, y = y0; i < n; i += BLOCK, advance(x, y))
        if (M[i] > 0.5f) {
            const Point p = unproject(Ki, x, y, D[i]);
            sx += p.x; sy += p.y; sz += p.z; cnt += 1.f;
        }
    cnt = block_reduce(cnt, lds, false);
    const float mx = block_reduce(sx, lds, false) / cnt, my = block_reduce(sy, lds, false) / cnt,
                mz = block_reduce(sz, lds, false) / cnt;
    float r = -INFINITY;
    for (int i = tid, x = x0, y = y0; i < n; i += BLOCK, advance(x, y))
        if (M[i] > 0.5f) {
            const Point p = unproject(Ki, x, y, D[i]);
            const float dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
            r = fmaxf(r, sqrtf(dx * dx + dy * dy + dz * dz));
        }
    r = block_reduce(r, lds, true);
    if (!(cnt > 0.f)) r = NAN;
    if (tid == 0) {
        mean[b * 3] = mx; mean[b * 3 + 1] = my; mean[b * 3 + 2] = mz;
        scale[b] = r;
    }
    // normalised point of pixel i = (x, y) (0 where invalid) - the value written to `seen`
    auto normalised_xy = [&](int i, int x, int y) -> Point {
        Point q = {0.f, 0.f, 0.f};
        if (M[i] > 0.5f) {
            const Point p = unproject(Ki, x, y, D[i]);
            q.x = (p.x - mx) / r; q.y = (p.y - my) / r; q.z = (p.z - mz) / r;
        }
        return q;
    };
    auto normalised = [&](int i) -> Point { return normalised_xy(i, i % W, i / W); };
    float *S = seen + (size_t)b * n * 3;
    // Same-size coordinate map (arch.depth.dsp = 1, the default encoder): the bilinear taps are (1, 0) x (1, 0), so
    // interpolate_coordmap reduces to q / (1 + 1e-6) on valid pixels - written in this pass instead of a fourth one
    // that unprojects every pixel four more times.
    const bool same = coord_dsp && Ho == H && Wo == W;
    const float inv_eps = 1.0f + 1.e-6f;
    for (int i = tid, x = x0, y = y0; i < n; i += BLOCK, advance(x, y)) {
        const Point q = normalised_xy(i, x, y);
        S[i * 3] = q.x; S[i * 3 + 1] = q.y; S[i * 3 + 2] = q.z;
        if (same) {
            const bool valid = M[i] > 0.5f;
            float *O = coord_dsp + (size_t)b * 3 * n;
            O[i] = valid ? q.x / inv_eps : 0.f;
            O[n + i] = valid ? q.y / inv_eps : 0.f;
            O[2 * n + i] = valid ? q.z / inv_eps : 0.f;
            mask_dsp[(size_t)b * n + i] = valid ? 1.f : 0.f;
        }
    }
    if (!coord_dsp || same) return;
    const int no = Ho * Wo;
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    float *O = coord_dsp + (size_t)b * 3 * no;
    for (int o = tid; o < no; o += BLOCK) {
        const Tap ty = bilinear_tap(o / Wo, sh, H), tx = bilinear_tap(o % Wo, sw, W);
        const int a00 = ty.i0 * W + tx.i0, a01 = ty.i0 * W + tx.i1, a10 = ty.i1 * W + tx.i0, a11 = ty.i1 * W + tx.i1;
        const Point q00 = normalised(a00), q01 = normalised(a01), q10 = normalised(a10), q11 = normalised(a11);
        const float m = bilinear_mix(ty, tx, M[a00] > 0.5f ? 1.f : 0.f, M[a01] > 0.5f ? 1.f : 0.f,
                                     M[a10] > 0.5f ? 1.f : 0.f, M[a11] > 0.5f ? 1.f : 0.f);
        O[o] = masked_out(bilinear_mix(ty, tx, q00.x, q01.x, q10.x, q11.x), m, 0.f);
        O[no + o] = masked_out(bilinear_mix(ty, tx, q00.y, q01.y, q10.y, q11.y), m, 0.f);
        O[2 * no + o] = masked_out(bilinear_mix(ty, tx, q00.z, q01.z, q10.z, q11.z), m, 0.f);
        mask_dsp[(size_t)b * no + o] = m > 0.5f ? 1.f : 0.f;
    }
}

// ---- the same chain on SEEN_CH workgroups per image (three launches) ----
// One 1024-lane workgroup per image is 80 us at batch 1 (three dependent passes of 49 pixels per lane, and 1.4 MB of
// stores through one CU).  Here every image is cut into SEEN_CH pixel chunks: launch 1 = per-chunk masked sums, launch 2 =
// mean (the chunk sums in chunk order, the same value in every workgroup) + per-chunk max radius, launch 3 = scale (max of
// the chunk maxima) + normalise / resample of the chunk.  ws: [B][SEEN_CH][8] floats.
constexpr int SEEN_CH = 64, SEEN_T = 256;

__device__ __forceinline__ float block_reduce256(float v, float *lds, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float y = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, y) : v + y;
    }
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3])) : (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__device__ __forceinline__ void chunk_range(int n, int c, int &lo, int &hi) {
    const int per = (n + SEEN_CH - 1) / SEEN_CH;
    lo = c * per;
    hi = min(n, lo + per);
}

__global__ __launch_bounds__(SEEN_T) void seen_sums_kernel(const float *__restrict__ depth, const float *__restrict__ intr,
                                                           const float *__restrict__ mask, int H, int W, float *__restrict__ ws) {
    __shared__ float lds[4];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, n = H * W;
    const float *D = depth + (size_t)b * n, *M = mask + (size_t)b * n;
    float Ki[9];
    inverse3x3(intr + (size_t)b * 9, Ki);
    int lo, hi;
    chunk_range(n, c, lo, hi);
    float sx = 0.f, sy = 0.f, sz = 0.f, cnt = 0.f;
    for (int i = lo + tid; i < hi; i += SEEN_T)
        if (M[i] > 0.5f) {
            const Point p = unproject(Ki, i % W, i / W, D[i]);
            sx += p.x; sy += p.y; sz += p.z; cnt += 1.f;
        }
    sx = block_reduce256(sx, lds, false);
    sy = block_reduce256(sy, lds, false);
    sz = block_reduce256(sz, lds, false);
    cnt = block_reduce256(cnt, lds, false);
    if (tid == 0) {
        float *o = ws + ((size_t)b * SEEN_CH + c) * 8;
        o[0] = sx; o[1] = sy; o[2] = sz; o[3] = cnt;
    }
}

// mean of the image from the chunk sums, in chunk order
__device__ __forceinline__ void seen_mean(const float *__restrict__ ws, int b, float &mx, float &my, float &mz, float &cnt) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    cnt = 0.f;
    for (int c = 0; c < SEEN_CH; c++) {
        const float *o = ws + ((size_t)b * SEEN_CH + c) * 8;
        sx += o[0]; sy += o[1]; sz += o[2]; cnt += o[3];
    }
    mx = sx / cnt; my = sy / cnt; mz = sz / cnt;
}

__global__ __launch_bounds__(SEEN_T) void seen_radius_kernel(const float *__restrict__ depth, const float *__restrict__ intr,
                                                             const float *__restrict__ mask, int H, int W, float *__restrict__ ws) {
    __shared__ float lds[4];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, n = H * W;
    const float *D = depth + (size_t)b * n, *M = mask + (size_t)b * n;
    float Ki[9];
    inverse3x3(intr + (size_t)b * 9, Ki);
    float mx, my, mz, cnt;
    seen_mean(ws, b, mx, my, mz, cnt);
    int lo, hi;
    chunk_range(n, c, lo, hi);
    float r = -INFINITY;
    for (int i = lo + tid; i < hi; i += SEEN_T)
        if (M[i] > 0.5f) {
            const Point p = unproject(Ki, i % W, i / W, D[i]);
            const float dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
            r = fmaxf(r, sqrtf(dx * dx + dy * dy + dz * dz));
        }
    r = block_reduce256(r, lds, true);
    if (tid == 0) ws[((size_t)b * SEEN_CH + c) * 8 + 4] = r;
}

__global__ __launch_bounds__(SEEN_T) void seen_apply_kernel(
    const float *__restrict__ depth, const float *__restrict__ intr, const float *__restrict__ mask, int H, int W,
    int Ho, int Wo, const float *__restrict__ ws, float *__restrict__ seen, float *__restrict__ mean, float *__restrict__ scale,
    float *__restrict__ coord_dsp, float *__restrict__ mask_dsp) {
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, n = H * W;
    const float *D = depth + (size_t)b * n, *M = mask + (size_t)b * n;
    float Ki[9];
    inverse3x3(intr + (size_t)b * 9, Ki);
    float mx, my, mz, cnt;
    seen_mean(ws, b, mx, my, mz, cnt);
    float r = -INFINITY;
    for (int k = 0; k < SEEN_CH; k++) r = fmaxf(r, ws[((size_t)b * SEEN_CH + k) * 8 + 4]);
    if (!(cnt > 0.f)) r = NAN;
    if (c == 0 && tid == 0) {
        mean[b * 3] = mx; mean[b * 3 + 1] = my; mean[b * 3 + 2] = mz;
        scale[b] = r;
    }
    auto normalised = [&](int i) -> Point {
        Point q = {0.f, 0.f, 0.f};
        if (M[i] > 0.5f) {
            const Point p = unproject(Ki, i % W, i / W, D[i]);
            q.x = (p.x - mx) / r; q.y = (p.y - my) / r; q.z = (p.z - mz) / r;
        }
        return q;
    };
    float *S = seen + (size_t)b * n * 3;
    const bool same = coord_dsp && Ho == H && Wo == W;
    const float inv_eps = 1.0f + 1.e-6f;
    int lo, hi;
    chunk_range(n, c, lo, hi);
    for (int i = lo + tid; i < hi; i += SEEN_T) {
        const Point q = normalised(i);
        S[i * 3] = q.x; S[i * 3 + 1] = q.y; S[i * 3 + 2] = q.z;
        if (same) {
            const bool valid = M[i] > 0.5f;
            float *O = coord_dsp + (size_t)b * 3 * n;
            O[i] = valid ? q.x / inv_eps : 0.f;
            O[n + i] = valid ? q.y / inv_eps : 0.f;
            O[2 * n + i] = valid ? q.z / inv_eps : 0.f;
            mask_dsp[(size_t)b * n + i] = valid ? 1.f : 0.f;
        }
    }
    if (!coord_dsp || same) return;
    const int no = Ho * Wo;
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    float *O = coord_dsp + (size_t)b * 3 * no;
    chunk_range(no, c, lo, hi);
    for (int o = lo + tid; o < hi; o += SEEN_T) {
        const Tap ty = bilinear_tap(o / Wo, sh, H), tx = bilinear_tap(o % Wo, sw, W);
        const int a00 = ty.i0 * W + tx.i0, a01 = ty.i0 * W + tx.i1, a10 = ty.i1 * W + tx.i0, a11 = ty.i1 * W + tx.i1;
        const Point q00 = normalised(a00), q01 = normalised(a01), q10 = normalised(a10), q11 = normalised(a11);
        const float m = bilinear_mix(ty, tx, M[a00] > 0.5f ? 1.f : 0.f, M[a01] > 0.5f ? 1.f : 0.f,
                                     M[a10] > 0.5f ? 1.f : 0.f, M[a11] > 0.5f ? 1.f : 0.f);
        O[o] = masked_out(bilinear_mix(ty, tx, q00.x, q01.x, q10.x, q11.x), m, 0.f);
        O[no + o] = masked_out(bilinear_mix(ty, tx, q00.y, q01.y, q10.y, q11.y), m, 0.f);
        O[2 * no + o] = masked_out(bilinear_mix(ty, tx, q00.z, q01.z, q10.z, q11.z), m, 0.f);
        mask_dsp[(size_t)b * no + o] = m > 0.5f ? 1.f : 0.f;
    }
}

bool bad_hw(const char *what, int batch, int H, int W) {
    if (batch < 0 || H <= 0 || W <= 0 || (long long)H * W > (1 << 28)) {
        zs::set_err("%s: bad size (batch=%d H=%d W=%d)", what, batch, H, W);
        return true;
    }
    return false;
}

}  // namespace

extern "C" int zs_intr_param2mtx(const float *params, int batch, int H, int W, float *intr, void *stream) {
    if (bad_hw("zs_intr_param2mtx", batch, H, W)) return 0;
    if (batch == 0) return 1;
    if (!params || !intr) { zs::set_err("zs_intr_param2mtx: null pointer"); return 0; }
    hipLaunchKernelGGL(intr_param2mtx_kernel, dim3((batch + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), params, batch, (float)H, (float)W, intr);
    return zs::check_launch("zs_intr_param2mtx") ? 1 : 0;
}

extern "C" int zs_unproj_depth(const float *depth, const float *intr, int batch, int H, int W, float *points,
                               void *stream) {
    if (bad_hw("zs_unproj_depth", batch, H, W)) return 0;
    if (batch == 0) return 1;
    if (!depth || !intr || !points) { zs::set_err("zs_unproj_depth: null pointer"); return 0; }
    if (batch > 65535) { zs::set_err("zs_unproj_depth: batch %d > 65535", batch); return 0; }
    hipLaunchKernelGGL(unproj_kernel, dim3((H * W + 255) / 256, batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), depth, intr, H, W, points);
    return zs::check_launch("zs_unproj_depth") ? 1 : 0;
}

extern "C" int zs_valid_norm_fac(const float *points, const uint8_t *mask, int batch, int n, float *mean,
                                 float *max_dist, void *stream) {
    if (batch < 0 || n <= 0) { zs::set_err("zs_valid_norm_fac: bad size (batch=%d n=%d)", batch, n); return 0; }
    if (batch == 0) return 1;
    if (!points || !mask || !mean || !max_dist) { zs::set_err("zs_valid_norm_fac: null pointer"); return 0; }
    hipLaunchKernelGGL(norm_fac_kernel, dim3(batch), dim3(BLOCK), 0, static_cast<hipStream_t>(stream), points, mask,
                       n, mean, max_dist);
    return zs::check_launch("zs_valid_norm_fac") ? 1 : 0;
}

extern "C" int zs_masked_resample(const float *map, const float *mask, int batch, int channels, int H, int W,
                                  int Ho, int Wo, float bg, float *out, float *mask_out, void *stream) {
    if (bad_hw("zs_masked_resample", batch, H, W) || bad_hw("zs_masked_resample", batch, Ho, Wo)) return 0;
    if (channels <= 0) { zs::set_err("zs_masked_resample: channels=%d", channels); return 0; }
    if (batch == 0) return 1;
    if (!map || !mask || !out || !mask_out) { zs::set_err("zs_masked_resample: null pointer"); return 0; }
    if (batch > 65535) { zs::set_err("zs_masked_resample: batch %d > 65535", batch); return 0; }
    hipLaunchKernelGGL(masked_resample_kernel, dim3((Ho * Wo + 255) / 256, batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), map, mask, channels, H, W, Ho, Wo, bg, out, mask_out);
    return zs::check_launch("zs_masked_resample") ? 1 : 0;
}

extern "C" int zs_seen_surface(const float *depth, const float *intr, const float *mask, int batch, int H, int W,
                               int Ho, int Wo, float *seen_points, float *mean, float *scale, float *coord_dsp,
                               float *mask_dsp, void *stream) {
    if (bad_hw("zs_seen_surface", batch, H, W)) return 0;
    if (coord_dsp && bad_hw("zs_seen_surface", batch, Ho, Wo)) return 0;
    if (batch == 0) return 1;
    if (!depth || !intr || !mask || !seen_points || !mean || !scale || (coord_dsp && !mask_dsp)) {
        zs::set_err("zs_seen_surface: null pointer");
        return 0;
    }
    hipLaunchKernelGGL(seen_surface_kernel, dim3(batch), dim3(BLOCK), 0, static_cast<hipStream_t>(stream), depth,
                       intr, mask, H, W, Ho, Wo, seen_points, mean, scale, coord_dsp, mask_dsp);
    return zs::check_launch("zs_seen_surface") ? 1 : 0;
}

extern "C" size_t zs_seen_surface_workspace_bytes(int batch) {
    return batch > 0 ? (size_t)batch * SEEN_CH * 8 * sizeof(float) : 0;
}

extern "C" int zs_seen_surface_ws(const float *depth, const float *intr, const float *mask, int batch, int H, int W,
                                  int Ho, int Wo, float *seen_points, float *mean, float *scale, float *coord_dsp,
                                  float *mask_dsp, void *workspace, void *stream) {
    if (!workspace)
        return zs_seen_surface(depth, intr, mask, batch, H, W, Ho, Wo, seen_points, mean, scale, coord_dsp, mask_dsp, stream);
    if (bad_hw("zs_seen_surface", batch, H, W)) return 0;
    if (coord_dsp && bad_hw("zs_seen_surface", batch, Ho, Wo)) return 0;
    if (batch == 0) return 1;
    if (!depth || !intr || !mask || !seen_points || !mean || !scale || (coord_dsp && !mask_dsp)) {
        zs::set_err("zs_seen_surface: null pointer");
        return 0;
    }
    if (batch > 65535) { zs::set_err("zs_seen_surface_ws: batch %d > 65535", batch); return 0; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *ws = static_cast<float *>(workspace);
    const dim3 grid(SEEN_CH, batch);
    hipLaunchKernelGGL(seen_sums_kernel, grid, dim3(SEEN_T), 0, st, depth, intr, mask, H, W, ws);
    hipLaunchKernelGGL(seen_radius_kernel, grid, dim3(SEEN_T), 0, st, depth, intr, mask, H, W, ws);
    hipLaunchKernelGGL(seen_apply_kernel, grid, dim3(SEEN_T), 0, st, depth, intr, mask, H, W, Ho, Wo, ws, seen_points, mean, scale,
                       coord_dsp, mask_dsp);
    return zs::check_launch("zs_seen_surface") ? 1 : 0;
}
