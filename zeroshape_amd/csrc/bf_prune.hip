// Lower bounds for the brute-force pose search (utils/eval_3D.py:140-170): which of the 6912
// rotations can still beat the best Chamfer-L1 found so far?
//
// The reference evaluates the exact Chamfer distance for every rotation (1.4e12 point
// pairs per sample).  Most rotations leave the two clouds far apart; for those a cheap,
// RIGOROUS lower bound of the Chamfer-L1 proves they cannot win, and only the survivors
// need the exact kernels.  The search result is unchanged: a pruned rotation has a strictly
// larger distance than the winner (host side: zeroshape_amd/utils/eval_3D.py).
//
// Bound: a cloud is rasterised into a 32^3 occupancy grid over its bounding cube, and
// LB[c] = cell_size * min over occupied cells c' of sqrt(sum_axis max(0, |c_a - c'_a| - 1)^2)
// is a lower bound of the distance from ANY point of cell c to the cloud (points outside the
// cube are projected onto it first, which cannot increase distances to points inside).
//   direction 1: pred'_i = (R p_i - mu) / s  (normalize_pc, :93-102)  -> LB_gt[cell(pred'_i)]
//   direction 2: |g_j - pred'_i| = |R^T (mu + s g_j) - p_i| / s       -> LB_pred[cell(R^T(mu + s g_j))] / s
// lb[k] = (mean_i dir1 + mean_j dir2) / 2  <=  cd[k] up to rounding; the host applies a margin.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace {

constexpr int LB_AXIS = 32;
constexpr int LB_CELLS = LB_AXIS * LB_AXIS * LB_AXIS;
constexpr int LB_META = 8;  // min x,y,z, cell size, 1/cell size

__device__ __forceinline__ float block_reduce(float v, float *lds, int op) {  // 0 sum, 1 min, 2 max
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float y = __shfl_xor(v, o, 64);
        v = op == 0 ? v + y : op == 1 ? fminf(v, y) : fmaxf(v, y);
    }
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = lds[0];
    for (int w = 1; w < nw; w++) r = op == 0 ? r + lds[w] : op == 1 ? fminf(r, lds[w]) : fmaxf(r, lds[w]);
    return r;
}

// grid: [LB_META floats | LB_CELLS floats]; scratch: LB_CELLS + 1 ints (occupancy -> compact list)
__global__ __launch_bounds__(1024) void lb_build_kernel(const float *__restrict__ cloud, int n,
                                                        float *__restrict__ grid, int *__restrict__ scratch, int compact) {
    __shared__ float lds[16];
    __shared__ int n_occ;
    const int tid = threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += 1024)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = cloud[(size_t)i * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    float ext = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        mn[a] = block_reduce(mn[a], lds, 1);
        mx[a] = block_reduce(mx[a], lds, 2);
        ext = fmaxf(ext, mx[a] - mn[a]);
    }
    if (!(ext > 0.f) || !isfinite(ext)) ext = 1.0f;
    const float h = ext * (1.0f + 1e-5f) / LB_AXIS, inv = 1.0f / h;   // cube slightly larger than the box
    if (tid == 0) {
        grid[0] = mn[0]; grid[1] = mn[1]; grid[2] = mn[2]; grid[3] = h; grid[4] = inv;
        n_occ = 0;
    }
    int *occ = scratch;              // [LB_CELLS] flags, then reused as the compact list
    for (int c = tid; c < LB_CELLS; c += 1024) occ[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        int cc[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float t = (cloud[(size_t)i * 3 + a] - mn[a]) * inv;
            cc[a] = (int)fminf(fmaxf(t, 0.f), (float)(LB_AXIS - 1));
        }
        occ[(cc[2] * LB_AXIS + cc[1]) * LB_AXIS + cc[0]] = 1;
    }
    __syncthreads();
    if (!compact) return;            // the sweeps read the flags themselves
    // compact the occupied cells into grid-space indices (order irrelevant)
    int *list = scratch + LB_CELLS + 1;
    for (int c = tid; c < LB_CELLS; c += 1024)
        if (occ[c]) list[atomicAdd(&n_occ, 1)] = c;
    __syncthreads();
    if (tid == 0) scratch[LB_CELLS] = n_occ;
}

// The same field by three sweeps (round 6).  The cost sum_axis max(|d_a| - 1, 0)^2 is a SUM of per-axis terms, so the minimum
// over the occupied cells separates: along x  A[c] = min_ox (occupied(ox, cy, cz) ? g(|cx - ox|) : inf), along y
// B[c] = min_oy (A[cx, oy, cz] + g(|cy - oy|)), along z C[c] = min_oz (B[cx, cy, oz] + g(|cz - oz|)) - 32 candidates per cell
// and sweep instead of every occupied cell (a surface of 10k points occupies ~3,000): integer arithmetic, the SAME squared
// cell distance as lb_field_kernel below (kept as the reference form: ZS_BF_FIELD_BRUTE=1), 215 -> 3 x 5 us per cloud.
template <int AXIS>
__global__ __launch_bounds__(256) void lb_sweep_kernel(const int *__restrict__ src, int *__restrict__ dst,
                                                       float *__restrict__ grid) {
    constexpr int INF = 1 << 28;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int stride = AXIS == 0 ? 1 : AXIS == 1 ? LB_AXIS : LB_AXIS * LB_AXIS;
    const int pos = (c / stride) % LB_AXIS, base = c - pos * stride;
    int best = INF;
#pragma unroll
    for (int o = 0; o < LB_AXIS; o++) {
        const int v = src[base + o * stride];
        const int d = max(abs(o - pos) - 1, 0);
        const int cand = (AXIS == 0 ? (v ? 0 : INF) : v) + d * d;          // the first sweep reads the occupancy flags
        best = min(best, cand);
    }
    if (AXIS < 2) dst[c] = best;
    else grid[LB_META + c] = best < INF ? grid[3] * sqrtf((float)best) * (1.0f - 1e-4f) : 0.f;   // (no occupied cell: 0)
}

__global__ __launch_bounds__(256) void lb_field_kernel(float *__restrict__ grid, const int *__restrict__ scratch) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int n_occ = scratch[LB_CELLS];
    const int *list = scratch + LB_CELLS + 1;
    const int cx = c % LB_AXIS, cy = (c / LB_AXIS) % LB_AXIS, cz = c / (LB_AXIS * LB_AXIS);
    int best = 0x7fffffff;
    for (int t = 0; t < n_occ; t++) {
        const int o = list[t];
        const int dx = max(abs(o % LB_AXIS - cx) - 1, 0), dy = max(abs((o / LB_AXIS) % LB_AXIS - cy) - 1, 0),
                  dz = max(abs(o / (LB_AXIS * LB_AXIS) - cz) - 1, 0);
        best = min(best, dx * dx + dy * dy + dz * dz);
    }
    // exact integer cell distance, then one rounding each; shaved by 1e-4 for the roundings of
    // the cell assignment
    grid[LB_META + c] = n_occ > 0 ? grid[3] * sqrtf((float)best) * (1.0f - 1e-4f) : 0.f;
}

__device__ __forceinline__ float lb_lookup(const float *__restrict__ grid, float x, float y, float z) {
    const float inv = grid[4];
    const int cx = (int)fminf(fmaxf((x - grid[0]) * inv, 0.f), (float)(LB_AXIS - 1));
    const int cy = (int)fminf(fmaxf((y - grid[1]) * inv, 0.f), (float)(LB_AXIS - 1));
    const int cz = (int)fminf(fmaxf((z - grid[2]) * inv, 0.f), (float)(LB_AXIS - 1));
    return grid[LB_META + (cz * LB_AXIS + cy) * LB_AXIS + cx];
}

// one block per rotation
__global__ __launch_bounds__(256) void bf_lower_bound_kernel(
    const float *__restrict__ pred, int n, const float *__restrict__ gt, int m,
    const float *__restrict__ rot, const float *__restrict__ grid_gt,
    const float *__restrict__ grid_pred, float *__restrict__ lb) {
    __shared__ float lds[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = rot[(size_t)k * 9 + i];
    // normalize_pc statistics of the rotated cloud
    float sx = 0.f, sy = 0.f, sz = 0.f, xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int i = tid; i < n; i += 256) {
        const float px = pred[(size_t)i * 3], py = pred[(size_t)i * 3 + 1], pz = pred[(size_t)i * 3 + 2];
        const float x = R[0] * px + R[1] * py + R[2] * pz, y = R[3] * px + R[4] * py + R[5] * pz,
                    z = R[6] * px + R[7] * py + R[8] * pz;
        sx += x; sy += y; sz += z;
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
        ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
    }
    const float mux = block_reduce(sx, lds, 0) / n, muy = block_reduce(sy, lds, 0) / n,
                muz = block_reduce(sz, lds, 0) / n;
    const float ex = block_reduce(xmax, lds, 2) - block_reduce(xmin, lds, 1);
    const float ey = block_reduce(ymax, lds, 2) - block_reduce(ymin, lds, 1);
    const float s = fmaxf(ex, ey) + 1e-7f, inv_s = 1.0f / s;
    float a1 = 0.f, a2 = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float px = pred[(size_t)i * 3], py = pred[(size_t)i * 3 + 1], pz = pred[(size_t)i * 3 + 2];
        const float x = (R[0] * px + R[1] * py + R[2] * pz - mux) * inv_s,
                    y = (R[3] * px + R[4] * py + R[5] * pz - muy) * inv_s,
                    z = (R[6] * px + R[7] * py + R[8] * pz - muz) * inv_s;
        a1 += lb_lookup(grid_gt, x, y, z);
    }
    for (int j = tid; j < m; j += 256) {
        const float wx = mux + s * gt[(size_t)j * 3], wy = muy + s * gt[(size_t)j * 3 + 1],
                    wz = muz + s * gt[(size_t)j * 3 + 2];
        // R^T w
        const float x = R[0] * wx + R[3] * wy + R[6] * wz, y = R[1] * wx + R[4] * wy + R[7] * wz,
                    z = R[2] * wx + R[5] * wy + R[8] * wz;
        a2 += lb_lookup(grid_pred, x, y, z);
    }
    a1 = block_reduce(a1, lds, 0);
    a2 = block_reduce(a2, lds, 0);
    if (tid == 0) lb[k] = 0.5f * (a1 / n + a2 * inv_s / m);
}

}  // namespace

extern "C" size_t zs_bf_grid_bytes(void) { return (size_t)(LB_META + LB_CELLS) * sizeof(float); }
extern "C" size_t zs_bf_scratch_bytes(void) { return (size_t)(2 * LB_CELLS + 2) * sizeof(int); }

extern "C" int zs_bf_lower_bounds(const float *pred, int n, const float *gt_normalized, int m,
                                  const float *rotations, int k, float *grid_gt, float *grid_pred,
                                  void *scratch, float *lower_bounds, void *stream) {
    if (n <= 0 || m <= 0 || k < 0) {
        zs::set_err("zs_bf_lower_bounds: bad size (n=%d m=%d k=%d)", n, m, k);
        return 0;
    }
    if (k == 0) return 1;
    if (!pred || !gt_normalized || !rotations || !grid_gt || !grid_pred || !scratch || !lower_bounds) {
        zs::set_err("zs_bf_lower_bounds: null pointer");
        return 0;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int *sc = static_cast<int *>(scratch);
    const char *env = getenv("ZS_BF_FIELD_BRUTE");                          // A/B switch, read per call: the all-occupied-cells form
    const bool brute = env && atoi(env) != 0;
    auto field = [&](const float *cloud, int count, float *grid) {
        hipLaunchKernelGGL(lb_build_kernel, dim3(1), dim3(1024), 0, s, cloud, count, grid, sc, brute ? 1 : 0);
        if (brute) {
            hipLaunchKernelGGL(lb_field_kernel, dim3(LB_CELLS / 256), dim3(256), 0, s, grid, sc);
            return;
        }
        int *flags = sc, *tmp = sc + LB_CELLS + 1;       // [flags | (count) | second buffer]: ping-pong between the two areas
        hipLaunchKernelGGL((lb_sweep_kernel<0>), dim3(LB_CELLS / 256), dim3(256), 0, s, flags, tmp, grid);
        hipLaunchKernelGGL((lb_sweep_kernel<1>), dim3(LB_CELLS / 256), dim3(256), 0, s, tmp, flags, grid);
        hipLaunchKernelGGL((lb_sweep_kernel<2>), dim3(LB_CELLS / 256), dim3(256), 0, s, flags, tmp, grid);
    };
    field(gt_normalized, m, grid_gt);
    field(pred, n, grid_pred);
    hipLaunchKernelGGL(bf_lower_bound_kernel, dim3(k), dim3(256), 0, s, pred, n, gt_normalized, m,
                       rotations, grid_gt, grid_pred, lower_bounds);
    return zs::check_launch("zs_bf_lower_bounds") ? 1 : 0;
}
