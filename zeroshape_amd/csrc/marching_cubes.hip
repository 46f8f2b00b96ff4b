// Iso-surface extraction + area-weighted surface sampling on the GPU (gfx950).
//
// Replaces the host-side step of the reference's evaluation, utils/eval_3D.py:233-263
// (convert_to_explicit: PyMCubes marching_cubes at iso 0.5 -> trimesh -> mesh.sample(10000)),
// which forces a device->host copy of the whole level grid and serialises the GPU behind
// Python threads.  Here the grid never leaves HBM: count -> scan -> emit is one read of the
// volume per pass (HBM-bound, G^3 * 4 bytes), sampling is a binary search per point.
//
// Conventions (zeroshape_amd/mc_tables.py generates the case tables): corner/edge
// numbering of the classic algorithm, case bit set when value < iso, vertices in voxel
// index space mapped to world space as v * scale + offset with scale = (max - min) / G
// (the reference divides by S = G = N+1, utils/eval_3D.py:252-255 - reproduced, not fixed).
// Every edge vertex is interpolated from its lower-coordinate endpoint, so the two to four
// cubes sharing an edge produce bit-identical vertices.  Output is a triangle soup
// [n][3][3] in cube order (x slowest, z fastest), table order within a cube: deterministic.
//
// PyMCubes / trimesh are not installable in this environment: parity with them is
// "unpinned" (DESIGN.md section 5); the oracle is oracle/mc_ref.py (same tables, numpy).
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int MC_THREADS = 256;

// edge e: (corner a, corner b) listed low-coordinate endpoint first, its axis, and the
// corner offsets (mc_tables.py numbering)
__device__ const int kEdgeA[12] = {0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3};
__device__ const int kEdgeB[12] = {1, 2, 2, 3, 5, 6, 6, 7, 4, 5, 6, 7};
__device__ const int kEdgeAxis[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
__device__ const int kCornerX[8] = {0, 1, 1, 0, 0, 1, 1, 0};
__device__ const int kCornerY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
__device__ const int kCornerZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};

__device__ __forceinline__ int cube_case(const float *__restrict__ vol, int G, int i, int j, int k,
                                         float iso, float f[8]) {
    const size_t gg = (size_t)G * G;
    const float *p = vol + (size_t)i * gg + (size_t)j * G + k;
    f[0] = p[0];
    f[1] = p[gg];
    f[2] = p[gg + G];
    f[3] = p[G];
    f[4] = p[1];
    f[5] = p[gg + 1];
    f[6] = p[gg + G + 1];
    f[7] = p[G + 1];
    int c = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) c |= (f[b] < iso) ? (1 << b) : 0;
    return c;
}

// block-wide exclusive scan of one int per thread (MC_THREADS threads); returns the block total
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MC_THREADS / 64; w++) {
        const int s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(MC_THREADS) void mc_count_kernel(const float *__restrict__ vol, int G,
                                                              float iso,
                                                              const uint8_t *__restrict__ tri_count,
                                                              int *__restrict__ block_sums) {
    __shared__ int lds[MC_THREADS / 64];
    const int C = G - 1;
    const long long cube = (long long)blockIdx.x * MC_THREADS + threadIdx.x;
    int n = 0;
    if (cube < (long long)C * C * C) {
        const int k = (int)(cube % C), j = (int)((cube / C) % C), i = (int)(cube / ((long long)C * C));
        float f[8];
        n = tri_count[cube_case(vol, G, i, j, k, iso, f)];
    }
    int total;
    block_exclusive_scan(n, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// exclusive scan of `n` ints in place by ONE block (n <= a few 10^4); writes the grand total
__global__ __launch_bounds__(1024) void scan_small_kernel(int *__restrict__ data, int n,
                                                          int *__restrict__ total_out) {
    __shared__ int lds[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int idx = base + threadIdx.x;
        const int v = idx < n ? data[idx] : 0;
        lds[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int y = threadIdx.x >= o ? lds[threadIdx.x - o] : 0;
            __syncthreads();
            lds[threadIdx.x] += y;
            __syncthreads();
        }
        const int incl = lds[threadIdx.x];
        const int c = carry;
        if (idx < n) data[idx] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(MC_THREADS) void mc_emit_kernel(
    const float *__restrict__ vol, int G, float iso, const int8_t *__restrict__ tri_table,
    int table_stride, const uint8_t *__restrict__ tri_count, const int *__restrict__ block_offsets,
    float scale, float offset, float *__restrict__ tris, int max_tris) {
    __shared__ int lds[MC_THREADS / 64];
    const int C = G - 1;
    const long long cube = (long long)blockIdx.x * MC_THREADS + threadIdx.x;
    int n = 0, cs = 0, i = 0, j = 0, k = 0;
    float f[8];
    if (cube < (long long)C * C * C) {
        k = (int)(cube % C);
        j = (int)((cube / C) % C);
        i = (int)(cube / ((long long)C * C));
        cs = cube_case(vol, G, i, j, k, iso, f);
        n = tri_count[cs];
    }
    int total;
    const int local = block_exclusive_scan(n, lds, total);
    if (n == 0) return;
    const int first = block_offsets[blockIdx.x] + local;
    for (int t = 0; t < n; t++) {
        if (first + t >= max_tris) return;
        float *o = tris + (size_t)(first + t) * 9;
#pragma unroll
        for (int v = 0; v < 3; v++) {
            const int e = tri_table[cs * table_stride + 3 * t + v];
            const int a = kEdgeA[e], b = kEdgeB[e], axis = kEdgeAxis[e];
            const float tt = (iso - f[a]) / (f[b] - f[a]);
            const float px = (float)(i + kCornerX[a]) + (axis == 0 ? tt : 0.0f);
            const float py = (float)(j + kCornerY[a]) + (axis == 1 ? tt : 0.0f);
            const float pz = (float)(k + kCornerZ[a]) + (axis == 2 ? tt : 0.0f);
            const float p[3] = {px, py, pz};
            o[3 * v + 0] = fmaf(p[0], scale, offset);
            o[3 * v + 1] = fmaf(p[1], scale, offset);
            o[3 * v + 2] = fmaf(p[2], scale, offset);
        }
    }
}

// ---- area-weighted sampling ------------------------------------------------------------- //
__global__ __launch_bounds__(256) void tri_area_kernel(const float *__restrict__ tris, int n,
                                                       double *__restrict__ area) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float *p = tris + (size_t)t * 9;
    const float ux = p[3] - p[0], uy = p[4] - p[1], uz = p[5] - p[2];
    const float vx = p[6] - p[0], vy = p[7] - p[1], vz = p[8] - p[2];
    const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    area[t] = 0.5 * sqrt((double)nx * nx + (double)ny * ny + (double)nz * nz);
}

// inclusive scan of doubles in place, one block, sequential chunks (n ~ 10^5: a few 100 us)
__global__ __launch_bounds__(1024) void scan_f64_kernel(double *__restrict__ data, int n) {
    __shared__ double lds[1024];
    __shared__ double carry;
    if (threadIdx.x == 0) carry = 0.0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int idx = base + threadIdx.x;
        lds[threadIdx.x] = idx < n ? data[idx] : 0.0;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const double y = threadIdx.x >= o ? lds[threadIdx.x - o] : 0.0;
            __syncthreads();
            lds[threadIdx.x] += y;
            __syncthreads();
        }
        const double incl = lds[threadIdx.x] + carry;
        if (idx < n) data[idx] = incl;
        __syncthreads();
        if (threadIdx.x == 1023) carry = incl;
        __syncthreads();
    }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint64_t seed, uint64_t ctr) {
    return (float)(splitmix64(seed ^ splitmix64(ctr)) >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void mesh_sample_kernel(const float *__restrict__ tris, int n,
                                                          const double *__restrict__ cum,
                                                          uint64_t seed, int n_samples,
                                                          float *__restrict__ pts) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_samples) return;
    const double total = cum[n - 1];
    const double target = (double)u01(seed, 3ull * s) * total;
    int lo = 0, hi = n - 1;  // first index with cum[idx] > target
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] > target) hi = mid; else lo = mid + 1;
    }
    float r1 = u01(seed, 3ull * s + 1), r2 = u01(seed, 3ull * s + 2);
    if (r1 + r2 > 1.0f) {  // reflect into the triangle (trimesh.sample does the same)
        r1 = 1.0f - r1;
        r2 = 1.0f - r2;
    }
    const float *p = tris + (size_t)lo * 9;
#pragma unroll
    for (int c = 0; c < 3; c++)
        pts[(size_t)s * 3 + c] = fmaf(r2, p[6 + c] - p[c], fmaf(r1, p[3 + c] - p[c], p[c]));
}

}  // namespace

extern "C" size_t zs_mc_scratch_bytes(int G) {
    if (G < 2) return 0;
    const long long cubes = (long long)(G - 1) * (G - 1) * (G - 1);
    return (size_t)((cubes + MC_THREADS - 1) / MC_THREADS + 1) * sizeof(int);
}

extern "C" int zs_mc_count(const float *vol, int G, float iso, const uint8_t *tri_count,
                           void *scratch, int *total, void *stream) {
    if (G < 2 || G > 2048) {
        zs::set_err("zs_mc_count: bad grid size %d", G);
        return 0;
    }
    if (!vol || !tri_count || !scratch || !total) {
        zs::set_err("zs_mc_count: null pointer");
        return 0;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long cubes = (long long)(G - 1) * (G - 1) * (G - 1);
    const int nb = (int)((cubes + MC_THREADS - 1) / MC_THREADS);
    int *sums = static_cast<int *>(scratch);
    hipLaunchKernelGGL(mc_count_kernel, dim3(nb), dim3(MC_THREADS), 0, s, vol, G, iso, tri_count, sums);
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, s, sums, nb, total);
    return zs::check_launch("zs_mc_count") ? 1 : 0;
}

extern "C" int zs_mc_emit(const float *vol, int G, float iso, const int8_t *tri_table,
                          int table_stride, const uint8_t *tri_count, const void *scratch, float scale,
                          float offset, float *tris, int n_tris, void *stream) {
    if (G < 2 || G > 2048 || n_tris < 0) {
        zs::set_err("zs_mc_emit: bad size (G=%d n_tris=%d)", G, n_tris);
        return 0;
    }
    if (n_tris == 0) return 1;
    if (!vol || !tri_table || !tri_count || !scratch || !tris) {
        zs::set_err("zs_mc_emit: null pointer");
        return 0;
    }
    const long long cubes = (long long)(G - 1) * (G - 1) * (G - 1);
    const int nb = (int)((cubes + MC_THREADS - 1) / MC_THREADS);
    hipLaunchKernelGGL(mc_emit_kernel, dim3(nb), dim3(MC_THREADS), 0, static_cast<hipStream_t>(stream),
                       vol, G, iso, tri_table, table_stride, tri_count, static_cast<const int *>(scratch),
                       scale, offset, tris, n_tris);
    return zs::check_launch("zs_mc_emit") ? 1 : 0;
}

extern "C" int zs_mesh_sample(const float *tris, int n_tris, int n_samples, uint64_t seed,
                              double *cum_area, float *points, void *stream) {
    if (n_tris < 0 || n_samples < 0) {
        zs::set_err("zs_mesh_sample: negative size");
        return 0;
    }
    if (n_samples == 0) return 1;
    if (!points) {
        zs::set_err("zs_mesh_sample: null pointer");
        return 0;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_tris == 0) {  // empty mesh -> zeros (utils/eval_3D.py:262)
        (void)hipMemsetAsync(points, 0, (size_t)n_samples * 3 * sizeof(float), s);
        return zs::check_launch("zs_mesh_sample") ? 1 : 0;
    }
    if (!tris || !cum_area) {
        zs::set_err("zs_mesh_sample: null pointer");
        return 0;
    }
    hipLaunchKernelGGL(tri_area_kernel, dim3((n_tris + 255) / 256), dim3(256), 0, s, tris, n_tris, cum_area);
    hipLaunchKernelGGL(scan_f64_kernel, dim3(1), dim3(1024), 0, s, cum_area, n_tris);
    hipLaunchKernelGGL(mesh_sample_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, s, tris, n_tris,
                       cum_area, seed, n_samples, points);
    return zs::check_launch("zs_mesh_sample") ? 1 : 0;
}
