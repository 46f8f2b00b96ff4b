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
constexpr int MC_PER = 4;                  // consecutive cubes of one k-row per thread (count pass)
constexpr int MC_BLOCK = MC_THREADS * MC_PER;   // cubes per workgroup of the count pass

// Cubes are enumerated row by row (a row = the C cubes of one (i, j), k fastest), every row padded to a multiple of
// MC_PER "virtual" cubes, so that a thread of the count pass owns MC_PER cubes of ONE row and reads their corners as
// four 16-byte loads (round 3; the output order - x slowest, z fastest - does not change: padding cubes emit nothing).
struct CubeSpace {
    int C, per_row;            // cubes per row, virtual cubes per row
    long long total;           // virtual cubes
};
__host__ __device__ inline CubeSpace cube_space(int G) {
    const int C = G - 1, per_row = (C + MC_PER - 1) / MC_PER * MC_PER;
    return CubeSpace{C, per_row, (long long)C * C * per_row};
}
// virtual cube -> (i, j, k); k >= C marks padding
__device__ __forceinline__ void virtual_ijk(const CubeSpace &cs, long long v, int &i, int &j, int &k) {
    if (cs.total < (1LL << 31)) {
        const unsigned c = (unsigned)v, row = c / (unsigned)cs.per_row;
        k = (int)(c - row * (unsigned)cs.per_row);
        i = (int)(row / (unsigned)cs.C);
        j = (int)(row - (unsigned)i * (unsigned)cs.C);
    } else {
        const long long row = v / cs.per_row;
        k = (int)(v - row * cs.per_row);
        i = (int)(row / cs.C);
        j = (int)(row - (long long)i * cs.C);
    }
}

// edge e: (corner a, corner b) listed low-coordinate endpoint first, its axis, and the
// corner offsets (mc_tables.py numbering)
__device__ const int kEdgeA[12] = {0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3};
__device__ const int kEdgeB[12] = {1, 2, 2, 3, 5, 6, 6, 7, 4, 5, 6, 7};
__device__ const int kEdgeAxis[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
__device__ const int kCornerX[8] = {0, 1, 1, 0, 0, 1, 1, 0};
__device__ const int kCornerY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
__device__ const int kCornerZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};

__device__ __forceinline__ void cube_corners(const float *__restrict__ vol, int G, int i, int j, int k, float f[8]) {
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
}
__device__ __forceinline__ int case_of(const float f[8], float iso) {
    int c = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) c |= (f[b] < iso) ? (1 << b) : 0;
    return c;
}

// linear cube index (k fastest) -> (i, j, k); 32-bit divisions whenever the index fits (G <= 1291): the 64-bit ones
// cost more than the eight loads of the cube
__device__ __forceinline__ void cube_ijk(long long cube, int C, int &i, int &j, int &k) {
    if ((long long)C * C * C < (1LL << 31)) {
        const unsigned c = (unsigned)cube, q = c / (unsigned)C;
        k = (int)(c - q * (unsigned)C);
        i = (int)(q / (unsigned)C);
        j = (int)(q - (unsigned)i * (unsigned)C);
    } else {
        k = (int)(cube % C);
        j = (int)((cube / C) % C);
        i = (int)(cube / ((long long)C * C));
    }
}

// block-wide exclusive scan of one int per thread (64 * NW threads); returns the block total
template <int NW = MC_THREADS / 64>
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
    for (int w = 0; w < NW; w++) {
        const int s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + x - v;
}

// Count pass.  Round 2 read every corner with its own 4-byte load (32 loads per thread, each wave instruction a
// 1 KiB span with a quarter of it used): 1.4 TB/s at 257^3, bound by the texture-address path, not by HBM.  Now a
// thread owns four consecutive cubes of one k-row: the four corner rows (i | i+1, j | j+1) arrive as four
// global_load_dwordx4 (a wave covers 1 KiB contiguous per instruction), the fifth value of each row comes from the
// next lane (same row) or one extra load at a row's end, and the 20 comparisons against iso are formed once.
__global__ __launch_bounds__(MC_THREADS) void mc_count_kernel(const float *__restrict__ vol, int G,
                                                              float iso,
                                                              const uint8_t *__restrict__ tri_count,
                                                              int *__restrict__ block_sums, long long n_blocks) {
    const CubeSpace cs = cube_space(G);
    const long long first = ((long long)blockIdx.x * MC_THREADS + threadIdx.x) * MC_PER;
    const int lane = threadIdx.x & 63;
    int n = 0;
    const bool live = first < cs.total;
    int i = 0, j = 0, k0 = 0;
    if (live) virtual_ijk(cs, first, i, j, k0);
    const size_t gg = (size_t)G * G;
    const float *p = vol + (size_t)i * gg + (size_t)j * G + k0;
    const float *rows[4] = {p, p + gg, p + gg + G, p + G};      // corners 0/4, 1/5, 2/6, 3/7 at k and k + 1
    unsigned below[4];                                          // bit e: value k0 + e of the row is below iso
    const bool wide = live && k0 + MC_PER - 1 <= G - 1;         // the 16-byte load stays inside the row
    float head[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float v[MC_PER] = {0.f, 0.f, 0.f, 0.f};
        if (wide) {
            // (rows are G floats apart: 4-byte aligned only, which global_load_dwordx4 accepts)
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const f4u q = *reinterpret_cast<const f4u *>(rows[r]);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else if (live) {
#pragma unroll
            for (int e = 0; e < MC_PER; e++) v[e] = rows[r][min(e, G - 1 - k0)];
        }
        head[r] = v[0];
        below[r] = 0;
#pragma unroll
        for (int e = 0; e < MC_PER; e++) below[r] |= (v[e] < iso) ? (1u << e) : 0u;
    }
    // value k0 + 4 of every row: the next lane's first value when that lane continues this row
    const bool next_same_row = lane < 63 && k0 + MC_PER < cs.per_row && first + MC_PER < cs.total;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float v = __shfl_down(head[r], 1, 64);
        if (live && !next_same_row) v = rows[r][min(MC_PER, G - 1 - k0)];
        below[r] |= (v < iso) ? (1u << MC_PER) : 0u;
    }
    if (live) {
#pragma unroll
        for (int q = 0; q < MC_PER; q++) {
            if (k0 + q >= cs.C) break;
            int c = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                c |= ((below[r] >> q) & 1u) << r;
                c |= ((below[r] >> (q + 1)) & 1u) << (4 + r);
            }
            n += tri_count[c];
        }
    }
    // a wave's 64 x MC_PER consecutive virtual cubes are one block of MC_THREADS cubes of the emit pass: its sum is that
    // block's entry, no exchange between the waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    const long long blk = (long long)blockIdx.x * (MC_BLOCK / MC_THREADS) + (threadIdx.x >> 6);
    if (lane == 0 && blk < n_blocks) block_sums[blk] = n;
    if (blockIdx.x == 0 && threadIdx.x == 0) block_sums[n_blocks] = 0;     // the scan turns it into the total
}

// ---- scans over many workgroups, two launches -------------------------------------------------------------- //
// launch 1: a workgroup scans SCAN_TILE consecutive elements in place (a thread loads SCAN_ITEMS adjacent ones, so
// a wave reads one contiguous span; wave shuffles + one LDS exchange) and leaves its total in tile_tot[tile];
// launch 2: every workgroup adds the sum of the tile totals before its own (a few hundred values, summed in index
// order by each workgroup alike - deterministic) to its elements.  A single workgroup walking 10^5 elements took
// 50-240 us; this takes two launches of a few microseconds.
constexpr int SCAN_ITEMS = 8, SCAN_TILE = 256 * SCAN_ITEMS;
template <typename T>
__device__ __forceinline__ T block256_exclusive(T v, T *lds, T &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const T t = lds[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + x - v;
}
// INCLUSIVE = false: data[i] <- sum of data[tile start .. i);  true: .. i]
template <typename T, bool INCLUSIVE>
__global__ __launch_bounds__(256) void scan_tiles_kernel(T *__restrict__ data, long long n, T *__restrict__ tile_tot) {
    __shared__ T lds[4];
    const long long b = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; e++) {
        v[e] = b + e < n ? data[b + e] : (T)0;
        s += v[e];
    }
    T total;
    T run = block256_exclusive<T>(s, lds, total);
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; e++) {
        if (INCLUSIVE) run += v[e];
        if (b + e < n) data[b + e] = run;
        if (!INCLUSIVE) run += v[e];
    }
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = total;
}
template <typename T>
__global__ __launch_bounds__(256) void scan_offsets_kernel(T *__restrict__ data, long long n, const T *__restrict__ tile_tot,
                                                           int tiles, T *__restrict__ total_out) {
    __shared__ T lds[4];
    __shared__ T offset;
    // sum of the totals of the tiles before this one, in a fixed order: 256 strided partial sums, then the scan's tree
    T part = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) part += tile_tot[t];
    T total;
    (void)block256_exclusive<T>(part, lds, total);
    if (threadIdx.x == 0) offset = total;
    __syncthreads();
    const T off = offset;
    const long long b = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; e++)
        if (b + e < n) data[b + e] += off;
    if (total_out && blockIdx.x == tiles - 1 && threadIdx.x == 0) *total_out = off + tile_tot[tiles - 1];
}
static inline int scan_tiles(long long n) { return (int)((n + SCAN_TILE - 1) / SCAN_TILE); }

// one cube per thread; a workgroup's MC_THREADS cubes are one entry of the count pass's block offsets
__global__ __launch_bounds__(MC_THREADS) void mc_emit_kernel(
    const float *__restrict__ vol, int G, float iso, const int8_t *__restrict__ tri_table,
    int table_stride, const uint8_t *__restrict__ tri_count, const int *__restrict__ block_offsets,
    float scale, float offset, float *__restrict__ tris, int max_tris) {
    __shared__ int lds[MC_THREADS / 64];
    // a block without triangles (most of them: the surface meets a few per cent of the k-rows) leaves at once, without
    // touching the volume - the count pass already knows
    const int block_first = block_offsets[blockIdx.x];
    if (block_offsets[blockIdx.x + 1] == block_first) return;
    const CubeSpace space = cube_space(G);
    const long long cube = (long long)blockIdx.x * MC_THREADS + threadIdx.x;
    int n = 0, cs = 0, i = 0, j = 0, k = 0;
    float f[8];
    if (cube < space.total) {
        virtual_ijk(space, cube, i, j, k);
        if (k < space.C) {
            cube_corners(vol, G, i, j, k, f);
            cs = case_of(f, iso);
            n = tri_count[cs];
        }
    }
    int total;
    const int local = block_exclusive_scan(n, lds, total);
    if (n == 0) return;
    const int first = block_first + local;
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
            o[3 * v + 0] = fmaf(px, scale, offset);
            o[3 * v + 1] = fmaf(py, scale, offset);
            o[3 * v + 2] = fmaf(pz, scale, offset);
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

static inline long long mc_blocks(int G) { return (cube_space(G).total + MC_THREADS - 1) / MC_THREADS; }

extern "C" size_t zs_mc_scratch_bytes(int G) {
    if (G < 2) return 0;
    const long long nb = mc_blocks(G);
    return (size_t)(nb + 2 + scan_tiles(nb + 1) + 1) * sizeof(int);    // block offsets + total | totals of the scan's tiles
}

extern "C" size_t zs_mesh_sample_scratch_doubles(int n_tris) {
    return n_tris < 0 ? 0 : (size_t)n_tris + scan_tiles(n_tris) + 1;     // cumulative areas | totals of the scan's tiles
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
    const long long cubes = cube_space(G).total;
    const long long nb = mc_blocks(G);
    int *sums = static_cast<int *>(scratch);
    hipLaunchKernelGGL(mc_count_kernel, dim3((unsigned)((cubes + MC_BLOCK - 1) / MC_BLOCK)), dim3(MC_THREADS), 0, s, vol, G, iso,
                       tri_count, sums, nb);
    // exclusive scan over the nb block sums and one trailing zero: offsets[b + 1] - offsets[b] = triangles of block b
    int *tile_tot = sums + nb + 2;
    const int tiles = scan_tiles(nb + 1);
    hipLaunchKernelGGL((scan_tiles_kernel<int, false>), dim3(tiles), dim3(256), 0, s, sums, nb + 1, tile_tot);
    hipLaunchKernelGGL(scan_offsets_kernel<int>, dim3(tiles), dim3(256), 0, s, sums, nb + 1, tile_tot, tiles, total);
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
    const int nb = (int)mc_blocks(G);
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
    double *tile_tot = cum_area + n_tris;
    const int tiles = scan_tiles(n_tris);
    hipLaunchKernelGGL((scan_tiles_kernel<double, true>), dim3(tiles), dim3(256), 0, s, cum_area, (long long)n_tris, tile_tot);
    hipLaunchKernelGGL(scan_offsets_kernel<double>, dim3(tiles), dim3(256), 0, s, cum_area, (long long)n_tris, tile_tot, tiles,
                       static_cast<double *>(nullptr));
    hipLaunchKernelGGL(mesh_sample_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, s, tris, n_tris,
                       cum_area, seed, n_samples, points);
    return zs::check_launch("zs_mesh_sample") ? 1 : 0;
}
