// Uniform-grid exact nearest neighbour: the machinery shared by csrc/chamfer_grid.hip (Chamfer both ways on given
// clouds) and csrc/pose_search.hip (the same search on clouds that are rotated + normalised on the fly).
//
// A candidate cloud is binned into <= 16^3 cells over its bounding box (point_grid_build: one 1024-lane workgroup:
// bounding box, histogram, scan, scatter of (x, y, z, index) records sorted by cell); a query visits the cells around
// it ring by ring until a conservative lower bound on the distance to every unvisited cell exceeds the running minimum
// (point_grid_nearest).  A skipped candidate provably has a strictly larger d, and among evaluated candidates the
// winner is the lexicographic minimum of (d, index), with d = fma(dz,dz, fma(dy,dy, dx*dx)), (dx,dy,dz) = candidate -
// query: the same bits as the brute-force scan.
//
// Slot layout (floats): [16 words meta | 4112 words of cell starts | mc x (x, y, z, index)].
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace zs {
namespace pgrid {

constexpr int GRID_MAX_AXIS = 16;
constexpr int GRID_MAX_CELLS = GRID_MAX_AXIS * GRID_MAX_AXIS * GRID_MAX_AXIS;
constexpr int META_WORDS = 16;
constexpr int CELL_WORDS = GRID_MAX_CELLS + 16;   // 4097 cell starts, padded: records stay 16-byte aligned
constexpr int BUILD_THREADS = 1024;

__host__ __device__ inline size_t slot_words(int mc) {
    return (size_t)META_WORDS + CELL_WORDS + 4 * (size_t)mc;
}
__host__ __device__ inline int axis_cells(int mc) {  // ~2-3 candidates per cell, at most 16^3 cells
    int a = 1;
    while (a < GRID_MAX_AXIS && (long long)(a + 1) * (a + 1) * (a + 1) * 2 <= mc) a++;
    return a;
}

struct GridMeta {
    float minx, miny, minz;     // bounding box minimum
    float cx, cy, cz;           // cell size per axis
    float ix, iy, iz;           // cells per unit length (0 for a flat axis)
    float slack;                // absolute safety margin of the pruning bound
    int na;                     // cells per axis
};

__device__ __forceinline__ int cell_coord(float p, float mn, float inv, int na) {
    float t = (p - mn) * inv;
    t = fminf(fmaxf(t, 0.0f), (float)(na - 1));   // also maps NaN to 0
    return (int)t;
}

// One workgroup of BUILD_THREADS lanes bins the mc candidates point(i, x, y, z) into `slot`.
// Shared memory is the caller's: cnt[GRID_MAX_CELLS], red[6][BUILD_THREADS / 64], wsum[BUILD_THREADS / 64].
template <typename POINT>
__device__ __forceinline__ void point_grid_build(POINT point, int mc, float *__restrict__ slot, int *cnt,
                                                 float (*red)[BUILD_THREADS / 64], int *wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < mc; i += BUILD_THREADS) {
        float v[3];
        point(i, v[0], v[1], v[2]);
#pragma unroll
        for (int a = 0; a < 3; a++) {
            mn[a] = fminf(mn[a], v[a]);
            mx[a] = fmaxf(mx[a], v[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
        if (lane == 0) {
            red[a][wave] = mn[a];
            red[3 + a][wave] = mx[a];
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float lo = INFINITY, hi = -INFINITY;
        for (int w = 0; w < BUILD_THREADS / 64; w++) {
            lo = fminf(lo, red[a][w]);
            hi = fmaxf(hi, red[3 + a][w]);
        }
        mn[a] = lo;
        mx[a] = hi;
    }
    const int na = axis_cells(mc);
    GridMeta g;
    g.na = na;
    float ext[3], maxabs = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        ext[a] = mx[a] - mn[a];
        if (!(ext[a] > 0.f) || !isfinite(ext[a])) ext[a] = 0.f;   // flat or non-finite axis: one slab
        maxabs = fmaxf(maxabs, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
    }
    if (!isfinite(maxabs)) maxabs = 0.f;
    g.minx = mn[0]; g.miny = mn[1]; g.minz = mn[2];
    g.cx = ext[0] / na; g.cy = ext[1] / na; g.cz = ext[2] / na;
    g.ix = ext[0] > 0.f ? na / ext[0] : 0.f;
    g.iy = ext[1] > 0.f ? na / ext[1] : 0.f;
    g.iz = ext[2] > 0.f ? na / ext[2] : 0.f;
    // cell assignment and cell faces are computed with different roundings: a few ulps of the
    // coordinates / extents of slack keep the bound conservative
    g.slack = 8e-7f * (maxabs + ext[0] + ext[1] + ext[2]);
    if (tid == 0) {
        slot[0] = g.minx; slot[1] = g.miny; slot[2] = g.minz;
        slot[3] = g.cx; slot[4] = g.cy; slot[5] = g.cz;
        slot[6] = g.ix; slot[7] = g.iy; slot[8] = g.iz;
        slot[9] = g.slack;
        reinterpret_cast<int *>(slot)[10] = na;
    }
    for (int c = tid; c < GRID_MAX_CELLS; c += BUILD_THREADS) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < mc; i += BUILD_THREADS) {
        float px, py, pz;
        point(i, px, py, pz);
        const int c = (cell_coord(pz, g.minz, g.iz, na) * na + cell_coord(py, g.miny, g.iy, na)) * na +
                      cell_coord(px, g.minx, g.ix, na);
        atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    // exclusive scan of cnt[0..4096): 4 cells per thread
    int local[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        local[j] = cnt[tid * 4 + j];
        sum += local[j];
    }
    int x = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    int run = base + x - sum;
    int *cell_start = reinterpret_cast<int *>(slot) + META_WORDS;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        cnt[tid * 4 + j] = run;            // becomes the scatter cursor
        cell_start[tid * 4 + j] = run;
        run += local[j];
    }
    if (tid == BUILD_THREADS - 1) cell_start[GRID_MAX_CELLS] = run;
    __syncthreads();
    float *sorted = slot + META_WORDS + CELL_WORDS;
    for (int i = tid; i < mc; i += BUILD_THREADS) {
        float px, py, pz;
        point(i, px, py, pz);
        const int c = (cell_coord(pz, g.minz, g.iz, na) * na + cell_coord(py, g.miny, g.iy, na)) * na +
                      cell_coord(px, g.minx, g.ix, na);
        const int pos = atomicAdd(&cnt[c], 1);
        float4 rec;
        rec.x = px; rec.y = py; rec.z = pz; rec.w = __int_as_float(i);
        reinterpret_cast<float4 *>(sorted)[pos] = rec;
    }
}

// nearest candidate of (qx, qy, qz) in the grid of `slot`: squared distance and lowest index among equal minima
__device__ __forceinline__ void point_grid_nearest(const float *__restrict__ slot, float qx, float qy, float qz,
                                                   float &best, int &best_i) {
    const float minx = slot[0], miny = slot[1], minz = slot[2];
    const float csx = slot[3], csy = slot[4], csz = slot[5];
    const float ix = slot[6], iy = slot[7], iz = slot[8], slack = slot[9];
    const int na = reinterpret_cast<const int *>(slot)[10];
    const int *cell_start = reinterpret_cast<const int *>(slot) + META_WORDS;
    const float4 *sorted = reinterpret_cast<const float4 *>(slot + META_WORDS + CELL_WORDS);
    const int c0x = cell_coord(qx, minx, ix, na), c0y = cell_coord(qy, miny, iy, na),
              c0z = cell_coord(qz, minz, iz, na);
    best = INFINITY;
    best_i = 0;
    for (int r = 0; r < na; r++) {
        const int zlo = max(c0z - r, 0), zhi = min(c0z + r, na - 1);
        const int ylo = max(c0y - r, 0), yhi = min(c0y + r, na - 1);
        const int xlo = max(c0x - r, 0), xhi = min(c0x + r, na - 1);
        for (int cz = zlo; cz <= zhi; cz++)
            for (int cy = ylo; cy <= yhi; cy++) {
                const bool shell_zy = (abs(cz - c0z) == r) || (abs(cy - c0y) == r);
                // inside the slab only the two end cells of the x run belong to the shell
                const int xstep = shell_zy ? 1 : max(xhi - xlo, 1);
                for (int cx = xlo; cx <= xhi; cx += xstep) {
                    if (!shell_zy && abs(cx - c0x) != r) continue;
                    const int c = (cz * na + cy) * na + cx;
                    const int e = cell_start[c + 1];
                    for (int p = cell_start[c]; p < e; p++) {
                        const float4 rec = sorted[p];
                        const float dx = rec.x - qx, dy = rec.y - qy, dz = rec.z - qz;
                        const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        const int id = __float_as_int(rec.w);
                        if (d < best || (d == best && id < best_i)) {
                            best = d;
                            best_i = id;
                        }
                    }
                }
            }
        // every unvisited candidate lies outside the visited block along at least one axis;
        // sides of the block that coincide with the bounding box have nothing beyond them
        float bound = INFINITY;
        if (c0x - r > 0) bound = fminf(bound, qx - (minx + (float)(c0x - r) * csx));
        if (c0x + r < na - 1) bound = fminf(bound, (minx + (float)(c0x + r + 1) * csx) - qx);
        if (c0y - r > 0) bound = fminf(bound, qy - (miny + (float)(c0y - r) * csy));
        if (c0y + r < na - 1) bound = fminf(bound, (miny + (float)(c0y + r + 1) * csy) - qy);
        if (c0z - r > 0) bound = fminf(bound, qz - (minz + (float)(c0z - r) * csz));
        if (c0z + r < na - 1) bound = fminf(bound, (minz + (float)(c0z + r + 1) * csz) - qz);
        bound -= slack;
        if (bound > 0.f && bound * bound * (1.0f - 1e-5f) > best) break;   // also true for bound = +inf
    }
}

}  // namespace pgrid
}  // namespace zs
