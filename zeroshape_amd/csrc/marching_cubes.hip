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

// edge e joins corner a (its low-coordinate endpoint) and corner b along `axis` (mc_tables.py numbering):
//   a    = {0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3}, axis = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2}   (packed below, kEdge*Packed)
//   corner c sits at (x, y, z) = ((0x66 >> c) & 1, (0xCC >> c) & 1, (0xF0 >> c) & 1)
// The corner values of the MC_PER cubes a lane owns: rows[r][e] = value e (k0 .. k0 + MC_PER) of corner row r
// (corners 0/4, 1/5, 2/6, 3/7 at k and k + 1).  Round 2 read every corner with its own 4-byte load (32 loads per
// thread, each wave instruction a 1 KiB span with a quarter of it used): 1.4 TB/s at 257^3, bound by the
// texture-address path, not by HBM.  Now the four rows arrive as four global_load_dwordx4 (a wave covers 1 KiB
// contiguous per instruction) and the fifth value of each row comes from the next lane (same row) or one extra load at
// a row's end.  `first` = the lane's first virtual cube; lanes of one wave own consecutive groups.
// ROWWAVE (cs.per_row == 64 * MC_PER, e.g. 257^3): a wave is exactly one k-row, so the one value no lane's 16-byte load
// covers - the row's last - sits at a wave-uniform address and is read through the scalar cache (s_load_dword) instead
// of a vector-memory instruction with one live lane.
template <bool ROWWAVE = false>
__device__ __forceinline__ bool load_cube_rows(const float *__restrict__ vol, int G, const CubeSpace &cs, long long first,
                                               int lane, int &i, int &j, int &k0, float (&rows)[4][MC_PER + 1]) {
    const bool live = first < cs.total;
    i = j = k0 = 0;
    if (live) virtual_ijk(cs, first, i, j, k0);
    const size_t gg = (size_t)G * G;
    const float *p = vol + (size_t)i * gg + (size_t)j * G + k0;
    const float *src[4] = {p, p + gg, p + gg + G, p + G};
    const bool wide = live && k0 + MC_PER - 1 <= G - 1;         // the 16-byte load stays inside the row
    const bool next_same_row = lane < 63 && k0 + MC_PER < cs.per_row && first + MC_PER < cs.total;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int e = 0; e < MC_PER; e++) rows[r][e] = 0.f;
    // ONE branch around all four rows: with the branch inside the row loop hipcc closes every row's region with its own
    // s_waitcnt vmcnt(0) - four dependent round trips per thread instead of four loads in flight
    if (wide) {
        // (rows are G floats apart: 4-byte aligned only, which global_load_dwordx4 accepts)
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const f4u q0 = *reinterpret_cast<const f4u *>(src[0]), q1 = *reinterpret_cast<const f4u *>(src[1]);
        const f4u q2 = *reinterpret_cast<const f4u *>(src[2]), q3 = *reinterpret_cast<const f4u *>(src[3]);
        rows[0][0] = q0.x; rows[0][1] = q0.y; rows[0][2] = q0.z; rows[0][3] = q0.w;
        rows[1][0] = q1.x; rows[1][1] = q1.y; rows[1][2] = q1.z; rows[1][3] = q1.w;
        rows[2][0] = q2.x; rows[2][1] = q2.y; rows[2][2] = q2.z; rows[2][3] = q2.w;
        rows[3][0] = q3.x; rows[3][1] = q3.y; rows[3][2] = q3.z; rows[3][3] = q3.w;
    } else if (live) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int e = 0; e < MC_PER; e++) rows[r][e] = src[r][min(e, G - 1 - k0)];
    }
    if (ROWWAVE) {
        // all lanes of the wave are live or none (whole rows); lane 0's (i, j) is the wave's
        const int wi = __builtin_amdgcn_readfirstlane(i), wj = __builtin_amdgcn_readfirstlane(j);
        const bool wlive = __builtin_amdgcn_readfirstlane(live ? 1 : 0) != 0;
        const float *pe = vol + (wlive ? (size_t)wi * gg + (size_t)wj * G + (G - 1) : 0);
        const float last[4] = {pe[0], pe[gg], pe[gg + G], pe[G]};        // a dead wave reads row (0, 0): in range, unused
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = __shfl_down(rows[r][0], 1, 64);
            rows[r][MC_PER] = lane == 63 ? last[r] : v;
        }
        return live;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float v = __shfl_down(rows[r][0], 1, 64);     // value k0 + MC_PER: the next lane's first one ...
        if (live && !next_same_row) v = src[r][min(MC_PER, G - 1 - k0)];     // ... unless that lane starts another row
        rows[r][MC_PER] = v;
    }
    return live;
}
// case index of cube q of the lane: bit c set when corner c is below iso (corner c = rows[c & 3][q + (c >> 2)])
__device__ __forceinline__ int case_of_rows(const float (&rows)[4][MC_PER + 1], int q, float iso) {
    int c = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        c |= (rows[r][q] < iso) ? (1 << r) : 0;
        c |= (rows[r][q + 1] < iso) ? (1 << (4 + r)) : 0;
    }
    return c;
}

// Count pass: a wave's 64 x MC_PER consecutive virtual cubes are one UNIT of the emit pass; its entry is
// (unit has triangles) << 32 | triangles, so one scan yields the triangle offsets and the list of non-empty units.
// The 256-entry triangle-count table is copied to LDS once per workgroup: the four look-ups per lane are ds_read_u8 instead
// of four more vector-memory gathers (the pass is bound by its vector-memory instructions - 4 row loads, 4 row-end loads and
// 4 table gathers per wave before round 3's last change - not by HBM).
template <bool ROWWAVE>
__global__ __launch_bounds__(MC_THREADS) void mc_count_kernel(const float *__restrict__ vol, int G,
                                                              float iso,
                                                              const uint8_t *__restrict__ tri_count,
                                                              unsigned long long *__restrict__ unit_sums, long long n_units) {
    __shared__ uint8_t tri_lds[256];
    static_assert(MC_THREADS == 256, "one table entry per thread");
    tri_lds[threadIdx.x] = tri_count[threadIdx.x];
    const CubeSpace cs = cube_space(G);
    const long long first = ((long long)blockIdx.x * MC_THREADS + threadIdx.x) * MC_PER;
    const int lane = threadIdx.x & 63;
    int i, j, k0;
    float rows[4][MC_PER + 1];
    const bool live = load_cube_rows<ROWWAVE>(vol, G, cs, first, lane, i, j, k0, rows);
    __syncthreads();
    int n = 0;
    if (live) {
#pragma unroll
        for (int q = 0; q < MC_PER; q++)
            if (k0 + q < cs.C) n += tri_lds[case_of_rows(rows, q, iso)];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    const long long unit = (long long)blockIdx.x * (MC_BLOCK / MC_THREADS) + (threadIdx.x >> 6);
    if (lane == 0 && unit < n_units) unit_sums[unit] = ((unsigned long long)(n > 0 ? 1 : 0) << 32) | (unsigned)n;
    if (blockIdx.x == 0 && threadIdx.x == 0) unit_sums[n_units] = 0;     // the scan turns it into the totals
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

// second scan launch of the count pass: offsets of the units (exclusive, both halves) and the list of non-empty ones
__global__ __launch_bounds__(256) void mc_scan_offsets_kernel(const unsigned long long *__restrict__ sums,
                                                              unsigned long long *__restrict__ offsets, long long n,
                                                              const unsigned long long *__restrict__ tile_tot, int tiles,
                                                              int *__restrict__ active_list, int *__restrict__ total_out,
                                                              int *__restrict__ active_out) {
    __shared__ unsigned long long lds[4];
    __shared__ unsigned long long offset;
    unsigned long long part = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) part += tile_tot[t];
    unsigned long long total;
    (void)block256_exclusive<unsigned long long>(part, lds, total);
    if (threadIdx.x == 0) offset = total;
    __syncthreads();
    const unsigned long long off = offset;
    const long long b = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; e++)
        if (b + e < n) {
            const unsigned long long x = offsets[b + e] + off;
            offsets[b + e] = x;
            if (sums[b + e] >> 32) active_list[x >> 32] = (int)(b + e);
        }
    if (blockIdx.x == tiles - 1 && threadIdx.x == 0) {
        const unsigned long long all = off + tile_tot[tiles - 1];
        *total_out = (int)(all & 0xffffffffull);
        *active_out = (int)(all >> 32);
    }
}
// first scan launch with separate input and output (the unit sums stay intact for the second one)
__global__ __launch_bounds__(256) void mc_scan_tiles_kernel(const unsigned long long *__restrict__ src,
                                                            unsigned long long *__restrict__ dst, long long n,
                                                            unsigned long long *__restrict__ tile_tot) {
    __shared__ unsigned long long lds[4];
    const long long b = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; e++) {
        v[e] = b + e < n ? src[b + e] : 0ull;
        sum += v[e];
    }
    unsigned long long total;
    unsigned long long run = block256_exclusive<unsigned long long>(sum, lds, total);
#pragma unroll
    for (int e = 0; e < SCAN_ITEMS; e++) {
        if (b + e < n) dst[b + e] = run;
        run += v[e];
    }
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = total;
}

// Emit pass (round 3): one WAVE per non-empty unit (the count pass's list), so the volume is read only where the
// surface is, with the count pass's 16-byte row loads; the unit's triangles are then spread over the lanes - lane t
// builds triangle t of the unit from the owner cube's corner values in LDS - instead of every owner lane looping
// over its own (most lanes of a unit own none).  Same vertex arithmetic, same order: bit-identical soup.
constexpr unsigned long long kEdgeAPacked = 0x321047540310ull;      // edge e -> low-coordinate corner, 4 bits each
constexpr unsigned kEdgeAxisPacked = 0xAA4444u;                      // edge e -> axis, 2 bits each
constexpr int EMIT_VALS = 4 * (MC_PER + 1) + 1;                      // floats per lane in LDS (+1: bank skew)

__global__ __launch_bounds__(MC_THREADS) void mc_emit_kernel(
    const float *__restrict__ vol, int G, float iso, const int8_t *__restrict__ tri_table,
    int table_stride, const uint8_t *__restrict__ tri_count, const unsigned long long *__restrict__ offsets,
    const int *__restrict__ active_list, const int *__restrict__ n_active, float scale, float offset,
    float *__restrict__ tris, int max_tris) {
    __shared__ float s_vals[MC_THREADS / 64][64 * EMIT_VALS];
    __shared__ unsigned short s_rec[MC_THREADS / 64][64 * MC_PER * 5];      // per triangle: lane << 5 | q << 3 | t
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long w = (long long)blockIdx.x * (MC_THREADS / 64) + wave;
    if (w >= *n_active) return;
    const int unit = active_list[w];
    const int base = (int)(offsets[unit] & 0xffffffffull);
    const CubeSpace cs = cube_space(G);
    const long long first = ((long long)unit * 64 + lane) * MC_PER;
    int i, j, k0;
    float rows[4][MC_PER + 1];
    const bool live = load_cube_rows(vol, G, cs, first, lane, i, j, k0, rows);
    int cases[MC_PER], n[MC_PER], mine = 0;
#pragma unroll
    for (int q = 0; q < MC_PER; q++) {
        cases[q] = 0;
        n[q] = 0;
        if (live && k0 + q < cs.C) {
            cases[q] = case_of_rows(rows, q, iso);
            n[q] = tri_count[cases[q]];
        }
        mine += n[q];
    }
    int incl = mine;                                    // wave-inclusive scan of the lanes' triangle counts
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    const int total = __shfl(incl, 63, 64);
    float *sv = &s_vals[wave][lane * EMIT_VALS];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int e = 0; e <= MC_PER; e++) sv[r * (MC_PER + 1) + e] = rows[r][e];
    int at = incl - mine;
#pragma unroll
    for (int q = 0; q < MC_PER; q++)
        for (int t = 0; t < n[q]; t++) s_rec[wave][at++] = (unsigned short)((lane << 5) | (q << 3) | t);
    // (one wave: LDS writes above are ordered before the reads below by the s_waitcnt the compiler places; no barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int tix = t0 + lane;
        if (tix >= total || base + tix >= max_tris) continue;
        const int rec = s_rec[wave][tix];
        const int owner = rec >> 5, q = (rec >> 3) & 3, t = rec & 7;
        const float *ov = &s_vals[wave][owner * EMIT_VALS];
        // the owner lane's cube: its case and position (all lanes of a unit share the row unless the unit wraps)
        const long long ofirst = ((long long)unit * 64 + owner) * MC_PER;
        int oi, oj, ok;
        virtual_ijk(cs, ofirst, oi, oj, ok);
        ok += q;
        int cse = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            cse |= (ov[r * (MC_PER + 1) + q] < iso) ? (1 << r) : 0;
            cse |= (ov[r * (MC_PER + 1) + q + 1] < iso) ? (1 << (4 + r)) : 0;
        }
        float *o = tris + (size_t)(base + tix) * 9;
#pragma unroll
        for (int v = 0; v < 3; v++) {
            const int e = tri_table[cse * table_stride + 3 * t + v];
            const int ca = (int)((kEdgeAPacked >> (4 * e)) & 15ull), axis = (int)((kEdgeAxisPacked >> (2 * e)) & 3u);
            const int cb = ca + (axis == 0 ? ((ca & 3) == 0 ? 1 : -1) : axis == 1 ? ((ca & 3) == 0 ? 3 : 1) : 4);
            const float fa = ov[(ca & 3) * (MC_PER + 1) + q + (ca >> 2)];
            const float fb = ov[(cb & 3) * (MC_PER + 1) + q + (cb >> 2)];
            const float tt = (iso - fa) / (fb - fa);
            const float px = (float)(oi + ((0x66 >> ca) & 1)) + (axis == 0 ? tt : 0.0f);
            const float py = (float)(oj + ((0xCC >> ca) & 1)) + (axis == 1 ? tt : 0.0f);
            const float pz = (float)(ok + ((0xF0 >> ca) & 1)) + (axis == 2 ? tt : 0.0f);
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

static inline long long mc_units(int G) { return (cube_space(G).total + MC_THREADS - 1) / MC_THREADS; }

// scratch of the count / emit passes, in 8-byte words: unit sums [nu + 1] | unit offsets [nu + 1] | scan tile totals |
// active-unit list (ints) | n_active (int)
struct McScratch {
    unsigned long long *sums, *offsets, *tile_tot;
    int *active, *n_active;
    size_t bytes;
};
static inline McScratch mc_scratch(void *base, int G) {
    const long long nu = mc_units(G);
    const int tiles = scan_tiles(nu + 1);
    unsigned long long *w = static_cast<unsigned long long *>(base);
    McScratch m;
    m.sums = w;
    m.offsets = w + nu + 1;
    m.tile_tot = m.offsets + nu + 1;
    m.active = reinterpret_cast<int *>(m.tile_tot + tiles + 1);
    m.n_active = m.active + nu + 1;
    m.bytes = (size_t)(2 * (nu + 1) + tiles + 1) * 8 + (size_t)(nu + 4) * 4;
    return m;
}

extern "C" size_t zs_mc_scratch_bytes(int G) {
    if (G < 2) return 0;
    return mc_scratch(nullptr, G).bytes;
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
    if (reinterpret_cast<uintptr_t>(scratch) & 7) {
        zs::set_err("zs_mc_count: scratch must be 8-byte aligned");
        return 0;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long cubes = cube_space(G).total;
    const long long nu = mc_units(G);
    const McScratch m = mc_scratch(scratch, G);
    if (cube_space(G).per_row == 64 * MC_PER) hipLaunchKernelGGL(mc_count_kernel<true>, dim3((unsigned)((cubes + MC_BLOCK - 1) / MC_BLOCK)), dim3(MC_THREADS), 0, s, vol, G, iso,
                       tri_count, m.sums, nu);
    else hipLaunchKernelGGL(mc_count_kernel<false>, dim3((unsigned)((cubes + MC_BLOCK - 1) / MC_BLOCK)), dim3(MC_THREADS), 0, s, vol, G, iso,
                       tri_count, m.sums, nu);
    // exclusive scan over the nu unit sums and one trailing zero: triangle offsets in the low halves, the index of every
    // non-empty unit among the non-empty ones in the high halves
    const int tiles = scan_tiles(nu + 1);
    hipLaunchKernelGGL(mc_scan_tiles_kernel, dim3(tiles), dim3(256), 0, s, static_cast<const unsigned long long *>(m.sums),
                       m.offsets, nu + 1, m.tile_tot);
    hipLaunchKernelGGL(mc_scan_offsets_kernel, dim3(tiles), dim3(256), 0, s, static_cast<const unsigned long long *>(m.sums),
                       m.offsets, nu + 1, static_cast<const unsigned long long *>(m.tile_tot), tiles, m.active, total, m.n_active);
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
    // one wave per non-empty unit; their number stays on the device (at most min(units, triangles))
    const McScratch m = mc_scratch(const_cast<void *>(scratch), G);
    const long long waves = mc_units(G) < (long long)n_tris ? mc_units(G) : (long long)n_tris;
    hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)((waves + MC_THREADS / 64 - 1) / (MC_THREADS / 64))), dim3(MC_THREADS), 0,
                       static_cast<hipStream_t>(stream), vol, G, iso, tri_table, table_stride, tri_count,
                       static_cast<const unsigned long long *>(m.offsets), static_cast<const int *>(m.active),
                       static_cast<const int *>(m.n_active), scale, offset, tris, n_tris);
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
