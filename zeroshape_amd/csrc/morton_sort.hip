// Spatial sorts of a point cloud for the culled nearest-neighbour scan of the pose search
// (csrc/pose_search.hip: pose_nn_soa_kernel).
//
// A cloud whose runs of 64 consecutive points are spatially compact stays so under any rigid rotation - which
// is what lets the pose search skip every (query, 64 candidates) block whose box is farther away than the
// nearest neighbour found so far (utils/eval_3D.py:140-170 evaluates every pair).  Two orders:
//   zs_morton_sort   along the Z-order curve of the bounding box (one sort; runs straddle the curve's jumps)
//   zs_str_sort      sort-tile-recursive packing (the R-tree bulk-loading order): sort by x, cut into slabs of
//                    whole leaves; inside a slab sort by y, cut into strips; inside a strip sort by z.  Three
//                    sorts, leaves of 64 points with near-cubic boxes: on the benchmark clouds the culled scan
//                    evaluates 12 % of the pairs with it against 22 % with the Z-order (CPU model of the kernel's
//                    decisions; a k-d split order reaches 10 % at eight sorts).  The default of the pose search.
//
// Morton path:
//   morton_bbox_kernel    one workgroup: min / max of the finite coordinates
//   morton_key_kernel     30-bit key (10 bits per axis, x lowest) of every point, value = its index;
//                         points with a non-finite coordinate sort to the end
//   rocprim::radix_sort_pairs   stable LSD radix sort -> the permutation is a pure function of the
//                         coordinates (equal keys keep their input order): summation orders that follow
//                         it are reproducible from call to call
//   morton_gather_kernel  sorted[i] = points[perm[i]], and per tile of `tile` consecutive sorted points
//                         its axis-aligned box (lo[3], hi[3]; non-finite coordinates ignored)
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int MS_THREADS = 256;

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// lo[3], hi[3] over the workgroup of the points fetch(i), i in [begin, end) -> out[6]; non-finite coordinates ignored
template <typename F>
__device__ __forceinline__ void block_box(F fetch, int begin, int end, float *out, float (*red)[MS_THREADS / 64]) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = begin + threadIdx.x; i < end; i += MS_THREADS) {
        float v[3];
        fetch(i, v);
#pragma unroll
        for (int a = 0; a < 3; a++)
            if (fabsf(v[a]) < INFINITY) {
                lo[a] = fminf(lo[a], v[a]);
                hi[a] = fmaxf(hi[a], v[a]);
            }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = wave_min(lo[a]);
        hi[a] = wave_max(hi[a]);
        if (lane == 0) {
            red[a][wave] = lo[a];
            red[3 + a][wave] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < MS_THREADS / 64; w++)
            v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        out[threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(MS_THREADS) void morton_bbox_kernel(const float *__restrict__ p, int n, float *bbox) {
    __shared__ float red[6][MS_THREADS / 64];
    block_box([&](int i, float *v) {
        v[0] = p[(size_t)i * 3];
        v[1] = p[(size_t)i * 3 + 1];
        v[2] = p[(size_t)i * 3 + 2];
    }, 0, n, bbox, red);
}

__device__ __forceinline__ unsigned spread10(unsigned v) {   // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(MS_THREADS) void morton_key_kernel(const float *__restrict__ p, int n,
                                                                const float *__restrict__ bbox, unsigned *keys,
                                                                unsigned *vals) {
    const int i = blockIdx.x * MS_THREADS + threadIdx.x;
    if (i >= n) return;
    unsigned key = 0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float v = p[(size_t)i * 3 + a], lo = bbox[a], ext = bbox[3 + a] - lo;
        ok = ok && fabsf(v) < INFINITY;
        const float t = ext > 0.f ? (v - lo) / ext : 0.f;
        const int c = min(1023, max(0, (int)(t * 1024.f)));
        key |= spread10((unsigned)c) << a;
    }
    keys[i] = ok ? key : 0x7fffffffu;
    vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(MS_THREADS) void morton_gather_kernel(const float *__restrict__ p, int n,
                                                                   const unsigned *__restrict__ perm,
                                                                   float *__restrict__ sorted, int *__restrict__ perm_out,
                                                                   float *__restrict__ boxes, int tile) {
    __shared__ float red[6][MS_THREADS / 64];
    const int begin = blockIdx.x * tile, end = min(n, begin + tile);
    float scratch6[6];
    block_box([&](int i, float *v) {
        const unsigned s = perm[i];
        v[0] = p[(size_t)s * 3];
        v[1] = p[(size_t)s * 3 + 1];
        v[2] = p[(size_t)s * 3 + 2];
        sorted[(size_t)i * 3] = v[0];
        sorted[(size_t)i * 3 + 1] = v[1];
        sorted[(size_t)i * 3 + 2] = v[2];
        if (perm_out) perm_out[i] = (int)s;
    }, begin, end, boxes ? boxes + (size_t)blockIdx.x * 6 : scratch6, red);
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t sort_temp_bytes(int n) {
    size_t bytes = 0;
    unsigned *none = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, none, none, none, none, (size_t)n, 0u, 31u);
    return bytes;
}

// ---- sort-tile-recursive order --------------------------------------------------------------------------- //
constexpr int STR_LEAF = 64;

struct StrPlan {       // the cut positions are a function of n alone
    int n, slab_points, max_strips;
};
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int ceil_cbrt(int v) {
    int r = 1;
    while (r * r * r < v) r++;
    return r;
}
__host__ __device__ inline int ceil_sqrt(int v) {
    int r = 1;
    while (r * r < v) r++;
    return r;
}
__host__ __device__ inline StrPlan str_plan(int n) {
    const int leaves = ceil_div(n, STR_LEAF), slabs = ceil_cbrt(leaves);
    const int per = ceil_div(leaves, slabs);
    return StrPlan{n, per * STR_LEAF, ceil_sqrt(per)};
}
// position -> slab (pass 1) or global strip id (pass 2): both in whole leaves
__host__ __device__ inline int str_group(const StrPlan &pl, int i, int pass) {
    const int slab = i / pl.slab_points;
    if (pass == 1) return slab;
    const int begin = slab * pl.slab_points, len = min(pl.slab_points, pl.n - begin);
    const int leaves = ceil_div(len, STR_LEAF), strips = ceil_sqrt(leaves), per = ceil_div(leaves, strips);
    return slab * pl.max_strips + (i - begin) / (per * STR_LEAF);
}

__device__ __forceinline__ unsigned ordered_bits(float v) {    // monotone float -> unsigned
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// pass 0: key = x; pass 1: (slab, y); pass 2: (strip, z).  A point with a non-finite coordinate sorts last in pass 0
// and keeps the end of its group afterwards.
__global__ __launch_bounds__(MS_THREADS) void str_key_kernel(const float *__restrict__ p, StrPlan pl, int pass,
                                                             unsigned long long *keys, unsigned *vals) {
    const int i = blockIdx.x * MS_THREADS + threadIdx.x;
    if (i >= pl.n) return;
    const float x = p[(size_t)i * 3], y = p[(size_t)i * 3 + 1], z = p[(size_t)i * 3 + 2];
    const bool ok = fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY;
    const float v = pass == 0 ? x : pass == 1 ? y : z;
    const unsigned lo = ok ? ordered_bits(v) : 0xffffffffu;
    const unsigned hi = pass == 0 ? 0u : (unsigned)str_group(pl, i, pass);
    keys[i] = ((unsigned long long)hi << 32) | lo;
    vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(MS_THREADS) void str_gather_kernel(const float *__restrict__ p, const int *__restrict__ perm_in,
                                                                int n, const unsigned *__restrict__ vals,
                                                                float *__restrict__ out, int *__restrict__ perm_out) {
    const int i = blockIdx.x * MS_THREADS + threadIdx.x;
    if (i >= n) return;
    const unsigned s = vals[i];
    out[(size_t)i * 3] = p[(size_t)s * 3];
    out[(size_t)i * 3 + 1] = p[(size_t)s * 3 + 1];
    out[(size_t)i * 3 + 2] = p[(size_t)s * 3 + 2];
    if (perm_out) perm_out[i] = perm_in ? perm_in[s] : (int)s;
}

size_t sort64_temp_bytes(int n) {
    size_t bytes = 0;
    unsigned long long *k = nullptr;
    unsigned *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, 64u);
    return bytes;
}

}  // namespace

extern "C" size_t zs_str_scratch_bytes(int n) {
    if (n <= 0) return 0;
    // keys in / out (8 B), values in / out, one intermediate cloud, two permutations, sort workspace
    return 2 * align256((size_t)n * 8) + 2 * align256((size_t)n * 4) + align256((size_t)n * 12) + 2 * align256((size_t)n * 4) +
           align256(sort64_temp_bytes(n));
}

extern "C" int zs_str_sort(const float *points, int n, float *sorted, int *perm, void *scratch, void *stream) {
    if (n < 0) {
        zs::set_err("zs_str_sort: negative size");
        return 0;
    }
    if (n == 0) return 1;
    if (!points || !sorted || !scratch || points == sorted) {
        zs::set_err("zs_str_sort: null or aliased pointer");
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(scratch);
    const size_t a8 = align256((size_t)n * 8), a4 = align256((size_t)n * 4), a12 = align256((size_t)n * 12);
    unsigned long long *keys_in = reinterpret_cast<unsigned long long *>(base);
    unsigned long long *keys_out = reinterpret_cast<unsigned long long *>(base + a8);
    unsigned *vals_in = reinterpret_cast<unsigned *>(base + 2 * a8);
    unsigned *vals_out = reinterpret_cast<unsigned *>(base + 2 * a8 + a4);
    float *mid = reinterpret_cast<float *>(base + 2 * a8 + 2 * a4);
    int *perm_a = reinterpret_cast<int *>(base + 2 * a8 + 2 * a4 + a12);
    int *perm_b = perm_a + a4 / 4;
    void *temp = base + 2 * a8 + 2 * a4 + a12 + 2 * a4;
    size_t temp_bytes = sort64_temp_bytes(n);
    const StrPlan pl = str_plan(n);
    const dim3 grid((n + MS_THREADS - 1) / MS_THREADS), block(MS_THREADS);
    // pass 0: points -> sorted (x); pass 1: sorted -> mid (slab, y); pass 2: mid -> sorted (strip, z)
    const float *src[3] = {points, sorted, mid};
    float *dst[3] = {sorted, mid, sorted};
    const int *pin[3] = {nullptr, perm_a, perm_b};
    int *pout[3] = {perm_a, perm_b, perm};
    for (int pass = 0; pass < 3; pass++) {
        hipLaunchKernelGGL(str_key_kernel, grid, block, 0, st, src[pass], pl, pass, keys_in, vals_in);
        if (rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 64u, st) != hipSuccess) {
            zs::set_err("zs_str_sort: rocprim::radix_sort_pairs failed");
            return 0;
        }
        hipLaunchKernelGGL(str_gather_kernel, grid, block, 0, st, src[pass], pin[pass], n, static_cast<const unsigned *>(vals_out),
                           dst[pass], pass == 2 ? perm : pout[pass]);
    }
    return zs::check_launch("zs_str_sort") ? 1 : 0;
}

namespace {
}  // namespace

extern "C" size_t zs_morton_scratch_bytes(int n) {
    if (n <= 0) return 0;
    return 256 + 4 * align256((size_t)n * 4) + align256(sort_temp_bytes(n));
}

extern "C" int zs_morton_sort(const float *points, int n, float *sorted, int *perm, float *tile_boxes, int tile,
                              void *scratch, void *stream) {
    if (n < 0 || (tile_boxes && tile <= 0)) {
        zs::set_err("zs_morton_sort: bad size (n=%d tile=%d)", n, tile);
        return 0;
    }
    if (n == 0) return 1;
    if (!points || !sorted || !scratch || points == sorted) {
        zs::set_err("zs_morton_sort: null or aliased pointer");
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(scratch);
    float *bbox = reinterpret_cast<float *>(base);
    const size_t arr = align256((size_t)n * 4);
    unsigned *keys_in = reinterpret_cast<unsigned *>(base + 256);
    unsigned *keys_out = reinterpret_cast<unsigned *>(base + 256 + arr);
    unsigned *vals_in = reinterpret_cast<unsigned *>(base + 256 + 2 * arr);
    unsigned *vals_out = reinterpret_cast<unsigned *>(base + 256 + 3 * arr);
    void *temp = base + 256 + 4 * arr;
    size_t temp_bytes = sort_temp_bytes(n);
    hipLaunchKernelGGL(morton_bbox_kernel, dim3(1), dim3(MS_THREADS), 0, st, points, n, bbox);
    hipLaunchKernelGGL(morton_key_kernel, dim3((n + MS_THREADS - 1) / MS_THREADS), dim3(MS_THREADS), 0, st, points, n, bbox,
                       keys_in, vals_in);
    if (rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 31u, st) !=
        hipSuccess) {
        zs::set_err("zs_morton_sort: rocprim::radix_sort_pairs failed");
        return 0;
    }
    const int t = tile_boxes ? tile : 1024;
    hipLaunchKernelGGL(morton_gather_kernel, dim3((n + t - 1) / t), dim3(MS_THREADS), 0, st, points, n, vals_out, sorted, perm,
                       tile_boxes, t);
    return zs::check_launch("zs_morton_sort") ? 1 : 0;
}
