// Fused brute-force pose search for MI355X (gfx950): one batch of rotations = three launches and
// no tensor of rotated clouds, distances or indices.
//
// Replaces, per batch of rotations, the reference's (utils/eval_3D.py:149-168)
//     rotate (bmm) -> normalize_pc (:93-102) -> chamfer_distance (:265-269, the NN kernels of
//     external/chamfer3D/chamfer3D.cu:12-134 + sqrt) -> compute_fscore (:215-231) -> means ->
//     strict-minimum update of the running best
// and, stand-alone, normalize_pc and compute_fscore themselves (zs_normalize_pc / zs_fscore).
//
//   pose_stats_kernel   one workgroup per rotation: R p for every prediction point (fmaf chain
//                       in k order), mean (double sums, rounded once), x / y extents of the
//                       zero-meaned cloud -> the normalize_pc scale; 16 floats per rotation.
//   pose_nn_kernel      the Chamfer scan of csrc/chamfer.hip (same arithmetic: d = fma(dz,dz,
//                       fma(dy,dy,dx*dx)), lowest index wins ties) with the rotated + normalised
//                       prediction generated on the fly - as queries (direction 0) and while
//                       staging candidates into LDS (direction 1) - and an epilogue that reduces
//                       sqrt(d) and the six F-score counters per workgroup in a FIXED order
//                       (bit-reproducible: the same rotation gives the same bits in any batch).
//   pose_nn_cull_kernel the same scan over MORTON-SORTED clouds (csrc/morton_sort.hip) that skips the
//                       (run of 64 queries, sub-tile of 64 candidates) blocks whose bounding boxes are
//                       farther apart than the worst nearest neighbour the run has found so far, tiles
//                       visited nearest first: the same minima - a skipped candidate is provably not
//                       closer - from a fraction of the pairs (round 3; the default).
//   pose_finish_kernel  per batch: sums the partials in order, acc / comp / cd / F-score per
//                       rotation, lexicographic (cd, rotation index) winner, merged into the running
//                       best record on the device.
// Every kernel first compares the batch's smallest lower bound (csrc/bf_prune.hip) with the
// running best and returns if the batch cannot win: the host enqueues all batches without a
// single synchronisation and reads the record once at the end.
#include "zs_common.h"
#include "zs_point_grid.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int PS_THREADS = 256;
constexpr int PS_Q = 2;                       // queries per lane
constexpr int PS_TILE = 1024;                 // candidates per LDS tile
constexpr int PS_STRIDE = PS_TILE + 4;
constexpr int PS_STAT = 16;                   // floats per rotation: R[9], mean[3], denom, pad
constexpr int PS_PART = 8;                    // per workgroup: sum sqrt(d), 6 counts, pad
constexpr int PS_BEST = 16;                   // cd, index (int bits), acc, comp, f[6], evaluated, scanned (ints),
                                              // [12] external bound: the best cd another rank has already found

struct Xform {  // normalize_pc(R p): ((R p) - mean) / denom
    float r[9], mu[3], den;
    __device__ __forceinline__ void load(const float *s) {
#pragma unroll
        for (int i = 0; i < 9; i++) r[i] = s[i];
        mu[0] = s[9]; mu[1] = s[10]; mu[2] = s[11];
        den = s[12];
    }
    __device__ __forceinline__ void rotate(float x, float y, float z, float &ox, float &oy, float &oz) const {
        ox = fmaf(r[2], z, fmaf(r[1], y, r[0] * x));
        oy = fmaf(r[5], z, fmaf(r[4], y, r[3] * x));
        oz = fmaf(r[8], z, fmaf(r[7], y, r[6] * x));
    }
    __device__ __forceinline__ void apply(float x, float y, float z, float &ox, float &oy, float &oz) const {
        rotate(x, y, z, ox, oy, oz);
        ox = (ox - mu[0]) / den;
        oy = (oy - mu[1]) / den;
        oz = (oz - mu[2]) / den;
    }
};

// the batch cannot beat the running best (margin: the roundings of the bound and of the exact path;
// zeroshape_amd/utils/eval_3D.py)
// best[12] (+inf unless a multi-GPU search stored the other ranks' best there, zeroshape_amd/utils/eval_3D.py) tightens
// every pruning decision: a rotation strictly worse than ANY rank's best cannot be the global lexicographic minimum.
// It never enters the record this rank reports (pose_finish_kernel compares with best[0] alone).
__device__ __forceinline__ float prune_bound(const float *best) { return fminf(best[0], best[12]); }
__device__ __forceinline__ bool batch_pruned(const float *lower_bound, const float *best) {
    return lower_bound && lower_bound[0] * (1.0f - 1e-3f) - 1e-6f > prune_bound(best);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- statistics of R p: mean and the normalize_pc scale ----------------------------------- //
// rotations == nullptr: identity (plain normalize_pc of cloud[b]); order == nullptr: rotation b
__global__ __launch_bounds__(PS_THREADS) void pose_stats_kernel(
    const float *__restrict__ cloud, size_t cloud_stride, int n, const float *__restrict__ rotations,
    const int *__restrict__ order, float *__restrict__ stats, const float *__restrict__ lower_bound,
    const float *__restrict__ best) {
    if (batch_pruned(lower_bound, best)) return;
    __shared__ double sd[3][PS_THREADS / 64];
    __shared__ float sf[4][PS_THREADS / 64];
    const int b = blockIdx.x;
    const float *p = cloud + (size_t)b * cloud_stride;
    Xform t;
    if (rotations) {
        const float *r = rotations + (size_t)(order ? order[b] : b) * 9;
#pragma unroll
        for (int i = 0; i < 9; i++) t.r[i] = r[i];
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) t.r[i] = (i % 4 == 0) ? 1.f : 0.f;
    }
    double sx = 0, sy = 0, sz = 0;
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int i = threadIdx.x; i < n; i += PS_THREADS) {
        float x = p[(size_t)i * 3], y = p[(size_t)i * 3 + 1], z = p[(size_t)i * 3 + 2];
        if (rotations) {
            float ox, oy, oz;
            t.rotate(x, y, z, ox, oy, oz);
            x = ox; y = oy; z = oz;
        }
        sx += x; sy += y; sz += z;
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
        ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
    }
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, o, 64)); xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
        ymin = fminf(ymin, __shfl_xor(ymin, o, 64)); ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sd[0][wave] = sx; sd[1][wave] = sy; sd[2][wave] = sz;
        sf[0][wave] = xmin; sf[1][wave] = xmax; sf[2][wave] = ymin; sf[3][wave] = ymax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < PS_THREADS / 64; w++) {
            sx += sd[0][w]; sy += sd[1][w]; sz += sd[2][w];
            xmin = fminf(xmin, sf[0][w]); xmax = fmaxf(xmax, sf[1][w]);
            ymin = fminf(ymin, sf[2][w]); ymax = fmaxf(ymax, sf[3][w]);
        }
        const float mx = (float)(sx / n), my = (float)(sy / n), mz = (float)(sz / n);
        // extents of the zero-meaned coordinates, as the reference forms them (:97-100)
        const float lx = (xmax - mx) - (xmin - mx), ly = (ymax - my) - (ymin - my);
        float *s = stats + (size_t)b * PS_STAT;
#pragma unroll
        for (int i = 0; i < 9; i++) s[i] = t.r[i];
        s[9] = mx; s[10] = my; s[11] = mz;
        s[12] = fmaxf(lx, ly) + 1.e-7f;
        s[13] = s[14] = s[15] = 0.f;
    }
}

// out[b][i] = normalize_pc(R_b cloud[b])[i]
__global__ __launch_bounds__(PS_THREADS) void pose_apply_kernel(const float *__restrict__ cloud, size_t cloud_stride,
                                                                int n, const float *__restrict__ stats,
                                                                float *__restrict__ out) {
    const int b = blockIdx.y;
    Xform t;
    t.load(stats + (size_t)b * PS_STAT);
    const float *p = cloud + (size_t)b * cloud_stride;
    float *o = out + (size_t)b * n * 3;
    for (int i = blockIdx.x * PS_THREADS + threadIdx.x; i < n; i += gridDim.x * PS_THREADS) {
        float x, y, z;
        t.apply(p[(size_t)i * 3], p[(size_t)i * 3 + 1], p[(size_t)i * 3 + 2], x, y, z);
        o[(size_t)i * 3] = x; o[(size_t)i * 3 + 1] = y; o[(size_t)i * 3 + 2] = z;
    }
}

// ---- nearest neighbours both ways + per-workgroup metric partials ------------------------- //
// Staged drop with a device-side MERGE (round 4).  The query blocks of a batch are scanned in stages with a pose_kill_kernel
// between them; when nothing can be dropped (an unalignable ground truth) the stages are pure cost - small launches that leave
// CUs idle (45.6 ms staged vs 37 ms unstaged over the whole sphere).  The host cannot look (every batch is enqueued up front),
// so the first kill kernel decides: if at least 3/4 of the batch survived the first stage it sets dead[PS_THREADS] = 1, the
// SECOND stage's launch - whose grid always spans all remaining blocks - then scans them all, and the later stages' launches
// return at once.  `stage` = (mode << 24) | end of the stage's own block range; mode 0: plain, 1: second stage, 2: later stage.
// Which launch forms a partial sum does not change the sum: records are bit-identical either way.
constexpr int PS_MERGE_SLOT = PS_THREADS;       // index of the merge flag behind the per-rotation dead flags
constexpr int PS_DEAD_INTS = PS_THREADS + 64;
__device__ __forceinline__ bool stage_skips(int stage, int bx, const int *__restrict__ dead) {
    const int mode = (stage >> 24) & 3, end = stage & 0xffffff;
    if (mode == 0 || !dead) return false;
    const bool merged = dead[PS_MERGE_SLOT] != 0;
    return mode == 1 ? (bx >= end && !merged) : merged;
}

// grid (ceil(max(n, m) / (256 Q)), rotations in the batch, 2)
// The launch covers query blocks x_begin .. x_begin + gridDim.x - 1 of blocks_x; `dead` (nullable): rotations that the
// probe (block 0, see zs_pose_search_batch) has already shown to lose - they are skipped.
__global__ __launch_bounds__(PS_THREADS) void pose_nn_kernel(
    const float *__restrict__ pred, int n, const float *__restrict__ gt, int m,
    const float *__restrict__ stats, const float *__restrict__ thresholds, float *__restrict__ partial,
    const float *__restrict__ lower_bound, const float *__restrict__ best, int x_begin, int blocks_x,
    const int *__restrict__ dead, int stage) {
    if (batch_pruned(lower_bound, best)) return;
    __shared__ __attribute__((aligned(16))) float tile[3 * PS_STRIDE];
    __shared__ float red[PS_PART][PS_THREADS / 64];
    const int dir = blockIdx.z, rot = blockIdx.y, bx = x_begin + blockIdx.x;
    if (stage_skips(stage, bx, dead)) return;
    if (dead && dead[rot]) return;
    const int nq = dir == 0 ? n : m;      // queries
    const int nc = dir == 0 ? m : n;      // candidates
    const int q_base = bx * (PS_THREADS * PS_Q);
    if (q_base >= nq) return;
    Xform t;
    t.load(stats + (size_t)rot * PS_STAT);

    float qx[PS_Q], qy[PS_Q], qz[PS_Q], bestd[PS_Q];
#pragma unroll
    for (int q = 0; q < PS_Q; q++) {
        int j = q_base + q * PS_THREADS + threadIdx.x;
        j = j < nq ? j : nq - 1;
        if (dir == 0) {
            t.apply(pred[(size_t)j * 3], pred[(size_t)j * 3 + 1], pred[(size_t)j * 3 + 2], qx[q], qy[q], qz[q]);
        } else {
            qx[q] = gt[(size_t)j * 3]; qy[q] = gt[(size_t)j * 3 + 1]; qz[q] = gt[(size_t)j * 3 + 2];
        }
        bestd[q] = INFINITY;
    }
    for (int k0 = 0; k0 < nc; k0 += PS_TILE) {
        const int cnt = min(PS_TILE, nc - k0);
        const int cnt4 = (cnt + 3) & ~3;
        __syncthreads();
        for (int c = threadIdx.x; c < cnt4; c += PS_THREADS) {
            float x = INFINITY, y = INFINITY, z = INFINITY;
            if (c < cnt) {
                const size_t o = (size_t)(k0 + c) * 3;
                if (dir == 0) {
                    x = gt[o]; y = gt[o + 1]; z = gt[o + 2];
                } else {
                    t.apply(pred[o], pred[o + 1], pred[o + 2], x, y, z);
                }
            }
            tile[c] = x;
            tile[PS_STRIDE + c] = y;
            tile[2 * PS_STRIDE + c] = z;
        }
        __syncthreads();
        float4 cx = *reinterpret_cast<const float4 *>(&tile[0]);
        float4 cy = *reinterpret_cast<const float4 *>(&tile[PS_STRIDE]);
        float4 cz = *reinterpret_cast<const float4 *>(&tile[2 * PS_STRIDE]);
        for (int k = 0; k < cnt4; k += 4) {
            const float4 nx = *reinterpret_cast<const float4 *>(&tile[k + 4]);
            const float4 ny = *reinterpret_cast<const float4 *>(&tile[PS_STRIDE + k + 4]);
            const float4 nz = *reinterpret_cast<const float4 *>(&tile[2 * PS_STRIDE + k + 4]);
            const float ax[4] = {cx.x, cx.y, cx.z, cx.w};
            const float ay[4] = {cy.x, cy.y, cy.z, cy.w};
            const float az[4] = {cz.x, cz.y, cz.z, cz.w};
            // only the minimum is needed here (no argmin): min is order-independent, so no
            // compare/select chain at all
#pragma unroll
            for (int q = 0; q < PS_Q; q++) {
                float d[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float dx = ax[c] - qx[q];
                    const float dy = ay[c] - qy[q];
                    const float dz = az[c] - qz[q];
                    d[c] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                }
                bestd[q] = fminf(bestd[q], fminf(fminf(d[0], d[1]), fminf(d[2], d[3])));
            }
            cx = nx; cy = ny; cz = nz;
        }
    }
    // epilogue: sum of sqrt(d) and the threshold counters of this workgroup's queries, fixed order
    float v[PS_PART];
#pragma unroll
    for (int i = 0; i < PS_PART; i++) v[i] = 0.f;
    float thr[6];
#pragma unroll
    for (int i = 0; i < 6; i++) thr[i] = thresholds[i];
#pragma unroll
    for (int q = 0; q < PS_Q; q++) {
        const int j = q_base + q * PS_THREADS + threadIdx.x;
        if (j < nq) {
            const float s = sqrtf(bestd[q]);
            v[0] += s;
#pragma unroll
            for (int i = 0; i < 6; i++) v[1 + i] += s < thr[i] ? 1.f : 0.f;   // counts <= 512: exact in fp32
        }
    }
#pragma unroll
    for (int i = 0; i < 7; i++) v[i] = wave_sum(v[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 7; i++) red[i][wave] = v[i];
    __syncthreads();
    if (threadIdx.x < 7) {
        float s = red[threadIdx.x][0];
        for (int w = 1; w < PS_THREADS / 64; w++) s += red[threadIdx.x][w];
        partial[(((size_t)rot * 2 + dir) * blocks_x + bx) * PS_PART + threadIdx.x] = s;
    }
}

// ---- the same scan with box culling (round 3) --------------------------------------------- //
// Both clouds arrive Morton-sorted in their own frames (zs_morton_sort): 64 consecutive points ("sub-tile") are a
// compact patch, also after the rotation.  pose_pack_kernel writes a cloud - the ground truth once per search, the
// rotated + normalised prediction once per rotation of the batch - as a PACK:
//     [sub-tile][x | y | z][64] coordinates (+inf padded)   |   [sub-tile] lo[3], hi[3] exact box   |   [tile of 16] box
// pose_nn_soa_kernel keeps pose_nn_kernel's grid, query mapping and epilogue, but its four waves walk the candidates
// independently, without LDS and without barriers:
//   * a candidate sub-tile is UNIFORM data for the wave: its box and its 192 coordinates come through the scalar
//     cache into SGPRs (s_load_dwordx8) and enter the distance arithmetic as scalar operands - the role the LDS
//     broadcast reads play in pose_nn_kernel;
//   * tiles are visited in order of increasing box distance from the wave's 128 queries, and the walk ends at the
//     first tile that is farther than every query's current nearest neighbour;
//   * inside a tile, lane l first measures sub-tile l against the boxes of the wave's two runs of 64 queries; the
//     sub-tiles that can still matter to a run are taken nearest first, each tested per QUERY - point-to-box distance
//     against the lane's own running minimum - and scanned only if some lane still needs it.
// d = fma(dz,dz,fma(dy,dy,dx*dx)) is evaluated exactly as in pose_nn_kernel for every pair that is not skipped, and a
// skipped candidate has d >= (point-to-box distance)^2 (1 - 8 * 2^-24) > the query's running minimum (PS_SKIP leaves
// a 16x margin for the roundings of both sides) - so every query ends with the same minimum, bit for bit, and the
// epilogue sums them in the same fixed order: same record as the all-pairs scan of the same sorted clouds.
constexpr int PS_SUB = 64;
constexpr int PS_NSUB = PS_TILE / PS_SUB;            // 16 sub-tiles per tile
constexpr int PS_SUB_FLOATS = 3 * PS_SUB;
constexpr float PS_SKIP = 1.0f - 0x1p-20f;
static_assert(PS_Q == 2 && PS_NSUB == 16, "pose_nn_soa_kernel: two query runs per wave, 16 sub-tiles per tile");

__host__ __device__ inline int pack_subs(int n) { return (n + PS_TILE - 1) / PS_TILE * PS_NSUB; }
__host__ __device__ inline size_t pack_floats(int n) {
    const size_t ns = (size_t)pack_subs(n);
    return ns * PS_SUB_FLOATS + ns * 8 + ns / PS_NSUB * 8;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// wave-wide min / max at VALU speed: quad swaps, half-row and row mirrors, then the four row results (wave-uniform).
// The DPP modifier sits on the v_min / v_max itself (inline asm: through fminf(v, dpp(v)) hipcc emits v_mov_dpp + two
// canonicalising v_max + v_min per step); "s_nop 1" covers the VALU-write -> DPP-read hazard of the chained steps.  Inputs here
// are box gaps and running minima - never NaN - so v_min / v_max return exactly what fminf / fmaxf do.
#define ZS_DPP_STEP(OP, CTRL) asm volatile("s_nop 1\n\t" OP " %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); v = r
__device__ __forceinline__ float wave_minf(float v) {
    float r;
    ZS_DPP_STEP("v_min_f32_dpp", "quad_perm:[1,0,3,2]");
    ZS_DPP_STEP("v_min_f32_dpp", "quad_perm:[2,3,0,1]");
    ZS_DPP_STEP("v_min_f32_dpp", "row_half_mirror");
    ZS_DPP_STEP("v_min_f32_dpp", "row_mirror");
    return fminf(fminf(lane_f(v, 0), lane_f(v, 16)), fminf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ float wave_maxf(float v) {
    float r;
    ZS_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2]");
    ZS_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1]");
    ZS_DPP_STEP("v_max_f32_dpp", "row_half_mirror");
    ZS_DPP_STEP("v_max_f32_dpp", "row_mirror");
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
#undef ZS_DPP_STEP
// squared distance between two axis-aligned boxes (0 when they overlap; +inf when one is empty: lo = +inf, hi = -inf)
__device__ __forceinline__ float box_gap2(const float *alo, const float *ahi, const float *blo, const float *bhi) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float g = fmaxf(0.f, fmaxf(alo[a] - bhi[a], blo[a] - ahi[a]));
        s = fmaf(g, g, s);
    }
    return s;
}
__device__ __forceinline__ float point_gap2(float x, float y, float z, const float *lo, const float *hi) {
    const float gx = fmaxf(0.f, fmaxf(lo[0] - x, x - hi[0]));
    const float gy = fmaxf(0.f, fmaxf(lo[1] - y, y - hi[1]));
    const float gz = fmaxf(0.f, fmaxf(lo[2] - z, z - hi[2]));
    return fmaf(gz, gz, fmaf(gy, gy, gx * gx));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// point_gap2 of the lane's two queries at once: v_pk_add / v_pk_mul / v_pk_fma on (query 0, query 1) pairs, the box bounds
// broadcast; per lane the operations and their order are point_gap2's, so the two results carry the same bits
__device__ __forceinline__ f32x2 point_gap2_pair(const float (&qx)[PS_Q], const float (&qy)[PS_Q], const float (&qz)[PS_Q],
                                                 const float *lo, const float *hi) {
    const f32x2 x = {qx[0], qx[1]}, y = {qy[0], qy[1]}, z = {qz[0], qz[1]};
    const f32x2 ax = f32x2{lo[0], lo[0]} - x, bx = x - f32x2{hi[0], hi[0]};
    const f32x2 ay = f32x2{lo[1], lo[1]} - y, by = y - f32x2{hi[1], hi[1]};
    const f32x2 az = f32x2{lo[2], lo[2]} - z, bz = z - f32x2{hi[2], hi[2]};
    const f32x2 gx = {fmaxf(0.f, fmaxf(ax.x, bx.x)), fmaxf(0.f, fmaxf(ax.y, bx.y))};
    const f32x2 gy = {fmaxf(0.f, fmaxf(ay.x, by.x)), fmaxf(0.f, fmaxf(ay.y, by.y))};
    const f32x2 gz = {fmaxf(0.f, fmaxf(az.x, bz.x)), fmaxf(0.f, fmaxf(az.y, bz.y))};
    return __builtin_elementwise_fma(gz, gz, __builtin_elementwise_fma(gy, gy, gx * gx));
}

// grid (tiles of the cloud, clouds): cloud b = normalize_pc(R_b src) (stats) or src itself (stats == nullptr) -> pack b
__global__ __launch_bounds__(PS_TILE) void pose_pack_kernel(const float *__restrict__ src, int n,
                                                            const float *__restrict__ stats, float *__restrict__ packs,
                                                            size_t pack_stride, const float *__restrict__ lower_bound,
                                                            const float *__restrict__ best) {
    if (batch_pruned(lower_bound, best)) return;
    __shared__ float s_box[PS_NSUB][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ns = pack_subs(n), sub = blockIdx.x * PS_NSUB + wave;
    float *pack = packs + (size_t)blockIdx.y * pack_stride;
    const int i = sub * PS_SUB + lane;
    const bool in = i < n;
    float x = INFINITY, y = INFINITY, z = INFINITY;
    if (in) {
        x = src[(size_t)i * 3]; y = src[(size_t)i * 3 + 1]; z = src[(size_t)i * 3 + 2];
        if (stats) {
            Xform t;
            t.load(stats + (size_t)blockIdx.y * PS_STAT);
            float ox, oy, oz;
            t.apply(x, y, z, ox, oy, oz);
            x = ox; y = oy; z = oz;
        }
    }
    float *c = pack + (size_t)sub * PS_SUB_FLOATS;
    c[lane] = x;
    c[PS_SUB + lane] = y;
    c[2 * PS_SUB + lane] = z;
    // exact box of the finite coordinates (fminf / fmaxf drop NaN; infinities and the padding are masked)
    const bool fx = in && fabsf(x) < INFINITY, fy = in && fabsf(y) < INFINITY, fz = in && fabsf(z) < INFINITY;
    const float b[6] = {wave_minf(fx ? x : INFINITY), wave_minf(fy ? y : INFINITY), wave_minf(fz ? z : INFINITY),
                        wave_maxf(fx ? x : -INFINITY), wave_maxf(fy ? y : -INFINITY), wave_maxf(fz ? z : -INFINITY)};
    if (lane < 8) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) v = lane == k ? b[k] : v;
        pack[(size_t)ns * PS_SUB_FLOATS + (size_t)sub * 8 + lane] = v;
        if (lane < 6) s_box[wave][lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        float v = 0.f;
        if (threadIdx.x < 6) {
            v = s_box[0][threadIdx.x];
            for (int w = 1; w < PS_NSUB; w++) v = threadIdx.x < 3 ? fminf(v, s_box[w][threadIdx.x]) : fmaxf(v, s_box[w][threadIdx.x]);
        }
        pack[(size_t)ns * (PS_SUB_FLOATS + 8) + (size_t)blockIdx.x * 8 + threadIdx.x] = v;
    }
}

#ifdef ZS_POSE_COUNT   // measurement build only (tools/pose_pairs.py): (run of 64 queries) x (64 candidates) blocks
__device__ unsigned long long g_pose_blocks[2];   // [0] evaluated, [1] all
#endif

// 64 candidates of one sub-tile (uniform address -> scalar loads) against this lane's queries.  Two buffers of eight
// candidates in ping-pong: the coordinates of the next eight are requested before the current eight are used.  hipcc
// merges and sinks plain loads next to their use (one s_load_dwordx16 triple per 16 candidates, waited for at once), so
// the loads and their wait are asm: scalar loads return out of order, only lgkmcnt(0) can wait for them, and with one
// chunk in flight that wait covers exactly the chunk needed next.  The wait statement "returns" the chunk's registers, so
// every consumer is ordered behind it.  (No LDS operation is outstanding inside the walk; all loads are drained on return.)
typedef float f32x8 __attribute__((ext_vector_type(8)));
struct Chunk8 {
    f32x8 x, y, z;
    __device__ __forceinline__ void request(const float *c) {
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x100\n\ts_load_dwordx8 %2, %3, 0x200"
                     : "=&s"(x), "=&s"(y), "=&s"(z) : "s"(c) : "memory");
    }
    // the same, pinned IN FRONT of the arithmetic on `held`: the statement "returns" held's registers too, so hipcc cannot
    // sink the request below the distance code that overlaps its latency
    __device__ __forceinline__ void request_before(const float *c, Chunk8 &held) {
        asm volatile("s_load_dwordx8 %0, %6, 0x0\n\ts_load_dwordx8 %1, %6, 0x100\n\ts_load_dwordx8 %2, %6, 0x200"
                     : "=&s"(x), "=&s"(y), "=&s"(z), "+s"(held.x), "+s"(held.y), "+s"(held.z) : "s"(c) : "memory");
    }
    __device__ __forceinline__ void arrived() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x), "+s"(y), "+s"(z) : : "memory"); }
};
static_assert(PS_SUB * sizeof(float) == 0x100, "Chunk8::request addresses the y / z rows as immediates");
template <bool D0, bool D1>
__device__ __forceinline__ void scan_chunk(const Chunk8 &a, const float (&qx)[PS_Q], const float (&qy)[PS_Q],
                                           const float (&qz)[PS_Q], float (&bestd)[PS_Q]) {
    // Two candidates per instruction (round 4): the packed fp32 forms v_pk_add / v_pk_mul / v_pk_fma take an SGPR PAIR - two
    // consecutive candidates of the chunk - as one operand and the lane's query coordinate broadcast by op_sel, so a pair of
    // distances costs 3 + 1 + 2 packed instructions instead of 12.  Lane by lane they are the same IEEE operations in the same
    // order (c + (-q) == c - q exactly; v_pk_fma_f32 is a fused multiply-add): every d, hence every minimum, keeps its bits.
    // 6.4 -> 3.8 VALU instructions per pair; the culled scan was at 91 % of the VALU issue rate (profiles/r03_eval_sq_*).
#pragma unroll
    for (int h = 0; h < 8; h += 4)
#pragma unroll
        for (int q = 0; q < PS_Q; q++) {
            if ((q == 0 && !D0) || (q == 1 && !D1)) continue;
            const f32x2 qx2 = {qx[q], qx[q]}, qy2 = {qy[q], qy[q]}, qz2 = {qz[q], qz[q]};
            f32x2 d[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const f32x2 cx = {a.x[h + 2 * e], a.x[h + 2 * e + 1]}, cy = {a.y[h + 2 * e], a.y[h + 2 * e + 1]},
                            cz = {a.z[h + 2 * e], a.z[h + 2 * e + 1]};
                const f32x2 dx = cx - qx2, dy = cy - qy2, dz = cz - qz2;
                d[e] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            }
            bestd[q] = fminf(bestd[q], fminf(fminf(d[0].x, d[0].y), fminf(d[1].x, d[1].y)));
        }
}
template <bool D0, bool D1>
__device__ __forceinline__ void scan_subtile(const float *__restrict__ c, const float (&qx)[PS_Q], const float (&qy)[PS_Q],
                                             const float (&qz)[PS_Q], float (&bestd)[PS_Q]) {
    Chunk8 a, b;
    a.request(c);
    a.arrived();
#ifdef ZS_POSE_EXP_NOLOAD       // timing experiment (wrong results): the arithmetic of a sub-tile on its first chunk only
#pragma unroll
    for (int k = 0; k < PS_SUB; k += 8) {
        scan_chunk<D0, D1>(a, qx, qy, qz, bestd);
        asm volatile("" : "+s"(a.x), "+s"(a.y), "+s"(a.z));
    }
    return;
#endif
#pragma unroll
    for (int k = 0; k < PS_SUB; k += 16) {
        b.request_before(c + k + 8, a);
        scan_chunk<D0, D1>(a, qx, qy, qz, bestd);
        b.arrived();
        if (k + 16 < PS_SUB) a.request_before(c + k + 16, b);
        scan_chunk<D0, D1>(b, qx, qy, qz, bestd);
        if (k + 16 < PS_SUB) a.arrived();
    }
}

// grid and partial layout of pose_nn_kernel.  pred_packs: one pack per rotation of the batch (pose_pack_kernel);
// gt_pack: the ground truth's.  CULL = false scans every sub-tile in index order (the all-pairs scan on packs).
template <bool CULL>
__global__ __launch_bounds__(PS_THREADS) void pose_nn_soa_kernel(
    const float *__restrict__ pred_packs, size_t pack_stride, int n, const float *__restrict__ gt_pack, int m,
    const float *__restrict__ thresholds, float *__restrict__ partial, const float *__restrict__ lower_bound,
    const float *__restrict__ best, int x_begin, int blocks_x, const int *__restrict__ dead, int stage) {
    if (batch_pruned(lower_bound, best)) return;
    __shared__ float red[PS_PART][PS_THREADS / 64];
    const int dir = blockIdx.z, rot = blockIdx.y, bx = x_begin + blockIdx.x;
    if (stage_skips(stage, bx, dead)) return;
    if (dead && dead[rot]) return;
    const int nq = dir == 0 ? n : m;      // queries
    const int nc = dir == 0 ? m : n;      // candidates
    const int q_base = bx * (PS_THREADS * PS_Q);
    if (q_base >= nq) return;
    const float *ppack = pred_packs + (size_t)rot * pack_stride;
    const float *Q = dir == 0 ? ppack : gt_pack;
    const float *C = dir == 0 ? gt_pack : ppack;
    const int ncs = pack_subs(nc);
    const float *Csub = C + (size_t)ncs * PS_SUB_FLOATS, *Ctile = Csub + (size_t)ncs * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float qx[PS_Q], qy[PS_Q], qz[PS_Q], bestd[PS_Q];
    float rlo[PS_Q][3], rhi[PS_Q][3];       // boxes of this wave's two runs of 64 queries (wave-uniform)
#pragma unroll
    for (int q = 0; q < PS_Q; q++) {
        int j = q_base + q * PS_THREADS + threadIdx.x;
        j = j < nq ? j : nq - 1;          // (a duplicate of the last query: inside the cloud, harmless for the boxes)
        const float *src = Q + (size_t)(j >> 6) * PS_SUB_FLOATS + (j & 63);
        qx[q] = src[0]; qy[q] = src[PS_SUB]; qz[q] = src[2 * PS_SUB];
        bestd[q] = INFINITY;
        if (CULL) {
            rlo[q][0] = wave_minf(qx[q]); rhi[q][0] = wave_maxf(qx[q]);
            rlo[q][1] = wave_minf(qy[q]); rhi[q][1] = wave_maxf(qy[q]);
            rlo[q][2] = wave_minf(qz[q]); rhi[q][2] = wave_maxf(qz[q]);
        }
    }
    const int T = ncs / PS_NSUB;
#ifdef ZS_POSE_COUNT
    unsigned scanned = 0;
#define ZS_COUNT(k) scanned += (k)
#else
#define ZS_COUNT(k)
#endif
    if (!CULL || T > 64) {
        for (int sidx = 0; sidx < ncs; sidx++)
            scan_subtile<true, true>(C + (size_t)sidx * PS_SUB_FLOATS, qx, qy, qz, bestd);
        ZS_COUNT(2 * ncs);
    } else {
        float wlo[3], whi[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            wlo[a] = fminf(rlo[0][a], rlo[1][a]);
            whi[a] = fmaxf(rhi[0][a], rhi[1][a]);
        }
        // lane t holds the distance from the wave's queries to tile t; tiles are taken nearest first
        float tg = INFINITY;
        if (lane < T) {
            float lo[3], hi[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                lo[a] = Ctile[(size_t)lane * 8 + a];
                hi[a] = Ctile[(size_t)lane * 8 + 3 + a];
            }
            tg = box_gap2(wlo, whi, lo, hi);
        }
        bool todo = lane < T;
        float w0 = INFINITY, w1 = INFINITY;
        for (int ti = 0; ti < T; ti++) {
            const float nearest = wave_minf(todo ? tg : INFINITY);
            if (!(nearest * PS_SKIP <= fmaxf(w0, w1))) break;      // ... and every remaining tile is farther still
            const unsigned long long pick = __ballot(todo && tg == nearest);
            const int t = __builtin_ctzll(pick);                   // (the lowest index among equals)
            todo = todo && lane != t;
            // lane l keeps the box of sub-tile l of this tile and its distance to the wave's two run boxes; the sub-tiles
            // that can still matter to a run are then taken nearest first (every scan shrinks the running minima and
            // with them the set of sub-tiles that survive the per-query test), their boxes broadcast with v_readlane
            float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
            float gm = INFINITY;
            bool cand = false;
            if (lane < PS_NSUB) {
                // (read from global memory per tile; a copy of all boxes in LDS was measured: no faster, the other waves
                // of the SIMD cover this round trip)
                const float *bsub = Csub + ((size_t)t * PS_NSUB + lane) * 8;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    slo[a] = bsub[a];
                    shi[a] = bsub[3 + a];
                }
                const float g0 = box_gap2(rlo[0], rhi[0], slo, shi), g1 = box_gap2(rlo[1], rhi[1], slo, shi);
                cand = g0 * PS_SKIP <= w0 || g1 * PS_SKIP <= w1;
                gm = fminf(g0, g1);
            }
#ifdef ZS_POSE_ITER_MIN   // rounds 3-5 (A/B): one wave-wide minimum per candidate
            while (__ballot(cand) != 0ull) {
                int sl;
                if ((stage >> 30) & 1) {           // measurement (ZS_POSE_LANE_ORDER=1): candidates in index order
                    sl = __builtin_ctzll(__ballot(cand));
                } else {
                    const float near_sub = wave_minf(cand ? gm : INFINITY);
                    sl = __builtin_ctzll(__ballot(cand && gm == near_sub));
                }
                cand = cand && lane != sl;
#else
            // The order is fixed when the tile is entered, so it is computed once: a candidate's rank among the tile's candidates
            // by (gap, lane) - the lane index rides in the four low mantissa bits of the non-negative gap, whose bit pattern orders
            // like an unsigned integer - from sixteen uniform compares, instead of one wave-wide minimum (~19 VALU instructions)
            // per candidate.  The order only decides how soon the running minima shrink, never a result.
            unsigned key = 0xffffffffu;
            if (cand) key = ((stage >> 30) & 1) ? (unsigned)lane     // measurement (ZS_POSE_LANE_ORDER=1): index order
                                                : ((__builtin_bit_cast(unsigned, gm) & ~15u) | (unsigned)lane);
            int rank = 0;
#pragma unroll
            for (int j = 0; j < PS_NSUB; j++) rank += (unsigned)__builtin_amdgcn_readlane((int)key, j) < key ? 1 : 0;
            const int ncand = __popcll(__ballot(cand));
            for (int r = 0; r < ncand; r++) {
                const int sl = __builtin_ctzll(__ballot(cand && rank == r));
#endif
                const float lo[3] = {lane_f(slo[0], sl), lane_f(slo[1], sl), lane_f(slo[2], sl)};
                const float hi[3] = {lane_f(shi[0], sl), lane_f(shi[1], sl), lane_f(shi[2], sl)};
                // both runs' point-to-box distances in packed form (the same operations per lane as point_gap2)
                const f32x2 pg = point_gap2_pair(qx, qy, qz, lo, hi) * f32x2{PS_SKIP, PS_SKIP};
                const bool d0 = __ballot(pg.x <= bestd[0]) != 0ull;
                const bool d1 = __ballot(pg.y <= bestd[1]) != 0ull;
                const float *c = C + ((size_t)t * PS_NSUB + sl) * PS_SUB_FLOATS;        // uniform: scalar loads
                if (d0 && d1)
                    scan_subtile<true, true>(c, qx, qy, qz, bestd);
                else if (d0)
                    scan_subtile<true, false>(c, qx, qy, qz, bestd);
                else if (d1)
                    scan_subtile<false, true>(c, qx, qy, qz, bestd);
                ZS_COUNT((d0 ? 1 : 0) + (d1 ? 1 : 0));
            }
            w0 = wave_maxf(bestd[0]);
            w1 = wave_maxf(bestd[1]);
        }
    }
#ifdef ZS_POSE_COUNT
    if (lane == 0) {
        atomicAdd(&g_pose_blocks[0], (unsigned long long)scanned);
        atomicAdd(&g_pose_blocks[1], (unsigned long long)(2 * ((nc + PS_SUB - 1) / PS_SUB)));
    }
#endif
#undef ZS_COUNT
    // epilogue: pose_nn_kernel's, statement for statement
    float v[PS_PART];
#pragma unroll
    for (int i = 0; i < PS_PART; i++) v[i] = 0.f;
    float thr[6];
#pragma unroll
    for (int i = 0; i < 6; i++) thr[i] = thresholds[i];
#pragma unroll
    for (int q = 0; q < PS_Q; q++) {
        const int j = q_base + q * PS_THREADS + threadIdx.x;
        if (j < nq) {
            const float sq = sqrtf(bestd[q]);
            v[0] += sq;
#pragma unroll
            for (int i = 0; i < 6; i++) v[1 + i] += sq < thr[i] ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 7; i++) v[i] = wave_sum(v[i]);
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 7; i++) red[i][wave] = v[i];
    __syncthreads();
    if (threadIdx.x < 7) {
        float sum = red[threadIdx.x][0];
        for (int w = 1; w < PS_THREADS / 64; w++) sum += red[threadIdx.x][w];
        partial[(((size_t)rot * 2 + dir) * blocks_x + bx) * PS_PART + threadIdx.x] = sum;
    }
}

// A rotation whose first `done` query blocks alone already give (s0 / n + s1 / m) / 2 > best cannot win: acc >= s0 / n and
// comp >= s1 / m (fixed-order sums of non-negative terms only grow; division, addition and halving are monotone under
// rounding), so cd = (acc + comp) / 2 is at least that - strictly above the running best, so ties are not affected.  First batch of a search: best = +inf, nobody dies.
__device__ __forceinline__ bool kill_test(const float *__restrict__ partial, int count, int n, int m, int blocks_x, int done,
                                          const float *__restrict__ best, int rot) {
    const int nb[2] = {(n + PS_THREADS * PS_Q - 1) / (PS_THREADS * PS_Q), (m + PS_THREADS * PS_Q - 1) / (PS_THREADS * PS_Q)};
    float s[2] = {0.f, 0.f};
    for (int dir = 0; dir < 2; dir++)
        for (int b = 0; b < min(done, nb[dir]); b++) s[dir] += partial[(((size_t)rot * 2 + dir) * blocks_x + b) * PS_PART];
    const float bst = prune_bound(best);
    return (s[0] / (float)n + s[1] / (float)m) / 2.f > bst;
}
// decide_merge: this is the kill behind the FIRST stage - at most a quarter of the batch dropped -> merge the remaining stages
__global__ __launch_bounds__(PS_THREADS) void pose_kill_kernel(const float *__restrict__ partial, int count, int n, int m,
                                                               int blocks_x, int done, const float *__restrict__ best,
                                                               int *__restrict__ dead,
                                                               const float *__restrict__ lower_bound, int decide_merge) {
    if (batch_pruned(lower_bound, best)) return;
    __shared__ int killed;
    if (threadIdx.x == 0) killed = 0;
    __syncthreads();
    const int rot = threadIdx.x;
    bool dies = false;
    if (rot < count) dies = kill_test(partial, count, n, m, blocks_x, done, best, rot);
    if (dies) {
        dead[rot] = 1;
        atomicAdd(&killed, 1);
    }
    __syncthreads();
    if (decide_merge && threadIdx.x == 0) dead[PS_MERGE_SLOT] = 4 * killed <= count ? 1 : 0;
}

// ---- the same search through uniform grids (csrc/zs_point_grid.h): ~10^2 instead of 10^4 distance evaluations per
// query, the same minima bit for bit (a skipped candidate provably has a strictly larger d) ------------------------ //
// The ground truth is binned once per search, normalize_pc(R p) once per rotation; queries and the per-workgroup
// partial sums are laid out exactly as in pose_nn_kernel, so pose_finish_kernel forms the same sums in the same order.
__global__ __launch_bounds__(zs::pgrid::BUILD_THREADS) void pose_gt_grid_kernel(const float *__restrict__ gt, int m,
                                                                                float *__restrict__ slot) {
    using namespace zs::pgrid;
    __shared__ int cnt[GRID_MAX_CELLS];
    __shared__ float red[6][BUILD_THREADS / 64];
    __shared__ int wsum[BUILD_THREADS / 64];
    point_grid_build([&](int i, float &x, float &y, float &z) {
        x = gt[(size_t)i * 3];
        y = gt[(size_t)i * 3 + 1];
        z = gt[(size_t)i * 3 + 2];
    }, m, slot, cnt, red, wsum);
}

__global__ __launch_bounds__(zs::pgrid::BUILD_THREADS) void pose_pred_grid_kernel(
    const float *__restrict__ pred, int n, const float *__restrict__ stats, float *__restrict__ slots, size_t slot_words_n,
    const float *__restrict__ lower_bound, const float *__restrict__ best) {
    using namespace zs::pgrid;
    if (batch_pruned(lower_bound, best)) return;
    __shared__ int cnt[GRID_MAX_CELLS];
    __shared__ float red[6][BUILD_THREADS / 64];
    __shared__ int wsum[BUILD_THREADS / 64];
    const int rot = blockIdx.x;
    Xform t;
    t.load(stats + (size_t)rot * PS_STAT);
    point_grid_build([&](int i, float &x, float &y, float &z) {
        t.apply(pred[(size_t)i * 3], pred[(size_t)i * 3 + 1], pred[(size_t)i * 3 + 2], x, y, z);
    }, n, slots + (size_t)rot * slot_words_n, cnt, red, wsum);
}

// grid (ceil(max(n, m) / (256 Q)), rotations in the batch, 2): pose_nn_kernel with the scan replaced by the grid walk
__global__ __launch_bounds__(PS_THREADS) void pose_nn_grid_kernel(
    const float *__restrict__ pred, int n, const float *__restrict__ gt, int m, const float *__restrict__ stats,
    const float *__restrict__ thresholds, float *__restrict__ partial, const float *__restrict__ gt_slot,
    const float *__restrict__ pred_slots, size_t slot_words_n, const float *__restrict__ lower_bound,
    const float *__restrict__ best) {
    if (batch_pruned(lower_bound, best)) return;
    __shared__ float red[PS_PART][PS_THREADS / 64];
    const int dir = blockIdx.z, rot = blockIdx.y;
    const int nq = dir == 0 ? n : m;      // queries
    const int q_base = blockIdx.x * (PS_THREADS * PS_Q);
    if (q_base >= nq) return;
    Xform t;
    t.load(stats + (size_t)rot * PS_STAT);
    const float *slot = dir == 0 ? gt_slot : pred_slots + (size_t)rot * slot_words_n;
    float bestd[PS_Q];
#pragma unroll
    for (int q = 0; q < PS_Q; q++) {
        int j = q_base + q * PS_THREADS + threadIdx.x;
        j = j < nq ? j : nq - 1;
        float qx, qy, qz;
        if (dir == 0) {
            t.apply(pred[(size_t)j * 3], pred[(size_t)j * 3 + 1], pred[(size_t)j * 3 + 2], qx, qy, qz);
        } else {
            qx = gt[(size_t)j * 3]; qy = gt[(size_t)j * 3 + 1]; qz = gt[(size_t)j * 3 + 2];
        }
        int who;
        zs::pgrid::point_grid_nearest(slot, qx, qy, qz, bestd[q], who);
    }
    // epilogue: pose_nn_kernel's, statement for statement
    float v[PS_PART];
#pragma unroll
    for (int i = 0; i < PS_PART; i++) v[i] = 0.f;
    float thr[6];
#pragma unroll
    for (int i = 0; i < 6; i++) thr[i] = thresholds[i];
#pragma unroll
    for (int q = 0; q < PS_Q; q++) {
        const int j = q_base + q * PS_THREADS + threadIdx.x;
        if (j < nq) {
            const float s = sqrtf(bestd[q]);
            v[0] += s;
#pragma unroll
            for (int i = 0; i < 6; i++) v[1 + i] += s < thr[i] ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 7; i++) v[i] = wave_sum(v[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 7; i++) red[i][wave] = v[i];
    __syncthreads();
    if (threadIdx.x < 7) {
        float s = red[threadIdx.x][0];
        for (int w = 1; w < PS_THREADS / 64; w++) s += red[threadIdx.x][w];
        partial[(((size_t)rot * 2 + dir) * gridDim.x + blockIdx.x) * PS_PART + threadIdx.x] = s;
    }
}

// F-score of utils/eval_3D.py:215-231 from the two hit fractions
__device__ __forceinline__ float fscore_of(float precision, float recall) {
    const float f = 2.f * precision * recall / (precision + recall);
    return f != f ? 0.f : f;   // 0 / 0 -> 0
}

// one workgroup per batch: per-rotation metrics (thread = rotation), then the batch winner
__global__ __launch_bounds__(PS_THREADS) void pose_finish_kernel(
    const float *__restrict__ partial, int count, int n, int m, int blocks_x, const int *__restrict__ order,
    int index_offset, float *__restrict__ best, const float *__restrict__ lower_bound, const int *__restrict__ dead) {
    if (batch_pruned(lower_bound, best)) return;
    __shared__ float s_cd[PS_THREADS];
    __shared__ int s_idx[PS_THREADS];
    __shared__ int s_alive;
    const int tid = threadIdx.x;
    if (tid == 0) s_alive = 0;
    __syncthreads();
    float cd = INFINITY, acc = 0.f, comp = 0.f, f[6];
    int gidx = 0x7fffffff;
    const bool alive = tid < count && !(dead && dead[tid]);
    if (alive) atomicAdd(&s_alive, 1);
    if (tid < count && !alive) gidx = (order ? order[tid] : tid) + index_offset;    // cd stays +inf
    if (alive) {
        float tot[2][7];
        const int nb[2] = {(n + PS_THREADS * PS_Q - 1) / (PS_THREADS * PS_Q), (m + PS_THREADS * PS_Q - 1) / (PS_THREADS * PS_Q)};
        for (int dir = 0; dir < 2; dir++) {
#pragma unroll
            for (int i = 0; i < 7; i++) tot[dir][i] = 0.f;
            for (int b = 0; b < nb[dir]; b++) {
                const float *p = partial + (((size_t)tid * 2 + dir) * blocks_x + b) * PS_PART;
#pragma unroll
                for (int i = 0; i < 7; i++) tot[dir][i] += p[i];
            }
        }
        acc = tot[0][0] / (float)n;
        comp = tot[1][0] / (float)m;
        cd = (acc + comp) / 2.f;
#pragma unroll
        for (int i = 0; i < 6; i++) f[i] = fscore_of(tot[0][1 + i] / (float)n, tot[1][1 + i] / (float)m);
        gidx = (order ? order[tid] : tid) + index_offset;
        if (cd != cd) cd = INFINITY;   // a NaN distance never wins (`cd < best_cd`, :162) and must not shadow the others
    }
    s_cd[tid] = cd;
    s_idx[tid] = gidx;
    __syncthreads();
    for (int o = PS_THREADS / 2; o > 0; o >>= 1) {   // lexicographic minimum of (cd, index)
        if (tid < o) {
            const float c2 = s_cd[tid + o];
            const int i2 = s_idx[tid + o];
            if (c2 < s_cd[tid] || (c2 == s_cd[tid] && i2 < s_idx[tid])) {
                s_cd[tid] = c2;
                s_idx[tid] = i2;
            }
        }
        __syncthreads();
    }
    const float win_cd = s_cd[0];
    const int win_idx = s_idx[0];
    __syncthreads();
    int *ibest = reinterpret_cast<int *>(best);
    if (tid == 0) {
        ibest[10] += count;     // rotations of batches that the lower bounds did not prune (diagnostics)
        ibest[11] += s_alive;   // ... of which scanned in full (the others lost after their first query block)
    }
    if (tid < count && gidx == win_idx) {
        const float old_cd = best[0];
        const int old_idx = ibest[1];
        if (win_cd < old_cd || (win_cd == old_cd && win_idx < old_idx)) {   // NaN never wins, like `cd < best`
            best[0] = cd;
            ibest[1] = gidx;
            best[2] = acc;
            best[3] = comp;
#pragma unroll
            for (int i = 0; i < 6; i++) best[4 + i] = f[i];
        }
    }
}

// compute_fscore (utils/eval_3D.py:215-231) on un-squared distances: one workgroup per batch row
__global__ __launch_bounds__(PS_THREADS) void fscore_kernel(const float *__restrict__ d1, int n,
                                                            const float *__restrict__ d2, int m,
                                                            const float *__restrict__ thresholds, int nt,
                                                            float *__restrict__ out) {
    __shared__ int cnt[2][PS_THREADS / 64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = 0; t < nt; t++) {
        const float thr = thresholds[t];
        int c1 = 0, c2 = 0;
        for (int i = threadIdx.x; i < n; i += PS_THREADS) c1 += d1[(size_t)b * n + i] < thr ? 1 : 0;
        for (int i = threadIdx.x; i < m; i += PS_THREADS) c2 += d2[(size_t)b * m + i] < thr ? 1 : 0;
        c1 = wave_sum(c1);
        c2 = wave_sum(c2);
        __syncthreads();
        if (lane == 0) {
            cnt[0][wave] = c1;
            cnt[1][wave] = c2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < PS_THREADS / 64; w++) {
                c1 += cnt[0][w];
                c2 += cnt[1][w];
            }
            out[(size_t)b * nt + t] = fscore_of((float)c1 / (float)n, (float)c2 / (float)m);
        }
    }
}

int blocks_x_of(int n, int m) {
    const int nmax = n > m ? n : m;
    return (nmax + PS_THREADS * PS_Q - 1) / (PS_THREADS * PS_Q);
}

}  // namespace

extern "C" int zs_pose_max_batch(void) { return PS_THREADS; }

extern "C" size_t zs_pose_scratch_bytes(int n, int m, int count) {
    if (n <= 0 || m <= 0 || count <= 0) return 0;
    return ((size_t)count * PS_STAT + (size_t)count * 2 * blocks_x_of(n, m) * PS_PART + (size_t)PS_DEAD_INTS) * sizeof(float);
}

extern "C" size_t zs_pose_best_bytes(void) { return PS_BEST * sizeof(float); }

extern "C" int zs_pose_best_init(float *best, void *stream) {
    if (!best) {
        zs::set_err("zs_pose_best_init: null pointer");
        return 0;
    }
    float h[PS_BEST];
    for (int i = 0; i < PS_BEST; i++) h[i] = 0.f;
    h[0] = INFINITY;
    h[12] = INFINITY;
    const int big = 0x7fffffff;
    __builtin_memcpy(&h[1], &big, 4);
    // (pageable host source: the copy is staged before the call returns)
    if (hipMemcpyAsync(best, h, sizeof(h), hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)) != hipSuccess) {
        zs::set_err("zs_pose_best_init: hipMemcpyAsync failed");
        return 0;
    }
    return 1;
}

namespace {
// pred: the cloud in its caller's order (statistics: the mean's double sum depends on the order, and zs_pose_apply
// runs on this order); pred_nn / gt_nn: the clouds the scans read (the same, or Morton-sorted).  mode 0: pose_nn_kernel
// (all pairs, candidates staged in LDS); 1 / 2: pose_nn_soa_kernel on packs, all pairs / box-culled.
int search_batch(const float *pred, const float *pred_nn, int n, const float *gt_nn, const float *gt_pack, int m, int mode,
                 const float *rotations, const int *order, int count, int index_offset, const float *lower_bound,
                 const float *thresholds6, float *best, void *scratch, void *stream) {
    if (n < 0 || m < 0 || count < 0) {
        zs::set_err("zs_pose_search_batch: negative size (n=%d m=%d count=%d)", n, m, count);
        return 0;
    }
    if (count == 0) return 1;
    if (n == 0 || m == 0) {
        zs::set_err("zs_pose_search_batch: empty cloud (n=%d m=%d)", n, m);
        return 0;
    }
    if (count > PS_THREADS) {
        zs::set_err("zs_pose_search_batch: %d rotations per batch exceed %d", count, PS_THREADS);
        return 0;
    }
    if (!pred || !pred_nn || !rotations || !thresholds6 || !best || !scratch || (mode == 0 ? !gt_nn : !gt_pack) || mode < 0 ||
        mode > 2) {
        zs::set_err("zs_pose_search_batch: null pointer or bad mode");
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *stats = static_cast<float *>(scratch);
    float *partial = stats + (size_t)count * PS_STAT;
    const int bx = blocks_x_of(n, m);
    hipLaunchKernelGGL(pose_stats_kernel, dim3(count), dim3(PS_THREADS), 0, st, pred, (size_t)0, n, rotations, order,
                       stats, lower_bound, best);
    // The query blocks are scanned in stages of 1, 2, 4 and the rest; after each stage the rotations whose partial sums
    // already prove them worse than the running best are dropped (pose_kill_kernel).  Exact: a dropped rotation could
    // not have won, and the survivors' sums are formed as before.  On an alignable ground truth most of the sphere dies
    // after 5-15 % of its queries.
    int *dead = reinterpret_cast<int *>(partial + (size_t)count * 2 * bx * PS_PART);
    if (hipMemsetAsync(dead, 0, (size_t)PS_DEAD_INTS * sizeof(int), st) != hipSuccess) {
        zs::set_err("zs_pose_search_batch: hipMemsetAsync failed");
        return 0;
    }
    float *packs = reinterpret_cast<float *>(dead + PS_DEAD_INTS);
    const size_t pstride = pack_floats(n);
    if (mode != 0)
        hipLaunchKernelGGL(pose_pack_kernel, dim3(pack_subs(n) / PS_NSUB, count), dim3(PS_TILE), 0, st, pred_nn, n,
                           static_cast<const float *>(stats), packs, pstride, lower_bound, static_cast<const float *>(best));
    // ZS_POSE_STAGES="a,b,c" moves the stage boundaries (measurement; "0,0,0" = one stage: no staged drop); ZS_POSE_MERGE=0
    // keeps every stage to its own blocks.  Defaults from tools/pose_stages.py (6,912 rotations, 10k x 10k points, batch 256):
    // stages 1,3,7: unalignable 41.3 / alignable 2.86 ms; 2,6,6: 39.5 / 2.62; one stage: 35.4 / 3.58.
    static const char *stage_env = getenv("ZS_POSE_STAGES");
    static const bool merge_on = getenv("ZS_POSE_MERGE") == nullptr || atoi(getenv("ZS_POSE_MERGE")) != 0;
    int stage_end[4] = {2, 6, 6, bx};
    if (stage_env) {
        int a = 1, b = 3, c = 7;
        if (sscanf(stage_env, "%d,%d,%d", &a, &b, &c) == 3) { stage_end[0] = a; stage_end[1] = b; stage_end[2] = c; }
    }
    int lo = 0, launched = 0;
    for (int sidx = 0; sidx < 4 && lo < bx; sidx++) {
        const int hi = stage_end[sidx] < bx ? stage_end[sidx] : bx;
        if (hi <= lo) continue;
        // the second launch spans every remaining block (it scans them all when the first kill merged the stages)
        const int smode = !merge_on || launched == 0 ? 0 : launched == 1 ? 1 : 2;
        const int span = smode == 1 ? bx - lo : hi - lo;
        static const int lane_order = getenv("ZS_POSE_LANE_ORDER") && atoi(getenv("ZS_POSE_LANE_ORDER")) ? 1 : 0;
        const int stage = (lane_order << 30) | (smode << 24) | hi;
        const dim3 grid(span, count, 2);
        if (mode == 2)
            hipLaunchKernelGGL(pose_nn_soa_kernel<true>, grid, dim3(PS_THREADS), 0, st, static_cast<const float *>(packs), pstride,
                               n, gt_pack, m, thresholds6, partial, lower_bound, static_cast<const float *>(best), lo, bx,
                               static_cast<const int *>(dead), stage);
        else if (mode == 1)
            hipLaunchKernelGGL(pose_nn_soa_kernel<false>, grid, dim3(PS_THREADS), 0, st, static_cast<const float *>(packs), pstride,
                               n, gt_pack, m, thresholds6, partial, lower_bound, static_cast<const float *>(best), lo, bx,
                               static_cast<const int *>(dead), stage);
        else
            hipLaunchKernelGGL(pose_nn_kernel, grid, dim3(PS_THREADS), 0, st, pred_nn, n, gt_nn, m, stats, thresholds6, partial,
                               lower_bound, best, lo, bx, static_cast<const int *>(dead), stage);
        if (hi < bx)
            hipLaunchKernelGGL(pose_kill_kernel, dim3(1), dim3(PS_THREADS), 0, st, partial, count, n, m, bx, hi, best, dead,
                               lower_bound, merge_on && launched == 0 ? 1 : 0);
        lo = hi;
        launched++;
    }
    hipLaunchKernelGGL(pose_finish_kernel, dim3(1), dim3(PS_THREADS), 0, st, partial, count, n, m, bx, order,
                       index_offset, best, lower_bound, static_cast<const int *>(dead));
    return zs::check_launch("zs_pose_search_batch") ? 1 : 0;
}
}  // namespace

extern "C" int zs_pose_search_batch(const float *pred, int n, const float *gt_normalized, int m,
                                    const float *rotations, const int *order, int count, int index_offset,
                                    const float *lower_bound, const float *thresholds6, float *best,
                                    void *scratch, void *stream) {
    return search_batch(pred, pred, n, gt_normalized, nullptr, m, 0, rotations, order, count, index_offset, lower_bound,
                        thresholds6, best, scratch, stream);
}

#ifdef ZS_POSE_COUNT
extern "C" int zs_pose_debug_counters(unsigned long long *out2, int reset) {
    if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_pose_blocks), 16) != hipSuccess) return 0;
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_pose_blocks), z, 16) != hipSuccess) return 0;
    }
    return 1;
}
#endif

extern "C" size_t zs_pose_pack_bytes(int points) { return points > 0 ? pack_floats(points) * sizeof(float) : 0; }

extern "C" size_t zs_pose_sorted_scratch_bytes(int n, int m, int count) {
    if (n <= 0 || m <= 0 || count <= 0) return 0;
    return zs_pose_scratch_bytes(n, m, count) + (size_t)count * zs_pose_pack_bytes(n);
}

extern "C" int zs_pose_pack(const float *sorted_points, int points, float *pack, void *stream) {
    if (points <= 0 || !sorted_points || !pack) {
        zs::set_err("zs_pose_pack: bad arguments (points=%d)", points);
        return 0;
    }
    hipLaunchKernelGGL(pose_pack_kernel, dim3(pack_subs(points) / PS_NSUB, 1), dim3(PS_TILE), 0, static_cast<hipStream_t>(stream),
                       sorted_points, points, static_cast<const float *>(nullptr), pack, (size_t)0,
                       static_cast<const float *>(nullptr), static_cast<const float *>(nullptr));
    return zs::check_launch("zs_pose_pack") ? 1 : 0;
}

extern "C" int zs_pose_search_batch_sorted(const float *pred, const float *pred_sorted, int n, const float *gt_sorted,
                                           const float *gt_pack, int m, const float *rotations, const int *order, int count,
                                           int index_offset, const float *lower_bound, const float *thresholds6, float *best,
                                           void *scratch, int mode, void *stream) {
    return search_batch(pred, pred_sorted, n, gt_sorted, gt_pack, m, mode, rotations, order, count, index_offset, lower_bound,
                        thresholds6, best, scratch, stream);
}

extern "C" size_t zs_pose_grid_bytes(int n, int m, int count) {
    if (n <= 0 || m <= 0 || count <= 0) return 0;
    return (zs::pgrid::slot_words(m) + (size_t)count * zs::pgrid::slot_words(n)) * sizeof(float);
}

extern "C" int zs_pose_gt_grid(const float *gt_normalized, int m, void *grids, void *stream) {
    if (m <= 0 || !gt_normalized || !grids || (reinterpret_cast<uintptr_t>(grids) & 15)) {
        zs::set_err("zs_pose_gt_grid: bad arguments (m=%d; 16-byte aligned buffer)", m);
        return 0;
    }
    hipLaunchKernelGGL(pose_gt_grid_kernel, dim3(1), dim3(zs::pgrid::BUILD_THREADS), 0, static_cast<hipStream_t>(stream),
                       gt_normalized, m, static_cast<float *>(grids));
    return zs::check_launch("zs_pose_gt_grid") ? 1 : 0;
}

extern "C" int zs_pose_search_batch_grid(const float *pred, int n, const float *gt_normalized, int m,
                                         const float *rotations, const int *order, int count, int index_offset,
                                         const float *lower_bound, const float *thresholds6, float *best,
                                         void *scratch, void *grids, const float *pred_stats, void *stream) {
    if (n <= 0 || m <= 0 || count < 0) {
        zs::set_err("zs_pose_search_batch_grid: bad size (n=%d m=%d count=%d)", n, m, count);
        return 0;
    }
    if (count == 0) return 1;
    if (count > PS_THREADS) {
        zs::set_err("zs_pose_search_batch_grid: %d rotations per batch exceed %d", count, PS_THREADS);
        return 0;
    }
    if (!pred || !gt_normalized || !rotations || !thresholds6 || !best || !scratch || !grids ||
        (reinterpret_cast<uintptr_t>(grids) & 15)) {
        zs::set_err("zs_pose_search_batch_grid: null or misaligned pointer");
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *stats = static_cast<float *>(scratch);
    float *partial = stats + (size_t)count * PS_STAT;
    float *gt_slot = static_cast<float *>(grids);
    float *pred_slots = gt_slot + zs::pgrid::slot_words(m);
    const size_t sw = zs::pgrid::slot_words(n);
    const int bx = blocks_x_of(n, m);
    hipLaunchKernelGGL(pose_stats_kernel, dim3(count), dim3(PS_THREADS), 0, st, pred_stats ? pred_stats : pred, (size_t)0, n,
                       rotations, order, stats, lower_bound, best);
    hipLaunchKernelGGL(pose_pred_grid_kernel, dim3(count), dim3(zs::pgrid::BUILD_THREADS), 0, st, pred, n, stats, pred_slots,
                       sw, lower_bound, best);
    hipLaunchKernelGGL(pose_nn_grid_kernel, dim3(bx, count, 2), dim3(PS_THREADS), 0, st, pred, n, gt_normalized, m, stats,
                       thresholds6, partial, gt_slot, pred_slots, sw, lower_bound, best);
    hipLaunchKernelGGL(pose_finish_kernel, dim3(1), dim3(PS_THREADS), 0, st, partial, count, n, m, bx, order,
                       index_offset, best, lower_bound, static_cast<const int *>(nullptr));
    return zs::check_launch("zs_pose_search_batch_grid") ? 1 : 0;
}

extern "C" int zs_pose_apply(const float *pred, int n, const float *rotations, const int *index, float *out,
                             void *scratch, void *stream) {
    if (n < 0) {
        zs::set_err("zs_pose_apply: negative size");
        return 0;
    }
    if (n == 0) return 1;
    if (!pred || !rotations || !out || !scratch) {
        zs::set_err("zs_pose_apply: null pointer");
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *stats = static_cast<float *>(scratch);
    hipLaunchKernelGGL(pose_stats_kernel, dim3(1), dim3(PS_THREADS), 0, st, pred, (size_t)0, n, rotations, index, stats,
                       static_cast<const float *>(nullptr), static_cast<const float *>(nullptr));
    hipLaunchKernelGGL(pose_apply_kernel, dim3((n + PS_THREADS - 1) / PS_THREADS, 1), dim3(PS_THREADS), 0, st, pred,
                       (size_t)0, n, stats, out);
    return zs::check_launch("zs_pose_apply") ? 1 : 0;
}

extern "C" int zs_normalize_pc(const float *pc, int b, int n, float *out, void *scratch, void *stream) {
    if (b < 0 || n < 0) {
        zs::set_err("zs_normalize_pc: negative size (b=%d n=%d)", b, n);
        return 0;
    }
    if (b == 0 || n == 0) return 1;
    if (b > 65535) {
        zs::set_err("zs_normalize_pc: batch %d > 65535", b);
        return 0;
    }
    if (!pc || !out || !scratch) {
        zs::set_err("zs_normalize_pc: null pointer");
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *stats = static_cast<float *>(scratch);
    hipLaunchKernelGGL(pose_stats_kernel, dim3(b), dim3(PS_THREADS), 0, st, pc, (size_t)n * 3, n,
                       static_cast<const float *>(nullptr), static_cast<const int *>(nullptr), stats,
                       static_cast<const float *>(nullptr), static_cast<const float *>(nullptr));
    int bx = (n + PS_THREADS - 1) / PS_THREADS;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(pose_apply_kernel, dim3(bx, b), dim3(PS_THREADS), 0, st, pc, (size_t)n * 3, n, stats, out);
    return zs::check_launch("zs_normalize_pc") ? 1 : 0;
}

extern "C" int zs_fscore(const float *dist1, int n, const float *dist2, int m, int b, const float *thresholds,
                         int n_thresholds, float *out, void *stream) {
    if (b < 0 || n < 0 || m < 0 || n_thresholds < 0) {
        zs::set_err("zs_fscore: negative size");
        return 0;
    }
    if (b == 0 || n_thresholds == 0) return 1;
    if (!dist1 || !dist2 || !thresholds || !out) {
        zs::set_err("zs_fscore: null pointer");
        return 0;
    }
    hipLaunchKernelGGL(fscore_kernel, dim3(b), dim3(PS_THREADS), 0, static_cast<hipStream_t>(stream), dist1, n, dist2,
                       m, thresholds, n_thresholds, out);
    return zs::check_launch("zs_fscore") ? 1 : 0;
}
