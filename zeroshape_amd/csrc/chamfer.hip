// Chamfer-3D nearest-neighbour kernels for MI355X (gfx950), written from scratch.
//
// Replaces external/chamfer3D/chamfer3D.cu (NmDistanceKernel :12-134,
// NmDistanceGradKernel :155-174) behind the C ABI of include/zeroshape_hip.h.
//
// Design (not a translation of the CUDA kernel):
//  * one launch covers BOTH directions (blockIdx.z) and all batches (blockIdx.y);
//  * each lane keeps Q queries and their running (best, argbest) in registers for
//    the whole scan - the reference round-trips the running minimum through
//    global memory every 512 reference points (chamfer3D.cu:126-129);
//  * the other cloud is staged through LDS as structure-of-arrays tiles
//    (x[T] | y[T] | z[T]) so that 4 candidates come back from three broadcast
//    ds_read_b128 (all lanes read the same address: conflict-free), and the
//    tail of the last tile is padded with +inf so the inner loop has no bounds
//    checks (an +inf candidate can never win the strict '<');
//  * arithmetic is the reference's, spelled with explicit fmaf so the result is
//    bit-identical: d = fma(dz,dz, fma(dy,dy, dx*dx)), (dx,dy,dz) = other - self,
//    strict '<' while scanning in index order => lowest index wins ties.
//
//  * 4 candidates per step; when no lane of a wave improves any of its queries in a step
//    (the common case: the chance decays like 1/k) the compare/select chain is skipped.
//
// The kernel is fp32-VALU bound: 6 VALU ops per point pair for the distance + 0.75 for the
// group minimum (+3 compare/select in the rare improving steps); no HBM traffic to speak
// of ((n+m)*B*20 bytes per call).
//
// Round 6: zs_chamfer_forward launches nn_both_sgpr_kernel (below: candidates as scalar operands of packed fp32
// instructions, no LDS) - same bits, 0.92 -> 0.82 ms on [24, 10k] x [24, 10k]; ZS_CHAMFER_LDS=1 keeps nn_both_kernel.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 1024;  // candidates per LDS tile (12 KiB as SoA)
constexpr int NN_STRIDE = NN_TILE + 4;  // +4: the look-ahead read of the last step stays in bounds

template <int Q>
__global__ __launch_bounds__(NN_THREADS) void nn_both_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int n1, int n2,
    float *__restrict__ dist1, float *__restrict__ dist2, int *__restrict__ idx1,
    int *__restrict__ idx2) {
    __shared__ __attribute__((aligned(16))) float tile[3 * NN_STRIDE];

    const int dir = blockIdx.z;
    const int batch = blockIdx.y;
    const int n = dir == 0 ? n1 : n2;  // queries
    const int m = dir == 0 ? n2 : n1;  // candidates
    const int q_base = blockIdx.x * (NN_THREADS * Q);
    if (q_base >= n || m == 0) return;  // uniform per block

    const float *__restrict__ qry = (dir == 0 ? xyz1 : xyz2) + (size_t)batch * n * 3;
    const float *__restrict__ cand = (dir == 0 ? xyz2 : xyz1) + (size_t)batch * m * 3;
    float *__restrict__ out_d = (dir == 0 ? dist1 : dist2) + (size_t)batch * n;
    int *__restrict__ out_i = (dir == 0 ? idx1 : idx2) + (size_t)batch * n;

    float qx[Q], qy[Q], qz[Q], best[Q];
    int best_i[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        int j = q_base + q * NN_THREADS + threadIdx.x;
        j = j < n ? j : n - 1;  // clamp: duplicates are computed but not stored
        qx[q] = qry[j * 3 + 0];
        qy[q] = qry[j * 3 + 1];
        qz[q] = qry[j * 3 + 2];
        best[q] = INFINITY;
        best_i[q] = 0;
    }

    for (int k0 = 0; k0 < m; k0 += NN_TILE) {
        const int cnt = min(NN_TILE, m - k0);
        const int cnt4 = (cnt + 3) & ~3;
        __syncthreads();  // previous tile fully consumed
        for (int t = threadIdx.x; t < cnt4; t += NN_THREADS) {
            float x = INFINITY, y = INFINITY, z = INFINITY;
            if (t < cnt) {
                x = cand[(size_t)(k0 + t) * 3 + 0];
                y = cand[(size_t)(k0 + t) * 3 + 1];
                z = cand[(size_t)(k0 + t) * 3 + 2];
            }
            tile[t] = x;
            tile[NN_STRIDE + t] = y;
            tile[2 * NN_STRIDE + t] = z;
        }
        __syncthreads();
        // 4 candidates per step; the next step's LDS reads are issued before this step's
        // arithmetic (register double buffer).  Fast path: if no lane of the wave improves any
        // of its queries in this group (the common case once the running minima have settled:
        // the chance decays like 1/k), the compare/select chain is skipped altogether - exact,
        // because a candidate that is not strictly smaller changes nothing.
        float4 cx = *reinterpret_cast<const float4 *>(&tile[0]);
        float4 cy = *reinterpret_cast<const float4 *>(&tile[NN_STRIDE]);
        float4 cz = *reinterpret_cast<const float4 *>(&tile[2 * NN_STRIDE]);
        for (int k = 0; k < cnt4; k += 4) {
            const float4 nx = *reinterpret_cast<const float4 *>(&tile[k + 4]);  // padded: always in bounds
            const float4 ny = *reinterpret_cast<const float4 *>(&tile[NN_STRIDE + k + 4]);
            const float4 nz = *reinterpret_cast<const float4 *>(&tile[2 * NN_STRIDE + k + 4]);
            const float ax[4] = {cx.x, cx.y, cx.z, cx.w};
            const float ay[4] = {cy.x, cy.y, cy.z, cy.w};
            const float az[4] = {cz.x, cz.y, cz.z, cz.w};
            float d[Q][4];
            bool any = false;
#pragma unroll
            for (int q = 0; q < Q; q++) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float dx = ax[c] - qx[q];
                    const float dy = ay[c] - qy[q];
                    const float dz = az[c] - qz[q];
                    d[q][c] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                }
                const float mn = fminf(fminf(fminf(d[q][0], d[q][1]), d[q][2]), d[q][3]);
                any |= mn < best[q];
            }
            if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const bool better = d[q][c] < best[q];
                        best[q] = better ? d[q][c] : best[q];
                        best_i[q] = better ? (k0 + k + c) : best_i[q];
                    }
                }
            }
            cx = nx;
            cy = ny;
            cz = nz;
        }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int j = q_base + q * NN_THREADS + threadIdx.x;
        if (j < n) {
            out_d[j] = best[q];
            out_i[j] = best_i[q];
        }
    }
}

// ---- the same scan with the candidates as SCALAR operands (round 6) ------------------------------------------------ //
// nn_both_kernel moves every candidate through LDS (three broadcast ds_read_b128 per four candidates and wave) and spends
// 7 VALU instructions per pair; the pose search's all-pairs scan (csrc/pose_search.hip) runs the same arithmetic at 1.7x its
// pair rate because there a candidate is UNIFORM data: it arrives through the scalar cache in SGPRs and enters packed fp32
// instructions (v_pk_add / v_pk_mul / v_pk_fma on an SGPR pair = two candidates, the lane's query broadcast).  This kernel
// does that for the plugin's own [B][m][3] layout: a wave reads 8 consecutive candidates = 24 floats with scalar loads, the
// x / y / z of candidates (2e, 2e + 1) are gathered into SGPR pairs by s_mov (the scalar unit is otherwise idle), no LDS, no
// barrier - the four waves of a workgroup walk independently.  Per lane the operations and their order are nn_both_kernel's
// (c + (-q) == c - q exactly; v_pk_fma_f32 is a fused multiply-add; strict '<' in index order), so every distance and every
// index keeps its bits.  The last m % 8 candidates are taken one at a time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
struct Chunk24 {                      // eight [x y z] candidates in SGPRs
    f32x8 a, b, c;                    // floats 0-7 | 8-15 | 16-23
    __device__ __forceinline__ void request(const float *p) {
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x20\n\ts_load_dwordx8 %2, %3, 0x40"
                     : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(p) : "memory");
    }
    // the same, pinned IN FRONT of the arithmetic on `held` (the statement "returns" held's registers too)
    __device__ __forceinline__ void request_before(const float *p, Chunk24 &held) {
        asm volatile("s_load_dwordx8 %0, %6, 0x0\n\ts_load_dwordx8 %1, %6, 0x20\n\ts_load_dwordx8 %2, %6, 0x40"
                     : "=&s"(a), "=&s"(b), "=&s"(c), "+s"(held.a), "+s"(held.b), "+s"(held.c) : "s"(p) : "memory");
    }
    __device__ __forceinline__ void arrived() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c) : : "memory"); }
    __device__ __forceinline__ float at(int i) const { return i < 8 ? a[i] : i < 16 ? b[i - 8] : c[i - 16]; }
};

template <int Q>
__global__ __launch_bounds__(NN_THREADS) void nn_both_sgpr_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int n1, int n2,
    float *__restrict__ dist1, float *__restrict__ dist2, int *__restrict__ idx1,
    int *__restrict__ idx2) {
    const int dir = blockIdx.z;
    const int batch = blockIdx.y;
    const int n = dir == 0 ? n1 : n2;  // queries
    const int m = dir == 0 ? n2 : n1;  // candidates
    const int q_base = blockIdx.x * (NN_THREADS * Q);
    if (q_base >= n || m == 0) return;  // uniform per block

    const float *__restrict__ qry = (dir == 0 ? xyz1 : xyz2) + (size_t)batch * n * 3;
    const float *__restrict__ cand = (dir == 0 ? xyz2 : xyz1) + (size_t)batch * m * 3;
    float *__restrict__ out_d = (dir == 0 ? dist1 : dist2) + (size_t)batch * n;
    int *__restrict__ out_i = (dir == 0 ? idx1 : idx2) + (size_t)batch * n;

    float qx[Q], qy[Q], qz[Q], best[Q];
    int best_i[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        int j = q_base + q * NN_THREADS + threadIdx.x;
        j = j < n ? j : n - 1;  // clamp: duplicates are computed but not stored
        qx[q] = qry[j * 3 + 0];
        qy[q] = qry[j * 3 + 1];
        qz[q] = qry[j * 3 + 2];
        best[q] = INFINITY;
        best_i[q] = 0;
    }

    // eight candidates = 24 floats per chunk, two chunks in ping-pong: the next chunk is requested before the arithmetic on the
    // current one (asm, as in csrc/pose_search.hip: hipcc sinks plain scalar loads next to their use and waits at once)
    auto scan8 = [&](const Chunk24 &ch, int k) {
#pragma unroll
        for (int g = 0; g < 2; g++) {                            // two groups of four candidates, as nn_both_kernel's steps
            float d[Q][4];                                       // (one test per eight candidates was measured: slower, 0.845 vs 0.82 ms)
            bool any = false;
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const f32x2 qx2 = {qx[q], qx[q]}, qy2 = {qy[q], qy[q]}, qz2 = {qz[q], qz[q]};
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int a = (4 * g + 2 * e) * 3;           // candidates (4g + 2e, 4g + 2e + 1)
                    const f32x2 cx = {ch.at(a), ch.at(a + 3)}, cy = {ch.at(a + 1), ch.at(a + 4)}, cz = {ch.at(a + 2), ch.at(a + 5)};
                    const f32x2 dx = cx - qx2, dy = cy - qy2, dz = cz - qz2;
                    const f32x2 dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    d[q][2 * e] = dd.x;
                    d[q][2 * e + 1] = dd.y;
                }
                const float mn = fminf(fminf(fminf(d[q][0], d[q][1]), d[q][2]), d[q][3]);
                any |= mn < best[q];
            }
            if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const bool better = d[q][cc] < best[q];
                        best[q] = better ? d[q][cc] : best[q];
                        best_i[q] = better ? (k + 4 * g + cc) : best_i[q];
                    }
                }
            }
        }
    };
    const int m8 = m & ~7;
    if (m8 > 0) {
        Chunk24 A, B;
        A.request(cand);
        A.arrived();
        int k = 0;
        for (; k + 16 <= m8; k += 16) {
            B.request_before(cand + (size_t)(k + 8) * 3, A);
            scan8(A, k);
            B.arrived();
            if (k + 16 < m8) A.request_before(cand + (size_t)(k + 16) * 3, B);
            scan8(B, k + 8);
            if (k + 16 < m8) A.arrived();
        }
        if (k < m8) scan8(A, k);                                  // an odd number of chunks: the last one is already here
    }
    for (int k = m8; k < m; k++) {                               // the last m % 8 candidates
        const float cx = cand[(size_t)k * 3], cy = cand[(size_t)k * 3 + 1], cz = cand[(size_t)k * 3 + 2];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const float dx = cx - qx[q], dy = cy - qy[q], dz = cz - qz[q];
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            const bool better = d < best[q];
            best[q] = better ? d : best[q];
            best_i[q] = better ? k : best_i[q];
        }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int j = q_base + q * NN_THREADS + threadIdx.x;
        if (j < n) {
            out_d[j] = best[q];
            out_i[j] = best_i[q];
        }
    }
}

// chamfer3D.cu:155-174 semantics; one thread per (batch, point), both directions.
__global__ __launch_bounds__(256) void nn_grad_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int b, int n1, int n2,
    const float *__restrict__ gd1, const float *__restrict__ gd2, const int *__restrict__ idx1,
    const int *__restrict__ idx2, float *__restrict__ g1, float *__restrict__ g2) {
    const int dir = blockIdx.z;
    const int n = dir == 0 ? n1 : n2;
    const int m = dir == 0 ? n2 : n1;
    const float *__restrict__ a = dir == 0 ? xyz1 : xyz2;
    const float *__restrict__ c = dir == 0 ? xyz2 : xyz1;
    const float *__restrict__ gd = dir == 0 ? gd1 : gd2;
    const int *__restrict__ idx = dir == 0 ? idx1 : idx2;
    float *__restrict__ ga = dir == 0 ? g1 : g2;
    float *__restrict__ gc = dir == 0 ? g2 : g1;
    const size_t total = (size_t)b * n;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (size_t)gridDim.x * blockDim.x) {
        const size_t i = t / n;
        const int j2 = idx[t];
        const float x1 = a[t * 3 + 0], y1 = a[t * 3 + 1], z1 = a[t * 3 + 2];
        const size_t o = (i * m + j2) * 3;
        const float x2 = c[o + 0], y2 = c[o + 1], z2 = c[o + 2];
        const float g = gd[t] * 2;
        atomicAdd(&ga[t * 3 + 0], g * (x1 - x2));
        atomicAdd(&ga[t * 3 + 1], g * (y1 - y2));
        atomicAdd(&ga[t * 3 + 2], g * (z1 - z2));
        atomicAdd(&gc[o + 0], -(g * (x1 - x2)));
        atomicAdd(&gc[o + 1], -(g * (y1 - y2)));
        atomicAdd(&gc[o + 2], -(g * (z1 - z2)));
    }
}

}  // namespace

extern "C" int zs_chamfer_forward(const float *xyz1, const float *xyz2, int b, int n, int m,
                                  float *dist1, float *dist2, int *idx1, int *idx2,
                                  void *stream) {
    if (b < 0 || n < 0 || m < 0) {
        zs::set_err("zs_chamfer_forward: negative size (b=%d n=%d m=%d)", b, n, m);
        return 0;
    }
    if (b == 0 || n == 0 || m == 0) return 1;  // nothing is written (see header)
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2) {
        zs::set_err("zs_chamfer_forward: null pointer");
        return 0;
    }
    if (b > 65535) {
        zs::set_err("zs_chamfer_forward: batch %d > 65535", b);
        return 0;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nmax = n > m ? n : m;
    // Q queries per lane: more queries amortise the LDS reads and add independent chains,
    // fewer give more blocks; ZS_CHAMFER_Q overrides for experiments
    int Q = ((long long)b * nmax >= 64 * 1024) ? 2 : 1;
    if (const char *e = getenv("ZS_CHAMFER_Q")) Q = atoi(e);
    dim3 grid((nmax + NN_THREADS * Q - 1) / (NN_THREADS * Q), b, 2);
    const bool lds_scan = getenv("ZS_CHAMFER_LDS") != nullptr;             // A/B (read per call): rounds 1-5's LDS-staged scan
    if (!lds_scan && (Q == 4 || Q == 2 || Q == 1)) {
        if (Q == 4)
            hipLaunchKernelGGL(nn_both_sgpr_kernel<4>, grid, dim3(NN_THREADS), 0, s, xyz1, xyz2, n, m, dist1,
                               dist2, idx1, idx2);
        else if (Q == 2)
            hipLaunchKernelGGL(nn_both_sgpr_kernel<2>, grid, dim3(NN_THREADS), 0, s, xyz1, xyz2, n, m, dist1,
                               dist2, idx1, idx2);
        else
            hipLaunchKernelGGL(nn_both_sgpr_kernel<1>, grid, dim3(NN_THREADS), 0, s, xyz1, xyz2, n, m, dist1,
                               dist2, idx1, idx2);
        return zs::check_launch("zs_chamfer_forward") ? 1 : 0;
    }
    if (Q == 4)
        hipLaunchKernelGGL(nn_both_kernel<4>, grid, dim3(NN_THREADS), 0, s, xyz1, xyz2, n, m, dist1,
                           dist2, idx1, idx2);
    else if (Q == 2)
        hipLaunchKernelGGL(nn_both_kernel<2>, grid, dim3(NN_THREADS), 0, s, xyz1, xyz2, n, m, dist1,
                           dist2, idx1, idx2);
    else if (Q == 1)
        hipLaunchKernelGGL(nn_both_kernel<1>, grid, dim3(NN_THREADS), 0, s, xyz1, xyz2, n, m, dist1,
                           dist2, idx1, idx2);
    else {
        zs::set_err("zs_chamfer_forward: unsupported ZS_CHAMFER_Q=%d", Q);
        return 0;
    }
    return zs::check_launch("zs_chamfer_forward") ? 1 : 0;
}

extern "C" int zs_chamfer_backward(const float *xyz1, const float *xyz2, int b, int n, int m,
                                   float *gradxyz1, float *gradxyz2, const float *graddist1,
                                   const float *graddist2, const int *idx1, const int *idx2,
                                   void *stream) {
    if (b < 0 || n < 0 || m < 0) {
        zs::set_err("zs_chamfer_backward: negative size (b=%d n=%d m=%d)", b, n, m);
        return 0;
    }
    if (b == 0 || n == 0 || m == 0) return 1;
    if (!xyz1 || !xyz2 || !gradxyz1 || !gradxyz2 || !graddist1 || !graddist2 || !idx1 || !idx2) {
        zs::set_err("zs_chamfer_backward: null pointer");
        return 0;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long nmax = (long long)b * (n > m ? n : m);
    int blocks = (int)((nmax + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nn_grad_kernel, dim3(blocks, 1, 2), dim3(256), 0, s, xyz1, xyz2, b, n, m,
                       graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2);
    return zs::check_launch("zs_chamfer_backward") ? 1 : 0;
}
