// Batch-1 pointwise GEMM of the split-fp16 inference engine: the 197-token matrices of the ViT blocks, the 1 x 1 layers on 14 x 14
// and 7 x 7 maps, the one-pixel fc head - problems whose 128 x 128 tiling would leave most CUs idle.
//
// What tools/ubench/small_gemm.hip measured on MI355X inside a dependent chain of launches with cold weights (round 4):
//   * the CU's vector-memory path moves ~30 B/clk of streamed operands (64 B/clk on hits): a launch costs
//     ~1.8 us + bytes through its busiest CU / ~77 GB/s, so the tiling must use all 256 CUs ONCE (a second round of
//     workgroups costs a whole workgroup latency: 252 tiles dealt 35 to an XCD of 32 CUs ran 16.3 us, dealt evenly 8.8 us),
//   * hipcc sinks every operand load next to its MFMA and waits vmcnt(0) (ISA of the first version) - the operand ring
//     therefore lives in registers only inline asm writes, with counted waits,
//   * a ring deeper than ~3 K = 16 steps buys nothing (the issue of the loads, not their latency, is the bound), per-step
//     vector address arithmetic costs as much as the loads (operands are addressed as uniform base + fixed lane offset),
//   * staging A through LDS in full lines (LDS-DMA, swizzled) measured no better than fragment-shaped register loads here.
// Structure: workgroup = 32 MI rows x 32 NJ columns x a range of K; its NW waves split that range (operands straight from
// global memory into MFMA registers, nothing shared), partial tiles summed through LDS in wave order.  The kernel body is
// the micro-benchmark's, which tools/ubench/small_gemm.hip (RACE=1) screens bit-identical over 60 x 8 launches at every
// shape; a first rewrite with a tap walker inside the same ring showed sporadic whole-tile corruption and is not used.
#include "nn_gemm_stream.h"
#include "zs_common.h"
#include "zs_split16.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

namespace zs {
namespace stream_gemm {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Geo { int mtiles, ntiles, splits, raw; };      // raw: K ranges write raw partial sums [z][M][N], no ticket

__device__ __forceinline__ float activate(float v, int act) {
    if (act == ZS_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ZS_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == ZS_ACT_RELU_CLAMP1) return fminf(fmaxf(v, 0.f), 1.f);
    return v;
}

__device__ __forceinline__ void activate4(f32x4 &v, int act) {          // one decision per quad, not three per value (nn_conv.hip)
    if (act == ZS_ACT_NONE) return;
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = activate(v[e], act);
}

template <int NW, int MI, int NJ, int DEPTH, bool LN>
__global__ __launch_bounds__(64 * NW) void stream_gemm_kernel(Args a, Geo g) {
    constexpr int SM = 32 * MI, SN = 32 * NJ, PAD = SN + 4;
    __shared__ __attribute__((aligned(16))) float part[NW][SM][PAD];
    __shared__ float rowtab[LN ? 2 : 1][LN ? SM : 1];       // LN: (rstd, -mean rstd) of the tile's rows
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mtiles = g.mtiles, ntiles = g.ntiles;
    // XCD-aware tile order: workgroup id b runs on XCD b % 8; the row tiles of one column slab share an XCD (the slab's
    // weights are fetched into ONE L2)
    int tm, tn;
    const int z = blockIdx.y;
    {
        const int b = blockIdx.x;
        const int T = mtiles * ntiles, xcd = b & 7, local = b >> 3;
        const int lo = (int)((long long)xcd * T / 8), hi = (int)((long long)(xcd + 1) * T / 8);
        const int t = lo + local;
        if (t >= hi) return;
        tn = t / mtiles;
        tm = t - tn * mtiles;
    }
    const int m0 = tm * SM, n0 = tn * SN;
    const int S = (a.K + 15) / 16;                        // K = 16 steps
    const int zs0 = (int)((long long)z * S / g.splits), zs1 = (int)((long long)(z + 1) * S / g.splits);
    const int per_w = (zs1 - zs0 + NW - 1) / NW;
    const int s_begin = zs0 + wave * per_w, s_end = min(zs1, s_begin + per_w);
    const int ns = max(s_end - s_begin, 0);

    // operand addresses = uniform base (SGPR pair, advanced per step by scalar adds) + per-lane 32-bit offset (fixed): no
    // vector address arithmetic in the loop
    unsigned aoff[MI];
    bool rok[MI];
#pragma unroll
    for (int i = 0; i < MI; i++) {
        const int m = m0 + 32 * i + l32;
        rok[i] = m < a.M;
        aoff[i] = a.a_k16 ? (unsigned)((rok[i] ? m : 0) * 64 + 16 * half)
                          : (unsigned)(((size_t)(rok[i] ? m : 0) * a.lda + 4 * half) * 4);
    }
    const size_t astep = a.a_k16 ? (size_t)a.M * 64 : 64;         // bytes from one K = 16 step of A to the next (uniform)
    unsigned boff = (unsigned)(((size_t)half * a.CoutPad + n0 + l32) * 16);
    const char *abase = reinterpret_cast<const char *>(a.a);
    const char *bbase = reinterpret_cast<const char *>(a.w);
    const size_t bstep2 = (size_t)2 * a.CoutPad * 16;              // two weight quad rows

    constexpr int L = 2 * MI + 2 * NJ;
    static_assert((DEPTH - 1) * L < 64, "vmcnt is a 6-bit counter");
    f32x4 ra[DEPTH][MI][2];
    f32x4 rb[DEPTH][NJ][2];
#define ZS_GLDS(dst, voff, sbase, IMM) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory")
    auto load = [&](int slot, int s) {
        const int sc = min(s, S - 1);               // clamped: always a valid address
        const int sa = sc;
        const char *pb0 = bbase + (size_t)sc * 2 * bstep2, *pb1 = pb0 + bstep2;
        const char *pa = abase + (size_t)sa * astep;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            ZS_GLDS(rb[slot][j][0], boff, pb0, 512 * j);
            ZS_GLDS(rb[slot][j][1], boff, pb1, 512 * j);
        }
#pragma unroll
        for (int i = 0; i < MI; i++) {
            ZS_GLDS(ra[slot][i][0], aoff[i], pa, 0);
            ZS_GLDS(ra[slot][i][1], aoff[i], pa, 32);
        }
    };
    auto landed = [&](int slot) {
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((DEPTH - 1) * L) : "memory");
#pragma unroll
        for (int j = 0; j < NJ; j++) asm volatile("" : "+v"(rb[slot][j][0]), "+v"(rb[slot][j][1]));
#pragma unroll
        for (int i = 0; i < MI; i++) asm volatile("" : "+v"(ra[slot][i][0]), "+v"(ra[slot][i][1]));
    };
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) { rb[d][j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[d][j][1] = rb[d][j][0]; }
#pragma unroll
        for (int i = 0; i < MI; i++) { ra[d][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ra[d][i][1] = ra[d][i][0]; }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) load(d, s_begin + d);
    // LN: LayerNorm of the input rows (gamma / beta live in the weights) from the producer's (sum, M2 about its own mean) per row
    // and column tile: eight lanes per row split the tiles; mean from the sums, then M2 = sum of (M2_t + n_t (mean_t - mean)^2).
    // The operand loads above are in flight meanwhile.
    float row_s[MI], row_t[MI];
#pragma unroll
    for (int i = 0; i < MI; i++) { row_s[i] = 1.f; row_t[i] = 0.f; }
    if (LN) {
        const int tiles = a.in_tiles;
        const float nb = (float)a.K / (float)tiles;
        for (int r0 = 0; r0 < SM; r0 += 8 * NW) {
            const int row = r0 + (tid >> 3), sub = tid & 7, m = m0 + row;
            float sums[4], m2s[4], S1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int tl = sub + 8 * k;
                float2 e = {0.f, 0.f};
                if (row < SM && m < a.M && tl < tiles) e = *reinterpret_cast<const float2 *>(a.in_stats + ((size_t)m * tiles + tl) * 2);
                sums[k] = e.x;
                m2s[k] = e.y;
                S1 += e.x;
            }
            S1 += __shfl_xor(S1, 1, 64); S1 += __shfl_xor(S1, 2, 64); S1 += __shfl_xor(S1, 4, 64);
            const float mean = S1 / (float)a.K;
            float M2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dd = sums[k] / nb - mean;
                if (sub + 8 * k < tiles) M2 += m2s[k] + nb * dd * dd;
            }
            M2 += __shfl_xor(M2, 1, 64); M2 += __shfl_xor(M2, 2, 64); M2 += __shfl_xor(M2, 4, 64);
            if (sub == 0 && row < SM) {
                const float rstd = 1.0f / sqrtf(M2 / (float)a.K + a.in_eps);
                rowtab[0][row] = m < a.M ? rstd : 0.f;
                rowtab[LN ? 1 : 0][row] = m < a.M ? -mean * rstd : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; i++) { row_s[i] = rowtab[0][32 * i + l32]; row_t[i] = rowtab[LN ? 1 : 0][32 * i + l32]; }
    }
    for (int base = 0; base < ns; base += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const bool live = base + d < ns;
            landed(d);
            u32x4 ah[MI], al[MI];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                f32x4 q0 = ra[d][i][0], q1 = ra[d][i][1];
                if (LN) { q0 = q0 * row_s[i] + row_t[i]; q1 = q1 * row_s[i] + row_t[i]; }
                if (!(live && rok[i])) { q0 = f32x4{0.f, 0.f, 0.f, 0.f}; q1 = q0; }
                if (a.in_relu) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { q0[e] = fmaxf(q0[e], 0.f); q1[e] = fmaxf(q1[e], 0.f); }
                }
                zs::s16::split8(q0, q1, ah[i], al[i]);
            }
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    zs::s16::mfma3(acc[i][j], __builtin_bit_cast(u32x4, rb[d][j][0]), __builtin_bit_cast(u32x4, rb[d][j][1]), ah[i], al[i]);   // transposed: lane = pixel
            load(d, s_begin + base + DEPTH + d);
        }
    }
    // The refills of the last DEPTH steps are still in flight.  The compiler does not know that: to it a ring register is dead
    // after its last MFMA, and it reused some for the epilogue's addresses AHEAD of this wait (tools/ring_audit.py on the ISA) - a
    // load landing late then overwrote an LDS / global address.  Pinning every ring register after the wait keeps all of them
    // allocated until the loads have landed.  (Root cause of the round-4 "layout dependent" faults of the K split and of the
    // 32 x 96 tiles; the unsplit variants had the same window and were lucky.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) asm volatile("" : "+v"(rb[d][j][0]), "+v"(rb[d][j][1]));
#pragma unroll
        for (int i = 0; i < MI; i++) asm volatile("" : "+v"(ra[d][i][0]), "+v"(ra[d][i][1]));
    }
    // wave partials -> LDS: register 4q + e of lane (l32, half) = channel 8q + 4 half + e of pixel l32
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<f32x4 *>(&part[wave][32 * i + l32][32 * j + 8 * q + 4 * half]) =
                    f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    __syncthreads();
    constexpr int QPR = SN / 4, QUADS = SM * QPR, THREADS = 64 * NW, PASSES = QUADS / THREADS;
    static_assert(QUADS % THREADS == 0, "whole passes");
    f32x4 v[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int e = tid + THREADS * ps, p = e / QPR, c = e % QPR;
        v[ps] = *reinterpret_cast<const f32x4 *>(&part[0][p][4 * c]);
#pragma unroll
        for (int w = 1; w < NW; w++) v[ps] += *reinterpret_cast<const f32x4 *>(&part[w][p][4 * c]);
    }
    if (g.splits > 1 && g.raw) {
        // two-launch split: this range's raw partial sums, laid out like the output; the caller's reduce launch sums the ranges
        // in range order and runs the epilogue (and the statistics)
        float *dstz = a.parts + (size_t)z * a.M * a.N;
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int e = tid + THREADS * ps, p = e / QPR, c = e % QPR;
            const int m = m0 + p, n = n0 + 4 * c;
            if (m < a.M && n < a.N) *reinterpret_cast<f32x4 *>(dstz + (size_t)m * a.N + n) = v[ps];
        }
        return;
    }
    if (g.splits > 1) {
        // K split across blockIdx.y: publish this range's partial tile, take a ticket; the last arriver sums all ranges in range
        // order - the same sum whoever it is
        const int tile = tn * mtiles + tm;
        float *mine = a.parts + ((size_t)tile * g.splits + z) * (SM * SN);
        int *flag = reinterpret_cast<int *>(&part[0][0][0]);
        {
            // partial tiles as agent-scope atomics (written through to / read at the coherence point of the eight L2s): no cache
            // write-back or invalidate of a whole L2 per workgroup
            // 16-byte stores with sc0 sc1 (system-coherent write-through: a wave writes whole 128-byte lines; the same data as dword
            // atomics - what __hip_atomic_store gives - is a partial-line write per lane and 4x the instructions)
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++) {
                const float *dst = mine + (size_t)(tid + THREADS * ps) * 4;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(v[ps]) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                // ACQ_REL at agent scope: release = this workgroup's partial tile (drained above), acquire = the last arriver reads
                // the others' (one thread per workgroup pays the L2 write-back / invalidate; +0.5-1 us over RELAXED, measured)
                const int old = __hip_atomic_fetch_add(&a.tickets[tile], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == g.splits - 1;
                if (last) __hip_atomic_store(&a.tickets[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = last;
            }
            __syncthreads();
            if (!*flag) return;
            const float *all = a.parts + (size_t)tile * g.splits * (SM * SN);
            // four ranges' loads in flight at a time (one range per trip costs a memory round trip per range); summed in range order
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++) v[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int z0 = 0; z0 < g.splits; z0 += 4) {
                f32x4 t[4][PASSES];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int zz = min(z0 + u, g.splits - 1);
#pragma unroll
                    for (int ps = 0; ps < PASSES; ps++) {
                        const float *src = all + (size_t)zz * (SM * SN) + (size_t)(tid + THREADS * ps) * 4;
                        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(t[u][ps]) : "v"(src) : "memory");
                    }
                }
                // the compiler does not know these loads are in flight: wait here, and "return" every destination from the wait
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int ps = 0; ps < PASSES; ps++) asm volatile("" : "+v"(t[u][ps]));
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (z0 + u < g.splits) {
#pragma unroll
                        for (int ps = 0; ps < PASSES; ps++) v[ps] += t[u][ps];
                    }
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int e = tid + THREADS * ps, p = e / QPR, c = e % QPR;
        const int m = m0 + p, n = n0 + 4 * c;
        const bool valid = m < a.M && n < a.N;
        if (!valid && !a.out_stats) continue;
        f32x4 r = v[ps];
        if (valid) {
            const size_t o = a.out_k16 ? ((size_t)(n >> 4) * a.M + m) * 16 + (n & 15) : (size_t)m * a.N + n;
            if (a.scale) r *= *reinterpret_cast<const f32x4 *>(a.scale + n);
            if (a.shift) r += *reinterpret_cast<const f32x4 *>(a.shift + n);
            if (a.res1) r += *reinterpret_cast<const f32x4 *>(a.res1 + o);
            if (a.res2) r += *reinterpret_cast<const f32x4 *>(a.res2 + o);
            activate4(r, a.act);
            *reinterpret_cast<f32x4 *>(a.out + o) = r;
        }
        if constexpr ((QPR & (QPR - 1)) == 0) {
            // statistics of the stored rows for the consumer's fused LayerNorm (zs_conv_fuse.out_mode 2): (sum, M2 about the
            // mean of this column tile) per (row, column tile); the QPR lanes of a row are neighbours in the wave
            if (a.out_stats) {
                float rs = valid ? (r[0] + r[1]) + (r[2] + r[3]) : 0.f;
#pragma unroll
                for (int sh = 1; sh < QPR; sh <<= 1) rs += __shfl_xor(rs, sh, 64);
                const float mean = rs * (1.0f / SN);
                float d2 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) d2 += (r[k] - mean) * (r[k] - mean);
                if (!valid) d2 = 0.f;
#pragma unroll
                for (int sh = 1; sh < QPR; sh <<= 1) d2 += __shfl_xor(d2, sh, 64);
                if (valid && c == 0) {
                    float *dst = a.out_stats + ((size_t)m * ntiles + tn) * 2;
                    dst[0] = rs;
                    dst[1] = d2;
                }
            }
        }
    }
}

#undef ZS_GLDS

// Tile shape of a problem.  Cost model from the measurements above: time ~ bytes through the busiest CU / 77 GB/s; more than
// one workgroup per CU runs in rounds.
struct Plan { int mi, nj, splits, mtiles, ntiles; };

Plan plan(const Args &a) {
    static const int cus = getenv("ZS_STREAM_CUS") ? atoi(getenv("ZS_STREAM_CUS")) : 256;
    const int steps = (a.K + 15) / 16;
    Plan best = {1, 2, 1, 0, 0};
    double best_cost = 1e30;
    const bool can_split = a.parts && a.tickets && !a.a_k16 && !a.out_k16;
    // candidates: 32 / 64 rows x 32 / 64 / 96 columns (96: the weight rows of the last tile must exist, nt * 96 <= CoutPad)
    for (int mi = 1; mi <= 2; mi++) {
        if (mi == 2 && a.M <= 32) continue;
        const long long mt = (a.M + 32 * mi - 1) / (32 * mi);
        for (int nj = 1; nj <= 3; nj++) {
            const long long nt = (a.N + 32 * nj - 1) / (32 * nj), T = mt * nt;
            if (a.out_stats && 32 * nj != a.stats_cols) continue;
            if (nj == 3 && nt * 96 > a.CoutPad) continue;
            if (T > a.max_tickets) continue;
            for (int z = 1; z <= 16; z++) {
                if (z > 1 && (!can_split || steps / (z * 4) < 2 || T * z > cus ||
                              (size_t)T * z * 32 * mi * 32 * nj * 4 > a.parts_bytes)) break;
                const double rounds = (double)((T * z + cus - 1) / cus);
                const double bytes = (32.0 * mi + 32.0 * nj) * 64.0 * ((steps + z - 1) / z);
                const double tile_bytes = 32.0 * mi * 32.0 * nj * 4.0;
                const double cost = rounds * bytes / 77e3 + 0.4 * mi * nj + (z > 1 ? 2.0 + (z + 1) * tile_bytes / 77e3 : 0.0);
                if (cost < best_cost) { best_cost = cost; best = Plan{mi, nj, z, (int)mt, (int)nt}; }
            }
        }
    }
    if (const char *f = getenv("ZS_STREAM_FORCE")) {       // measurement / debugging override: "mi,nj,z"
        int mi = 0, nj = 0, z = 1;
        sscanf(f, "%d,%d,%d", &mi, &nj, &z);
        if (mi >= 1 && mi <= 2 && nj >= 1 && nj <= 3 && (nj < 3 || (a.N + 95) / 96 * 96 <= a.CoutPad) && z >= 1 &&
            (z == 1 || can_split) && steps / (z * 4) >= 1)
            best = Plan{mi, nj, z, (int)((a.M + 32 * mi - 1) / (32 * mi)), (int)((a.N + 32 * nj - 1) / (32 * nj))};
    }
    return best;
}

}  // namespace

bool launch(const Args &a, hipStream_t st, bool dry_run) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K & 15) || (a.N & 3) || (a.lda & 3)) return false;
    if (a.a_k16 && a.in_stats) return false;
    if (a.out_k16 && ((a.N & 15) || a.res1 || a.res2 || a.out_stats)) return false;
    if (a.out_stats && ((a.stats_cols != 32 && a.stats_cols != 64) || a.N % a.stats_cols)) return false;
    if (a.in_stats && (a.in_tiles <= 0 || a.in_tiles > 32 || a.lda != a.K)) return false;
    if ((size_t)a.M * a.lda * 4 >= ((size_t)1 << 32)) return false;        // 32-bit lane offsets
    if (const char *only = getenv("ZS_STREAM_ONLY")) {                // debugging: the kernel for one (K, N) only
        int k = 0, n = 0;
        if (sscanf(only, "%d,%d", &k, &n) == 2 && (k != a.K || n != a.N)) return false;
    }
    static const int max_k = getenv("ZS_STREAM_MAX_K") ? atoi(getenv("ZS_STREAM_MAX_K")) : (1 << 30);   // measurement: longer K -> small-tile kernel
    if (a.K > max_k) return false;
    const Plan p = plan(a);
    if (a.out_stats && 32 * p.nj != a.stats_cols) return false;
    const long long T = (long long)p.mtiles * p.ntiles;
    // The kernel wins where its tiles fill the chip once (ViT qkv / fc1 at batch 1: 8.8 vs 13.6 us, 11 vs 19.6 us); with few
    // tiles the small-tile kernel's narrower tiles (two workgroups per CU) are faster (proj 7.8 vs 8.6 us, 1 x 1 layers of
    // 256 channels on 14 x 14 maps 4.9 vs 7.0 us) - tools/ubench/small_gemm.hip.  ZS_STREAM_MIN_TILES moves the line.
    static const long long min_tiles = getenv("ZS_STREAM_MIN_TILES") ? atoll(getenv("ZS_STREAM_MIN_TILES")) : 160;      // 96 / 128 / 160 / 192 / 256: 2.934 / 2.921 / 2.917 / 2.947 / 3.03 ms per batch-1 forward
    // ... or where splitting a long contraction puts the whole chip on a layer that has few tiles (ViT fc2: 84 tiles x 3)
    // OFF by default (ZS_STREAM_SPLIT=1 enables it).  It is correct now (the round-4 faults were ring registers reused ahead of the
    // final vmcnt wait - see the pin block after the K loop and tools/ring_audit.py; 900 back-to-back split launches and the
    // encoder tests are clean with it on) but it does not pay: every extra range costs 2-3 us of exchange (write-through partial
    // tile, agent-scope ticket, read-back) - ViT fc2 23.2 (no split) -> 18.8 (2 ranges) -> 21.8 (3) vs 20.4 us for the small-tile
    // kernel; proj 9.7 -> 11.5 -> 14.7 vs 8.3; the 2,048 -> 512 layer at 7 x 7 15.9 -> 11.1 (4) vs 12.8 (tools/stream_shapes.py).
    static const bool allow_split = getenv("ZS_STREAM_SPLIT") != nullptr && atoi(getenv("ZS_STREAM_SPLIT")) != 0;
    if (T * p.splits < min_tiles || (p.splits > 1 && !allow_split)) return false;
    if (const char *only = getenv("ZS_STREAM_SPLIT_ONLY")) {          // debugging: the K split for one (K, N) only
        int k = 0, n = 0;
        if (p.splits > 1 && sscanf(only, "%d,%d", &k, &n) == 2 && (k != a.K || n != a.N)) return false;
    }
    Geo g = {p.mtiles, p.ntiles, p.splits, 0};
    if (a.ranges) *a.ranges = 1;
    // two-launch K split (the caller reduces).  OFF by default (ZS_STREAM_2L_K = smallest K that splits): for the ViT fc2 at batch 1
    // (168 tiles of 32 x 32, K 3,072, every workgroup streams 786 KB) two / three ranges + the reduce launch measured 2.71 / 2.68 ms
    // per forward against 2.67 without - the reduce launch eats what the shorter ranges save.
    static const int two_k = getenv("ZS_STREAM_2L_K") ? atoi(getenv("ZS_STREAM_2L_K")) : (1 << 30);
    static const int two_target = getenv("ZS_STREAM_2L_TARGET") ? atoi(getenv("ZS_STREAM_2L_TARGET")) : 384;
    if (a.two_launch_max > 1 && a.ranges && a.parts && p.splits == 1 && a.K >= two_k && !a.a_k16 && !a.out_k16) {
        long long z = two_target / T;
        if (z > a.two_launch_max) z = a.two_launch_max;
        if (z > (a.K / 16) / 8) z = (a.K / 16) / 8;                               // at least eight K = 16 steps per range
        while (z > 1 && (size_t)z * a.M * a.N * 4 > a.parts_bytes) z--;
        if (z > 1) {
            g.splits = (int)z;
            g.raw = 1;
            *a.ranges = (int)z;
        }
    }
    if (dry_run) return true;
    const dim3 grid((unsigned)(8 * ((T + 7) / 8)), (unsigned)g.splits);
#define ZS_SG(MI_, NJ_)                                                                                              \
    do {                                                                                                             \
        if (a.in_stats) hipLaunchKernelGGL((stream_gemm_kernel<4, MI_, NJ_, 3, true>), grid, dim3(256), 0, st, a, g); \
        else hipLaunchKernelGGL((stream_gemm_kernel<4, MI_, NJ_, 3, false>), grid, dim3(256), 0, st, a, g);           \
    } while (0)
    if (p.mi == 2) { if (p.nj == 1) ZS_SG(2, 1); else if (p.nj == 2) ZS_SG(2, 2); else ZS_SG(2, 3); }
    else if (p.nj == 1) ZS_SG(1, 1);
    else if (p.nj == 2) ZS_SG(1, 2);
    else ZS_SG(1, 3);
#undef ZS_SG
    return true;
}

}  // namespace stream_gemm
}  // namespace zs
