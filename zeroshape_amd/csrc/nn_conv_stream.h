// Batch-1 GEMM / convolution kernel of the split-fp16 inference engine (included by nn_conv.hip): the layers whose 128 x 128
// tiling would leave most CUs idle - 14 x 14 and 7 x 7 maps, 197-token matrices, the one-pixel fc head.
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
// global memory into MFMA registers, nothing shared), partial tiles summed through LDS in wave order; K may also be split
// across blockIdx.y (`zsplit`): partial tiles go to the workspace, the last workgroup to arrive (a ticket per tile) sums them
// in split order and runs the epilogue - the same sum whoever arrives last.  Geometry: pointwise, or kh x kw taps with
// Cin % 16 == 0 (a K = 16 step lies inside one tap).  Fused: input ReLU, LayerNorm of the input rows (XF 2), scale / shift /
// two residuals / activation, row statistics of the output (out_mode 2).
#pragma once

namespace stream {

constexpr int MAX_WAVES = 8;

struct Geo {            // per-launch constants beyond ConvArgs
    int mtiles, ntiles, zsplit, steps;      // steps = K16 steps of the whole contraction
    int *tickets;                           // one per tile, zero between launches
    float *parts;                           // [tile][zsplit][32 MI x 32 NJ] partial tiles
};

#define ZS_GLDS(dst, voff, sbase, IMM) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory")

template <int NW, int MI, int NJ, int DEPTH, bool PW, int XF>
__global__ __launch_bounds__(64 * NW) void conv_stream_kernel(ConvArgs a, Geo g) {
    constexpr int TM_ = 32 * MI, TN_ = 32 * NJ, PAD = TN_ + 4, THREADS = 64 * NW;
    constexpr int L = 2 * MI + 2 * NJ;                      // vector loads per ring slot
    static_assert((DEPTH - 1) * L < 64, "vmcnt is a 6-bit counter");
    // one LDS object (a second one makes hipcc drain vmcnt before LDS reads): wave partial tiles, then flags / row tables
    __shared__ __attribute__((aligned(16))) float lds[NW * TM_ * PAD + 2 * TM_ + 4];
    float (*part)[TM_][PAD] = reinterpret_cast<float (*)[TM_][PAD]>(lds);
    float *rowtab = lds + NW * TM_ * PAD;                   // XF 2: [2][TM_] (rstd, -mean rstd)
    int *flag = reinterpret_cast<int *>(lds + NW * TM_ * PAD + 2 * TM_);
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // balanced XCD-aware tile order: workgroup id b runs on XCD b % 8; XCD x owns the contiguous tile range
    // [x T / 8, (x + 1) T / 8) in (column tile major, row tile minor) order: a column slab's weights land in ONE L2 and no XCD
    // gets more tiles than CUs while another idles
    int tm, tn;
    {
        const int T = g.mtiles * g.ntiles, b = blockIdx.x, xcd = b & 7, local = b >> 3;
        const int lo = (int)((long long)xcd * T / 8), hi = (int)((long long)(xcd + 1) * T / 8);
        if (lo + local >= hi) return;
        const int t = lo + local;
        tn = t / g.mtiles;
        tm = t - tn * g.mtiles;
    }
    const int z = blockIdx.y;
    const int m0 = tm * TM_, n0 = tn * TN_;
    const int zs0 = (int)((long long)z * g.steps / g.zsplit), zs1 = (int)((long long)(z + 1) * g.steps / g.zsplit);
    const int per_w = (zs1 - zs0 + NW - 1) / NW;
    const int s_begin = zs0 + wave * per_w, s_end = min(zs1, s_begin + per_w), ns = max(s_end - s_begin, 0);

    // ---- operand addressing: uniform base (scalar, advanced per step) + per-lane 32-bit byte offset ----
    bool rok[MI];
    unsigned aoff[MI];                                      // current lane offset of the A rows (PW: fixed; taps: per tap)
    int iy0[MI], ix0[MI];
    unsigned pbase[MI];                                     // taps: byte offset of the row's sample
#pragma unroll
    for (int i = 0; i < MI; i++) {
        const int m = m0 + 32 * i + l32;
        rok[i] = m < a.M;
        if (PW) {
            aoff[i] = (unsigned)(((size_t)(rok[i] ? m : 0) * a.Cin + 4 * half) * 4);
        } else {
            int pb = 0, py = 0, px = 0;
            if (rok[i]) {
                pb = m / (a.Hout * a.Wout);
                const int rem = m - pb * a.Hout * a.Wout;
                py = rem / a.Wout;
                px = rem - py * a.Wout;
            }
            iy0[i] = py * a.stride - a.pad_t;
            ix0[i] = px * a.stride - a.pad_l;
            pbase[i] = (unsigned)((size_t)pb * a.Hin * a.Win * a.Cin * 4);
            aoff[i] = 0;
        }
    }
    const unsigned boff = (unsigned)(((size_t)half * a.CoutPad + n0 + l32) * 16);
    const char *abase = reinterpret_cast<const char *>(a.in);
    const char *bbase = reinterpret_cast<const char *>(a.w);
    const size_t bstep2 = (size_t)2 * a.CoutPad * 16;       // two weight quad rows = one lane-half's hi -> lo distance
    // tap walker of the LOADS (uniform): K = 16 step s = (tap, channel offset cc)
    int l_tap = 0, l_cc = 0, l_ky = 0, l_kx = 0;
    unsigned okbits = 0;                                    // bit (slot * MI + i): the slot's tap is inside the image for row block i
    auto set_tap = [&]() {                                  // per-lane offsets of the current tap
#pragma unroll
        for (int i = 0; i < MI; i++) {
            const int iy = iy0[i] + l_ky, ix = ix0[i] + l_kx;
            const bool ok = rok[i] && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            aoff[i] = ok ? pbase[i] + (unsigned)((iy * a.Win + ix) * a.Cin + 4 * half) * 4u : (unsigned)(16 * half);
            okbits = (okbits & ~(1u << (24 + i))) | ((ok ? 1u : 0u) << (24 + i));      // bits 24.. = the current tap's flags
        }
    };
    if (!PW) {
        const int k0 = 16 * min(s_begin, g.steps - 1);
        l_tap = k0 / a.Cin;
        l_cc = k0 - l_tap * a.Cin;
        l_ky = l_tap / a.kw;
        l_kx = l_tap - l_ky * a.kw;
        set_tap();
    }

    f32x4 ra[DEPTH][MI][2], rb[DEPTH][NJ][2];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) { rb[d][j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[d][j][1] = rb[d][j][0]; }
#pragma unroll
        for (int i = 0; i < MI; i++) { ra[d][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ra[d][i][1] = ra[d][i][0]; }
    }
    int l_s = s_begin;                                      // next step to load
    auto load = [&](int slot) {
        const int sc = min(l_s, g.steps - 1);               // clamped: always a valid address (the products are masked at use)
        const char *pb0 = bbase + (size_t)sc * 2 * bstep2, *pb1 = pb0 + bstep2;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            ZS_GLDS(rb[slot][j][0], boff, pb0, 512 * j);
            ZS_GLDS(rb[slot][j][1], boff, pb1, 512 * j);
        }
        const char *pa = PW ? abase + (size_t)sc * 64 : abase + (size_t)l_cc * 4;
#pragma unroll
        for (int i = 0; i < MI; i++) {
            ZS_GLDS(ra[slot][i][0], aoff[i], pa, 0);
            ZS_GLDS(ra[slot][i][1], aoff[i], pa, 32);
            if (!PW) okbits = (okbits & ~(1u << (slot * MI + i))) | (((okbits >> (24 + i)) & 1u) << (slot * MI + i));
        }
        l_s++;
        if (!PW && l_s < g.steps) {
            l_cc += 16;
            if (l_cc >= a.Cin) {                            // uniform
                l_cc = 0;
                l_kx++;
                if (l_kx == a.kw) { l_kx = 0; l_ky++; }
                set_tap();
            }
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
    for (int d = 0; d < DEPTH; d++) load(d);

    // ---- XF 2: LayerNorm of the input rows from the producer's (sum, M2) per column tile (8 lanes per row) ----
    float row_s[MI], row_t[MI];
#pragma unroll
    for (int i = 0; i < MI; i++) { row_s[i] = 1.f; row_t[i] = 0.f; }
    if (XF == 2) {
        const int tiles = a.fz.in_tiles;
        const float nb = (float)a.Cin / (float)tiles;
        for (int r0 = 0; r0 < TM_; r0 += THREADS / 8) {
            const int row = r0 + (tid >> 3), sub = tid & 7, m = m0 + row;
            float sums[4], m2s[4], S = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int tl = sub + 8 * k;
                float2 e = {0.f, 0.f};
                if (row < TM_ && m < a.M && tl < tiles) e = *reinterpret_cast<const float2 *>(a.fz.in_stats + ((size_t)m * tiles + tl) * 2);
                sums[k] = e.x;
                m2s[k] = e.y;
                S += e.x;
            }
            S += __shfl_xor(S, 1, 64); S += __shfl_xor(S, 2, 64); S += __shfl_xor(S, 4, 64);
            const float mean = S / (float)a.Cin;
            float M2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float dd = sums[k] / nb - mean;
                if (sub + 8 * k < tiles) M2 += m2s[k] + nb * dd * dd;
            }
            M2 += __shfl_xor(M2, 1, 64); M2 += __shfl_xor(M2, 2, 64); M2 += __shfl_xor(M2, 4, 64);
            if (sub == 0 && row < TM_) {
                const float rstd = 1.0f / sqrtf(M2 / (float)a.Cin + a.fz.in_eps);
                rowtab[row] = m < a.M ? rstd : 0.f;
                rowtab[TM_ + row] = m < a.M ? -mean * rstd : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; i++) { row_s[i] = rowtab[32 * i + l32]; row_t[i] = rowtab[TM_ + 32 * i + l32]; }
    }

    const float relu_floor = a.in_relu ? 0.f : -INFINITY;
    for (int base = 0; base < ns; base += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const bool live = base + d < ns;
            landed(d);
            u32x4 ah[MI], al[MI];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                f32x4 q0 = ra[d][i][0], q1 = ra[d][i][1];
                const bool ok = live && (PW ? rok[i] : ((okbits >> (d * MI + i)) & 1u) != 0);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (XF == 2) { q0[e] = q0[e] * row_s[i] + row_t[i]; q1[e] = q1[e] * row_s[i] + row_t[i]; }
                    q0[e] = ok ? fmaxf(q0[e], relu_floor) : 0.f;
                    q1[e] = ok ? fmaxf(q1[e], relu_floor) : 0.f;
                }
                zs::s16::split8(q0, q1, ah[i], al[i]);
            }
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)      // transposed product: lane = pixel, registers = channels
                    zs::s16::mfma3(acc[i][j], __builtin_bit_cast(u32x4, rb[d][j][0]), __builtin_bit_cast(u32x4, rb[d][j][1]), ah[i], al[i]);
            load(d);                              // the refill goes out after the MFMAs that read the slot (same registers)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- wave partials -> LDS: register 4 q + e of lane (l32, half) = channel 8 q + 4 half + e of pixel l32 ----
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<f32x4 *>(&part[wave][32 * i + l32][32 * j + 8 * q + 4 * half]) =
                    f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    __syncthreads();
    constexpr int QPR = TN_ / 4, QUADS = TM_ * QPR;
    static_assert(QUADS % THREADS == 0, "whole passes");
    constexpr int PASSES = QUADS / THREADS;
    const int tile = tn * g.mtiles + tm;
    f32x4 v[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int e = tid + THREADS * ps, p = e / QPR, c = e % QPR;
        v[ps] = *reinterpret_cast<const f32x4 *>(&part[0][p][4 * c]);
#pragma unroll
        for (int w = 1; w < NW; w++) v[ps] += *reinterpret_cast<const f32x4 *>(&part[w][p][4 * c]);
    }
    if (g.zsplit > 1) {
        // publish this range's partial tile (plain 16-byte stores, one agent-scope release per workgroup), take a ticket;
        // the last arriver acquires and sums all ranges in range order
        float *mine = g.parts + ((size_t)tile * g.zsplit + z) * (TM_ * TN_);
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) *reinterpret_cast<f32x4 *>(mine + (size_t)(tid + THREADS * ps) * 4) = v[ps];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int old = __hip_atomic_fetch_add(&g.tickets[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == g.zsplit - 1;
            if (last) {
                __hip_atomic_store(&g.tickets[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        const float *all = g.parts + (size_t)tile * g.zsplit * (TM_ * TN_);
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            f32x4 sum = *reinterpret_cast<const f32x4 *>(all + (size_t)(tid + THREADS * ps) * 4);
            for (int zz = 1; zz < g.zsplit; zz++)
                sum += *reinterpret_cast<const f32x4 *>(all + (size_t)zz * (TM_ * TN_) + (size_t)(tid + THREADS * ps) * 4);
            v[ps] = sum;
        }
    }
    // ---- epilogue: scale / shift / residuals / activation, 16 bytes per thread; row statistics for a LayerNorm consumer ----
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int e = tid + THREADS * ps, p = e / QPR, c = e % QPR;
        const int m = m0 + p, n = n0 + 4 * c;
        const bool valid = m < a.M && n < a.Cout;
        f32x4 o = v[ps];
        if (valid) {
            const size_t off = (size_t)m * a.Cout + n;
            if (a.scale) o *= *reinterpret_cast<const f32x4 *>(a.scale + n);
            if (a.shift) o += *reinterpret_cast<const f32x4 *>(a.shift + n);
            if (a.res1) o += *reinterpret_cast<const f32x4 *>(a.res1 + off);
            if (a.res2) o += *reinterpret_cast<const f32x4 *>(a.res2 + off);
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = activate(o[k], a.act);
            *reinterpret_cast<f32x4 *>(a.out + off) = o;
        } else {
            o = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (a.fz.out_mode == 2) {               // (sum, M2 about the tile-row mean) per row and column tile
            float rs = (o[0] + o[1]) + (o[2] + o[3]);
#pragma unroll
            for (int sh = 1; sh < QPR; sh <<= 1) rs += __shfl_xor(rs, sh, 64);
            const float mean = rs * (1.0f / TN_);
            float d2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) d2 += (o[k] - mean) * (o[k] - mean);
#pragma unroll
            for (int sh = 1; sh < QPR; sh <<= 1) d2 += __shfl_xor(d2, sh, 64);
            if (valid && c == 0) {
                float *dst = a.fz.out_stats + ((size_t)m * g.ntiles + tn) * 2;
                dst[0] = rs;
                dst[1] = d2;
            }
        }
    }
}

#undef ZS_GLDS

// tile shape and K split of a problem.  Cost model from the measurements above: a launch's time ~ the bytes its busiest CU
// streams / 77 GB/s, + ~1.5 us and the partial tiles' bytes for an in-launch reduce; more than one workgroup per CU runs in
// rounds.  Candidates: 32 MI x 32 NJ tiles, NJ in {1, 2, 3} (3 only without row statistics), K split z-fold while every
// wave keeps >= 2 steps.
struct Plan { int mi, nj, zsplit, mtiles, ntiles; };

static Plan plan(long long M, int Cout, int CoutPad, int steps, int stats_nj) {
    static const int cus = getenv("ZS_STREAM_CUS") ? atoi(getenv("ZS_STREAM_CUS")) : 256;
    Plan best = {1, 2, 1, 0, 0};
    double best_cost = 1e30;
    const int mi = (M > 32 && M <= 64) ? 2 : 1;            // one row tile for 7 x 7 maps: the weights are read once
    const long long mt = (M + 32 * mi - 1) / (32 * mi);
    for (int nj = 1; nj <= 3; nj++) {
        if (stats_nj && nj != stats_nj) continue;              // row statistics: the consumer counts on zs_conv2d_fused_cols()
        const long long nt = (Cout + 32 * nj - 1) / (32 * nj), T = mt * nt;
        if (nj == 3 && (mi == 2 || nt * 96 > CoutPad)) continue;      // the weight rows are padded to 128 columns, not 96
        for (int z = 1; z <= 32; z++) {
            if (z > 1 && (steps / (z * 4) < 2 || T * z > cus)) break;
            const double rounds = (double)((T * z + cus - 1) / cus);
            const double bytes = (32.0 * mi + 32.0 * nj) * 64.0 * ((steps + z - 1) / z);      // 16 k x 4 B per step and row / column
            const double tile_bytes = 32.0 * mi * 32.0 * nj * 4.0;
            double cost = rounds * bytes / 77e3 + (z > 1 ? 1.5 + (z + 1) * tile_bytes / 77e3 : 0.0);
            cost += 0.4 * mi * nj;                              // epilogue / LDS reduction per tile
            if (cost < best_cost) { best_cost = cost; best = Plan{mi, nj, z, (int)mt, (int)nt}; }
        }
    }
    // measurement / debugging override: ZS_STREAM_FORCE="nj,z"
    if (const char *f = getenv("ZS_STREAM_FORCE")) {
        int nj = 0, z = 0;
        if (sscanf(f, "%d,%d", &nj, &z) == 2 && nj >= 1 && nj <= 3 && z >= 1) {
            const long long nt = (Cout + 32 * nj - 1) / (32 * nj);
            if (!(nj == 3 && (mi == 2 || nt * 96 > CoutPad)) && steps / (z * 4) >= 1) best = Plan{mi, nj, z, (int)mt, (int)nt};
        }
    }
    return best;
}

}  // namespace stream
