// 256 x 256 ping-pong tiles for the large pointwise layers (ViT qkv / fc1 / fc2 at batch 28, the window stage's layers of the
// transformer coordinate encoder).  Included by nn_conv.hip inside its anonymous namespace (ConvArgs, dma16, activate4_t, the
// vector types).  Design notes, the stamp history of the schedule and what was tried and dropped: DESIGN.md section 10.7.
//
// Why another GEMM kernel (round 5; profiles/r04_conv_gemm_pmc.txt): the 128 x 128 LDS-DMA kernel moves 16 KiB into LDS per
// 384 MFMA cycles - 43 B/clk/CU at full matrix rate - through a path that delivered 14 (MFMA pipe 34 % busy), and it
// fetched the activations as 64-byte row pieces: sixteen half-used cache lines per DMA instruction.  Here
//   * the tile is 256 x 256 (half the LDS bytes per MFMA: 21 B/clk/CU at full rate), eight waves of 64 x 128;
//   * a stage is K = 32: an activation row piece is one whole 128-byte line (a DMA instruction = 8 rows x 128 B), the
//     weights' k-quad rows are 4 KiB runs; two 64 KiB stages, the next one in flight under the current one's 96 MFMAs
//     per wave (3,072 matrix-pipe cycles per SIMD: longer than a loaded LDS-DMA's issue-to-landed time);
//   * the two waves of a SIMD run a segment apart ("ping-pong"): while one issues the 24 MFMAs of a K = 16 group - with its
//     DMAs of the next stage between them, in ONE asm statement - the other reads / splits the fragments of its group; two
//     workgroup barriers per stage, static priority for the group that opens an interval with MFMAs, and no wait on the DMA
//     queue closer than 1.5 intervals to the issue;
//   * workgroups are dealt to the XCDs in contiguous runs of tiles (neighbouring tiles share operand panels in ONE L2).
// Arithmetic per output element: the same three MFMAs per K = 16 group in the same k order as conv_gemm_dma_kernel -
// whole tiles are bit-identical to that kernel's results.
//
// Tail (layers whose tile count is a little more than a multiple of the 256 CUs - at M = 5,516 every ViT layer is):
// the last `tail` tiles are cut into `splits` K ranges, one workgroup each, dispatched behind the whole tiles; the ranges'
// raw partial tiles go to the workspace and pp256_tail_kernel sums them in range order (deterministic) + epilogue.
// (A last-arriver reduction inside the launch would read splits x 256 KiB serially on ONE CU: ~15 us.)
namespace pp256 {

constexpr int TM = 256, TN = 256, SK = 32, SQ = SK / 4;          // tile, K per stage, k-quads per stage
constexpr int A_BYTES = TM * SK * 4, B_BYTES = TN * SK * 4, STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int MI = 2, NJ = 4;                                     // 32 x 32 MFMA blocks per wave: 64 rows x 128 columns

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512) void conv_gemm_pp256_kernel(ConvArgs a) {
    __shared__ f32x4 lds[2 * STAGE_BYTES / 16];                   // [stage][A: [row][8 slots] | B: [k-quad][column]]
    __shared__ __attribute__((aligned(16))) float ep_lds[2][TN];   // the tile's per-channel scale | shift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                    // waves w and w + 4 share a SIMD: one of each group per SIMD
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4 *)&lds[0]);
    const int l32 = lane & 31, half = lane >> 5;
    const int wm = (wave & 3) * 64, wn = grp * 128;
    const int ntn = (a.CoutPad + TN - 1) / TN, ntm = (a.M + TM - 1) / TM, ntiles = ntm * ntn;
    const int steps_all = a.K / SK;

    // workgroup -> unit.  Units [0, full) are whole tiles, dealt so that each XCD (workgroup b runs on XCD b % 8) owns a
    // contiguous run of them (bijective for any count); units beyond are the K ranges of the tail tiles, round-robin over the
    // XCDs as dispatched: unit full + s * tail + tt = range s of tail tile tt
    const int full = ntiles - a.sk_per;                           // a.sk_per = number of tail tiles, a.splits = ranges per tail tile
    int tile, t_lo = 0, steps = steps_all, range = -1;
    if ((int)blockIdx.x < full) {
        const int xcd = (int)blockIdx.x & 7, q8 = full >> 3, r8 = full & 7;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
    } else {
        int u = (int)blockIdx.x - full;
        if (full == 0) {      // every tile is cut (fc2: 66 tiles x 3 ranges): XCD-contiguous runs here too - neighbouring tiles of ONE K
                              // range share operand panels (r05 counters: TCC hit rate 38 % with the round-robin deal)
            const int nu = (int)gridDim.x, xcd = u & 7, q8 = nu >> 3, r8 = nu & 7;
            u = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (u >> 3);
        }
        tile = full + u % a.sk_per;
        range = u / a.sk_per;
        t_lo = (int)(((long long)range * steps_all) / a.splits);
        steps = (int)(((long long)(range + 1) * steps_all) / a.splits) - t_lo;
    }
    // tile order: row-major, n fastest - an XCD's run of tiles shares few row panels, and its column panels stream in step
    const int m0 = (tile / ntn) * TM, n0 = (tile % ntn) * TN;

    {   // visible to the epilogue through the barriers of the K loop
        const int which = tid >> 8, c = tid & (TN - 1), n = n0 + c;
        const float *src = which ? a.shift : a.scale;
        ep_lds[which][c] = (src && n < a.Cout) ? src[n] : (which ? 0.0f : 1.0f);
    }

    // ---- this wave's DMA sources: A rows 32 wave + 8 j + lane / 8 (j < 4), 16-byte chunk slot ^ swizzle(row); B k-quad
    // `wave` of the stage, columns 64 c + lane (c < 4)
    // Rows beyond M re-read row M - 1 and columns beyond CoutPad the last column: finite garbage that only reaches outputs the
    // epilogue never stores (an output depends on its own row and column only) - every pointer advances by the same constant
    const char *apo[4], *bpo[4];                                  // source of piece k minus 1024 k (see mfmas_dma)
    const int a_inc = SK * 4, b_inc = SQ * a.CoutPad * 16;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = 32 * wave + 8 * j + (lane >> 3), m = min(m0 + row, a.M - 1);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        apo[j] = reinterpret_cast<const char *>(a.in + (size_t)m * a.Cin + (size_t)t_lo * SK + 4 * chunk) - 1024 * j;
        const int n = min(n0 + 64 * j + lane, a.CoutPad - 1);
        bpo[j] = reinterpret_cast<const char *>(reinterpret_cast<const f32x4 *>(a.w) + ((size_t)t_lo * SQ + wave) * a.CoutPad + n) - 1024 * j;
    }
    // this wave's eight DMAs of a stage back to back (prologue; the loop's are inside mfmas_dma).  Piece k < 4: A rows 8 k ..
    // 8 k + 7 of this wave's 32; 4 + k: columns 64 k .. of this wave's k-quad.  more = 0: the pointers stay (last stage)
    auto issue = [&](int buf, int more) {
        const unsigned da = lds_base + buf * STAGE_BYTES + wave * (32 * 128);
        const unsigned db = lds_base + buf * STAGE_BYTES + A_BYTES + wave * (TN * 16);
#pragma unroll
        for (int k = 0; k < 4; k++) {
#if !defined(ZS_EXP_PP_NO_DMA) && !defined(ZS_EXP_PP_NO_DMA_A)
            dma16(apo[k] + 1024 * k, da + k * 1024);
#endif
            apo[k] += more ? a_inc : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
#if !defined(ZS_EXP_PP_NO_DMA) && !defined(ZS_EXP_PP_NO_DMA_B)
            dma16(bpo[k] + 1024 * k, db + k * 1024);
#endif
            bpo[k] += more ? b_inc : 0;
        }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    u32x4 ah[MI], al[MI], bh[NJ], bl[NJ];
    auto frags = [&](int buf, int g) {                            // the K = 16 group g of a stage: 4 + 8 ds_read_b128, 2 split8
        const f32x4 *sa = &lds[buf * (STAGE_BYTES / 16)], *sb = sa + A_BYTES / 16;
        f32x4 fa[MI][2];
#pragma unroll
        for (int i = 0; i < MI; i++) {
            const int R = wm + 32 * i + l32, sw = (R >> 1) & 7;
            fa[i][0] = sa[R * 8 + ((4 * g + half) ^ sw)];
            fa[i][1] = sa[R * 8 + ((4 * g + half + 2) ^ sw)];
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            bh[j] = __builtin_bit_cast(u32x4, sb[(4 * g + half) * TN + wn + 32 * j + l32]);
            bl[j] = __builtin_bit_cast(u32x4, sb[(4 * g + half + 2) * TN + wn + 32 * j + l32]);
        }
#pragma unroll
        for (int i = 0; i < MI; i++) zs::s16::split8(fa[i][0], fa[i][1], ah[i], al[i]);
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm();                                              // every read of this stage has returned before the barrier that frees it
    };
    // A segment's 24 MFMAs, term-major (eight other accumulators between two MFMAs on the same one - the order hipcc picks
    // for the builtins - and per accumulator the order of zs_split16.h's mfma3), as ONE asm statement; mfmas_dma() carries
    // this wave's eight DMAs of the next stage between them, one per three MFMAs.  An LDS-DMA costs its wave 75-125 issue
    // cycles (tools/pp256_stamps.py: 600-1,000 cycles for eight back to back in front of a segment); spread out, the matrix
    // pipe's queue covers them.  Why asm: hipcc neither schedules around inline-asm DMAs nor keeps its registers when
    // sched_barrier()s pin them between MFMA builtins (1,139 spills), and a loop that mixes builtin and asm segments
    // shuffles the accumulators between two assignments (411 spills).  The instruction offset of a DMA applies to its
    // source AND its LDS address: piece k sits at M0 + 1024 k, so its source pointer is kept 1024 k low (apo / bpo).
#define PP_MFMA(C, W, X) "v_mfma_f32_32x32x16_f16 %[" #C "], %[" #W "], %[" #X "], %[" #C "]\n\t"
#define PP_DMA(P, OFF) "global_load_lds_dwordx4 %[" #P "], off" OFF "\n\t"
#define PP_ACCS [c0] "+v"(acc[0][0]), [c1] "+v"(acc[0][1]), [c2] "+v"(acc[0][2]), [c3] "+v"(acc[0][3]),   \
                [c4] "+v"(acc[1][0]), [c5] "+v"(acc[1][1]), [c6] "+v"(acc[1][2]), [c7] "+v"(acc[1][3])
#define PP_FRAGS [h0] "v"(bh[0]), [h1] "v"(bh[1]), [h2] "v"(bh[2]), [h3] "v"(bh[3]),                       \
                 [l0] "v"(bl[0]), [l1] "v"(bl[1]), [l2] "v"(bl[2]), [l3] "v"(bl[3]),                       \
                 [xh0] "v"(ah[0]), [xh1] "v"(ah[1]), [xl0] "v"(al[0]), [xl1] "v"(al[1])
    auto mfmas = [&]() {
#ifdef ZS_EXP_PP_NO_MFMA
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][j][e] += __builtin_bit_cast(float, ah[i][e] ^ al[i][e] ^ bh[j][e] ^ bl[j][e]);
#else
        asm volatile(
            PP_MFMA(c0, l0, xh0) PP_MFMA(c1, l1, xh0) PP_MFMA(c2, l2, xh0) PP_MFMA(c3, l3, xh0)       // weights lo x activations hi
            PP_MFMA(c4, l0, xh1) PP_MFMA(c5, l1, xh1) PP_MFMA(c6, l2, xh1) PP_MFMA(c7, l3, xh1)
            PP_MFMA(c0, h0, xl0) PP_MFMA(c1, h1, xl0) PP_MFMA(c2, h2, xl0) PP_MFMA(c3, h3, xl0)       // weights hi x activations lo
            PP_MFMA(c4, h0, xl1) PP_MFMA(c5, h1, xl1) PP_MFMA(c6, h2, xl1) PP_MFMA(c7, h3, xl1)
            PP_MFMA(c0, h0, xh0) PP_MFMA(c1, h1, xh0) PP_MFMA(c2, h2, xh0) PP_MFMA(c3, h3, xh0)       // weights hi x activations hi
            PP_MFMA(c4, h0, xh1) PP_MFMA(c5, h1, xh1) PP_MFMA(c6, h2, xh1) PP_MFMA(c7, h3, xh1)
            : PP_ACCS : PP_FRAGS);
#endif
    };
    auto mfmas_dma = [&](int buf, int more) {                     // more: 0 = the stage being fetched is the last one (the pointers stay)
#if defined(ZS_EXP_PP_NO_MFMA) || defined(ZS_EXP_PP_NO_DMA) || defined(ZS_EXP_PP_NO_DMA_A) || defined(ZS_EXP_PP_NO_DMA_B)
        issue(buf, more);
        mfmas();
#else
        const unsigned da = lds_base + buf * STAGE_BYTES + wave * (32 * 128);
        asm volatile(
            "s_mov_b32 m0, %[da]\n\t"
            PP_MFMA(c0, l0, xh0) PP_MFMA(c1, l1, xh0)
            PP_DMA(a0, "")
            PP_MFMA(c2, l2, xh0) PP_MFMA(c3, l3, xh0) PP_MFMA(c4, l0, xh1)
            PP_DMA(a1, " offset:1024")
            PP_MFMA(c5, l1, xh1) PP_MFMA(c6, l2, xh1) PP_MFMA(c7, l3, xh1)
            PP_DMA(a2, " offset:2048")
            PP_MFMA(c0, h0, xl0) PP_MFMA(c1, h1, xl0) PP_MFMA(c2, h2, xl0)
            PP_DMA(a3, " offset:3072")
            "s_add_u32 m0, m0, 0x8000\n\t"
            PP_MFMA(c3, h3, xl0) PP_MFMA(c4, h0, xl1) PP_MFMA(c5, h1, xl1)
            PP_DMA(b0, "")
            PP_MFMA(c6, h2, xl1) PP_MFMA(c7, h3, xl1) PP_MFMA(c0, h0, xh0)
            PP_DMA(b1, " offset:1024")
            PP_MFMA(c1, h1, xh0) PP_MFMA(c2, h2, xh0) PP_MFMA(c3, h3, xh0)
            PP_DMA(b2, " offset:2048")
            PP_MFMA(c4, h0, xh1) PP_MFMA(c5, h1, xh1) PP_MFMA(c6, h2, xh1)
            PP_DMA(b3, " offset:3072")
            PP_MFMA(c7, h3, xh1)
            : PP_ACCS
            : PP_FRAGS, [a0] "v"(apo[0]), [a1] "v"(apo[1]), [a2] "v"(apo[2]), [a3] "v"(apo[3]),
              [b0] "v"(bpo[0]), [b1] "v"(bpo[1]), [b2] "v"(bpo[2]), [b3] "v"(bpo[3]), [da] "s"(da)
            : "memory", "scc", "m0");
        const int ia = more ? a_inc : 0, ib = more ? b_inc : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            apo[k] += ia;
            bpo[k] += ib;
        }
#endif
    };
#undef PP_MFMA
#undef PP_DMA
#undef PP_ACCS
#undef PP_FRAGS

    // ---- the schedule.  Stage t lives in buffer t & 1; R = read + split the fragments of a K = 16 group, M = its 24 MFMAs.
    // Two workgroup barriers per stage, A(t) and B(t); between two barriers the waves of a SIMD do complementary work:
    //   B(t - 1) .. A(t): group 0: R(t, 0) then M(t, 0) + its DMAs of stage t + 1 | group 1: M(t - 1, 1) + its DMAs of stage t + 1, then R(t, 0)
    //   A(t) .. B(t)    : group 0: R(t, 1) then M(t, 1), wait for its DMAs        | group 1: M(t, 0), then R(t, 1), wait for its DMAs
    // (the wave that starts an interval with R finds the matrix pipe busy with its partner's M and queues behind it: an interval
    // is two M segments long, measured 1.1 x that).  Stage t + 1 goes into the buffer of stage t - 1, whose last reads are
    // before B(t - 1); every wave waits for its own DMAs of stage t + 1 before B(t), the first reads of that stage come behind it.
    // (Round-5 history, tools/pp256_stamps.py: four barriers - one per segment - cost ~250 cycles each on 768-cycle segments.)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef ZS_EXP_PP_STAMPS   // timing experiment (tools/pp256_stamps.py): shader-clock stamps of workgroup 0, stages 8-11, every wave
    unsigned *dbg = reinterpret_cast<unsigned *>(a.ws + WS_COUNTER_FLOATS) + (64u << 20) / 4;
#define PP_PHASE(k)                                                                              \
    if (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) {                                        \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                             \
        if (lane == 0) dbg[512 + ((blockIdx.x == 0 ? 0 : 8) + wave) * 4 + (k)] = (unsigned)now;  \
    }
#define PP_STAMP(k)                                                                              \
    if (blockIdx.x == 0 && t >= 8 && t < 12) {                                                   \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                             \
        if (lane == 0) dbg[(wave * 4 + (t - 8)) * 8 + (k)] = (unsigned)now;                      \
    }
#else
#define PP_STAMP(k)
#define PP_PHASE(k)
#endif
    PP_PHASE(0)
    // Past the last stage the DMAs fetch it once more, into the buffer nobody reads again: no branch around the asm segments
    if (grp == 0) {
        issue(0, steps > 1);
        wait_vm<0>();
#pragma unroll 1
        for (int t = 0; t < steps; t++) {
            const int buf = t & 1;
            bar();                                                // B(t - 1)
            PP_STAMP(0)
            frags(buf, 0);
            PP_STAMP(1)
            mfmas_dma(buf ^ 1, t + 2 < steps);                    // stage t + 1
            PP_STAMP(2)
#ifndef ZS_EXP_PP_ONE_BARRIER
            bar();                                                // A(t)
#endif
            PP_STAMP(3)
            frags(buf, 1);
            PP_STAMP(4)
            mfmas();
            PP_STAMP(5)
            wait_vm<0>();
            PP_STAMP(6)
        }
        bar();
    } else {
        // The matrix pipe serves the OLDER wave of a SIMD first (waves 0-3): behind a barrier, group 0's M segment overtook the
        // one group 1 had started and pushed group 1's R segment to the end of the interval, where nothing runs beside it
        // (stamps: 2,576-cycle intervals).  Static priority for the group that starts an interval with M restores the order.
#ifndef ZS_EXP_PP_NO_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        issue(0, steps > 1);
        issue(1, steps > 2);
        wait_vm<8>();
        bar();                                                    // B(-1)
#pragma unroll 1
        for (int t = 0; t < steps; t++) {
            const int buf = t & 1;
            PP_STAMP(0)
            frags(buf, 0);
            PP_STAMP(1)
#ifndef ZS_EXP_PP_ONE_BARRIER
            bar();                                                // A(t)
#endif
            PP_STAMP(2)
            mfmas();
            PP_STAMP(3)
            frags(buf, 1);
            PP_STAMP(4)
            wait_vm<0>();
            PP_STAMP(5)
            bar();                                                // B(t)
            PP_STAMP(6)
            mfmas_dma(buf, t + 3 < steps);                        // stage t + 2
            PP_STAMP(7)
        }
        wait_vm<0>();
    }
    PP_PHASE(1)
#undef PP_STAMP

    // ---- epilogue.  The MFMAs took the weights as their A operand: register 4 q + e of lane (l32, half) is output channel
    // 8 q + 4 half + e of row l32 of the block - four consecutive channels of one pixel, 16 bytes per lane.
    // Every load the epilogue needs is issued BEFORE the stores of its half: vmcnt counts loads and stores in one queue, so a
    // load behind a store waits for the store's acknowledgement (and hipcc cannot hoist it: `out` may alias).  The first
    // form - scale / shift / residual loaded per 16-byte piece between the stores - spent 32-38 k cycles per tile here, with
    // the stores redirected to a 4 MiB window as well (tools/pp256_stamps.py): 32 dependent store -> load round trips.
    // ... and the activation is a compile-time constant of the store loop (one switch per tile, activate4_t): decided per
    // value it cost 500 taken branches per wave through an instruction stream far larger than the instruction cache
    if (range < 0) {
        // the pointers as opaque scalar registers: hipcc otherwise re-loads a.out / a.res2 from the kernel arguments in front of
        // EVERY store (s_load + s_waitcnt lgkmcnt(0), twice per store)
        // (generic pointers: flat_ stores.  Casting them to the global address space made hipcc spill 214 registers in these
        // loops - 46.8 k cycles per tile instead of 21.5 k)
        float *out = a.out;
        const float *res1 = a.res1, *res2 = a.res2;
        asm volatile("" : "+s"(out), "+s"(res1), "+s"(res2));
        const int Cout = a.Cout;
        // FAST: the wave's 64 x 128 block lies inside the matrix and there is no second residual - no per-lane predicates
        auto store_tile = [&](auto act_tag, auto fast_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
            for (int i = 0; i < MI; i++) {
                const int m = m0 + wm + 32 * i + l32;
                const bool mok = FAST || m < a.M;
                const size_t row = (size_t)(mok ? m : 0) * Cout;
#pragma unroll
                for (int jp = 0; jp < NJ; jp += 2) {              // half a row block at a time: 8 residual quads in flight, then 8 stores
                    f32x4 r[2][4];                                // res1 (res2, if any, is read between the stores: no pointwise layer of the encoders has one)
                    if (res1) {
#pragma unroll
                        for (int j = 0; j < 2; j++)
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const int n = n0 + wn + 32 * (jp + j) + 8 * q + 4 * half;
                                r[j][q] = *reinterpret_cast<const f32x4 *>(res1 + row + (FAST || n < Cout ? n : 0));
                            }
                    }
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int c = wn + 32 * (jp + j) + 8 * q + 4 * half, n = n0 + c;
                            if (!FAST && (!mok || n >= Cout)) continue;        // Cout % 4 == 0 (checked by the launcher)
                            const f32x16 &d = acc[i][jp + j];
                            f32x4 v = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
                            v = v * *reinterpret_cast<const f32x4 *>(&ep_lds[0][c]) + *reinterpret_cast<const f32x4 *>(&ep_lds[1][c]);
                            if (res1) v += r[j][q];
                            if (!FAST && res2) v += *reinterpret_cast<const f32x4 *>(res2 + row + n);
                            activate4_t<ACT>(v);
                            *reinterpret_cast<f32x4 *>(out + row + n) = v;
                        }
                }
            }
        };
        const bool fast = m0 + wm + 64 <= a.M && n0 + wn + 128 <= Cout && !res2;      // wave-uniform
        auto run = [&](auto act_tag) {
            if (fast) store_tile(act_tag, std::true_type{}); else store_tile(act_tag, std::false_type{});
        };
        switch (a.act) {
            case ZS_ACT_RELU: run(std::integral_constant<int, ZS_ACT_RELU>{}); break;
            case ZS_ACT_GELU: run(std::integral_constant<int, ZS_ACT_GELU>{}); break;
            case ZS_ACT_RELU_CLAMP1: run(std::integral_constant<int, ZS_ACT_RELU_CLAMP1>{}); break;
            default: run(std::integral_constant<int, ZS_ACT_NONE>{}); break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_PHASE(2)
        return;
    }
    // ---- a K range of a tail tile: the raw register image to the workspace, slot (tail tile, range); pp256_tail_kernel
    // (the next launch) sums a tile's ranges in range order and runs the epilogue
    float *mine = a.ws + WS_COUNTER_FLOATS + ((size_t)(tile - full) * a.splits + range) * (TM * TN) + tid * 4;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<f32x4 *>(mine + ((i * NJ + j) * 4 + q) * 2048) =
                    f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
}

// grid (32, tail tiles) x 512 threads: block x = the register-image slot (i, j, q) of every thread of the tile's workgroup
__global__ __launch_bounds__(512) void pp256_tail_kernel(ConvArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int slot = blockIdx.x, q = slot & 3, j = (slot >> 2) % NJ, i = slot / (4 * NJ);
    const int ntn = (a.CoutPad + TN - 1) / TN, ntm = (a.M + TM - 1) / TM, full = ntm * ntn - a.sk_per;
    const int tile = full + blockIdx.y;
    const int m = (tile / ntn) * TM + (wave & 3) * 64 + 32 * i + l32;
    const int n = (tile % ntn) * TN + (wave >> 2) * 128 + 32 * j + 8 * q + 4 * half;
    if (m >= a.M || n >= a.Cout) return;
    const float *p = a.ws + WS_COUNTER_FLOATS + (size_t)blockIdx.y * a.splits * (TM * TN) + slot * 2048 + tid * 4;
    f32x4 v = *reinterpret_cast<const f32x4 *>(p);
    for (int s = 1; s < a.splits; s++) v += *reinterpret_cast<const f32x4 *>(p + (size_t)s * (TM * TN));
    const size_t o = (size_t)m * a.Cout + n;
    if (a.scale) v *= *reinterpret_cast<const f32x4 *>(a.scale + n);
    if (a.shift) v += *reinterpret_cast<const f32x4 *>(a.shift + n);
    if (a.res1) v += *reinterpret_cast<const f32x4 *>(a.res1 + o);
    if (a.res2) v += *reinterpret_cast<const f32x4 *>(a.res2 + o);
    activate4(v, a.act);
    *reinterpret_cast<f32x4 *>(a.out + o) = v;
}

}  // namespace pp256
