// 3 x 3, stride-1, pad-1 layers of the split-fp16 convolution engine with the INPUT PATCH kept in LDS across the nine taps.
// Included by nn_conv.hip inside its anonymous namespace (ConvArgs, dma16, zs_zero_page, activate, the vector types).
//
// conv_gemm_dma_kernel is bound by the bytes it moves L2 -> LDS (2.6x faster with the DMAs compiled out), and as
// an implicit GEMM it moves every input value nine times - once per tap - as a row of the A operand.  Here a workgroup
// owns a 2-D tile of output pixels of ONE image and stages, per 16-channel slab, the tile's input patch (tile + halo)
// once: the A fragments of the nine taps are the same LDS bytes read at nine shifted pixel offsets.  The patch is split
// into its fp16 halves once per slab, in place (a pixel's 16 fp32 channels = 64 B become 2 x 16 B of hi and 2 x 16 B of
// lo halves: the fragment reads fetch finished MFMA operands, where the GEMM kernel splits every value once per tap and
// consuming wave).  K is walked (slab outer, tap inner); the weights are the pre-split [K16/4][CoutPad][4] array
// addressed by (tap, slab), so nothing is re-packed.  Same products as the GEMM kernels, summed in another order.
//
// conv3x3_patch32_kernel: Cout <= 32 (DPT's 128 -> 32 head layer at 224 x 224: the GEMM kernel pads N to 128 and moves
// 16 KiB per k-step for 128 x 32 useful outputs).  Tile 16 x 16 pixels (M = 256), wave w = tile rows 4w .. 4w+3 (two
// 32 x 32 MFMA blocks), N = 32.  Per slab a stage holds the split 18 x 18 patch (20.25 KiB, padded to 21 KiB) and the nine
// taps' weights (9 x 4 row quads x 32 columns = 18 KiB); two stages = 78 KiB, two workgroups per CU.
// The weights go global -> LDS by DMA; the patch goes through REGISTERS: four lanes per pixel load its 64-byte slab (the
// DMA's coalescing), hold it across the current slab's MFMAs, split it and write the halves where the fragment reads expect
// them.  (First version: patch by DMA as raw fp32 + an in-place split pass: 515 us on the head layer - 220 us for the DMAs
// alone, i.e. the same ~8 TB/s the GEMM kernel's DMA stream saturates at, + 95 us for the split pass and its barrier, not
// overlapped with the 300 us of fragment reads and MFMAs; tools/bench_patch.py with the ZS_EXP_P32_* builds.)  Per slab:
//   barrier -> issue the next slab's patch loads (6 x 16 B per lane) and weight DMAs (18 KiB) -> nine taps x (4 A + 2 B
//   ds_read_b128, 6 MFMAs) per wave from this stage -> loads landed: split, 2 ds_write_b64 per quad into the other stage.
// Bytes into LDS per 256 x 32 outputs and slab: 39 KiB (GEMM kernel: 2 x 9 x 16 = 288 KiB).

namespace patch32 {
constexpr int PT = 16, PP = PT + 2, PPIX = PP * PP;        // tile side, patch side, patch pixels (324)
constexpr int A_PIX = 336;                                 // patch pixels padded to whole waves of the loader (21 x 16)
constexpr int A_QUADS = A_PIX * 4;                         // f32x4 slots of the (padded) patch
constexpr int A_ROUNDS = (PPIX * 4 + 255) / 256;           // loader rounds of 256 (pixel, k-quad) units: 5 full + 16 lanes
constexpr int B_DMAS = 18;                                 // 9 taps x 2 (row-quad pairs) x 32 columns x 16 B
constexpr int B_QUADS = B_DMAS * 64;
constexpr int STAGE_QUADS = A_QUADS + B_QUADS;             // 2,496 x 16 B = 39,936 B
}  // namespace patch32

// Row m (0..31) of a 32 x 32 MFMA block -> pixel 16 ty + tx of the block's 2 x 16 pixels.  ds_read_b128 serves a wave in
// the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) (MI355X_MICROARCH.md, LDS): a group must touch 64
// distinct banks.  With the obvious mapping (ty = m / 16) a group reads 8 pixels of one patch row and 8 of the next, 18
// pixels further - two of them on the same banks whatever the swizzle (2 LDS cycles per group, measured as 2x the
// fragment-read time).  Here a group IS one row of 16 consecutive patch pixels, conflict-free under slot ^= (pixel >> 2) & 3
// at any alignment: group {0-3, 12-15, 20-27} = tile row 0, x = 0..15; group {4-11, 16-19, 28-31} = tile row 1.
__device__ __forceinline__ int block_pixel(int m) {
    return m < 4 ? m : m < 12 ? 16 + m - 4 : m < 16 ? m - 8 : m < 20 ? 16 + m - 8 : m < 28 ? m - 12 : 16 + m - 16;
}

// NW waves per workgroup share the tile's eight 2 x 16-pixel MFMA blocks: NW = 8 -> one block per wave, four waves per SIMD
// with two workgroups per CU (a wave's slab is barrier, issue, 27 MFMAs, land, one after the other: the other waves of
// the SIMD are what keeps the matrix pipe busy meanwhile; four waves, two blocks each: 475 vs ... us on the head layer).
// UP2 (zs_conv3x3_tail_nhwc with ZS_CONV_IN_UPSAMPLE2; DPT's head: Interpolate(x2, bilinear, align_corners) in front of the
// 128 -> 32 layer): the layer's input is the x2 up-sampling of a.in [B][Hin][Win][Cin] (a.Hout = 2 Hin, a.Wout = 2 Win), which is
// never materialised (719 MB at 224 x 224 x 128 x 28).  An 18 x 18 patch of the up-sampled map needs at most 11 x 11 source
// pixels: ONE 16-byte load per lane and slab (instead of three) brings the source patch; landing it = raw quads to LDS (the
// start of the stage that is being filled), barrier, every lane interpolates its three (pixel, k-quad) units from there
// (upsample2x_vec_kernel's formula), barrier, split halves written over the same stage.
template <bool RELU, int NW, bool UP2 = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2, NW / 2))) void conv3x3_patch32_kernel(ConvArgs a, int tiles_x, int tiles_y) {
    using namespace patch32;
    static_assert(!UP2 || NW == 8, "the up-sampling loader is written for 512 lanes");
    constexpr int NT = 64 * NW, MB = 8 / NW;                   // threads, MFMA blocks per wave
    constexpr int SRC = 11;                                    // UP2: source patch side
    constexpr int AR = (PPIX * 4 + NT - 1) / NT;               // loader rounds of NT (pixel, k-quad) units; the last one is partial
    constexpr int BR = (B_DMAS + NW - 1) / NW;
    static_assert(AR * NT <= A_PIX * 4 + NT - 64 && (AR == 6 || AR == 3), "loader rounds");
    __shared__ f32x4 lds[2][STAGE_QUADS];
    __shared__ f32x4 tail_lds[3][8];                           // fused tail: scale | shift | tail weights of the 32 channels
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4 *)&lds[0][0]);
    const f32x4 *zero = zs_zero_page;
    if (a.tail_w && tid < 96) {                                // read by the epilogue, many barriers from here
        const int which = tid >> 5, n = tid & 31;
        const float *src = which == 0 ? a.scale : which == 1 ? a.shift : a.tail_w;
        reinterpret_cast<float *>(tail_lds)[tid] = (src && n < a.Cout) ? src[n] : (which == 0 ? 1.0f : 0.0f);
    }

    const int per_image = tiles_x * tiles_y;
    const int b = (int)blockIdx.x / per_image, trem = (int)blockIdx.x - b * per_image;
    const int y0 = (trem / tiles_x) * PT, x0 = (trem % tiles_x) * PT;
    const int slabs = a.Cin / BK;

    // A: unit u = tid + NT r = (patch pixel u / 4, k-quad u % 4): four lanes read a pixel's 64-byte slab.  The quad's four
    // values become 8 B of the hi operand and 8 B of the lo operand of lane half h = quad & 1 (its k values 4h..4h+3 first,
    // 4h+8..4h+11 second): slot h ^ ((pixel >> 2) & 3) of the pixel holds hi, that slot ^ 2 lo (conflict-free fragment
    // reads, as in conv_gemm_dma_kernel).  Halo pixels outside the image read the zero page (increment 0).
    const f32x4 *ga[AR];
    int ga_inc[AR];
    int a_dst[AR];                                             // 8-byte units inside the stage
#pragma unroll
    for (int r = 0; r < AR; r++) {
        const int u = tid + NT * r, p = u >> 2, quad = u & 3;
        const int py = p / PP, px = p - py * PP, iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = p < PPIX && iy >= 0 && iy < a.Hout && ix >= 0 && ix < a.Wout;       // (Hout == Hin unless UP2)
        ga[r] = (ok && !UP2) ? reinterpret_cast<const f32x4 *>(a.in + (((size_t)b * a.Hin + iy) * a.Win + ix) * a.Cin + 4 * quad) : zero;
        ga_inc[r] = (ok && !UP2) ? BK / 4 : 0;
        a_dst[r] = p * 8 + ((quad & 1) ^ ((p >> 2) & 3)) * 2 + (quad >> 1);
    }
    // UP2: the taps of unit (patch pixel p, k-quad) inside the SRC x SRC source patch, recomputed at every landing (cheaper than
    // nine registers held across the MFMA loop: the kernel sits at the 128-register line of four waves per SIMD)
    const float up_sy = UP2 ? (float)(a.Hin - 1) / (float)(a.Hout - 1) : 0.f, up_sx = UP2 ? (float)(a.Win - 1) / (float)(a.Wout - 1) : 0.f;
    const int up_y0 = (int)(up_sy * (float)max(y0 - 1, 0)), up_x0 = (int)(up_sx * (float)max(x0 - 1, 0));
    auto up_taps = [&](int r, int &o, int &dx, int &dy, float &ly, float &lx) -> bool {
        int t = tid;
        asm volatile("" : "+v"(t));        // opaque: otherwise the taps are hoisted out of the slab loop and held (then spilled)
        const int u = t + NT * r, p = u >> 2, py = p / PP, px = p - py * PP, iy = y0 - 1 + py, ix = x0 - 1 + px;
        const float fy = up_sy * (float)max(iy, 0), fx = up_sx * (float)max(ix, 0);
        const int ys = (int)fy, xs = (int)fx;
        ly = fy - (float)ys;
        lx = fx - (float)xs;
        o = ((ys - up_y0) * SRC + (xs - up_x0)) * 4 + (u & 3);
        dx = xs < a.Win - 1 ? 4 : 0;
        dy = ys < a.Hin - 1 ? SRC * 4 : 0;
        return p < PPIX && iy >= 0 && iy < a.Hout && ix >= 0 && ix < a.Wout;
    };
    const bool last_round = 64 * wave + NT * (AR - 1) < PPIX * 4;      // waves with units in the partial round (uniform)
    const f32x4 *gs = zero;                                    // UP2: this lane's (source pixel, k-quad) = tid
    int gs_inc = 0;
    if (UP2 && tid < SRC * SRC * 4) {
        const int sp = tid >> 2, gy = min(up_y0 + sp / SRC, a.Hin - 1), gx = min(up_x0 + sp % SRC, a.Win - 1);
        gs = reinterpret_cast<const f32x4 *>(a.in + (((size_t)b * a.Hin + gy) * a.Win + gx) * a.Cin + 4 * (tid & 3));
        gs_inc = BK / 4;
    }
    // B: DMA db = (tap, e) covers the row quads 2e, 2e + 1 (lane / 32) of the tap's K = 16 step, columns lane % 32
    const char *gb[BR];
#pragma unroll
    for (int i = 0; i < BR; i++) {
        const int db = wave + NW * i, tap = db >> 1, q = 2 * (db & 1) + (lane >> 5);
        gb[i] = reinterpret_cast<const char *>(reinterpret_cast<const f32x4 *>(a.w) + (size_t)(tap * (a.Cin >> 2) + q) * a.CoutPad + (lane & 31));
    }
    const int gb_inc = KQ * a.CoutPad * 16;

    // Workgroups start their walk over the slabs at different slabs (and wrap): a slab is the same 64 bytes of every pixel's
    // channel vector, so workgroups that run in step - they all start together - would all be reading the same few address
    // bits 6.. of every pixel at any one time, i.e. the same few memory channels.
#ifndef ZS_EXP_P32_NO_ROTATE
    int slab_at = (int)((blockIdx.x >> 3) % (unsigned)slabs);
#else
    int slab_at = 0;
#endif
#pragma unroll
    for (int r = 0; r < AR; r++) ga[r] += (size_t)slab_at * ga_inc[r];
    gs += (size_t)slab_at * gs_inc;
#pragma unroll
    for (int i = 0; i < BR; i++) gb[i] += (size_t)slab_at * gb_inc;
    f32x4 stage_a[AR];
    auto issue = [&](int stage) {                              // patch loads -> registers, weight DMAs -> LDS
        const bool wrap = ++slab_at == slabs;
        if (wrap) slab_at = 0;
        if (UP2) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stage_a[0]) : "v"(gs) : "memory");
            gs += wrap ? -(ptrdiff_t)(slabs - 1) * gs_inc : gs_inc;
        } else {
#pragma unroll
            for (int r = 0; r < AR; r++) {
                if (r < AR - 1 || last_round)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stage_a[r]) : "v"(ga[r]) : "memory");
                ga[r] += wrap ? -(ptrdiff_t)(slabs - 1) * ga_inc[r] : ga_inc[r];
            }
        }
        const unsigned dst = lds_base + stage * (STAGE_QUADS * 16) + A_QUADS * 16;
#pragma unroll
        for (int i = 0; i < BR; i++) {
            if (wave + NW * i < B_DMAS) dma16(gb[i], dst + (wave + NW * i) * 1024);
            gb[i] += wrap ? -(ptrdiff_t)(slabs - 1) * gb_inc : gb_inc;
        }
    };
    auto land = [&](int stage) {                               // loads landed: split, write the halves
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        if constexpr (UP2) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(stage_a[0]) : : "memory");
            f32x4 *scratch = &lds[stage][0];                   // SRC x SRC source pixels x 4 quads (<= 512 f32x4) at the stage's start
            scratch[tid] = stage_a[0];
            __syncthreads();
            u32x2_t hi[AR], lo[AR];
#pragma unroll
            for (int r = 0; r < AR; r++) {
                f32x4 q = {0.f, 0.f, 0.f, 0.f};
                int o, dx, dy;
                float ly, lx;
                if ((r < AR - 1 || last_round) && up_taps(r, o, dx, dy, ly, lx)) {
                    const f32x4 v00 = scratch[o], v01 = scratch[o + dx], v10 = scratch[o + dy], v11 = scratch[o + dy + dx];
                    q = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
                    if (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; e++) q[e] = fmaxf(q[e], 0.f);
                    }
                }
                unsigned h0, l0, h1, l1;
                zs::s16::split2(q.x, q.y, h0, l0);
                zs::s16::split2(q.z, q.w, h1, l1);
                hi[r] = u32x2_t{h0, h1};
                lo[r] = u32x2_t{l0, l1};
                __builtin_amdgcn_sched_barrier(0);             // one unit's sixteen tap registers at a time (no spills)
            }
            __syncthreads();                                   // every tap read before the halves overwrite the source patch
            u32x2_t *sa = reinterpret_cast<u32x2_t *>(&lds[stage][0]);
#pragma unroll
            for (int r = 0; r < AR; r++)
                if (r < AR - 1 || last_round) {
                    sa[a_dst[r]] = hi[r];
                    sa[a_dst[r] ^ 4] = lo[r];
                }
            return;
        }
        if constexpr (AR == 6)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(stage_a[0]), "+v"(stage_a[1]), "+v"(stage_a[2]), "+v"(stage_a[3]),
                         "+v"(stage_a[4]), "+v"(stage_a[5]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(stage_a[0]), "+v"(stage_a[1]), "+v"(stage_a[2]) : : "memory");
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 *sa = reinterpret_cast<u32x2 *>(&lds[stage][0]);
#pragma unroll
        for (int r = 0; r < AR; r++) {
            if (r < AR - 1 || last_round) {
                f32x4 q = stage_a[r];
                if (RELU) {
#pragma unroll
                    for (int e = 0; e < 4; e++) q[e] = fmaxf(q[e], 0.f);
                }
                unsigned h0, l0, h1, l1;
                zs::s16::split2(q.x, q.y, h0, l0);
                zs::s16::split2(q.z, q.w, h1, l1);
                sa[a_dst[r]] = u32x2{h0, h1};
                sa[a_dst[r] ^ 4] = u32x2{l0, l1};
            }
        }
    };

    f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

    // patch pixel under this lane's output pixel at tap (0, 0), per MFMA block (two tile rows, see block_pixel)
    int pb[MB];
#pragma unroll
    for (int i = 0; i < MB; i++) pb[i] = (2 * (MB * wave + i) + (block_pixel(l32) >> 4)) * PP + (block_pixel(l32) & 15);

#ifdef ZS_EXP_P32_STAMPS   // tools/stamp_patch.py: s_memtime of (block ZS_EXP_P32_STAMPS, every wave) per slab phase -> workspace
    unsigned long long *dbg = (a.ws && (int)blockIdx.x == ZS_EXP_P32_STAMPS && lane == 0)
                                  ? reinterpret_cast<unsigned long long *>(a.ws + WS_COUNTER_FLOATS) + wave * 64 : nullptr;
#define ZS_P32_STAMP(i) do { if (dbg) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ZS_P32_STAMP(i) do { } while (0)
#endif
    ZS_P32_STAMP(62);
    issue(0);
    land(0);
    for (int s = 0; s < slabs; s++) {
        const f32x4 *sa = &lds[s & 1][0];
        const f32x4 *sb = sa + A_QUADS;
        ZS_P32_STAMP(5 * s + 0);
        __syncthreads();                  // this stage is complete (own DMAs and writes, then everybody's); the other one is free
        ZS_P32_STAMP(5 * s + 1);
        if (s + 1 < slabs) issue((s + 1) & 1);
        ZS_P32_STAMP(5 * s + 2);
        // fragments of tap t + 1 are requested before the MFMAs of tap t are issued (sched_barrier keeps the scheduler from
        // sinking the reads next to their use, one wait per read)
        u32x4 ah[2][MB], al[2][MB], bh[2], bl[2];
        auto fetch = [&](int tap, int buf) {
            const int shift = (tap / 3) * PP + tap % 3;
#pragma unroll
            for (int i = 0; i < MB; i++) {
                const int p = pb[i] + shift, s0 = p * 4 + (half ^ ((p >> 2) & 3));
                ah[buf][i] = __builtin_bit_cast(u32x4, sa[s0]);
                al[buf][i] = __builtin_bit_cast(u32x4, sa[s0 ^ 2]);
            }
            bh[buf] = __builtin_bit_cast(u32x4, sb[tap * 128 + half * 32 + l32]);
            bl[buf] = __builtin_bit_cast(u32x4, sb[tap * 128 + (half + 2) * 32 + l32]);
        };
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            if (tap + 1 < 9) fetch(tap + 1, (tap + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MB; i++) zs::s16::mfma3(acc[i], bh[tap & 1], bl[tap & 1], ah[tap & 1][i], al[tap & 1][i]);   // rows = channels
            __builtin_amdgcn_sched_barrier(0);
        }
        ZS_P32_STAMP(5 * s + 3);
        if (s + 1 < slabs) land((s + 1) & 1);
        ZS_P32_STAMP(5 * s + 4);
    }

    // transposed products (weights as the MFMA's A operand): register 4 q + e of lane (l32, half) = channel 8 q + 4 half + e of
    // pixel l32 of the block; 16 bytes per lane and instruction (see conv3x3_patch128_kernel)
    {
        const int rr = block_pixel(l32);
        const bool vec = (a.Cout & 3) == 0;
#pragma unroll
        for (int i = 0; i < MB; i++) {
            const int y = y0 + 2 * (MB * wave + i) + (rr >> 4), x = x0 + (rr & 15);
            if (y >= a.Hout || x >= a.Wout) continue;
            if (a.tail_w) {
                // fused pointwise tail to one channel (zs_conv3x3_tail_nhwc): this lane's 16 channels, then the other lane
                // half's (the pixel's other 16 channels sit in lane ^ 32); channels >= Cout contribute nothing
                float t = 0.f;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x4 sc = tail_lds[0][2 * q + half], sh = tail_lds[1][2 * q + half], tw = tail_lds[2][2 * q + half];
#pragma unroll
                    for (int e = 0; e < 4; e++)                 // channels >= Cout: scale 1, shift 0, weight 0
                        t = fmaf(activate(acc[i][4 * q + e] * sc[e] + sh[e], a.act), tw[e], t);
                }
                t += __shfl_xor(t, 32, 64);
                if (half == 0)
                    a.out[((size_t)b * a.Hout + y) * a.Wout + x] = activate(t + (a.tail_b ? a.tail_b[0] : 0.f), a.tail_act);
                continue;
            }
            const size_t pix = (((size_t)b * a.Hout + y) * a.Wout + x) * a.Cout;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n = 8 * q + 4 * half;
                if (n >= a.Cout) continue;
                f32x4 v = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                if (vec) {
                    if (a.scale) v *= *reinterpret_cast<const f32x4 *>(a.scale + n);
                    if (a.shift) v += *reinterpret_cast<const f32x4 *>(a.shift + n);
                    if (a.res1) v += *reinterpret_cast<const f32x4 *>(a.res1 + pix + n);
                    if (a.res2) v += *reinterpret_cast<const f32x4 *>(a.res2 + pix + n);
                    activate4(v, a.act);
                    *reinterpret_cast<f32x4 *>(a.out + pix + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (n + e >= a.Cout) continue;
                        const size_t o = pix + n + e;
                        float t = v[e] * (a.scale ? a.scale[n + e] : 1.0f) + (a.shift ? a.shift[n + e] : 0.0f);
                        if (a.res1) t += a.res1[o];
                        if (a.res2) t += a.res2[o];
                        a.out[o] = activate(t, a.act);
                    }
                }
            }
        }
    }
    ZS_P32_STAMP(63);
#undef ZS_P32_STAMP
}

// ---- conv3x3_patch128_kernel: 128 output channels per workgroup ---------------------------------------------------- //
// The 128 x 128 tiling of conv_gemm_dma_kernel (four waves, wave tile 64 x 64, weights streamed tap by tap through a three-
// stage LDS ring by DMA, one barrier per tap-step) with the A operand taken from a patch: the workgroup's 128 rows are an
// 8 x 16-pixel tile of one image, its 10 x 18 patch of a 16-channel slab goes global -> registers -> split -> LDS once per
// slab (three 16-byte loads per lane, issued at taps 0..2 of the previous slab, landed at tap 6; two patch stages), and the
// nine taps read it at shifted offsets.  Per slab and workgroup 72 KiB of weights + 11.25 KiB of patch enter LDS; the GEMM
// kernel moves 144 KiB and splits every A value once per tap.  48 KiB of LDS, three workgroups per CU.
namespace patch128 {
constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PPIX = PH * PW;   // tile, patch (180 pixels)
constexpr int A_PIX = 192;                                                 // padded to the loader's 3 rounds of 256 units
constexpr int A_QUADS = A_PIX * 4;
constexpr int B_QUADS = KQ * BN;                                           // one tap-step of weights: 4 row quads x 128 columns
constexpr int NS = 3, DEPTH = NS - 1;
}  // namespace patch128

template <bool RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void conv3x3_patch128_kernel(ConvArgs a, int tiles_x, int tiles_y) {
    using namespace patch128;
    __shared__ f32x4 lds_a[2][A_QUADS];
    __shared__ f32x4 lds_b[NS][B_QUADS];
    __shared__ __attribute__((aligned(16))) float ep_lds[2][BN];   // per-channel scale | shift of this column tile, for the epilogue
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ldsb_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4 *)&lds_b[0][0]);
    const f32x4 *zero = zs_zero_page;

    const int per_image = tiles_x * tiles_y;
    // a.sk_per = number of column tiles when the launch is 1-D and XCD-aware: workgroup L runs on XCD L % 8 (round-robin
    // dispatch), so the column tiles of one pixel tile are given consecutive slots of ONE XCD - the second one finds the patch in
    // that XCD's L2 - and consecutive pixel tiles (shared halos) go round the XCDs as before.  0: grid (pixel tiles, column tiles)
    int mt = (int)blockIdx.x, nt = (int)blockIdx.y;
    if (a.sk_per > 0) {
        const int k = (int)blockIdx.x >> 3;
        nt = k % a.sk_per;
        mt = (k / a.sk_per) * 8 + ((int)blockIdx.x & 7);
        if (mt >= a.B * per_image) return;
    }
    const int b = mt / per_image, trem = mt - b * per_image;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int n0 = nt * BN;
    {   // read by the epilogue, many barriers from here
        const int which = tid >> 7, c = tid & (BN - 1), n = n0 + c;
        const float *src = which ? a.shift : a.scale;
        ep_lds[which][c] = (src && n < a.Cout) ? src[n] : (which ? 0.0f : 1.0f);
    }
    // split-K (a.splits > 1, layers of few tiles - 14 x 14 maps, batch 1): blockIdx.z owns a contiguous range of the slabs and
    // writes its raw partial tile to ws[z][pixel][cout]; conv_splitk_reduce_kernel sums the ranges in order (deterministic)
    const int all_slabs = a.Cin / BK;
    const int slab_lo = (int)(((long long)blockIdx.z * all_slabs) / a.splits);
    const int slabs = (int)(((long long)(blockIdx.z + 1) * all_slabs) / a.splits) - slab_lo;
    const int wm = wave & 1, wn = (wave >> 1) * 64;            // wave tile: tile rows 4 wm .. 4 wm + 3, columns wn .. wn + 63

    // patch loader: unit u = tid + 256 r = (patch pixel u / 4, k-quad u % 4), as in conv3x3_patch32_kernel
    const f32x4 *ga[3];
    int ga_inc[3], a_dst[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int u = tid + 256 * r, p = u >> 2, quad = u & 3;
        const int py = p / PW, px = p - py * PW, iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = p < PPIX && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        ga[r] = ok ? reinterpret_cast<const f32x4 *>(a.in + (((size_t)b * a.Hin + iy) * a.Win + ix) * a.Cin + 4 * quad) : zero;
        ga_inc[r] = ok ? BK / 4 : 0;
        a_dst[r] = p * 8 + ((quad & 1) ^ ((p >> 2) & 3)) * 2 + (quad >> 1);
    }
    // weights: per tap-step this wave fetches row quad `wave` of the step, columns n0 + lane and n0 + 64 + lane
    const f32x4 *gb = reinterpret_cast<const f32x4 *>(a.w) + (size_t)wave * a.CoutPad + n0 + lane;
    const size_t b_tap = (size_t)(a.Cin >> 2) * a.CoutPad, b_slab = (size_t)KQ * a.CoutPad;     // f32x4 units
    // workgroups start at different slabs of their range and wrap (see conv3x3_patch32_kernel); slab_at counts inside the range
    int slab_at = (int)(((unsigned)mt >> 3) % (unsigned)slabs);
#pragma unroll
    for (int r = 0; r < 3; r++) ga[r] += (size_t)(slab_lo + slab_at) * ga_inc[r];
    gb += (size_t)(slab_lo + slab_at) * b_slab;

    f32x4 stage_a[3];
    auto load_a = [&](int r) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stage_a[r]) : "v"(ga[r]) : "memory"); };
    int a_at = slab_at;                                        // the slab the patch pointers are at (one ahead of the weights)
    auto next_slab_a = [&]() {                                 // after the three loads of a slab: pointers to the following one
        const bool wrap = a_at + 1 == slabs;
#pragma unroll
        for (int r = 0; r < 3; r++) ga[r] += wrap ? -(ptrdiff_t)(slabs - 1) * ga_inc[r] : ga_inc[r];
        a_at = wrap ? 0 : a_at + 1;
    };
    auto land_a = [&](int stage) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 *sa = reinterpret_cast<u32x2 *>(&lds_a[stage][0]);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            f32x4 q = stage_a[r];
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 4; e++) q[e] = fmaxf(q[e], 0.f);
            }
            unsigned h0, l0, h1, l1;
            zs::s16::split2(q.x, q.y, h0, l0);
            zs::s16::split2(q.z, q.w, h1, l1);
            sa[a_dst[r]] = u32x2{h0, h1};
            sa[a_dst[r] ^ 4] = u32x2{l0, l1};
        }
    };
    auto dma_b = [&](const f32x4 *g, int stage, bool live) {   // 2 DMAs: this wave's row quad of a step
        const unsigned dst = ldsb_base + stage * (B_QUADS * 16) + wave * (BN * 16);
        dma16(live ? reinterpret_cast<const char *>(g) : reinterpret_cast<const char *>(zero), dst);
        dma16(live ? reinterpret_cast<const char *>(g + 64) : reinterpret_cast<const char *>(zero), dst + 1024);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    int pb[2];
#pragma unroll
    for (int i = 0; i < 2; i++) pb[i] = (2 * (2 * wm + i) + (block_pixel(l32) >> 4)) * PW + (block_pixel(l32) & 15);

    // prologue: patch of the first slab, weights of its first two taps
#pragma unroll
    for (int r = 0; r < 3; r++) load_a(r);
    next_slab_a();
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(stage_a[0]), "+v"(stage_a[1]), "+v"(stage_a[2]) : : "memory");
    land_a(0);
    dma_b(gb, 0, true);
    dma_b(gb + b_tap, 1, true);

    for (int s = 0; s < slabs; s++) {
        const f32x4 *sa = &lds_a[s & 1][0];
        const bool last = s + 1 == slabs;
        const bool wrap = slab_at + 1 == slabs;
        const f32x4 *gb_next = gb + (wrap ? -(ptrdiff_t)(slabs - 1) * (ptrdiff_t)b_slab : (ptrdiff_t)b_slab);
        auto tap_step = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            // vector-memory operations younger than the DMAs of this step: the patch load of tap T - 2 (if it had one), the
            // two DMAs and the patch load of tap T - 1
            constexpr int YOUNGER = ((T >= 2 && T - 2 < 3) ? 1 : 0) + 2 + ((T >= 1 && T - 1 < 3) ? 1 : 0);
            if (T == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(YOUNGER) : "memory");   // + the patch writes of tap 6
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(YOUNGER) : "memory");
            // the step two ahead: taps T + 2 of this slab, or taps 0 / 1 of the next one
            if (T + DEPTH < 9) dma_b(gb + (size_t)(T + DEPTH) * b_tap, (T + DEPTH) % NS, true);
            else dma_b(gb_next + (size_t)(T + DEPTH - 9) * b_tap, (T + DEPTH) % NS, !last);
            if (T < 3) load_a(T);                              // next slab's patch (the last slab re-reads its own: the counts stay)
            if (T == 2) next_slab_a();
            if (T == 6) {
                asm volatile("" : "+v"(stage_a[0]), "+v"(stage_a[1]), "+v"(stage_a[2]));   // loads retired by this step's wait
                if (!last) land_a((s + 1) & 1);
            }
            // (reading the patch fragments above the barrier - they do not depend on it - costs six registers, the third wave
            // per SIMD with them, and was slower: 384 vs 359 us on the 256 -> 256 layer at 56 x 56 x 28)
            const f32x4 *sb = &lds_b[T % NS][0];
            const int shift = (T / 3) * PW + T % 3;
            u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int p = pb[i] + shift, s0 = p * 4 + (half ^ ((p >> 2) & 3));
                ah[i] = __builtin_bit_cast(u32x4, sa[s0]);
                al[i] = __builtin_bit_cast(u32x4, sa[s0 ^ 2]);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                bh[j] = __builtin_bit_cast(u32x4, sb[half * BN + wn + 32 * j + l32]);
                bl[j] = __builtin_bit_cast(u32x4, sb[(half + 2) * BN + wn + 32 * j + l32]);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) zs::s16::mfma3(acc[i][j], bh[j], bl[j], ah[i], al[i]);   // D = W^T A^T: rows = channels
        };
        tap_step(std::integral_constant<int, 0>());
        tap_step(std::integral_constant<int, 1>());
        tap_step(std::integral_constant<int, 2>());
        tap_step(std::integral_constant<int, 3>());
        tap_step(std::integral_constant<int, 4>());
        tap_step(std::integral_constant<int, 5>());
        tap_step(std::integral_constant<int, 6>());
        tap_step(std::integral_constant<int, 7>());
        tap_step(std::integral_constant<int, 8>());
        gb = gb_next;
        slab_at = wrap ? 0 : slab_at + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the zero-page DMAs and the spare patch loads of the last slab

    // The products are formed transposed (weights as the MFMA's A operand): accumulator register 4 q + e of lane (l32, half) is
    // channel 8 q + 4 half + e of pixel l32 - four consecutive channels of ONE pixel per register quad, so the epilogue moves
    // 16 bytes per lane and instruction (residuals in, result out) instead of 4: a quarter of the vector-memory instructions,
    // which is what the tiles' last phase queues behind (tools/stamp_patch.py).
    float *part = a.splits > 1 ? a.ws + WS_COUNTER_FLOATS + (size_t)blockIdx.z * a.M * a.Cout : nullptr;
    const bool vec = (a.Cout & 3) == 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int rr = block_pixel(l32);
        const int y = y0 + 2 * (2 * wm + i) + (rr >> 4), x = x0 + (rr & 15);
        if (y >= a.Hout || x >= a.Wout) continue;
        const size_t pix = (((size_t)b * a.Hout + y) * a.Wout + x) * a.Cout;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n = n0 + wn + 32 * j + 8 * q + 4 * half;
                if (n >= a.Cout) continue;
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                if (vec) {                                     // n + 3 < Cout as well
                    if (part) { *reinterpret_cast<f32x4 *>(part + pix + n) = v; continue; }
                    v = v * *reinterpret_cast<const f32x4 *>(&ep_lds[0][n - n0]) + *reinterpret_cast<const f32x4 *>(&ep_lds[1][n - n0]);
                    if (a.res1) v += *reinterpret_cast<const f32x4 *>(a.res1 + pix + n);
                    if (a.res2) v += *reinterpret_cast<const f32x4 *>(a.res2 + pix + n);
                    activate4(v, a.act);
                    *reinterpret_cast<f32x4 *>(a.out + pix + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (n + e >= a.Cout) continue;
                        const size_t o = pix + n + e;
                        if (part) { part[o] = v[e]; continue; }
                        float t = v[e] * (a.scale ? a.scale[n + e] : 1.0f) + (a.shift ? a.shift[n + e] : 0.0f);
                        if (a.res1) t += a.res1[o];
                        if (a.res2) t += a.res2[o];
                        a.out[o] = activate(t, a.act);
                    }
                }
            }
    }
}
