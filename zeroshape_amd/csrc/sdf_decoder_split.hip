// Split-fp16 ("f16x3") variant of the fused implicit-occupancy decoder for MI355X (gfx950).
//
// Same algorithm, same per-point dataflow and the same decoder program as csrc/sdf_decoder.hip
// (Implicit.forward, model/shape/implicit.py:251-288, hoisted latent half in the prologue), but
// every contraction runs on the 16-bit matrix pipe (v_mfma_f32_32x32x16_f16, 16x the rate of the
// fp32 MFMA) with both operands split into two fp16 halves, x ~= hi + lo:
//     A B  ~=  A_hi B_hi + A_hi B_lo + A_lo B_hi          (fp32 accumulation; A_lo B_lo dropped)
// 3 MFMAs of 32 cycles per K = 16 instead of 8 fp32 MFMAs of 64 cycles: 5.3x fewer matrix
// cycles at ~2^-22 relative operand error (fp32 itself is 1e-6 from fp64 on the seeded network; the
// same scheme on bf16 halves, same cost, measured 1.9e-5 and 10 flips; plain bf16 8e-3).  Halves
// are rounded to nearest even (v_cvt_pk_f16_f32, round 3; rounds 1-2 truncated with
// v_cvt_pkrtz_f16_f32, whose one-signed errors accumulate - csrc/zs_split16.h has the numbers):
// |x| < 65,520 keeps the full precision, far beyond what LayerNorm-ed activations and these weights
// reach (beyond it hi is inf and the result NaN); values below ~1e-4 lose relative (not absolute)
// precision to fp16 subnormals.  tests/test_gpu_decoder_split.py.
//
// What changes against the fp32 kernel, and why:
//  * The transposed chain survives: accumulator registers 8j..8j+7 of a 32x32 output tile are,
//    after a split into packed (hi, lo) fp16 pairs, exactly the B operand of K-block j of the
//    next layer.  The matching A operand of K-block j of a 32x32 weight unit is the fp32
//    program's records 8j..8j+7 of that unit, split the same way - so the split program has the
//    SAME unit order and byte size as the fp32 one and is derived from it on the device
//    (split_program_kernel), K/V records of the image included.
//  * Weight stream.  At this rate four waves streaming private copies of the 10 MB program would
//    pull ~50 TB/s through the vector memory path (64 B/clk/CU, L2 34 TB/s).  The four waves of a
//    workgroup run the same program in lock step, so each 16 KiB chunk (8 K-blocks) is staged ONCE
//    per workgroup into LDS by LDS-DMA (global_load_lds_dwordx4, two K-blocks per wave, two
//    buffers) and read back as A operands with conflict-free ds_read_b128 by all four waves.
//    One raw s_barrier per chunk (AStream below).  Ordering rules: cdna_hip_programming.md
//    section 5 (counted vmcnt by the issuing wave, then a barrier the reader has passed; restage
//    after an lgkmcnt-retired read + barrier).
//  * LDS: [params window 16 KiB][A staging 2 x 16 KiB][4 x 28 KiB activation slabs] = 160 KiB
//    (tile 7 of a slab-resident array lives in registers).  Params are paged in five windows
//    per tile instead of two.
//  * No asm register ring: LDS reads and MFMAs are builtins, scheduled and hazard-padded by
//    hipcc (VGPR-form accumulators: -mllvm -amdgpu-mfma-vgpr-form, zeroshape_amd/build.py); only
//    the DMA, the barrier and the prefetch pin are asm.
//  * Activations of tile t (GELU, softplus, operand split) are fed value by value between the
//    MFMA groups of tile t+1 (software pipelines over the hidden tiles of the MLP - the split
//    program permutes those K-blocks accordingly - and over the output tiles of impl_mlp).
//  * The attention map (implicit.py:277) is only produced by the fp32 kernel.
#include "zs_common.h"
#include <stdlib.h>
#include "sdf_layout.h"
#include "sdf_math.h"
#include "zs_split16.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>
#include <type_traits>

namespace {

using namespace zs::lay;
using zs::dm::gelu_erf;
using zs::dm::softplus100;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float S_GUARD = 64.0f;  // program.py: S_GUARD
constexpr int WAVES = 4;
constexpr int PTS_PER_WAVE = 32;
constexpr int PTS_PER_BLOCK = WAVES * PTS_PER_WAVE;
constexpr int MAX_WGS = 256;                        // one persistent workgroup per CU
constexpr int KB_U4 = 128;                          // one K-block: [hi: 64 lanes x 16 B][lo: 64 lanes x 16 B]
constexpr int CK = 8;                               // K-blocks per staged chunk: two per wave
constexpr int KPW = CK / WAVES;                     // K-blocks a wave stages per chunk
constexpr int CHUNK_BYTES = CK * KB_U4 * 16;        // 16 KiB
constexpr int NBUF = 2;
constexpr int KB_TOTAL = G_TOTAL / 2;               // 4,928 K-blocks = 14,784 MFMAs per wave tile
constexpr int PRM_WINDOW = 4096;                    // floats of params resident in LDS at a time
constexpr int STAGE_FLOATS = NBUF * CHUNK_BYTES / 4;
constexpr int SLAB_TILES = NT - 1;                  // tiles of an activation array kept in LDS (the last: registers)
constexpr int SLAB_U4 = SLAB_TILES * 4 * 64;        // ... as packed (hi, lo) K-blocks: 28 KiB per wave
constexpr int ZTILES_F4 = NT * 4 * 64;              // one skip layer's fp32 feat partial products (workspace)
constexpr int ZSLAB_F4 = 3 * ZTILES_F4;
constexpr int LDS_FLOATS = PRM_WINDOW + STAGE_FLOATS + WAVES * SLAB_U4 * 4;
static_assert(LDS_FLOATS * 4 == 160 * 1024, "the kernel owns the whole LDS of a CU");
static_assert(KB_TOTAL % CK == 0 && CK % WAVES == 0 && KPW * 2048 <= 4096, "chunking");
// the stream prefetches NBUF chunks past its position: the last reads run into the zero tail of
// the records and (harmlessly) the params section behind it
static_assert((RING * GROUP_FLOATS + PARAM_FLOATS) * 4 >= NBUF * CHUNK_BYTES, "prefetch stays inside the program");

// params windows (floats, relative to the params section of the program; <= PRM_WINDOW each)
constexpr int W_PP = P_PP;                                   // point_proj table (1,024)
constexpr int W_BLK = P_BLK0;                                // + blk * P_BLK_STRIDE: one attention block (3,328)
constexpr int W_IA = P_LNFG;                                 // final norm, impl layers 0, 1, pair 0 (3,072)
constexpr int W_IB = P_IMPL_PAIR + P_IMPL_PAIR_STRIDE;       // pairs 1, 2, layer 8 (2,832)
static_assert(P_BLK_STRIDE <= PRM_WINDOW && W_IB - W_IA <= PRM_WINDOW && P_USED - W_IB <= PRM_WINDOW,
              "params windows");

#define DEV __device__ __forceinline__
// fence between the MFMAs of a K-block and the side work placed in their dependency gaps
// (ZS_FENCE_SCHED: a hard scheduling barrier; default: source order only, which hipcc keeps)
#ifdef ZS_FENCE_SCHED
#define ZS_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define ZS_FENCE do { } while (0)
#endif

// pin(ahi, alo): the prefetch reads issued above stay above (memory clobber) and the MFMAs that
// consume this K-block's A operand stay below (they read the statement's outputs).  Without it
// hipcc hoists MFMAs over the prefetch or sinks the prefetch down to its first use as soon as
// independent VALU work is around - either way the LDS latency is exposed at every K-block.
DEV void pin(u32x4 &a, u32x4 &b) { asm volatile("" : "+v"(a), "+v"(b) : : "memory"); }

// as_h / pk_f16 / split2 (x -> packed fp16 hi, lo: v_cvt_pk_f16_f32 + v_fma_mix_f32, 2 VALU per
// value) / mfma3: csrc/zs_split16.h, shared with the convolution engine
using zs::s16::mfma3;
using zs::s16::split2;

// a 32-feature x 32-point activation tile as the B operands of its two K-blocks:
// v[2 j + 0] = hi, v[2 j + 1] = lo of K-block j (accumulator registers 8 j .. 8 j + 7)
struct PT {
    u32x4 v[4];
};
template <typename T>
DEV PT pack_tile(const T &x) {
    PT p;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned h, l;
            split2(x[8 * j + 2 * i], x[8 * j + 2 * i + 1], h, l);
            p.v[2 * j][i] = h;
            p.v[2 * j + 1][i] = l;
        }
    return p;
}

// One K-block (hi and lo halves, 2 x 64 lanes x 16 B): global -> LDS[lds_dst + 16 lane (+ 1024)].
// The instruction offset moves the global and the LDS address alike.  M0 (the LDS-DMA destination
// base) is not saved: hipcc has no use for it in this kernel (no LDS-direct, GWS, movrel or
// interpolation instructions; tools/check_split_isa.py audits the ISA for it).
DEV void glds_kblocks(const char *gsrc, unsigned lds_dst) {  // this wave's two K-blocks of a chunk
#ifdef ZS_EXP_NO_DMA   // energy ablation (tools/energy_ablation.sh): results are garbage, the timing is the point
    if (gsrc != nullptr) return;
#endif
#ifndef ZS_DMA_MOD      // cache-policy modifiers of the weight stream's loads (experiments: " nt", " sc1", ...)
#define ZS_DMA_MOD ""
#endif
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ZS_DMA_MOD
                 "\n\tglobal_load_lds_dwordx4 %0, off offset:1024" ZS_DMA_MOD
                 "\n\tglobal_load_lds_dwordx4 %0, off offset:2048" ZS_DMA_MOD
                 "\n\tglobal_load_lds_dwordx4 %0, off offset:3072" ZS_DMA_MOD
                 :
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// ---- weight stream: LDS-DMA staged chunks shared by the four waves ------------------ //
// Two buffers of 8 K-blocks: while chunk c is consumed, chunk c+1 lands.  One synchronisation per
// chunk, before its LAST K-block is multiplied (that K-block's A operand is already in registers):
// own share of chunk c+1 landed (vmcnt) and own reads of chunk c retired (lgkmcnt) -> s_barrier ->
// chunk c+2 is staged into the buffer just freed and the first A read of chunk c+1 is issued.
// Lead time of a chunk: one chunk period = 8 K-blocks.
// tools/ubench/mfma_f16_stream.hip prices the pieces (cycles per K-block of 3 MFMAs, one wave
// per SIMD, all CUs busy, 4-K-block chunks): MFMAs alone 96.1; + two ds_read_b128 105.5; + the
// barrier every four K-blocks 124.1 (~74 cycles per barrier: the skew of four waves); + the
// LDS-DMA 131.6; + two more ds_read_b128 for B operands from the slab 146.9.
// Measured alternatives on the whole kernel (129^3 grid): 4-K-block chunks: two buffers (lead 4
// K-blocks) +6 ms - the DMA latency is exposed at every barrier - three buffers the baseline of
// the 8-K-block version; register staging (global_load -> ds_write_b128 behind the barrier) +3
// ms; the DMAs moved from behind the barrier into the shadows of the first MFMAs of the chunk
// +0.5 ms; reads two K-blocks ahead +0.6 ms; two alternating accumulator chains +2 ms (extra
// adds; a single chain already issues back to back).
struct AStream {
    const u32x4 *buf[NBUF];   // this lane's view of the buffers, buf[0] = chunk being consumed
    unsigned dst[NBUF];       // LDS byte address of this wave's K-blocks in each (wave-uniform)
    u32x4 hi, lo;             // A operand of the next K-block (read in flight)
    const char *gsrc;         // this lane's source of this wave's K-blocks of the next chunk to stage

    DEV void init(const char *prog, u32x4 *stage, unsigned stage_addr, int wave, int lane) {
        const char *g = prog + wave * (KPW * KB_U4 * 16) + lane * 16;
        // the previous tile's reads and DMAs are retired in every wave before the buffers are reused
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < NBUF; i++) {
            buf[i] = stage + i * (CK * KB_U4) + lane;
            dst[i] = stage_addr + i * CHUNK_BYTES + wave * (KPW * KB_U4 * 16);
            glds_kblocks(g + i * CHUNK_BYTES, dst[i]);
        }
        gsrc = g + NBUF * CHUNK_BYTES;
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");  // chunk 0 landed everywhere
        hi = buf[0][0];
        lo = buf[0][64];
    }
    // A operand of the K-block at position `pos` (0..7) of the current chunk; issues the read of
    // the next.  EXTRA_VM: vector-memory loads the caller has issued since the previous
    // synchronisation and does not need yet (they are younger than the DMAs waited for here:
    // the counted wait lets them stay in flight).
    template <int EXTRA_VM = 0>
    DEV void step(int pos, u32x4 &ahi, u32x4 &alo) {
        ahi = hi;
        alo = lo;
        if (pos == CK - 1) {
            static_assert(EXTRA_VM == 0 || EXTRA_VM == 4, "wait-count variants");
            if (EXTRA_VM == 4)
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            glds_kblocks(gsrc, dst[0]);
            gsrc += CHUNK_BYTES;
            const u32x4 *t = buf[0];
            const unsigned u = dst[0];
            buf[0] = buf[1];
            dst[0] = dst[1];
            buf[1] = t;
            dst[1] = u;
            hi = buf[0][0];
            lo = buf[0][64];
        } else {
#ifndef ZS_EXP_NO_AREAD   // energy ablation: one A read per chunk instead of eight
            hi = buf[0][(pos + 1) * KB_U4];
            lo = buf[0][(pos + 1) * KB_U4 + 64];
#endif
        }
    }
    DEV void drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
};

// The LDS-resident activation array of a wave: tiles 0..6 in the slab, tile 7 in registers (those
// 4 KiB per wave are what lets the staging buffers hold 8 K-blocks, i.e. half the barriers).
struct Slab {
    u32x4 *fl;  // this lane's view: [(tile * 4 + j * 2 + hl) * 64]
    PT t7;
    DEV void store(int tile, const PT &p) {
        if (tile == SLAB_TILES) {
            t7 = p;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) fl[(tile * 4 + q) * 64] = p.v[q];
        }
    }
};

// One K-block = zs::s16::mfma3: three dependent MFMAs on one accumulator.  A dependent
// v_mfma_f32_32x32x16_f16 issues 32 cycles behind its producer (tools/ubench/mfma_f16_stream.hip:
// one chain and two alternating chains both run 96.1 cycles per K-block), so a second chain buys
// nothing.
DEV f32x16 zero16() {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = 0.f;
    return v;
}

// ---- side work in the dependency gaps ---------------------------------------------------- //
// The three MFMAs of a K-block form a dependent chain (one accumulator): the 2nd and 3rd each wait
// ~32 cycles for their predecessor, and an in-order wave issues NOTHING behind a waiting MFMA.
// hipcc keeps source order (MFMA MFMA MFMA, then the activation code), so round 1's side work ran
// with only the third MFMA in flight.  tools/ubench/mfma_valu_place.hip (cycles per K-block, ideal
// 96): 12 plain VALU behind the three MFMAs 126.6, the same 12 as 4 + 4 + 4 in the gaps 101.6;
// 18: 138.6 vs 113.1; but 24: 157 either way, and a transcendental shares a gap with at most one
// plain instruction (exp + 1: 99.6; exp + 2: 124.6; exp + 3: 135-156).  Hence: side(kb, g) is
// called behind the g-th MFMA of K-block kb, fenced by sched_barrier(0) so hipcc cannot regroup
// it, and the activations are cut into gap-sized stages (GeluStager, SoftplusStager below).
struct NoSide {
    DEV void operator()(int, int) const {}
};
template <typename SIDE>
DEV void mfma3_gaps(f32x16 &acc, const u32x4 &ahi, const u32x4 &alo, const u32x4 &bhi, const u32x4 &blo, int kb,
                    SIDE &side) {
    using zs::s16::as_h;
    if (std::is_same<SIDE, NoSide>::value) {   // nothing to place: leave the three MFMAs to hipcc
        mfma3(acc, ahi, alo, bhi, blo);
        return;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(alo), as_h(bhi), acc, 0, 0, 0);
    ZS_FENCE;
    side(kb, 0);
    ZS_FENCE;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ahi), as_h(blo), acc, 0, 0, 0);
    ZS_FENCE;
    side(kb, 1);
    ZS_FENCE;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ahi), as_h(bhi), acc, 0, 0, 0);
    ZS_FENCE;
    side(kb, 2);
    ZS_FENCE;
}

// Collects the split B operands of an activation tile, one value at a time (index 0..15)
struct TilePacker {
    PT p;
    float prev;
    DEV void feed(int kb, float v) {
        if (kb & 1) {
            unsigned h, l;
            split2(prev, v, h, l);
            p.v[2 * (kb >> 3)][(kb & 7) >> 1] = h;
            p.v[2 * (kb >> 3) + 1][(kb & 7) >> 1] = l;
        }
        prev = v;
    }
};

// zs::dm::gelu_erf (same operations in the same order: bit-identical) cut into five stages:
// 3 plain | 1 plain + rcp | exp + 1 plain | 5 plain | 2 plain
struct GeluStager {
    float x, u, q, t, e, p;
    DEV void s0(float v) {
        x = v;
        u = fabsf(v) * 0.70710678118654752440f;
        q = (u * u) * -1.44269504088896340736f;
    }
    DEV void s1() { t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f)); }
    DEV void s2() {
        e = __builtin_amdgcn_exp2f(q);
        p = fmaf(0.75052702f, t, -1.02753365f);
    }
    DEV void s3() {
        p = fmaf(p, t, 1.00509130f);
        p = fmaf(p, t, -0.20116957f);
        p = fmaf(p, t, 0.18019173f);
        p = p * t;
        u = u * p;
    }
    DEV float s4() const { return fmaf(-u, e, fmaxf(x, 0.0f)); }
    DEV float all(float v) {
        s0(v); s1(); s2(); s3();
        return s4();
    }
    // stage st (0..5; 5 = idle) of value `v`; the finished value goes to e.feed(idx, .)
    DEV void stage(int st, float v, int idx, TilePacker &pk) {
        if (st == 0) s0(v);
        else if (st == 1) s1();
        else if (st == 2) s2();
        else if (st == 3) s3();
        else if (st == 4) pk.feed(idx, s4());
    }
};
// zs::dm::softplus100 in three stages: 1 plain + exp | 1 plain + log | 2 plain
struct SoftplusStager {
    float x, t;
    DEV void s0(float v) {
        x = v;
        t = __builtin_amdgcn_exp2f(fabsf(v) * -144.26950408889634074f);
    }
    DEV void s1() { t = __builtin_amdgcn_logf(1.0f + t); }
    DEV float s2() const { return fmaf(t, 0.0069314718055994530942f, fmaxf(x, 0.0f)); }
};

// K-blocks with the activation of ONE value written into the dependency gaps by hand (inline asm:
// hipcc neither places side work there by itself nor survives being fenced into it - register
// spills, and its AGPR-copy rewrite pass crashes).  Same operations in the same order as
// zs::dm::gelu_erf / softplus100: bit-identical results.  The A / B operands are ordinary asm inputs
// (hipcc waits for their LDS reads in front of the statement); the VALU part touches no MFMA operand.
// gelu: 4 plain | rcp, exp, 1 plain | 7 plain
DEV float kblock_gelu(f32x16 &acc, const u32x4 &ahi, const u32x4 &alo, const u32x4 &bhi, const u32x4 &blo, float x) {
    float u, q, d, t, e, m, p;
    const float c0 = 0.70710678118654752440f, c1 = 0.3275911f, c2 = -1.02753365f;
    asm volatile(
#ifndef ZS_EXP_TWO_TERM
                 "v_mfma_f32_32x32x16_f16 %[acc], %[alo], %[bh], %[acc]\n\t"
#endif
                 "v_mul_f32_e64 %[u], |%[x]|, %[c0]\n\t"
                 "v_mul_f32_e32 %[q], %[u], %[u]\n\t"
                 "v_mul_f32_e32 %[q], 0xbfb8aa3b, %[q]\n\t"
                 "v_fma_f32 %[d], %[u], %[c1], 1.0\n\t"
                 "v_mfma_f32_32x32x16_f16 %[acc], %[ahi], %[bl], %[acc]\n\t"
                 "v_rcp_f32_e32 %[t], %[d]\n\t"
                 "v_exp_f32_e32 %[e], %[q]\n\t"
                 "v_max_f32_e32 %[m], 0, %[x]\n\t"
                 "v_mfma_f32_32x32x16_f16 %[acc], %[ahi], %[bh], %[acc]\n\t"
                 "v_fmamk_f32 %[p], %[t], 0x3f40228a, %[c2]\n\t"
                 "v_fmaak_f32 %[p], %[p], %[t], 0x3f80a6d5\n\t"
                 "v_fmaak_f32 %[p], %[p], %[t], 0xbe4dff65\n\t"
                 "v_fmaak_f32 %[p], %[p], %[t], 0x3e38842e\n\t"
                 "v_mul_f32_e64 %[p], %[t], -%[p]\n\t"
                 "v_mul_f32_e32 %[p], %[u], %[p]\n\t"
                 "v_fmac_f32_e32 %[m], %[p], %[e]"
                 : [acc] "+v"(acc), [u] "=&v"(u), [q] "=&v"(q), [d] "=&v"(d), [t] "=&v"(t), [e] "=&v"(e), [m] "=&v"(m),
                   [p] "=&v"(p)
                 : [alo] "v"(alo), [ahi] "v"(ahi), [bh] "v"(bhi), [bl] "v"(blo), [x] "v"(x), [c0] "s"(c0), [c1] "s"(c1),
                   [c2] "v"(c2));
    return m;
}
// softplus100: 1 plain + exp | 2 plain + log | 1 plain
DEV float kblock_softplus(f32x16 &acc, const u32x4 &ahi, const u32x4 &alo, const u32x4 &bhi, const u32x4 &blo, float x) {
    float t, m;
    const float c0 = -144.26950408889634074f;
    asm volatile(
#ifndef ZS_EXP_TWO_TERM
                 "v_mfma_f32_32x32x16_f16 %[acc], %[alo], %[bh], %[acc]\n\t"
#endif
                 "v_mul_f32_e64 %[t], |%[x]|, %[c0]\n\t"
                 "v_exp_f32_e32 %[t], %[t]\n\t"
                 "v_mfma_f32_32x32x16_f16 %[acc], %[ahi], %[bl], %[acc]\n\t"
                 "v_max_f32_e32 %[m], 0, %[x]\n\t"
                 "v_add_f32_e32 %[t], 1.0, %[t]\n\t"
                 "v_log_f32_e32 %[t], %[t]\n\t"
                 "v_mfma_f32_32x32x16_f16 %[acc], %[ahi], %[bh], %[acc]\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f32_e32 %[m], 0x3be32166, %[t]"
                 : [acc] "+v"(acc), [t] "=&v"(t), [m] "=&v"(m)
                 : [alo] "v"(alo), [ahi] "v"(ahi), [bh] "v"(bhi), [bl] "v"(blo), [x] "v"(x), [c0] "s"(c0));
    return m;
}

// acc += W_tile X, X = KT packed tiles in registers; starts at a chunk boundary.
// SIDE: side(kb, g) runs behind the g-th MFMA of K-block kb
template <int KT, int EXTRA_VM = 0, typename SIDE = NoSide>
DEV void gemm_reg(AStream &s, const PT *X, f32x16 &acc, SIDE side = SIDE()) {  // starts chunk-aligned
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            u32x4 ahi, alo;
            s.step((kt * 2 + j) & (CK - 1), ahi, alo);
            pin(ahi, alo);
            mfma3_gaps(acc, ahi, alo, X[kt].v[2 * j], X[kt].v[2 * j + 1], kt * 2 + j, side);
        }
}
// ... with X pinned in the ACCUMULATOR half of the register file (B operands of an MFMA may be AGPRs).
// For an array that lives through a whole phase and is only ever a B operand - feat / sqrt(2) in
// impl_mlp - this tells hipcc where it belongs: left to itself it shuffled such arrays between the
// two halves (hundreds of v_accvgpr moves per phase) and spilled.  Hazards hipcc does not pad for
// asm: the A operands arrive through its own s_waitcnt (they are asm inputs), the leading s_nop
// covers a VALU write of the accumulator just before the statement.
template <typename SIDE>
DEV void mfma3_breg(f32x16 &acc, const u32x4 &ahi, const u32x4 &alo, const u32x4 &bhi, const u32x4 &blo, int kb,
                    SIDE &side) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(alo), "a"(bhi));
    ZS_FENCE;
    side(kb, 0);
    ZS_FENCE;
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ahi), "a"(blo));
    ZS_FENCE;
    side(kb, 1);
    ZS_FENCE;
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ahi), "a"(bhi));
    ZS_FENCE;
    side(kb, 2);
    ZS_FENCE;
}
template <int KT, typename SIDE = NoSide>
DEV void gemm_areg(AStream &s, const PT *X, f32x16 &acc, SIDE side = SIDE()) {  // starts chunk-aligned
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            u32x4 ahi, alo;
            s.step((kt * 2 + j) & (CK - 1), ahi, alo);
            pin(ahi, alo);
            mfma3_breg(acc, ahi, alo, X[kt].v[2 * j], X[kt].v[2 * j + 1], kt * 2 + j, side);
        }
}
// ... with the ACCUMULATOR pinned in the accumulator half of the register file: the residual stream
// y (128 registers, only ever updated by MFMAs and read by the LayerNorms) belongs there.  Left to
// hipcc (-amdgpu-mfma-vgpr-form) every update of a y tile moved it to VGPRs and back, and with
// side work between those MFMAs its "Rewrite AGPR-Copy-MFMA" pass crashes (ROCm 7.2).
template <typename SIDE>
DEV void mfma3_cacc(f32x16 &acc, const u32x4 &ahi, const u32x4 &alo, const u32x4 &bhi, const u32x4 &blo, int kb,
                    SIDE &side) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(alo), "v"(bhi));
    ZS_FENCE;
    side(kb, 0);
    ZS_FENCE;
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(ahi), "v"(blo));
    ZS_FENCE;
    side(kb, 1);
    ZS_FENCE;
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(ahi), "v"(bhi));
    ZS_FENCE;
    side(kb, 2);
    ZS_FENCE;
}
// one input tile (2 K-blocks) starting at chunk position pos0 (even); side sees K-blocks kb0, kb0 + 1.
// YACC: acc is a tile of the residual stream (AGPR-resident, see mfma3_cacc)
template <bool YACC = false, typename SIDE = NoSide>
DEV void gemm_one(AStream &s, const PT &X, f32x16 &acc, int pos0, int kb0 = 0, SIDE side = SIDE()) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
        u32x4 ahi, alo;
        s.step((pos0 + j) & (CK - 1), ahi, alo);
        pin(ahi, alo);
        if (YACC)
            mfma3_cacc(acc, ahi, alo, X.v[2 * j], X.v[2 * j + 1], kb0 + j, side);
        else
            mfma3_gaps(acc, ahi, alo, X.v[2 * j], X.v[2 * j + 1], kb0 + j, side);
    }
}
// 8 input tiles with the B operands read from the wave's LDS slab ([k-block][hi | lo][lane]),
// one K-block ahead of their use
// B operand of K-block kb of the slab-resident array (tile 7 lives in registers)
DEV void slab_b(const Slab &sl, int kb, u32x4 &bh, u32x4 &bl) {
    if (kb >= SLAB_TILES * 2) {
        bh = sl.t7.v[(kb & 1) * 2];
        bl = sl.t7.v[(kb & 1) * 2 + 1];
    } else {
        bh = sl.fl[kb * KB_U4];
        bl = sl.fl[kb * KB_U4 + 64];
    }
}
// pos0: chunk position of the first K-block (0 or 4: a head consumes 92 K-blocks)
template <typename SIDE = NoSide>
DEV void gemm_lds(AStream &s, const Slab &sl, f32x16 &acc, int pos0 = 0, SIDE side = SIDE()) {
    u32x4 bh, bl;
    slab_b(sl, 0, bh, bl);
#pragma unroll
    for (int kb = 0; kb < NT * 2; kb++) {
        u32x4 ahi, alo;
        s.step((pos0 + kb) & (CK - 1), ahi, alo);
        u32x4 nh = bh, nl = bl;
        if (kb + 1 < NT * 2) slab_b(sl, kb + 1, nh, nl);  // after the step: its lgkmcnt(0) must not wait for this read
        pin(ahi, alo);
        mfma3_gaps(acc, ahi, alo, bh, bl, kb, side);
        bh = nh;
        bl = nl;
    }
}

// gemm_lds / gemm_reg with the activation ACT (0 gelu, 1 softplus100) of cur[kb] computed in the gaps of
// K-block kb (kblock_* above); fin(kb, value) receives the result, tail(kb) runs behind the K-block
template <int ACT, typename FIN, typename TAIL>
DEV void gemm_lds_act(AStream &s, const Slab &sl, f32x16 &acc, const f32x16 &cur, FIN fin, TAIL tail) {
    u32x4 bh, bl;
    slab_b(sl, 0, bh, bl);
#pragma unroll
    for (int kb = 0; kb < NT * 2; kb++) {
        u32x4 ahi, alo;
        s.step(kb & (CK - 1), ahi, alo);
        u32x4 nh = bh, nl = bl;
        if (kb + 1 < NT * 2) slab_b(sl, kb + 1, nh, nl);
        const float r = ACT == 0 ? kblock_gelu(acc, ahi, alo, bh, bl, cur[kb]) : kblock_softplus(acc, ahi, alo, bh, bl, cur[kb]);
        bh = nh;
        bl = nl;
        fin(kb, r);
        tail(kb);
    }
}
template <int ACT, typename FIN, typename TAIL>
DEV void gemm_reg_act(AStream &s, const PT *X, f32x16 &acc, const f32x16 &cur, FIN fin, TAIL tail) {
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            u32x4 ahi, alo;
            const int kb = kt * 2 + j;
            s.step(kb & (CK - 1), ahi, alo);
            const float r = ACT == 0 ? kblock_gelu(acc, ahi, alo, X[kt].v[2 * j], X[kt].v[2 * j + 1], cur[kb])
                                     : kblock_softplus(acc, ahi, alo, X[kt].v[2 * j], X[kt].v[2 * j + 1], cur[kb]);
            fin(kb, r);
            tail(kb);
        }
}

// Sum / maximum of a value over the two lane halves of a point (lanes l and l ^ 32), on every lane:
// __shfl_xor(v, 32) = ds_bpermute, a ~150-cycle LDS round trip exposed on a lone wave, 60-70 times per wave tile.
// gfx950 has v_permlane32_swap: with two copies (a, b) of v it leaves a = [lo | lo], b = [hi | hi], so the sum is a + b
// without LDS.  Round 2 measured 1.2 % of the cycles for it and could not make it correct; round 3 pinned it down
// (tools/ubench/permlane_check.hip, 4 M lanes per form; tools/ab_permlane.py, tools/race_screen_split.py):
//   * __builtin_amdgcn_permlane32_swap is MISCOMPILED by this hipcc (ROCm 7.2): the generated code adds the first
//     result to itself (r[0] + r[0]) - every lane wrong, with or without wait states in front.  That is round 2's
//     "upper halves come back un-swapped";
//   * the asm form with `s_nop 4` on both sides (operands are two fresh v_mov copies, which hipcc pads against MFMA
//     results itself) is correct in isolation - 0 mismatches for VALU- and MFMA-produced values - and inside this kernel
//     the full grid stays 3.46e-6 from the fp32 kernel, the same maximum as the LDS form.  But identical launches are
//     then no longer BIT-reproducible: 0.6 % of the values differ in their last bits from run to run (40 repeats, 1.1 M
//     of 172 M values; also with only the sums switched over), where the LDS form differs in none.  The cause is not
//     identified (suspect: the swap issued under an MFMA that is still in flight); a fused decoder whose occupancy
//     indices are a parity contract cannot ship that.
//   * the gain it would buy: 28.66 vs 28.80 ms per grid (0.5 %) - the kernel's time is its energy.
// So the LDS form stays; -DZS_SPLIT_PERMLANE builds the asm form for whoever continues (ZS_SPLIT_PERMLANE_SUM_ONLY: sums only).
#ifndef ZS_SPLIT_PERMLANE
DEV float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }
DEV float half_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
#else
DEV void half_pair(float v, float &lo, float &hi) {
    lo = v;
    hi = v;
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(lo), "+v"(hi));
}
DEV float half_sum(float v) {
    float lo, hi;
    half_pair(v, lo, hi);
    return lo + hi;
}
#ifdef ZS_SPLIT_PERMLANE_SUM_ONLY
DEV float half_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
#else
DEV float half_max(float v) {
    float lo, hi;
    half_pair(v, lo, hi);
    return fmaxf(lo, hi);
}
#endif
#endif

// row-param read from LDS: 16 floats for (tile, lane half)
DEV void rp(const float *prm, int off, int tile, int hi, float *v) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(prm + off + tile * 32 + hi * 16);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f32x4 a = q[i];
        v[4 * i + 0] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
    }
}
DEV f32x16 rp16(const float *prm, int off, int tile, int hi) {
    float t[16];
    rp(prm, off, tile, hi, t);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = t[r];
    return v;
}
// w.w + w.x*x + w.y*y + w.z*z for the 16 registers of (tile, hi): [tile][hi][r][4] table
DEV f32x16 xyz_affine(const float *prm, int off, int tile, int hi, float x, float y, float z) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(prm + off + tile * 128 + hi * 64);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const f32x4 w = q[r];
        v[r] = fmaf(w.z, z, fmaf(w.y, y, fmaf(w.x, x, w.w)));
    }
    return v;
}

// Accumulator initialisers of the NEXT output tile, produced one step per K-block beside the current
// tile's GEMM (gemm side work).  A wave alone on its SIMD has nobody to hide an LDS round trip
// behind, and hipcc cannot hoist these reads over the stream's asm statements: the read of step
// r + 1 is issued in step r, ~100 cycles before its use.  (Round 1 read the whole table in front of
// every tile: 16 serialised round trips, ~1,000 cycles per tile with the matrix pipe idle.)
struct XyzInit {  // w.w + w.x x + w.y y + w.z z for the 16 registers of (tile, hi): [tile][hi][r][4] table
    const f32x4 *q;
    f32x4 w;
    f32x16 v;
    DEV void start(const float *prm, int off, int tile, int hi) {
        q = reinterpret_cast<const f32x4 *>(prm + off + tile * 128 + hi * 64);
        w = q[0];
    }
    DEV void step(int r, float x, float y, float z) {
        v[r] = fmaf(w.z, z, fmaf(w.y, y, fmaf(w.x, x, w.w)));
        if (r + 1 < 16) w = q[r + 1];
    }
};
struct RowInit {  // 16 floats of a row-param vector (bias) for (tile, hi)
    const f32x4 *q;
    f32x4 w;
    f32x16 v;
    DEV void start(const float *prm, int off, int tile, int hi) {
        q = reinterpret_cast<const f32x4 *>(prm + off + tile * 32 + hi * 16);
        w = q[0];
    }
    DEV void step(int r) {
        if ((r & 3) == 0) {
            v[r] = w.x; v[r + 1] = w.y; v[r + 2] = w.z; v[r + 3] = w.w;
            if (r + 4 < 16) w = q[(r >> 2) + 1];
        }
    }
};

DEV void ln_stats(const f32x16 *x, float &mean, float &rstd) {
    float s = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += x[kt][r];
    s = half_sum(s);
    mean = s * (1.0f / 256.0f);
    float v = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float d = x[kt][r] - mean;
            v = fmaf(d, d, v);
        }
    v = half_sum(v);
    rstd = 1.0f / sqrtf(v * (1.0f / 256.0f) + 1e-6f);
}
// LayerNorm -> packed B operands in the wave's LDS slab
DEV void layer_norm_lds(const f32x16 *x, Slab &sl, const float *prm, int g_off, int b_off, int hi) {
    float mean, rstd;
    ln_stats(x, mean, rstd);
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        float g[16], b[16], t[16];
        rp(prm, g_off, kt, hi, g);
        rp(prm, b_off, kt, hi, b);
#pragma unroll
        for (int r = 0; r < 16; r++) t[r] = fmaf((x[kt][r] - mean) * rstd, g[r], b[r]);
        sl.store(kt, pack_tile(t));
    }
}

// one wave-uniform window of the params section -> LDS (all four waves)
DEV void load_params(float *prm, const float *prog_params, int start, int count) {
    __syncthreads();  // everyone is done with the previous window
    const f32x4 *src = reinterpret_cast<const f32x4 *>(prog_params + start);
    f32x4 *dst = reinterpret_cast<f32x4 *>(prm);
    for (int i = threadIdx.x; i < count / 4; i += WAVES * 64) dst[i] = src[i];
    __syncthreads();
}

// One latent tile of the point->latent attention of one head: S = K_tile q (2 K-blocks), online
// softmax update in the log2 domain, o += V_tile^T P (2 K-blocks).  MASK: last tile, rows >= 197
// are padding.
template <bool MASK>
DEV void attn_tile(AStream &s, const PT &q, f32x16 &o, float &m_run, float &z_run, float c, int hi,
                   int pos0) {
    f32x16 S = zero16();
    gemm_one(s, q, S, pos0);
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (MASK) {
            const int rw = (r & 3) + 8 * (r >> 2) + 4 * hi;
            S[r] = rw < L - 32 * (LT - 1) ? S[r] : -INFINITY;
        }
        mt = fmaxf(mt, S[r]);
    }
    mt *= c;
    // Lazy reference: the running maximum (shared by the two lane halves of a point) is only moved -
    // with the exchange of the halves' maxima through LDS, the rescale factor and the rescaling of o -
    // when some logit of the wave exceeds it by more than 2^LAZY; otherwise the probabilities of this
    // tile are taken relative to the old reference (p <= 2^8: far inside the fp16 halves' range and
    // exact to the same relative precision).  After the first tile that is the rule, and it saves the
    // ~150-cycle LDS round trip and ~20 VALU instructions per tile.
    constexpr float LAZY = 8.0f;
    if (__builtin_amdgcn_ballot_w64(mt > m_run + LAZY) != 0) {   // wave-uniform
        mt = half_max(mt);
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        z_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] *= alpha;
        m_run = m_new;
    }
    float zs_ = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(S[r], c, -m_run));
        S[r] = p;
        zs_ += p;
    }
    z_run += zs_;
    gemm_one(s, pack_tile(S), o, pos0 + 2);
}

#ifdef ZS_EXP_TIMING  // tools/phase_timing_split.py: cycle stamps of (block 0, wave 0, first tile) -> workspace tail
#define ZS_STAMP(i) do { if (dbg) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ZS_STAMP(i) do { } while (0)
#endif
// One wave: 32 points (lane & 31; both lane halves carry the same point).
DEV float decode_tile(const char *prog, float *prm, u32x4 *stage, unsigned stage_addr, u32x4 *slab,
                      f32x4 *zs, float px, float py, float pz, int wave, int lane, float &guard,
                      unsigned long long *dbg) {
    const int hi = lane >> 5;
    const float *prog_params = reinterpret_cast<const float *>(prog) + REC_FLOATS;
    AStream s;
    s.init(prog, stage, stage_addr, wave, lane);
    Slab sl;
    sl.fl = slab;
    ZS_STAMP(0);

    // point_proj (implicit.py:128-131); y is the residual stream, kept as accumulators
    load_params(prm, prog_params, W_PP, 1024);
    f32x16 y[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) y[kt] = xyz_affine(prm, 0, kt, hi, px, py, pz);

    ZS_STAMP(1);

    const float scale = 0.17677669529663688110f;  // 32 ** -0.5

#pragma unroll 1
    for (int blk = 0; blk < BLOCKS; blk++) {
        load_params(prm, prog_params, W_BLK + blk * P_BLK_STRIDE, P_BLK_STRIDE);
        layer_norm_lds(y, sl, prm, PB_LN1G, PB_LN1B, hi);
        ZS_STAMP(2 + blk * 4);
        // y = x + proj_bias + sum_heads Wproj_h o_h
#pragma unroll
        for (int nt = 0; nt < NT; nt++) y[nt] += rp16(prm, PB_BPROJ, nt, hi);

        // one head: 48 (q, k, v) + 28 (7 latent tiles) + 16 (proj) = 92 K-blocks = 11.5 chunks, so
        // heads alternate between chunk positions 0 and 4 (P); all positions are compile-time
        auto head = [&](int hd, auto Ptag) {
            constexpr int P = decltype(Ptag)::value;
            f32x16 q = rp16(prm, PB_BQKV, hd * 3 + 0, hi);
            gemm_lds(s, sl, q, P);
            f32x16 k = rp16(prm, PB_BQKV, hd * 3 + 1, hi);
            gemm_lds(s, sl, k, P);
            f32x16 v = rp16(prm, PB_BQKV, hd * 3 + 2, hi);
            gemm_lds(s, sl, v, P);

            // logits are kept in the log2 domain: c = d^-1/2 * log2(e), softmax = 2^(c s - m)
            const float c = scale * 1.44269504088896340736f;
            float s_self = 0.f;  // self logit (implicit.py:44), fp32
            float qq = 0.f;      // |q|^2: envelope guard, see ENVELOPE below
#pragma unroll
            for (int r = 0; r < 16; r++) {
                s_self = fmaf(q[r], k[r], s_self);
                qq = fmaf(q[r], q[r], qq);
            }
            s_self = half_sum(s_self) * c;
            {   // (scale |q| max_l |k_l|)^2, the Cauchy-Schwarz bound of this head's latent logits
                const float km = prog_params[P_KMAX + blk * HEADS + hd] * scale;  // scalar load
                guard = fmaxf(guard, half_sum(qq) * (km * km));
            }
            const PT qp = pack_tile(q);

            float m_run = -INFINITY, z_run = 0.f;
            f32x16 o = zero16();
#pragma unroll 1
            for (int l2 = 0; l2 < (LT - 1) / 2; l2++) {
                attn_tile<false>(s, qp, o, m_run, z_run, c, hi, P);
                attn_tile<false>(s, qp, o, m_run, z_run, c, hi, (P + 4) & (CK - 1));
            }
            attn_tile<true>(s, qp, o, m_run, z_run, c, hi, P);
            {
                const float m_new = fmaxf(m_run, s_self);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                const float p_self = __builtin_amdgcn_exp2f(s_self - m_new);
                const float z = fmaf(half_sum(z_run), alpha, p_self);
                const float inv = 1.0f / z;
                const float a_i = alpha * inv, p_i = p_self * inv;
#pragma unroll
                for (int r = 0; r < 16; r++) o[r] = fmaf(p_i, v[r], o[r] * a_i);
            }
            // y += Wproj[:, head] o_h
            const PT op = pack_tile(o);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) gemm_one(s, op, y[nt], (P + 4 + 2 * nt) & (CK - 1));
        };
        static_assert((LT - 1) % 2 == 0 && (3 * NT * 2 + LT * 4 + NT * 2) % CK == 4, "head = 11.5 chunks");
#pragma unroll 1
        for (int hd = 0; hd < HEADS; hd += 2) {
            head(hd, std::integral_constant<int, 0>());
            head(hd + 1, std::integral_constant<int, 4>());
        }

        ZS_STAMP(3 + blk * 4);
        // MLP (timm Mlp): y += b2 + W2 gelu(W1 LN2(y) + b1), one hidden tile at a time
        layer_norm_lds(y, sl, prm, PB_LN2G, PB_LN2B, hi);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) y[nt] += rp16(prm, PB_B2, nt, hi);
        ZS_STAMP(4 + blk * 4);
        // software pipeline over the hidden tiles (stream order: fc1(0), [fc1(t+1), fc2(t)]..., fc2(31)):
        // fc1 of tile t+1 carries the GELU + operand split of tile t and the bias fetch of tile t+2
        RowInit bi;
        f32x16 hid = rp16(prm, PB_B1, 0, hi);
        bi.start(prm, PB_B1, 1, hi);
        gemm_lds(s, sl, hid, 0, [&](int kb, int g) {
            if (g == 2 && (kb & 3) == 1) bi.step(kb & ~3);
        });
#pragma unroll 1
        for (int ht = 0; ht < HT - 1; ht++) {
            f32x16 nxt = bi.v;
            bi.start(prm, PB_B1, ht + 2 < HT ? ht + 2 : HT - 1, hi);
            TilePacker e;
            gemm_lds_act<0>(s, sl, nxt, hid, [&](int kb, float r) { e.feed(kb, r); },
                            [&](int kb) { if ((kb & 3) == 1) bi.step(kb & ~3); });
#pragma unroll
            for (int nt = 0; nt < NT; nt++) gemm_one(s, e.p, y[nt], (2 * nt) & (CK - 1));
            hid = nxt;
        }
        {
#pragma unroll
            for (int r = 0; r < 16; r++) hid[r] = gelu_erf(hid[r]);
            const PT hp = pack_tile(hid);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) gemm_one(s, hp, y[nt], (2 * nt) & (CK - 1));
        }
        ZS_STAMP(5 + blk * 4);
    }

    // final norm (implicit.py:275) -> feat, fp32 in registers
    load_params(prm, prog_params, W_IA, W_IB - W_IA);
    float h[NT * 16];
    {
        float mean, rstd;
        ln_stats(y, mean, rstd);
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            float g[16], b[16];
            rp(prm, P_LNFG - W_IA, kt, hi, g);
            rp(prm, P_LNFB - W_IA, kt, hi, b);
#pragma unroll
            for (int r = 0; r < 16; r++) h[kt * 16 + r] = fmaf((y[kt][r] - mean) * rstd, g[r], b[r]);
        }
    }
    ZS_STAMP(10);
    PT hp[NT];

    // impl_mlp (implicit.py:168-184): inputs = cat[xyz, feat].  Layer 0: feat (regs) -> LDS
#pragma unroll
    for (int kt = 0; kt < NT; kt++) hp[kt] = pack_tile(h + kt * 16);
    // (software pipeline over the output tiles: tile nt's GEMM carries tile nt-1's activation)
    {
        f32x16 prev;
        XyzInit ini;
        ini.v = xyz_affine(prm, P_IMPL0 - W_IA, 0, hi, px, py, pz);
#pragma unroll
        for (int nt = 0; nt <= NT; nt++) {
            TilePacker e;
            if (nt < NT) {
                f32x16 acc = ini.v;
                if (nt + 1 < NT) ini.start(prm, P_IMPL0 - W_IA, nt + 1, hi);
                if (nt > 0)
                    gemm_reg_act<1>(s, hp, acc, prev, [&](int kb, float r) { e.feed(kb, r); },
                                    [&](int kb) { if (nt + 1 < NT) ini.step(kb, px, py, pz); });
                else
                    gemm_reg<NT, 0>(s, hp, acc, [&](int kb, int g) { if (g == 2) ini.step(kb, px, py, pz); });
                if (nt > 0) sl.store(nt - 1, e.p);
                prev = acc;
            } else {
#pragma unroll
                for (int kb = 0; kb < 16; kb++) e.feed(kb, softplus100(prev[kb]));
                sl.store(nt - 1, e.p);
            }
        }
    }
    ZS_STAMP(11);
    // the skip layers consume cat[x, xyz, feat] / sqrt(2): feat / sqrt(2) stays packed in registers
    // (the residual stream is dead by now) and is contracted inside each skip layer as 16 more
    // K-blocks per output tile - the split stream interleaves [x part | feat part] per tile
    // (split_source_kblock) - so nothing is parked in memory (round 1 parked 96 KiB of fp32
    // partial products per wave tile: 12.8 GB per 129^3 launch)
    const float rsqrt2 = 0.70710678118654752440f;
    PT fp[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; r++) t[r] = h[kt * 16 + r] * rsqrt2;
        fp[kt] = pack_tile(t);
    }
    const float sx = px * rsqrt2, sy = py * rsqrt2, sz = pz * rsqrt2;

    ZS_STAMP(12);
    // layer 1 (plain): LDS -> registers, pre-divided by sqrt(2) because layer 2 is a skip layer
    {
        f32x16 prev;
        RowInit ini;
        ini.v = rp16(prm, P_IMPL1 - W_IA, 0, hi);
#pragma unroll
        for (int nt = 0; nt <= NT; nt++) {
            TilePacker e;
            if (nt < NT) {
                f32x16 acc = ini.v;
                if (nt + 1 < NT) ini.start(prm, P_IMPL1 - W_IA, nt + 1, hi);
                if (nt > 0)
                    gemm_lds_act<1>(s, sl, acc, prev, [&](int kb, float r) { e.feed(kb, r * rsqrt2); },
                                    [&](int kb) { if (nt + 1 < NT) ini.step(kb); });
                else
                    gemm_lds(s, sl, acc, 0, [&](int kb, int g) { if (g == 2) ini.step(kb); });
                if (nt > 0) hp[nt - 1] = e.p;
                prev = acc;
            } else {
#pragma unroll
                for (int kb = 0; kb < 16; kb++) e.feed(kb, softplus100(prev[kb]) * rsqrt2);
                hp[nt - 1] = e.p;
            }
        }
    }
    ZS_STAMP(13);
    float out = 0.f;
#pragma unroll 1
    for (int i = 0; i < 3; i++) {
        if (i == 1) load_params(prm, prog_params, W_IB, P_USED - W_IB);
        const int pp = i == 0 ? P_IMPL_PAIR - W_IA : (i - 1) * P_IMPL_PAIR_STRIDE;
        // skip layer 2+2i: registers (x / sqrt(2)) and feat / sqrt(2) -> LDS
        {
            f32x16 prev;
            XyzInit ini;
            ini.v = xyz_affine(prm, pp, 0, hi, sx, sy, sz);
#pragma unroll
            for (int nt = 0; nt <= NT; nt++) {
                TilePacker e;
                if (nt < NT) {
                    f32x16 acc = ini.v;
                    if (nt + 1 < NT) ini.start(prm, pp, nt + 1, hi);
                    // the x part carries the previous tile's activation, the feat part the next tile's initialiser
                    if (nt > 0)
                        gemm_reg_act<1>(s, hp, acc, prev, [&](int kb, float r) { e.feed(kb, r); }, [&](int) {});
                    else
                        gemm_reg<NT>(s, hp, acc);
                    gemm_areg<NT>(s, fp, acc, [&](int kb, int g) {
                        if (g == 2 && nt + 1 < NT) ini.step(kb, sx, sy, sz);
                    });
                    if (nt > 0) sl.store(nt - 1, e.p);
                    prev = acc;
                } else {
#pragma unroll
                    for (int kb = 0; kb < 16; kb++) e.feed(kb, softplus100(prev[kb]));
                    sl.store(nt - 1, e.p);
                }
            }
        }
        // plain layer 3+2i: LDS -> registers (/ sqrt(2) when the next layer is a skip layer);
        // the last one feeds layer 8 (256 -> 1), evaluated in fp32 on the spot
        const float post = i < 2 ? rsqrt2 : 1.0f;
        {
            f32x16 prev;
            RowInit ini, w8;  // bias of the next tile; layer 8 weights of the previous tile (last pair only)
            const float wsel = i == 2 ? 1.0f : 0.0f;
            ini.v = rp16(prm, pp + 1024, 0, hi);
#pragma unroll
            for (int r = 0; r < 16; r++) w8.v[r] = 0.f;
#pragma unroll
            for (int nt = 0; nt <= NT; nt++) {
                TilePacker e;
                // weights of tile nt - 1, fetched beside the previous GEMM; pairs 0 and 1 read whatever
                // finite parameters sit at that offset of their window and scale them away (a
                // branch on i here crashes hipcc's AGPR-copy rewrite pass, ROCm 7.2)
                const f32x16 w = w8.v * wsel;
                auto finish = [&](int kb, float v) {
                    const float t = v * post;
                    e.feed(kb, t);
                    out = fmaf(t, w[kb], out);
                };
                if (nt < NT) {
                    f32x16 acc = ini.v;
                    if (nt + 1 < NT) ini.start(prm, pp + 1024, nt + 1, hi);
                    w8.start(prm, P_W8 - W_IB, nt, hi);
                    auto tail = [&](int kb) {
                        if (nt + 1 < NT) ini.step(kb);
                        w8.step(kb);
                    };
                    if (nt > 0)
                        gemm_lds_act<1>(s, sl, acc, prev, [&](int kb, float r) { finish(kb, r); }, tail);
                    else
                        gemm_lds(s, sl, acc, 0, [&](int kb, int g) { if (g == 2) tail(kb); });
                    if (nt > 0) hp[nt - 1] = e.p;
                    prev = acc;
                } else {
#pragma unroll
                    for (int kb = 0; kb < 16; kb++) finish(kb, softplus100(prev[kb]));
                    hp[nt - 1] = e.p;
                }
            }
        }
    }
    ZS_STAMP(14);
    s.drain();
    out = half_sum(out);
    ZS_STAMP(15);
    return out + prm[P_B8 - W_IB];
}

template <bool GRID>
__global__ __launch_bounds__(WAVES * 64, 1) void sdf_decode_split_kernel(
    const char *__restrict__ programs, size_t program_stride_bytes, int batch,
    const float *__restrict__ points,  // !GRID: [batch][m][3]
    const float *__restrict__ axis,    //  GRID: [G]
    int G, long long first_point,      //  GRID: linear index of the first grid point
    int m,                             // points per image handled by this launch
    float *__restrict__ out, int apply_sigmoid, f32x4 *__restrict__ workspace,
    int *__restrict__ tile_flags,    // [tiles] zeroed by the caller, or null
    int static_order) {              // 1: tile += gridDim.x, no counter (ZS_SPLIT_STATIC_TILES=1: the A/B arm of tools/ab_tile_order.py)
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float *prm = lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4 *stage = reinterpret_cast<u32x4 *>(lds + PRM_WINDOW);
    const unsigned stage_addr = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(lds + PRM_WINDOW));
    u32x4 *fl = reinterpret_cast<u32x4 *>(lds + PRM_WINDOW + STAGE_FLOATS) + wave * SLAB_U4 + lane;
    f32x4 *zslab = workspace + ((size_t)blockIdx.x * WAVES + wave) * ZSLAB_F4 + lane;

    const int tiles_per_img = (m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK;
    const int total = tiles_per_img * batch;
    // DYNAMIC tile order (round 4).  Every workgroup takes tile blockIdx.x first, then fetches the next index from a counter in
    // the workspace tail (one agent-scope atomic per 128-point tile of ~0.44 ms, issued when the tile is done: nothing of this
    // kernel's own is in flight then, so the hand-counted vmcnt waits of decode_tile are not disturbed).  A static
    // tile += gridDim.x deal makes the launch as slow as its unluckiest workgroup: one whose CU was still busy when the launch
    // started (the per-image probe launches of Implicit.prepare run beside it), or the 98 of 256 that get a ninth tile at
    // vox 64.  Thread 0 fetches, the index reaches the other waves through a per-workgroup slot in global memory (the kernel
    // owns all 160 KiB of LDS) behind one barrier.  D = total - gridDim.x indices are real; each workgroup's terminal fetch
    // is >= D, and the one that draws D + gridDim.x - 1 - the last fetch of the launch - puts the counter back to zero for
    // the next launch.  Which workgroup evaluates a tile does not change its result.
    // The counter is the LIBRARY's: launch_counter_reset() zeroes it on the stream in front of every launch (a
    // caller's un-zeroed workspace, or a faulted launch that left it non-zero, must not decide which tiles get evaluated),
    // and a fetched index is used only when it is a real one (unsigned compare).
    int *tile_counter = reinterpret_cast<int *>(workspace + (size_t)MAX_WGS * WAVES * ZSLAB_F4) + 256;   // 1 KiB into the tail
    int *tile_slot = tile_counter + 16 + blockIdx.x;
    const int dyn_tiles = total > (int)gridDim.x ? total - (int)gridDim.x : 0;
    for (int tile = blockIdx.x;;) {
      if (tile < total) {
        const int img = tile / tiles_per_img;
        const int t = tile - img * tiles_per_img;
        const char *prog = programs + (size_t)img * program_stride_bytes;

        const int p = t * PTS_PER_BLOCK + wave * PTS_PER_WAVE + (lane & 31);
        const int pc = p < m ? p : m - 1;  // clamp: tail lanes recompute the last point
        float px, py, pz;
        if (GRID) {
            const long long gp = first_point + pc;
            const long long gg = (long long)G * G;
            const int ix = (int)(gp / gg);
            const int rem = (int)(gp - (long long)ix * gg);
            const int iy = rem / G;
            const int iz = rem - iy * G;
            px = axis[ix];
            py = axis[iy];
            pz = axis[iz];
        } else {
            const float *q = points + ((size_t)img * m + pc) * 3;
            px = q[0];
            py = q[1];
            pz = q[2];
        }
#ifdef ZS_EXP_TIMING
        unsigned long long *dbg = (blockIdx.x == 0 && threadIdx.x == 0 && tile == 0)
            ? reinterpret_cast<unsigned long long *>(workspace + (size_t)MAX_WGS * WAVES * ZSLAB_F4) : nullptr;
#else
        unsigned long long *dbg = nullptr;
#endif
        float guard = 0.f;
        float logit = decode_tile(prog, prm, stage, stage_addr, fl, zslab, px, py, pz, wave, lane, guard, dbg);
        if (apply_sigmoid) logit = 1.0f / (1.0f + expf(-logit));
        // ENVELOPE (zeroshape_amd/program.py, S_GUARD): outside it the tile is flagged for the
        // exact-fp32 kernel, and a program with non-finite / out-of-range operands yields NaN like
        // the reference's arithmetic would (integer tests: this file is built with -fno-honor-nans)
        const unsigned bad = reinterpret_cast<const unsigned *>(prog)[REC_FLOATS + P_FLAG];
        const bool pt_bad = (__builtin_bit_cast(unsigned, px) & 0x7f800000u) == 0x7f800000u ||
                            (__builtin_bit_cast(unsigned, py) & 0x7f800000u) == 0x7f800000u ||
                            (__builtin_bit_cast(unsigned, pz) & 0x7f800000u) == 0x7f800000u;
        if (bad || pt_bad) logit = __builtin_bit_cast(float, 0x7fc00000u);
        if (tile_flags) {
            const bool over = !(__builtin_bit_cast(unsigned, guard) <= __builtin_bit_cast(unsigned, S_GUARD * S_GUARD));
            if (__builtin_amdgcn_ballot_w64(over || bad != 0) != 0 && lane == 0) tile_flags[tile] = 1;
        }
        if (lane < 32 && p < m) out[(size_t)img * m + p] = logit;
      }
        if (static_order) {                   // wave-uniform
            tile += (int)gridDim.x;
            if (tile >= total) break;
            continue;
        }
        if (threadIdx.x == 0) {
            const int d = __hip_atomic_fetch_add(tile_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == dyn_tiles + (int)gridDim.x - 1)
                __hip_atomic_store(tile_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tile_slot, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();                      // (also drains thread 0's store: s_waitcnt vmcnt(0) in front of the barrier)
        const int d = __builtin_amdgcn_readfirstlane(__hip_atomic_load(tile_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if ((unsigned)d >= (unsigned)dyn_tiles) break;
        tile = (int)gridDim.x + d;
    }
}

// not finite, or beyond the fp16 range the split operands saturate at (integer test: NaN-proof)
DEV bool out_of_range(const f32x4 &v) {
    const unsigned lim = 0x477fe000u;  // 65504.0f
    return (__builtin_bit_cast(unsigned, v.x) & 0x7fffffffu) > lim || (__builtin_bit_cast(unsigned, v.y) & 0x7fffffffu) > lim ||
           (__builtin_bit_cast(unsigned, v.z) & 0x7fffffffu) > lim || (__builtin_bit_cast(unsigned, v.w) & 0x7fffffffu) > lim;
}

// Order of the split stream: the fp32 program's, except (a) inside the two MLP sections, where the
// kernel runs a software pipeline over the 32 hidden tiles (fc1 of tile t+1 before fc2 of tile
// t): fc1(0), [fc1(1), fc2(0)], ..., [fc1(31), fc2(30)], fc2(31), 16 K-blocks each; (b) in
// impl_mlp, where the feat halves of the skip layers are contracted inside their layers.
__host__ __device__ inline int split_source_kblock(int kb) {
    constexpr int KB_BLOCK = G_BLOCK / 2, KB_ATT = HEADS * G_HEAD / 2;
    static_assert(KB_BLOCK - KB_ATT == HT * 32, "an MLP section is 32 K-blocks per hidden tile");
    constexpr int KB_IMPL0 = BLOCKS * KB_BLOCK, KB_L = NT * NT * 2;  // impl_mlp section; K-blocks of a 256 x 256 layer
    if (kb >= KB_IMPL0 + G_IMPL / 2) return kb;  // zero tail
    if (kb >= KB_IMPL0) {
        // fp32 program: L0 | Z2 Z4 Z6 (feat halves of the skip layers) | L1 | L2x L3 | L4x L5 | L6x L7
        // split stream: L0 | L1 | for each pair: for each output tile [L(2+2i)x tile | Z(2+2i) tile], L(3+2i)
        const int p = kb - KB_IMPL0;
        if (p < KB_L) return kb;                                  // layer 0
        if (p < 2 * KB_L) return KB_IMPL0 + 4 * KB_L + (p - KB_L);  // layer 1
        const int q = p - 2 * KB_L, pair = q / (3 * KB_L), r = q - pair * 3 * KB_L;
        if (r >= 2 * KB_L) return KB_IMPL0 + (5 + 2 * pair + 1) * KB_L + (r - 2 * KB_L);  // plain layer 3 + 2 pair
        const int nt = r / (4 * NT), w = r - nt * 4 * NT;         // 16 x-part + 16 feat-part K-blocks per tile
        return w < 2 * NT ? KB_IMPL0 + (5 + 2 * pair) * KB_L + nt * 2 * NT + w
                          : KB_IMPL0 + (1 + pair) * KB_L + nt * 2 * NT + (w - 2 * NT);
    }
    const int blk = kb / KB_BLOCK, p = kb - blk * KB_BLOCK - KB_ATT;
    if (p < 16) return kb;  // attention section, or fc1(0)
    const int q = p - 16, it = q >> 5, r = q & 31;
    int src;
    if (it < HT - 1)
        src = r < 16 ? (it + 1) * 32 + r : it * 32 + r;  // fc1(it + 1) | fc2(it)
    else
        src = it * 32 + 16 + r;                          // fc2(31)
    return blk * KB_BLOCK + KB_ATT + src;
}

// fp32 decoder program -> split program: same units and size; K-block j of a unit holds
// records 8 j .. 8 j + 7 of that unit as [hi: lane x 8 fp16][lo: lane x 8 fp16]; params copied.
__global__ __launch_bounds__(256) void split_program_kernel(const float *__restrict__ src,
                                                            size_t src_stride_floats,
                                                            u32x4 *__restrict__ dst,
                                                            size_t dst_stride_u4) {
    const int img = blockIdx.y;
    const float *s = src + (size_t)img * src_stride_floats;
    u32x4 *d = dst + (size_t)img * dst_stride_u4;
    const int e = blockIdx.x * 256 + threadIdx.x;
    constexpr int REC_KB = REC_FLOATS / (KB_U4 * 4);  // K-blocks incl. the zero tail
    if (e < REC_KB * 64) {
        const int kb = e >> 6, lane = e & 63;
        const f32x4 *g = reinterpret_cast<const f32x4 *>(s) + (size_t)split_source_kblock(kb) * 128;  // two fp32 groups
        const f32x4 a = g[lane], b = g[64 + lane];
        if (out_of_range(a) || out_of_range(b)) atomicOr(reinterpret_cast<unsigned *>(d) + REC_FLOATS + P_FLAG, 1u);
        unsigned h[4], l[4];
        split2(a.x, a.y, h[0], l[0]);
        split2(a.z, a.w, h[1], l[1]);
        split2(b.x, b.y, h[2], l[2]);
        split2(b.z, b.w, h[3], l[3]);
        const u32x4 hi = {h[0], h[1], h[2], h[3]}, lo = {l[0], l[1], l[2], l[3]};
        d[(size_t)kb * KB_U4 + lane] = hi;
        d[(size_t)kb * KB_U4 + 64 + lane] = lo;
    } else {
        const int i = e - REC_KB * 64;
        // (the quads of P_FLAG - zeroed by the launcher - and P_KMAX are written elsewhere)
        if (i < PARAM_FLOATS / 4 && i != P_FLAG / 4 && (i < P_KMAX / 4 || i >= (P_KMAX + BLOCKS * HEADS) / 4)) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(s + REC_FLOATS)[i];
            if (i < (P_USED + 3) / 4 && out_of_range(v))
                atomicOr(reinterpret_cast<unsigned *>(d) + REC_FLOATS + P_FLAG, 1u);
            d[REC_FLOATS / 4 + i] = __builtin_bit_cast(u32x4, v);
        }
    }
}

// largest |k_l| over the latent rows of (image, block, head), from the K records of the fp32
// program (record (lt, r) of lane l holds K[32 lt + (l & 31)][row(r, l >> 5)]) -> params[P_KMAX]
__global__ __launch_bounds__(256) void k_bound_kernel(const float *__restrict__ src, size_t src_stride_floats,
                                                      u32x4 *__restrict__ dst, size_t dst_stride_u4) {
    __shared__ float red[256];
    const int img = blockIdx.y, bh = blockIdx.x, blk = bh / HEADS, hd = bh - blk * HEADS;
    const float *kv = src + (size_t)img * src_stride_floats +
                      (size_t)(blk * G_BLOCK + hd * G_HEAD + G_QKV_HEAD) * GROUP_FLOATS;
    float sq = 0.f;
    const int t = threadIdx.x;
    if (t < LT * 32) {
        const int lt = t >> 5;
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 v = reinterpret_cast<const f32x4 *>(kv + (lt * 8 + g) * GROUP_FLOATS)[(t & 31) + 32 * half];
                sq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, sq))));
            }
    }
    red[t] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] = fmaxf(red[t], red[t + o]);
        __syncthreads();
    }
    if (t == 0)
        reinterpret_cast<float *>(dst + (size_t)img * dst_stride_u4)[REC_FLOATS + P_KMAX + bh] = sqrtf(red[0]);
}

int decode_grid_size(int batch, int m) {
    const long long tiles = (long long)batch * ((m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK);
    return (int)(tiles < MAX_WGS ? tiles : MAX_WGS);
}

bool check_programs(const char *what, const void *programs, size_t stride) {
    if (!programs) {
        zs::set_err("%s: null pointer", what);
        return false;
    }
    if (stride % 16 != 0 || stride < zs_sdf_program_bytes()) {
        zs::set_err("%s: bad program stride %zu", what, stride);
        return false;
    }
    return true;
}

}  // namespace

extern "C" int zs_sdf_split_programs(const void *programs, size_t program_stride_bytes,
                                     void *split_programs, size_t split_stride_bytes, int batch,
                                     void *stream) {
    if (batch < 0 || batch > 65535) {
        zs::set_err("zs_sdf_split_programs: bad batch %d", batch);
        return 0;
    }
    if (batch == 0) return 1;
    if (!check_programs("zs_sdf_split_programs", programs, program_stride_bytes) ||
        !check_programs("zs_sdf_split_programs", split_programs, split_stride_bytes))
        return 0;
    const int elems = (REC_FLOATS / (KB_U4 * 4)) * 64 + PARAM_FLOATS / 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemset2DAsync(static_cast<char *>(split_programs) + (size_t)(REC_FLOATS + P_FLAG) * 4, split_stride_bytes,
                         0, 16, batch, st) != hipSuccess) {
        zs::set_err("zs_sdf_split_programs: hipMemset2DAsync failed");
        return 0;
    }
    hipLaunchKernelGGL(split_program_kernel, dim3((elems + 255) / 256, batch), dim3(256), 0, st,
                       static_cast<const float *>(programs), program_stride_bytes / sizeof(float),
                       static_cast<u32x4 *>(split_programs), split_stride_bytes / sizeof(u32x4));
    hipLaunchKernelGGL(k_bound_kernel, dim3(BLOCKS * HEADS, batch), dim3(256), 0, st,
                       static_cast<const float *>(programs), program_stride_bytes / sizeof(float),
                       static_cast<u32x4 *>(split_programs), split_stride_bytes / sizeof(u32x4));
    return zs::check_launch("zs_sdf_split_programs") ? 1 : 0;
}

// in front of every split launch: the dynamic tile order's counter (1 KiB into the workspace tail) starts at zero whatever
// the caller's workspace held (a memset node when the stream is being captured); -> static_order of the launch
static int launch_counter_reset(void *workspace, hipStream_t st) {
    const char *e = getenv("ZS_SPLIT_STATIC_TILES");                  // read per launch: tools/ab_tile_order.py alternates the arms in one process
    if (e && atoi(e) != 0) return 1;
    int *tile_counter = reinterpret_cast<int *>(static_cast<f32x4 *>(workspace) + (size_t)MAX_WGS * WAVES * ZSLAB_F4) + 256;
    (void)hipMemsetAsync(tile_counter, 0, sizeof(int), st);
    return 0;
}

extern "C" int zs_sdf_query_points_split(const void *split_programs, size_t program_stride_bytes,
                                         int batch, const float *points, int m, float *logits,
                                         int *tile_flags, void *workspace, void *stream) {
    if (batch < 0 || m < 0) {
        zs::set_err("zs_sdf_query_points_split: negative size (batch=%d m=%d)", batch, m);
        return 0;
    }
    if (batch == 0 || m == 0) return 1;
    if (!points || !logits || !workspace) {
        zs::set_err("zs_sdf_query_points_split: null pointer");
        return 0;
    }
    if (!check_programs("zs_sdf_query_points_split", split_programs, program_stride_bytes)) return 0;
    if ((long long)batch * ((m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK) > 0x7fffffffLL) {
        zs::set_err("zs_sdf_query_points_split: too many tiles");
        return 0;
    }
    const int static_order = launch_counter_reset(workspace, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL((sdf_decode_split_kernel<false>), dim3(decode_grid_size(batch, m)),
                       dim3(WAVES * 64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char *>(split_programs), program_stride_bytes, batch, points,
                       nullptr, 0, 0LL, m, logits, 0, static_cast<f32x4 *>(workspace), tile_flags, static_order);
    return zs::check_launch("zs_sdf_query_points_split") ? 1 : 0;
}

extern "C" int zs_sdf_query_grid_range_split(const void *split_programs, size_t program_stride_bytes,
                                             int batch, const float *axis, int G, long long point_begin,
                                             long long point_end, int apply_sigmoid, float *out,
                                             int *tile_flags, void *workspace, void *stream) {
    const long long P = (long long)G * G * G;
    if (batch < 0 || G <= 0 || point_begin < 0 || point_end > P || point_begin > point_end) {
        zs::set_err("zs_sdf_query_grid_range_split: bad range (batch=%d G=%d points=[%lld,%lld))", batch, G,
                    point_begin, point_end);
        return 0;
    }
    const long long mm = point_end - point_begin;
    if (batch == 0 || mm == 0) return 1;
    if (!axis || !out || !workspace) {
        zs::set_err("zs_sdf_query_grid_range_split: null pointer");
        return 0;
    }
    if (mm > 0x7fffffffLL - PTS_PER_BLOCK) {
        zs::set_err("zs_sdf_query_grid_range_split: %lld points per launch exceed 2^31; split the range", mm);
        return 0;
    }
    if (!check_programs("zs_sdf_query_grid_range_split", split_programs, program_stride_bytes)) return 0;
    const int m = (int)mm;
    const int static_order = launch_counter_reset(workspace, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL((sdf_decode_split_kernel<true>), dim3(decode_grid_size(batch, m)),
                       dim3(WAVES * 64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char *>(split_programs), program_stride_bytes, batch, nullptr,
                       axis, G, point_begin, m, out, apply_sigmoid, static_cast<f32x4 *>(workspace), tile_flags, static_order);
    return zs::check_launch("zs_sdf_query_grid_range_split") ? 1 : 0;
}

extern "C" int zs_sdf_query_grid_split(const void *split_programs, size_t program_stride_bytes,
                                       int batch, const float *axis, int G, int slice_begin,
                                       int slice_end, int apply_sigmoid, float *out, int *tile_flags,
                                       void *workspace, void *stream) {
    if (batch < 0 || G <= 0 || slice_begin < 0 || slice_end > G || slice_begin > slice_end) {
        zs::set_err("zs_sdf_query_grid_split: bad range (batch=%d G=%d slices=[%d,%d))", batch, G,
                    slice_begin, slice_end);
        return 0;
    }
    const long long mm = (long long)(slice_end - slice_begin) * G * G;
    if (batch == 0 || mm == 0) return 1;
    if (!axis || !out || !workspace) {
        zs::set_err("zs_sdf_query_grid_split: null pointer");
        return 0;
    }
    if (mm > 0x7fffffffLL - PTS_PER_BLOCK) {
        zs::set_err("zs_sdf_query_grid_split: %lld points per launch exceed 2^31; split the slab", mm);
        return 0;
    }
    if (!check_programs("zs_sdf_query_grid_split", split_programs, program_stride_bytes)) return 0;
    const int m = (int)mm;
    const int static_order = launch_counter_reset(workspace, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL((sdf_decode_split_kernel<true>), dim3(decode_grid_size(batch, m)),
                       dim3(WAVES * 64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char *>(split_programs), program_stride_bytes, batch, nullptr,
                       axis, G, (long long)slice_begin * G * G, m, out, apply_sigmoid,
                       static_cast<f32x4 *>(workspace), tile_flags, static_order);
    return zs::check_launch("zs_sdf_query_grid_split") ? 1 : 0;
}
