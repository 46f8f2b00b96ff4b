// Fused implicit-occupancy decoder for MI355X (gfx950): one launch evaluates
// Implicit.forward (model/shape/implicit.py:251-288) for every query point, with the
// point-independent latent half hoisted into the prologue (csrc/sdf_prologue.hip).
//
// Mapping to the hardware (zeroshape_amd/program.py):
//  * persistent workgroups (one per CU, 4 waves = one per SIMD); a wave owns 32 query
//    points for the whole network;
//  * every layer is computed transposed (Y^T = W X^T) with v_mfma_f32_32x32x2_f32, so the
//    accumulator layout of one layer IS the B-operand layout of the next: no transposes;
//  * register budget (the 512-entry file could hold everything, but hipcc spilled 3.7 GB
//    per launch to scratch in the first version, profiles/r01_v0_*): accumulators and ONE
//    activation array live in registers, the other activation array of a layer pair lives
//    in the wave's private 32 KiB LDS slab (read back as B operands with conflict-free
//    ds_read_b128), and the feat halves of the three skip layers are computed right after
//    the final LayerNorm and parked in a per-wave global workspace (L2-resident);
//  * weights arrive as one linear stream of pre-packed A operands ("records", 4 per
//    16-byte load), prefetched 8 loads (32 MFMAs) ahead through a register ring of
//    inline-asm loads with hand-counted s_waitcnt;
//  * biases / LayerNorm affine / xyz columns sit in LDS (32 KiB, two phases);
//  * exact-fp32 MFMA (bitwise an fmaf chain) - parity mode; 39,424 MFMAs per wave tile
//    = 5.05 MFLOP per point including the 224-vs-197 latent padding.
//
// Per-point arithmetic follows the reference op for op (LayerNorm eps 1e-6, softmax
// over 197 latent logits + 1 self logit, exact-erf GELU, softplus(beta=100,
// threshold=20), cat(..)/sqrt(2) skips); only summation order differs.
#include "zs_common.h"
#include "sdf_layout.h"
#include "sdf_math.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

namespace {

using namespace zs::lay;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 4;
constexpr int PTS_PER_WAVE = 32;
constexpr int PTS_PER_BLOCK = WAVES * PTS_PER_WAVE;
constexpr int MAX_WGS = 256;                       // one persistent workgroup per CU
constexpr int SLAB_F4 = NT * 4 * 64;               // one activation array as float4 groups: 32 KiB
constexpr int ZSLAB_F4 = 3 * SLAB_F4;              // three skip layers' feat partial products
constexpr int ATTN_P_F4 = BLOCKS * HEADS * LT * 4 * 64;      // raw P tiles per wave tile (float4)
constexpr int ATTN_ST_F = BLOCKS * HEADS * 9 * 64;          // 7 reference maxima + final max + 1/Z per lane
constexpr int ATTN_WT_F4 = ATTN_P_F4 + ATTN_ST_F / 4;       // per wave tile
constexpr size_t WORKSPACE_BYTES = (size_t)MAX_WGS * WAVES * ZSLAB_F4 * sizeof(f32x4) + 4096;  // + debug tail

#define DEV __device__ __forceinline__

DEV const f32x4 *uniform_ptr(const f32x4 *p) {  // make wave-uniformity provable ("s" operands)
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const f32x4 *>(((uint64_t)hi << 32) | lo);
}

// ---- weight stream: register ring, 8 x 16-byte loads in flight per lane ------------- //
// Left to itself hipcc sinks each global_load next to its first use (vmcnt(1) pattern, one
// load in flight), which exposed the L2 latency on every group - 32 % of all wave cycles
// parked in s_waitcnt in the first profile (profiles/r01_v0_*).  The ring is therefore
// inline asm with hand-counted waits, and - because hipcc may copy, split or spill any
// asm OUTPUT register while its load is still in flight (it did: v_mov copies of in-flight
// ring registers at loop back-edges and exits) - the ring lives in 32 fixed registers the
// compiler cannot see: amdgpu_num_vgpr(240) caps BOTH halves of the unified file at 240
// (on gfx90a+ with AGPR-using asm the budget is split evenly), so v[240:255] and a[240:255]
// are never allocated; the asm statements name them literally and list them as clobbers
// (which also makes the kernel descriptor allocate all 512), and global loads may write
// AGPRs directly.  Landed data is copied out with v_mov / v_accvgpr_read into ordinary
// compiler-owned values (4 VALU per 4 MFMAs, hidden under the MFMAs).
//
// Protocol: ONE queue position per 16-byte load, consumed in issue order.  At position p
// the wave waits vmcnt(RING-1) (the 7 younger loads stay in flight; loads return in
// order, and any extra compiler VMEM op only makes the wait more conservative), copies
// slot p % 8 out and re-issues that slot for position p + 8 in the same asm statement.
// Most positions are weight groups; in the skip layers 4 positions per output tile fetch
// the tile's parked feat partial product from the workspace instead.
// tools/check_asm_ring.py audits the .s: no compiler instruction may mention v240..v255
// or a240..a255.
// slots 0-3 live in v[240:255], slots 4-7 in a[240:255]
#define ZS_TAKE_ISSUE_V(A, B, C, D, SUFFIX)                                                          \
    asm volatile("s_waitcnt vmcnt(7)\n\tv_mov_b32 %0, v" #A "\n\tv_mov_b32 %1, v" #B               \
                 "\n\tv_mov_b32 %2, v" #C "\n\tv_mov_b32 %3, v" #D                                 \
                 "\n\tglobal_load_dwordx4 v[" #A ":" #D "], %4, %5" SUFFIX "\n\ts_nop 1"             \
                 : "=&v"(o.x), "=&v"(o.y), "=&v"(o.z), "=&v"(o.w)                                    \
                 : "v"(voff), "s"(src)                                                               \
                 : "v" #A, "v" #B, "v" #C, "v" #D)
#define ZS_TAKE_ISSUE_A(A, B, C, D, SUFFIX)                                                          \
    asm volatile("s_waitcnt vmcnt(7)\n\tv_accvgpr_read_b32 %0, a" #A "\n\tv_accvgpr_read_b32 %1, a" #B \
                 "\n\tv_accvgpr_read_b32 %2, a" #C "\n\tv_accvgpr_read_b32 %3, a" #D                 \
                 "\n\tglobal_load_dwordx4 a[" #A ":" #D "], %4, %5" SUFFIX "\n\ts_nop 1"             \
                 : "=&v"(o.x), "=&v"(o.y), "=&v"(o.z), "=&v"(o.w)                                    \
                 : "v"(voff), "s"(src)                                                               \
                 : "a" #A, "a" #B, "a" #C, "a" #D)
#define ZS_ISSUE_V(A, B, C, D)                                                                       \
    asm volatile("global_load_dwordx4 v[" #A ":" #D "], %0, %1" : : "v"(voff), "s"(src)              \
                 : "v" #A, "v" #B, "v" #C, "v" #D)
#define ZS_ISSUE_A(A, B, C, D)                                                                       \
    asm volatile("global_load_dwordx4 a[" #A ":" #D "], %0, %1" : : "v"(voff), "s"(src)              \
                 : "a" #A, "a" #B, "a" #C, "a" #D)

// A whole weight position in one statement: wait for the slot, run the group's 4 MFMAs with
// the A operands read STRAIGHT from the ring registers (no copy-out: on this chip every
// extra VALU instruction costs MFMA time, see the note at gelu_erf), then re-issue the slot.
// The accumulator is a tied AGPR tuple ("+a"); B operands are ordinary VGPR values.
// Hazards hipcc does not pad for asm (cdna_hip_programming.md section 5.7 item 2):
//  * the leading s_nop 1 covers compiler VALU / v_accvgpr_write results feeding this MFMA;
//  * 4 MFMAs accumulating into the same tuple back to back need no padding;
//  * the reload is issued after the MFMAs, which read their operands long before a load
//    can return;
//  * MFMA result -> any non-accumulating reader: mfma_done() below, after the last group.
#define ZS_MFMA4(ACC, RF, A, B, C, D, SUFFIX)                                                        \
    asm volatile("s_waitcnt vmcnt(7)\n\ts_nop 1"                                                     \
                 "\n\tv_mfma_f32_32x32x2_f32 %0, " RF #A ", %1, %0"                                  \
                 "\n\tv_mfma_f32_32x32x2_f32 %0, " RF #B ", %2, %0"                                  \
                 "\n\tv_mfma_f32_32x32x2_f32 %0, " RF #C ", %3, %0"                                  \
                 "\n\tv_mfma_f32_32x32x2_f32 %0, " RF #D ", %4, %0"                                  \
                 "\n\tglobal_load_dwordx4 " RF "[" #A ":" #D "], %5, %6" SUFFIX                      \
                 : ACC(acc)                                                                          \
                 : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(voff), "s"(src)                           \
                 : RF #A, RF #B, RF #C, RF #D)

struct Stream {
    const f32x4 *abase;  // wave-uniform (SGPR pair): next weight group to fetch
    unsigned voff;       // lane * 16 bytes

    // wait for slot, copy it out, re-issue it from `src` (plain: weights; sc1: workspace)
    DEV f32x4 take_issue(int slot, const f32x4 *src, bool sc1) {
        f32x4 o;
        if (!sc1) {
            switch (slot) {
                case 0: ZS_TAKE_ISSUE_V(240, 241, 242, 243, ""); break;
                case 1: ZS_TAKE_ISSUE_V(244, 245, 246, 247, ""); break;
                case 2: ZS_TAKE_ISSUE_V(248, 249, 250, 251, ""); break;
                case 3: ZS_TAKE_ISSUE_V(252, 253, 254, 255, ""); break;
                case 4: ZS_TAKE_ISSUE_A(240, 241, 242, 243, ""); break;
                case 5: ZS_TAKE_ISSUE_A(244, 245, 246, 247, ""); break;
                case 6: ZS_TAKE_ISSUE_A(248, 249, 250, 251, ""); break;
                default: ZS_TAKE_ISSUE_A(252, 253, 254, 255, ""); break;
            }
        } else {  // served by L2: this wave wrote the workspace earlier in the launch
            switch (slot) {
                case 0: ZS_TAKE_ISSUE_V(240, 241, 242, 243, " sc1"); break;
                case 1: ZS_TAKE_ISSUE_V(244, 245, 246, 247, " sc1"); break;
                case 2: ZS_TAKE_ISSUE_V(248, 249, 250, 251, " sc1"); break;
                case 3: ZS_TAKE_ISSUE_V(252, 253, 254, 255, " sc1"); break;
                case 4: ZS_TAKE_ISSUE_A(240, 241, 242, 243, " sc1"); break;
                case 5: ZS_TAKE_ISSUE_A(244, 245, 246, 247, " sc1"); break;
                case 6: ZS_TAKE_ISSUE_A(248, 249, 250, 251, " sc1"); break;
                default: ZS_TAKE_ISSUE_A(252, 253, 254, 255, " sc1"); break;
            }
        }
        return o;
    }
    // weight position: 4 MFMAs + re-issue of the slot from `src`.  AV = false: the accumulator
    // is a long-lived AGPR tuple (the residual stream y); AV = true: a short-lived VGPR tuple
    // whose result is post-processed by VALU code right away (q/k/v, S, o, hidden, impl
    // layers) - saves the v_accvgpr_read/write round trips, which are full-price VALU ops.
    template <bool VACC>
    DEV void mfma4_from(int slot, f32x16 &acc, float b0, float b1, float b2, float b3,
                        const f32x4 *src, bool sc1) {
#define ZS_ACC_A(x) "+a"(x)
#define ZS_ACC_V(x) "+v"(x)
#define ZS_SLOTS(ACC, SUFFIX)                                              \
        switch (slot) {                                                    \
            case 0: ZS_MFMA4(ACC, "v", 240, 241, 242, 243, SUFFIX); break; \
            case 1: ZS_MFMA4(ACC, "v", 244, 245, 246, 247, SUFFIX); break; \
            case 2: ZS_MFMA4(ACC, "v", 248, 249, 250, 251, SUFFIX); break; \
            case 3: ZS_MFMA4(ACC, "v", 252, 253, 254, 255, SUFFIX); break; \
            case 4: ZS_MFMA4(ACC, "a", 240, 241, 242, 243, SUFFIX); break; \
            case 5: ZS_MFMA4(ACC, "a", 244, 245, 246, 247, SUFFIX); break; \
            case 6: ZS_MFMA4(ACC, "a", 248, 249, 250, 251, SUFFIX); break; \
            default: ZS_MFMA4(ACC, "a", 252, 253, 254, 255, SUFFIX); break; \
        }
        if (VACC) {
            if (!sc1) { ZS_SLOTS(ZS_ACC_V, "") } else { ZS_SLOTS(ZS_ACC_V, " sc1") }
        } else {
            if (!sc1) { ZS_SLOTS(ZS_ACC_A, "") } else { ZS_SLOTS(ZS_ACC_A, " sc1") }
        }
#undef ZS_SLOTS
    }
    template <bool VACC>
    DEV void mfma4(int slot, f32x16 &acc, float b0, float b1, float b2, float b3) {
        mfma4_from<VACC>(slot, acc, b0, b1, b2, b3, abase, false);
        abase += 64;
    }
    DEV f32x4 next(int slot) {  // weight position whose slot is re-used by a weight position
        const f32x4 a = take_issue(slot, abase, false);
        abase += 64;
        return a;
    }
    DEV f32x4 next_then_z(int slot, const f32x4 *zsrc) {  // ... re-used by a workspace position
        return take_issue(slot, zsrc, true);
    }
    DEV void init(const f32x4 *b, int ln) {
        voff = ln * 16;
        const f32x4 *src = b;
        ZS_ISSUE_V(240, 241, 242, 243); src += 64;
        ZS_ISSUE_V(244, 245, 246, 247); src += 64;
        ZS_ISSUE_V(248, 249, 250, 251); src += 64;
        ZS_ISSUE_V(252, 253, 254, 255); src += 64;
        ZS_ISSUE_A(240, 241, 242, 243); src += 64;
        ZS_ISSUE_A(244, 245, 246, 247); src += 64;
        ZS_ISSUE_A(248, 249, 250, 251); src += 64;
        ZS_ISSUE_A(252, 253, 254, 255); src += 64;
        abase = src;
    }
    // retire the 8 loads still in flight past the end of the stream (padding groups)
    DEV void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// MFMA results leave asm-land: 16-pass MFMA D -> any reader other than an accumulating MFMA
// needs ~19 wait states that hipcc does not know about.
template <bool VACC>
DEV void mfma_done(f32x16 &acc) {
    if (VACC)
        asm volatile("s_nop 15\n\ts_nop 5" : "+v"(acc));
    else
        asm volatile("s_nop 15\n\ts_nop 5" : "+a"(acc));
}

// acc += W_tile * X.  X = KT activation tiles as 16*KT scalars in registers (register r of
// tile kt is X[16*kt + r]); consumes KT*4 groups starting at ring slot `phase` (kt-major,
// then register) - the order program.py packs them in.  `phase` (0 or 4) and every index
// are compile-time constants after unrolling.  Register-resident activations are plain
// scalars, not f32x16 tuples: only accumulators need 16-register tuples.
template <int KT, bool VACC>
DEV void gemm_tile(Stream &s, const float *X, f32x16 &acc, int phase) {
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
#pragma unroll
        for (int g = 0; g < 4; g++)
            s.mfma4<VACC>((phase + kt * 4 + g) & (RING - 1), acc, X[kt * 16 + 4 * g + 0],
                          X[kt * 16 + 4 * g + 1], X[kt * 16 + 4 * g + 2], X[kt * 16 + 4 * g + 3]);
    }
    mfma_done<VACC>(acc);
}

// one-tile variant whose B operand is a 16-register activation tile (q, P, o, hidden);
// `last` = the caller reads acc next (several of these can chain on one accumulator)
template <bool VACC, typename T>
DEV void gemm_tile_v(Stream &s, const T &X, f32x16 &acc, int phase, bool last = true) {
#pragma unroll
    for (int g = 0; g < 4; g++)
        s.mfma4<VACC>((phase + g) & (RING - 1), acc, X[4 * g + 0], X[4 * g + 1], X[4 * g + 2], X[4 * g + 3]);
    if (last) mfma_done<VACC>(acc);
}

// 8-tile variant with the B operands read from the wave's LDS slab
// ([kt][g][lane] float4: lane-contiguous -> conflict-free ds_read_b128).  The read for
// group g+1 is issued BEFORE the asm statement of group g (hipcc cannot hoist it itself:
// LDS reads do not cross the asm statements), i.e. a whole group of MFMAs ahead of its
// use; otherwise every group paid the LDS latency with the MFMA pipe idle.
template <bool VACC>
DEV void gemm_tile_lds(Stream &s, const f32x4 *fl, f32x16 &acc, int phase) {
    f32x4 b = fl[0];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            f32x4 bn = b;
            if (kt * 4 + g + 1 < NT * 4) bn = fl[(kt * 4 + g + 1) * 64];
            s.mfma4<VACC>((phase + kt * 4 + g) & (RING - 1), acc, b.x, b.y, b.z, b.w);
            b = bn;
        }
    }
    mfma_done<VACC>(acc);
}

// one output tile of a skip layer: 32 weight positions (B = x / sqrt(2) in registers), then
// 4 workspace positions that add the tile's parked feat partial product.  36 positions:
// `phase` alternates 0 / 4 from tile to tile.  Positions 24..27 re-issue their slots for
// positions 32..35 = the workspace tile.
DEV void skip_tile(Stream &s, const float *X, f32x16 &acc, const f32x4 *ztile, int phase) {
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const int slot = (phase + q) & (RING - 1);
        if (q >= 24 && q < 28)
            s.mfma4_from<true>(slot, acc, X[4 * q + 0], X[4 * q + 1], X[4 * q + 2], X[4 * q + 3],
                               ztile + (q - 24) * 64, true);
        else
            s.mfma4<true>(slot, acc, X[4 * q + 0], X[4 * q + 1], X[4 * q + 2], X[4 * q + 3]);
    }
    mfma_done<true>(acc);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const f32x4 a = s.next((phase + 32 + j) & (RING - 1));
        acc[4 * j + 0] += a.x;
        acc[4 * j + 1] += a.y;
        acc[4 * j + 2] += a.z;
        acc[4 * j + 3] += a.w;
    }
}

DEV float xhalf(float v) { return __shfl_xor(v, 32, 64); }  // value of lane l ^ 32

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

// LayerNorm statistics over the 256 features of each point: 8 accumulator tuples
// (16 regs each) x 2 lane halves
DEV void ln_stats(const f32x16 *x, float &mean, float &rstd) {
    float s = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += x[kt][r];
    s += xhalf(s);
    mean = s * (1.0f / 256.0f);
    float v = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float d = x[kt][r] - mean;
            v = fmaf(d, d, v);
        }
    v += xhalf(v);
    rstd = 1.0f / sqrtf(v * (1.0f / 256.0f) + 1e-6f);
}
// LayerNorm -> the wave's LDS slab (B operands of the following GEMMs)
DEV void layer_norm_lds(const f32x16 *x, f32x4 *fl, const float *prm, int g_off, int b_off, int hi) {
    float mean, rstd;
    ln_stats(x, mean, rstd);
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        float g[16], b[16];
        rp(prm, g_off, kt, hi, g);
        rp(prm, b_off, kt, hi, b);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x4 t;
            t.x = fmaf((x[kt][4 * j + 0] - mean) * rstd, g[4 * j + 0], b[4 * j + 0]);
            t.y = fmaf((x[kt][4 * j + 1] - mean) * rstd, g[4 * j + 1], b[4 * j + 1]);
            t.z = fmaf((x[kt][4 * j + 2] - mean) * rstd, g[4 * j + 2], b[4 * j + 2]);
            t.w = fmaf((x[kt][4 * j + 3] - mean) * rstd, g[4 * j + 3], b[4 * j + 3]);
            fl[(kt * 4 + j) * 64] = t;
        }
    }
}
// LayerNorm -> registers
DEV void layer_norm_reg(const f32x16 *x, float *h, const float *prm, int g_off, int b_off, int hi) {
    float mean, rstd;
    ln_stats(x, mean, rstd);
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        float g[16], b[16];
        rp(prm, g_off, kt, hi, g);
        rp(prm, b_off, kt, hi, b);
#pragma unroll
        for (int r = 0; r < 16; r++) h[kt * 16 + r] = fmaf((x[kt][r] - mean) * rstd, g[r], b[r]);
    }
}

// NOTE on VALU cost: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate and does not
// overlap with VALU instructions of the same SIMD (tools/ubench/mfma_chain.hip: every VALU
// op beside a dependent MFMA chain adds its full 4 cycles), so each VALU instruction in
// this kernel costs MFMA time.  The activation functions are therefore written for the
// fewest instructions that still sit 2+ orders of magnitude inside the 1e-4 contract;
// tests/test_device_math.py checks the same formulas against fp64 on the host.

// gelu_erf / softplus100: csrc/sdf_math.h (shared with the split-fp16 decoder)
using zs::dm::gelu_erf;
using zs::dm::softplus100;

DEV void store_tile_lds(f32x4 *fl, int tile, const float *v) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f32x4 t;
        t.x = v[4 * j + 0]; t.y = v[4 * j + 1]; t.z = v[4 * j + 2]; t.w = v[4 * j + 3];
        fl[(tile * 4 + j) * 64] = t;
    }
}

// One wave: 32 points (lane & 31; both lane halves carry the same point).
// `prm`: LDS params region (phase A: program params [0, P_PHASE_B); phase B: the rest);
// `fl`: this wave's LDS slab, `zs`: this wave's workspace slab - both already offset by lane
// except `zs_u`, the same workspace slab as a wave-uniform pointer for the asm loads.
// One latent tile of the point->latent attention of one head: S = K_tile q (16 MFMAs),
// online softmax update in the log2 domain, o += V_tile^T P (16 MFMAs).  MASK: the tile is
// the last one and its rows >= 197 are padding.
// ATTN: also dump the un-normalised probabilities and their reference maximum for the
// attention-visualisation output (normalised and averaged by attn_reduce_kernel).
template <bool MASK, bool ATTN>
DEV void attn_tile(Stream &s, const f32x16 &q, f32x16 &o, float &m_run, float &z_run, float c, int hi,
                   f32x4 *praw, float *pstat) {
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; r++) S[r] = 0.f;
    gemm_tile_v<true>(s, q, S, 0);
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (MASK) {
            const int rw = (r & 3) + 8 * (r >> 2) + 4 * hi;
            S[r] = rw < L - 32 * (LT - 1) ? S[r] : -INFINITY;
        }
        mt = fmaxf(mt, S[r]);
    }
    mt = fmaxf(mt, xhalf(mt)) * c;
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float zs_ = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(S[r], c, -m_new));
        S[r] = p;
        zs_ += p;
    }
    z_run = fmaf(z_run, alpha, zs_);
    if (ATTN) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x4 t;
            t.x = S[4 * j + 0]; t.y = S[4 * j + 1]; t.z = S[4 * j + 2]; t.w = S[4 * j + 3];
            praw[j * 64] = t;
        }
        *pstat = m_new;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] *= alpha;
    gemm_tile_v<true>(s, S, o, 4);
    m_run = m_new;
}

#ifdef ZS_EXP_TIMING  // tools/phase_timing.py: cycle stamps of (block 0, wave 0, first tile) -> workspace tail
#define ZS_STAMP(i) do { if (dbg) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ZS_STAMP(i) do { } while (0)
#endif
template <bool ATTN>
DEV float decode_tile(const f32x4 *recs, const float *__restrict__ prog_params, float *prm, f32x4 *fl,
                      f32x4 *zs, const f32x4 *zs_u, float px, float py, float pz, int lane,
                      unsigned long long *dbg, f32x4 *araw) {
    const int hi = lane >> 5;
    ZS_STAMP(0);
    Stream s;
    s.init(recs, lane);

    // point_proj (implicit.py:128-131); y is the residual stream, kept as accumulators
    f32x16 y[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) y[kt] = xyz_affine(prm, P_PP, kt, hi, px, py, pz);

    const float scale = 0.17677669529663688110f;  // 32 ** -0.5

#pragma unroll 1
    for (int blk = 0; blk < BLOCKS; blk++) {
        const int pb = P_BLK0 + blk * P_BLK_STRIDE;
        ZS_STAMP(1 + blk * 4);
        layer_norm_lds(y, fl, prm, pb + PB_LN1G, pb + PB_LN1B, hi);
        ZS_STAMP(2 + blk * 4);
        // y = x + proj_bias + sum_heads Wproj_h o_h
#pragma unroll
        for (int nt = 0; nt < NT; nt++) y[nt] += rp16(prm, pb + PB_BPROJ, nt, hi);

#pragma unroll 1
        for (int hd = 0; hd < HEADS; hd++) {
            f32x16 q = rp16(prm, pb + PB_BQKV, hd * 3 + 0, hi);
            gemm_tile_lds<true>(s, fl, q, 0);
            f32x16 k = rp16(prm, pb + PB_BQKV, hd * 3 + 1, hi);
            gemm_tile_lds<true>(s, fl, k, 0);
            f32x16 v = rp16(prm, pb + PB_BQKV, hd * 3 + 2, hi);
            gemm_tile_lds<true>(s, fl, v, 0);

            // logits are kept in the log2 domain: c = d^-1/2 * log2(e), softmax = 2^(c s - m)
            const float c = scale * 1.44269504088896340736f;
            // self logit (implicit.py:44)
            float s_self = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) s_self = fmaf(q[r], k[r], s_self);
            s_self = (s_self + xhalf(s_self)) * c;

            // online softmax over 7 latent tiles (+ self), o = sum P V
            float m_run = -INFINITY, z_run = 0.f;
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; r++) o[r] = 0.f;
            // attention-vis dump slots of this (block, head): raw P tiles + per-lane statistics
            f32x4 *praw = ATTN ? araw + ((blk * HEADS + hd) * LT) * 256 + lane : nullptr;
            float *pstat = ATTN ? reinterpret_cast<float *>(araw + ATTN_P_F4) + (blk * HEADS + hd) * 9 * 64 + lane
                                : nullptr;
#pragma unroll 1
            for (int lt = 0; lt < LT - 1; lt++)
                attn_tile<false, ATTN>(s, q, o, m_run, z_run, c, hi, praw + lt * 256, pstat + lt * 64);
            attn_tile<true, ATTN>(s, q, o, m_run, z_run, c, hi, praw + (LT - 1) * 256,
                                  pstat + (LT - 1) * 64);  // last tile: rows >= 197 masked
            {
                const float m_new = fmaxf(m_run, s_self);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                const float p_self = __builtin_amdgcn_exp2f(s_self - m_new);
                const float z = fmaf(z_run + xhalf(z_run), alpha, p_self);
                const float inv = 1.0f / z;
                if (ATTN) {
                    pstat[7 * 64] = m_new;
                    pstat[8 * 64] = inv;
                }
                const float a_i = alpha * inv, p_i = p_self * inv;
#pragma unroll
                for (int r = 0; r < 16; r++) o[r] = fmaf(p_i, v[r], o[r] * a_i);
            }
            // y += Wproj[:, head] o_h
#pragma unroll
            for (int nt = 0; nt < NT; nt++) gemm_tile_v<false>(s, o, y[nt], (nt & 1) * 4, nt == NT - 1);
        }

        ZS_STAMP(3 + blk * 4);
        // MLP (timm Mlp): y += b2 + W2 gelu(W1 LN2(y) + b1), one hidden tile at a time
        layer_norm_lds(y, fl, prm, pb + PB_LN2G, pb + PB_LN2B, hi);
        ZS_STAMP(4 + blk * 4);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) y[nt] += rp16(prm, pb + PB_B2, nt, hi);
#pragma unroll 1
        for (int ht = 0; ht < HT; ht++) {
            f32x16 hid = rp16(prm, pb + PB_B1, ht, hi);
            gemm_tile_lds<true>(s, fl, hid, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) hid[r] = gelu_erf(hid[r]);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) gemm_tile_v<false>(s, hid, y[nt], (nt & 1) * 4, nt == NT - 1);
        }
    }

    ZS_STAMP(9);
    // final norm (implicit.py:275) -> feat, in registers
    float h[NT * 16];
    layer_norm_reg(y, h, prm, P_LNFG, P_LNFB, hi);

    // phase B params (impl_mlp) replace the phase A ones in LDS
    __syncthreads();
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(prog_params + P_PHASE_B);
        f32x4 *dst = reinterpret_cast<f32x4 *>(prm);
        for (int i = threadIdx.x; i < (P_USED - P_PHASE_B + 3) / 4; i += WAVES * 64) dst[i] = src[i];
    }
    __syncthreads();

    ZS_STAMP(10);
    // impl_mlp (implicit.py:168-184): inputs = cat[xyz, feat].  Layer 0: feat (regs) -> LDS
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        f32x16 acc = xyz_affine(prm, P_IMPL0 - P_PHASE_B, nt, hi, px, py, pz);
        gemm_tile<NT, true>(s, h, acc, 0);
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; r++) t[r] = softplus100(acc[r]);
        store_tile_lds(fl, nt, t);
    }
    ZS_STAMP(11);
    // the skip layers consume cat[x, xyz, feat] / sqrt(2): their feat halves are computed now,
    // while feat is in registers, and parked in the workspace (Z tiles)
    // (x * (1/sqrt 2) instead of the reference's x / sqrt 2: <= 1 ulp apart, 10x fewer VALU ops)
    const float rsqrt2 = 0.70710678118654752440f;
#pragma unroll
    for (int i = 0; i < NT * 16; i++) h[i] = h[i] * rsqrt2;
    const float sx = px * rsqrt2, sy = py * rsqrt2, sz = pz * rsqrt2;
#pragma unroll 1
    for (int li = 0; li < 3; li++) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            gemm_tile<NT, true>(s, h, acc, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4 t;
                t.x = acc[4 * j + 0]; t.y = acc[4 * j + 1]; t.z = acc[4 * j + 2]; t.w = acc[4 * j + 3];
                zs[(li * SLAB_F4) + (nt * 4 + j) * 64] = t;
            }
        }
    }
    // the Z stores must have reached L2 before the sc1 read-backs (>= 256 groups later; this
    // drain is a formality that costs one ring refill per tile)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    ZS_STAMP(12);
    // layer 1 (plain): LDS -> registers, pre-divided by sqrt(2) because layer 2 is a skip layer
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        f32x16 acc = rp16(prm, P_IMPL1 - P_PHASE_B, nt, hi);
        gemm_tile_lds<true>(s, fl, acc, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) h[nt * 16 + r] = softplus100(acc[r]) * rsqrt2;
    }
    ZS_STAMP(13);
#pragma unroll 1
    for (int i = 0; i < 3; i++) {
        const int pp = P_IMPL_PAIR - P_PHASE_B + i * P_IMPL_PAIR_STRIDE;
        const f32x4 *zl = zs_u + i * SLAB_F4;
        // skip layer 2+2i: registers (x / sqrt(2)) + parked feat half -> LDS
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            f32x16 acc = xyz_affine(prm, pp, nt, hi, sx, sy, sz);
            skip_tile(s, h, acc, zl + nt * 256, (nt & 1) * 4);
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = softplus100(acc[r]);
            store_tile_lds(fl, nt, t);
        }
        // plain layer 3+2i: LDS -> registers (/ sqrt(2) when the next layer is a skip layer)
        const float post = i < 2 ? rsqrt2 : 1.0f;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            f32x16 acc = rp16(prm, pp + 1024, nt, hi);
            gemm_tile_lds<true>(s, fl, acc, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) h[nt * 16 + r] = softplus100(acc[r]) * post;
        }
    }
    ZS_STAMP(14);
    s.drain();
    // layer 8: 256 -> 1
    float out = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        float w[16];
        rp(prm, P_W8 - P_PHASE_B, kt, hi, w);
#pragma unroll
        for (int r = 0; r < 16; r++) out = fmaf(h[kt * 16 + r], w[r], out);
    }
    out += xhalf(out);
    ZS_STAMP(15);
    return out + prm[P_B8 - P_PHASE_B];
}

template <bool GRID, bool ATTN>
__global__ __launch_bounds__(WAVES * 64, 1) __attribute__((amdgpu_num_vgpr(240))) void sdf_decode_kernel(
    const float *__restrict__ programs, size_t program_stride_floats, int batch,
    const float *__restrict__ points,  // !GRID: [batch][m][3]
    const float *__restrict__ axis,    //  GRID: [G]
    int G, long long first_point,      //  GRID: linear index of the first grid point
    int m,                             // points per image handled by this launch
    float *__restrict__ out, int apply_sigmoid, f32x4 *__restrict__ workspace,
    f32x4 *__restrict__ attn_raw,    // ATTN: [wave tile][ATTN_WT_F4]
    const int *__restrict__ tile_mask) {  // non-null: only the 128-point tiles with a non-zero entry
    // LDS: [params 32 KiB][4 x 32 KiB activation slabs] = 160 KiB, one workgroup per CU
    __shared__ __attribute__((aligned(16))) float lds[P_PHASE_B + WAVES * SLAB_F4 * 4];
    float *prm = lds;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    f32x4 *fl = reinterpret_cast<f32x4 *>(lds + P_PHASE_B) + wave * SLAB_F4 + lane;
    f32x4 *zslab = workspace + ((size_t)blockIdx.x * WAVES + wave) * ZSLAB_F4;
    const f32x4 *zs_u = uniform_ptr(zslab);

    const int tiles_per_img = (m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK;
    const int total = tiles_per_img * batch;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        if (tile_mask && tile_mask[tile] == 0) continue;  // workgroup-uniform
        const int img = tile / tiles_per_img;
        const int t = tile - img * tiles_per_img;
        const float *prog = programs + (size_t)img * program_stride_floats;
        __syncthreads();  // previous tile done with the phase B params
        {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(prog + REC_FLOATS);
            f32x4 *dst = reinterpret_cast<f32x4 *>(prm);
            for (int i = threadIdx.x; i < P_PHASE_B / 4; i += WAVES * 64) dst[i] = src[i];
        }
        __syncthreads();

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
        const f32x4 *recs = uniform_ptr(reinterpret_cast<const f32x4 *>(prog));
#ifdef ZS_EXP_TIMING
        unsigned long long *dbg = (blockIdx.x == 0 && threadIdx.x == 0 && tile == 0)
            ? reinterpret_cast<unsigned long long *>(workspace + (size_t)MAX_WGS * WAVES * ZSLAB_F4) : nullptr;
#else
        unsigned long long *dbg = nullptr;
#endif
        f32x4 *araw = ATTN ? attn_raw + ((size_t)tile * WAVES + wave) * ATTN_WT_F4 : nullptr;
        float logit = decode_tile<ATTN>(recs, prog + REC_FLOATS, prm, fl, zslab + lane, zs_u, px, py, pz,
                                        lane, dbg, araw);
        if (apply_sigmoid) logit = 1.0f / (1.0f + expf(-logit));
        if (lane < 32 && p < m) out[(size_t)img * m + p] = logit;
    }
}

// attention-visualisation epilogue (implicit.py:63,79,277): attn[img][p][l] = mean over the 16
// (block, head) pairs of softmax probabilities of latent l, from the raw tiles dumped by the
// ATTN kernel: P_raw * 2^(m_ref - m_final) / Z.
__global__ __launch_bounds__(256) void attn_reduce_kernel(const f32x4 *__restrict__ raw,
                                                          float *__restrict__ attn, int batch, int m) {
    const int tiles_per_img = (m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK;
    const long long total = (long long)batch * m * L;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        const int l = (int)(e % L);
        const long long ip = e / L;
        const int p = (int)(ip % m);
        const int img = (int)(ip / m);
        const int t = p / PTS_PER_BLOCK, wave = (p % PTS_PER_BLOCK) / PTS_PER_WAVE, tp = p % PTS_PER_WAVE;
        const f32x4 *w = raw + (((size_t)img * tiles_per_img + t) * WAVES + wave) * ATTN_WT_F4;
        const float *st = reinterpret_cast<const float *>(w + ATTN_P_F4);
        const int lt = l >> 5, row = l & 31;
        const int hi = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
        const int lane = tp + 32 * hi;
        float acc = 0.f;
        for (int bh = 0; bh < BLOCKS * HEADS; bh++) {
            const f32x4 pv = w[((bh * LT + lt) * 4 + (r >> 2)) * 64 + lane];
            const float pr = (r & 3) == 0 ? pv.x : (r & 3) == 1 ? pv.y : (r & 3) == 2 ? pv.z : pv.w;
            const float *sb = st + bh * 9 * 64 + lane;
            acc += pr * __builtin_amdgcn_exp2f(sb[lt * 64] - sb[7 * 64]) * sb[8 * 64];
        }
        attn[e] = acc * (1.0f / (BLOCKS * HEADS));
    }
}

int decode_grid_size(int batch, int m) {
    const long long tiles = (long long)batch * ((m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK);
    return (int)(tiles < MAX_WGS ? tiles : MAX_WGS);
}

}  // namespace

extern "C" size_t zs_sdf_program_bytes(void) { return (size_t)PROGRAM_FLOATS * sizeof(float); }
extern "C" size_t zs_sdf_workspace_bytes(void) { return WORKSPACE_BYTES; }
extern "C" size_t zs_sdf_attn_scratch_bytes(int batch, int m) {
    if (batch <= 0 || m <= 0) return 0;
    const size_t wave_tiles = (size_t)batch * ((m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK) * WAVES;
    return wave_tiles * ATTN_WT_F4 * sizeof(f32x4);
}

extern "C" int zs_sdf_query_points(const void *programs, size_t program_stride_bytes, int batch,
                                   const float *points, int m, float *logits, float *attn,
                                   const int *tile_mask, void *workspace, void *stream) {
    if (batch < 0 || m < 0) {
        zs::set_err("zs_sdf_query_points: negative size (batch=%d m=%d)", batch, m);
        return 0;
    }
    if (batch == 0 || m == 0) return 1;
    if (!programs || !points || !logits || !workspace) {
        zs::set_err("zs_sdf_query_points: null pointer");
        return 0;
    }
    if (program_stride_bytes % 16 != 0 || program_stride_bytes < zs_sdf_program_bytes()) {
        zs::set_err("zs_sdf_query_points: bad program stride %zu", program_stride_bytes);
        return 0;
    }
    if ((long long)batch * ((m + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK) > 0x7fffffffLL) {
        zs::set_err("zs_sdf_query_points: too many tiles");
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!attn) {
        hipLaunchKernelGGL((sdf_decode_kernel<false, false>), dim3(decode_grid_size(batch, m)),
                           dim3(WAVES * 64), 0, st, static_cast<const float *>(programs),
                           program_stride_bytes / sizeof(float), batch, points, nullptr, 0, 0LL, m, logits,
                           0, static_cast<f32x4 *>(workspace), nullptr, tile_mask);
    } else {
        if (tile_mask) {
            zs::set_err("zs_sdf_query_points: the attention map cannot be combined with a tile mask");
            return 0;
        }
        // raw tiles live behind the fixed part of the workspace (zs_sdf_attn_scratch_bytes)
        f32x4 *raw = reinterpret_cast<f32x4 *>(static_cast<char *>(workspace) + WORKSPACE_BYTES);
        hipLaunchKernelGGL((sdf_decode_kernel<false, true>), dim3(decode_grid_size(batch, m)),
                           dim3(WAVES * 64), 0, st, static_cast<const float *>(programs),
                           program_stride_bytes / sizeof(float), batch, points, nullptr, 0, 0LL, m, logits,
                           0, static_cast<f32x4 *>(workspace), raw, nullptr);
        const long long total = (long long)batch * m * L;
        int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(attn_reduce_kernel, dim3(blocks), dim3(256), 0, st, raw, attn, batch, m);
    }
    return zs::check_launch("zs_sdf_query_points") ? 1 : 0;
}

extern "C" int zs_sdf_query_grid_range(const void *programs, size_t program_stride_bytes, int batch,
                                       const float *axis, int G, long long point_begin,
                                       long long point_end, int apply_sigmoid, float *out,
                                       const int *tile_mask, void *workspace, void *stream) {
    const long long P = (long long)G * G * G;
    if (batch < 0 || G <= 0 || point_begin < 0 || point_end > P || point_begin > point_end) {
        zs::set_err("zs_sdf_query_grid_range: bad range (batch=%d G=%d points=[%lld,%lld))", batch, G,
                    point_begin, point_end);
        return 0;
    }
    const long long mm = point_end - point_begin;
    if (batch == 0 || mm == 0) return 1;
    if (!programs || !axis || !out || !workspace) {
        zs::set_err("zs_sdf_query_grid_range: null pointer");
        return 0;
    }
    if (mm > 0x7fffffffLL - PTS_PER_BLOCK) {
        zs::set_err("zs_sdf_query_grid_range: %lld points per launch exceed 2^31; split the range", mm);
        return 0;
    }
    if (program_stride_bytes % 16 != 0 || program_stride_bytes < zs_sdf_program_bytes()) {
        zs::set_err("zs_sdf_query_grid_range: bad program stride %zu", program_stride_bytes);
        return 0;
    }
    const int m = (int)mm;
    hipLaunchKernelGGL((sdf_decode_kernel<true, false>), dim3(decode_grid_size(batch, m)),
                       dim3(WAVES * 64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(programs), program_stride_bytes / sizeof(float), batch,
                       nullptr, axis, G, point_begin, m, out, apply_sigmoid,
                       static_cast<f32x4 *>(workspace), nullptr, tile_mask);
    return zs::check_launch("zs_sdf_query_grid_range") ? 1 : 0;
}

extern "C" int zs_sdf_query_grid(const void *programs, size_t program_stride_bytes, int batch,
                                 const float *axis, int G, int slice_begin, int slice_end,
                                 int apply_sigmoid, float *out, const int *tile_mask, void *workspace,
                                 void *stream) {
    if (batch < 0 || G <= 0 || slice_begin < 0 || slice_end > G || slice_begin > slice_end) {
        zs::set_err("zs_sdf_query_grid: bad range (batch=%d G=%d slices=[%d,%d))", batch, G,
                    slice_begin, slice_end);
        return 0;
    }
    const long long mm = (long long)(slice_end - slice_begin) * G * G;
    if (batch == 0 || mm == 0) return 1;
    if (!programs || !axis || !out || !workspace) {
        zs::set_err("zs_sdf_query_grid: null pointer");
        return 0;
    }
    if (mm > 0x7fffffffLL - PTS_PER_BLOCK) {
        zs::set_err("zs_sdf_query_grid: %lld points per launch exceed 2^31; split the slab", mm);
        return 0;
    }
    if (program_stride_bytes % 16 != 0 || program_stride_bytes < zs_sdf_program_bytes()) {
        zs::set_err("zs_sdf_query_grid: bad program stride %zu", program_stride_bytes);
        return 0;
    }
    const int m = (int)mm;
    hipLaunchKernelGGL((sdf_decode_kernel<true, false>), dim3(decode_grid_size(batch, m)),
                       dim3(WAVES * 64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(programs), program_stride_bytes / sizeof(float), batch,
                       nullptr, axis, G, (long long)slice_begin * G * G, m, out, apply_sigmoid,
                       static_cast<f32x4 *>(workspace), nullptr, tile_mask);
    return zs::check_launch("zs_sdf_query_grid") ? 1 : 0;
}
