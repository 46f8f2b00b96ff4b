// fp32 implicit-GEMM convolution / linear layer on the MFMA pipe (v_mfma_f32_32x32x2_f32), the
// one GEMM engine of the image encoders (DPT-hybrid depth model, ResNet-50 coordinate encoder,
// intrinsics head: SURVEY.md section 8 rows a19-a23).
//
// Activations are channels-last: in [B][Hin][Win][Cin], out [B][Hout][Wout][Cout]; a token
// matrix [n][C] is the 1x1 case with Hin = Win-less geometry (B=1, H=1, W=n).  GEMM view:
//   M = B*Hout*Wout output pixels, N = Cout, K = kh*kw*Cin (tap-major, channel-minor)
//   D[pixel][cout] = sum_k A[pixel][k] * Wt[k][cout]
// Weights are packed once on the host as [K16/4][CoutPad][4] floats (K16 = K rounded up to 16,
// CoutPad = Cout rounded up to 128, zero padded; zeroshape_amd/nn/pack.py) so the B tile is a
// straight float4 copy and a lane's four k values of one MFMA group sit in one register quad.
//
// Tiling: 128 pixels x 128 couts per 256-lane workgroup, wave = 64 x 64 (2 x 2 MFMA tiles, 64
// accumulator registers), K step 16 through a double-buffered LDS ring (2 x 16 KiB) with the
// next step's global loads in flight under the current step's 64 MFMAs per wave.
// LDS layout [k/4][row][4]: lane (row = l%32, half = l/32) reads ONE float4 = its k values
// {4*(2t+half)+s, s=0..3} - conflict-free ds_read_b128, any k permutation is legal because both
// operands use the same one.
//
// Fused into the launch: input transform (ReLU of the input, or a*in_scale+in_shift on in-bounds
// taps, e.g. DPT's 2x-1), per-channel scale/shift (bias or folded BatchNorm), up to two residual
// adds, activation (ReLU / GELU(erf) / ReLU+clamp1).
#include "zs_common.h"
#include "zs_split16.h"
#include "nn_gemm_stream.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, KQ = BK / 4;

struct ConvArgs {
    const float *in, *w, *scale, *shift, *res1, *res2;
    float *out;
    int B, Hin, Win, Cin, Hout, Wout, Cout, CoutPad, kh, kw, stride, pad_t, pad_l, K, M;
    int in_relu, act;
    float in_scale, in_shift;
    int dil;   // 1, or 2: the input is read as if zero-stuffed to (H-1)*2+1 rows / columns (the
               // data-gradient of a stride-2 convolution is a stride-1 convolution over that)
    // split-K across workgroups (small-tile variant): blockIdx.z takes one of `splits` ranges of K and
    // writes its raw partial tile to ws[split][M][Cout]; conv_splitk_reduce_kernel sums them in order
    float *ws;
    int splits;
    // stream-K (large-tile variant): the (tile, k-step) iteration space is cut into gridDim.x equal contiguous
    // ranges; sk_per = iterations per workgroup, ws = [tile arrival counters | partial tiles]
    int sk_per;
    int w_split;   // F16: the weights are already split (zs_conv2d_presplit_weight): quads 4s+q hold the hi halves and
                   // 4s+q+2 the lo halves of the K = 16 operand the lane half q contracts, q = 0, 1
    int slab_major;   // LDS-DMA kernel, kh * kw > 1: walk K as (16-channel slab, tap) instead of (tap, channel) - see SlabWalk
    // conv3x3_patch32_kernel only: fused pointwise tail to one channel (zs_conv3x3_tail_nhwc); out is then [B][H][W]
    const float *tail_w, *tail_b;
    int tail_act;
    // fused normalisations (zs_conv2d_nhwc_fused, small-tile kernel only; include/zeroshape_hip.h: zs_conv_fuse)
    zs_conv_fuse fz;
    int fz_tiles_per_sample;      // input-statistics tiles per sample (in_mode 1 / 3), output tiles per sample (out_mode 1)
};

// workspace layout (floats): [SK_COUNTERS ints, zero between launches][partial tiles / split-K partial sums]
constexpr size_t WS_COUNTER_FLOATS = (size_t)1 << 18;          // 1 MiB: up to 262,144 tiles
constexpr int SK_MAX_WGS = 1024;
constexpr size_t WS_SPLITK_BYTES = (size_t)16 << 20;           // small-tile split-K partial sums
constexpr size_t WS_PARTS_BYTES = (size_t)SK_MAX_WGS * 2 * 128 * 128 * 4;     // the partial-tile region (stream-K shares, patch-kernel split-K)

__device__ __forceinline__ float activate(float v, int act) {
    if (act == ZS_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ZS_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == ZS_ACT_RELU_CLAMP1) return fminf(fmaxf(v, 0.f), 1.f);
    return v;
}

// Four values at once with the activation code known at compile time (ACT) or decided ONCE per quad: the per-element form
// above costs three wave-uniform compare-and-branch pairs per VALUE inside the unrolled epilogues - 500 taken branches per
// wave and tile, through ~60 KiB of instructions, were most of the 30,000-cycle epilogue of the 256 x 256 kernel (round 5)
template <int ACT>
__device__ __forceinline__ void activate4_t(f32x4 &v) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (ACT == ZS_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
        if (ACT == ZS_ACT_GELU) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
        if (ACT == ZS_ACT_RELU_CLAMP1) v[e] = fminf(fmaxf(v[e], 0.f), 1.f);
    }
}
__device__ __forceinline__ void activate4(f32x4 &v, int act) {
    if (act == ZS_ACT_NONE) return;
    if (act == ZS_ACT_RELU) activate4_t<ZS_ACT_RELU>(v);
    else if (act == ZS_ACT_GELU) activate4_t<ZS_ACT_GELU>(v);
    else activate4_t<ZS_ACT_RELU_CLAMP1>(v);
}

// position of a lane's current k (4 consecutive channels of one tap) in (ky, kx, channel) form,
// advanced incrementally: no integer division in the K loop
struct TapIter {
    int c, ky, kx;
    __device__ __forceinline__ void init(int k, int Cin, int kw) {
        const int tap = k / Cin;
        c = k - tap * Cin;
        ky = tap / kw;
        kx = tap - ky * kw;
    }
    // k += step, with step = adv_tap * Cin + adv_c precomputed; branch-free (selects), exact while
    // adv_tap + 1 <= 2 * kw, and monotone in ky beyond that (positions past K only need ky >= kh)
    __device__ __forceinline__ void advance(int adv_tap, int adv_c, int Cin, int kw) {
        c += adv_c;
        const int wrap = c >= Cin ? 1 : 0;
        c -= wrap ? Cin : 0;
        kx += adv_tap + wrap;
        int w1 = kx >= kw ? 1 : 0;
        kx -= w1 ? kw : 0;
        ky += w1;
        w1 = kx >= kw ? 1 : 0;
        kx -= w1 ? kw : 0;
        ky += w1;
        ky += kx >= kw ? 1 << 20 : 0;          // still out of range: far beyond K, poison ky
    }
};

// Tap-major walk (Cin a multiple of the k step): a whole k step lies inside ONE tap, so validity
// and the pixel offset are recomputed only when the tap changes (every Cin / step iterations, a
// wave-uniform branch) and the per-step work is one add - instead of the TapIter's ~35 VALU
// instructions per quad, which the fp32 MFMA pipe cannot hide (DESIGN.md sections 3, 11.5).
struct TapWalk {
    int tap, cc, ky, kx;          // uniform across the workgroup / wave
    bool ok;                      // this lane's pixel has an in-range input at the current tap
    size_t base;                  // offset of that input pixel (channel 0)
    __device__ __forceinline__ void set_tap(const ConvArgs &a, bool pix_ok, int iy0, int ix0) {
        const int vy = iy0 + ky, vx = ix0 + kx, sh = a.dil - 1;
        const int iy = vy >> sh, ix = vx >> sh;
        ok = pix_ok && ky < a.kh && vy >= 0 && iy < a.Hin && vx >= 0 && ix < a.Win && ((vy | vx) & sh) == 0;
        base = ok ? ((size_t)iy * a.Win + ix) * a.Cin : 0;
    }
    __device__ __forceinline__ void init(const ConvArgs &a, int k, bool pix_ok, int iy0, int ix0) {
        tap = k / a.Cin;
        cc = k - tap * a.Cin;
        ky = tap / a.kw;
        kx = tap - ky * a.kw;
        set_tap(a, pix_ok, iy0, ix0);
    }
    __device__ __forceinline__ void advance(const ConvArgs &a, int step, bool pix_ok, int iy0, int ix0) {
        cc += step;
        if (cc >= a.Cin) {        // uniform
            cc -= a.Cin;
            tap++;
            kx++;
            if (kx == a.kw) { kx = 0; ky++; }
            set_tap(a, pix_ok, iy0, ix0);
        }
    }
};

// Slab-major walk (round 3 experiment, LDS-DMA kernel, opt-in ZS_CONV_SLAB_MAJOR=1): the k steps of a kh x kw layer in
// the order (16-channel slab outer, tap inner) instead of (tap outer, channels inner).  Hypothesis: the
// tap-major order streams all channels of a tap - 131 KiB per 128-row tile at 56 x 56 x 256 - before the next tap touches
// the same input pixels again, 32 concurrent tiles per XCD overflow its 4 MiB L2, so every tap re-reads its activations
// from beyond L2; slab-major puts the nine reads of an input pixel's 64-byte slab back to back.  MEASURED (tools/
// conv_shapes.py, batch 28, same box): SLOWER - the 3 x 3 256 -> 256 layer at 56 x 56 516 vs 432 us, the 128 -> 32 head
// at 224 x 224 1,611 vs 1,370 us, 20.2 vs 19.3 ms of convolutions per forward.  Where the re-reads are served from does
// not matter: the kernel is bound by the BYTES it moves into LDS (the same in both orders), and tap-major reads each
// pixel's channels as one sequential run over consecutive steps.  Only staging fewer bytes helps (an input patch kept
// in LDS across the nine taps) - not built.  Same products, summed in another order; weights addressed by (tap, slab).
struct SlabWalk {
    int tap, cc, ky, kx;          // uniform across the workgroup / wave
    bool ok;
    size_t base;
    __device__ __forceinline__ void set_tap(const ConvArgs &a, bool pix_ok, int iy0, int ix0) {
        const int vy = iy0 + ky, vx = ix0 + kx, sh = a.dil - 1;
        const int iy = vy >> sh, ix = vx >> sh;
        ok = pix_ok && ky < a.kh && vy >= 0 && iy < a.Hin && vx >= 0 && ix < a.Win && ((vy | vx) & sh) == 0;
        base = ok ? ((size_t)iy * a.Win + ix) * a.Cin : 0;
    }
    __device__ __forceinline__ void init(const ConvArgs &a, int step, bool pix_ok, int iy0, int ix0) {
        const int ntaps = a.kh * a.kw;
        if (a.slab_major) {
            const int slab = step / ntaps;
            tap = step - slab * ntaps;
            cc = slab * BK;
        } else {
            const int k = BK * step;
            tap = k / a.Cin;
            cc = k - tap * a.Cin;
        }
        ky = tap / a.kw;
        kx = tap - ky * a.kw;
        set_tap(a, pix_ok, iy0, ix0);
    }
    __device__ __forceinline__ int kquad(const ConvArgs &a) const { return (tap * a.Cin + cc) >> 2; }   // row quad of the weights
    __device__ __forceinline__ void advance(const ConvArgs &a, bool pix_ok, int iy0, int ix0) {
        if (a.slab_major) {       // uniform
            tap++;
            kx++;
            if (kx == a.kw) { kx = 0; ky++; }
            if (tap == a.kh * a.kw) { tap = 0; kx = 0; ky = 0; cc += BK; }
            set_tap(a, pix_ok, iy0, ix0);
        } else {
            cc += BK;
            if (cc >= a.Cin) {
                cc -= a.Cin;
                tap++;
                kx++;
                if (kx == a.kw) { kx = 0; ky++; }
                set_tap(a, pix_ok, iy0, ix0);
            }
        }
    }
};

// One A operand quad: issued unconditionally from a clamped address so the load stays in flight
// (no branch, no wait); `ok` is applied when the value is used.
struct AQuad { f32x4 v; bool ok; };

// PW: pointwise fast path (1x1, stride 1, no padding / dilation / input transform - every linear layer
// and bottleneck 1x1): the operand address is pixel * Cin + k, no tap bookkeeping and no transform, which
// removes most of the VALU work that the fp32 MFMA pipe cannot hide.
// MODE 0: generic taps.  MODE 2 (TM): tap-major walk, Cin % 16 == 0.  PLAIN: no input transform.
// F16: split-fp16 arithmetic (zs_split16.h).  A thread stages the quads kq_lo and kq_lo + 2 of its row -
// exactly the eight k values the MFMA lane (row, half = kq_lo) contracts over - so it splits them
// once into the hi / lo operand halves and stores those where the two fp32 quads went; the MFMA
// loop then issues 3 K = 16 MFMAs per tile pair instead of 8 K = 2 ones at a quarter of the rate.
template <bool PW, int MODE = 0, bool PLAIN = false, bool F16 = false, bool SK = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs a) {
    constexpr bool TM = MODE == 2;
    __shared__ f32x4 lds_a[2][KQ][BM];
    __shared__ f32x4 lds_b[2][KQ][BN];
    __shared__ int sk_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksteps = (a.K + BK - 1) / BK;
    const int ntiles = a.CoutPad / BN;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64, l32 = lane & 31, half = lane >> 5;
    const int prow = tid & (BM - 1), kq_lo = tid >> 7;       // this thread stages row prow, k-quads kq_lo and kq_lo + 2
    const float relu_floor = a.in_relu ? 0.f : -INFINITY;
    const int adv_tap = BK / a.Cin, adv_c = BK % a.Cin;
    const f32x4 *wq = reinterpret_cast<const f32x4 *>(a.w);

    // stream-K: this workgroup's contiguous range of the iteration space (tile-major, n fastest, then k-steps);
    // otherwise one tile per workgroup, all of K
    int it = SK ? (int)blockIdx.x * a.sk_per : ((int)blockIdx.x * ntiles + (int)blockIdx.y) * ksteps;
    const int it_begin = it;
    const int it_end = SK ? min(it + a.sk_per, (int)(((a.M + BM - 1) / BM) * ntiles) * ksteps) : it + ksteps;

    while (it < it_end) {
    const int tile = it / ksteps, kb = it - tile * ksteps, ke = min(ksteps, kb + (it_end - it));
    const int m0 = (tile / ntiles) * BM, n0 = (tile % ntiles) * BN;

    // this thread's A pixel
    const int pix = m0 + prow;
    const bool pix_ok = pix < a.M;
    int pb = 0, py = 0, px = 0;
    if (pix_ok) {
        pb = pix / (a.Hout * a.Wout);
        const int rem = pix - pb * a.Hout * a.Wout;
        py = rem / a.Wout;
        px = rem - py * a.Wout;
    }
    const int iy0 = py * a.stride - a.pad_t, ix0 = px * a.stride - a.pad_l;
    const float *in_b = a.in + (size_t)pb * a.Hin * a.Win * a.Cin;

    auto load_a = [&](const TapIter &ti) -> AQuad {            // 4 consecutive channels of one tap
        const int vy = iy0 + ti.ky, vx = ix0 + ti.kx, sh = a.dil - 1;      // dil 1 -> 0, dil 2 -> 1
        const int iy = vy >> sh, ix = vx >> sh;
        AQuad q;
        q.ok = pix_ok && ti.ky < a.kh && vy >= 0 && iy < a.Hin && vx >= 0 && ix < a.Win && ((vy | vx) & sh) == 0;
        const size_t off = q.ok ? ((size_t)iy * a.Win + ix) * a.Cin + ti.c : 0;
        q.v = *reinterpret_cast<const f32x4 *>(in_b + off);
        return q;
    };
    auto finish_a = [&](const AQuad &q) -> f32x4 {             // input transform, zero for padding
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = q.ok ? fmaxf(q.v[e], relu_floor) * a.in_scale + a.in_shift : 0.f;
        return v;
    };
    TapIter it0, it1;
    if (!PW && !TM) {
        it0.init(BK * kb + 4 * kq_lo, a.Cin, a.kw);
        it1.init(BK * kb + 4 * (kq_lo + 2), a.Cin, a.kw);
    }
    const float *in_pix = a.in + (size_t)(pix_ok ? pix : 0) * a.Cin;       // PW
    int ka0 = BK * kb + 4 * kq_lo, ka1 = BK * kb + 4 * (kq_lo + 2);
    auto load_pw = [&](int k) -> AQuad {
        AQuad q;
        q.ok = pix_ok && k < a.K;
        q.v = *reinterpret_cast<const f32x4 *>(in_pix + (q.ok ? k : 0));
        return q;
    };
    auto finish_pw = [&](const AQuad &q) -> f32x4 { return q.ok ? q.v : f32x4{0.f, 0.f, 0.f, 0.f}; };
    TapWalk tw;                                                             // TM
    if (TM) tw.init(a, BK * kb, pix_ok, iy0, ix0);
    auto load_tm = [&](int kq) -> AQuad {                                   // quad kq (0..3) of the current 16-chunk
        AQuad q;
        q.ok = tw.ok;
        q.v = *reinterpret_cast<const f32x4 *>(in_b + tw.base + (tw.ok ? tw.cc + 4 * kq : 0));
        return q;
    };
    auto finish_any = [&](const AQuad &q) -> f32x4 {
        if (PW || PLAIN) return finish_pw(q);
        return finish_a(q);
    };
    auto load_b = [&](int kq, int n) -> f32x4 { return wq[(size_t)kq * a.CoutPad + n0 + n]; };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    AQuad ra[2];
    f32x4 rb[2];
    ra[0] = PW ? load_pw(ka0) : (TM ? load_tm(kq_lo) : load_a(it0));
    ra[1] = PW ? load_pw(ka1) : (TM ? load_tm(kq_lo + 2) : load_a(it1));
    rb[0] = load_b(kb * KQ + kq_lo, prow);
    rb[1] = load_b(kb * KQ + kq_lo + 2, prow);
    auto stage = [&](int buf) {
        if (F16) {
            u32x4 hi, lo;
            zs::s16::split8(finish_any(ra[0]), finish_any(ra[1]), hi, lo);
            lds_a[buf][kq_lo][prow] = __builtin_bit_cast(f32x4, hi);
            lds_a[buf][kq_lo + 2][prow] = __builtin_bit_cast(f32x4, lo);
            if (a.w_split) {
                lds_b[buf][kq_lo][prow] = rb[0];
                lds_b[buf][kq_lo + 2][prow] = rb[1];
            } else {
                zs::s16::split8(rb[0], rb[1], hi, lo);
                lds_b[buf][kq_lo][prow] = __builtin_bit_cast(f32x4, hi);
                lds_b[buf][kq_lo + 2][prow] = __builtin_bit_cast(f32x4, lo);
            }
        } else {
            lds_a[buf][kq_lo][prow] = finish_any(ra[0]);
            lds_a[buf][kq_lo + 2][prow] = finish_any(ra[1]);
            lds_b[buf][kq_lo][prow] = rb[0];
            lds_b[buf][kq_lo + 2][prow] = rb[1];
        }
    };
    stage(0);
    __syncthreads();

    for (int ks = kb; ks < ke; ks++) {
        const int cur = (ks - kb) & 1;
        const bool more = ks + 1 < ke;
        if (more) {
            if (PW) {
                ka0 += BK;
                ka1 += BK;
                ra[0] = load_pw(ka0);
                ra[1] = load_pw(ka1);
            } else if (TM) {
                tw.advance(a, BK, pix_ok, iy0, ix0);
                ra[0] = load_tm(kq_lo);
                ra[1] = load_tm(kq_lo + 2);
            } else {
                it0.advance(adv_tap, adv_c, a.Cin, a.kw);
                it1.advance(adv_tap, adv_c, a.Cin, a.kw);
                ra[0] = load_a(it0);
                ra[1] = load_a(it1);
            }
            rb[0] = load_b((ks + 1) * KQ + kq_lo, prow);
            rb[1] = load_b((ks + 1) * KQ + kq_lo + 2, prow);
        }
        if (F16) {
            u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                ah[i] = __builtin_bit_cast(u32x4, lds_a[cur][half][wm + 32 * i + l32]);
                al[i] = __builtin_bit_cast(u32x4, lds_a[cur][2 + half][wm + 32 * i + l32]);
                bh[i] = __builtin_bit_cast(u32x4, lds_b[cur][half][wn + 32 * i + l32]);
                bl[i] = __builtin_bit_cast(u32x4, lds_b[cur][2 + half][wn + 32 * i + l32]);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) zs::s16::mfma3(acc[i][j], bh[j], bl[j], ah[i], al[i]);   // transposed: rows = channels
        } else
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f32x4 fa[2], fb[2];
            fa[0] = lds_a[cur][2 * t + half][wm + l32];
            fa[1] = lds_a[cur][2 * t + half][wm + 32 + l32];
            fb[0] = lds_b[cur][2 * t + half][wn + l32];
            fb[1] = lds_b[cur][2 * t + half][wn + 32 + l32];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][s], fa[i][s], acc[i][j], 0, 0, 0);   // D^T
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
    }

    // epilogue of one 32x32 tile.  The products are formed transposed (the weights are the MFMA's A operand): register
    // 4 q + e of lane (l32, half) = D[pixel row l32][channel 8 q + 4 half + e] - four consecutive channels of one pixel,
    // moved as 16 bytes per lane (a quarter of the epilogue's vector-memory instructions; same sums, same order)
    auto epilogue = [&](int i, int j, const f32x16 &d) {
        const int m = m0 + wm + 32 * i + l32;
        if (m >= a.M) return;
        const size_t row = (size_t)m * a.Cout;
        const bool vec = (a.Cout & 3) == 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n = n0 + wn + 32 * j + 8 * q + 4 * half;
            if (n >= a.Cout) continue;
            f32x4 v = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
            if (vec) {
                if (a.scale) v *= *reinterpret_cast<const f32x4 *>(a.scale + n);
                if (a.shift) v += *reinterpret_cast<const f32x4 *>(a.shift + n);
                if (a.res1) v += *reinterpret_cast<const f32x4 *>(a.res1 + row + n);
                if (a.res2) v += *reinterpret_cast<const f32x4 *>(a.res2 + row + n);
                activate4(v, a.act);
                *reinterpret_cast<f32x4 *>(a.out + row + n) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (n + e >= a.Cout) continue;
                    const size_t o = row + n + e;
                    float t = v[e] * (a.scale ? a.scale[n + e] : 1.0f) + (a.shift ? a.shift[n + e] : 0.0f);
                    if (a.res1) t += a.res1[o];
                    if (a.res2) t += a.res2[o];
                    a.out[o] = activate(t, a.act);
                }
            }
        }
    };
    if (!SK || (kb == 0 && ke == ksteps)) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) epilogue(i, j, acc[i][j]);
    } else {
        // A share of a tile: park the raw accumulators (slot 0 = this workgroup's first segment, slot 1 = its last),
        // count arrivals; the workgroup that arrives last sums all shares in k order - the same sum whoever does
        // it - and runs the epilogue.  No workgroup ever waits for another.  Shares move as agent-scope relaxed atomics
        // (sc1: written through to / read from memory, the coherence point of the eight XCDs' L2s), drained by the store
        // counter before the workgroup barrier; the arrival counter is an ACQ_REL read-modify-write by one thread per
        // workgroup (release: the shares written before the barrier; acquire: the last arriver reads the others' shares) -
        // the ordering no longer rests on what gfx950 and hipcc happen to do.  (A __threadfence() in every
        // thread costs 3x the whole kernel here, measured.)  The sum is formed in fresh registers, tile by tile: the
        // accumulators stay in the AGPR half and the kernel keeps three workgroups per CU.
        float *parts = a.ws + WS_COUNTER_FLOATS;
        int *counters = reinterpret_cast<int *>(a.ws);
        float *mine = parts + ((size_t)blockIdx.x * 2 + (it == it_begin ? 0 : 1)) * (BM * BN) + tid;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    __hip_atomic_store(&mine[((i * 2 + j) * 16 + r) * 256], acc[i][j][r], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::: "memory");                 // one tile's registers at a time
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's shares have reached memory
        __syncthreads();
        const int w_first = (tile * ksteps) / a.sk_per, w_last = ((tile + 1) * ksteps - 1) / a.sk_per;
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(&counters[tile], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            sk_last = old == w_last - w_first;
            if (sk_last)                                      // ready for the next launch
                __hip_atomic_store(&counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (sk_last) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    f32x16 d;
#pragma unroll
                    for (int r = 0; r < 16; r++) d[r] = 0.f;
#pragma unroll 1
                    for (int w = w_first; w <= w_last; w++) {
                        const int w_tile0 = (w * a.sk_per) / ksteps;   // the tile of w's first iteration
                        const float *p = parts + ((size_t)w * 2 + (w_tile0 == tile ? 0 : 1)) * (BM * BN) + tid;
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            d[r] += __hip_atomic_load(&p[((i * 2 + j) * 16 + r) * 256], __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
                    }
                    epilogue(i, j, d);
                    asm volatile("" ::: "memory");
                }
        }
        __syncthreads();          // sk_last is rewritten by the next segment
    }
    it += ke - kb;
    }
}


// ---- split-fp16, LDS-DMA pipelined variant of the 128x128 tiling ----------------------------------------- //
// The kernel above keeps one k-step of operands in flight in registers; measured (tools/prof_conv.sh, ViT fc1 at
// batch 28) a wave spends ~2,800 cycles per step of 384 MFMA cycles waiting for those loads, three waves per SIMD
// do not cover it (MFMA pipe 25 % busy).  Here both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4:
// no registers, no VALU), NS stages deep, DEPTH = NS - 1 steps ahead: A as raw fp32 quads (split into the fp16
// halves by the consuming wave, on the LDS -> register path), B from the pre-split weights
// (zs_conv2d_presplit_weight).  Out-of-range rows / taps read a 16-byte zero page.  Per step and wave: wait for
// the own DMAs of this step (counted vmcnt) -> s_barrier (everybody's landed; everybody is done reading the
// stage about to be refilled) -> issue the DMAs of step + DEPTH -> 8 ds_read_b128 -> 2 split8 -> 12 MFMAs.
// Handles the pointwise and the tap-major (Cin % 16 == 0) geometries without input scale / shift.
__device__ f32x4 zs_zero_page[4];


__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {   // 64 lanes x 16 B -> LDS[dst + 16 lane]
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vm_then_barrier() {       // all but the N youngest vector-memory operations done
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(N) : "memory");
}

// DMA_NS stages of 16 KiB: 3 -> three workgroups per CU, two steps of lead.  (9 stages, one workgroup per CU, eight
// steps of lead measured SLOWER - 632 vs 471 us on the 3x3 layer: the limit is the rate of the stream, not its latency.)
// MI = 32-row MFMA tiles per wave along M: 2 -> 128 x 128 workgroup tile (three workgroups per CU), 4 -> 256 x 128 (two
// per CU, 25 % fewer bytes through LDS per MFMA: 24 KiB instead of 2 x 16 per 256 x 128 outputs and k-step)
template <bool PW, bool RELU, bool SK, int DMA_NS, int MI = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DMA_NS == 2 ? 4 : 2, 8))) void conv_gemm_dma_kernel(ConvArgs a) {
    constexpr int DMA_DEPTH = DMA_NS - 1;
    constexpr int TMB = 64 * MI;                              // rows of the workgroup tile
    constexpr int NDMA = MI + 2;                              // DMAs per wave and step: A row blocks of 64, B columns x 2
    constexpr int STAGE_BYTES = (TMB + BN) * BK * 4;
    __shared__ f32x4 lds[DMA_NS][KQ * (TMB + BN)];            // [stage][A: [k-quad][row] | B: [k-quad][column]]
    __shared__ __attribute__((aligned(16))) float ep_lds[2][BN];   // the tile's per-channel scale | shift (epilogue reads them as
                                                                  // ds_read_b128: no vector-memory instructions for them)
    __shared__ int sk_last;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4 *)&lds[0][0]);
    const int ksteps = (a.K + BK - 1) / BK;
    const int ntiles = a.CoutPad / BN;
    const int wm = (wave & 1) * (32 * MI), wn = (wave >> 1) * 64, l32 = lane & 31, half = lane >> 5;
    const char *zero = reinterpret_cast<const char *>(zs_zero_page);
    const f32x4 *wq = reinterpret_cast<const f32x4 *>(a.w);

    // plain split-K (a.splits > 1, not SK): blockIdx.z owns the k-steps [z, z + 1) * ksteps / splits of its tile and writes the
    // raw partial tile to ws[z][M][Cout]; conv_splitk_reduce_kernel sums the ranges in order and applies the epilogue
    const int split_lo = (!SK && a.splits > 1) ? (int)(((long long)blockIdx.z * ksteps) / a.splits) : 0;
    const int split_n = (!SK && a.splits > 1) ? (int)(((long long)(blockIdx.z + 1) * ksteps) / a.splits) - split_lo : ksteps;
    int it = SK ? (int)blockIdx.x * a.sk_per : ((int)blockIdx.x * ntiles + (int)blockIdx.y) * ksteps + split_lo;
    const int it_begin = it;
    const int it_end = SK ? min(it + a.sk_per, (int)(((a.M + TMB - 1) / TMB) * ntiles) * ksteps) : it + split_n;

    while (it < it_end) {
    const int tile = it / ksteps, kb = it - tile * ksteps, ke = min(ksteps, kb + (it_end - it));
    const int m0 = (tile / ntiles) * TMB, n0 = (tile % ntiles) * BN;
    {   // visible to the epilogue through the barriers of the k loop (the previous tile's epilogue is behind a barrier too)
        const int which = tid >> 7, c = tid & (BN - 1), n = n0 + c;
        const float *src = which ? a.shift : a.scale;
        ep_lds[which][c] = (src && n < a.Cout) ? src[n] : (which ? 0.0f : 1.0f);
    }

    // A is staged COALESCED: one DMA covers 16 rows x 64 B (lane = 4 (row % 16) + slot: each row's 16 channels of the
    // step are one contiguous 64 B run - 16 cache lines per instruction; a lane per row, the obvious mapping, touches 64
    // lines per instruction and is bound by the vector cache's tag rate, not by bytes).  LDS holds A as
    // [row][4 slots], slot = quad ^ ((row >> 2) & 3): the MFMA fragment reads (32 rows, one quad) stay conflict-free.
    // This wave issues instructions wave * MI + h (rows 16 (wave MI + h) + lane / 4), B: k-quad `wave`, columns lane, + 64.
    const int a_slot = lane & 3;
    SlabWalk tw[MI];
    const float *src[MI];                                     // PW: the pixel's row; TM: the image
    bool pok[MI];
    int iy0[MI], ix0[MI];
#pragma unroll
    for (int h = 0; h < MI; h++) {
        const int arow = 16 * (wave * MI + h) + (lane >> 2);
        const int pix = m0 + arow;
        pok[h] = pix < a.M;
        if (PW) {
            src[h] = a.in + (size_t)(pok[h] ? pix : 0) * a.Cin;
        } else {
            int pb = 0, py = 0, px = 0;
            if (pok[h]) {
                pb = pix / (a.Hout * a.Wout);
                const int rem = pix - pb * a.Hout * a.Wout;
                py = rem / a.Wout;
                px = rem - py * a.Wout;
            }
            iy0[h] = py * a.stride - a.pad_t;
            ix0[h] = px * a.stride - a.pad_l;
            src[h] = a.in + (size_t)pb * a.Hin * a.Win * a.Cin;
            tw[h].init(a, kb, pok[h], iy0[h], ix0[h]);
        }
    }
    const f32x4 *wcol = wq + n0 + lane;
    int ks_issue = kb;                                        // next step to stage
    // the k-quad this lane fetches of its rows (the slot it lands in is lane & 3); rows of one instruction share row >> 2 & 3
    // only pairwise, so the quad is per lane: quad = slot ^ ((row >> 2) & 3), row = 16 (..) + lane / 4 -> (lane >> 4) & 3
    const int a_quad = a_slot ^ ((lane >> 4) & 3);
    auto issue = [&]() {                                      // NDMA DMAs: A row blocks x MI, B columns x 2 (k-quad = wave)
        const int st = (ks_issue - kb) % DMA_NS;
        const unsigned dst = lds_base + st * STAGE_BYTES + wave * (MI * 1024);
        const unsigned dst_b = lds_base + st * STAGE_BYTES + KQ * TMB * 16 + wave * (BN * 16);
        const bool live = ks_issue < ke;                      // past the end: keep the vmcnt arithmetic, fetch zeros
        const int wquad = PW ? ks_issue * KQ : tw[0].kquad(a);    // first weight row quad of this step (uniform)
#pragma unroll
        for (int h = 0; h < MI; h++) {
            const char *g;
            if (PW) {
                const int k = BK * ks_issue + 4 * a_quad;
                g = (live && pok[h] && k < a.K) ? reinterpret_cast<const char *>(src[h] + k) : zero;
            } else {
                g = (live && tw[h].ok) ? reinterpret_cast<const char *>(src[h] + tw[h].base + tw[h].cc + 4 * a_quad) : zero;
                if (live) tw[h].advance(a, pok[h], iy0[h], ix0[h]);
            }
#if !defined(ZS_EXP_CONV_NO_DMA) && !defined(ZS_EXP_CONV_NO_DMA_A)
            dma16(g, dst + h * 1024);
#endif
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const char *g = live ? reinterpret_cast<const char *>(wcol + (size_t)(wquad + wave) * a.CoutPad + 64 * h) : zero;
#if !defined(ZS_EXP_CONV_NO_DMA) && !defined(ZS_EXP_CONV_NO_DMA_B)
            dma16(g, dst_b + h * 1024);
#endif
        }
        ks_issue++;
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // everybody is done with the stages (previous segment's reads) before they are refilled
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int d = 0; d < DMA_DEPTH; d++) issue();

    const float relu_floor = RELU ? 0.f : -INFINITY;
    for (int ks = kb; ks < ke; ks++) {
        static_assert(NDMA * (DMA_DEPTH - 1) < 64, "vmcnt is a 6-bit counter");
        wait_vm_then_barrier<NDMA * (DMA_DEPTH - 1)>();       // the younger steps' DMAs (NDMA each) may stay in flight
        issue();
        const int st = (ks - kb) % DMA_NS;
        u32x4 ah[MI], al[MI], bh[2], bl[2];
        f32x4 fa[MI][2];
        const f32x4 *sa = &lds[st][0], *sb = &lds[st][KQ * TMB];
#pragma unroll
        for (int i = 0; i < MI; i++) {
            const int R = wm + 32 * i + l32, sw = (R >> 2) & 3;
            fa[i][0] = sa[R * 4 + (half ^ sw)];
            fa[i][1] = sa[R * 4 + ((half + 2) ^ sw)];
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            bh[j] = __builtin_bit_cast(u32x4, sb[half * BN + wn + 32 * j + l32]);
            bl[j] = __builtin_bit_cast(u32x4, sb[(half + 2) * BN + wn + 32 * j + l32]);
        }
#pragma unroll
        for (int i = 0; i < MI; i++) {
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    fa[i][0][e] = fmaxf(fa[i][0][e], relu_floor);
                    fa[i][1][e] = fmaxf(fa[i][1][e], relu_floor);
                }
            }
#ifdef ZS_EXP_CONV_NO_SPLIT
            ah[i] = __builtin_bit_cast(u32x4, fa[i][0]);
            al[i] = __builtin_bit_cast(u32x4, fa[i][1]);
#else
            zs::s16::split8(fa[i][0], fa[i][1], ah[i], al[i]);
#endif
        }
#ifdef ZS_EXP_CONV_NO_MFMA
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][j][e] += __builtin_bit_cast(float, ah[i][e] ^ al[i][e] ^ bh[j][e] ^ bl[j][e]);
#else
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) zs::s16::mfma3(acc[i][j], bh[j], bl[j], ah[i], al[i]);   // transposed: rows = channels
#endif
    }

    // The MFMAs take the weights as their A operand: register 4 q + e of lane (l32, half) is output channel 8 q + 4 half + e
    // of row l32 of the block - four consecutive channels of one pixel, moved as 16 bytes per lane (a quarter of the
    // epilogue's vector-memory instructions; the 1x1 layers with K = 64..256 are mostly epilogue).
    auto epilogue = [&](int i, int j, const f32x16 &d) {
        const int m = m0 + wm + 32 * i + l32;
        if (m >= a.M) return;
        float *part = (!SK && a.splits > 1) ? a.ws + WS_COUNTER_FLOATS + (size_t)blockIdx.z * a.M * a.Cout : nullptr;
        const size_t row = (size_t)m * a.Cout;
        const bool vec = (a.Cout & 3) == 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n = n0 + wn + 32 * j + 8 * q + 4 * half;
            if (n >= a.Cout) continue;
            f32x4 v = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
            if (vec) {
                if (part) { *reinterpret_cast<f32x4 *>(part + row + n) = v; continue; }
                v = v * *reinterpret_cast<const f32x4 *>(&ep_lds[0][n - n0]) + *reinterpret_cast<const f32x4 *>(&ep_lds[1][n - n0]);
                if (a.res1) v += *reinterpret_cast<const f32x4 *>(a.res1 + row + n);
                if (a.res2) v += *reinterpret_cast<const f32x4 *>(a.res2 + row + n);
                activate4(v, a.act);
                *reinterpret_cast<f32x4 *>(a.out + row + n) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (n + e >= a.Cout) continue;
                    const size_t o = row + n + e;
                    if (part) { part[o] = v[e]; continue; }
                    float t = v[e] * (a.scale ? a.scale[n + e] : 1.0f) + (a.shift ? a.shift[n + e] : 0.0f);
                    if (a.res1) t += a.res1[o];
                    if (a.res2) t += a.res2[o];
                    a.out[o] = activate(t, a.act);
                }
            }
        }
    };
    if (!SK || (kb == 0 && ke == ksteps)) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < MI; i++) epilogue(i, j, acc[i][j]);
    } else {   // a share of a tile: see conv_gemm_kernel
        float *parts = a.ws + WS_COUNTER_FLOATS;
        int *counters = reinterpret_cast<int *>(a.ws);
        float *mine = parts + ((size_t)blockIdx.x * 2 + (it == it_begin ? 0 : 1)) * (TMB * BN) + tid;
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    __hip_atomic_store(&mine[((i * 2 + j) * 16 + r) * 256], acc[i][j][r], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::: "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int w_first = (tile * ksteps) / a.sk_per, w_last = ((tile + 1) * ksteps - 1) / a.sk_per;
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(&counters[tile], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            sk_last = old == w_last - w_first;
            if (sk_last) __hip_atomic_store(&counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (sk_last) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < MI; i++) {
                    f32x16 d;
#pragma unroll
                    for (int r = 0; r < 16; r++) d[r] = 0.f;
#pragma unroll 1
                    for (int w = w_first; w <= w_last; w++) {
                        const int w_tile0 = (w * a.sk_per) / ksteps;
                        const float *p = parts + ((size_t)w * 2 + (w_tile0 == tile ? 0 : 1)) * (TMB * BN) + tid;
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            d[r] += __hip_atomic_load(&p[((i * 2 + j) * 16 + r) * 256], __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
                    }
                    epilogue(i, j, d);
                    asm volatile("" ::: "memory");
                }
        }
        __syncthreads();
    }
    it += ke - kb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the zero-page DMAs issued past the end
}

// ---- 256 x 256 tiles, eight waves, stream-K: the variant for large layers ------------------------------------ //
// tools/conv_abl.sh: with the DMAs removed the 128 x 128 kernel above runs 2.6x faster (3x3 256 -> 256 at 56 x 56 x
// 28: 186 vs 490 us) - it is bound by the bytes it moves from L2 into LDS (1 KiB per k for 128 x 128 outputs, ~9 TB/s
// in aggregate), not by latency or the matrix pipe.  A 256 x 256 tile moves half the bytes per MFMA.  One workgroup
// of eight waves per CU (wave tile 64 x 128, two waves per SIMD), four 32 KiB stages, three steps of lead, always
// as stream-K over 256 persistent workgroups (with 256 slots every layer shape would otherwise round up badly).
constexpr int TM2 = 256, TN2 = 256, NS2 = 4, DEPTH2 = NS2 - 1;
constexpr int STAGE2_BYTES = (TM2 + TN2) * BK * 4;           // 32 KiB: A [4 quads][256 rows][16 B], then B the same

template <bool PW, bool RELU>
__global__ __launch_bounds__(512) void conv_gemm_dma256_kernel(ConvArgs a) {
    __shared__ f32x4 lds[NS2][2][KQ][TM2];
    __shared__ int sk_last;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4 *)&lds[0][0][0][0]);
    const int ksteps = (a.K + BK - 1) / BK;
    const int ntiles = (a.CoutPad + TN2 - 1) / TN2;
    const int wm = (wave & 3) * 64, wn = (wave >> 2) * 128, l32 = lane & 31, half = lane >> 5;
    const int kq_w = wave & 3, blk_w = wave >> 2;              // this wave stages k-quad kq_w of row / column block blk_w
    const char *zero = reinterpret_cast<const char *>(zs_zero_page);
    const f32x4 *wq = reinterpret_cast<const f32x4 *>(a.w);

    // workgroups are dealt to the eight XCDs round-robin: give each XCD one contiguous eighth of the iteration space
    // (neighbouring tiles share A rows / B columns through that XCD's own L2)
#ifdef ZS_EXP_CONV_NO_XCD_MAP
    const int wg = (int)blockIdx.x;
#else
    const int wg = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#endif
    int it = wg * a.sk_per;
    const int it_begin = it;
    const int it_end = min(it + a.sk_per, (int)(((a.M + TM2 - 1) / TM2) * ntiles) * ksteps);

    while (it < it_end) {
    const int tile = it / ksteps, kb = it - tile * ksteps, ke = min(ksteps, kb + (it_end - it));
    const int m0 = (tile / ntiles) * TM2, n0 = (tile % ntiles) * TN2;

    TapWalk tw[2];
    const float *src[2];
    bool pok[2], nok[2];
    int iy0[2], ix0[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int pix = m0 + blk_w * 128 + 64 * h + lane;
        pok[h] = pix < a.M;
        nok[h] = n0 + blk_w * 128 + 64 * h + lane < a.CoutPad;
        if (PW) {
            src[h] = a.in + (size_t)(pok[h] ? pix : 0) * a.Cin;
        } else {
            int pb = 0, py = 0, px = 0;
            if (pok[h]) {
                pb = pix / (a.Hout * a.Wout);
                const int rem = pix - pb * a.Hout * a.Wout;
                py = rem / a.Wout;
                px = rem - py * a.Wout;
            }
            iy0[h] = py * a.stride - a.pad_t;
            ix0[h] = px * a.stride - a.pad_l;
            src[h] = a.in + (size_t)pb * a.Hin * a.Win * a.Cin;
            tw[h].init(a, BK * kb, pok[h], iy0[h], ix0[h]);
        }
    }
    const f32x4 *wcol = wq + n0 + blk_w * 128 + lane;
    int ks_issue = kb;
    auto issue = [&]() {                                      // 4 DMAs per wave: A rows x 2, B columns x 2
        const int st = (ks_issue - kb) % NS2;
        const unsigned dst = lds_base + st * STAGE2_BYTES + (kq_w * TM2 + blk_w * 128) * 16;
        const bool live = ks_issue < ke;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const char *g;
            if (PW) {
                const int k = BK * ks_issue + 4 * kq_w;
                g = (live && pok[h] && k < a.K) ? reinterpret_cast<const char *>(src[h] + k) : zero;
            } else {
                g = (live && tw[h].ok) ? reinterpret_cast<const char *>(src[h] + tw[h].base + tw[h].cc + 4 * kq_w) : zero;
                if (live) tw[h].advance(a, BK, pok[h], iy0[h], ix0[h]);
            }
#ifndef ZS_EXP_CONV_NO_DMA
            dma16(g, dst + h * 1024);
#endif
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const char *g = (live && nok[h]) ? reinterpret_cast<const char *>(wcol + (size_t)(ks_issue * KQ + kq_w) * a.CoutPad + 64 * h) : zero;
#ifndef ZS_EXP_CONV_NO_DMA
            dma16(g, dst + KQ * TM2 * 16 + h * 1024);
#endif
        }
        ks_issue++;
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int d = 0; d < DEPTH2; d++) issue();

    for (int ks = kb; ks < ke; ks++) {
        wait_vm_then_barrier<4 * (DEPTH2 - 1)>();
        issue();
        const int st = (ks - kb) % NS2;
        u32x4 ah[2], al[2], bh[4], bl[4];
        f32x4 fa[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            fa[i][0] = lds[st][0][half][wm + 32 * i + l32];
            fa[i][1] = lds[st][0][half + 2][wm + 32 * i + l32];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bh[j] = __builtin_bit_cast(u32x4, lds[st][1][half][wn + 32 * j + l32]);
            bl[j] = __builtin_bit_cast(u32x4, lds[st][1][half + 2][wn + 32 * j + l32]);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    fa[i][0][e] = fmaxf(fa[i][0][e], 0.f);
                    fa[i][1][e] = fmaxf(fa[i][1][e], 0.f);
                }
            }
            zs::s16::split8(fa[i][0], fa[i][1], ah[i], al[i]);
        }
#ifdef ZS_EXP_CONV_NO_MFMA
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][j][e] += __builtin_bit_cast(float, ah[i][e] ^ al[i][e] ^ bh[j][e] ^ bl[j][e]);
#else
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) zs::s16::mfma3(acc[i][j], ah[i], al[i], bh[j], bl[j]);
#endif
    }

    auto epilogue = [&](int i, int j, const f32x16 &d) {
        const int n = n0 + wn + 32 * j + l32;
        if (n >= a.Cout) return;
        const float sc = a.scale ? a.scale[n] : 1.0f, sh = a.shift ? a.shift[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * half + (r & 3);
            if (m >= a.M) continue;
            const size_t o = (size_t)m * a.Cout + n;
            float v = d[r] * sc + sh;
            if (a.res1) v += a.res1[o];
            if (a.res2) v += a.res2[o];
            a.out[o] = activate(v, a.act);
        }
    };
    if (kb == 0 && ke == ksteps) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) epilogue(i, j, acc[i][j]);
    } else {   // a share of a tile: see conv_gemm_kernel
        float *parts = a.ws + WS_COUNTER_FLOATS;
        int *counters = reinterpret_cast<int *>(a.ws);
        float *mine = parts + ((size_t)wg * 2 + (it == it_begin ? 0 : 1)) * (TM2 * TN2) + tid;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    __hip_atomic_store(&mine[((i * 4 + j) * 16 + r) * 512], acc[i][j][r], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::: "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int w_first = (tile * ksteps) / a.sk_per, w_last = ((tile + 1) * ksteps - 1) / a.sk_per;
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(&counters[tile], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            sk_last = old == w_last - w_first;
            if (sk_last) __hip_atomic_store(&counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (sk_last) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    f32x16 d;
#pragma unroll
                    for (int r = 0; r < 16; r++) d[r] = 0.f;
#pragma unroll 1
                    for (int w = w_first; w <= w_last; w++) {
                        const int w_tile0 = (w * a.sk_per) / ksteps;
                        const float *p = parts + ((size_t)w * 2 + (w_tile0 == tile ? 0 : 1)) * (TM2 * TN2) + tid;
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            d[r] += __hip_atomic_load(&p[((i * 4 + j) * 16 + r) * 512], __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
                    }
                    epilogue(i, j, d);
                    asm volatile("" ::: "memory");
                }
        }
        __syncthreads();
    }
    it += ke - kb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#include "nn_conv_patch.h"
#include "nn_conv_pp256.h"

// ---- small-problem variant: many small tiles, K split across the four waves ----
// The 128x128 tiling needs >= ~256 tiles to fill 256 CUs; a 14x14 feature map or a 197-token
// matrix gives a handful.  Here a workgroup owns 32 pixels x 64 couts and its four waves each
// take a contiguous quarter of K straight from global memory into MFMA operand registers (the
// packed weight layout makes the B fragment one coalesced float4 per lane; the A fragment is one
// float4 per lane = 4 channels of one tap), so a layer launches (M/32)*(Cout/64) workgroups and
// a wave's dependent MFMA chain is K/4 long instead of K.  Partials are summed through LDS in a
// fixed order (deterministic), then the same fused epilogue.
#ifndef ZS_SMALL_SU
#define ZS_SMALL_SU 4
#endif
constexpr int SM = 32, SU = ZS_SMALL_SU;       // tile rows, and t-steps (8 k each) per prefetch chunk
// NJ = 32-column MFMA tiles per wave: the tile is 32 x 32*NJ (NJ = 1 when even 32x64 tiles leave CUs idle)

// F16: two consecutive t-steps give a lane the eight k values of one K = 16 MFMA operand (A and B in
// the same order), split into hi / lo halves in registers.
// XF: fused input normalisation (zs_conv_fuse.in_mode): 1 = GroupNorm + ReLU of the input from per-tile group sums,
// 2 = LayerNorm of the input rows from per-tile row sums (gamma / beta live in the weights).  The statistics come from
// the producing launch's epilogue (out_mode): no normalisation launch, no extra pass over the tensor.  (A third mode -
// relu(GN(in) + residual) formed on load and written back by column tile 0 - was built and measured: every column
// tile re-reads the residual, 21.7 vs 9.8 us for the 1,024 -> 256 layer at 14 x 14; zs_group_norm_apply_stats does
// that step in one pass instead.)
constexpr int XF_MAXC = 1024;     // channels of a normalised input
template <int NJ, bool PW, int MODE = 0, bool PLAIN = false, bool F16 = false, int XF = 0>
__global__ __launch_bounds__(256) void conv_gemm_small_kernel(ConvArgs a) {
    constexpr bool TM = MODE == 2;            // tap-major walk, Cin % 8 == 0
    constexpr int SN = 32 * NJ;
    __shared__ float part[4][SM][SN + 1];
    __shared__ __attribute__((aligned(16))) float xf_tab[2][XF == 0 ? 4 : (XF == 2 ? SM : XF_MAXC)];
    __shared__ float xf_red[XF == 1 ? 8 : 1][32][2], xf_stat[32][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * SM, n0 = blockIdx.y * SN;
    const int pix = m0 + l32;
    const bool pix_ok = pix < a.M;
    int pb = 0, py = 0, px = 0;
    if (pix_ok) {
        pb = pix / (a.Hout * a.Wout);
        const int rem = pix - pb * a.Hout * a.Wout;
        py = rem / a.Wout;
        px = rem - py * a.Wout;
    }
    const int iy0 = py * a.stride - a.pad_t, ix0 = px * a.stride - a.pad_l;
    const float *in_b = a.in + (size_t)pb * a.Hin * a.Win * a.Cin;
    const float relu_floor = a.in_relu ? 0.f : -INFINITY;
    auto load_a = [&](const TapIter &it) -> AQuad {
        const int vy = iy0 + it.ky, vx = ix0 + it.kx, sh = a.dil - 1;      // dil 1 -> 0, dil 2 -> 1
        const int iy = vy >> sh, ix = vx >> sh;
        AQuad q;
        q.ok = pix_ok && it.ky < a.kh && vy >= 0 && iy < a.Hin && vx >= 0 && ix < a.Win && ((vy | vx) & sh) == 0;
        const size_t off = q.ok ? ((size_t)iy * a.Win + ix) * a.Cin + it.c : 0;
        q.v = *reinterpret_cast<const f32x4 *>(in_b + off);
        return q;
    };
    const f32x4 *wq = reinterpret_cast<const f32x4 *>(a.w);

    const int T = (a.K + BK - 1) / BK * 2;                 // t-steps of 8 k (= 2 weight quads)
    // this workgroup's share of K (all of it unless split across workgroups), then a quarter per wave
    // (ranges start at even t: a pair of t-steps is one K = 16 operand, and pre-split weights are stored per pair)
    const int tsplit = ((T + a.splits - 1) / a.splits + 1) & ~1, ts0 = blockIdx.z * tsplit, ts1 = min(T, ts0 + tsplit);
    const int per = ((max(ts1 - ts0, 0) + 3) / 4 + 1) & ~1, t_begin = ts0 + wave * per, t_end = min(ts1, t_begin + per);
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    AQuad fa[2][SU];
    f32x4 fb[2][SU][NJ];
    TapIter it;                                   // this lane's k = 8 t + 4 half, fetched in t order
    it.init(8 * t_begin + 4 * half, a.Cin, a.kw);
    const float *in_pix = a.in + (size_t)(pix_ok ? pix : 0) * a.Cin;       // PW: address = pixel * Cin + k
    int ka = 8 * t_begin + 4 * half;
    TapWalk tw;                                                             // TM: the wave's current 8-chunk
    if (TM) tw.init(a, 8 * t_begin, pix_ok, iy0, ix0);
    const int adv_tap = 8 / a.Cin, adv_c = 8 % a.Cin;
    // weight quads of t-steps past t_end exist (zero padding up to K16) or are clamped to the last
    // one; their A quads are flagged invalid, so the loop needs no tail branch
    const int kq_last = (a.K + BK - 1) / BK * KQ - 1;
    const f32x4 *wlane = wq + n0 + l32;
    auto fetch = [&](int buf, int t0) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int t = t0 + u;
            if (PW) {
                fa[buf][u].ok = pix_ok && ka < a.K && t < t_end;
                fa[buf][u].v = *reinterpret_cast<const f32x4 *>(in_pix + (fa[buf][u].ok ? ka : 0));
                ka += 8;
            } else if (TM) {
                fa[buf][u].ok = tw.ok && t < t_end;
                fa[buf][u].v = *reinterpret_cast<const f32x4 *>(in_b + tw.base + (tw.ok ? tw.cc + 4 * half : 0));
                tw.advance(a, 8, pix_ok, iy0, ix0);
            } else {
                fa[buf][u] = load_a(it);
                fa[buf][u].ok = fa[buf][u].ok && t < t_end;
                it.advance(adv_tap, adv_c, a.Cin, a.kw);
            }
            const int kq = min(2 * t + half, kq_last);
#pragma unroll
            for (int j = 0; j < NJ; j++) fb[buf][u][j] = wlane[(size_t)kq * a.CoutPad + 32 * j];
        }
    };
    float row_s = 1.0f, row_t = 0.0f;              // XF 2: this lane's row
    // XF: channel of the t-step being consumed (uniform; a t-step of 8 lies inside one tap: Cin % 8 == 0)
    int xc = XF == 1 ? (8 * t_begin) % a.Cin : 0;
    auto a_quad = [&](int buf, int u) -> f32x4 {
        f32x4 av;
        if (XF == 1) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(&xf_tab[0][xc + 4 * half]);
            const f32x4 sh = *reinterpret_cast<const f32x4 *>(&xf_tab[1][xc + 4 * half]);
#pragma unroll
            for (int e = 0; e < 4; e++) av[e] = fa[buf][u].ok ? fmaxf(fa[buf][u].v[e] * sc[e] + sh[e], 0.f) : 0.f;
            xc += 8;
            if (xc >= a.Cin) xc -= a.Cin;
        } else if (XF == 2) {
#pragma unroll
            for (int e = 0; e < 4; e++) av[e] = fa[buf][u].ok ? fa[buf][u].v[e] * row_s + row_t : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                av[e] = (PW || PLAIN) ? (fa[buf][u].ok ? fa[buf][u].v[e] : 0.f)
                                      : (fa[buf][u].ok ? fmaxf(fa[buf][u].v[e], relu_floor) * a.in_scale + a.in_shift : 0.f);
        }
        return av;
    };
    auto consume = [&](int buf) {
        if (F16) {
            static_assert(SU % 2 == 0, "pairs of t-steps");
#pragma unroll
            for (int u = 0; u < SU; u += 2) {
                u32x4 ah, al;
                const f32x4 q0 = a_quad(buf, u);           // in t order: the XF walker advances per call
                const f32x4 q1 = a_quad(buf, u + 1);
                zs::s16::split8(q0, q1, ah, al);
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    u32x4 bh, bl;
                    if (a.w_split) {
                        bh = __builtin_bit_cast(u32x4, fb[buf][u][j]);
                        bl = __builtin_bit_cast(u32x4, fb[buf][u + 1][j]);
                    } else {
                        zs::s16::split8(fb[buf][u][j], fb[buf][u + 1][j], bh, bl);
                    }
                    zs::s16::mfma3(acc[j], ah, al, bh, bl);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const f32x4 av = a_quad(buf, u);
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], fb[buf][u][j][s], acc[j], 0, 0, 0);
        }
    };
    if (t_begin < t_end) fetch(0, t_begin);       // the first operands travel while the statistics are turned into tables
    if (XF == 1) {
        // group statistics of this workgroup's sample: the producer wrote (sum, sum of squares) per 32-row tile and group;
        // summed here in a fixed order (8 strided slices in float, the slices and the moments in double: the same value in
        // every workgroup and on every run), then one (scale, shift) per channel
        const int sample = a.fz_tiles_per_sample > 0 ? m0 / (a.Hout * a.Wout) : 0;
        const int gshift = a.fz.in_gshift;                          // channels per group = 1 << gshift
        const double cnt = (double)a.Hin * a.Win * (double)(1 << gshift);
        const int tiles = a.fz.in_tiles, g = tid & 31, slice = tid >> 5;
        float gam[XF_MAXC / 256], bet[XF_MAXC / 256];             // this thread's channels' affine: requested before the sums are
#pragma unroll                                                     // waited for (one memory round trip instead of two)
        for (int k = 0; k < XF_MAXC / 256; k++) {
            const int c = tid + 256 * k;
            gam[k] = c < a.Cin ? a.fz.in_gamma[c] : 0.f;
            bet[k] = c < a.Cin ? a.fz.in_beta[c] : 0.f;
        }
        float S = 0.f, Q = 0.f;
        for (int tl = slice; tl < tiles; tl += 8) {
            const float2 e = *reinterpret_cast<const float2 *>(a.fz.in_stats + ((size_t)(sample * tiles + tl) * 32 + g) * 2);
            S += e.x;
            Q += e.y;
        }
        xf_red[slice][g][0] = S;
        xf_red[slice][g][1] = Q;
        __syncthreads();
        if (tid < 32) {
            double s2 = 0.0, q2 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) { s2 += (double)xf_red[k][tid][0]; q2 += (double)xf_red[k][tid][1]; }
            const double mean = s2 / cnt, var = fmax(q2 / cnt - mean * mean, 0.0);
            xf_stat[tid][0] = (float)mean;
            xf_stat[tid][1] = (float)(1.0 / sqrt(var + (double)a.fz.in_eps));
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < XF_MAXC / 256; k++) {
            const int c = tid + 256 * k;
            if (c < a.Cin) {
                const int g2 = c >> gshift;
                const float sc = gam[k] * xf_stat[g2][1];
                xf_tab[0][c] = sc;
                xf_tab[1][c] = bet[k] - xf_stat[g2][0] * sc;
            }
        }
        __syncthreads();
    }
    if (XF == 2) {
        // row statistics: the producer wrote (sum, M2 about its own mean) per row and column tile.  Eight lanes per row
        // split the tiles; mean from the sums, then M2 = sum of (M2_t + n_t (mean_t - mean)^2): two xor-shuffle trees
        const int row = tid >> 3, sub = tid & 7, m = m0 + row, tiles = a.fz.in_tiles;
        const float nb = (float)a.Cin / (float)tiles;
        float sums[4], m2s[4];                                      // up to 32 column tiles
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int tl = sub + 8 * k;
            float2 e = {0.f, 0.f};
            if (m < a.M && tl < tiles) e = *reinterpret_cast<const float2 *>(a.fz.in_stats + ((size_t)m * tiles + tl) * 2);
            sums[k] = e.x;
            m2s[k] = e.y;
            S += e.x;
        }
        S += __shfl_xor(S, 1, 64); S += __shfl_xor(S, 2, 64); S += __shfl_xor(S, 4, 64);
        const float mean = S / (float)a.Cin;
        float M2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float d = sums[k] / nb - mean;
            if (sub + 8 * k < tiles) M2 += m2s[k] + nb * d * d;
        }
        M2 += __shfl_xor(M2, 1, 64); M2 += __shfl_xor(M2, 2, 64); M2 += __shfl_xor(M2, 4, 64);
        if (sub == 0) {
            const float rstd = 1.0f / sqrtf(M2 / (float)a.Cin + a.fz.in_eps);
            xf_tab[0][row] = m < a.M ? rstd : 0.f;
            xf_tab[1][row] = m < a.M ? -mean * rstd : 0.f;
        }
        __syncthreads();
    }
    if (XF == 2) {
        row_s = xf_tab[0][l32];
        row_t = xf_tab[1][l32];
    }
    if (t_begin < t_end) {
        for (int t0 = t_begin; t0 < t_end; t0 += 2 * SU) {
            fetch(1, t0 + SU);
            consume(0);
            fetch(0, t0 + 2 * SU);
            consume(1);
        }
    }
    // D[row = 8*(r/4) + 4*half + r%4][col = l32] -> LDS, then a fixed-order 4-way sum
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) part[wave][8 * (r >> 2) + 4 * half + (r & 3)][32 * j + l32] = acc[j][r];
    __syncthreads();
    if ((a.Cout & 3) == 0) {      // four consecutive channels per thread: 16-byte residual loads and stores
        // out_mode: statistics of the values this launch stores, for the consumer's fused normalisation.  1: (sum, sum of
        // squares) per (32-row tile, group of Cout / groups channels); 2: (sum, M2 about the tile-row mean) per (row,
        // column tile).  A thread's quads of all passes share their column quad, hence their group(s).
        const int om = a.fz.out_mode, gw = om == 1 ? a.Cout / a.fz.out_groups : 4;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
        constexpr int QPR = SN / 4, PASSES = SM * QPR / 256;       // quads per row; NJ = 1: one pass, NJ = 2: two
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int e = tid + 256 * ps;
            const int row = e / QPR, col = 4 * (e % QPR), m = m0 + row, n = n0 + col;
            const bool valid = m < a.M && n < a.Cout;
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; c++)
                v[c] = (part[0][row][col + c] + part[1][row][col + c]) + (part[2][row][col + c] + part[3][row][col + c]);
            const size_t o = (size_t)(valid ? m : 0) * a.Cout + (valid ? n : 0);
            if (a.splits > 1) {       // raw partial; the epilogue runs in conv_splitk_reduce_kernel
                if (valid) *reinterpret_cast<f32x4 *>(a.ws + WS_COUNTER_FLOATS + (size_t)blockIdx.z * a.M * a.Cout + o) = v;
                continue;
            }
            if (valid) {
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = v[c] * (a.scale ? a.scale[n + c] : 1.0f) + (a.shift ? a.shift[n + c] : 0.0f);
                if (a.res1) v += *reinterpret_cast<const f32x4 *>(a.res1 + o);
                if (a.res2) v += *reinterpret_cast<const f32x4 *>(a.res2 + o);
                activate4(v, a.act);
                *reinterpret_cast<f32x4 *>(a.out + o) = v;
            } else {
                v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (om == 1) {
                if (gw >= 4) {
                    s0 += (v[0] + v[1]) + (v[2] + v[3]);
                    q0 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                } else {               // two channels per group: two groups per quad
                    s0 += v[0] + v[1]; q0 += v[0] * v[0] + v[1] * v[1];
                    s1 += v[2] + v[3]; q1 += v[2] * v[2] + v[3] * v[3];
                }
            } else if (om == 2) {      // one pass per row here: rows of later passes are other rows
                float rs = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
                for (int sh = 1; sh < QPR; sh <<= 1) rs += __shfl_xor(rs, sh, 64);
                const float mean = rs * (1.0f / SN);
                float d2 = 0.f;
#pragma unroll
                for (int c = 0; c < 4; c++) d2 += (v[c] - mean) * (v[c] - mean);
#pragma unroll
                for (int sh = 1; sh < QPR; sh <<= 1) d2 += __shfl_xor(d2, sh, 64);
                if (valid && (e % QPR) == 0) {
                    float *dst = a.fz.out_stats + ((size_t)m * gridDim.y + blockIdx.y) * 2;
                    dst[0] = rs;
                    dst[1] = d2;
                }
            }
        }
        if (om == 1 && a.splits == 1) {
            // lanes of a wave: low bits = column quad (QPR of them), high bits = row.  Sum over the rows of the wave and
            // over the quads of a group by xor shuffles (a fixed tree), then over the four waves through LDS in wave order.
            const int qpg = gw >= 4 ? gw / 4 : 1;                   // quads per group (1, 2, 4, 8)
#pragma unroll
            for (int sh = QPR; sh < 64; sh <<= 1) {
                s0 += __shfl_xor(s0, sh, 64); q0 += __shfl_xor(q0, sh, 64);
                s1 += __shfl_xor(s1, sh, 64); q1 += __shfl_xor(q1, sh, 64);
            }
            for (int sh = 1; sh < qpg; sh <<= 1) { s0 += __shfl_xor(s0, sh, 64); q0 += __shfl_xor(q0, sh, 64); }
            __syncthreads();                                        // the partial tiles have been read: reuse part[0]
            float *wsum = &part[0][0][0];                           // [wave][group in tile][2 or 4]
            const int gpt = gw >= 4 ? QPR / qpg : 2 * QPR;          // groups per tile
            if (lane < QPR && (lane % qpg) == 0) {
                if (gw >= 4) {
                    wsum[(wave * gpt + lane / qpg) * 2] = s0;
                    wsum[(wave * gpt + lane / qpg) * 2 + 1] = q0;
                } else {
                    wsum[(wave * gpt + 2 * lane) * 2] = s0;     wsum[(wave * gpt + 2 * lane) * 2 + 1] = q0;
                    wsum[(wave * gpt + 2 * lane + 1) * 2] = s1; wsum[(wave * gpt + 2 * lane + 1) * 2 + 1] = q1;
                }
            }
            __syncthreads();
            if (tid < gpt) {
                const int g = n0 / gw + tid;
                if (g < a.fz.out_groups) {
                    float ts = 0.f, tq = 0.f;
                    for (int w = 0; w < 4; w++) { ts += wsum[(w * gpt + tid) * 2]; tq += wsum[(w * gpt + tid) * 2 + 1]; }
                    float *dst = a.fz.out_stats + ((size_t)blockIdx.x * a.fz.out_groups + g) * 2;
                    dst[0] = ts;
                    dst[1] = tq;
                }
            }
        }
        return;
    }
    for (int e = tid; e < SM * SN; e += 256) {
        const int row = e / SN, col = e % SN, m = m0 + row, n = n0 + col;
        if (m >= a.M || n >= a.Cout) continue;
        float v = (part[0][row][col] + part[1][row][col]) + (part[2][row][col] + part[3][row][col]);
        const size_t o = (size_t)m * a.Cout + n;
        if (a.splits > 1) {       // raw partial; the epilogue runs in conv_splitk_reduce_kernel
            a.ws[WS_COUNTER_FLOATS + (size_t)blockIdx.z * a.M * a.Cout + o] = v;
            continue;
        }
        v = v * (a.scale ? a.scale[n] : 1.0f) + (a.shift ? a.shift[n] : 0.0f);
        if (a.res1) v += a.res1[o];
        if (a.res2) v += a.res2[o];
        a.out[o] = activate(v, a.act);
    }
}

// fp32 packed weights [K16/4][CoutPad][4] -> the same shape with, per K = 16 step s and lane half q, the hi halves of
// the eight k values {4q..4q+3, 4q+8..4q+11} in quad 4s+q and their lo halves in quad 4s+q+2 (what the F16 kernels
// otherwise compute from the fp32 quads on every use)
__global__ __launch_bounds__(256) void presplit_weight_kernel(const f32x4 *__restrict__ w, f32x4 *__restrict__ out,
                                                              size_t pairs, int CoutPad) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;         // (s, q, n)
    if (i >= pairs) return;
    const size_t n = i % CoutPad, q = (i / CoutPad) & 1, s = i / CoutPad / 2;
    u32x4 hi, lo;
    zs::s16::split8(w[(4 * s + q) * CoutPad + n], w[(4 * s + q + 2) * CoutPad + n], hi, lo);
    out[(4 * s + q) * CoutPad + n] = __builtin_bit_cast(f32x4, hi);
    out[(4 * s + q + 2) * CoutPad + n] = __builtin_bit_cast(f32x4, lo);
}

// the same for a table of operands in ONE launch (the training step splits every layer's operand after each optimiser step):
// pair g of the concatenated pair space belongs to entry e with prefix[e] <= g < prefix[e + 1]
constexpr int PRESPLIT_PER_THREAD = 16;
__global__ __launch_bounds__(256) void presplit_weight_multi_kernel(const f32x4 *const *__restrict__ src, f32x4 *const *__restrict__ dst,
                                                                    const int *__restrict__ cout_pad,
                                                                    const unsigned long long *__restrict__ prefix, int n) {
    const unsigned long long total = prefix[n];
    unsigned long long g = (unsigned long long)blockIdx.x * (256 * PRESPLIT_PER_THREAD) + threadIdx.x;
    int e = 0;
#pragma unroll 1
    for (int k = 0; k < PRESPLIT_PER_THREAD; k++, g += 256) {
        if (g >= total) return;
        if (k == 0 || g >= prefix[e + 1]) {           // binary search (the first pair; afterwards only when the entry changes)
            int lo = 0, hi = n - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (prefix[mid] <= g) lo = mid; else hi = mid - 1;
            }
            e = lo;
        }
        const size_t i = (size_t)(g - prefix[e]), cp = (size_t)cout_pad[e];
        const size_t col = i % cp, q = (i / cp) & 1, st = i / cp / 2;
        const f32x4 *w = src[e];
        f32x4 *out = dst[e];
        u32x4 hi4, lo4;
        zs::s16::split8(w[(4 * st + q) * cp + col], w[(4 * st + q + 2) * cp + col], hi4, lo4);
        out[(4 * st + q) * cp + col] = __builtin_bit_cast(f32x4, hi4);
        out[(4 * st + q + 2) * cp + col] = __builtin_bit_cast(f32x4, lo4);
    }
}

// out = epilogue(sum over the splits, in split order: deterministic)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvArgs a) {
    const size_t total = (size_t)a.M * a.Cout;
    if ((a.Cout & 3) == 0 && total < ((size_t)1 << 32)) {
        // four channels per lane, 32-bit indices (round 5: the scalar loop below pays a 64-bit modulo per value; 30 launches of it
        // per batch-28 forward, 34 per batch-1 forward).  Per element the same sums in the same order: the same values
        const unsigned quads = (unsigned)(total >> 2), cq = (unsigned)a.Cout >> 2;
        const f32x4 *ws4 = reinterpret_cast<const f32x4 *>(a.ws + WS_COUNTER_FLOATS);
        for (unsigned o = blockIdx.x * 256 + threadIdx.x; o < quads; o += gridDim.x * 256) {
            const unsigned n = (o % cq) << 2;
            f32x4 v = ws4[o];
            for (int sp = 1; sp < a.splits; sp++) v += ws4[(size_t)sp * quads + o];
            const f32x4 sc = a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
            const f32x4 sh = a.shift ? *reinterpret_cast<const f32x4 *>(a.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            v = v * sc + sh;                                  // (one contraction per component, as the scalar form)
            if (a.res1) v += reinterpret_cast<const f32x4 *>(a.res1)[o];
            if (a.res2) v += reinterpret_cast<const f32x4 *>(a.res2)[o];
            activate4(v, a.act);
            reinterpret_cast<f32x4 *>(a.out)[o] = v;
        }
        return;
    }
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int n = (int)(o % a.Cout);
        const float *ws = a.ws + WS_COUNTER_FLOATS;
        float v = ws[o];
        for (int sp = 1; sp < a.splits; sp++) v += ws[(size_t)sp * total + o];
        v = v * (a.scale ? a.scale[n] : 1.0f) + (a.shift ? a.shift[n] : 0.0f);
        if (a.res1) v += a.res1[o];
        if (a.res2) v += a.res2[o];
        a.out[o] = activate(v, a.act);
    }
}

// The same for a layer that also writes statistics of what it stores (zs_conv_fuse.out_mode 1 / 2): one workgroup per
// (32-row tile, 32 NJ columns) of the output - the tile shape, thread <-> quad map and reduction trees of
// conv_gemm_small_kernel's vector epilogue, so the statistics land in the layout (and the order of summation within a tile)
// the unsplit launch would have produced; the value of each element is the sum of the ranges' partials in range order.
template <int NJ>
__global__ __launch_bounds__(256) void conv_splitk_reduce_stats_kernel(ConvArgs a) {
    constexpr int SN = 32 * NJ, QPR = SN / 4, PASSES = SM * QPR / 256;
    __shared__ float wsum[4 * 2 * QPR * 2 + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * SM, n0 = blockIdx.y * SN;
    const size_t total = (size_t)a.M * a.Cout;
    const float *ws = a.ws + WS_COUNTER_FLOATS;
    const int om = a.fz.out_mode, gw = om == 1 ? a.Cout / a.fz.out_groups : 4;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int e = tid + 256 * ps;
        const int row = e / QPR, col = 4 * (e % QPR), m = m0 + row, n = n0 + col;
        const bool valid = m < a.M && n < a.Cout;
        const size_t o = (size_t)(valid ? m : 0) * a.Cout + (valid ? n : 0);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            v = *reinterpret_cast<const f32x4 *>(ws + o);
            for (int sp = 1; sp < a.splits; sp++) v += *reinterpret_cast<const f32x4 *>(ws + (size_t)sp * total + o);
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = v[c] * (a.scale ? a.scale[n + c] : 1.0f) + (a.shift ? a.shift[n + c] : 0.0f);
            if (a.res1) v += *reinterpret_cast<const f32x4 *>(a.res1 + o);
            if (a.res2) v += *reinterpret_cast<const f32x4 *>(a.res2 + o);
            activate4(v, a.act);
            *reinterpret_cast<f32x4 *>(a.out + o) = v;
        }
        if (om == 1) {
            if (gw >= 4) {
                s0 += (v[0] + v[1]) + (v[2] + v[3]);
                q0 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            } else {
                s0 += v[0] + v[1]; q0 += v[0] * v[0] + v[1] * v[1];
                s1 += v[2] + v[3]; q1 += v[2] * v[2] + v[3] * v[3];
            }
        } else if (om == 2) {
            float rs = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
            for (int sh = 1; sh < QPR; sh <<= 1) rs += __shfl_xor(rs, sh, 64);
            const float mean = rs * (1.0f / SN);
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) d2 += (v[c] - mean) * (v[c] - mean);
#pragma unroll
            for (int sh = 1; sh < QPR; sh <<= 1) d2 += __shfl_xor(d2, sh, 64);
            if (valid && (e % QPR) == 0) {
                float *dst = a.fz.out_stats + ((size_t)m * gridDim.y + blockIdx.y) * 2;
                dst[0] = rs;
                dst[1] = d2;
            }
        }
    }
    if (om == 1) {
        const int qpg = gw >= 4 ? gw / 4 : 1;
#pragma unroll
        for (int sh = QPR; sh < 64; sh <<= 1) {
            s0 += __shfl_xor(s0, sh, 64); q0 += __shfl_xor(q0, sh, 64);
            s1 += __shfl_xor(s1, sh, 64); q1 += __shfl_xor(q1, sh, 64);
        }
        for (int sh = 1; sh < qpg; sh <<= 1) { s0 += __shfl_xor(s0, sh, 64); q0 += __shfl_xor(q0, sh, 64); }
        const int gpt = gw >= 4 ? QPR / qpg : 2 * QPR;
        if (lane < QPR && (lane % qpg) == 0) {
            if (gw >= 4) {
                wsum[(wave * gpt + lane / qpg) * 2] = s0;
                wsum[(wave * gpt + lane / qpg) * 2 + 1] = q0;
            } else {
                wsum[(wave * gpt + 2 * lane) * 2] = s0;     wsum[(wave * gpt + 2 * lane) * 2 + 1] = q0;
                wsum[(wave * gpt + 2 * lane + 1) * 2] = s1; wsum[(wave * gpt + 2 * lane + 1) * 2 + 1] = q1;
            }
        }
        __syncthreads();
        if (tid < gpt) {
            const int g = n0 / gw + tid;
            if (g < a.fz.out_groups) {
                float ts = 0.f, tq = 0.f;
                for (int w = 0; w < 4; w++) { ts += wsum[(w * gpt + tid) * 2]; tq += wsum[(w * gpt + tid) * 2 + 1]; }
                float *dst = a.fz.out_stats + ((size_t)blockIdx.x * a.fz.out_groups + g) * 2;
                dst[0] = ts;
                dst[1] = tq;
            }
        }
    }
}

}  // namespace

extern "C" size_t zs_conv2d_splitk_workspace_bytes(void) {
    static_assert(WS_PARTS_BYTES == (size_t)SK_MAX_WGS * 2 * BM * BN * 4 && WS_PARTS_BYTES >= WS_SPLITK_BYTES, "workspace regions");
    return WS_COUNTER_FLOATS * 4 + WS_PARTS_BYTES;
}

extern "C" int zs_conv2d_nhwc(const float *in, const float *packed_w, const float *scale, const float *shift,
                              const float *res1, const float *res2, float *out, int batch, int Hin, int Win,
                              int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t,
                              int pad_l, int flags, float in_scale, float in_shift, int act, void *stream) {
    return zs_conv2d_nhwc_ws(in, packed_w, scale, shift, res1, res2, out, batch, Hin, Win, Cin, Hout, Wout, Cout, kh, kw,
                             stride, pad_t, pad_l, flags, in_scale, in_shift, act, nullptr, stream);
}

static int conv2d_impl(const float *in, const float *packed_w, const float *scale, const float *shift,
                       const float *res1, const float *res2, float *out, int batch, int Hin, int Win,
                       int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t,
                       int pad_l, int flags, float in_scale, float in_shift, int act, const zs_conv_fuse *fuse,
                       void *workspace, void *stream);

extern "C" int zs_conv2d_nhwc_ws(const float *in, const float *packed_w, const float *scale, const float *shift,
                                 const float *res1, const float *res2, float *out, int batch, int Hin, int Win,
                                 int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t,
                                 int pad_l, int flags, float in_scale, float in_shift, int act, void *workspace,
                                 void *stream) {
    return conv2d_impl(in, packed_w, scale, shift, res1, res2, out, batch, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, stride,
                       pad_t, pad_l, flags, in_scale, in_shift, act, nullptr, workspace, stream);
}

static bool small_is_narrow(long long M, int Cout) {
    static const long long narrow_below = getenv("ZS_CONV_NARROW_BELOW") ? atoll(getenv("ZS_CONV_NARROW_BELOW")) : 384;
    return ((M + SM - 1) / SM) * ((Cout + 63) / 64) < narrow_below;      // 32x32 tiles: twice the workgroups for the smallest problems
}

// out_mode 2 (row statistics per column tile): the small-tile kernel's own tile width; the streaming kernel follows it
static bool stream_enabled() {
    static const bool on = getenv("ZS_CONV_STREAM") == nullptr || atoi(getenv("ZS_CONV_STREAM")) != 0;
    return on;
}
static long long stream_max_rows() {
    static const long long rows = getenv("ZS_STREAM_MAX_ROWS") ? atoll(getenv("ZS_STREAM_MAX_ROWS")) : 1024;
    return rows;
}

// Column-tile width of the row statistics (out_mode 2): the small-tile kernel's preference (the streaming kernel writes them
// too - 32 or 64 columns - but with its K split off it refuses these few-tile layers, and the small kernel's narrow tiles are
// faster on few rows: ViT proj 8.3 vs 9.7 us, fc2 20.4 vs 23.2 us, tools/stream_shapes.py)
extern "C" int zs_conv2d_fused_cols(int M, int Cout) { return small_is_narrow(M, Cout) || Cout % 64 ? 32 : 64; }

extern "C" int zs_conv2d_nhwc_fused(const float *in, const float *packed_w, const float *scale, const float *shift,
                                    const float *res1, const float *res2, float *out, int batch, int Hin, int Win,
                                    int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t,
                                    int pad_l, int flags, float in_scale, float in_shift, int act, const zs_conv_fuse *fuse,
                                    void *workspace, void *stream) {
    if (!fuse) { zs::set_err("zs_conv2d_nhwc_fused: null fuse descriptor"); return 0; }
    const int need = ZS_CONV_F16X3 | ZS_CONV_W_PRESPLIT;
    const zs_conv_fuse &f = *fuse;
    const long long hw_in = (long long)Hin * Win, hw_out = (long long)Hout * Wout;
    const bool pw = kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && Hin == Hout && Win == Wout;
    const bool gn_in = f.in_mode == 1;
    int gshift = 0;
    while ((32 << gshift) < Cin) gshift++;
    if ((flags & need) != need || (flags & (ZS_CONV_IN_RELU | ZS_CONV_IN_DILATE2 | ZS_CONV_FORCE_LARGE)) ||
        (f.in_mode && (in_scale != 1.0f || in_shift != 0.0f || (Cin & 7))) ||      // (without an input mode: any layer, e.g. the 7 x 7 stem)
        f.in_mode < 0 || f.in_mode > 2 || f.out_mode < 0 || f.out_mode > 2 || (Cout & 3) ||
        (f.in_mode && !f.in_stats) || (f.out_mode && !f.out_stats) ||
        (gn_in && (!f.in_gamma || !f.in_beta || f.in_groups != 32 || (32 << gshift) != Cin || Cin > XF_MAXC || f.in_tiles <= 0 ||
                   (batch > 1 && hw_in % SM) || (batch > 1 && hw_out % SM))) ||
        (f.in_mode == 2 && (!pw || f.in_tiles <= 0 || f.in_tiles > 32 || Cin % f.in_tiles)) ||
        (f.out_mode == 1 && (f.out_groups <= 0 || Cout % f.out_groups || (batch > 1 && hw_out % SM))) ||
        (f.out_mode == 2 && Cout % zs_conv2d_fused_cols((int)(batch * hw_out), Cout))) {
        zs::set_err("zs_conv2d_nhwc_fused: unsupported combination (flags %d in_mode %d out_mode %d Cin %d Cout %d k %dx%d s %d)",
                    flags, f.in_mode, f.out_mode, Cin, Cout, kh, kw, stride);
        return 0;
    }
    if (f.out_mode == 1) {
        const int gw = Cout / f.out_groups;
        if (!(gw == 2 || gw == 4 || gw == 8 || gw == 16 || gw == 32)) {
            zs::set_err("zs_conv2d_nhwc_fused: out_mode 1 takes 2..32 channels per group (Cout %d, groups %d)", Cout, f.out_groups);
            return 0;
        }
    }
    zs_conv_fuse f2 = f;
    f2.in_gshift = gshift;
    return conv2d_impl(in, packed_w, scale, shift, res1, res2, out, batch, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, stride,
                       pad_t, pad_l, (flags | ZS_CONV_FORCE_SMALL) & ~ZS_CONV_SPLIT_SMALL, in_scale, in_shift, act, &f2, workspace, stream);
}

// the streaming kernel's description of a pointwise layer with K16-major operands (shape, fusion and workspace fields; the
// caller adds the tensors)
static void k16_stream_args(zs::stream_gemm::Args &g, const ConvArgs &a, int Cin, int Cout, int flags, const zs_conv_fuse *fuse,
                            void *workspace) {
    memset(&g, 0, sizeof g);
    g.M = a.M; g.K = a.K; g.N = Cout; g.CoutPad = a.CoutPad; g.lda = Cin; g.act = a.act; g.in_relu = 0;
    g.in_stats = fuse && fuse->in_mode == 2 ? fuse->in_stats : nullptr;
    g.in_tiles = fuse ? fuse->in_tiles : 0;
    g.in_eps = fuse ? fuse->in_eps : 0.f;
    g.out_stats = fuse && fuse->out_mode == 2 ? fuse->out_stats : nullptr;
    g.stats_cols = zs_conv2d_fused_cols(a.M, Cout);
    g.parts = workspace ? static_cast<float *>(workspace) + WS_COUNTER_FLOATS : nullptr;
    g.tickets = static_cast<int *>(workspace);
    g.parts_bytes = WS_PARTS_BYTES;
    g.max_tickets = (int)WS_COUNTER_FLOATS;
    g.two_launch_max = 1;
    g.a_k16 = (flags & ZS_CONV_IN_K16) ? 1 : 0;
    g.out_k16 = (flags & ZS_CONV_OUT_K16) ? 1 : 0;
}

// Would zs_conv2d_nhwc[_fused] accept this pointwise layer with the given K16 flags?  (ln_in: fuse.in_mode 2 with `in_tiles` tiles;
// row_stats_out: fuse.out_mode 2; has_res: a residual is added.)  The caller asks before choosing the layout of a tensor.
extern "C" int zs_conv2d_k16_ok(int M, int Cin, int Cout, int flags, int ln_in_tiles, int row_stats_out, int has_res) {
    if (M <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 15) || (Cout & 3) || !stream_enabled() || M > stream_max_rows()) return 0;
    if (!(flags & (ZS_CONV_IN_K16 | ZS_CONV_OUT_K16))) return 0;
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.M = M; a.K = Cin; a.CoutPad = (Cout + BN - 1) / BN * BN; a.act = 0;
    zs_conv_fuse f;
    memset(&f, 0, sizeof f);
    static float dummy[2];
    if (ln_in_tiles > 0) { f.in_mode = 2; f.in_tiles = ln_in_tiles; f.in_stats = dummy; }
    if (row_stats_out) {
        if (Cout % zs_conv2d_fused_cols(M, Cout)) return 0;
        f.out_mode = 2; f.out_stats = dummy;
    }
    zs::stream_gemm::Args g;
    k16_stream_args(g, a, Cin, Cout, flags, &f, dummy);          // (a workspace is always there in the inference engine)
    g.res1 = has_res ? dummy : nullptr;
    int ranges = 1;
    g.ranges = &ranges;
    return zs::stream_gemm::launch(g, nullptr, true) ? 1 : 0;
}

static int conv2d_impl(const float *in, const float *packed_w, const float *scale, const float *shift,
                       const float *res1, const float *res2, float *out, int batch, int Hin, int Win,
                       int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t,
                       int pad_l, int flags, float in_scale, float in_shift, int act, const zs_conv_fuse *fuse,
                       void *workspace, void *stream) {
    if (batch < 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || (Cin & 3) || Hout <= 0 || Wout <= 0 || Cout <= 0 ||
        kh <= 0 || kw <= 0 || stride <= 0 || act < 0 || act > ZS_ACT_RELU_CLAMP1) {
        zs::set_err("zs_conv2d_nhwc: bad geometry (B=%d in %dx%dx%d out %dx%dx%d k %dx%d s %d act %d; Cin must be "
                    "a multiple of 4)", batch, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, stride, act);
        return 0;
    }
    if (batch == 0) return 1;
    if (!in || !packed_w || !out) { zs::set_err("zs_conv2d_nhwc: null pointer"); return 0; }
    const long long M = (long long)batch * Hout * Wout;
    if (M > (1LL << 30)) { zs::set_err("zs_conv2d_nhwc: %lld output pixels", M); return 0; }
    ConvArgs a;
    a.slab_major = 0;
    a.tail_w = a.tail_b = nullptr;
    a.tail_act = 0;
    if (fuse) a.fz = *fuse; else memset(&a.fz, 0, sizeof a.fz);
    a.fz_tiles_per_sample = fuse && batch > 1 ? 1 : 0;
    a.in = in; a.w = packed_w; a.scale = scale; a.shift = shift; a.res1 = res1; a.res2 = res2; a.out = out;
    a.B = batch; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
    a.CoutPad = (Cout + BN - 1) / BN * BN;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
    a.K = kh * kw * Cin; a.M = (int)M;
    a.in_relu = (flags & ZS_CONV_IN_RELU) ? 1 : 0;
    a.act = act; a.in_scale = in_scale; a.in_shift = in_shift;
    a.dil = (flags & ZS_CONV_IN_DILATE2) ? 2 : 1;
    a.ws = static_cast<float *>(workspace);
    a.splits = 1;
    a.sk_per = 0;
    a.w_split = (flags & ZS_CONV_W_PRESPLIT) ? 1 : 0;
    if (a.w_split && !(flags & ZS_CONV_F16X3)) { zs::set_err("zs_conv2d_nhwc: ZS_CONV_W_PRESPLIT needs ZS_CONV_F16X3"); return 0; }
    // fewer than ~3/4 of a wave of 128x128 tiles over the 256 CUs: use the small-tile split-K variant
    const long long big_tiles = ((M + BM - 1) / BM) * (a.CoutPad / BN);
    static const long long big_min = getenv("ZS_CONV_BIG_MIN") ? atoll(getenv("ZS_CONV_BIG_MIN")) : 192;   // (128 is ~1 % faster at batch 28 - 15.3 vs 15.5 ms - but moves the training kernels too: the 300-iteration from-scratch run of tests/test_gpu_trained_weights.py then takes the trajectory on which the untrained depth head dies; kept at 192)
    // With a workspace - the inference engine (nn/ops.py); the training path calls zs_conv2d_nhwc without one and keeps 192 - the
    // large-tile kernels start at 96 tiles (round 5, tools/enc_b28.py on one box: 192 / 128 / 96 / 80 / 64 / 40 tiles = 14.24 / 14.08 /
    // 14.11 / 14.06 / 14.61 / 14.87 ms per batch-28 forward; batch 1 has no layer in that range)
    static const long long big_min_ws = getenv("ZS_CONV_BIG_MIN_WS") ? atoll(getenv("ZS_CONV_BIG_MIN_WS")) : 96;
    const long long big_min_eff = workspace && !getenv("ZS_CONV_BIG_MIN") ? big_min_ws : big_min;
    const bool small = (flags & ZS_CONV_FORCE_SMALL) || (!(flags & ZS_CONV_FORCE_LARGE) && big_tiles < big_min_eff);
    const bool pw = kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && a.dil == 1 && !a.in_relu &&
                    in_scale == 1.0f && in_shift == 0.0f && Hin == Hout && Win == Wout;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool plain = !a.in_relu && in_scale == 1.0f && in_shift == 0.0f;
    static const bool no_tm = getenv("ZS_CONV_NO_TAPMAJOR") != nullptr;       // A/B switch for measurements
    const bool f16 = (flags & ZS_CONV_F16X3) != 0;
#define ZS_LAUNCH1(KERNEL, F, ...)                                                                                  \
    do {                                                                                                            \
        if (pw) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ true, 0, false, F>), grid, dim3(256), 0, st, a);              \
        else if (tm && plain) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, 2, true, F>), grid, dim3(256), 0, st, a); \
        else if (tm) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, 2, false, F>), grid, dim3(256), 0, st, a);        \
        else hipLaunchKernelGGL((KERNEL<__VA_ARGS__ false, 0, false, F>), grid, dim3(256), 0, st, a);                \
    } while (0)
#define ZS_LAUNCH(KERNEL, ...)                            \
    do {                                                  \
        if (f16) ZS_LAUNCH1(KERNEL, true, __VA_ARGS__);   \
        else ZS_LAUNCH1(KERNEL, false, __VA_ARGS__);      \
    } while (0)
#define ZS_LAUNCH_BIG(F, SKF)                                                                                              \
    do {                                                                                                                   \
        if (pw) hipLaunchKernelGGL((conv_gemm_kernel<true, 0, false, F, SKF>), grid, dim3(256), 0, st, a);                  \
        else if (tm && plain) hipLaunchKernelGGL((conv_gemm_kernel<false, 2, true, F, SKF>), grid, dim3(256), 0, st, a);    \
        else if (tm) hipLaunchKernelGGL((conv_gemm_kernel<false, 2, false, F, SKF>), grid, dim3(256), 0, st, a);            \
        else hipLaunchKernelGGL((conv_gemm_kernel<false, 0, false, F, SKF>), grid, dim3(256), 0, st, a);                    \
    } while (0)
    // ---- K16-major activations (ZS_CONV_IN_K16 / ZS_CONV_OUT_K16): only the streaming GEMM kernel reads / writes that layout ----
    if (flags & (ZS_CONV_IN_K16 | ZS_CONV_OUT_K16)) {
        const bool ok = kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && a.dil == 1 && Hin == Hout && Win == Wout &&
                        f16 && a.w_split && in_scale == 1.0f && in_shift == 0.0f && !a.in_relu && stream_enabled() &&
                        (!fuse || ((fuse->in_mode == 0 || fuse->in_mode == 2) && (fuse->out_mode == 0 || fuse->out_mode == 2)));
        zs::stream_gemm::Args g;
        k16_stream_args(g, a, Cin, Cout, flags, fuse, workspace);
        g.a = in; g.w = packed_w; g.scale = scale; g.shift = shift; g.res1 = res1; g.res2 = res2; g.out = out;
        int ranges = 1;
        g.ranges = &ranges;
        if (!ok || !zs::stream_gemm::launch(g, st)) {
            zs::set_err("zs_conv2d_nhwc: the K16-major layout flags need a pointwise split-fp16 layer the streaming kernel takes "
                        "(zs_conv2d_k16_ok; M=%lld Cin=%d Cout=%d)", M, Cin, Cout);
            return 0;
        }
        return zs::check_launch("zs_conv2d_nhwc") ? 1 : 0;
    }
    // 3 x 3 stride-1 pad-1 layers in split-fp16 with pre-split weights: the input-patch kernels (nn_conv_patch.h)
    static const bool no_patch = getenv("ZS_CONV_NO_PATCH") != nullptr;          // A/B switch for measurements
    const bool patch_geom = f16 && a.w_split && !no_patch && kh == 3 && kw == 3 && stride == 1 && a.dil == 1 && pad_t == 1 &&
                            pad_l == 1 && Hin == Hout && Win == Wout && (Cin % BK) == 0 && in_scale == 1.0f && in_shift == 0.0f &&
                            !(flags & (ZS_CONV_FORCE_SMALL | ZS_CONV_FORCE_LARGE));
    if (patch_geom && Cout <= 32 && Hout >= 8 && Wout >= 8) {
        const int tx = (Wout + patch32::PT - 1) / patch32::PT, ty = (Hout + patch32::PT - 1) / patch32::PT;
        const dim3 grid((unsigned)((long long)batch * tx * ty));
        static const int nw = getenv("ZS_CONV_PATCH_WAVES") ? atoi(getenv("ZS_CONV_PATCH_WAVES")) : 8;      // A/B: 4 or 8
        if (nw == 4) {
            if (a.in_relu) hipLaunchKernelGGL((conv3x3_patch32_kernel<true, 4>), grid, dim3(256), 0, st, a, tx, ty);
            else hipLaunchKernelGGL((conv3x3_patch32_kernel<false, 4>), grid, dim3(256), 0, st, a, tx, ty);
        } else {
            if (a.in_relu) hipLaunchKernelGGL((conv3x3_patch32_kernel<true, 8>), grid, dim3(512), 0, st, a, tx, ty);
            else hipLaunchKernelGGL((conv3x3_patch32_kernel<false, 8>), grid, dim3(512), 0, st, a, tx, ty);
        }
        return zs::check_launch("zs_conv2d_nhwc") ? 1 : 0;
    }
    static const long long patch_min = getenv("ZS_CONV_PATCH_MIN") ? atoll(getenv("ZS_CONV_PATCH_MIN")) : 192;
    if (patch_geom && Cout > 32 && Hout >= 8 && Wout >= 8) {
        const int tx = (Wout + patch128::TW - 1) / patch128::TW, ty = (Hout + patch128::TH - 1) / patch128::TH;
        const long long wgs = (long long)batch * tx * ty * (a.CoutPad / BN);
        // few tiles (14 x 14 maps at batch 28: 112; everything at batch 1): split the slabs across ~384 workgroups, partial
        // tiles through the workspace, summed in range order by conv_splitk_reduce_kernel (which applies the epilogue)
        static const long long split_target = getenv("ZS_CONV_PATCH_SPLIT_TARGET") ? atoll(getenv("ZS_CONV_PATCH_SPLIT_TARGET")) : 384;
        static const long long split_min = getenv("ZS_CONV_PATCH_SPLIT_MIN") ? atoll(getenv("ZS_CONV_PATCH_SPLIT_MIN")) : 32;   // (64 until round 4: the 56 x 56 fusion layers at batch 1 - 50 workgroups - took the LDS-DMA kernel + split-K; batch 1 3.00 -> 2.95 ms)
        long long sp = 1;
        if (wgs < patch_min && workspace && wgs >= split_min) {
            sp = (split_target + wgs - 1) / wgs;
            const long long slabs = Cin / BK, cap = (long long)(WS_PARTS_BYTES / 4) / (M * (long long)Cout);
            if (sp > slabs / 2) sp = slabs / 2;            // at least two slabs (18 tap-steps) per range
            if (sp > cap) sp = cap;
        }
        if (wgs >= patch_min || sp > 1) {
            a.splits = (int)(sp > 1 ? sp : 1);
            static const bool xcd_map = getenv("ZS_CONV_PATCH_XCD") ? atoi(getenv("ZS_CONV_PATCH_XCD")) != 0 : true;
            const long long mtiles = (long long)batch * tx * ty, ntl = a.CoutPad / BN;
            dim3 grid((unsigned)mtiles, (unsigned)ntl, (unsigned)a.splits);
            if (xcd_map && ntl > 1) {
                a.sk_per = (int)ntl;
                grid = dim3((unsigned)((mtiles + 7) / 8 * 8 * ntl), 1, (unsigned)a.splits);
            }
            if (a.in_relu) hipLaunchKernelGGL((conv3x3_patch128_kernel<true>), grid, dim3(256), 0, st, a, tx, ty);
            else hipLaunchKernelGGL((conv3x3_patch128_kernel<false>), grid, dim3(256), 0, st, a, tx, ty);
            if (a.splits > 1) {
                const long long total = M * (long long)Cout;
                const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
                hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
            }
            return zs::check_launch("zs_conv2d_nhwc") ? 1 : 0;
        }
    }
    // ---- large pointwise layers: 256 x 256 ping-pong tiles (nn_conv_pp256.h) ----
    {
        static const int pp_on = getenv("ZS_CONV_PP256") ? atoi(getenv("ZS_CONV_PP256")) : 1;
        // (tools/bench_pp256.py, old -> new us: fc1 110 -> 97, qkv 91 -> 73, fc2 142 -> 95, 16,384 x 3,072 x 768 312 -> 236; it loses
        // where a tile's K loop is short against its 256 KiB epilogue and one workgroup per CU cannot overlap the two: proj
        // (66 tiles x K 768) 38 -> 44, K <= 512 everywhere)
        static const long long pp_min_tiles = getenv("ZS_CONV_PP256_MIN_TILES") ? atoll(getenv("ZS_CONV_PP256_MIN_TILES")) : 48;
        static const long long pp_min_k = getenv("ZS_CONV_PP256_MIN_K") ? atoll(getenv("ZS_CONV_PP256_MIN_K")) : 768;
        static const long long pp_long_k = getenv("ZS_CONV_PP256_LONG_K") ? atoll(getenv("ZS_CONV_PP256_LONG_K")) : 2048;
        static const int pp_max_split = getenv("ZS_CONV_PP256_MAX_SPLIT") ? atoi(getenv("ZS_CONV_PP256_MAX_SPLIT")) : 8;
        static const int pp_min_steps = getenv("ZS_CONV_PP256_MIN_STEPS") ? atoi(getenv("ZS_CONV_PP256_MIN_STEPS")) : 4;
        static const int cus = [] { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        const long long T = ((M + pp256::TM - 1) / pp256::TM) * ((a.CoutPad + pp256::TN - 1) / pp256::TN);
        const bool pp_forced = (flags & ZS_CONV_FORCE_TILE256) != 0;
        // ... and from K = 256 where the tiles make four rounds or more (the window stage of the transformer coordinate encoder at
        // batch 28: 356,720 rows - qkv 770 -> 562 us, fc1 1,096 -> 695, fc2 807 -> 560, proj 272 -> 245; at 343 tiles the same K
        // loses: 56 -> 68): over many rounds the XCDs drift apart and one tile's epilogue runs beside another's loop
        static const long long pp_many_k = getenv("ZS_CONV_PP256_MANY_K") ? atoll(getenv("ZS_CONV_PP256_MANY_K")) : 256;
        const bool pp_pays = pp_on && T >= pp_min_tiles &&
                             ((a.K >= pp_min_k && (T >= cus / 2 || a.K >= pp_long_k)) || (a.K >= pp_many_k && T >= 4LL * cus));
        if ((pp_forced || pp_pays) && pw && f16 && a.w_split && !fuse && (Cin % pp256::SK) == 0 && (Cout & 3) == 0 &&
            !(flags & (ZS_CONV_FORCE_SMALL | ZS_CONV_FORCE_LARGE))) {
            const long long steps = a.K / pp256::SK;
            // the tail: tiles beyond the last whole round of the CUs (or all of them, when they fill less than half of one)
            long long tail = T <= cus / 2 ? T : (T % cus < cus / 2 ? T % cus : 0), sp = 1;
            if (T < cus && T > cus / 2) tail = 0;
            if (tail && workspace) {
                sp = cus / tail;
                if (sp > pp_max_split) sp = pp_max_split;
                if (sp > steps / pp_min_steps) sp = steps / pp_min_steps;
                const long long cap = (long long)(WS_PARTS_BYTES / 4) / ((long long)pp256::TM * pp256::TN);
                if (tail * sp > cap) sp = cap / tail;
            }
            if (sp < 2) { tail = 0; sp = 1; }
            a.sk_per = (int)tail;
            a.splits = (int)sp;
            const dim3 grid((unsigned)(T - tail + tail * sp));
            hipLaunchKernelGGL(pp256::conv_gemm_pp256_kernel, grid, dim3(512), 0, st, a);
            if (tail) hipLaunchKernelGGL(pp256::pp256_tail_kernel, dim3(pp256::MI * pp256::NJ * 4, (unsigned)tail), dim3(512), 0, st, a);
            return zs::check_launch("zs_conv2d_nhwc") ? 1 : 0;
        }
    }
    // few 128 x 128 tiles but a long contraction (3 x 3 768 -> 768 on 7 x 7 maps at batch 28: 66 tiles, 432 k-steps): the
    // LDS-DMA kernel with the k-steps split across blockIdx.z, ~384 workgroups, >= 24 steps each
    // (tools/conv_shapes.py + tools/bench_encoder.py, same box: batch 28 18.23 -> 17.89 ms - the 768 -> 768 head layers 156 ->
    // 100 us; below ~40 tiles - the ViT fc2 at batch 1: 12 tiles - the reduce launch costs more than the ranges save)
    static const long long dsplit_min_steps = getenv("ZS_CONV_DMA_SPLIT_MIN_STEPS") ? atoll(getenv("ZS_CONV_DMA_SPLIT_MIN_STEPS")) : 96;
    static const long long dsplit_min_tiles = getenv("ZS_CONV_DMA_SPLIT_MIN_TILES") ? atoll(getenv("ZS_CONV_DMA_SPLIT_MIN_TILES")) : 40;
    {
        const bool pw_ = kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && a.dil == 1 && !a.in_relu &&
                         in_scale == 1.0f && in_shift == 0.0f && Hin == Hout && Win == Wout;
        const bool dma_ok = f16 && a.w_split && (Cin % BK) == 0 && in_scale == 1.0f && in_shift == 0.0f && a.dil == 1 &&
                            getenv("ZS_CONV_NO_DMA") == nullptr;
        const long long ksteps = (a.K + BK - 1) / BK;
        if (small && !(flags & ZS_CONV_FORCE_SMALL) && workspace && dma_ok && ksteps >= dsplit_min_steps && big_tiles >= dsplit_min_tiles) {
            long long sp = (384 + big_tiles - 1) / big_tiles;
            const long long cap = (long long)(WS_PARTS_BYTES / 4) / (M * (long long)Cout);
            if (sp > ksteps / 24) sp = ksteps / 24;
            if (sp > cap) sp = cap;
            if (sp > 1) {
                a.splits = (int)sp;
                const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(a.CoutPad / BN), (unsigned)sp);
                if (pw_) hipLaunchKernelGGL((conv_gemm_dma_kernel<true, false, false, 3, 2>), grid, dim3(256), 0, st, a);
                else if (a.in_relu) hipLaunchKernelGGL((conv_gemm_dma_kernel<false, true, false, 3, 2>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((conv_gemm_dma_kernel<false, false, false, 3, 2>), grid, dim3(256), 0, st, a);
                const long long total = M * (long long)Cout;
                const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
                hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
                return zs::check_launch("zs_conv2d_nhwc") ? 1 : 0;
            }
        }
    }
    if (small) {
        const bool tm = !pw && !no_tm && (Cin % 8) == 0;
        const bool narrow = (fuse && fuse->out_mode == 2) ? zs_conv2d_fused_cols((int)M, Cout) == 32 : small_is_narrow(M, Cout);
        // ---- pointwise layers of few rows: the streaming GEMM kernel (csrc/nn_gemm_stream.hip) ----
        const bool pw_geom = kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && a.dil == 1 && Hin == Hout && Win == Wout;
        if (stream_enabled() && pw_geom && f16 && a.w_split && in_scale == 1.0f && in_shift == 0.0f && M <= stream_max_rows() &&
            (!fuse || ((fuse->in_mode == 0 || fuse->in_mode == 2) && (fuse->out_mode == 0 || fuse->out_mode == 2)))) {
            zs::stream_gemm::Args g;
            g.a = in; g.w = packed_w; g.scale = scale; g.shift = shift; g.res1 = res1; g.res2 = res2; g.out = out;
            g.M = (int)M; g.K = a.K; g.N = Cout; g.CoutPad = a.CoutPad; g.lda = Cin; g.act = act; g.in_relu = a.in_relu;
            g.in_stats = fuse && fuse->in_mode == 2 ? fuse->in_stats : nullptr;
            g.in_tiles = fuse ? fuse->in_tiles : 0;
            g.in_eps = fuse ? fuse->in_eps : 0.f;
            g.out_stats = fuse && fuse->out_mode == 2 ? fuse->out_stats : nullptr;
            g.stats_cols = zs_conv2d_fused_cols((int)M, Cout);
            g.parts = workspace ? static_cast<float *>(workspace) + WS_COUNTER_FLOATS : nullptr;
            g.tickets = static_cast<int *>(workspace);
            g.parts_bytes = WS_PARTS_BYTES;
            g.max_tickets = (int)WS_COUNTER_FLOATS;
            int ranges = 1;
            g.two_launch_max = workspace && (Cout & 3) == 0 ? 4 : 1;
            g.ranges = &ranges;
            g.a_k16 = g.out_k16 = 0;
            if (zs::stream_gemm::launch(g, st)) {
                if (ranges > 1) {           // the kernel wrote raw partial sums of `ranges` K ranges: sum + epilogue (+ statistics) here
                    a.splits = ranges;
                    if (g.out_stats) {
                        const int cols = g.stats_cols;
                        const dim3 rgrid((unsigned)((M + SM - 1) / SM), (unsigned)((Cout + cols - 1) / cols));
                        if (cols == 32) hipLaunchKernelGGL(conv_splitk_reduce_stats_kernel<1>, rgrid, dim3(256), 0, st, a);
                        else hipLaunchKernelGGL(conv_splitk_reduce_stats_kernel<2>, rgrid, dim3(256), 0, st, a);
                    } else {
                        const long long total = M * (long long)Cout;
                        const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
                        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
                    }
                }
                return zs::check_launch("zs_conv2d_nhwc") ? 1 : 0;
            }
        }
        const long long wgs = ((M + SM - 1) / SM) * ((Cout + (narrow ? 31 : 63)) / (narrow ? 32 : 64));
        // Split K across workgroups while the launch would leave most of the 256 CUs idle (14x14 maps,
        // 197-token matrices at batch 1: 28-84 workgroups): ~512 workgroups, >= 16 t-steps (K = 128) per
        // split, partial tiles inside the caller's workspace (zs_conv2d_splitk_workspace_bytes()).
        static const int split_target = getenv("ZS_CONV_SPLIT_TARGET") ? atoi(getenv("ZS_CONV_SPLIT_TARGET")) : 512;
        // ... and, whatever the caller's flag, layers with a LONG contraction on few workgroups (3 x 3 768 -> 768 on 7 x 7 maps at
        // batch 1: 48 workgroups stream 21 MB of weights in 33.7 us): K >= ZS_CONV_SPLIT_LONG_K on <= ZS_CONV_SPLIT_LONG_WGS workgroups
        // split towards ZS_CONV_SPLIT_LONG_TARGET workgroups.  Round 3 had tried the split everywhere at target 512 and found it a
        // loss at batch 1; restricted to long contractions and with FEW ranges it pays (tools/enc_b1.py, batch-1 forward):
        //   off 2.923 ms | K >= 6912: 2.838 | 4096: 2.844 | 2304: 2.792 (target 512) | 2304 at target 384 / 256 / 192: 2.785 / 2.787 /
        //   2.726 | K >= 2048 at target 192 / 160 / 128 / 96: 2.712 / 2.717 / 2.752 / 2.788 | K >= 1024 at 160: 2.723
        static const long long long_k = getenv("ZS_CONV_SPLIT_LONG_K") ? atoll(getenv("ZS_CONV_SPLIT_LONG_K")) : 2048;
        static const long long long_wgs = getenv("ZS_CONV_SPLIT_LONG_WGS") ? atoll(getenv("ZS_CONV_SPLIT_LONG_WGS")) : 64;
        static const int long_target = getenv("ZS_CONV_SPLIT_LONG_TARGET") ? atoi(getenv("ZS_CONV_SPLIT_LONG_TARGET")) : 192;
        const bool flagged = (flags & ZS_CONV_SPLIT_SMALL) != 0;
        // (fused layers too - their statistics are then written by conv_splitk_reduce_stats_kernel; the vector epilogue needs Cout % 4)
        const bool long_split = workspace && (!fuse || (Cout & 3) == 0) && a.K >= long_k && wgs <= long_wgs;
        if (workspace && (flagged || long_split) && wgs < 256) {
            const int T = (a.K + BK - 1) / BK * 2;
            const int target = flagged ? split_target : long_target;
            long long sp = (target + wgs - 1) / wgs;
            if (sp > T / 16) sp = T / 16;
            if (sp > 16) sp = 16;
            const long long cap = (long long)(WS_SPLITK_BYTES / 4) / (M * (long long)Cout);
            if (sp > cap) sp = cap;
            if (sp > 1) a.splits = (int)sp;
        }
        if (fuse && fuse->in_mode) {
            // fused input normalisation: pointwise (in_mode 1, 2, 3) or tap-major plain (in_mode 1) geometry, split-fp16
            if (!(pw || (tm && plain && fuse->in_mode == 1))) {
                zs::set_err("zs_conv2d_nhwc_fused: in_mode %d needs a pointwise layer (or, for in_mode 1, Cin %% 8 == 0)", fuse->in_mode);
                return 0;
            }
#define ZS_LAUNCH_XF(NJ_, XF_)                                                                                                 \
    do {                                                                                                                       \
        if (pw) hipLaunchKernelGGL((conv_gemm_small_kernel<NJ_, true, 0, false, true, XF_>), grid, dim3(256), 0, st, a);          \
        else hipLaunchKernelGGL((conv_gemm_small_kernel<NJ_, false, 2, true, true, (XF_ == 1 ? 1 : 0)>), grid, dim3(256), 0, st, a); \
    } while (0)
            const dim3 grid((unsigned)((M + SM - 1) / SM), (unsigned)((Cout + (narrow ? 31 : 63)) / (narrow ? 32 : 64)), (unsigned)a.splits);
            if (narrow) {
                if (fuse->in_mode == 1) ZS_LAUNCH_XF(1, 1); else ZS_LAUNCH_XF(1, 2);
            } else {
                if (fuse->in_mode == 1) ZS_LAUNCH_XF(2, 1); else ZS_LAUNCH_XF(2, 2);
            }
#undef ZS_LAUNCH_XF
        } else if (narrow) {
            const dim3 grid((unsigned)((M + SM - 1) / SM), (unsigned)((Cout + 31) / 32), (unsigned)a.splits);
            ZS_LAUNCH(conv_gemm_small_kernel, 1,);
        } else {
            const dim3 grid((unsigned)((M + SM - 1) / SM), (unsigned)((Cout + 63) / 64), (unsigned)a.splits);
            ZS_LAUNCH(conv_gemm_small_kernel, 2,);
        }
        if (a.splits > 1) {
            if (fuse && fuse->out_mode) {           // epilogue + statistics of the summed ranges, in the unsplit launch's layout
                const dim3 rgrid((unsigned)((M + SM - 1) / SM), (unsigned)((Cout + (narrow ? 31 : 63)) / (narrow ? 32 : 64)));
                if (narrow) hipLaunchKernelGGL(conv_splitk_reduce_stats_kernel<1>, rgrid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL(conv_splitk_reduce_stats_kernel<2>, rgrid, dim3(256), 0, st, a);
            } else {
                const long long total = M * (long long)Cout;
                const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
                hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
            }
        }
    } else {
        const bool tm = !pw && !no_tm && (Cin % BK) == 0;
        // Stream-K (ZS_CONV_STREAM_K + a workspace): a fixed number of workgroups share the (tile, k-step) space
        // evenly, so a layer whose tile count is 1.03 x the number of CUs (ViT fc2 / proj at batch 28: 264 tiles)
        // does not run a second, almost empty round.  Whole multiples of the machine keep one tile per workgroup.
        static const int sk_wgs = getenv("ZS_CONV_SK_WGS") ? atoi(getenv("ZS_CONV_SK_WGS")) : 512;
        const long long ksteps = (a.K + BK - 1) / BK, iters = big_tiles * ksteps;
        // (for the 128 x 128 kernels only where one round would leave most of the 768 slots empty and K is long:
        // elsewhere the kernels are bound by the bytes they move, a short last round runs faster, and the exchange of
        // shares costs more than the balance gains - measured, tools/conv_ns.sh)
        static const bool sk_env = getenv("ZS_CONV_SK_ALWAYS") != nullptr;
        const bool sk_always = sk_env || (flags & ZS_CONV_STREAM_K_ALWAYS);
        const bool sk = workspace && (flags & ZS_CONV_STREAM_K) && sk_wgs >= 1 && sk_wgs <= SK_MAX_WGS &&
                        big_tiles <= (long long)WS_COUNTER_FLOATS && iters < (1LL << 30) && iters >= 4LL * sk_wgs &&
                        (sk_always || (big_tiles * 2 <= sk_wgs && ksteps >= 128));
        // split-fp16 with pre-split weights, pointwise or tap-major geometry, no input scale / shift: the LDS-DMA
        // pipelined kernel (ZS_CONV_NO_DMA=1: the register-staged one, for A/B measurements)
        static const bool no_dma = getenv("ZS_CONV_NO_DMA") != nullptr;
        const bool dma = f16 && a.w_split && !no_dma && (pw || tm) && in_scale == 1.0f && in_shift == 0.0f && (Cin % BK) == 0;
        // ZS_CONV_SLAB_MAJOR=1: kh x kw layers walk K slab-major (SlabWalk).  OFF: measured slower (see SlabWalk)
        static const bool slab_major = getenv("ZS_CONV_SLAB_MAJOR") && atoi(getenv("ZS_CONV_SLAB_MAJOR")) != 0;
        a.slab_major = dma && !pw && slab_major && a.kh * a.kw > 1 ? 1 : 0;
        // Variants kept for A/B runs, all measured inside the batch-28 encoder (tools/conv_shapes.py, 19.8 ms of
        // GEMM time with the default): ZS_CONV_DMA_MI=4 = 256 x 128 workgroup tiles (wave tile 128 x 64, two workgroups
        // per CU, 25 % fewer bytes through LDS per MFMA): SLOWER, 21.1 ms (3x3 476 vs 425 us, fc1 179 vs 140 us);
        // ZS_CONV_DMA_NS=2 = two stages, four workgroups per CU, one step of lead: the same, 19.7 ms.  Fewer bytes with
        // fewer loads in flight loses; more workgroups with less lead changes nothing.
        static const int dma_mi = getenv("ZS_CONV_DMA_MI") ? atoi(getenv("ZS_CONV_DMA_MI")) : 2;
        const long long tall_tiles = ((M + 255) / 256) * (a.CoutPad / BN);
        const bool tall = dma_mi == 4 && tall_tiles >= 1;
        // 64 x 128 tiles (MI = 1, four workgroups per CU) where 128-row tiles give most CUs a single workgroup - ViT fc2 /
        // proj at batch 28: 264 tiles on 256 CUs, four waves per CU.  tools/conv_shapes.py, batch 28, threshold 0 / 300 / 600 / 1200:
        // 17.17 / 16.83 / 16.44 / 16.94 ms of convolutions (fc2 167 -> 148 us, proj 73 -> 55, the 344-tile 1x1 layers 43 -> 32)
        static const long long half_below = getenv("ZS_CONV_HALF_BELOW") ? atoll(getenv("ZS_CONV_HALF_BELOW")) : 600;
        const bool half_tall = dma_mi == 1 || (dma_mi == 2 && big_tiles < half_below);
#define ZS_LAUNCH_DMA1(SKF, NS, MI_)                                                                                     \
    do {                                                                                                                 \
        if (pw) hipLaunchKernelGGL((conv_gemm_dma_kernel<true, false, SKF, NS, MI_>), grid, dim3(256), 0, st, a);         \
        else if (a.in_relu) hipLaunchKernelGGL((conv_gemm_dma_kernel<false, true, SKF, NS, MI_>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((conv_gemm_dma_kernel<false, false, SKF, NS, MI_>), grid, dim3(256), 0, st, a);           \
    } while (0)
        static const int dma_ns = getenv("ZS_CONV_DMA_NS") ? atoi(getenv("ZS_CONV_DMA_NS")) : 3;
#define ZS_LAUNCH_DMA(SKF)                              \
    do {                                                \
        if (dma_ns == 2) ZS_LAUNCH_DMA1(SKF, 2, 2);     \
        else ZS_LAUNCH_DMA1(SKF, 3, 2);                 \
    } while (0)
        // 256 x 256 tiles over 256 persistent workgroups (layers with >= 192 output channels): OPT-IN
        // (ZS_CONV_256_MIN_KSTEPS=128 selects it for K >= 2048).  Run back to back on hot caches it beats the
        // 128 x 128 kernel on the long contractions (tools/conv_ns.sh, batch 28: ViT fc2 152 vs 177 us, 3x3 256 -> 256
        // at 56 x 56 419 vs 476 us; fc1 158 vs 142, proj 95 vs 57: the exchange of 256 KiB tile shares), but inside
        // the encoder, where every layer's input has just been written, it loses everywhere (tools/conv_shapes.py:
        // fc2 242 vs 195 us, the 3x3 462 vs 424 us): one workgroup per CU has a third of the loads in flight.
        static const bool no_256 = getenv("ZS_CONV_NO_256") != nullptr;
        static const long long min_ksteps_256 = getenv("ZS_CONV_256_MIN_KSTEPS") ? atoll(getenv("ZS_CONV_256_MIN_KSTEPS")) : (1LL << 40);
        const long long tiles2 = ((M + TM2 - 1) / TM2) * ((a.CoutPad + TN2 - 1) / TN2), iters2 = tiles2 * ksteps;
        if (dma && workspace && (flags & ZS_CONV_STREAM_K) && !no_256 && Cout >= 192 && (ksteps >= min_ksteps_256 || sk_always) && iters2 >= 8LL * 256 &&
            iters2 < (1LL << 30) && tiles2 <= (long long)WS_COUNTER_FLOATS) {
            a.sk_per = (int)((iters2 + 255) / 256);
            const dim3 grid((unsigned)((iters2 + a.sk_per - 1) / a.sk_per));
            if (pw) hipLaunchKernelGGL((conv_gemm_dma256_kernel<true, false>), grid, dim3(512), 0, st, a);
            else if (a.in_relu) hipLaunchKernelGGL((conv_gemm_dma256_kernel<false, true>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((conv_gemm_dma256_kernel<false, false>), grid, dim3(512), 0, st, a);
        } else if (sk) {
            a.sk_per = (int)((iters + sk_wgs - 1) / sk_wgs);
            const dim3 grid((unsigned)((iters + a.sk_per - 1) / a.sk_per));
            if (dma) ZS_LAUNCH_DMA(true);
            else if (f16) ZS_LAUNCH_BIG(true, true);
            else ZS_LAUNCH_BIG(false, true);
        } else if (dma && tall) {
            const dim3 grid((unsigned)((M + 255) / 256), (unsigned)(a.CoutPad / BN));
            ZS_LAUNCH_DMA1(false, 3, 4);
        } else if (dma && half_tall) {
            const dim3 grid((unsigned)((M + 63) / 64), (unsigned)(a.CoutPad / BN));
            ZS_LAUNCH_DMA1(false, 3, 1);
        } else {
            const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(a.CoutPad / BN));
            if (dma) ZS_LAUNCH_DMA(false);
            else if (f16) ZS_LAUNCH_BIG(true, false);
            else ZS_LAUNCH_BIG(false, false);
        }
#undef ZS_LAUNCH_DMA
#undef ZS_LAUNCH_DMA1
    }
#undef ZS_LAUNCH_BIG
#undef ZS_LAUNCH
#undef ZS_LAUNCH1
    return zs::check_launch("zs_conv2d_nhwc") ? 1 : 0;
}

extern "C" int zs_conv3x3_tail_nhwc(const float *in, const float *packed_w, const float *scale, const float *shift, float *out,
                                    int batch, int H, int W, int Cin, int Cout, int flags, int act, const float *tail_w,
                                    const float *tail_b, int tail_act, void *stream) {
    const int need = ZS_CONV_F16X3 | ZS_CONV_W_PRESPLIT;
    if (batch < 0 || H < 8 || W < 8 || Cin <= 0 || (Cin % BK) || Cout <= 0 || Cout > 32 || (flags & need) != need ||
        (flags & ~(need | ZS_CONV_IN_RELU | ZS_CONV_IN_UPSAMPLE2)) || act < 0 || act > ZS_ACT_RELU_CLAMP1 || tail_act < 0 ||
        tail_act > ZS_ACT_RELU_CLAMP1 || ((flags & ZS_CONV_IN_UPSAMPLE2) && ((H | W) & 1))) {
        zs::set_err("zs_conv3x3_tail_nhwc: takes 3x3 stride-1 pad-1 layers of <= 32 output channels, Cin %% 16 == 0, maps >= 8x8, "
                    "flags ZS_CONV_F16X3 | ZS_CONV_W_PRESPLIT [| ZS_CONV_IN_RELU] (B=%d %dx%dx%d -> %d, flags %d)", batch, H, W,
                    Cin, Cout, flags);
        return 0;
    }
    if (batch == 0) return 1;
    if (!in || !packed_w || !out || !tail_w) { zs::set_err("zs_conv3x3_tail_nhwc: null pointer"); return 0; }
    const long long M = (long long)batch * H * W;
    if (M > (1LL << 30)) { zs::set_err("zs_conv3x3_tail_nhwc: %lld output pixels", M); return 0; }
    ConvArgs a;
    a.slab_major = 0;
    memset(&a.fz, 0, sizeof a.fz);
    a.fz_tiles_per_sample = 0;
    a.in = in; a.w = packed_w; a.scale = scale; a.shift = shift; a.res1 = nullptr; a.res2 = nullptr; a.out = out;
    const bool up2 = (flags & ZS_CONV_IN_UPSAMPLE2) != 0;
    a.B = batch; a.Hin = up2 ? H / 2 : H; a.Win = up2 ? W / 2 : W; a.Cin = Cin; a.Hout = H; a.Wout = W; a.Cout = Cout;
    a.CoutPad = (Cout + BN - 1) / BN * BN;
    a.kh = 3; a.kw = 3; a.stride = 1; a.pad_t = 1; a.pad_l = 1; a.K = 9 * Cin; a.M = (int)M;
    a.in_relu = (flags & ZS_CONV_IN_RELU) ? 1 : 0;
    a.act = act; a.in_scale = 1.0f; a.in_shift = 0.0f; a.dil = 1; a.ws = nullptr; a.splits = 1; a.sk_per = 0; a.w_split = 1;
    a.tail_w = tail_w; a.tail_b = tail_b; a.tail_act = tail_act;
    const int tx = (W + patch32::PT - 1) / patch32::PT, ty = (H + patch32::PT - 1) / patch32::PT;
    const dim3 grid((unsigned)((long long)batch * tx * ty));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (up2) {
        if (a.in_relu) hipLaunchKernelGGL((conv3x3_patch32_kernel<true, 8, true>), grid, dim3(512), 0, st, a, tx, ty);
        else hipLaunchKernelGGL((conv3x3_patch32_kernel<false, 8, true>), grid, dim3(512), 0, st, a, tx, ty);
    } else if (a.in_relu) hipLaunchKernelGGL((conv3x3_patch32_kernel<true, 8>), grid, dim3(512), 0, st, a, tx, ty);
    else hipLaunchKernelGGL((conv3x3_patch32_kernel<false, 8>), grid, dim3(512), 0, st, a, tx, ty);
    return zs::check_launch("zs_conv3x3_tail_nhwc") ? 1 : 0;
}

extern "C" int zs_conv2d_presplit_weight(const float *packed_w, float *split_w, int Cin, int Cout, int kh, int kw,
                                         void *stream) {
    if (!packed_w || !split_w || packed_w == split_w || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0) {
        zs::set_err("zs_conv2d_presplit_weight: bad arguments");
        return 0;
    }
    const size_t K16 = ((size_t)kh * kw * Cin + BK - 1) / BK * BK, CoutPad = ((size_t)Cout + BN - 1) / BN * BN;
    const size_t pairs = K16 / BK * 2 * CoutPad;
    hipLaunchKernelGGL(presplit_weight_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const f32x4 *>(packed_w),
                       reinterpret_cast<f32x4 *>(split_w), pairs, (int)CoutPad);
    return zs::check_launch("zs_conv2d_presplit_weight") ? 1 : 0;
}

extern "C" int zs_conv2d_presplit_weight_multi(const void *const *packed, void *const *split, const int *cout_pad,
                                               const unsigned long long *pair_prefix, int n, unsigned long long total_pairs,
                                               void *stream) {
    if (n < 0) { zs::set_err("zs_conv2d_presplit_weight_multi: bad arguments"); return 0; }
    if (n == 0 || total_pairs == 0) return 1;
    if (!packed || !split || !cout_pad || !pair_prefix) { zs::set_err("zs_conv2d_presplit_weight_multi: null pointer"); return 0; }
    const unsigned long long per = 256ull * PRESPLIT_PER_THREAD, blocks = (total_pairs + per - 1) / per;
    if (blocks > 0x7fffffffull) { zs::set_err("zs_conv2d_presplit_weight_multi: %llu pairs", total_pairs); return 0; }
    hipLaunchKernelGGL(presplit_weight_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const f32x4 *const *>(packed), reinterpret_cast<f32x4 *const *>(split), cout_pad,
                       pair_prefix, n);
    return zs::check_launch("zs_conv2d_presplit_weight_multi") ? 1 : 0;
}

extern "C" size_t zs_conv2d_packed_floats(int Cin, int Cout, int kh, int kw) {
    const size_t K16 = ((size_t)kh * kw * Cin + BK - 1) / BK * BK;
    const size_t CoutPad = ((size_t)Cout + BN - 1) / BN * BN;
    return K16 * CoutPad;
}
