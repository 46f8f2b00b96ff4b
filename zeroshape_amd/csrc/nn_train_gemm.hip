// Training-side GEMM work of the convolution / linear layers (fp32 MFMA, v_mfma_f32_32x32x2_f32):
//   zs_pack_conv_weight  torch-layout weights -> the [K16/4][CoutPad][4] operand of zs_conv2d_nhwc,
//                        either for the forward product or for the data gradient (taps flipped,
//                        Cin <-> Cout swapped), on the GPU so it can run every optimiser step
//   zs_conv2d_wgrad      dW[cout][k] = sum_pixels dY[pixel][cout] * A[pixel][k]  (A = the same
//                        implicit im2col operand as the forward pass, incl. its input transform)
//   zs_standardize_weight / _bwd   timm StdConv2d weight standardisation and its adjoint
// The data gradient itself is zs_conv2d_nhwc on dY with the flipped pack (ZS_CONV_IN_DILATE2 for
// stride 2).  Used by: Implicit (model/shape/implicit.py) in training, DPT / ResNet encoders
// (training half).
#include "zs_common.h"
#include "zs_split16.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }
inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

// ---------------------------------------------------------------------------------------------
// weight packing.  Source: w[cout * ld + (cin0 + c) * kh*kw + tap], c < Cin (a channel sub-range
// of a torch [Cout][CinTot][kh][kw] tensor, ld = CinTot*kh*kw).
//   forward pack : GEMM K = taps * CinP  (CinP = Cin rounded up to 4, zero filled), N = Cout
//                  k = tap * CinP + c           -> w[n][c][tap]
//   dgrad pack   : GEMM K = taps * CoutP (CoutP = Cout rounded up to 4), N = Cin
//                  k = tap' * CoutP + co        -> w[co][n][taps-1-tap']   (both axes flipped)
// dst[(k/4) * NPad * 4 + n * 4 + k%4], NPad = N rounded up to 128, K16 = K rounded up to 16.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ dst,
                                                          int Cout, int Cin, int cin0, int ld, int taps, int dgrad,
                                                          int K16, int NPad) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)K16 * NPad;
    if (i >= total) return;
    const int e = i & 3, n = (i >> 2) % NPad, k = (int)((i >> 2) / NPad) * 4 + e;
    float v = 0.f;
    if (!dgrad) {
        const int CinP = (Cin + 3) & ~3, tap = k / CinP, c = k - tap * CinP;
        if (tap < taps && c < Cin && n < Cout) v = w[(size_t)n * ld + (size_t)(cin0 + c) * taps + tap];
    } else {
        const int CoutP = (Cout + 3) & ~3, tap = k / CoutP, co = k - tap * CoutP;
        if (tap < taps && co < Cout && n < Cin) v = w[(size_t)co * ld + (size_t)(cin0 + n) * taps + (taps - 1 - tap)];
    }
    dst[i] = v;
}

// the same over a device table of layers: one launch re-packs every operand of the model after an
// optimiser step (chunk c = elements [chunk_start[c], +PACK_CHUNK) of entry chunk_entry[c])
struct PackEntry {               // mirrors include/zeroshape_hip.h zs_pack_entry
    const float *src;
    float *dst;
    int Cout, Cin, cin0, ld, taps, dgrad, K16, NPad;
};
constexpr int PACK_CHUNK = 16384;
constexpr int PACK_TILE_TAPS = 9;          // kernels up to 3x3 repack through LDS tiles
constexpr int PACK_ROWS = 64;              // operand columns (n) per tile
__host__ __device__ inline int pack_ct(int taps) { return taps == 1 ? 64 : 16; }      // K-side channels per tile
// chunks of one entry: LDS tiles (64 n x CT channels x all taps) for small kernels, PACK_CHUNK elements otherwise
static int pack_entry_chunks(int Cout, int Cin, int taps, int dgrad, int K16, int NPad) {
    if (taps > PACK_TILE_TAPS) return (int)(((size_t)K16 * NPad + PACK_CHUNK - 1) / PACK_CHUNK);
    const int Kc = ((dgrad ? Cout : Cin) + 3) / 4 * 4, CT = pack_ct(taps);
    return ((Kc + CT - 1) / CT) * (NPad / PACK_ROWS);
}
constexpr int PACK_LDS_FLOATS = 64 * 145;  // >= 64 x (16*9 | 1), 16 x 64*9, 64 x 65

// A repack is a transpose: the source is contiguous along (channel, tap) of one output channel, the operand along n.
// A tile is read with whole contiguous source spans per row (64 B .. 2.3 KB) into LDS and written as 16-byte
// (k-group, n) cells that are contiguous over n (1 KB runs).  Reading the source cell by cell instead made every
// 64-byte line travel to four (pointwise) or nine (3x3) workgroups on different XCDs.
// The K padding rows of the operand (k >= taps * Kc) are written once by zs_pack_conv_weight and never change.
// split_dst (optional, one pointer per entry, NULL entries allowed): the operand's fp16 halves in the layout of
// zs_conv2d_presplit_weight, written from the same LDS tile - for entries pack_inline_split() accepts (tile path, K side a
// multiple of 16 channels: a tile then holds whole K = 16 groups, k-quads 4 s + q and 4 s + q + 2 meet in one thread).  With
// split_only the fp32 operand of such an entry is not written at all (optim.amp: every consumer reads the halves) - the
// re-pack + split of 191 M parameters moved 6.1 GB per optimiser step in two launches, this way 3.1 GB in one.
static bool pack_inline_split(int Cout, int Cin, int taps, int dgrad) {
    const int Kc = ((dgrad ? Cout : Cin) + 3) / 4 * 4;
    return taps <= PACK_TILE_TAPS && Kc % 16 == 0;
}
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackEntry *__restrict__ tab,
                                                                const int *__restrict__ chunk_entry,
                                                                const unsigned long long *__restrict__ chunk_start,
                                                                float *const *__restrict__ split_dst, int split_only) {
    extern __shared__ __attribute__((aligned(16))) float plds[];
    const int entry = chunk_entry[blockIdx.x];
    const PackEntry t = tab[entry];
    float *const sp = split_dst ? split_dst[entry] : nullptr;      // (host: non-NULL only for pack_inline_split entries)
    const int CinP = (t.Cin + 3) & ~3, CoutP = (t.Cout + 3) & ~3;
    const unsigned ord = (unsigned)chunk_start[blockIdx.x];
    if (t.taps > PACK_TILE_TAPS) {
        const size_t total = (size_t)t.K16 * t.NPad, s0 = (size_t)ord * PACK_CHUNK;
        const size_t s1 = s0 + PACK_CHUNK < total ? s0 + PACK_CHUNK : total;
        for (size_t i = s0 + 4 * (size_t)threadIdx.x; i < s1; i += 1024) {
            const unsigned q = (unsigned)(i >> 2);                 // 32-bit divisions: K16 * NPad / 4 < 2^32
            const int n = (int)(q % (unsigned)t.NPad), k = (int)(q / (unsigned)t.NPad) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (!t.dgrad) {
                const int tap = k / CinP, c = k - tap * CinP;      // CinP % 4 == 0: one tap per k-group
                if (tap < t.taps && n < t.Cout)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (c + e < t.Cin) v[e] = t.src[(size_t)n * t.ld + (size_t)(t.cin0 + c + e) * t.taps + tap];
            } else {
                const int tap = k / CoutP, co = k - tap * CoutP;
                if (tap < t.taps && n < t.Cin)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (co + e < t.Cout)
                            v[e] = t.src[(size_t)(co + e) * t.ld + (size_t)(t.cin0 + n) * t.taps + (t.taps - 1 - tap)];
            }
            *reinterpret_cast<f32x4 *>(t.dst + i) = v;
        }
        return;
    }
    const int taps = t.taps, CT = pack_ct(taps), NT = t.NPad / PACK_ROWS;
    const int n0 = (int)(ord % (unsigned)NT) * PACK_ROWS, c0 = (int)(ord / (unsigned)NT) * CT;
    const int quads = CT / 4, cells = taps * quads * PACK_ROWS;
    const bool vec_ok = (t.ld & 3) == 0 && (t.cin0 & 3) == 0 && (reinterpret_cast<size_t>(t.src) & 15) == 0;
    if (!t.dgrad) {
        // rows = output channels n, span = CT channels x taps of one row
        const int span = CT * taps, stride = span | 1;             // odd row stride: the cell gather below walks rows
        const float *src = t.src + (size_t)(t.cin0 + c0) * taps;
        if (taps == 1 && vec_ok && c0 + CT <= t.Cin) {            // pointwise: 16-byte loads of whole 256-byte rows
            for (int idx = threadIdx.x; idx < PACK_ROWS * (CT / 4); idx += 256) {
                const int r = idx / (CT / 4), j = 4 * (idx - r * (CT / 4)), n = n0 + r;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < t.Cout) v = *reinterpret_cast<const f32x4 *>(src + (size_t)n * t.ld + j);
#pragma unroll
                for (int e = 0; e < 4; e++) plds[r * stride + j + e] = v[e];
            }
        } else {
            for (int idx = threadIdx.x; idx < PACK_ROWS * span; idx += 256) {
                const int r = idx / span, j = idx - r * span, n = n0 + r, c = c0 + j / taps;
                plds[r * stride + j] = (n < t.Cout && c < t.Cin) ? src[(size_t)n * t.ld + j] : 0.f;
            }
        }
        __syncthreads();
        if (sp) {           // pairs of k-quads (cq, cq + 2), cq % 4 < 2: one K = 16 step's lane half
            for (int idx = threadIdx.x; idx < cells / 2; idx += 256) {
                const int nl = idx & (PACK_ROWS - 1), q = idx >> 6, pq = q % (quads / 2), tap = q / (quads / 2);
                const int cq = (pq >> 1) * 4 + (pq & 1), c = c0 + 4 * cq;
                if (c >= CinP) continue;
                const float *cell = plds + nl * stride + 4 * cq * taps + tap;
                const f32x4 v0 = {cell[0], cell[taps], cell[2 * taps], cell[3 * taps]};
                const f32x4 v1 = {cell[8 * taps], cell[9 * taps], cell[10 * taps], cell[11 * taps]};
                const size_t o = ((size_t)(tap * (CinP / 4) + c / 4) * t.NPad + n0 + nl) * 4, o2 = o + (size_t)2 * t.NPad * 4;
                zs::s16::u32x4 hi, lo;
                zs::s16::split8(v0, v1, hi, lo);
                *reinterpret_cast<f32x4 *>(sp + o) = __builtin_bit_cast(f32x4, hi);
                *reinterpret_cast<f32x4 *>(sp + o2) = __builtin_bit_cast(f32x4, lo);
                if (!split_only) {
                    *reinterpret_cast<f32x4 *>(t.dst + o) = v0;
                    *reinterpret_cast<f32x4 *>(t.dst + o2) = v1;
                }
            }
            return;
        }
        for (int idx = threadIdx.x; idx < cells; idx += 256) {
            const int nl = idx & (PACK_ROWS - 1), q = idx >> 6, cq = q % quads, tap = q / quads, c = c0 + 4 * cq;
            if (c >= CinP) continue;
            const float *cell = plds + nl * stride + 4 * cq * taps + tap;
            const f32x4 v = {cell[0], cell[taps], cell[2 * taps], cell[3 * taps]};
            *reinterpret_cast<f32x4 *>(t.dst + ((size_t)(tap * (CinP / 4) + c / 4) * t.NPad + n0 + nl) * 4) = v;
        }
    } else {
        // rows = output channels co (the K side), span = 64 input channels n x taps of one row; taps flipped
        const int span = PACK_ROWS * taps;
        const float *src = t.src + (size_t)(t.cin0 + n0) * taps;
        if (taps == 1 && vec_ok && n0 + PACK_ROWS <= t.Cin) {
            for (int idx = threadIdx.x; idx < CT * (PACK_ROWS / 4); idx += 256) {
                const int r = idx / (PACK_ROWS / 4), j = 4 * (idx - r * (PACK_ROWS / 4)), co = c0 + r;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (co < t.Cout) v = *reinterpret_cast<const f32x4 *>(src + (size_t)co * t.ld + j);
                *reinterpret_cast<f32x4 *>(plds + r * span + j) = v;
            }
        } else {
            for (int idx = threadIdx.x; idx < CT * span; idx += 256) {
                const int r = idx / span, j = idx - r * span, co = c0 + r, n = n0 + j / taps;
                plds[r * span + j] = (co < t.Cout && n < t.Cin) ? src[(size_t)co * t.ld + j] : 0.f;
            }
        }
        __syncthreads();
        if (sp) {
            for (int idx = threadIdx.x; idx < cells / 2; idx += 256) {
                const int nl = idx & (PACK_ROWS - 1), q = idx >> 6, pq = q % (quads / 2), tap = q / (quads / 2);
                const int cq = (pq >> 1) * 4 + (pq & 1), co = c0 + 4 * cq;
                if (co >= CoutP) continue;
                const float *cell = plds + 4 * cq * span + nl * taps + (taps - 1 - tap);
                const f32x4 v0 = {cell[0], cell[span], cell[2 * span], cell[3 * span]};
                const f32x4 v1 = {cell[8 * span], cell[9 * span], cell[10 * span], cell[11 * span]};
                const size_t o = ((size_t)(tap * (CoutP / 4) + co / 4) * t.NPad + n0 + nl) * 4, o2 = o + (size_t)2 * t.NPad * 4;
                zs::s16::u32x4 hi, lo;
                zs::s16::split8(v0, v1, hi, lo);
                *reinterpret_cast<f32x4 *>(sp + o) = __builtin_bit_cast(f32x4, hi);
                *reinterpret_cast<f32x4 *>(sp + o2) = __builtin_bit_cast(f32x4, lo);
                if (!split_only) {
                    *reinterpret_cast<f32x4 *>(t.dst + o) = v0;
                    *reinterpret_cast<f32x4 *>(t.dst + o2) = v1;
                }
            }
            return;
        }
        for (int idx = threadIdx.x; idx < cells; idx += 256) {
            const int nl = idx & (PACK_ROWS - 1), q = idx >> 6, cq = q % quads, tap = q / quads, co = c0 + 4 * cq;
            if (co >= CoutP) continue;
            const float *cell = plds + 4 * cq * span + nl * taps + (taps - 1 - tap);
            const f32x4 v = {cell[0], cell[span], cell[2 * span], cell[3 * span]};
            *reinterpret_cast<f32x4 *>(t.dst + ((size_t)(tap * (CoutP / 4) + co / 4) * t.NPad + n0 + nl) * 4) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  Workgroup tile: 128 couts x 128 k over a range of pixels; the pixel range is
// split over blockIdx.z and the partial sums are reduced in a fixed order by wgrad_reduce_kernel
// (deterministic).  LDS holds the two operand tiles in their natural [pixel][column] layout; an
// MFMA step contracts two pixels: lane (l32, half) supplies dY[p+half][cout l32] and
// A[p+half][k l32] - plain ds_read_b32, rows padded to 160 floats so the two halves hit
// disjoint banks.
// ---------------------------------------------------------------------------------------------
constexpr int WP = 16;

struct WgradArgs {
    const float *in, *dy;
    float *partial;                 // [splits][CoutP][K]
    float *bias_partial;            // [splits][CoutP] column sums of dY (the bias gradient), or NULL
    int B, Hin, Win, Cin, Hout, Wout, CoutP, kh, kw, stride, pad_t, pad_l, K, M;
    int in_relu;
    float in_scale, in_shift;
    int pix_per_split;
};

// T = tile edge: 128 (a wave owns 2x2 MFMA tiles) for layers with many output pixels, 64 (one MFMA
// tile per wave) for big weight matrices over few pixels (token matrices, 14x14 maps): four times
// the workgroups without splitting the short pixel range into slivers whose partial tiles would
// cost more to write and reduce than to compute.
// PW: pointwise layers (1x1, stride 1, no padding, no input transform): A[p][k] = in[p * Cin + k], no pixel
// decoding (two integer divisions per load) and no transform - VALU work the fp32 MFMA pipe cannot hide.
// MODE 0: generic (pixel decode by division, 64-bit offsets, input transform).  MODE 1 = PW.
// MODE 2 = FAST: no input transform, pixel decode advanced incrementally (Wout >= 6, Hout >= 2), 32-bit
// offsets (the input has < 2^31 elements) - every 3x3 layer of the encoders.
template <int T, int MODE>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    constexpr bool PW = MODE == 1, FAST = MODE == 2;
    constexpr int WLD = T + 32;                 // row stride % 64 == 32: the two k-halves hit disjoint banks
    constexpr int QUADS = T / 4, ROWS = 256 / QUADS, PASSES = WP / ROWS, NI = T / 64;
    __shared__ __attribute__((aligned(16))) float lds_y[2][WP][WLD];
    __shared__ __attribute__((aligned(16))) float lds_a[2][WP][WLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int c0 = blockIdx.x * T, k0 = blockIdx.y * T;
    const int p_begin = blockIdx.z * a.pix_per_split, p_end = min(a.M, p_begin + a.pix_per_split);

    // loader role: rows (pixels) tid/QUADS (+ROWS) of the step, column quad tid%QUADS
    const int prow = tid / QUADS, quad = tid % QUADS;
    const int yc = c0 + 4 * quad;
    const bool yc_ok = yc < a.CoutP;
    const int kk = k0 + 4 * quad;              // A column quad: k fixed for the whole loop
    const bool k_ok = kk < a.K;
    int kc = 0, ky = 0, kx = 0;
    if (k_ok) {
        const int tap = kk / a.Cin;
        kc = kk - tap * a.Cin;
        ky = tap / a.kw;
        kx = tap - ky * a.kw;
    }
    const float relu_floor = a.in_relu ? 0.f : -INFINITY;
    const int HW = a.Hout * a.Wout;

    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    struct Frag { f32x4 y, x; bool ok; };
    // this thread's pixels p_begin + prow + ROWS*q + 16*step, decoded to (image, row, column) once and
    // then advanced by 16 per step with compares and subtracts (no division in the loop); needs
    // 16 <= 3 * Wout and 3 <= Hout... otherwise (tiny maps) the divisions stay
    constexpr bool incremental = FAST;
    int qb[PASSES], qy[PASSES], qx[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; q++) {
        const int p = p_begin + prow + ROWS * q;
        qb[q] = p / HW;
        const int rem = p - qb[q] * HW;
        qy[q] = rem / a.Wout;
        qx[q] = rem - qy[q] * a.Wout;
    }
    auto load = [&](int p, int q) -> Frag {
        Frag f;
        const bool p_ok = p < p_end;
        const int pc = p_ok ? p : p_begin;
        f.y = (yc_ok && p_ok) ? *reinterpret_cast<const f32x4 *>(a.dy + (size_t)pc * a.CoutP + yc) : f32x4{0, 0, 0, 0};
        if (PW) {
            f.ok = p_ok && k_ok;
            f.x = *reinterpret_cast<const f32x4 *>(a.in + (f.ok ? (size_t)pc * a.Cin + kk : 0));
            return f;
        }
        int pb, py, px;
        if (incremental) {
            pb = qb[q]; py = qy[q]; px = qx[q];
            qx[q] += WP;                                   // next step's pixel
#pragma unroll
            for (int r = 0; r < 3; r++)
                if (qx[q] >= a.Wout) { qx[q] -= a.Wout; qy[q]++; }
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (qy[q] >= a.Hout) { qy[q] -= a.Hout; qb[q]++; }
        } else {
            pb = pc / HW;
            const int rem = pc - pb * HW;
            py = rem / a.Wout;
            px = rem - py * a.Wout;
        }
        const int iy = py * a.stride - a.pad_t + ky, ix = px * a.stride - a.pad_l + kx;
        f.ok = p_ok && k_ok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        if (FAST) {
            const int off = f.ok ? ((pb * a.Hin + iy) * a.Win + ix) * a.Cin + kc : 0;
            f.x = *reinterpret_cast<const f32x4 *>(a.in + off);
        } else {
            const size_t off = f.ok ? (((size_t)pb * a.Hin + iy) * a.Win + ix) * a.Cin + kc : 0;
            f.x = *reinterpret_cast<const f32x4 *>(a.in + off);
        }
        return f;
    };
    auto store = [&](int buf, int row, const Frag &f) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++)
            v[e] = (PW || FAST) ? (f.ok ? f.x[e] : 0.f) : (f.ok ? fmaxf(f.x[e], relu_floor) * a.in_scale + a.in_shift : 0.f);
        *reinterpret_cast<f32x4 *>(&lds_y[buf][row][4 * quad]) = f.y;
        *reinterpret_cast<f32x4 *>(&lds_a[buf][row][4 * quad]) = v;
        bsum += f.y;                 // every dY row of the tile passes through exactly one thread per column quad
    };

    f32x16 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int wm = (wave & 1) * (T / 2), wn = (wave >> 1) * (T / 2);
    const int steps = (p_end - p_begin + WP - 1) / WP;
    if (steps > 0) {
        Frag f[PASSES];
#pragma unroll
        for (int q = 0; q < PASSES; q++) f[q] = load(p_begin + prow + ROWS * q, q);
#pragma unroll
        for (int q = 0; q < PASSES; q++) store(0, prow + ROWS * q, f[q]);
        __syncthreads();
        for (int s = 0; s < steps; s++) {
            const int cur = s & 1;
            const bool more = s + 1 < steps;
            if (more) {
#pragma unroll
                for (int q = 0; q < PASSES; q++) f[q] = load(p_begin + (s + 1) * WP + prow + ROWS * q, q);
            }
#pragma unroll
            for (int t = 0; t < WP / 2; t++) {
                float ya[NI], xb[NI];
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    ya[i] = lds_y[cur][2 * t + half][wm + 32 * i + l32];
                    xb[i] = lds_a[cur][2 * t + half][wn + 32 * i + l32];
                }
#pragma unroll
                for (int i = 0; i < NI; i++)
#pragma unroll
                    for (int j = 0; j < NI; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[i], xb[j], acc[i][j], 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int q = 0; q < PASSES; q++) store(cur ^ 1, prow + ROWS * q, f[q]);
            }
            __syncthreads();
        }
    }
    if (a.bias_partial && blockIdx.y == 0) {          // bias gradient: column sums of this tile's dY rows
        __syncthreads();
        *reinterpret_cast<f32x4 *>(&lds_y[0][prow][4 * quad]) = bsum;
        __syncthreads();
        if (prow == 0 && yc_ok) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < ROWS; r++) t += *reinterpret_cast<const f32x4 *>(&lds_y[0][r][4 * quad]);
            *reinterpret_cast<f32x4 *>(a.bias_partial + (size_t)blockIdx.z * a.CoutP + yc) = t;
        }
    }
    float *dst = a.partial + (size_t)blockIdx.z * a.CoutP * a.K;
#pragma unroll
    for (int j = 0; j < NI; j++) {
        const int k = k0 + wn + 32 * j + l32;
        if (k >= a.K) continue;
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = c0 + wm + 32 * i + 8 * (r >> 2) + 4 * half + (r & 3);
                if (co < a.CoutP) dst[(size_t)co * a.K + k] = acc[i][j][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient in split-fp16 arithmetic (round 3; optim.amp): dW = dY^T A contracts over PIXELS, the
// row index of both operands in memory, so the MFMA operands (eight consecutive k = pixels per lane) are columns of
// the staged tiles.  Gathering them at the MFMA (8 LDS reads + a split per operand and K-block) is VALU-bound - why
// round 2 priced this kernel and did not build it.  Here the LOADER splits once and stores PIXEL PAIRS: a thread that
// holds the channel quad of two adjacent pixels packs (pixel p, pixel p + 1) of each channel into one 32-bit word of
// fp16 heads and one of fp16 remainders (csrc/zs_split16.h); LDS is [pixel pair 0..7][column], so its four channels
// are ONE ds_write_b128 per operand half (a wave writes 1 KiB contiguous), and the MFMA operand of a lane - eight
// consecutive pixels of one column - is the four words (pair 4 half + t, column), t = 0..3: plain ds_read_b32 with
// consecutive lanes on consecutive banks (rows of T + 8 words put the two lane halves on disjoint banks), no VALU.
// A step of 16 pixels is 3 MFMAs of 32 cycles per 32 x 32 output tile instead of 8 fp32 MFMAs of 64.  Same tiles,
// pixel splits, partial layout and reduce kernel as wgrad_kernel; the bias gradient stays an fp32 sum.
// ---------------------------------------------------------------------------------------------
template <int T, int MODE>
__global__ __launch_bounds__(256) void wgrad_split_kernel(WgradArgs a) {
    using zs::s16::u32x4;
    constexpr bool PW = MODE == 1, FAST = MODE == 2;
    constexpr int QUADS = T / 4, ROWS = 256 / QUADS, PASSES = WP / ROWS, NI = T / 64;
    static_assert(WP == 16 && (PASSES == 1 || PASSES == 2), "one K-block of 16 pixels per step");
    constexpr int RS = T + 8;                   // words per pixel-pair row
    // [buffer][operand: dY | A][half: hi | lo][pixel pair][column], a word = fp16 (pixel 2 w) | fp16 (pixel 2 w + 1) << 16
    __shared__ __attribute__((aligned(16))) unsigned lds[2][2][2][(WP / 2) * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int c0 = blockIdx.x * T, k0 = blockIdx.y * T;
    const int p_begin = blockIdx.z * a.pix_per_split, p_end = min(a.M, p_begin + a.pix_per_split);
    // loader role: column quad tid % QUADS of the pixel rows PASSES * (tid / QUADS) + q of the step (adjacent rows)
    const int prow = tid / QUADS, quad = tid % QUADS;
    const int yc = c0 + 4 * quad;
    const bool yc_ok = yc < a.CoutP;
    const int kk = k0 + 4 * quad;
    const bool k_ok = kk < a.K;
    int kc = 0, ky = 0, kx = 0;
    if (k_ok) {
        const int tap = kk / a.Cin;
        kc = kk - tap * a.Cin;
        ky = tap / a.kw;
        kx = tap - ky * a.kw;
    }
    const float relu_floor = a.in_relu ? 0.f : -INFINITY;
    const int HW = a.Hout * a.Wout;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    struct Frag { f32x4 y, x; bool ok; };
    int qb[PASSES], qy[PASSES], qx[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; q++) {
        const int p = p_begin + PASSES * prow + q;
        qb[q] = p / HW;
        const int rem = p - qb[q] * HW;
        qy[q] = rem / a.Wout;
        qx[q] = rem - qy[q] * a.Wout;
    }
    auto load = [&](int p, int q) -> Frag {
        Frag f;
        const bool p_ok = p < p_end;
        const int pc = p_ok ? p : p_begin;
        f.y = (yc_ok && p_ok) ? *reinterpret_cast<const f32x4 *>(a.dy + (size_t)pc * a.CoutP + yc) : f32x4{0, 0, 0, 0};
        if (PW) {
            f.ok = p_ok && k_ok;
            f.x = *reinterpret_cast<const f32x4 *>(a.in + (f.ok ? (size_t)pc * a.Cin + kk : 0));
            return f;
        }
        int pb, py, px;
        if (FAST) {
            pb = qb[q]; py = qy[q]; px = qx[q];
            qx[q] += WP;
#pragma unroll
            for (int r = 0; r < 3; r++)
                if (qx[q] >= a.Wout) { qx[q] -= a.Wout; qy[q]++; }
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (qy[q] >= a.Hout) { qy[q] -= a.Hout; qb[q]++; }
        } else {
            pb = pc / HW;
            const int rem = pc - pb * HW;
            py = rem / a.Wout;
            px = rem - py * a.Wout;
        }
        const int iy = py * a.stride - a.pad_t + ky, ix = px * a.stride - a.pad_l + kx;
        f.ok = p_ok && k_ok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        if (FAST) {
            const int off = f.ok ? ((pb * a.Hin + iy) * a.Win + ix) * a.Cin + kc : 0;
            f.x = *reinterpret_cast<const f32x4 *>(a.in + off);
        } else {
            const size_t off = f.ok ? (((size_t)pb * a.Hin + iy) * a.Win + ix) * a.Cin + kc : 0;
            f.x = *reinterpret_cast<const f32x4 *>(a.in + off);
        }
        return f;
    };
    auto operand = [&](const Frag &f) -> f32x4 {       // the A operand's values of one pixel (transform + padding applied)
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++)
            v[e] = (PW || FAST) ? (f.ok ? f.x[e] : 0.f) : (f.ok ? fmaxf(f.x[e], relu_floor) * a.in_scale + a.in_shift : 0.f);
        return v;
    };
    auto store = [&](int buf, const Frag (&f)[PASSES]) {
        if (PASSES == 2) {       // words of (pixel 2 prow, pixel 2 prow + 1) of the thread's four columns: one b128 each
            const f32x4 x0 = operand(f[0]), x1 = operand(f[PASSES - 1]);
            u32x4 hy, ly, hx, lx;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                unsigned h, l;
                zs::s16::split2(f[0].y[e], f[PASSES - 1].y[e], h, l);
                hy[e] = h; ly[e] = l;
                zs::s16::split2(x0[e], x1[e], h, l);
                hx[e] = h; lx[e] = l;
            }
            const int w = prow * RS + 4 * quad;
            *reinterpret_cast<u32x4 *>(&lds[buf][0][0][w]) = hy;
            *reinterpret_cast<u32x4 *>(&lds[buf][0][1][w]) = ly;
            *reinterpret_cast<u32x4 *>(&lds[buf][1][0][w]) = hx;
            *reinterpret_cast<u32x4 *>(&lds[buf][1][1][w]) = lx;
            bsum += f[0].y + f[PASSES - 1].y;
        } else {                 // one pixel per thread: its half of each word, 16-bit stores
            const f32x4 x0 = operand(f[0]);
            const int base = ((prow >> 1) * RS + 4 * quad) * 2 + (prow & 1);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                unsigned h, l;
                zs::s16::split2(f[0].y[e], f[0].y[e + 1], h, l);
                reinterpret_cast<unsigned short *>(lds[buf][0][0])[base + 2 * e] = (unsigned short)(h & 0xffffu);
                reinterpret_cast<unsigned short *>(lds[buf][0][0])[base + 2 * e + 2] = (unsigned short)(h >> 16);
                reinterpret_cast<unsigned short *>(lds[buf][0][1])[base + 2 * e] = (unsigned short)(l & 0xffffu);
                reinterpret_cast<unsigned short *>(lds[buf][0][1])[base + 2 * e + 2] = (unsigned short)(l >> 16);
                zs::s16::split2(x0[e], x0[e + 1], h, l);
                reinterpret_cast<unsigned short *>(lds[buf][1][0])[base + 2 * e] = (unsigned short)(h & 0xffffu);
                reinterpret_cast<unsigned short *>(lds[buf][1][0])[base + 2 * e + 2] = (unsigned short)(h >> 16);
                reinterpret_cast<unsigned short *>(lds[buf][1][1])[base + 2 * e] = (unsigned short)(l & 0xffffu);
                reinterpret_cast<unsigned short *>(lds[buf][1][1])[base + 2 * e + 2] = (unsigned short)(l >> 16);
            }
            bsum += f[0].y;
        }
    };

    f32x16 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int wm = (wave & 1) * (T / 2), wn = (wave >> 1) * (T / 2);
    const int steps = (p_end - p_begin + WP - 1) / WP;
    if (steps > 0) {
        // the global loads run TWO steps ahead of the MFMAs (f: step s + 1, loaded during step s - 1 and stored to LDS during
        // step s; g: step s + 2, requested now): a step is 3-12 MFMAs and a barrier, a load takes longer than that - with one
        // step of distance every step waited for its loads (ViT fc layers at batch 4: 45 us for 50 steps)
        Frag f[PASSES], g[PASSES];
#pragma unroll
        for (int q = 0; q < PASSES; q++) f[q] = load(p_begin + PASSES * prow + q, q);
        store(0, f);
        if (steps > 1) {
#pragma unroll
            for (int q = 0; q < PASSES; q++) f[q] = load(p_begin + WP + PASSES * prow + q, q);
        }
#pragma unroll
        for (int q = 0; q < PASSES; q++) g[q] = f[q];
        __syncthreads();
        for (int s = 0; s < steps; s++) {
            const int cur = s & 1;
            const bool more = s + 1 < steps;
            if (s + 2 < steps) {
#pragma unroll
                for (int q = 0; q < PASSES; q++) g[q] = load(p_begin + (s + 2) * WP + PASSES * prow + q, q);
            }
            u32x4 ah[NI], al[NI], bh[NI], bl[NI];
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int oa = (4 * half + t) * RS + wm + 32 * i + l32, ob = (4 * half + t) * RS + wn + 32 * i + l32;
                    ah[i][t] = lds[cur][0][0][oa];
                    al[i][t] = lds[cur][0][1][oa];
                    bh[i][t] = lds[cur][1][0][ob];
                    bl[i][t] = lds[cur][1][1][ob];
                }
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NI; j++) zs::s16::mfma3(acc[i][j], ah[i], al[i], bh[j], bl[j]);
            if (more) store(cur ^ 1, f);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PASSES; q++) f[q] = g[q];
        }
    }
    if (a.bias_partial && blockIdx.y == 0) {          // bias gradient: column sums of this tile's dY rows (fp32)
        __shared__ __attribute__((aligned(16))) float bred[256 / (T / 4)][T];
        __syncthreads();
        *reinterpret_cast<f32x4 *>(&bred[prow][4 * quad]) = bsum;
        __syncthreads();
        if (prow == 0 && yc_ok) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < ROWS; r++) t += *reinterpret_cast<const f32x4 *>(&bred[r][4 * quad]);
            *reinterpret_cast<f32x4 *>(a.bias_partial + (size_t)blockIdx.z * a.CoutP + yc) = t;
        }
    }
    float *dst = a.partial + (size_t)blockIdx.z * a.CoutP * a.K;
#pragma unroll
    for (int j = 0; j < NI; j++) {
        const int k = k0 + wn + 32 * j + l32;
        if (k >= a.K) continue;
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = c0 + wm + 32 * i + 8 * (r >> 2) + 4 * half + (r & 3);
                if (co < a.CoutP) dst[(size_t)co * a.K + k] = acc[i][j][r];
            }
    }
}

// Pointwise layers on 64 x 64 tiles (every ViT layer at batch 4).  In wgrad_split_kernel<64, PW> a thread stages ONE pixel of
// both operands and writes its fp16 halves with sixteen 16-bit LDS stores per step; the loader, not the three MFMAs per wave
// and step, bounded the kernel (a build whose loader only copied pre-split words ran the ViT fc layers in 23.8 instead of
// 33.1 us; its pre-pass launches cost more than that - DESIGN 11.6).  Here the workgroup's halves take one operand each:
// threads 0..127 the dY tile, 128..255 the input tile, a thread the channel quad of TWO adjacent pixels - the pixel-pair
// words of four channels are one ds_write_b128 per half, as in the 128 x 128 form.  Branch-free (the role only selects
// pointers), same words, same MFMA order: the weight AND the bias gradient are bit-identical.
__global__ __launch_bounds__(256) void wgrad_split_pw64_kernel(WgradArgs a) {
    using zs::s16::u32x4;
    constexpr int T = 64, QUADS = T / 4, RS = T + 8;
    __shared__ __attribute__((aligned(16))) unsigned lds[2][2][2][(WP / 2) * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int c0 = blockIdx.x * T, k0 = blockIdx.y * T;
    const int p_begin = blockIdx.z * a.pix_per_split, p_end = min(a.M, p_begin + a.pix_per_split);
    const int role = tid >> 7, prow = (tid & 127) / QUADS, quad = tid & (QUADS - 1);      // role 0: dY, 1: the input
    const int width = role ? a.Cin : a.CoutP, col = (role ? k0 : c0) + 4 * quad;
    const bool col_ok = col < (role ? a.K : a.CoutP);
    const float *base = (role ? a.in : a.dy) + (col_ok ? col : 0);
    unsigned *const my_hi = &lds[0][role][0][prow * RS + 4 * quad];
    constexpr int BUF = 4 * (WP / 2) * RS, HALF = (WP / 2) * RS;
    f32x4 bsum0 = {0.f, 0.f, 0.f, 0.f}, bsum1 = bsum0;       // per pixel row of the step, like the one-pixel loaders: same sums
    struct Pair { f32x4 v0, v1; };
    auto load = [&](int s) -> Pair {
        const int p = p_begin + s * WP + 2 * prow;
        Pair f;
        f.v0 = (col_ok && p < p_end) ? *reinterpret_cast<const f32x4 *>(base + (size_t)p * width) : f32x4{0.f, 0.f, 0.f, 0.f};
        f.v1 = (col_ok && p + 1 < p_end) ? *reinterpret_cast<const f32x4 *>(base + (size_t)(p + 1) * width)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
        return f;
    };
    auto store = [&](int buf, const Pair &f) {
        u32x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            unsigned hh, ll;
            zs::s16::split2(f.v0[e], f.v1[e], hh, ll);
            h[e] = hh; l[e] = ll;
        }
        *reinterpret_cast<u32x4 *>(my_hi + buf * BUF) = h;
        *reinterpret_cast<u32x4 *>(my_hi + buf * BUF + HALF) = l;
        bsum0 += f.v0;                          // (meaningful for role 0 only)
        bsum1 += f.v1;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int wm = (wave & 1) * (T / 2), wn = (wave >> 1) * (T / 2);
    const int steps = (p_end - p_begin + WP - 1) / WP;
    if (steps > 0) {
        Pair f = load(0), g;
        store(0, f);
        if (steps > 1) f = load(1);
        g = f;
        __syncthreads();
        for (int s = 0; s < steps; s++) {
            const int cur = s & 1;
            if (s + 2 < steps) g = load(s + 2);
            u32x4 ah, al, bh, bl;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int oa = (4 * half + t) * RS + wm + l32, ob = (4 * half + t) * RS + wn + l32;
                ah[t] = lds[cur][0][0][oa];
                al[t] = lds[cur][0][1][oa];
                bh[t] = lds[cur][1][0][ob];
                bl[t] = lds[cur][1][1][ob];
            }
            zs::s16::mfma3(acc, ah, al, bh, bl);
            if (s + 1 < steps) store(cur ^ 1, f);
            __syncthreads();
            f = g;
        }
    }
    if (a.bias_partial && blockIdx.y == 0) {          // bias gradient: column sums of this tile's dY rows (fp32)
        __shared__ __attribute__((aligned(16))) float bred[WP][T];
        __syncthreads();
        if (role == 0) {
            *reinterpret_cast<f32x4 *>(&bred[2 * prow][4 * quad]) = bsum0;
            *reinterpret_cast<f32x4 *>(&bred[2 * prow + 1][4 * quad]) = bsum1;
        }
        __syncthreads();
        if (role == 0 && prow == 0 && col_ok) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < WP; r++) t += *reinterpret_cast<const f32x4 *>(&bred[r][4 * quad]);
            *reinterpret_cast<f32x4 *>(a.bias_partial + (size_t)blockIdx.z * a.CoutP + col) = t;
        }
    }
    float *dst = a.partial + (size_t)blockIdx.z * a.CoutP * a.K;
    const int k = k0 + wn + l32;
    if (k < a.K)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = c0 + wm + 8 * (r >> 2) + 4 * half + (r & 3);
            if (co < a.CoutP) dst[(size_t)co * a.K + k] = acc[r];
        }
}

// partial [splits][CoutP][K] (k = tap*CinP + c) -> dw[cout*ld + (cin0+c)*taps + tap], cout < Cout, c < Cin; the bias
// gradient (column sums of dY per split) rides as Cout extra outputs.  An output is summed by ZL adjacent lanes
// (splits strided over them, combined by a fixed butterfly): small weights with hundreds of splits must not be a
// serial walk on one workgroup.  A thread covers all taps of its (cout, c): the taps of one weight row are
// contiguous in dw, so a wave writes one contiguous span instead of every ninth float of nine.
template <int ZL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw,
                                                           int splits, int CoutP, int K, int Cout, int Cin, int CinP,
                                                           int cin0, int ld, int taps, int accumulate,
                                                           const float *__restrict__ bias_partial,
                                                           float *__restrict__ db) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, i = t / ZL, n_w = (size_t)Cout * CinP;
    const int zl = (int)(t % ZL);
    const size_t n_all = n_w + (db ? (size_t)Cout : 0);
    if (i >= n_all) return;                                   // whole ZL groups leave together
    if (i >= n_w) {
        const size_t co = i - n_w;
        float sb = 0.f;
        for (int z = zl; z < splits; z += ZL) sb += bias_partial[(size_t)z * CoutP + co];
#pragma unroll
        for (int o = ZL / 2; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
        if (zl == 0) db[co] = sb;
        return;
    }
    const int c = (int)((unsigned)i % (unsigned)CinP), co = (int)((unsigned)i / (unsigned)CinP);   // Cout * CinP < 2^32
    if (c >= Cin) return;
    float *o = dw + (size_t)co * ld + (size_t)(cin0 + c) * taps;
    const size_t zstride = (size_t)CoutP * K;
    for (int tap = 0; tap < taps; tap++) {
        const float *src = partial + (size_t)co * K + (size_t)tap * CinP + c;
        float s0 = 0.f, s1 = 0.f;
        int z = zl;
        for (; z + ZL < splits; z += 2 * ZL) {
            s0 += src[(size_t)z * zstride];
            s1 += src[(size_t)(z + ZL) * zstride];
        }
        if (z < splits) s0 += src[(size_t)z * zstride];
        float sum = s0 + s1;
#pragma unroll
        for (int off = ZL / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (zl == 0) o[tap] = accumulate ? o[tap] + sum : sum;
    }
}

// ---- StdConv2d weight standardisation: per output channel (w - mean) / sqrt(biased var + eps) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum256(float v, float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__global__ __launch_bounds__(256) void std_weight_kernel(const float *__restrict__ w, float *__restrict__ out, int n,
                                                         float eps) {
    __shared__ float lds[4];
    const float *r = w + (size_t)blockIdx.x * n;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) s += r[e];
    const float mean = block_sum256(s, lds) / n;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) { const float d = r[e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum256(q, lds) / n + eps);
    for (int e = threadIdx.x; e < n; e += 256) out[(size_t)blockIdx.x * n + e] = (r[e] - mean) * rstd;
}

// the same for a device table of weights in ONE launch (the training step re-standardises the 52 StdConv weights of the
// DPT-hybrid backbone after every optimiser step): block b works on global row b, row_prefix[e] <= b < row_prefix[e + 1]
struct StdEntry {               // mirrors include/zeroshape_hip.h zs_std_entry
    const float *w;
    float *out;
    int rows, n;
    float eps;
    int pad;
};
__global__ __launch_bounds__(256) void std_weight_multi_kernel(const StdEntry *__restrict__ tab,
                                                               const int *__restrict__ row_prefix, int n_entries) {
    __shared__ float lds[4];
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (row_prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const StdEntry t = tab[lo];
    const int row = (int)blockIdx.x - row_prefix[lo], n = t.n;
    const float *r = t.w + (size_t)row * n;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) s += r[e];
    const float mean = block_sum256(s, lds) / n;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) { const float d = r[e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum256(q, lds) / n + t.eps);
    for (int e = threadIdx.x; e < n; e += 256) t.out[(size_t)row * n + e] = (r[e] - mean) * rstd;
}

// dw = rstd * (g - mean(g) - what * mean(g * what))
__global__ __launch_bounds__(256) void std_weight_bwd_kernel(const float *__restrict__ w, const float *__restrict__ g,
                                                             float *__restrict__ dw, int n, float eps) {
    __shared__ float lds[4];
    const float *r = w + (size_t)blockIdx.x * n, *gr = g + (size_t)blockIdx.x * n;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) s += r[e];
    const float mean = block_sum256(s, lds) / n;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) { const float d = r[e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum256(q, lds) / n + eps);
    float sg = 0.f, sgx = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) { sg += gr[e]; sgx += gr[e] * (r[e] - mean) * rstd; }
    const float mg = block_sum256(sg, lds) / n, mgx = block_sum256(sgx, lds) / n;
    for (int e = threadIdx.x; e < n; e += 256)
        dw[(size_t)blockIdx.x * n + e] = rstd * (gr[e] - mg - (r[e] - mean) * rstd * mgx);
}

// ---- data gradient of a convolution with <= 4 input channels (the network stems) ----
// As a GEMM this has N = 3 columns in a 128-wide tile, and at stride 2 three quarters of the
// zero-stuffed taps are padding: 2 % useful work.  Direct form instead: one thread per input pixel
// gathers the (at most ceil(k/s)^2) output pixels that read it; weights sit in LDS as
// [tap][cout][4 channels] (every lane of a wave that needs an entry reads the same address).
__global__ __launch_bounds__(256) void dgrad_small_cin_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                              float *__restrict__ dx, int B, int H, int W, int CinP,
                                                              int Ho, int Wo, int Cout, int kh, int kw, int stride,
                                                              int pad_t, int pad_l, int Cin, int cin0, int ld,
                                                              float scale) {
    extern __shared__ f32x4 wl[];                    // [taps][Cout]
    const int taps = kh * kw;
    for (int e = threadIdx.x; e < taps * Cout; e += 256) {
        const int tap = e / Cout, co = e - tap * Cout;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < Cin; c++) v[c] = w[(size_t)co * ld + (size_t)(cin0 + c) * taps + tap];
        wl[e] = v;
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)B * H * W;
    if (i >= total) return;
    const int ix = i % W, iy = (i / W) % H, b = i / W / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < kh; ky++) {
        const int vy = iy + pad_t - ky;
        if (vy < 0 || vy % stride) continue;
        const int oy = vy / stride;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < kw; kx++) {
            const int vx = ix + pad_l - kx;
            if (vx < 0 || vx % stride) continue;
            const int ox = vx / stride;
            if (ox >= Wo) continue;
            const f32x4 *g = reinterpret_cast<const f32x4 *>(dy + (((size_t)b * Ho + oy) * Wo + ox) * Cout);
            const f32x4 *wt = wl + (ky * kw + kx) * Cout;
            for (int q = 0; q < Cout / 4; q++) {
                const f32x4 gv = g[q];
#pragma unroll
                for (int e = 0; e < 4; e++) acc += wt[4 * q + e] * gv[e];
            }
        }
    }
    f32x4 out = acc * scale;
    for (int c = Cin; c < 4; c++) out[c] = 0.f;
    float *o = dx + i * CinP;
    for (int c = 0; c < CinP; c++) o[c] = out[c];
}

}  // namespace

#define ZS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            zs::set_err(__VA_ARGS__);    \
            return 0;                    \
        }                                \
    } while (0)

extern "C" int zs_pack_conv_weight(const float *w, float *packed, int Cout, int Cin, int cin0, int CinTot, int kh,
                                   int kw, int dgrad, void *stream) {
    ZS_REQUIRE(Cout > 0 && Cin > 0 && cin0 >= 0 && cin0 + Cin <= CinTot && kh > 0 && kw > 0,
               "zs_pack_conv_weight: bad geometry (Cout=%d Cin=%d cin0=%d CinTot=%d k %dx%d)", Cout, Cin, cin0, CinTot,
               kh, kw);
    ZS_REQUIRE(w && packed, "zs_pack_conv_weight: null pointer");
    const int taps = kh * kw;
    const int Kc = dgrad ? (Cout + 3) / 4 * 4 : (Cin + 3) / 4 * 4, N = dgrad ? Cin : Cout;
    const int K16 = (taps * Kc + 15) / 16 * 16, NPad = (N + 127) / 128 * 128;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks_for((size_t)K16 * NPad)), dim3(256), 0, S(stream), w, packed,
                       Cout, Cin, cin0, CinTot * taps, taps, dgrad ? 1 : 0, K16, NPad);
    return zs::check_launch("zs_pack_conv_weight") ? 1 : 0;
}

static_assert(sizeof(PackEntry) == sizeof(zs_pack_entry), "PackEntry must mirror zs_pack_entry");

extern "C" int zs_pack_chunk_elems(void) { return PACK_CHUNK; }

extern "C" int zs_pack_entry_chunks(int Cout, int Cin, int taps, int dgrad, int K16, int NPad) {
    ZS_REQUIRE(Cout > 0 && Cin > 0 && taps > 0 && K16 > 0 && NPad > 0 && NPad % 128 == 0,
               "zs_pack_entry_chunks: bad operand (Cout=%d Cin=%d taps=%d K16=%d NPad=%d)", Cout, Cin, taps, K16, NPad);
    return pack_entry_chunks(Cout, Cin, taps, dgrad, K16, NPad);
}

extern "C" int zs_pack_entry_inline_split(int Cout, int Cin, int taps, int dgrad) {
    return pack_inline_split(Cout, Cin, taps, dgrad) ? 1 : 0;
}

extern "C" int zs_pack_conv_weight_multi_split(const zs_pack_entry *table, const int *chunk_entry,
                                               const unsigned long long *chunk_start, int n_chunks, float *const *split_dst,
                                               int split_only, void *stream) {
    ZS_REQUIRE(n_chunks >= 0, "zs_pack_conv_weight_multi: bad arguments");
    if (n_chunks == 0) return 1;
    ZS_REQUIRE(table && chunk_entry && chunk_start, "zs_pack_conv_weight_multi: null pointer");
    hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(n_chunks), dim3(256), PACK_LDS_FLOATS * sizeof(float), S(stream),
                       reinterpret_cast<const PackEntry *>(table), chunk_entry, chunk_start, split_dst,
                       split_dst && split_only ? 1 : 0);
    return zs::check_launch("zs_pack_conv_weight_multi") ? 1 : 0;
}

extern "C" int zs_pack_conv_weight_multi(const zs_pack_entry *table, const int *chunk_entry,
                                         const unsigned long long *chunk_start, int n_chunks, void *stream) {
    return zs_pack_conv_weight_multi_split(table, chunk_entry, chunk_start, n_chunks, nullptr, 0, stream);
}

// tile edge and number of pixel-range splits of one weight gradient
static void wgrad_plan(long long M, int CoutP, int K, int *tile, int *splits_out) {
    static const long long target = getenv("ZS_WGRAD_TARGET") ? atoll(getenv("ZS_WGRAD_TARGET")) : 768;
    const long long tiles128 = (long long)((CoutP + 127) / 128) * ((K + 127) / 128);
    const long long tiles64 = (long long)((CoutP + 63) / 64) * ((K + 63) / 64);
    // big weights over few pixels, or tiny weights over many: more, smaller tiles instead of slivers of
    // the pixel range (every split costs a partial tile to write and to reduce); measured per shape
    const bool small_tiles = (M < 4096 && tiles128 < 192) || tiles128 <= 12;
    const long long tiles = small_tiles ? tiles64 : tiles128;
    static const long long target_small = getenv("ZS_WGRAD_TARGET_SMALL") ? atoll(getenv("ZS_WGRAD_TARGET_SMALL")) : 1024;
    long long splits = ((small_tiles ? target_small : target) + tiles - 1) / tiles;     // aim at 2-3 workgroups per CU
    const long long max_by_pixels = (M + 127) / 128;                           // at least 128 pixels per split
    if (splits > max_by_pixels) splits = max_by_pixels;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    *tile = small_tiles ? 64 : 128;
    *splits_out = (int)splits;
}

extern "C" size_t zs_conv2d_wgrad_workspace_bytes(int batch, int Hout, int Wout, int Cin, int Cout, int kh, int kw) {
    const int CoutP = (Cout + 3) / 4 * 4, K = kh * kw * ((Cin + 3) / 4 * 4);
    const long long M = (long long)batch * Hout * Wout;
    int tile, splits;
    wgrad_plan(M, CoutP, K, &tile, &splits);
    return (size_t)splits * CoutP * (K + 1) * sizeof(float);
}

extern "C" int zs_conv2d_wgrad(const float *in, const float *dy, float *dw, float *db, void *workspace, int batch, int Hin,
                               int Win, int CinP, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t,
                               int pad_l, int flags, float in_scale, float in_shift, int Cin, int cin0, int CinTot,
                               int accumulate, void *stream) {
    ZS_REQUIRE(batch >= 0 && Hin > 0 && Win > 0 && CinP > 0 && (CinP & 3) == 0 && Cin > 0 && Cin <= CinP &&
                   Hout > 0 && Wout > 0 && Cout > 0 && kh > 0 && kw > 0 && stride > 0 && cin0 >= 0 &&
                   cin0 + Cin <= CinTot,
               "zs_conv2d_wgrad: bad geometry (B=%d in %dx%dx%d out %dx%dx%d k %dx%d s %d; Cin %d at %d of %d)", batch,
               Hin, Win, CinP, Hout, Wout, Cout, kh, kw, stride, Cin, cin0, CinTot);
    if (batch == 0) return 1;
    ZS_REQUIRE(in && dy && dw && workspace, "zs_conv2d_wgrad: null pointer");
    const long long M = (long long)batch * Hout * Wout;
    ZS_REQUIRE(M <= (1LL << 30), "zs_conv2d_wgrad: %lld output pixels", M);
    WgradArgs a;
    a.in = in; a.dy = dy; a.partial = static_cast<float *>(workspace);
    a.B = batch; a.Hin = Hin; a.Win = Win; a.Cin = CinP; a.Hout = Hout; a.Wout = Wout;
    a.CoutP = (Cout + 3) / 4 * 4;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
    a.K = kh * kw * CinP; a.M = (int)M;
    a.in_relu = (flags & ZS_CONV_IN_RELU) ? 1 : 0;
    a.in_scale = in_scale; a.in_shift = in_shift;
    int tile, splits;
    wgrad_plan(M, a.CoutP, a.K, &tile, &splits);
    a.bias_partial = db ? a.partial + (size_t)splits * a.CoutP * a.K : nullptr;
    a.pix_per_split = (int)((M + splits - 1) / splits);
    a.pix_per_split = (a.pix_per_split + WP - 1) / WP * WP;
    const dim3 grid((a.CoutP + tile - 1) / tile, (a.K + tile - 1) / tile, splits);
    const bool plain = !a.in_relu && in_scale == 1.0f && in_shift == 0.0f;
    const bool pw = kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && plain && Hin == Hout && Win == Wout;
    const bool fast = !pw && plain && Wout >= 6 && Hout >= 2 && (long long)batch * Hin * Win * CinP < (1LL << 31);
    const int mode = pw ? 1 : (fast ? 2 : 0);
#define ZS_WGRAD(KERNEL, TT)                                                                                  \
    do {                                                                                                      \
        if (mode == 1) hipLaunchKernelGGL((KERNEL<TT, 1>), grid, dim3(256), 0, S(stream), a);                  \
        else if (mode == 2) hipLaunchKernelGGL((KERNEL<TT, 2>), grid, dim3(256), 0, S(stream), a);             \
        else hipLaunchKernelGGL((KERNEL<TT, 0>), grid, dim3(256), 0, S(stream), a);                            \
    } while (0)
    static const bool no_pw64 = getenv("ZS_WGRAD_NO_PW64") != nullptr;        // A/B switch
    if (flags & ZS_CONV_F16X3) {                 // split-fp16 arithmetic (optim.amp)
        if (tile == 64 && mode == 1 && !no_pw64) hipLaunchKernelGGL(wgrad_split_pw64_kernel, grid, dim3(256), 0, S(stream), a);
        else if (tile == 64) ZS_WGRAD(wgrad_split_kernel, 64);
        else ZS_WGRAD(wgrad_split_kernel, 128);
    } else {
        if (tile == 64) ZS_WGRAD(wgrad_kernel, 64);
        else ZS_WGRAD(wgrad_kernel, 128);
    }
#undef ZS_WGRAD
    if (!zs::check_launch("zs_conv2d_wgrad")) return 0;
    {
        const size_t outs = (size_t)Cout * CinP + (db ? Cout : 0);
        int zl = 1;                                            // lanes per output: enough threads for a few waves per CU
        while (zl < 64 && 2 * zl <= splits && outs * zl < (size_t)(1 << 17)) zl *= 2;
#define ZS_WRED(Z)                                                                                                     \
        hipLaunchKernelGGL((wgrad_reduce_kernel<Z>), dim3(blocks_for(outs * Z)), dim3(256), 0, S(stream), a.partial, dw,    \
                           splits, a.CoutP, a.K, Cout, Cin, CinP, cin0, CinTot * kh * kw, kh * kw, accumulate ? 1 : 0,      \
                           a.bias_partial, db)
        switch (zl) {
            case 1: ZS_WRED(1); break;
            case 2: ZS_WRED(2); break;
            case 4: ZS_WRED(4); break;
            case 8: ZS_WRED(8); break;
            case 16: ZS_WRED(16); break;
            case 32: ZS_WRED(32); break;
            default: ZS_WRED(64); break;
        }
#undef ZS_WRED
    }
    return zs::check_launch("zs_conv2d_wgrad(reduce)") ? 1 : 0;
}

extern "C" int zs_standardize_weight(const float *w, float *out, int Cout, int n, float eps, void *stream) {
    ZS_REQUIRE(Cout > 0 && n > 0 && w && out, "zs_standardize_weight: bad arguments");
    hipLaunchKernelGGL(std_weight_kernel, dim3(Cout), dim3(256), 0, S(stream), w, out, n, eps);
    return zs::check_launch("zs_standardize_weight") ? 1 : 0;
}

extern "C" int zs_standardize_weight_multi(const zs_std_entry *table, const int *row_prefix, int n_entries, int total_rows,
                                           void *stream) {
    static_assert(sizeof(StdEntry) == sizeof(zs_std_entry), "zs_std_entry layout");
    ZS_REQUIRE(n_entries >= 0 && total_rows >= 0, "zs_standardize_weight_multi: bad arguments");
    if (n_entries == 0 || total_rows == 0) return 1;
    ZS_REQUIRE(table && row_prefix, "zs_standardize_weight_multi: null pointer");
    hipLaunchKernelGGL(std_weight_multi_kernel, dim3(total_rows), dim3(256), 0, S(stream),
                       reinterpret_cast<const StdEntry *>(table), row_prefix, n_entries);
    return zs::check_launch("zs_standardize_weight_multi") ? 1 : 0;
}

extern "C" int zs_standardize_weight_bwd(const float *w, const float *grad_out, float *dw, int Cout, int n, float eps,
                                         void *stream) {
    ZS_REQUIRE(Cout > 0 && n > 0 && w && grad_out && dw, "zs_standardize_weight_bwd: bad arguments");
    hipLaunchKernelGGL(std_weight_bwd_kernel, dim3(Cout), dim3(256), 0, S(stream), w, grad_out, dw, n, eps);
    return zs::check_launch("zs_standardize_weight_bwd") ? 1 : 0;
}

extern "C" int zs_conv2d_dgrad_small_cin(const float *dy, const float *w, float *dx, int batch, int H, int W, int CinP,
                                         int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t, int pad_l,
                                         int Cin, int cin0, int CinTot, float scale, void *stream) {
    ZS_REQUIRE(batch >= 0 && H > 0 && W > 0 && CinP > 0 && CinP <= 4 && Cin > 0 && Cin <= CinP && Hout > 0 && Wout > 0 &&
                   Cout > 0 && (Cout & 3) == 0 && kh > 0 && kw > 0 && stride > 0 && cin0 >= 0 && cin0 + Cin <= CinTot &&
                   (size_t)kh * kw * Cout * 16 <= 160 * 1024,
               "zs_conv2d_dgrad_small_cin: bad geometry (B=%d in %dx%dx%d out %dx%dx%d k %dx%d s %d; Cin <= 4, "
               "Cout %% 4 == 0, taps*Cout*16 B of LDS)", batch, H, W, CinP, Hout, Wout, Cout, kh, kw, stride);
    if (batch == 0) return 1;
    ZS_REQUIRE(dy && w && dx, "zs_conv2d_dgrad_small_cin: null pointer");
    const size_t lds = (size_t)kh * kw * Cout * sizeof(f32x4);
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(dgrad_small_cin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(dgrad_small_cin_kernel, dim3(blocks_for((size_t)batch * H * W)), dim3(256), lds, S(stream), dy, w, dx,
                       batch, H, W, CinP, Hout, Wout, Cout, kh, kw, stride, pad_t, pad_l, Cin, cin0, CinTot * kh * kw, scale);
    return zs::check_launch("zs_conv2d_dgrad_small_cin") ? 1 : 0;
}
