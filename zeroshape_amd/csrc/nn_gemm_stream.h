// Launch interface of the batch-1 pointwise GEMM kernel (csrc/nn_gemm_stream.hip), used by zs_conv2d_nhwc's small-problem path.
#pragma once
#include <hip/hip_runtime.h>

namespace zs {
namespace stream_gemm {

struct Args {
    const float *a;                 // [M][lda] fp32 rows (channels-last activations of a pointwise layer)
    const float *w;                 // pre-split packed weights [K16/4][CoutPad][4] (zs_conv2d_presplit_weight)
    const float *scale, *shift;     // [N] or null
    const float *res1, *res2;       // [M][N] or null
    float *out;                     // [M][N]
    int M, K, N, CoutPad, lda, act, in_relu;
    // LayerNorm of the input rows from a producer's row statistics [M][in_tiles][2] (null: none)
    const float *in_stats;
    int in_tiles;
    float in_eps;
    // row statistics of the output [M][ceil(N / cols)][2] (null: none); `stats_cols` = 32 or 64 fixes the tile width
    float *out_stats;
    int stats_cols;
    // K split across workgroups: partial tiles + one ticket per tile (zero between launches); null: no split
    float *parts;
    int *tickets;
    size_t parts_bytes;
    int max_tickets;
    // two-launch K split: with `two_launch_max` > 1 the launcher may split a long contraction into up to that many ranges whose
    // raw partial sums go to parts[z][M][N]; it returns the number of ranges in *ranges (1: complete output written) and the
    // CALLER runs the reduction + epilogue (nn_conv.hip: conv_splitk_reduce[_stats]_kernel)
    int two_launch_max;
    int *ranges;
    // K16-major activations (round 6; the ViT MLP's hidden tensor at batch 1): [K / 16][M][16] instead of [M][K] - the 32 rows of a
    // tile's K = 16 step are 2 KiB of consecutive bytes instead of 32 pieces a row pitch apart.  a_k16: `a` is laid out that way
    // (lda is ignored; no in_stats); out_k16: `out` is written that way ([N / 16][M][16]; N % 16 == 0, no residuals, no out_stats).
    // Neither combines with a K split.
    int a_k16, out_k16;
};

// true when the kernel takes the problem (and has launched it); false: the caller uses another kernel
// dry_run: decide only (does the kernel take this problem as described?), launch nothing
bool launch(const Args &a, hipStream_t stream, bool dry_run = false);

}  // namespace stream_gemm
}  // namespace zs
