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
};

// true when the kernel takes the problem (and has launched it); false: the caller uses another kernel
bool launch(const Args &a, hipStream_t stream);

}  // namespace stream_gemm
}  // namespace zs
