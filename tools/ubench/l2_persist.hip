// Does data a kernel pulled into an XCD's L2 survive the kernel boundary?  (Round 6: would a "prefetch the NEXT layer's
// weights" hook in the batch-1 GEMM kernels pay?  Their launches stream cold weights at ~30 B/clk/CU against 64 B/clk on hits.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o l2_persist l2_persist.hip ; rocprofv3 --kernel-trace --stats -- ./l2_persist
// Workgroup b runs on XCD b % 8 and reads slice (b >> 3) of region (b % 8 + shift) % 8 of a buffer.  Per buffer, back to back
// on one stream:  rd_cold (first touch: HBM) -> rd_same (same mapping: that XCD's L2, if it survived) -> rd_shift (regions
// rotated by one XCD: another L2 - a hit can only come from the memory-side Infinity Cache).  40 buffers of 8 MB cycle, so
// every first touch is cold for the 256 MiB Infinity Cache too.  Three names = three rows in the kernel trace.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void body(const u32x4 *buf, size_t per_xcd, int wgs_per_xcd, int shift, u32x4 *sink) {
    const int b = blockIdx.x, xcd = (b + shift) & 7, local = b >> 3;
    const size_t n = per_xcd / wgs_per_xcd;
    const u32x4 *p = buf + (size_t)xcd * per_xcd + (size_t)local * n;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < n; i += 256) acc ^= p[i];
    if (acc.x == 0x12345u && acc.y == 0x777u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void rd_cold(const u32x4 *buf, size_t per_xcd, int w, u32x4 *sink) { body(buf, per_xcd, w, 0, sink); }
__global__ __launch_bounds__(256) void rd_same(const u32x4 *buf, size_t per_xcd, int w, u32x4 *sink) { body(buf, per_xcd, w, 0, sink); }
__global__ __launch_bounds__(256) void rd_shift(const u32x4 *buf, size_t per_xcd, int w, u32x4 *sink) { body(buf, per_xcd, w, 1, sink); }
__global__ __launch_bounds__(256) void rd_later(const u32x4 *buf, size_t per_xcd, int w, u32x4 *sink) { body(buf, per_xcd, w, 0, sink); }

int main(int argc, char **argv) {
    const int wgs_per_xcd = argc > 1 ? atoi(argv[1]) : 28;
    const size_t per_xcd_bytes = (argc > 2 ? atoi(argv[2]) : 1024) * 1024ull;
    const int NBUF = 40, reps = 3;
    const size_t per_xcd = per_xcd_bytes / 16 / wgs_per_xcd * wgs_per_xcd;
    u32x4 *bufs[NBUF], *sink;
    CK(hipMalloc(&sink, 64));
    for (int i = 0; i < NBUF; i++) {
        CK(hipMalloc(&bufs[i], per_xcd * 16 * 8));
        CK(hipMemset(bufs[i], i + 1, per_xcd * 16 * 8));
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(hipDeviceSynchronize());
    const dim3 grid(8 * wgs_per_xcd), block(256);
    for (int r = 0; r < reps; r++)
        for (int i = 0; i < NBUF; i++) {
            hipLaunchKernelGGL(rd_cold, grid, block, 0, st, bufs[i], per_xcd, wgs_per_xcd, sink);
            hipLaunchKernelGGL(rd_same, grid, block, 0, st, bufs[i], per_xcd, wgs_per_xcd, sink);
            hipLaunchKernelGGL(rd_shift, grid, block, 0, st, bufs[i], per_xcd, wgs_per_xcd, sink);
            // ... and buffer i - 2 once more with its own mapping: three other 8 MB buffers went through the L2s since
            hipLaunchKernelGGL(rd_later, grid, block, 0, st, bufs[(i + NBUF - 2) % NBUF], per_xcd, wgs_per_xcd, sink);
        }
    CK(hipStreamSynchronize(st));
    printf("l2_persist: %d workgroups per XCD, %zu KB per XCD and buffer, %d buffers x %d rounds\n", wgs_per_xcd, per_xcd * 16 / 1024, NBUF, reps);
    return 0;
}
