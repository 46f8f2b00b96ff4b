// Store throughput of ONE workgroup per CU on gfx950: 8 waves each writing 32 x 1 KiB (a 256 x 256 fp32 tile = 256 KiB per
// workgroup, the epilogue of csrc/nn_conv_pp256.h) as flat_store / global_store, dwordx4 / dwordx2 / dword, linear (1 KiB
// contiguous per wave instruction) or tile-shaped (32 rows x 32 B per instruction, row stride 12 KiB), with and without `nt`.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_rate.hip -o store_rate && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE: 0 flat x4, 1 global x4, 2 global x4 nt, 3 global x2 (twice as many), 4 global x1, 5 flat x4 + waitcnt after each,
//       6 global x4 sc0 sc1, 7 flat x4 nt
template <int MODE, bool TILE>
__global__ __launch_bounds__(512) void k(float *out, int reps, int ld, unsigned long long *cyc) {
    __shared__ float hog[130 * 256];                                 // one workgroup per CU, like the GEMM kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    hog[tid] = (float)tid;
    __syncthreads();
    f32x4 v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = f32x4{hog[(tid + i) & 511], 1.f, 2.f, (float)i};
    float *base = out + (size_t)blockIdx.x * (TILE ? 256 : 65536);   // TILE: 256 columns of a row-major [256 * ?][ld] matrix
    if (TILE) base = out + (size_t)(blockIdx.x / (ld / 256)) * 256 * ld + (size_t)(blockIdx.x % (ld / 256)) * 256;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            float *p;
            if (TILE) {              // wave: rows (wave & 3) * 64 + 32 (i >> 4) + l32, columns (wave >> 2) * 128 + 8 (i & 15) + 4 half
                p = base + (size_t)((wave & 3) * 64 + 32 * (i >> 4) + l32) * ld + (wave >> 2) * 128 + 8 * (i & 15) + 4 * half;
            } else {
                p = base + (wave * 32 + i) * 256 + lane * 4;
            }
            if (MODE == 0) asm volatile("flat_store_dwordx4 %0, %1" : : "v"(p), "v"(v[i]) : "memory");
            if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v[i]) : "memory");
            if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v[i]) : "memory");
            if (MODE == 3) {
                asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(f32x2{v[i].x, v[i].y}) : "memory");
                asm volatile("global_store_dwordx2 %0, %1, off offset:8" : : "v"(p), "v"(f32x2{v[i].z, v[i].w}) : "memory");
            }
            if (MODE == 4) {
                asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v[i].x) : "memory");
                asm volatile("global_store_dword %0, %1, off offset:4" : : "v"(p), "v"(v[i].y) : "memory");
                asm volatile("global_store_dword %0, %1, off offset:8" : : "v"(p), "v"(v[i].z) : "memory");
                asm volatile("global_store_dword %0, %1, off offset:12" : : "v"(p), "v"(v[i].w) : "memory");
            }
            if (MODE == 5) asm volatile("flat_store_dwordx4 %0, %1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : : "v"(p), "v"(v[i]) : "memory");
            if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v[i]) : "memory");
            if (MODE == 7) asm volatile("flat_store_dwordx4 %0, %1 nt" : : "v"(p), "v"(v[i]) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, bool TILE>
void run(const char *name, float *d, unsigned long long *cyc, int blocks) {
    const int reps = 4, ld = 3072;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, TILE>), dim3(blocks), dim3(512), 0, 0, d, 1, ld, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, TILE>), dim3(blocks), dim3(512), 0, 0, d, reps, ld, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= blocks;
    printf("%-26s %-6s %3d wgs: %8.1f us  %9.0f cycles per 256 KiB tile  = %5.1f B/clk/CU  (%6.2f TB/s chip)\n", name,
           TILE ? "tile" : "linear", blocks, ms * 1e3, mean / reps, 262144.0 * reps / mean, blocks * 262144.0 * reps / ms / 1e9);
}

int main() {
    float *d;
    unsigned long long *cyc;
    hipMalloc(&d, (size_t)256 * 65536 * 4 * 2);
    hipMalloc(&cyc, 256 * 8);
    hipMemset(d, 0, (size_t)256 * 65536 * 4 * 2);
    for (int blocks : {32, 256}) {
        run<0, false>("flat x4", d, cyc, blocks);
        run<0, true>("flat x4", d, cyc, blocks);
        run<1, false>("global x4", d, cyc, blocks);
        run<1, true>("global x4", d, cyc, blocks);
        run<2, true>("global x4 nt", d, cyc, blocks);
        run<7, true>("flat x4 nt", d, cyc, blocks);
        run<6, true>("global x4 sc0 sc1", d, cyc, blocks);
        run<3, true>("global x2", d, cyc, blocks);
        run<4, true>("global x1", d, cyc, blocks);
        run<5, true>("flat x4 + wait each", d, cyc, blocks);
    }
    return 0;
}
