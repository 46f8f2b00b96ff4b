// s_memtime tick rate vs s_memrealtime (100 MHz) under idle and under load
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void cal(unsigned long long *out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = fmaf(x, 1.0001f, 0.5f);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 64);
    for (int blocks : {1, 256, 2048}) for (int iters : {1000, 100000, 3000000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(cal, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("event %.3f ms | ", ms);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("blocks %d iters %d: memtime %llu realtime %llu -> %.1f MHz; %.2f ticks/iter\n", blocks, iters, h[0], h[1], h[0] * 100.0 / h[1], (double)h[0] / iters);
    }
    return 0;
}
