// Issue rate of packed fp32 VALU ops on gfx950: independent chains of v_fma_f32 / v_pk_fma_f32 with VGPR and with SGPR-pair
// operands, 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int N = 4096, U = 8;

template <int MODE>
__global__ void k(float *out, const float *sc, int iters) {
    f32x2 acc[U];
    for (int u = 0; u < U; u++) acc[u] = f32x2{(float)threadIdx.x, (float)u};
    const float a = out[threadIdx.x & 7];
    f32x2 s2;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s2) : "s"(sc) : "memory");
    const f32x2 a2 = {a, a};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (MODE == 0) {            // scalar fma, two per pair of lanes-ops (same flops as one packed)
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[u].x) : "v"(a));
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[u].y) : "v"(a));
            } else if (MODE == 1) {     // packed fma, VGPR operands
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[u]) : "v"(a2));
            } else if (MODE == 2) {     // packed add with an SGPR pair + broadcast VGPR (the pose kernel's form)
                asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[1,0]" : "+v"(acc[u]) : "s"(s2));
            } else if (MODE == 3) {     // scalar sub with SGPR
                asm volatile("v_sub_f32 %0, %1, %0" : "+v"(acc[u].x) : "s"(s2.x));
                asm volatile("v_sub_f32 %0, %1, %0" : "+v"(acc[u].y) : "s"(s2.y));
            } else if (MODE == 4) {     // packed mul VGPR
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[u]) : "v"(a2));
            }
        }
    }
    float r = 0;
    for (int u = 0; u < U; u++) r += acc[u].x + acc[u].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, float *d, float *sc) {
    for (int wps : {1, 2, 4, 8}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = 256 * wps;             // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sc, 16);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sc, N);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double lane_ops = (double)blocks * 256 * N * U * 2;         // two f32 results per (lane, slot, iteration)
        printf("%-34s %d waves/SIMD: %7.3f ms  %6.1f G lane-results/s  = %5.2f results/clk/SIMD at 2.4 GHz\n", name, wps, ms,
               lane_ops / ms / 1e6, lane_ops / ms / 1e6 / 1024 / 2.4);
    }
}

int main() {
    float *d, *sc;
    hipMalloc(&d, 256 * 8 * 256 * 4 + 64);
    hipMalloc(&sc, 64);
    hipMemset(d, 0, 256 * 8 * 256 * 4 + 64);
    hipMemset(sc, 0, 64);
    run<0>("2 x v_fma_f32", d, sc);
    run<1>("v_pk_fma_f32 (VGPR)", d, sc);
    run<2>("v_pk_add_f32 (SGPR pair, op_sel)", d, sc);
    run<3>("2 x v_sub_f32 (SGPR)", d, sc);
    run<4>("v_pk_mul_f32 (VGPR)", d, sc);
    return 0;
}
