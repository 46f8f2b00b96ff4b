// Micro-benchmark (round 2): WHERE must the VALU side work of the split-fp16 decoder sit relative to
// the three MFMAs of a K-block?  One wave per SIMD issues in order: an MFMA that waits for its
// predecessor's accumulator blocks everything behind it.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_valu_place mfma_valu_place.hip
// PLACE 0: MFMA MFMA MFMA | 3V VALU          (what hipcc emitted for round 1's source order)
//       1: MFMA V MFMA V MFMA V              (one chain, VALU in the dependency gaps)
// ACCS  1: one accumulator (dependent chain) | 3: three accumulators in rotation
// V = independent v_fma_f32 per gap; T = of which v_exp_f32 (transcendental)
// Register operands only (no LDS, no barrier): isolates the issue behaviour.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)
constexpr int ITER = 4096;

template <int V, int T, int TPOS = 0>
__device__ __forceinline__ void valu(float (&x)[24], int base) {
#pragma unroll
    for (int j = 0; j < V; j++) {
        float &r = x[(base + j) % 24];
        const bool tr = TPOS == 0 ? j < T : TPOS == 1 ? j >= V - T : (j >= (V - T) / 2 && j < (V - T) / 2 + T);
        if (tr) r = __builtin_amdgcn_exp2f(r);
        else r = fmaf(r, 0.999f, 0.25f);
    }
}

template <int PLACE, int ACCS, int V, int T, int TPOS = 0>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *w, float *out, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    f32x16 c0, c1, c2;
    for (int r = 0; r < 16; r++) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; }
    u32x4 a = w[lane], b = w[64 + lane], a2 = w[128 + lane], b2 = w[192 + lane];
    float x[24];
    for (int j = 0; j < 24; j++) x[j] = lane * 0.01f + j * 0.1f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++) {
        if (PLACE == 0) {
            c0 = MF(a2, b, c0);
            if (ACCS == 3) { c1 = MF(a, b2, c1); c2 = MF(a, b, c2); } else { c0 = MF(a, b2, c0); c0 = MF(a, b, c0); }
            __builtin_amdgcn_sched_barrier(0);
            valu<3 * V, 3 * T>(x, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (PLACE == 2) {   // plain VALU in the first two gaps, every transcendental behind the third MFMA
            c0 = MF(a2, b, c0);
            __builtin_amdgcn_sched_barrier(0);
            valu<V, 0>(x, 0);
            __builtin_amdgcn_sched_barrier(0);
            c0 = MF(a, b2, c0);
            __builtin_amdgcn_sched_barrier(0);
            valu<V, 0>(x, 8);
            __builtin_amdgcn_sched_barrier(0);
            c0 = MF(a, b, c0);
            __builtin_amdgcn_sched_barrier(0);
            valu<V - 3 * T + 3 * T, 3 * T, TPOS>(x, 16);   // V plain-equivalents: (V - 3T) fma + 3T exp ... see main
            __builtin_amdgcn_sched_barrier(0);
        } else {
            c0 = MF(a2, b, c0);
            __builtin_amdgcn_sched_barrier(0);
            valu<V, T, TPOS>(x, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ACCS == 3) c1 = MF(a, b2, c1); else c0 = MF(a, b2, c0);
            __builtin_amdgcn_sched_barrier(0);
            valu<V, T, TPOS>(x, 8);
            __builtin_amdgcn_sched_barrier(0);
            if (ACCS == 3) c2 = MF(a, b, c2); else c0 = MF(a, b, c0);
            __builtin_amdgcn_sched_barrier(0);
            valu<V, T, TPOS>(x, 16);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r];
    for (int j = 0; j < 24; j++) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int PLACE, int ACCS, int V, int T, int TPOS = 0>
void run(const u32x4 *w, float *out, unsigned long long *cyc) {
    hipLaunchKernelGGL((k<PLACE, ACCS, V, T, TPOS>), dim3(256), dim3(256), 0, 0, w, out, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<PLACE, ACCS, V, T, TPOS>), dim3(256), dim3(256), 0, 0, w, out, cyc);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s, %d acc, %d VALU per gap (%d transcendental, tpos %d): %7.1f cycles per K-block (3 MFMAs, ideal 96)\n",
           PLACE == 2 ? "MFMA V MFMA V MFMA V+T" : PLACE ? "MFMA V MFMA V MFMA V" : "MFMA MFMA MFMA | 3V ", ACCS, V, T, TPOS, (double)c / ITER);
    fflush(stdout);
}

template <int V, int T>
void all(const u32x4 *w, float *out, unsigned long long *cyc) {
    run<0, 1, V, T>(w, out, cyc);
    run<1, 1, V, T>(w, out, cyc);
    run<0, 3, V, T>(w, out, cyc);
    run<1, 3, V, T>(w, out, cyc);
}

int main() {
    u32x4 *w; float *out; unsigned long long *cyc;
    hipMalloc(&w, 1 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    unsigned *h = (unsigned *)malloc(1 << 20);
    for (int i = 0; i < (1 << 18); i++) h[i] = 0x3c003c00u + (i & 0xff);
    hipMemcpy(w, h, 1 << 20, hipMemcpyHostToDevice);
    if (getenv("ZS_UB_TRANS")) {
        run<1, 1, 4, 1, 0>(w, out, cyc);   // exp first in each gap
        run<1, 1, 4, 1, 1>(w, out, cyc);   // exp last in each gap
        run<1, 1, 4, 1, 2>(w, out, cyc);   // exp in the middle
        run<2, 1, 4, 1, 0>(w, out, cyc);   // 4 fma | 4 fma | 3 exp + 4 fma (exp first)
        run<2, 1, 4, 1, 1>(w, out, cyc);   // ... exp last
        run<1, 1, 5, 1, 2>(w, out, cyc);
        run<1, 1, 3, 1, 2>(w, out, cyc);
        run<1, 1, 2, 1, 1>(w, out, cyc);
        run<1, 1, 1, 1, 0>(w, out, cyc);   // one exp per gap alone
        run<0, 1, 1, 1, 0>(w, out, cyc);
        return 0;
    }
    all<0, 0>(w, out, cyc);
    all<2, 0>(w, out, cyc);
    all<4, 0>(w, out, cyc);
    all<4, 1>(w, out, cyc);
    all<6, 0>(w, out, cyc);
    all<6, 2>(w, out, cyc);
    all<8, 0>(w, out, cyc);
    return 0;
}
