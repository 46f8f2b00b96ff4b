// Micro-benchmark: what MFMA rate can one wave per SIMD sustain with the decoder's loop
// shapes?  hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
constexpr int ITER = 20000;  // groups of 4 MFMAs per acc-variant

template <int V>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(240))) void k(const f32x4 *w, float *out) {
    __shared__ f32x4 lds[64 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) lds[i] = w[i];
    __syncthreads();
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; r++) { acc0[r] = r; acc1[r] = r + 1; acc2[r] = r + 2; acc3[r] = r + 3; }
    float a = lane * 0.001f, b = lane * 0.002f;
    if (V == 0) {  // one dependent chain, register operands
        for (int i = 0; i < ITER; i++) {
            acc0 = MF(a, b, acc0); acc0 = MF(b, a, acc0); acc0 = MF(a, a, acc0); acc0 = MF(b, b, acc0);
        }
    } else if (V == 1) {  // 4 independent accumulators
        for (int i = 0; i < ITER / 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc0 = MF(a, b, acc0); acc1 = MF(b, a, acc1); acc2 = MF(a, a, acc2); acc3 = MF(b, b, acc3);
            }
        }
    } else if (V == 2 || V == 3) {  // decoder group pattern: asm ring + LDS B + 4 MFMAs
        unsigned voff = lane * 16;
        const f32x4 *src = w;
        asm volatile("global_load_dwordx4 a[240:243], %0, %1" ::"v"(voff), "s"(src) : "a240", "a241", "a242", "a243");
        asm volatile("global_load_dwordx4 a[244:247], %0, %1 offset:1024" ::"v"(voff), "s"(src) : "a244", "a245", "a246", "a247");
        f32x4 bq = lds[lane];
        for (int i = 0; i < ITER / 2; i++) {
            f32x4 o, bn;
            bn = lds[((2 * i + 1) & 63) * 64 + lane];
            asm volatile("s_waitcnt vmcnt(1)\n\tv_accvgpr_read_b32 %0, a240\n\tv_accvgpr_read_b32 %1, a241\n\tv_accvgpr_read_b32 %2, a242\n\tv_accvgpr_read_b32 %3, a243\n\tglobal_load_dwordx4 a[240:243], %4, %5\n\ts_nop 1"
                         : "=&v"(o.x), "=&v"(o.y), "=&v"(o.z), "=&v"(o.w) : "v"(voff), "s"(src) : "a240", "a241", "a242", "a243");
            acc0 = MF(o.x, bq.x, acc0); acc0 = MF(o.y, bq.y, acc0); acc0 = MF(o.z, bq.z, acc0); acc0 = MF(o.w, bq.w, acc0);
            bq = lds[((2 * i + 2) & 63) * 64 + lane];
            asm volatile("s_waitcnt vmcnt(1)\n\tv_accvgpr_read_b32 %0, a244\n\tv_accvgpr_read_b32 %1, a245\n\tv_accvgpr_read_b32 %2, a246\n\tv_accvgpr_read_b32 %3, a247\n\tglobal_load_dwordx4 a[244:247], %4, %5 offset:1024\n\ts_nop 1"
                         : "=&v"(o.x), "=&v"(o.y), "=&v"(o.z), "=&v"(o.w) : "v"(voff), "s"(src) : "a244", "a245", "a246", "a247");
            if (V == 2) { acc0 = MF(o.x, bn.x, acc0); acc0 = MF(o.y, bn.y, acc0); acc0 = MF(o.z, bn.z, acc0); acc0 = MF(o.w, bn.w, acc0); }
            else        { acc1 = MF(o.x, bn.x, acc1); acc1 = MF(o.y, bn.y, acc1); acc1 = MF(o.z, bn.z, acc1); acc1 = MF(o.w, bn.w, acc1); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (V == 4) {  // dependent chain + 3 independent VALU ops per MFMA (overlap test)
        float x0 = a, x1 = b, x2 = a + b;
        for (int i = 0; i < ITER; i++) {
            acc0 = MF(a, b, acc0); x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f);
            acc0 = MF(b, a, acc0); x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f);
            acc0 = MF(a, a, acc0); x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f);
            acc0 = MF(b, b, acc0); x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f);
        }
        acc1[0] += x0 + x1 + x2;
    } else if (V == 5) {  // 12 independent VALU ops per MFMA
        float x[12];
        for (int j = 0; j < 12; j++) x[j] = a + j;
        for (int i = 0; i < ITER; i++) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                acc0 = MF(a, b, acc0);
#pragma unroll
                for (int j = 0; j < 12; j++) x[j] = fmaf(x[j], 1.0001f, 0.5f);
            }
        }
        for (int j = 0; j < 12; j++) acc1[0] += x[j];
    }
    else if (V == 6) {  // 4 independent accumulators + 12 independent VALU ops per MFMA
        float x[12];
        for (int j = 0; j < 12; j++) x[j] = a + j;
        for (int i = 0; i < ITER / 4; i++) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                acc0 = MF(a, b, acc0);
#pragma unroll
                for (int j = 0; j < 12; j++) x[j] = fmaf(x[j], 1.0001f, 0.5f);
                acc1 = MF(b, a, acc1);
#pragma unroll
                for (int j = 0; j < 12; j++) x[j] = fmaf(x[j], 1.0001f, 0.5f);
                acc2 = MF(a, a, acc2);
#pragma unroll
                for (int j = 0; j < 12; j++) x[j] = fmaf(x[j], 1.0001f, 0.5f);
                acc3 = MF(b, b, acc3);
#pragma unroll
                for (int j = 0; j < 12; j++) x[j] = fmaf(x[j], 1.0001f, 0.5f);
            }
        }
        for (int j = 0; j < 12; j++) acc1[0] += x[j];
    } else if (V == 7) {  // chain + 4 transcendentals (v_exp) per MFMA
        float x[4];
        for (int j = 0; j < 4; j++) x[j] = a + j;
        for (int i = 0; i < ITER * 4; i++) {
            acc0 = MF(a, b, acc0);
#pragma unroll
            for (int j = 0; j < 4; j++) x[j] = __builtin_amdgcn_exp2f(x[j]);
        }
        for (int j = 0; j < 4; j++) acc1[0] += x[j];
    } else if (V == 8) {  // 16x16x4 f32 chain (8 passes) + 6 VALU per MFMA (same VALU:FLOP ratio as V5)
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        f32x4v c0 = {0, 1, 2, 3};
        float x[6];
        for (int j = 0; j < 6; j++) x[j] = a + j;
        for (int i = 0; i < ITER * 8; i++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 6; j++) x[j] = fmaf(x[j], 1.0001f, 0.5f);
        }
        acc0[0] += c0[0] + c0[1] + c0[2] + c0[3];
        for (int j = 0; j < 6; j++) acc1[0] += x[j];
    } else if (V == 9) {  // 16x16x4 f32 chain alone
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        f32x4v c0 = {0, 1, 2, 3}, c1 = {1, 2, 3, 4};
        for (int i = 0; i < ITER * 4; i++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        }
        acc0[0] += c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(const char *name, const f32x4 *w, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, w, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, w, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double mfmas = (double)ITER * 4 * 1024;   // per launch: 4 waves x 256 WGs
    double tf = mfmas * 4096 / (ms * 1e-3) / 1e12;
    double cyc = ms * 1e-3 * 2.4e9 / (ITER * 4);
    printf("%-44s %8.3f ms  %7.1f TF  ~%5.1f cyc/MFMA @2.4GHz\n", name, ms, tf, cyc);
    fflush(stdout);
}

int main(int argc, char **argv) {
    int only = argc > 1 ? atoi(argv[1]) : -1;
    f32x4 *w; float *out;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
    hipMalloc(&out, 256 * 256 * 4);
    if (only < 0 || only == 0) run<0>("V0 one dependent chain", w, out);
    if (only < 0 || only == 1) run<1>("V1 four accumulators", w, out);
    if (only < 0 || only == 2) run<2>("V2 decoder group pattern, one acc", w, out);
    if (only < 0 || only == 3) run<3>("V3 decoder group pattern, two accs", w, out);
    if (only < 0 || only == 4) run<4>("V4 chain + 3 VALU / MFMA", w, out);
    if (only < 0 || only == 5) run<5>("V5 chain + 12 VALU / MFMA", w, out);
    if (only < 0 || only == 6) run<6>("V6 four accs + 12 VALU / MFMA", w, out);
    if (only < 0 || only == 7) run<7>("V7 chain + 4 v_exp / MFMA", w, out);
    if (only < 0 || only == 8) run<8>("V8 16x16x4 chain + 6 VALU / MFMA (x2 MFMAs)", w, out);
    if (only < 0 || only == 9) run<9>("V9 16x16x4 two chains (x2 MFMAs)", w, out);
    return 0;
}
