// Micro-benchmark for the split-fp16 decoder's inner loop: what does one wave per SIMD sustain with
// v_mfma_f32_32x32x16_bf16 under the loop shapes of csrc/sdf_decoder_split.hip?
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_f16_stream mfma_f16_stream.hip
// Reports shader cycles (s_memtime) per K-block of 3 MFMAs (ideal 96) for wave 0 of block 0, all
// 256 CUs busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
constexpr int ITER = 4096;  // K-blocks

__device__ __forceinline__ void pin(u32x4 &a, u32x4 &b) { asm volatile("" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void glds(const char *g, unsigned dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
                 : : "v"(g), "s"(dst) : "memory");
}

// V: 0 one chain, register operands | 1 two alternating chains | 2 one chain + 2 ds_read_b128 (A) per
// K-block, distance 1 | 3 = 2 with two chains | 4 = 2 + barrier every 4 K-blocks | 5 = 4 + LDS-DMA
// (the decoder's stream) | 6 = 5 with two chains | 7 = 5 + 2 more ds_read_b128 (B) per K-block
// 8 / 9 / 10 = 5 + 4 / 8 / 16 independent v_fma_f32 behind every MFMA | 11 = 0 + 4 v_fma per MFMA
template <int V>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *w, float *out, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[10240];  // 160 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 10240; i += 256) lds[i] = w[i & 4095];
    __syncthreads();
    f32x16 c0, c1;
    for (int r = 0; r < 16; r++) { c0[r] = 0.f; c1[r] = 0.f; }
    u32x4 a = w[lane], b = w[64 + lane], a2 = w[128 + lane], b2 = w[192 + lane];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u32x4 *)lds;
    const char *g = reinterpret_cast<const char *>(w) + wave * 2048 + lane * 16;
    const u32x4 *A = lds + lane;            // staged chunks: 3 x 512 u32x4
    const u32x4 *B = lds + 2048 + wave * 2048 + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float x[16];
    for (int j = 0; j < 16; j++) x[j] = lane * 0.01f + j;
#define VALU(N) do { _Pragma("unroll") for (int j = 0; j < (N); j++) x[j] = fmaf(x[j], 1.0001f, 0.5f); } while (0)
    constexpr int NV = V == 8 ? 4 : V == 9 ? 8 : V == 10 ? 16 : V == 11 ? 4 : 0;
    if (V == 0 || V == 11) {
        for (int i = 0; i < ITER; i++) {
            c0 = MF(a, b, c0); VALU(NV); c0 = MF(a2, b, c0); VALU(NV); c0 = MF(a, b2, c0); VALU(NV);
        }
    } else if (V == 1) {
        for (int i = 0; i < ITER / 2; i++) {
            c0 = MF(a, b, c0); c1 = MF(a2, b, c1); c0 = MF(a, b2, c0);
            c1 = MF(a, b, c1); c0 = MF(a2, b, c0); c1 = MF(a, b2, c1);
        }
    } else {
        u32x4 hi = A[0], lo = A[64], bh = B[0], bl = B[64];
        int buf = 0;
        for (int i = 0; i < ITER / 4; i++) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                u32x4 ahi = hi, alo = lo;
                if (p == 3) {
                    if (V >= 4) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (V >= 5) glds(g + (i & 255) * 8192, base + buf * 8192 + wave * 2048);
                    buf = buf == 2 ? 0 : buf + 1;
                    hi = A[buf * 512];
                    lo = A[buf * 512 + 64];
                } else {
                    hi = A[buf * 512 + (p + 1) * 128];
                    lo = A[buf * 512 + (p + 1) * 128 + 64];
                }
                u32x4 ubh = bh, ubl = bl;
                if (V == 7) {
                    bh = B[((i * 4 + p + 1) & 15) * 128];
                    bl = B[((i * 4 + p + 1) & 15) * 128 + 64];
                }
                pin(ahi, alo);
                if (V == 3 || V == 6) {
                    if (p & 1) { c1 = MF(alo, ubh, c1); c0 = MF(ahi, ubl, c0); c1 = MF(ahi, ubh, c1); }
                    else       { c0 = MF(alo, ubh, c0); c1 = MF(ahi, ubl, c1); c0 = MF(ahi, ubh, c0); }
                } else {
                    c0 = MF(alo, ubh, c0); VALU(NV); c0 = MF(ahi, ubl, c0); VALU(NV); c0 = MF(ahi, ubh, c0); VALU(NV);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + x[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(const u32x4 *w, float *out, unsigned long long *cyc, const char *what) {
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, w, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, w, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("V%d %-62s %7.1f cycles/K-block  (%.3f ms, %.0f TFLOP/s executed)\n", V, what, (double)c / ITER, ms,
           256.0 * 4 * ITER * 3 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    u32x4 *w; float *out; unsigned long long *cyc;
    hipMalloc(&w, 4 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    unsigned *h = (unsigned *)malloc(4 << 20);
    for (int i = 0; i < (1 << 20); i++) h[i] = 0x3c003c00u + (i & 0xff);
    hipMemcpy(w, h, 4 << 20, hipMemcpyHostToDevice);
    run<0>(w, out, cyc, "one chain, register operands");
    run<1>(w, out, cyc, "two alternating chains, register operands");
    run<2>(w, out, cyc, "one chain + 2 ds_read_b128 per K-block");
    run<3>(w, out, cyc, "two chains + 2 ds_read_b128 per K-block");
    run<4>(w, out, cyc, "one chain + 2 ds_read + barrier per 4 K-blocks");
    run<5>(w, out, cyc, "one chain + 2 ds_read + barrier + LDS-DMA (decoder stream)");
    run<6>(w, out, cyc, "two chains + 2 ds_read + barrier + LDS-DMA");
    run<7>(w, out, cyc, "one chain + 4 ds_read + barrier + LDS-DMA");
    run<8>(w, out, cyc, "decoder stream (V5) + 4 v_fma_f32 per MFMA");
    run<9>(w, out, cyc, "decoder stream (V5) + 8 v_fma_f32 per MFMA");
    run<10>(w, out, cyc, "decoder stream (V5) + 16 v_fma_f32 per MFMA");
    run<11>(w, out, cyc, "one chain, register operands + 4 v_fma_f32 per MFMA");
    return 0;
}
