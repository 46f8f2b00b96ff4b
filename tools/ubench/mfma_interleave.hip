// Micro-benchmark (round 2): does the split-fp16 decoder's stream lose its time to DEPENDENT
// MFMAs separated by other instructions?  (MI355X_MICROARCH.md: one extra issue slot between two
// MFMAs on the same accumulator costs ~43 cycles; between MFMAs on different accumulators ~6.)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_interleave mfma_interleave.hip
// All variants run the decoder's weight stream: 2 x 16 KiB staging buffers of 8 K-blocks, filled by
// LDS-DMA (two K-blocks per wave per chunk), one raw s_barrier per chunk, A operands read back with
// ds_read_b128 one K-block ahead.  MODE:
//   0  one accumulator chain per output tile (the round-1 kernel): lo*bh, hi*bl, hi*bh -> acc
//   1  three accumulators per output tile in rotation (hi*bh -> M, lo*bh -> C1, hi*bl -> C2)
//   2  two OUTPUT tiles share each B operand (A of K-blocks p and p+1 of the chunk; 6 MFMAs alternate)
//   3  two POINT tiles share each A operand (half the A reads / DMA / barriers per product)
// F = independent v_fma_f32 behind every MFMA; BLDS = B operands re-read from the wave's LDS slab.
// Reports shader cycles per K-block PRODUCT (3 MFMAs, ideal 96) for wave 0 of block 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)
constexpr int PRODUCTS = 4096;  // K-block products per wave
constexpr int CK = 8;

__device__ __forceinline__ void pin(u32x4 &a, u32x4 &b) { asm volatile("" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void glds4(const char *g, unsigned dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 "\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
                 "\n\tglobal_load_lds_dwordx4 %0, off offset:2048"
                 "\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                 : : "v"(g), "s"(dst) : "memory");
}

template <int MODE, int F, bool BLDS>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *w, float *out, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[10240];  // 160 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 10240; i += 256) lds[i] = w[i & 4095];
    __syncthreads();
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; r++) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u32x4 *)lds;
    const char *g = reinterpret_cast<const char *>(w) + wave * 4096 + lane * 16;
    const u32x4 *A = lds + lane;                       // 2 buffers x 1024 u32x4
    const u32x4 *B = lds + 2048 + wave * 2048 + lane;  // the wave's slab
    u32x4 rb0h = w[lane], rb0l = w[64 + lane], rb1h = w[128 + lane], rb1l = w[192 + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pin(rb0h, rb0l);
    pin(rb1h, rb1l);
    float x[16];
    for (int j = 0; j < 16; j++) x[j] = lane * 0.01f + j;
#define VALU() do { _Pragma("unroll") for (int j = 0; j < F; j++) x[j] = fmaf(x[j], 1.0001f, 0.5f); } while (0)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int buf = 0;
    // products per A K-block: 1 (modes 0-2) or 2 (mode 3)
    constexpr int PPA = MODE == 3 ? 2 : 1;
    constexpr int CHUNKS = PRODUCTS / PPA / CK;
    if (MODE == 0 || MODE == 1 || MODE == 3) {
        u32x4 hi = A[0], lo = A[64];
        u32x4 b0h = rb0h, b0l = rb0l, b1h = rb1h, b1l = rb1l;
        for (int i = 0; i < CHUNKS; i++) {
#pragma unroll
            for (int p = 0; p < CK; p++) {
                u32x4 ahi = hi, alo = lo;
                if (p == CK - 1) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    glds4(g + (i & 127) * 16384, base + buf * 16384 + wave * 4096);
                    buf ^= 1;
                    hi = A[buf * 1024];
                    lo = A[buf * 1024 + 64];
                } else {
                    hi = A[buf * 1024 + (p + 1) * 128];
                    lo = A[buf * 1024 + (p + 1) * 128 + 64];
                }
                u32x4 u0h = b0h, u0l = b0l, u1h = b1h, u1l = b1l;
                if (BLDS) {
                    b0h = B[((p + 1) & 7) * 128];
                    b0l = B[((p + 1) & 7) * 128 + 64];
                    if (MODE == 3) {
                        b1h = B[1024 + ((p + 1) & 7) * 128];
                        b1l = B[1024 + ((p + 1) & 7) * 128 + 64];
                    }
                }
                pin(ahi, alo);
                if (MODE == 0) {
                    c0 = MF(alo, u0h, c0); VALU(); c0 = MF(ahi, u0l, c0); VALU(); c0 = MF(ahi, u0h, c0); VALU();
                } else if (MODE == 1) {
                    c1 = MF(alo, u0h, c1); VALU(); c2 = MF(ahi, u0l, c2); VALU(); c0 = MF(ahi, u0h, c0); VALU();
                } else {
                    c0 = MF(alo, u0h, c0); VALU(); c1 = MF(alo, u1h, c1); VALU();
                    c0 = MF(ahi, u0l, c0); VALU(); c1 = MF(ahi, u1l, c1); VALU();
                    c0 = MF(ahi, u0h, c0); VALU(); c1 = MF(ahi, u1h, c1); VALU();
                }
            }
        }
    } else if (MODE == 8 || MODE == 9) {
        u32x4 hi = A[0], lo = A[64];
        u32x4 b0h = rb0h, b0l = rb0l;
        for (int i = 0; i < CHUNKS; i++) {
            const char *gi = g + (i & 127) * 16384;
#pragma unroll
            for (int p = 0; p < CK; p++) {
                u32x4 ahi = hi, alo = lo;
                if (p == CK - 1) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    buf ^= 1;
                }
                hi = A[buf * 1024 + ((p + 1) & 7) * 128];
                lo = A[buf * 1024 + ((p + 1) & 7) * 128 + 64];
                pin(ahi, alo);
                // K-block p has gaps 3p, 3p+1, 3p+2; pieces only in K-blocks 0..5 (gaps 0..15 of 18)
#define GAP(G, MFMA_STMT)                                                                               \
                do {                                                                                    \
                    MFMA_STMT;                                                                          \
                    if ((G) < 16 && wave == ((G) & 3)) {                                                \
                        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off"               \
                                     : : "v"(gi + ((G) >> 2) * 1024),                                   \
                                         "s"(base + (buf ^ 1) * 16384 + wave * 4096 + ((G) >> 2) * 1024) : "memory"); \
                    }                                                                                   \
                } while (0)
                GAP(3 * p + 0, c0 = MF(alo, b0h, c0));
                GAP(3 * p + 1, c0 = MF(ahi, b0l, c0));
                GAP(3 * p + 2, c0 = MF(ahi, b0h, c0));
            }
        }
    } else if (MODE == 10) {
        const u32x4 *gw = w + wave * 0 + lane;   // all four waves read the same stream
        u32x4 hi = gw[0], lo = gw[64];
        u32x4 b0h = rb0h, b0l = rb0l;
        for (int i = 0; i < PRODUCTS; i += 8) {
#pragma unroll
            for (int p = 0; p < 8; p++) {
                u32x4 ahi = hi, alo = lo;
                const int nx = ((i + p + 1) & 1023) * 128;
                hi = __builtin_nontemporal_load(gw + nx);
                lo = __builtin_nontemporal_load(gw + nx + 64);
                pin(ahi, alo);
                c0 = MF(alo, b0h, c0); c0 = MF(ahi, b0l, c0); c0 = MF(ahi, b0h, c0);
            }
        }
    } else if (MODE >= 4) {
        u32x4 hi = A[0], lo = A[64], hi2 = A[128], lo2 = A[192];
        u32x4 b0h = rb0h, b0l = rb0l;
        for (int i = 0; i < CHUNKS; i++) {
#pragma unroll
            for (int p = 0; p < CK; p++) {
                u32x4 ahi = hi, alo = lo;
                if (MODE == 7) { hi = hi2; lo = lo2; }
                if (p == CK - 1) {
                    if (MODE >= 5) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    buf ^= 1;
                }
                if (MODE == 7) {
                    const int q = p + 2, bb = q >= CK ? (p == CK - 1 ? buf : buf ^ 1) : buf;
                    hi2 = A[bb * 1024 + (q & 7) * 128];
                    lo2 = A[bb * 1024 + (q & 7) * 128 + 64];
                } else {
                    const int q = p + 1;
                    hi = A[buf * 1024 + (q & 7) * 128];
                    lo = A[buf * 1024 + (q & 7) * 128 + 64];
                }
                pin(ahi, alo);
                if (MODE >= 6 && p < 4) {
                    // first MFMA of the K-block and ONE LDS-DMA piece behind it, in one statement
                    asm volatile("s_mov_b32 m0, %4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %3, off"
                                 : "+v"(c0) : "v"(alo), "v"(b0h), "v"(g + (i & 127) * 16384 + p * 1024),
                                   "s"(base + (buf ^ 1) * 16384 + wave * 4096 + p * 1024) : "memory");
                } else {
                    c0 = MF(alo, b0h, c0);
                }
                c0 = MF(ahi, b0l, c0); c0 = MF(ahi, b0h, c0);
            }
        }
    } else {  // MODE 2: K-blocks (p, p+1) of the chunk belong to two output tiles, same B
        u32x4 hi0 = A[0], lo0 = A[64], hi1 = A[128], lo1 = A[192];
        u32x4 b0h = rb0h, b0l = rb0l;
        for (int i = 0; i < CHUNKS; i++) {
#pragma unroll
            for (int p = 0; p < CK; p += 2) {
                u32x4 ah0 = hi0, al0 = lo0, ah1 = hi1, al1 = lo1;
                if (p == CK - 2) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    glds4(g + (i & 127) * 16384, base + buf * 16384 + wave * 4096);
                    buf ^= 1;
                    hi0 = A[buf * 1024]; lo0 = A[buf * 1024 + 64];
                    hi1 = A[buf * 1024 + 128]; lo1 = A[buf * 1024 + 192];
                } else {
                    hi0 = A[buf * 1024 + (p + 2) * 128]; lo0 = A[buf * 1024 + (p + 2) * 128 + 64];
                    hi1 = A[buf * 1024 + (p + 3) * 128]; lo1 = A[buf * 1024 + (p + 3) * 128 + 64];
                }
                u32x4 u0h = b0h, u0l = b0l;
                if (BLDS) {
                    b0h = B[((p / 2 + 1) & 7) * 128];
                    b0l = B[((p / 2 + 1) & 7) * 128 + 64];
                }
                pin(ah0, al0);
                pin(ah1, al1);
                c0 = MF(al0, u0h, c0); VALU(); c1 = MF(al1, u0h, c1); VALU();
                c0 = MF(ah0, u0l, c0); VALU(); c1 = MF(ah1, u0l, c1); VALU();
                c0 = MF(ah0, u0h, c0); VALU(); c1 = MF(ah1, u0h, c1); VALU();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r] + x[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int F, bool BLDS>
void run(const u32x4 *w, float *out, unsigned long long *cyc) {
    static const char *names[] = {"one chain", "3 accumulators in rotation", "2 output tiles share B",
                                  "2 point tiles share A", "A reads only", "A reads + barrier", "barrier + spread DMA",
                                  "spread DMA + reads 2 ahead", "staggered DMA (wave = gap & 3)", "", "A straight from global"};
    hipLaunchKernelGGL((k<MODE, F, BLDS>), dim3(256), dim3(256), 0, 0, w, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, F, BLDS>), dim3(256), dim3(256), 0, 0, w, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%-28s) F=%d B=%s  %7.1f cycles/product  (%.3f ms, %.0f TFLOP/s executed)\n", MODE, names[MODE],
           F, BLDS ? "lds" : "reg", (double)c / PRODUCTS, ms, 256.0 * 4 * PRODUCTS * 3 * 32768.0 / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

template <int F, bool BLDS>
void all_modes(const u32x4 *w, float *out, unsigned long long *cyc) {
    run<0, F, BLDS>(w, out, cyc);
    run<1, F, BLDS>(w, out, cyc);
    run<2, F, BLDS>(w, out, cyc);
    run<3, F, BLDS>(w, out, cyc);
}

int main() {
    u32x4 *w; float *out; unsigned long long *cyc;
    hipMalloc(&w, 4 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    unsigned *h = (unsigned *)malloc(4 << 20);
    for (int i = 0; i < (1 << 20); i++) h[i] = 0x3c003c00u + (i & 0xff);
    hipMemcpy(w, h, 4 << 20, hipMemcpyHostToDevice);
    run<4, 0, false>(w, out, cyc);
    run<5, 0, false>(w, out, cyc);
    run<0, 0, false>(w, out, cyc);
    run<6, 0, false>(w, out, cyc);
    run<7, 0, false>(w, out, cyc);
    run<8, 0, false>(w, out, cyc);
    run<10, 0, false>(w, out, cyc);
    if (getenv("ZS_UB_SHORT")) return 0;
    all_modes<0, false>(w, out, cyc);
    all_modes<0, true>(w, out, cyc);
    all_modes<2, false>(w, out, cyc);
    all_modes<2, true>(w, out, cyc);
    all_modes<4, false>(w, out, cyc);
    all_modes<4, true>(w, out, cyc);
    all_modes<6, true>(w, out, cyc);
    return 0;
}
