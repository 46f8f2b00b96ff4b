// v_permlane32_swap on gfx950: what it needs around it.  For three forms of "sum of a value over the two lane halves"
// (builtin alone, builtin behind an `s_nop 1` asm fence, all-asm with nops on both sides) the kernel computes
// v = VALU chain(lane), then the half sum, a million times with different values, and counts mismatches against
// ds_bpermute (__shfl_xor).  Variants place the swap (a) right behind VALU writes, (b) right behind an MFMA that
// writes the operand, (c) in front of a VALU read.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane_check.hip -o tools/ubench/permlane_check && tools/ubench/permlane_check
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FORM>
__device__ __forceinline__ float half_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = __builtin_bit_cast(unsigned, v);
    if (FORM == 0) {
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    } else if (FORM == 1) {
        asm volatile("s_nop 1" : "+v"(a), "+v"(b));
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    } else {
        asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(a), "+v"(b));
        return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
}

template <int FORM, bool MFMA>
__global__ void check(const float *in, int n, unsigned *bad, float *sample) {
    const int lane = threadIdx.x & 63;
    unsigned wrong = 0;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        float v = in[(size_t)i * 64 + lane];
        v = v * 1.25f + 0.5f;                       // VALU writes right in front
        if (MFMA) {                                 // ... or an MFMA result as the operand
            f32x16 acc;
            for (int r = 0; r < 16; r++) acc[r] = v + r;
            f16x8 x, y;
            for (int e = 0; e < 8; e++) { x[e] = (_Float16)(0.01f * (lane + e)); y[e] = (_Float16)(0.02f * (e + 1)); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
            v = acc[3];
        }
        const float want = v + __shfl_xor(v, 32, 64);
        const float got = half_sum<FORM>(v);
        wrong += (got != want) ? 1u : 0u;
        if (i == 0) sample[lane] = got - want;
    }
    atomicAdd(bad, wrong);
}

int main() {
    const int n = 1 << 16;
    float *in, *sample;
    unsigned *bad;
    hipMalloc(&in, (size_t)n * 64 * 4);
    hipMalloc(&sample, 256);
    hipMalloc(&bad, 4);
    float *h = new float[(size_t)n * 64];
    for (size_t i = 0; i < (size_t)n * 64; i++) h[i] = (float)((i * 2654435761u) % 100003) * 1e-3f - 50.0f;
    hipMemcpy(in, h, (size_t)n * 64 * 4, hipMemcpyHostToDevice);
#define RUN(FORM, MF)                                                                      \
    do {                                                                                   \
        unsigned z = 0;                                                                    \
        hipMemcpy(bad, &z, 4, hipMemcpyHostToDevice);                                      \
        hipLaunchKernelGGL((check<FORM, MF>), dim3(1024), dim3(64), 0, 0, in, n, bad, sample); \
        hipMemcpy(&z, bad, 4, hipMemcpyDeviceToHost);                                      \
        printf("form %d (%s) operand from %s: %u mismatching lanes of %d\n", FORM,          \
               FORM == 0 ? "builtin" : FORM == 1 ? "s_nop 1 fence + builtin" : "asm, s_nop 4 both sides", MF ? "MFMA" : "VALU", z, n * 64); \
    } while (0)
    RUN(0, false); RUN(1, false); RUN(2, false);
    RUN(0, true); RUN(1, true); RUN(2, true);
    return 0;
}
