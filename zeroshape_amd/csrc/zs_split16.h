// Split-fp16 operand helpers shared by the kernels that run fp32 contractions on the 16-bit matrix
// pipe (csrc/sdf_decoder_split.hip, csrc/nn_conv.hip): x ~= hi + lo with both halves fp16, rounded
// to nearest even (gfx950's v_cvt_pk_f16_f32), |x - hi - lo| <= 2^-22 |x| and the error has no sign
// preference.  Rounds 1-2 rounded toward zero (v_cvt_pkrtz_f16_f32): 2^-20 worst case and every operand
// error of one sign, which accumulated coherently over the K = 256..1024 sums and the ~25 dependent
// layers - on trained weights (logit scale 15) 4.2e-5 from the fp32 kernel with a mean of 1.4e-5; the CPU
// emulation of both roundings (tests/mfma_emulator.py) gives 7.2e-5 / mean -3.6e-5 (toward zero) against
// 1.8e-5 / mean -5.7e-6 (nearest) at that scale.  Same instruction count.  Range: |x| < 65,520 (beyond
// that hi is +-inf and the products NaN - loud, where the truncating form saturated silently);
//     A B ~= A_hi B_hi + A_hi B_lo + A_lo B_hi     (fp32 accumulation)
// = three v_mfma_f32_32x32x16_f16 per K = 16 instead of eight v_mfma_f32_32x32x2_f32 at a quarter
// of the rate each.
#pragma once
#include <hip/hip_runtime.h>

namespace zs {
namespace s16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define ZS_S16 __device__ __forceinline__ static

ZS_S16 f16x8 as_h(const u32x4 &v) { return __builtin_bit_cast(f16x8, v); }
ZS_S16 unsigned pk_f16(float a, float b) {
#ifdef ZS_SPLIT_RTZ   // A/B measurements of the rounding only (tools/build_variant_lib.py)
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
#else
    unsigned h;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    return h;
#endif
}
// two fp32 values -> packed fp16 heads and packed fp16 remainders.  The remainder x - hi comes from
// one v_fma_mix_f32 per value (fp16 operand read in place from the packed register, exact in fp32):
// 2 VALU instructions per value.
ZS_S16 void split2(float a, float b, unsigned &h, unsigned &l) {
    h = pk_f16(a, b);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(b));
    l = pk_f16(ra, rb);
}
// eight fp32 values (two quads) -> one K = 16 MFMA operand pair
ZS_S16 void split8(const f32x4 &q0, const f32x4 &q1, u32x4 &hi, u32x4 &lo) {
    unsigned h[4], l[4];
    split2(q0.x, q0.y, h[0], l[0]);
    split2(q0.z, q0.w, h[1], l[1]);
    split2(q1.x, q1.y, h[2], l[2]);
    split2(q1.z, q1.w, h[3], l[3]);
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
}
ZS_S16 void mfma3(f32x16 &acc, const u32x4 &ahi, const u32x4 &alo, const u32x4 &bhi, const u32x4 &blo) {
#ifndef ZS_EXP_TWO_TERM   // timing experiment only (tools/build_variant_lib.py): what does the third MFMA cost in wall time?
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(alo), as_h(bhi), acc, 0, 0, 0);
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ahi), as_h(blo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ahi), as_h(bhi), acc, 0, 0, 0);
}

#undef ZS_S16

}  // namespace s16
}  // namespace zs
