/* Host mirror of the activation formulas in zeroshape_amd/csrc/sdf_math.h (same
 * constants, same operation order, fmaf where the kernel uses fmaf; exp2f/log2f stand in
 * for v_exp_f32 / v_log_f32, 1.0f/x for v_rcp_f32).  tests/test_device_math.py bounds
 * their error against fp64. */
#include <math.h>
float zs_host_gelu(float x) {
    const float u = fabsf(x) * 0.70710678118654752440f;
    const float t = 1.0f / fmaf(0.3275911f, u, 1.0f);
    float p = 0.75052702f;
    p = fmaf(p, t, -1.02753365f);
    p = fmaf(p, t, 1.00509130f);
    p = fmaf(p, t, -0.20116957f);
    p = fmaf(p, t, 0.18019173f);
    p = p * t;
    const float e = exp2f((u * u) * -1.44269504088896340736f);
    return fmaf(-(u * p), e, fmaxf(x, 0.0f));
}
float zs_host_softplus100(float x) {
    const float t = exp2f(fabsf(x) * -144.26950408889634074f);
    const float l = log2f(1.0f + t);
    return fmaf(l, 0.0069314718055994530942f, fmaxf(x, 0.0f));
}
void zs_host_apply(int which, const float *x, float *y, int n) {
    for (int i = 0; i < n; i++) y[i] = which == 0 ? zs_host_gelu(x[i]) : zs_host_softplus100(x[i]);
}
