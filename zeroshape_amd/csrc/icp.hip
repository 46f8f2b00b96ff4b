// standardize_pc (utils/eval_3D.py:83-91) and one iteration of the reference's point-to-point ICP
// (utils/eval_3D.py:271-284) on the device.  The reference does, per iteration: Chamfer nearest neighbours ->
// gather -> two means -> a 3 x 3 cross-covariance by matmul -> torch.svd -> R = V U^T with its own sign rule ->
// X1 <- (X1 - t1) R^T + t2: a dozen ATen launches and an SVD round trip for 9 numbers.  Here the iteration after the
// Chamfer call (zs_chamfer_forward, which supplies idx1) is two launches:
//   icp_fit_kernel    one workgroup per batch element: sums of X1, of the gathered X2 and of the nine products in
//                     double (fixed order: wave butterflies, then the waves in order), then thread 0 forms
//                     H = sum (x1 - t1)(x2c - t2)^T, diagonalises H^T H by cyclic Jacobi rotations (3 x 3, double),
//                     u_i = H v_i / s_i, R = sum_i v_i u_i^T (independent of the order and signs of the singular
//                     triplets), negates row 2 of R when det R < 0 (the reference's rule, :282 - not Kabsch's) and
//                     stores R, t1, t2;
//   icp_apply_kernel  X1_out = (X1 - t1) R^T + t2.
// Memory-bound, tiny (10k points x 24 clouds = 2.9 MB per pass); the point is that no value leaves the device and no
// ATen / LAPACK call sits in the loop.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int ICP_THREADS = 256;
constexpr int ICP_FIT = 16;          // floats per batch element: R[9], t1[3], t2[3], pad

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the workgroup of NV doubles per thread, result in out[NV] for every thread (fixed order)
template <int NV>
__device__ __forceinline__ void block_sums(double (&v)[NV], double (*lds)[ICP_THREADS / 64], double (&out)[NV]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const double s = wave_sum_d(v[i]);
        if (lane == 0) lds[i][wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double s = lds[i][0];
        for (int w = 1; w < ICP_THREADS / 64; w++) s += lds[i][w];
        out[i] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(ICP_THREADS) void standardize_pc_kernel(const float *__restrict__ pc, int n,
                                                                     float *__restrict__ out) {
    __shared__ double lds[3][ICP_THREADS / 64];
    const float *p = pc + (size_t)blockIdx.x * n * 3;
    float *o = out + (size_t)blockIdx.x * n * 3;
    double s[3] = {0.0, 0.0, 0.0}, tot[3];
    for (int i = threadIdx.x; i < n; i += ICP_THREADS) {
        s[0] += p[(size_t)i * 3];
        s[1] += p[(size_t)i * 3 + 1];
        s[2] += p[(size_t)i * 3 + 2];
    }
    block_sums<3>(s, lds, tot);
    const float mx = (float)(tot[0] / n), my = (float)(tot[1] / n), mz = (float)(tot[2] / n);
    double q[1] = {0.0}, qt[1];
    for (int i = threadIdx.x; i < n; i += ICP_THREADS) {
        const float x = p[(size_t)i * 3] - mx, y = p[(size_t)i * 3 + 1] - my, z = p[(size_t)i * 3 + 2] - mz;
        const float d = sqrtf(x * x + y * y + z * z);              // origin_distance, then squared again (:88-89)
        q[0] += (double)(d * d);
    }
    block_sums<1>(q, lds, qt);
    const float scale2 = sqrtf((float)(qt[0] / n)) * 2.0f;
    for (int i = threadIdx.x; i < n; i += ICP_THREADS) {
        o[(size_t)i * 3] = (p[(size_t)i * 3] - mx) / scale2;
        o[(size_t)i * 3 + 1] = (p[(size_t)i * 3 + 1] - my) / scale2;
        o[(size_t)i * 3 + 2] = (p[(size_t)i * 3 + 2] - mz) / scale2;
    }
}

// eigenvectors of the symmetric 3 x 3 matrix a (destroyed): columns of v, eigenvalues on the diagonal of a
__device__ void jacobi3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 24; sweep++) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) {          // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {          // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {          // V <- V J
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

__global__ __launch_bounds__(ICP_THREADS) void icp_fit_kernel(const float *__restrict__ x1, int n,
                                                              const float *__restrict__ x2, int m,
                                                              const int *__restrict__ idx1, float *__restrict__ fit) {
    __shared__ double lds[15][ICP_THREADS / 64];
    const int b = blockIdx.x;
    const float *p = x1 + (size_t)b * n * 3, *g = x2 + (size_t)b * m * 3;
    const int *ix = idx1 + (size_t)b * n;
    double s[15], tot[15];
#pragma unroll
    for (int i = 0; i < 15; i++) s[i] = 0.0;
    for (int i = threadIdx.x; i < n; i += ICP_THREADS) {
        int j = ix[i];
        j = j < 0 ? 0 : (j >= m ? m - 1 : j);
        const double a[3] = {p[(size_t)i * 3], p[(size_t)i * 3 + 1], p[(size_t)i * 3 + 2]};
        const double c[3] = {g[(size_t)j * 3], g[(size_t)j * 3 + 1], g[(size_t)j * 3 + 2]};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s[k] += a[k];
            s[3 + k] += c[k];
#pragma unroll
            for (int l = 0; l < 3; l++) s[6 + 3 * k + l] += a[k] * c[l];
        }
    }
    block_sums<15>(s, lds, tot);
    if (threadIdx.x != 0) return;
    // the reference's means are fp32 tensors: the centred clouds are formed with those rounded values (:278-280)
    float t1[3], t2[3];
    for (int k = 0; k < 3; k++) {
        t1[k] = (float)(tot[k] / n);
        t2[k] = (float)(tot[3 + k] / n);
    }
    // H = sum (a - t1)(c - t2)^T = sum a c^T - t1 sum c^T - (sum a) t2^T + n t1 t2^T
    double H[3][3], S[3][3], V[3][3];
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++)
            H[k][l] = tot[6 + 3 * k + l] - (double)t1[k] * tot[3 + l] - tot[k] * (double)t2[l] + (double)n * t1[k] * t2[l];
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) S[k][l] = H[0][k] * H[0][l] + H[1][k] * H[1][l] + H[2][k] * H[2][l];   // H^T H
    jacobi3(S, V);
    // order the triplets by decreasing singular value; u_i = H v_i / s_i, the last one completed by a cross product
    // when its singular value vanishes (a planar or collinear cloud)
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (S[ord[j]][ord[j]] > S[ord[i]][ord[i]]) { const int tmp = ord[i]; ord[i] = ord[j]; ord[j] = tmp; }
    double U[3][3], Vs[3][3];
    const double smax = sqrt(fmax(S[ord[0]][ord[0]], 0.0));
    for (int i = 0; i < 3; i++) {
        const int c = ord[i];
        for (int k = 0; k < 3; k++) Vs[k][i] = V[k][c];
        const double sv = sqrt(fmax(S[c][c], 0.0));
        if (sv > 1e-12 * smax && sv > 0.0) {
            for (int k = 0; k < 3; k++) U[k][i] = (H[k][0] * V[0][c] + H[k][1] * V[1][c] + H[k][2] * V[2][c]) / sv;
        } else if (i == 2) {
            U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
            U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
            U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
        } else {
            for (int k = 0; k < 3; k++) U[k][i] = k == i ? 1.0 : 0.0;       // degenerate input (fewer than two directions)
        }
    }
    double R[3][3];
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) R[k][l] = Vs[k][0] * U[l][0] + Vs[k][1] * U[l][1] + Vs[k][2] * U[l][2];   // V U^T
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0.0)
        for (int l = 0; l < 3; l++) R[2][l] = -R[2][l];               // R[R.det() < 0, 2] *= -1  (:282)
    float *f = fit + (size_t)b * ICP_FIT;
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) f[3 * k + l] = (float)R[k][l];
    for (int k = 0; k < 3; k++) {
        f[9 + k] = t1[k];
        f[12 + k] = t2[k];
    }
    f[15] = 0.f;
}

__global__ __launch_bounds__(ICP_THREADS) void icp_apply_kernel(const float *__restrict__ x1, int n,
                                                                const float *__restrict__ fit, float *__restrict__ out) {
    const int b = blockIdx.y;
    const float *f = fit + (size_t)b * ICP_FIT;
    const float *p = x1 + (size_t)b * n * 3;
    float *o = out + (size_t)b * n * 3;
    for (int i = blockIdx.x * ICP_THREADS + threadIdx.x; i < n; i += gridDim.x * ICP_THREADS) {
        const float x = p[(size_t)i * 3] - f[9], y = p[(size_t)i * 3 + 1] - f[10], z = p[(size_t)i * 3 + 2] - f[11];
        // (X1 - t1) @ R^T: out_k = sum_l (x1 - t1)_l R[k][l], summed in l order like the matmul, then + t2
        o[(size_t)i * 3] = fmaf(z, f[2], fmaf(y, f[1], x * f[0])) + f[12];
        o[(size_t)i * 3 + 1] = fmaf(z, f[5], fmaf(y, f[4], x * f[3])) + f[13];
        o[(size_t)i * 3 + 2] = fmaf(z, f[8], fmaf(y, f[7], x * f[6])) + f[14];
    }
}

}  // namespace

extern "C" int zs_standardize_pc(const float *pc, int b, int n, float *out, void *stream) {
    if (b < 0 || n < 0) {
        zs::set_err("zs_standardize_pc: negative size (b=%d n=%d)", b, n);
        return 0;
    }
    if (b == 0 || n == 0) return 1;
    if (!pc || !out) {
        zs::set_err("zs_standardize_pc: null pointer");
        return 0;
    }
    hipLaunchKernelGGL(standardize_pc_kernel, dim3(b), dim3(ICP_THREADS), 0, static_cast<hipStream_t>(stream), pc, n, out);
    return zs::check_launch("zs_standardize_pc") ? 1 : 0;
}

extern "C" size_t zs_icp_scratch_bytes(int b) { return b > 0 ? (size_t)b * ICP_FIT * sizeof(float) : 0; }

extern "C" int zs_icp_step(const float *x1, int n, const float *x2, int m, const int *idx1, int b, float *x1_out,
                           void *scratch, void *stream) {
    if (b < 0 || n < 0 || m < 0) {
        zs::set_err("zs_icp_step: negative size (b=%d n=%d m=%d)", b, n, m);
        return 0;
    }
    if (b == 0 || n == 0) return 1;
    if (m == 0 || b > 65535) {
        zs::set_err("zs_icp_step: empty target cloud or batch > 65535 (b=%d m=%d)", b, m);
        return 0;
    }
    if (!x1 || !x2 || !idx1 || !x1_out || !scratch) {
        zs::set_err("zs_icp_step: null pointer");
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *fit = static_cast<float *>(scratch);
    hipLaunchKernelGGL(icp_fit_kernel, dim3(b), dim3(ICP_THREADS), 0, st, x1, n, x2, m, idx1, fit);
    int bx = (n + ICP_THREADS - 1) / ICP_THREADS;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(icp_apply_kernel, dim3(bx, b), dim3(ICP_THREADS), 0, st, x1, n, static_cast<const float *>(fit), x1_out);
    return zs::check_launch("zs_icp_step") ? 1 : 0;
}
