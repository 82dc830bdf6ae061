// Two-view linear triangulation, one lane per correspondence.
//
// Work of `triangulate_pts` (src/triangulation.py:10-32): for every correspondence the 6 x 4 matrix
// A = [ [x1]_x P1 ; [x2]_x P2 ] ([v]_x as src/utils.py:4-5 builds it) and its right singular vector of
// the smallest singular value, divided by its last entry and rounded to float32.  The reference
// takes the vector from LAPACK's SVD; here a one-sided Jacobi iteration orthogonalises the four
// columns of A in registers (rotations applied to A and accumulated in V), which finds the same
// vector to high relative accuracy without forming A^T A.  The column that ends with the smallest
// norm is the answer.
#include "ba_common.h"

namespace sfmba {

constexpr int kTriSweeps = 12;  // upper bound; a wave leaves as soon as all its lanes have converged

__global__ __launch_bounds__(256) void k_triangulate(const double* __restrict__ x1, const double* __restrict__ x2,
                                                     const double* __restrict__ P1, const double* __restrict__ P2,
                                                     int64_t n, float* __restrict__ Xh) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const int64_t ii = live ? i : 0;
  double a[6][4], v[4][4];
  {
    const double u1[3] = {x1[3 * ii], x1[3 * ii + 1], x1[3 * ii + 2]};
    const double u2[3] = {x2[3 * ii], x2[3 * ii + 1], x2[3 * ii + 2]};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double p0 = P1[c], p1 = P1[4 + c], p2 = P1[8 + c];
      a[0][c] = -u1[2] * p1 + u1[1] * p2;
      a[1][c] = u1[2] * p0 - u1[0] * p2;
      a[2][c] = -u1[1] * p0 + u1[0] * p1;
      const double q0 = P2[c], q1 = P2[4 + c], q2 = P2[8 + c];
      a[3][c] = -u2[2] * q1 + u2[1] * q2;
      a[4][c] = u2[2] * q0 - u2[0] * q2;
      a[5][c] = -u2[1] * q0 + u2[0] * q1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) v[r][c] = r == c ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < kTriSweeps; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          alpha = fma(a[r][p], a[r][p], alpha);
          beta = fma(a[r][q], a[r][q], beta);
          gamma = fma(a[r][p], a[r][q], gamma);
        }
        // columns already orthogonal to working precision: leave them
        if (fabs(gamma) > 1e-15 * sqrt(alpha * beta) && gamma != 0.0) {
          rotated = true;
          // tan of the rotation angle, t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with
          // zeta = (beta - alpha) / (2 gamma), written with one division and one square root
          const double d = beta - alpha, g2 = 2.0 * gamma;
          const double sgn = d * g2 >= 0.0 ? 1.0 : -1.0;
          const double t = sgn * fabs(g2) / (fabs(d) + sqrt(fma(d, d, g2 * g2)));
          const double c = rsqrt(fma(t, t, 1.0)), s = c * t;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const double ap = a[r][p], aq = a[r][q];
            a[r][p] = c * ap - s * aq;
            a[r][q] = s * ap + c * aq;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double vp = v[r][p], vq = v[r][q];
            v[r][p] = c * vp - s * vq;
            v[r][q] = s * vp + c * vq;
          }
        }
      }
    if (!__any(rotated)) break;
  }
  // smallest singular value = smallest column norm
  double best = 0.0;
  int bc = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double nrm = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) nrm = fma(a[r][c], a[r][c], nrm);
    if (c == 0 || nrm < best) {
      best = nrm;
      bc = c;
    }
  }
  double xv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) xv[r] = bc == 0 ? v[r][0] : (bc == 1 ? v[r][1] : (bc == 2 ? v[r][2] : v[r][3]));
  if (live) {
    float4 o;
    o.x = (float)(xv[0] / xv[3]);
    o.y = (float)(xv[1] / xv[3]);
    o.z = (float)(xv[2] / xv[3]);
    o.w = (float)(xv[3] / xv[3]);
    *reinterpret_cast<float4*>(Xh + 4 * i) = o;
  }
}

int launch_triangulate(const double* x1, const double* x2, const double* P1, const double* P2, int64_t n, float* Xh,
                       hipStream_t st) {
  if (n <= 0) return SFM_BA_OK;
  {
    ProfScope ps(KID_TRIANGULATE, st);
    hipLaunchKernelGGL(k_triangulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x1, x2, P1, P2, n, Xh);
  }
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, "k_triangulate launch");
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

}  // namespace sfmba
