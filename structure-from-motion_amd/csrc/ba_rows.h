// Rows of A formed from the state instead of read from memory, shared by the kernels that do so
// (ba_fused.hip: the point-major assembly and back-substitution; ba_kernels.hip: the camera-pair kernel).
#pragma once

#include "ba_common.h"

namespace sfmba {

// 1 / x to full precision: hardware estimate and two Newton steps (x is a depth or its square: never 0)
__device__ __forceinline__ double fz_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

// Row of A and residual of one observation, formed where they are used instead of being read back
// from memory: the arithmetic of jacobian_rows (src/jacobian.py:21-220) and of the projection
// (src/utils.py:8-11) on the camera's prepared record `rec` (56 doubles) and projection matrix `pm`
// (12) in LDS, with one reciprocal per quotient rule instead of fourteen divisions (a few units in
// the last place from the rows k_jacobian writes).  av[0..6] = row of A as k_jacobian lays it out,
// av[7] = x_true - x_pred; zeros for a lane without an observation.
template <bool RESIDUAL = true, bool FENCED = false, typename RecT = const double2*, typename PmT = const double2*>
__device__ __forceinline__ void fz_rows_from_state(RecT r2, PmT p2, double x0, double x1, double x2, double2 xt,
                                                   bool has, double2 (&av)[9]) {
  // (FENCED: in four parts fenced off from each other.  Left alone, the scheduler moves all 34 LDS reads
  // to the front: 68 registers, which a kernel compiled for two waves per SIMD does not have to spare;
  // a kernel with one wave per SIMD wants exactly that, nothing else hides the latency of the reads)
  const double z = 0.0;
  double d0, d1, d2, wi, ui, vi;
  {
    const double2 c01 = r2[0], c23 = r2[1], c45 = r2[2], c67 = r2[3], c89 = r2[4], cab = r2[5];
    const double mu0 = c01.x, mu1 = c01.y, mu2 = c23.x, mv0 = c23.y, mv1 = c45.x, mv2 = c45.y;
    const double mw0 = c67.x, mw1 = c67.y, mw2 = c89.x;
    d0 = x0 - c89.y;
    d1 = x1 - cab.x;
    d2 = x2 - cab.y;
    const double u = mu0 * d0 + mu1 * d1 + mu2 * d2;
    const double v = mv0 * d0 + mv1 * d1 + mv2 * d2;
    const double w = mw0 * d0 + mw1 * d1 + mw2 * d2;
    const double iw2 = fz_rcp(w * w);
    // a lane without an observation: zero scale factors make every entry below an exact zero (the
    // other factors are finite: its record and point are those of existing cameras and points)
    wi = has ? w * iw2 : z;
    ui = has ? u * iw2 : z;
    vi = has ? v * iw2 : z;
    // dF/dX rows (the point block); dF/dC = -dF/dX sits in columns 4..6 of A
    const double b00 = fma(wi, mu0, -ui * mw0), b01 = fma(wi, mu1, -ui * mw1), b02 = fma(wi, mu2, -ui * mw2);
    const double b10 = fma(wi, mv0, -vi * mw0), b11 = fma(wi, mv1, -vi * mw1), b12 = fma(wi, mv2, -vi * mw2);
    av[2] = make_double2(-b00, -b01);
    av[3].x = -b02;
    av[5].y = -b10;
    av[6] = make_double2(-b11, -b12);
  }
  if (FENCED) __builtin_amdgcn_sched_barrier(0);
  const double2 f01 = r2[26], f23 = r2[27];  // fx fy | ox oy
  if (FENCED) {
    // one column of the 9 x 4 table dR/dQ at a time (entry (3 r + i, q) at doubles 12 + 4 (3 r + i) + q):
    // nine 8-byte reads in flight instead of eighteen 16-byte ones
    const double* rd = reinterpret_cast<const double*>(&r2[0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double g[3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
        g[r] = d0 * rd[12 + 4 * (3 * r + 0) + q] + d1 * rd[12 + 4 * (3 * r + 1) + q] + d2 * rd[12 + 4 * (3 * r + 2) + q];
      const double du = f01.x * g[0] + f23.x * g[2];
      const double dv = f01.y * g[1] + f23.y * g[2];
      const double a0 = fma(wi, du, -ui * g[2]);
      const double a1 = fma(wi, dv, -vi * g[2]);
      if (q == 0) { av[0].x = a0; av[3].y = a1; }
      if (q == 1) { av[0].y = a0; av[4].x = a1; }
      if (q == 2) { av[1].x = a0; av[4].y = a1; }
      if (q == 3) { av[1].y = a0; av[5].x = a1; }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // columns 2 h, 2 h + 1 of the 9 x 4 table dR/dQ: row 3 r + i at doubles 12 + 4 (3 r + i)
    double g[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double2 t0 = r2[6 + 2 * (3 * r + 0) + h], t1 = r2[6 + 2 * (3 * r + 1) + h], t2 = r2[6 + 2 * (3 * r + 2) + h];
      g[r][0] = d0 * t0.x + d1 * t1.x + d2 * t2.x;
      g[r][1] = d0 * t0.y + d1 * t1.y + d2 * t2.y;
    }
    double a0[2], a1[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const double du = f01.x * g[0][c] + f23.x * g[2][c];
      const double dv = f01.y * g[1][c] + f23.y * g[2][c];
      a0[c] = fma(wi, du, -ui * g[2][c]);
      a1[c] = fma(wi, dv, -vi * g[2][c]);
    }
    if (h == 0) {
      av[0] = make_double2(a0[0], a0[1]);
      av[3].y = a1[0];
      av[4].x = a1[1];
    } else {
      av[1] = make_double2(a0[0], a0[1]);
      av[4].y = a1[0];
      av[5].x = a1[1];
    }
  }
  }
  if (RESIDUAL) {
    const double2 p01 = p2[0], p23 = p2[1], p45 = p2[2], p67 = p2[3], p89 = p2[4], pab = p2[5];
    const double h0 = p01.x * x0 + p01.y * x1 + p23.x * x2 + p23.y;
    const double h1 = p45.x * x0 + p45.y * x1 + p67.x * x2 + p67.y;
    const double h2 = p89.x * x0 + p89.y * x1 + pab.x * x2 + pab.y;
    const double ih2 = fz_rcp(h2);
    av[7] = make_double2(has ? xt.x - h0 * ih2 : z, has ? xt.y - h1 * ih2 : z);
  }
}


}  // namespace sfmba
