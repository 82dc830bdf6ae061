// Bundle-adjustment kernels for gfx950 (MI355X): per-camera records, per-observation
// Jacobians, Schur assembly, back-substitution, camera update and cost.  fp64 throughout.
//
// Layout conventions
//   A [N,2,7]  row a (0 = x pixel, 1 = y pixel) of observation i starts at A + 14*i + 7*a
//   B [N,2,3]  row a at B + 6*i + 3*a
//   "slot"     position of an observation in the plan's point-major order; perm[slot] is
//              its position in the caller's order, which is the order of A, B, x_true, x_pred
//   S          dense (7m+1) x 7m row-major; row 7m carries the right-hand side e
//
// Every reduction runs in a fixed order (no atomics), so results are reproducible bit for bit.
#include <cstdlib>
#include <mutex>

#include "ba_common.h"
#include "ba_rows.h"

namespace sfmba {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void st2(double* p, double a, double b) {
  *reinterpret_cast<double2*>(p) = make_double2(a, b);
}

// Rotation matrix (row-major) -> quaternion [x,y,z,w]; the four cases of the reference's
// conversion, selected by the same comparisons.
__device__ __forceinline__ void rot_to_quat(const double* R, double* Q) {
  const double tr = R[0] + R[4] + R[8];
  double x, y, z, w;
  if (tr > 0.0) {
    const double s = 0.5 / sqrt(tr + 1.0);
    w = 0.25 / s;
    x = (R[7] - R[5]) * s;
    y = (R[2] - R[6]) * s;
    z = (R[3] - R[1]) * s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = 2.0 * sqrt(1.0 + R[0] - R[4] - R[8]);
    w = (R[7] - R[5]) / s;
    x = 0.25 * s;
    y = (R[1] + R[3]) / s;
    z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = 2.0 * sqrt(1.0 + R[4] - R[0] - R[8]);
    w = (R[2] - R[6]) / s;
    x = (R[1] + R[3]) / s;
    y = 0.25 * s;
    z = (R[5] + R[7]) / s;
  } else {
    const double s = 2.0 * sqrt(1.0 + R[8] - R[0] - R[4]);
    w = (R[3] - R[1]) / s;
    x = (R[2] + R[6]) / s;
    y = (R[5] + R[7]) / s;
    z = 0.25 * s;
  }
  Q[0] = x; Q[1] = y; Q[2] = z; Q[3] = w;
}

__device__ __forceinline__ void quat_to_rot(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double xx = x * x, xy = x * y, xz = x * z, xw = x * w;
  const double yy = y * y, yz = y * z, yw = y * w, zz = z * z, zw = z * w;
  R[0] = 1 - 2 * (yy + zz); R[1] = 2 * (xy - zw);     R[2] = 2 * (xz + yw);
  R[3] = 2 * (xy + zw);     R[4] = 1 - 2 * (xx + zz); R[5] = 2 * (yz - xw);
  R[6] = 2 * (xz - yw);     R[7] = 2 * (yz + xw);     R[8] = 1 - 2 * (xx + yy);
}

__device__ __forceinline__ double det3(const double* R) {
  return R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
         R[2] * (R[3] * R[7] - R[4] * R[6]);
}

// ------------------------------------------------------------------------------------------
// per-camera record
// ------------------------------------------------------------------------------------------
// prepared record of one camera (layout: see sfm_ba_camera_prepare) from its rotation rr, centre c and
// intrinsics k; returns the flag bits of the camera
__device__ __forceinline__ int camera_record(const double* rr, const double* c, const double* k, double* o) {
  const double fx = k[0], fy = k[4], ox = k[2], oy = k[5];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[i] = fx * rr[i] + ox * rr[6 + i];
    o[3 + i] = fy * rr[3 + i] + oy * rr[6 + i];
    o[6 + i] = rr[6 + i];
    o[9 + i] = c[i];
  }
  double q[4];
  rot_to_quat(rr, q);
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  // dR/dQ, rows R11 R12 R13 R21 R22 R23 R31 R32 R33, columns d/dx d/dy d/dz d/dw,
  // with the signs of the reference's table
  const double g[36] = {
      0, -4 * y, -4 * z, 0,           //
      2 * y, 2 * x, -2 * w, 2 * z,    //
      2 * z, 2 * w, 2 * x, 2 * y,     //
      2 * y, 2 * x, 2 * w, 2 * z,     //
      -4 * x, 0, -4 * z, 0,           //
      -2 * w, 2 * z, 2 * y, 2 * x,    //
      2 * z, -2 * w, -2 * x, -2 * y,  //
      2 * w, 2 * z, 2 * y, 2 * x,     //
      -4 * x, -4 * y, 0, 0};
#pragma unroll
  for (int i = 0; i < 36; ++i) o[12 + i] = g[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[48 + i] = q[i];
  o[52] = fx; o[53] = fy; o[54] = ox; o[55] = oy;
  int f = 0;
  const double det = det3(rr);
  if (!(fabs(det - 1.0) <= 1e-8 + 1e-5)) f |= SFM_BA_FLAG_DET_NOT_ONE;
  double chk = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) chk += o[i];
  if (!isfinite(chk)) f |= SFM_BA_FLAG_NONFINITE;
  return f;
}

__global__ void k_camera_prepare(const double* __restrict__ R, const double* __restrict__ C,
                                 const double* __restrict__ K, int k_stride, int m,
                                 double* __restrict__ blk, int32_t* __restrict__ flags) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double rr[9], cc[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) rr[i] = R[9 * (int64_t)j + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) cc[i] = C[3 * (int64_t)j + i];
  const int f = camera_record(rr, cc, K + (int64_t)k_stride * j, blk + (int64_t)SFM_BA_CAM_BLK * j);
  if (flags && f) flags[j] |= f;  // sticky: the caller clears
}

// ------------------------------------------------------------------------------------------
// per-observation Jacobian blocks and prediction: one lane per observation.  The observation
// list is sorted by camera in the reference's order, so almost every wave sees one camera: its
// record is then read through the scalar cache (one fetch per wave instead of 56 doubles per
// lane through the vector cache).  A and B rows are parked in the wave's LDS slice and written
// out as whole 1 KiB wave-rows (lane l stores piece p*64+l), not as 112/48-byte rows per lane.
// ------------------------------------------------------------------------------------------
struct JacOut {
  double a0[4], a1[4], b0[3], b1[3];
};

template <typename RecPtr>
__device__ __forceinline__ void jacobian_rows(RecPtr cb, double x0, double x1, double x2, JacOut& o) {
  double mu_[3], mv_[3], mw_[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    mu_[k] = cb[k];
    mv_[k] = cb[3 + k];
    mw_[k] = cb[6 + k];
  }
  d[0] = x0 - cb[9];
  d[1] = x1 - cb[10];
  d[2] = x2 - cb[11];
  const double u = mu_[0] * d[0] + mu_[1] * d[1] + mu_[2] * d[2];
  const double v = mv_[0] * d[0] + mv_[1] * d[1] + mv_[2] * d[2];
  const double w = mw_[0] * d[0] + mw_[1] * d[1] + mw_[2] * d[2];
  const double w2 = w * w;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o.b0[k] = (w * mu_[k] - u * mw_[k]) / w2;
    o.b1[k] = (w * mv_[k] - v * mw_[k]) / w2;
  }
  const double fx = cb[52], fy = cb[53], ox = cb[54], oy = cb[55];
  // gq[r][q] = sum_i d[i] * dRdQ[3r+i][q]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double g0 = d[0] * cb[12 + 0 + q] + d[1] * cb[12 + 4 + q] + d[2] * cb[12 + 8 + q];
    const double g1 = d[0] * cb[12 + 12 + q] + d[1] * cb[12 + 16 + q] + d[2] * cb[12 + 20 + q];
    const double g2 = d[0] * cb[12 + 24 + q] + d[1] * cb[12 + 28 + q] + d[2] * cb[12 + 32 + q];
    const double du = fx * g0 + ox * g2;
    const double dv = fy * g1 + oy * g2;
    o.a0[q] = (w * du - u * g2) / w2;
    o.a1[q] = (w * dv - v * g2) / w2;
  }
}

__global__ __launch_bounds__(256) void k_jacobian(
    const double* __restrict__ blk, const double* __restrict__ P, const double* __restrict__ X,
    const int32_t* __restrict__ cam, const int32_t* __restrict__ pt, int64_t N,
    double* __restrict__ A, double* __restrict__ B, double* __restrict__ xp) {
  __shared__ __attribute__((aligned(16))) double sh[4][64 * 20];  // per wave: A rows [64][14], B rows [64][6]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wbase = (int64_t)blockIdx.x * 256 + 64 * wave;  // first observation of this wave
  if (wbase >= N) return;
  const int64_t i = wbase + lane;
  const bool live = i < N;
  const int64_t ic = live ? i : N - 1;
  const int c = cam[ic];
  const int p = pt[ic];
  const double x0 = X[3 * (int64_t)p], x1 = X[3 * (int64_t)p + 1], x2 = X[3 * (int64_t)p + 2];
  const int c_first = __builtin_amdgcn_readfirstlane(c);
  const bool uniform = __all(c == c_first);

  if (xp) {
    double h0, h1, h2;
    if (uniform) {
      const double* pm = P + 12 * (int64_t)c_first;
      h0 = pm[0] * x0 + pm[1] * x1 + pm[2] * x2 + pm[3];
      h1 = pm[4] * x0 + pm[5] * x1 + pm[6] * x2 + pm[7];
      h2 = pm[8] * x0 + pm[9] * x1 + pm[10] * x2 + pm[11];
    } else {
      const double* pm = P + 12 * (int64_t)c;
      h0 = pm[0] * x0 + pm[1] * x1 + pm[2] * x2 + pm[3];
      h1 = pm[4] * x0 + pm[5] * x1 + pm[6] * x2 + pm[7];
      h2 = pm[8] * x0 + pm[9] * x1 + pm[10] * x2 + pm[11];
    }
    if (live) st2(xp + 2 * i, h0 / h2, h1 / h2);
  }
  if (!A && !B) return;

  JacOut o;
  if (uniform)
    jacobian_rows(blk + (int64_t)SFM_BA_CAM_BLK * c_first, x0, x1, x2, o);
  else
    jacobian_rows(blk + (int64_t)SFM_BA_CAM_BLK * c, x0, x1, x2, o);

  double* sa = sh[wave];
  double* sb = sa + 64 * 14;
  {
    double* ra = sa + 14 * lane;
    st2(ra, o.a0[0], o.a0[1]);
    st2(ra + 2, o.a0[2], o.a0[3]);
    st2(ra + 4, -o.b0[0], -o.b0[1]);
    st2(ra + 6, -o.b0[2], o.a1[0]);
    st2(ra + 8, o.a1[1], o.a1[2]);
    st2(ra + 10, o.a1[3], -o.b1[0]);
    st2(ra + 12, -o.b1[1], -o.b1[2]);
    double* rb = sb + 6 * lane;
    st2(rb, o.b0[0], o.b0[1]);
    st2(rb + 2, o.b0[2], o.b1[0]);
    st2(rb + 4, o.b1[1], o.b1[2]);
  }
  // same wave wrote and reads: LDS operations of one wave complete in order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const int nlive = (N - wbase) < 64 ? (int)(N - wbase) : 64;
  if (A) {
    double* ga = A + 14 * wbase;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int piece = 64 * q + lane;  // 16-byte piece of the wave's [64][14] block
      if (piece < 7 * nlive) {
        const double2 v2 = ld2(sa + 2 * piece);
        st2(ga + 2 * piece, v2.x, v2.y);
      }
    }
  }
  if (B) {
    double* gb = B + 6 * wbase;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int piece = 64 * q + lane;
      if (piece < 3 * nlive) {
        const double2 v2 = ld2(sb + 2 * piece);
        st2(gb + 2 * piece, v2.x, v2.y);
      }
    }
  }
}

// One Jacobi rotation of a symmetric 3 x 3 matrix that annihilates a_pq; r is the third index.
// vp, vq: the matching columns of the accumulated rotations.
__device__ inline void jacobi_rot3(double& app, double& aqq, double& apq, double& arp, double& arq,
                                   double (&vp)[3], double (&vq)[3]) {
  if (apq == 0.0) return;
  const double theta = (aqq - app) / (2.0 * apq);
  const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
  app -= tt * apq;
  aqq += tt * apq;
  apq = 0.0;
  const double rp = c * arp - sn * arq, rq = sn * arp + c * arq;
  arp = rp;
  arq = rq;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double a = vp[i], b = vq[i];
    vp[i] = c * a - sn * b;
    vq[i] = sn * a + c * b;
  }
}

// Moore-Penrose inverse of a symmetric 3 x 3 matrix (00 01 02 11 12 22 in, same out) as
// numpy.linalg.pinv forms it (src/sparse_ba.py:201-212): eigenvalues whose magnitude is at most
// 1e-15 times the largest are dropped.  Cyclic Jacobi sweeps; taken only for (nearly) rank-deficient
// V_i, e.g. a point seen by a single camera with no damping.
__device__ __noinline__ void pinv_sym3(double a00, double a01, double a02, double a11, double a12, double a22,
                                       double (&out)[6]) {
  double v0[3] = {1, 0, 0}, v1[3] = {0, 1, 0}, v2[3] = {0, 0, 1};
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12);
    if (off == 0.0 || off <= 1e-300) break;
    jacobi_rot3(a00, a11, a01, a02, a12, v0, v1);
    jacobi_rot3(a00, a22, a02, a01, a12, v0, v2);
    jacobi_rot3(a11, a22, a12, a01, a02, v1, v2);
  }
  const double big = fmax(fabs(a00), fmax(fabs(a11), fabs(a22)));
  const double cut = 1e-15 * big;
  const double i0 = fabs(a00) > cut ? 1.0 / a00 : 0.0;
  const double i1 = fabs(a11) > cut ? 1.0 / a11 : 0.0;
  const double i2 = fabs(a22) > cut ? 1.0 / a22 : 0.0;
  out[0] = i0 * v0[0] * v0[0] + i1 * v1[0] * v1[0] + i2 * v2[0] * v2[0];
  out[1] = i0 * v0[0] * v0[1] + i1 * v1[0] * v1[1] + i2 * v2[0] * v2[1];
  out[2] = i0 * v0[0] * v0[2] + i1 * v1[0] * v1[2] + i2 * v2[0] * v2[2];
  out[3] = i0 * v0[1] * v0[1] + i1 * v1[1] * v1[1] + i2 * v2[1] * v2[1];
  out[4] = i0 * v0[1] * v0[2] + i1 * v1[1] * v1[2] + i2 * v2[1] * v2[2];
  out[5] = i0 * v0[2] * v0[2] + i1 * v1[2] * v1[2] + i2 * v2[2] * v2[2];
}

// What one observation brings to its point: the point block bb (B, or minus the camera-centre
// columns of A when no B is given), the weight w, bp = w bb and we = w (x_true - x_pred).
struct ObsTerms {
  double bb[6], bp[6], we[2];
  double w00, w01, w10, w11;
  double e0, e1;  // the residual itself
};

__device__ inline void obs_terms(int64_t pos, const double* __restrict__ A, const double* __restrict__ B,
                                 const double* __restrict__ xt, const double* __restrict__ xpred,
                                 const double* __restrict__ wgt, ObsTerms& o) {
  if (B) {
    const double* b = B + 6 * pos;
    const double2 b01 = ld2(b), b23 = ld2(b + 2), b45 = ld2(b + 4);
    o.bb[0] = b01.x; o.bb[1] = b01.y; o.bb[2] = b23.x; o.bb[3] = b23.y; o.bb[4] = b45.x; o.bb[5] = b45.y;
  } else {
    const double* ap = A + 14 * pos;
    const double2 a45 = ld2(ap + 4), a1213 = ld2(ap + 12);
    o.bb[0] = -a45.x; o.bb[1] = -a45.y; o.bb[2] = -ap[6]; o.bb[3] = -ap[11]; o.bb[4] = -a1213.x; o.bb[5] = -a1213.y;
  }
  const double2 xa = ld2(xt + 2 * pos), xb = ld2(xpred + 2 * pos);
  const double e0 = xa.x - xb.x, e1 = xa.y - xb.y;
  o.w00 = 1.0; o.w01 = 0.0; o.w10 = 0.0; o.w11 = 1.0;
  if (wgt) {
    const double* wp = wgt + 4 * pos;
    o.w00 = wp[0]; o.w01 = wp[1]; o.w10 = wp[2]; o.w11 = wp[3];
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    o.bp[q] = o.w00 * o.bb[q] + o.w01 * o.bb[3 + q];
    o.bp[3 + q] = o.w10 * o.bb[q] + o.w11 * o.bb[3 + q];
  }
  o.we[0] = o.w00 * e0 + o.w01 * e1;
  o.we[1] = o.w10 * e0 + o.w11 * e1;
  o.e0 = e0;
  o.e1 = e1;
}

// V^-1 (00 01 02 11 12 22) of the damped point block from the sums a[0..5] = B^T w B.
__device__ inline void point_inverse(const double* a, double mu, double (&inv)[6]) {
  const double v00 = a[0] + mu, v01 = a[1], v02 = a[2], v11 = a[3] + mu, v12 = a[4], v22 = a[5] + mu;
  const double c00 = v11 * v22 - v12 * v12;
  const double c01 = v02 * v12 - v01 * v22;
  const double c02 = v01 * v12 - v02 * v11;
  const double c11 = v00 * v22 - v02 * v02;
  const double c12 = v01 * v02 - v00 * v12;
  const double c22 = v00 * v11 - v01 * v01;
  const double det = v00 * c00 + v01 * c01 + v02 * c02;
  const double id = 1.0 / det;
  inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id; inv[3] = c11 * id; inv[4] = c12 * id; inv[5] = c22 * id;
  // (nearly) rank-deficient V_i - cond(V_i) beyond ~1e10, where the cofactor form has lost its
  // digits, or det <= 0: the pseudo-inverse, as the reference's pinv gives it
  const double tr = (v00 + v11 + v22) * (1.0 / 3.0);
  if (!(det > 1e-10 * tr * tr * tr)) pinv_sym3(v00, v01, v02, v11, v12, v22, inv);
}

// A point with more observations than a workgroup has lanes (its range holds this point alone):
// the lanes stride over the observations twice - sums for V_i and eps_b_i, reduced through LDS in a
// fixed order, then g and the k (k + 1) / 2 couplings, the second observation's terms re-read.
__device__ __noinline__ void point_long(int t, int p0, int pid, int s0, int ns, const int32_t* __restrict__ perm,
                                        const int32_t* __restrict__ tp_ptr, const int32_t* __restrict__ pair_entry,
                                        const double* __restrict__ A, const double* __restrict__ B,
                                        const double* __restrict__ xt, const double* __restrict__ xpred,
                                        const double* __restrict__ wgt, double mu, int fold_u,
                                        double* __restrict__ Vinv, double* __restrict__ eps_b, double* __restrict__ g,
                                        double* __restrict__ qsq, double* __restrict__ T, const double* __restrict__ X,
                                        double* __restrict__ prec, double* sRed /* 256 x 9 */) {
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = t; s < ns; s += 256) {
    ObsTerms o;
    obs_terms(perm[s0 + s], A, B, xt, xpred, wgt, o);
    acc[0] += o.bb[0] * o.bp[0] + o.bb[3] * o.bp[3];
    acc[1] += o.bb[0] * o.bp[1] + o.bb[3] * o.bp[4];
    acc[2] += o.bb[0] * o.bp[2] + o.bb[3] * o.bp[5];
    acc[3] += o.bb[1] * o.bp[1] + o.bb[4] * o.bp[4];
    acc[4] += o.bb[1] * o.bp[2] + o.bb[4] * o.bp[5];
    acc[5] += o.bb[2] * o.bp[2] + o.bb[5] * o.bp[5];
    acc[6] += o.bb[0] * o.we[0] + o.bb[3] * o.we[1];
    acc[7] += o.bb[1] * o.we[0] + o.bb[4] * o.we[1];
    acc[8] += o.bb[2] * o.we[0] + o.bb[5] * o.we[1];
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) sRed[9 * t + q] = acc[q];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
#pragma unroll
      for (int q = 0; q < 9; ++q) sRed[9 * t + q] += sRed[9 * (t + o) + q];
    }
    __syncthreads();
  }
  double a[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) a[q] = sRed[q];
  double iv[6];
  point_inverse(a, mu, iv);  // every lane, same numbers
  const double z0 = iv[0] * a[6] + iv[1] * a[7] + iv[2] * a[8];
  const double z1 = iv[1] * a[6] + iv[3] * a[7] + iv[4] * a[8];
  const double z2 = iv[2] * a[6] + iv[4] * a[7] + iv[5] * a[8];
  if (t == 0) {
    double* vo = Vinv + 9 * (int64_t)pid;
    vo[0] = iv[0]; vo[1] = iv[1]; vo[2] = iv[2];
    vo[3] = iv[1]; vo[4] = iv[3]; vo[5] = iv[4];
    vo[6] = iv[2]; vo[7] = iv[4]; vo[8] = iv[5];
    double* eo = eps_b + 3 * (int64_t)pid;
    eo[0] = a[6]; eo[1] = a[7]; eo[2] = a[8];
    if (prec) {
      double* pr = prec + kPointRec * (int64_t)p0;
      const double* xq = X + 3 * (int64_t)pid;
      pr[0] = xq[0]; pr[1] = xq[1]; pr[2] = xq[2];
      pr[3] = iv[0]; pr[4] = iv[1]; pr[5] = iv[2]; pr[6] = iv[3]; pr[7] = iv[4]; pr[8] = iv[5];
    }
  }
  const int64_t k = ns;
  for (int ai = t; ai < ns; ai += 256) {
    ObsTerms o;
    const int64_t pos = perm[s0 + ai];
    obs_terms(pos, A, B, xt, xpred, wgt, o);
    st2(g + 2 * pos, o.we[0] - (o.bp[0] * z0 + o.bp[1] * z1 + o.bp[2] * z2),
        o.we[1] - (o.bp[3] * z0 + o.bp[4] * z1 + o.bp[5] * z2));
    st2(qsq + 2 * pos, o.e0 * o.e0 + o.e1 * o.e1, 0.0);
    const double h00 = o.bp[0] * iv[0] + o.bp[1] * iv[1] + o.bp[2] * iv[2];
    const double h01 = o.bp[0] * iv[1] + o.bp[1] * iv[3] + o.bp[2] * iv[4];
    const double h02 = o.bp[0] * iv[2] + o.bp[1] * iv[4] + o.bp[2] * iv[5];
    const double h10 = o.bp[3] * iv[0] + o.bp[4] * iv[1] + o.bp[5] * iv[2];
    const double h11 = o.bp[3] * iv[1] + o.bp[4] * iv[3] + o.bp[5] * iv[4];
    const double h12 = o.bp[3] * iv[2] + o.bp[4] * iv[4] + o.bp[5] * iv[5];
    if (!T) continue;  // (the couplings are formed by the camera-pair kernel itself)
    int64_t ts = (int64_t)tp_ptr[p0] + (int64_t)ai * k - (int64_t)ai * (ai - 1) / 2;
    for (int b = ai; b < ns; ++b, ++ts) {
      ObsTerms ob;
      if (b == ai)
        ob = o;
      else
        obs_terms(perm[s0 + b], A, B, xt, xpred, wgt, ob);
      const double* q = ob.bp;
      double* out = T + 4 * (int64_t)pair_entry[ts];
      double t00 = h00 * q[0] + h01 * q[1] + h02 * q[2], t01 = h00 * q[3] + h01 * q[4] + h02 * q[5];
      double t10 = h10 * q[0] + h11 * q[1] + h12 * q[2], t11 = h10 * q[3] + h11 * q[4] + h12 * q[5];
      if (fold_u && b == ai) {
        t00 -= o.w00; t01 -= o.w01; t10 -= o.w10; t11 -= o.w11;
      }
      st2(out, t00, t01);
      st2(out + 2, t10, t11);
    }
  }
}

// ------------------------------------------------------------------------------------------
// per-point stage: V_i, V_i^-1, eps_b_i, then per observation g = w (eps - B V^-1 eps_b) and per
// observation pair (a <= b) of the point the 2x2 coupling T_ab = (wB)_a V^-1 (wB)_b^T.
// One workgroup owns a run of points with at most kGroupSlots observations; one lane per
// observation, then one lane per point.  pt_map (may be null): id under which point p of the plan
// keeps its V^-1 and eps_b (the far plan of a larger plan numbers its points from 0).
// T may be null: no couplings are written.  prec (may be null, needs X): per point of the plan its
// position and V^-1 (00 01 02 11 12 22) as one record of kPointRec doubles, what the camera-pair kernel
// that forms its rows itself (k_pair_state) fetches per entry.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_point(
    const int32_t* __restrict__ pt_map, const int32_t* __restrict__ grp_pt, const int32_t* __restrict__ pt_ptr,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ tp_ptr,
    const int32_t* __restrict__ pair_entry, const double* __restrict__ A,
    const double* __restrict__ B, const double* __restrict__ xt, const double* __restrict__ xpred,
    const double* __restrict__ wgt, double mu, int fold_u, double* __restrict__ Vinv,
    double* __restrict__ eps_b, double* __restrict__ g, double* __restrict__ qsq, double* __restrict__ T,
    const double* __restrict__ X, double* __restrict__ prec) {
  // qsq: per observation {squared residual, 0}, what the camera-pair kernel adds up per camera for the cost
  __shared__ double sBp[kGroupSlots * 6];
  __shared__ double sCon[kGroupSlots * 9];
  __shared__ int sPtr[kGroupSlots + 1];
  __shared__ int sLp[kGroupSlots];
  const int t = threadIdx.x;
  const int p0 = grp_pt[blockIdx.x];
  const int np = grp_pt[blockIdx.x + 1] - p0;
  for (int i = t; i <= np; i += 256) sPtr[i] = pt_ptr[p0 + i];
  __syncthreads();
  const int s0 = sPtr[0];
  const int ns = sPtr[np] - s0;
  if (ns > kGroupSlots) {  // one point with more observations than lanes (same for the whole workgroup)
    point_long(t, p0, pt_map ? pt_map[p0] : p0, s0, ns, perm, tp_ptr, pair_entry, A, B, xt, xpred, wgt, mu, fold_u, Vinv,
               eps_b, g, qsq, T, X, prec, sCon);
    return;
  }
  if (t < np)
    for (int s = sPtr[t]; s < sPtr[t + 1]; ++s) sLp[s - s0] = t;

  double bp[6], we[2];
  double w00 = 1.0, w01 = 0.0, w10 = 0.0, w11 = 1.0;
  int pos = 0;
  if (t < ns) {
    pos = perm[s0 + t];
    double bb[6];
    if (B) {
      const double* b = B + 6 * (int64_t)pos;
      const double2 b01 = ld2(b), b23 = ld2(b + 2), b45 = ld2(b + 4);
      bb[0] = b01.x; bb[1] = b01.y; bb[2] = b23.x; bb[3] = b23.y; bb[4] = b45.x; bb[5] = b45.y;
    } else {
      // no B given: the point block is minus the camera-centre block of A (columns 4..6 of both rows)
      const double* ap = A + 14 * (int64_t)pos;
      const double2 a45 = ld2(ap + 4), a1213 = ld2(ap + 12);
      bb[0] = -a45.x; bb[1] = -a45.y; bb[2] = -ap[6]; bb[3] = -ap[11]; bb[4] = -a1213.x; bb[5] = -a1213.y;
    }
    const double2 xa = ld2(xt + 2 * (int64_t)pos), xb = ld2(xpred + 2 * (int64_t)pos);
    const double e0 = xa.x - xb.x, e1 = xa.y - xb.y;
    st2(qsq + 2 * (int64_t)pos, e0 * e0 + e1 * e1, 0.0);
    if (wgt) {
      const double* wp = wgt + 4 * (int64_t)pos;
      w00 = wp[0]; w01 = wp[1]; w10 = wp[2]; w11 = wp[3];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      bp[q] = w00 * bb[q] + w01 * bb[3 + q];
      bp[3 + q] = w10 * bb[q] + w11 * bb[3 + q];
    }
    we[0] = w00 * e0 + w01 * e1;
    we[1] = w10 * e0 + w11 * e1;
    double* c = sCon + 9 * t;
    // B^T w B (symmetric: 00 01 02 11 12 22) and B^T w eps
    c[0] = bb[0] * bp[0] + bb[3] * bp[3];
    c[1] = bb[0] * bp[1] + bb[3] * bp[4];
    c[2] = bb[0] * bp[2] + bb[3] * bp[5];
    c[3] = bb[1] * bp[1] + bb[4] * bp[4];
    c[4] = bb[1] * bp[2] + bb[4] * bp[5];
    c[5] = bb[2] * bp[2] + bb[5] * bp[5];
    c[6] = bb[0] * we[0] + bb[3] * we[1];
    c[7] = bb[1] * we[0] + bb[4] * we[1];
    c[8] = bb[2] * we[0] + bb[5] * we[1];
#pragma unroll
    for (int q = 0; q < 6; ++q) sBp[6 * t + q] = bp[q];
  }
  __syncthreads();
  if (t < np) {
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = sPtr[t] - s0; s < sPtr[t + 1] - s0; ++s) {
#pragma unroll
      for (int q = 0; q < 9; ++q) a[q] += sCon[9 * s + q];
    }
    double iv[6];
    point_inverse(a, mu, iv);
    const double i00 = iv[0], i01 = iv[1], i02 = iv[2], i11 = iv[3], i12 = iv[4], i22 = iv[5];
    // V^-1 and V^-1 eps_b of the point go where the contributions of its first observation were:
    // only this thread has read them (a point without observations has no such place and nobody
    // reads its record)
    if (sPtr[t + 1] > sPtr[t]) {
      double* vz = sCon + 9 * (sPtr[t] - s0);
      vz[0] = i00; vz[1] = i01; vz[2] = i02; vz[3] = i11; vz[4] = i12; vz[5] = i22;
      vz[6] = i00 * a[6] + i01 * a[7] + i02 * a[8];
      vz[7] = i01 * a[6] + i11 * a[7] + i12 * a[8];
      vz[8] = i02 * a[6] + i12 * a[7] + i22 * a[8];
    }
    const int pid = pt_map ? pt_map[p0 + t] : p0 + t;
    double* vo = Vinv + 9 * (int64_t)pid;
    vo[0] = i00; vo[1] = i01; vo[2] = i02;
    vo[3] = i01; vo[4] = i11; vo[5] = i12;
    vo[6] = i02; vo[7] = i12; vo[8] = i22;
    double* eo = eps_b + 3 * (int64_t)pid;
    eo[0] = a[6]; eo[1] = a[7]; eo[2] = a[8];
    if (prec) {
      double* pr = prec + kPointRec * (int64_t)(p0 + t);
      const double* xq = X + 3 * (int64_t)pid;
      pr[0] = xq[0]; pr[1] = xq[1]; pr[2] = xq[2];
      pr[3] = i00; pr[4] = i01; pr[5] = i02; pr[6] = i11; pr[7] = i12; pr[8] = i22;
    }
  }
  __syncthreads();
  if (t < ns) {
    const int lp = sLp[t];
    const double* vz = sCon + 9 * (sPtr[lp] - s0);
    const double i00 = vz[0], i01 = vz[1], i02 = vz[2], i11 = vz[3], i12 = vz[4], i22 = vz[5];
    const double z0 = vz[6], z1 = vz[7], z2 = vz[8];
    st2(g + 2 * (int64_t)pos, we[0] - (bp[0] * z0 + bp[1] * z1 + bp[2] * z2),
        we[1] - (bp[3] * z0 + bp[4] * z1 + bp[5] * z2));
    // H = (wB)_a V^-1  (2x3)
    const double h00 = bp[0] * i00 + bp[1] * i01 + bp[2] * i02;
    const double h01 = bp[0] * i01 + bp[1] * i11 + bp[2] * i12;
    const double h02 = bp[0] * i02 + bp[1] * i12 + bp[2] * i22;
    const double h10 = bp[3] * i00 + bp[4] * i01 + bp[5] * i02;
    const double h11 = bp[3] * i01 + bp[4] * i11 + bp[5] * i12;
    const double h12 = bp[3] * i02 + bp[4] * i12 + bp[5] * i22;
    const int first = sPtr[lp] - s0;
    const int k = T ? sPtr[lp + 1] - sPtr[lp] : 0;  // (no T: the couplings are formed by the camera-pair kernel itself)
    const int a = t - first;
    int64_t ts = (int64_t)tp_ptr[p0 + lp] + (int64_t)a * k - (int64_t)a * (a - 1) / 2;
    // four pairs at a time: their entry positions are fetched together, then the four rows go out
    for (int b0 = a; b0 < k; b0 += 4, ts += 4) {
      int pe[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) pe[u] = b0 + u < k ? pair_entry[ts + u] : -1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + u;
        if (b >= k) break;
        const double* q = sBp + 6 * (first + b);
        double* o = T + 4 * (int64_t)pe[u];  // row of T = position of the pair's entry
        double t00 = h00 * q[0] + h01 * q[1] + h02 * q[2], t01 = h00 * q[3] + h01 * q[4] + h02 * q[5];
        double t10 = h10 * q[0] + h11 * q[1] + h12 * q[2], t11 = h10 * q[3] + h11 * q[4] + h12 * q[5];
        if (fold_u && b == a) {
          // the observation's own pair also carries its term of U_j = sum A^T w A: the camera-pair
          // kernel then produces U_j - sum Y W^T for the diagonal blocks by itself
          t00 -= w00; t01 -= w01; t10 -= w10; t11 -= w11;
        }
        st2(o, t00, t01);
        st2(o + 2, t10, t11);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// per-camera sums over a work item (a chunk of one camera's observations):
//   [0..27] upper triangle of sum A^T w A, [28..34] sum A^T w eps, [35..41] sum A^T g, [42] sum |eps|^2
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCamThreads) void k_camera_items(
    const int32_t* __restrict__ item_beg, const int32_t* __restrict__ item_end,
    const int32_t* __restrict__ cam_obs, const double* __restrict__ A, const double* __restrict__ xt,
    const double* __restrict__ xpred, const double* __restrict__ wgt, const double* __restrict__ g,
    double* __restrict__ part) {
  __shared__ double red[SFM_BA_CAM_RED * (kCamThreads + 1)];
  const int t = threadIdx.x;
  const int beg = item_beg[blockIdx.x], end = item_end[blockIdx.x];
  double acc[SFM_BA_CAM_RED];
#pragma unroll
  for (int v = 0; v < SFM_BA_CAM_RED; ++v) acc[v] = 0.0;
  for (int i = beg + t; i < end; i += kCamThreads) {
    const int64_t pos = cam_obs ? cam_obs[i] : i;
    const double* ap = A + 14 * pos;
    double a[14];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const double2 v2 = ld2(ap + 2 * q);
      a[2 * q] = v2.x;
      a[2 * q + 1] = v2.y;
    }
    const double2 xa = ld2(xt + 2 * pos), xb = ld2(xpred + 2 * pos), gg = ld2(g + 2 * pos);
    const double e0 = xa.x - xb.x, e1 = xa.y - xb.y;
    double w00 = 1.0, w01 = 0.0, w10 = 0.0, w11 = 1.0;
    if (wgt) {
      const double* wp = wgt + 4 * pos;
      w00 = wp[0]; w01 = wp[1]; w10 = wp[2]; w11 = wp[3];
    }
    const double we0 = w00 * e0 + w01 * e1, we1 = w10 * e0 + w11 * e1;
    int v = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const double wa0 = w00 * a[r] + w10 * a[7 + r];  // (A^T w) row r, column 0
      const double wa1 = w01 * a[r] + w11 * a[7 + r];
#pragma unroll
      for (int c = r; c < 7; ++c) acc[v++] += wa0 * a[c] + wa1 * a[7 + c];
    }
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      acc[28 + r] += a[r] * we0 + a[7 + r] * we1;
      acc[35 + r] += a[r] * gg.x + a[7 + r] * gg.y;
    }
    acc[42] += e0 * e0 + e1 * e1;
  }
#pragma unroll
  for (int v = 0; v < SFM_BA_CAM_RED; ++v) red[v * (kCamThreads + 1) + t] = acc[v];
  __syncthreads();
  // two lanes per value, each sums half of the lanes in ascending order, then the halves are added
  __shared__ double half[SFM_BA_CAM_RED * 2];
  if (t < 2 * SFM_BA_CAM_RED) {
    const int v = t >> 1, h = t & 1;
    const double* r = red + v * (kCamThreads + 1) + h * (kCamThreads / 2);
    double s = 0.0;
    for (int i = 0; i < kCamThreads / 2; ++i) s += r[i];
    half[t] = s;
  }
  __syncthreads();
  if (t < SFM_BA_CAM_RED) part[(int64_t)blockIdx.x * SFM_BA_CAM_RED + t] = half[2 * t] + half[2 * t + 1];
}

// one workgroup (64 lanes) per camera: add the camera's item partials in order, add the damping,
// expand U, form e = eps_a - sum Y eps_b, write row 7m of S
__global__ __launch_bounds__(64) void k_camera_finish(
    const int32_t* __restrict__ cam_item_ptr, const double* __restrict__ part, int m, double mu_u,
    double* __restrict__ S, double* __restrict__ e, double* __restrict__ U, double* __restrict__ eps_a,
    double* __restrict__ cam_red) {
  __shared__ double s[SFM_BA_CAM_RED];
  const int j = blockIdx.x, t = threadIdx.x;
  if (t < SFM_BA_CAM_RED) {
    double a = 0.0;
    for (int it = cam_item_ptr[j]; it < cam_item_ptr[j + 1]; ++it) a += part[(int64_t)it * SFM_BA_CAM_RED + t];
    s[t] = a;
    if (cam_red) cam_red[(int64_t)j * SFM_BA_CAM_RED + t] = a;
  }
  __syncthreads();
  if (t < 49 && U) {
    const int r = t / 7, c = t % 7;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int idx = lo * 7 - lo * (lo - 1) / 2 + (hi - lo);
    U[(int64_t)j * 49 + t] = s[idx] + (r == c ? mu_u : 0.0);
  }
  if (t < 7) {
    // e_j = sum A^T g  (g already holds w (eps - B V^-1 eps_b))
    const double ev = s[35 + t];
    if (e) e[(int64_t)j * 7 + t] = ev;
    if (eps_a) eps_a[(int64_t)j * 7 + t] = s[28 + t];
    if (S) S[(int64_t)(7 * m) * (7 * m) + 7 * j + t] = ev;
  }
}

// ------------------------------------------------------------------------------------------
// camera-pair blocks: S_jk = delta_jk (U_j) - sum over entries A_a^T T A_b.  One wave (= one
// workgroup) per work item (a chunk of one block's entries); a step is 16 entries.
//
// Loading: every operand byte is fetched once per wave, straight into the wave's LDS by
// global_load_lds (lane = 16-byte piece of an entry's A_a row, A_b row, T, or of the entry
// records themselves; the pieces land back to back, which is exactly the packed [entry][14] /
// [entry][4] layout).  The loop is latency bound (entry record -> row address -> row), so it runs
// as a software pipeline entirely on the transfer counter: in iteration s the rows of step s are
// consumed while the rows of steps s+1, s+2 and the entry records of steps s+3, s+4 are in
// flight.  Every iteration issues the same 6 transfers (past the item's end they refetch the last
// step), so `s_waitcnt vmcnt(6)` at the top of iteration s means "everything issued up to
// iteration s-2 has landed": the rows of step s and the records of step s+2.  All transfers are
// issued by all 64 lanes with a wave-uniform LDS base (lanes without a piece refetch another
// lane's piece into padding).  The LDS footprint (18 KiB) is what bounds the waves per SIMD, and
// with it the instruction issue rate, so images are kept small rather than deep.
// Arithmetic: lane = 4 * g + h, g = entry of the step, h = column pair {2h, 2h+1} of the block
// (h = 3 owns column 6 and a dummy); a lane keeps the 7 rows of its two columns, and the 16
// partial results are combined at the end by a fixed xor-butterfly.
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// 16 bytes per lane from `base` + `off` (32-bit byte offset per lane) to LDS byte address
// `lds_byte` + 16 * lane (wave-uniform base).  Issued from an asm statement so that the compiler
// does not know a transfer is in flight: it would otherwise guard every later ds_read with
// `s_waitcnt vmcnt(0)`, and the pipeline below counts the transfers itself.  M0 carries the LDS
// base and is restored in the same statement.
__device__ __forceinline__ void dma16(const void* base, unsigned off, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off), "s"(base), "s"(lds_byte)
      : "memory");
}

// Five such transfers with one save / restore of M0: four from `base` to LDS bytes lds_byte + 0,
// 1024, 2048, 3072 and one from `base2` to lds_byte + 4096.  The five LDS addresses are formed by
// the compiler and handed in as scalars: the statement itself touches no status bit.
__device__ __forceinline__ void dma16x5(const void* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3,
                                        const void* base2, unsigned o4, unsigned lds_byte) {
  unsigned keep;
  const unsigned l1 = lds_byte + 1024u, l2 = lds_byte + 2048u, l3 = lds_byte + 3072u, l4 = lds_byte + 4096u;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %6\n\t"
      "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
      "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
      "s_mov_b32 m0, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
      "s_mov_b32 m0, %12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %7\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(o4), "s"(base), "s"(base2), "s"(lds_byte), "s"(l1), "s"(l2),
        "s"(l3), "s"(l4)
      : "memory");
}

// LDS byte address of a pointer into a __shared__ array, as a wave-uniform scalar
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)p);
}

constexpr int kPairStep = 16;  // entries per step
// bytes of one step's row image: A_a rows [16][14] and A_b rows [16][14], each padded to 128
// pieces (2 KiB), then T [16][4] padded to 64 pieces (1 KiB)
constexpr int kPairImgB = 5120;
constexpr int kPairRing = 2;   // row images
constexpr int kPairERing = 4;  // entry-record slots (1 KiB each, 128 B used); a power of two: q % 4 is a mask

__global__ __launch_bounds__(64) void k_pair_blocks(
    int n_items, const int32_t* __restrict__ pit_blk, const int32_t* __restrict__ pit_beg,
    const int32_t* __restrict__ pit_end, const int32_t* __restrict__ pit_part,
    const int32_t* __restrict__ blk_j, const int32_t* __restrict__ blk_k,
    const PairEntry* __restrict__ ent, const double* __restrict__ A, const double* __restrict__ gT,
    unsigned t_off, unsigned q_off, int fold_u, const double* __restrict__ cam_part,
    const int32_t* __restrict__ cam_item_ptr, double mu_u, int m, double* __restrict__ S, double* __restrict__ e_out,
    double* __restrict__ ssq_out, double* __restrict__ spart) {
  // gT: start of the buffer that holds g [N][2] (at byte 0), T [pairs][4] (at byte t_off) and the squared
  // residuals [N][2] (at byte q_off).
  // fold_u: the self pairs carry U_j (see k_point) and the diagonal blocks also sum e_j = A^T g
  // one LDS object for row images and record slots
  __shared__ __attribute__((aligned(16))) double lds[(kPairRing * kPairImgB + kPairERing * 1024) / 8];
  double* const rows = lds;
  int32_t* const erec = reinterpret_cast<int32_t*>(lds + kPairRing * kPairImgB / 8);
  const unsigned rows_b = lds_addr(rows), erec_b = lds_addr(erec);
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Consecutive work items
  // belong to the same or neighbouring camera pairs and re-read the same A rows, so give every
  // XCD a contiguous eighth of the item list (bijective remap of the workgroup index).
  const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int lane = threadIdx.x;
  const int g = lane >> 2, h = lane & 3;
  const int beg = __builtin_amdgcn_readfirstlane(pit_beg[item]);
  const int end = __builtin_amdgcn_readfirstlane(pit_end[item]);
  const int blk = __builtin_amdgcn_readfirstlane(pit_blk[item]);
  const int bj = __builtin_amdgcn_readfirstlane(blk_j[blk]), bk = __builtin_amdgcn_readfirstlane(blk_k[blk]);
  const bool with_e = fold_u && bj == bk;  // every entry of a diagonal block pairs an observation with itself
  const int nsteps = (end - beg + kPairStep - 1) / kPairStep;
  const int last_base = beg + kPairStep * (nsteps - 1);
  // row pieces: a 16-entry row block is 112 pieces of 16 bytes; this lane carries piece `lane`
  // (row r0) and piece 64 + lane (row r1; lanes 48..63 have none and refetch their first piece)
  const int r0 = lane / 7, c0 = lane - 7 * r0;
  const bool has1 = lane < 48;
  const int r1 = has1 ? (64 + lane) / 7 : r0, c1 = has1 ? (64 + lane) - 7 * ((64 + lane) / 7) : c0;
  // entry records of step q -> slot q % kPairRing (8 pieces; the other lanes refetch them)
  auto edma = [&](int q) {
    const int i0 = min(beg + kPairStep * q, last_base);
    dma16(ent, 8u * (unsigned)(i0 + 2 * (lane & 7)), erec_b + 1024 * (q % kPairERing));
  };
  // rows and T of step q -> image q % kPairRing (5 transfers); needs the step's records in LDS
  const bool diag = bj == bk;  // a diagonal block's records hold (row of A, entry of g), see PairEntry
  auto rdma = [&](int q) {
    const int32_t* er = erec + 256 * (q % kPairERing);
    const unsigned base = rows_b + kPairImgB * (q % kPairRing);
    const int2 ea = *reinterpret_cast<const int2*>(er + 2 * r0);
    const int2 eb = *reinterpret_cast<const int2*>(er + 2 * r1);
    const int i0 = min(beg + kPairStep * q, last_base);
    // lanes 0..31: the 16 T rows; lanes 32..47 of a diagonal block: g of the step's 16 observations
    unsigned toff = t_off + 32u * (unsigned)(i0 + ((lane >> 1) & 15)) + 16u * (lane & 1);
    // (records past the item's end belong to the next block, whose second field is a row of A, not
    // an entry of g: those lanes fetch g[0], which is never used)
    if (with_e && lane >= 32 && lane < 48) toff = i0 + (lane - 32) < end ? (unsigned)er[2 * (lane - 32) + 1] : 0u;
    // lanes 48..63 of a diagonal block: the squared residuals of the same 16 observations
    if (with_e && lane >= 48) toff = q_off + (i0 + (lane - 48) < end ? (unsigned)er[2 * (lane - 48) + 1] : 0u);
    dma16x5(A, (unsigned)ea.x + 16u * c0, (unsigned)eb.x + 16u * c1, (unsigned)(diag ? ea.x : ea.y) + 16u * c0,
            (unsigned)(diag ? eb.x : eb.y) + 16u * c1, gT, toff, base);
  };
  double acc0[7] = {0, 0, 0, 0, 0, 0, 0}, acc1[7] = {0, 0, 0, 0, 0, 0, 0};
  double e0 = 0.0, e1 = 0.0, sq = 0.0;
  // `tail`: the step may reach past the item's last entry (only the last step can)
  auto consume = [&](int q, bool tail) {
    const int i0 = beg + kPairStep * q;
    const double* ba = rows + (kPairImgB / 8) * (q % kPairRing);
    const double* bb = ba + 256;
    const double* bt = ba + 512 + 4 * g;
    double2 t0 = ld2(bt), t1 = ld2(bt + 2);
    const bool dead = tail && i0 + g >= end;
    if (dead) t0 = t1 = make_double2(0.0, 0.0);
    const double2 b0 = ld2(bb + 14 * g + 2 * h);  // A_b[0][2h], A_b[0][2h+1]
    const double b10 = bb[14 * g + 7 + 2 * h], b11 = bb[14 * g + 8 + 2 * h];
    const double p00 = t0.x * b0.x + t0.y * b10, p10 = t1.x * b0.x + t1.y * b10;
    const double p01 = t0.x * b0.y + t0.y * b11, p11 = t1.x * b0.y + t1.y * b11;
    if (with_e) {
      double2 gv = ld2(ba + 512 + 64 + 2 * g);
      if (dead) gv = make_double2(0.0, 0.0);
      e0 = fma(b10, gv.y, fma(b0.x, gv.x, e0));
      e1 = fma(b11, gv.y, fma(b0.y, gv.x, e1));
      // (every lane of the entry's group reads it; lane h = 0 of the group is the one that counts)
      sq += dead ? 0.0 : ba[512 + 96 + 2 * g];
    }
    double a[14];
#pragma unroll
    for (int q2 = 0; q2 < 7; ++q2) {
      const double2 v2 = ld2(ba + 14 * g + 2 * q2);
      a[2 * q2] = v2.x;
      a[2 * q2 + 1] = v2.y;
    }
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      acc0[r] = fma(a[7 + r], p10, fma(a[r], p00, acc0[r]));
      acc1[r] = fma(a[7 + r], p11, fma(a[r], p01, acc1[r]));
    }
  };
  // 6 transfers per iteration (5 row/T + 1 record); records run two steps ahead of their rows
  edma(0);
  edma(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  rdma(0);
  edma(2);
  rdma(1);
  edma(3);
  for (int s = 0; s < nsteps - 1; ++s) {
    // all but the 6 youngest transfers have landed: the rows of step s, the records of step s+2
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    consume(s, false);
    rdma(s + 2);  // into the image just consumed
    edma(s + 4);
  }
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  consume(nsteps - 1, true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be landing when the LDS is released
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    double v = acc0[r], w = acc1[r];
#pragma unroll
    for (int sft = 4; sft < 64; sft <<= 1) {
      v += __shfl_xor(v, sft);
      w += __shfl_xor(w, sft);
    }
    acc0[r] = v;
    acc1[r] = w;
  }
  if (with_e) {
#pragma unroll
    for (int sft = 4; sft < 64; sft <<= 1) {
      e0 += __shfl_xor(e0, sft);
      e1 += __shfl_xor(e1, sft);
      sq += __shfl_xor(sq, sft);
    }
  }
  if (g != 0) return;
  const int part = pit_part[item];
  const int ncol = h < 3 ? 2 : 1;
  if (part >= 0) {
    double* o = spart + (int64_t)part * kPairPartStride;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      o[7 * r + 2 * h] = acc0[r];
      if (ncol == 2) o[7 * r + 2 * h + 1] = acc1[r];
    }
    if (with_e) {
      o[49 + 2 * h] = e0;
      if (ncol == 2) o[49 + 2 * h + 1] = e1;
      if (h == 0) o[56] = sq;
    }
    return;
  }
  const int j = bj, k = bk;
  const int64_t ld = 7 * (int64_t)m;
  if (with_e) {
    for (int cs = 0; cs < ncol; ++cs) {
      const double ev = cs == 0 ? e0 : e1;
      S[ld * ld + 7 * j + 2 * h + cs] = ev;
      if (e_out) e_out[7 * j + 2 * h + cs] = ev;
    }
    if (h == 0 && ssq_out) ssq_out[j] = sq;
  }
  for (int cs = 0; cs < ncol; ++cs) {
    const int c = 2 * h + cs;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const double av = cs == 0 ? acc0[r] : acc1[r];
      if (j == k) {
        // U_j from the per-camera item partials (upper triangle, fixed order)
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        const int idx = lo * 7 - lo * (lo - 1) / 2 + (hi - lo);
        double u = 0.0;
        if (!fold_u)
          for (int it = cam_item_ptr[j]; it < cam_item_ptr[j + 1]; ++it) u += cam_part[(int64_t)it * SFM_BA_CAM_RED + idx];
        if (r == c) u += mu_u;
        S[(7 * (int64_t)j + r) * ld + 7 * k + c] = u - av;
      } else {
        // (the solvers read the lower triangle only: a caller that wants nothing but the solve gets
        // no mirror image)
        if (!fold_u) S[(7 * (int64_t)j + r) * ld + 7 * k + c] = -av;
        S[(7 * (int64_t)k + c) * ld + 7 * j + r] = -av;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// camera-pair blocks from the state: what k_pair_blocks computes, with the two rows of A of an entry
// formed in the kernel (ba_rows.h) and its coupling T_ab = B_a V^-1 B_b^T formed from them, instead of
// two rows of A and a 2 x 2 block fetched per entry.  An entry then costs its point's record (position
// and V^-1, 80 bytes, written by k_point) and 4 bytes of index; on the diagonal also the observation's g
// and squared residual.  The block's two camera records are the same for every lane and come through
// the scalar cache.  One wave per work item, 64 entries per step, two phases per step:
//   1. lane = entry: rows A_a, A_b, T, P = T A_b; A_a (2 x 7) and P (2 x 7) go to the wave's LDS
//   2. lane = element (r, c) of the block (49 lanes), e_j[r] (7 lanes), the squared residuals (1 lane):
//      the lane walks the step's entries in order and adds A_a[:, r] . P[:, c] (A_a[:, r] . g for e_j)
// so every sum runs in entry order inside one lane: no reduction across lanes, reproducible bit for
// bit.  No weights; the self pairs carry U_j (T_aa - I), as with fold_u in k_pair_blocks.
// ------------------------------------------------------------------------------------------
constexpr int kPairStateRec = 33;  // doubles per entry in LDS: A_a (14), P (14), g (2), squared residual, 0.5, pad (odd stride)

__global__ __launch_bounds__(64, 3) void k_pair_state(
    int n_items, const int32_t* __restrict__ pit_blk, const int32_t* __restrict__ pit_beg,
    const int32_t* __restrict__ pit_end, const int32_t* __restrict__ pit_part, const int32_t* __restrict__ blk_j,
    const int32_t* __restrict__ blk_k, const PairEntry* __restrict__ ent, const int32_t* __restrict__ ent_pt,
    const double* __restrict__ cam_blk, const double* __restrict__ prec, const double* __restrict__ g,
    const double* __restrict__ qsq, double mu_u, int m, double* __restrict__ S, double* __restrict__ e_out,
    double* __restrict__ ssq_out, double* __restrict__ spart) {
  __shared__ double img[64 * kPairStateRec];
  // (same remap of the workgroup index as k_pair_blocks: every XCD walks a contiguous eighth of the items)
  const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int lane = threadIdx.x;
  const int beg = __builtin_amdgcn_readfirstlane(pit_beg[item]);
  const int end = __builtin_amdgcn_readfirstlane(pit_end[item]);
  const int blk = __builtin_amdgcn_readfirstlane(pit_blk[item]);
  const int bj = __builtin_amdgcn_readfirstlane(blk_j[blk]), bk = __builtin_amdgcn_readfirstlane(blk_k[blk]);
  const bool diag = bj == bk;
  const double2* rj = reinterpret_cast<const double2*>(cam_blk) + (SFM_BA_CAM_BLK / 2) * (int64_t)bj;
  const double2* rk = reinterpret_cast<const double2*>(cam_blk) + (SFM_BA_CAM_BLK / 2) * (int64_t)bk;
  // phase 2: what this lane multiplies per entry (offsets into the entry's record)
  //   lanes 0..48: A_a[0][r] P[0][c] + A_a[1][r] P[1][c];  49..55: A_a[0][r] g0 + A_a[1][r] g1;
  //   56: 0.5 q + 0.5 q (the squared residual, exactly); 57..63: nothing
  int oa0, oa1, op0, op1;
  {
    const int r = lane < 49 ? lane / 7 : lane - 49, c = lane < 49 ? lane - 7 * (lane / 7) : 0;
    oa0 = r;
    oa1 = 7 + r;
    op0 = lane < 49 ? 14 + c : 28;
    op1 = lane < 49 ? 21 + c : 29;
    if (lane >= 56) {
      oa0 = oa1 = 30;
      op0 = op1 = 31;
    }
  }
  const bool counts = lane < 49 || (diag && lane < 57);
  double acc0 = 0.0, acc1 = 0.0;
  for (int i0 = beg; i0 < end; i0 += 64) {
    // ---- phase 1: this lane's entry ----
    {
      const int i = i0 + lane;
      const bool live = i < end;
      const int ic = live ? i : end - 1;
      const double* pr = prec + kPointRec * (int64_t)ent_pt[ic];
      const double2 x01 = ld2(pr), x2v = ld2(pr + 2), v12 = ld2(pr + 4), v34 = ld2(pr + 6);
      const double v5 = pr[8];
      double2 gv = make_double2(0.0, 0.0);
      double qs = 0.0;
      if (diag) {
        // the second field of a diagonal block's record is 16 * observation: its entry of g and of the squared residuals
        const unsigned ob = ent[ic].row_b >> 3;  // in doubles
        const double2 gl = ld2(g + ob);
        const double ql = qsq[ob];
        gv = live ? gl : gv;
        qs = live ? ql : 0.0;
      }
      double* o = img + kPairStateRec * lane;
      const double i00 = x2v.y, i01 = v12.x, i02 = v12.y, i11 = v34.x, i12 = v34.y, i22 = v5;
      double2 aa[9], ab[9];
      fz_rows_from_state<false>(rj, static_cast<const double2*>(nullptr), x01.x, x01.y, x2v.x, make_double2(0.0, 0.0), live, aa);
#define ra_(q) (((q) & 1) ? aa[(q) >> 1].y : aa[(q) >> 1].x)
#define rb_(q) (((q) & 1) ? ab[(q) >> 1].y : ab[(q) >> 1].x)
      if (diag) {
#pragma unroll
        for (int q = 0; q < 7; ++q) ab[q] = aa[q];
      } else {
        fz_rows_from_state<false>(rk, static_cast<const double2*>(nullptr), x01.x, x01.y, x2v.x, make_double2(0.0, 0.0), live, ab);
      }
      // T = B_a V^-1 B_b^T with B = minus the camera-centre columns of A (the two signs cancel)
      const double h00 = ra_(4) * i00 + ra_(5) * i01 + ra_(6) * i02;
      const double h01 = ra_(4) * i01 + ra_(5) * i11 + ra_(6) * i12;
      const double h02 = ra_(4) * i02 + ra_(5) * i12 + ra_(6) * i22;
      const double h10 = ra_(11) * i00 + ra_(12) * i01 + ra_(13) * i02;
      const double h11 = ra_(11) * i01 + ra_(12) * i11 + ra_(13) * i12;
      const double h12 = ra_(11) * i02 + ra_(12) * i12 + ra_(13) * i22;
      double t00 = h00 * rb_(4) + h01 * rb_(5) + h02 * rb_(6), t01 = h00 * rb_(11) + h01 * rb_(12) + h02 * rb_(13);
      double t10 = h10 * rb_(4) + h11 * rb_(5) + h12 * rb_(6), t11 = h10 * rb_(11) + h11 * rb_(12) + h12 * rb_(13);
      if (diag && live) {
        t00 -= 1.0;
        t11 -= 1.0;
      }
#pragma unroll
      for (int q = 0; q < 14; ++q) o[q] = ra_(q);
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        o[14 + c] = t00 * rb_(c) + t01 * rb_(7 + c);
        o[21 + c] = t10 * rb_(c) + t11 * rb_(7 + c);
      }
#undef ra_
#undef rb_
      o[28] = gv.x;
      o[29] = gv.y;
      o[30] = qs;
      o[31] = 0.5;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- phase 2: the step's entries in order (two interleaved sums: even and odd entries); eight
    // entries' operands are fetched before the first of them is used ----
    {
      const int cnt = end - i0 < 64 ? end - i0 : 64;
      const double* q = img;
      int e = 0;
      for (; e + 8 <= cnt; e += 8, q += 8 * kPairStateRec) {
        double a0[8], a1[8], p0[8], p1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a0[u] = q[u * kPairStateRec + oa0];
          a1[u] = q[u * kPairStateRec + oa1];
          p0[u] = q[u * kPairStateRec + op0];
          p1[u] = q[u * kPairStateRec + op1];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          acc0 = fma(a1[u], p1[u], fma(a0[u], p0[u], acc0));
          acc1 = fma(a1[u + 1], p1[u + 1], fma(a0[u + 1], p0[u + 1], acc1));
        }
      }
      for (; e + 2 <= cnt; e += 2, q += 2 * kPairStateRec) {
        acc0 = fma(q[oa1], q[op1], fma(q[oa0], q[op0], acc0));
        acc1 = fma(q[kPairStateRec + oa1], q[kPairStateRec + op1], fma(q[kPairStateRec + oa0], q[kPairStateRec + op0], acc1));
      }
      if (e < cnt) acc0 = fma(q[oa1], q[op1], fma(q[oa0], q[op0], acc0));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (!counts) return;
  const double v = acc0 + acc1;
  const int part = pit_part[item];
  const int64_t ld = 7 * (int64_t)m;
  if (part >= 0) {
    spart[(int64_t)part * kPairPartStride + lane] = v;
    return;
  }
  if (lane == 56) {
    if (ssq_out) ssq_out[bj] = v;
  } else if (lane >= 49) {
    S[ld * ld + 7 * bj + (lane - 49)] = v;
    if (e_out) e_out[7 * bj + (lane - 49)] = v;
  } else {
    const int r = lane / 7, c = lane - 7 * r;
    if (diag)
      S[(7 * (int64_t)bj + r) * ld + 7 * bk + c] = (r == c ? mu_u : 0.0) - v;
    else
      S[(7 * (int64_t)bk + c) * ld + 7 * bj + r] = -v;
  }
}

// blocks whose entries were split over several work items: add the partials in item order
__global__ __launch_bounds__(64) void k_pair_finish(
    const int32_t* __restrict__ mb_blk, const int32_t* __restrict__ mb_part,
    const int32_t* __restrict__ mb_cnt, const int32_t* __restrict__ blk_j,
    const int32_t* __restrict__ blk_k, const double* __restrict__ spart,
    const double* __restrict__ cam_part, const int32_t* __restrict__ cam_item_ptr, double mu_u, int m,
    int fold_u, double* __restrict__ S, double* __restrict__ e_out, double* __restrict__ ssq_out) {
  const int t = threadIdx.x;
  if (t > 56) return;
  const int b = mb_blk[blockIdx.x];
  const int p0 = mb_part[blockIdx.x], cnt = mb_cnt[blockIdx.x];
  const int j = blk_j[b], k = blk_k[b];
  if (t >= 49 && !(fold_u && j == k)) return;
  double a = 0.0;
  for (int i = 0; i < cnt; ++i) a += spart[(int64_t)(p0 + i) * kPairPartStride + t];
  const int64_t ld = 7 * (int64_t)m;
  if (t == 56) {
    if (ssq_out) ssq_out[j] = a;  // the camera's squared residuals (slot 56)
    return;
  }
  if (t >= 49) {
    // e_j of a diagonal block (slots 49..55 of its partials)
    S[ld * ld + 7 * j + (t - 49)] = a;
    if (e_out) e_out[7 * j + (t - 49)] = a;
    return;
  }
  const int r = t / 7, c = t % 7;
  if (j == k) {
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int idx = lo * 7 - lo * (lo - 1) / 2 + (hi - lo);
    double u = 0.0;
    if (!fold_u)
      for (int it = cam_item_ptr[j]; it < cam_item_ptr[j + 1]; ++it) u += cam_part[(int64_t)it * SFM_BA_CAM_RED + idx];
    if (r == c) u += mu_u;
    S[(7 * (int64_t)j + r) * ld + 7 * k + c] = u - a;
  } else {
    if (!fold_u) S[(7 * (int64_t)j + r) * ld + 7 * k + c] = -a;
    S[(7 * (int64_t)k + c) * ld + 7 * j + r] = -a;
  }
}

// ------------------------------------------------------------------------------------------
// W = A^T w B and Y = W V^-1[point] per observation (only for callers that want them)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_calc_W_Y(int64_t N, const double* __restrict__ A,
                                                  const double* __restrict__ B,
                                                  const double* __restrict__ wgt,
                                                  const double* __restrict__ Vinv,
                                                  const int32_t* __restrict__ pt,
                                                  double* __restrict__ W, double* __restrict__ Y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double a[14], b[6];
#pragma unroll
  for (int q = 0; q < 14; ++q) a[q] = A[14 * i + q];
#pragma unroll
  for (int q = 0; q < 6; ++q) b[q] = B[6 * i + q];
  double w00 = 1.0, w01 = 0.0, w10 = 0.0, w11 = 1.0;
  if (wgt) {
    w00 = wgt[4 * i]; w01 = wgt[4 * i + 1]; w10 = wgt[4 * i + 2]; w11 = wgt[4 * i + 3];
  }
  double vi[9];
  const double* vp = Vinv + 9 * (int64_t)pt[i];
#pragma unroll
  for (int q = 0; q < 9; ++q) vi[q] = vp[q];
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const double wa0 = a[r] * w00 + a[7 + r] * w10;
    const double wa1 = a[r] * w01 + a[7 + r] * w11;
    double wr[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) wr[q] = wa0 * b[q] + wa1 * b[3 + q];
    if (W) {
#pragma unroll
      for (int q = 0; q < 3; ++q) W[21 * i + 3 * r + q] = wr[q];
    }
    if (Y) {
#pragma unroll
      for (int q = 0; q < 3; ++q) Y[21 * i + 3 * r + q] = wr[0] * vi[q] + wr[1] * vi[3 + q] + wr[2] * vi[6 + q];
    }
  }
}

__global__ void k_add_diagonal(double* S, int n7, double mu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n7) S[(int64_t)i * n7 + i] += mu;
}

// Lower triangle of S (row i: columns 0 .. i) followed by the right-hand side (row n7 of the
// system buffer), and back.  What the ranks of a sharded run with a dense S exchange: the solvers
// read the lower triangle only.  One workgroup per row; `mu` is added to the diagonal on the way back.
__global__ __launch_bounds__(256) void k_lower_pack(const double* __restrict__ S, int n7, double* __restrict__ tri) {
  const int i = blockIdx.x;
  const int64_t at = (int64_t)i * (i + 1) / 2;
  const int len = i < n7 ? i + 1 : n7;
  for (int c = threadIdx.x; c < len; c += 256) tri[at + c] = S[(int64_t)i * n7 + c];
}

__global__ __launch_bounds__(256) void k_lower_unpack(const double* __restrict__ tri, int n7, double mu,
                                                      double* __restrict__ S) {
  const int i = blockIdx.x;
  const int64_t at = (int64_t)i * (i + 1) / 2;
  const int len = i < n7 ? i + 1 : n7;
  for (int c = threadIdx.x; c < len; c += 256) S[(int64_t)i * n7 + c] = tri[at + c] + ((c == i) ? mu : 0.0);
}

// ------------------------------------------------------------------------------------------
// back-substitution: delta_b_i = V_i^-1 (eps_b_i - sum_a B_a^T w_a A_a delta_a[cam_a])
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_back_substitute(
    const int32_t* __restrict__ pt_map, const int32_t* __restrict__ grp_pt, const int32_t* __restrict__ pt_ptr,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ slot_cam,
    const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ wgt,
    const double* __restrict__ Vinv, const double* __restrict__ eps_b, const double* __restrict__ da,
    double* __restrict__ db, double* __restrict__ Xupd, double x_step) {
  __shared__ double sV[kGroupSlots * 3];
  __shared__ int sPtr[kGroupSlots + 1];
  const int t = threadIdx.x;
  const int p0 = grp_pt[blockIdx.x];
  const int np = grp_pt[blockIdx.x + 1] - p0;
  for (int i = t; i <= np; i += 256) sPtr[i] = pt_ptr[p0 + i];
  __syncthreads();
  const int s0 = sPtr[0];
  const int ns = sPtr[np] - s0;
  if (ns > kGroupSlots) {
    // one point with more observations than lanes: the lanes stride over them, sums through LDS
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int s = t; s < ns; s += 256) {
      const int64_t pos = perm[s0 + s];
      const double* dj = da + 7 * (int64_t)slot_cam[s0 + s];
      const double* ap = A + 14 * pos;
      double h0 = 0.0, h1 = 0.0;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        h0 += ap[q] * dj[q];
        h1 += ap[7 + q] * dj[q];
      }
      double w00 = 1.0, w01 = 0.0, w10 = 0.0, w11 = 1.0;
      if (wgt) {
        const double* wp = wgt + 4 * pos;
        w00 = wp[0]; w01 = wp[1]; w10 = wp[2]; w11 = wp[3];
      }
      const double k0 = w00 * h0 + w01 * h1, k1 = w10 * h0 + w11 * h1;
      if (B) {
        const double* b = B + 6 * pos;
        a0 += b[0] * k0 + b[3] * k1;
        a1 += b[1] * k0 + b[4] * k1;
        a2 += b[2] * k0 + b[5] * k1;
      } else {
        a0 += -ap[4] * k0 - ap[11] * k1;
        a1 += -ap[5] * k0 - ap[12] * k1;
        a2 += -ap[6] * k0 - ap[13] * k1;
      }
    }
    sV[3 * t] = a0; sV[3 * t + 1] = a1; sV[3 * t + 2] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) {
        sV[3 * t] += sV[3 * (t + o)];
        sV[3 * t + 1] += sV[3 * (t + o) + 1];
        sV[3 * t + 2] += sV[3 * (t + o) + 2];
      }
      __syncthreads();
    }
    if (t == 0) {
      const int64_t p = pt_map ? pt_map[p0] : p0;
      const double r0 = eps_b[3 * p] - sV[0], r1 = eps_b[3 * p + 1] - sV[1], r2 = eps_b[3 * p + 2] - sV[2];
      const double* v = Vinv + 9 * p;
      const double d0 = v[0] * r0 + v[1] * r1 + v[2] * r2;
      const double d1 = v[3] * r0 + v[4] * r1 + v[5] * r2;
      const double d2 = v[6] * r0 + v[7] * r1 + v[8] * r2;
      db[3 * p] = d0; db[3 * p + 1] = d1; db[3 * p + 2] = d2;
      if (Xupd) {
        Xupd[3 * p] += x_step * d0;
        Xupd[3 * p + 1] += x_step * d1;
        Xupd[3 * p + 2] += x_step * d2;
      }
    }
    return;
  }
  if (t < ns) {
    const int64_t pos = perm[s0 + t];
    const double* dj = da + 7 * (int64_t)slot_cam[s0 + t];
    const double* ap = A + 14 * pos;
    double h0 = 0.0, h1 = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      h0 += ap[q] * dj[q];
      h1 += ap[7 + q] * dj[q];
    }
    double w00 = 1.0, w01 = 0.0, w10 = 0.0, w11 = 1.0;
    if (wgt) {
      const double* wp = wgt + 4 * pos;
      w00 = wp[0]; w01 = wp[1]; w10 = wp[2]; w11 = wp[3];
    }
    const double k0 = w00 * h0 + w01 * h1, k1 = w10 * h0 + w11 * h1;
    if (B) {
      const double* b = B + 6 * pos;
#pragma unroll
      for (int q = 0; q < 3; ++q) sV[3 * t + q] = b[q] * k0 + b[3 + q] * k1;
    } else {
      // no B given: minus the camera-centre block of A
#pragma unroll
      for (int q = 0; q < 3; ++q) sV[3 * t + q] = -ap[4 + q] * k0 - ap[11 + q] * k1;
    }
  }
  __syncthreads();
  if (t < np) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int s = sPtr[t] - s0; s < sPtr[t + 1] - s0; ++s) {
      a0 += sV[3 * s];
      a1 += sV[3 * s + 1];
      a2 += sV[3 * s + 2];
    }
    const int64_t p = pt_map ? pt_map[p0 + t] : p0 + t;
    const double r0 = eps_b[3 * p] - a0, r1 = eps_b[3 * p + 1] - a1, r2 = eps_b[3 * p + 2] - a2;
    const double* v = Vinv + 9 * p;
    const double d0 = v[0] * r0 + v[1] * r1 + v[2] * r2;
    const double d1 = v[3] * r0 + v[4] * r1 + v[5] * r2;
    const double d2 = v[6] * r0 + v[7] * r1 + v[8] * r2;
    db[3 * p] = d0; db[3 * p + 1] = d1; db[3 * p + 2] = d2;
    if (Xupd) {
      Xupd[3 * p] += x_step * d0;
      Xupd[3 * p + 1] += x_step * d1;
      Xupd[3 * p + 2] += x_step * d2;
    }
  }
}

// ------------------------------------------------------------------------------------------
// camera update
// ------------------------------------------------------------------------------------------
__global__ void k_update_cameras(double* __restrict__ R, double* __restrict__ T, double* __restrict__ C,
                                 double* __restrict__ P, const double* __restrict__ K, int k_stride,
                                 const double* __restrict__ da, int m, int first, int held, double q_step,
                                 double c_step, int32_t* __restrict__ flags, double* __restrict__ blk,
                                 int32_t* __restrict__ blk_flags) {
  // blk (may be null): the prepared record of every camera in its new state, what k_camera_prepare
  // would compute next - the following iteration then needs no launch for it
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double r[9], q[4];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = R[9 * (int64_t)j + i];
  if (j < first || j == held) {  // cameras the loop holds: state kept, record still prepared
    if (blk) {
      double cc[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) cc[i] = C[3 * (int64_t)j + i];
      const int pf = camera_record(r, cc, K + (int64_t)k_stride * j, blk + (int64_t)SFM_BA_CAM_BLK * j);
      if (blk_flags && pf) blk_flags[j] |= pf;
    }
    return;
  }
  int f = 0;
  if (!(fabs(det3(r) - 1.0) <= 1e-8 + 1e-5)) f |= SFM_BA_FLAG_DET_NOT_ONE;
  rot_to_quat(r, q);
  const double* d = da + 7 * (int64_t)j;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] += q_step * d[i];
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(nrm > 1e-8)) f |= SFM_BA_FLAG_ZERO_QUAT;
  if (!isfinite(nrm)) f |= SFM_BA_FLAG_NONFINITE;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = q[i] / nrm;
  quat_to_rot(q, r);
  double c[3], t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) c[i] = C[3 * (int64_t)j + i] + c_step * d[4 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = -(r[3 * i] * c[0] + r[3 * i + 1] * c[1] + r[3 * i + 2] * c[2]);
  // C = -R^T T as Frame.compute_projection_matrix recomputes it
#pragma unroll
  for (int i = 0; i < 3; ++i) c[i] = -(r[i] * t[0] + r[3 + i] * t[1] + r[6 + i] * t[2]);
  const double* k = K + (int64_t)k_stride * j;
#pragma unroll
  for (int i = 0; i < 9; ++i) R[9 * (int64_t)j + i] = r[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    T[3 * (int64_t)j + i] = t[i];
    C[3 * (int64_t)j + i] = c[i];
  }
  if (P) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double r0 = b < 3 ? r[b] : t[0], r1 = b < 3 ? r[3 + b] : t[1], r2 = b < 3 ? r[6 + b] : t[2];
        P[12 * (int64_t)j + 4 * a + b] = k[3 * a] * r0 + k[3 * a + 1] * r1 + k[3 * a + 2] * r2;
      }
    }
  }
  if (flags && f) flags[j] |= f;  // sticky: the caller clears
  if (blk) {
    const int pf = camera_record(r, c, k, blk + (int64_t)SFM_BA_CAM_BLK * j);
    if (blk_flags && pf) blk_flags[j] |= pf;
  }
}

// ------------------------------------------------------------------------------------------
// cost: per-camera sum of squared residuals through the same per-camera work items
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCamThreads) void k_cost_items(
    const int32_t* __restrict__ item_beg, const int32_t* __restrict__ item_end,
    const int32_t* __restrict__ cam_obs, const double* __restrict__ P, const double* __restrict__ X,
    const int32_t* __restrict__ cam, const int32_t* __restrict__ pt, const double* __restrict__ xt,
    double* __restrict__ part) {
  __shared__ double red[kCamThreads];
  const int t = threadIdx.x;
  const int beg = item_beg[blockIdx.x], end = item_end[blockIdx.x];
  // every observation of an item belongs to one camera: its projection matrix through the scalar path
  const int64_t pos0 = cam_obs ? cam_obs[beg] : beg;
  const double* pm = P + 12 * (int64_t)__builtin_amdgcn_readfirstlane(cam[pos0]);
  double pr[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) pr[i] = pm[i];
  // the (at most kCamChunk / kCamThreads) observations of a thread: all index loads, then all gathers,
  // then the arithmetic, summed in the order of the observations
  constexpr int U = kCamChunk / kCamThreads;
  int64_t pos[U];
  bool on[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = beg + t + u * kCamThreads;
    on[u] = i < end;
    pos[u] = on[u] ? (cam_obs ? cam_obs[i] : i) : pos0;
  }
  int pts[U];
#pragma unroll
  for (int u = 0; u < U; ++u) pts[u] = pt[pos[u]];
  double xs[U][3];
  double2 xa[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double* x = X + 3 * (int64_t)pts[u];
    xs[u][0] = x[0];
    xs[u][1] = x[1];
    xs[u][2] = x[2];
    xa[u] = ld2(xt + 2 * pos[u]);
  }
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double x0 = xs[u][0], x1 = xs[u][1], x2 = xs[u][2];
    const double h0 = pr[0] * x0 + pr[1] * x1 + pr[2] * x2 + pr[3];
    const double h1 = pr[4] * x0 + pr[5] * x1 + pr[6] * x2 + pr[7];
    const double h2 = pr[8] * x0 + pr[9] * x1 + pr[10] * x2 + pr[11];
    const double e0 = xa[u].x - h0 / h2, e1 = xa[u].y - h1 / h2;
    if (on[u]) acc += e0 * e0 + e1 * e1;
  }
  red[t] = acc;
  __syncthreads();
  for (int s = kCamThreads / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (t == 0) part[blockIdx.x] = red[0];
}

__global__ void k_cost_camera_sum(const int32_t* __restrict__ cam_item_ptr, const double* __restrict__ part,
                                  int m, double* __restrict__ ssq) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double a = 0.0;
  for (int it = cam_item_ptr[j]; it < cam_item_ptr[j + 1]; ++it) a += part[it];
  ssq[j] = a;
}

__global__ __launch_bounds__(256) void k_cost_finish(const double* __restrict__ ssq, int m,
                                                     double* __restrict__ cost) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  double a = 0.0;
  for (int j = t; j < m; j += 256) a += sqrt(ssq[j]);
  red[t] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (t == 0) cost[0] = red[0] / (double)m;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, what);
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

int launch_camera_prepare(const double* R, const double* C, const double* K, int k_stride, int m,
                          double* blk, int32_t* flags, hipStream_t st) {
  if (m <= 0) return SFM_BA_OK;
  {
    ProfScope ps(KID_CAMERA_PREPARE, st);
    hipLaunchKernelGGL(k_camera_prepare, dim3((m + 63) / 64), dim3(64), 0, st, R, C, K, k_stride, m, blk, flags);
  }
  return check_launch("k_camera_prepare");
}

int launch_jacobian(const double* blk, const double* P, const double* X, const int32_t* cam,
                    const int32_t* pt, int64_t N, double* A, double* B, double* xp, hipStream_t st) {
  if (N <= 0) return SFM_BA_OK;
  const int64_t grid = (N + 255) / 256;
  {
    ProfScope ps(KID_JACOBIAN, st);
    hipLaunchKernelGGL(k_jacobian, dim3((unsigned)grid), dim3(256), 0, st, blk, P, X, cam, pt, N, A, B, xp);
  }
  return check_launch("k_jacobian");
}

int launch_assemble(const sfm_ba_plan* p, const double* A, const double* B, const double* xt,
                    const double* xp, const double* w, double mu, double mu_u, double* S, double* e,
                    double* U, double* eps_a, double* Vinv, double* eps_b, double* cam_red,
                    void* ws, hipStream_t st) {
  char* base = static_cast<char*>(ws);
  double* g = reinterpret_cast<double*>(base + p->ws_g);
  double* T = reinterpret_cast<double*>(base + p->ws_T);
  double* qsq = reinterpret_cast<double*>(base + p->ws_q);
  double* campart = reinterpret_cast<double*>(base + p->ws_campart);
  double* spart = reinterpret_cast<double*>(base + p->ws_spart);
  // Callers that want none of the per-camera by-products (U, eps_a, the all-reduce records) get
  // them folded into the camera-pair kernel: no separate pass over A.
  const int fold_u = (S && !U && !eps_a && !cam_red) ? 1 : 0;
  // ... and, where the plan found runs of points inside narrow camera windows, the whole assembly
  // is one pass over A (ba_fused.hip).  That path takes the point block from A and needs mu > 0.
  if (fold_u && !B && !w && mu > 0.0 && p->n_runs > 0 && p->fz_enabled)
    return launch_assemble_fused(p, A, xt, xp, mu, mu_u, S, e, Vinv, eps_b, ws, st);
  if (p->n_groups > 0) {
    {
    ProfScope ps(KID_POINT, st);
    hipLaunchKernelGGL(k_point, dim3(p->n_groups), dim3(256), 0, st, static_cast<const int32_t*>(nullptr), p->d_grp_pt,
                       p->d_pt_ptr, p->d_perm, p->d_tp_ptr, p->d_pair_entry, A, B, xt, xp, w, mu, fold_u, Vinv, eps_b, g, qsq, T,
                       static_cast<const double*>(nullptr), static_cast<double*>(nullptr));
  }
    int rc = check_launch("k_point");
    if (rc) return rc;
  }
  if (p->n_items > 0 && !fold_u) {
    {
    ProfScope ps(KID_CAMERA_ITEMS, st);
    hipLaunchKernelGGL(k_camera_items, dim3(p->n_items), dim3(kCamThreads), 0, st, p->d_item_beg,
                       p->d_item_end, p->d_cam_obs, A, xt, xp, w, g, campart);
  }
    int rc = check_launch("k_camera_items");
    if (rc) return rc;
  }
  int rc = SFM_BA_OK;
  if (!fold_u) {
    {
      ProfScope ps(KID_CAMERA_FINISH, st);
      hipLaunchKernelGGL(k_camera_finish, dim3(p->m), dim3(64), 0, st, p->d_cam_item_ptr, campart, p->m, mu_u, S,
                         e, U, eps_a, cam_red);
    }
    rc = check_launch("k_camera_finish");
    if (rc) return rc;
  }
  if (p->n_pair_items > 0 && S) {
    {
    ProfScope ps(KID_PAIR_BLOCKS, st);
    hipLaunchKernelGGL(k_pair_blocks, dim3(p->n_pair_items), dim3(64), 0, st, p->n_pair_items,
                       p->d_pit_blk, p->d_pit_beg, p->d_pit_end, p->d_pit_part, p->d_blk_j, p->d_blk_k,
                       p->d_entries, A, g, (unsigned)(p->ws_T - p->ws_g), (unsigned)(p->ws_q - p->ws_g), fold_u, campart,
                       p->d_cam_item_ptr, mu_u, p->m, S, e, static_cast<double*>(nullptr), spart);
  }
    rc = check_launch("k_pair_blocks");
    if (rc) return rc;
    if (p->n_multi > 0) {
      {
    ProfScope ps(KID_PAIR_FINISH, st);
    hipLaunchKernelGGL(k_pair_finish, dim3(p->n_multi), dim3(64), 0, st, p->d_mb_blk, p->d_mb_part,
                         p->d_mb_cnt, p->d_blk_j, p->d_blk_k, spart, campart, p->d_cam_item_ptr, mu_u, p->m,
                         fold_u, S, e, static_cast<double*>(nullptr));
  }
      rc = check_launch("k_pair_finish");
      if (rc) return rc;
    }
  }
  return SFM_BA_OK;
}

// The camera-pair path for the far plan of a larger plan (every block of such a plan keeps partial
// results): V^-1 and eps_b of its points under the ids pt_map gives them, and per block the partial sums
// of A_j^T T A_k - which on the diagonal hold sum Y W^T - U_j - and of A^T g, left in the workspace for the
// finishing kernel of the larger plan.  The point block is taken from A; no weights.
int launch_assemble_parts(const sfm_ba_plan* p, const int32_t* pt_map, const double* A, const double* xt,
                          const double* xp, double mu, double* Vinv, double* eps_b, void* ws, hipStream_t st) {
  // (the far points are long tracks: the gathering kernel fetches their rows of A again and again out of the
  // L2, and is the faster one for them - 88 us against 142 for k_pair_state at 500 cameras x 300 k points,
  // tracks visibility)
  if (!p->force_parts) return SFM_BA_EINVAL;
  char* base = static_cast<char*>(ws);
  double* g = reinterpret_cast<double*>(base + p->ws_g);
  double* T = reinterpret_cast<double*>(base + p->ws_T);
  double* qsq = reinterpret_cast<double*>(base + p->ws_q);
  double* spart = reinterpret_cast<double*>(base + p->ws_spart);
  if (p->n_groups > 0) {
    {
      ProfScope ps(KID_POINT, st);
      hipLaunchKernelGGL(k_point, dim3(p->n_groups), dim3(256), 0, st, pt_map, p->d_grp_pt, p->d_pt_ptr, p->d_perm,
                         p->d_tp_ptr, p->d_pair_entry, A, static_cast<const double*>(nullptr), xt, xp,
                         static_cast<const double*>(nullptr), mu, 1, Vinv, eps_b, g, qsq, T,
                         static_cast<const double*>(nullptr), static_cast<double*>(nullptr));
    }
    const int rc = check_launch("k_point");
    if (rc) return rc;
  }
  if (p->n_pair_items > 0) {
    {
      ProfScope ps(KID_PAIR_BLOCKS, st);
      hipLaunchKernelGGL(k_pair_blocks, dim3(p->n_pair_items), dim3(64), 0, st, p->n_pair_items, p->d_pit_blk, p->d_pit_beg,
                         p->d_pit_end, p->d_pit_part, p->d_blk_j, p->d_blk_k, p->d_entries, A, g,
                         (unsigned)(p->ws_T - p->ws_g), (unsigned)(p->ws_q - p->ws_g), 1,
                         static_cast<const double*>(nullptr), p->d_cam_item_ptr, 0.0, p->m, static_cast<double*>(nullptr),
                         static_cast<double*>(nullptr), static_cast<double*>(nullptr), spart);
    }
    return check_launch("k_pair_blocks");
  }
  return SFM_BA_OK;
}

// The camera-pair path of a whole plan inside the loop.  A: the rows sfm_ba_jacobian wrote from this state
// (the per-point stage takes the point blocks from them, the back-substitution reads them afterwards); the
// camera-pair products are formed from the state.  Writes the lower triangle of S, row 7m, e, V^-1, eps_b
// and (ssq) the per-camera sums of squared residuals.
int launch_assemble_pairs_state(const sfm_ba_plan* p, const double* A, const double* blk, const double* X,
                                const double* xt, const double* xp, double mu, double mu_u, double* S, double* e,
                                double* Vinv, double* eps_b, double* ssq, void* ws, hipStream_t st) {
  char* base = static_cast<char*>(ws);
  double* g = reinterpret_cast<double*>(base + p->ws_g);
  double* qsq = reinterpret_cast<double*>(base + p->ws_q);
  double* spart = reinterpret_cast<double*>(base + p->ws_spart);
  double* prec = reinterpret_cast<double*>(base + p->ws_prec);
  if (p->n_groups > 0) {
    {
      ProfScope ps(KID_POINT, st);
      hipLaunchKernelGGL(k_point, dim3(p->n_groups), dim3(256), 0, st, static_cast<const int32_t*>(nullptr), p->d_grp_pt,
                         p->d_pt_ptr, p->d_perm, p->d_tp_ptr, p->d_pair_entry, A, static_cast<const double*>(nullptr), xt, xp,
                         static_cast<const double*>(nullptr), mu, 1, Vinv, eps_b, g, qsq, static_cast<double*>(nullptr), X, prec);
    }
    const int rc = check_launch("k_point");
    if (rc) return rc;
  }
  if (p->n_pair_items > 0) {
    {
      ProfScope ps(KID_PAIR_STATE, st);
      hipLaunchKernelGGL(k_pair_state, dim3(p->n_pair_items), dim3(64), 0, st, p->n_pair_items, p->d_pit_blk, p->d_pit_beg,
                         p->d_pit_end, p->d_pit_part, p->d_blk_j, p->d_blk_k, p->d_entries, p->d_ent_pt, blk, prec, g, qsq, mu_u,
                         p->m, S, e, ssq, spart);
    }
    int rc = check_launch("k_pair_state");
    if (rc) return rc;
    if (p->n_multi > 0) {
      {
        ProfScope ps(KID_PAIR_FINISH, st);
        hipLaunchKernelGGL(k_pair_finish, dim3(p->n_multi), dim3(64), 0, st, p->d_mb_blk, p->d_mb_part, p->d_mb_cnt, p->d_blk_j,
                           p->d_blk_k, spart, static_cast<const double*>(nullptr), p->d_cam_item_ptr, mu_u, p->m, 1, S, e, ssq);
      }
      rc = check_launch("k_pair_finish");
      if (rc) return rc;
    }
  }
  return SFM_BA_OK;
}

int launch_calc_W_Y(const sfm_ba_plan* p, const double* A, const double* B, const double* w,
                    const double* Vinv, const int32_t* pt, double* W, double* Y, hipStream_t st) {
  const int64_t N = p->n_obs;
  if (N <= 0) return SFM_BA_OK;
  {
    ProfScope ps(KID_CALC_W_Y, st);
    hipLaunchKernelGGL(k_calc_W_Y, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, A, B, w, Vinv, pt, W, Y);
  }
  return check_launch("k_calc_W_Y");
}

int launch_add_diagonal(double* S, int n7, double mu, hipStream_t st) {
  if (n7 <= 0) return SFM_BA_OK;
  hipLaunchKernelGGL(k_add_diagonal, dim3((n7 + 255) / 256), dim3(256), 0, st, S, n7, mu);
  return check_launch("k_add_diagonal");
}

// [cost, info, OR of the camera flag words] as three doubles: what a caller needs to read after an
// iteration, in one buffer
__global__ __launch_bounds__(256) void k_status(const double* __restrict__ cost, const int32_t* __restrict__ info,
                                                const int32_t* __restrict__ fa, const int32_t* __restrict__ fb, int m,
                                                double* __restrict__ out) {
  __shared__ int s_or[4];
  int f = 0;
  for (int j = threadIdx.x; j < m; j += 256) f |= (fa ? fa[j] : 0) | (fb ? fb[j] : 0);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o);
  if ((threadIdx.x & 63) == 0) s_or[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = cost ? cost[0] : 0.0;
    out[1] = info ? (double)info[0] : 0.0;
    out[2] = (double)(s_or[0] | s_or[1] | s_or[2] | s_or[3]);
  }
}

int launch_status(const double* cost, const int32_t* info, const int32_t* fa, const int32_t* fb, int m, double* out,
                  hipStream_t st) {
  hipLaunchKernelGGL(k_status, dim3(1), dim3(256), 0, st, cost, info, fa, fb, m, out);
  return check_launch("k_status");
}

int launch_lower_pack(const double* S, int n7, double* tri, hipStream_t st) {
  if (n7 <= 0) return SFM_BA_OK;
  hipLaunchKernelGGL(k_lower_pack, dim3(n7 + 1), dim3(256), 0, st, S, n7, tri);
  return check_launch("k_lower_pack");
}

int launch_lower_unpack(const double* tri, int n7, double mu, double* S, hipStream_t st) {
  if (n7 <= 0) return SFM_BA_OK;
  hipLaunchKernelGGL(k_lower_unpack, dim3(n7 + 1), dim3(256), 0, st, tri, n7, mu, S);
  return check_launch("k_lower_unpack");
}

int launch_back_substitute(const sfm_ba_plan* p, const double* A, const double* B, const double* w,
                           const double* Vinv, const double* eps_b, const double* da, double* db,
                           double* Xupd, double x_step, hipStream_t st) {
  if (p->n_groups <= 0) return SFM_BA_OK;
  {
    ProfScope ps(KID_BACK, st);
    hipLaunchKernelGGL(k_back_substitute, dim3(p->n_groups), dim3(256), 0, st, static_cast<const int32_t*>(nullptr),
                       p->d_grp_pt, p->d_pt_ptr, p->d_perm, p->d_slot_cam, A, B, w, Vinv, eps_b, da, db, Xupd, x_step);
  }
  return check_launch("k_back_substitute");
}

// the same for the far plan of a larger plan: point p of the plan is point pt_map[p] of Vinv, eps_b, db, Xupd
int launch_back_substitute_mapped(const sfm_ba_plan* p, const int32_t* pt_map, const double* A, const double* Vinv,
                                  const double* eps_b, const double* da, double* db, double* Xupd, double x_step,
                                  hipStream_t st) {
  if (p->n_groups <= 0) return SFM_BA_OK;
  {
    ProfScope ps(KID_BACK, st);
    hipLaunchKernelGGL(k_back_substitute, dim3(p->n_groups), dim3(256), 0, st, pt_map, p->d_grp_pt, p->d_pt_ptr, p->d_perm,
                       p->d_slot_cam, A, static_cast<const double*>(nullptr), static_cast<const double*>(nullptr), Vinv,
                       eps_b, da, db, Xupd, x_step);
  }
  return check_launch("k_back_substitute");
}

int launch_update_cameras(double* R, double* T, double* C, double* P, const double* K, int k_stride,
                          const double* da, int m, int first, int held, double q_step, double c_step,
                          int32_t* flags, double* blk, int32_t* blk_flags, hipStream_t st) {
  if (m <= 0) return SFM_BA_OK;
  {
    ProfScope ps(KID_UPDATE, st);
    hipLaunchKernelGGL(k_update_cameras, dim3((m + 63) / 64), dim3(64), 0, st, R, T, C, P, K, k_stride, da, m,
                     first, held, q_step, c_step, flags, blk, blk_flags);
  }
  return check_launch("k_update_cameras");
}

int launch_cost_partial(const sfm_ba_plan* p, const double* P, const double* X, const int32_t* cam,
                        const int32_t* pt, const double* xt, double* ssq, void* ws, hipStream_t st) {
  double* campart = reinterpret_cast<double*>(static_cast<char*>(ws) + p->ws_campart);
  if (p->n_items > 0) {
    {
    ProfScope ps(KID_COST_ITEMS, st);
    hipLaunchKernelGGL(k_cost_items, dim3(p->n_items), dim3(kCamThreads), 0, st, p->d_item_beg, p->d_item_end,
                       p->d_cam_obs, P, X, cam, pt, xt, campart);
  }
    int rc = check_launch("k_cost_items");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_cost_camera_sum, dim3((p->m + 255) / 256), dim3(256), 0, st, p->d_cam_item_ptr, campart,
                     p->m, ssq);
  return check_launch("k_cost_camera_sum");
}

// single rank: per-camera sums and the cost in one launch (same order of additions as the two kernels)
__global__ __launch_bounds__(256) void k_cost_sum_finish(const int32_t* __restrict__ cam_item_ptr,
                                                         const double* __restrict__ part, int m,
                                                         double* __restrict__ ssq, double* __restrict__ cost) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  double a = 0.0;
  for (int j = t; j < m; j += 256) {
    double sj = 0.0;
    for (int it = cam_item_ptr[j]; it < cam_item_ptr[j + 1]; ++it) sj += part[it];
    ssq[j] = sj;
    a += sqrt(sj);
  }
  red[t] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (t == 0) cost[0] = red[0] / (double)m;
}

int launch_cost(const sfm_ba_plan* p, const double* P, const double* X, const int32_t* cam, const int32_t* pt,
                const double* xt, double* ssq, double* cost, void* ws, hipStream_t st) {
  double* campart = reinterpret_cast<double*>(static_cast<char*>(ws) + p->ws_campart);
  if (p->n_items > 0) {
    {
      ProfScope ps(KID_COST_ITEMS, st);
      hipLaunchKernelGGL(k_cost_items, dim3(p->n_items), dim3(kCamThreads), 0, st, p->d_item_beg, p->d_item_end,
                         p->d_cam_obs, P, X, cam, pt, xt, campart);
    }
    int rc = check_launch("k_cost_items");
    if (rc) return rc;
  }
  {
    ProfScope ps(KID_COST_FINISH, st);
    hipLaunchKernelGGL(k_cost_sum_finish, dim3(1), dim3(256), 0, st, p->d_cam_item_ptr, campart, p->m, ssq, cost);
  }
  return check_launch("k_cost_sum_finish");
}

// ------------------------------------------------------------------------------------------
// cost without any allocation: the per-item partial sums live in a file-scope device array instead
// of the caller's workspace.  Every GPU holds its own copy of the array at its own address; the
// address is looked up once per device (the array's name is only a host-side handle: handed to a
// kernel as it stands it is not a device address).
// ------------------------------------------------------------------------------------------
constexpr int kCostStaticItems = 65536;  // per-camera work items (of <= 1024 observations) the array holds
static __device__ __align__(256) double g_cost_scratch[kCostStaticItems];
constexpr int kMaxDevices = 64;
static double* g_cost_scratch_ptr[kMaxDevices];  // per device, null until looked up
static std::mutex g_cost_scratch_mu;

static int cost_scratch_for_current_device(double** out) {
  int dev = -1;
  SFMBA_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return SFM_BA_EINVAL;
  std::lock_guard<std::mutex> lk(g_cost_scratch_mu);
  if (!g_cost_scratch_ptr[dev]) {
    void* ptr = nullptr;
    SFMBA_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_cost_scratch)));
    g_cost_scratch_ptr[dev] = static_cast<double*>(ptr);
  }
  *out = g_cost_scratch_ptr[dev];
  return SFM_BA_OK;
}

}  // namespace sfmba

// Called when a plan is created (weakly bound there): the address of the array on the plan's device is
// looked up then, so that sfm_ba_cost_noalloc itself - which may run under stream capture - finds it
// and makes no runtime call besides its launches.
extern "C" int sfmba_cost_scratch_warm(void) {
  double* ptr = nullptr;
  return sfmba::cost_scratch_for_current_device(&ptr);
}

namespace sfmba {

int launch_cost_noalloc(const sfm_ba_plan* p, const double* P, const double* X, const int32_t* cam, const int32_t* pt,
                        const double* xt, double* ssq, double* cost, hipStream_t st) {
  if (p->n_items > kCostStaticItems) return SFM_BA_ETOOBIG;
  double* campart = nullptr;
  const int rc0 = cost_scratch_for_current_device(&campart);
  if (rc0) return rc0;
  // (no timing scopes here: they create events)
  if (p->n_items > 0) {
    hipLaunchKernelGGL(k_cost_items, dim3(p->n_items), dim3(kCamThreads), 0, st, p->d_item_beg, p->d_item_end,
                       p->d_cam_obs, P, X, cam, pt, xt, campart);
    const int rc = check_launch("k_cost_items");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_cost_sum_finish, dim3(1), dim3(256), 0, st, p->d_cam_item_ptr, campart, p->m, ssq, cost);
  return check_launch("k_cost_sum_finish");
}

int launch_cost_finish(const double* ssq, int m, double* cost, hipStream_t st) {
  {
    ProfScope ps(KID_COST_FINISH, st);
    hipLaunchKernelGGL(k_cost_finish, dim3(1), dim3(256), 0, st, ssq, m, cost);
  }
  return check_launch("k_cost_finish");
}

}  // namespace sfmba
