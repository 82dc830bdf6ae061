// Linear PnP: camera pose from 3D-2D correspondences (src/pose_estimation.py:21-144).
//
// The reference stacks, per correspondence, the three rows skew(x) [X 0 0; 0 X 0; 0 0 X] into a
// (3n x 12) matrix, takes its right singular vector of the smallest singular value as the 3x4
// [R|T] (or P), and snaps R to a rotation through a second, 3x3 SVD.  Here:
//   k_pnp_svd     one 64-lane workgroup per (hypothesis, chunk of <= 192 rows): one-sided Jacobi
//                 on the 12 columns (rows live in LDS, three per lane; a rotation of two columns is
//                 local to a row, the three inner products are wave reductions).  A chunk that is
//                 not alone hands on Sigma V^T (12 rows with the same Gram matrix, obtained by
//                 orthogonal transformations only, so nothing is squared); the next stage stacks
//                 16 of those per workgroup, until one workgroup is left, which emits the vector.
//   k_pnp_pose    per hypothesis: [R|T] from the vector (optionally K^-1 P), 3x3 Jacobi SVD,
//                 R = +-U V^T with det +1, T = +-T / d[0]      (cleanup_RT_mat, :21-30)
//   k_pnp_inliers per hypothesis: number of correspondences whose reprojection misses the
//                 observation by less than the threshold        (:123-133)
// Many hypotheses (RANSAC: 200 minimal samples of 6) run side by side in one launch.
#include <hip/hip_runtime.h>

#include "ba_common.h"

namespace sfmba {
namespace {

constexpr int kPnpRows = 192;  // rows per workgroup: 64 correspondences, or 16 handed-on blocks of 12
constexpr int kPnpLd = 13;     // leading dimension of the LDS images (12 columns + 1 pad)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

// rows: source of this stage's rows.  Stage 1 (pts != null): correspondence idx[h][c] (or c when
// idx is null) gives three rows; later stages: `rows_in` holds, per hypothesis, `n_in` rows of 12.
__global__ __launch_bounds__(64) void k_pnp_svd(const double* __restrict__ x, const double* __restrict__ X,
                                                const double* __restrict__ Kinv, const int32_t* __restrict__ idx,
                                                int s, const double* __restrict__ rows_in, int n_in, int chunks,
                                                double* __restrict__ rows_out, double* __restrict__ vec_out) {
  __shared__ double sa[kPnpRows * kPnpLd];
  __shared__ double sv[12 * kPnpLd];
  const int lane = threadIdx.x;
  const int h = blockIdx.x / chunks, chunk = blockIdx.x - h * chunks;
  int nrows;
  if (rows_in == nullptr) {
    const int c = chunk * 64 + lane;
    nrows = 3 * min(64, s - chunk * 64);
    double r[3][12];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 12; ++b) r[a][b] = 0.0;
    if (c < s) {
      const int64_t i = idx ? idx[(int64_t)h * s + c] : c;
      double u0 = x[3 * i], u1 = x[3 * i + 1], u2 = x[3 * i + 2];
      if (Kinv) {
        const double n0 = Kinv[0] * u0 + Kinv[1] * u1 + Kinv[2] * u2;
        const double n1 = Kinv[3] * u0 + Kinv[4] * u1 + Kinv[5] * u2;
        const double n2 = Kinv[6] * u0 + Kinv[7] * u1 + Kinv[8] * u2;
        u0 = n0; u1 = n1; u2 = n2;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double Xq = X[4 * i + q];
        // skew(u) = [0 -u2 u1; u2 0 -u0; -u1 u0 0]
        r[0][4 + q] = -u2 * Xq; r[0][8 + q] = u1 * Xq;
        r[1][q] = u2 * Xq;      r[1][8 + q] = -u0 * Xq;
        r[2][q] = -u1 * Xq;     r[2][4 + q] = u0 * Xq;
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 12; ++b) sa[(3 * lane + a) * kPnpLd + b] = r[a][b];
  } else {
    const int r0 = chunk * kPnpRows;
    nrows = min(kPnpRows, n_in - r0);
    const double* src = rows_in + ((int64_t)h * n_in + r0) * 12;
    for (int e = lane; e < kPnpRows * 12; e += 64) {
      const int rr = e / 12, cc = e - 12 * rr;
      sa[rr * kPnpLd + cc] = rr < nrows ? src[e] : 0.0;
    }
  }
  for (int e = lane; e < 144; e += 64) sv[(e / 12) * kPnpLd + (e % 12)] = (e / 12 == e % 12) ? 1.0 : 0.0;
  __syncthreads();

  // one-sided Jacobi: rotate column pairs until all are orthogonal to working precision.  Lane l
  // owns rows 3l..3l+2 of the image (zero rows past nrows) and, for l < 12, row l of V.
  double* ar = sa + 3 * lane * kPnpLd;
  double* vr = sv + (lane < 12 ? lane : 0) * kPnpLd;
  const double eps = 2.220446049250313e-16;
  for (int sweep = 0; sweep < 40; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < 11; ++p)
      for (int q = p + 1; q < 12; ++q) {
        const double a0 = ar[p], a1 = ar[kPnpLd + p], a2 = ar[2 * kPnpLd + p];
        const double b0 = ar[q], b1 = ar[kPnpLd + q], b2 = ar[2 * kPnpLd + q];
        const double pp = wave_sum(a0 * a0 + a1 * a1 + a2 * a2);
        const double qq = wave_sum(b0 * b0 + b1 * b1 + b2 * b2);
        const double pq = wave_sum(a0 * b0 + a1 * b1 + a2 * b2);
        if (fabs(pq) <= eps * sqrt(pp * qq) || pq == 0.0) continue;  // wave-uniform
        rotated = 1;
        const double zeta = (qq - pp) / (2.0 * pq);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        ar[p] = c * a0 - sn * b0;                 ar[q] = sn * a0 + c * b0;
        ar[kPnpLd + p] = c * a1 - sn * b1;        ar[kPnpLd + q] = sn * a1 + c * b1;
        ar[2 * kPnpLd + p] = c * a2 - sn * b2;    ar[2 * kPnpLd + q] = sn * a2 + c * b2;
        if (lane < 12) {
          const double vp = vr[p], vq = vr[q];
          vr[p] = c * vp - sn * vq;
          vr[q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  __syncthreads();
  // singular values = column norms
  __shared__ double sig[12];
  for (int j = 0; j < 12; ++j) {
    const double a0 = ar[j], a1 = ar[kPnpLd + j], a2 = ar[2 * kPnpLd + j];
    const double nn = wave_sum(a0 * a0 + a1 * a1 + a2 * a2);
    if (lane == 0) sig[j] = sqrt(nn);
  }
  __syncthreads();
  if (chunks > 1) {
    // hand on Sigma V^T: row j = sigma_j * (column j of V)
    double* dst = rows_out + ((int64_t)h * chunks + chunk) * 144;
    for (int e = lane; e < 144; e += 64) {
      const int j = e / 12, cc = e - 12 * j;
      dst[e] = sig[j] * sv[cc * kPnpLd + j];
    }
  } else if (lane < 12) {
    int best = 0;
    for (int j = 1; j < 12; ++j)
      if (sig[j] < sig[best]) best = j;
    vec_out[(int64_t)h * 12 + lane] = sv[lane * kPnpLd + best];
  }
}

__device__ __forceinline__ double det3x3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// one thread per hypothesis
__global__ void k_pnp_pose(const double* __restrict__ vec, const double* __restrict__ Kinv_post, int n_hyp,
                           double* __restrict__ R_out, double* __restrict__ T_out, int32_t* __restrict__ flags) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n_hyp) return;
  double rt[12];
  for (int q = 0; q < 12; ++q) rt[q] = vec[(int64_t)h * 12 + q];
  if (Kinv_post) {
    double t[12];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c)
        t[4 * r + c] = Kinv_post[3 * r] * rt[c] + Kinv_post[3 * r + 1] * rt[4 + c] + Kinv_post[3 * r + 2] * rt[8 + c];
    for (int q = 0; q < 12; ++q) rt[q] = t[q];
  }
  // a = R part (row-major 3x3), v = accumulated right rotations
  double a[9] = {rt[0], rt[1], rt[2], rt[4], rt[5], rt[6], rt[8], rt[9], rt[10]};
  double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double eps = 2.220446049250313e-16;
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double pp = 0, qq = 0, pq = 0;
        for (int r = 0; r < 3; ++r) {
          pp += a[3 * r + p] * a[3 * r + p];
          qq += a[3 * r + q] * a[3 * r + q];
          pq += a[3 * r + p] * a[3 * r + q];
        }
        if (fabs(pq) <= eps * sqrt(pp * qq) || pq == 0.0) continue;
        rotated = true;
        const double zeta = (qq - pp) / (2.0 * pq);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int r = 0; r < 3; ++r) {
          const double ap = a[3 * r + p], aq = a[3 * r + q];
          a[3 * r + p] = c * ap - sn * aq;
          a[3 * r + q] = sn * ap + c * aq;
          const double vp = v[3 * r + p], vq = v[3 * r + q];
          v[3 * r + p] = c * vp - sn * vq;
          v[3 * r + q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sg[3], d0 = 0.0;
  for (int j = 0; j < 3; ++j) {
    sg[j] = sqrt(a[j] * a[j] + a[3 + j] * a[3 + j] + a[6 + j] * a[6 + j]);
    d0 = fmax(d0, sg[j]);
  }
  // U V^T = sum_j (a_j / sigma_j) v_j^T
  double uv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 3; ++j) {
    const double inv = 1.0 / sg[j];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) uv[3 * r + c] += a[3 * r + j] * inv * v[3 * c + j];
  }
  const double det = rint(det3x3(uv));
  double sgn = 0.0;
  if (det == 1.0) sgn = 1.0;
  else if (det == -1.0) sgn = -1.0;
  flags[h] = sgn == 0.0 ? 1 : 0;  // the reference leaves its result unassigned here
  for (int q = 0; q < 9; ++q) R_out[(int64_t)h * 9 + q] = sgn * uv[q];
  for (int r = 0; r < 3; ++r) T_out[(int64_t)h * 3 + r] = sgn * rt[4 * r + 3] / d0;
}

__global__ __launch_bounds__(256) void k_pnp_inliers(const double* __restrict__ x, const double* __restrict__ X,
                                                     const double* __restrict__ K, const double* __restrict__ R,
                                                     const double* __restrict__ T, int64_t n, double threshold,
                                                     int32_t* __restrict__ counts, uint8_t* __restrict__ mask) {
  const int h = blockIdx.y;
  __shared__ double P[12];
  __shared__ int32_t wsum[4];
  if (threadIdx.x < 12) {
    const int r = threadIdx.x / 4, c = threadIdx.x % 4;
    const double* Rh = R + (int64_t)h * 9;
    const double* Th = T + (int64_t)h * 3;
    double acc = 0.0;
    for (int q = 0; q < 3; ++q) acc += K[3 * r + q] * (c < 3 ? Rh[3 * q + c] : Th[q]);
    P[threadIdx.x] = acc;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int in = 0;
  if (i < n) {
    const double X0 = X[4 * i], X1 = X[4 * i + 1], X2 = X[4 * i + 2], X3 = X[4 * i + 3];
    const double p0 = P[0] * X0 + P[1] * X1 + P[2] * X2 + P[3] * X3;
    const double p1 = P[4] * X0 + P[5] * X1 + P[6] * X2 + P[7] * X3;
    const double p2 = P[8] * X0 + P[9] * X1 + P[10] * X2 + P[11] * X3;
    const double d0 = p0 / p2 - x[3 * i], d1 = p1 / p2 - x[3 * i + 1], d2 = p2 / p2 - x[3 * i + 2];
    in = sqrt(d0 * d0 + d1 * d1 + d2 * d2) < threshold ? 1 : 0;
    if (mask) mask[(int64_t)h * n + i] = (uint8_t)in;
  }
  const int cnt = __popcll(__ballot(in));
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&counts[h], wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

}  // namespace

int64_t pnp_workspace_bytes(int n_hyp, int s) {
  // two row buffers (stage outputs alternate) and the vectors
  const int64_t chunks = (s + 63) / 64;
  return ((int64_t)n_hyp * chunks * 144 * 2 + (int64_t)n_hyp * 12) * (int64_t)sizeof(double) + 512;
}

int launch_pnp_solve(const double* x, const double* X, const double* Kinv_pre, const double* Kinv_post,
                     const int32_t* idx, int n_hyp, int s, void* ws, double* R, double* T, int32_t* flags,
                     hipStream_t st) {
  if (n_hyp <= 0 || s <= 0) return SFM_BA_EINVAL;
  int chunks = (s + 63) / 64;
  if ((int64_t)n_hyp * chunks >= (int64_t(1) << 31)) return SFM_BA_ETOOBIG;
  double* buf0 = static_cast<double*>(ws);
  double* buf1 = buf0 + (int64_t)n_hyp * chunks * 144;
  double* vec = buf1 + (int64_t)n_hyp * chunks * 144;
  k_pnp_svd<<<n_hyp * chunks, 64, 0, st>>>(x, X, Kinv_pre, idx, s, nullptr, 0, chunks, buf0, vec);
  double *in = buf0, *out = buf1;
  while (chunks > 1) {
    const int n_in = chunks * 12;
    chunks = (n_in + kPnpRows - 1) / kPnpRows;
    k_pnp_svd<<<n_hyp * chunks, 64, 0, st>>>(nullptr, nullptr, nullptr, nullptr, 0, in, n_in, chunks, out, vec);
    double* t = in;
    in = out;
    out = t;
  }
  k_pnp_pose<<<(n_hyp + 63) / 64, 64, 0, st>>>(vec, Kinv_post, n_hyp, R, T, flags);
  SFMBA_HIP(hipGetLastError());
  return SFM_BA_OK;
}

int launch_pnp_cleanup(const double* RT, const double* Kinv_post, int n, double* R, double* T, int32_t* flags,
                       hipStream_t st) {
  if (n <= 0) return SFM_BA_EINVAL;
  k_pnp_pose<<<(n + 63) / 64, 64, 0, st>>>(RT, Kinv_post, n, R, T, flags);
  SFMBA_HIP(hipGetLastError());
  return SFM_BA_OK;
}

int launch_pnp_inliers(const double* x, const double* X, const double* K, const double* R, const double* T, int n_hyp,
                       int64_t n, double threshold, int32_t* counts, uint8_t* mask, hipStream_t st) {
  if (n_hyp <= 0 || n <= 0 || n_hyp > 65535) return SFM_BA_EINVAL;
  SFMBA_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * n_hyp, st));
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)n_hyp);
  k_pnp_inliers<<<grid, 256, 0, st>>>(x, X, K, R, T, n, threshold, counts, mask);
  SFMBA_HIP(hipGetLastError());
  return SFM_BA_OK;
}

}  // namespace sfmba
