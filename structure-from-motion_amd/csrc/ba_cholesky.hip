// Reduced camera system: blocked right-looking Cholesky of the dense symmetric positive definite
// matrix S (n x n, row-major, leading dimension n, lower triangle referenced) with the right-hand
// side carried as row n of the same buffer, so that the forward substitution L y = e happens as
// part of the panel solves.  Then a blocked backward substitution L^T x = y.
//
// Per panel of NB columns:
//   k_chol_diag   : one workgroup factors the NB x NB diagonal block in LDS and inverts its factor
//   k_chol_panel  : rows below the block (and the right-hand-side row) times inv(L_kk)^T
//   k_chol_update : trailing lower-triangle tiles  A_ij -= P_i P_j^T
#include "ba_common.h"

namespace sfmba {

constexpr int NB = 64;       // panel width = tile edge
constexpr int LDT = NB + 1;  // padded LDS row

// workspace: [n_panels][NB*NB] inverse diagonal factors, then y[n] (running right-hand side of the
// backward substitution)
int64_t cholesky_workspace_bytes(int n) {
  const int64_t np = (n + NB - 1) / NB;
  return (np * NB * NB + (int64_t)np * NB + 64) * (int64_t)sizeof(double);
}

__global__ __launch_bounds__(256) void k_chol_diag(double* __restrict__ S, int n, int k0, int nb,
                                                   double* __restrict__ invL, int32_t* __restrict__ info) {
  __shared__ double L[NB * LDT];
  __shared__ double X[NB * LDT];
  __shared__ double piv[NB];
  const int t = threadIdx.x;
  const int tr = t >> 4, tc = t & 15;
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    double v = 0.0;
    if (r < nb && c <= r) v = S[(int64_t)(k0 + r) * n + k0 + c];
    L[r * LDT + c] = v;
  }
  for (int j = 0; j < nb; ++j) {
    __syncthreads();
    const double d = L[j * LDT + j];
    if (t == 0) {
      piv[j] = d;
      if (!(d > 0.0) && atomicCAS(info, 0, k0 + j + 1) == 0) {
      }
    }
    const double id = 1.0 / d;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int r = tr + 16 * a;
      if (r <= j || r >= nb) continue;
      const double lrj = L[r * LDT + j] * id;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int c = tc + 16 * b;
        if (c > j && c <= r) L[r * LDT + c] -= lrj * L[c * LDT + j];
      }
    }
  }
  __syncthreads();
  // scale the columns: L[r][j] = numerators / sqrt(pivot)
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    if (r < nb && c <= r) {
      const double s = sqrt(piv[c]);
      const double v = (r == c) ? s : L[r * LDT + c] / s;
      L[r * LDT + c] = v;
      S[(int64_t)(k0 + r) * n + k0 + c] = v;
    }
  }
  __syncthreads();
  // inverse of the lower-triangular factor, one lane per column
  if (t < NB) {
    const int c = t;
    for (int r = 0; r < NB; ++r) X[r * LDT + c] = 0.0;
    if (c < nb) {
      X[c * LDT + c] = 1.0 / L[c * LDT + c];
      for (int r = c + 1; r < nb; ++r) {
        double s = 0.0;
        for (int q = c; q < r; ++q) s += L[r * LDT + q] * X[q * LDT + c];
        X[r * LDT + c] = -s / L[r * LDT + r];
      }
    }
  }
  __syncthreads();
  for (int idx = t; idx < NB * NB; idx += 256) invL[idx] = X[(idx / NB) * LDT + (idx % NB)];
}

// P = A_tile * invL^T for the 64-row tiles below the diagonal block; rows up to and including n
// (the right-hand-side row).  grid.x = number of row tiles.
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ S, int n, int k0, int nb,
                                                    const double* __restrict__ invL) {
  __shared__ double At[NB * LDT];
  __shared__ double It[NB * LDT];
  const int t = threadIdx.x;
  const int row0 = k0 + nb + blockIdx.x * NB;
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    double v = 0.0;
    if (row0 + r <= n && c < nb) v = S[(int64_t)(row0 + r) * n + k0 + c];
    At[r * LDT + c] = v;
    It[r * LDT + c] = invL[idx];
  }
  __syncthreads();
  const int tr = t >> 4, tc = t & 15;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int q = 0; q < nb; ++q) {
    double av[4], iv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) av[a] = At[(tr + 16 * a) * LDT + q];
#pragma unroll
    for (int b = 0; b < 4; ++b) iv[b] = It[(tc + 16 * b) * LDT + q];  // invL[c][q]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * iv[b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = row0 + tr + 16 * a;
    if (r > n) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int c = tc + 16 * b;
      if (c < nb) S[(int64_t)r * n + k0 + c] = acc[a][b];
    }
  }
}

// trailing update A_ij -= P_i P_j^T over lower-triangle tiles (ti >= tj); grid.x enumerates them
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ S, int n, int k0, int nb, int ntile) {
  __shared__ double Pi[NB * LDT];
  __shared__ double Pj[NB * LDT];
  // decode (ti, tj), ti >= tj, from the linear index
  const int lin = blockIdx.x;
  int ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= lin) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > lin) --ti;
  const int tj = lin - ti * (ti + 1) / 2;
  if (ti >= ntile) return;
  const int t = threadIdx.x;
  const int base = k0 + nb;
  const int row0 = base + ti * NB, col0 = base + tj * NB;
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    double vi = 0.0, vj = 0.0;
    if (c < nb) {
      if (row0 + r <= n) vi = S[(int64_t)(row0 + r) * n + k0 + c];
      if (col0 + r < n) vj = S[(int64_t)(col0 + r) * n + k0 + c];
    }
    Pi[r * LDT + c] = vi;
    Pj[r * LDT + c] = vj;
  }
  __syncthreads();
  const int tr = t >> 4, tc = t & 15;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int q = 0; q < nb; ++q) {
    double av[4], bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) av[a] = Pi[(tr + 16 * a) * LDT + q];
#pragma unroll
    for (int b = 0; b < 4; ++b) bv[b] = Pj[(tc + 16 * b) * LDT + q];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = row0 + tr + 16 * a;
    if (r > n) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int c = col0 + tc + 16 * b;
      if (c < n && c <= r) S[(int64_t)r * n + c] -= acc[a][b];
    }
  }
}

// backward substitution step for panel k0: x_k = invL_kk^T y_k (recomputed by every workgroup), then
// workgroup w removes L[k0.., cols of tile w]^T x_k from y.  Workgroup 0 also stores x_k.
__global__ __launch_bounds__(256) void k_chol_back(const double* __restrict__ S, int n, int k0, int nb,
                                                   const double* __restrict__ invL, double* __restrict__ y,
                                                   double* __restrict__ x) {
  __shared__ double xk[NB];
  __shared__ double part[4][NB];
  const int t = threadIdx.x;
  if (t < NB) {
    double s = 0.0;
    if (t < nb)
      for (int r = t; r < nb; ++r) s += invL[r * NB + t] * y[k0 + r];
    xk[t] = s;
    if (blockIdx.x == 0 && t < nb) x[k0 + t] = s;
  }
  __syncthreads();
  if (k0 == 0) return;
  const int c = blockIdx.x * NB + (t & 63);
  const int rq = t >> 6;  // 4 row groups
  double s = 0.0;
  if (c < k0)
    for (int r = rq; r < nb; r += 4) s += S[(int64_t)(k0 + r) * n + c] * xk[r];
  part[rq][t & 63] = s;
  __syncthreads();
  if (t < NB && c < k0) y[c] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

__global__ void k_copy_row(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

int launch_solve_reduced(double* S, int n, double* da, void* ws, int32_t* info, hipStream_t st) {
  if (n <= 0) return SFM_BA_OK;
  ProfScope ps(KID_SOLVE, st);
  const int np = (n + NB - 1) / NB;
  double* invL = static_cast<double*>(ws);
  double* y = invL + (int64_t)np * NB * NB;
  hipError_t err = hipMemsetAsync(info, 0, sizeof(int32_t), st);
  if (err != hipSuccess) {
    set_hip_error(err, "hipMemsetAsync(info)");
    return SFM_BA_EHIP;
  }
  for (int k = 0; k < np; ++k) {
    const int k0 = k * NB;
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, st, S, n, k0, nb, invL + (int64_t)k * NB * NB, info);
    const int rows_below = n + 1 - (k0 + nb);  // includes the right-hand-side row
    const int ntile = (rows_below + NB - 1) / NB;
    if (ntile > 0) {
      hipLaunchKernelGGL(k_chol_panel, dim3(ntile), dim3(256), 0, st, S, n, k0, nb, invL + (int64_t)k * NB * NB);
      if (k0 + nb < n) {
        const int64_t nt = (int64_t)ntile * (ntile + 1) / 2;
        hipLaunchKernelGGL(k_chol_update, dim3((unsigned)nt), dim3(256), 0, st, S, n, k0, nb, ntile);
      }
    }
  }
  hipLaunchKernelGGL(k_copy_row, dim3((n + 255) / 256), dim3(256), 0, st, S + (int64_t)n * n, y, n);
  for (int k = np - 1; k >= 0; --k) {
    const int k0 = k * NB;
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    const int nwg = k0 > 0 ? (k0 + NB - 1) / NB : 1;
    hipLaunchKernelGGL(k_chol_back, dim3(nwg), dim3(256), 0, st, S, n, k0, nb, invL + (int64_t)k * NB * NB, y, da);
  }
  err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, "cholesky launches");
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

}  // namespace sfmba
