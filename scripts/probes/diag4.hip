// Diagnostic: a 64 x 64 factorisation without cross-lane broadcasts (diag_core4, below: 4 x 4 pivot
// blocks factored by every thread for itself, panel rows by forward substitution, rank-4 updates from
// LDS, two workgroup barriers per four columns) against the library's lane-broadcast one (diag_core):
// cycles (s_memtime around the call, thread 0) and accuracy against the defining products.
// Measured on MI355X: 39 k cycles against 27-29 k, same accuracy.  Per four columns: pivot to LDS and
// barrier 230-480, reading and factoring the pivot block 820-890 (about 95 dependent fp64
// instructions at 8-9 cycles each with one wave per SIMD), row solve and stores 590-720, barrier
// 100-170, update 200-490.  The dependent fp64 chain per column is as long as in the library's
// version; the barriers and the redundant arithmetic come on top.  Not used by the library.
#define SFMBA_STAMPS 1
#include "../../structure-from-motion_amd/csrc/ba_cholesky.hip"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
namespace sfmba {
void set_hip_error(hipError_t, const char*) {}
ProfScope::ProfScope(int id, hipStream_t st) : id_(id), st_(st), start_(nullptr) {}
ProfScope::~ProfScope() {}
}
namespace sfmba {
// ---- the same factorisation without cross-lane traffic ------------------------------------------
// value of `v` in lane 4 (lane / 4) + Q
template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, Q * 0x55, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, Q * 0x55, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// Factor the 64 x 64 block held in LDS (L: lower triangle, identity-padded past nb) and invert the
// factor into X (every entry of L and X is written: zeros above the diagonals); lower triangle of X
// to invL, the factor to S unless S is null.  256 threads, `pv` = 16 doubles of LDS.
//
// Thread (r, g) = (t / 4, t % 4) keeps entry (r, 4 u + g) of the block in a[u] and the same entry of
// an identity matrix stacked below it in b[u]; column operations on the stack turn the top into L
// and the bottom into inv(L)^T.  Columns are eliminated four at a time: the 4 x 4 pivot block is
// put into LDS by its owners and factored by every thread for itself (the two reciprocal square
// roots of a column pair side by side, as in factor16), the thread's row of the panel follows by
// forward substitution from the four entries its quad holds (DPP), and the remaining columns take
// a rank-4 update whose column factors are read back from LDS (broadcast reads).  No value ever
// crosses lanes through v_readlane; the price is two workgroup barriers per four columns.
__device__ __forceinline__ void diag_core4(double* __restrict__ L, double* __restrict__ X, double* __restrict__ pv,
                                           double* __restrict__ S, int n, int k0, int nb, double* __restrict__ invL,
                                           int32_t* __restrict__ info) {
  const int t = threadIdx.x;
  const int r = t >> 2, g = t & 3;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  double a[16], b[16];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = 4 * u + g;
    a[u] = c <= r ? L[r * LDT + c] : 0.0;
    b[u] = c == r ? 1.0 : 0.0;
  }
  int bad = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int j0 = 4 * k;
    if (k == 0 || k == 8) STAMP(20 + k);
    if ((t >> 4) == k) pv[t & 15] = a[k];
    __syncthreads();
    if (k == 0 || k == 8) STAMP(21 + k);
    const double p00 = pv[0], p10 = pv[4], p11 = pv[5], p20 = pv[8], p21 = pv[9], p22 = pv[10];
    const double p30 = pv[12], p31 = pv[13], p32 = pv[14], p33 = pv[15];
    // columns j0, j0 + 1
    const double w0 = p10 * p10;
    const double det0 = fma(p11, p00, -w0) - fma(p10, p10, -w0);
    const double rs0 = rsqrt(p00), rd0 = rsqrt(det0);
    const double l00 = p00 * rs0;
    const double rs1 = rd0 * l00;
    const double l10 = p10 * rs0, l20 = p20 * rs0, l30 = p30 * rs0;
    const double l21 = fma(-l20, l10, p21) * rs1, l31 = fma(-l30, l10, p31) * rs1;
    // columns j0 + 2, j0 + 3
    const double q22 = fma(-l21, l21, fma(-l20, l20, p22));
    const double q32 = fma(-l31, l21, fma(-l30, l20, p32));
    const double q33 = fma(-l31, l31, fma(-l30, l30, p33));
    const double w1 = q32 * q32;
    const double det1 = fma(q33, q22, -w1) - fma(q32, q32, -w1);
    const double rs2 = rsqrt(q22), rd1 = rsqrt(det1);
    const double l22 = q22 * rs2;
    const double rs3 = rd1 * l22;
    const double l32 = q32 * rs2;
    if (bad == 0) {
      if (!(p00 > 0.0)) bad = j0 + 1;
      else if (!(det0 > 0.0)) bad = j0 + 2;
      else if (!(q22 > 0.0)) bad = j0 + 3;
      else if (!(det1 > 0.0)) bad = j0 + 4;
    }
    if (k == 0 || k == 8) STAMP(22 + k);
    // this thread's rows of the panel: top (block) and bottom (identity)
    double l0, l1, l2, l3, m0, m1, m2, m3;
    {
      const double x0 = quad_bcast<0>(a[k]), x1 = quad_bcast<1>(a[k]), x2 = quad_bcast<2>(a[k]), x3 = quad_bcast<3>(a[k]);
      l0 = x0 * rs0;
      l1 = fma(-l0, l10, x1) * rs1;
      l2 = fma(-l1, l21, fma(-l0, l20, x2)) * rs2;
      l3 = fma(-l2, l32, fma(-l1, l31, fma(-l0, l30, x3))) * rs3;
      // entries above the diagonal (and rows above the pivot block) are not part of the factor
      l0 = r >= j0 ? l0 : 0.0;
      l1 = r >= j0 + 1 ? l1 : 0.0;
      l2 = r >= j0 + 2 ? l2 : 0.0;
      l3 = r >= j0 + 3 ? l3 : 0.0;
      const double y0 = quad_bcast<0>(b[k]), y1 = quad_bcast<1>(b[k]), y2 = quad_bcast<2>(b[k]), y3 = quad_bcast<3>(b[k]);
      m0 = y0 * rs0;
      m1 = fma(-m0, l10, y1) * rs1;
      m2 = fma(-m1, l21, fma(-m0, l20, y2)) * rs2;
      m3 = fma(-m2, l32, fma(-m1, l31, fma(-m0, l30, y3))) * rs3;
    }
    {
      const double lg = g == 0 ? l0 : (g == 1 ? l1 : (g == 2 ? l2 : l3));
      const double mg = g == 0 ? m0 : (g == 1 ? m1 : (g == 2 ? m2 : m3));
      L[r * LDT + j0 + g] = lg;
      X[(j0 + g) * LDT + r] = mg;  // bottom row r is column r of inv(L)
    }
    if (k == 0 || k == 8) STAMP(23 + k);
    __syncthreads();
    if (k == 0 || k == 8) STAMP(24 + k);
    if (k < 15) {
      // rows above the pivot block have nothing left to update; bottom rows below it are still zero
      const bool top = 16 * wave + 15 >= j0 + 4, bot = 16 * wave <= j0 + 3;
#pragma unroll
      for (int u = k + 1; u < 16; ++u) {
        const double2* lc = reinterpret_cast<const double2*>(L + (4 * u + g) * LDT + j0);
        const double2 c01 = lc[0], c23 = lc[1];
        if (top) a[u] = fma(-l3, c23.y, fma(-l2, c23.x, fma(-l1, c01.y, fma(-l0, c01.x, a[u]))));
        if (bot) b[u] = fma(-m3, c23.y, fma(-m2, c23.x, fma(-m1, c01.y, fma(-m0, c01.x, b[u]))));
      }
    }
    if (k == 0 || k == 8) STAMP(25 + k);
  }
  if (bad && t == 0 && bad <= nb) atomicCAS(info, 0, k0 + bad);
  // (the last barrier of the loop made L and X complete)
  {
    double xv[16], lv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int idx = t + 256 * it, rr = idx / NB, cc = idx % NB;
      xv[it] = X[rr * LDT + cc];
      lv[it] = L[rr * LDT + cc];
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int idx = t + 256 * it, rr = idx / NB, cc = idx % NB;
      if (cc <= rr) invL[rr * NB + cc] = xv[it];
      if (S != nullptr && rr < nb && cc <= rr) S[(int64_t)(k0 + rr) * n + k0 + cc] = lv[it];
    }
  }
}

}  // namespace sfmba
using namespace sfmba;
template <int WHICH>
__global__ __launch_bounds__(256) void k_probe(const double* S, int n, int nb, double* Lout, double* invL, int32_t* info,
                                               long long* cyc) {
  __shared__ double L[NB * LDT];
  __shared__ double X[NB * LDT];
  __shared__ double pv[16];
  const int t = threadIdx.x;
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    double v = r == c ? 1.0 : 0.0;
    if (r < nb && c < nb) v = c <= r ? S[r * n + c] : 0.0;
    L[r * LDT + c] = v;
    X[r * LDT + c] = 0.0;
  }
  __syncthreads();
  long long t0 = (long long)__builtin_amdgcn_s_memtime();
  if (WHICH == 0)
    diag_core(L, X, Lout, n, 0, nb, invL, info);
  else
    diag_core4(L, X, pv, Lout, n, 0, nb, invL, info);
  __syncthreads();
  long long t1 = (long long)__builtin_amdgcn_s_memtime();
  if (t == 0) cyc[0] = t1 - t0;
}
int main() {
  const int n = 64;
  for (int nb : {64, 44}) {
    std::mt19937 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> G(n * (n + 8)), S(n * n, 0.0);
    for (auto& v : G) v = nd(rng);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = i == j ? 1e-4 : 0.0;
        for (int k = 0; k < n + 8; ++k) s += G[i * (n + 8) + k] * G[j * (n + 8) + k] * (k < 20 ? 1e3 : 1.0);
        S[i * n + j] = s;
      }
    double *dS, *dL, *dI; int32_t* dinfo; long long* dc;
    (void)hipMalloc(&dS, n * n * 8); (void)hipMalloc(&dL, n * n * 8); (void)hipMalloc(&dI, n * n * 8);
    (void)hipMalloc(&dinfo, 4); (void)hipMalloc(&dc, 8);
    (void)hipMemcpy(dS, S.data(), n * n * 8, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
      long long cyc = 0; int info = 0;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(dL, 0, n * n * 8); (void)hipMemset(dI, 0, n * n * 8); (void)hipMemset(dinfo, 0, 4);
        if (which == 0) k_probe<0><<<1, 256>>>(dS, n, nb, dL, dI, dinfo, dc);
        else k_probe<1><<<1, 256>>>(dS, n, nb, dL, dI, dinfo, dc);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        (void)hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
      }
      std::vector<double> Lh(n * n), Ih(n * n);
      (void)hipMemcpy(Lh.data(), dL, n * n * 8, hipMemcpyDeviceToHost);
      (void)hipMemcpy(Ih.data(), dI, n * n * 8, hipMemcpyDeviceToHost);
      // L L^T = S on the nb x nb part, invL L = I on all 64 (identity padding)
      double e1 = 0, e2 = 0, smax = 0;
      for (int i = 0; i < nb; ++i)
        for (int j = 0; j <= i; ++j) {
          double s1 = 0;
          for (int k = 0; k <= j; ++k) s1 += Lh[i * n + k] * Lh[j * n + k];
          e1 = std::fmax(e1, std::fabs(s1 - S[i * n + j]));
          smax = std::fmax(smax, std::fabs(S[i * n + j]));
        }
      for (int i = 0; i < nb; ++i)
        for (int j = 0; j <= i; ++j) {
          double s2 = 0;
          for (int k = j; k <= i; ++k) s2 += Ih[i * n + k] * Lh[k * n + j];
          e2 = std::fmax(e2, std::fabs(s2 - (i == j ? 1.0 : 0.0)));
        }
      if (which == 1) {
        long long st[48];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(sfmba::g_stamps), sizeof(st));
        for (int k : {0, 8})
          printf("  step %d: pivot to LDS + barrier %lld, read + factor %lld, row solve + stores %lld, barrier %lld, update %lld\n", k,
                 st[21 + k] - st[20 + k], st[22 + k] - st[21 + k], st[23 + k] - st[22 + k], st[24 + k] - st[23 + k], st[25 + k] - st[24 + k]);
      }
      printf("nb=%d %s: %lld cycles, info %d, |L L^T - S| / |S| = %.2e, |inv(L) L - I| = %.2e\n", nb,
             which == 0 ? "diag_core " : "diag_core4", cyc, info, e1 / smax, e2);
    }
  }
  return 0;
}
