// Diagnostic: the 16 x 16 factorisation + inverse with lane = column and 64-bit DPP row broadcasts
// (one wave, four replicas of the block, the identity's rows dealt to the replicas), against the
// row-per-lane v_readlane version of the library (3 660 against 3 580 cycles: the wave issues in order, so a
// column's remaining multiply-adds stand between its pivot and the next column's reciprocal chain).  ./factor16_dpp
#include "../../structure-from-motion_amd/csrc/ba_cholesky.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
namespace sfmba {
void set_hip_error(hipError_t, const char*) {}
ProfScope::ProfScope(int id, hipStream_t st) : id_(id), st_(st), start_(nullptr) {}
ProfScope::~ProfScope() {}
}
using namespace sfmba;
namespace sfmba {
// d += x[lane K of this lane's row of 16 lanes] * b;  x[lane K of the row]
template <int K>
__device__ __forceinline__ void dpp_fmac_self(double& d, double b) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(b), "n"(K));
}
template <int K>
__device__ __forceinline__ double dpp_bcast(double x) {
  double m;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(x), "n"(K));
  return m;
}

// The 16 x 16 factorisation with lane = column: lane 16 g + c holds column c of the symmetric block, one
// row per register (d[r] = entry (r, c), both triangles), the same in each of the wave's four rows of 16
// lanes, and four rows of the identity below it (e[j] = entry (4 j + g, c) of the identity).  Column k is
// eliminated without its square root: entry (r, c) -= entry (r, k) entry (k, c) / pivot_k, where entry
// (r, k) is lane k's register r - a 64-bit DPP row broadcast folded into the multiply-add - and entry
// (k, c) is the lane's own register k (symmetry); lanes up to k keep their (finished) columns.  At the end
// lane c scales its column by 1 / sqrt(pivot_c): d[r] = L(r, c) (zero above the diagonal) and
// e[j] = inv(L)(c, 4 j + g).  Returns the index + 1 of the first non-positive pivot or 0.
template <int K>
__device__ __forceinline__ void factor16_dpp_step(double (&d)[SB], double (&e)[4], int c, int& bad) {
  const double dk = dpp_bcast<K>(d[K]);
  if (!(dk > 0.0) && bad == 0) bad = K + 1;
  // 1 / pivot to full precision: hardware estimate and two Newton steps
  double y = __builtin_amdgcn_rcp(dk);
  y = fma(fma(-dk, y, 1.0), y, y);
  y = fma(fma(-dk, y, 1.0), y, y);
  const double nt = c > K ? -d[K] * y : 0.0;
#pragma unroll
  for (int r = K + 1; r < SB; ++r) dpp_fmac_self<K>(d[r], nt);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * j <= K) dpp_fmac_self<K>(e[j], nt);
  if constexpr (K + 1 < SB) factor16_dpp_step<K + 1>(d, e, c, bad);
}
__device__ __forceinline__ int factor16_dpp(double (&d)[SB], double (&e)[4], int c) {
  int bad = 0;
  factor16_dpp_step<0>(d, e, c, bad);
  double piv = 0.0;
#pragma unroll
  for (int r = 0; r < SB; ++r) piv = r == c ? d[r] : piv;
  const double sc = rsqrt(piv);
#pragma unroll
  for (int r = 0; r < SB; ++r) d[r] = r >= c ? d[r] * sc : 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) e[j] *= sc;
  return bad;
}

}  // namespace sfmba

__global__ void k_probe(const double* in, double* outL, double* outX, long long* st) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  double d[16], e[4];
#pragma unroll
  for (int r = 0; r < 16; ++r) d[r] = in[(r > c ? r : c) * 16 + (r > c ? c : r)];
#pragma unroll
  for (int j = 0; j < 4; ++j) e[j] = (c == 4 * j + g) ? 1.0 : 0.0;
  long long t0, t1;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(d[0]), "+v"(d[15])::"memory");
  const int bad = factor16_dpp(d, e, c);
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(d[0]), "+v"(d[15]), "+v"(e[0]), "+v"(e[3])::"memory");
  if (g == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) outL[r * 16 + c] = d[r];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) outX[c * 16 + 4 * j + g] = e[j];
  if (threadIdx.x == 0) { st[0] = t1 - t0; st[1] = bad; }
}

int main() {
  std::mt19937 rng(2);
  std::normal_distribution<double> nd;
  std::vector<double> G(16 * 24), A(256);
  for (auto& v : G) v = nd(rng);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = i == j ? 0.5 : 0.0;
      for (int k = 0; k < 24; ++k) s += G[i * 24 + k] * G[j * 24 + k];
      A[i * 16 + j] = s;
    }
  double *din, *dL, *dX; long long* dst; long long h[2];
  (void)hipMalloc(&din, 256 * 8); (void)hipMalloc(&dL, 256 * 8); (void)hipMalloc(&dX, 256 * 8); (void)hipMalloc(&dst, 16);
  (void)hipMemcpy(din, A.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int i = 0; i < 3; ++i) { k_probe<<<1, 64>>>(din, dL, dX, dst); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h, dst, 16, hipMemcpyDeviceToHost);
  std::vector<double> L(256), X(256);
  (void)hipMemcpy(L.data(), dL, 256 * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(X.data(), dX, 256 * 8, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, up = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k < 16; ++k) { s1 += L[i * 16 + k] * L[j * 16 + k]; s2 += X[i * 16 + k] * L[k * 16 + j]; }
      e1 = fmax(e1, fabs(s1 - A[i * 16 + j]));
      e2 = fmax(e2, fabs(s2 - (i == j ? 1.0 : 0.0)));
      if (j > i) up = fmax(up, fmax(fabs(L[i * 16 + j]), fabs(X[i * 16 + j])));
    }
  printf("factor16_dpp: %lld cycles, bad=%lld, max |L L^T - A| = %.3e, max |X L - I| = %.3e, largest entry above the diagonal %.1e\n",
         h[0], h[1], e1, e2, up);
  return 0;
}
