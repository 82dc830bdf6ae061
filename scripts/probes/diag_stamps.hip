// Diagnostic: phase timing of the 64 x 64 diagonal factorisation (s_memtime stamps, wave 0 lane 0).
#define SFMBA_STAMPS 1
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
__global__ void k_factor_only(const double* in, double* out, long long* st, int nw) {
  double a[16];
  const int lane = threadIdx.x & 63;
  if ((int)(threadIdx.x >> 6) >= nw) return;
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = in[(lane & 15) * 16 + c] * (lane < 16 ? 1.0 : 0.5);
  long long t0, t1;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a[0]), "+v"(a[15])::"memory");
  const int bad = sfmba::factor16(a, lane);
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a[0]), "+v"(a[15])::"memory");
#pragma unroll
  for (int c = 0; c < 16; ++c) out[lane * 16 + c] = a[c];
  if (threadIdx.x == 0) { st[0] = t1 - t0; st[1] = bad; }
}
int main() {
  {
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
    double *din, *dout; long long* dst; long long h[2];
    (void)hipMalloc(&din, 256 * 8); (void)hipMalloc(&dout, 64 * 16 * 8); (void)hipMalloc(&dst, 16);
    (void)hipMemcpy(din, A.data(), 256 * 8, hipMemcpyHostToDevice);
    for (int nw = 1; nw <= 4; ++nw) {
      for (int i = 0; i < 3; ++i) { k_factor_only<<<1, 256>>>(din, dout, dst, nw); (void)hipDeviceSynchronize(); }
      (void)hipMemcpy(h, dst, 16, hipMemcpyDeviceToHost);
      printf("factor16 alone, %d wave(s) of a 256-thread workgroup running it: %lld cycles, bad=%lld\n", nw, h[0], h[1]);
    }
  }
  const int n = 64;
  std::mt19937 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> G(n * (n + 8)), S((n + 1) * n, 0.0);
  for (auto& v : G) v = nd(rng);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = i == j ? 0.5 : 0.0;
      for (int k = 0; k < n + 8; ++k) s += G[i * (n + 8) + k] * G[j * (n + 8) + k];
      S[i * n + j] = s;
    }
  double *dS, *dI; int32_t* dinfo;
  (void)hipMalloc(&dS, S.size() * 8); (void)hipMalloc(&dI, 64 * 64 * 8); (void)hipMalloc(&dinfo, 4);
  long long st[48];
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemset(dinfo, 0, 4);
    sfmba::k_chol_diag<<<1, 256>>>(dS, n, 0, n, dI, dinfo);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(sfmba::g_stamps), sizeof(st));
  }
  for (int kk = 0; kk < 4; ++kk)
    printf(" kk=%d factor %lld  barrier %lld  jobs %lld  barrier %lld\n", kk, st[2 + 4 * kk] - st[1 + 4 * kk],
           st[3 + 4 * kk] - st[2 + 4 * kk], st[4 + 4 * kk] - st[3 + 4 * kk], (kk < 3 ? st[5 + 4 * kk] : st[16]) - st[4 + 4 * kk]);
  for (int kk = 0; kk < 4; ++kk)
    printf(" kk=%d load %lld factor16 %lld store %lld\n", kk, st[20 + kk] - st[1 + 4 * kk], st[24 + kk] - st[20 + kk], st[2 + 4 * kk] - st[24 + kk]);
  printf(" loop %lld, tail jobs %lld, barrier %lld, stores %lld, all %lld\n", st[16] - st[0], st[17] - st[16], st[18] - st[17], st[19] - st[18], st[19] - st[0]);
  // check: L L^T = S and invL L = I
  std::vector<double> Lh((n + 1) * n), Ih(64 * 64);
  (void)hipMemcpy(Lh.data(), dS, Lh.size() * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(Ih.data(), dI, Ih.size() * 8, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k <= j; ++k) s1 += Lh[i * n + k] * Lh[j * n + k];
      for (int k = j; k <= i; ++k) s2 += Ih[i * 64 + k] * Lh[k * n + j];  // lower triangle of invL only
      e1 = fmax(e1, fabs(s1 - S[i * n + j]));
      e2 = fmax(e2, fabs(s2 - (i == j ? 1.0 : 0.0)));
    }
  printf(" max |L L^T - S| = %.3e   max |invL L - I| = %.3e\n", e1, e2);
  return 0;
}
