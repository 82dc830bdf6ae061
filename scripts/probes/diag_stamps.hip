// Diagnostic: phase timing of the 64 x 64 diagonal factorisation (s_memtime stamps, wave 0 lane 0).
#define SFMBA_STAMPS 1
#include "../../structure-from-motion_amd/csrc/ba_cholesky.hip"
#include <cstdio>
#include <vector>
#include <random>
namespace sfmba {
void set_hip_error(hipError_t, const char*) {}
ProfScope::ProfScope(int id, hipStream_t st) : id_(id), st_(st), start_(nullptr) {}
ProfScope::~ProfScope() {}
}
int main() {
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
  long long st[32];
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemset(dinfo, 0, 4);
    sfmba::k_chol_diag<<<1, 256>>>(dS, n, 0, n, dI, dinfo);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(sfmba::g_stamps), sizeof(st));
  }
  printf("cycles since core start (100 MHz ticks? s_memtime = shader clock):\n");
  for (int kk = 0; kk < 4; ++kk)
    printf(" kk=%d factor %lld  wait/A..roles %lld  barrierB %lld\n", kk, st[2 + 4 * kk] - st[1 + 4 * kk], st[3 + 4 * kk] - st[2 + 4 * kk], st[4 + 4 * kk] - st[3 + 4 * kk]);
  printf(" loop total %lld, update tail+sync %lld, assembly %lld, stores %lld, all %lld\n", st[16] - st[0], st[17] - st[16], st[18] - st[17], st[19] - st[18], st[19] - st[0]);
  return 0;
}
