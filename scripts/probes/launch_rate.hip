// Diagnostic: how fast one-wave workgroups are launched (time of a grid of N workgroups of 64 threads that do
// almost nothing, with and without an LDS allocation / a large register allocation).  ./launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_DOUBLES, int REGS>
__global__ __launch_bounds__(64) void k_small(double* out, int n) {
  __shared__ double img[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
  double acc[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) acc[i] = out[(threadIdx.x + i) & 63];
  if (LDS_DOUBLES > 0) img[threadIdx.x] = acc[0];
  double s = LDS_DOUBLES > 0 ? img[(threadIdx.x + 1) & 63] : 0.0;
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += acc[i] * acc[(i + 1) % REGS];
  if (s == 12345.678 && blockIdx.x == (unsigned)n) out[0] = s;
}
template <int L, int R>
static void run(const char* what, double* d, int n) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    k_small<L, R><<<n, 64>>>(d, n);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-44s %7d workgroups: %.1f us = %.2f ns per workgroup\n", what, n, best * 1e3f, best * 1e6f / n);
}
int main() {
  double* d; (void)hipMalloc(&d, 4096); (void)hipMemset(d, 0, 4096);
  for (int n : {16384, 125250, 500000}) {
    run<0, 4>("no LDS, few registers", d, n);
    run<1056, 4>("8.4 KB of LDS, few registers", d, n);
    run<2112, 4>("16.9 KB of LDS, few registers", d, n);
    run<1056, 80>("8.4 KB of LDS, 160+ registers", d, n);
  }
  return 0;
}
