// Diagnostic: issue rate of v_fma_f64 (three register operands) per SIMD with 1, 2 and 4 waves per SIMD,
// and the same with one scalar operand.  ./dp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k_rate(double* out, long long* cyc, double sc) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-3 * l, b = 1e-3 + 1e-6 * l;
  double c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = i;
  __syncthreads();
  long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0)
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
      else
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "s"(sc));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i];
  out[l] = s;
  if (l == 0) cyc[0] = t1 - t0;
}
__global__ void k_long(double* out, long long* cyc) {
  double a = 1.0 + 1e-3 * threadIdx.x, b = 1e-9, c = 0.0;
  long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int rep = 0; rep < 400000; ++rep) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[threadIdx.x] = c;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* d; long long* c;
  (void)hipMalloc(&d, 1024 * 8); (void)hipMalloc(&c, 64);
  {
    // what a tick of s_memtime is: a long kernel timed with events and with the counter
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_long<<<1, 64>>>(d, c); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_long<<<1, 64>>>(d, c);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long ticks; (void)hipMemcpy(&ticks, c, 8, hipMemcpyDeviceToHost);
    printf("s_memtime: %lld ticks in %.1f us of kernel time = %.1f ticks per us\n", ticks, ms * 1e3, ticks / (ms * 1e3));
  }
  for (int waves : {1, 4, 8, 16}) {
    long long h0, h1;
    k_rate<0><<<1, 64 * waves>>>(d, c, 1e-3); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h0, c, 8, hipMemcpyDeviceToHost);
    k_rate<1><<<1, 64 * waves>>>(d, c, 1e-3); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h1, c, 8, hipMemcpyDeviceToHost);
    const double per_simd = waves < 4 ? 1.0 : waves / 4.0;
    printf("%2d waves on one CU (%.0f per SIMD): 1024 v_fma_f64 per wave in %lld cycles (%.2f cycles per instruction and SIMD); with a scalar operand %lld (%.2f)\n",
           waves, per_simd, h0, h0 / (1024.0 * per_simd), h1, h1 / (1024.0 * per_simd));
  }
  return 0;
}
