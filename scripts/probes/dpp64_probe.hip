// Diagnostic: semantics and issue rate of 64-bit DPP (row_newbcast) on gfx950:
//   v_fmac_f64_dpp d, a, b row_newbcast:n   ->  d += a[lane n of this lane's row of 16] * b
// ./dpp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_sem(double* out) {
  const int l = threadIdx.x;
  double a = 100.0 + l, b = 2.0, c = 0.5;
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b));
  double m;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(a));
  out[l] = c;
  out[64 + l] = m;
}
template <int MODE>
__global__ void k_rate(double* out, long long* cyc) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-3 * l, b = 1e-3;
  double c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = i;
  long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
  for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c[i]) : "v"(a), "v"(b));
      } else if (MODE == 1) {
        asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(c[i]) : "v"(a), "v"(b));
      } else {
        // what the register factorisation does today: two lane reads and a fused multiply-add on the scalar pair
        int lo = __builtin_amdgcn_readlane(__double2loint(a), 3), hi = __builtin_amdgcn_readlane(__double2hiint(a), 3);
        c[i] = fma(__hiloint2double(hi, lo), b, c[i]);
        asm volatile("" : "+v"(c[i]));
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i];
  out[l] = s;
  if (l == 0) cyc[MODE] = t1 - t0;
}
int main() {
  double* d; long long* c;
  (void)hipMalloc(&d, 256 * 8); (void)hipMalloc(&c, 64);
  k_sem<<<1, 64>>>(d);
  double h[128];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const double want_c = 0.5 + (100.0 + (l & ~15) + 3) * 2.0, want_m = 100.0 + (l & ~15) + 5;
    if (h[l] != want_c || h[64 + l] != want_m) ++bad;
  }
  printf("semantics: %d lanes differ from d += a[row lane n] * b (lane 20: %.1f, expected %.1f)\n", bad, h[20], 0.5 + 119.0 * 2);
  k_rate<0><<<1, 64>>>(d, c); k_rate<1><<<1, 64>>>(d, c); k_rate<2><<<1, 64>>>(d, c);
  (void)hipDeviceSynchronize();
  long long hc[3];
  (void)hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
  printf("256 instructions, one wave: fmac_f64_dpp %lld cycles, fmac_f64 %lld cycles, readlane x2 + fma %lld cycles\n", hc[0], hc[1], hc[2]);
  return 0;
}
