// Probe: do fp64 MFMA (v_mfma_f64_16x16x4) and fp64 VALU FMAs of two waves on one SIMD overlap?
// A workgroup of 512 threads puts two waves on every SIMD (waves 0-3 and 4-7).  Mode 0: all waves
// run the MFMA loop; mode 1: all run the VALU loop; mode 2: waves 0-3 MFMA, waves 4-7 VALU; mode 3:
// every wave alternates MFMA and VALU phases (the fused assembly's shape).  Prints time per mode.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_block(d4_t (&acc)[6], double a, double b) {
#pragma unroll
  for (int i = 0; i < 60; ++i) acc[i % 6] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % 6], 0, 0, 0);
}
__device__ __forceinline__ void valu_block(double (&v)[16], double a) {
#pragma unroll
  for (int i = 0; i < 512; ++i) v[i % 16] = fma(v[i % 16], a, 1e-9);
}

__global__ __launch_bounds__(512) void k_probe(int mode, int iters, double* out) {
  const int wave = threadIdx.x >> 6;
  d4_t acc[6];
  double v[16];
  for (int i = 0; i < 6; ++i) acc[i] = d4_t{0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3 + i;
  const double a = 1.0 + 1e-12 * threadIdx.x, b = 0.5;
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4) || mode == 3;
  const bool do_valu = mode == 1 || (mode == 2 && wave >= 4) || mode == 3;
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) mfma_block(acc, a, b);
    if (do_valu) valu_block(v, a);
  }
  double s = 0;
  for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 512 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 200;
  for (int mode = 0; mode < 6; ++mode) {
    // modes 4, 5: modes 0 and 3 with one wave per SIMD (256 threads)
    const int threads = mode >= 4 ? 256 : 512;
    const int kmode = mode == 4 ? 0 : (mode == 5 ? 3 : mode);
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(threads), 0, 0, kmode, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(threads), 0, 0, kmode, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: two waves; mode 0: 2 * iters * 60 MFMAs; mode 1: 2 * iters * 512 FMAs
    printf("mode %d: %.3f ms  (per iteration per wave pair: %.0f ns)\n", mode, ms, ms * 1e6 / iters);
  }
  return 0;
}
