// Diagnostic: issue cost of v_readlane_b32 broadcasts against plain fp64 FMAs (one wave, s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long stamp(double& dep) {
  long long t;
  // the stamp is ordered after the value `dep` is produced and before any later use of it
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep)::"memory");
  return t;
}
__global__ void k(double* out, long long* st, double x0) {
  double a[16], l = x0 + threadIdx.x;
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = x0 * c + threadIdx.x;
  long long t0 = stamp(a[15]);
  // A: 16 x 15 independent fma with a per-lane multiplier (no broadcast)
#pragma unroll
  for (int rep = 0; rep < 16; ++rep)
#pragma unroll
    for (int c = 1; c < 16; ++c) a[c] = fma(-l, a[c - 1 == rep ? 0 : c - 1], a[c]);
  long long t1 = stamp(a[15]);
  // B: column elimination pattern without the reciprocal square root (120 broadcast + fma pairs)
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double lj = a[j] * 0.999;
    a[j] = lj;
#pragma unroll
    for (int c = j + 1; c < 16; ++c) a[c] = fma(-lj, lane_bcast(lj, c), a[c]);
  }
  long long t2 = stamp(a[15]);
  // C: dependent fma chain of 64
  double z = a[3];
#pragma unroll
  for (int i = 0; i < 64; ++i) z = fma(z, l, a[i & 15]);
  long long t3 = stamp(z);
  // D: dependent rsqrt chain of 8
  double q = fabs(z) + 1.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) q = rsqrt(q) + 1.0;
  long long t4 = stamp(q);
  // E: broadcast through LDS: one ds_write_b64 per lane then 8 ds_read_b128 of the same address
  __shared__ double sh[64];
  double acc = 0.0;
#pragma unroll
  for (int rep = 0; rep < 8; ++rep) {
    sh[threadIdx.x] = q + rep + acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int c = 0; c < 16; ++c) acc = fma(sh[c], a[c], acc);
  }
  long long t5 = stamp(acc);
  double s = z + q + acc;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += a[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) {
    st[0] = t1 - t0; st[1] = t2 - t1; st[2] = t3 - t2; st[3] = t4 - t3; st[4] = t5 - t4;
  }
}
int main() {
  double* d; long long* s; long long h[5];
  (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&s, 5 * 8);
  for (int i = 0; i < 3; ++i) { k<<<1, 64>>>(d, s, 1.000001); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h, s, sizeof(h), hipMemcpyDeviceToHost);
  printf("240 fma: %lld cyc (%.1f each)\n16 columns of bcast+fma (120 pairs): %lld cyc (%.1f each)\n64 dependent fma: %lld (%.1f each)\n8 dependent rsqrt+add: %lld (%.1f each)\n8 x (lds write + 16 broadcast reads + 16 dependent fma): %lld (%.1f per rep)\n",
         h[0], h[0] / 240.0, h[1], h[1] / 120.0, h[2], h[2] / 64.0, h[3], h[3] / 8.0, h[4], h[4] / 8.0);
  return 0;
}
