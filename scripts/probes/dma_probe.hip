// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes (full and exec-masked waves)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__global__ void k(const double* A, double* out) {
  __shared__ __attribute__((aligned(16))) double sh[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) sh[i] = -1.0;
  __syncthreads();
  double* base = sh + 256;
  const double* src = A + 2 * lane;            // lane l supplies doubles 2l, 2l+1
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)base, 16, 0, 0);
  if (lane < 48) __builtin_amdgcn_global_load_lds((glb_void*)(src + 1000), (lds_void*)(base + 128), 16, 0, 0);
  if (lane >= 8 && lane < 16) __builtin_amdgcn_global_load_lds((glb_void*)(src + 2000), (lds_void*)(base + 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = sh[i];
}
int main() {
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  double *dA, *dO;
  hipMalloc(&dA, 4096 * 8); hipMalloc(&dO, 1024 * 8);
  hipMemcpy(dA, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dO);
  std::vector<double> o(1024);
  hipMemcpy(o.data(), dO, 1024 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) if (o[256 + i] != i) { if (bad < 5) printf("full: sh[256+%d]=%g\n", i, o[256 + i]); ++bad; }
  for (int i = 0; i < 96; ++i) if (o[384 + i] != 1000 + i) { if (bad < 10) printf("masked48: sh[384+%d]=%g\n", i, o[384 + i]); ++bad; }
  for (int i = 96; i < 128; ++i) if (o[384 + i] != -1) { if (bad < 15) printf("masked48 tail: sh[384+%d]=%g\n", i, o[384 + i]); ++bad; }
  for (int i = 0; i < 128; ++i) { double want = (i >= 16 && i < 32) ? 2000 + i : -1; if (o[512 + i] != want) { if (bad < 20) printf("mid-mask: sh[512+%d]=%g want %g\n", i, o[512 + i], want); ++bad; } }
  printf("dma probe: %d mismatches\n", bad);
  return bad != 0;
}
