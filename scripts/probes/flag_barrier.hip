// Diagnostic: cost of a counter barrier between the workgroups of one launch (release fence, atomic
// add, spin with bounded iterations, acquire fence) against the cost of a launch boundary between
// dependent kernels.  ./flag_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_rounds(unsigned* counters, double* data, int rounds, int G, int stride8,
                                                unsigned* timeout) {
  // with stride8 only workgroups whose id is a multiple of 8 take part (one XCD if ids go round-robin)
  int wg = blockIdx.x;
  if (stride8) {
    if (wg & 7) return;
    wg >>= 3;
  }
  const int t = threadIdx.x;
  for (int r = 0; r < rounds; ++r) {
    // some payload that the next round's neighbour reads
    data[(size_t)wg * 256 + t] = (r == 0 ? 1.0 : data[(size_t)((wg + 1) % G) * 256 + t]) + 1.0;
    __syncthreads();
    if (t == 0) {
      __threadfence();
      __hip_atomic_fetch_add(&counters[r], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(&counters[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000u) {
          atomicAdd(timeout, 1u);
          break;
        }
      }
      __threadfence();
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_one(double* data, int G, int r) {
  const int wg = blockIdx.x, t = threadIdx.x;
  data[(size_t)wg * 256 + t] = (r == 0 ? 1.0 : data[(size_t)((wg + 1) % G) * 256 + t]) + 1.0;
}

int main() {
  const int rounds = 64;
  unsigned *dc, *dto;
  double* dd;
  (void)hipMalloc(&dc, rounds * 4);
  (void)hipMalloc(&dto, 4);
  (void)hipMalloc(&dd, 256 * 256 * 8 * 2);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int stride8 = 0; stride8 < 2; ++stride8)
    for (int G : {1, 8, 28, 32, 64, 224}) {
      if (stride8 && G > 64) continue;
      float best = 1e9f;
      unsigned to = 0;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(dc, 0, rounds * 4);
        (void)hipMemset(dto, 0, 4);
        (void)hipEventRecord(e0);
        k_rounds<<<stride8 ? 8 * G : G, 256>>>(dc, dd, rounds, G, stride8, dto);
        (void)hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        (void)hipMemcpy(&to, dto, 4, hipMemcpyDeviceToHost);
      }
      std::vector<double> h(256);
      (void)hipMemcpy(h.data(), dd, 256 * 8, hipMemcpyDeviceToHost);
      printf("%s G=%3d: %.2f us per round (value %.0f, expected %d, timeouts %u)\n", stride8 ? "ids 0,8,16,.." : "all ids      ", G,
             best * 1e3f / rounds, h[0], rounds + 1, to);
    }
  for (int G : {28, 224}) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      for (int r = 0; r < rounds; ++r) k_one<<<G, 256>>>(dd, G, r);
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("separate launches G=%3d: %.2f us per round\n", G, best * 1e3f / rounds);
  }
  return 0;
}
