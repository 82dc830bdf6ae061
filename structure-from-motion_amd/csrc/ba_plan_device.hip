// Visibility plan built on the device: the same index structures as the host builder in
// ba_plan.cpp (array for array), from index lists that already sit in device memory.  The work of
// `get_indices` / `Indices.__init__` (src/bundle_adjustment.py:44-67, src/sparse_ba.py:58-97)
// becomes two radix sorts (observations by (point, camera); pair entries by camera-pair block),
// a handful of prefix sums and small fill kernels.  Sorts and prefix sums come from rocPRIM; both
// are stable / integer, so every array is reproducible run to run.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <new>
#include <vector>

#include "ba_common.h"

namespace sfmba {
namespace {

constexpr int kT = 256;
inline unsigned blocks_for(int64_t n) { return (unsigned)std::max<int64_t>((n + kT - 1) / kT, 1); }

// temporary device memory, released when the builder returns (hipFree waits for the device)
struct Scratch {
  std::vector<void*> held;
  ~Scratch() {
    for (void* q : held) (void)hipFree(q);
  }
  template <typename T>
  hipError_t get(T** out, int64_t count) {
    void* q = nullptr;
    const hipError_t e = hipMalloc(&q, (size_t)std::max<int64_t>(count, 1) * sizeof(T) + 256);
    if (e == hipSuccess) held.push_back(q);
    *out = static_cast<T*>(q);
    return e;
  }
};

// statistics the host reads back between phases
enum Stat {
  ST_MAX_CAM = 0, ST_MAX_PT, ST_NEGATIVE, ST_CAM_UNSORTED, ST_KMAX, ST_EMPTY_CAM, ST_EMPTY_PT, ST_NONUNIQUE,
  ST_MAX_DIST, ST_SPAN_MAX, ST_KSHORT, ST_COUNT
};

__global__ void k_idx_stats(const int32_t* __restrict__ cam, const int32_t* __restrict__ pt, int64_t N,
                            int32_t* __restrict__ stats) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  const bool live = i < N;
  int32_t c = live ? cam[i] : 0, p = live ? pt[i] : 0;
  const bool negative = c < 0 || p < 0;
  const bool unsorted = live && i > 0 && c < cam[i - 1];
  // one atomic per wave and statistic: the maxima are reduced across the lanes first
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    c = max(c, __shfl_xor(c, s));
    p = max(p, __shfl_xor(p, s));
  }
  const bool any_negative = __any(negative), any_unsorted = __any(unsorted);
  if ((threadIdx.x & 63) == 0) {
    if (any_negative) stats[ST_NEGATIVE] = 1;
    if (any_unsorted) stats[ST_CAM_UNSORTED] = 1;
    atomicMax(&stats[ST_MAX_CAM], c);
    atomicMax(&stats[ST_MAX_PT], p);
  }
}

// histograms of camera and point ids; a wave whose lanes all carry one camera (the frame-major
// order the reference emits) adds its count once
__global__ void k_idx_count(const int32_t* __restrict__ cam, const int32_t* __restrict__ pt, int64_t N,
                            int32_t* __restrict__ cam_cnt, int32_t* __restrict__ pt_cnt) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  const bool live = i < N;
  const int32_t c = live ? cam[i] : -1;
  if (live) atomicAdd(&pt_cnt[pt[i]], 1);
  const int32_t c0 = __builtin_amdgcn_readfirstlane(c);
  const unsigned long long same = __ballot(live && c == c0);
  const unsigned long long any = __ballot(live);
  if (same == any) {
    if (live && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(any)) atomicAdd(&cam_cnt[c0], (int32_t)__builtin_popcountll(any));
  } else if (live) {
    atomicAdd(&cam_cnt[c], 1);
  }
}

__global__ void k_idx_keys(const int32_t* __restrict__ cam, const int32_t* __restrict__ pt, int64_t N, int cam_bits,
                           uint64_t* __restrict__ key, int32_t* __restrict__ val, uint32_t* __restrict__ ckey) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= N) return;
  key[i] = ((uint64_t)(uint32_t)pt[i] << cam_bits) | (uint32_t)cam[i];
  val[i] = (int32_t)i;
  if (ckey) ckey[i] = (uint32_t)cam[i];
}

// sorted (point, camera) keys -> camera and point of every slot, duplicate detection
__global__ void k_idx_unpack(const uint64_t* __restrict__ key, int64_t N, int cam_bits, int32_t* __restrict__ slot_cam,
                             int32_t* __restrict__ slot_pt, int32_t* __restrict__ stats) {
  const int64_t s = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (s >= N) return;
  const uint64_t k = key[s];
  slot_cam[s] = (int32_t)(k & ((uint64_t(1) << cam_bits) - 1));
  slot_pt[s] = (int32_t)(k >> cam_bits);
  if (s > 0 && key[s - 1] == k) stats[ST_NONUNIQUE] = 1;
}

// per point: pairs k (k + 1) / 2, group units max(k, 1); largest k; empty points.  Element n of
// both outputs is 0 so that an exclusive prefix sum over n + 1 elements ends with the total.
__global__ void k_point_counts(const int32_t* __restrict__ pt_cnt, int32_t n, int64_t* __restrict__ pairs,
                               int32_t* __restrict__ units, int32_t* __restrict__ stats) {
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    pairs[n] = 0;
    units[n] = 0;
    return;
  }
  const int32_t k = pt_cnt[i];
  pairs[i] = (int64_t)k * (k + 1) / 2;
  units[i] = (k > 0 && k <= kGroupSlots) ? k : 1;  // a point with more views than a range holds gets a range of its own
  atomicMax(&stats[ST_KMAX], k);
  if (k <= kGroupSlots) atomicMax(&stats[ST_KSHORT], k);
  if (k == 0) stats[ST_EMPTY_PT] = 1;
}

__global__ void k_camera_counts(const int32_t* __restrict__ cam_cnt, int32_t m, int32_t* __restrict__ chunks,
                                int32_t* __restrict__ stats) {
  const int32_t j = blockIdx.x * kT + threadIdx.x;
  if (j > m) return;
  if (j == m) {
    chunks[m] = 0;
    return;
  }
  const int32_t c = cam_cnt[j];
  chunks[j] = (c + kCamChunk - 1) / kCamChunk;
  if (c == 0) stats[ST_EMPTY_CAM] = 1;
}

__global__ void k_narrow(const int64_t* __restrict__ in, int32_t n, int32_t* __restrict__ out) {
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}

// a point opens a workgroup range of the per-point kernels when its first unit falls into another
// window of `window` units than its predecessor's (unit_ptr = prefix sum of max(k, 1))
__global__ void k_group_flags(const int32_t* __restrict__ unit_ptr, const int32_t* __restrict__ pt_cnt, int32_t n,
                              int32_t window, int32_t* __restrict__ flag) {
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  if (i > n) return;
  flag[i] = i < n && (i == 0 || unit_ptr[i] / window != unit_ptr[i - 1] / window || pt_cnt[i] > kGroupSlots ||
                      pt_cnt[i - 1] > kGroupSlots)
                ? 1
                : 0;
}

__global__ void k_group_fill(const int32_t* __restrict__ flag, const int32_t* __restrict__ gid, int32_t n,
                             int32_t* __restrict__ grp_pt) {
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  if (i > n) return;
  if (i == n)
    grp_pt[gid[n]] = n;  // gid[n] = number of groups
  else if (flag[i])
    grp_pt[gid[i]] = i;
}

__global__ void k_camera_items_fill(const int32_t* __restrict__ cam_ptr, const int32_t* __restrict__ item_ptr, int32_t m,
                                    int32_t* __restrict__ item_cam, int32_t* __restrict__ item_beg,
                                    int32_t* __restrict__ item_end) {
  const int32_t j = blockIdx.x * kT + threadIdx.x;
  if (j >= m) return;
  const int32_t end = cam_ptr[j + 1];
  int32_t it = item_ptr[j];
  for (int32_t b = cam_ptr[j]; b < end; b += kCamChunk, ++it) {
    item_cam[it] = j;
    item_beg[it] = b;
    item_end[it] = min(b + kCamChunk, end);
  }
}

// mark the camera pairs (j <= k) that share a point: one thread per slot, pairs with later slots
__global__ void k_mark_blocks(const int32_t* __restrict__ pt_ptr, const int32_t* __restrict__ slot_pt,
                              const int32_t* __restrict__ slot_cam, int64_t N, int32_t m, int32_t* __restrict__ seen) {
  const int64_t s = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (s >= N) return;
  const int32_t end = pt_ptr[slot_pt[s] + 1];
  const int64_t row = (int64_t)slot_cam[s] * m;
  for (int32_t b = (int32_t)s; b < end; ++b) seen[row + slot_cam[b]] = 1;
}

__global__ void k_block_list(const int32_t* __restrict__ seen, const int32_t* __restrict__ blk_of, int32_t m,
                             int32_t* __restrict__ blk_j, int32_t* __restrict__ blk_k, int32_t* __restrict__ stats) {
  const int64_t c = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (c >= (int64_t)m * m || !seen[c]) return;
  const int32_t j = (int32_t)(c / m), k = (int32_t)(c - (int64_t)j * m);
  const int32_t id = blk_of[c];
  blk_j[id] = j;
  blk_k[id] = k;
  atomicMax(&stats[ST_MAX_DIST], k - j);
}

// one thread per slot a of point i: its pairs (a, b >= a) get their block id as sort key, the pair
// slot as value, and the record the camera-pair kernel walks
__global__ void k_pair_keys(const int32_t* __restrict__ pt_ptr, const int32_t* __restrict__ tp_ptr,
                            const int32_t* __restrict__ slot_pt, const int32_t* __restrict__ slot_cam,
                            const int32_t* __restrict__ perm, const int32_t* __restrict__ blk_of, int64_t N, int32_t m,
                            uint32_t* __restrict__ key, int32_t* __restrict__ val, PairEntry* __restrict__ rec) {
  const int64_t s = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (s >= N) return;
  const int32_t i = slot_pt[s];
  const int32_t first = pt_ptr[i], k = pt_ptr[i + 1] - first, a = (int32_t)s - first;
  const int64_t row = (int64_t)slot_cam[s] * m;
  const uint32_t pa = (uint32_t)perm[s];
  int32_t ts = tp_ptr[i] + a * k - a * (a - 1) / 2;
  for (int32_t b = a; b < k; ++b, ++ts) {
    const int32_t id = blk_of[row + slot_cam[first + b]];
    key[ts] = (uint32_t)id;
    val[ts] = ts;
    PairEntry en;
    en.row_a = 112u * pa;
    en.row_b = b == a ? 16u * pa : 112u * (uint32_t)perm[first + b];
    rec[ts] = en;
  }
}

__global__ void k_place_entries(const int32_t* __restrict__ sorted_slot, const PairEntry* __restrict__ rec, int64_t E,
                                PairEntry* __restrict__ entries, int32_t* __restrict__ pair_entry) {
  const int64_t pos = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (pos >= E) return;
  const int32_t ts = sorted_slot[pos];
  entries[pos] = rec[ts];
  pair_entry[ts] = (int32_t)pos;
}

// first entry of every block from the sorted keys (every block id occurs: a block exists because a
// pair belongs to it); blk_ptr[nb] = number of entries
__global__ void k_block_starts(const uint32_t* __restrict__ key, int64_t E, int32_t nb, int32_t* __restrict__ blk_ptr) {
  const int64_t pos = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (pos >= E) return;
  if (pos == 0 || key[pos] != key[pos - 1]) blk_ptr[key[pos]] = (int32_t)pos;
  if (pos == E - 1) blk_ptr[nb] = (int32_t)E;
}

// per block: work items (chunks of kPairChunk entries), whether it is split, its partial results
__global__ void k_block_chunks(const int32_t* __restrict__ blk_ptr, int32_t nb, int32_t* __restrict__ chunks,
                               int32_t* __restrict__ multi, int32_t* __restrict__ parts) {
  const int32_t b = blockIdx.x * kT + threadIdx.x;
  if (b > nb) return;
  const int32_t ch = b < nb ? (blk_ptr[b + 1] - blk_ptr[b] + kPairChunk - 1) / kPairChunk : 0;
  chunks[b] = ch;
  multi[b] = ch > 1 ? 1 : 0;
  parts[b] = ch > 1 ? ch : 0;
}

__global__ void k_pair_items_fill(const int32_t* __restrict__ blk_ptr, const int32_t* __restrict__ item_off,
                                  const int32_t* __restrict__ multi_off, const int32_t* __restrict__ part_off, int32_t nb,
                                  int32_t* __restrict__ pit_blk, int32_t* __restrict__ pit_beg,
                                  int32_t* __restrict__ pit_end, int32_t* __restrict__ pit_part,
                                  int32_t* __restrict__ mb_blk, int32_t* __restrict__ mb_part,
                                  int32_t* __restrict__ mb_cnt) {
  const int32_t b = blockIdx.x * kT + threadIdx.x;
  if (b >= nb) return;
  const int32_t beg = blk_ptr[b], end = blk_ptr[b + 1];
  const int32_t ch = item_off[b + 1] - item_off[b];
  const int32_t it0 = item_off[b], p0 = part_off[b];
  for (int32_t c = 0; c < ch; ++c) {
    pit_blk[it0 + c] = b;
    pit_beg[it0 + c] = beg + c * kPairChunk;
    pit_end[it0 + c] = min(beg + (c + 1) * kPairChunk, end);
    pit_part[it0 + c] = ch > 1 ? p0 + c : -1;
  }
  if (ch > 1) {
    const int32_t q = multi_off[b];
    mb_blk[q] = b;
    mb_part[q] = p0;
    mb_cnt[q] = ch;
  }
}

// widest range of camera ids seen by one point (cameras ascend inside a point's slots)
__global__ void k_point_span(const int32_t* __restrict__ pt_ptr, const int32_t* __restrict__ slot_cam, int32_t n,
                             int32_t* __restrict__ stats) {
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const int32_t b = pt_ptr[i], e = pt_ptr[i + 1];
  if (e > b) atomicMax(&stats[ST_SPAN_MAX], slot_cam[e - 1] - slot_cam[b] + 1);
}

// runs of the fused assembly: a point opens a run when its window key (first camera / group)
// differs from its predecessor's or its id is a multiple of pmax
__global__ void k_run_flags(const int32_t* __restrict__ pt_ptr, const int32_t* __restrict__ slot_cam, int32_t n,
                            int32_t group, int32_t pmax, int32_t* __restrict__ flag) {
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  if (i > n) return;
  int32_t f = 0;
  if (i < n) {
    if (i == 0 || i % pmax == 0)
      f = 1;
    else
      f = slot_cam[pt_ptr[i]] / group != slot_cam[pt_ptr[i - 1]] / group ? 1 : 0;
  }
  flag[i] = f;
}

__global__ void k_run_fill(const int32_t* __restrict__ flag, const int32_t* __restrict__ rid,
                           const int32_t* __restrict__ pt_ptr, const int32_t* __restrict__ slot_cam, int32_t n,
                           int32_t group, int32_t* __restrict__ run_pt0, int32_t* __restrict__ run_base,
                           uint32_t* __restrict__ run_key, int32_t* __restrict__ run_id, int32_t pmax,
                           int32_t* __restrict__ rng_run) {
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    run_pt0[rid[n]] = n;  // rid[n] = number of runs
    rng_run[(n + pmax - 1) / pmax] = rid[n];
  } else if (flag[i]) {
    if (i % pmax == 0) rng_run[i / pmax] = rid[i];
    const int32_t r = rid[i], key = slot_cam[pt_ptr[i]] / group;
    run_pt0[r] = i;
    run_base[r] = key * group;
    run_key[r] = (uint32_t)key;
    run_id[r] = r;
  }
}

// steps of 8 points: per run its number of steps (element n_runs: 0, for the prefix sum), then the steps
__global__ void k_run_step_counts(const int32_t* __restrict__ run_pt0, int32_t n_runs, int32_t* __restrict__ cnt) {
  const int32_t r = blockIdx.x * kT + threadIdx.x;
  if (r > n_runs) return;
  cnt[r] = r < n_runs ? (run_pt0[r + 1] - run_pt0[r] + 7) / 8 : 0;
}

__global__ void k_step_fill(const int32_t* __restrict__ run_pt0, const int32_t* __restrict__ run_step0, int32_t n_runs,
                            int32_t* __restrict__ step_pt0, int32_t* __restrict__ step_run) {
  const int32_t r = blockIdx.x * kT + threadIdx.x;
  if (r > n_runs) return;
  if (r == n_runs) {
    step_run[run_step0[n_runs]] = n_runs;
    return;
  }
  int32_t k = run_step0[r];
  for (int32_t f = run_pt0[r]; f < run_pt0[r + 1]; f += 8, ++k) {
    step_pt0[k] = f;
    step_run[k] = r;
  }
}

__global__ void k_range_steps(const int32_t* __restrict__ rng_run, const int32_t* __restrict__ run_step0, int32_t n_ranges,
                              int32_t* __restrict__ rng_step) {
  const int32_t g = blockIdx.x * kT + threadIdx.x;
  if (g <= n_ranges) rng_step[g] = run_step0[rng_run[g]];
}

// observation of every (point, window slot)
__global__ void k_slot_obs_fill(const int32_t* __restrict__ pt_ptr, const int32_t* __restrict__ slot_pt,
                                const int32_t* __restrict__ slot_cam, const int32_t* __restrict__ perm, int64_t N,
                                int32_t group, int32_t* __restrict__ slot_obs) {
  const int64_t s = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (s >= N) return;
  const int32_t i = slot_pt[s];
  const int32_t base = slot_cam[pt_ptr[i]] / group * group;
  slot_obs[(int64_t)i * kFzLanesPerPoint + (slot_cam[s] - base)] = perm[s];
}

// first position of every key in the sorted key list (lower bound), key_ptr[n_keys] = n_runs
__global__ void k_key_ptr(const uint32_t* __restrict__ sorted_key, int32_t n_runs, int32_t n_keys,
                          int32_t* __restrict__ key_ptr) {
  const int32_t q = blockIdx.x * kT + threadIdx.x;
  if (q > n_keys) return;
  int32_t lo = 0, hi = n_runs;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (sorted_key[mid] < (uint32_t)q)
      lo = mid + 1;
    else
      hi = mid;
  }
  key_ptr[q] = lo;
}

template <typename T>
int scan_exclusive(const T* in, T* out, int64_t count, Scratch& scratch, hipStream_t st) {
  size_t bytes = 0;
  SFMBA_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), (size_t)count, rocprim::plus<T>(), st));
  char* tmp = nullptr;
  SFMBA_HIP(scratch.get(&tmp, (int64_t)bytes));
  SFMBA_HIP(rocprim::exclusive_scan(tmp, bytes, in, out, T(0), (size_t)count, rocprim::plus<T>(), st));
  return SFM_BA_OK;
}

template <typename K>
int sort_pairs(const K* key_in, K* key_out, const int32_t* val_in, int32_t* val_out, int64_t count, int bits,
               Scratch& scratch, hipStream_t st) {
  size_t bytes = 0;
  SFMBA_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key_in, key_out, val_in, val_out, (size_t)count, 0u, (unsigned)bits, st));
  char* tmp = nullptr;
  SFMBA_HIP(scratch.get(&tmp, (int64_t)bytes));
  SFMBA_HIP(rocprim::radix_sort_pairs(tmp, bytes, key_in, key_out, val_in, val_out, (size_t)count, 0u, (unsigned)bits, st));
  return SFM_BA_OK;
}

inline int bits_for(int64_t values) {  // bits that hold 0 .. values - 1 (at least 1)
  int b = 1;
  while ((int64_t(1) << b) < values) ++b;
  return b;
}

template <typename T>
int keep(T** field, int64_t count, int64_t* bytes) {
  *field = nullptr;
  const size_t sz = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
  SFMBA_HIP(hipMalloc(reinterpret_cast<void**>(field), sz));
  *bytes += (int64_t)sz;
  return SFM_BA_OK;
}

#define SFMBA_TRY(call)            \
  do {                             \
    const int rc__ = (call);       \
    if (rc__ != SFM_BA_OK) return rc__; \
  } while (0)

int build(const int32_t* cam, const int32_t* pt, int64_t N, int32_t m_hint, int32_t n_hint, int check, hipStream_t st,
          sfm_ba_plan* p) {
  Scratch scratch;
  int64_t bytes = 0;
  int32_t* d_stats = nullptr;
  int32_t h[ST_COUNT];
  SFMBA_HIP(scratch.get(&d_stats, ST_COUNT));
  auto read_stats = [&]() -> int {
    SFMBA_HIP(hipMemcpyAsync(h, d_stats, sizeof(h), hipMemcpyDeviceToHost, st));
    SFMBA_HIP(hipStreamSynchronize(st));
    return SFM_BA_OK;
  };
  {
    int32_t init[ST_COUNT] = {-1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    SFMBA_HIP(hipMemcpyAsync(d_stats, init, sizeof(init), hipMemcpyHostToDevice, st));
    SFMBA_HIP(hipStreamSynchronize(st));  // `init` leaves scope
  }
  k_idx_stats<<<blocks_for(N), kT, 0, st>>>(cam, pt, N, d_stats);
  SFMBA_TRY(read_stats());
  if (h[ST_NEGATIVE]) return SFM_BA_EINVAL;
  const int32_t m = m_hint > 0 ? m_hint : h[ST_MAX_CAM] + 1;
  const int32_t n = n_hint > 0 ? n_hint : h[ST_MAX_PT] + 1;
  if (h[ST_MAX_CAM] >= m || h[ST_MAX_PT] >= n) return SFM_BA_EINVAL;
  if ((int64_t)m * m >= (int64_t(1) << 31) - 1) return SFM_BA_ETOOBIG;
  const bool cam_sorted = !h[ST_CAM_UNSORTED];
  p->m = m;
  p->n = n;
  p->n_obs = N;

  // ---- counts and row pointers by point and by camera ----
  int32_t *cam_cnt, *pt_cnt, *cam_ptr;
  SFMBA_HIP(scratch.get(&cam_cnt, m + 1));
  SFMBA_HIP(scratch.get(&pt_cnt, n + 1));
  SFMBA_HIP(scratch.get(&cam_ptr, m + 1));
  SFMBA_HIP(hipMemsetAsync(cam_cnt, 0, sizeof(int32_t) * (m + 1), st));
  SFMBA_HIP(hipMemsetAsync(pt_cnt, 0, sizeof(int32_t) * (n + 1), st));
  k_idx_count<<<blocks_for(N), kT, 0, st>>>(cam, pt, N, cam_cnt, pt_cnt);
  SFMBA_TRY(keep(&p->d_pt_ptr, n + 1, &bytes));
  SFMBA_TRY(scan_exclusive(pt_cnt, p->d_pt_ptr, n + 1, scratch, st));
  SFMBA_TRY(scan_exclusive(cam_cnt, cam_ptr, m + 1, scratch, st));

  int64_t* pairs64;
  int32_t *units, *unit_ptr, *cam_chunks;
  SFMBA_HIP(scratch.get(&pairs64, 2 * ((int64_t)n + 1)));
  SFMBA_HIP(scratch.get(&units, n + 1));
  SFMBA_HIP(scratch.get(&unit_ptr, n + 1));
  SFMBA_HIP(scratch.get(&cam_chunks, m + 1));
  k_point_counts<<<blocks_for(n + 1), kT, 0, st>>>(pt_cnt, n, pairs64, units, d_stats);
  k_camera_counts<<<blocks_for(m + 1), kT, 0, st>>>(cam_cnt, m, cam_chunks, d_stats);
  int64_t* tp64 = pairs64 + (n + 1);
  SFMBA_TRY(scan_exclusive(pairs64, tp64, n + 1, scratch, st));
  SFMBA_TRY(scan_exclusive(units, unit_ptr, n + 1, scratch, st));
  SFMBA_TRY(keep(&p->d_cam_item_ptr, m + 1, &bytes));
  SFMBA_TRY(scan_exclusive(cam_chunks, p->d_cam_item_ptr, m + 1, scratch, st));
  int64_t pairs = 0;
  int32_t n_items = 0;
  SFMBA_HIP(hipMemcpyAsync(&pairs, tp64 + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  SFMBA_HIP(hipMemcpyAsync(&n_items, p->d_cam_item_ptr + m, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SFMBA_TRY(read_stats());
  if (check) {
    if (m_hint <= 0 && h[ST_EMPTY_CAM]) return SFM_BA_EMISSING_VIEW;
    if (n_hint <= 0 && h[ST_EMPTY_PT]) return SFM_BA_EMISSING_POINT;
  }

  // ---- observations sorted by (point, camera) ----
  const int cam_bits = bits_for(m), pt_bits = bits_for(n);
  uint64_t *key64, *key64_sorted;
  int32_t *ident, *slot_pt;
  uint32_t* cam_key = nullptr;
  SFMBA_HIP(scratch.get(&key64, N));
  SFMBA_HIP(scratch.get(&key64_sorted, N));
  SFMBA_HIP(scratch.get(&ident, N));
  SFMBA_HIP(scratch.get(&slot_pt, N));
  if (!cam_sorted) SFMBA_HIP(scratch.get(&cam_key, 2 * N));
  SFMBA_TRY(keep(&p->d_perm, N, &bytes));
  SFMBA_TRY(keep(&p->d_slot_cam, N, &bytes));
  k_idx_keys<<<blocks_for(N), kT, 0, st>>>(cam, pt, N, cam_bits, key64, ident, cam_key);
  SFMBA_TRY(sort_pairs(key64, key64_sorted, ident, p->d_perm, N, cam_bits + pt_bits, scratch, st));
  k_idx_unpack<<<blocks_for(N), kT, 0, st>>>(key64_sorted, N, cam_bits, p->d_slot_cam, slot_pt, d_stats);
  k_point_span<<<blocks_for(n), kT, 0, st>>>(p->d_pt_ptr, p->d_slot_cam, n, d_stats);
  if (!cam_sorted) {
    SFMBA_TRY(keep(&p->d_cam_obs, N, &bytes));
    SFMBA_TRY(sort_pairs(cam_key, cam_key + N, ident, p->d_cam_obs, N, cam_bits, scratch, st));
  }
  SFMBA_TRY(read_stats());
  if (check && h[ST_NONUNIQUE]) return SFM_BA_ENONUNIQUE;
  const int32_t span_max = h[ST_SPAN_MAX];
  const bool empty_point = h[ST_EMPTY_PT] != 0;
  if (N * 112 >= (int64_t(1) << 32)) return SFM_BA_ETOOBIG;  // 32-bit byte offsets into A
  // 32-bit byte offsets into T, counted from the start of g that precedes it in the workspace
  if (N * 16 + 256 + (pairs + kPairPad) * 32 >= (int64_t(1) << 32)) return SFM_BA_ETOOBIG;
  p->n_pair_slots = pairs;
  p->n_entries = pairs;
  p->n_items = n_items;

  SFMBA_TRY(keep(&p->d_tp_ptr, n + 1, &bytes));
  k_narrow<<<blocks_for(n + 1), kT, 0, st>>>(tp64, n + 1, p->d_tp_ptr);

  // ---- workgroup ranges of the per-point kernels ----
  {
    int32_t *flag, *gid;
    SFMBA_HIP(scratch.get(&flag, n + 1));
    SFMBA_HIP(scratch.get(&gid, n + 1));
    const int32_t window = kGroupSlots - std::max(h[ST_KSHORT], 1) + 1;
    k_group_flags<<<blocks_for(n + 1), kT, 0, st>>>(unit_ptr, pt_cnt, n, window, flag);
    SFMBA_TRY(scan_exclusive(flag, gid, n + 1, scratch, st));
    int32_t n_groups = 0;
    SFMBA_HIP(hipMemcpyAsync(&n_groups, gid + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SFMBA_HIP(hipStreamSynchronize(st));
    p->n_groups = n_groups;
    SFMBA_TRY(keep(&p->d_grp_pt, n_groups + 1, &bytes));
    k_group_fill<<<blocks_for(n + 1), kT, 0, st>>>(flag, gid, n, p->d_grp_pt);
  }

  // ---- per-camera work items ----
  SFMBA_TRY(keep(&p->d_item_cam, n_items, &bytes));
  SFMBA_TRY(keep(&p->d_item_beg, n_items, &bytes));
  SFMBA_TRY(keep(&p->d_item_end, n_items, &bytes));
  k_camera_items_fill<<<blocks_for(m), kT, 0, st>>>(cam_ptr, p->d_cam_item_ptr, m, p->d_item_cam, p->d_item_beg,
                                                    p->d_item_end);

  // ---- camera-pair blocks ----
  const int64_t cells = (int64_t)m * m;
  int32_t *seen, *blk_of;
  SFMBA_HIP(scratch.get(&seen, cells + 1));
  SFMBA_HIP(scratch.get(&blk_of, cells + 1));
  SFMBA_HIP(hipMemsetAsync(seen, 0, sizeof(int32_t) * (cells + 1), st));
  k_mark_blocks<<<blocks_for(N), kT, 0, st>>>(p->d_pt_ptr, slot_pt, p->d_slot_cam, N, m, seen);
  SFMBA_TRY(scan_exclusive(seen, blk_of, cells + 1, scratch, st));
  int32_t nb = 0;
  SFMBA_HIP(hipMemcpyAsync(&nb, blk_of + cells, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SFMBA_HIP(hipStreamSynchronize(st));
  p->n_blocks = nb;
  SFMBA_TRY(keep(&p->d_blk_j, nb, &bytes));
  SFMBA_TRY(keep(&p->d_blk_k, nb, &bytes));
  k_block_list<<<blocks_for(cells), kT, 0, st>>>(seen, blk_of, m, p->d_blk_j, p->d_blk_k, d_stats);

  // ---- pair entries ordered by (block, point) ----
  uint32_t *pkey, *pkey_sorted;
  int32_t *pval, *pval_sorted, *blk_ptr;
  PairEntry* rec;
  SFMBA_HIP(scratch.get(&pkey, pairs));
  SFMBA_HIP(scratch.get(&pkey_sorted, pairs));
  SFMBA_HIP(scratch.get(&pval, pairs));
  SFMBA_HIP(scratch.get(&pval_sorted, pairs));
  SFMBA_HIP(scratch.get(&rec, pairs));
  SFMBA_HIP(scratch.get(&blk_ptr, nb + 1));
  k_pair_keys<<<blocks_for(N), kT, 0, st>>>(p->d_pt_ptr, p->d_tp_ptr, slot_pt, p->d_slot_cam, p->d_perm, blk_of, N, m, pkey,
                                           pval, rec);
  SFMBA_TRY(sort_pairs(pkey, pkey_sorted, pval, pval_sorted, pairs, bits_for(nb), scratch, st));
  SFMBA_TRY(keep(&p->d_entries, pairs + kPairPad, &bytes));
  SFMBA_TRY(keep(&p->d_pair_entry, pairs, &bytes));
  // the camera-pair kernel fetches whole steps: valid (zero) records behind the last entry
  SFMBA_HIP(hipMemsetAsync(p->d_entries + pairs, 0, sizeof(PairEntry) * kPairPad, st));
  k_place_entries<<<blocks_for(pairs), kT, 0, st>>>(pval_sorted, rec, pairs, p->d_entries, p->d_pair_entry);
  k_block_starts<<<blocks_for(pairs), kT, 0, st>>>(pkey_sorted, pairs, nb, blk_ptr);

  // ---- pair work items and the partial results of split blocks ----
  int32_t *chunks, *multi, *parts, *item_off, *multi_off, *part_off;
  SFMBA_HIP(scratch.get(&chunks, 3 * ((int64_t)nb + 1)));
  SFMBA_HIP(scratch.get(&item_off, 3 * ((int64_t)nb + 1)));
  multi = chunks + (nb + 1);
  parts = multi + (nb + 1);
  multi_off = item_off + (nb + 1);
  part_off = multi_off + (nb + 1);
  k_block_chunks<<<blocks_for(nb + 1), kT, 0, st>>>(blk_ptr, nb, chunks, multi, parts);
  SFMBA_TRY(scan_exclusive(chunks, item_off, nb + 1, scratch, st));
  SFMBA_TRY(scan_exclusive(multi, multi_off, nb + 1, scratch, st));
  SFMBA_TRY(scan_exclusive(parts, part_off, nb + 1, scratch, st));
  int32_t totals[3] = {0, 0, 0};
  SFMBA_HIP(hipMemcpyAsync(&totals[0], item_off + nb, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SFMBA_HIP(hipMemcpyAsync(&totals[1], multi_off + nb, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SFMBA_HIP(hipMemcpyAsync(&totals[2], part_off + nb, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SFMBA_TRY(read_stats());
  p->max_cam_dist = h[ST_MAX_DIST];
  p->n_pair_items = totals[0];
  p->n_multi = totals[1];
  p->n_parts = totals[2];
  SFMBA_TRY(keep(&p->d_pit_blk, totals[0], &bytes));
  SFMBA_TRY(keep(&p->d_pit_beg, totals[0], &bytes));
  SFMBA_TRY(keep(&p->d_pit_end, totals[0], &bytes));
  SFMBA_TRY(keep(&p->d_pit_part, totals[0], &bytes));
  SFMBA_TRY(keep(&p->d_mb_blk, totals[1], &bytes));
  SFMBA_TRY(keep(&p->d_mb_part, totals[1], &bytes));
  SFMBA_TRY(keep(&p->d_mb_cnt, totals[1], &bytes));
  k_pair_items_fill<<<blocks_for(nb), kT, 0, st>>>(blk_ptr, item_off, multi_off, part_off, nb, p->d_pit_blk, p->d_pit_beg,
                                                   p->d_pit_end, p->d_pit_part, p->d_mb_blk, p->d_mb_part, p->d_mb_cnt);

  // ---- runs of the fused point-major assembly (same rule as the host builder) ----
  FzShape sh;
  if (fz_shape(n, span_max, empty_point, &sh)) {
    int32_t *flag, *rid;
    SFMBA_HIP(scratch.get(&flag, n + 1));
    SFMBA_HIP(scratch.get(&rid, n + 1));
    k_run_flags<<<blocks_for(n + 1), kT, 0, st>>>(p->d_pt_ptr, p->d_slot_cam, n, sh.group, sh.pmax, flag);
    SFMBA_TRY(scan_exclusive(flag, rid, n + 1, scratch, st));
    int32_t n_runs = 0;
    SFMBA_HIP(hipMemcpyAsync(&n_runs, rid + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SFMBA_HIP(hipStreamSynchronize(st));
    if (fz_worthwhile(n, n_runs)) {
      p->fz_slots = sh.slots;
      p->fz_group = sh.group;
      p->fz_pmax = sh.pmax;
      p->n_runs = n_runs;
      p->n_keys = (m - 1) / sh.group + 1;
      uint32_t *run_key, *run_key_sorted;
      int32_t* run_id;
      SFMBA_HIP(scratch.get(&run_key, n_runs));
      SFMBA_HIP(scratch.get(&run_key_sorted, n_runs));
      SFMBA_HIP(scratch.get(&run_id, n_runs));
      SFMBA_TRY(keep(&p->d_run_pt0, n_runs + 1, &bytes));
      SFMBA_TRY(keep(&p->d_run_base, n_runs, &bytes));
      SFMBA_TRY(keep(&p->d_slot_obs, (int64_t)n * kFzLanesPerPoint, &bytes));
      SFMBA_TRY(keep(&p->d_key_ptr, p->n_keys + 1, &bytes));
      SFMBA_TRY(keep(&p->d_key_run, n_runs, &bytes));
      p->n_ranges = (n + sh.pmax - 1) / sh.pmax;
      SFMBA_TRY(keep(&p->d_rng_run, p->n_ranges + 1, &bytes));
      k_run_fill<<<blocks_for(n + 1), kT, 0, st>>>(flag, rid, p->d_pt_ptr, p->d_slot_cam, n, sh.group, p->d_run_pt0,
                                                  p->d_run_base, run_key, run_id, sh.pmax, p->d_rng_run);
      SFMBA_HIP(hipMemsetAsync(p->d_slot_obs, 0xFF, sizeof(int32_t) * (size_t)n * kFzLanesPerPoint, st));
      k_slot_obs_fill<<<blocks_for(N), kT, 0, st>>>(p->d_pt_ptr, slot_pt, p->d_slot_cam, p->d_perm, N, sh.group,
                                                   p->d_slot_obs);
      {
        int32_t *step_cnt, *run_step0;
        SFMBA_HIP(scratch.get(&step_cnt, n_runs + 1));
        SFMBA_HIP(scratch.get(&run_step0, n_runs + 1));
        k_run_step_counts<<<blocks_for(n_runs + 1), kT, 0, st>>>(p->d_run_pt0, n_runs, step_cnt);
        SFMBA_TRY(scan_exclusive(step_cnt, run_step0, n_runs + 1, scratch, st));
        int32_t n_steps = 0;
        SFMBA_HIP(hipMemcpyAsync(&n_steps, run_step0 + n_runs, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        SFMBA_HIP(hipStreamSynchronize(st));
        p->n_steps = n_steps;
        {
          std::vector<int32_t> rr((size_t)p->n_ranges + 1);
          SFMBA_HIP(hipMemcpy(rr.data(), p->d_rng_run, sizeof(int32_t) * rr.size(), hipMemcpyDeviceToHost));
          for (int32_t g = 0; g < p->n_ranges; ++g)
            p->fz_max_range_runs = std::max(p->fz_max_range_runs, rr[(size_t)g + 1] - rr[(size_t)g]);
        }
        SFMBA_TRY(keep(&p->d_step_pt0, n_steps, &bytes));
        SFMBA_TRY(keep(&p->d_step_run, n_steps + 1, &bytes));
        SFMBA_TRY(keep(&p->d_rng_step, p->n_ranges + 1, &bytes));
        k_step_fill<<<blocks_for(n_runs + 1), kT, 0, st>>>(p->d_run_pt0, run_step0, n_runs, p->d_step_pt0, p->d_step_run);
        k_range_steps<<<blocks_for(p->n_ranges + 1), kT, 0, st>>>(p->d_rng_run, run_step0, p->n_ranges, p->d_rng_step);
      }
      SFMBA_TRY(sort_pairs(run_key, run_key_sorted, run_id, p->d_key_run, n_runs, bits_for(p->n_keys), scratch, st));
      k_key_ptr<<<blocks_for(p->n_keys + 1), kT, 0, st>>>(run_key_sorted, n_runs, p->n_keys, p->d_key_ptr);
    }
  }
  SFMBA_HIP(hipGetLastError());
  SFMBA_HIP(hipStreamSynchronize(st));
  p->device_bytes = bytes;
  plan_set_workspace(p);
  return SFM_BA_OK;
}

}  // namespace
}  // namespace sfmba

extern "C" int sfm_ba_plan_create_device(const int32_t* d_cam_idx, const int32_t* d_pt_idx, int64_t n_obs,
                                         int32_t n_cams_hint, int32_t n_pts_hint, int check, void* stream,
                                         sfm_ba_plan** out_plan) {
  if (!out_plan) return SFM_BA_EINVAL;
  *out_plan = nullptr;
  if (!d_cam_idx || !d_pt_idx || n_obs <= 0 || n_obs >= (int64_t(1) << 31)) return SFM_BA_EINVAL;
  sfm_ba_plan* p = new (std::nothrow) sfm_ba_plan();
  if (!p) return SFM_BA_EINVAL;
  const int rc = sfmba::build(d_cam_idx, d_pt_idx, n_obs, n_cams_hint, n_pts_hint, check, static_cast<hipStream_t>(stream), p);
  if (rc != SFM_BA_OK) {
    sfm_ba_plan_destroy(p);
    return rc;
  }
  *out_plan = p;
  return SFM_BA_OK;
}
