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
  ST_MAX_DIST, ST_KSHORT, ST_N_CAND, ST_COUNT
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
                                const int32_t* __restrict__ pt, PairEntry* __restrict__ entries,
                                int32_t* __restrict__ pair_entry, int32_t* __restrict__ ent_pt) {
  const int64_t pos = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (pos >= E) return;
  const int32_t ts = sorted_slot[pos];
  const PairEntry en = rec[ts];
  entries[pos] = en;
  pair_entry[ts] = (int32_t)pos;
  ent_pt[pos] = pt[en.row_a / 112u];  // (both observations of an entry belong to one point)
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
// (force: every block keeps partial results - the far plan of a larger plan)
__global__ void k_block_chunks(const int32_t* __restrict__ blk_ptr, int32_t nb, int force, int32_t* __restrict__ chunks,
                               int32_t* __restrict__ multi, int32_t* __restrict__ parts) {
  const int32_t b = blockIdx.x * kT + threadIdx.x;
  if (b > nb) return;
  const int32_t ch = b < nb ? (blk_ptr[b + 1] - blk_ptr[b] + kPairChunk - 1) / kPairChunk : 0;
  const bool with_parts = ch > 1 || (force && ch > 0);
  chunks[b] = ch;
  multi[b] = with_parts ? 1 : 0;
  parts[b] = with_parts ? ch : 0;
}

__global__ void k_pair_items_fill(const int32_t* __restrict__ blk_ptr, const int32_t* __restrict__ item_off,
                                  const int32_t* __restrict__ multi_off, const int32_t* __restrict__ part_off, int32_t nb,
                                  int force, int32_t* __restrict__ pit_blk, int32_t* __restrict__ pit_beg,
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
    pit_part[it0 + c] = (ch > 1 || force) ? p0 + c : -1;
  }
  if (ch > 1 || force) {
    const int32_t q = multi_off[b];
    mb_blk[q] = b;
    mb_part[q] = p0;
    mb_cnt[q] = ch;
  }
}

// ---- near / far points and the runs of the fused assembly (the rules of ba_common.h, as the host builder applies them) ----
// observations of the points with v cameras / a camera span of v, v <= 8
__global__ void k_fz_hist(const int32_t* __restrict__ pt_ptr, const int32_t* __restrict__ slot_cam, int32_t n,
                          unsigned long long* __restrict__ hist) {
  __shared__ unsigned long long sh[18];
  if (threadIdx.x < 18) sh[threadIdx.x] = 0;
  __syncthreads();
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  if (i < n) {
    const int32_t b = pt_ptr[i], e = pt_ptr[i + 1], k = e - b;
    if (k >= 1 && k <= 8) {
      atomicAdd(&sh[k], (unsigned long long)k);
      const int32_t span = slot_cam[e - 1] - slot_cam[b] + 1;
      if (span <= 8) atomicAdd(&sh[9 + span], (unsigned long long)k);
    }
  }
  __syncthreads();
  if (threadIdx.x < 18 && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

// window and sort key of every point that no earlier class took; a point that fits no window of this
// class gets the largest key
__global__ void k_fz_desc(const int32_t* __restrict__ pt_ptr, const int32_t* __restrict__ slot_cam,
                          const int32_t* __restrict__ taken, int32_t n, int slots, int lanes, int32_t m,
                          int32_t* __restrict__ desc, uint64_t* __restrict__ key, int32_t* __restrict__ ident,
                          int32_t* __restrict__ n_cand) {
  const int32_t i = blockIdx.x * kT + threadIdx.x;
  bool ok = false;
  if (i < n) {
    int32_t d[kFzMaxLanes];
    uint64_t k64 = 0;
    ok = taken[i] < 0 && fz_point_window(slot_cam + pt_ptr[i], pt_ptr[i + 1] - pt_ptr[i], slots, lanes, m, d, &k64);
    for (int s = 0; s < lanes; ++s) desc[(int64_t)lanes * i + s] = ok ? d[s] : -1;
    key[i] = ok ? k64 : (uint64_t(1) << 33);
    ident[i] = i;
  }
  const unsigned long long any = __ballot(ok);
  if (any && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(any)) atomicAdd(n_cand, (int32_t)__builtin_popcountll(any));
}

__device__ inline bool fz_same_window(const int32_t* __restrict__ desc, int lanes, int32_t a, int32_t b) {
  bool same = true;
  for (int s = 0; s < lanes; ++s) same = same && desc[(int64_t)lanes * a + s] == desc[(int64_t)lanes * b + s];
  return same;
}

// candidates in sorted order: a window starts where the camera list changes (element n_cand: 0)
__global__ void k_fz_win_flags(const int32_t* __restrict__ sorted_pt, const int32_t* __restrict__ desc, int lanes,
                               int32_t n_cand, int32_t* __restrict__ flag) {
  const int32_t q = blockIdx.x * kT + threadIdx.x;
  if (q > n_cand) return;
  flag[q] = q < n_cand && (q == 0 || !fz_same_window(desc, lanes, sorted_pt[q], sorted_pt[q - 1])) ? 1 : 0;
}

// first candidate of every window (element n_win: n_cand)
__global__ void k_fz_win_starts(const int32_t* __restrict__ flag, const int32_t* __restrict__ wid_excl, int32_t n_cand,
                                int32_t* __restrict__ wstart) {
  const int32_t q = blockIdx.x * kT + threadIdx.x;
  if (q > n_cand) return;
  if (q == n_cand)
    wstart[wid_excl[n_cand]] = n_cand;
  else if (flag[q])
    wstart[wid_excl[q]] = q;
}

// candidates and windows that stay: windows with at least min_pts points (elements n_cand / n_win: 0)
__global__ void k_fz_keep(const int32_t* __restrict__ flag, const int32_t* __restrict__ wid_excl,
                          const int32_t* __restrict__ wstart, int32_t n_cand, int32_t n_win, int32_t min_pts,
                          int32_t* __restrict__ keep, int32_t* __restrict__ wkeep) {
  const int32_t q = blockIdx.x * kT + threadIdx.x;
  if (q <= n_cand) {
    int32_t kp = 0;
    if (q < n_cand) {
      const int32_t w = wid_excl[q] + flag[q] - 1;
      kp = wstart[w + 1] - wstart[w] >= min_pts ? 1 : 0;
    }
    keep[q] = kp;
  }
  if (q <= n_win) wkeep[q] = q < n_win && wstart[q + 1] - wstart[q] >= min_pts ? 1 : 0;
}

// near points at their positions: point id, window, whether a window / a run starts there, position of every point
__global__ void k_fz_place(const int32_t* __restrict__ sorted_pt, const int32_t* __restrict__ flag,
                           const int32_t* __restrict__ wid_excl, const int32_t* __restrict__ keep,
                           const int32_t* __restrict__ newpos, const int32_t* __restrict__ newwid,
                           const int32_t* __restrict__ desc, int lanes, int32_t n_cand, int32_t n_near, int32_t pmax,
                           int32_t* __restrict__ fz_pt, int32_t* __restrict__ near_pos, int32_t* __restrict__ pos_key,
                           int32_t* __restrict__ win_start, int32_t* __restrict__ run_flag, int32_t* __restrict__ key_cam) {
  const int32_t q = blockIdx.x * kT + threadIdx.x;
  if (q == n_cand) run_flag[n_near] = 0;
  if (q >= n_cand || !keep[q]) return;
  const int32_t pos = newpos[q], i = sorted_pt[q];
  const int32_t key = newwid[wid_excl[q] + flag[q] - 1];
  fz_pt[pos] = i;
  near_pos[i] = pos;
  pos_key[pos] = key;
  win_start[pos] = flag[q];
  run_flag[pos] = (flag[q] || pos % pmax == 0) ? 1 : 0;
  if (flag[q])
    for (int s = 0; s < lanes; ++s) key_cam[(int64_t)lanes * key + s] = desc[(int64_t)lanes * i + s];
}

__global__ void k_fz_runs(const int32_t* __restrict__ run_flag, const int32_t* __restrict__ rid,
                          const int32_t* __restrict__ win_start, const int32_t* __restrict__ pos_key,
                          const int32_t* __restrict__ key_cam, int lanes, int32_t n_near, int32_t n_keys, int32_t pmax,
                          int32_t* __restrict__ run_pt0, int32_t* __restrict__ run_key, int32_t* __restrict__ run_cam,
                          int32_t* __restrict__ key_run, int32_t* __restrict__ rng_run) {
  const int32_t pos = blockIdx.x * kT + threadIdx.x;
  if (pos > n_near) return;
  if (pos == n_near) {
    const int32_t n_runs = rid[n_near];
    run_pt0[n_runs] = n_near;
    key_run[n_keys] = n_runs;
    rng_run[(n_near + pmax - 1) / pmax] = n_runs;
    return;
  }
  if (!run_flag[pos]) return;
  const int32_t r = rid[pos], key = pos_key[pos];
  run_pt0[r] = pos;
  run_key[r] = key;
  const int32_t first = key_cam[(int64_t)lanes * key];
  for (int s = 0; s < lanes; ++s) {
    const int32_t c = key_cam[(int64_t)lanes * key + s];
    run_cam[(int64_t)lanes * r + s] = c >= 0 ? c : first;
  }
  if (win_start[pos]) key_run[key] = r;
  if (pos % pmax == 0) rng_run[pos / pmax] = r;
}

// observation of every (position, window slot) of one class; the points of the class are marked as taken
__global__ void k_slot_obs_fill(const int32_t* __restrict__ slot_pt, const int32_t* __restrict__ slot_cam,
                                const int32_t* __restrict__ perm, const int32_t* __restrict__ near_pos,
                                const int32_t* __restrict__ desc, int lanes, int cls, int64_t N,
                                int32_t* __restrict__ slot_obs, int32_t* __restrict__ taken) {
  const int64_t s = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (s >= N) return;
  const int32_t i = slot_pt[s];
  const int32_t pos = near_pos[i];
  if (pos < 0) return;
  const int32_t c = slot_cam[s];
  int sl = 0;
  for (int q = 0; q < lanes; ++q)
    if (desc[(int64_t)lanes * i + q] == c) sl = q;
  slot_obs[(int64_t)pos * lanes + sl] = perm[s];
  taken[i] = cls;  // (every slot of the point writes the same value)
}

// far points (element n: 0) and far observations (element N: 0), for the prefix sums that number them
__global__ void k_far_flags(const int32_t* __restrict__ taken, const int32_t* __restrict__ pt, int32_t n, int64_t N,
                            int32_t* __restrict__ pflag, int32_t* __restrict__ oflag) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i <= n) pflag[i] = i < n && taken[i] < 0 ? 1 : 0;
  if (i <= N) oflag[i] = i < N && taken[pt[i]] < 0 ? 1 : 0;
}

__global__ void k_far_fill(const int32_t* __restrict__ pflag, const int32_t* __restrict__ pnum,
                           const int32_t* __restrict__ oflag, const int32_t* __restrict__ onum,
                           const int32_t* __restrict__ cam, const int32_t* __restrict__ pt, int32_t n, int64_t N,
                           int32_t* __restrict__ far_pt, int32_t* __restrict__ far_obs, int32_t* __restrict__ far_cam,
                           int32_t* __restrict__ far_lpt, int32_t* __restrict__ far_gpt) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i < n && pflag[i]) far_pt[pnum[i]] = (int32_t)i;
  if (i < N && oflag[i]) {
    const int32_t at = onum[i];
    far_obs[at] = (int32_t)i;
    far_cam[at] = cam[i];
    far_lpt[at] = pnum[pt[i]];
    far_gpt[at] = pt[i];
  }
}

// what the finishing kernel adds for every block: one candidate per (window, slot a <= slot b) of a class;
// its sort key is the block of the two cameras (nb: none - a slot without a camera, or a pair without a
// shared point).  `at`: first candidate of this class in the list (class 0 first).
__global__ void k_fb_candidates(const int32_t* __restrict__ key_cam, int32_t n_keys, int lanes, int cls, int64_t at,
                                const int32_t* __restrict__ seen, const int32_t* __restrict__ blk_of, int32_t m, int32_t nb,
                                uint32_t* __restrict__ key, int32_t* __restrict__ val) {
  const int pairs = lanes * (lanes + 1) / 2;
  const int64_t t = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (t >= (int64_t)n_keys * pairs) return;
  const int32_t q = (int32_t)(t / pairs);
  int pr = (int)(t - (int64_t)q * pairs), a = 0;
  while (pr >= lanes - a) {  // pairs (a, b >= a) in the order a = 0: `lanes` of them, a = 1: lanes - 1, ...
    pr -= lanes - a;
    ++a;
  }
  const int b = a + pr;
  const int32_t ca = key_cam[(int64_t)lanes * q + a], cb = key_cam[(int64_t)lanes * q + b];
  int32_t id = nb;
  if (ca >= 0 && cb >= 0 && seen[(int64_t)ca * m + cb]) id = blk_of[(int64_t)ca * m + cb];
  key[at + t] = (uint32_t)id;
  val[at + t] = fz_fb_pack(cls, q, a, b);
}

// first position of every value 0 .. n_values in a sorted key list (lower bound)
__global__ void k_lower_bounds(const uint32_t* __restrict__ sorted_key, int64_t count, int32_t n_values,
                               int32_t* __restrict__ out) {
  const int32_t q = blockIdx.x * kT + threadIdx.x;
  if (q > n_values) return;
  int64_t lo = 0, hi = count;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_key[mid] < (uint32_t)q)
      lo = mid + 1;
    else
      hi = mid;
  }
  out[q] = (int32_t)lo;
}

// partial results the far plan keeps for every block of this plan (the far plan's blocks are sorted by (j, k))
__global__ void k_fb_far(const int32_t* __restrict__ blk_j, const int32_t* __restrict__ blk_k, int32_t nb,
                         const int32_t* __restrict__ far_j, const int32_t* __restrict__ far_k,
                         const int32_t* __restrict__ far_part, const int32_t* __restrict__ far_cnt, int32_t nfb, int32_t m,
                         int32_t* __restrict__ out_part, int32_t* __restrict__ out_cnt) {
  const int32_t b = blockIdx.x * kT + threadIdx.x;
  if (b >= nb) return;
  const int64_t want = (int64_t)blk_j[b] * m + blk_k[b];
  int32_t lo = 0, hi = nfb;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if ((int64_t)far_j[mid] * m + far_k[mid] < want)
      lo = mid + 1;
    else
      hi = mid;
  }
  const bool found = lo < nfb && (int64_t)far_j[lo] * m + far_k[lo] == want;
  out_part[b] = found ? far_part[lo] : -1;
  out_cnt[b] = found ? far_cnt[lo] : 0;
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

enum PlanFlags { kPlanNoFused = 1, kPlanForceParts = 2 };

int build(const int32_t* cam, const int32_t* pt, int64_t N, int32_t m_hint, int32_t n_hint, int check, int flags,
          hipStream_t st, sfm_ba_plan* p) {
  const int force = (flags & kPlanForceParts) ? 1 : 0;
  p->force_parts = force;
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
    static_assert(ST_COUNT == 11, "one initial value per statistic");
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
  if (!cam_sorted) {
    SFMBA_TRY(keep(&p->d_cam_obs, N, &bytes));
    SFMBA_TRY(sort_pairs(cam_key, cam_key + N, ident, p->d_cam_obs, N, cam_bits, scratch, st));
  }
  SFMBA_TRY(read_stats());
  if (check && h[ST_NONUNIQUE]) return SFM_BA_ENONUNIQUE;
  if (N * 112 >= (int64_t(1) << 32)) return SFM_BA_ETOOBIG;  // 32-bit byte offsets into A
  // 32-bit byte offsets into T and the per-observation array behind it, counted from the start of g
  // that precedes them in the workspace
  if (N * 32 + 512 + (pairs + kPairPad) * 32 >= (int64_t(1) << 32)) return SFM_BA_ETOOBIG;
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
  SFMBA_TRY(keep(&p->d_ent_pt, pairs, &bytes));
  k_place_entries<<<blocks_for(pairs), kT, 0, st>>>(pval_sorted, rec, pairs, pt, p->d_entries, p->d_pair_entry, p->d_ent_pt);
  k_block_starts<<<blocks_for(pairs), kT, 0, st>>>(pkey_sorted, pairs, nb, blk_ptr);

  // ---- pair work items and the partial results of split blocks ----
  int32_t *chunks, *multi, *parts, *item_off, *multi_off, *part_off;
  SFMBA_HIP(scratch.get(&chunks, 3 * ((int64_t)nb + 1)));
  SFMBA_HIP(scratch.get(&item_off, 3 * ((int64_t)nb + 1)));
  multi = chunks + (nb + 1);
  parts = multi + (nb + 1);
  multi_off = item_off + (nb + 1);
  part_off = multi_off + (nb + 1);
  k_block_chunks<<<blocks_for(nb + 1), kT, 0, st>>>(blk_ptr, nb, force, chunks, multi, parts);
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
  k_pair_items_fill<<<blocks_for(nb), kT, 0, st>>>(blk_ptr, item_off, multi_off, part_off, nb, force, p->d_pit_blk, p->d_pit_beg,
                                                   p->d_pit_end, p->d_pit_part, p->d_mb_blk, p->d_mb_part, p->d_mb_cnt);

  // ---- near points of the fused point-major assembly, far points (same rules as the host builder) ----
  // (a list that names a (camera, point) pair twice - accepted only without checks - stays on the
  // camera-pair path, which adds both; the slot table of the fused path has room for one)
  if (!(flags & kPlanNoFused) && !h[ST_NONUNIQUE]) {
    unsigned long long* hist;
    SFMBA_HIP(scratch.get(&hist, 18));
    SFMBA_HIP(hipMemsetAsync(hist, 0, 18 * sizeof(unsigned long long), st));
    k_fz_hist<<<blocks_for(n), kT, 0, st>>>(p->d_pt_ptr, p->d_slot_cam, n, hist);
    unsigned long long hh[18];
    SFMBA_HIP(hipMemcpyAsync(hh, hist, sizeof(hh), hipMemcpyDeviceToHost, st));
    SFMBA_HIP(hipStreamSynchronize(st));
    int64_t obs_by_k[9], obs_by_span[9];
    for (int v = 0; v < 9; ++v) {
      obs_by_k[v] = (int64_t)hh[v];
      obs_by_span[v] = (int64_t)hh[9 + v];
    }
    int32_t* taken;  // class in which the point is near, -1: none (yet)
    SFMBA_HIP(scratch.get(&taken, n));
    SFMBA_HIP(hipMemsetAsync(taken, 0xFF, sizeof(int32_t) * (size_t)std::max(n, 1), st));
    for (int c = 0; c < kFzClasses; ++c) {
      const int slots = c == 0 ? fz_choose_slots(obs_by_k, obs_by_span) : 16;
      if (slots <= 0) continue;
      const int lanes = fz_lanes(slots);
      // windows and sort keys; candidates sorted by key (stable: ascending point id inside a key)
      int32_t *desc, *ident, *sorted_pt;
      uint64_t *wkey, *wkey_sorted;
      SFMBA_HIP(scratch.get(&desc, (int64_t)lanes * n));
      SFMBA_HIP(scratch.get(&ident, n));
      SFMBA_HIP(scratch.get(&sorted_pt, n));
      SFMBA_HIP(scratch.get(&wkey, n));
      SFMBA_HIP(scratch.get(&wkey_sorted, n));
      SFMBA_HIP(hipMemsetAsync(d_stats + ST_N_CAND, 0, sizeof(int32_t), st));
      k_fz_desc<<<blocks_for(n), kT, 0, st>>>(p->d_pt_ptr, p->d_slot_cam, taken, n, slots, lanes, m, desc, wkey, ident,
                                             d_stats + ST_N_CAND);
      SFMBA_TRY(sort_pairs(wkey, wkey_sorted, ident, sorted_pt, n, 34, scratch, st));
      SFMBA_TRY(read_stats());
      const int32_t n_cand = h[ST_N_CAND];
      // windows of the sorted candidates, the ones long enough to keep, positions of the near points
      int32_t *flag, *wid_excl, *wstart, *keepf, *wkeep, *newpos, *newwid;
      SFMBA_HIP(scratch.get(&flag, n_cand + 1));
      SFMBA_HIP(scratch.get(&wid_excl, n_cand + 1));
      k_fz_win_flags<<<blocks_for(n_cand + 1), kT, 0, st>>>(sorted_pt, desc, lanes, n_cand, flag);
      SFMBA_TRY(scan_exclusive(flag, wid_excl, n_cand + 1, scratch, st));
      int32_t n_win = 0;
      SFMBA_HIP(hipMemcpyAsync(&n_win, wid_excl + n_cand, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SFMBA_HIP(hipStreamSynchronize(st));
      SFMBA_HIP(scratch.get(&wstart, n_win + 1));
      SFMBA_HIP(scratch.get(&keepf, n_cand + 1));
      SFMBA_HIP(scratch.get(&newpos, n_cand + 1));
      SFMBA_HIP(scratch.get(&wkeep, n_win + 1));
      SFMBA_HIP(scratch.get(&newwid, n_win + 1));
      k_fz_win_starts<<<blocks_for(n_cand + 1), kT, 0, st>>>(flag, wid_excl, n_cand, wstart);
      k_fz_keep<<<blocks_for(std::max(n_cand, n_win) + 1), kT, 0, st>>>(flag, wid_excl, wstart, n_cand, n_win,
                                                                       fz_min_window_points(c), keepf, wkeep);
      SFMBA_TRY(scan_exclusive(keepf, newpos, n_cand + 1, scratch, st));
      SFMBA_TRY(scan_exclusive(wkeep, newwid, n_win + 1, scratch, st));
      int32_t n_near = 0, n_keys = 0;
      SFMBA_HIP(hipMemcpyAsync(&n_near, newpos + n_cand, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SFMBA_HIP(hipMemcpyAsync(&n_keys, newwid + n_win, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SFMBA_HIP(hipStreamSynchronize(st));
      if (n_near <= 0) continue;
      const int32_t pmax = fz_pmax_for(n_near);
      int32_t n_runs = 0;
      int32_t *near_pos, *fz_pt, *pos_key, *win_start, *run_flag, *rid, *key_cam;
      SFMBA_HIP(scratch.get(&near_pos, n));
      SFMBA_HIP(hipMemsetAsync(near_pos, 0xFF, sizeof(int32_t) * (size_t)std::max(n, 1), st));
      SFMBA_HIP(scratch.get(&fz_pt, n_near));
      SFMBA_HIP(scratch.get(&pos_key, n_near));
      SFMBA_HIP(scratch.get(&win_start, n_near));
      SFMBA_HIP(scratch.get(&run_flag, n_near + 1));
      SFMBA_HIP(scratch.get(&rid, n_near + 1));
      SFMBA_HIP(scratch.get(&key_cam, (int64_t)lanes * n_keys));
      k_fz_place<<<blocks_for(n_cand + 1), kT, 0, st>>>(sorted_pt, flag, wid_excl, keepf, newpos, newwid, desc, lanes, n_cand,
                                                       n_near, pmax, fz_pt, near_pos, pos_key, win_start, run_flag, key_cam);
      SFMBA_TRY(scan_exclusive(run_flag, rid, n_near + 1, scratch, st));
      SFMBA_HIP(hipMemcpyAsync(&n_runs, rid + n_near, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SFMBA_HIP(hipStreamSynchronize(st));
      if (!fz_worthwhile(c, n, n_near, n_runs)) continue;
      FzClass& f = p->fz[c];
      f.slots = slots;
      f.lanes = lanes;
      f.pmax = pmax;
      f.n_near = n_near;
      f.n_runs = n_runs;
      f.n_keys = n_keys;
      f.n_ranges = (n_near + pmax - 1) / pmax;
      p->n_runs += n_runs;
      SFMBA_TRY(keep(&f.d_pt, n_near, &bytes));
      SFMBA_HIP(hipMemcpyAsync(f.d_pt, fz_pt, sizeof(int32_t) * (size_t)n_near, hipMemcpyDeviceToDevice, st));
      SFMBA_TRY(keep(&f.d_key_cam, (int64_t)lanes * n_keys, &bytes));
      SFMBA_HIP(hipMemcpyAsync(f.d_key_cam, key_cam, sizeof(int32_t) * lanes * (size_t)n_keys, hipMemcpyDeviceToDevice, st));
      SFMBA_TRY(keep(&f.d_run_pt0, n_runs + 1, &bytes));
      SFMBA_TRY(keep(&f.d_run_key, n_runs, &bytes));
      SFMBA_TRY(keep(&f.d_run_cam, (int64_t)lanes * n_runs, &bytes));
      SFMBA_TRY(keep(&f.d_key_run, n_keys + 1, &bytes));
      SFMBA_TRY(keep(&f.d_rng_run, f.n_ranges + 1, &bytes));
      k_fz_runs<<<blocks_for(n_near + 1), kT, 0, st>>>(run_flag, rid, win_start, pos_key, f.d_key_cam, lanes, n_near, n_keys,
                                                      pmax, f.d_run_pt0, f.d_run_key, f.d_run_cam, f.d_key_run, f.d_rng_run);
      SFMBA_TRY(keep(&f.d_slot_obs, (int64_t)n_near * lanes, &bytes));
      SFMBA_HIP(hipMemsetAsync(f.d_slot_obs, 0xFF, sizeof(int32_t) * (size_t)n_near * lanes, st));
      k_slot_obs_fill<<<blocks_for(N), kT, 0, st>>>(slot_pt, p->d_slot_cam, p->d_perm, near_pos, desc, lanes, c, N, f.d_slot_obs,
                                                   taken);
      {
        // steps, most runs of a range: from the run boundaries, on the host
        std::vector<int32_t> rp((size_t)n_runs + 1), rr((size_t)f.n_ranges + 1);
        SFMBA_HIP(hipMemcpyAsync(rp.data(), f.d_run_pt0, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, st));
        SFMBA_HIP(hipMemcpyAsync(rr.data(), f.d_rng_run, sizeof(int32_t) * rr.size(), hipMemcpyDeviceToHost, st));
        SFMBA_HIP(hipStreamSynchronize(st));
        const int pps = 64 / lanes;
        for (int32_t r = 0; r < n_runs; ++r) f.n_steps += (rp[(size_t)r + 1] - rp[(size_t)r] + pps - 1) / pps;
        for (int32_t g = 0; g < f.n_ranges; ++g) f.max_range_runs = std::max(f.max_range_runs, rr[(size_t)g + 1] - rr[(size_t)g]);
      }
    }
    if (p->n_runs > 0) {
      // far points (those no class took) and their observations
      int32_t *pflag, *pnum, *oflag, *onum;
      SFMBA_HIP(scratch.get(&pflag, n + 1));
      SFMBA_HIP(scratch.get(&pnum, n + 1));
      SFMBA_HIP(scratch.get(&oflag, N + 1));
      SFMBA_HIP(scratch.get(&onum, N + 1));
      k_far_flags<<<blocks_for(std::max<int64_t>(n, N) + 1), kT, 0, st>>>(taken, pt, n, N, pflag, oflag);
      SFMBA_TRY(scan_exclusive(pflag, pnum, n + 1, scratch, st));
      SFMBA_TRY(scan_exclusive(oflag, onum, N + 1, scratch, st));
      int32_t n_far = 0, n_far_obs = 0;
      SFMBA_HIP(hipMemcpyAsync(&n_far, pnum + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SFMBA_HIP(hipMemcpyAsync(&n_far_obs, onum + N, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SFMBA_HIP(hipStreamSynchronize(st));
      p->n_far = n_far;
      p->n_far_obs = n_far_obs;
      SFMBA_TRY(keep(&p->d_far_pt, n_far, &bytes));
      SFMBA_TRY(keep(&p->d_far_obs, n_far_obs, &bytes));
      SFMBA_TRY(keep(&p->d_far_cam, n_far_obs, &bytes));
      SFMBA_TRY(keep(&p->d_far_lpt, n_far_obs, &bytes));
      SFMBA_TRY(keep(&p->d_far_gpt, n_far_obs, &bytes));
      k_far_fill<<<blocks_for(std::max<int64_t>(n, N)), kT, 0, st>>>(pflag, pnum, oflag, onum, cam, pt, n, N, p->d_far_pt,
                                                                  p->d_far_obs, p->d_far_cam, p->d_far_lpt, p->d_far_gpt);
      if (n_far_obs > 0) {
        p->far = new (std::nothrow) sfm_ba_plan();
        if (!p->far) return SFM_BA_EINVAL;
        SFMBA_TRY(build(p->d_far_cam, p->d_far_lpt, n_far_obs, m, n_far, 0, kPlanNoFused | kPlanForceParts, st, p->far));
      }
      // entries of the finishing kernel: candidates (class 0 first) sorted by block (stable: ascending
      // window inside a block and class)
      int64_t n_fc = 0, at[kFzClasses];
      for (int c = 0; c < kFzClasses; ++c) {
        at[c] = n_fc;
        n_fc += (int64_t)p->fz[c].n_keys * (p->fz[c].lanes * (p->fz[c].lanes + 1) / 2);
      }
      uint32_t *fkey, *fkey_sorted;
      int32_t *fval, *fval_sorted;
      SFMBA_HIP(scratch.get(&fkey, n_fc));
      SFMBA_HIP(scratch.get(&fkey_sorted, n_fc));
      SFMBA_HIP(scratch.get(&fval, n_fc));
      SFMBA_HIP(scratch.get(&fval_sorted, n_fc));
      for (int c = 0; c < kFzClasses; ++c) {
        const FzClass& f = p->fz[c];
        if (f.n_keys <= 0) continue;
        k_fb_candidates<<<blocks_for((int64_t)f.n_keys * (f.lanes * (f.lanes + 1) / 2)), kT, 0, st>>>(
            f.d_key_cam, f.n_keys, f.lanes, c, at[c], seen, blk_of, m, nb, fkey, fval);
      }
      SFMBA_TRY(sort_pairs(fkey, fkey_sorted, fval, fval_sorted, n_fc, bits_for((int64_t)nb + 1), scratch, st));
      SFMBA_TRY(keep(&p->d_fb_ptr, nb + 1, &bytes));
      k_lower_bounds<<<blocks_for(nb + 1), kT, 0, st>>>(fkey_sorted, n_fc, nb, p->d_fb_ptr);
      int32_t n_fb = 0;
      SFMBA_HIP(hipMemcpyAsync(&n_fb, p->d_fb_ptr + nb, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SFMBA_HIP(hipStreamSynchronize(st));
      p->n_fb = n_fb;
      SFMBA_TRY(keep(&p->d_fb_ent, n_fb, &bytes));
      SFMBA_HIP(hipMemcpyAsync(p->d_fb_ent, fval_sorted, sizeof(int32_t) * (size_t)n_fb, hipMemcpyDeviceToDevice, st));
      SFMBA_TRY(keep(&p->d_fb_far_part, nb, &bytes));
      SFMBA_TRY(keep(&p->d_fb_far_cnt, nb, &bytes));
      const sfm_ba_plan* fp = p->far;
      k_fb_far<<<blocks_for(nb), kT, 0, st>>>(p->d_blk_j, p->d_blk_k, nb, fp ? fp->d_blk_j : nullptr,
                                             fp ? fp->d_blk_k : nullptr, fp ? fp->d_mb_part : nullptr,
                                             fp ? fp->d_mb_cnt : nullptr, fp ? fp->n_blocks : 0, m, p->d_fb_far_part,
                                             p->d_fb_far_cnt);
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

extern "C" int sfmba_cost_scratch_warm(void);  // ba_kernels.hip

extern "C" int sfm_ba_plan_create_device(const int32_t* d_cam_idx, const int32_t* d_pt_idx, int64_t n_obs,
                                         int32_t n_cams_hint, int32_t n_pts_hint, int check, void* stream,
                                         sfm_ba_plan** out_plan) {
  if (!out_plan) return SFM_BA_EINVAL;
  *out_plan = nullptr;
  if (!d_cam_idx || !d_pt_idx || n_obs <= 0 || n_obs >= (int64_t(1) << 31)) return SFM_BA_EINVAL;
  sfm_ba_plan* p = new (std::nothrow) sfm_ba_plan();
  if (!p) return SFM_BA_EINVAL;
  const int rc = sfmba::build(d_cam_idx, d_pt_idx, n_obs, n_cams_hint, n_pts_hint, check, 0, static_cast<hipStream_t>(stream), p);
  if (rc != SFM_BA_OK) {
    sfm_ba_plan_destroy(p);
    return rc;
  }
  (void)sfmba_cost_scratch_warm();  // (the array sfm_ba_cost_noalloc uses: looked up now, not under a capture)
  *out_plan = p;
  return SFM_BA_OK;
}
