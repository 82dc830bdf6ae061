// Visibility plan builder (host side): validates the (point, viewpoint) index lists the way the
// reference's Indices class does and builds the index structures the kernels walk.
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ba_common.h"

namespace sfmba {

static thread_local std::string g_hip_error;

void set_hip_error(hipError_t err, const char* what) {
  g_hip_error = std::string(what) + ": " + hipGetErrorString(err);
}

const char* last_hip_error() { return g_hip_error.c_str(); }

template <typename T>
static int upload(const std::vector<T>& h, T** d, int64_t* bytes) {
  *d = nullptr;
  const size_t sz = std::max<size_t>(h.size(), 1) * sizeof(T);
  SFMBA_HIP(hipMalloc(reinterpret_cast<void**>(d), sz));
  if (!h.empty()) SFMBA_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *bytes += (int64_t)sz;
  return SFM_BA_OK;
}

static inline int64_t align256(int64_t v) { return (v + 255) & ~int64_t(255); }

// workspace section offsets (bytes) from the plan's counts
void plan_set_workspace(sfm_ba_plan* p) {
  int64_t off = 0;
  p->ws_g = off;
  off = align256(off + p->n_obs * 2 * (int64_t)sizeof(double));
  p->ws_T = off;
  off = align256(off + (p->n_pair_slots + kPairPad) * 4 * (int64_t)sizeof(double));
  p->ws_campart = off;
  off = align256(off + (int64_t)std::max(p->n_items, 1) * SFM_BA_CAM_RED * (int64_t)sizeof(double));
  p->ws_spart = off;
  off = align256(off + (int64_t)std::max(p->n_parts, 1) * kPairPartStride * (int64_t)sizeof(double));
  // fused point-major assembly: one partial result per run, one per window
  p->ws_fpart = off;
  off = align256(off + (int64_t)p->n_runs * fz_part_stride(p->fz_slots) * (int64_t)sizeof(double));
  p->ws_fkey = off;
  off = align256(off + (int64_t)p->n_keys * fz_part_stride(p->fz_slots) * (int64_t)sizeof(double));
  p->ws_fzero = off;  // a row of zeros
  off = align256(off + 256);
  p->ws_total = off;
}

}  // namespace sfmba

using namespace sfmba;

extern "C" const char* sfm_ba_last_hip_error(void) { return sfmba::last_hip_error(); }

extern "C" void sfm_ba_plan_destroy(sfm_ba_plan* p) {
  if (!p) return;
  void* ptrs[] = {p->d_pt_ptr,   p->d_perm,     p->d_slot_cam, p->d_tp_ptr,  p->d_grp_pt,  p->d_cam_obs, p->d_pair_entry,
                  p->d_item_cam, p->d_item_beg, p->d_item_end, p->d_cam_item_ptr, p->d_blk_j, p->d_blk_k,
                  p->d_entries,  p->d_pit_blk,  p->d_pit_beg,  p->d_pit_end, p->d_pit_part, p->d_mb_blk,
                  p->d_mb_part,  p->d_mb_cnt,   p->d_run_pt0,  p->d_run_base, p->d_slot_obs, p->d_key_ptr,
                  p->d_key_run,  p->d_rng_run,  p->d_step_pt0, p->d_step_run, p->d_rng_step};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  delete p;
}

extern "C" int sfm_ba_plan_create(const int32_t* cam_idx, const int32_t* pt_idx, int64_t N, int32_t m_hint,
                                  int32_t n_hint, int check, sfm_ba_plan** out) {
  if (!out) return SFM_BA_EINVAL;
  *out = nullptr;
  if (!cam_idx || !pt_idx || N <= 0 || N >= (int64_t(1) << 31)) return SFM_BA_EINVAL;
  int32_t maxc = -1, maxp = -1;
  for (int64_t i = 0; i < N; ++i) {
    if (cam_idx[i] < 0 || pt_idx[i] < 0) return SFM_BA_EINVAL;
    maxc = std::max(maxc, cam_idx[i]);
    maxp = std::max(maxp, pt_idx[i]);
  }
  const int32_t m = m_hint > 0 ? m_hint : maxc + 1;
  const int32_t n = n_hint > 0 ? n_hint : maxp + 1;
  if (maxc >= m || maxp >= n) return SFM_BA_EINVAL;

  // ---- sort observations by (point, camera): counting sort by camera, then stable by point ----
  std::vector<int32_t> cam_cnt(m + 1, 0), pt_cnt(n + 1, 0);
  bool cam_sorted = true;
  for (int64_t i = 0; i < N; ++i) {
    ++cam_cnt[cam_idx[i] + 1];
    ++pt_cnt[pt_idx[i] + 1];
    if (i > 0 && cam_idx[i] < cam_idx[i - 1]) cam_sorted = false;
  }
  if (check) {
    if (m_hint <= 0)
      for (int32_t j = 0; j < m; ++j)
        if (cam_cnt[j + 1] == 0) return SFM_BA_EMISSING_VIEW;
    if (n_hint <= 0)
      for (int32_t i = 0; i < n; ++i)
        if (pt_cnt[i + 1] == 0) return SFM_BA_EMISSING_POINT;
  }
  std::vector<int32_t> cam_ptr(m + 1, 0), pt_ptr(n + 1, 0);
  for (int32_t j = 0; j < m; ++j) cam_ptr[j + 1] = cam_ptr[j] + cam_cnt[j + 1];
  for (int32_t i = 0; i < n; ++i) pt_ptr[i + 1] = pt_ptr[i] + pt_cnt[i + 1];
  std::vector<int32_t> by_cam(N);
  {
    std::vector<int32_t> fill(cam_ptr.begin(), cam_ptr.end() - 1);
    for (int64_t i = 0; i < N; ++i) by_cam[fill[cam_idx[i]]++] = (int32_t)i;
  }
  std::vector<int32_t> perm(N), slot_cam(N);
  {
    std::vector<int32_t> fill(pt_ptr.begin(), pt_ptr.end() - 1);
    for (int64_t q = 0; q < N; ++q) {
      const int32_t i = by_cam[q];
      const int32_t s = fill[pt_idx[i]]++;
      perm[s] = i;
      slot_cam[s] = cam_idx[i];
    }
  }
  int32_t kshort = 0;  // kshort: the most views among the points that fit one workgroup range
  for (int32_t i = 0; i < n; ++i) {
    const int32_t k = pt_ptr[i + 1] - pt_ptr[i];
    if (k <= kGroupSlots) kshort = std::max(kshort, k);
    if (check)
      for (int32_t s = pt_ptr[i] + 1; s < pt_ptr[i + 1]; ++s)
        if (slot_cam[s] == slot_cam[s - 1]) return SFM_BA_ENONUNIQUE;
  }
  if (N * 112 >= (int64_t(1) << 32)) return SFM_BA_ETOOBIG;  // 32-bit byte offsets into A

  sfm_ba_plan* p = new (std::nothrow) sfm_ba_plan();
  if (!p) return SFM_BA_EINVAL;
  p->m = m;
  p->n = n;
  p->n_obs = N;

  // ---- per-point pair slots and workgroup ranges ----
  // A point takes max(k, 1) units; it opens a new workgroup range when its first unit falls into
  // another window of kGroupSlots - kshort + 1 units than its predecessor's.  A range then holds at
  // most kGroupSlots observations and at most kGroupSlots points, and the rule needs no sequential
  // pass (the device builder applies it with a prefix sum).  A point seen by more than kGroupSlots
  // cameras takes one unit and a range of its own (the kernels stride over its observations).
  std::vector<int32_t> tp_ptr(n + 1, 0), grp_pt;
  int64_t pairs = 0;
  {
    const int32_t window = kGroupSlots - std::max(kshort, 1) + 1;
    int64_t unit = 0, prev_unit = 0;
    bool prev_long = false;
    for (int32_t i = 0; i < n; ++i) {
      const int32_t k = pt_ptr[i + 1] - pt_ptr[i];
      const bool is_long = k > kGroupSlots;
      if (i == 0 || unit / window != prev_unit / window || is_long || prev_long) grp_pt.push_back(i);
      prev_unit = unit;
      prev_long = is_long;
      unit += is_long ? 1 : std::max(k, 1);
      tp_ptr[i] = (int32_t)pairs;
      pairs += (int64_t)k * (k + 1) / 2;
      // 32-bit byte offsets into T, counted from the start of g that precedes it in the workspace
      if (N * 16 + 256 + (pairs + kPairPad) * 32 >= (int64_t(1) << 32)) {
        delete p;
        return SFM_BA_ETOOBIG;
      }
    }
    tp_ptr[n] = (int32_t)pairs;
    grp_pt.push_back(n);
  }
  p->n_groups = (int32_t)grp_pt.size() - 1;
  p->n_pair_slots = pairs;

  // ---- per-camera work items ----
  std::vector<int32_t> item_cam, item_beg, item_end, cam_item_ptr(m + 1, 0);
  for (int32_t j = 0; j < m; ++j) {
    cam_item_ptr[j] = (int32_t)item_cam.size();
    for (int32_t b = cam_ptr[j]; b < cam_ptr[j + 1]; b += kCamChunk) {
      item_cam.push_back(j);
      item_beg.push_back(b);
      item_end.push_back(std::min(b + kCamChunk, cam_ptr[j + 1]));
    }
  }
  cam_item_ptr[m] = (int32_t)item_cam.size();
  p->n_items = (int32_t)item_cam.size();

  // ---- camera-pair blocks (j <= k) and their entries, ordered by (j, k) then by point ----
  std::vector<int32_t> blk_j, blk_k, blk_ptr;
  std::vector<PairEntry> entries;
  std::vector<int32_t> pair_entry((size_t)pairs);
  int32_t max_cam_dist = 0;
  {
    std::vector<int64_t> cnt((size_t)m * m, 0);
    for (int32_t i = 0; i < n; ++i)
      for (int32_t a = pt_ptr[i]; a < pt_ptr[i + 1]; ++a)
        for (int32_t b = a; b < pt_ptr[i + 1]; ++b) ++cnt[(size_t)slot_cam[a] * m + slot_cam[b]];
    int64_t total = 0;
    for (int32_t j = 0; j < m; ++j)
      for (int32_t k = j; k < m; ++k) {
        const int64_t c = cnt[(size_t)j * m + k];
        if (c == 0) {
          cnt[(size_t)j * m + k] = -1;
          continue;
        }
        blk_j.push_back(j);
        blk_k.push_back(k);
        max_cam_dist = std::max(max_cam_dist, k - j);
        blk_ptr.push_back((int32_t)total);
        cnt[(size_t)j * m + k] = total;
        total += c;
      }
    blk_ptr.push_back((int32_t)total);
    entries.resize((size_t)total);
    for (int32_t i = 0; i < n; ++i) {
      const int32_t k = pt_ptr[i + 1] - pt_ptr[i];
      for (int32_t a = 0; a < k; ++a)
        for (int32_t b = a; b < k; ++b) {
          const int32_t sa = pt_ptr[i] + a, sb = pt_ptr[i] + b;
          int64_t& at = cnt[(size_t)slot_cam[sa] * m + slot_cam[sb]];
          PairEntry en;
          en.row_a = 112u * (uint32_t)perm[sa];
          en.row_b = sa == sb ? 16u * (uint32_t)perm[sa] : 112u * (uint32_t)perm[sb];
          pair_entry[(size_t)tp_ptr[i] + a * k - a * (a - 1) / 2 + (b - a)] = (int32_t)at;
          entries[(size_t)at++] = en;
        }
    }
  }
  p->n_blocks = (int32_t)blk_j.size();
  p->max_cam_dist = max_cam_dist;
  p->n_entries = (int64_t)entries.size();
  // the camera-pair kernel fetches whole 32-entry steps: keep one step of valid records (and, in
  // the workspace, of T rows) behind the last entry
  entries.resize(entries.size() + kPairPad, PairEntry{0, 0});

  std::vector<int32_t> pit_blk, pit_beg, pit_end, pit_part, mb_blk, mb_part, mb_cnt;
  int32_t n_parts = 0;
  for (int32_t b = 0; b < p->n_blocks; ++b) {
    const int32_t beg = blk_ptr[b], end = blk_ptr[b + 1];
    const int32_t chunks = (end - beg + kPairChunk - 1) / kPairChunk;
    if (chunks > 1) {
      mb_blk.push_back(b);
      mb_part.push_back(n_parts);
      mb_cnt.push_back(chunks);
    }
    for (int32_t c = 0; c < chunks; ++c) {
      pit_blk.push_back(b);
      pit_beg.push_back(beg + c * kPairChunk);
      pit_end.push_back(std::min(beg + (c + 1) * kPairChunk, end));
      pit_part.push_back(chunks > 1 ? n_parts++ : -1);
    }
  }
  p->n_pair_items = (int32_t)pit_blk.size();
  p->n_parts = n_parts;
  p->n_multi = (int32_t)mb_blk.size();

  // ---- runs of the fused point-major assembly ----
  // Every point's cameras must fit a window of 6 or 8 consecutive ids.  `group` consecutive
  // first-cameras share a window (key = first camera / group, base camera = key * group); a run is
  // a maximal range of consecutive points with one key, cut at multiples of pmax.  The rule needs
  // no sequential pass either.
  std::vector<int32_t> run_pt0, run_base, slot_obs, key_ptr, key_run, rng_run, step_pt0, step_run, rng_step;
  {
    int32_t span_max = 0;
    bool empty_point = false;
    for (int32_t i = 0; i < n; ++i) {
      if (pt_ptr[i + 1] == pt_ptr[i]) {
        empty_point = true;
        continue;
      }
      span_max = std::max(span_max, slot_cam[pt_ptr[i + 1] - 1] - slot_cam[pt_ptr[i]] + 1);
    }
    FzShape sh;
    if (fz_shape(n, span_max, empty_point, &sh)) {
      int32_t prev_key = -1;
      for (int32_t i = 0; i < n; ++i) {
        const int32_t key = slot_cam[pt_ptr[i]] / sh.group;
        if (i == 0 || key != prev_key || i % sh.pmax == 0) {
          run_pt0.push_back(i);
          run_base.push_back(key * sh.group);
        }
        prev_key = key;
      }
      const int32_t n_runs = (int32_t)run_pt0.size();
      run_pt0.push_back(n);
      if (fz_worthwhile(n, n_runs)) {
        p->fz_slots = sh.slots;
        p->fz_group = sh.group;
        p->fz_pmax = sh.pmax;
        p->n_runs = n_runs;
        p->n_keys = (m - 1) / sh.group + 1;
        // a run starts at every multiple of pmax: ranges of pmax points own whole runs
        p->n_ranges = (n + sh.pmax - 1) / sh.pmax;
        rng_run.assign(p->n_ranges + 1, n_runs);
        for (int32_t r = n_runs - 1; r >= 0; --r)
          if (run_pt0[r] % sh.pmax == 0) rng_run[run_pt0[r] / sh.pmax] = r;
        for (int32_t g = 0; g < p->n_ranges; ++g)
          p->fz_max_range_runs = std::max(p->fz_max_range_runs, rng_run[g + 1] - rng_run[g]);
        // steps of 8 points, run by run
        std::vector<int32_t> run_step0(n_runs + 1, 0);
        for (int32_t r = 0; r < n_runs; ++r) {
          run_step0[r] = (int32_t)step_pt0.size();
          for (int32_t f = run_pt0[r]; f < run_pt0[r + 1]; f += 8) {
            step_pt0.push_back(f);
            step_run.push_back(r);
          }
        }
        run_step0[n_runs] = (int32_t)step_pt0.size();
        p->n_steps = (int32_t)step_pt0.size();
        step_run.push_back(n_runs);
        rng_step.resize(p->n_ranges + 1);
        for (int32_t g = 0; g <= p->n_ranges; ++g) rng_step[g] = run_step0[rng_run[g]];
        slot_obs.assign((size_t)n * kFzLanesPerPoint, -1);
        for (int32_t i = 0; i < n; ++i) {
          const int32_t base = slot_cam[pt_ptr[i]] / sh.group * sh.group;
          for (int32_t s = pt_ptr[i]; s < pt_ptr[i + 1]; ++s)
            slot_obs[(size_t)i * kFzLanesPerPoint + (slot_cam[s] - base)] = perm[s];
        }
        key_ptr.assign(p->n_keys + 1, 0);
        for (int32_t r = 0; r < n_runs; ++r) ++key_ptr[run_base[r] / sh.group + 1];
        for (int32_t q = 0; q < p->n_keys; ++q) key_ptr[q + 1] += key_ptr[q];
        key_run.resize(n_runs);
        std::vector<int32_t> fill(key_ptr.begin(), key_ptr.end() - 1);
        for (int32_t r = 0; r < n_runs; ++r) key_run[fill[run_base[r] / sh.group]++] = r;
      } else {
        run_pt0.clear();
        run_base.clear();
      }
    }
  }

  // ---- upload ----
  int rc = SFM_BA_OK;
  int64_t bytes = 0;
  auto up = [&](auto& h, auto** d) {
    if (rc == SFM_BA_OK) rc = upload(h, d, &bytes);
  };
  up(pt_ptr, &p->d_pt_ptr);
  up(perm, &p->d_perm);
  up(slot_cam, &p->d_slot_cam);
  up(tp_ptr, &p->d_tp_ptr);
  up(pair_entry, &p->d_pair_entry);
  up(grp_pt, &p->d_grp_pt);
  if (!cam_sorted) up(by_cam, &p->d_cam_obs);
  up(item_cam, &p->d_item_cam);
  up(item_beg, &p->d_item_beg);
  up(item_end, &p->d_item_end);
  up(cam_item_ptr, &p->d_cam_item_ptr);
  up(blk_j, &p->d_blk_j);
  up(blk_k, &p->d_blk_k);
  up(entries, &p->d_entries);
  up(pit_blk, &p->d_pit_blk);
  up(pit_beg, &p->d_pit_beg);
  up(pit_end, &p->d_pit_end);
  up(pit_part, &p->d_pit_part);
  up(mb_blk, &p->d_mb_blk);
  up(mb_part, &p->d_mb_part);
  up(mb_cnt, &p->d_mb_cnt);
  if (p->n_runs > 0) {
    up(run_pt0, &p->d_run_pt0);
    up(run_base, &p->d_run_base);
    up(slot_obs, &p->d_slot_obs);
    up(key_ptr, &p->d_key_ptr);
    up(key_run, &p->d_key_run);
    up(rng_run, &p->d_rng_run);
    up(step_pt0, &p->d_step_pt0);
    up(step_run, &p->d_step_run);
    up(rng_step, &p->d_rng_step);
  }
  if (rc != SFM_BA_OK) {
    sfm_ba_plan_destroy(p);
    return rc;
  }
  p->device_bytes = bytes;

  plan_set_workspace(p);
  *out = p;
  return SFM_BA_OK;
}

extern "C" int sfm_ba_plan_sizes(const sfm_ba_plan* p, int64_t what[8]) {
  if (!p || !what) return SFM_BA_EINVAL;
  what[0] = p->m;
  what[1] = p->n;
  what[2] = p->n_obs;
  what[3] = p->n_blocks;
  what[4] = p->n_entries;
  what[5] = p->n_pair_slots;
  what[6] = p->n_items;
  what[7] = p->device_bytes;
  return SFM_BA_OK;
}

extern "C" int sfm_ba_plan_read(const sfm_ba_plan* p, int field, void* h_dst, int64_t capacity, int64_t* bytes_out) {
  if (!p || !bytes_out) return SFM_BA_EINVAL;
  const void* src = nullptr;
  int64_t count = 0, width = sizeof(int32_t);
  switch (field) {
    case 0: src = p->d_pt_ptr; count = p->n + 1; break;
    case 1: src = p->d_perm; count = p->n_obs; break;
    case 2: src = p->d_slot_cam; count = p->n_obs; break;
    case 3: src = p->d_tp_ptr; count = p->n + 1; break;
    case 4: src = p->d_pair_entry; count = p->n_pair_slots; break;
    case 5: src = p->d_grp_pt; count = p->n_groups + 1; break;
    case 6: src = p->d_cam_obs; count = p->d_cam_obs ? p->n_obs : 0; break;
    case 7: src = p->d_item_cam; count = p->n_items; break;
    case 8: src = p->d_item_beg; count = p->n_items; break;
    case 9: src = p->d_item_end; count = p->n_items; break;
    case 10: src = p->d_cam_item_ptr; count = p->m + 1; break;
    case 11: src = p->d_blk_j; count = p->n_blocks; break;
    case 12: src = p->d_blk_k; count = p->n_blocks; break;
    case 13: src = p->d_entries; count = p->n_entries + kPairPad; width = sizeof(PairEntry); break;
    case 14: src = p->d_pit_blk; count = p->n_pair_items; break;
    case 15: src = p->d_pit_beg; count = p->n_pair_items; break;
    case 16: src = p->d_pit_end; count = p->n_pair_items; break;
    case 17: src = p->d_pit_part; count = p->n_pair_items; break;
    case 18: src = p->d_mb_blk; count = p->n_multi; break;
    case 19: src = p->d_mb_part; count = p->n_multi; break;
    case 20: src = p->d_mb_cnt; count = p->n_multi; break;
    case 21: src = p->d_run_pt0; count = p->n_runs > 0 ? p->n_runs + 1 : 0; break;
    case 22: src = p->d_run_base; count = p->n_runs; break;
    case 23: src = p->d_slot_obs; count = p->n_runs > 0 ? (int64_t)p->n * kFzLanesPerPoint : 0; break;
    case 24: src = p->d_key_ptr; count = p->n_runs > 0 ? p->n_keys + 1 : 0; break;
    case 25: src = p->d_key_run; count = p->n_runs; break;
    case 26: src = p->d_rng_run; count = p->n_runs > 0 ? p->n_ranges + 1 : 0; break;
    case 27: src = p->d_step_pt0; count = p->n_steps; break;
    case 28: src = p->d_step_run; count = p->n_runs > 0 ? p->n_steps + 1 : 0; break;
    case 29: src = p->d_rng_step; count = p->n_runs > 0 ? p->n_ranges + 1 : 0; break;
    default: return SFM_BA_EINVAL;
  }
  *bytes_out = count * width;
  if (!h_dst) return SFM_BA_OK;
  if (capacity < count * width) return SFM_BA_EINVAL;
  if (count > 0) SFMBA_HIP(hipMemcpy(h_dst, src, (size_t)(count * width), hipMemcpyDeviceToHost));
  return SFM_BA_OK;
}

extern "C" int32_t sfm_ba_plan_half_bandwidth(const sfm_ba_plan* p) {
  // two cameras j <= k share a point only if k - j <= max_cam_dist: the furthest non-zero of S from
  // its diagonal couples the first parameter of camera j with the last of camera j + max_cam_dist
  return p ? 7 * p->max_cam_dist + 6 : -1;
}

extern "C" int sfm_ba_plan_fused_info(const sfm_ba_plan* p, int32_t info[8]) {
  if (!p || !info) return SFM_BA_EINVAL;
  info[0] = p->n_runs > 0 && p->fz_enabled ? 1 : 0;
  info[1] = p->fz_slots;
  info[2] = p->fz_group;
  info[3] = p->fz_pmax;
  info[4] = p->n_runs;
  info[5] = p->n_keys;
  info[6] = p->n_steps;
  info[7] = p->n_ranges;
  return SFM_BA_OK;
}

extern "C" int sfm_ba_plan_set_fused(sfm_ba_plan* p, int enable) {
  if (!p) return SFM_BA_EINVAL;
  p->fz_enabled = enable ? 1 : 0;
  return SFM_BA_OK;
}

extern "C" int64_t sfm_ba_plan_workspace_bytes(const sfm_ba_plan* p) { return p ? p->ws_total : 0; }
