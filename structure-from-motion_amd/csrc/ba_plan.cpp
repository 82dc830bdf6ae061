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
  p->ws_q = off;  // per observation: its squared residual and a zero (addressed like g, 16 bytes per observation)
  off = align256(off + p->n_obs * 2 * (int64_t)sizeof(double));
  p->ws_campart = off;
  off = align256(off + (int64_t)std::max(p->n_items, 1) * SFM_BA_CAM_RED * (int64_t)sizeof(double));
  p->ws_spart = off;
  off = align256(off + (int64_t)std::max(p->n_parts, 1) * kPairPartStride * (int64_t)sizeof(double));
  // fused point-major assembly: per class one partial result per run, one per window
  for (int c = 0; c < kFzClasses; ++c) {
    FzClass& f = p->fz[c];
    f.ws_part = off;
    off = align256(off + (int64_t)f.n_runs * (f.slots ? fz_part_stride(f.slots) : 0) * (int64_t)sizeof(double));
    f.ws_key = off;
    off = align256(off + (int64_t)f.n_keys * (f.slots ? fz_part_stride(f.slots) : 0) * (int64_t)sizeof(double));
  }
  p->ws_prec = off;  // per point: position and V^-1, for the camera-pair kernel that forms its rows itself
  off = align256(off + (int64_t)std::max(p->n, 1) * kPointRec * (int64_t)sizeof(double));
  p->ws_fzero = off;  // a row of zeros
  off = align256(off + 256);
  // far points: their rows of A, observed and predicted pixels in the far plan's observation order,
  // then that plan's own workspace
  p->ws_far_A = off;
  off = align256(off + p->n_far_obs * 14 * (int64_t)sizeof(double));
  p->ws_far_xt = off;
  off = align256(off + p->n_far_obs * 2 * (int64_t)sizeof(double));
  p->ws_far_xp = off;
  off = align256(off + p->n_far_obs * 2 * (int64_t)sizeof(double));
  p->ws_far = off;
  off = align256(off + (p->far ? p->far->ws_total : 0));
  p->ws_total = off;
}

}  // namespace sfmba

using namespace sfmba;

// defined in ba_kernels.hip; absent from host-only builds of this file (tests/native/plan_check.cpp)
extern "C" int sfmba_cost_scratch_warm(void) __attribute__((weak));

extern "C" const char* sfm_ba_last_hip_error(void) { return sfmba::last_hip_error(); }

extern "C" void sfm_ba_plan_destroy(sfm_ba_plan* p) {
  if (!p) return;
  void* ptrs[] = {p->d_pt_ptr,   p->d_perm,     p->d_slot_cam, p->d_tp_ptr,  p->d_grp_pt,  p->d_cam_obs, p->d_pair_entry,
                  p->d_item_cam, p->d_item_beg, p->d_item_end, p->d_cam_item_ptr, p->d_blk_j, p->d_blk_k,
                  p->d_entries,  p->d_ent_pt,   p->d_pit_blk,  p->d_pit_beg,  p->d_pit_end, p->d_pit_part, p->d_mb_blk,
                  p->d_mb_part,  p->d_mb_cnt,   p->d_fb_ptr,   p->d_fb_ent,
                  p->d_fb_far_part, p->d_fb_far_cnt, p->d_far_pt, p->d_far_obs, p->d_far_cam, p->d_far_lpt, p->d_far_gpt};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  for (int c = 0; c < kFzClasses; ++c) {
    const FzClass& f = p->fz[c];
    void* fp[] = {f.d_pt, f.d_slot_obs, f.d_run_pt0, f.d_run_key, f.d_run_cam, f.d_rng_run, f.d_key_cam, f.d_key_run};
    for (void* q : fp)
      if (q) (void)hipFree(q);
  }
  sfm_ba_plan_destroy(p->far);
  delete p;
}

namespace {
enum PlanFlags { kPlanNoFused = 1, kPlanForceParts = 2 };
// what the builder of a plan with far points needs to know of the far plan's blocks
struct HostBlocks {
  std::vector<int32_t> blk_j, blk_k, mb_part, mb_cnt;
};

int build_host(const int32_t* cam_idx, const int32_t* pt_idx, int64_t N, int32_t m_hint, int32_t n_hint, int check,
               int flags, HostBlocks* blocks_out, sfm_ba_plan** out) {
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
  bool nonunique = false;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t k = pt_ptr[i + 1] - pt_ptr[i];
    if (k <= kGroupSlots) kshort = std::max(kshort, k);
    for (int32_t s = pt_ptr[i] + 1; s < pt_ptr[i + 1]; ++s)
      if (slot_cam[s] == slot_cam[s - 1]) nonunique = true;
  }
  if (check && nonunique) return SFM_BA_ENONUNIQUE;
  if (N * 112 >= (int64_t(1) << 32)) return SFM_BA_ETOOBIG;  // 32-bit byte offsets into A

  sfm_ba_plan* p = new (std::nothrow) sfm_ba_plan();
  if (!p) return SFM_BA_EINVAL;
  p->m = m;
  p->n = n;
  p->n_obs = N;
  p->force_parts = (flags & kPlanForceParts) ? 1 : 0;

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
      // 32-bit byte offsets into T and the per-observation array behind it, counted from the start of g
      // that precedes them in the workspace
      if (N * 32 + 512 + (pairs + kPairPad) * 32 >= (int64_t(1) << 32)) {
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
    const bool parts = chunks > 1 || p->force_parts;  // (a far plan's blocks are all finished by the plan above it)
    if (parts) {
      mb_blk.push_back(b);
      mb_part.push_back(n_parts);
      mb_cnt.push_back(chunks);
    }
    for (int32_t c = 0; c < chunks; ++c) {
      pit_blk.push_back(b);
      pit_beg.push_back(beg + c * kPairChunk);
      pit_end.push_back(std::min(beg + (c + 1) * kPairChunk, end));
      pit_part.push_back(parts ? n_parts++ : -1);
    }
  }
  p->n_pair_items = (int32_t)pit_blk.size();
  p->n_parts = n_parts;
  p->n_multi = (int32_t)mb_blk.size();

  if (blocks_out) {
    blocks_out->blk_j = blk_j;
    blocks_out->blk_k = blk_k;
    blocks_out->mb_part = mb_part;
    blocks_out->mb_cnt = mb_cnt;
  }

  // ---- near points of the fused point-major assembly, far points (same rules as the device builder) ----
  // per class: point ids by position, slot table, runs, windows
  struct HostClass {
    std::vector<int32_t> pt, slot_obs, run_pt0, run_key, run_cam, rng_run, key_cam, key_run;
  } hc[kFzClasses];
  std::vector<int32_t> fb_ptr, fb_ent, fb_far_part, fb_far_cnt, far_pt, far_obs, far_cam, far_lpt, far_gpt;
  // (a list that names a (camera, point) pair twice - accepted only without checks - stays on the
  // camera-pair path, which adds both; the slot table of the fused path has room for one)
  if (!(flags & kPlanNoFused) && !nonunique) {
    int64_t obs_by_k[9] = {0}, obs_by_span[9] = {0};
    for (int32_t i = 0; i < n; ++i) {
      const int32_t k = pt_ptr[i + 1] - pt_ptr[i];
      if (k < 1 || k > 8) continue;
      obs_by_k[k] += k;
      const int32_t span = slot_cam[pt_ptr[i + 1] - 1] - slot_cam[pt_ptr[i]] + 1;
      if (span <= 8) obs_by_span[span] += k;
    }
    std::vector<int32_t> cls_of((size_t)n, -1);  // class in which the point is near
    for (int c = 0; c < kFzClasses; ++c) {
      const int slots = c == 0 ? fz_choose_slots(obs_by_k, obs_by_span) : 16;
      if (slots <= 0) continue;
      const int lanes = fz_lanes(slots);
      // candidates: the points no earlier class took, sorted by the key of their window (ties: ascending id)
      std::vector<int32_t> desc((size_t)n * lanes, -1), cand, near;
      std::vector<uint64_t> key((size_t)n, 0);
      for (int32_t i = 0; i < n; ++i)
        if (cls_of[i] < 0 && fz_point_window(slot_cam.data() + pt_ptr[i], pt_ptr[i + 1] - pt_ptr[i], slots, lanes, m,
                                             &desc[(size_t)lanes * i], &key[i]))
          cand.push_back(i);
      std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
      auto same_window = [&](int32_t a, int32_t b) {
        return memcmp(&desc[(size_t)lanes * a], &desc[(size_t)lanes * b], lanes * sizeof(int32_t)) == 0;
      };
      // windows = maximal stretches of equal camera lists in the sorted order; short ones are given up
      for (size_t q = 0; q < cand.size();) {
        size_t r = q + 1;
        while (r < cand.size() && same_window(cand[q], cand[r])) ++r;
        if (r - q >= (size_t)fz_min_window_points(c)) near.insert(near.end(), cand.begin() + q, cand.begin() + r);
        q = r;
      }
      const int32_t n_near = (int32_t)near.size();
      const int32_t pmax = fz_pmax_for(n_near);
      HostClass& h = hc[c];
      for (int32_t q = 0; q < n_near; ++q) {
        const bool new_window = q == 0 || !same_window(near[q], near[q - 1]);
        if (new_window) {
          h.key_run.push_back((int32_t)h.run_pt0.size());
          h.key_cam.insert(h.key_cam.end(), &desc[(size_t)lanes * near[q]], &desc[(size_t)lanes * near[q]] + lanes);
        }
        if (new_window || q % pmax == 0) {
          h.run_pt0.push_back(q);
          h.run_key.push_back((int32_t)h.key_run.size() - 1);
          const int32_t* d = &desc[(size_t)lanes * near[q]];
          for (int s = 0; s < lanes; ++s) h.run_cam.push_back(d[s] >= 0 ? d[s] : d[0]);
        }
      }
      const int32_t n_runs = (int32_t)h.run_pt0.size();
      if (!fz_worthwhile(c, n, n_near, n_runs)) {
        h = HostClass();
        continue;
      }
      FzClass& f = p->fz[c];
      f.slots = slots;
      f.lanes = lanes;
      f.pmax = pmax;
      f.n_near = n_near;
      f.n_runs = n_runs;
      f.n_keys = (int32_t)h.key_run.size();
      p->n_runs += n_runs;
      h.key_run.push_back(n_runs);
      h.run_pt0.push_back(n_near);
      f.n_ranges = (n_near + pmax - 1) / pmax;
      h.rng_run.assign(f.n_ranges + 1, n_runs);
      for (int32_t r = n_runs - 1; r >= 0; --r)
        if (h.run_pt0[r] % pmax == 0) h.rng_run[h.run_pt0[r] / pmax] = r;
      for (int32_t g = 0; g < f.n_ranges; ++g) f.max_range_runs = std::max(f.max_range_runs, h.rng_run[g + 1] - h.rng_run[g]);
      const int pps = 64 / lanes;  // points per step
      for (int32_t r = 0; r < n_runs; ++r) f.n_steps += (h.run_pt0[r + 1] - h.run_pt0[r] + pps - 1) / pps;
      h.pt = near;
      h.slot_obs.assign((size_t)n_near * lanes, -1);
      for (int32_t q = 0; q < n_near; ++q) {
        const int32_t i = near[q];
        cls_of[i] = c;
        for (int32_t s = pt_ptr[i]; s < pt_ptr[i + 1]; ++s)
          h.slot_obs[(size_t)q * lanes + fz_slot_of(&desc[(size_t)lanes * i], lanes, slot_cam[s])] = perm[s];
      }
    }
    if (p->n_runs > 0) {
      // far points (those no class took, among them the points without observations that plans built with
      // size hints allow) and their observations
      std::vector<int32_t> far_num((size_t)n, -1);
      for (int32_t i = 0; i < n; ++i)
        if (cls_of[i] < 0) {
          far_num[i] = (int32_t)far_pt.size();
          far_pt.push_back(i);
        }
      for (int64_t o = 0; o < N; ++o)
        if (far_num[pt_idx[o]] >= 0) {
          far_obs.push_back((int32_t)o);
          far_cam.push_back(cam_idx[o]);
          far_lpt.push_back(far_num[pt_idx[o]]);
          far_gpt.push_back(pt_idx[o]);
        }
      p->n_far = (int32_t)far_pt.size();
      p->n_far_obs = (int64_t)far_obs.size();
      HostBlocks fb;
      if (p->n_far_obs > 0) {
        const int rc_far = build_host(far_cam.data(), far_lpt.data(), p->n_far_obs, m, p->n_far, 0,
                                      kPlanNoFused | kPlanForceParts, &fb, &p->far);
        if (rc_far != SFM_BA_OK) {
          sfm_ba_plan_destroy(p);
          return rc_far;
        }
      }
      // what the finishing kernel adds up for every block: the windows that hold both cameras (class 0 first,
      // ascending), then the far plan's partial results
      std::vector<int32_t> blk_of((size_t)m * m, -1);
      for (int32_t b = 0; b < p->n_blocks; ++b) blk_of[(size_t)blk_j[b] * m + blk_k[b]] = b;
      fb_ptr.assign((size_t)p->n_blocks + 1, 0);
      for (int pass = 0; pass < 2; ++pass) {
        std::vector<int32_t> fill(fb_ptr.begin(), fb_ptr.end() - 1);
        for (int c = 0; c < kFzClasses; ++c) {
          const int lanes = p->fz[c].lanes;
          for (int32_t q = 0; q < p->fz[c].n_keys; ++q)
            for (int a = 0; a < lanes; ++a) {
              const int32_t ca = hc[c].key_cam[(size_t)lanes * q + a];
              if (ca < 0) continue;
              for (int b = a; b < lanes; ++b) {
                const int32_t cb = hc[c].key_cam[(size_t)lanes * q + b];
                if (cb < 0) continue;
                const int32_t id = blk_of[(size_t)ca * m + cb];
                if (id < 0) continue;
                if (pass == 0)
                  ++fb_ptr[(size_t)id + 1];
                else
                  fb_ent[(size_t)fill[id]++] = fz_fb_pack(c, q, a, b);
              }
            }
        }
        if (pass == 0) {
          for (int32_t b = 0; b < p->n_blocks; ++b) fb_ptr[(size_t)b + 1] += fb_ptr[b];
          fb_ent.resize((size_t)fb_ptr[p->n_blocks]);
        }
      }
      p->n_fb = (int32_t)fb_ent.size();
      fb_far_part.assign((size_t)p->n_blocks, -1);
      fb_far_cnt.assign((size_t)p->n_blocks, 0);
      for (size_t q = 0; q < fb.blk_j.size(); ++q) {
        const int32_t id = blk_of[(size_t)fb.blk_j[q] * m + fb.blk_k[q]];  // a far block is a block of the whole
        fb_far_part[id] = fb.mb_part[q];
        fb_far_cnt[id] = fb.mb_cnt[q];
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
  {
    // point of every entry (both of its observations belong to it)
    std::vector<int32_t> ent_pt((size_t)p->n_entries);
    for (int64_t e = 0; e < p->n_entries; ++e) ent_pt[(size_t)e] = pt_idx[entries[(size_t)e].row_a / 112];
    up(ent_pt, &p->d_ent_pt);
  }
  up(pit_blk, &p->d_pit_blk);
  up(pit_beg, &p->d_pit_beg);
  up(pit_end, &p->d_pit_end);
  up(pit_part, &p->d_pit_part);
  up(mb_blk, &p->d_mb_blk);
  up(mb_part, &p->d_mb_part);
  up(mb_cnt, &p->d_mb_cnt);
  if (p->n_runs > 0) {
    for (int c = 0; c < kFzClasses; ++c) {
      if (p->fz[c].n_runs <= 0) continue;
      up(hc[c].pt, &p->fz[c].d_pt);
      up(hc[c].slot_obs, &p->fz[c].d_slot_obs);
      up(hc[c].run_pt0, &p->fz[c].d_run_pt0);
      up(hc[c].run_key, &p->fz[c].d_run_key);
      up(hc[c].run_cam, &p->fz[c].d_run_cam);
      up(hc[c].rng_run, &p->fz[c].d_rng_run);
      up(hc[c].key_cam, &p->fz[c].d_key_cam);
      up(hc[c].key_run, &p->fz[c].d_key_run);
    }
    up(fb_ptr, &p->d_fb_ptr);
    up(fb_ent, &p->d_fb_ent);
    up(fb_far_part, &p->d_fb_far_part);
    up(fb_far_cnt, &p->d_fb_far_cnt);
    up(far_pt, &p->d_far_pt);
    up(far_obs, &p->d_far_obs);
    up(far_cam, &p->d_far_cam);
    up(far_lpt, &p->d_far_lpt);
    up(far_gpt, &p->d_far_gpt);
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
}  // namespace

extern "C" int sfm_ba_plan_create(const int32_t* cam_idx, const int32_t* pt_idx, int64_t N, int32_t m_hint,
                                  int32_t n_hint, int check, sfm_ba_plan** out) {
  const int rc = build_host(cam_idx, pt_idx, N, m_hint, n_hint, check, 0, nullptr, out);
  // (the array sfm_ba_cost_noalloc uses: its address on this device is looked up now, not under a capture)
  if (rc == SFM_BA_OK && sfmba_cost_scratch_warm) (void)sfmba_cost_scratch_warm();
  return rc;
}

extern "C" const sfm_ba_plan* sfm_ba_plan_far(const sfm_ba_plan* p) { return p ? p->far : nullptr; }

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
  // fields 21..28 belong to class 0 of the near points, 121..128 to class 1
  const int cls = field >= 100 ? 1 : 0;
  const FzClass& f = p->fz[cls];
  const bool on = f.n_runs > 0;
  if (field >= 100) field -= 100;
  if (cls == 1 && (field < 21 || field > 28)) return SFM_BA_EINVAL;
  switch (field) {
    case 21: src = f.d_run_pt0; count = on ? f.n_runs + 1 : 0; break;
    case 22: src = f.d_run_key; count = f.n_runs; break;
    case 23: src = f.d_slot_obs; count = (int64_t)f.n_near * f.lanes; break;
    case 24: src = f.d_key_cam; count = (int64_t)f.n_keys * f.lanes; break;
    case 25: src = f.d_key_run; count = on ? f.n_keys + 1 : 0; break;
    case 26: src = f.d_rng_run; count = on ? f.n_ranges + 1 : 0; break;
    case 27: src = f.d_pt; count = f.n_near; break;
    case 28: src = f.d_run_cam; count = (int64_t)f.n_runs * f.lanes; break;
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
    case 29: src = p->d_fb_ptr; count = p->n_runs > 0 ? p->n_blocks + 1 : 0; break;
    case 30: src = p->d_fb_ent; count = p->n_fb; break;
    case 31: src = p->d_fb_far_part; count = p->n_runs > 0 ? p->n_blocks : 0; break;
    case 32: src = p->d_fb_far_cnt; count = p->n_runs > 0 ? p->n_blocks : 0; break;
    case 33: src = p->d_far_pt; count = p->n_far; break;
    case 34: src = p->d_far_obs; count = p->n_far_obs; break;
    case 35: src = p->d_far_cam; count = p->n_far_obs; break;
    case 36: src = p->d_far_lpt; count = p->n_far_obs; break;
    case 37: src = p->d_far_gpt; count = p->n_far_obs; break;
    case 38: src = p->d_ent_pt; count = p->n_entries; break;
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

extern "C" int sfm_ba_plan_fused_info(const sfm_ba_plan* p, int32_t info[16]) {
  if (!p || !info) return SFM_BA_EINVAL;
  for (int c = 0; c < kFzClasses; ++c) {
    const FzClass& f = p->fz[c];
    int32_t* o = info + 8 * c;
    o[0] = f.n_runs > 0 && p->fz_enabled ? 1 : 0;
    o[1] = f.slots;
    o[2] = f.n_near;
    o[3] = f.pmax;
    o[4] = f.n_runs;
    o[5] = f.n_keys;
    o[6] = f.n_steps;
    o[7] = f.n_ranges;
  }
  return SFM_BA_OK;
}

extern "C" int sfm_ba_plan_set_fused(sfm_ba_plan* p, int enable) {
  if (!p) return SFM_BA_EINVAL;
  p->fz_enabled = enable ? 1 : 0;
  return SFM_BA_OK;
}

extern "C" int64_t sfm_ba_plan_workspace_bytes(const sfm_ba_plan* p) { return p ? p->ws_total : 0; }
