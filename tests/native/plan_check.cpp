// Host-only check of the visibility plan: builds the plan with device allocation replaced by host
// memory and verifies the invariants the kernels rely on.  Usage: plan_check m n k mode seed
//   mode 0: every point is seen by k cameras drawn uniformly; mode 1: by k consecutive cameras
//   around a centre that moves with the point index (banded); mode 2: by k consecutive cameras from
//   a first camera that ascends with the point index (the order in which an incremental
//   reconstruction creates points); mode 3: as 2 with the point ids shuffled; mode 4: tracks of
//   consecutive cameras whose length is geometric (mean 6, up to 30), ids in random order; mode 5: as
//   2, and every 50th point is seen from one of eight fixed sets of cameras that lie far apart.
//   Optional sixth argument: 1 / 0 = the plan must / must not hold runs of the fused assembly;
//   optional seventh: the least share (percent) of the points that must be near.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <tuple>

static hipError_t host_alloc(void** p, size_t sz) {
  *p = malloc(sz);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
#define hipMalloc(p, sz) host_alloc((void**)(p), (sz))
#define hipMemcpy(d, s, n, kind) (memcpy((d), (s), (n)), hipSuccess)
#define hipFree(p) (free(p), hipSuccess)
#define hipGetErrorString(e) "host"
#include "../../structure-from-motion_amd/csrc/ba_plan.cpp"

#define REQUIRE(c)                                            \
  do {                                                        \
    if (!(c)) {                                               \
      printf("FAILED line %d: %s\n", __LINE__, #c);           \
      return 1;                                               \
    }                                                         \
  } while (0)

// camera-pair structures of a plan built from the lists (cam, pt)
static int check_pairs(const sfm_ba_plan* p, const std::vector<int32_t>& cam, const std::vector<int32_t>& pt) {
  const int64_t N = (int64_t)cam.size();
  const int64_t E = p->n_entries;
  REQUIRE(E == p->n_pair_slots);
  // pair slot -> entry position is a bijection
  std::vector<char> hit((size_t)E, 0);
  for (int64_t s = 0; s < E; ++s) {
    const int32_t e = p->d_pair_entry[s];
    REQUIRE(e >= 0 && e < E && !hit[(size_t)e]);
    hit[(size_t)e] = 1;
  }
  // items tile [0, E) in order; every entry of an item belongs to the item's block
  int64_t at = 0;
  std::vector<int> part_seen((size_t)std::max(p->n_parts, 1), 0), blk_items((size_t)p->n_blocks, 0);
  for (int it = 0; it < p->n_pair_items; ++it) {
    REQUIRE(p->d_pit_beg[it] == at);
    REQUIRE(p->d_pit_end[it] > p->d_pit_beg[it] && p->d_pit_end[it] - p->d_pit_beg[it] <= sfmba::kPairChunk);
    at = p->d_pit_end[it];
    const int b = p->d_pit_blk[it];
    REQUIRE(b >= 0 && b < p->n_blocks);
    ++blk_items[(size_t)b];
    const int j = p->d_blk_j[b], kk = p->d_blk_k[b];
    REQUIRE(j <= kk);
    for (int64_t e = p->d_pit_beg[it]; e < p->d_pit_end[it]; ++e) {
      const sfmba::PairEntry en = p->d_entries[e];
      REQUIRE(en.row_a % 112 == 0 && en.row_a / 112 < N);
      const int64_t oa = en.row_a / 112;
      REQUIRE(cam[(size_t)oa] == j);
      if (j == kk) {
        REQUIRE(en.row_b == 16 * oa);
      } else {
        REQUIRE(en.row_b % 112 == 0 && en.row_b / 112 < N);
        const int64_t ob = en.row_b / 112;
        REQUIRE(cam[(size_t)ob] == kk && pt[(size_t)ob] == pt[(size_t)oa]);
      }
    }
    const int part = p->d_pit_part[it];
    if (part >= 0) {
      REQUIRE(part < p->n_parts && !part_seen[(size_t)part]);
      part_seen[(size_t)part] = 1 + b;
    }
  }
  REQUIRE(at == E);
  // the records kept behind the last entry are valid rows
  for (int64_t e = E; e < E + sfmba::kPairPad; ++e) REQUIRE(p->d_entries[e].row_a == 0 && p->d_entries[e].row_b == 0);
  // blocks with partial results (split blocks; every block of a far plan): contiguous, all theirs, complete
  int multi = 0;
  for (int b = 0; b < p->n_blocks; ++b) multi += (blk_items[(size_t)b] > 1 || p->force_parts);
  REQUIRE(multi == p->n_multi);
  for (int q = 0; q < p->n_multi; ++q) {
    const int b = p->d_mb_blk[q];
    REQUIRE(p->d_mb_cnt[q] == blk_items[(size_t)b] && (p->d_mb_cnt[q] > 1 || p->force_parts));
    REQUIRE(!p->force_parts || b == q);
    for (int i = 0; i < p->d_mb_cnt[q]; ++i) REQUIRE(part_seen[(size_t)(p->d_mb_part[q] + i)] == 1 + b);
  }
  for (int it = 0; it < p->n_pair_items; ++it)
    REQUIRE((p->d_pit_part[it] < 0) == (blk_items[(size_t)p->d_pit_blk[it]] == 1 && !p->force_parts));
  // every block has entries
  for (int b = 0; b < p->n_blocks; ++b) REQUIRE(blk_items[(size_t)b] >= 1);
  return 0;
}

int main(int argc, char** argv) {
  const int m = atoi(argv[1]), n = atoi(argv[2]), k = atoi(argv[3]), mode = atoi(argv[4]), seed = atoi(argv[5]);
  std::mt19937 rng(seed);
  std::vector<int32_t> cam, pt;
  std::vector<int32_t> label((size_t)n);
  for (int i = 0; i < n; ++i) label[(size_t)i] = i;
  if (mode == 3 || mode == 4)
    for (int i = n - 1; i > 0; --i) std::swap(label[(size_t)i], label[(size_t)(rng() % (unsigned)(i + 1))]);
  for (int i = 0; i < n; ++i) {
    std::set<int> cs;
    const int kk = 2 + (int)(rng() % (unsigned)(k - 1));  // 2..k views
    if (mode == 0) {
      while ((int)cs.size() < std::min(kk, m)) cs.insert((int)(rng() % (unsigned)m));
    } else if (mode == 1) {
      const int c0 = (int)((int64_t)i * m / n);
      for (int q = 0; q < kk; ++q) cs.insert(std::min(m - 1, std::max(0, c0 - kk / 2 + q)));
    } else if (mode == 4) {
      int len = 2;
      while (len < 30 && rng() % 6 != 0) ++len;
      const int c0 = (int)(rng() % (unsigned)m);
      for (int q = 0; q < len; ++q) cs.insert(std::min(m - 1, c0 + q));
    } else if (mode == 5 && i % 50 == 7) {
      const int f = (int)(rng() % 8u);
      for (int q = 0; q < std::min(kk, 3); ++q) cs.insert((f + q) % m);
      for (int q = 0; q < std::min(kk, 3); ++q) cs.insert((m / 2 + 3 * f + q) % m);
    } else {
      const int c0 = (int)((int64_t)i * m / n);
      for (int q = 0; q < kk; ++q) cs.insert(std::min(m - 1, c0 + q));
    }
    for (int c : cs) {
      cam.push_back(c);
      pt.push_back(label[(size_t)i]);
    }
  }
  // caller order: shuffled
  const int64_t N = (int64_t)cam.size();
  for (int64_t i = N - 1; i > 0; --i) {
    const int64_t j = (int64_t)(rng() % (uint64_t)(i + 1));
    std::swap(cam[i], cam[j]);
    std::swap(pt[i], pt[j]);
  }
  sfm_ba_plan* p = nullptr;
  REQUIRE(sfm_ba_plan_create(cam.data(), pt.data(), N, m, n, 0, &p) == SFM_BA_OK);
  if (check_pairs(p, cam, pt)) return 1;
  // near points of the fused assembly, per class: their positions are tiled by runs, every observation of
  // a near point sits at the slot of its camera in its run's window, every other slot is empty
  int64_t near_obs = 0;
  int n_near_all = 0;
  if (p->n_runs > 0) {
    std::vector<int> cls_of((size_t)n, -1);
    std::vector<int64_t> seen_obs((size_t)N, 0);
    std::set<std::tuple<int, int, int>> need;  // (block, class, window) the finishing kernel must find
    std::vector<int> blk_of((size_t)m * m, -1);
    for (int b = 0; b < p->n_blocks; ++b) blk_of[(size_t)p->d_blk_j[b] * m + p->d_blk_k[b]] = b;
    int runs_all = 0;
    for (int c = 0; c < sfmba::kFzClasses; ++c) {
      const sfmba::FzClass& f = p->fz[c];
      if (f.n_runs == 0) {
        REQUIRE(f.n_near == 0 && f.n_keys == 0);
        continue;
      }
      runs_all += f.n_runs;
      n_near_all += f.n_near;
      const int L = f.lanes, pps = 64 / L;
      REQUIRE(c == 0 ? (f.slots >= 4 && f.slots <= 8 && L == 8) : (f.slots == 16 && L == 16));
      REQUIRE(f.pmax >= 8 && f.n_near > 0 && f.n_near <= n);
      REQUIRE(f.d_run_pt0[0] == 0 && f.d_run_pt0[f.n_runs] == f.n_near);
      for (int q = 0; q < f.n_near; ++q) {
        const int i = f.d_pt[q];
        REQUIRE(i >= 0 && i < n && cls_of[(size_t)i] < 0);
        cls_of[(size_t)i] = c;
      }
      int steps = 0;
      for (int r = 0; r < f.n_runs; ++r) {
        const int key = f.d_run_key[r];
        REQUIRE(key >= 0 && key < f.n_keys);
        REQUIRE(f.d_run_pt0[r + 1] > f.d_run_pt0[r] && f.d_run_pt0[r + 1] - f.d_run_pt0[r] <= f.pmax);
        REQUIRE(r == 0 || f.d_run_key[r - 1] == key || f.d_run_key[r - 1] == key - 1);
        steps += (f.d_run_pt0[r + 1] - f.d_run_pt0[r] + pps - 1) / pps;
        const int32_t* kc = f.d_key_cam + L * key;
        for (int sl = 0; sl < L; ++sl) {
          REQUIRE(kc[sl] >= -1 && kc[sl] < m && (sl < f.slots || kc[sl] < 0));
          const int rc = f.d_run_cam[L * r + sl];
          REQUIRE(rc >= 0 && rc < m && (kc[sl] < 0 || rc == kc[sl]));
        }
        for (int q = f.d_run_pt0[r]; q < f.d_run_pt0[r + 1]; ++q) {
          int cams_q[16], nq = 0;
          for (int sl = 0; sl < L; ++sl) {
            const int32_t o = f.d_slot_obs[(size_t)q * L + sl];
            if (o < 0) continue;
            REQUIRE(sl < f.slots && o < N && pt[(size_t)o] == f.d_pt[q] && cam[(size_t)o] == kc[sl]);
            ++seen_obs[(size_t)o];
            cams_q[nq++] = kc[sl];
          }
          REQUIRE(nq >= 1);
          for (int a = 0; a < nq; ++a)
            for (int b = a; b < nq; ++b) {
              const int lo = std::min(cams_q[a], cams_q[b]), hi = std::max(cams_q[a], cams_q[b]);
              REQUIRE(blk_of[(size_t)lo * m + hi] >= 0);
              need.insert({blk_of[(size_t)lo * m + hi], c, key});
            }
        }
      }
      REQUIRE(steps == f.n_steps);
      // ranges of pmax positions own whole runs
      REQUIRE(f.n_ranges == (f.n_near + f.pmax - 1) / f.pmax);
      REQUIRE(f.d_rng_run[0] == 0 && f.d_rng_run[f.n_ranges] == f.n_runs);
      int max_range_runs = 0;
      for (int g = 0; g < f.n_ranges; ++g) {
        REQUIRE(f.d_rng_run[g + 1] > f.d_rng_run[g]);
        REQUIRE(f.d_run_pt0[f.d_rng_run[g]] == g * f.pmax);
        max_range_runs = std::max(max_range_runs, f.d_rng_run[g + 1] - f.d_rng_run[g]);
      }
      REQUIRE(max_range_runs == f.max_range_runs);
      // the runs of a window are consecutive; windows differ from their neighbours and hold enough points
      REQUIRE(f.d_key_run[0] == 0 && f.d_key_run[f.n_keys] == f.n_runs);
      for (int q = 0; q < f.n_keys; ++q) {
        REQUIRE(f.d_key_run[q + 1] > f.d_key_run[q]);
        for (int r = f.d_key_run[q]; r < f.d_key_run[q + 1]; ++r) REQUIRE(f.d_run_key[r] == q);
        REQUIRE(f.d_run_pt0[f.d_key_run[q + 1]] - f.d_run_pt0[f.d_key_run[q]] >= sfmba::fz_min_window_points(c));
        if (q > 0) REQUIRE(memcmp(f.d_key_cam + L * q, f.d_key_cam + L * (q - 1), 4 * L) != 0);
      }
    }
    REQUIRE(runs_all == p->n_runs);
    std::vector<int> pos_of(cls_of);  // >= 0: near
    for (int64_t o = 0; o < N; ++o) {
      REQUIRE(seen_obs[(size_t)o] == (pos_of[(size_t)pt[(size_t)o]] >= 0 ? 1 : 0));
      near_obs += seen_obs[(size_t)o];
    }
    // entries of the finishing kernel: per block class 0 before class 1, ascending windows, the right
    // cameras at the slots, and every (block, class, window) that holds a point's pair of cameras is there
    REQUIRE(p->d_fb_ptr[0] == 0 && p->d_fb_ptr[p->n_blocks] == p->n_fb);
    for (int b = 0; b < p->n_blocks; ++b) {
      for (int e = p->d_fb_ptr[b]; e < p->d_fb_ptr[b + 1]; ++e) {
        const int ent = p->d_fb_ent[e], c = (ent >> 30) & 1, key = (ent >> 8) & 0x3FFFFF, sj = (ent >> 4) & 15, sk = ent & 15;
        const sfmba::FzClass& f = p->fz[c];
        REQUIRE(key >= 0 && key < f.n_keys && sj < f.lanes && sk < f.lanes);
        REQUIRE(f.d_key_cam[f.lanes * key + sj] == p->d_blk_j[b] && f.d_key_cam[f.lanes * key + sk] == p->d_blk_k[b]);
        REQUIRE(e == p->d_fb_ptr[b] || (p->d_fb_ent[e - 1] >> 8) < (ent >> 8));
        need.erase({b, c, key});
      }
    }
    REQUIRE(need.empty());
    // far points: the rest, ascending, with their observations in the caller's relative order
    REQUIRE(p->n_far == n - n_near_all);
    for (int q = 0; q < p->n_far; ++q) {
      REQUIRE(pos_of[(size_t)p->d_far_pt[q]] < 0);
      REQUIRE(q == 0 || p->d_far_pt[q - 1] < p->d_far_pt[q]);
    }
    REQUIRE(p->n_far_obs == N - near_obs);
    std::vector<int32_t> fcam, fpt;
    for (int64_t i = 0; i < p->n_far_obs; ++i) {
      const int32_t o = p->d_far_obs[i];
      REQUIRE(o >= 0 && o < N && (i == 0 || p->d_far_obs[i - 1] < o));
      REQUIRE(p->d_far_cam[i] == cam[(size_t)o] && p->d_far_pt[p->d_far_lpt[i]] == pt[(size_t)o]);
      REQUIRE(p->d_far_gpt[i] == pt[(size_t)o]);
      fcam.push_back(p->d_far_cam[i]);
      fpt.push_back(p->d_far_lpt[i]);
    }
    REQUIRE((p->far != nullptr) == (p->n_far_obs > 0));
    std::vector<int> far_seen((size_t)p->n_blocks, 0);
    if (p->far) {
      const sfm_ba_plan* f = p->far;
      REQUIRE(f->m == m && f->n == p->n_far && f->n_obs == p->n_far_obs && f->force_parts == 1 && f->n_runs == 0);
      if (check_pairs(f, fcam, fpt)) return 1;
      for (int q = 0; q < f->n_blocks; ++q) {
        const int b = blk_of[(size_t)f->d_blk_j[q] * m + f->d_blk_k[q]];
        REQUIRE(b >= 0 && p->d_fb_far_part[b] == f->d_mb_part[q] && p->d_fb_far_cnt[b] == f->d_mb_cnt[q]);
        far_seen[(size_t)b] = 1;
      }
    }
    for (int b = 0; b < p->n_blocks; ++b) REQUIRE(far_seen[(size_t)b] || p->d_fb_far_cnt[b] == 0);
    REQUIRE(p->ws_total >= p->ws_far + (p->far ? p->far->ws_total : 0));
  }
  if (argc > 6) REQUIRE((p->n_runs > 0) == (atoi(argv[6]) != 0));
  if (argc > 7) REQUIRE(100.0 * n_near_all >= (double)atoi(argv[7]) * n);
  printf("ok runs=%d windows=%d+%d slots=%d+%d near=%d+%d (%.1f%% of the observations) far=%d ", p->n_runs, p->fz[0].n_keys,
         p->fz[1].n_keys, p->fz[0].slots, p->fz[1].slots, p->fz[0].n_near, p->fz[1].n_near, 100.0 * (double)near_obs / (double)N,
         p->n_far);
  printf("m=%d n=%d N=%lld blocks=%d entries=%lld items=%d parts=%d half_bandwidth=%d\n", m, n, (long long)N,
         p->n_blocks, (long long)p->n_entries, p->n_pair_items, p->n_parts, sfm_ba_plan_half_bandwidth(p));
  sfm_ba_plan_destroy(p);
  return 0;
}
