// Host-only check of the visibility plan: builds the plan with device allocation replaced by host
// memory and verifies the invariants the kernels rely on.  Usage: plan_check m n k mode seed
//   mode 0: every point is seen by k cameras drawn uniformly; mode 1: by k consecutive cameras
//   around a centre that moves with the point index (banded); mode 2: by k consecutive cameras from
//   a first camera that ascends with the point index (the order in which an incremental
//   reconstruction creates points).  Optional sixth argument: 1 / 0 =
//   the plan must / must not hold runs of the fused assembly.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>

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

int main(int argc, char** argv) {
  const int m = atoi(argv[1]), n = atoi(argv[2]), k = atoi(argv[3]), mode = atoi(argv[4]), seed = atoi(argv[5]);
  std::mt19937 rng(seed);
  std::vector<int32_t> cam, pt;
  for (int i = 0; i < n; ++i) {
    std::set<int> cs;
    const int kk = 2 + (int)(rng() % (unsigned)(k - 1));  // 2..k views
    if (mode == 0) {
      while ((int)cs.size() < std::min(kk, m)) cs.insert((int)(rng() % (unsigned)m));
    } else if (mode == 1) {
      const int c0 = (int)((int64_t)i * m / n);
      for (int q = 0; q < kk; ++q) cs.insert(std::min(m - 1, std::max(0, c0 - kk / 2 + q)));
    } else {
      const int c0 = (int)((int64_t)i * m / n);
      for (int q = 0; q < kk; ++q) cs.insert(std::min(m - 1, c0 + q));
    }
    for (int c : cs) {
      cam.push_back(c);
      pt.push_back(i);
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
  // split blocks: their partials are contiguous, all theirs, and complete
  int multi = 0;
  for (int b = 0; b < p->n_blocks; ++b) multi += blk_items[(size_t)b] > 1;
  REQUIRE(multi == p->n_multi);
  for (int q = 0; q < p->n_multi; ++q) {
    const int b = p->d_mb_blk[q];
    REQUIRE(p->d_mb_cnt[q] == blk_items[(size_t)b] && p->d_mb_cnt[q] > 1);
    for (int i = 0; i < p->d_mb_cnt[q]; ++i) REQUIRE(part_seen[(size_t)(p->d_mb_part[q] + i)] == 1 + b);
  }
  for (int it = 0; it < p->n_pair_items; ++it)
    REQUIRE((p->d_pit_part[it] < 0) == (blk_items[(size_t)p->d_pit_blk[it]] == 1));
  // every block has entries
  for (int b = 0; b < p->n_blocks; ++b) REQUIRE(blk_items[(size_t)b] >= 1);
  // runs of the fused assembly: they tile the points, every observation sits at the slot of its
  // camera in its run's window, every other slot is empty, and the runs are grouped by window
  if (p->n_runs > 0) {
    REQUIRE(p->fz_slots >= 4 && p->fz_slots <= 8);
    REQUIRE(p->fz_group >= 1 && p->fz_pmax >= 8);
    REQUIRE(p->d_run_pt0[0] == 0 && p->d_run_pt0[p->n_runs] == n);
    std::vector<int64_t> seen_obs((size_t)N, 0);
    for (int r = 0; r < p->n_runs; ++r) {
      const int b = p->d_run_base[r];
      REQUIRE(p->d_run_pt0[r + 1] > p->d_run_pt0[r] && p->d_run_pt0[r + 1] - p->d_run_pt0[r] <= p->fz_pmax);
      REQUIRE(b >= 0 && b % p->fz_group == 0);
      for (int i = p->d_run_pt0[r]; i < p->d_run_pt0[r + 1]; ++i)
        for (int sl = 0; sl < 8; ++sl) {
          const int32_t o = p->d_slot_obs[(size_t)i * 8 + sl];
          if (o < 0) continue;
          REQUIRE(sl < p->fz_slots && o < N && pt[(size_t)o] == i && cam[(size_t)o] == b + sl);
          ++seen_obs[(size_t)o];
        }
    }
    for (int64_t o = 0; o < N; ++o) REQUIRE(seen_obs[(size_t)o] == 1);
    // ranges of pmax points own whole runs
    REQUIRE(p->n_ranges == (n + p->fz_pmax - 1) / p->fz_pmax);
    REQUIRE(p->d_rng_run[0] == 0 && p->d_rng_run[p->n_ranges] == p->n_runs);
    for (int g = 0; g < p->n_ranges; ++g) {
      REQUIRE(p->d_rng_run[g + 1] > p->d_rng_run[g]);
      REQUIRE(p->d_run_pt0[p->d_rng_run[g]] == g * p->fz_pmax);
    }
    REQUIRE(p->n_keys == (m - 1) / p->fz_group + 1);
    REQUIRE(p->d_key_ptr[0] == 0 && p->d_key_ptr[p->n_keys] == p->n_runs);
    std::vector<char> run_hit((size_t)p->n_runs, 0);
    for (int q = 0; q < p->n_keys; ++q)
      for (int i = p->d_key_ptr[q]; i < p->d_key_ptr[q + 1]; ++i) {
        const int r = p->d_key_run[i];
        REQUIRE(r >= 0 && r < p->n_runs && !run_hit[(size_t)r] && p->d_run_base[r] == q * p->fz_group);
        REQUIRE(i == p->d_key_ptr[q] || p->d_key_run[i - 1] < r);
        run_hit[(size_t)r] = 1;
      }
  }
  if (argc > 6) REQUIRE((p->n_runs > 0) == (atoi(argv[6]) != 0));
  printf("ok runs=%d slots=%d group=%d pmax=%d ", p->n_runs, p->fz_slots, p->fz_group, p->fz_pmax);
  printf("m=%d n=%d N=%lld blocks=%d entries=%lld items=%d parts=%d half_bandwidth=%d\n", m, n, (long long)N,
         p->n_blocks, (long long)E, p->n_pair_items, p->n_parts, sfm_ba_plan_half_bandwidth(p));
  sfm_ba_plan_destroy(p);
  return 0;
}
