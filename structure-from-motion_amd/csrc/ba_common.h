// Internal declarations shared by the translation units of libsfm_ba.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sfm_ba.h"

namespace sfmba {

// ---- tunables -------------------------------------------------------------------------------
constexpr int kGroupSlots = 256;   // observations handled by one workgroup of the per-point kernels
constexpr int kCamChunk = 1024;    // observations per per-camera work item
constexpr int kCamThreads = 128;   // threads of a per-camera work item
constexpr int kPairChunk = 512;   // (obs, obs) entries per camera-pair work item (one wave each)
constexpr int kPairPad = 64;
constexpr int kPointRec = 10;       // doubles per point record of the camera-pair kernel that forms its rows itself:
                                    // position (3), V^-1 as 00 01 02 11 12 22 (6), one of padding (80 bytes)
constexpr int kPairPartStride = 64;  // doubles per partial result of a split block: 7x7 block, e_j (49..55), the
                                     // camera's sum of squared residuals (56); diagonal blocks only fill 49..56

// Fused point-major assembly.  The plan sorts the points it can handle this way ("near" points) by the
// window of cameras they are seen from and walks them by position in that order:
//   window  a list of up to `slots` cameras.  A point whose cameras span at most `slots` consecutive ids
//           uses a window of consecutive cameras (base = first camera rounded down to a multiple of
//           slots - span + 1, so that points with nearby first cameras share it); a point with at most
//           `slots` cameras that lie further apart uses its own camera set as the window
//   run     consecutive positions of one window, cut at multiples of pmax; lane (point, slot) of a
//           wave then owns one observation or none
// There are two classes of near points: class 0 with windows of 4..8 cameras (8 lanes per point, 8 points
// per step; the width is chosen per plan), and class 1 with windows of 16 cameras (16 lanes per point, 4
// points per step) for the points class 0 leaves over - the longer tracks.  Points with more cameras
// than that, and points whose window would hold too few points to be worth a partial result of its own,
// are "far": they go through the camera-pair kernels of a second plan over their observations, and the
// finishing kernel adds all of it into S.
constexpr int kFzClasses = 2;
constexpr int kFzMaxLanes = 16;      // lanes (= slot positions of the tables) per point: 8 (class 0) or 16 (class 1)
constexpr int kFzMaxPmax = 1024;     // longest range of points one wave walks (and with it: runs per range)
constexpr int kFzMaxRunsPerRange = 256;  // what the kernel that forms A itself keeps of a range's run boundaries
constexpr int kFzSlotSums = 36;      // per-camera-slot sums behind a partial's tiles: 28 (upper triangle of
                                     // sum A^T A) + 7 (sum A^T g) + 1 (sum |eps|^2); 7 more per extra row
__host__ __device__ inline int fz_lanes(int slots) { return slots > 8 ? 16 : 8; }
// a window with fewer points is not worth a partial result of its own (16 KiB .. 61 KiB)
inline int fz_min_window_points(int cls) { return cls == 0 ? 4 : 16; }
// Cameras per window of class 0, from two histograms over the points with at most 8 cameras: obs_by_k[v]
// / obs_by_span[v] = observations of the points with v cameras / with a camera span of v (spans
// beyond 8: not counted).  The smallest width that keeps 98 % of what a width of 8 would keep; measured
// on the span when at least half of those observations belong to points that fit a window of
// consecutive cameras, otherwise on the camera count.  0: no point qualifies.
inline int fz_choose_slots(const int64_t obs_by_k[9], const int64_t obs_by_span[9]) {
  int64_t tot_k = 0, tot_s = 0;
  for (int v = 1; v <= 8; ++v) {
    tot_k += obs_by_k[v];
    tot_s += obs_by_span[v];
  }
  if (tot_k == 0) return 0;
  const int64_t* h = 2 * tot_s >= tot_k ? obs_by_span : obs_by_k;
  const int64_t tot = 2 * tot_s >= tot_k ? tot_s : tot_k;
  int64_t acc = 0;
  for (int v = 1; v <= 8; ++v) {
    acc += h[v];
    if (v >= 4 && 50 * acc >= 49 * tot) return v;
  }
  return 8;
}
// a run is cut wherever the position is a multiple of this: one wave per range of pmax near points
inline int fz_pmax_for(int32_t n_near) {
  static const int ranges = [] {
    const char* v = getenv("SFM_BA_FZ_RANGES");
    const int r = v ? atoi(v) : 0;
    return r >= 64 ? r : 2048;
  }();
  const int pm = ((n_near + ranges - 1) / ranges + 7) & ~7;
  return pm < 32 ? 32 : (pm > kFzMaxPmax ? kFzMaxPmax : pm);
}
// A class is kept when its runs hold enough points on average (a run costs a partial result of its own:
// about what two steps cost in class 0, and 61 KiB written and read in class 1) and its points are a
// fair share of all points.
inline bool fz_worthwhile(int cls, int32_t n, int32_t n_near, int32_t n_runs) {
  if (n_near <= 0) return false;
  if (cls == 0) return (int64_t)n_runs * 8 <= (int64_t)n_near + 1536 && (int64_t)n_near * 8 >= (int64_t)n;
  return (int64_t)n_runs * 24 <= (int64_t)n_near + 1536 && (int64_t)n_near * 32 >= (int64_t)n;
}
// sort key of a near candidate: windows of consecutive cameras first, by base camera, then the
// camera-set windows by a hash of the set (equal sets get equal keys; different sets with equal
// hashes only cost run length, the windows are told apart by comparing the cameras)
__host__ __device__ inline uint32_t fz_hash_cams(const int32_t* cams, int k) {
  uint32_t h = 2166136261u;
  for (int i = 0; i < k; ++i) {
    h ^= (uint32_t)cams[i] + 0x9e3779b9u + (h << 6) + (h >> 2);
    h *= 16777619u;
  }
  return h;
}
// Window of a point with cameras cams[0..k) (ascending) in a class with `slots` cameras per window
// (`lanes` table positions) and m cameras: fills desc[0..lanes) (-1: no camera at that slot) and the sort
// key; false when the point does not fit a window (k = 0 or k > slots).
__host__ __device__ inline bool fz_point_window(const int32_t* cams, int k, int slots, int lanes, int m, int32_t* desc,
                                                uint64_t* key) {
  if (k < 1 || k > slots) return false;
  const int span = cams[k - 1] - cams[0] + 1;
  if (span <= slots) {
    const int g = slots - span + 1;
    int base = cams[0] / g * g;
    const int last = m - slots > 0 ? m - slots : 0;  // windows do not reach past the last camera
    if (base > last) base = last;
    for (int s = 0; s < lanes; ++s) desc[s] = (s < slots && base + s < m) ? base + s : -1;
    *key = (uint64_t)(uint32_t)base;
  } else {
    for (int s = 0; s < lanes; ++s) desc[s] = s < k ? cams[s] : -1;
    *key = (uint64_t(1) << 32) | fz_hash_cams(cams, k);
  }
  return true;
}
// slot of camera c in the window desc (the camera is there)
__host__ __device__ inline int fz_slot_of(const int32_t* desc, int lanes, int32_t c) {
  for (int s = 0; s < lanes; ++s)
    if (desc[s] == c) return s;
  return -1;
}
// the 7 * slots rows of a window: 16 per row tile of the matrix cores, the rest ("extra rows") summed by the lanes
__host__ __device__ inline int fz_row_tiles(int slots) { return slots <= 4 ? 2 : (slots <= 7 ? 3 : (slots <= 8 ? 4 : 7)); }
__host__ __device__ inline int fz_extra_rows(int slots) { return slots == 7 ? 1 : 0; }
// doubles of one partial result: lower-triangle tiles (16 x 16), then the per-slot sums as
// [value][slot] (value i of slot s at lanes i + s: 28 of U, 7 of e, the squared residuals, 7 per extra
// row), rounded up to whole 256-byte units
__host__ __device__ constexpr int64_t fz_part_stride_c(int tiles, int extra, int lanes) {
  return (int64_t)(tiles * (tiles + 1) / 2) * 256 + (((kFzSlotSums + 7 * extra) * lanes + 31) & ~31);
}
__host__ __device__ inline int64_t fz_part_stride(int slots) {
  return fz_part_stride_c(fz_row_tiles(slots), fz_extra_rows(slots), fz_lanes(slots));
}
// entry of the finishing kernel: class, window, slot of camera j, slot of camera k
__host__ __device__ inline int32_t fz_fb_pack(int cls, int key, int sj, int sk) { return (cls << 30) | (key << 8) | (sj << 4) | sk; }

// one class of near points: its tables on the device
struct FzClass {
  int32_t slots = 0;  // cameras per window (0: the class is empty)
  int32_t lanes = 8;  // table positions per point
  int32_t pmax = 0, n_near = 0, n_runs = 0, n_keys = 0, n_ranges = 0, n_steps = 0;
  int32_t max_range_runs = 0;     // most runs inside one range of pmax positions
  int32_t* d_pt = nullptr;        // [n_near] point id at every position
  int32_t* d_slot_obs = nullptr;  // [n_near][lanes] observation (caller order) of the point at that position by the
                                  // camera at that slot of its window, or -1
  int32_t* d_run_pt0 = nullptr;   // [n_runs+1] first position of each run
  int32_t* d_run_key = nullptr;   // [n_runs] window of the run
  int32_t* d_run_cam = nullptr;   // [n_runs][lanes] cameras of the run's window, for the kernels' loads: a slot
                                  // without a camera repeats the window's first camera
  int32_t* d_rng_run = nullptr;   // [n_ranges+1] first run of every range of pmax positions (one wave per range)
  int32_t* d_key_cam = nullptr;   // [n_keys][lanes] camera at every slot of the window, -1: none
  int32_t* d_key_run = nullptr;   // [n_keys+1] first run of every window (the runs of a window are consecutive)
  int64_t ws_part = 0, ws_key = 0;  // workspace: one partial result per run, one per window
};

// One (observation, observation) pair of a camera-pair block, as byte offsets (the kernel that walks
// these adds a piece offset and issues the fetch; N * 112 < 2^32 is checked when the plan is built)
struct PairEntry {
  uint32_t row_a;  // 112 * observation (caller order) of the block's row camera j: its row of A
  uint32_t row_b;  // off the diagonal (j < k): 112 * observation of camera k, same point.  On the
                   // diagonal the pair couples an observation with itself, row_b would repeat
                   // row_a, and the field holds 16 * observation instead: its entry of g
};

}  // namespace sfmba

// Opaque plan: device-resident index structures of one visibility pattern.
struct sfm_ba_plan {
  int32_t m = 0, n = 0;
  int64_t n_obs = 0;
  // observations grouped by point ("slots"; inside a point, cameras ascend)
  int32_t* d_pt_ptr = nullptr;    // [n+1]
  int32_t* d_perm = nullptr;      // [N] slot -> observation position in caller order
  int32_t* d_slot_cam = nullptr;  // [N] camera of the slot
  int32_t* d_tp_ptr = nullptr;    // [n+1] first pair slot of the point (k(k+1)/2 per point)
  int32_t* d_pair_entry = nullptr;  // [pairs] pair slot -> position of its entry in d_entries (= row of T)
  int32_t* d_grp_pt = nullptr;    // [n_groups+1] first point of each workgroup's range
  int32_t n_groups = 0;
  int64_t n_pair_slots = 0;
  // per-camera work items over the camera-sorted observation list
  int32_t* d_cam_obs = nullptr;   // [N] or null when the caller order is already camera-sorted
  int32_t* d_item_cam = nullptr;  // [n_items]
  int32_t* d_item_beg = nullptr;  // [n_items] range in the camera-sorted list
  int32_t* d_item_end = nullptr;
  int32_t* d_cam_item_ptr = nullptr;  // [m+1]
  int32_t n_items = 0;
  // camera-pair blocks (j <= k) with at least one shared point
  int32_t n_blocks = 0;
  int32_t max_cam_dist = 0;  // largest k - j over camera pairs (j <= k) that share a point
  int32_t* d_blk_j = nullptr;
  int32_t* d_blk_k = nullptr;
  int64_t n_entries = 0;
  sfmba::PairEntry* d_entries = nullptr;
  int32_t* d_ent_pt = nullptr;    // [n_entries] point (of this plan) the entry's two observations belong to
  // pair work items: chunk of one block's entries
  int32_t n_pair_items = 0;
  int32_t* d_pit_blk = nullptr;
  int32_t* d_pit_beg = nullptr;
  int32_t* d_pit_end = nullptr;
  int32_t* d_pit_part = nullptr;  // -1: the block has one item and is written directly
  int32_t n_parts = 0;            // partial 7x7 results of multi-item blocks
  int32_t n_multi = 0;            // blocks with more than one item
  int32_t* d_mb_blk = nullptr;
  int32_t* d_mb_part = nullptr;   // first partial
  int32_t* d_mb_cnt = nullptr;    // number of partials
  // fused point-major assembly over the near points (n_runs = 0: the plan has none)
  sfmba::FzClass fz[sfmba::kFzClasses];
  int32_t fz_enabled = 1;
  int32_t n_runs = 0;  // runs of both classes
  // per camera-pair block (the plan's blocks, j <= k): where the finishing kernel finds its sums
  int32_t* d_fb_ptr = nullptr;    // [n_blocks+1] entries of the block in d_fb_ent
  int32_t* d_fb_ent = nullptr;    // [n_fb] fz_fb_pack(class, window, slot of camera j, slot of camera k)
  int32_t* d_fb_far_part = nullptr;  // [n_blocks] first partial result of the block in the far plan's workspace
  int32_t* d_fb_far_cnt = nullptr;   // [n_blocks] number of those partials (0: the far points do not touch the block)
  int32_t n_fb = 0;
  // far points: a plan of their own over their observations, points renumbered 0 .. n_far - 1 in
  // ascending id, observations in the caller's relative order
  sfm_ba_plan* far = nullptr;
  int32_t n_far = 0;
  int64_t n_far_obs = 0;
  int32_t* d_far_pt = nullptr;    // [n_far] point id of far point q
  int32_t* d_far_obs = nullptr;   // [n_far_obs] observation (caller order) of far observation i
  int32_t* d_far_cam = nullptr;   // [n_far_obs] its camera
  int32_t* d_far_lpt = nullptr;   // [n_far_obs] its far point number
  int32_t* d_far_gpt = nullptr;   // [n_far_obs] its point id (= d_far_pt[d_far_lpt[i]])
  int32_t force_parts = 0;        // this plan is the far plan of another: every block writes partial results
  int64_t device_bytes = 0;
  // workspace section offsets (bytes)
  int64_t ws_g = 0, ws_T = 0, ws_q = 0, ws_campart = 0, ws_spart = 0, ws_prec = 0, ws_fzero = 0, ws_far_A = 0,
          ws_far_xt = 0, ws_far_xp = 0, ws_far = 0, ws_total = 0;
};

namespace sfmba {

void set_hip_error(hipError_t err, const char* what);
void plan_set_workspace(sfm_ba_plan* p);  // workspace section offsets from the plan's counts

// per-kernel timing (ba_profile.cpp): a ProfScope around a launch records two events on the
// launch stream while profiling is enabled, and costs one branch otherwise
enum KernelId {
  KID_CAMERA_PREPARE = 0, KID_JACOBIAN, KID_POINT, KID_CAMERA_ITEMS, KID_CAMERA_FINISH, KID_PAIR_BLOCKS,
  KID_PAIR_FINISH, KID_SOLVE, KID_BACK, KID_UPDATE, KID_COST_ITEMS, KID_COST_FINISH, KID_CALC_W_Y, KID_TRIANGULATE, KID_FUSED_RUNS,
  KID_FUSED_REDUCE, KID_FUSED_FINISH, KID_FUSED_STATE, KID_BACK_STATE, KID_PAIR_STATE, KID_COUNT
};
bool profiling_enabled();
class ProfScope {
 public:
  ProfScope(int id, hipStream_t st);
  ~ProfScope();
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;

 private:
  int id_;
  hipStream_t st_;
  void* start_;
  bool shared_ = false;
};

#define SFMBA_HIP(call)                         \
  do {                                          \
    hipError_t err__ = (call);                  \
    if (err__ != hipSuccess) {                  \
      sfmba::set_hip_error(err__, #call);       \
      return SFM_BA_EHIP;                       \
    }                                           \
  } while (0)

// launchers implemented in ba_kernels.hip / ba_cholesky.hip
int launch_camera_prepare(const double* R, const double* C, const double* K, int k_stride, int m,
                          double* blk, int32_t* flags, hipStream_t st);
int launch_jacobian(const double* blk, const double* P, const double* X, const int32_t* cam,
                    const int32_t* pt, int64_t N, double* A, double* B, double* xp, hipStream_t st);
int launch_assemble(const sfm_ba_plan* p, const double* A, const double* B, const double* xt,
                    const double* xp, const double* w, double mu, double mu_u, double* S, double* e,
                    double* U, double* eps_a, double* Vinv, double* eps_b, double* cam_red,
                    void* ws, hipStream_t st);
// the fused path of launch_assemble (plan with runs, no weights, no by-products, B taken from A)
int launch_assemble_fused(const sfm_ba_plan* p, const double* A, const double* xt, const double* xp, double mu,
                          double mu_u, double* S, double* e, double* Vinv, double* eps_b, void* ws, hipStream_t st);
// the camera-pair path of a far plan (see sfm_ba_plan::far): partial results only, points renumbered by pt_map
int launch_assemble_parts(const sfm_ba_plan* p, const int32_t* pt_map, const double* A, const double* xt,
                          const double* xp, double mu, double* Vinv, double* eps_b, void* ws, hipStream_t st);
// the camera-pair path of a whole plan in the loop: A as sfm_ba_jacobian wrote it from this state (the per-point
// stage and the back-substitution read it), camera-pair products formed from the state; S, e, the per-camera
// squared residuals
int launch_assemble_pairs_state(const sfm_ba_plan* p, const double* A, const double* blk, const double* X,
                                const double* xt, const double* xp, double mu, double mu_u, double* S, double* e,
                                double* Vinv, double* eps_b, double* ssq, void* ws, hipStream_t st);
int launch_back_substitute_mapped(const sfm_ba_plan* p, const int32_t* pt_map, const double* A, const double* Vinv,
                                  const double* eps_b, const double* da, double* db, double* Xupd, double x_step,
                                  hipStream_t st);
int launch_calc_W_Y(const sfm_ba_plan* p, const double* A, const double* B, const double* w,
                    const double* Vinv, const int32_t* pt, double* W, double* Y, hipStream_t st);
int launch_add_diagonal(double* S, int n7, double mu, hipStream_t st);
bool fz_state_applies(const sfm_ba_plan* p);
int launch_assemble_state(const sfm_ba_plan* p, const double* blk, const double* P, const double* X, const double* xt,
                          double mu, double mu_u, double* S, double* e, double* Vinv, double* eps_b, double* ssq, void* ws,
                          hipStream_t st);
int launch_back_substitute_state(const sfm_ba_plan* p, const double* blk, const double* Vinv, const double* eps_b,
                                 const double* da, double* db, double* X, int update, double x_step, void* ws,
                                 hipStream_t st);
int launch_status(const double* cost, const int32_t* info, const int32_t* fa, const int32_t* fb, int m, double* out,
                  hipStream_t st);
int launch_lower_pack(const double* S, int n7, double* tri, hipStream_t st);
int launch_lower_unpack(const double* tri, int n7, double mu, double* S, hipStream_t st);
int launch_back_substitute(const sfm_ba_plan* p, const double* A, const double* B, const double* w,
                           const double* Vinv, const double* eps_b, const double* da, double* db,
                           double* Xupd, double x_step, hipStream_t st);
int launch_update_cameras(double* R, double* T, double* C, double* P, const double* K, int k_stride,
                          const double* da, int m, int first, int held, double q_step, double c_step,
                          int32_t* flags, double* blk, int32_t* blk_flags, hipStream_t st);
int launch_cost(const sfm_ba_plan* p, const double* P, const double* X, const int32_t* cam, const int32_t* pt,
                const double* xt, double* ssq, double* cost, void* ws, hipStream_t st);
int launch_cost_partial(const sfm_ba_plan* p, const double* P, const double* X, const int32_t* cam,
                        const int32_t* pt, const double* xt, double* ssq, void* ws, hipStream_t st);
int launch_cost_finish(const double* ssq, int m, double* cost, hipStream_t st);
int launch_cost_noalloc(const sfm_ba_plan* p, const double* P, const double* X, const int32_t* cam, const int32_t* pt,
                        const double* xt, double* ssq, double* cost, hipStream_t st);
int64_t cholesky_workspace_bytes(int n7);
// half_bandwidth < 0: dense; otherwise S[r][c] = 0 wherever |r - c| > half_bandwidth
int launch_solve_reduced(double* S, int n7, int half_bandwidth, double* da, void* ws, int32_t* info, hipStream_t st);
int reduced_solve_block(int n7, int half_bandwidth);
int launch_permute_system(const double* src, int n7, const int32_t* order, int half_bandwidth, double* dst,
                          hipStream_t st);

int64_t band_packed_elements(int n7);
int launch_band_pack(const double* S, int n7, double* packed, hipStream_t st);
int64_t band_compact_elements(int n7, int half_bandwidth);
int launch_band_compact(const double* S, int n7, int half_bandwidth, double* compact, hipStream_t st);
int launch_band_expand(const double* compact, int n7, int half_bandwidth, double* packed, hipStream_t st);
int launch_solve_band_packed(double* packed, int n7, double mu_diag, double* da, void* ws, int32_t* info,
                             hipStream_t st);

int launch_triangulate(const double* x1, const double* x2, const double* P1, const double* P2, int64_t n, float* Xh,
                       hipStream_t st);

int64_t pnp_workspace_bytes(int n_hyp, int s);
int launch_pnp_solve(const double* x, const double* X, const double* Kinv_pre, const double* Kinv_post,
                     const int32_t* idx, int n_hyp, int s, void* ws, double* R, double* T, int32_t* flags,
                     hipStream_t st);
int launch_pnp_cleanup(const double* RT, const double* Kinv_post, int n, double* R, double* T, int32_t* flags,
                       hipStream_t st);
int launch_pnp_inliers(const double* x, const double* X, const double* K, const double* R, const double* T, int n_hyp,
                       int64_t n, double threshold, int32_t* counts, uint8_t* mask, hipStream_t st);


}  // namespace sfmba
