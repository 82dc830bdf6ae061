// Internal declarations shared by the translation units of libsfm_ba.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sfm_ba.h"

namespace sfmba {

// ---- tunables -------------------------------------------------------------------------------
constexpr int kGroupSlots = 256;   // observations handled by one workgroup of the per-point kernels
constexpr int kCamChunk = 1024;    // observations per per-camera work item
constexpr int kCamThreads = 128;   // threads of a per-camera work item
constexpr int kPairChunk = 512;   // (obs, obs) entries per camera-pair work item (one wave each)
constexpr int kPairPad = 64;
constexpr int kPairPartStride = 56;  // doubles per partial result of a split block: 7x7 block + e_j        // valid records / T rows kept behind the last entry

// One (observation, observation) pair of a camera-pair block, as byte offsets (the kernel that walks
// these adds a piece offset and issues the fetch; N * 112 < 2^32 is checked when the plan is built)
// Fused point-major assembly ("runs"): a run is a range of consecutive points whose cameras all lie in
// one window of `slots` consecutive camera ids starting at the run's base camera; lane (point, slot)
// of a wave then owns one observation or none, 8 points per step.
constexpr int kFzLanesPerPoint = 8;  // lanes (= slot positions of the table) per point
constexpr int kFzMaxPmax = 1024;     // longest range of points one wave walks (and with it: runs per range)
constexpr int kFzMaxRunsPerRange = 256;  // what the kernel that forms A itself keeps of a range's run boundaries
constexpr int kFzSlotSums = 35;      // per-camera-slot sums behind a partial's tiles: 28 (upper triangle of
                                     // sum A^T A) + 7 (sum A^T g); 7 more per extra row
struct FzShape {
  int slots;  // camera slots of a window: 4 .. 8
  int group;  // consecutive first-cameras that share a window: slots - span + 1
  int pmax;   // a run is cut wherever the point id is a multiple of this
};
// Shape of the runs for n points whose cameras span at most span_max consecutive ids; false when the
// fused path does not apply (a point without observations, or a span beyond 8 cameras).
inline bool fz_shape(int32_t n, int32_t span_max, bool empty_point, FzShape* o) {
  if (empty_point || span_max < 1 || span_max > 8) return false;
  o->slots = span_max < 4 ? 4 : span_max;
  o->group = o->slots - span_max + 1;
  int pm = ((n + 2047) / 2048 + 7) & ~7;
  o->pmax = pm < 32 ? 32 : (pm > kFzMaxPmax ? kFzMaxPmax : pm);
  return true;
}
// runs must be long enough for their fixed cost (one partial result each) to be small
inline bool fz_worthwhile(int32_t n, int32_t n_runs) { return (int64_t)n_runs * 24 <= (int64_t)n + 1536; }
// the 7 * slots rows of a window: 16 per row tile of the matrix cores, the rest ("extra rows") summed by the lanes
__host__ __device__ inline int fz_row_tiles(int slots) { return slots <= 4 ? 2 : (slots <= 7 ? 3 : 4); }
__host__ __device__ inline int fz_extra_rows(int slots) { return slots == 7 ? 1 : 0; }
// doubles of one partial result: lower-triangle tiles (16 x 16), then the per-slot sums as
// [value][slot] (value i of slot s at 8 i + s: 28 of U, 7 of e, 7 per extra row), rounded up to whole
// 256-byte units
__host__ __device__ constexpr int64_t fz_part_stride_c(int tiles, int extra) {
  return (int64_t)(tiles * (tiles + 1) / 2) * 256 + (((kFzSlotSums + 7 * extra) * 8 + 31) & ~31);
}
__host__ __device__ inline int64_t fz_part_stride(int slots) {
  return fz_part_stride_c(fz_row_tiles(slots), fz_extra_rows(slots));
}

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
  // fused point-major assembly (fz_slots = 0: the visibility pattern does not allow it)
  int32_t fz_slots = 0, fz_group = 0, fz_pmax = 0, fz_enabled = 1;
  int32_t n_runs = 0, n_keys = 0, n_ranges = 0, n_steps = 0;
  int32_t fz_max_range_runs = 0;  // most runs inside one range of fz_pmax points
  int32_t* d_run_pt0 = nullptr;   // [n_runs+1] first point of each run
  int32_t* d_run_base = nullptr;  // [n_runs] first camera of the run's window
  int32_t* d_rng_run = nullptr;   // [n_ranges+1] first run of every range of fz_pmax points (one wave per range)
  // steps: up to 8 consecutive points of one run; the steps of a run, of a range and of the plan are consecutive
  int32_t* d_step_pt0 = nullptr;  // [n_steps] first point of the step
  int32_t* d_step_run = nullptr;  // [n_steps + 1] its run (entry n_steps: n_runs)
  int32_t* d_rng_step = nullptr;  // [n_ranges+1] first step of every range
  int32_t* d_slot_obs = nullptr;  // [n][8] observation (caller order) of point i at camera base + slot, or -1
  int32_t* d_key_ptr = nullptr;   // [n_keys+1] runs grouped by window (key = base / group), ascending run id
  int32_t* d_key_run = nullptr;   // [n_runs]
  int64_t device_bytes = 0;
  // workspace section offsets (bytes)
  int64_t ws_g = 0, ws_T = 0, ws_campart = 0, ws_spart = 0, ws_fpart = 0, ws_fkey = 0, ws_fzero = 0, ws_total = 0;
};

namespace sfmba {

void set_hip_error(hipError_t err, const char* what);
void plan_set_workspace(sfm_ba_plan* p);  // workspace section offsets from the plan's counts

// per-kernel timing (ba_profile.cpp): a ProfScope around a launch records two events on the
// launch stream while profiling is enabled, and costs one branch otherwise
enum KernelId {
  KID_CAMERA_PREPARE = 0, KID_JACOBIAN, KID_POINT, KID_CAMERA_ITEMS, KID_CAMERA_FINISH, KID_PAIR_BLOCKS,
  KID_PAIR_FINISH, KID_SOLVE, KID_BACK, KID_UPDATE, KID_COST_ITEMS, KID_COST_FINISH, KID_CALC_W_Y, KID_TRIANGULATE, KID_FUSED_RUNS,
  KID_FUSED_REDUCE, KID_FUSED_FINISH, KID_FUSED_STATE, KID_BACK_STATE, KID_COUNT
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
// (steps: the step blocks of launch_jacobian_steps, or null to gather the operands from A, xt, xp)
int launch_assemble_fused(const sfm_ba_plan* p, const double* A, const double* xt, const double* xp,
                          const double* steps, double mu, double mu_u, double* S, double* e, double* Vinv,
                          double* eps_b, void* ws, hipStream_t st);
int launch_jacobian_steps(const sfm_ba_plan* p, const double* blk, const double* P, const double* X, const double* xt,
                          double* steps, hipStream_t st);
int launch_back_substitute_steps(const sfm_ba_plan* p, const double* steps, const double* Vinv, const double* eps_b,
                                 const double* da, double* db, double* Xupd, double x_step, hipStream_t st);
int launch_calc_W_Y(const sfm_ba_plan* p, const double* A, const double* B, const double* w,
                    const double* Vinv, const int32_t* pt, double* W, double* Y, hipStream_t st);
int launch_add_diagonal(double* S, int n7, double mu, hipStream_t st);
bool fz_state_applies(const sfm_ba_plan* p);
int launch_assemble_state(const sfm_ba_plan* p, const double* blk, const double* P, const double* X, const double* xt,
                          double mu, double mu_u, double* S, double* e, double* Vinv, double* eps_b, void* ws,
                          hipStream_t st);
int launch_back_substitute_state(const sfm_ba_plan* p, const double* blk, const double* Vinv, const double* eps_b,
                                 const double* da, double* db, double* X, int update, double x_step, hipStream_t st);
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
