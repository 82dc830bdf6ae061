// Optional per-kernel timing with HIP events recorded on the launch stream.
#include <mutex>
#include <vector>

#include "ba_common.h"

namespace sfmba {

static const char* kNames[KID_COUNT] = {
    "k_camera_prepare", "k_jacobian",   "k_point",        "k_camera_items",  "k_camera_finish",
    "k_pair_blocks",    "k_pair_finish", "solve_reduced",  "k_back_substitute", "k_update_cameras",
    "k_cost_items",     "k_cost_finish", "k_calc_W_Y",     "k_triangulate",
    "k_fused_runs",     "k_fused_reduce", "k_fused_finish", "k_fused_state", "k_back_substitute_state",
    "k_pair_state"};

struct Pending {
  int id;
  hipEvent_t start, stop;
  bool start_shared;  // start is the stop event of the scope before (chained mode)
};

static bool g_enabled = false;
static bool g_chained = false;        // a scope that follows another one directly starts at its stop event
static uint64_t g_mask = ~uint64_t(0);  // kernel ids that are timed
static hipEvent_t g_chain_event = nullptr;
static hipStream_t g_chain_stream = nullptr;
static bool g_chain_valid = false;
static std::mutex g_mu;
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;

bool profiling_enabled() { return g_enabled; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  // no system-scope acquire / release around a record: nobody reads results on the host at these
  // points (only the time stamps are used), and the fences are most of what a default event costs
  // the kernels around it
  if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(int id, hipStream_t st) : id_(id), st_(st), start_(nullptr) {
  if (!g_enabled) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!((g_mask >> id) & 1)) {
    g_chain_valid = false;  // an untimed launch follows: the next timed scope needs its own start
    return;
  }
  if (g_chained && g_chain_valid && g_chain_stream == st_) {
    start_ = g_chain_event;
    shared_ = true;
    return;
  }
  start_ = get_event();
  if (start_) (void)hipEventRecord(static_cast<hipEvent_t>(start_), st_);
}

ProfScope::~ProfScope() {
  if (!start_) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t stop = get_event();
  if (!stop) return;
  (void)hipEventRecord(stop, st_);
  g_pending.push_back(Pending{id_, static_cast<hipEvent_t>(start_), stop, shared_});
  g_chain_event = stop;
  g_chain_stream = st_;
  g_chain_valid = true;
}

}  // namespace sfmba

using namespace sfmba;

extern "C" int sfm_ba_profile_enable(int mode) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = mode != 0;
  g_chained = mode == 2;
  g_chain_valid = false;
  return SFM_BA_OK;
}

extern "C" int sfm_ba_profile_select(uint64_t kernel_mask) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_mask = kernel_mask;
  g_chain_valid = false;
  return SFM_BA_OK;
}

extern "C" int sfm_ba_profile_kernel_count(void) { return KID_COUNT; }

extern "C" const char* sfm_ba_profile_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? kNames[id] : ""; }

extern "C" int sfm_ba_profile_read(double* ms_total, int64_t* launches) {
  if (!ms_total || !launches) return SFM_BA_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < KID_COUNT; ++i) {
    ms_total[i] = 0.0;
    launches[i] = 0;
  }
  for (const Pending& p : g_pending) {
    SFMBA_HIP(hipEventSynchronize(p.stop));
    float ms = 0.f;
    SFMBA_HIP(hipEventElapsedTime(&ms, p.start, p.stop));
    ms_total[p.id] += ms;
    launches[p.id] += 1;
    if (!p.start_shared) g_pool.push_back(p.start);
    g_pool.push_back(p.stop);
  }
  g_pending.clear();
  g_chain_valid = false;
  return SFM_BA_OK;
}
