// Fused point-major Schur assembly for gfx950 (MI355X): calc_V_inv, calc_W, calc_Y, calc_S, calc_U,
// calc_epsilon_*, calc_e of the reference (src/sparse_ba.py:147-271) in one pass over A.
//
// Applies to the plan's "near" points: the plan keeps them sorted by the window of (up to 8) cameras
// they are seen from and cuts that order into runs of one window each (ba_common.h).  Then
//   S_window -= sum_i Y_i W_i^T,   Y_i = W_i V_i^-1,   W_i = [A_ij^T B_ij]_j  (7 rows per camera j)
// is a dense contraction over the points of the run: with V_i^-1 = G_i^T G_i and Z_i = W_i G_i^T it
// is the symmetric rank-3 update sum_i Z_i Z_i^T, and a wave accumulates it with v_mfma_f64_16x16x4
// (rows 7 slot + r of the window, K = 3 columns per point).  On gfx950 the fp64 MFMA and the fp64
// vector instructions of a SIMD share one pipe (they do not overlap, measured with
// scripts/probes/dp_coexec.hip), so the kernel's time is the sum of its MFMA and VALU cycles: the
// window is as small as the visibility allows (no tile of padding) and the per-point arithmetic
// avoids divisions.
//
// One wave per range of pmax positions (a range holds whole runs), no barriers; the operand fetches
// run two steps ahead and do not stop at the end of a run.  Lane (p, s) of a step owns the observation of point p (8 points per
// step) by the camera at slot s of the window (or none: it then contributes zeros):
//   1. the row of A, x_true and x_pred come straight into registers (loads for the next step are
//      issued before the matrix phase of this one);
//   2. V_i, eps_b_i are summed over the 8 lanes of a point by DPP, inverted in every lane;
//   3. the lane adds its A^T A and A^T g to 35 running sums (U_j and e_j of its slot; reduced over
//      the 8 points once per run) and writes its 7 x 3 block of Z into the wave's LDS panel;
//   4. 6 K-steps of 4 columns: every lane reads one double per row tile and issues the MFMAs of the
//      lower-triangle tiles.
// At the end of the run the wave stores its tiles and per-slot sums as one partial result.  Partials
// are added in a fixed order (k_fused_reduce: runs of one window; k_fused_finish: windows of one
// camera pair), so results are reproducible bit for bit.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ba_common.h"
#include "ba_rows.h"

namespace sfmba {

typedef double d4_t __attribute__((ext_vector_type(4)));

// Diagnostic builds (-DSFMBA_FZ_STAMPS) add up, per phase of k_fused_runs, the shader cycles every
// wave spends there; the launcher prints and clears the sums.  The library build has no stamps.
#ifdef SFMBA_FZ_STAMPS
__device__ unsigned long long g_fz_stamps[16];
#define FZ_STAMP_DECL long long stamp_acc_[16] = {0}, stamp_prev_ = (long long)__builtin_amdgcn_s_memtime()
#define FZ_STAMP(i)                                                  \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime();  \
    stamp_acc_[i] += now_ - stamp_prev_;                             \
    stamp_prev_ = now_;                                              \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)
#define FZ_STAMP_WAIT(what) asm volatile("s_waitcnt " what ::: "memory")
#define FZ_STAMP_FLUSH                                                                      \
  do {                                                                                      \
    if (threadIdx.x == 0)                                                                   \
      for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_fz_stamps[i_], (unsigned long long)stamp_acc_[i_]); \
  } while (0)
#else
#define FZ_STAMP_DECL
#define FZ_STAMP(i)
#define FZ_STAMP_WAIT(what)
#define FZ_STAMP_FLUSH
#endif

__device__ __forceinline__ double2 ldg2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void sts2(double* p, double a, double b) {
  *reinterpret_cast<double2*>(p) = make_double2(a, b);
}

template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// sum over the 8 lanes that share lane >> 3 (every lane receives it, all eight bit-identical:
// each step adds the same two numbers in both partners)
__device__ __forceinline__ double sum8(double v) {
  v += dpp64<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp64<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp64<0x141>(v);  // row_half_mirror: the other quad of the 8 (all four lanes of a quad agree by now)
  return v;
}

// sum over the LPP (8 or 16) lanes of a point, in every lane of the group
template <int LPP>
__device__ __forceinline__ double sum_point(double v) {
  v = sum8(v);
  if (LPP == 16) v += dpp64<0x140>(v);  // row_mirror: the other half of the row of 16 (all eight lanes of a half agree)
  return v;
}

// value of lane SX (0..7) of this lane's group of 8, in every lane of the group
template <int SX>
__device__ __forceinline__ double bcast8(double v, int s) {
  constexpr int c = SX & 3;
  const double t = dpp64<(c | (c << 2) | (c << 4) | (c << 6))>(v);  // every quad: its lane c
  const double o = dpp64<0x141>(t);                                  // row_half_mirror: the other quad's
  return (s >> 2) == (SX >> 2) ? t : o;
}

// 1 / sqrt(x): hardware estimate refined by three Newton steps (the estimate is good to roughly
// single precision; the third step is there for margin).  No special cases: x is a positive pivot.
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double t = x * y;
    const double e = fma(-t, y, 1.0);
    y = fma(0.5 * y, e, y);
  }
  return y;
}

// W camera slots per window, rows 7 slot + r (dense: no padding rows between cameras); the first
// 16 T rows go through the matrix cores, the E rows behind them are accumulated by the lanes
// themselves (21 FMAs per row and step: cheaper than another row of tiles for one row; three such
// rows for W = 5 on two row tiles cost 42 registers more than a third row of tiles and spilled).
//   W = 4: T 2, E 0 (28 rows)   W = 5: T 3, E 0 (35)   W = 6: T 3, E 0 (42)   W = 7: T 3, E 1 (49)
//   W = 8: T 4, E 0 (56)
// The operands are gathered row by row from A, x_true and x_pred in the caller's order through the
// slot table; V^-1 and eps_b go to the point's id (fz_pt: point id at every position).
// what lanes of k_fused_runs without an observation read as their operands: zeros, never written
__device__ __align__(256) const double g_fz_zero_row[32] = {0.0};

template <int W, int T, int E, int WPS, int LPP>  // WPS: waves per SIMD the register allocation leaves room for;
                                                 // LPP: lanes (= camera slots of the tables) per point
__global__ __launch_bounds__(64, WPS) void k_fused_runs(
    const int32_t* __restrict__ rng_run, const int32_t* __restrict__ run_pt0,
    const int32_t* __restrict__ slot_obs, const int32_t* __restrict__ fz_pt, const double* __restrict__ A,
    const double* __restrict__ xt, const double* __restrict__ xp, double mu, double* __restrict__ Vinv,
    double* __restrict__ eps_b, double* __restrict__ part, int pmax,
    int n_pts) {
  static_assert(16 * T + E >= 7 * W && 16 * T <= 7 * W + 15, "tiles and extra rows must cover the window");
  static_assert(E == 0 || (16 * T) / 7 == (16 * T + E - 1) / 7, "the extra rows belong to one camera slot");
  static_assert(E == 0 || (16 * T) / 7 == 4 || (16 * T) / 7 == 6, "extra rows are wired for slots 4 and 6");
  static_assert(LPP == 8 || LPP == 16, "8 or 16 lanes per point");
  static_assert(E == 0 || LPP == 8, "extra rows are wired for 8 lanes per point");
  constexpr int PPS = 64 / LPP;   // points per step
  constexpr int KS = 8 * LPP;     // doubles per K column of the panel: 8 per camera slot (7 rows + a zero)
  constexpr int NK = 3 * PPS / 4; // K steps of 4 columns per step
  constexpr int NT = T * (T + 1) / 2;
  constexpr int64_t STRIDE = fz_part_stride_c(T, E, LPP);
  __shared__ __attribute__((aligned(16))) double Zp[3 * PPS * KS];  // [K column = 3 point + q][8 slot + r]
  __shared__ int bnd[kFzMaxPmax + 1];  // first point of every run of this wave's range, and the range's end
#define FZ_RESIDUAL_IN_ROWS false
  const int lane = threadIdx.x;
  const double* zero_row = g_fz_zero_row;
  const int p = lane / LPP, s = lane % LPP;
  const int li = lane & 15, lk = lane >> 4;
  // panel offset of dense row 16 t + li for the operand reads (rows past the window: a zero entry)
  int roff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int rho = 16 * t + li;
    roff[t] = rho < 7 * W ? 8 * (rho / 7) + rho % 7 : 7;
  }
  // A wave walks the runs of one range of pmax points (ranges hold equal numbers of points) as one
  // stream of steps: a step holds up to 8 consecutive points of one run, and the operands of the
  // next step - the first step of the next run included - are fetched while this one is worked on.
  // The boundaries of the range's runs sit in LDS (at most pmax + 1 of them).
  FZ_STAMP_DECL;
  const int run0 = rng_run[blockIdx.x];
  const int nr = rng_run[blockIdx.x + 1] - run0;
  int f0 = blockIdx.x * pmax, r0 = 0;  // first point of the step, its run (relative to run0)
  // slot table entry of this lane for a step that starts at point `first` in a run that ends before `lim`
  // (a 32-bit index: the address is then the kernel argument plus a lane offset; as a 64-bit per-lane
  // pointer it is kept in registers across the steps, spilled, and every reload of a spill waits for
  // all loads in flight - the next step's operands)
#define FZ_FETCH_POS(first, lim) (((first) + p) < (lim) ? slot_obs[(unsigned)(LPP * ((first) + p) + s)] : -1)
  // row of A (a_(0..13)), x_true (a_(14..15)) and x_pred (a_(16..17)) of observation `pos`; a lane
  // without an observation reads the row of zeros the launcher keeps in the workspace (selecting the
  // address keeps the loads unconditional, so that they land in the registers of av[] themselves)
#define FZ_FETCH_ROWS(pos)                                                    \
  do {                                                                        \
    const bool has_ = (pos) >= 0;                                             \
    const double* ap_ = has_ ? A + 14 * (int64_t)(pos) : zero_row;            \
    const double* tp_ = has_ ? xt + 2 * (int64_t)(pos) : zero_row;            \
    const double* pp_ = has_ ? xp + 2 * (int64_t)(pos) : zero_row;            \
    _Pragma("unroll") for (int q_ = 0; q_ < 7; ++q_) av[q_] = ldg2(ap_ + 2 * q_); \
    av[7] = ldg2(tp_);                                                        \
    av[8] = ldg2(pp_);                                                        \
  } while (0)
  // step after the one that starts at `first` in run `r`: 8 points on, or the start of the next run
#define FZ_ADVANCE(first, r)                                                    \
  do {                                                                          \
    (first) += PPS;                                                             \
    if ((first) >= __builtin_amdgcn_readfirstlane(bnd[(r) + 1])) {              \
      (r) += 1;                                                                 \
      (first) = __builtin_amdgcn_readfirstlane(bnd[(r)]);                       \
    }                                                                           \
  } while (0)
#define FZ_LIM(r) __builtin_amdgcn_readfirstlane(bnd[(r) + 1])
  // the first step's table entries do not need the boundaries: they are on their way meanwhile
  // point id of the lane's point in the step that starts at position `first`
#define FZ_FETCH_PID(first) fz_pt[(unsigned)(((first) + p) < n_pts ? (first) + p : n_pts - 1)]
  const int pt_first = f0 + p < n_pts ? f0 + p : n_pts - 1;
  const int pos_raw = slot_obs[(unsigned)(LPP * pt_first + s)];
  int pid = fz_pt[(unsigned)pt_first];
  for (int i = lane; i <= nr; i += 64) bnd[i] = run_pt0[run0 + i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // (kept as the 16-byte pieces the loads deliver: the values carried around the loop are then the
  // load results themselves; as 18 separate doubles they get registers of their own and every step
  // ends with a wait for the loads and 18 copies)
  double2 av[9];
#define a_(i) (((i) & 1) ? av[(i) >> 1].y : av[(i) >> 1].x)
  int pos = f0 + p < FZ_LIM(0) ? pos_raw : -1;
  FZ_FETCH_ROWS(pos);
  int f1 = f0, r1 = r0;
  FZ_ADVANCE(f1, r1);
  int pos_next = r1 < nr ? FZ_FETCH_POS(f1, FZ_LIM(r1)) : -1;
  int pid_next = r1 < nr ? FZ_FETCH_PID(f1) : 0;
  int f2 = f1, r2 = r1;
  if (r2 < nr) FZ_ADVANCE(f2, r2);
  int pos_next2 = r2 < nr ? FZ_FETCH_POS(f2, FZ_LIM(r2)) : -1;

  d4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = d4_t{0.0, 0.0, 0.0, 0.0};
  double u[28], ev[7], ssq = 0.0, xa[E > 0 ? E : 1][7];
#pragma unroll
  for (int i = 0; i < 28; ++i) u[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) ev[i] = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int i = 0; i < 7; ++i) xa[e][i] = 0.0;

  FZ_STAMP(0);  // set-up
  while (r0 < nr) {  // runs of the range
  // steps of the run, as a counted loop of its own: the partial result is written behind it, and
  // code of that part which the compiler moved into the steps (it does so when the end of a run is
  // a branch inside one flat loop) makes every step wait for its matrix instructions
  const int run = run0 + r0;
  int nsteps = (FZ_LIM(r0) - f0 + PPS - 1) / PPS;
  asm volatile("" : "+s"(nsteps));
  int step = 0;
  do {  // (at least one step: the loop's exit then has a block of its own for the copies the write-out needs)
    const int pt = f0 + p;
    const int pt1 = FZ_LIM(r0);
    FZ_STAMP_WAIT("vmcnt(0)");
    FZ_STAMP(1);  // operands of this step

#include "ba_fused_step_a.inc"
    // ---- next step's operands on their way while the matrix cores work ----
    FZ_STAMP(3);  // Z and its stores
    FZ_STAMP_WAIT("lgkmcnt(0)");
    FZ_STAMP(4);  // panel landed
    f0 = f1; r0 = r1;
    // (the fetch overwrites av[]: it must stay below the last use of the old values, or the new ones
    // get registers of their own and every step ends with a wait for the loads and 14 copies.  Loads
    // are not tied to a scheduling barrier when the instructions are selected; a memory clobber is.)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pos = pos_next;
    pid = pid_next;
    FZ_FETCH_ROWS(pos);
    f1 = f2; r1 = r2;
    pos_next = pos_next2;
    pid_next = r1 < nr ? FZ_FETCH_PID(f1) : 0;
    if (r2 < nr) FZ_ADVANCE(f2, r2);
    pos_next2 = r2 < nr ? FZ_FETCH_POS(f2, FZ_LIM(r2)) : -1;
    FZ_STAMP(5);  // fetches of the next step issued
#include "ba_fused_step_b.inc"
    FZ_STAMP(6);  // matrix phase issued
  } while (++step < nsteps);
#include "ba_fused_flush.inc"
  }
  FZ_STAMP_FLUSH;
#undef FZ_FETCH_POS
#undef FZ_FETCH_PID
#undef FZ_FETCH_ROWS
#undef FZ_ADVANCE
#undef FZ_LIM
#undef a_
#undef FZ_RESIDUAL_IN_ROWS
}

// ---- the same assembly straight from the state (rows of A: ba_rows.h) ----------------------------
// k_fused_runs with the rows of A and the residuals formed in the kernel from the camera records,
// the points and x_true: A, B and x_pred never exist in memory.  The records and projection
// matrices of the 8 cameras of the current run's window sit in LDS (4 352 bytes, reloaded when the
// next step belongs to another run); the point and x_true of the next step are fetched at the top of
// a step and turned into its operands behind the panel stores, where k_fused_runs issues its gathers.
template <int W, int T, int E, int WPS, int LPP>
__global__ __launch_bounds__(64, WPS) void k_fused_state(
    const int32_t* __restrict__ rng_run, const int32_t* __restrict__ run_pt0, const int32_t* __restrict__ run_cam,
    const int32_t* __restrict__ slot_obs, const int32_t* __restrict__ fz_pt, const double* __restrict__ blk,
    const double* __restrict__ P,
    const double* __restrict__ X, const double* __restrict__ xt, double mu, double* __restrict__ Vinv,
    double* __restrict__ eps_b, double* __restrict__ part, int pmax, int n_pts, int m) {
  static_assert(16 * T + E >= 7 * W && 16 * T <= 7 * W + 15, "tiles and extra rows must cover the window");
  static_assert(LPP == 8 || LPP == 16, "8 or 16 lanes per point");
  static_assert(E == 0 || LPP == 8, "extra rows are wired for 8 lanes per point");
  constexpr int PPS = 64 / LPP;
  constexpr int KS = 8 * LPP;
  constexpr int NK = 3 * PPS / 4;
  constexpr int NT = T * (T + 1) / 2;
  constexpr int64_t STRIDE = fz_part_stride_c(T, E, LPP);
  constexpr int REC = SFM_BA_CAM_BLK;
  constexpr int RS = REC + 2;  // record stride in LDS: 58 doubles = 116 dwords, the cameras' 16-byte reads hit different bank groups
  // the lane keeps its camera's record in registers for the whole run (one wave per SIMD, 8 lanes per point;
  // with 16 lanes per point the accumulator tiles leave no room for it)
  constexpr bool REGS = WPS == 1 && LPP == 8;
  __shared__ __attribute__((aligned(16))) double Zp[3 * PPS * KS];
  __shared__ __attribute__((aligned(16))) double sRec[LPP * (RS + 12)];  // LPP records, then LPP projection matrices
  __shared__ int bnd[kFzMaxRunsPerRange + 1];
#define FZ_RESIDUAL_IN_ROWS true
  const int lane = threadIdx.x;
  const int p = lane / LPP, s = lane % LPP;
  const int li = lane & 15, lk = lane >> 4;
  int roff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int rho = 16 * t + li;
    roff[t] = rho < 7 * W ? 8 * (rho / 7) + rho % 7 : 7;
  }
  FZ_STAMP_DECL;
  const int run0 = rng_run[blockIdx.x];
  const int nr = rng_run[blockIdx.x + 1] - run0;
  int f0 = blockIdx.x * pmax, r0 = 0;
#define FZ_FETCH_POS(first, lim) (((first) + p) < (lim) ? slot_obs[(unsigned)(LPP * ((first) + p) + s)] : -1)
#define FZ_ADVANCE(first, r)                                                    \
  do {                                                                          \
    (first) += PPS;                                                             \
    if ((first) >= __builtin_amdgcn_readfirstlane(bnd[(r) + 1])) {              \
      (r) += 1;                                                                 \
      (first) = __builtin_amdgcn_readfirstlane(bnd[(r)]);                       \
    }                                                                           \
  } while (0)
#define FZ_LIM(r) __builtin_amdgcn_readfirstlane(bnd[(r) + 1])
  // records and projection matrices of the window of run `rr` (relative to run0) into LDS (the plan
  // lists a camera for every slot: slots without one repeat the first, they hold no observation)
#define FZ_LOAD_WINDOW(rr)                                                                         \
  do {                                                                                             \
    const int32_t* wc_ = run_cam + LPP * (int64_t)(run0 + (rr));                                   \
    const double2* src_ = reinterpret_cast<const double2*>(blk);                                   \
    double2* dst_ = reinterpret_cast<double2*>(sRec);                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < (LPP * REC / 2 + 63) / 64; ++i_) {                     \
      const int e_ = lane + 64 * i_;                                                               \
      if (e_ < LPP * REC / 2) {                                                                    \
        const int c_ = wc_[e_ / (REC / 2)];                                                        \
        dst_[(e_ / (REC / 2)) * (RS / 2) + e_ % (REC / 2)] = src_[(int64_t)c_ * (REC / 2) + e_ % (REC / 2)]; \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < (6 * LPP + 63) / 64; ++i_) {                           \
      const int e_ = lane + 64 * i_;                                                               \
      if (e_ < 6 * LPP) {                                                                          \
        const int c_ = wc_[e_ / 6];                                                                \
        dst_[LPP * RS / 2 + e_] = reinterpret_cast<const double2*>(P)[(int64_t)c_ * 6 + e_ % 6];   \
      }                                                                                            \
    }                                                                                              \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                         \
    __builtin_amdgcn_wave_barrier();                                                               \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                         \
  } while (0)
  // point (id pid_) and pixel of this lane's observation (lanes without one: the step's point all the
  // same - it exists - and the first pixel)
#define FZ_FETCH_PID(first) fz_pt[(unsigned)(((first) + p) < n_pts ? (first) + p : n_pts - 1)]
#define FZ_FETCH_STATE(pid_, pos_, X0, X1, X2, XT)                         \
  do {                                                                     \
    const double* xp_ = X + 3 * (int64_t)(pid_);                           \
    X0 = xp_[0];                                                           \
    X1 = xp_[1];                                                           \
    X2 = xp_[2];                                                           \
    XT = ldg2(xt + 2 * (int64_t)((pos_) >= 0 ? (pos_) : 0));               \
  } while (0)
  // With one wave per SIMD (512 registers) the lane keeps its camera's record and projection matrix
  // in registers for the whole run: they change with the window only.  With two waves per SIMD they
  // are read from LDS at every step.
  double2 rreg[REGS ? 28 : 1], preg[REGS ? 6 : 1];
#define FZ_RECORD_TO_REGS()                                                                                 \
  do {                                                                                                      \
    if (REGS) {                                                                                             \
      const double2* r_ = reinterpret_cast<const double2*>(sRec + RS * s);                                  \
      const double2* q_ = reinterpret_cast<const double2*>(sRec + LPP * RS + 12 * s);                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 28; ++i_) rreg[REGS ? i_ : 0] = r_[i_];                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) preg[REGS ? i_ : 0] = q_[i_];                        \
    }                                                                                                       \
  } while (0)
#define FZ_ROWS(X0, X1, X2, XT, HAS)                                                                        \
  do {                                                                                                      \
    if (REGS)                                                                                               \
      fz_rows_from_state<true, false>(rreg, preg, X0, X1, X2, XT, HAS, av);                                 \
    else                                                                                                    \
      fz_rows_from_state<true, true>(reinterpret_cast<const double2*>(sRec + RS * s),                     \
                                     reinterpret_cast<const double2*>(sRec + LPP * RS + 12 * s), X0, X1, X2, XT, HAS, av); \
  } while (0)
  const int pt_first = f0 + p < n_pts ? f0 + p : n_pts - 1;
  const int pos_raw = slot_obs[(unsigned)(LPP * pt_first + s)];
  int pid = fz_pt[(unsigned)pt_first];
  for (int i = lane; i <= nr; i += 64) bnd[i] = run_pt0[run0 + i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double2 av[9];
#define a_(i) (((i) & 1) ? av[(i) >> 1].y : av[(i) >> 1].x)
  int pos = f0 + p < FZ_LIM(0) ? pos_raw : -1;
  int r_win = 0;  // run whose window sits in LDS
  FZ_LOAD_WINDOW(0);
  FZ_RECORD_TO_REGS();
  double xn0, xn1, xn2;
  double2 xtn;
  FZ_FETCH_STATE(pid, pos, xn0, xn1, xn2, xtn);
  FZ_ROWS(xn0, xn1, xn2, xtn, pos >= 0);
  int f1 = f0, r1 = r0;
  FZ_ADVANCE(f1, r1);
  int pos_next = r1 < nr ? FZ_FETCH_POS(f1, FZ_LIM(r1)) : -1;
  int pid_next = r1 < nr ? FZ_FETCH_PID(f1) : 0;
  int f2 = f1, r2 = r1;
  if (r2 < nr) FZ_ADVANCE(f2, r2);
  int pos_next2 = r2 < nr ? FZ_FETCH_POS(f2, FZ_LIM(r2)) : -1;
  int pid_next2 = r2 < nr ? FZ_FETCH_PID(f2) : 0;

  d4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = d4_t{0.0, 0.0, 0.0, 0.0};
  double u[28], ev[7], ssq = 0.0, xa[E > 0 ? E : 1][7];
#pragma unroll
  for (int i = 0; i < 28; ++i) u[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) ev[i] = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int i = 0; i < 7; ++i) xa[e][i] = 0.0;

  FZ_STAMP(0);
  while (r0 < nr) {  // runs of the range
  const int run = run0 + r0;
  int nsteps = (FZ_LIM(r0) - f0 + PPS - 1) / PPS;
  asm volatile("" : "+s"(nsteps));
  int step = 0;
  do {
    const int pt = f0 + p;
    const int pt1 = FZ_LIM(r0);
    // the next step's point and pixel: on their way during this step's arithmetic (with two waves per
    // SIMD the five values would not fit into the registers across it: fetched where they are used,
    // and the other wave covers the wait)
    if (WPS == 1 && r1 < nr) FZ_FETCH_STATE(pid_next, pos_next, xn0, xn1, xn2, xtn);
    FZ_STAMP(1);
#include "ba_fused_step_a.inc"
    FZ_STAMP(3);
    FZ_STAMP_WAIT("lgkmcnt(0)");
    FZ_STAMP(4);
    f0 = f1; r0 = r1;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pos = pos_next;
    pid = pid_next;
    if (r0 < nr) {
      if (WPS != 1) FZ_FETCH_STATE(pid, pos, xn0, xn1, xn2, xtn);
      if (r0 != r_win) {
        FZ_LOAD_WINDOW(r0);
        FZ_RECORD_TO_REGS();
        r_win = r0;
      }
      FZ_ROWS(xn0, xn1, xn2, xtn, pos >= 0);
    }
    f1 = f2; r1 = r2;
    pos_next = pos_next2;
    pid_next = pid_next2;
    if (r2 < nr) FZ_ADVANCE(f2, r2);
    pos_next2 = r2 < nr ? FZ_FETCH_POS(f2, FZ_LIM(r2)) : -1;
    pid_next2 = r2 < nr ? FZ_FETCH_PID(f2) : 0;
    FZ_STAMP(5);
#include "ba_fused_step_b.inc"
    FZ_STAMP(6);
  } while (++step < nsteps);
#include "ba_fused_flush.inc"
  }
  FZ_STAMP_FLUSH;
#undef FZ_FETCH_POS
#undef FZ_FETCH_PID
#undef FZ_FETCH_STATE
#undef FZ_ROWS
#undef FZ_RECORD_TO_REGS
#undef FZ_LOAD_WINDOW
#undef FZ_ADVANCE
#undef FZ_LIM
#undef a_
#undef FZ_RESIDUAL_IN_ROWS
}

// Back-substitution straight from the state (src/sparse_ba.py:279-290 and the point update of
// src/bundle_adjustment.py:100): delta_b_i = V_i^-1 (eps_b_i - sum_s B_s^T A_s delta_a[cam_s]) with the
// rows of A formed in the kernel as in k_fused_state (B = minus the camera-centre columns of A).
// One wave per range of positions, a step of 64 / LPP points at a time; the window's camera records and
// its entries of delta_a sit in LDS.  X is read before it is updated: a point's rows depend on no other point.
template <int LPP>
__global__ __launch_bounds__(64) void k_back_substitute_state(
    const int32_t* __restrict__ rng_run, const int32_t* __restrict__ run_pt0, const int32_t* __restrict__ run_cam,
    const int32_t* __restrict__ slot_obs, const int32_t* __restrict__ fz_pt, const double* __restrict__ blk,
    const double* __restrict__ Vinv, const double* __restrict__ eps_b, const double* __restrict__ da,
    double* __restrict__ db, double* __restrict__ X, int update, double x_step) {
  constexpr int REC = SFM_BA_CAM_BLK;
  constexpr int PPS = 64 / LPP;
  __shared__ __attribute__((aligned(16))) double sRec[LPP * REC];
  __shared__ double sDa[7 * LPP];
  const int lane = threadIdx.x;
  const int p = lane / LPP, s = lane % LPP;
  const int run_first = rng_run[blockIdx.x], run_last = rng_run[blockIdx.x + 1];
  for (int run = run_first; run < run_last; ++run) {
    const int beg = __builtin_amdgcn_readfirstlane(run_pt0[run]), end = __builtin_amdgcn_readfirstlane(run_pt0[run + 1]);
    {
      const int32_t* wc = run_cam + LPP * (int64_t)run;
      const double2* src = reinterpret_cast<const double2*>(blk);
      double2* dst = reinterpret_cast<double2*>(sRec);
#pragma unroll
      for (int i = 0; i < (LPP * REC / 2 + 63) / 64; ++i) {
        const int e = lane + 64 * i;
        if (e < LPP * REC / 2) dst[e] = src[(int64_t)wc[e / (REC / 2)] * (REC / 2) + e % (REC / 2)];
      }
#pragma unroll
      for (int i = 0; i < (7 * LPP + 63) / 64; ++i) {
        const int e = lane + 64 * i;
        if (e < 7 * LPP) sDa[e] = da[7 * (int64_t)wc[e / 7] + e % 7];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the point id of a step is fetched two steps ahead, its table entry and its point one step ahead, V^-1
    // and eps_b of the step at its top: nothing below waits for a load it has just issued
    int pos_n, pid_n, pid_n2;
    double xn0, xn1, xn2;
#define FZ_BS_PID(first) fz_pt[(unsigned)((first) + p < end ? (first) + p : end - 1)]
#define FZ_BS_FETCH(first, pid_)                                                          \
  do {                                                                                    \
    const int ptn_ = (first) + p < end ? (first) + p : end - 1;                           \
    pos_n = (first) + p < end ? slot_obs[(unsigned)(LPP * ptn_ + s)] : -1;                \
    const double* xq_ = X + 3 * (int64_t)(pid_);                                          \
    xn0 = xq_[0];                                                                         \
    xn1 = xq_[1];                                                                         \
    xn2 = xq_[2];                                                                         \
  } while (0)
    pid_n = FZ_BS_PID(beg);
    pid_n2 = FZ_BS_PID(beg + PPS);
    FZ_BS_FETCH(beg, pid_n);
    for (int f = beg; f < end; f += PPS) {
      const bool live = f + p < end;
      const int pos = pos_n;
      const int pid = pid_n;
      const double x0 = xn0, x1 = xn1, x2 = xn2;
      pid_n = pid_n2;
      if (f + PPS < end) {
        FZ_BS_FETCH(f + PPS, pid_n);
        pid_n2 = FZ_BS_PID(f + 2 * PPS);
      }
      double vv[9], eb[3];
      if (s == 0 && live) {
        const double* v = Vinv + 9 * (int64_t)pid;
#pragma unroll
        for (int i = 0; i < 9; ++i) vv[i] = v[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) eb[i] = eps_b[3 * (int64_t)pid + i];
      }
      double2 av[9];
      fz_rows_from_state<false>(reinterpret_cast<const double2*>(sRec + REC * s), static_cast<const double2*>(nullptr), x0,
                                x1, x2, make_double2(0.0, 0.0), pos >= 0, av);
#define a_(i) (((i) & 1) ? av[(i) >> 1].y : av[(i) >> 1].x)
      const double* dj = sDa + 7 * s;
      double h0 = 0.0, h1 = 0.0;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        h0 += a_(q) * dj[q];
        h1 += a_(7 + q) * dj[q];
      }
      const double c0 = sum_point<LPP>(-a_(4) * h0 - a_(11) * h1);
      const double c1 = sum_point<LPP>(-a_(5) * h0 - a_(12) * h1);
      const double c2 = sum_point<LPP>(-a_(6) * h0 - a_(13) * h1);
#undef a_
      if (s == 0 && live) {
        const double r0 = eb[0] - c0, r1 = eb[1] - c1, r2 = eb[2] - c2;
        const double d0 = vv[0] * r0 + vv[1] * r1 + vv[2] * r2;
        const double d1 = vv[3] * r0 + vv[4] * r1 + vv[5] * r2;
        const double d2 = vv[6] * r0 + vv[7] * r1 + vv[8] * r2;
        db[3 * (int64_t)pid] = d0; db[3 * (int64_t)pid + 1] = d1; db[3 * (int64_t)pid + 2] = d2;
        if (update) {
          X[3 * (int64_t)pid] = x0 + x_step * d0;
          X[3 * (int64_t)pid + 1] = x1 + x_step * d1;
          X[3 * (int64_t)pid + 2] = x2 + x_step * d2;
        }
      }
    }
#undef FZ_BS_FETCH
#undef FZ_BS_PID
    // the next run's window overwrites the records: every lane is done reading them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// partial results of the runs of one window (they are consecutive), added in run order
__global__ __launch_bounds__(256) void k_fused_reduce(const int32_t* __restrict__ key_run, int64_t stride,
                                                      const double* __restrict__ part, double* __restrict__ kpart) {
  const int key = blockIdx.x;
  const int beg = key_run[key], end = key_run[key + 1];
  for (int64_t el = (int64_t)blockIdx.y * 256 + threadIdx.x; el < stride; el += (int64_t)gridDim.y * 256) {
    double sum = 0.0;
    int i = beg;
    for (; i + 4 <= end; i += 4) {
      const double v0 = part[(int64_t)i * stride + el];
      const double v1 = part[(int64_t)(i + 1) * stride + el];
      const double v2 = part[(int64_t)(i + 2) * stride + el];
      const double v3 = part[(int64_t)(i + 3) * stride + el];
      sum = (((sum + v0) + v1) + v2) + v3;
    }
    for (; i < end; ++i) sum += part[(int64_t)i * stride + el];
    kpart[(int64_t)key * stride + el] = sum;
  }
}

// one 64-lane workgroup per camera pair (j <= k) that shares a point: add the windows that hold both
// cameras (the plan lists them with the cameras' slots, ascending) and then the partial results the far
// plan's camera-pair kernel left for the pair, write the block of S below the diagonal (diagonal blocks
// in full, with U_j and the damping), and for diagonal blocks e_j
__global__ __launch_bounds__(64) void k_fused_finish(const int32_t* __restrict__ blk_j, const int32_t* __restrict__ blk_k,
                                                     const int32_t* __restrict__ fb_ptr, const int32_t* __restrict__ fb_ent,
                                                     const int32_t* __restrict__ fb_far_part,
                                                     const int32_t* __restrict__ fb_far_cnt, int slots0, int slots1,
                                                     const double* __restrict__ kpart0, const double* __restrict__ kpart1,
                                                     const double* __restrict__ far_part, double mu_u, int m,
                                                     double* __restrict__ S, double* __restrict__ e_out,
                                                     double* __restrict__ ssq_out) {
  // threads 0..48: the 7 x 7 block; 49..55: e_j, 56: the camera's sum of squared residuals (diagonal blocks)
  const int t = threadIdx.x;
  const int j = blk_j[blockIdx.x], k = blk_k[blockIdx.x];
  if (t >= 57 || (t >= 49 && j != k)) return;
  const int r = t / 7, c = t - 7 * r;
  const int ulo = r < c ? r : c, uhi = r < c ? c : r;
  const int uidx = ulo * 7 - ulo * (ulo - 1) / 2 + (uhi - ulo);
  double sum = 0.0;
  const int e_end = fb_ptr[blockIdx.x + 1];
  for (int en = fb_ptr[blockIdx.x]; en < e_end; ++en) {
    const int ent = fb_ent[en];
    const int cls = (ent >> 30) & 1, sj = (ent >> 4) & 15, sk = ent & 15;
    const int slots = cls ? slots1 : slots0;
    const int T = fz_row_tiles(slots), lanes = fz_lanes(slots);
    const int nt = T * (T + 1) / 2;
    const double* pb = (cls ? kpart1 : kpart0) + (int64_t)((ent >> 8) & 0x3FFFFF) * fz_part_stride(slots);
    if (t < 49) {
      // (Y_j W_k^T)[r][c] = D[7 sk + c][7 sj + r]; D is symmetric and its lower triangle is stored
      // (a diagonal block takes both of its triangles from the lower one)
      int row = 7 * sk + c, col = 7 * sj + r;
      if (col > row) {
        const int tmp = row;
        row = col;
        col = tmp;
      }
      double v;
      if (row < 16 * T) {
        const int ti = row >> 4, tj = col >> 4;
        v = pb[(ti * (ti + 1) / 2 + tj) * 256 + (row & 15) * 16 + (col & 15)];
      } else {
        v = pb[nt * 256 + (kFzSlotSums + 7 * (row - 16 * T) + col % 7) * lanes + col / 7];
      }
      if (j == k) v = pb[nt * 256 + uidx * lanes + sj] - v;
      sum += v;
    } else {
      sum += pb[nt * 256 + (28 + (t - 49)) * lanes + sj];  // (t = 56: value 35, the squared residuals)
    }
  }
  // the far points' share: sum A_j^T T A_k over their entries of the pair, which on the diagonal holds
  // sum Y W^T - U_j (see k_point), and sum A^T g in slots 49..55, sum |eps|^2 in slot 56 of a diagonal
  // block's partials
  {
    const int cnt = fb_far_cnt[blockIdx.x];
    const double* fp = far_part + (int64_t)fb_far_part[blockIdx.x] * kPairPartStride + t;
    double far = 0.0;
    for (int i = 0; i < cnt; ++i) far += fp[(int64_t)i * kPairPartStride];
    if (cnt > 0) sum = (t >= 49 || j != k) ? sum + far : sum - far;
  }
  const int64_t ld = 7 * (int64_t)m;
  if (t == 56) {
    if (ssq_out) ssq_out[j] = sum;
  } else if (t >= 49) {
    S[ld * ld + 7 * j + (t - 49)] = sum;
    if (e_out) e_out[7 * j + (t - 49)] = sum;
  } else if (j == k) {
    S[(7 * (int64_t)j + r) * ld + 7 * j + c] = sum + (r == c ? mu_u : 0.0);
  } else {
    S[(7 * (int64_t)k + c) * ld + 7 * j + r] = -sum;
  }
}

// Far points of a plan whose far points have no observation at all (ids nobody names, allowed on plans
// built with size hints; with observations among them the far plan's kernels do this): V_i = mu I,
// eps_b_i = 0 (src/sparse_ba.py:201-212, 163-171 with an empty sum), and delta_b_i = 0.
__global__ void k_far_unobserved(const int32_t* __restrict__ far_pt, int n_far, double mu, double* __restrict__ Vinv,
                                 double* __restrict__ eps_b, double* __restrict__ db) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_far) return;
  const int64_t i = far_pt[q];
  if (Vinv) {
    const double d = 1.0 / mu;
    double* v = Vinv + 9 * i;
    v[0] = d; v[1] = 0; v[2] = 0; v[3] = 0; v[4] = d; v[5] = 0; v[6] = 0; v[7] = 0; v[8] = d;
    eps_b[3 * i] = 0; eps_b[3 * i + 1] = 0; eps_b[3 * i + 2] = 0;
  }
  if (db) {
    db[3 * i] = 0; db[3 * i + 1] = 0; db[3 * i + 2] = 0;
  }
}

// rows of A, observed and predicted pixels of the far observations, gathered into the far plan's order
__global__ __launch_bounds__(256) void k_far_gather(const int32_t* __restrict__ far_obs, int64_t nf,
                                                    const double* __restrict__ A, const double* __restrict__ xt,
                                                    const double* __restrict__ xp, double* __restrict__ A_far,
                                                    double* __restrict__ xt_far, double* __restrict__ xp_far) {
  // 9 pieces of 16 bytes per observation: 7 of the row of A, x_true, x_pred
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= 9 * nf) return;
  const int64_t i = g / 9;
  const int q = (int)(g - 9 * i);
  const int64_t o = far_obs[i];
  if (q < 7) {
    if (A) *reinterpret_cast<double2*>(A_far + 14 * i + 2 * q) = ldg2(A + 14 * o + 2 * q);
  } else if (q == 7) {
    *reinterpret_cast<double2*>(xt_far + 2 * i) = ldg2(xt + 2 * o);
  } else if (xp) {
    *reinterpret_cast<double2*>(xp_far + 2 * i) = ldg2(xp + 2 * o);
  }
}

static inline int check_launch_fz(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, what);
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

#ifdef SFMBA_FZ_STAMPS
static void fz_print_stamps() {
  unsigned long long h[16];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fz_stamps), sizeof(h));
  fprintf(stderr, "fz stamps (previous launch): setup %llu operands %llu valu %llu z %llu panel %llu fetch %llu mfma %llu tiles out %llu sums parked %llu added %llu rest %llu\n",
          h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[8], h[9], h[10], h[7]);
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fz_stamps), h, sizeof(h));
}
#endif
// reduce and finish shared by the two kernels that produce the partial results
static int launch_fused_tail(const sfm_ba_plan* p, double mu_u, double* S, double* e, double* ssq, char* ws,
                             const double* far_part, hipStream_t st);

bool fz_state_applies(const sfm_ba_plan* p) {
  if (p->n_runs <= 0 || !p->fz_enabled) return false;
  for (int c = 0; c < kFzClasses; ++c)
    if (p->fz[c].max_range_runs > kFzMaxRunsPerRange) return false;
  return true;
}

static int launch_far_unobserved(const sfm_ba_plan* p, double mu, double* Vinv, double* eps_b, double* db,
                                 hipStream_t st) {
  if (p->far || p->n_far <= 0) return SFM_BA_OK;
  hipLaunchKernelGGL(k_far_unobserved, dim3((p->n_far + 255) / 256), dim3(256), 0, st, p->d_far_pt, p->n_far, mu, Vinv,
                     eps_b, db);
  return check_launch_fz("k_far_unobserved");
}

// The far points' share of the assembly: V^-1, eps_b of those points, g and T in the far plan's
// workspace, and the partial results of its camera-pair blocks, which k_fused_finish adds into S.
// A_far, xt_far, xp_far: rows of A and pixels in the far plan's observation order.
static int launch_far_assemble(const sfm_ba_plan* p, const double* A_far, const double* xt_far, const double* xp_far,
                               double mu, double* Vinv, double* eps_b, char* ws, hipStream_t st) {
  if (!p->far) return SFM_BA_OK;
  return launch_assemble_parts(p->far, p->d_far_pt, A_far, xt_far, xp_far, mu, Vinv, eps_b, ws + p->ws_far, st);
}

static const double* far_partials(const sfm_ba_plan* p, const char* ws) {
  return p->far ? reinterpret_cast<const double*>(ws + p->ws_far + p->far->ws_spart) : nullptr;
}

// waves per SIMD of k_fused_state for a window width: with windows of up to six cameras the kernel fits 256
// registers (what it spills is only reloaded when the window changes) and two waves cover each other's
// waits (364 against 387 us on 1 800 x 1 M); from seven cameras on it spills inside the steps, and one wave
// with 512 registers - the camera record held in them for the whole run - is the faster one (142 against 147 us)
static int state_waves_per_simd(int slots) {
  static const int wps_env = [] {
    const char* v = getenv("SFM_BA_FZ_STATE_WPS");
    return v && (v[0] == '1' || v[0] == '2') ? v[0] - '0' : 0;
  }();
  return wps_env ? wps_env : (slots <= 6 ? 2 : 1);
}

int launch_assemble_state(const sfm_ba_plan* p, const double* blk, const double* P, const double* X, const double* xt,
                          double mu, double mu_u, double* S, double* e, double* Vinv, double* eps_b, double* ssq, void* ws,
                          hipStream_t st) {
  if (!fz_state_applies(p)) return SFM_BA_EINVAL;
  char* base = static_cast<char*>(ws);
  if (p->far) {
    // far points: their rows of A and predictions from the state (the Jacobian kernel on the far plan's
    // observation list), their observed pixels gathered beside them
    double* A_far = reinterpret_cast<double*>(base + p->ws_far_A);
    double* xt_far = reinterpret_cast<double*>(base + p->ws_far_xt);
    double* xp_far = reinterpret_cast<double*>(base + p->ws_far_xp);
    int rc = launch_jacobian(blk, P, X, p->d_far_cam, p->d_far_gpt, p->n_far_obs, A_far, nullptr, xp_far, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_far_gather, dim3((unsigned)((9 * p->n_far_obs + 255) / 256)), dim3(256), 0, st, p->d_far_obs,
                       p->n_far_obs, static_cast<const double*>(nullptr), xt, static_cast<const double*>(nullptr), A_far,
                       xt_far, xp_far);
    rc = check_launch_fz("k_far_gather");
    if (rc) return rc;
    rc = launch_far_assemble(p, A_far, xt_far, xp_far, mu, Vinv, eps_b, base, st);
    if (rc) return rc;
  } else {
    const int rc = launch_far_unobserved(p, mu, Vinv, eps_b, nullptr, st);
    if (rc) return rc;
  }
  for (int c = 0; c < kFzClasses; ++c) {
    const FzClass& f = p->fz[c];
    if (f.n_runs <= 0) continue;
    double* part = reinterpret_cast<double*>(base + f.ws_part);
    {
      ProfScope ps(KID_FUSED_STATE, st);
      const int wps_state = state_waves_per_simd(f.slots);
#define FZ_ARGS dim3(f.n_ranges), dim3(64), 0, st, f.d_rng_run, f.d_run_pt0, f.d_run_cam, f.d_slot_obs, f.d_pt, blk, P, X, xt, \
                mu, Vinv, eps_b, part, f.pmax, f.n_near, p->m
#define FZ_LAUNCH_STATE(W, T, E)                                       \
  do {                                                                 \
    if (wps_state == 2)                                                \
      hipLaunchKernelGGL((k_fused_state<W, T, E, 2, 8>), FZ_ARGS);     \
    else                                                               \
      hipLaunchKernelGGL((k_fused_state<W, T, E, 1, 8>), FZ_ARGS);     \
  } while (0)
      switch (f.slots) {
        case 4: FZ_LAUNCH_STATE(4, 2, 0); break;
        case 5: FZ_LAUNCH_STATE(5, 3, 0); break;
        case 6: FZ_LAUNCH_STATE(6, 3, 0); break;
        case 7: FZ_LAUNCH_STATE(7, 3, 1); break;
        case 8: FZ_LAUNCH_STATE(8, 4, 0); break;
        default: hipLaunchKernelGGL((k_fused_state<16, 7, 0, 1, 16>), FZ_ARGS); break;
      }
#undef FZ_LAUNCH_STATE
#undef FZ_ARGS
    }
#ifdef SFMBA_FZ_STAMPS
    fz_print_stamps();
#endif
    const int rc = check_launch_fz("k_fused_state");
    if (rc) return rc;
  }
  if (!S) return SFM_BA_OK;
  return launch_fused_tail(p, mu_u, S, e, ssq, base, far_partials(p, base), st);
}

int launch_back_substitute_state(const sfm_ba_plan* p, const double* blk, const double* Vinv, const double* eps_b,
                                 const double* da, double* db, double* X, int update, double x_step, void* ws,
                                 hipStream_t st) {
  if (p->n_runs <= 0) return SFM_BA_EINVAL;
  for (int c = 0; c < kFzClasses; ++c) {
    const FzClass& f = p->fz[c];
    if (f.n_runs <= 0) continue;
    {
      ProfScope ps(KID_BACK_STATE, st);
      if (f.lanes == 8)
        hipLaunchKernelGGL(k_back_substitute_state<8>, dim3(f.n_ranges), dim3(64), 0, st, f.d_rng_run, f.d_run_pt0, f.d_run_cam,
                           f.d_slot_obs, f.d_pt, blk, Vinv, eps_b, da, db, X, update, x_step);
      else
        hipLaunchKernelGGL(k_back_substitute_state<16>, dim3(f.n_ranges), dim3(64), 0, st, f.d_rng_run, f.d_run_pt0,
                           f.d_run_cam, f.d_slot_obs, f.d_pt, blk, Vinv, eps_b, da, db, X, update, x_step);
    }
    const int rc = check_launch_fz("k_back_substitute_state");
    if (rc) return rc;
  }
  if (!p->far) return launch_far_unobserved(p, 1.0, nullptr, nullptr, db, st);
  // far points: from the rows the assembly left in the workspace
  const double* A_far = reinterpret_cast<const double*>(static_cast<char*>(ws) + p->ws_far_A);
  return launch_back_substitute_mapped(p->far, p->d_far_pt, A_far, Vinv, eps_b, da, db, update ? X : nullptr, x_step, st);
}

int launch_assemble_fused(const sfm_ba_plan* p, const double* A, const double* xt, const double* xp, double mu,
                          double mu_u, double* S, double* e, double* Vinv, double* eps_b, void* ws, hipStream_t st) {
  if (p->n_runs <= 0) return SFM_BA_OK;
  char* base = static_cast<char*>(ws);
  if (p->far) {
    double* A_far = reinterpret_cast<double*>(base + p->ws_far_A);
    double* xt_far = reinterpret_cast<double*>(base + p->ws_far_xt);
    double* xp_far = reinterpret_cast<double*>(base + p->ws_far_xp);
    hipLaunchKernelGGL(k_far_gather, dim3((unsigned)((9 * p->n_far_obs + 255) / 256)), dim3(256), 0, st, p->d_far_obs,
                       p->n_far_obs, A, xt, xp, A_far, xt_far, xp_far);
    int rc = check_launch_fz("k_far_gather");
    if (rc) return rc;
    rc = launch_far_assemble(p, A_far, xt_far, xp_far, mu, Vinv, eps_b, base, st);
    if (rc) return rc;
  } else {
    const int rc = launch_far_unobserved(p, mu, Vinv, eps_b, nullptr, st);
    if (rc) return rc;
  }
  // waves per SIMD the kernel is compiled for: two (256 registers each) hide memory latency better,
  // unless the variant then spills - a reload from scratch memory waits for every load in flight
  static const int wps_env = [] {
    const char* v = getenv("SFM_BA_FZ_WPS");
    return v && (v[0] == '1' || v[0] == '2') ? v[0] - '0' : 0;
  }();
  const int wps = wps_env ? wps_env : 2;
  for (int c = 0; c < kFzClasses; ++c) {
    const FzClass& f = p->fz[c];
    if (f.n_runs <= 0) continue;
    double* part = reinterpret_cast<double*>(base + f.ws_part);
    {
      ProfScope ps(KID_FUSED_RUNS, st);
#define FZ_ARGS dim3(f.n_ranges), dim3(64), 0, st, f.d_rng_run, f.d_run_pt0, f.d_slot_obs, f.d_pt, A, xt, xp, mu, Vinv, eps_b, \
                part, f.pmax, f.n_near
#define FZ_LAUNCH(W, T, E)                                          \
  do {                                                              \
    if (wps == 1)                                                   \
      hipLaunchKernelGGL((k_fused_runs<W, T, E, 1, 8>), FZ_ARGS);   \
    else                                                            \
      hipLaunchKernelGGL((k_fused_runs<W, T, E, 2, 8>), FZ_ARGS);   \
  } while (0)
      switch (f.slots) {
        case 4: FZ_LAUNCH(4, 2, 0); break;
        case 5: FZ_LAUNCH(5, 3, 0); break;
        case 6: FZ_LAUNCH(6, 3, 0); break;
        case 7: FZ_LAUNCH(7, 3, 1); break;
        case 8: FZ_LAUNCH(8, 4, 0); break;
        default: hipLaunchKernelGGL((k_fused_runs<16, 7, 0, 1, 16>), FZ_ARGS); break;
      }
#undef FZ_LAUNCH
#undef FZ_ARGS
    }
#ifdef SFMBA_FZ_STAMPS
    fz_print_stamps();
#endif
    const int rc = check_launch_fz("k_fused_runs");
    if (rc) return rc;
  }
  if (!S) return SFM_BA_OK;
  return launch_fused_tail(p, mu_u, S, e, nullptr, base, far_partials(p, base), st);
}

static int launch_fused_tail(const sfm_ba_plan* p, double mu_u, double* S, double* e, double* ssq, char* ws,
                             const double* far_part, hipStream_t st) {
  const double* kpart[kFzClasses] = {nullptr, nullptr};
  for (int c = 0; c < kFzClasses; ++c) {
    const FzClass& f = p->fz[c];
    if (f.n_runs <= 0) continue;
    double* part = reinterpret_cast<double*>(ws + f.ws_part);
    double* kp = reinterpret_cast<double*>(ws + f.ws_key);
    kpart[c] = kp;
    const int64_t stride = fz_part_stride(f.slots);
    {
      ProfScope ps(KID_FUSED_REDUCE, st);
      hipLaunchKernelGGL(k_fused_reduce, dim3(f.n_keys, c == 0 ? 3 : 8), dim3(256), 0, st, f.d_key_run, stride, part, kp);
    }
    const int rc = check_launch_fz("k_fused_reduce");
    if (rc) return rc;
  }
  if (p->n_blocks > 0) {
    ProfScope ps(KID_FUSED_FINISH, st);
    hipLaunchKernelGGL(k_fused_finish, dim3(p->n_blocks), dim3(64), 0, st, p->d_blk_j, p->d_blk_k, p->d_fb_ptr,
                       p->d_fb_ent, p->d_fb_far_part, p->d_fb_far_cnt, p->fz[0].slots, p->fz[1].slots, kpart[0], kpart[1],
                       far_part, mu_u, p->m, S, e, ssq);
  }
  return check_launch_fz("k_fused_finish");
}

}  // namespace sfmba
