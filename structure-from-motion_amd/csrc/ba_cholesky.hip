// Reduced camera system S x = e, S symmetric positive definite (n x n, row-major, leading dimension
// n, lower triangle referenced), right-hand side in row n of the same buffer.  Three paths share one
// 64 x 64 diagonal-block factorisation (diag_core):
//
// General path - blocked right-looking Cholesky, NB = 64, in place.  The right-hand side is carried
// as row n, so the forward substitution L y = e happens as part of the panel solves; then a blocked
// backward substitution L^T x = y.
//   k_chol_diag       : first panel only: one workgroup factors and inverts the 64 x 64 diagonal block.
//   k_chol_step       : one launch per panel (ba_chol_step.inc): the rows below the block times
//                       inv(L_kk)^T, the trailing update A_ij -= P_i P_j^T and the next diagonal
//                       factorisation as roles of one kernel                               [MFMA]
//                       All-zero 64-row tiles of the factor are flagged and skipped: S is block sparse
//                       whenever far-apart cameras share no point, and the factor keeps those zeros
//                       outside the profile.
//   k_chol_back_group : four panels of L^T x = y per launch, last panel first.
//
// Narrow band (no non-zero further than 64 places from the diagonal; S is then block tridiagonal
// in 64 x 64 blocks and is only read):
//   k_bcr_*       : block cyclic reduction - log2(n / 64) rounds of parallel eliminations.
//   k_chol_band   : fewer than four blocks - one workgroup walks the chain in a single launch.
//
// Tile products use v_mfma_f64_16x16x4_f64 with operands staged in LDS at a row stride of 66 doubles
// (conflict-free for the one-double-per-lane operand reads).
#include "ba_common.h"

namespace sfmba {

constexpr int NB = 64;   // panel width = tile edge
constexpr int LDT = 66;  // LDS row stride in doubles
constexpr int SB = 16;   // sub-block edge inside the diagonal kernel

typedef double d4_t __attribute__((ext_vector_type(4)));

// Diagnostic builds (scripts/probes/diag_stamps.hip) define SFMBA_STAMPS to time the phases of the
// diagonal factorisation; the library build compiles the stamps away.
#ifdef SFMBA_STAMPS
__device__ long long g_stamps[48];
#define STAMP(i)                                                   \
  do {                                                             \
    if (threadIdx.x == 0) g_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define STAMP(i) \
  do {           \
  } while (0)
#endif

// workspace: [n_panels][NB*NB] inverse diagonal factors, then y[n_panels * NB + 64] (running
// right-hand side of the substitutions), then either one int per (panel, row tile below it) - 1 when
// that 64 x 64 tile of the factor is entirely zero (general path) - or [n_panels][NB*NB]
// sub-diagonal factor tiles (band path)
int wide_band_tiles(int n, int half_bandwidth);
int64_t wide_workspace_doubles(int n, int C);

int64_t cholesky_workspace_bytes(int n) {
  const int64_t np = (n + NB - 1) / NB;
  // widest superblock the wide-band reduction may pick for this n (it wants four superblocks)
  int64_t wide = 0;
  for (int C = 2; C <= 4; C *= 2)
    if (wide_band_tiles(n, NB * C) == C) wide = wide_workspace_doubles(n, C) * (int64_t)sizeof(double);
  // zero-tile flags, then the six tiles the look-ahead of the one-launch-per-panel path passes on
  const int64_t flags = (np * (np + 2) + 64) * (int64_t)sizeof(int32_t) + 6 * NB * NB * (int64_t)sizeof(double) + 8;
  const int64_t tiles = np * NB * NB * (int64_t)sizeof(double);
  const int64_t general = (np * NB * NB + np * NB + 64) * (int64_t)sizeof(double) + (flags > tiles ? flags : tiles);
  const int64_t reduction = (4 * np * NB * NB + np * NB + 64) * (int64_t)sizeof(double);  // D, B, P_u, P_d, e per block
  const int64_t narrow = general > reduction ? general : reduction;
  return narrow > wide ? narrow : wide;
}

// value of `v` in lane `lane` (wave-uniform lane index)
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// Factor the 16 x 16 lower-triangular-stored block whose row r sits in lane r (a[0..15], entries
// above the diagonal ignored); lanes >= 16 may hold further rows of the same 16 columns, which
// come out forward-substituted (row * L^-T).  On return a[] holds row r of L (zeros above the
// diagonal for r < 16).  Returns the index+1 of the first non-positive pivot or 0.
//
// Columns are eliminated in pairs so that the two reciprocal square roots of a pair - the long
// dependent part of a column step - run side by side: with the leading 2 x 2 block [a00 a10; a10 a11]
// and det = a00 a11 - a10^2 the second pivot is det / a00, so 1/sqrt(pivot_1) = rsqrt(det) sqrt(a00).
// det is formed with the product errors compensated (fma residuals), which keeps it as accurate
// as the pivot of a column-by-column elimination.
__device__ __forceinline__ int factor16(double (&a)[SB], int r) {
  int bad = 0;
#pragma unroll
  for (int j = 0; j < SB; j += 2) {
    const double a00 = lane_bcast(a[j], j), a10 = lane_bcast(a[j], j + 1), a11 = lane_bcast(a[j + 1], j + 1);
    const double w = a10 * a10;
    const double e = fma(a10, a10, -w);
    const double f = fma(a11, a00, -w);
    const double det = f - e;
    if (!(a00 > 0.0) && bad == 0) bad = j + 1;
    if (!(det > 0.0) && bad == 0) bad = j + 2;
    const double rs0 = rsqrt(a00), rdet = rsqrt(det);
    const double l00 = a00 * rs0;
    const double rs1 = rdet * l00;
    const double l10 = a10 * rs0;
    const double l0 = a[j] * rs0;
    const double l1 = fma(-l0, l10, a[j + 1]) * rs1;
    a[j] = l0;
    a[j + 1] = l1;
#pragma unroll
    for (int c = j + 2; c < SB; ++c) {
      a[c] = fma(-l0, lane_bcast(l0, c), a[c]);
      a[c] = fma(-l1, lane_bcast(l1, c), a[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < SB; ++c)
    if (c > r) a[c] = 0.0;
  return bad;
}

// Factor the 64 x 64 block held in LDS (L, lower triangle, identity-padded past nb), invert the
// factor (X, any contents on entry), store the factor into S (unless S is null) and the lower triangle of the inverse into
// invL (entries above the diagonal of invL are never written: readers mask them).  256 threads.
//
// Phase kk = 16-column block column kk.  Waves 0 and 1 share one copy of the register
// factorisation: wave 0 holds one row of block column kk per lane (rows 16 kk .. 63) - the lanes
// below the diagonal block see exactly the same column operations, which is the forward
// substitution that finishes their rows - and wave 1 holds the diagonal block and the identity,
// which comes out as the transposed inverse of the diagonal block at no extra latency.  Between
// two factorisations only the three tiles of the next block column take a rank-16 MFMA update;
// everything else (the other tiles' updates, block row kk of the inverse
// X_kj = -X_kk sum_{q=j}^{kk-1} L_kq X_qj, the stores of finished block columns / rows) is done by
// waves 2 and 3 while waves 0 and 1 factor the next block column.
struct NoShadow {
  __device__ __forceinline__ void operator()(int, int) const {}
};

// `shadow(kk, tt)` is run by waves 2 and 3 (tt = 0..127) at the start of phase kk, while waves 0 and
// 1 factor block column kk: these two waves have slack in every phase (none of their own work in
// phase 0), so callers hang independent work there.
template <class Shadow = NoShadow>
__device__ __forceinline__ void diag_core(double* __restrict__ L, double* __restrict__ X, double* __restrict__ S,
                                          int n, int k0, int nb, double* __restrict__ invL,
                                          int32_t* __restrict__ info, Shadow shadow = Shadow()) {
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int NS = NB / SB;
  const int li = lane & 15, lk = lane >> 4;

  // tile (bi, bc) -= L[rows of bi, block column kc] L[rows of bc, block column kc]^T
  auto upd_tile = [&](int bi, int bc, int kc, bool dup = false) {
    d4_t c4;
    double av[4], bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) c4[g] = L[(bi * SB + lk + 4 * g) * LDT + bc * SB + li];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      av[q] = -L[(bi * SB + li) * LDT + kc * SB + 4 * q + lk];
      bv[q] = L[(bc * SB + li) * LDT + kc * SB + 4 * q + lk];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], c4, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) L[(bi * SB + lk + 4 * g) * LDT + bc * SB + li] = c4[g];
    if (dup) {
#pragma unroll
      for (int g = 0; g < 4; ++g) X[(bi * SB + lk + 4 * g) * LDT + bc * SB + li] = c4[g];
    }
  };
  // dst tile = -X_(bi,bi) sum_{q = q_lo}^{q_hi - 1} L[rows of bi, 4q..4q+3] X[4q..4q+3, cols of bj].  The
  // inner sum leaves the accumulator in exactly the lane layout the second product wants for its
  // B operand (row 4 s + (lane >> 4) in register s): no LDS round trip between the two products.
  auto x_job = [&](int bi, int bj, int q_lo, int q_hi, double* dst) {
    constexpr int QMAX = (NS - 1) * SB / 4;
    const int cnt = q_hi - q_lo;  // wave-uniform
    double av[QMAX], bv[QMAX], xv[4];
#pragma unroll
    for (int i = 0; i < QMAX; ++i) {
      const int q = q_lo + i;
      av[i] = i < cnt ? L[(bi * SB + li) * LDT + 4 * q + lk] : 0.0;
      bv[i] = i < cnt ? X[(4 * q + lk) * LDT + bj * SB + li] : 0.0;
    }
#pragma unroll
    for (int sft = 0; sft < 4; ++sft) xv[sft] = -X[(bi * SB + li) * LDT + bi * SB + 4 * sft + lk];
    d4_t m = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < QMAX; ++i)
      if (i < cnt) m = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], m, 0, 0, 0);
    d4_t xo = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sft = 0; sft < 4; ++sft) xo = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[sft], m[sft], xo, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) dst[(lk + 4 * g) * LDT + li] = xo[g];
  };
  // one wave: block column kc of the factor (rows 16 kc .. 63) to S
  auto store_l_col = [&](int kc) {
    const int cnt = (NB - kc * SB) * SB;
    for (int idx = lane; idx < cnt; idx += 64) {
      const int r = kc * SB + idx / SB, c = kc * SB + idx % SB;
      if (S != nullptr && r < nb && c <= r) S[(int64_t)(k0 + r) * n + k0 + c] = L[r * LDT + c];
    }
  };
  // one wave: block row br of the inverse (columns 0 .. 16 br + 15) to invL
  auto store_x_row = [&](int br) {
    const int w = (br + 1) * SB;
    for (int idx = lane; idx < SB * w; idx += 64) {
      const int r = br * SB + idx / w, c = idx % w;
      if (c <= r) invL[r * NB + c] = X[r * LDT + c];
    }
  };

  __syncthreads();
  // Wave 1 factors a copy of the diagonal 16 x 16 block of every phase, and wave 0 stores the factored
  // block column over the original inside the same phase: wave 1 must not read it from there (nothing
  // orders its loads before wave 0's stores - seen going wrong about once in 20 000 calls when the waves
  // of a workgroup drift apart).  It reads the block from X's diagonal block of the phase instead, which
  // holds nothing until wave 1 itself writes the block's inverse there: the first one is copied here, the
  // later ones are left there by the update that completes them (upd_tile with dup).
  if (t < SB * SB) X[(t / SB) * LDT + t % SB] = L[(t / SB) * LDT + t % SB];
  __syncthreads();
  STAMP(0);
  for (int kk = 0; kk < NS; ++kk) {
    const int o = kk * SB;
    STAMP(1 + 4 * kk);
    if (wave < 2) {
      const int row = o + lane;
      const bool from_l = wave == 0 ? row < NB : lane < SB;
      double a[SB];
#pragma unroll
      for (int c = 0; c < SB; ++c) {
        a[c] = from_l ? (wave == 0 ? L : X)[row * LDT + o + c] : 0.0;
        if (wave == 1 && lane - SB == c) a[c] = 1.0;
      }
      STAMP(20 + kk);
      const int bad = factor16(a, lane);
      STAMP(24 + kk);
      if (wave == 0) {
        if (row < NB) {
#pragma unroll
          for (int c = 0; c < SB; ++c) L[row * LDT + o + c] = a[c];
        }
        if (bad && lane == 0 && o + bad <= nb) atomicCAS(info, 0, k0 + o + bad);
      } else if (lane >= SB && lane < 2 * SB) {
        // lane 16 + q holds row q of inv(L_kk)^T
#pragma unroll
        for (int c = 0; c < SB; ++c) X[(o + c) * LDT + o + lane - SB] = a[c];
      }
    } else {
      shadow(kk, t - 128);
    }
    if (wave >= 2 && kk > 0) {
      // in the shadow of the factorisation: what the previous block column left to do
      const int pk = kk - 1;
      if (wave == 2) {
        if (pk == 0) {
          upd_tile(2, 2, 0);
          upd_tile(3, 3, 0);
        } else if (pk == 1) {
          upd_tile(3, 3, 1);
          x_job(1, 0, 0, 4, X + (1 * SB) * LDT);
        } else {
          x_job(2, 0, 0, 8, X + (2 * SB) * LDT);
        }
      } else {
        if (pk == 0) upd_tile(3, 2, 0);
        if (pk == 2) x_job(2, 1, 4, 8, X + (2 * SB) * LDT + SB);
        store_l_col(pk);
        if (pk >= 1) store_x_row(pk - 1);
      }
    }
    STAMP(2 + 4 * kk);
    __syncthreads();  // block column kk of L and X_kk are final
    STAMP(3 + 4 * kk);
    if (kk < NS - 1 && wave < NS - 1 - kk) upd_tile(kk + 1 + wave, kk + 1, kk, wave == 0);
    STAMP(4 + 4 * kk);
    if (kk < NS - 1) __syncthreads();
  }
  // block row 3 of the inverse; the long sum of X_30 is split over two waves, the second half parked
  // in the (structurally zero, never stored) block (0, 3) of X
  if (wave == 0) {
    x_job(3, 0, 0, 6, X + (3 * SB) * LDT);
  } else if (wave == 1) {
    x_job(3, 1, 4, 12, X + (3 * SB) * LDT + SB);
  } else if (wave == 2) {
    x_job(3, 2, 8, 12, X + (3 * SB) * LDT + 2 * SB);
  } else {
    x_job(3, 0, 6, 12, X + 3 * SB);
  }
  STAMP(17);
  __syncthreads();
  // fold the parked half of X_30 in and clear its parking place: X is then the complete inverse
  // (lower blocks; blocks above the diagonal are never touched by this function)
  {
    const int r = t / SB, c = t % SB;
    X[(3 * SB + r) * LDT + c] += X[r * LDT + 3 * SB + c];
    X[r * LDT + 3 * SB + c] = 0.0;
  }
  __syncthreads();
  STAMP(18);
  {
    // what is left to store: block rows 2 and 3 of the inverse, the last diagonal block of the factor
    double xv[8], lv;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = t + 256 * it, r = 2 * SB + idx / NB, c = idx % NB;
      xv[it] = X[r * LDT + c];
    }
    const int lr = 3 * SB + t / SB, lc = 3 * SB + t % SB;
    lv = L[lr * LDT + lc];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = t + 256 * it, r = 2 * SB + idx / NB, c = idx % NB;
      if (c <= r) invL[r * NB + c] = xv[it];
    }
    if (S != nullptr && lr < nb && lc <= lr) S[(int64_t)(k0 + lr) * n + k0 + lc] = lv;
  }
  STAMP(19);
}

// first panel only: later diagonal blocks are factored by the chain workgroup of k_chol_step that
// produced them
__global__ __launch_bounds__(256) void k_chol_diag(double* __restrict__ S, int n, int k0, int nb,
                                                   double* __restrict__ invL, int32_t* __restrict__ info) {
  __shared__ double L[NB * LDT];
  __shared__ double X[NB * LDT];
  const int t = threadIdx.x;
  {
    // all 16 loads of a lane are issued before the first LDS store (one memory latency, not 16)
    double v[NB * NB / 256];
#pragma unroll
    for (int it = 0; it < NB * NB / 256; ++it) {
      const int idx = t + 256 * it, r = idx / NB, c = idx % NB;
      v[it] = (r == c) ? 1.0 : 0.0;  // identity padding keeps the padded factor well defined
      if (r < nb && c < nb) v[it] = (c <= r) ? S[(int64_t)(k0 + r) * n + k0 + c] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < NB * NB / 256; ++it) {
      const int idx = t + 256 * it, r = idx / NB, c = idx % NB;
      L[r * LDT + c] = v[it];
    }
  }
  diag_core(L, X, S, n, k0, nb, invL, info);
}

// 64 x 64 x 64 product on the matrix cores: acc (2 x 2 MFMA tiles of this wave's 32 x 32 quadrant)
// += sign * Pa[rows] Pb[cols]^T with both operands stored [row][k] in LDS.
__device__ __forceinline__ void mfma_quadrant(const double* __restrict__ Pa, const double* __restrict__ Pb,
                                              int wr, int wc, int lane, bool negate, d4_t (&acc)[2][2]) {
  const int lr = lane & 15, lk = lane >> 4;
  const double* pa0 = Pa + (wr * 32 + lr) * LDT + lk;
  const double* pa1 = pa0 + 16 * LDT;
  const double* pb0 = Pb + (wc * 32 + lr) * LDT + lk;
  const double* pb1 = pb0 + 16 * LDT;
#pragma unroll 4
  for (int kk = 0; kk < NB; kk += 4) {
    double a0 = pa0[kk], a1 = pa1[kk];
    const double b0 = pb0[kk], b1 = pb1[kk];
    if (negate) {
      a0 = -a0;
      a1 = -a1;
    }
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}

#include "ba_chol_step.inc"

// Backward substitution L^T x = y, a group of up to four panels kTop, kTop - 1, ... in one launch: every workgroup walks the
// group's own chain (x of the panel, its contribution to the next panel's right-hand side, ...) - a few
// 64 x 64 matrix-vector products, done by all workgroups alike - and then takes the group's contribution
// off its own 64 columns of y.  A quarter of the launches of the panel-by-panel form; the factor's tiles
// of the workgroup's own columns are fetched before the chain starts.
constexpr int kBackGroup = 4;
__global__ __launch_bounds__(256) void k_chol_back_group(const double* __restrict__ S, int n, int kTop, int m,
                                                         const double* __restrict__ invL_all, double* __restrict__ y,
                                                         double* __restrict__ x, const int32_t* __restrict__ zflags,
                                                         int np) {
  __shared__ double xs[kBackGroup][NB];
  __shared__ double ys[NB];
  __shared__ double part[4][NB];
  const int t = threadIdx.x;
  const int col = t & 63, rq = t >> 6;
  const int j = blockIdx.x;
  const bool has_cols = j <= kTop - m;  // a column tile below the group (workgroup 0 always runs: it stores x)
  // Everything the chain reads from memory is requested here, before the first sum: none of the addresses
  // depends on x (few workgroups, one per CU: registers are not what limits this kernel).
  //   lv: own columns - rows of the group's panels, column j * 64 + col
  //   iv: column `col` of each panel's (lower-triangular) inverse, rows rq, rq + 4, ...
  //   gv: the group's own tiles - rows of panel kTop - sp, column `col` of panel kTop - s, sp < s
  //   yv: the panels' entries of y
  double lv[kBackGroup][NB / 4], iv[kBackGroup][NB / 4], gv[kBackGroup * (kBackGroup - 1) / 2][NB / 4], yv[kBackGroup];
#pragma unroll
  for (int s = 0; s < kBackGroup; ++s) {
    const int kg = kTop - s, k0 = kg * NB;
    const bool in = s < m;
    const int nb = in ? ((n - k0) < NB ? (n - k0) : NB) : 0;
    const bool on = has_cols && in && zflags[(int64_t)j * (np + 1) + (kg - j - 1)] == 0;
    const double* invL = invL_all + (int64_t)(in ? kg : 0) * NB * NB;
#pragma unroll
    for (int i = 0; i < NB / 4; ++i) {
      const int r = rq + 4 * i;
      lv[s][i] = (on && r < nb) ? S[(int64_t)(k0 + r) * n + j * NB + col] : 0.0;
      iv[s][i] = (r < nb && col <= r) ? invL[r * NB + col] : 0.0;
    }
    yv[s] = (t < nb) ? y[k0 + t] : 0.0;
#pragma unroll
    for (int sp = 0; sp < s; ++sp) {
      const int kp = kTop - sp, k0p = kp * NB;
      const int nbp = (n - k0p) < NB ? (n - k0p) : NB;
      const bool gon = in && zflags[(int64_t)kg * (np + 1) + (kp - kg - 1)] == 0;
#pragma unroll
      for (int i = 0; i < NB / 4; ++i) {
        const int r = rq + 4 * i;
        gv[s * (s - 1) / 2 + sp][i] = (gon && r < nbp && col < nb) ? S[(int64_t)(k0p + r) * n + k0 + col] : 0.0;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < kBackGroup; ++s) {
    if (s < m) {
      const int k0 = (kTop - s) * NB;
      const int nb = (n - k0) < NB ? (n - k0) : NB;
      // right-hand side of this panel: y minus what the panels above it in the group contribute
      double acc = 0.0;
#pragma unroll
      for (int sp = 0; sp < s; ++sp)
#pragma unroll
        for (int i = 0; i < NB / 4; ++i) acc += gv[s * (s - 1) / 2 + sp][i] * xs[sp][rq + 4 * i];
      part[rq][col] = acc;
      __syncthreads();
      if (t < NB) ys[t] = yv[s] - ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t]));
      __syncthreads();
      double s2 = 0.0;
#pragma unroll
      for (int i = 0; i < NB / 4; ++i) s2 += iv[s][i] * ys[rq + 4 * i];
      part[rq][col] = s2;
      __syncthreads();
      if (t < NB) {
        const double v = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        xs[s][t] = v;
        if (j == 0 && t < nb) x[k0 + t] = v;
      }
      __syncthreads();
    }
  }
  if (!has_cols) return;
  double a = 0.0;
#pragma unroll
  for (int s = 0; s < kBackGroup; ++s)
    if (s < m) {
#pragma unroll
      for (int i = 0; i < NB / 4; ++i) a += lv[s][i] * xs[s][rq + 4 * i];
    }
  part[rq][col] = a;
  __syncthreads();
  if (t < NB) y[j * NB + t] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

// ---- band path -------------------------------------------------------------------------------
// When S has no non-zero further than 64 places from its diagonal (cameras share points only with
// near neighbours in index), each panel has one non-zero 64 x 64 tile below its diagonal block and
// the factorisation is a chain with no parallel slack at all: panel k+1 cannot start before panel k
// is done.  One workgroup then walks the whole chain without leaving the chip: the diagonal block
// lives in LDS from its update to its factorisation, the sub-diagonal tile and the next diagonal
// block are fetched by the two waves that idle at the start of each diagonal factorisation, and
// both substitutions run in the same launch.  S is only read.
//
// The two 64^3 products per panel (P = B inv(L)^T and D' -= P P^T) are done per 16 x 16 block on the
// matrix cores, skipping blocks that are structurally zero: inv(L) is lower triangular, and of the
// sub-diagonal tile B only the blocks near its upper right corner are non-zero when the band is
// narrower than 64 (a bit mask of the non-zero blocks of B is taken while it is loaded).
//   invL_ws [np][64*64] inverse diagonal factors (lower triangle), p_ws [np][64*64] sub-diagonal
//   factor tiles, y_ws [np*64] forward-substituted right-hand side.

// c += (negate ? -1 : 1) * A[block row I, block col kb] B[block row J, block col kb]^T summed over the
// block columns kb < kb_end whose bit is set in both ma and mb
__device__ __forceinline__ void block_prod(const double* __restrict__ A, int I, unsigned ma,
                                           const double* __restrict__ B, int J, unsigned mb, int kb_end,
                                           bool negate, int li, int lk, d4_t& c) {
  constexpr int NS = NB / SB;
  double av[NS][4], bv[NS][4];
  const unsigned on = ma & mb & ((1u << kb_end) - 1u);
#pragma unroll
  for (int kb = 0; kb < NS; ++kb)
    if ((on >> kb) & 1u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double a = A[(I * SB + li) * LDT + kb * SB + 4 * q + lk];
        av[kb][q] = negate ? -a : a;
        bv[kb][q] = B[(J * SB + li) * LDT + kb * SB + 4 * q + lk];
      }
    }
#pragma unroll
  for (int kb = 0; kb < NS; ++kb)
    if ((on >> kb) & 1u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kb][q], bv[kb][q], c, 0, 0, 0);
    }
}

__global__ __launch_bounds__(256) void k_chol_band(const double* __restrict__ S, int n, double* __restrict__ invL_ws,
                                                   double* __restrict__ p_ws, double* __restrict__ y_ws,
                                                   double* __restrict__ x_out, int32_t* __restrict__ info) {
  constexpr int NS = NB / SB;
  __shared__ double Dbuf[2][NB * LDT];  // diagonal block -> its factor; the other one receives the next block
  __shared__ double X[NB * LDT];        // inverse of the factor
  __shared__ double Bt[NB * LDT];       // sub-diagonal tile of S -> of the factor
  __shared__ double tv[NB];             // right-hand side of the current panel minus what earlier panels took
  __shared__ double yv[NB];
  __shared__ double en[NB];
  __shared__ double part[4][NB];
  __shared__ unsigned bmask_sh;         // bit 4 I + K: block (I, K) of the sub-diagonal tile has a non-zero
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int np = (n + NB - 1) / NB;
  const double* rhs = S + (int64_t)n * n;
  for (int idx = t; idx < NB * LDT; idx += 256) X[idx] = 0.0;
  {
    const int nb = n < NB ? n : NB;
    double v[NB * NB / 256];
#pragma unroll
    for (int it = 0; it < NB * NB / 256; ++it) {
      const int idx = t + 256 * it, r = idx / NB, c = idx % NB;
      v[it] = (r == c) ? 1.0 : 0.0;  // identity padding keeps the padded factor well defined
      if (r < nb && c < nb) v[it] = (c <= r) ? S[(int64_t)r * n + c] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < NB * NB / 256; ++it) {
      const int idx = t + 256 * it, r = idx / NB, c = idx % NB;
      Dbuf[0][r * LDT + c] = v[it];
    }
    if (t < NB) tv[t] = t < nb ? rhs[t] : 0.0;
    if (t == 0) bmask_sh = 0u;
  }
  for (int k = 0; k < np; ++k) {
    double* D = Dbuf[k & 1];
    double* Dn = Dbuf[(k + 1) & 1];
    const int k0 = k * NB;
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    const bool has_next = k + 1 < np;
    const int k1 = k0 + NB;
    const int nb1 = has_next ? ((n - k1) < NB ? (n - k1) : NB) : 0;
    // waves 2 and 3, while waves 0 and 1 factor the first block column: the tile below the diagonal
    // block (with the mask of its non-zero 16 x 16 blocks), the next diagonal block, the next piece
    // of the right-hand side
    auto fetch_next = [&](int kk, int tt) {
      if (!has_next || kk > 1) return;
      double v[NB * NB / 128];
      if (kk == 0) {
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) {
          const int idx = tt + 128 * it, r = idx / NB, c = idx % NB;
          v[it] = r < nb1 ? S[(int64_t)(k1 + r) * n + k0 + c] : 0.0;
        }
        unsigned m = 0u;
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) {
          const int idx = tt + 128 * it, r = idx / NB, c = idx % NB;
          Bt[r * LDT + c] = v[it];
          // a wave holds one row here: 16 lanes per block column
          const unsigned long long nz = __ballot(v[it] != 0.0);
          const int I = r / SB;
#pragma unroll
          for (int kb = 0; kb < NS; ++kb)
            if ((nz >> (16 * kb)) & 0xffffull) m |= 1u << (4 * I + kb);
        }
        if (lane == 0 && m) atomicOr(&bmask_sh, m);
      } else {
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) {
          const int idx = tt + 128 * it, r = idx / NB, c = idx % NB;
          v[it] = (r == c) ? 1.0 : 0.0;
          if (r < nb1 && c < nb1) v[it] = (c <= r) ? S[(int64_t)(k1 + r) * n + k1 + c] : 0.0;
        }
        const double ev = (tt < nb1) ? rhs[k1 + tt] : 0.0;
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) {
          const int idx = tt + 128 * it, r = idx / NB, c = idx % NB;
          Dn[r * LDT + c] = v[it];
        }
        if (tt < NB) en[tt] = ev;
      }
    };
    diag_core(D, X, nullptr, n, k0, nb, invL_ws + (int64_t)k * NB * NB, info, fetch_next);
    // forward substitution of this panel: y_k = inv(L_kk) tv
    {
      const int r = t & 63, q = t >> 6;
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) sacc += X[r * LDT + 16 * q + c] * tv[16 * q + c];
      part[q][r] = sacc;
    }
    __syncthreads();
    if (t < NB) {
      const double yk = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
      yv[t] = yk;
      y_ws[k0 + t] = yk;
    }
    if (!has_next) break;
    // P = B inv(L_kk)^T.  Wave w owns the four blocks (I, (w - I) mod 4); block (I, J) sums over block
    // columns kb <= J (inv(L) is lower triangular) in which B has a non-zero block.
    const unsigned bmask = bmask_sh;
    // structural mask of P: block (I, J) is non-zero if B has a non-zero block (I, kb) with kb <= J
    unsigned pmask = 0u;
#pragma unroll
    for (int I = 0; I < NS; ++I)
#pragma unroll
      for (int J = 0; J < NS; ++J)
        if ((bmask >> (4 * I)) & ((2u << J) - 1u)) pmask |= 1u << (4 * I + J);
    d4_t pacc[NS];
#pragma unroll
    for (int I = 0; I < NS; ++I) {
      const int J = (wave - I) & 3;
      pacc[I] = d4_t{0.0, 0.0, 0.0, 0.0};
      block_prod(Bt, I, (bmask >> (4 * I)) & 15u, X, J, 15u, J + 1, false, li, lk, pacc[I]);
    }
    __syncthreads();  // every wave is done reading Bt (and yv is visible)
    double* pk = p_ws + (int64_t)k * NB * NB;
#pragma unroll
    for (int I = 0; I < NS; ++I) {
      const int J = (wave - I) & 3;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = I * SB + lk + 4 * g, c = J * SB + li;
        Bt[r * LDT + c] = pacc[I][g];
        pk[r * NB + c] = pacc[I][g];
      }
    }
    if (t == 0) bmask_sh = 0u;  // for the next panel's fetch (a barrier away from its first use)
    __syncthreads();
    // next diagonal block -= P P^T, lower blocks only: wave w owns the blocks (I, J), J <= I, with
    // (I + J) mod 4 = w; next tv = e - P y
#pragma unroll
    for (int I = 0; I < NS; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J)
        if (((I + J) & 3) == wave) {
          const unsigned on = (pmask >> (4 * I)) & (pmask >> (4 * J)) & 15u;
          if (on) {
            d4_t c4 = {0.0, 0.0, 0.0, 0.0};
            block_prod(Bt, I, on, Bt, J, on, NS, true, li, lk, c4);
#pragma unroll
            for (int g = 0; g < 4; ++g) Dn[(I * SB + lk + 4 * g) * LDT + J * SB + li] += c4[g];
          }
        }
    {
      const int r = t & 63, q = t >> 6;
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) sacc += Bt[r * LDT + 16 * q + c] * yv[16 * q + c];
      part[q][r] = sacc;
    }
    __syncthreads();
    if (t < NB) tv[t] = en[t] - ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t]));
    // (diag_core opens with a workgroup barrier: Dn and tv are visible to every wave after it)
  }
  __syncthreads();
  // backward substitution, last panel first: x_k = inv(L_kk)^T (y_k - P_k^T x_{k+1}); yv holds y of the
  // last panel already.  Thread (col = t & 63, rq = t >> 6) sums rows rq, rq+4, ... of its column.
  const int col = t & 63, rq = t >> 6;
  double xi[NB / 4], pv[NB / 4];
  {
    const double* il = invL_ws + (int64_t)(np - 1) * NB * NB;
#pragma unroll
    for (int i = 0; i < NB / 4; ++i) {
      const int r = rq + 4 * i;
      xi[i] = col <= r ? il[r * NB + col] : 0.0;
      pv[i] = 0.0;
    }
  }
  for (int k = np - 1; k >= 0; --k) {
    const int k0 = k * NB;
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    // z = y_k - P_k^T x_{k+1}   (tv holds x_{k+1}; pv is zero for the last panel)
    double sacc = 0.0;
#pragma unroll
    for (int i = 0; i < NB / 4; ++i) sacc += pv[i] * tv[rq + 4 * i];
    part[rq][col] = sacc;
    __syncthreads();
    if (t < NB) yv[t] = yv[t] - ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t]));
    __syncthreads();
    sacc = 0.0;
#pragma unroll
    for (int i = 0; i < NB / 4; ++i) sacc += xi[i] * yv[rq + 4 * i];
    // tiles of the next step are fetched before this step's last reduction
    double yn = 0.0;
    if (k > 0) {
      const double* il = invL_ws + (int64_t)(k - 1) * NB * NB;
      const double* pl = p_ws + (int64_t)(k - 1) * NB * NB;
#pragma unroll
      for (int i = 0; i < NB / 4; ++i) {
        const int r = rq + 4 * i;
        xi[i] = col <= r ? il[r * NB + col] : 0.0;
        pv[i] = pl[r * NB + col];
      }
      if (t < NB) yn = y_ws[k0 - NB + t];
    }
    __syncthreads();  // every thread has read part[] of the first product
    part[rq][col] = sacc;
    __syncthreads();
    if (t < NB) {
      const double xk = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
      tv[t] = xk;
      if (t < nb) x_out[k0 + t] = xk;
      yv[t] = yn;
    }
    __syncthreads();
  }
}

// ---- band path, parallel form: block cyclic reduction -----------------------------------------
// The same band (no non-zero further than 64 places from the diagonal) makes S block tridiagonal in
// 64 x 64 blocks: diagonal blocks D_i and couplings B_i = S(block i+1, block i).  Eliminating every
// second block at once leaves a block tridiagonal system in the remaining half - a Cholesky
// factorisation in odd-even order - so the chain of n/64 dependent diagonal factorisations becomes
// log2(n/64) rounds, each factoring its blocks on separate compute units.  At stride s the active
// blocks are 0, s, 2s, ...; block i = j s with odd j is eliminated:
//     D_i = L_i L_i^T,  y_i = inv(L_i) e_i,
//     P_u = S(u, i) inv(L_i)^T  (u = i - s),   P_d = S(d, i) inv(L_i)^T  (d = i + s),
// and the remaining blocks take
//     D_u -= P_u P_u^T,  D_d -= P_d P_d^T,  e_u -= P_u y_i,  e_d -= P_d y_i,  S(d, u) = -P_d P_u^T.
// Backwards, last round first:  x_i = inv(L_i)^T (y_i - P_u^T x_u - P_d^T x_d).
// Workspace per block: D (then the inverse of its factor), B, P_u, P_d (64 x 64 each), e (64).

// c += sign * A[block row I, cols] B[block row J, cols]^T over block columns kb < kb_end.  With
// trans_a the first operand is read transposed: A^T[block row I] from a tile stored [k][row].
__device__ __forceinline__ void tile_mma(const double* __restrict__ A, bool trans_a, int I,
                                         const double* __restrict__ B, int J, int kb_end, bool negate,
                                         int li, int lk, d4_t& c) {
  constexpr int NS = NB / SB;
  double av[NS][4], bv[NS][4];
#pragma unroll
  for (int kb = 0; kb < NS; ++kb)
    if (kb < kb_end) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kcol = kb * SB + 4 * q + lk;
        const double a = trans_a ? A[kcol * LDT + I * SB + li] : A[(I * SB + li) * LDT + kcol];
        av[kb][q] = negate ? -a : a;
        bv[kb][q] = B[(J * SB + li) * LDT + kcol];
      }
    }
#pragma unroll
  for (int kb = 0; kb < NS; ++kb)
    if (kb < kb_end) {
#pragma unroll
      for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kb][q], bv[kb][q], c, 0, 0, 0);
    }
}

// 64 x 64 row-major tile in global memory -> LDS tile with row stride LDT (all 256 threads)
__device__ __forceinline__ void tile_to_lds(const double* __restrict__ g, double* __restrict__ l, int t) {
  double v[NB * NB / 256];
#pragma unroll
  for (int it = 0; it < NB * NB / 256; ++it) v[it] = g[t + 256 * it];
#pragma unroll
  for (int it = 0; it < NB * NB / 256; ++it) {
    const int idx = t + 256 * it;
    l[(idx / NB) * LDT + idx % NB] = v[it];
  }
}

// S -> block tridiagonal form: D_i (lower triangle, identity past the matrix edge), B_i, e_i
__global__ __launch_bounds__(256) void k_bcr_extract(const double* __restrict__ S, int n, int np,
                                                     double* __restrict__ Dk, double* __restrict__ Bk,
                                                     double* __restrict__ ek) {
  const int i = blockIdx.x, t = threadIdx.x;
  const int k0 = i * NB, k1 = k0 + NB;
  const int nb = (n - k0) < NB ? (n - k0) : NB;
  const int nb1 = (n - k1) < NB ? (n - k1) : NB;  // <= 0 for the last block
  double vd[NB * NB / 256], vb[NB * NB / 256];
#pragma unroll
  for (int it = 0; it < NB * NB / 256; ++it) {
    const int idx = t + 256 * it, r = idx / NB, c = idx % NB;
    vd[it] = (r == c) ? 1.0 : 0.0;
    if (r < nb && c < nb) vd[it] = (c <= r) ? S[(int64_t)(k0 + r) * n + k0 + c] : 0.0;
    vb[it] = (r < nb1) ? S[(int64_t)(k1 + r) * n + k0 + c] : 0.0;
  }
#pragma unroll
  for (int it = 0; it < NB * NB / 256; ++it) {
    Dk[(int64_t)i * NB * NB + t + 256 * it] = vd[it];
    Bk[(int64_t)i * NB * NB + t + 256 * it] = vb[it];
  }
  if (t < NB) ek[k0 + t] = t < nb ? S[(int64_t)n * n + k0 + t] : 0.0;
}

// one workgroup per eliminated block: j = j0 + 2 blockIdx, i = j s
__global__ __launch_bounds__(256) void k_bcr_eliminate(int n, int np, int s, int j0, double* __restrict__ Dk,
                                                       const double* __restrict__ Bk, double* __restrict__ Pu,
                                                       double* __restrict__ Pd, double* __restrict__ ek,
                                                       int32_t* __restrict__ info) {
  constexpr int NS = NB / SB;
  __shared__ double L[NB * LDT];
  __shared__ double X[NB * LDT];
  __shared__ double Bu[NB * LDT];  // S(i, u) = coupling stored with block u; read transposed
  __shared__ double Bd[NB * LDT];  // S(d, i) = coupling stored with block i
  __shared__ double ev[NB];
  __shared__ double part[4][NB];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int i = (j0 + 2 * (int)blockIdx.x) * s;
  const bool has_u = i - s >= 0, has_d = i + s < np;
  const int k0 = i * NB;
  const int nb = (n - k0) < NB ? (n - k0) : NB;
  double* Di = Dk + (int64_t)i * NB * NB;
  tile_to_lds(Di, L, t);
  for (int idx = t; idx < NB * LDT; idx += 256) X[idx] = 0.0;
  // the coupling tiles and the right-hand side arrive while the first block columns are factored
  auto fetch = [&](int kk, int tt) {
    if (kk > 1) return;
    const bool on = kk == 0 ? has_u : has_d;
    if (on) {
      const double* g = Bk + (int64_t)(kk == 0 ? i - s : i) * NB * NB;
      double* l = kk == 0 ? Bu : Bd;
      double v[NB * NB / 128];
#pragma unroll
      for (int it = 0; it < NB * NB / 128; ++it) v[it] = g[tt + 128 * it];
#pragma unroll
      for (int it = 0; it < NB * NB / 128; ++it) {
        const int idx = tt + 128 * it;
        l[(idx / NB) * LDT + idx % NB] = v[it];
      }
    }
    if (kk == 0 && tt < NB) ev[tt] = ek[k0 + tt];
  };
  diag_core(L, X, nullptr, n, k0, nb, Di, info, fetch);
  // y_i = inv(L_i) e_i
  {
    const int r = t & 63, q = t >> 6;
    double sacc = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) sacc += X[r * LDT + 16 * q + c] * ev[16 * q + c];
    part[q][r] = sacc;
  }
  __syncthreads();
  if (t < NB) ek[k0 + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
  // P_u = S(u, i) inv(L_i)^T = Bu^T X^T,  P_d = Bd X^T.  Wave w owns the blocks (I, (w - I) mod 4);
  // inv(L_i) is lower triangular: block column kb contributes to block (I, J) only for kb <= J.
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    if (side == 0 ? !has_u : !has_d) continue;
    const double* Bsrc = side == 0 ? Bu : Bd;
    double* Pout = (side == 0 ? Pu : Pd) + (int64_t)i * NB * NB;
#pragma unroll
    for (int I = 0; I < NS; ++I) {
      const int J = (wave - I) & 3;
      d4_t c4 = {0.0, 0.0, 0.0, 0.0};
      tile_mma(Bsrc, side == 0, I, X, J, J + 1, false, li, lk, c4);
#pragma unroll
      for (int g = 0; g < 4; ++g) Pout[(I * SB + lk + 4 * g) * NB + J * SB + li] = c4[g];
    }
  }
}

// Eight workgroups per remaining block i = 2 (blockIdx >> 3) s: task (blockIdx >> 2) & 1, block row
// q = blockIdx & 3 of the 64 x 64 result, one 16 x 16 tile per wave.  Task 0 updates D_i and e_i from
// the eliminated neighbours on both sides; task 1 forms the new coupling S(i + 2s, i).
__global__ __launch_bounds__(256) void k_bcr_update(int n, int np, int s, double* __restrict__ Dk,
                                                    double* __restrict__ Bk, const double* __restrict__ Pu,
                                                    const double* __restrict__ Pd, double* __restrict__ ek) {
  constexpr int NS = NB / SB;
  __shared__ double Ta[NB * LDT];
  __shared__ double Tb[NB * LDT];
  __shared__ double ya[NB], yb[NB];
  __shared__ double part[4][SB];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int i = 2 * ((int)blockIdx.x >> 3) * s;
  const int task = (blockIdx.x >> 2) & 1;
  const int q = blockIdx.x & 3;  // block row of the result
  const int o1 = i - s, o2 = i + s;
  const bool has1 = o1 >= 0, has2 = o2 < np;
  if (task == 1) {
    // S(i + 2s, i) = -P_d[o2] P_u[o2]^T: tile (q, wave)
    if (!has2 || i + 2 * s >= np) return;
    tile_to_lds(Pd + (int64_t)o2 * NB * NB, Ta, t);
    tile_to_lds(Pu + (int64_t)o2 * NB * NB, Tb, t);
    __syncthreads();
    double* Bout = Bk + (int64_t)i * NB * NB;
    d4_t c4 = {0.0, 0.0, 0.0, 0.0};
    tile_mma(Ta, false, q, Tb, wave, NS, true, li, lk, c4);
#pragma unroll
    for (int g = 0; g < 4; ++g) Bout[(q * SB + lk + 4 * g) * NB + wave * SB + li] = c4[g];
    return;
  }
  if (!has1 && !has2) return;
  // zero tiles stand in for a missing neighbour
  if (has1) {
    tile_to_lds(Pd + (int64_t)o1 * NB * NB, Ta, t);
  } else {
    for (int idx = t; idx < NB * LDT; idx += 256) Ta[idx] = 0.0;
  }
  if (has2) {
    tile_to_lds(Pu + (int64_t)o2 * NB * NB, Tb, t);
  } else {
    for (int idx = t; idx < NB * LDT; idx += 256) Tb[idx] = 0.0;
  }
  if (t < NB) {
    ya[t] = has1 ? ek[o1 * NB + t] : 0.0;
    yb[t] = has2 ? ek[o2 * NB + t] : 0.0;
  }
  __syncthreads();
  double* Di = Dk + (int64_t)i * NB * NB;
  // lower tiles (q, wave), wave <= q
  if (wave <= q) {
    d4_t c4;
#pragma unroll
    for (int g = 0; g < 4; ++g) c4[g] = Di[(q * SB + lk + 4 * g) * NB + wave * SB + li];
    if (has1) tile_mma(Ta, false, q, Ta, wave, NS, true, li, lk, c4);
    if (has2) tile_mma(Tb, false, q, Tb, wave, NS, true, li, lk, c4);
#pragma unroll
    for (int g = 0; g < 4; ++g) Di[(q * SB + lk + 4 * g) * NB + wave * SB + li] = c4[g];
  }
  // rows 16 q .. 16 q + 15 of e_i -= P_d[o1] y_o1 + P_u[o2] y_o2: thread (row r, column quarter cq)
  {
    const int r = t & 15, cq = (t >> 4) & 3;
    if (t < 64) {
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        sacc += Ta[(q * SB + r) * LDT + 16 * cq + c] * ya[16 * cq + c] + Tb[(q * SB + r) * LDT + 16 * cq + c] * yb[16 * cq + c];
      part[cq][r] = sacc;
    }
  }
  __syncthreads();
  if (t < SB) ek[i * NB + q * SB + t] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

// x_i = inv(L_i)^T (y_i - P_u^T x_u - P_d^T x_d) for block i with neighbours iu, id (negative: none).
// xu, xd, zv: 64 doubles of LDS each, part: 4 x 64.  All 256 threads.
__device__ __forceinline__ void bcr_back_block(int n, int i, int iu, int id, const double* __restrict__ Dk,
                                               const double* __restrict__ Pu, const double* __restrict__ Pd,
                                               double* __restrict__ ek, double* __restrict__ x_out, double* xu,
                                               double* xd, double* zv, double (*part)[NB]) {
  const int t = threadIdx.x;
  const int col = t & 63, rq = t >> 6;
  const bool has_u = iu >= 0, has_d = id >= 0;
  const int k0 = i * NB;
  const int nb = (n - k0) < NB ? (n - k0) : NB;
  // thread (col, rq) sums rows rq, rq + 4, ... of its column: all three tiles are fetched up front
  double pu[NB / 4], pd[NB / 4], xi[NB / 4];
#pragma unroll
  for (int q = 0; q < NB / 4; ++q) {
    const int r = rq + 4 * q;
    pu[q] = has_u ? Pu[(int64_t)i * NB * NB + r * NB + col] : 0.0;
    pd[q] = has_d ? Pd[(int64_t)i * NB * NB + r * NB + col] : 0.0;
    xi[q] = col <= r ? Dk[(int64_t)i * NB * NB + r * NB + col] : 0.0;  // lower triangle holds inv(L_i)
  }
  if (t < NB) {
    xu[t] = has_u ? ek[iu * NB + t] : 0.0;
    xd[t] = has_d ? ek[id * NB + t] : 0.0;
  }
  __syncthreads();
  double sacc = 0.0;
#pragma unroll
  for (int q = 0; q < NB / 4; ++q) sacc += pu[q] * xu[rq + 4 * q] + pd[q] * xd[rq + 4 * q];
  part[rq][col] = sacc;
  __syncthreads();
  if (t < NB) zv[t] = ek[k0 + t] - ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t]));
  __syncthreads();
  sacc = 0.0;
#pragma unroll
  for (int q = 0; q < NB / 4; ++q) sacc += xi[q] * zv[rq + 4 * q];
  part[rq][col] = sacc;
  __syncthreads();
  if (t < NB) {
    const double xk = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    ek[k0 + t] = xk;
    if (t < nb) x_out[k0 + t] = xk;
  }
}

// the blocks eliminated at stride s: one workgroup each
__global__ __launch_bounds__(256) void k_bcr_back(int n, int np, int s, int j0, const double* __restrict__ Dk,
                                                  const double* __restrict__ Pu, const double* __restrict__ Pd,
                                                  double* __restrict__ ek, double* __restrict__ x_out) {
  __shared__ double xu[NB], xd[NB], zv[NB];
  __shared__ double part[4][NB];
  const int i = (j0 + 2 * (int)blockIdx.x) * s;
  bcr_back_block(n, i, i - s >= 0 ? i - s : -1, i + s < np ? i + s : -1, Dk, Pu, Pd, ek, x_out, xu, xd, zv, part);
}

// The last rounds of the reduction as one launch: when two or three blocks 0, s, 2s are left, one
// workgroup walks them as a chain (factor, couple, update the next block, factor, ...; then the
// substitutions backwards).  A round of the reduction costs an elimination, an update and later a
// back-substitution launch for a single block each; the chain has the same factorisations on its
// critical path and none of the launches between them.
__global__ __launch_bounds__(256) void k_bcr_chain(int n, int np, int s, int cnt, double* __restrict__ Dk,
                                                   const double* __restrict__ Bk, double* __restrict__ Pd,
                                                   double* __restrict__ ek, double* __restrict__ x_out,
                                                   int32_t* __restrict__ info) {
  constexpr int NS = NB / SB;
  __shared__ double L[NB * LDT];
  __shared__ double X[NB * LDT];
  __shared__ double Bt[NB * LDT];  // coupling S(next block, this block)
  __shared__ double Pt[NB * LDT];  // the same times inv(L)^T
  __shared__ double ev[NB], yv[NB], xu[NB], xd[NB], zv[NB];
  __shared__ double part[4][NB];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, lk = lane >> 4;
  for (int idx = t; idx < NB * LDT; idx += 256) X[idx] = 0.0;
  for (int c = 0; c < cnt; ++c) {
    const int i = c * s;
    const int k0 = i * NB;
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    const bool has_next = c + 1 < cnt;
    double* Di = Dk + (int64_t)i * NB * NB;
    __syncthreads();  // the previous block's readers of L are done
    tile_to_lds(Di, L, t);
    if (t < NB) ev[t] = ek[k0 + t];
    __syncthreads();
    if (c > 0) {
      // D_i -= P P^T (lower tiles (q, wave), wave <= q) and e_i -= P y of the block before
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (wave <= q) {
          d4_t c4;
#pragma unroll
          for (int g = 0; g < 4; ++g) c4[g] = L[(q * SB + lk + 4 * g) * LDT + wave * SB + li];
          tile_mma(Pt, false, q, Pt, wave, NS, true, li, lk, c4);
#pragma unroll
          for (int g = 0; g < 4; ++g) L[(q * SB + lk + 4 * g) * LDT + wave * SB + li] = c4[g];
        }
      {
        const int r = t & 63, q = t >> 6;
        double sacc = 0.0;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) sacc += Pt[r * LDT + 16 * q + cc] * yv[16 * q + cc];
        part[q][r] = sacc;
      }
      __syncthreads();
      if (t < NB) ev[t] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    }
    // the coupling to the next block arrives while the first block columns are factored
    auto fetch = [&](int kk, int tt) {
      if (kk != 0 || !has_next) return;
      const double* g = Bk + (int64_t)i * NB * NB;
      double v[NB * NB / 128];
#pragma unroll
      for (int it = 0; it < NB * NB / 128; ++it) v[it] = g[tt + 128 * it];
#pragma unroll
      for (int it = 0; it < NB * NB / 128; ++it) {
        const int idx = tt + 128 * it;
        Bt[(idx / NB) * LDT + idx % NB] = v[it];
      }
    };
    diag_core(L, X, nullptr, n, k0, nb, Di, info, fetch);
    // y_i = inv(L_i) e_i
    {
      const int r = t & 63, q = t >> 6;
      double sacc = 0.0;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) sacc += X[r * LDT + 16 * q + cc] * ev[16 * q + cc];
      part[q][r] = sacc;
    }
    __syncthreads();
    if (t < NB) {
      const double y = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
      yv[t] = y;
      ek[k0 + t] = y;
    }
    if (has_next) {
      // P = S(next, i) inv(L_i)^T: wave w owns the blocks (I, (w - I) mod 4)
      d4_t c4[NS];
#pragma unroll
      for (int I = 0; I < NS; ++I) {
        const int J = (wave - I) & 3;
        c4[I] = d4_t{0.0, 0.0, 0.0, 0.0};
        tile_mma(Bt, false, I, X, J, J + 1, false, li, lk, c4[I]);
      }
      double* Pout = Pd + (int64_t)i * NB * NB;
#pragma unroll
      for (int I = 0; I < NS; ++I) {
        const int J = (wave - I) & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Pt[(I * SB + lk + 4 * g) * LDT + J * SB + li] = c4[I][g];
          Pout[(I * SB + lk + 4 * g) * NB + J * SB + li] = c4[I][g];
        }
      }
    }
  }
  __syncthreads();
  // backwards: x_i = inv(L_i)^T (y_i - P_i^T x_next); P_i sits where an eliminated block keeps its P_d
  for (int c = cnt - 1; c >= 0; --c) {
    bcr_back_block(n, c * s, -1, c + 1 < cnt ? (c + 1) * s : -1, Dk, Pd, Pd, ek, x_out, xu, xd, zv, part);
    __syncthreads();
  }
}

// packed block-tridiagonal form: [np][64*64] diagonal blocks, [np][64*64] couplings, [np*64] right-hand side
int64_t band_packed_elements(int n) {
  const int64_t np = (n + NB - 1) / NB;
  return 2 * np * NB * NB + np * NB;
}

int launch_band_pack(const double* S, int n, double* packed, hipStream_t st) {
  const int np = (n + NB - 1) / NB;
  double* Dk = packed;
  double* Bk = Dk + (int64_t)np * NB * NB;
  double* ek = Bk + (int64_t)np * NB * NB;
  hipLaunchKernelGGL(k_bcr_extract, dim3(np), dim3(256), 0, st, S, n, np, Dk, Bk, ek);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, "k_bcr_extract launch");
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

// Compact form of the same blocks, for the exchange between the ranks of a sharded run: per block
// the lower triangle of D_i (2080), the entries of the coupling that a half bandwidth hb leaves
// non-zero - S(k1 + r, k0 + c) with c >= r + 64 - hb - and e_i (64).
__host__ __device__ inline int compact_b_row_start(int r, int off) {
  // entries of rows 0 .. r-1: sum of max(0, 64 - off - r')
  const int w = NB - off;  // length of row 0
  const int full = r < w ? r : w;
  return full * w - full * (full - 1) / 2;
}

int64_t band_compact_elements(int n, int half_bandwidth) {
  const int64_t np = (n + NB - 1) / NB;
  const int off = NB - half_bandwidth < 0 ? 0 : NB - half_bandwidth;
  return np * (NB * (NB + 1) / 2 + compact_b_row_start(NB, off) + NB);
}

// one workgroup per block: S -> compact (to_compact != 0), or compact -> [D | B | e] with zero fill
__global__ __launch_bounds__(256) void k_bcr_compact(const double* __restrict__ S, int n, int np, int off,
                                                     double* __restrict__ compact, double* __restrict__ Dk,
                                                     double* __restrict__ Bk, double* __restrict__ ek, int to_compact) {
  const int i = blockIdx.x, t = threadIdx.x;
  const int k0 = i * NB, k1 = k0 + NB;
  const int nb = (n - k0) < NB ? (n - k0) : NB;
  const int nb1 = (n - k1) < NB ? (n - k1) : NB;  // <= 0 for the last block
  const int nd = NB * (NB + 1) / 2, nbz = compact_b_row_start(NB, off);
  double* cb = compact + (int64_t)i * (nd + nbz + NB);
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    const bool in_d = c <= r, in_b = c >= r + off;
    const int pd = r * (r + 1) / 2 + c;
    const int pb = nd + compact_b_row_start(r, off) + (c - r - off);
    if (to_compact) {
      if (in_d) {
        double v = (r == c) ? 1.0 : 0.0;  // identity past the matrix edge
        if (r < nb && c < nb) v = S[(int64_t)(k0 + r) * n + k0 + c];
        cb[pd] = v;
      }
      if (in_b) cb[pb] = (r < nb1) ? S[(int64_t)(k1 + r) * n + k0 + c] : 0.0;
    } else {
      Dk[(int64_t)i * NB * NB + idx] = in_d ? cb[pd] : 0.0;
      Bk[(int64_t)i * NB * NB + idx] = in_b ? cb[pb] : 0.0;
    }
  }
  if (t < NB) {
    if (to_compact)
      cb[nd + nbz + t] = t < nb ? S[(int64_t)n * n + k0 + t] : 0.0;
    else
      ek[k0 + t] = cb[nd + nbz + t];
  }
}

int launch_band_compact(const double* S, int n, int half_bandwidth, double* compact, hipStream_t st) {
  const int np = (n + NB - 1) / NB;
  const int off = NB - half_bandwidth < 0 ? 0 : NB - half_bandwidth;
  hipLaunchKernelGGL(k_bcr_compact, dim3(np), dim3(256), 0, st, S, n, np, off, compact, nullptr, nullptr, nullptr, 1);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, "k_bcr_compact launch");
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

int launch_band_expand(const double* compact, int n, int half_bandwidth, double* packed, hipStream_t st) {
  const int np = (n + NB - 1) / NB;
  const int off = NB - half_bandwidth < 0 ? 0 : NB - half_bandwidth;
  double* Dk = packed;
  double* Bk = Dk + (int64_t)np * NB * NB;
  double* ek = Bk + (int64_t)np * NB * NB;
  hipLaunchKernelGGL(k_bcr_compact, dim3(np), dim3(256), 0, st, nullptr, n, np, off, const_cast<double*>(compact), Dk, Bk,
                     ek, 0);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, "k_bcr_compact launch");
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

// damping added after a sharded run's all-reduce: the diagonal of S lives in the diagonal blocks
__global__ void k_bcr_add_diagonal(double* __restrict__ Dk, int n, double mu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Dk[(int64_t)(i / NB) * NB * NB + (i % NB) * (NB + 1)] += mu;
}

// cyclic reduction on the packed form (destroyed); P_u, P_d go to `ws` (2 np 64*64 doubles)
int launch_solve_band_packed(double* packed, int n, double mu_diag, double* da, void* ws, int32_t* info,
                             hipStream_t st) {
  const int np = (n + NB - 1) / NB;
  double* Dk = packed;
  double* Bk = Dk + (int64_t)np * NB * NB;
  double* ek = Bk + (int64_t)np * NB * NB;
  double* Pu = static_cast<double*>(ws);
  double* Pd = Pu + (int64_t)np * NB * NB;
  if (mu_diag != 0.0) hipLaunchKernelGGL(k_bcr_add_diagonal, dim3((n + 255) / 256), dim3(256), 0, st, Dk, n, mu_diag);
  int s = 1;
  for (; (np + s - 1) / s > 3; s *= 2) {
    const int m = (np + s - 1) / s;  // active blocks 0, s, 2s, ...
    hipLaunchKernelGGL(k_bcr_eliminate, dim3(m / 2), dim3(256), 0, st, n, np, s, 1, Dk, Bk, Pu, Pd, ek, info);
    hipLaunchKernelGGL(k_bcr_update, dim3(8 * ((m + 1) / 2)), dim3(256), 0, st, n, np, s, Dk, Bk, Pu, Pd, ek);
  }
  // at most three blocks are left: one workgroup walks them as a chain, substitutions included
  hipLaunchKernelGGL(k_bcr_chain, dim3(1), dim3(256), 0, st, n, np, s, (np + s - 1) / s, Dk, Bk, Pd, ek, da, info);
  for (s /= 2; s >= 1; s /= 2) {
    const int m = (np + s - 1) / s;
    hipLaunchKernelGGL(k_bcr_back, dim3(m / 2), dim3(256), 0, st, n, np, s, 1, Dk, Pu, Pd, ek, da);
  }
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, "cyclic reduction launches");
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

#include "ba_bcr_wide.inc"

// SFM_BA_WIDE_BAND=0 sends bands wider than 64 to the general factorisation (comparison runs);
// SFM_BA_WIDE_BAND=<tiles> caps the superblock edge at that many tiles.
static int wide_band_limit() {
  static const int v = [] {
    const char* e = getenv("SFM_BA_WIDE_BAND");
    return e ? atoi(e) : kWideMaxTiles;
  }();
  return v;
}

// Edge of the blocks the reduced solve works on for this size and band: 64 (cyclic reduction /
// block chain), 128 or 256 (reduction on superblocks), 0 for the general factorisation.
int reduced_solve_block(int n, int half_bandwidth) {
  if (half_bandwidth >= 0 && half_bandwidth <= NB) return NB;
  const int Cw = wide_band_tiles(n, half_bandwidth);
  return Cw > 0 && Cw <= wide_band_limit() ? NB * Cw : 0;
}

// S with its cameras renumbered: row / column 7 p + a of `dst` is row / column 7 order[p] + a of `src`
// (7 x 7 blocks move whole).  Only the lower triangle of src is read; of dst the entries with
// r - c <= half_bandwidth of the lower triangle and the right-hand-side row are written.
__global__ __launch_bounds__(256) void k_permute_system(const double* __restrict__ src, int n,
                                                        const int32_t* __restrict__ order, int hb,
                                                        double* __restrict__ dst) {
  const int r = blockIdx.x;  // 0 .. n: row n is the right-hand side
  if (r == n) {
    for (int c = threadIdx.x; c < n; c += 256) dst[(int64_t)n * n + c] = src[(int64_t)n * n + 7 * order[c / 7] + c % 7];
    return;
  }
  const int rs = 7 * order[r / 7] + r % 7;
  const int c_lo = r - hb < 0 ? 0 : r - hb;
  for (int c = c_lo + threadIdx.x; c <= r; c += 256) {
    const int cs = 7 * order[c / 7] + c % 7;
    dst[(int64_t)r * n + c] = rs >= cs ? src[(int64_t)rs * n + cs] : src[(int64_t)cs * n + rs];
  }
}

int launch_permute_system(const double* src, int n, const int32_t* order, int half_bandwidth, double* dst,
                          hipStream_t st) {
  if (n <= 0) return SFM_BA_OK;
  hipLaunchKernelGGL(k_permute_system, dim3(n + 1), dim3(256), 0, st, src, n, order, half_bandwidth, dst);
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, "k_permute_system launch");
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

__global__ void k_copy_row(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

int launch_solve_reduced(double* S, int n, int half_bandwidth, double* da, void* ws, int32_t* info, hipStream_t st) {
  if (n <= 0) return SFM_BA_OK;
  ProfScope ps(KID_SOLVE, st);
  const int np = (n + NB - 1) / NB;
  double* invL = static_cast<double*>(ws);
  double* y = invL + (int64_t)np * NB * NB;
  int32_t* zflags = reinterpret_cast<int32_t*>(y + (int64_t)np * NB + 64);  // [np][np+1]
  hipError_t err = hipSuccess;
  if (half_bandwidth >= 0 && half_bandwidth <= NB) {
    if (np >= 4) {
      // packed form in the first part of the workspace, P_u / P_d behind it
      double* packed = static_cast<double*>(ws);
      const int rc = launch_band_pack(S, n, packed, st);
      if (rc != SFM_BA_OK) return rc;
      return launch_solve_band_packed(packed, n, 0.0, da, packed + band_packed_elements(n), info, st);
    }
    double* p_ws = reinterpret_cast<double*>(zflags);
    hipLaunchKernelGGL(k_chol_band, dim3(1), dim3(256), 0, st, S, n, invL, p_ws, y, da, info);
    err = hipGetLastError();
    if (err != hipSuccess) {
      set_hip_error(err, "k_chol_band launch");
      return SFM_BA_EHIP;
    }
    return SFM_BA_OK;
  }
  if (half_bandwidth > NB) {
    const int Cw = wide_band_tiles(n, half_bandwidth);
    if (Cw > 0 && Cw <= wide_band_limit()) return launch_solve_wide_band(S, n, Cw, da, ws, info, st);
  }
  // diagnostic switches: SFM_BA_CHOL_SWITCH=<unknowns left from which every tile of the trailing matrix is
  // updated in every launch> (0: never), SFM_BA_CHOL_LOOK=0 (no look-ahead)
  static const int switch_env = [] {
    const char* e = getenv("SFM_BA_CHOL_SWITCH");
    return e ? atoi(e) : 9000;
  }();
  static const int look_env = [] {
    const char* e = getenv("SFM_BA_CHOL_LOOK");
    return e ? atoi(e) : 1;
  }();
  // one launch per panel (ba_chol_step.inc).  While more than ~9 000 unknowns are left the tiles of the
  // trailing matrix take two panels per pass, every second launch (half the traffic); below that the chain
  // of diagonal factorisations is what bounds a launch, every tile is updated in every launch and the
  // look-ahead shortens the chain (measured crossover).
  int ks = switch_env > 0 ? (n - switch_env + NB - 1) / NB : np + 1;
  if (ks < 2) ks = 0;
  const int look = look_env != 0;
  double* side = reinterpret_cast<double*>(zflags + ((((int64_t)np * (np + 2) + 64) + 1) & ~int64_t(1)));
  if (look && ks == 0 && 2 * NB < n)
    hipLaunchKernelGGL(k_chol_tile_copy, dim3(2), dim3(256), 0, st, S, n, NB, 0, 2 * NB, NB, side);
  hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, st, S, n, 0, n < NB ? n : NB, invL, info);
  for (int k = 0; k < np; ++k) {
    const int k0 = k * NB;
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    const int base = k0 + nb;
    const bool has_next = base < n;
    const int ntile = (n + 1 - base + NB - 1) / NB;  // row tiles below the panel, the right-hand side's included
    int64_t bulk = 0;
    if (has_next && k >= 1) {
      const int64_t nT = n / 128 - base / 128 + 1;  // 128-row tiles from the one that holds row `base` to row n's
      bulk = nT * (nT + 1) / 2;
    }
    int64_t grid = ntile + bulk;
    if (grid <= 0) continue;
    if (grid > 256) grid += 2;  // two workgroups without a role, see k_chol_step
    hipLaunchKernelGGL(k_chol_step, dim3((unsigned)grid), dim3(256), 0, st, S, n, k, np, ks, look, ntile, zflags, invL, side,
                       info);
  }
  hipLaunchKernelGGL(k_copy_row, dim3((n + 255) / 256), dim3(256), 0, st, S + (int64_t)n * n, y, n);
  for (int k = np - 1; k >= 0; k -= kBackGroup) {
    const int m = k + 1 < kBackGroup ? k + 1 : kBackGroup;
    const int nwg = k - m + 1 > 0 ? k - m + 1 : 1;
    hipLaunchKernelGGL(k_chol_back_group, dim3(nwg), dim3(256), 0, st, S, n, k, m, invL, y, da, zflags, np);
  }
  err = hipGetLastError();
  if (err != hipSuccess) {
    set_hip_error(err, "cholesky launches");
    return SFM_BA_EHIP;
  }
  return SFM_BA_OK;
}

}  // namespace sfmba
