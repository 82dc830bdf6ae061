// Fused point-major Schur assembly for gfx950 (MI355X): calc_V_inv, calc_W, calc_Y, calc_S, calc_U,
// calc_epsilon_*, calc_e of the reference (src/sparse_ba.py:147-271) in one pass over A.
//
// Applies when the plan found "runs": ranges of consecutive points whose cameras all lie in a window
// of 6 or 8 consecutive camera ids (the visibility an incremental reconstruction produces).  Then
//   S_window -= sum_i Y_i W_i^T,   Y_i = W_i V_i^-1,   W_i = [A_ij^T B_ij]_j  (7 rows per camera j)
// is a dense contraction over the points of the run: with V_i^-1 = L_i L_i^T and Z_i = W_i L_i it is
// the symmetric rank-3 update sum_i Z_i Z_i^T, and a wave accumulates it with v_mfma_f64_16x16x4
// (16 rows = two cameras of 7 parameters + 1 padding row each; K = 3 columns per point).
//
// One wave per run, no barriers.  Lane (p, s) of a step owns the observation of point p (8 points per
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
#include "ba_common.h"

namespace sfmba {

typedef double d4_t __attribute__((ext_vector_type(4)));

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

// sum over the 8 lanes that share lane & 7
__device__ __forceinline__ double sum_over_points(double v) {
  v += dpp64<0x128>(v);  // row_ror:8 = lane ^ 8 inside a row of 16
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

template <int T>  // row tiles of 16: 3 (6 camera slots) or 4 (8 slots)
__global__ __launch_bounds__(64, 2) void k_fused_runs(
    const int32_t* __restrict__ run_pt0, const int32_t* __restrict__ slot_obs, const double* __restrict__ A,
    const double* __restrict__ xt, const double* __restrict__ xp, double mu, double* __restrict__ Vinv,
    double* __restrict__ eps_b, double* __restrict__ part) {
  constexpr int KS = 16 * T;  // rows of the panel
  constexpr int NT = T * (T + 1) / 2;
  constexpr int64_t STRIDE = (int64_t)NT * 256 + 8 * kFzSlotRec;
  __shared__ __attribute__((aligned(16))) double Zp[24 * KS];  // [K column = 3 point + q][row = 8 slot + r]
  const int lane = threadIdx.x;
  const int p = lane >> 3, s = lane & 7;
  const int li = lane & 15, lk = lane >> 4;
  const int run = blockIdx.x;
  const int pt0 = run_pt0[run], pt1 = run_pt0[run + 1];

  d4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = d4_t{0.0, 0.0, 0.0, 0.0};
  double u[28], ev[7];
#pragma unroll
  for (int i = 0; i < 28; ++i) u[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) ev[i] = 0.0;

  // slot table entry of this lane for the step that starts at point `first` (-1: no observation)
#define FZ_FETCH_POS(first) (((first) + p) < pt1 ? slot_obs[kFzLanesPerPoint * (int64_t)((first) + p) + s] : -1)
  // row of A (a[0..13]), x_true (a[14..15]) and x_pred (a[16..17]) of observation `pos`, zeros for none
#define FZ_FETCH_ROWS(pos)                                          \
  do {                                                              \
    if ((pos) >= 0) {                                               \
      const double* ap_ = A + 14 * (int64_t)(pos);                  \
      _Pragma("unroll") for (int q_ = 0; q_ < 7; ++q_) {            \
        const double2 v_ = ldg2(ap_ + 2 * q_);                      \
        a[2 * q_] = v_.x;                                           \
        a[2 * q_ + 1] = v_.y;                                       \
      }                                                             \
      const double2 t_ = ldg2(xt + 2 * (int64_t)(pos));             \
      const double2 p_ = ldg2(xp + 2 * (int64_t)(pos));             \
      a[14] = t_.x; a[15] = t_.y; a[16] = p_.x; a[17] = p_.y;       \
    } else {                                                        \
      _Pragma("unroll") for (int q_ = 0; q_ < 18; ++q_) a[q_] = 0.0; \
    }                                                               \
  } while (0)
  double a[18];
  int pos = FZ_FETCH_POS(pt0);
  int pos_next = FZ_FETCH_POS(pt0 + 8);
  FZ_FETCH_ROWS(pos);

  for (int first = pt0; first < pt1; first += 8) {
    const int pt = first + p;
    // ---- per observation: point block (minus the camera-centre columns of A), residual ----
    const double b0 = -a[4], b1 = -a[5], b2 = -a[6], b3 = -a[11], b4 = -a[12], b5 = -a[13];
    const double e0 = a[14] - a[16], e1 = a[15] - a[17];
    // B^T B (00 01 02 11 12 22) and B^T eps, summed over the observations of the point
    const double v00 = sum8(b0 * b0 + b3 * b3) + mu;
    const double v01 = sum8(b0 * b1 + b3 * b4);
    const double v02 = sum8(b0 * b2 + b3 * b5);
    const double v11 = sum8(b1 * b1 + b4 * b4) + mu;
    const double v12 = sum8(b1 * b2 + b4 * b5);
    const double v22 = sum8(b2 * b2 + b5 * b5) + mu;
    const double q0 = sum8(b0 * e0 + b3 * e1);
    const double q1 = sum8(b1 * e0 + b4 * e1);
    const double q2 = sum8(b2 * e0 + b5 * e1);
    // V^-1 in closed form (V is symmetric positive definite for mu > 0)
    const double c00 = v11 * v22 - v12 * v12;
    const double c01 = v02 * v12 - v01 * v22;
    const double c02 = v01 * v12 - v02 * v11;
    const double c11 = v00 * v22 - v02 * v02;
    const double c12 = v01 * v02 - v00 * v12;
    const double c22 = v00 * v11 - v01 * v01;
    const double det = v00 * c00 + v01 * c01 + v02 * c02;
    const double id = 1.0 / det;
    const double i00 = c00 * id, i01 = c01 * id, i02 = c02 * id, i11 = c11 * id, i12 = c12 * id, i22 = c22 * id;
    if (pt < pt1) {
      // lane s < 3 of the point stores row s of V^-1 and entry s of eps_b.  Three masked blocks on
      // laundered copies of s: a selection by s (of values or of addresses) is turned into a table
      // in scratch memory by the compiler
      double* vo = Vinv + 9 * (int64_t)pt;
      double* eo = eps_b + 3 * (int64_t)pt;
      int sa = s, sb = s, sc = s;
      asm volatile("" : "+v"(sa));
      asm volatile("" : "+v"(sb));
      asm volatile("" : "+v"(sc));
      if (sa == 0) { vo[0] = i00; vo[1] = i01; vo[2] = i02; eo[0] = q0; }
      if (sb == 1) { vo[3] = i01; vo[4] = i11; vo[5] = i12; eo[1] = q1; }
      if (sc == 2) { vo[6] = i02; vo[7] = i12; vo[8] = i22; eo[2] = q2; }
    }
    const double z0 = i00 * q0 + i01 * q1 + i02 * q2;
    const double z1 = i01 * q0 + i11 * q1 + i12 * q2;
    const double z2 = i02 * q0 + i12 * q1 + i22 * q2;
    const double g0 = e0 - (b0 * z0 + b1 * z1 + b2 * z2);
    const double g1 = e1 - (b3 * z0 + b4 * z1 + b5 * z2);
    // ---- per camera slot: U_j += A^T A, e_j += A^T g ----
    {
      int v = 0;
#pragma unroll
      for (int r = 0; r < 7; ++r) {
#pragma unroll
        for (int c = r; c < 7; ++c, ++v) u[v] = fma(a[7 + r], a[7 + c], fma(a[r], a[c], u[v]));
        ev[r] = fma(a[7 + r], g1, fma(a[r], g0, ev[r]));
      }
    }
    // ---- Z = A^T (B L), V^-1 = L L^T ----
    const double l00 = sqrt(i00);
    const double rl0 = 1.0 / l00;
    const double l10 = i01 * rl0, l20 = i02 * rl0;
    const double l11 = sqrt(i11 - l10 * l10);
    const double l21 = (i12 - l20 * l10) / l11;
    const double l22 = sqrt(i22 - l20 * l20 - l21 * l21);
    const double m00 = b0 * l00 + b1 * l10 + b2 * l20, m01 = b1 * l11 + b2 * l21, m02 = b2 * l22;
    const double m10 = b3 * l00 + b4 * l10 + b5 * l20, m11 = b4 * l11 + b5 * l21, m12 = b5 * l22;
    if (s < 2 * T) {
      double* zp = Zp + (3 * p) * KS + 8 * s;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double ma = q == 0 ? m00 : (q == 1 ? m01 : m02);
        const double mb = q == 0 ? m10 : (q == 1 ? m11 : m12);
        double z[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) z[r] = a[r] * ma + a[7 + r] * mb;
        sts2(zp + q * KS, z[0], z[1]);
        sts2(zp + q * KS + 2, z[2], z[3]);
        sts2(zp + q * KS + 4, z[4], z[5]);
        sts2(zp + q * KS + 6, z[6], 0.0);
      }
    }
    // ---- next step's operands on their way while the matrix cores work ----
    pos = pos_next;
    FZ_FETCH_ROWS(pos);
    pos_next = FZ_FETCH_POS(first + 16);
    // one wave wrote the panel and reads it: its LDS operations complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
      double zt[T];
#pragma unroll
      for (int t = 0; t < T; ++t) zt[t] = Zp[(4 * kk + lk) * KS + 16 * t + li];
      int idx = 0;
#pragma unroll
      for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj, ++idx)
          acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(zt[ti], zt[tj], acc[idx], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

#undef FZ_FETCH_POS
#undef FZ_FETCH_ROWS

  // ---- the run's partial result: tiles as they lie in the registers, then the per-slot sums ----
  double* po = part + (int64_t)run * STRIDE;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) po[t * 256 + (lk + 4 * v) * 16 + li] = acc[t][v];
  double* uo = po + NT * 256 + s * kFzSlotRec;
#pragma unroll
  for (int i = 0; i < 28; ++i) {
    const double v = sum_over_points(u[i]);
    if (p == 0) uo[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const double v = sum_over_points(ev[i]);
    if (p == 0) uo[28 + i] = v;
  }
}

// partial results of the runs of one window, added in run order
__global__ __launch_bounds__(256) void k_fused_reduce(const int32_t* __restrict__ key_ptr,
                                                      const int32_t* __restrict__ key_run, int64_t stride,
                                                      const double* __restrict__ part, double* __restrict__ kpart) {
  const int key = blockIdx.x;
  const int beg = key_ptr[key], end = key_ptr[key + 1];
  if (beg == end) return;
  for (int64_t el = (int64_t)blockIdx.y * 256 + threadIdx.x; el < stride; el += (int64_t)gridDim.y * 256) {
    double sum = 0.0;
    int i = beg;
    for (; i + 4 <= end; i += 4) {
      const double v0 = part[(int64_t)key_run[i] * stride + el];
      const double v1 = part[(int64_t)key_run[i + 1] * stride + el];
      const double v2 = part[(int64_t)key_run[i + 2] * stride + el];
      const double v3 = part[(int64_t)key_run[i + 3] * stride + el];
      sum = (((sum + v0) + v1) + v2) + v3;
    }
    for (; i < end; ++i) sum += part[(int64_t)key_run[i] * stride + el];
    kpart[(int64_t)key * stride + el] = sum;
  }
}

// one 64-lane workgroup per camera pair (j <= k) that shares a point: add the windows that hold both
// cameras, write the block of S below the diagonal (diagonal blocks in full, with U_j and the
// damping), and for diagonal blocks e_j
__global__ __launch_bounds__(64) void k_fused_finish(const int32_t* __restrict__ blk_j, const int32_t* __restrict__ blk_k,
                                                     const int32_t* __restrict__ key_ptr, int slots, int group,
                                                     const double* __restrict__ kpart, double mu_u, int m,
                                                     double* __restrict__ S, double* __restrict__ e_out) {
  const int t = threadIdx.x;
  const int j = blk_j[blockIdx.x], k = blk_k[blockIdx.x];
  if (t >= 56 || (t >= 49 && j != k)) return;
  const int nt = fz_tiles(slots);
  const int64_t stride = fz_part_stride(slots);
  const int lo = k - slots + 1 > 0 ? k - slots + 1 : 0;
  const int key_lo = (lo + group - 1) / group, key_hi = j / group;
  const int r = t / 7, c = t - 7 * r;
  const int ulo = r < c ? r : c, uhi = r < c ? c : r;
  const int uidx = ulo * 7 - ulo * (ulo - 1) / 2 + (uhi - ulo);
  double sum = 0.0;
  for (int key = key_lo; key <= key_hi; ++key) {
    if (key_ptr[key + 1] == key_ptr[key]) continue;
    const int base = key * group;
    const int sj = j - base, sk = k - base;
    const double* pb = kpart + (int64_t)key * stride;
    if (t < 49) {
      // (Y_j W_k^T)[r][c] = D[8 sk + c][8 sj + r], a lower-triangle tile (sk >= sj)
      const int ti = sk >> 1, tj = sj >> 1;
      double v = pb[(ti * (ti + 1) / 2 + tj) * 256 + ((sk & 1) * 8 + c) * 16 + (sj & 1) * 8 + r];
      if (j == k) v = pb[nt * 256 + sj * kFzSlotRec + uidx] - v;
      sum += v;
    } else {
      sum += pb[nt * 256 + sj * kFzSlotRec + 28 + (t - 49)];
    }
  }
  const int64_t ld = 7 * (int64_t)m;
  if (t >= 49) {
    S[ld * ld + 7 * j + (t - 49)] = sum;
    if (e_out) e_out[7 * j + (t - 49)] = sum;
  } else if (j == k) {
    S[(7 * (int64_t)j + r) * ld + 7 * j + c] = sum + (r == c ? mu_u : 0.0);
  } else {
    S[(7 * (int64_t)k + c) * ld + 7 * j + r] = -sum;
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

int launch_assemble_fused(const sfm_ba_plan* p, const double* A, const double* xt, const double* xp, double mu,
                          double mu_u, double* S, double* e, double* Vinv, double* eps_b, void* ws, hipStream_t st) {
  if (p->n_runs <= 0) return SFM_BA_OK;
  char* base = static_cast<char*>(ws);
  double* part = reinterpret_cast<double*>(base + p->ws_fpart);
  double* kpart = reinterpret_cast<double*>(base + p->ws_fkey);
  const int64_t stride = fz_part_stride(p->fz_slots);
  {
    ProfScope ps(KID_FUSED_RUNS, st);
    if (p->fz_slots == 6)
      hipLaunchKernelGGL(k_fused_runs<3>, dim3(p->n_runs), dim3(64), 0, st, p->d_run_pt0, p->d_slot_obs, A, xt, xp, mu,
                         Vinv, eps_b, part);
    else
      hipLaunchKernelGGL(k_fused_runs<4>, dim3(p->n_runs), dim3(64), 0, st, p->d_run_pt0, p->d_slot_obs, A, xt, xp, mu,
                         Vinv, eps_b, part);
  }
  int rc = check_launch_fz("k_fused_runs");
  if (rc) return rc;
  if (!S) return SFM_BA_OK;
  {
    ProfScope ps(KID_FUSED_REDUCE, st);
    hipLaunchKernelGGL(k_fused_reduce, dim3(p->n_keys, 3), dim3(256), 0, st, p->d_key_ptr, p->d_key_run, stride, part,
                       kpart);
  }
  rc = check_launch_fz("k_fused_reduce");
  if (rc) return rc;
  if (p->n_blocks > 0) {
    ProfScope ps(KID_FUSED_FINISH, st);
    hipLaunchKernelGGL(k_fused_finish, dim3(p->n_blocks), dim3(64), 0, st, p->d_blk_j, p->d_blk_k, p->d_key_ptr,
                       p->fz_slots, p->fz_group, kpart, mu_u, p->m, S, e);
  }
  return check_launch_fz("k_fused_finish");
}

}  // namespace sfmba
