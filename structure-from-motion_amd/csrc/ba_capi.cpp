// extern "C" surface of libsfm_ba.so: argument checks, then the kernel launchers.
#include "ba_common.h"

using namespace sfmba;

extern "C" {

int sfm_ba_version(void) { return SFM_BA_VERSION; }

const char* sfm_ba_error_string(int code) {
  switch (code) {
    case SFM_BA_OK: return "ok";
    case SFM_BA_EINVAL: return "invalid argument";
    case SFM_BA_EHIP: return "HIP runtime error";
    case SFM_BA_ENONUNIQUE: return "Found non-unique (i, j) pair";
    case SFM_BA_EMISSING_POINT: return "point indices are not contiguous";
    case SFM_BA_EMISSING_VIEW: return "viewpoint indices are not contiguous";
    case SFM_BA_ENOTPD: return "reduced camera system is not positive definite";
    case SFM_BA_ETOOBIG: return "a point has too many observations or the problem exceeds int32 indexing";
    default: return "unknown error";
  }
}

int sfm_ba_camera_prepare(const double* d_R, const double* d_C, const double* d_K, int k_stride, int32_t m,
                          double* d_cam_blk, int32_t* d_flags, void* stream) {
  if (!d_R || !d_C || !d_K || !d_cam_blk || m < 0 || (k_stride != 0 && k_stride != 9)) return SFM_BA_EINVAL;
  return launch_camera_prepare(d_R, d_C, d_K, k_stride, m, d_cam_blk, d_flags, static_cast<hipStream_t>(stream));
}

int sfm_ba_jacobian(const double* d_cam_blk, const double* d_P, const double* d_X, const int32_t* d_cam_idx,
                    const int32_t* d_pt_idx, int64_t n_obs, double* d_A, double* d_B, double* d_x_pred,
                    void* stream) {
  if (!d_cam_blk || !d_X || !d_cam_idx || !d_pt_idx || n_obs < 0) return SFM_BA_EINVAL;
  if (d_x_pred && !d_P) return SFM_BA_EINVAL;
  if (n_obs >= (int64_t(1) << 31)) return SFM_BA_ETOOBIG;
  return launch_jacobian(d_cam_blk, d_P, d_X, d_cam_idx, d_pt_idx, n_obs, d_A, d_B, d_x_pred,
                         static_cast<hipStream_t>(stream));
}

int sfm_ba_schur_assemble(const sfm_ba_plan* plan, const double* d_A, const double* d_B, const double* d_x_true,
                          const double* d_x_pred, const double* d_weights, double mu, double mu_u, double* d_S,
                          double* d_e, double* d_U, double* d_eps_a, double* d_V_inv, double* d_eps_b,
                          double* d_cam_red, void* d_workspace, void* stream) {
  if (!plan || !d_A || !d_x_true || !d_x_pred || !d_V_inv || !d_eps_b || !d_workspace) return SFM_BA_EINVAL;
  if (!(mu >= 0.0) || !(mu_u >= 0.0)) return SFM_BA_EINVAL;
  return launch_assemble(plan, d_A, d_B, d_x_true, d_x_pred, d_weights, mu, mu_u, d_S, d_e, d_U, d_eps_a, d_V_inv,
                         d_eps_b, d_cam_red, d_workspace, static_cast<hipStream_t>(stream));
}

int sfm_ba_schur_assemble_pairs_state(const sfm_ba_plan* plan, const double* d_A, const double* d_cam_blk,
                                      const double* d_X, const double* d_x_true, const double* d_x_pred, double mu,
                                      double mu_u, double* d_S, double* d_e, double* d_V_inv, double* d_eps_b,
                                      double* d_cam_ssq, void* d_workspace, void* stream) {
  if (!plan || !d_A || !d_cam_blk || !d_X || !d_x_true || !d_x_pred || !d_S || !d_V_inv || !d_eps_b || !d_workspace)
    return SFM_BA_EINVAL;
  if (!(mu >= 0.0) || !(mu_u >= 0.0)) return SFM_BA_EINVAL;
  return launch_assemble_pairs_state(plan, d_A, d_cam_blk, d_X, d_x_true, d_x_pred, mu, mu_u, d_S, d_e, d_V_inv, d_eps_b,
                                     d_cam_ssq, d_workspace, static_cast<hipStream_t>(stream));
}

int sfm_ba_plan_state_path(const sfm_ba_plan* plan) { return plan && fz_state_applies(plan) ? 1 : 0; }

int sfm_ba_schur_assemble_state(const sfm_ba_plan* plan, const double* d_cam_blk, const double* d_P, const double* d_X,
                                const double* d_x_true, double mu, double mu_u, double* d_S, double* d_e,
                                double* d_V_inv, double* d_eps_b, double* d_cam_ssq, void* d_workspace, void* stream) {
  if (!plan || !d_cam_blk || !d_P || !d_X || !d_x_true || !d_V_inv || !d_eps_b || !d_workspace) return SFM_BA_EINVAL;
  if (!fz_state_applies(plan) || !(mu > 0.0) || !(mu_u >= 0.0)) return SFM_BA_EINVAL;
  return launch_assemble_state(plan, d_cam_blk, d_P, d_X, d_x_true, mu, mu_u, d_S, d_e, d_V_inv, d_eps_b, d_cam_ssq,
                               d_workspace, static_cast<hipStream_t>(stream));
}

int sfm_ba_back_substitute_state(const sfm_ba_plan* plan, const double* d_cam_blk, const double* d_V_inv,
                                 const double* d_eps_b, const double* d_delta_a, double* d_delta_b, double* d_X,
                                 int update_points, double x_step, void* d_workspace, void* stream) {
  if (!plan || !d_cam_blk || !d_V_inv || !d_eps_b || !d_delta_a || !d_delta_b || !d_X || !d_workspace)
    return SFM_BA_EINVAL;
  if (!fz_state_applies(plan)) return SFM_BA_EINVAL;
  return launch_back_substitute_state(plan, d_cam_blk, d_V_inv, d_eps_b, d_delta_a, d_delta_b, d_X, update_points,
                                      x_step, d_workspace, static_cast<hipStream_t>(stream));
}

int sfm_ba_calc_W_Y(const sfm_ba_plan* plan, const double* d_A, const double* d_B, const double* d_weights,
                    const double* d_V_inv, const int32_t* d_pt_idx, double* d_W, double* d_Y, void* stream) {
  if (!plan || !d_A || !d_B || !d_V_inv || !d_pt_idx) return SFM_BA_EINVAL;
  return launch_calc_W_Y(plan, d_A, d_B, d_weights, d_V_inv, d_pt_idx, d_W, d_Y, static_cast<hipStream_t>(stream));
}

int sfm_ba_add_diagonal(double* d_S, int32_t n7, double mu, void* stream) {
  if (!d_S || n7 < 0) return SFM_BA_EINVAL;
  return launch_add_diagonal(d_S, n7, mu, static_cast<hipStream_t>(stream));
}

int sfm_ba_iteration_status(const double* d_cost, const int32_t* d_info, const int32_t* d_flags_a,
                            const int32_t* d_flags_b, int32_t m, double* d_status, void* stream) {
  if (!d_status || m < 0) return SFM_BA_EINVAL;
  return launch_status(d_cost, d_info, d_flags_a, d_flags_b, m, d_status, static_cast<hipStream_t>(stream));
}

int64_t sfm_ba_lower_elements(int32_t n7) { return n7 > 0 ? (int64_t)n7 * (n7 + 1) / 2 + n7 : 0; }

int sfm_ba_lower_pack(const double* d_S, int32_t n7, double* d_tri, void* stream) {
  if (!d_S || !d_tri || n7 < 0) return SFM_BA_EINVAL;
  return launch_lower_pack(d_S, n7, d_tri, static_cast<hipStream_t>(stream));
}

int sfm_ba_lower_unpack(const double* d_tri, int32_t n7, double mu, double* d_S, void* stream) {
  if (!d_S || !d_tri || n7 < 0) return SFM_BA_EINVAL;
  return launch_lower_unpack(d_tri, n7, mu, d_S, static_cast<hipStream_t>(stream));
}

int64_t sfm_ba_cholesky_workspace_bytes(int32_t n7) { return n7 > 0 ? cholesky_workspace_bytes(n7) : 0; }

int sfm_ba_solve_reduced(double* d_S, int32_t n7, double* d_delta_a, void* d_chol_ws, int32_t* d_info,
                         void* stream) {
  if (!d_S || !d_delta_a || !d_chol_ws || !d_info || n7 <= 0) return SFM_BA_EINVAL;
  return launch_solve_reduced(d_S, n7, -1, d_delta_a, d_chol_ws, d_info, static_cast<hipStream_t>(stream));
}

int sfm_ba_solve_reduced_banded(double* d_S, int32_t n7, int32_t half_bandwidth, double* d_delta_a,
                                void* d_chol_ws, int32_t* d_info, void* stream) {
  if (!d_S || !d_delta_a || !d_chol_ws || !d_info || n7 <= 0 || half_bandwidth < 0) return SFM_BA_EINVAL;
  return launch_solve_reduced(d_S, n7, half_bandwidth, d_delta_a, d_chol_ws, d_info,
                              static_cast<hipStream_t>(stream));
}

int32_t sfm_ba_reduced_solve_block(int32_t n7, int32_t half_bandwidth) {
  return n7 > 0 && half_bandwidth >= 0 ? reduced_solve_block(n7, half_bandwidth) : 0;
}

int sfm_ba_permute_system(const double* d_S, int32_t n7, const int32_t* d_cam_order, int32_t half_bandwidth,
                          double* d_S_out, void* stream) {
  if (!d_S || !d_cam_order || !d_S_out || d_S == d_S_out || n7 < 0 || n7 % 7 != 0 || half_bandwidth < 0)
    return SFM_BA_EINVAL;
  return launch_permute_system(d_S, n7, d_cam_order, half_bandwidth, d_S_out, static_cast<hipStream_t>(stream));
}

int sfm_ba_triangulate(const double* d_x1, const double* d_x2, const double* d_P1, const double* d_P2, int64_t n,
                       float* d_Xh, void* stream) {
  if (n < 0 || (n > 0 && (!d_x1 || !d_x2 || !d_P1 || !d_P2 || !d_Xh))) return SFM_BA_EINVAL;
  return launch_triangulate(d_x1, d_x2, d_P1, d_P2, n, d_Xh, static_cast<hipStream_t>(stream));
}

int64_t sfm_ba_pnp_workspace_bytes(int32_t n_hyp, int32_t n_per_hyp) {
  return n_hyp > 0 && n_per_hyp > 0 ? pnp_workspace_bytes(n_hyp, n_per_hyp) : 0;
}

int sfm_ba_pnp_solve(const double* d_x, const double* d_X, const double* d_Kinv_pre, const double* d_Kinv_post,
                     const int32_t* d_idx, int32_t n_hyp, int32_t n_per_hyp, void* d_workspace, double* d_R, double* d_T,
                     int32_t* d_flags, void* stream) {
  if (!d_x || !d_X || !d_workspace || !d_R || !d_T || !d_flags || n_hyp <= 0 || n_per_hyp <= 0) return SFM_BA_EINVAL;
  return launch_pnp_solve(d_x, d_X, d_Kinv_pre, d_Kinv_post, d_idx, n_hyp, n_per_hyp, d_workspace, d_R, d_T, d_flags,
                          static_cast<hipStream_t>(stream));
}

int sfm_ba_pnp_cleanup(const double* d_RT, const double* d_Kinv_post, int32_t n, double* d_R, double* d_T,
                       int32_t* d_flags, void* stream) {
  if (!d_RT || !d_R || !d_T || !d_flags || n <= 0) return SFM_BA_EINVAL;
  return launch_pnp_cleanup(d_RT, d_Kinv_post, n, d_R, d_T, d_flags, static_cast<hipStream_t>(stream));
}

int sfm_ba_pnp_inliers(const double* d_x, const double* d_X, const double* d_K, const double* d_R, const double* d_T,
                       int32_t n_hyp, int64_t n, double threshold, int32_t* d_counts, uint8_t* d_mask, void* stream) {
  if (!d_x || !d_X || !d_K || !d_R || !d_T || !d_counts) return SFM_BA_EINVAL;
  return launch_pnp_inliers(d_x, d_X, d_K, d_R, d_T, n_hyp, n, threshold, d_counts, d_mask,
                            static_cast<hipStream_t>(stream));
}

int64_t sfm_ba_band_packed_elements(int32_t n7) { return n7 > 0 ? band_packed_elements(n7) : 0; }

int sfm_ba_band_pack(const double* d_S, int32_t n7, double* d_packed, void* stream) {
  if (!d_S || !d_packed || n7 <= 0) return SFM_BA_EINVAL;
  return launch_band_pack(d_S, n7, d_packed, static_cast<hipStream_t>(stream));
}

int64_t sfm_ba_band_compact_elements(int32_t n7, int32_t half_bandwidth) {
  return n7 > 0 && half_bandwidth >= 0 && half_bandwidth <= 64 ? band_compact_elements(n7, half_bandwidth) : 0;
}

int sfm_ba_band_compact(const double* d_S, int32_t n7, int32_t half_bandwidth, double* d_compact, void* stream) {
  if (!d_S || !d_compact || n7 <= 0 || half_bandwidth < 0 || half_bandwidth > 64) return SFM_BA_EINVAL;
  return launch_band_compact(d_S, n7, half_bandwidth, d_compact, static_cast<hipStream_t>(stream));
}

int sfm_ba_band_expand(const double* d_compact, int32_t n7, int32_t half_bandwidth, double* d_packed, void* stream) {
  if (!d_compact || !d_packed || n7 <= 0 || half_bandwidth < 0 || half_bandwidth > 64) return SFM_BA_EINVAL;
  return launch_band_expand(d_compact, n7, half_bandwidth, d_packed, static_cast<hipStream_t>(stream));
}

int sfm_ba_solve_band_packed(double* d_packed, int32_t n7, double mu_diag, double* d_delta_a, void* d_chol_ws,
                             int32_t* d_info, void* stream) {
  if (!d_packed || !d_delta_a || !d_chol_ws || !d_info || n7 <= 0) return SFM_BA_EINVAL;
  return launch_solve_band_packed(d_packed, n7, mu_diag, d_delta_a, d_chol_ws, d_info,
                                  static_cast<hipStream_t>(stream));
}

int sfm_ba_back_substitute(const sfm_ba_plan* plan, const double* d_A, const double* d_B, const double* d_weights,
                           const double* d_V_inv, const double* d_eps_b, const double* d_delta_a,
                           double* d_delta_b, double* d_X_update, double x_step, void* stream) {
  if (!plan || !d_A || !d_V_inv || !d_eps_b || !d_delta_a || !d_delta_b) return SFM_BA_EINVAL;
  return launch_back_substitute(plan, d_A, d_B, d_weights, d_V_inv, d_eps_b, d_delta_a, d_delta_b, d_X_update,
                                x_step, static_cast<hipStream_t>(stream));
}

int sfm_ba_update_cameras(double* d_R, double* d_T, double* d_C, double* d_P, const double* d_K, int k_stride,
                          const double* d_delta_a, int32_t m, int32_t first_cam, double q_step, double c_step,
                          int32_t* d_flags, void* stream) {
  if (!d_R || !d_T || !d_C || !d_K || !d_delta_a || m < 0 || first_cam < 0 || (k_stride != 0 && k_stride != 9))
    return SFM_BA_EINVAL;
  return launch_update_cameras(d_R, d_T, d_C, d_P, d_K, k_stride, d_delta_a, m, first_cam, -1, q_step, c_step, d_flags,
                               nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int sfm_ba_update_cameras_prepare(double* d_R, double* d_T, double* d_C, double* d_P, const double* d_K, int k_stride,
                                  const double* d_delta_a, int32_t m, int32_t first_cam, double q_step, double c_step,
                                  int32_t* d_flags, double* d_cam_blk, int32_t* d_blk_flags, void* stream) {
  if (!d_R || !d_T || !d_C || !d_K || !d_delta_a || !d_cam_blk || m < 0 || first_cam < 0 ||
      (k_stride != 0 && k_stride != 9))
    return SFM_BA_EINVAL;
  return launch_update_cameras(d_R, d_T, d_C, d_P, d_K, k_stride, d_delta_a, m, first_cam, -1, q_step, c_step, d_flags,
                               d_cam_blk, d_blk_flags, static_cast<hipStream_t>(stream));
}

int sfm_ba_update_cameras_held(double* d_R, double* d_T, double* d_C, double* d_P, const double* d_K, int k_stride,
                               const double* d_delta_a, int32_t m, int32_t held_cam, double q_step, double c_step,
                               int32_t* d_flags, double* d_cam_blk, int32_t* d_blk_flags, void* stream) {
  if (!d_R || !d_T || !d_C || !d_K || !d_delta_a || m < 0 || held_cam < -1 || held_cam >= m ||
      (k_stride != 0 && k_stride != 9))
    return SFM_BA_EINVAL;
  return launch_update_cameras(d_R, d_T, d_C, d_P, d_K, k_stride, d_delta_a, m, 0, held_cam, q_step, c_step, d_flags,
                               d_cam_blk, d_blk_flags, static_cast<hipStream_t>(stream));
}

int sfm_ba_cost(const sfm_ba_plan* plan, const double* d_P, const double* d_X, const int32_t* d_cam_idx,
                const int32_t* d_pt_idx, const double* d_x_true, double* d_cam_ssq, double* d_cost, void* d_workspace,
                void* stream) {
  if (!plan || !d_P || !d_X || !d_cam_idx || !d_pt_idx || !d_x_true || !d_cam_ssq || !d_cost || !d_workspace)
    return SFM_BA_EINVAL;
  return launch_cost(plan, d_P, d_X, d_cam_idx, d_pt_idx, d_x_true, d_cam_ssq, d_cost, d_workspace,
                     static_cast<hipStream_t>(stream));
}

int sfm_ba_cost_noalloc(const sfm_ba_plan* plan, const double* d_P, const double* d_X, const int32_t* d_cam_idx,
                        const int32_t* d_pt_idx, const double* d_x_true, double* d_cam_ssq, double* d_cost,
                        void* stream) {
  if (!plan || !d_P || !d_X || !d_cam_idx || !d_pt_idx || !d_x_true || !d_cam_ssq || !d_cost) return SFM_BA_EINVAL;
  return launch_cost_noalloc(plan, d_P, d_X, d_cam_idx, d_pt_idx, d_x_true, d_cam_ssq, d_cost,
                             static_cast<hipStream_t>(stream));
}

int sfm_ba_cost_partial(const sfm_ba_plan* plan, const double* d_P, const double* d_X, const int32_t* d_cam_idx,
                        const int32_t* d_pt_idx, const double* d_x_true, double* d_cam_ssq, void* d_workspace,
                        void* stream) {
  if (!plan || !d_P || !d_X || !d_cam_idx || !d_pt_idx || !d_x_true || !d_cam_ssq || !d_workspace)
    return SFM_BA_EINVAL;
  return launch_cost_partial(plan, d_P, d_X, d_cam_idx, d_pt_idx, d_x_true, d_cam_ssq, d_workspace,
                             static_cast<hipStream_t>(stream));
}

int sfm_ba_cost_finish(const double* d_cam_ssq, int32_t m, double* d_cost, void* stream) {
  if (!d_cam_ssq || !d_cost || m <= 0) return SFM_BA_EINVAL;
  return launch_cost_finish(d_cam_ssq, m, d_cost, static_cast<hipStream_t>(stream));
}

}  // extern "C"
