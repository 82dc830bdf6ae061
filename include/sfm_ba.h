/*
 * sfm_ba.h - C ABI of libsfm_ba.so, the MI355X (gfx950) bundle-adjustment library.
 *
 * Every entry point takes plain pointers and sizes.  Pointers named d_* are DEVICE
 * pointers (HBM of the current HIP device); pointers named h_* are host pointers.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).  All
 * floating point data is IEEE fp64, all indices are int32.  Functions return
 * SFM_BA_OK (0) or a negative SFM_BA_E* code; sfm_ba_error_string() names it.
 * Kernels are enqueued on `stream` and the call returns without synchronising
 * unless stated otherwise.
 *
 * The reference is pure Python (numpy); there is no FFI in it to bind.  Each entry
 * point below states the reference function(s) whose work it performs, with paths
 * relative to the reference checkout.
 */
#ifndef SFM_BA_H
#define SFM_BA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_BA_VERSION 1

/* status codes */
#define SFM_BA_OK 0
#define SFM_BA_EINVAL -1        /* bad argument (null pointer, negative size, bad stride) */
#define SFM_BA_EHIP -2          /* a HIP runtime call failed; see sfm_ba_last_hip_error() */
#define SFM_BA_ENONUNIQUE -3    /* "Found non-unique (i, j) pair"         src/sparse_ba.py:61-62 */
#define SFM_BA_EMISSING_POINT -4 /* point ids not contiguous 0..max        src/sparse_ba.py:89 */
#define SFM_BA_EMISSING_VIEW -5 /* viewpoint ids not contiguous 0..max    src/sparse_ba.py:91 */
#define SFM_BA_ENOTPD -6        /* reduced camera system not positive definite (numpy raises LinAlgError
                                   for a singular S at src/sparse_ba.py:275) */
#define SFM_BA_ETOOBIG -7       /* the problem exceeds the plan's 32-bit offsets (2^32 / 112 observations,
                                   about 134 M same-point observation pairs) or an entry point's stated maximum */

/* doubles per prepared camera record (see sfm_ba_camera_prepare) */
#define SFM_BA_CAM_BLK 56
/* doubles per per-camera reduction record: 28 (upper triangle of U) + 7 (eps_a) + 7 (sum Y eps_b) + 1 (sum |eps|^2) */
#define SFM_BA_CAM_RED 43

/* flag bits written by sfm_ba_camera_prepare / sfm_ba_update_cameras into d_flags[camera] */
#define SFM_BA_FLAG_DET_NOT_ONE 1   /* |det R - 1| > 1e-8 + 1e-5   src/utils.py:74-76 */
#define SFM_BA_FLAG_ZERO_QUAT 2     /* |Q + dQ| ~ 0                src/bundle_adjustment.py:89-91 */
#define SFM_BA_FLAG_NONFINITE 4     /* a NaN/Inf reached the camera record */

int sfm_ba_version(void);
const char* sfm_ba_error_string(int code);
/* text of the last failing HIP call on the calling thread ("" if none) */
const char* sfm_ba_last_hip_error(void);

/* ------------------------------------------------------------------------------------------
 * Visibility plan.  Performs the work of `Indices.__init__` (src/sparse_ba.py:58-97): checks
 * that (point, viewpoint) pairs are unique and that ids are contiguous, and builds the index
 * structures the kernels walk: observations grouped by point, per-camera work items, and for
 * every camera pair (j <= k) that shares a point the list of (obs of j, obs of k) pairs -
 * the information `Indices.shared_point_indices` (src/sparse_ba.py:107-128) recomputes from a
 * dense mask on every call.  Built on the host from host index arrays, then uploaded.
 *   h_cam_idx, h_pt_idx : n_obs entries, any order (the reference emits frame-major order,
 *                         src/bundle_adjustment.py:44-67)
 *   n_cams_hint, n_pts_hint : 0 to derive max+1 as the reference does; a rank of a sharded
 *                         run passes the global camera count (cameras it does not observe
 *                         are then allowed) and its local point count
 *   check              : non-zero to run the uniqueness / contiguity checks
 * On success *out_plan receives an opaque handle; free it with sfm_ba_plan_destroy.
 * Synchronous (blocks until the upload finished).
 * ---------------------------------------------------------------------------------------- */
typedef struct sfm_ba_plan sfm_ba_plan;

int sfm_ba_plan_create(const int32_t* h_cam_idx, const int32_t* h_pt_idx, int64_t n_obs,
                       int32_t n_cams_hint, int32_t n_pts_hint, int check, sfm_ba_plan** out_plan);
/* The same plan (array for array) built on the device from index lists that already sit in device
 * memory: the per-call index work of `get_indices` (src/bundle_adjustment.py:44-67) and
 * `Indices.__init__` (src/sparse_ba.py:58-97) as two radix sorts and a few prefix sums, without
 * the host pass and the upload of the finished plan.  Same checks, same return codes.  Allocates
 * temporary device memory (about 24 bytes per pair entry) and returns after `stream` has drained. */
int sfm_ba_plan_create_device(const int32_t* d_cam_idx, const int32_t* d_pt_idx, int64_t n_obs,
                              int32_t n_cams_hint, int32_t n_pts_hint, int check, void* stream,
                              sfm_ba_plan** out_plan);
void sfm_ba_plan_destroy(sfm_ba_plan* plan);
/* Copies one index array of the plan to the host (for tools and tests; the reference exposes the
 * same information as attributes of `Indices`, src/sparse_ba.py:64-97).  field: 0 pt_ptr, 1 perm,
 * 2 slot_cam, 3 tp_ptr, 4 pair_entry, 5 grp_pt, 6 cam_obs (empty when the caller order is
 * camera-sorted), 7-9 per-camera items (camera, begin, end), 10 cam_item_ptr, 11-12 block cameras
 * (j, k), 13 pair entries (two uint32 each), 14-17 pair items (block, begin, end, partial), 18-20
 * split blocks (block, first partial, partials); the fused assembly (all empty when the plan has no
 * runs), per class of near points - fields 21-28 for class 0 (windows of up to 8 cameras), 121-128 for
 * class 1 (windows of 16): 21 first position of each run, 22 window of each run, 23 observation per
 * (position, window slot), 24 cameras of each window, 25 first run of each window, 26 first run of every
 * range of positions one wave walks, 27 point id at each position, 28 cameras of each run (for the
 * kernels' loads); 29-30 per block the (class, window, slot, slot) entries the finishing kernel adds,
 * 31-32 per block the far plan's partial results, 33 far point ids, 34-37 per far observation:
 * observation, camera, far point number, point id.  *bytes receives the array's size; h_dst may be
 * null to query it, otherwise `capacity` must be at least that size. */
int sfm_ba_plan_read(const sfm_ba_plan* plan, int field, void* h_dst, int64_t capacity, int64_t* bytes);
/* sizes: what[0]=n_cams, [1]=n_pts, [2]=n_obs, [3]=number of camera-pair blocks (j<=k) with a shared
 * point, [4]=number of (obs,obs) pair entries, [5]=number of per-point pair slots (doubles/4 of d_T),
 * [6]=number of per-camera work items, [7]=bytes of device memory held by the plan */
int sfm_ba_plan_sizes(const sfm_ba_plan* plan, int64_t what[8]);
/* Half bandwidth of the reduced camera matrix this plan's observations produce: S[r][c] = 0
 * wherever |r - c| exceeds it (7 (k - j) + 6 for the furthest-apart cameras j <= k that share a
 * point).  `Indices.shared_point_indices` (src/sparse_ba.py:107-128) is where the reference finds
 * the same empty blocks, one pair at a time.  A sharded run takes the maximum over its ranks. */
int32_t sfm_ba_plan_half_bandwidth(const sfm_ba_plan* plan);
/* The fused assembly and its near / far split of the points.  A point seen by few cameras is "near":
 * the plan sorts the near points by the window of cameras they are seen from - consecutive cameras where
 * the point's cameras span no more than the window (what an incremental reconstruction produces:
 * `Map.update_map`, src/map.py:17-37, appends points as frames are registered; any order of the point
 * ids), otherwise the point's own camera set - and cuts that order into runs of one window each;
 * sfm_ba_schur_assemble_state (and sfm_ba_schur_assemble called without d_B, weights, d_U, d_eps_a and
 * d_cam_red, with mu > 0) then forms V^-1, eps_b and the near points' share of S and e in a single pass.
 * Two classes of near points: class 0 with windows of 4..8 cameras (the width is chosen per plan from the
 * observations' histogram), class 1 with windows of 16 cameras for the points class 0 leaves over (the
 * longer tracks).  Points with more cameras, and points whose window would hold too few points (4 in
 * class 0, 16 in class 1), are "far": they form a plan of their own (sfm_ba_plan_far, NULL when there is
 * none) that goes through the camera-pair kernels, and the finishing kernel adds all shares into S.  The
 * reference's `Indices` / calc_S (src/sparse_ba.py:58-97, 237-258) take any index order and any camera
 * set; so does this.
 *   info[8 c + 0] = 1 when class c will be used, [8 c + 1] = cameras per window (0: the class is empty),
 *   [8 c + 2] = near points, [8 c + 3] = largest run length in points, [8 c + 4] = runs,
 *   [8 c + 5] = windows, [8 c + 6] = steps (groups of up to 8 / 4 consecutive points of one run),
 *   [8 c + 7] = ranges of positions (one wave each).  sfm_ba_plan_set_fused(plan, 0) turns the path off
 *   for this plan (tests compare the two paths); both give the same S and e to rounding. */
int sfm_ba_plan_fused_info(const sfm_ba_plan* plan, int32_t info[16]);
int sfm_ba_plan_set_fused(sfm_ba_plan* plan, int enable);
/* the plan over the far points' observations (owned by `plan`; NULL: every point is near, or the plan
 * has no fused path).  Its points are the far points renumbered 0 .. in ascending id. */
const sfm_ba_plan* sfm_ba_plan_far(const sfm_ba_plan* plan);
/* bytes of the scratch workspace the assemble / back-substitution entry points need */
int64_t sfm_ba_plan_workspace_bytes(const sfm_ba_plan* plan);

/* ------------------------------------------------------------------------------------------
 * Per-camera records.  Work of `rotation_matrix_to_quaternion` (src/utils.py:62-105, four
 * branches, quaternion order [x,y,z,w]), `compute_dR_dQ_mat` (src/jacobian.py:171-203, signs as
 * written there) and the mixed rows fx*R0+ox*R2, fy*R1+oy*R2, R2 that every Jacobian function
 * recomputes per observation (src/jacobian.py:29-49).
 *   d_R [m,9] row-major rotations; d_C [m,3] centres; d_K [m,9] or [1,9] row-major intrinsics
 *   (k_stride = 9 or 0; the Jacobians read fx=K[0], fy=K[4], ox=K[2], oy=K[5] only, as
 *   src/jacobian.py:22-25 does); d_cam_blk [m, SFM_BA_CAM_BLK] out:
 *     [0..8] mixed rows, [9..11] C, [12..47] dR/dQ (9x4 row-major), [48..51] Q, [52..55] fx,fy,ox,oy
 *   d_flags [m] int32 in/out: SFM_BA_FLAG_DET_NOT_ONE is OR-ed in where the reference would assert.
 *   Flag words are sticky: the kernels only ever set bits, so a caller that enqueues several
 *   iterations and checks once sees what any of them raised; the caller zeroes the array.
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_camera_prepare(const double* d_R, const double* d_C, const double* d_K, int k_stride,
                          int32_t m, double* d_cam_blk, int32_t* d_flags, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-observation Jacobian blocks and predictions.  Work of `compute_A_B`
 * (src/bundle_adjustment.py:22-41) = `compute_pose_jacobian_mat` + `compute_X_jacobian_mat`
 * (src/jacobian.py:215-220, :75-126) per observation, and of the projection inside
 * `get_indices` (src/bundle_adjustment.py:55, src/utils.py:8-11).
 *   d_P [m,12] projection matrices K[R|T] (may be NULL when d_x_pred is NULL)
 *   d_X [n,3] points; d_cam_idx, d_pt_idx [N]
 *   d_A [N,2,7] = [dF/dQ | dF/dC], d_B [N,2,3] = dF/dX  (either may be NULL to skip)
 *   d_x_pred [N,2] (may be NULL)
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_jacobian(const double* d_cam_blk, const double* d_P, const double* d_X,
                    const int32_t* d_cam_idx, const int32_t* d_pt_idx, int64_t n_obs,
                    double* d_A, double* d_B, double* d_x_pred, void* stream);

/* ------------------------------------------------------------------------------------------
 * Schur assembly.  Work of calc_U, calc_V_inv, calc_W, calc_Y, calc_S, calc_epsilon,
 * calc_epsilon_a, calc_epsilon_b and calc_e (src/sparse_ba.py:147-271) as called by
 * `SBA.compute` (src/sparse_ba.py:393-402).
 *   d_B may be NULL here and in sfm_ba_back_substitute: the point block is then taken as minus
 *   the camera-centre block of d_A (columns 4..6), which is what the reference's Jacobians satisfy
 *   exactly (src/jacobian.py:29-49: dF/dC = -dF/dX); the loop then never materialises B.
 *   d_A [N,2,7], d_B [N,2,3], d_x_true [N,2], d_x_pred [N,2] in the observation order the plan
 *   was built from; d_weights [N,2,2] symmetric or NULL (identity, src/sparse_ba.py:387-388)
 *   mu     : damping added to every V_i (src/sparse_ba.py:206,210)
 *   mu_u   : damping added to every U_j (src/sparse_ba.py:194,197); a rank of a sharded run
 *            passes 0 and adds mu after the all-reduce (sfm_ba_add_diagonal)
 * Outputs:
 *   d_S   [(7m+1), 7m] row-major: rows 0..7m-1 = S (both triangles; blocks of camera pairs
 *         without a shared point are left untouched - zero the buffer once before the first call),
 *         row 7m = e flattened.  With d_U, d_eps_a and d_cam_red all NULL (a caller that only
 *         wants the solve) the blocks above the diagonal are not written: the solvers read the
 *         lower triangle only
 *   d_e [m,7], d_U [m,7,7], d_eps_a [m,7]      (d_U, d_eps_a may be NULL)
 *   d_V_inv [n,3,3], d_eps_b [n,3]
 *   d_cam_red [m, SFM_BA_CAM_RED] per-camera sums without damping (may be NULL): what a sharded
 *         run all-reduces; [42] is the camera's sum of squared residuals
 *   d_workspace : sfm_ba_plan_workspace_bytes(plan) bytes; holds per-observation and per-pair
 *         intermediates that sfm_ba_back_substitute reads again
 * V_i is inverted in closed form; the reference uses numpy.linalg.pinv, identical for the
 * full-rank, well-conditioned V_i that any mu > 0 produces.
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_schur_assemble(const sfm_ba_plan* plan, const double* d_A, const double* d_B,
                          const double* d_x_true, const double* d_x_pred, const double* d_weights,
                          double mu, double mu_u, double* d_S, double* d_e, double* d_U,
                          double* d_eps_a, double* d_V_inv, double* d_eps_b, double* d_cam_red,
                          void* d_workspace, void* stream);

/* The same assembly (calc_V_inv ... calc_e, src/sparse_ba.py:147-271, no weights, no by-products) for any
 * visibility inside a loop that owns the state: d_A holds the rows sfm_ba_jacobian wrote from the state
 * (d_cam_blk, d_X) - the per-point stage takes the point blocks out of them - and the camera-pair products
 * A_j^T T A_k are formed from the state inside the kernel: an entry costs its point's position and V^-1
 * (80 bytes) and an index instead of two rows of A and a coupling (256 bytes), and no coupling is written.
 * The rows formed there agree with d_A's to a few units in the last place (one reciprocal per quotient rule
 * instead of a division per entry).  Writes the lower triangle of d_S, row 7m, d_e (may be NULL), d_V_inv,
 * d_eps_b and - d_cam_ssq [m], may be NULL - per camera the sum of squared residuals of its observations
 * (see sfm_ba_schur_assemble_state).  mu_u as in sfm_ba_schur_assemble. */
int sfm_ba_schur_assemble_pairs_state(const sfm_ba_plan* plan, const double* d_A, const double* d_cam_blk,
                                      const double* d_X, const double* d_x_true, const double* d_x_pred, double mu,
                                      double mu_u, double* d_S, double* d_e, double* d_V_inv, double* d_eps_b,
                                      double* d_cam_ssq, void* d_workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Jacobian blocks and Schur assembly in one pass, for plans with a fused path (sfm_ba_plan_state_path
 * returns 1): the rows of A and the residuals - compute_A_B and the projection of get_indices
 * (src/bundle_adjustment.py:22-67, src/jacobian.py:21-220, src/utils.py:8-11) - are formed inside
 * the assembly kernel from the prepared camera records (sfm_ba_camera_prepare), the projection
 * matrices d_P [m,12], the points d_X [n,3] and d_x_true [N,2], and consumed there by calc_V_inv ...
 * calc_e (src/sparse_ba.py:147-271): for the near points A, B and x_pred are never written to memory
 * (the rows of the far points' observations go to the workspace and through the camera-pair kernels).
 * Outputs as sfm_ba_schur_assemble writes them without by-products (d_S lower triangle and row 7m, d_e,
 * d_V_inv, d_eps_b); mu > 0.  The rows agree with those sfm_ba_jacobian writes to a few units in the last
 * place (one reciprocal per quotient rule instead of a division per entry).
 *   d_cam_ssq [m] out, may be NULL: per camera the sum of squared residuals of its observations at this
 *   state - the sums `compute_reprojection_error_total` (src/bundle_adjustment.py:9-19) takes the roots
 *   of, a by-product of the residuals the kernels form anyway: sfm_ba_cost_finish(d_cam_ssq, ...) then
 *   gives the cost of the state the system was assembled at without a pass of its own (equal to
 *   sfm_ba_cost's to rounding; the sums are associated differently).  Entries of cameras without
 *   observations are not written (zero the array once).
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_plan_state_path(const sfm_ba_plan* plan);
int sfm_ba_schur_assemble_state(const sfm_ba_plan* plan, const double* d_cam_blk, const double* d_P, const double* d_X,
                                const double* d_x_true, double mu, double mu_u, double* d_S, double* d_e,
                                double* d_V_inv, double* d_eps_b, double* d_cam_ssq, void* d_workspace, void* stream);
/* calc_delta_b (src/sparse_ba.py:279-290) the same way: the rows of A are formed again from the camera
 * records and d_X (the state the system was assembled at: call it before the cameras are updated).
 * d_workspace: the workspace the preceding sfm_ba_schur_assemble_state call on this state used (it
 * holds the rows of the far points' observations, which are not formed a second time).
 * update_points != 0: d_X += x_step * delta_b afterwards (src/bundle_adjustment.py:100), point by point. */
int sfm_ba_back_substitute_state(const sfm_ba_plan* plan, const double* d_cam_blk, const double* d_V_inv,
                                 const double* d_eps_b, const double* d_delta_a, double* d_delta_b, double* d_X,
                                 int update_points, double x_step, void* d_workspace, void* stream);

/* W [N,7,3] and Y [N,7,3] as the reference materialises them (calc_W, calc_Y,
 * src/sparse_ba.py:215-234).  Not needed by the solve; provided for callers that inspect them. */
int sfm_ba_calc_W_Y(const sfm_ba_plan* plan, const double* d_A, const double* d_B,
                    const double* d_weights, const double* d_V_inv, const int32_t* d_pt_idx,
                    double* d_W, double* d_Y, void* stream);

/* Rebuild S rows/e from all-reduced per-camera records and add the damping: S_jj += U_j + mu I
 * is already inside S for a single-rank run; a sharded run calls this after the all-reduce to
 * add mu to the 7m diagonal entries of d_S (the one place mu enters U, src/sparse_ba.py:194-197). */
int sfm_ba_add_diagonal(double* d_S, int32_t n7, double mu, void* stream);

/* One read for everything the reference's loop looks at after an iteration
 * (src/bundle_adjustment.py:75-105: the cost it prints and compares, and the exceptions numpy or its
 * own asserts would have raised): d_status [3] doubles = { d_cost[0], d_info[0], OR over the m
 * cameras of d_flags_a[j] | d_flags_b[j] }.  Any input may be NULL (reads as 0). */
int sfm_ba_iteration_status(const double* d_cost, const int32_t* d_info, const int32_t* d_flags_a,
                            const int32_t* d_flags_b, int32_t m, double* d_status, void* stream);

/* What the ranks of a sharded run exchange when S is dense: the lower triangle of d_S (row i:
 * columns 0 .. i, at offset i (i + 1) / 2) followed by the right-hand side (row n7 of the system
 * buffer) - sfm_ba_lower_elements(n7) = n7 (n7 + 1) / 2 + n7 doubles, half of the buffer.  The sums
 * over points of calc_S / calc_e (src/sparse_ba.py:237-271) are formed per rank; S is symmetric and
 * the solvers read its lower triangle only.  sfm_ba_lower_pack fills d_tri from d_S, the ranks
 * all-reduce d_tri, sfm_ba_lower_unpack writes it back into the lower triangle and row n7 of d_S
 * and adds mu to the diagonal (the one place mu enters U, src/sparse_ba.py:194-197). */
int64_t sfm_ba_lower_elements(int32_t n7);
int sfm_ba_lower_pack(const double* d_S, int32_t n7, double* d_tri, void* stream);
int sfm_ba_lower_unpack(const double* d_tri, int32_t n7, double mu, double* d_S, void* stream);

/* ------------------------------------------------------------------------------------------
 * Reduced camera system.  Work of `calc_delta_a` (src/sparse_ba.py:274-276): solves
 * S delta_a = e.  The reference calls LAPACK dgesv (LU); S is symmetric positive definite
 * (it is the Schur complement of J^T J + mu I), and this library factors it by a blocked
 * Cholesky on the device, in place.
 *   d_S [(n7+1), n7] as written by sfm_ba_schur_assemble (row n7 = right-hand side); destroyed
 *   d_delta_a [n7] out
 *   d_chol_ws : sfm_ba_cholesky_workspace_bytes(n7) bytes
 *   d_info [1] int32 in/out: zeroed by the caller; a factorisation that meets a non-positive pivot
 *          writes 1+its index if the word still holds 0 (SFM_BA_ENOTPD semantics), so the word keeps
 *          the first failure of all the solves enqueued since the caller cleared it; read it after
 *          synchronising the stream
 * ---------------------------------------------------------------------------------------- */
int64_t sfm_ba_cholesky_workspace_bytes(int32_t n7);
int sfm_ba_solve_reduced(double* d_S, int32_t n7, double* d_delta_a, void* d_chol_ws,
                         int32_t* d_info, void* stream);
/* The same solve (src/sparse_ba.py:274-276) for a caller that knows S[r][c] = 0 wherever
 * |r - c| > half_bandwidth (sfm_ba_plan_half_bandwidth).  With half_bandwidth <= 64 the factorisation
 * is a chain of 64 x 64 blocks that one workgroup walks in a single launch, substitutions included,
 * and d_S is only read; with 64 < half_bandwidth <= 256 and enough unknowns (7m above 384 for a
 * band up to 128, above 1280 up to 256) the same reduction runs on superblocks of 128 or 256
 * unknowns, d_S again only read; anything else takes the general path of sfm_ba_solve_reduced
 * (d_S destroyed).  The band paths read whole blocks - entries up to twice the block edge
 * (sfm_ba_reduced_solve_block) below the diagonal, which must hold their zeros - and nothing beyond. */
int sfm_ba_solve_reduced_banded(double* d_S, int32_t n7, int32_t half_bandwidth, double* d_delta_a,
                                void* d_chol_ws, int32_t* d_info, void* stream);
/* Which of those paths a size and band take: the edge of the blocks the solve works on - 64 (cyclic
 * reduction / block chain), 128 or 256 (reduction on superblocks) - or 0 for the general
 * factorisation.  No counterpart in the reference (numpy.linalg.solve has one path). */
int32_t sfm_ba_reduced_solve_block(int32_t n7, int32_t half_bandwidth);
/* The system of src/sparse_ba.py:274-276 with its cameras renumbered, for a caller that assembles in
 * the frames' order and solves in an order with a narrower band: row / column 7 p + a of d_S_out is
 * row / column 7 d_cam_order[p] + a of d_S (both [(n7+1) x n7], right-hand side in row n7; d_cam_order
 * [n7 / 7] = the camera at position p: a permutation of 0 .. n7/7 - 1, which the call cannot check -
 * an index outside that range is an out-of-bounds read on the device).  Only the lower triangle of d_S is read; of d_S_out the
 * lower-triangle entries within half_bandwidth of the diagonal and row n7 are written (pass n7 to
 * get the whole lower triangle; for a band path pass 2 * sfm_ba_reduced_solve_block(...) - 1, what
 * that path reads).  The solution of the renumbered system is delta_a in the new order:
 * delta_a of camera d_cam_order[p] is its block p. */
int sfm_ba_permute_system(const double* d_S, int32_t n7, const int32_t* d_cam_order, int32_t half_bandwidth,
                          double* d_S_out, void* stream);


/* Packed form of a narrow-band system (half bandwidth <= 64), for sharded runs: S is block
 * tridiagonal in 64 x 64 blocks, and [diagonal blocks | couplings | right-hand side] is all of it -
 * 8 (2 * 64 * 7m + 7m) bytes rounded up to whole blocks instead of the 8 (7m + 1) 7m of the dense
 * buffer (3.6 MB instead of 98 MB at m = 500).  Packing is a copy, so ranks pack their local sums,
 * all-reduce the packed buffers and solve from the packed form; `mu_diag` is the damping of U
 * (src/sparse_ba.py:194-197), added once after the reduction.  No counterpart in the reference,
 * whose calc_S (src/sparse_ba.py:237-258) always fills the dense matrix.
 *   d_packed  : sfm_ba_band_packed_elements(n7) doubles; destroyed by the solve
 *   d_chol_ws : sfm_ba_cholesky_workspace_bytes(n7) bytes
 * Entries of d_S outside the band are not read. */
int64_t sfm_ba_band_packed_elements(int32_t n7);
int sfm_ba_band_pack(const double* d_S, int32_t n7, double* d_packed, void* stream);
/* What the ranks of a sharded run exchange when the band is narrow: per 64-row block only the lower
 * triangle of the diagonal block, the coupling entries a half bandwidth hb <= 64 leaves non-zero
 * (S(k1 + r, k0 + c) with c >= r + 64 - hb) and the right-hand side - about half of the packed
 * form.  sfm_ba_band_compact fills it from S, the ranks all-reduce it, sfm_ba_band_expand turns
 * it into the packed form sfm_ba_solve_band_packed takes. */
int64_t sfm_ba_band_compact_elements(int32_t n7, int32_t half_bandwidth);
int sfm_ba_band_compact(const double* d_S, int32_t n7, int32_t half_bandwidth, double* d_compact, void* stream);
int sfm_ba_band_expand(const double* d_compact, int32_t n7, int32_t half_bandwidth, double* d_packed,
                       void* stream);
int sfm_ba_solve_band_packed(double* d_packed, int32_t n7, double mu_diag, double* d_delta_a,
                             void* d_chol_ws, int32_t* d_info, void* stream);

/* ------------------------------------------------------------------------------------------
 * Back-substitution.  Work of `calc_delta_b` (src/sparse_ba.py:279-290):
 * delta_b_i = V_inv_i (eps_b_i - sum_j W_ij^T delta_a_j).
 *   d_delta_b [n,3] out.  If d_X_update is not NULL, X_update[i] += x_step * delta_b[i]
 *   (the point update of src/bundle_adjustment.py:100 with x_step = 0.5).
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_back_substitute(const sfm_ba_plan* plan, const double* d_A, const double* d_B,
                           const double* d_weights, const double* d_V_inv, const double* d_eps_b,
                           const double* d_delta_a, double* d_delta_b, double* d_X_update,
                           double x_step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Camera update.  Work of the per-frame body of `optimize_using_BA`
 * (src/bundle_adjustment.py:82-98) with `quaternion_to_rotation_matrix` (src/utils.py:25-59) and
 * `Frame.compute_projection_matrix` (src/frame.py:50-53): for every camera >= first_cam
 * Q = quat(R) + q_step*delta_a[0:4], normalised; R = rot(Q); C' = C + c_step*delta_a[4:7];
 * T = -R C'; C = -R^T T; P = K [R|T] (d_K [m,9] or [1,9], k_stride 9 or 0; d_P may be NULL).
 * The reference skips frame 0 (first_cam = 1) and uses
 * q_step = 1, c_step = 0.1.  SFM_BA_FLAG_ZERO_QUAT is OR-ed into d_flags [m] where the reference
 * would raise ValueError (sticky, see sfm_ba_camera_prepare).
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_update_cameras(double* d_R, double* d_T, double* d_C, double* d_P, const double* d_K,
                          int k_stride, const double* d_delta_a, int32_t m, int32_t first_cam,
                          double q_step, double c_step, int32_t* d_flags, void* stream);
/* The same, and the prepared record of every camera in its new state (d_cam_blk, d_blk_flags as
 * sfm_ba_camera_prepare writes them): the next iteration then starts at sfm_ba_jacobian. */
int sfm_ba_update_cameras_prepare(double* d_R, double* d_T, double* d_C, double* d_P, const double* d_K,
                                  int k_stride, const double* d_delta_a, int32_t m, int32_t first_cam,
                                  double q_step, double c_step, int32_t* d_flags, double* d_cam_blk,
                                  int32_t* d_blk_flags, void* stream);
/* The same for a caller that keeps the cameras in another order than the frames' (an order that
 * narrows the band of S, see sfm_ba_solve_reduced_banded): every camera but held_cam is updated
 * (held_cam = -1: all of them) - the frame the loop of src/bundle_adjustment.py:84 starts after
 * need not sit at index 0.  d_cam_blk / d_blk_flags may be NULL (no records prepared). */
int sfm_ba_update_cameras_held(double* d_R, double* d_T, double* d_C, double* d_P, const double* d_K,
                               int k_stride, const double* d_delta_a, int32_t m, int32_t held_cam,
                               double q_step, double c_step, int32_t* d_flags, double* d_cam_blk,
                               int32_t* d_blk_flags, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cost.  Work of `compute_reprojection_error_total` (src/bundle_adjustment.py:9-19): per camera
 * the sum of squared residuals of its observations; d_cam_ssq [m] out.  The caller (after an
 * all-reduce in a sharded run) takes mean_j sqrt(ssq_j); sfm_ba_cost_finish does that on the
 * device: d_cost[0] = sum_j sqrt(d_cam_ssq[j]) / m.
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_cost_partial(const sfm_ba_plan* plan, const double* d_P, const double* d_X,
                        const int32_t* d_cam_idx, const int32_t* d_pt_idx, const double* d_x_true,
                        double* d_cam_ssq, void* d_workspace, void* stream);
int sfm_ba_cost_finish(const double* d_cam_ssq, int32_t m, double* d_cost, void* stream);
/* Both steps in two launches instead of three, for a run that has nothing to all-reduce in between:
 * d_cam_ssq [m] and d_cost[0] as above, bit for bit. */
int sfm_ba_cost(const sfm_ba_plan* plan, const double* d_P, const double* d_X, const int32_t* d_cam_idx,
                const int32_t* d_pt_idx, const double* d_x_true, double* d_cam_ssq, double* d_cost,
                void* d_workspace, void* stream);

/* The same cost (d_cam_ssq [m], d_cost[0], bit for bit what sfm_ba_cost writes) for callers that cannot
 * allocate or hand over a workspace: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  No counterpart in the reference.  The per-item partial sums go to a device
 * array that is part of the library (512 KiB on every GPU the library runs on, kept until the process
 * ends), and the call performs no allocation of any kind (no timing events either: it is never part
 * of the per-kernel profile).  Limit: SFM_BA_COST_NOALLOC_MAX_ITEMS per-camera work items, i.e. sum over
 * cameras of ceil(observations of the camera / 1024); a larger plan is refused with SFM_BA_ETOOBIG.
 * All calls on one GPU share the array: calls that overlap on one GPU (different streams) race.
 * The array's address on a device is looked up (hipGetSymbolAddress, once per device) when a plan is
 * created on that device, so the call itself issues its two launches and nothing else and may be
 * captured into a graph; with a plan created on another device the first call does the lookup. */
#define SFM_BA_COST_NOALLOC_MAX_ITEMS 65536
int sfm_ba_cost_noalloc(const sfm_ba_plan* plan, const double* d_P, const double* d_X, const int32_t* d_cam_idx,
                        const int32_t* d_pt_idx, const double* d_x_true, double* d_cam_ssq, double* d_cost,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * Two-view linear triangulation: the step that produces the points bundle adjustment starts from.
 * Work of `triangulate_pts` (src/triangulation.py:10-32): per correspondence the null direction of
 * [ [x1]_x P1 ; [x2]_x P2 ] (src/utils.py:4-5 for [v]_x), divided by its last entry, as float32.
 *   d_x1, d_x2 [n,3] homogeneous pixel coordinates (fp64); d_P1, d_P2 [12] row-major 3x4
 *   d_Xh [n,4] float32 out (last column 1)
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_triangulate(const double* d_x1, const double* d_x2, const double* d_P1, const double* d_P2,
                       int64_t n, float* d_Xh, void* stream);

/* ------------------------------------------------------------------------------------------
 * Linear PnP: pose of a camera from 3D-2D correspondences, the other producer of the poses bundle
 * adjustment starts from.  sfm_ba_pnp_solve does the work of `estimate_pose_Linear_PnP_n`
 * (src/pose_estimation.py:73-105) or `estimate_pose_Linear_PnP` (:41-70) for n_hyp subsets of the
 * correspondences at once - the 200 minimal samples of `estimate_pose_Linear_PnP_RANSAC`
 * (:108-144) are one call - and sfm_ba_pnp_inliers counts, per pose, the correspondences that
 * reproject within the threshold (:123-133).
 *   d_x [n,3] homogeneous pixels, d_X [n,4] homogeneous points (fp64)
 *   d_Kinv_pre  [9] or NULL : K^-1 applied to x before the rows are built       (:85, PnP_n)
 *   d_Kinv_post [9] or NULL : K^-1 applied to the recovered 3x4 matrix          (:34, PnP)
 *   d_idx [n_hyp, n_per_hyp] rows of d_x / d_X each hypothesis uses, or NULL for 0..n_per_hyp-1
 *   d_workspace : sfm_ba_pnp_workspace_bytes(n_hyp, n_per_hyp) bytes
 *   d_R [n_hyp,9], d_T [n_hyp,3] out: rotation (det +1) and translation after `cleanup_RT_mat`
 *   (:21-30); d_flags [n_hyp] out: 1 where det(U V^T) rounds to neither +1 nor -1 (the reference
 *   then fails with an unassigned result)
 *   sfm_ba_pnp_inliers: d_K [9]; d_counts [n_hyp] out; d_mask [n_hyp, n] bytes out or NULL
 * The right singular vector comes from one-sided Jacobi rotations on the stacked rows (blocks of
 * 192 rows, reduced by orthogonal transformations only), not from the normal equations.
 * ---------------------------------------------------------------------------------------- */
int64_t sfm_ba_pnp_workspace_bytes(int32_t n_hyp, int32_t n_per_hyp);
int sfm_ba_pnp_solve(const double* d_x, const double* d_X, const double* d_Kinv_pre, const double* d_Kinv_post,
                     const int32_t* d_idx, int32_t n_hyp, int32_t n_per_hyp, void* d_workspace, double* d_R,
                     double* d_T, int32_t* d_flags, void* stream);
/* `cleanup_RT_mat` / `decompose_projection_mat` alone (src/pose_estimation.py:21-38) for n 3x4
 * matrices d_RT [n,12] row-major (K^-1 applied first when d_Kinv_post is given) */
int sfm_ba_pnp_cleanup(const double* d_RT, const double* d_Kinv_post, int32_t n, double* d_R, double* d_T,
                       int32_t* d_flags, void* stream);
int sfm_ba_pnp_inliers(const double* d_x, const double* d_X, const double* d_K, const double* d_R,
                       const double* d_T, int32_t n_hyp, int64_t n, double threshold, int32_t* d_counts,
                       uint8_t* d_mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing (no counterpart in the reference, which has no profiler).  While enabled,
 * every kernel launch made by the entry points above is bracketed by two HIP events recorded on
 * the launch stream; sfm_ba_profile_read waits for them, adds the elapsed milliseconds and launch
 * counts per kernel id into the two arrays of sfm_ba_profile_kernel_count() entries and forgets
 * them.  "solve_reduced" is one id for the whole factor-and-solve sequence.
 *   sfm_ba_profile_enable(mode): 0 off; 1 on; 2 on, and a timed launch that directly follows another
 *   timed launch on the same stream starts at that launch's stop event instead of recording one of
 *   its own (half the event records; the caller asserts that it enqueues nothing else on the stream
 *   between the library's calls - anything it did enqueue would be counted into the next launch).
 *   sfm_ba_profile_select(mask): bit i set = kernel id i is timed (default: all).
 * ---------------------------------------------------------------------------------------- */
int sfm_ba_profile_enable(int mode);
int sfm_ba_profile_select(uint64_t kernel_mask);
int sfm_ba_profile_kernel_count(void);
const char* sfm_ba_profile_kernel_name(int id);
int sfm_ba_profile_read(double* ms_total, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* SFM_BA_H */
