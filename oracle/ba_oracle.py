"""CPU oracle for the bundle-adjustment hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's Gauss-Newton inner loop
(per-observation Jacobians, Schur assembly, reduced solve, back-substitution and
the outer update loop).  It exists so that the HIP path can be checked on any
machine: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it, and only as the checker / timed CPU comparator.
The product package never imports it.

Parity pin: the reference has no test, golden vector or known-answer file for
this path, so the pin is ``tests/golden/*.npz`` - outputs of the reference's own
modules run in the build container by ``tests/golden/make_golden.py``.
``tests/test_oracle_golden.py`` checks every function below against them.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).  Loops of the reference over observations / points /
cameras are restated as whole-array numpy operations; sums are taken in the same
index order (``np.add.at`` / ``np.bincount`` accumulate sequentially).
"""
from __future__ import annotations

import numpy as np

try:  # scipy is only needed for the large-problem form of calc_S
    import scipy.sparse as _sp
except Exception:  # pragma: no cover
    _sp = None


# --------------------------------------------------------------------------- #
# src/utils.py
# --------------------------------------------------------------------------- #
def project_3D_to_2D(X, P):
    """src/utils.py:8-11 - homogeneous projection, rows divided by the last row."""
    x = np.dot(P, X.T)
    return (x / x[-1]).T


def quaternion_to_rotation_matrix(q):
    """src/utils.py:25-59 - q = [x, y, z, w] (the docstring there says w first; the
    unpacking at :40 is x, y, z, w).  Asserts unit norm (:37-38)."""
    q = np.asarray(q, dtype=np.float64)
    assert q.shape[-1] == 4, "Input quaternion must have 4 elements"
    assert np.all(np.isclose(np.linalg.norm(q, axis=-1), 1.0)), "Input quaternion must be unit quaternion"
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    xx, xy, xz, xw = x * x, x * y, x * z, x * w
    yy, yz, yw = y * y, y * z, y * w
    zz, zw = z * z, z * w
    R = np.stack(
        [
            np.stack([1 - 2 * (yy + zz), 2 * (xy - zw), 2 * (xz + yw)], -1),
            np.stack([2 * (xy + zw), 1 - 2 * (xx + zz), 2 * (yz - xw)], -1),
            np.stack([2 * (xz - yw), 2 * (yz + xw), 1 - 2 * (xx + yy)], -1),
        ],
        -2,
    )
    return R


def rotation_matrix_to_quaternion(R):
    """src/utils.py:62-105 - four-branch conversion, returns [x, y, z, w].
    Batched over leading axes; asserts det(R) ~ 1 (:74-76)."""
    R = np.asarray(R, dtype=np.float64)
    single = R.ndim == 2
    Rb = R.reshape(-1, 3, 3)
    assert np.all(np.isclose(np.linalg.det(Rb), 1.0)), "Input matrix must have determinant 1 (be a rotation matrix)"
    out = np.empty((Rb.shape[0], 4))
    tr = Rb[:, 0, 0] + Rb[:, 1, 1] + Rb[:, 2, 2]
    r00, r11, r22 = Rb[:, 0, 0], Rb[:, 1, 1], Rb[:, 2, 2]
    b1 = tr > 0
    b2 = ~b1 & (r00 > r11) & (r00 > r22)
    b3 = ~b1 & ~b2 & (r11 > r22)
    b4 = ~b1 & ~b2 & ~b3
    if b1.any():
        M = Rb[b1]
        s = 0.5 / np.sqrt(tr[b1] + 1.0)
        out[b1] = np.stack(
            [(M[:, 2, 1] - M[:, 1, 2]) * s, (M[:, 0, 2] - M[:, 2, 0]) * s, (M[:, 1, 0] - M[:, 0, 1]) * s, 0.25 / s], -1
        )
    if b2.any():
        M = Rb[b2]
        s = 2.0 * np.sqrt(1.0 + M[:, 0, 0] - M[:, 1, 1] - M[:, 2, 2])
        out[b2] = np.stack(
            [0.25 * s, (M[:, 0, 1] + M[:, 1, 0]) / s, (M[:, 0, 2] + M[:, 2, 0]) / s, (M[:, 2, 1] - M[:, 1, 2]) / s], -1
        )
    if b3.any():
        M = Rb[b3]
        s = 2.0 * np.sqrt(1.0 + M[:, 1, 1] - M[:, 0, 0] - M[:, 2, 2])
        out[b3] = np.stack(
            [(M[:, 0, 1] + M[:, 1, 0]) / s, 0.25 * s, (M[:, 1, 2] + M[:, 2, 1]) / s, (M[:, 0, 2] - M[:, 2, 0]) / s], -1
        )
    if b4.any():
        M = Rb[b4]
        s = 2.0 * np.sqrt(1.0 + M[:, 2, 2] - M[:, 0, 0] - M[:, 1, 1])
        out[b4] = np.stack(
            [(M[:, 0, 2] + M[:, 2, 0]) / s, (M[:, 1, 2] + M[:, 2, 1]) / s, 0.25 * s, (M[:, 1, 0] - M[:, 0, 1]) / s], -1
        )
    return out[0] if single else out.reshape(R.shape[:-2] + (4,))


# --------------------------------------------------------------------------- #
# src/jacobian.py
# --------------------------------------------------------------------------- #
def compute_dR_dQ_mat(Q):
    """src/jacobian.py:171-203 - 9x4 derivative of the rotation entries w.r.t.
    (x, y, z, w), reproduced entry by entry (rows R11,R12,R13,R21,...,R33).
    Three entries carry the opposite sign of the true derivative of the matrix
    in src/utils.py:51-57 (row R12 col w at :179, row R23 col w at :183, row R31
    col z at :184); they are part of the contract and are kept."""
    Q = np.asarray(Q, dtype=np.float64)
    x, y, z, w = Q[..., 0], Q[..., 1], Q[..., 2], Q[..., 3]
    o = np.zeros_like(x)
    rows = [
        [o, -4 * y, -4 * z, o],
        [2 * y, 2 * x, -2 * w, 2 * z],
        [2 * z, 2 * w, 2 * x, 2 * y],
        [2 * y, 2 * x, 2 * w, 2 * z],
        [-4 * x, o, -4 * z, o],
        [-2 * w, 2 * z, 2 * y, 2 * x],
        [2 * z, -2 * w, -2 * x, -2 * y],
        [2 * w, 2 * z, 2 * y, 2 * x],
        [-4 * x, -4 * y, o, o],
    ]
    return np.stack([np.stack(r, -1) for r in rows], -2)


def _uvw(K, R, C, X):
    """Shared head of src/jacobian.py:21-49, :75-103, :129-157: mixed rows
    m_u = fx R0 + ox R2, m_v = fy R1 + oy R2, m_w = R2 and u, v, w = m . (X - C).
    K, R, C, X are batched over the leading axis: (N,3,3), (N,3,3), (N,3), (N,3)."""
    fx, fy, ox, oy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]
    mu_ = fx[:, None] * R[:, 0, :] + ox[:, None] * R[:, 2, :]
    mv_ = fy[:, None] * R[:, 1, :] + oy[:, None] * R[:, 2, :]
    mw_ = R[:, 2, :]
    d = X - C
    u = np.einsum("ni,ni->n", mu_, d)
    v = np.einsum("ni,ni->n", mv_, d)
    w = np.einsum("ni,ni->n", mw_, d)
    return (fx, fy, ox, oy), (mu_, mv_, mw_), d, (u, v, w)


def jacobians_batched(K, R, C, X, Q=None):
    """A (N,2,7) = [dF/dQ | dF/dC] and B (N,2,3) = dF/dX for N (camera, point)
    pairs given per-observation K, R, C (camera) and X (point).

    dF/dX: src/jacobian.py:75-126; dF/dC: :21-72 (the negated rows of dF/dX);
    dF/dR: :129-168; dF/dQ = dF/dR . dR/dQ: :206-212; A = hstack: :215-220.
    ``Q`` (N,4) may be passed to skip the per-observation rotation->quaternion
    conversion the reference performs at :207."""
    (fx, fy, ox, oy), (mu_, mv_, mw_), d, (u, v, w) = _uvw(K, R, C, X)
    w2 = (w**2)[:, None]
    B = np.stack([(w[:, None] * mu_ - u[:, None] * mw_) / w2, (w[:, None] * mv_ - v[:, None] * mw_) / w2], 1)
    dF_dC = np.stack([(w[:, None] * (-mu_) - u[:, None] * (-mw_)) / w2, (w[:, None] * (-mv_) - v[:, None] * (-mw_)) / w2], 1)
    z = np.zeros_like(d)
    du_dR = np.concatenate([fx[:, None] * d, z, ox[:, None] * d], 1)
    dv_dR = np.concatenate([z, fy[:, None] * d, oy[:, None] * d], 1)
    dw_dR = np.concatenate([z, z, d], 1)
    dF_dR = np.stack([(w[:, None] * du_dR - u[:, None] * dw_dR) / w2, (w[:, None] * dv_dR - v[:, None] * dw_dR) / w2], 1)
    if Q is None:
        Q = rotation_matrix_to_quaternion(R)
    dR_dQ = compute_dR_dQ_mat(Q)
    dF_dQ = np.einsum("nar,nrq->naq", dF_dR, dR_dQ)
    A = np.concatenate([dF_dQ, dF_dC], 2)
    return A, B


def compute_X_jacobian_mat(frame, X):
    """src/jacobian.py:75-126 for one frame and one X (3,1)."""
    _, B = jacobians_batched(frame.K[None], frame.R[None], frame.C.reshape(1, 3), np.asarray(X).reshape(1, 3))
    return B[0]


def compute_pose_jacobian_mat(frame, X):
    """src/jacobian.py:215-220 for one frame and one X (3,1)."""
    A, _ = jacobians_batched(frame.K[None], frame.R[None], frame.C.reshape(1, 3), np.asarray(X).reshape(1, 3))
    return A[0]


# --------------------------------------------------------------------------- #
# src/bundle_adjustment.py (harness)
# --------------------------------------------------------------------------- #
def compute_reprojection_error_total(frames, map_):
    """src/bundle_adjustment.py:9-19 - mean over frames of the Frobenius norm of
    the frame's (k,3) residual matrix (third column is identically zero)."""
    total = 0.0
    for frame in frames:
        X = map_.X[frame.index_kp_3d]
        x_proj = project_3D_to_2D(X, frame.P)
        x = frame.keypoints[frame.triangulated_idx]
        total += np.linalg.norm(x - x_proj)
    return total / len(frames)


def get_indices(frames, map_):
    """src/bundle_adjustment.py:44-67 - frame-major flattening of the observations."""
    viewpoints_indices, point_indices, xt, xp = [], [], [np.zeros((0, 2))], [np.zeros((0, 2))]
    for frame_n, frame in enumerate(frames):
        frame_x = frame.keypoints[frame.triangulated_idx][:, :2]
        frame_X = map_.X[frame.index_kp_3d]
        frame_proj_x = project_3D_to_2D(frame_X, frame.P)[:, :2]
        viewpoints_indices.extend([frame_n] * len(frame.triangulated_idx))
        point_indices.extend(frame.index_kp_3d)
        xt.append(frame_x)
        xp.append(frame_proj_x)
    return viewpoints_indices, point_indices, np.vstack(xt), np.vstack(xp)


def compute_A_B(frames, map_, viewpoints_indices, point_indices):
    """src/bundle_adjustment.py:22-41 - Jacobian blocks of every observation."""
    vi = np.asarray(viewpoints_indices, dtype=np.int64)
    pi = np.asarray(point_indices, dtype=np.int64)
    Ks = np.stack([f.K for f in frames])
    Rs = np.stack([f.R for f in frames])
    Cs = np.stack([np.asarray(f.C).reshape(3) for f in frames])
    Qs = rotation_matrix_to_quaternion(Rs)
    X = map_.X[pi][:, :3]
    return jacobians_batched(Ks[vi], Rs[vi], Cs[vi], X, Q=Qs[vi])


# --------------------------------------------------------------------------- #
# src/sparse_ba.py
# --------------------------------------------------------------------------- #
def identities2x2(n):
    """src/sparse_ba.py:15-18."""
    I = np.zeros((n, 2, 2))
    I[:, [0, 1], [0, 1]] = 1
    return I


def check_indices(viewpoint_indices, point_indices):
    """src/sparse_ba.py:52-62, :66-67, :89-91 - unique (i,j) pairs; every id in
    0..max present.  Returns (n_viewpoints, n_points)."""
    vi = np.asarray(viewpoint_indices)
    pi = np.asarray(point_indices)
    assert len(vi) == len(pi)
    unique = np.unique(np.vstack((pi, vi)), axis=1)
    if unique.shape[1] != len(pi):
        raise ValueError("Found non-unique (i, j) pair")
    m = int(vi.max()) + 1
    n = int(pi.max()) + 1
    assert len(np.unique(pi)) == n
    assert len(np.unique(vi)) == m
    return m, n


def check_args(m, n, n_visible, x_true, x_pred, A, B, weights, mu):
    """src/sparse_ba.py:21-49."""
    assert A.shape[0] == B.shape[0] == n_visible
    assert x_true.shape[0] == x_pred.shape[0] == n_visible
    assert A.shape[1] == B.shape[1] == 2
    assert mu >= 0
    if not (2 * n_visible >= A.shape[2] * m + B.shape[2] * n):
        raise ValueError("n_rows(J) must be greater than n_cols(J)")
    assert weights.shape[1:3] == (2, 2)
    if not np.allclose(weights[:, 0, 1], weights[:, 1, 0]):
        raise ValueError("All weights must be symmetric")


def _segment_sum(idx, values, size):
    """Sum rows of ``values`` (N, ...) into ``size`` bins in ascending-position
    order (what the reference's per-camera / per-point Python loops do)."""
    flat = values.reshape(values.shape[0], -1)
    out = np.empty((size, flat.shape[1]))
    for c in range(flat.shape[1]):
        out[:, c] = np.bincount(idx, weights=flat[:, c], minlength=size)
    return out.reshape((size,) + values.shape[1:])


def calc_S(vi, pi, m, U, Y, W, force_sparse=False):
    """src/sparse_ba.py:237-258 - S_jk = delta_jk U_j - sum over points seen by
    both j and k of Y_ij W_ik^T; blocks without shared points stay zero.

    Small problems: explicit per-point pair expansion accumulated with np.add.at.
    Large problems: the same sum written as the block-sparse product
    Ymat (7m x 3n) @ Wmat^T, evaluated by scipy.sparse."""
    p = U.shape[1]
    N = len(vi)
    n = int(pi.max()) + 1
    if not force_sparse and N <= 200_000:
        order = np.argsort(pi, kind="stable")
        ps = pi[order]
        starts = np.searchsorted(ps, np.arange(n))
        counts = np.diff(np.append(starts, N))
        S4 = np.zeros((m, m, p, p))
        for kk in np.unique(counts):
            pts = np.nonzero(counts == kk)[0]
            if kk == 0 or len(pts) == 0:
                continue
            obs = order[starts[pts][:, None] + np.arange(kk)[None, :]]  # (np, kk)
            for s in range(0, len(pts), 4096):
                ob = obs[s : s + 4096]
                contrib = np.einsum("pajc,pbkc->pabjk", Y[ob], W[ob])
                cj = np.broadcast_to(vi[ob][:, :, None], contrib.shape[:3])
                ck = np.broadcast_to(vi[ob][:, None, :], contrib.shape[:3])
                np.subtract.at(S4, (cj.reshape(-1), ck.reshape(-1)), contrib.reshape(-1, p, p))
        S = S4.transpose(0, 2, 1, 3).reshape(m * p, m * p).copy()
    else:
        q = W.shape[2]
        rows = (p * vi[:, None, None] + np.arange(p)[None, :, None] + np.zeros((1, 1, q), dtype=np.int64)).reshape(-1)
        cols = (q * pi[:, None, None] + np.zeros((1, p, 1), dtype=np.int64) + np.arange(q)[None, None, :]).reshape(-1)
        Ym = _sp.csr_matrix((Y.reshape(-1), (rows, cols)), shape=(p * m, q * n))
        Wm = _sp.csr_matrix((W.reshape(-1), (rows, cols)), shape=(p * m, q * n))
        S = -np.asarray((Ym @ Wm.T).todense())
    for j in range(m):
        S[p * j : p * (j + 1), p * j : p * (j + 1)] += U[j]
    return S


def sba_compute(viewpoint_indices, point_indices, x_true, x_pred, A, B, weights=None, mu=0.0, do_check_args=True, force_sparse=False):
    """src/sparse_ba.py:336-406 - one damped Gauss-Newton step.  Returns a dict
    with every intermediate the reference forms: U, V_inv, W, Y, S, epsilon,
    epsilon_a, epsilon_b, e, delta_a, delta_b."""
    vi = np.asarray(viewpoint_indices, dtype=np.int64)
    pi = np.asarray(point_indices, dtype=np.int64)
    m, n = check_indices(vi, pi)
    N = len(vi)
    if weights is None:
        weights = identities2x2(N)
    if do_check_args:
        check_args(m, n, N, x_true, x_pred, A, B, weights, mu)
    p, q = A.shape[2], B.shape[2]

    AtW = np.einsum("nap,nab->npb", A, weights)  # A^T w  (N, 7, 2)
    BtW = np.einsum("naq,nab->nqb", B, weights)  # B^T w  (N, 3, 2)
    # :189-198  U_j = sum A^T w A + mu I
    U = _segment_sum(vi, np.einsum("npb,nbr->npr", AtW, A), m) + mu * np.identity(p)
    # :201-212  V_i^-1 = pinv(sum B^T w B + mu I)
    V = _segment_sum(pi, np.einsum("nqb,nbr->nqr", BtW, B), n) + mu * np.identity(q)
    V_inv = np.linalg.pinv(V)
    # :215-225  W = A^T w B ; :228-234  Y = W V_inv[point]
    W = np.einsum("npb,nbq->npq", AtW, B)
    Y = np.einsum("npq,nqr->npr", W, V_inv[pi])
    # :237-258
    S = calc_S(vi, pi, m, U, Y, W, force_sparse=force_sparse)
    # :147-171
    epsilon = x_true - x_pred
    epsilon_a = _segment_sum(vi, np.einsum("npb,nb->np", AtW, epsilon), m)
    epsilon_b = _segment_sum(pi, np.einsum("nqb,nb->nq", BtW, epsilon), n)
    # :261-271  e_j = eps_a_j - sum_i Y_ij eps_b_i
    d = np.einsum("npq,nq->np", Y, epsilon_b[pi])
    e = epsilon_a - _segment_sum(vi, d, m)
    # :274-276
    delta_a = np.linalg.solve(S, e.flatten()).reshape(e.shape)
    # :279-290  delta_b_i = V_inv_i (eps_b_i - sum_j W_ij^T delta_a_j)
    dd = np.einsum("npq,np->nq", W, delta_a[vi])
    delta_b = np.einsum("nqr,nr->nq", V_inv, epsilon_b - _segment_sum(pi, dd, n))
    return dict(
        U=U, V_inv=V_inv, W=W, Y=Y, S=S, epsilon=epsilon, epsilon_a=epsilon_a,
        epsilon_b=epsilon_b, e=e, delta_a=delta_a, delta_b=delta_b,
    )


# --------------------------------------------------------------------------- #
# src/bundle_adjustment.py:70-105 (outer loop) and src/frame.py:50-53
# --------------------------------------------------------------------------- #
def apply_update(frames, map_, del_A, del_B):
    """src/bundle_adjustment.py:82-100: frame 0 is skipped; Q += dQ then
    normalised; C += 0.1 dC; T = -R C; projection rebuilt (src/frame.py:50-53);
    X += 0.5 dX."""
    for f, frame in enumerate(frames):
        if f == 0:
            continue
        Q = rotation_matrix_to_quaternion(frame.R)
        upd = del_A[f]
        new_Q = Q + upd[0:4]
        norm = np.linalg.norm(new_Q)
        if np.isclose(norm, 0.0):
            raise ValueError("Cannot normalize quaternion with zero norm")
        new_Q = new_Q / norm
        new_R = quaternion_to_rotation_matrix(new_Q)
        new_C = frame.C + 0.1 * upd[4:7].reshape(3, 1)
        new_T = -np.dot(new_R, new_C)
        frame.R = new_R
        frame.T = new_T
        frame.C = -np.dot(frame.R.T, frame.T)
        frame.RT = np.hstack((frame.R, frame.T))
        frame.P = np.dot(frame.K, frame.RT)
    map_.X[:, :3] = map_.X[:, :3] + 0.5 * del_B


def optimize_using_BA(frames, map_, max_iter=50, threshold=100.0, mu=0.5, min_iter=0):
    """src/bundle_adjustment.py:70-105 without the two plotting calls.  Returns
    the list of per-iteration costs (the values the reference prints at :102).
    ``min_iter`` forces iterations past the cost threshold (benchmark use)."""
    costs = []
    avg_error = 1e10
    n_iter = 0
    while (avg_error > threshold or n_iter < min_iter) and n_iter < max_iter:
        vi, pi, x_t, x_p = get_indices(frames, map_)
        A, B = compute_A_B(frames, map_, vi, pi)
        out = sba_compute(vi, pi, x_t, x_p, A, B, weights=None, mu=mu)
        apply_update(frames, map_, out["delta_a"], out["delta_b"])
        avg_error = compute_reprojection_error_total(frames, map_)
        costs.append(float(avg_error))
        n_iter += 1
    return costs


# --------------------------------------------------------------------------- #
# Array-form loop used for large problems and the CPU baseline timing
# --------------------------------------------------------------------------- #
def gn_iteration_arrays(K, R, T, X, cam_idx, pt_idx, x_true, mu=0.5, force_sparse=True):
    """One iteration of src/bundle_adjustment.py:75-101 on structure-of-arrays
    state (K (3,3); R (m,3,3); T (m,3); X (n,3)).  Updates R, T, X in place and
    returns (cost after the update, delta_a, delta_b)."""
    m = R.shape[0]
    vi = cam_idx.astype(np.int64)
    pi = pt_idx.astype(np.int64)
    C = -np.einsum("mji,mj->mi", R, T)
    P = K[None] @ np.concatenate([R, T[:, :, None]], 2)
    Xh = np.concatenate([X[pi], np.ones((len(pi), 1))], 1)
    proj = np.einsum("nij,nj->ni", P[vi], Xh)
    x_pred = proj[:, :2] / proj[:, 2:3]
    Ks = np.broadcast_to(K, (m, 3, 3))
    Qs = rotation_matrix_to_quaternion(R)
    A, B = jacobians_batched(Ks[vi], R[vi], C[vi], X[pi], Q=Qs[vi])
    out = sba_compute(vi, pi, x_true, x_pred, A, B, None, mu, force_sparse=force_sparse)
    da, db = out["delta_a"], out["delta_b"]
    newQ = Qs + da[:, :4]
    newQ[0] = Qs[0]
    nrm = np.linalg.norm(newQ, axis=1)
    if np.any(np.isclose(nrm[1:], 0.0)):
        raise ValueError("Cannot normalize quaternion with zero norm")
    newQ[1:] = newQ[1:] / nrm[1:, None]
    newR = quaternion_to_rotation_matrix(newQ[1:])
    newC = C[1:] + 0.1 * da[1:, 4:7]
    R[1:] = newR
    T[1:] = -np.einsum("mij,mj->mi", newR, newC)
    X += 0.5 * db
    return cost_arrays(K, R, T, X, vi, pi, x_true), da, db


def cost_arrays(K, R, T, X, cam_idx, pt_idx, x_true):
    """src/bundle_adjustment.py:9-19 on arrays: mean over cameras of
    sqrt(sum over the camera's observations of |x - proj|^2)."""
    m = R.shape[0]
    P = K[None] @ np.concatenate([R, T[:, :, None]], 2)
    Xh = np.concatenate([X[pt_idx], np.ones((len(pt_idx), 1))], 1)
    proj = np.einsum("nij,nj->ni", P[cam_idx], Xh)
    r = x_true - proj[:, :2] / proj[:, 2:3]
    ssq = np.bincount(cam_idx, weights=(r * r).sum(1), minlength=m)
    return float(np.sqrt(ssq).sum() / m)
