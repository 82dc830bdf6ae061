"""Linear PnP (/root/reference/src/pose_estimation.py:21-144).

CPU tests pin the oracle with the reference's own known-answer cases, restated here as data from
/root/reference/test/test_linear_pnp.py (the `cleanup_RT_mat` matrix, the crafted six-point pose,
its RANSAC variant) with the reference's own assertions.  GPU tests compare the library with the
oracle: the same cases, then seeded noisy problems of 6 to 5000 correspondences.

Tolerances: a pose is compared at 1e-9 of its largest entry on well-conditioned data (the GPU uses
one-sided Jacobi rotations, LAPACK a bidiagonalisation; both are backward stable, and the smallest
singular vector moves by eps * sigma_max / gap); `T / d[0]` is compared at 1e-13 relative where
the reference's test asserts equality with numpy's own `d[0]`.
"""
import numpy as np
import pytest

from oracle import pnp_oracle as orc

R_CLEANUP = np.array([
    [8.00594113e-04, 2.07275507e-15, 2.21756931e-13],
    [9.63817239e-15, 1.71564047e-10, -8.00594199e-04],
    [-2.49159865e-14, 8.00594113e-04, 3.76291759e-13],
])
T_CLEANUP = np.array([0.00020015, 0.00020015, -0.00020015])
X6 = np.array([
    [0.0, 5.0, 2.0, 1.0], [2.0, 5.0, 3.0, 1.0], [2.0, 4.0, 2.0, 1.0],
    [3.0, 5.0, 2.0, 1.0], [2.0, 4.0, 3.0, 1.0], [3.0, 4.0, 2.0, 1.0],
])
K6 = np.array([[700.0, 0.0, 320.0], [0.0, 933.3333, 240.0], [0.0, 0.0, 1.0]])
RT6 = np.array([[1.0, 0.0, 0.0, 0.25], [0.0, 0.0, -1.0, 0.25], [0.0, 1.0, 0.0, -0.25]])


def _check_crafted(R, T):
    x = orc.project_3D_to_2D(X6, K6 @ RT6)
    RT = np.hstack((R, T))
    assert np.linalg.norm(orc.project_3D_to_2D(X6, K6 @ RT) - x) < 0.05
    assert np.array_equal(np.round(RT6, 2), np.round(RT, 2) + 0.0)


def _problem(n, seed, noise=0.3, outliers=0.0):
    rng = np.random.default_rng(seed)
    K = np.array([[1200.0, 0.0, 960.0], [0.0, 1150.0, 540.0], [0.0, 0.0, 1.0]])
    w = rng.normal(size=3)
    w *= rng.uniform(0.2, 1.0) / np.linalg.norm(w)
    th = np.linalg.norm(w)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    T = rng.normal(size=(3, 1)) * 0.5 + np.array([[0.0], [0.0], [6.0]])
    Xw = np.hstack([rng.uniform(-2, 2, size=(n, 3)), np.ones((n, 1))])
    x = orc.project_3D_to_2D(Xw, K @ np.hstack((R, T)))
    x[:, :2] += rng.normal(size=(n, 2)) * noise
    bad = rng.random(n) < outliers
    x[bad, :2] += rng.uniform(-400, 400, size=(int(bad.sum()), 2))
    return x, Xw, K, R, T


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))


# ---- oracle against the reference's own cases (CPU) ---------------------------------------------
def test_oracle_cleanup_RT():
    R_c, T_c = orc.cleanup_RT_mat(R_CLEANUP, T_CLEANUP)
    _, d, _ = np.linalg.svd(R_CLEANUP)
    assert np.array_equal(T_c, T_CLEANUP / d[0])
    assert np.round(np.linalg.det(R_c)) == 1


def test_oracle_crafted_pose():
    x = orc.project_3D_to_2D(X6, K6 @ RT6)
    _check_crafted(*orc.estimate_pose_Linear_PnP(x, X6, K6))
    _check_crafted(*orc.estimate_pose_Linear_PnP_n(x, X6, K6))


def test_oracle_crafted_ransac():
    x = orc.project_3D_to_2D(X6, K6 @ RT6)
    np.random.seed(0)
    R, T, n_in = orc.estimate_pose_Linear_PnP_RANSAC(x, X6, K6)
    assert n_in == 6
    _check_crafted(R, T)


def test_oracle_rows_match_the_per_point_construction():
    """the vectorised row builder against the literal vstack / hstack of the source (:53-65)"""
    x, Xw, *_ = _problem(7, 1)
    A = np.zeros((0, 12))
    for i in range(len(x)):
        v = x[i]
        skew = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        z = np.zeros((1, 4))
        Xt = Xw[i].reshape(1, -1)
        Xm = np.vstack((np.hstack((Xt, z, z)), np.hstack((z, Xt, z)), np.hstack((z, z, Xt))))
        A = np.vstack((A, skew @ Xm))
    assert np.array_equal(A, orc._rows(x, Xw))


# ---- library against the oracle (GPU) ---------------------------------------------------------
@pytest.mark.gpu
def test_gpu_cleanup_RT():
    from sfm_amd import pose_estimation as pe

    R_c, T_c = pe.cleanup_RT_mat(R_CLEANUP, T_CLEANUP)
    R_o, T_o = orc.cleanup_RT_mat(R_CLEANUP, T_CLEANUP)
    assert T_c.shape == T_CLEANUP.shape
    assert rel(T_c, T_o) < 1e-13 and rel(R_c, R_o) < 1e-12
    assert np.round(np.linalg.det(R_c)) == 1
    # reflection: both are negated
    R_c, T_c = pe.cleanup_RT_mat(-R_CLEANUP, T_CLEANUP.reshape(3, 1))
    R_o, T_o = orc.cleanup_RT_mat(-R_CLEANUP, T_CLEANUP.reshape(3, 1))
    assert T_c.shape == (3, 1) and rel(T_c, T_o) < 1e-13 and rel(R_c, R_o) < 1e-12
    with pytest.raises(UnboundLocalError):
        pe.cleanup_RT_mat(np.zeros((3, 3)), T_CLEANUP)


@pytest.mark.gpu
def test_gpu_crafted_pose():
    from sfm_amd import pose_estimation as pe

    x = orc.project_3D_to_2D(X6, K6 @ RT6)
    for fn, ofn in ((pe.estimate_pose_Linear_PnP, orc.estimate_pose_Linear_PnP),
                    (pe.estimate_pose_Linear_PnP_n, orc.estimate_pose_Linear_PnP_n)):
        R, T = fn(x, X6, K6)
        assert R.shape == (3, 3) and T.shape == (3, 1)
        _check_crafted(R, T)
        R_o, T_o = ofn(x, X6, K6)
        assert rel(R, R_o) < 1e-9 and rel(T, T_o) < 1e-9
    R, T = pe.decompose_projection_mat(K6 @ RT6 * 3.0, K6)
    R_o, T_o = orc.decompose_projection_mat(K6 @ RT6 * 3.0, K6)
    assert rel(R, R_o) < 1e-12 and rel(T, T_o) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n", [6, 7, 50, 64, 65, 500, 1025, 5000])
def test_gpu_pose_matches_oracle(n):
    from sfm_amd import pose_estimation as pe

    x, Xw, K, R_gt, T_gt = _problem(n, 10 + n)
    R, T = pe.estimate_pose_Linear_PnP_n(x, Xw, K)
    R_o, T_o = orc.estimate_pose_Linear_PnP_n(x, Xw, K)
    assert rel(R, R_o) < 1e-9 and rel(T, T_o) < 1e-9
    if n >= 50:
        assert rel(R, R_gt) < 2e-2
    R, T = pe.estimate_pose_Linear_PnP(x, Xw, K)
    R_o, T_o = orc.estimate_pose_Linear_PnP(x, Xw, K)
    assert rel(R, R_o) < 1e-7 and rel(T, T_o) < 1e-7  # pixel-scaled rows: sigma_max / gap is ~1e3 larger


@pytest.mark.gpu
def test_gpu_ransac_matches_oracle():
    from sfm_amd import pose_estimation as pe

    x6 = orc.project_3D_to_2D(X6, K6 @ RT6)
    np.random.seed(0)
    R, T, n_in = pe.estimate_pose_Linear_PnP_RANSAC(x6, X6, K6)
    assert n_in == 6
    _check_crafted(R, T)

    x, Xw, K, R_gt, _ = _problem(800, 3, noise=0.5, outliers=0.3)
    for thr in (2000, 8.0):
        np.random.seed(7)
        R, T, n_in = pe.estimate_pose_Linear_PnP_RANSAC(x, Xw, K, threshold=thr)
        np.random.seed(7)
        R_o, T_o, n_o = orc.estimate_pose_Linear_PnP_RANSAC(x, Xw, K, threshold=thr)
        assert n_in == n_o
        assert rel(R, R_o) < 1e-9 and rel(T, T_o) < 1e-9
    assert n_in < 800 and rel(R, R_gt) < 2e-2  # the tight threshold rejected the outliers


@pytest.mark.gpu
def test_gpu_many_hypotheses_at_once():
    """every hypothesis of a batch equals the single-subset call"""
    from sfm_amd import pose_estimation as pe

    x, Xw, K, *_ = _problem(300, 21)
    rng = np.random.default_rng(0)
    idx = np.stack([rng.choice(300, 70, replace=False) for _ in range(9)])
    corr = pe.Correspondences(x, Xw)
    Kinv = np.linalg.inv(K)
    R, T, flags = corr.solve(idx=idx, Kinv_pre=Kinv)
    assert not flags.any().item()
    counts, mask = corr.inliers(K, R, T, 3.0, want_mask=True)
    R, T, counts, mask = R.cpu().numpy(), T.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy()
    for h in range(9):
        R_o, T_o = orc.estimate_pose_Linear_PnP_n(x[idx[h]], Xw[idx[h]], K)
        assert rel(R[h].reshape(3, 3), R_o) < 1e-9 and rel(T[h].reshape(3, 1), T_o) < 1e-9
        err = np.linalg.norm(orc.project_3D_to_2D(Xw, K @ np.hstack((R_o, T_o))) - x, axis=1)
        sure = np.abs(err - 3.0) > 1e-6
        assert np.array_equal(mask[h][sure] != 0, (err < 3.0)[sure])
        assert counts[h] == int(mask[h].sum())
