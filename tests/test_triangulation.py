"""Two-view triangulation: the CPU oracle against the reference's outputs (tests/golden/
triangulation.npz, made by tests/golden/make_triangulation.py) and the reference's own known-answer
cases (test/test_triangulation.py:6-58), then the GPU kernel against both.

Outputs are float32 (the reference stacks them into a float32 array), so the bar is float32
rounding: 2e-6 of the largest coordinate on the noisy cases, where the point is the minimiser of an
over-determined system and the two SVD algorithms agree to ~1e-12 before rounding.
"""
import numpy as np
import pytest

from oracle import triangulation_oracle as orc
from tests.helpers import load


def _check_against_golden(fn, tol):
    d = load("triangulation")
    for c in range(int(d["n_cases"])):
        X = fn(d[f"c{c}_x1"], d[f"c{c}_x2"], d[f"c{c}_P1"], d[f"c{c}_P2"])
        ref = d[f"c{c}_Xh"]
        assert X.dtype == np.float32 and X.shape == ref.shape
        assert np.max(np.abs(X.astype(np.float64) - ref)) <= tol * np.max(np.abs(ref))
    X = fn(d["crafted_x1"], d["crafted_x2"], d["crafted_P1"], d["crafted_P2"])
    assert np.max(np.abs(X.astype(np.float64) - d["crafted_Xh"])) <= tol * np.max(np.abs(d["crafted_Xh"]))
    # the reference's assertions on its crafted case
    assert np.array_equal(np.round(d["crafted_X"], 2), np.round(X, 2))
    P1, P2 = d["crafted_P1"], d["crafted_P2"]
    for P, x in ((P1, d["crafted_x1"]), (P2, d["crafted_x2"])):
        xr = (P @ X.astype(np.float64).T)
        xr = (xr / xr[-1]).T
        assert np.linalg.norm(xr - x) < 0.05
    assert fn(np.zeros((0, 3)), np.zeros((0, 3)), P1, P2).shape == (0, 4)


def test_oracle_matches_reference():
    _check_against_golden(orc.triangulate_pts, 2e-6)


@pytest.mark.gpu
def test_gpu_matches_reference():
    import sfm_amd  # noqa: F401
    from sfm_amd.triangulation import triangulate_pts

    _check_against_golden(triangulate_pts, 2e-6)


@pytest.mark.gpu
def test_gpu_large_batch_against_oracle_and_reprojection():
    """2^20 correspondences (ragged against the 256-lane workgroups): exact projections must
    triangulate back onto the points, noisy ones must match the oracle's SVD."""
    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic
    from sfm_amd.triangulation import triangulate_pts

    rng = np.random.default_rng(3)
    n = (1 << 20) + 77
    K = synthetic.K_DEFAULT
    X = np.concatenate([rng.uniform(-2, 2, size=(n, 3)) + np.array([0, 0, 8.0]), np.ones((n, 1))], axis=1)
    R2 = synthetic.rodrigues(np.array([0.02, -0.05, 0.01]))
    P1 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = K @ np.hstack([R2, np.array([[0.7], [0.05], [-0.1]])])
    x1 = (P1 @ X.T); x1 = (x1 / x1[-1]).T
    x2 = (P2 @ X.T); x2 = (x2 / x2[-1]).T
    Xh = triangulate_pts(x1, x2, P1, P2)
    assert Xh.shape == (n, 4) and np.all(Xh[:, 3] == 1.0)
    assert np.max(np.abs(Xh[:, :3] - X[:, :3])) < 2e-5  # float32 rounding of coordinates up to 10
    sel = rng.choice(n, size=20000, replace=False)
    x1n = x1[sel] + np.concatenate([rng.normal(0, 0.7, size=(sel.size, 2)), np.zeros((sel.size, 1))], axis=1)
    x2n = x2[sel] + np.concatenate([rng.normal(0, 0.7, size=(sel.size, 2)), np.zeros((sel.size, 1))], axis=1)
    a = triangulate_pts(x1n, x2n, P1, P2)
    b = orc.triangulate_pts(x1n, x2n, P1, P2)
    assert np.max(np.abs(a.astype(np.float64) - b)) <= 2e-6 * np.max(np.abs(b))
