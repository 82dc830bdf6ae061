"""A sequence that closes a loop: the cameras kept in the folded order on the device (reorder=True)
against the same problem in acquisition order.  In acquisition order S has blocks in its far corners
and goes through the general factorisation; folded it is a band (narrow: cyclic reduction on
64-blocks; up to 256: on superblocks).  Results must agree to the parity bar and come back in the
caller's camera order, with the caller's camera 0 held."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(p, reorder, iters=3):
    from sfm_amd.device import DeviceProblem

    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, reorder=reorder)
    c0 = prob.cost()
    costs = prob.run(max_iter=iters, threshold=0.0)
    R, T, C, P = prob.cameras()
    return prob, c0, np.asarray(costs), R, T, C, P, prob.points()


# (cameras, points, views): 7m = 280 (band chain / general), 2100 (superblocks of 128: 17 of them),
# 1400 with per-point view counts 3..5 (narrow after folding: 64-blocks)
@pytest.mark.parametrize("m,n,k", [(40, 3000, 5), (300, 30000, (6, 7)), (200, 20000, (3, 4))])
def test_loop_reordered_matches_acquisition_order(m, n, k):
    import sfm_amd  # noqa: F401
    from sfm_amd import ordering, synthetic

    p = synthetic.make_problem(m, n, k, visibility="loop", seed=8)
    a, c0a, ca, Ra, Ta, Ca, Pa, Xa = _run(p, False)
    b, c0b, cb, Rb, Tb, Cb, Pb, Xb = _run(p, True)
    assert a.cam_order is None and a.half_bandwidth == 7 * (m - 1) + 6
    assert b.cam_order is not None and b.natural_half_bandwidth == a.half_bandwidth
    # the plan's band is the one the host predicted for the new order
    assert b.half_bandwidth == ordering.half_bandwidth(b.cam_pos, p.cam_idx, p.pt_idx, p.n)
    assert 2 * b.half_bandwidth <= a.half_bandwidth
    assert abs(c0a - c0b) <= 1e-12 * c0a
    assert np.max(np.abs(ca - cb) / ca) < 1e-9
    assert np.max(np.abs(Ra - Rb)) < 1e-8 and np.max(np.abs(Ta - Tb)) < 1e-8
    assert np.max(np.abs(Pa - Pb)) < 1e-8 * np.max(np.abs(Pa))
    assert np.max(np.abs(Xa - Xb)) < 1e-8 * np.max(np.abs(Xa))
    # the caller's first camera is the one held, bit for bit
    assert np.array_equal(Rb[0], p.R[0]) and np.array_equal(Tb[0], p.T[0])
    assert not np.array_equal(Rb[1], p.R[1])


def test_loop_against_oracle():
    """Two iterations of the folded loop against the CPU oracle (which solves in acquisition order).
    Cost bar 5e-9: the oracle's LU solution of this system (condition number ~1e9) is itself ~1e-9
    relative from an extended-precision one (tests/diag_solve.py); measured 1.05e-9 / 3e-10."""
    import sfm_amd  # noqa: F401
    from oracle import ba_oracle as orc
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    p = synthetic.make_problem(60, 4000, 5, visibility="loop", seed=9)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, reorder=True)
    assert prob.cam_order is not None
    R, T, X = p.R.copy(), p.T.copy(), p.X.copy()
    for it in range(2):
        c_ref, da_ref, db_ref = orc.gn_iteration_arrays(p.K, R, T, X, p.cam_idx, p.pt_idx, p.x_true)
        prob.iterate_async()
        c = float(prob.cost_buf.item())
        prob.check_flags()
        print("iteration", it, "relative cost difference", abs(c - c_ref) / c_ref)
        assert abs(c - c_ref) <= 5e-9 * c_ref, (it, c, c_ref)
        # delta_a lives in the device's camera order
        da = prob.delta_a.cpu().numpy().reshape(-1, 7)[prob.cam_pos]
        assert np.max(np.abs(da - da_ref.reshape(-1, 7))) <= 1e-7 * np.max(np.abs(da_ref))
    Rg, Tg, _, _ = prob.cameras()
    assert np.max(np.abs(Rg - R)) < 1e-8 and np.max(np.abs(Tg - T)) < 1e-8
    assert np.max(np.abs(prob.points() - X)) < 1e-8 * np.max(np.abs(X))


def test_band_and_dense_problems_keep_their_order():
    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    for vis, m in (("banded", 40), ("uniform", 12)):
        p = synthetic.make_problem(m, 2000, 4, visibility=vis, seed=3)
        prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, reorder=True)
        assert prob.cam_order is None


def test_update_cameras_held_argument_checks():
    import torch

    import sfm_amd  # noqa: F401
    from sfm_amd import _lib

    lib = _lib.load()
    z = torch.zeros(64, dtype=torch.float64, device="cuda")
    f = torch.zeros(4, dtype=torch.int32, device="cuda")
    p = _lib.ptr
    for held in (-2, 4):
        rc = lib.sfm_ba_update_cameras_held(p(z), p(z), p(z), None, p(z), 0, p(z), 4, held, 1.0, 0.1, p(f), None, None,
                                            None)
        assert rc == _lib.EINVAL


def test_drop_in_entry_with_reorder():
    """optimize_using_BA(frames, map_, reorder=True) on a closed loop: same frames and map as without."""
    import copy

    import sfm_amd  # noqa: F401
    from sfm_amd import bundle_adjustment as ba
    from sfm_amd import synthetic
    from sfm_amd.frame import Frame, Map

    p = synthetic.make_problem(40, 2500, 5, visibility="loop", seed=12)
    frames_a, map_a = synthetic.to_frames_and_map(p, Frame, Map, extra_keypoints=3)
    frames_b, map_b = copy.deepcopy(frames_a), copy.deepcopy(map_a)
    ca = ba.optimize_using_BA(frames_a, map_a, verbose=False, max_iter=3, threshold=0.0)
    cb = ba.optimize_using_BA(frames_b, map_b, verbose=False, max_iter=3, threshold=0.0, reorder=True)
    assert np.max(np.abs(np.asarray(ca) - np.asarray(cb)) / np.asarray(ca)) < 1e-9
    for fa, fb in zip(frames_a, frames_b):
        assert np.max(np.abs(fa.R - fb.R)) < 1e-8 and np.max(np.abs(fa.T - fb.T)) < 1e-8
        assert np.max(np.abs(fa.P - fb.P)) < 1e-8 * np.max(np.abs(fa.P))
    assert np.array_equal(frames_b[0].R, p.R[0])
    assert np.max(np.abs(map_a.X - map_b.X)) < 1e-8 * np.max(np.abs(map_a.X))


@pytest.mark.parametrize("m,n,k,block", [(300, 30000, (6, 7), 128), (500, 40000, (12, 14), 256)])
def test_superblock_solve_accuracy_on_a_real_system(m, n, k, block):
    """The reduction on superblocks on an assembled, ill-conditioned S (not a diagonally dominant
    random matrix): at least as close to an extended-precision solution as numpy's LU, within 5x."""
    import sfm_amd  # noqa: F401
    from sfm_amd import _lib, synthetic
    from sfm_amd.device import DeviceProblem

    p = synthetic.make_problem(m, n, k, visibility="loop", seed=31)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, reorder=True)
    n7 = prob.n7
    assert int(_lib.load().sfm_ba_reduced_solve_block(n7, prob.half_bandwidth)) == block
    prob.camera_prepare()
    prob.jacobian()
    prob.assemble(0.5)
    S = prob.S[:n7].cpu().numpy().copy()
    e = prob.S[n7].cpu().numpy().copy()
    S = np.tril(S) + np.tril(S, -1).T
    prob.solve()
    prob.check_flags()
    x_dev = prob.delta_a.cpu().numpy().reshape(-1)
    x_lu = np.linalg.solve(S, e)
    Sl, el, x = S.astype(np.longdouble), e.astype(np.longdouble), x_lu.astype(np.longdouble)
    for _ in range(6):
        x = x + np.linalg.solve(S, (el - Sl @ x).astype(np.float64)).astype(np.longdouble)
    x = x.astype(np.float64)
    err_dev = np.linalg.norm(x_dev - x) / np.linalg.norm(x)
    err_lu = np.linalg.norm(x_lu - x) / np.linalg.norm(x)
    print(f"7m = {n7}, half bandwidth {prob.half_bandwidth}: device {err_dev:.2e}, numpy LU {err_lu:.2e}")
    assert err_dev <= max(5.0 * err_lu, 1e-10)
