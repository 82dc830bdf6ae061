"""SplitProblem (near points through the fused assembly in acquisition order, far points through the
camera-pair kernels, the sum solved in the folded camera order) against DeviceProblem on the same
closed-loop problems: same costs, poses and points to the parity bar."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# (cameras, points, views, visibility) -> path of the solve on the folded system:
# 7m = 2100, band ~90: superblocks of 128; 7m = 1400, band <= 64: 64-blocks; 7m = 280: general path
# (whole lower triangle renumbered); two laps, 7m = 3500, band ~240: superblocks of 256
# and the benchmark's size (500 cameras x 300 000 points x 1.95 M observations, one lap)
CASES = [(300, 30000, (6, 7), "loop"), (200, 20000, (3, 4), "loop"), (40, 3000, 5, "loop"),
         (500, 40000, (6, 7), "closures"), (500, 300000, (6, 7), "loop")]


@pytest.mark.parametrize("m,n,k,vis", CASES)
def test_split_matches_unsplit(m, n, k, vis):
    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem
    from sfm_amd.split import SplitProblem

    p = synthetic.make_problem(m, n, k, visibility=vis, seed=21)
    a = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    b = SplitProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    assert b.near.plan.fused and b.far.n > 0 and b.far.n * 10 < p.n
    assert b.half_bandwidth * 2 <= b.natural_half_bandwidth == a.half_bandwidth
    c0a, c0b = a.cost(), b.cost()
    assert abs(c0a - c0b) <= 1e-12 * c0a
    ca = np.asarray(a.run(max_iter=3, threshold=0.0))
    cb = np.asarray(b.run(max_iter=3, threshold=0.0))
    assert np.max(np.abs(ca - cb) / ca) < 1e-9
    Ra, Ta, Ca, Pa = a.cameras()
    Rb, Tb, Cb, Pb = b.cameras()
    assert np.max(np.abs(Ra - Rb)) < 1e-8 and np.max(np.abs(Ta - Tb)) < 1e-8
    assert np.max(np.abs(Pa - Pb)) < 1e-8 * np.max(np.abs(Pa))
    assert np.max(np.abs(a.points() - b.points())) < 1e-8 * np.max(np.abs(a.points()))
    assert np.array_equal(Rb[0], p.R[0]) and np.array_equal(Tb[0], p.T[0])
    da_a = a.delta_a.cpu().numpy()
    da_b = b.delta_a.cpu().numpy()
    assert np.max(np.abs(da_a - da_b)) <= 1e-6 * np.max(np.abs(da_a))


def test_split_against_oracle():
    import sfm_amd  # noqa: F401
    from oracle import ba_oracle as orc
    from sfm_amd import synthetic
    from sfm_amd.split import SplitProblem

    p = synthetic.make_problem(60, 4000, 5, visibility="loop", seed=22)
    prob = SplitProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    R, T, X = p.R.copy(), p.T.copy(), p.X.copy()
    for it in range(2):
        c_ref, da_ref, db_ref = orc.gn_iteration_arrays(p.K, R, T, X, p.cam_idx, p.pt_idx, p.x_true)
        prob.iterate_async()
        c = float(prob.cost_buf.item())
        prob.check_flags()
        assert abs(c - c_ref) <= 1e-9 * c_ref, (it, c, c_ref)
    Rg, Tg, _, _ = prob.cameras()
    assert np.max(np.abs(Rg - R)) < 1e-8 and np.max(np.abs(Tg - T)) < 1e-8
    assert np.max(np.abs(prob.points() - X)) < 1e-8 * np.max(np.abs(X))


def test_split_refuses_what_it_cannot_split():
    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic
    from sfm_amd.split import SplitProblem

    for vis, m, msg in (("banded", 40, "nothing to split"), ("uniform", 40, "tenth")):
        p = synthetic.make_problem(m, 2000, 4, visibility=vis, seed=3)
        with pytest.raises(ValueError, match=msg):
            SplitProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)


def test_permute_system_by_hand():
    import torch

    import sfm_amd  # noqa: F401
    from sfm_amd import _lib

    lib = _lib.load()
    m = 5
    n7 = 7 * m
    rng = np.random.default_rng(0)
    G = rng.standard_normal((n7, n7))
    S = G + G.T
    e = rng.standard_normal(n7)
    order = np.array([3, 0, 4, 1, 2], dtype=np.int32)
    idx = (7 * order[:, None] + np.arange(7)[None, :]).reshape(-1)
    want = S[np.ix_(idx, idx)]
    src = np.vstack([np.tril(S), e[None, :]])  # lower triangle only
    d_src = torch.from_numpy(src).cuda()
    d_ord = torch.from_numpy(order).cuda()
    p = _lib.ptr
    for hb in (n7, 9):
        d_dst = torch.full((n7 + 1, n7), np.nan, dtype=torch.float64, device="cuda")
        _lib.check(lib.sfm_ba_permute_system(p(d_src), n7, p(d_ord), hb, p(d_dst), None))
        got = d_dst.cpu().numpy()
        r, c = np.indices((n7, n7))
        band = (c <= r) & (r - c <= hb)
        assert np.array_equal(got[:n7][band], want[band])
        assert np.all(np.isnan(got[:n7][~band]))
        assert np.array_equal(got[n7], e[idx])
    assert lib.sfm_ba_permute_system(p(d_src), n7, p(d_ord), 5, p(d_src), None) == _lib.EINVAL
    assert lib.sfm_ba_reduced_solve_block(3500, 60) == 64
    assert lib.sfm_ba_reduced_solve_block(3500, 90) == 128
    assert lib.sfm_ba_reduced_solve_block(3500, 200) == 256
    assert lib.sfm_ba_reduced_solve_block(3500, 300) == 0
    assert lib.sfm_ba_reduced_solve_block(700, 90) == 128
    assert lib.sfm_ba_reduced_solve_block(350, 90) == 0
    assert lib.sfm_ba_reduced_solve_block(1200, 200) == 0
