"""The fused point-major assembly (runs of points inside narrow camera windows, one pass over A,
fp64 MFMA accumulation) against the camera-pair path of the same library, against the CPU oracle,
and its index structures against the host builder.

Bars: S, e, V_inv, eps_b of the two device paths agree to 1e-12 of the largest entry (the sums are
associated differently); one Gauss-Newton iteration against the numpy oracle as in
tests/test_gpu_parity.py (cost 1e-9 relative on these small problems, updates 1e-7)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MU = 0.5


def _problem(m, n, k, seed=0):
    from sfm_amd import synthetic

    return synthetic.make_problem(m, n, k, visibility="banded", seed=seed)


# m, n, views per point, cameras per window
CASES = [
    (30, 6000, (5, 7), 7),    # 3 row tiles + 1 extra row
    (16, 2000, 8, 8),         # 4 row tiles
    (40, 9000, (3, 5), 5),    # 3 row tiles (35 of their 48 rows used)
    (25, 4000, 6, 6),         # 3 row tiles
    (64, 20001, (2, 4), 4),   # 2 row tiles
    (50, 8000, (2, 3), 4),    # several first-cameras per window
]


@pytest.mark.parametrize("m,n,k,slots", CASES)
def test_fused_matches_pair_path(m, n, k, slots):
    import torch

    from sfm_amd.device import DeviceProblem, Plan

    p = _problem(m, n, k, seed=m)
    host = Plan(p.cam_idx, p.pt_idx, where="host")
    dev = Plan(p.cam_idx, p.pt_idx, where="device")
    assert host.fused == dev.fused == 1
    assert dev.fused_slots == slots and dev.classes[0]["n_near"] == n and dev.far is None
    for f in Plan.FUSED_FIELDS:
        a, b = host.read(f), dev.read(f)
        assert a.shape == b.shape and np.array_equal(a, b), f
    assert host.workspace_bytes == dev.workspace_bytes
    # the table holds every observation exactly once, at the slot of its camera in the window of its run
    so = dev.read("slot_obs").reshape(n, 8)
    key_of_pos = np.repeat(dev.read("run_key"), np.diff(dev.read("run_pt0")))
    kc = dev.read("key_cam").reshape(-1, 8)
    pos, sl = np.nonzero(so >= 0)
    obs = so[pos, sl]
    assert np.array_equal(np.sort(obs), np.arange(p.n_obs))
    assert np.array_equal(p.pt_idx[obs], dev.read("fz_pt")[pos]) and np.array_equal(p.cam_idx[obs], kc[key_of_pos[pos], sl])

    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    assert prob.plan.fused == 1
    prob.camera_prepare()
    prob.jacobian(want_B=False)
    n7 = 7 * m
    res = {}
    for fused in (0, 1):
        prob.plan.set_fused(fused)
        prob.sys.zero_()
        prob.V_inv.fill_(float("nan"))
        prob.eps_b.fill_(float("nan"))
        prob.assemble(MU, by_products=False, B=False)
        res[fused] = [torch.tril(prob.S[:n7]).clone(), prob.S[n7].clone(), prob.e.clone(), prob.V_inv.clone(),
                      prob.eps_b.clone()]
    for name, a, b in zip(("S", "e (row 7m)", "e", "V_inv", "eps_b"), res[0], res[1]):
        scale = float(a.abs().max().item())
        assert bool(torch.isfinite(b).all().item()), name
        assert float((a - b).abs().max().item()) <= 1e-12 * scale, name
    # same zero pattern below the diagonal: blocks of camera pairs without a common point stay untouched
    assert bool(((res[0][0] == 0) == (res[1][0] == 0)).all().item())
    # bit-reproducible
    prob.sys.zero_()
    prob.assemble(MU, by_products=False, B=False)
    assert bool((torch.tril(prob.S[:n7]) == res[1][0]).all().item()) and bool((prob.S[n7] == res[1][1]).all().item())


@pytest.mark.parametrize("m,n,k", [(30, 6000, (5, 7)), (40, 9000, (3, 5))])
def test_fused_iteration_against_oracle(m, n, k):
    """Two iterations of the loop against the numpy oracle, with the fused assembly and with the
    camera-pair path.  cond(S) is ~1e9 on these problems and numpy's LU solution (the oracle's) is
    itself ~1e-9 relative from the exact one (tests/test_gpu_parity.py::test_ladybug_scale_against_oracle),
    so the cost bar is 2e-8 as there; the two device paths must agree to 1e-9 (rounding differences of S, amplified by the solve)."""
    from oracle import ba_oracle as orc
    from sfm_amd.device import DeviceProblem
    from tests.helpers import relerr

    p = _problem(m, n, k, seed=7)
    R, T, X = p.R.copy(), p.T.copy(), p.X.copy()
    ref = [orc.gn_iteration_arrays(p.K, R, T, X, p.cam_idx, p.pt_idx, p.x_true) for _ in range(2)]
    costs = {}
    # 1: rows of A formed inside the fused kernels; 0: camera-pair path; 3: fused assembly on the rows
    # of A written by the Jacobian kernel
    for fused in (1, 0, 3):
        prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, state=(fused == 1))
        assert prob.plan.fused == 1 and prob.from_state == (fused == 1)
        prob.plan.set_fused(0 if fused == 0 else 1)
        costs[fused] = []
        for it in range(2):
            c_ref, da_ref, db_ref = ref[it]
            prob.iterate_async()
            c = float(prob.cost_buf.item())
            prob.check_flags()
            costs[fused].append(c)
            print(f"fused={fused} iteration {it}: cost {c!r} oracle {c_ref!r} rel {abs(c - c_ref) / c_ref:.2e}")
            assert abs(c - c_ref) <= 2e-8 * c_ref, (fused, it, c, c_ref)
            assert relerr(prob.delta_a.cpu().numpy(), da_ref) < 1e-7
            assert relerr(prob.delta_b.cpu().numpy(), db_ref) < 1e-7
        assert relerr(prob.points(), X) < 1e-8
    for a, b, d in zip(costs[0], costs[1], costs[3]):
        assert abs(a - b) <= 1e-9 * a and abs(a - d) <= 1e-9 * a, costs
