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


# m, n, views per point, cameras per window, first-cameras per window
CASES = [
    (30, 6000, (5, 7), 7, 1),    # 3 row tiles + 1 extra row
    (16, 2000, 8, 8, 1),         # 4 row tiles
    (40, 9000, (3, 5), 5, 1),    # 3 row tiles (35 of their 48 rows used)
    (25, 4000, 6, 6, 1),         # 3 row tiles
    (64, 20001, (2, 4), 4, 1),   # 2 row tiles
    (50, 8000, (2, 3), 4, 2),    # two first-cameras per window
]


@pytest.mark.parametrize("m,n,k,slots,group", CASES)
def test_fused_matches_pair_path(m, n, k, slots, group):
    import torch

    from sfm_amd.device import DeviceProblem, Plan

    p = _problem(m, n, k, seed=m)
    host = Plan(p.cam_idx, p.pt_idx, where="host")
    dev = Plan(p.cam_idx, p.pt_idx, where="device")
    assert host.fused == dev.fused == 1
    assert (dev.fused_slots, dev.fused_group) == (slots, group)
    for f in ("run_pt0", "run_base", "slot_obs", "key_ptr", "key_run", "rng_run", "step_pt0", "step_run", "rng_step"):
        a, b = host.read(f), dev.read(f)
        assert a.shape == b.shape and np.array_equal(a, b), f
    assert host.workspace_bytes == dev.workspace_bytes
    # the table holds every observation exactly once, at the slot of its camera
    so = dev.read("slot_obs").reshape(n, 8)
    base = np.repeat(dev.read("run_base"), np.diff(dev.read("run_pt0")))
    pts, sl = np.nonzero(so >= 0)
    obs = so[pts, sl]
    assert np.array_equal(np.sort(obs), np.arange(p.n_obs))
    assert np.array_equal(p.pt_idx[obs], pts) and np.array_equal(p.cam_idx[obs], base[pts] + sl)

    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, steps=True)
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

    # the same stages on step blocks (rows of A and residuals in the order the kernel walks them):
    # the blocks hold exactly the rows of A and x_true - x_pred, the assembly gives the same sums
    # bit for bit (same arithmetic, other addresses), the back-substitution the same update to rounding
    assert dev.n_steps > 0 and prob.steps is not None and prob.steps.numel() == 1024 * dev.n_steps
    prob.jacobian(want_B=False)
    prob.steps.fill_(float("nan"))
    prob.jacobian_steps()
    blocks = prob.steps.view(dev.n_steps, 8, 64, 2).cpu().numpy()
    step_pt0, step_run, run_pt0 = dev.read("step_pt0"), dev.read("step_run"), dev.read("run_pt0")
    A_h = prob.A.cpu().numpy()
    eps_h = p.x_true - prob.x_pred.cpu().numpy()
    rows = blocks[:, :7].transpose(0, 2, 1, 3).reshape(dev.n_steps, 64, 14)   # [step][lane][14]
    eps_b = blocks[:, 7]                                                       # [step][lane][2]
    pt_of = step_pt0[:, None] + (np.arange(64) >> 3)[None, :]
    valid = pt_of < run_pt0[step_run[:-1] + 1][:, None]
    obs_of = np.where(valid, so[np.minimum(pt_of, n - 1), (np.arange(64) & 7)[None, :]], -1)
    has = obs_of >= 0
    assert np.array_equal(rows[has], A_h[obs_of[has]])
    assert np.max(np.abs(eps_b[has] - eps_h[obs_of[has]])) <= 1e-12 * np.max(np.abs(eps_h))
    assert not rows[~has].any() and not eps_b[~has].any()
    assert int(has.sum()) == p.n_obs
    prob.sys.zero_()
    prob.V_inv.fill_(float("nan"))
    prob.eps_b.fill_(float("nan"))
    prob.assemble(MU, steps=True)
    got = [torch.tril(prob.S[:n7]).clone(), prob.S[n7].clone(), prob.e.clone(), prob.V_inv.clone(), prob.eps_b.clone()]
    for name, a, b in zip(("S", "e (row 7m)", "e", "V_inv", "eps_b"), res[1], got):
        scale = float(a.abs().max().item())
        assert float((a - b).abs().max().item()) <= 1e-13 * scale, name
    prob.solve()
    prob.back_substitute(B=False)
    db_rows = prob.delta_b.clone()
    prob.back_substitute(steps=True)
    assert float((db_rows - prob.delta_b).abs().max().item()) <= 1e-12 * float(db_rows.abs().max().item())


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
    # 1: rows of A formed inside the fused kernels; 0: camera-pair path; 2: fused assembly on step
    # blocks; 3: fused assembly on the rows of A written by the Jacobian kernel
    for fused in (1, 0, 2, 3):
        prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, steps=(fused == 2), state=(fused == 1))
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
    for a, b, c, d in zip(costs[0], costs[1], costs[2], costs[3]):
        assert abs(a - b) <= 1e-9 * a and abs(a - c) <= 1e-9 * a and abs(a - d) <= 1e-9 * a, costs


def test_ineligible_patterns_keep_the_pair_path():
    from sfm_amd import synthetic
    from sfm_amd.device import Plan

    p = synthetic.make_problem(20, 2000, 6, visibility="uniform", seed=1)
    assert Plan(p.cam_idx, p.pt_idx).fused == 0
    # banded, but point ids in random order: windows change from point to point
    q = _problem(30, 3000, 6, seed=2)
    rng = np.random.default_rng(0)
    relabel = rng.permutation(q.n).astype(np.int32)
    for where in ("host", "device"):
        assert Plan(q.cam_idx, relabel[q.pt_idx], where=where).fused == 0
    # nine consecutive cameras do not fit a window
    r = _problem(30, 3000, 9, seed=3)
    assert Plan(r.cam_idx, r.pt_idx).fused == 0
