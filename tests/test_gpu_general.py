"""The fused point-major assembly as the general path: point ids in any order, near and far points in
one plan (far points through the camera-pair kernels of the plan's far plan, added by the finishing
kernel), windows made of a point's own camera set where its cameras lie far apart.

What is compared: the plan arrays of the host and the device builder (equal array for array); the
state path (rows of A formed in the kernels) and the fused path on materialised rows against the
camera-pair path of the same library on the same problem - S, e, eps_b within 1e-12 of the largest
entry, V_inv 1e-11, delta_b 1e-11 (bars of tests/test_gpu_state.py) -; the loop against the numpy
oracle (bars of tests/test_gpu_fused.py); and the reference-generated golden fixtures through the state
path: S, e, V_inv, eps_b of the first Gauss-Newton step against the reference's own at 1e-12 of the
largest entry (V_inv: 1e-11, see test_golden_step_through_the_state_path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MU = 0.5


def mixed_problem(m, n, seed, long_every=20, closure_every=37, unique_every=211):
    """Banded visibility (4..7 consecutive cameras, ids shuffled) in which every `long_every`-th point is a
    long track (9..14 cameras), every `closure_every`-th point is seen from one of six fixed camera sets
    made of two distant groups of three (windows of the points' own camera sets), and every
    `unique_every`-th point from a random set of 5 cameras that no other point shares (far: its window
    would hold one point)."""
    from sfm_amd import synthetic

    rng = np.random.default_rng(seed)
    pt, cam = [], []
    fixed = [np.r_[np.arange(a, a + 3), np.arange(b, b + 3)] for a, b in
             [(0, m // 2), (2, m // 2 + 7), (5, m - 4), (9, m // 2 + 1), (m // 4, m - 9), (1, m // 3 + 11)]]
    kinds = np.zeros(n, dtype=np.int64)
    for i in range(n):
        if i % unique_every == 5:
            cs = np.sort(rng.choice(m, size=5, replace=False))
            kinds[i] = 3
        elif i % closure_every == 3:
            cs = fixed[int(rng.integers(len(fixed)))]
            kinds[i] = 2
        elif i % long_every == 1:
            k = int(rng.integers(9, 15))
            c0 = int(rng.integers(0, m - k + 1))
            cs = np.arange(c0, c0 + k)
            kinds[i] = 1
        else:
            k = int(rng.integers(4, 8))
            c0 = int(rng.integers(0, m - k + 1))
            cs = np.arange(c0, c0 + k)
        pt += [i] * len(cs)
        cam += list(cs)
    pt, cam = np.asarray(pt), np.asarray(cam)
    X = rng.uniform(-2.0, 2.0, size=(n, 3)) + np.array([0.0, 0.0, 8.0])
    R = synthetic.rodrigues(rng.normal(0.0, 0.08, size=(m, 3)))
    Cc = rng.normal(0.0, 0.5, size=(m, 3))
    T = -np.einsum("mij,mj->mi", R, Cc)
    p = synthetic.observe_and_perturb(rng, synthetic.K_DEFAULT.copy(), R, T, X, pt, cam)
    p.visibility = "mixed"
    return p, kinds


def _plans_equal(a, b):
    from sfm_amd.device import Plan

    for f in Plan.FIELDS:
        x, y = a.read(f), b.read(f)
        assert x.shape == y.shape and np.array_equal(x, y), f
    for f in Plan.CLASS_FIELDS:
        x, y = a.read(f, cls=1), b.read(f, cls=1)
        assert x.shape == y.shape and np.array_equal(x, y), (f, "class 1")
    assert a.workspace_bytes == b.workspace_bytes
    assert a.classes == b.classes
    assert (a.far is None) == (b.far is None)
    if a.far is not None:
        _plans_equal(a.far, b.far)


def _compare_paths(p, expect_far=None, where="device"):
    """State path and fused path on rows against the camera-pair path, then both back-substitutions."""
    import torch

    from sfm_amd.device import DeviceProblem

    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, plan_where=where)
    assert prob.plan.fused == 1 and prob.from_state
    if expect_far is not None:
        assert (prob.plan.far is not None) == expect_far
    n7 = 7 * p.m
    prob.camera_prepare()
    prob.jacobian(want_B=False)

    def snapshot():
        return [torch.tril(prob.S[:n7]).clone(), prob.S[n7].clone(), prob.e.clone(), prob.V_inv.clone(), prob.eps_b.clone()]

    def clear():
        prob.sys.zero_()
        prob.V_inv.fill_(float("nan"))
        prob.eps_b.fill_(float("nan"))
        prob.e.fill_(float("nan"))

    clear()
    prob.plan.set_fused(0)
    prob.assemble(MU, by_products=False, B=False)
    ref = snapshot()
    prob.solve()
    prob.back_substitute(B=False)
    db_ref = prob.delta_b.clone()
    da = prob.delta_a.clone()
    prob.plan.set_fused(1)
    names = ("S", "e row", "e", "V_inv", "eps_b")
    for label, run in (("rows", lambda: prob.assemble(MU, by_products=False, B=False)), ("state", lambda: prob.assemble_state(MU))):
        clear()
        run()
        got = snapshot()
        for name, a, b in zip(names, ref, got):
            assert bool(torch.isfinite(b).all().item()), (label, name)
            err = float((a - b).abs().max().item()) / float(a.abs().max().item())
            print(label, name, err)
            assert err <= (1e-11 if name == "V_inv" else 1e-12), (label, name, err)
        assert bool(((ref[0] == 0) == (got[0] == 0)).all().item()), label  # same blocks written
        first = got
        clear()
        run()
        again = snapshot()
        for a, b in zip(first, again):
            assert bool((a == b).all().item()), label  # bit-reproducible
    # back-substitution from the state (the workspace still holds the far points' rows of the last assembly)
    prob.delta_a.copy_(da)
    X0 = prob.X.clone()
    prob.delta_b.fill_(float("nan"))
    prob.back_substitute_state(update_points=True)
    err = float((db_ref - prob.delta_b).abs().max().item()) / float(db_ref.abs().max().item())
    print("delta_b", err)
    assert err <= 1e-11
    assert float((prob.X - (X0 + 0.5 * prob.delta_b)).abs().max().item()) <= 1e-15 * float(X0.abs().max().item())
    return prob


# m, n, views per point: every window width, ids shuffled
SHUFFLED = [(30, 6000, (5, 7)), (16, 2000, 8), (40, 9000, (3, 5)), (25, 4000, 6), (64, 20001, (2, 4)), (50, 8000, (2, 3))]


@pytest.mark.parametrize("m,n,k", SHUFFLED)
def test_shuffled_point_ids(m, n, k):
    from sfm_amd import synthetic
    from sfm_amd.device import Plan

    p = synthetic.make_problem(m, n, k, visibility="banded", seed=m, shuffle_ids=True)
    host = Plan(p.cam_idx, p.pt_idx, where="host")
    dev = Plan(p.cam_idx, p.pt_idx, where="device")
    _plans_equal(host, dev)
    assert dev.fused == 1 and dev.classes[0]["n_near"] == n and dev.far is None
    # every observation sits once in the table, at the slot of its camera in its run's window
    so = dev.read("slot_obs").reshape(-1, 8)
    key_of_pos = np.repeat(dev.read("run_key"), np.diff(dev.read("run_pt0")))
    kc = dev.read("key_cam").reshape(-1, 8)
    pos, sl = np.nonzero(so >= 0)
    obs = so[pos, sl]
    assert np.array_equal(np.sort(obs), np.arange(p.n_obs))
    assert np.array_equal(p.pt_idx[obs], dev.read("fz_pt")[pos]) and np.array_equal(p.cam_idx[obs], kc[key_of_pos[pos], sl])
    _compare_paths(p, expect_far=False)


@pytest.mark.parametrize("m,n,seed,where,long_every,expect_wide", [
    (60, 8000, 1, "device", 20, False), (60, 8000, 1, "host", 20, False), (90, 30000, 2, "device", 20, False),
    (40, 30000, 3, "device", 4, True)])
def test_near_and_far_points_in_one_plan(m, n, seed, where, long_every, expect_wide):
    from sfm_amd.device import Plan

    p, kinds = mixed_problem(m, n, seed, long_every=long_every)
    host = Plan(p.cam_idx, p.pt_idx, where="host")
    dev = Plan(p.cam_idx, p.pt_idx, where="device")
    _plans_equal(host, dev)
    assert dev.fused == 1 and dev.fused_slots == 7 and dev.far is not None
    near0 = np.zeros(n, dtype=bool)
    near0[dev.read("fz_pt")] = True
    near1 = np.zeros(n, dtype=bool)
    near1[dev.read("fz_pt", cls=1)] = True
    assert not (near0 & near1).any()
    assert near0[kinds == 0].all()          # short tracks: windows of 7 consecutive cameras
    assert not near0[kinds == 1].any()      # long tracks (9..14 cameras): windows of 16 where enough of them share one, else far
    assert near0[kinds == 2].all()          # distant camera sets that many points share: windows of their own
    assert not near0[kinds == 3].any()      # camera sets of a single point
    print("long tracks in windows of 16:", int(near1[kinds == 1].sum()), "of", int((kinds == 1).sum()), dev.classes[1])
    if expect_wide:
        assert dev.classes[1]["on"] and near1[kinds == 1].sum() * 2 > (kinds == 1).sum()
    near = near0 | near1
    assert np.array_equal(dev.read("far_pt"), np.flatnonzero(~near))
    assert dev.far.n == int((~near).sum()) and dev.far.n_obs == int((~near)[p.pt_idx].sum())
    _compare_paths(p, expect_far=True, where=where)


@pytest.mark.parametrize("m,n,k", [(40, 12000, (9, 16)), (30, 6000, (10, 13)), (64, 30000, (3, 16))])
def test_windows_of_sixteen_cameras(m, n, k):
    """Points seen by 9..16 consecutive cameras: the class of near points with windows of 16 cameras (16
    lanes per point, 4 points per step), alone and beside the class with windows of up to 8."""
    from sfm_amd import synthetic
    from sfm_amd.device import Plan

    p = synthetic.make_problem(m, n, k, visibility="banded", seed=m, shuffle_ids=True)
    host = Plan(p.cam_idx, p.pt_idx, where="host")
    dev = Plan(p.cam_idx, p.pt_idx, where="device")
    _plans_equal(host, dev)
    c1 = dev.classes[1]
    assert dev.fused == 1 and c1["on"] and c1["slots"] == 16 and c1["n_near"] > 0
    so = dev.read("slot_obs", cls=1).reshape(-1, 16)
    key_of_pos = np.repeat(dev.read("run_key", cls=1), np.diff(dev.read("run_pt0", cls=1)))
    kc = dev.read("key_cam", cls=1).reshape(-1, 16)
    pos, sl = np.nonzero(so >= 0)
    obs = so[pos, sl]
    assert np.unique(obs).size == obs.size
    assert np.array_equal(p.pt_idx[obs], dev.read("fz_pt", cls=1)[pos]) and np.array_equal(p.cam_idx[obs], kc[key_of_pos[pos], sl])
    counts = np.bincount(p.pt_idx, minlength=n)
    assert (counts[dev.read("fz_pt", cls=1)] > dev.fused_slots).all() or dev.classes[0]["n_near"] == 0
    print("classes", dev.classes, "far", dev.n_far)
    _compare_paths(p)


@pytest.mark.parametrize("visibility,m,n", [("tracks", 60, 20000), ("mixed", 50, 6000), ("long", 40, 12000)])
def test_general_iterations_against_oracle(visibility, m, n):
    """Two iterations of the loop on plans with far points against the numpy oracle: the state path and
    the camera-pair path (bars of tests/test_gpu_fused.py::test_fused_iteration_against_oracle)."""
    from oracle import ba_oracle as orc
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem
    from tests.helpers import relerr

    if visibility == "tracks":
        p = synthetic.make_problem(m, n, 6, visibility="tracks", seed=11)
    elif visibility == "long":
        p = synthetic.make_problem(m, n, (9, 16), visibility="banded", seed=4, shuffle_ids=True)
    else:
        p, _ = mixed_problem(m, n, 5)
    R, T, X = p.R.copy(), p.T.copy(), p.X.copy()
    ref = [orc.gn_iteration_arrays(p.K, R, T, X, p.cam_idx, p.pt_idx, p.x_true) for _ in range(2)]
    costs = {}
    for fused in (1, 0):
        prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
        assert prob.plan.fused == 1 and prob.from_state
        assert (prob.plan.far is not None) == (visibility != "long")
        prob.plan.set_fused(fused)
        costs[fused] = []
        for it in range(2):
            c_ref, da_ref, db_ref = ref[it]
            prob.iterate_async()
            c = float(prob.cost_buf.item())
            prob.check_flags()
            costs[fused].append(c)
            print(f"{visibility} fused={fused} iteration {it}: cost {c!r} oracle {c_ref!r} rel {abs(c - c_ref) / c_ref:.2e}")
            assert abs(c - c_ref) <= 2e-8 * c_ref, (fused, it, c, c_ref)
            assert relerr(prob.delta_a.cpu().numpy(), da_ref) < 1e-7
            assert relerr(prob.delta_b.cpu().numpy(), db_ref) < 1e-7
        assert relerr(prob.points(), X) < 1e-8
    # (the two paths against each other: 1e-9, as in tests/test_gpu_fused.py; 5e-9 on the problem of long
    # tracks, whose cost rises by half per iteration - the reference's loop diverges there - and with it
    # every rounding difference of the first step)
    for a, b in zip(costs[0], costs[1]):
        assert abs(a - b) <= (5e-9 if visibility == "long" else 1e-9) * a, costs


@pytest.mark.parametrize("case", ["ba_4x16_bigrot", "ba_8x400", "ba_20x3000", "ba_6x3000_long", "real_box", "real_boot"])
def test_golden_step_through_the_state_path(case):
    """The reference's own S, e, V_inv, eps_b of the first Gauss-Newton step (golden fixtures, generated by
    the reference's code) against the state path: k_fused_state for the near points, the far plan for the
    rest.  1e-12 of the largest entry, V_inv 1e-11 (the kernel forms the rows with one reciprocal per
    quotient rule; cond(V_i) amplifies the last-place differences)."""
    import torch

    from sfm_amd import bundle_adjustment as ba
    from tests.helpers import frames_from_fixture, load, relerr

    d = load(case)
    frames, map_ = frames_from_fixture(d)
    prob, vi, pi, x_t = ba._problem_from(frames, map_)
    assert prob.plan.fused == 1 and prob.from_state, "the fixture must run through the fused state path"
    print(case, "near points", prob.plan.n_near, "of", prob.n, "windows", prob.plan.n_keys, "runs", prob.plan.n_runs,
          "slots", prob.plan.fused_slots)
    if "step_S" not in d.files:
        pytest.skip("fixture holds the loop only")
    n7 = 7 * prob.m
    prob.camera_prepare()
    prob.assemble_state(MU)
    S = torch.tril(prob.S[:n7]).cpu().numpy()
    S = S + np.tril(S, -1).T
    assert relerr(S, d["step_S"]) < 1e-12
    assert np.array_equal(S == 0, d["step_S"] == 0)
    assert relerr(prob.e.cpu().numpy(), d["step_e"]) < 1e-11
    assert relerr(prob.V_inv.cpu().numpy().reshape(-1, 3, 3), d["step_V_inv"]) < 1e-11
    assert relerr(prob.eps_b.cpu().numpy(), d["step_eps_b"]) < 1e-12
    prob.solve()
    prob.back_substitute_state()
    assert relerr(prob.delta_a.cpu().numpy(), d["step_delta_a"]) < 1e-6
    assert relerr(prob.delta_b.cpu().numpy(), d["step_delta_b"]) < 1e-6


def test_patterns_without_a_fused_path():
    from sfm_amd import synthetic
    from sfm_amd.device import Plan

    p = synthetic.make_problem(50, 20000, 8, visibility="uniform", seed=1)
    for where in ("host", "device"):
        pl = Plan(p.cam_idx, p.pt_idx, where=where)
        assert pl.fused == 0 and pl.n_near == 0 and pl.far is None
    # seventeen consecutive cameras do not fit a window
    r = synthetic.make_problem(40, 3000, 17, visibility="banded", seed=3)
    assert Plan(r.cam_idx, r.pt_idx).fused == 0
