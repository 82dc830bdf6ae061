"""Assembly and back-substitution straight from the state (the rows of A formed inside the kernels
that use them, never written to memory) against the same stages on the rows sfm_ba_jacobian writes.
The rows differ by a few units in the last place (one reciprocal per quotient rule instead of a
division per entry): S, e, eps_b within 1e-12 of the largest entry, V_inv within 1e-11 (cond(V_i)
reaches 1e4 for points seen from two nearby cameras), delta_b within 1e-11."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CASES = [(30, 6000, (5, 7)), (16, 2000, 8), (40, 9000, (3, 5)), (25, 4000, 6), (64, 20001, (2, 4)), (50, 8000, (2, 3))]


@pytest.mark.parametrize("m,n,k", CASES)
def test_state_assembly_matches_rows_assembly(m, n, k):
    import torch
    from sfm_amd import synthetic, _lib
    from sfm_amd.device import DeviceProblem

    p = synthetic.make_problem(m, n, k, visibility="banded", seed=m)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    assert prob.plan.fused == 1 and _lib.load().sfm_ba_plan_state_path(prob.plan.handle) == 1
    prob.camera_prepare()
    prob.jacobian(want_B=False)
    n7 = 7 * m
    prob.sys.zero_()
    prob.assemble(0.5, by_products=False, B=False)
    ref = [torch.tril(prob.S[:n7]).clone(), prob.S[n7].clone(), prob.e.clone(), prob.V_inv.clone(), prob.eps_b.clone()]
    prob.sys.zero_(); prob.V_inv.fill_(float("nan")); prob.eps_b.fill_(float("nan")); prob.e.fill_(float("nan"))
    prob.assemble_state(0.5)
    got = [torch.tril(prob.S[:n7]).clone(), prob.S[n7].clone(), prob.e.clone(), prob.V_inv.clone(), prob.eps_b.clone()]
    for name, a, b in zip(("S", "e row", "e", "V_inv", "eps_b"), ref, got):
        assert bool(torch.isfinite(b).all().item()), name
        err = float((a - b).abs().max().item()) / float(a.abs().max().item())
        print(name, err)
        assert err <= (1e-11 if name == "V_inv" else 1e-12), name


@pytest.mark.parametrize("m,n,k", CASES[:3])
def test_state_back_substitution(m, n, k):
    import torch
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    p = synthetic.make_problem(m, n, k, visibility="banded", seed=m)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    prob.camera_prepare()
    prob.jacobian(want_B=False)
    prob.assemble(0.5, by_products=False, B=False)
    prob.solve()
    prob.back_substitute(B=False)
    ref = prob.delta_b.clone()
    X0 = prob.X.clone()
    prob.delta_b.fill_(float("nan"))
    prob.back_substitute_state(update_points=True)
    err = float((ref - prob.delta_b).abs().max().item()) / float(ref.abs().max().item())
    print("delta_b", err)
    assert err <= 1e-11
    assert float((prob.X - (X0 + 0.5 * prob.delta_b)).abs().max().item()) <= 1e-15 * float(X0.abs().max().item())
