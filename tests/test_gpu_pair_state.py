"""The camera-pair kernel that forms its rows of A from the state (sfm_ba_schur_assemble_pairs_state)
against the one that gathers rows of A and couplings (sfm_ba_schur_assemble) on the same state: S (lower
triangle), e, eps_b within 1e-12 of the largest entry, V_inv 1e-11 (bars of tests/test_gpu_state.py: the rows
formed in the kernel use one reciprocal per quotient rule where the Jacobian kernel divides); the
per-camera squared residuals against sfm_ba_cost; and the loop on a plan without runs against the numpy
oracle (bars of tests/test_gpu_parity.py for that size)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MU = 0.5


def _rel(a, b):
    return float((a - b).abs().max().item() / b.abs().max().item())


@pytest.mark.parametrize("visibility,m,n,k,damped", [("uniform", 24, 3000, 5, True), ("uniform", 60, 9000, 7, False),
                                                     ("banded", 40, 400, 30, True), ("tracks", 50, 6000, 6, True)])
def test_pairs_from_state_against_gathered_rows(visibility, m, n, k, damped):
    import torch

    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    p = synthetic.make_problem(m, n, k, visibility=visibility, seed=11)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, state=False)
    prob.plan.set_fused(False)   # the whole plan on the camera-pair path
    prob.camera_prepare()
    prob.jacobian(want_B=False)
    mu_u = MU if damped else 0.0
    n7 = prob.n7

    def run(which):
        prob.sys.zero_()
        prob.V_inv.fill_(float("nan"))
        prob.eps_b.fill_(float("nan"))
        if which == "rows":
            prob.assemble(MU, by_products=False, B=False)
            if not damped:
                prob.S[:n7].diagonal().sub_(MU)
        else:
            from sfm_amd import _lib

            ptr = _lib.ptr
            _lib.check(prob.lib.sfm_ba_schur_assemble_pairs_state(
                prob.plan.handle, ptr(prob.A), ptr(prob.cam_blk), ptr(prob.X), ptr(prob.x_true), ptr(prob.x_pred), MU, mu_u,
                ptr(prob.S), ptr(prob.e), ptr(prob.V_inv), ptr(prob.eps_b), ptr(prob.ssq), ptr(prob.workspace), None))
        torch.cuda.synchronize()
        return (torch.tril(prob.S[:n7]).clone(), prob.S[n7].clone(), prob.e.clone(), prob.V_inv.clone(), prob.eps_b.clone())

    S0, r0, e0, V0, b0 = run("rows")
    S1, r1, e1, V1, b1 = run("state")
    assert _rel(S1, S0) <= 1e-12 and _rel(r1, r0) <= 1e-12 and _rel(e1, e0) <= 1e-12
    assert _rel(b1, b0) <= 1e-12 and _rel(V1, V0) <= 1e-11
    ssq = prob.ssq.clone()
    prob.cost()
    assert float(((ssq - prob.ssq).abs() / prob.ssq).max().item()) <= 1e-13
    # twice the same bits
    S2 = run("state")[0]
    assert torch.equal(S1, S2)


def test_loop_without_runs_follows_the_oracle():
    from oracle import ba_oracle as orc
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    p = synthetic.make_problem(30, 4000, 6, visibility="uniform", seed=5)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    prob.plan.set_fused(False)   # every point through the camera-pair kernels
    R, T, X = p.R.copy(), p.T.copy(), p.X.copy()
    for it in range(4):
        prob.iterate_async()
        c_gpu = float(prob.cost_buf.item())
        c_sep = prob.cost()
        c_ref = orc.gn_iteration_arrays(p.K, R, T, X, p.cam_idx, p.pt_idx, p.x_true)[0]
        assert abs(c_gpu - c_sep) <= 1e-13 * c_sep
        assert abs(c_gpu - c_ref) <= 2e-8 * c_ref, (it, c_gpu, c_ref)
    prob.check_flags()
    Rg, Tg, _, _ = prob.cameras()
    assert np.max(np.abs(Rg - R)) <= 1e-7 * np.max(np.abs(R))
    assert np.max(np.abs(prob.points() - X)) <= 1e-7 * np.max(np.abs(X))
