"""The cost as a by-product of the assembly: per-camera sums of squared residuals out of
sfm_ba_schur_assemble_state (near points: per-slot sums of the fused kernel; far points: the diagonal
blocks of the camera-pair kernel) against sfm_ba_cost on the same state - equal to 1e-13 relative per
camera (the sums are associated differently) - and the loop's costs against the separate cost pass."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("visibility,m,n,k", [("banded", 30, 6000, (5, 7)), ("banded", 16, 2000, 8), ("tracks", 60, 20000, 6),
                                                ("banded", 40, 9000, (3, 5))])
def test_cost_from_the_assembly(visibility, m, n, k):
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    p = synthetic.make_problem(m, n, k, visibility=visibility, seed=3, shuffle_ids=True)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    assert prob.plan.fused and prob.from_state
    c_ref = prob.cost()
    ssq_ref = prob.ssq.clone()
    prob.camera_prepare()
    prob.ssq.fill_(float("nan"))
    prob.cost_buf.fill_(float("nan"))
    prob.assemble_state(0.5, want_cost=True)
    ssq = prob.ssq.clone()
    c = float(prob.cost_buf.item())
    rel = float(((ssq - ssq_ref).abs() / ssq_ref).max().item())
    print("per-camera relative difference", rel, "cost", c, c_ref)
    assert rel <= 1e-13 and abs(c - c_ref) <= 1e-13 * c_ref
    # the loop: costs of three iterations against a cost pass of their own after each
    costs = []
    for _ in range(3):
        prob.iterate_async()
        c_it = float(prob.cost_buf.item())
        c_sep = prob.cost()
        assert abs(c_it - c_sep) <= 1e-13 * c_sep
        costs.append(c_it)
    prob.check_flags()
    # ... and the same loop with the rows of A materialised and the separate cost pass
    ref = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, state=False)
    for c_it in costs:
        ref.iterate_async()
        c_ref = float(ref.cost_buf.item())
        assert abs(c_it - c_ref) <= 1e-9 * c_ref
