"""sfm_ba_cost_noalloc (partial sums in a device array of the library, no allocation, no workspace)
against sfm_ba_cost through the C ABI: per-camera sums and the cost must be bit-identical, on two
calls in a row of which the second has the smaller input.  (The refusal of a plan beyond the array's
65536 work items is not exercised: such a plan needs tens of thousands of cameras, whose camera-pair
table alone takes gigabytes.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(prob):
    import torch

    from sfm_amd import _lib

    lib, p = _lib.load(), _lib.ptr
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ssq_a, cost_a = torch.full_like(prob.ssq, -1.0), torch.full((1,), -1.0, dtype=torch.float64, device=prob.device)
    ssq_b, cost_b = torch.full_like(prob.ssq, -2.0), torch.full((1,), -2.0, dtype=torch.float64, device=prob.device)
    _lib.check(lib.sfm_ba_cost(prob.plan.handle, p(prob.P), p(prob.X), p(prob.cam_idx), p(prob.pt_idx), p(prob.x_true),
                               p(ssq_a), p(cost_a), p(prob.workspace), st))
    _lib.check(lib.sfm_ba_cost_noalloc(prob.plan.handle, p(prob.P), p(prob.X), p(prob.cam_idx), p(prob.pt_idx),
                                       p(prob.x_true), p(ssq_b), p(cost_b), st))
    torch.cuda.synchronize()
    return ssq_a.cpu().numpy(), float(cost_a.item()), ssq_b.cpu().numpy(), float(cost_b.item())


def test_cost_without_allocation_matches_cost():
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    big = synthetic.make_problem(40, 30000, 6, visibility="banded", seed=5)     # several work items per camera
    small = synthetic.make_problem(7, 300, 3, visibility="uniform", seed=6)    # one item per camera
    for p in (big, small):
        prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
        ssq_a, cost_a, ssq_b, cost_b = _both(prob)
        assert np.array_equal(ssq_a, ssq_b)
        assert cost_a == cost_b and np.isfinite(cost_b) and cost_b > 0
        # and once more on the same problem: the array keeps the last call's contents, nothing depends on them
        _, _, ssq_c, cost_c = _both(prob)
        assert np.array_equal(ssq_b, ssq_c) and cost_c == cost_b


def test_no_allocation_and_capture():
    """The call leaves the device's free memory as it was, and can be captured into a graph (the array's
    address was looked up when the plan was created): the replay writes the same bits."""
    import torch

    from sfm_amd import _lib, synthetic
    from sfm_amd.device import DeviceProblem

    p = synthetic.make_problem(12, 4000, 5, visibility="banded", seed=8)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    lib, ptr = _lib.load(), _lib.ptr
    ssq_ref, cost_ref, _, _ = _both(prob)
    ssq = torch.zeros_like(prob.ssq)
    cost = torch.zeros((1,), dtype=torch.float64, device=prob.device)

    def call():
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.sfm_ba_cost_noalloc(prob.plan.handle, ptr(prob.P), ptr(prob.X), ptr(prob.cam_idx), ptr(prob.pt_idx),
                                           ptr(prob.x_true), ptr(ssq), ptr(cost), st))

    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    reserved0 = torch.cuda.memory_reserved()
    call()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0 and torch.cuda.memory_reserved() == reserved0
    assert np.array_equal(ssq.cpu().numpy(), ssq_ref) and float(cost.item()) == cost_ref
    ssq.zero_()
    cost.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    torch.cuda.synchronize()
    ssq.zero_()
    cost.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(ssq.cpu().numpy(), ssq_ref) and float(cost.item()) == cost_ref
