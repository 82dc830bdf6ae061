"""Pin the CPU oracle to the golden vectors captured from the reference.

Tolerances: the oracle restates Python loops as whole-array operations, so sums
may associate differently; every stage must agree to a few ulp of the largest
entry (1e-12 relative), the solve outputs to 1e-7 of the largest entry (the
reduced system has condition number 1e8-1e9).
"""
import json
import os

import numpy as np
import pytest

from oracle import ba_oracle as orc
from tests.helpers import GOLDEN, frames_from_fixture, load, relerr

CASES = ["ba_4x16_bigrot", "ba_8x400", "ba_20x3000"]


def test_quaternion_branches():
    d = load("quaternion")
    assert set(d["branch"].tolist()) == {0, 1, 2, 3}
    Q = orc.rotation_matrix_to_quaternion(d["R"])
    assert np.max(np.abs(Q - d["Q"])) < 1e-15
    for i in (0, 1, 2, 3):
        k = int(np.nonzero(d["branch"] == i)[0][0])
        assert np.max(np.abs(orc.rotation_matrix_to_quaternion(d["R"][k]) - d["Q"][k])) < 1e-15
    R = orc.quaternion_to_rotation_matrix(d["q_in"])
    assert np.max(np.abs(R - d["R"])) < 1e-15
    assert np.array_equal(orc.compute_dR_dQ_mat(d["Q"]), d["dR_dQ"])


def test_single_jacobians():
    d = load("single_jacobian")
    m, n = d["R"].shape[0], d["X"].shape[0]
    Kb = np.broadcast_to(d["K"], (m * n, 3, 3))
    Rb = np.repeat(d["R"], n, axis=0)
    Cb = np.repeat(d["C"], n, axis=0)
    Xb = np.tile(d["X"], (m, 1))
    A, B = orc.jacobians_batched(Kb, Rb, Cb, Xb)
    assert relerr(A, d["J_pose"]) < 1e-13
    assert relerr(B, d["J_X"]) < 1e-13
    assert relerr(A[:, :, 4:], d["J_C"]) < 1e-13
    assert relerr(A[:, :, :4], d["J_Q"]) < 1e-13
    assert np.array_equal(A[:, :, 4:], -B)  # dF/dC is exactly -dF/dX in the reference

    class F:
        pass

    f = F()
    f.K, f.R, f.C = d["K"], d["R"][1], d["C"][1].reshape(3, 1)
    assert relerr(orc.compute_pose_jacobian_mat(f, d["X"][2].reshape(3, 1)), d["J_pose"][n + 2]) < 1e-13
    assert relerr(orc.compute_X_jacobian_mat(f, d["X"][2].reshape(3, 1)), d["J_X"][n + 2]) < 1e-13


@pytest.mark.parametrize("case", CASES)
def test_indices_and_jacobians(case):
    d = load(case)
    frames, map_ = frames_from_fixture(d)
    vi, pi, x_t, x_p = orc.get_indices(frames, map_)
    assert np.array_equal(np.asarray(vi), d["step_vi"])
    assert np.array_equal(np.asarray(pi), d["step_pi"])
    assert np.array_equal(x_t, d["step_x_t"])
    assert relerr(x_p, d["step_x_p"]) < 1e-14
    if "step_A" in d:
        A, B = orc.compute_A_B(frames, map_, vi, pi)
        assert relerr(A, d["step_A"]) < 1e-13
        assert relerr(B, d["step_B"]) < 1e-13
    assert abs(orc.compute_reprojection_error_total(frames, map_) - d["loop_cost0"]) < 1e-12 * d["loop_cost0"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("sparse", [False, True])
def test_gauss_newton_step(case, sparse):
    d = load(case)
    frames, map_ = frames_from_fixture(d)
    vi, pi, x_t, x_p = orc.get_indices(frames, map_)
    A, B = orc.compute_A_B(frames, map_, vi, pi)
    out = orc.sba_compute(vi, pi, x_t, x_p, A, B, None, 0.5, force_sparse=sparse)
    for k in ("U", "V_inv", "eps_a", "eps_b"):
        ok = {"eps_a": "epsilon_a", "eps_b": "epsilon_b"}.get(k, k)
        assert relerr(out[ok], d["step_" + k]) < 1e-12, k
    if "step_W" in d:
        assert relerr(out["W"], d["step_W"]) < 1e-12
        assert relerr(out["Y"], d["step_Y"]) < 1e-12
    assert relerr(out["S"], d["step_S"]) < 1e-12
    assert relerr(out["e"], d["step_e"]) < 1e-11
    assert relerr(out["delta_a"], d["step_delta_a"]) < 1e-7
    assert relerr(out["delta_b"], d["step_delta_b"]) < 1e-7
    # zero pattern of S: blocks of camera pairs without a shared point are exactly zero
    assert np.array_equal(out["S"] == 0, d["step_S"] == 0)


def test_weighted_step():
    d = load("ba_8x400")
    frames, map_ = frames_from_fixture(d)
    vi, pi, x_t, x_p = orc.get_indices(frames, map_)
    A, B = orc.compute_A_B(frames, map_, vi, pi)
    out = orc.sba_compute(vi, pi, x_t, x_p, A, B, d["weights"], 2.0)
    for k in ("U", "V_inv", "S"):
        assert relerr(out[k], d["wstep_" + k]) < 1e-12, k
    assert relerr(out["e"], d["wstep_e"]) < 1e-11
    assert relerr(out["delta_a"], d["wstep_delta_a"]) < 1e-8
    assert relerr(out["delta_b"], d["wstep_delta_b"]) < 1e-8


@pytest.mark.parametrize("case", CASES)
def test_outer_loop(case):
    """Loop exit, per-iteration cost (<= 1e-9 relative) and final state (<= 1e-8)."""
    d = load(case)
    frames, map_ = frames_from_fixture(d)
    ref_costs = d["loop_costs"]
    costs = orc.optimize_using_BA(frames, map_)
    assert len(costs) == len(ref_costs)
    assert np.max(np.abs(np.asarray(costs) - ref_costs) / ref_costs) < 1e-9
    R = np.stack([f.R for f in frames])
    T = np.stack([f.T.reshape(3) for f in frames])
    assert relerr(R, d["loop_R_final"]) < 1e-8
    assert relerr(T, d["loop_T_final"]) < 1e-8
    assert relerr(map_.X[:, :3], d["loop_X_final"]) < 1e-8
    assert relerr(np.stack([f.P for f in frames]), d["loop_P_final"]) < 1e-8


def test_outer_loop_50_iterations():
    """The long fixture never reaches the exit threshold, so all 50 iterations run.

    Its reduced system has condition number ~1e9 and the loop's fixed step scales
    (1.0 / 0.1 / 0.5) let solve round-off leak into the cost: perturbing the
    initial points by 1e-14 relative already moves the per-iteration cost by
    1e-9..1e-8 relative.  The bar here is therefore 1e-9 on the first iteration,
    1e-6 relative on every later iteration and 1e-6 on the final state."""
    d = load("ba_6x3000_long")
    frames, map_ = frames_from_fixture(d)
    ref_costs = d["loop_costs"]
    assert len(ref_costs) == 50
    costs = np.asarray(orc.optimize_using_BA(frames, map_))
    assert len(costs) == 50
    rel = np.abs(costs - ref_costs) / ref_costs
    assert rel[0] < 1e-9
    assert rel.max() < 1e-6
    assert relerr(np.stack([f.R for f in frames]), d["loop_R_final"]) < 1e-6
    assert relerr(map_.X[:, :3], d["loop_X_final"]) < 1e-6


@pytest.mark.parametrize("case", ["real_box", "real_boot"])
def test_real_geometry_loop(case):
    """Cameras and point cloud of the reference's own stage-1 outputs (46 x 5189 x 38117 and
    50 x 8123 x 53553), visibility by projection, the reference's loop stopped after 4 iterations
    (tests/golden/make_real_geometry.py).  On this geometry the reference's iteration raises the
    cost every step (181 -> 651 on `box`), cond(S) is ~1e9 and the LU solve of the reference is
    itself ~1e-9 relative from the exact solution, so differences in summation order reach the cost
    at the 1e-9 level and grow with the iteration count: measured 8e-10 (first) to 4e-9 (fourth).
    Bar: 2e-9 on the first iteration, 2e-8 afterwards, 1e-7 on the final state."""
    d = load(case)
    frames, map_ = frames_from_fixture(d)
    ref_costs = d["loop_costs"]
    costs = np.asarray(orc.optimize_using_BA(frames, map_, max_iter=len(ref_costs)))
    rel = np.abs(costs - ref_costs) / ref_costs
    assert rel[0] < 2e-9
    assert rel.max() < 2e-8
    assert relerr(np.stack([f.R for f in frames]), d["loop_R_final"]) < 1e-7
    assert relerr(np.stack([f.T.reshape(3) for f in frames]), d["loop_T_final"]) < 1e-7
    assert relerr(map_.X[:, :3], d["loop_X_final"]) < 1e-7


def test_error_cases():
    with open(os.path.join(GOLDEN, "errors.json")) as fh:
        cases = json.load(fh)

    def outcome(fn):
        try:
            fn()
            return None, None
        except Exception as ex:  # noqa: BLE001
            return type(ex).__name__, str(ex)

    assert outcome(lambda: orc.check_indices([0, 0, 1, 1], [0, 0, 0, 1])) == (
        cases["duplicate_pair"]["type"], cases["duplicate_pair"]["message"])
    assert outcome(lambda: orc.check_indices([0, 0, 1, 1], [0, 1, 0, 3]))[0] == cases["missing_point_id"]["type"]
    assert outcome(lambda: orc.check_indices([0, 0, 2, 2], [0, 1, 0, 1]))[0] == cases["missing_viewpoint_id"]["type"]
    assert outcome(lambda: orc.check_indices([0, 0, 1], [0, 1, 0, 1]))[0] == cases["length_mismatch"]["type"]
    A, B, x = np.ones((4, 2, 7)), np.ones((4, 2, 3)), np.zeros((4, 2))
    assert outcome(lambda: orc.sba_compute([0, 0, 1, 1], [0, 1, 0, 1], x, x, A, B, None, 0.5)) == (
        cases["too_few_rows"]["type"], cases["too_few_rows"]["message"])
    vi2 = [0] * 12 + [1] * 12 + [2] * 12
    pi2 = list(range(12)) * 3
    rng = np.random.default_rng(0)
    A2, B2, x2 = rng.normal(size=(36, 2, 7)), rng.normal(size=(36, 2, 3)), rng.normal(size=(36, 2))
    wbad = np.tile(np.array([[1.0, 0.2], [0.1, 1.0]]), (36, 1, 1))
    assert outcome(lambda: orc.sba_compute(vi2, pi2, x2, x2, A2, B2, wbad, 0.5)) == (
        cases["asymmetric_weights"]["type"], cases["asymmetric_weights"]["message"])
    assert outcome(lambda: orc.sba_compute(vi2, pi2, x2, x2, A2, B2, None, -1.0))[0] == cases["negative_mu"]["type"]
    assert cases["ok_36obs"]["type"] is None
    assert outcome(lambda: orc.sba_compute(vi2, pi2, x2, x2 * 0.5, A2, B2, None, 0.5))[0] is None
    assert outcome(lambda: orc.quaternion_to_rotation_matrix(np.array([1.0, 1.0, 0.0, 0.0])))[0] == cases["quat_not_unit"]["type"]
    assert outcome(lambda: orc.rotation_matrix_to_quaternion(2.0 * np.eye(3)))[0] == cases["rot_det_not_one"]["type"]
