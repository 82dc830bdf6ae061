"""The benchmark's workloads at full size - 500 cameras x 300 000 points x 1.95 M observations with
banded visibility (the default bench line: fused assembly, cyclic-reduction solve) and with uniform
visibility (every camera pair shares points: camera-pair assembly, a 3 500 x 3 500 system with
cond ~ 1e9 through the general MFMA Cholesky), and 1 800 cameras x 1 M points x 5 M observations
banded - checked through properties that do not need the CPU oracle at that size:

* the plan built on the device equals the plan built on the host, array for array;
* S is symmetric to rounding, and sampled 7 x 7 blocks of S and entries of e equal the defining sums
  delta_jk U_j - sum_i Y_ij W_ik^T and eps_a_j - sum_i Y_ij eps_b_i (/root/reference/src/sparse_ba.py:189-271)
  recomputed in numpy from the device's own A, B and residuals;
* the solve leaves a residual |S x - e| / |e| below 1e-9, and the fast path of the loop (no B, folded
  U) produces the same system as the path with every by-product;
* sampled points satisfy V_i delta_b_i = eps_b_i - sum_j W_ij^T delta_a_j (:279-290);
* the cost falls over three iterations;
* (uniform) the device solution of the dense system is as close to an extended-precision solution
  as numpy's LU, the reference's solver (/root/reference/src/sparse_ba.py:274-276), to within a
  factor of five, and both are two orders of magnitude inside eps * cond(S).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MU = 0.5


@pytest.fixture(scope="module", params=[("c4_500x300k", "banded"), ("c4_500x300k", "uniform"), ("venice1800", "banded")],
                ids=lambda wv: "-".join(wv))
def problem(request):
    from sfm_amd import synthetic

    w, v = request.param
    p = synthetic.make_workload(w, visibility=v, seed=0)
    p.meta["visibility_mode"] = v
    return p


def test_device_plan_equals_host_plan(problem):
    from sfm_amd.device import Plan

    host = Plan(problem.cam_idx, problem.pt_idx, where="host")
    dev = Plan(problem.cam_idx, problem.pt_idx, where="device")
    assert host.n_blocks == dev.n_blocks and host.half_bandwidth == dev.half_bandwidth
    banded = problem.meta["visibility_mode"] == "banded"
    assert (host.half_bandwidth <= 64) == banded and host.fused == dev.fused == int(banded)
    for f in Plan.FIELDS:
        assert np.array_equal(host.read(f), dev.read(f)), f


def test_system_and_step_properties(problem):
    import torch

    from sfm_amd.device import DeviceProblem

    p = problem
    m, n = p.m, p.n
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    prob.camera_prepare()
    prob.jacobian()
    prob.assemble(MU)  # every by-product: U, eps_a, e, V_inv, eps_b
    n7 = 7 * m
    S = prob.S[:n7].clone()
    e = prob.e.clone()
    # off-diagonal blocks are mirrored exactly; a diagonal block is a sum of A^T T A terms whose (r, c)
    # and (c, r) entries round separately
    assert float((S - S.T).abs().max().item()) <= 1e-13 * float(S.abs().max().item())

    # the loop's fast path builds the same system
    prob.sys.zero_()
    prob.assemble(MU, by_products=False, B=False)
    # (it writes the lower triangle only, which is all the solvers read)
    S_fast, e_fast = torch.tril(prob.S[:n7]), prob.S[n7].view(m, 7).clone()
    S_fast = S_fast + torch.tril(S_fast, -1).T
    scale = float(S.abs().max().item())
    assert float((torch.tril(S_fast) - torch.tril(S)).abs().max().item()) <= 1e-12 * scale
    assert float((e_fast - e).abs().max().item()) <= 1e-12 * float(e.abs().max().item())

    # defining sums for a few camera pairs, from the device's A, B, residuals
    A = prob.A.cpu().numpy().reshape(-1, 2, 7)
    B = prob.B.cpu().numpy().reshape(-1, 2, 3)
    eps = p.x_true - prob.x_pred.cpu().numpy()
    Vinv = prob.V_inv.cpu().numpy().reshape(n, 3, 3)
    eps_b = prob.eps_b.cpu().numpy()
    Sh, eh = S.cpu().numpy(), e.cpu().numpy()
    cam, pt = p.cam_idx, p.pt_idx
    W = np.einsum("nri,nrj->nij", A, B)  # (N,7,3)  A^T B
    rng = np.random.default_rng(0)
    for j in rng.choice(m, 4, replace=False):
        oj = np.nonzero(cam == j)[0]
        for k in (j, min(j + 1, m - 1), min(j + 3, m - 1)):
            ok = np.nonzero(cam == k)[0]
            common, ij, ik = np.intersect1d(pt[oj], pt[ok], return_indices=True)
            Y = W[oj[ij]] @ Vinv[common]
            blk = -np.einsum("nij,nkj->ik", Y, W[ok[ik]])
            if j == k:
                blk = blk + np.einsum("nri,nrj->ij", A[oj], A[oj]) + MU * np.eye(7)
            got = Sh[7 * j:7 * j + 7, 7 * k:7 * k + 7]
            assert np.max(np.abs(got - blk)) <= 1e-10 * np.max(np.abs(blk)), (j, k)
        Yj = W[oj] @ Vinv[pt[oj]]
        ej = np.einsum("nri,nr->i", A[oj], eps[oj]) - np.einsum("nij,nj->i", Yj, eps_b[pt[oj]])
        assert np.max(np.abs(eh[j] - ej)) <= 1e-10 * np.max(np.abs(ej))

    # solve: residual against the system as assembled (the narrow-band solve only reads S; the
    # general one factors it in place)
    prob.solve()
    prob.check_flags()
    x = prob.delta_a.view(-1)
    r = S_fast @ x - e_fast.reshape(-1)
    res = float(r.norm().item()) / float(e_fast.norm().item())
    # normwise backward error: what a backward-stable factorisation bounds whatever cond(S) is
    bwd = float(r.norm().item()) / (float(torch.linalg.matrix_norm(S_fast, ord="fro").item()) * float(x.norm().item())
                                    + float(e_fast.norm().item()))
    print(f"{p.meta['visibility_mode']} m={m}: solve residual {res:.2e} of |e|, backward error {bwd:.2e}")
    assert res <= 1e-9 and bwd <= 1e-14
    if p.meta["visibility_mode"] == "uniform":
        Sd, ed = S_fast.cpu().numpy(), e_fast.reshape(-1).cpu().numpy()
        da_lu = np.linalg.solve(Sd, ed)
        Sl, el, xx = Sd.astype(np.longdouble), ed.astype(np.longdouble), da_lu.astype(np.longdouble)
        for _ in range(6):
            xx = xx + np.linalg.solve(Sd, (el - Sl @ xx).astype(np.float64)).astype(np.longdouble)
        xx = xx.astype(np.float64)
        err_gpu = np.linalg.norm(x.cpu().numpy() - xx) / np.linalg.norm(xx)
        err_lu = np.linalg.norm(da_lu - xx) / np.linalg.norm(xx)
        print(f"dense solve error vs extended precision at {n7} unknowns, cond {np.linalg.cond(Sd):.2e}: "
              f"device {err_gpu:.2e}, numpy LU {err_lu:.2e}")
        # both are far inside the eps * cond ~ 2.5e-7 that either factorisation guarantees; the
        # Cholesky factors' error (1.6e-9 measured) is a few times LU's with partial pivoting (4.6e-10)
        assert err_gpu <= max(5.0 * err_lu, 1e-10) and err_gpu <= 0.05 * 2.2e-16 * np.linalg.cond(Sd)
        del Sd, Sl
        # the factorisation overwrote S: the next stages rebuild it

    # back-substitution identity on sampled points
    prob.back_substitute(B=False)
    da = prob.delta_a.cpu().numpy()
    db = prob.delta_b.cpu().numpy()
    for i in rng.choice(n, 50, replace=False):
        oi = np.nonzero(pt == i)[0]
        rhs = eps_b[i] - np.einsum("nij,ni->j", W[oi], da[cam[oi]])
        V = np.linalg.inv(Vinv[i])
        assert np.max(np.abs(V @ db[i] - rhs)) <= 1e-9 * max(np.max(np.abs(rhs)), 1e-300)

    # three iterations of the loop: the cost falls
    del A, B, W
    costs = [prob.cost()]
    for _ in range(3):
        prob.iterate_async()
        costs.append(float(prob.cost_buf.item()))
    prob.check_flags()
    assert all(b < a for a, b in zip(costs, costs[1:])), costs
    torch.cuda.synchronize()
