"""The sharded GPU path: two processes share the one GPU of the test box, each owns half of the
points, and the camera system is summed between them.  The ranks meet over gloo, which reduces a
host copy of the exchange buffer (``DeviceProblem._all_reduce``); on a multi-GPU node the same
code hands the device buffer to RCCL - that branch needs one GPU per rank and is not exercised
here.  Results must match the single-process GPU run."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# (cameras, points, views, visibility): 84 unknowns = 2 blocks of 64; 420 unknowns = 7 blocks (three
# rounds of the cyclic reduction).  The banded ones exchange the packed block-tridiagonal system;
# "dense" (every pair of cameras shares points, half bandwidth 97 > 64) exchanges the lower triangle
# of S and solves with the general Cholesky - blocks this rank's points never touch arrive from the peer.
# "wide": 12 consecutive cameras per point, half bandwidth 83 of 1120 unknowns: lower-triangle exchange,
# then the cyclic reduction on superblocks of 128.
# "loop": one lap that closes on itself; every rank keeps the cameras in the folded order that the
# unsharded lists give (computed by each rank alike before the points are dealt out), in which the
# system is a narrow band again: packed exchange, cyclic reduction.
PROBLEMS = {"small": (12, 3000, 4, "banded"), "chain7": (60, 6000, 5, "banded"), "dense": (14, 3000, 4, "uniform"),
            "wide": (160, 8000, 12, "banded"), "loop": (60, 6000, 4, "loop")}


def _worker(rank, world, port, q, which="small"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sfm_amd  # noqa: F401
        from sfm_amd import synthetic
        from sfm_amd.device import DeviceProblem
        from sfm_amd.dist import local_problem

        m, n, k, vis = PROBLEMS[which]
        full = synthetic.make_problem(m, n, k=k, visibility=vis, seed=6)
        loc, (p0, p1) = local_problem(full, rank, world)
        order = None
        if vis == "loop":
            from sfm_amd import ordering

            hb_nat = ordering.half_bandwidth(np.arange(m), full.cam_idx, full.pt_idx, full.n)
            hb_new, pos, order = ordering.narrowest_order(full.cam_idx, full.pt_idx, m, full.n, hb_nat)
            assert pos is not None and hb_new <= 64 < hb_nat
        prob = DeviceProblem(loc.K, loc.R, loc.T, loc.X, loc.cam_idx, loc.pt_idx, loc.x_true, n_cams=full.m,
                             group=dist.group.WORLD, check=False, cam_order=order)
        assert prob.band_sharded == ((vis == "banded" and k <= 9) or vis == "loop")
        if not prob.band_sharded:
            assert prob.half_bandwidth > 64
        c0 = prob.cost()
        costs = prob.run(max_iter=3, threshold=0.0)
        R, T, C, P = prob.cameras()
        q.put((rank, p0, p1, c0, costs, R, T, prob.points(), prob.delta_a.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["small", "chain7", "dense", "wide", "loop"])
def test_two_ranks_match_single_process(which):
    import torch.multiprocessing as mp

    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, which)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda r: r[0])

    m, n, k, vis = PROBLEMS[which]
    full = synthetic.make_problem(m, n, k=k, visibility=vis, seed=6)
    one = DeviceProblem(full.K, full.R, full.T, full.X, full.cam_idx, full.pt_idx, full.x_true, reorder=(vis == "loop"))
    assert (one.cam_order is not None) == (vis == "loop")
    c0 = one.cost()
    costs = one.run(max_iter=3, threshold=0.0)
    R1, T1, _, _ = one.cameras()
    X1 = one.points()
    X = np.zeros_like(X1)
    for rank, p0, p1, c0r, costs_r, R, T, Xl, da in results:
        assert abs(c0r - c0) <= 1e-12 * c0
        assert len(costs_r) == 3
        # sums over ranks associate differently from the single-process sums: same bar as parity
        assert np.max(np.abs(np.asarray(costs_r) - np.asarray(costs)) / np.asarray(costs)) < 1e-9
        assert np.max(np.abs(R - R1)) < 1e-9 and np.max(np.abs(T - T1)) < 1e-9
        X[p0:p1] = Xl
    assert np.max(np.abs(X - X1)) < 1e-8 * np.max(np.abs(X1))
    assert np.array_equal(results[0][5], results[1][5])  # replicated cameras are bitwise identical


def test_forced_sharded_flow_single_rank():
    """sharded=True on a 1-rank group takes the reduce-then-damp path and must equal the plain path."""
    import torch.distributed as dist

    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic
    from sfm_amd.device import DeviceProblem

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29900 + (os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        # uniform visibility: dense system, exchange of the lower triangle; banded: packed exchange
        for vis, packed in (("uniform", False), ("banded", True)):
            p = synthetic.make_problem(10 if vis == "uniform" else 40, 2000, k=4, visibility=vis, seed=2)
            a = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
            b = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, group=dist.group.WORLD,
                              sharded=True)
            assert b.band_sharded == packed
            # a dense exchange first (the by-hand API), then the loop's packed one: must not mix up
            b.camera_prepare()
            b.jacobian()
            b.assemble(0.5)
            ca = a.run(max_iter=3, threshold=0.0)
            cb = b.run(max_iter=3, threshold=0.0)
            assert np.max(np.abs(np.asarray(ca) - np.asarray(cb)) / np.asarray(ca)) < 1e-10
            assert np.max(np.abs(a.points() - b.points())) < 1e-10
    finally:
        dist.destroy_process_group()


def test_rccl_group_of_one_rank():
    """The collectives of the sharded flow through RCCL on device buffers.  The test box has one
    GPU and RCCL takes one rank per device, so the group has one rank: what is checked is that
    the device tensors go to RCCL as they lie and the flow equals the unsharded one, for the packed
    band exchange and for the lower-triangle exchange.  Runs in a child process, which ends with
    its process group."""
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + (os.getpid() % 500)))
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "scripts", "probes", "rccl_one_rank.py")],
                       capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "band exchange" in r.stdout and "triangle exchange" in r.stdout and "rccl one-rank ok" in r.stdout
