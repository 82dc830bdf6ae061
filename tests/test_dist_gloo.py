"""Point-id sharding over ranks, exercised with world_size 2 and 3 on CPU (gloo).

Each rank takes its shard from ``sfm_amd.dist.local_problem``, forms its partial camera system
with the CPU oracle (damping left out of U, as a rank of the GPU path does), the partial systems
are summed with ``torch.distributed.all_reduce`` in the packed layout the GPU path uses, the
damping is added once, and the result must equal the single-process system.  The replicated
solve and the local back-substitution then reproduce the single-process updates.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _partial_system(p, m, mu):
    """Oracle partial system of a shard: S without damping on U, e, per-camera squared residuals."""
    from oracle import ba_oracle as orc

    vi, pi = p.cam_idx.astype(np.int64), p.pt_idx.astype(np.int64)
    C = -np.einsum("mji,mj->mi", p.R, p.T)
    P = p.K[None] @ np.concatenate([p.R, p.T[:, :, None]], 2)
    Xh = np.concatenate([p.X[pi], np.ones((len(pi), 1))], 1)
    proj = np.einsum("nij,nj->ni", P[vi], Xh)
    x_pred = proj[:, :2] / proj[:, 2:3]
    Ks = np.broadcast_to(p.K, (m, 3, 3))
    A, B = orc.jacobians_batched(Ks[vi], p.R[vi], C[vi], p.X[pi], Q=orc.rotation_matrix_to_quaternion(p.R)[vi])
    N = len(vi)
    n = p.n
    eps = p.x_true - x_pred
    U = orc._segment_sum(vi, np.einsum("nap,nar->npr", A, A), m)            # no damping
    V = orc._segment_sum(pi, np.einsum("naq,nar->nqr", B, B), n) + mu * np.identity(3)
    V_inv = np.linalg.inv(V)
    W = np.einsum("nap,naq->npq", A, B)
    Y = np.einsum("npq,nqr->npr", W, V_inv[pi])
    eps_a = orc._segment_sum(vi, np.einsum("nap,na->np", A, eps), m)
    eps_b = orc._segment_sum(pi, np.einsum("naq,na->nq", B, eps), n)
    e = eps_a - orc._segment_sum(vi, np.einsum("npq,nq->np", Y, eps_b[pi]), m)
    S = np.zeros((7 * m, 7 * m))
    order = np.argsort(pi, kind="stable")
    starts = np.searchsorted(pi[order], np.arange(n + 1))
    for i in range(n):
        ob = order[starts[i]:starts[i + 1]]
        for a in ob:
            for b in ob:
                S[7 * vi[a]:7 * vi[a] + 7, 7 * vi[b]:7 * vi[b] + 7] -= Y[a] @ W[b].T
    for j in range(m):
        S[7 * j:7 * j + 7, 7 * j:7 * j + 7] += U[j]
    ssq = np.bincount(vi, weights=(eps * eps).sum(1), minlength=m)
    return S, e, ssq, (A, B, V_inv, W, eps_b)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sfm_amd  # noqa: F401
        from sfm_amd import synthetic
        from sfm_amd.dist import local_problem, packed_system_elements, shard_bounds

        m, n, mu = 6, 150, 0.5
        full = synthetic.make_problem(m, n, k=3, visibility="uniform", seed=4)
        local, (p0, p1) = local_problem(full, rank, world)
        bounds = shard_bounds(full.pt_idx, full.n, world)
        assert bounds[0] == 0 and bounds[-1] == n and np.all(np.diff(bounds) > 0)
        assert local.n == p1 - p0 and local.pt_idx.min() >= 0 and local.pt_idx.max() < local.n
        assert np.all(np.diff(local.cam_idx) >= 0)  # frame-major order is preserved inside a shard

        S, e, ssq, (A, B, V_inv, W, eps_b) = _partial_system(local, m, mu)
        n7 = 7 * m
        from sfm_amd.dist import pack_lower, unpack_lower

        nt = n7 * (n7 + 1) // 2 + n7
        assert packed_system_elements(m) == nt + m  # lower triangle of S, e, per-camera cost sums
        buf = torch.zeros(packed_system_elements(m), dtype=torch.float64)
        buf[:nt] = torch.from_numpy(pack_lower(S, e))
        buf[nt:] = torch.from_numpy(ssq)
        dist.all_reduce(buf)
        Sg, eg = unpack_lower(buf[:nt].numpy(), n7)
        Sg = Sg + mu * np.identity(n7)   # damping enters once
        cost = float(np.sqrt(buf[nt:].numpy()).sum() / m)
        delta_a = np.linalg.solve(Sg, eg).reshape(m, 7)
        vi, pi = local.cam_idx.astype(np.int64), local.pt_idx.astype(np.int64)
        from oracle import ba_oracle as orc

        dd = np.einsum("npq,np->nq", W, delta_a[vi])
        delta_b = np.einsum("nqr,nr->nq", V_inv, eps_b - orc._segment_sum(pi, dd, local.n))
        q.put((rank, p0, p1, Sg, eg, cost, delta_a, delta_b))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_system_matches_single_process(world):
    sys.path.insert(0, ROOT)
    import sfm_amd  # noqa: F401
    from oracle import ba_oracle as orc
    from sfm_amd import synthetic

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda r: r[0])

    m, n, mu = 6, 150, 0.5
    full = synthetic.make_problem(m, n, k=3, visibility="uniform", seed=4)
    S1, e1, ssq1, _ = _partial_system(full, m, mu)
    S1 = S1 + mu * np.identity(7 * m)
    # the single-process system equals the oracle's own sba_compute
    vi, pi = full.cam_idx.astype(np.int64), full.pt_idx.astype(np.int64)
    R, T, X = full.R.copy(), full.T.copy(), full.X.copy()
    c_after, da_ref, db_ref = orc.gn_iteration_arrays(full.K, R, T, X, vi, pi, full.x_true, mu=mu, force_sparse=False)
    cost1 = float(np.sqrt(ssq1).sum() / m)
    assert abs(cost1 - orc.cost_arrays(full.K, full.R, full.T, full.X, vi, pi, full.x_true)) < 1e-12 * cost1
    db = np.zeros((n, 3))
    for rank, p0, p1, Sg, eg, cost, delta_a, delta_b in results:
        assert np.max(np.abs(Sg - S1)) <= 1e-12 * np.max(np.abs(S1))
        assert np.max(np.abs(eg - e1.reshape(-1))) <= 1e-11 * np.max(np.abs(e1))
        assert abs(cost - cost1) <= 1e-12 * cost1
        assert np.max(np.abs(delta_a - da_ref)) <= 1e-6 * np.max(np.abs(da_ref))
        db[p0:p1] = delta_b
    assert np.max(np.abs(db - db_ref)) <= 1e-6 * np.max(np.abs(db_ref))
    # every rank solved the same replicated system
    assert all(np.array_equal(results[0][3], r[3]) for r in results[1:])


def test_shard_bounds_balance():
    sys.path.insert(0, ROOT)
    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic
    from sfm_amd.dist import local_problem, shard_bounds

    p = synthetic.make_problem(20, 5000, k=(3, 7), visibility="banded", seed=1)
    for world in (1, 2, 4, 8):
        b = shard_bounds(p.pt_idx, p.n, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == p.n
        sizes = [local_problem(p, r, world)[0].n_obs for r in range(world)]
        assert sum(sizes) == p.n_obs
        assert max(sizes) - min(sizes) <= 8  # within one point's observations of each other


def test_shard_bounds_never_empty():
    """Skewed counts (one point holds almost every observation) and as many ranks as points: every
    rank still owns a point; fewer points than ranks is an error raised before any collective."""
    sys.path.insert(0, ROOT)
    import sfm_amd  # noqa: F401
    from sfm_amd.dist import shard_bounds

    pt = np.concatenate([np.zeros(1000, dtype=np.int32), np.arange(1, 6, dtype=np.int32)])
    for world in (2, 3, 6):
        b = shard_bounds(pt, 6, world)
        assert b[0] == 0 and b[-1] == 6 and np.all(np.diff(b) >= 1), b
    pt = np.concatenate([np.arange(5, dtype=np.int32), np.full(1000, 5, dtype=np.int32)])
    for world in (2, 4, 6):
        b = shard_bounds(pt, 6, world)
        assert b[0] == 0 and b[-1] == 6 and np.all(np.diff(b) >= 1), b
    with pytest.raises(ValueError):
        shard_bounds(pt, 6, 7)


def test_lower_triangle_layout():
    sys.path.insert(0, ROOT)
    import sfm_amd  # noqa: F401
    from sfm_amd.dist import pack_lower, packed_system_elements, unpack_lower

    rng = np.random.default_rng(0)
    n7 = 21
    M = rng.normal(size=(n7, n7))
    S, e = M + M.T, rng.normal(size=n7)
    buf = pack_lower(S, e)
    assert buf.size == packed_system_elements(3) - 3
    for i in (0, 5, 20):
        assert np.array_equal(buf[i * (i + 1) // 2: i * (i + 1) // 2 + i + 1], S[i, : i + 1])
    S2, e2 = unpack_lower(buf, n7)
    assert np.array_equal(S2, S) and np.array_equal(e2, e)
    # 49 MB instead of 98 MB at 500 cameras
    assert abs(packed_system_elements(500) * 8 / 1e6 - 49.1) < 0.1
