#!/usr/bin/env python3
"""Benchmark of the bundle-adjustment Gauss-Newton iteration on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--visibility MODE]
                    [--scaling weak|strong]

A step is one full Gauss-Newton iteration of the reference's loop (per-camera records,
Jacobian blocks and predictions, Schur assembly, reduced camera solve, back-substitution with
the point update, camera update, cost) on synthetic data that is resident in HBM before the
timed region starts.  With N > 1 there is one process per GPU: either the ranks come from
`python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, when
WORLD_SIZE is unset, this script starts the N rank processes itself before it touches a GPU.
`--scaling weak` (default): every rank holds the workload's point count (N x points, same
cameras).  `--scaling strong`: ONE problem (default workload venice1800, 1 M points) is
sharded by point id over the N ranks.  The camera system is all-reduced over RCCL each
iteration.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 vector = matrix peak (datasheet)


def algorithmic_bytes(N, n, m, s_blocks, with_B=False):
    """Bytes per Gauss-Newton iteration a kernel group must move at minimum (fp64 values, int32
    indices).  `s_blocks` = number of 7 x 7 blocks of S that hold a non-zero (camera pairs with a
    shared point, both triangles): m * m when every pair of cameras shares a point, far fewer for
    banded visibility - only those blocks are written.  The loop reads dF/dX = -dF/dC out of the
    rows of A and never materialises B: 136 / 152 / 120 bytes per observation for the three stages.
    `with_B` prices B (48 bytes per observation) as a materialised array as well, which is what the
    reference's interface hands between its stages: 184 / 200 / 168."""
    b = 48 if with_B else 0
    return {
        "jacobian": (136 + b) * N + 24 * n + 128 * m,
        "assembly": (152 + b) * N + 96 * n + 392 * s_blocks + 560 * m,
        "back_substitution": (120 + b) * N + 120 * n + 56 * m,
    }


# kernels that implement each stage (a kernel absent from a run's profile is skipped: the per-camera
# kernels only run when a caller asks for U / eps_a)
GROUPS = {
    "jacobian": ["k_jacobian"],
    "assembly": ["k_fused_state", "k_fused_runs", "k_fused_reduce", "k_fused_finish", "k_point", "k_camera_items",
                 "k_camera_finish", "k_pair_blocks", "k_pair_state", "k_pair_finish"],
    "back_substitution": ["k_back_substitute", "k_back_substitute_state"],
}


def measured_traffic(workload, visibility):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE with the
    gfx950 read-side correction), or {} when none was taken for this workload.  The newest file that
    holds the workload wins (kernels change between rounds of measurements)."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f).get(f"{workload}/{visibility}")
        except (OSError, ValueError):
            continue
        if doc:
            return doc
    return {}


def cpu_baseline(workload, visibility, seed, sample_points=100000, iters=4):
    """The numpy oracle (a port of the reference's algorithm) on a bounded sample of the workload."""
    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic
    from oracle import ba_oracle as orc

    p = synthetic.make_workload(workload, visibility=visibility, seed=seed, points=min(sample_points, synthetic.WORKLOADS[workload][1]))
    R, T, X = p.R.copy(), p.T.copy(), p.X.copy()
    t0 = time.perf_counter()
    for _ in range(iters):
        orc.gn_iteration_arrays(p.K, R, T, X, p.cam_idx, p.pt_idx, p.x_true)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info

        threads = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    except Exception:  # noqa: BLE001
        threads = os.cpu_count() or 1
    return {
        "value": p.n_obs * iters / dt,
        "unit": "obs/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"{workload} restricted to its first {p.n} points ({p.n_obs} observations, {p.m} cameras), "
                  f"{iters} Gauss-Newton iterations of the vectorised numpy oracle in {dt:.1f} s "
                  f"(BLAS threads = cores; the Python-level work is single-threaded)",
        "iters_per_sec": iters / dt,
    }


def triangulation_line(n, reps):
    """`--triangulate N`: throughput of the two-view triangulation kernel with inputs resident in HBM
    (64 algorithmic bytes per correspondence: x1, x2 in, X out) and the numpy oracle beside it."""
    import torch

    import sfm_amd  # noqa: F401
    from sfm_amd import _lib, synthetic

    lib = _lib.load()
    rng = np.random.default_rng(0)
    K = synthetic.K_DEFAULT
    X = np.concatenate([rng.uniform(-2, 2, size=(n, 3)) + np.array([0, 0, 8.0]), np.ones((n, 1))], axis=1)
    P1 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = K @ np.hstack([synthetic.rodrigues(np.array([0.02, -0.05, 0.01])), np.array([[0.7], [0.05], [-0.1]])])
    noise = lambda: np.concatenate([rng.normal(0, 0.5, size=(n, 2)), np.zeros((n, 1))], axis=1)  # noqa: E731
    x1 = P1 @ X.T
    x1 = (x1 / x1[-1]).T + noise()
    x2 = P2 @ X.T
    x2 = (x2 / x2[-1]).T + noise()
    d1, d2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    p1, p2 = torch.from_numpy(P1.reshape(-1).copy()).cuda(), torch.from_numpy(P2.reshape(-1).copy()).cuda()
    out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    p = _lib.ptr
    for _ in range(3):
        _lib.check(lib.sfm_ba_triangulate(p(d1), p(d2), p(p1), p(p2), n, p(out), None))
    torch.cuda.synchronize()
    _lib.profile_read()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(lib.sfm_ba_triangulate(p(d1), p(d2), p(p1), p(p2), n, p(out), None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    _lib.profile_enable(False)
    ms, cnt = _lib.profile_read()["k_triangulate"]
    kernel_s = ms / cnt * 1e-3
    from oracle import triangulation_oracle as orc

    ns = min(n, 400_000)
    t1 = time.perf_counter()
    ref = orc.triangulate_pts(x1[:ns], x2[:ns], P1, P2)
    tc = time.perf_counter() - t1
    err = float(np.max(np.abs(out[:ns].cpu().numpy().astype(np.float64) - ref)) / np.max(np.abs(ref)))
    print(json.dumps({
        "metric": "two-view triangulation, correspondences/s", "n": n, "value": n / dt, "unit": "correspondences/s",
        "kernel_us": kernel_s * 1e6, "dtype": "f64", "data": "synthetic",
        "roofline": {"bound": "hbm", "achieved": 64.0 * n / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": 64.0 * n / kernel_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "note": "bound by the fp64 arithmetic of the Jacobi sweeps, not by memory"},
        "cpu_baseline": {"value": ns / tc, "unit": "correspondences/s", "cores": 1, "kind": "port",
                         "sample": f"numpy batched SVD oracle on the first {ns} correspondences in {tc:.2f} s"},
        "max_rel_diff_vs_oracle": err}))


def pnp_line(n, reps):
    """`--pnp N`: the reference's RANSAC pose estimation (200 minimal samples, inlier counts, refit) on N
    3D-2D correspondences, whole call including sampling and transfers, and the numpy oracle beside it."""
    import torch

    import sfm_amd  # noqa: F401
    from sfm_amd import pose_estimation as pe, synthetic

    rng = np.random.default_rng(0)
    K = synthetic.K_DEFAULT
    R = synthetic.rodrigues(np.array([0.3, -0.2, 0.1]))
    T = np.array([[0.2], [-0.1], [7.0]])
    X = np.concatenate([rng.uniform(-2, 2, size=(n, 3)), np.ones((n, 1))], axis=1)
    x = K @ np.hstack([R, T]) @ X.T
    x = (x / x[-1]).T
    x[:, :2] += rng.normal(0, 0.5, size=(n, 2))
    bad = rng.random(n) < 0.3
    x[bad, :2] += rng.uniform(-300, 300, size=(int(bad.sum()), 2))
    np.random.seed(1)
    pe.estimate_pose_Linear_PnP_RANSAC(x, X, K, threshold=8.0)
    torch.cuda.synchronize()
    np.random.seed(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        Rg, Tg, ng = pe.estimate_pose_Linear_PnP_RANSAC(x, X, K, threshold=8.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    from oracle import pnp_oracle as orc

    np.random.seed(1)
    t1 = time.perf_counter()
    Ro, To, no = orc.estimate_pose_Linear_PnP_RANSAC(x, X, K, threshold=8.0)
    tc = time.perf_counter() - t1
    np.random.seed(1)
    Rg, Tg, ng = pe.estimate_pose_Linear_PnP_RANSAC(x, X, K, threshold=8.0)
    print(json.dumps({
        "metric": "RANSAC linear PnP, pose estimations/s", "n": n, "value": 1.0 / dt, "unit": "calls/s",
        "ms_per_call": dt * 1e3, "dtype": "f64", "data": "synthetic, 30 % outliers",
        "cpu_baseline": {"value": 1.0 / tc, "unit": "calls/s", "cores": 1, "kind": "port",
                         "sample": f"numpy oracle, one call on the same {n} correspondences in {tc:.2f} s"},
        "inliers": int(ng), "inliers_oracle": int(no),
        "max_rel_diff_vs_oracle": float(max(np.max(np.abs(Rg - Ro)) / np.max(np.abs(Ro)),
                                            np.max(np.abs(Tg - To)) / np.max(np.abs(To))))}))


def free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n_ranks):
    """Start `n_ranks` copies of this command line as rank processes (one per GPU) and wait for them.
    Runs before anything in this process has touched a GPU: the children are fresh interpreters with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, as torch.distributed.run would set
    them; rank 0's standard output (the JSON line) is this process's.  Returns the exit code: 0 when
    every rank ended with 0, else the first non-zero code (the other ranks are ended then)."""
    import subprocess

    backend = os.environ.get("SFM_BA_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        import torch

        have = torch.cuda.device_count()  # counts devices without initialising the runtime
        if have < n_ranks:
            print(f"bench.py: --gpus {n_ranks} needs {n_ranks} visible GPUs, found {have} "
                  f"(SFM_BA_BENCH_BACKEND=gloo rehearses the multi-rank path on one GPU)", file=sys.stderr)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        for pr in list(pending):
            code = pr.poll()
            if code is None:
                continue
            pending.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:   # a rank died: its peers would wait in a collective for ever
                    other.terminate()
        time.sleep(0.05)
    return rc


def stage_times(prob, steps, warmup, world, barrier):
    """(wall seconds for `steps` iterations, per-kernel (ms total, launches) of the stage kernels, cost
    after the last of those iterations)."""
    from sfm_amd import _lib

    for _ in range(warmup):
        prob.iterate_async()
    barrier()
    prob.check_flags()
    _lib.profile_read()
    # HIP events on the launch stream around the kernels of the three HBM stages and around the
    # reduced solve (what the roofline object is computed from).  A single rank enqueues nothing
    # between the library's launches, so consecutive timed launches share an event and the records
    # cost ~0.02 ms per iteration: they stay inside the timed region.  With several ranks the
    # collectives sit between the launches, every launch needs two records of its own (~0.09 ms per
    # iteration, which would be charged to the scaling curve): the timed region then runs without
    # records and the kernels are timed in a pass of their own behind it.
    timed = sorted({k for names in GROUPS.values() for k in names} | {"solve_reduced"})
    if world == 1:
        _lib.profile_enable(True, chained=True, kernels=timed)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        prob.iterate_async()
    barrier()
    dt = time.perf_counter() - t0
    final_cost = float(prob.cost_buf.item())  # after warm-up + K iterations, whatever the world size
    if world > 1:
        _lib.profile_enable(True, chained=False, kernels=timed)
        for _ in range(steps):
            prob.iterate_async()
        barrier()
    _lib.profile_enable(False)
    return dt, _lib.profile_read(), final_cost


STATE_KERNELS = ("k_fused_state", "k_back_substitute_state")


def stage_groups(prof, steps, N, n, m, s_blocks, pmc):
    """Per-stage record: time, what bounds the stage, the stage's algorithmic bytes (and the same with B
    priced as a materialised array), counter traffic where a PMC pass of these kernels is on file.
    A stage whose kernel reads rows of A from memory is HBM-bound and `GBps` = algorithmic bytes / time
    is its achieved bandwidth.  A stage whose kernel forms the rows itself (k_fused_state,
    k_back_substitute_state) moves a fraction of those bytes and is bound by the fp64 pipe: for it the
    same quotient is reported as `priced_GBps` - the rate at which the stage's algorithmic bytes are
    retired, which may exceed the HBM peak and is not a bandwidth - and `traffic_GBps` is what it moves."""
    alg, alg_b = algorithmic_bytes(N, n, m, s_blocks), algorithmic_bytes(N, n, m, s_blocks, with_B=True)
    stages = dict(GROUPS)
    if "k_fused_state" in prof:
        # the rows of A are formed inside the assembly kernel: one stage does the work of two and is
        # priced with the bytes of both (what the two stages would have to move between them).  The far
        # points of the plan go through the Jacobian and camera-pair kernels: same stage.
        stages = {"jacobian+assembly": list(GROUPS["jacobian"]) + list(GROUPS["assembly"]),
                  "back_substitution": GROUPS["back_substitution"]}
        for d in (alg, alg_b):
            d["jacobian+assembly"] = d["jacobian"] + d["assembly"]
    groups = {}
    for g, names in stages.items():
        present = [nm for nm in names if nm in prof]
        ms = sum(prof[nm][0] for nm in present) / steps
        traffic = None
        if present and all(nm in pmc for nm in present):
            traffic = sum(pmc[nm]["bytes"] * prof[nm][1] / steps for nm in present)
        forms_rows = any(nm in STATE_KERNELS for nm in present)
        rate = alg[g] / (ms * 1e-3) / 1e9 if ms > 0 else None
        rate_b = alg_b[g] / (ms * 1e-3) / 1e9 if ms > 0 else None
        rec = {"ms": ms, "kernels": present, "bound": "fp64" if forms_rows else "hbm",
               "algorithmic_bytes": alg[g], "algorithmic_bytes_with_B": alg_b[g]}
        if forms_rows:
            rec.update({"priced_GBps": rate, "priced_GBps_with_B": rate_b})
        else:
            rec.update({"GBps": rate, "GBps_with_B": rate_b})
        rec.update({"traffic": traffic,
                    "traffic_GBps": traffic / (ms * 1e-3) / 1e9 if traffic and ms > 0 else None})
        groups[g] = rec
    return groups


def roofline_record(prof, groups, plan):
    """The `roofline` object of the line: the stage that takes the longest, against the ceiling that
    bounds its slowest kernel."""
    dominant = max(groups, key=lambda g: groups[g]["ms"])
    gr = groups[dominant]
    slowest = max(gr["kernels"], key=lambda nm: prof[nm][0])
    name = (f"{dominant} stage ({'+'.join(gr['kernels'])}; slowest kernel {slowest}, "
            f"{prof[slowest][0] / prof[slowest][1] * 1e3:.1f} us per launch)")
    if gr["bound"] == "hbm":
        return {
            "bound": "hbm", "kernel": name, "achieved": gr["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gr["GBps"] / HBM_PEAK_GBS, "traffic": gr["traffic"],
            "achieved_with_B": gr["GBps_with_B"], "frac_with_B": gr["GBps_with_B"] / HBM_PEAK_GBS, "groups": groups,
            "note": "achieved = algorithmic bytes of the stage / HIP-event time of its kernels: 136 / 152 / 120 bytes per "
                    "observation for Jacobian / assembly / back-substitution (B = -dF/dC is part of the rows of A and "
                    "never written); *_with_B adds B as a materialised array (184 / 200 / 168)",
        }
    # the fused kernel forms the rows of A itself: what it moves is a fraction of the stage's algorithmic
    # bytes and what bounds it is the fp64 pipe of the SIMDs, which the matrix and the vector instructions share
    kn = "k_fused_state"
    mfma_flops = 0.0
    for c in plan.classes:
        if not c["n_runs"]:
            continue
        slots = c["slots"]
        tiles = 2 if slots <= 4 else (3 if slots <= 7 else (4 if slots <= 8 else 7))
        k_steps = 6 if slots <= 8 else 3  # 8 points x 3 / 4 columns per matrix instruction; 4 points x 3 / 4
        mfma_flops += c["n_steps"] * (tiles * (tiles + 1) // 2) * k_steps * 2048.0
    launches_per_step = sum(1 for c in plan.classes if c["n_runs"])
    t = prof[kn][0] / prof[kn][1] * launches_per_step * 1e-3  # (one launch per class of near points)
    tf = mfma_flops / t / 1e12
    return {
        "bound": "mfma", "kernel": name, "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": tf / FP64_PEAK_TFLOPS, "traffic": gr["traffic"],
        "traffic_GBps": gr["traffic_GBps"],
        "traffic_frac_of_hbm_peak": gr["traffic_GBps"] / HBM_PEAK_GBS if gr["traffic_GBps"] else None,
        "matrix_flops": mfma_flops,
        "achieved_algorithmic": gr["priced_GBps"], "achieved_algorithmic_unit": "GB/s",
        "frac_algorithmic": gr["priced_GBps"] / HBM_PEAK_GBS,
        "achieved_algorithmic_with_B": gr["priced_GBps_with_B"],
        "frac_algorithmic_with_B": gr["priced_GBps_with_B"] / HBM_PEAK_GBS,
        "groups": groups,
        "note": "the dominant kernel forms the rows of A in registers and is bound by the fp64 pipe, not by HBM: achieved = "
                "v_mfma_f64_16x16x4 flops it executes per iteration (lower-triangle tiles x 6 per step of 8 points; x 3 per "
                "step of 4 points in windows of 16 cameras) / its "
                "HIP-event time, against the dense fp64 peak; the same pipe also executes the kernel's vector fp64 "
                "arithmetic, about as many cycles again.  traffic = HBM bytes per iteration of the stage from the "
                "committed PMC passes (null: no pass for this workload).  achieved_algorithmic prices the bytes the two "
                "stages it replaces would have to move (136 + 152 bytes per observation, 184 + 200 with B) over the "
                "stage's time: the rate at which that work is retired, not bytes moved",
    }


def solve_record(prof, steps, n7, hb):
    solve_ms = prof.get("solve_reduced", (0.0, 1))[0] / steps
    from sfm_amd import _lib

    # edge of the blocks the library's band paths work on for this size and band (0: general path)
    blk = int(_lib.load().sfm_ba_reduced_solve_block(int(n7), int(hb)))
    wide = blk if blk > 64 else 0
    if hb <= 64:
        # band Cholesky + two band substitutions
        flops, path = n7 * (hb + 1.0) ** 2 + 4.0 * n7 * (hb + 1.0), "block cyclic reduction (band)"
    elif wide:
        flops = n7 * (hb + 1.0) ** 2 + 4.0 * n7 * (hb + 1.0)
        path = f"block cyclic reduction on superblocks of {wide} unknowns (band)"
    else:
        flops, path = n7 ** 3 / 3.0 + 2.0 * n7 * n7, "blocked Cholesky (dense)"
    return {
        "bound": "latency" if hb <= 64 or wide else "mfma",
        "path": path,
        "half_bandwidth": hb,
        "flops": flops,
        "ms": solve_ms,
        "achieved": flops / (solve_ms * 1e-3) / 1e12 if solve_ms > 0 else None,
        "peak": FP64_PEAK_TFLOPS,
        "unit": "TFLOP/s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None,
                    help="default: c4_500x300k (weak scaling), venice1800 (strong scaling)")
    ap.add_argument("--visibility", default="banded", choices=["banded", "uniform", "closures", "loop", "tracks"])
    ap.add_argument("--shuffle-ids", action="store_true",
                    help="relabel the point ids at random (the plan sorts the points itself)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's point count per GPU; strong: one problem sharded over the GPUs")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-uniform-record", "--no-extra-records", dest="no_uniform_record", action="store_true",
                    help="skip the short extra runs (other visibility patterns of the same size, the 1 M-point workload)")
    ap.add_argument("--triangulate", type=int, default=0, metavar="N",
                    help="benchmark the two-view triangulation kernel on N correspondences instead")
    ap.add_argument("--pnp", type=int, default=0, metavar="N",
                    help="benchmark RANSAC linear PnP on N correspondences instead")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "venice1800" if args.scaling == "strong" else "c4_500x300k"
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if args.pnp > 0:
        pnp_line(args.pnp, max(args.steps, 1))
        return
    if args.triangulate > 0:
        triangulation_line(args.triangulate, max(args.steps, 1))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    if os.environ.get("SFM_BA_BENCH_SELFTEST"):
        # launcher check without a GPU: the ranks meet over gloo, add up their ranks, rank 0 reports
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        if os.environ.get("SFM_BA_BENCH_SELFTEST") == f"fail{rank}":
            os._exit(7)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "scaling": args.scaling, "workload": args.workload,
                              "rank_sum": float(t.item())}))
        dist.destroy_process_group()
        return

    import sfm_amd  # noqa: F401
    from sfm_amd import _lib, ordering, synthetic
    from sfm_amd.device import DeviceProblem
    from sfm_amd.dist import local_problem, packed_system_elements, shard_bounds

    if os.environ.get("SFM_BA_BENCH_WATCHDOG"):
        # a rank that is still running after this many seconds prints where every thread stands and exits
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["SFM_BA_BENCH_WATCHDOG"]), exit=True)
    # Rehearsal on a box with fewer GPUs than ranks: SFM_BA_BENCH_BACKEND=gloo puts every rank on
    # GPU 0 and moves the buffers with gloo (RCCL refuses two ranks on one device).  The numbers of
    # such a run mean nothing; it exists to exercise the multi-rank code path.
    backend = os.environ.get("SFM_BA_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPUs are visible")
        torch.cuda.set_device(local_rank)
    else:
        torch.cuda.set_device(0)
    group = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD
    plan_where = os.environ.get("SFM_BA_PLAN_WHERE", "device")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def describe(prob):
        pl = prob.plan
        if not (pl.fused and prob.from_state):
            return "fused point-major on materialised rows of A" if pl.fused else "camera-pair"
        txt = "fused point-major, rows of A formed in the kernel (" + " + ".join(
            f"{c['n_near']} near points in {c['n_keys']} windows of {c['slots']} cameras" for c in pl.classes if c["n_runs"])
        if pl.n_far:
            txt += f"; {pl.n_far} far points with {pl.n_far_obs} observations through the camera-pair kernels"
        return txt + ")"

    def order_text(prob):
        if prob.cam_order is None:
            return "acquisition order"
        nat = prob.natural_half_bandwidth
        return (f"folded (half bandwidth {nat if nat is not None else '?'} in acquisition order, {prob.half_bandwidth} "
                f"folded; on the device)")

    def sub_record(workload, visibility, steps, shuffle_ids=False, reorder=False, n_points=None):
        """A short run of another workload / visibility on this GPU: stage times, solve, what bounds what."""
        mm = synthetic.WORKLOADS[workload][0]
        q = synthetic.make_workload(workload, visibility=visibility, seed=args.seed, points=n_points, shuffle_ids=shuffle_ids)
        pr = DeviceProblem(q.K, q.R, q.T, q.X, q.cam_idx, q.pt_idx, q.x_true, n_cams=mm, check=True, plan_where=plan_where,
                           reorder=reorder)
        sdt, sprof, cost = stage_times(pr, steps, 2, 1, barrier)
        pr.check_flags()
        nb = 2 * pr.plan.n_blocks - mm
        groups = stage_groups(sprof, steps, pr.N, pr.n, mm, nb, measured_traffic(workload, visibility))
        rec = {
            "workload": f"{workload}: {mm} cameras x {q.n} points x {q.n_obs} observations, {visibility} visibility"
                        + (", point ids relabelled at random" if shuffle_ids else ""),
            "steps": steps, "ms_per_step": sdt / steps * 1e3, "value": q.n_obs * steps / sdt, "unit": "obs/s",
            "nonzero_S_blocks": nb, "camera_order": order_text(pr), "assembly": describe(pr), "final_cost": cost,
            "roofline": roofline_record(sprof, groups, pr.plan),
            "solve": solve_record(sprof, steps, 7 * mm, pr.half_bandwidth),
        }
        del pr, q
        torch.cuda.empty_cache()
        return rec

    m, n_workload, k = synthetic.WORKLOADS[args.workload]
    n_total = n_workload * world if args.scaling == "weak" else n_workload
    full = synthetic.make_workload(args.workload, visibility=args.visibility, seed=args.seed, points=n_total,
                                   shuffle_ids=args.shuffle_ids)
    N_total = full.n_obs
    revisits = args.visibility in ("loop", "closures") and os.environ.get("SFM_BA_REORDER", "1") != "0"
    cam_order = None
    imbalance = None
    if world > 1:
        if revisits:
            # every rank computes the same camera order from the unsharded lists (deterministic), so that
            # the shards agree on the band they exchange
            hb_nat = ordering.half_bandwidth(np.arange(m), full.cam_idx, full.pt_idx, full.n)
            hb_new, pos, order = ordering.narrowest_order(full.cam_idx, full.pt_idx, m, full.n, hb_nat)
            if pos is not None and 2 * hb_new <= hb_nat:
                cam_order = order
        prob_h, _ = local_problem(full, rank, world)
        bounds = shard_bounds(full.pt_idx, full.n, world)
        per_rank = np.diff(np.searchsorted(np.sort(full.pt_idx), bounds))
        imbalance = float(per_rank.max() / per_rank.mean())
    else:
        prob_h = full
    del full
    prob = DeviceProblem(prob_h.K, prob_h.R, prob_h.T, prob_h.X, prob_h.cam_idx, prob_h.pt_idx, prob_h.x_true,
                         n_cams=m, group=group, check=(world == 1), plan_where=plan_where,
                         reorder=revisits and world == 1, cam_order=cam_order)
    n_blocks = prob.plan.n_blocks
    N_loc, n_loc = prob.N, prob.n

    # timed region: K iterations after W warm-up iterations
    dt, prof, final_cost = stage_times(prob, args.steps, args.warmup, world, barrier)
    prob.check_flags()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # the same K iterations without the event records, for the cost of the instrumentation
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        prob.iterate_async()
    barrier()
    dt_plain = time.perf_counter() - t1

    # and once more with every kernel timed, for the table of the small ones (outside the timed region)
    _lib.profile_enable(True)
    for _ in range(args.steps):
        prob.iterate_async()
    barrier()
    _lib.profile_enable(False)
    for name, v in _lib.profile_read().items():
        prof.setdefault(name, v)

    allreduce_ms = None
    if world > 1:
        # the exchange alone: K all-reduces of the buffer an iteration exchanges, outside the timed region
        buf = prob.compact if prob.band_sharded else prob.tri
        if buf is not None:
            barrier()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                prob._all_reduce(buf)
            barrier()
            allreduce_ms = (time.perf_counter() - t2) / args.steps * 1e3

    if rank == 0:
        per_kernel = {name: {"ms_avg": ms / cnt, "launches_per_step": cnt / args.steps} for name, (ms, cnt) in prof.items()}
        s_blocks = 2 * n_blocks - m  # plan counts pairs j <= k
        pmc = measured_traffic(args.workload, args.visibility) if world == 1 and not args.shuffle_ids else {}
        groups = stage_groups(prof, args.steps, N_loc, n_loc, m, s_blocks, pmc)
        roofline = roofline_record(prof, groups, prob.plan)
        n7 = 7 * m
        solve = solve_record(prof, args.steps, n7, prob.half_bandwidth)
        out = {
            "metric": "BA Gauss-Newton observations/sec (synthetic %d cams x %d pts)" % (m, n_total),
            "value": N_total * args.steps / dt,
            "unit": "obs/s",
            "iters_per_sec": args.steps / dt,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_without_events": dt_plain / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {m} cameras x {n_total} points x {N_total} observations, "
                            f"{args.visibility} visibility" + (", point ids relabelled at random" if args.shuffle_ids else "")
                            + f", seed {args.seed}, "
                            + (f"{n_workload} points per GPU" if args.scaling == "weak"
                               else f"one problem sharded by point id over {world} GPU(s)"),
                "cameras": m, "points": n_total, "observations": N_total, "visibility": args.visibility,
                "nonzero_S_blocks": s_blocks,
                "camera_order": order_text(prob),
                "assembly": describe(prob),
                "parallelism": "1 gpu" if world == 1 else f"point-id sharding x{world}, all-reduce of "
                               f"{packed_system_elements(m, prob.half_bandwidth) * 8 / 1e6:.1f} MB per iteration",
            },
            "final_cost": final_cost,
            "roofline": roofline,
            "solve": solve,
            "kernels": per_kernel,
        }
        if world > 1:
            out["multi_gpu"] = {
                "allreduce_ms": allreduce_ms, "allreduce_MB": packed_system_elements(m, prob.half_bandwidth) * 8 / 1e6,
                "solve_ms": solve["ms"], "observations_per_rank_max_over_mean": imbalance,
                "note": "allreduce_ms: the exchange buffer of one iteration all-reduced alone, outside the timed region; "
                        "solve_ms: the replicated reduced solve (rank 0).  Both are the parts of an iteration that do not "
                        "shrink with the number of GPUs",
            }
        if world == 1 and args.visibility == "banded" and not args.shuffle_ids and not args.no_uniform_record:
            # short runs of the same size with other visibility patterns, and of the 1 M-point workload
            del prob
            torch.cuda.empty_cache()
            usteps = max(2, min(args.steps, 5))
            out["shuffled_ids"] = sub_record(args.workload, "banded", usteps, shuffle_ids=True, n_points=n_total)
            out["tracks_visibility"] = sub_record(args.workload, "tracks", usteps, n_points=n_total)
            out["uniform_visibility"] = sub_record(args.workload, "uniform", usteps, n_points=n_total)
            if m >= 4 * (k if isinstance(k, int) else k[1]):
                # one lap that ends where it started (a closed loop): cameras kept in the folded order
                out["loop_visibility"] = sub_record(args.workload, "loop", usteps, reorder=True, n_points=n_total)
            if args.workload != "venice1800":
                out["venice1800_banded"] = sub_record("venice1800", "banded", usteps)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.visibility, args.seed)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
