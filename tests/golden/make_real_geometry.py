"""Golden fixtures on the geometry of the reference's own pipeline outputs.

The datasets' images and correspondences are not part of the reference checkout, so the
"boot" / "box" pipeline configurations cannot run; what the checkout does hold are the
cameras and the point clouds those runs produced (data/stage1/<set>/estimated_camera_parameters.json,
estimated_points.ply, camera intrinsics).  This script takes that geometry as ground truth,
derives a visibility list from it (a point is seen by the cameras it projects into, image
1920 x 1080, at most 8 of them), synthesises noisy observations and a perturbed start as
``synthetic.observe_and_perturb`` does, and runs the reference's ``optimize_using_BA`` on it.
Inputs, per-iteration costs and the final state are stored as numbers in ``real_<set>.npz``.

Runs ONLY in the build container (reference mounted at /root/reference); see make_golden.py for
how the reference is imported.

    MPLBACKEND=Agg python tests/golden/make_real_geometry.py [box] [boot]
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (registers the stub modules and imports the reference)
from sfm_amd import synthetic  # noqa: E402

DATA = "/root/reference/data/stage1"
WIDTH, HEIGHT, MAX_VIEWS = 1920.0, 1080.0, 8


def read_ply_xyz(path):
    """Vertices of a binary little-endian PLY whose vertex record is 3 x float32 + 4 x uint8."""
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line == "end_header":
                break
        raw = f.read(16 * n)
    rec = np.frombuffer(raw, dtype=np.dtype([("xyz", "<f4", 3), ("rgba", "u1", 4)]), count=n)
    return rec["xyz"].astype(np.float64)


def load_geometry(name):
    d = os.path.join(DATA, name)
    est = json.load(open(os.path.join(d, "estimated_camera_parameters.json")))
    intr_file = "camera_parameters.json" if os.path.exists(os.path.join(d, "camera_parameters.json")) else "gt_camera_parameters.json"
    K = np.asarray(json.load(open(os.path.join(d, intr_file)))["intrinsics"], dtype=np.float64)
    names = sorted(est["extrinsics"])
    RT = np.stack([np.asarray(est["extrinsics"][k], dtype=np.float64)[:3, :4] for k in names])
    # the estimates are orthonormal only to ~1e-8: project onto rotations so that the reference's
    # det(R) == 1 assertion holds for the ground truth as it does inside the pipeline
    R = []
    for j in range(RT.shape[0]):
        u, _, vt = np.linalg.svd(RT[j, :, :3])
        R.append(u @ vt)
    X = read_ply_xyz(os.path.join(d, "estimated_points.ply"))
    return K, np.stack(R), RT[:, :, 3].copy(), X


def visibility(K, R, T, X):
    xc = np.einsum("mij,nj->mni", R, X) + T[:, None, :]
    uvw = xc @ K.T
    w = uvw[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = uvw[..., 0] / w, uvw[..., 1] / w
    vis = (w > 0.1) & (u >= 0) & (u < WIDTH) & (v >= 0) & (v < HEIGHT)  # (m, n)
    pt, cam = [], []
    for i in range(X.shape[0]):
        cams = np.nonzero(vis[:, i])[0][:MAX_VIEWS]
        if cams.size >= 2:
            pt += [i] * cams.size
            cam += list(cams)
    pt, cam = np.asarray(pt), np.asarray(cam)
    # contiguous ids: drop unseen points / cameras
    keep_p, pt = np.unique(pt, return_inverse=True)
    keep_c, cam = np.unique(cam, return_inverse=True)
    return keep_p, keep_c, pt.astype(np.int64), cam.astype(np.int64)


class _LastCost(float):
    """A cost value that prints and stores like the float it is but answers ``> threshold`` with
    False: handed to the reference's loop as the result of its cost function, it ends the loop
    after that iteration without touching the reference's code."""

    def __gt__(self, other):
        return False


def run_loop_bounded(frames, map_, iterations):
    """The reference's optimize_using_BA, stopped after `iterations` iterations.  On this geometry
    its cost grows from iteration to iteration (the loop would run all 50 and end far from the
    start, in a regime where rounding differences are amplified beyond any parity bar)."""
    true_cost = mg.ref_ba.compute_reprojection_error_total
    calls = {"n": 0}

    def counted(fr, mp):
        c = true_cost(fr, mp)
        calls["n"] += 1
        return _LastCost(c) if calls["n"] >= iterations else c

    mg.ref_ba.compute_reprojection_error_total = counted
    try:
        cost0 = true_cost(frames, map_)
        out = mg.run_loop(frames, map_)
    finally:
        mg.ref_ba.compute_reprojection_error_total = true_cost
    out["cost0"] = np.float64(cost0)
    out["cost_final"] = np.float64(true_cost(frames, map_))
    return out


def make(name, seed=0, iterations=5):
    K, R, T, X = load_geometry(name)
    keep_p, keep_c, pt, cam = visibility(K, R, T, X)
    R, T, X = R[keep_c], T[keep_c], X[keep_p]
    rng = np.random.default_rng(seed)
    # the point cloud's scale differs from the synthetic one: perturb the points relative to their spread
    spread = float(np.median(np.linalg.norm(X - np.median(X, axis=0), axis=1)))
    problem = synthetic.observe_and_perturb(rng, K, R, T, X, pt, cam, pixel_noise=0.5,
                                            point_perturb=0.002 * spread, rot_perturb=0.002,
                                            trans_perturb=0.001 * spread)
    problem.visibility = "projected"
    frames, map_ = synthetic.to_frames_and_map(problem, mg.RefFrame, mg.RefMap)
    data = mg.pack_inputs(problem, frames, map_)
    data["extra_keypoints"] = np.int64(0)
    loop = run_loop_bounded(frames, map_, iterations)
    data["loop_iterations"] = np.int64(iterations)
    data.update({f"loop_{k}": v for k, v in loop.items()})
    # keypoints are float32 by construction; store them as such
    path = os.path.join(HERE, f"real_{name}.npz")
    np.savez_compressed(path, **data)
    print(f"real_{name}: m={problem.m} n={problem.n} N={problem.n_obs} cost0={loop['cost0']:.4f} "
          f"iters={len(loop['costs'])} costs={loop['costs'][:4]}... -> {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["box", "boot"]):
        make(nm)
