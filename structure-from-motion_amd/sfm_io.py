"""File formats either side of the bundle-adjustment path (host side, numpy only).

* the reference's outputs: camera poses as ``estimated_camera_parameters.json`` and the point cloud
  as a binary PLY (``/root/reference/src/pipeline.py:157-169``; the PLY there is written by trimesh:
  little-endian, per vertex 3 x float32 + 4 x uint8);
* the evaluation metrics of ``/root/reference/data/stage1/eval.py:7-63``;
* a loader for "bundle adjustment in the large" text problems, as a source of real large inputs.
"""
from __future__ import annotations

import json

import numpy as np

from .synthetic import BAProblem, rodrigues

PLY_HEADER = ("ply\nformat binary_little_endian 1.0\ncomment written by sfm_amd\nelement vertex {n}\n"
              "property float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\nend_header\n")
_VERTEX = np.dtype([("xyz", "<f4", 3), ("rgba", "u1", 4)])


def write_camera_parameters(path, frames, image_names):
    """``{"extrinsics": {image name: 3x4 [R|T]}}`` as src/pipeline.py:160-169 writes it."""
    assert len(frames) == len(image_names)
    out = {"extrinsics": {name: np.asarray(fr.RT, dtype=np.float64).tolist() for name, fr in zip(image_names, frames)}}
    with open(path, "w") as fp:
        json.dump(out, fp)


def read_camera_parameters(path):
    """Returns (names sorted, RT (m,3,4), K or None) from an ``*_camera_parameters.json``."""
    with open(path) as fp:
        doc = json.load(fp)
    names = sorted(doc.get("extrinsics", {}))
    RT = np.asarray([np.asarray(doc["extrinsics"][k], dtype=np.float64)[:3, :4] for k in names]).reshape(-1, 3, 4)
    K = np.asarray(doc["intrinsics"], dtype=np.float64) if "intrinsics" in doc else None
    return names, RT, K


def write_points_ply(path, X, colors=None):
    """Point cloud as the reference's ``estimated_points.ply`` (src/pipeline.py:158-159): binary
    little-endian, x y z float32 and red green blue alpha uint8 per vertex.  ``colors``: (n,3) or
    (n,4) uint8; black, opaque when omitted."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, np.asarray(X).shape[-1])[:, :3]
    n = X.shape[0]
    rec = np.zeros(n, dtype=_VERTEX)
    rec["xyz"] = X.astype(np.float32)
    rec["rgba"][:, 3] = 255
    if colors is not None and n:
        c = np.asarray(colors, dtype=np.uint8).reshape(n, -1)
        rec["rgba"][:, : c.shape[1]] = c[:, :4]
    with open(path, "wb") as fp:
        fp.write(PLY_HEADER.format(n=n).encode("ascii"))
        fp.write(rec.tobytes())


def read_points_ply(path):
    """Vertices (n,3) float64 and colours (n,4) uint8 of a binary little-endian PLY with the vertex
    layout above (the layout of the reference's committed ``estimated_points.ply`` files)."""
    with open(path, "rb") as fp:
        n, props, fmt = None, [], None
        while True:
            line = fp.readline()
            if not line:
                raise ValueError("PLY header does not end")
            line = line.decode("ascii").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property"):
                props.append(tuple(line.split()[1:]))
            elif line == "end_header":
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError("only binary little-endian PLY point clouds are supported")
        if props[:7] != [("float", "x"), ("float", "y"), ("float", "z"), ("uchar", "red"), ("uchar", "green"),
                         ("uchar", "blue"), ("uchar", "alpha")]:
            raise ValueError("unexpected PLY vertex layout: %r" % (props,))
        rec = np.frombuffer(fp.read(_VERTEX.itemsize * n), dtype=_VERTEX, count=n)
    return rec["xyz"].astype(np.float64), rec["rgba"].copy()


# ---- data/stage1/eval.py metrics -------------------------------------------------------------------
def chamfer_distance(A, B):
    """Mean nearest-neighbour distance A->B plus B->A (eval.py:7-15)."""
    from scipy.spatial import cKDTree

    return float(np.mean(cKDTree(B).query(A)[0]) + np.mean(cKDTree(A).query(B)[0]))


def crop_points_to_bbox(points, bb):
    """Points inside the axis-aligned box bb = (min, max), eval.py:18-27."""
    points = np.asarray(points)
    mask = np.logical_and(np.all(points >= bb[0], axis=1), np.all(points <= bb[1], axis=1))
    return points[mask]


def compute_transformation_error(t1, t2):
    """Rotation angle between and translation distance of two [R|T] (eval.py:30-46)."""
    eps = 1e-6
    r1, r2 = t1[:3, :3], t2[:3, :3]
    rot = np.arccos(np.clip((np.trace(r1 @ r2.T) - 1) / 2, -1.0 + eps, 1.0 - eps))
    return rot, np.linalg.norm(t1[:3, 3] - t2[:3, 3])


def pose_estimate(d1, d2, scale):
    """Mean rotation / translation error over the cameras of two extrinsics dictionaries, the second
    one's translations divided by ``scale`` (eval.py:49-69; float32 as there)."""
    rot = tr = 0.0
    for camera in d1.keys():
        a = np.array(d1[camera], dtype=np.float32)
        b = np.array(d2[camera], dtype=np.float32)
        b[:3, 3] /= scale
        r, t = compute_transformation_error(a, b)
        rot += r
        tr += t
    return rot / len(d1), tr / len(d1)


# ---- "bundle adjustment in the large" problems -----------------------------------------------------
def load_bal(path) -> BAProblem:
    """Text problem file: ``m n N``; N lines ``camera point x y``; 9 numbers per camera (Rodrigues
    vector, translation, focal length, two radial distortion coefficients); 3 per point.  That
    camera model images p = -P/P_z scaled by f (and a distortion polynomial), which is this code's
    pinhole model with K = diag(-f, -f, 1); the distortion coefficients are dropped.  Returns a
    problem with one K per camera, observations ordered frame-major as ``get_indices`` emits them."""
    with open(path) as fp:
        tok = fp.read().split()
    m, n, N = int(tok[0]), int(tok[1]), int(tok[2])
    obs = np.asarray(tok[3: 3 + 4 * N], dtype=np.float64).reshape(N, 4)
    cams = np.asarray(tok[3 + 4 * N: 3 + 4 * N + 9 * m], dtype=np.float64).reshape(m, 9)
    pts = np.asarray(tok[3 + 4 * N + 9 * m: 3 + 4 * N + 9 * m + 3 * n], dtype=np.float64).reshape(n, 3)
    cam_idx, pt_idx = obs[:, 0].astype(np.int64), obs[:, 1].astype(np.int64)
    order = np.lexsort((pt_idx, cam_idx))
    K = np.zeros((m, 3, 3))
    K[:, 0, 0] = K[:, 1, 1] = -cams[:, 6]
    K[:, 2, 2] = 1.0
    p = BAProblem(K=K, R=rodrigues(cams[:, :3]), T=cams[:, 3:6].copy(), X=pts, cam_idx=cam_idx[order].astype(np.int32),
                  pt_idx=pt_idx[order].astype(np.int32), x_true=obs[order, 2:4].copy(), visibility="file")
    p.meta = dict(m=m, n=n, source=str(path), dropped_distortion=cams[:, 7:9].copy())
    return p
