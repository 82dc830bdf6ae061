"""Host-side file formats and evaluation metrics (sfm_amd.sfm_io) against the formats of the
reference's committed outputs and the formulas of its data/stage1/eval.py."""
import json
import os

import numpy as np
import pytest

import sfm_amd  # noqa: F401
from sfm_amd import sfm_io, synthetic
from sfm_amd.frame import Frame


def test_ply_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    X = rng.normal(size=(37, 4))
    col = rng.integers(0, 256, size=(37, 3), dtype=np.uint8)
    path = tmp_path / "pts.ply"
    sfm_io.write_points_ply(path, X, col)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n")
    lines = head.decode().splitlines()
    # the vertex layout of the reference's estimated_points.ply: 16 bytes per vertex
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
    assert "element vertex 37" in lines
    assert [l for l in lines if l.startswith("property")] == [
        "property float x", "property float y", "property float z", "property uchar red", "property uchar green",
        "property uchar blue", "property uchar alpha"]
    assert len(body) == 37 * 16
    Xr, cr = sfm_io.read_points_ply(path)
    assert np.array_equal(Xr, X[:, :3].astype(np.float32).astype(np.float64))
    assert np.array_equal(cr[:, :3], col) and np.all(cr[:, 3] == 255)
    sfm_io.write_points_ply(path, np.zeros((0, 3)))
    assert sfm_io.read_points_ply(path)[0].shape == (0, 3)


def test_camera_parameters_round_trip(tmp_path):
    p = synthetic.make_problem(4, 30, k=3, seed=1)
    frames, _ = synthetic.to_frames_and_map(p, Frame, sfm_amd.frame.Map)
    names = [f"{i:05d}.jpg" for i in range(4)]
    path = tmp_path / "cams.json"
    sfm_io.write_camera_parameters(path, frames, names)
    doc = json.load(open(path))
    assert list(doc) == ["extrinsics"] and sorted(doc["extrinsics"]) == names
    got_names, RT, K = sfm_io.read_camera_parameters(path)
    assert got_names == names and K is None
    for j, fr in enumerate(frames):
        assert np.array_equal(RT[j], fr.RT)


def test_eval_metrics():
    # chamfer distance of two shifted grids: every nearest neighbour is 0.1 away
    g = np.stack(np.meshgrid(np.arange(4.0), np.arange(4.0), np.arange(4.0)), -1).reshape(-1, 3)
    assert abs(sfm_io.chamfer_distance(g, g + np.array([0.1, 0, 0])) - 0.2) < 1e-12
    bb = (np.array([0.5, 0.5, 0.5]), np.array([2.5, 2.5, 2.5]))
    assert sfm_io.crop_points_to_bbox(g, bb).shape == (8, 3)
    # rotation by 0.3 rad about z, translation offset (0, 0, 2)
    c, s = np.cos(0.3), np.sin(0.3)
    t1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    t2 = np.hstack([np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]), np.array([[0.0], [0.0], [2.0]])])
    rot, tr = sfm_io.compute_transformation_error(t1, t2)
    assert abs(rot - 0.3) < 1e-12 and abs(tr - 2.0) < 1e-12
    r, t = sfm_io.pose_estimate({"a": t1.tolist(), "b": t1.tolist()}, {"a": t2.tolist(), "b": t1.tolist()}, scale=2.0)
    # identical rotations still score arccos(1 - 1e-6): the clip of the reference's formula (eval.py:37)
    assert abs(r - (0.3 + np.arccos(1.0 - 1e-6)) / 2) < 1e-4 and abs(t - 0.5) < 1e-6


def test_bal_loader(tmp_path):
    """A two-camera, three-point file in the text layout; the loader's pinhole equivalent must
    reproduce the file's projection model p = -f P / P_z (distortion coefficients zero)."""
    rng = np.random.default_rng(5)
    m, n = 2, 3
    cams = np.concatenate([rng.normal(0, 0.05, size=(m, 3)), rng.normal(0, 0.2, size=(m, 3)), np.array([[500.0], [650.0]]),
                           np.zeros((m, 2))], axis=1)
    pts = rng.uniform(-1, 1, size=(n, 3)) + np.array([0, 0, -6.0])
    R = synthetic.rodrigues(cams[:, :3])
    lines, expect = [], {}
    for i in range(n):
        for j in (1, 0):  # deliberately not frame-major
            P = R[j] @ pts[i] + cams[j, 3:6]
            xy = -cams[j, 6] * P[:2] / P[2]
            lines.append(f"{j} {i} {float(xy[0])!r} {float(xy[1])!r}")
            expect[(j, i)] = xy
    text = f"{m} {n} {len(lines)}\n" + "\n".join(lines) + "\n" + "\n".join(repr(float(v)) for v in cams.reshape(-1)) + "\n" + \
        "\n".join(repr(float(v)) for v in pts.reshape(-1)) + "\n"
    path = tmp_path / "problem.txt"
    path.write_text(text)
    p = sfm_io.load_bal(path)
    assert p.m == m and p.n == n and p.n_obs == len(lines) and p.K.shape == (m, 3, 3)
    assert np.all(np.diff(p.cam_idx) >= 0)
    for c, i, x in zip(p.cam_idx, p.pt_idx, p.x_true):
        assert np.allclose(x, expect[(int(c), int(i))], rtol=0, atol=1e-12)
        uvw = p.K[c] @ (p.R[c] @ p.X[i] + p.T[c])
        assert np.allclose(uvw[:2] / uvw[2], x, rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_bal_problem_runs_on_device(tmp_path):
    """Per-camera intrinsics through the whole device loop: a file problem with exact observations
    has zero cost."""
    from sfm_amd.device import DeviceProblem

    rng = np.random.default_rng(9)
    m, n = 6, 200
    cams = np.concatenate([rng.normal(0, 0.03, size=(m, 3)), rng.normal(0, 0.2, size=(m, 3)),
                           rng.uniform(400, 700, size=(m, 1)), np.zeros((m, 2))], axis=1)
    pts = rng.uniform(-1, 1, size=(n, 3)) + np.array([0, 0, -6.0])
    R = synthetic.rodrigues(cams[:, :3])
    lines = []
    for j in range(m):
        for i in range(n):
            P = R[j] @ pts[i] + cams[j, 3:6]
            xy = -cams[j, 6] * P[:2] / P[2]
            lines.append(f"{j} {i} {float(xy[0])!r} {float(xy[1])!r}")
    path = tmp_path / "problem.txt"
    path.write_text(f"{m} {n} {len(lines)}\n" + "\n".join(lines) + "\n" + "\n".join(repr(float(v)) for v in cams.reshape(-1)) + "\n" +
                    "\n".join(repr(float(v)) for v in pts.reshape(-1)) + "\n")
    p = sfm_io.load_bal(path)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    assert prob.cost() < 1e-9
    prob.iterate_async()
    prob.check_flags()
    assert float(prob.cost_buf.item()) < 1e-6


@pytest.mark.parametrize("scene", ["box", "boot"])
def test_readers_against_the_references_committed_outputs(scene, tmp_path):
    """The readers on bytes of the files the reference has committed under data/stage1/<scene>/
    (tests/golden/stage1_io.npz, built by tests/golden/make_stage1_io_fixture.py: the PLY header as
    trimesh wrote it with the first 32 and the last vertex record, the first three extrinsics of
    each camera file; expected values decoded there with struct / json)."""
    import os

    from sfm_amd import sfm_io

    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage1_io.npz"))
    header = fx[f"{scene}_ply_header"].tobytes()
    n = int(fx[f"{scene}_ply_n"])
    assert f"element vertex {n}\n".encode() in header and header.startswith(b"ply\nformat binary_little_endian 1.0\n")
    # a 33-vertex file with the reference's own header text (count patched) and its own records
    small = header.replace(f"element vertex {n}\n".encode(), b"element vertex 33\n")
    path = tmp_path / "points.ply"
    path.write_bytes(small + fx[f"{scene}_ply_first32"].tobytes() + fx[f"{scene}_ply_last"].tobytes())
    xyz, rgba = sfm_io.read_points_ply(str(path))
    assert xyz.shape == (33, 3) and rgba.shape == (33, 4)
    assert np.array_equal(xyz, fx[f"{scene}_ply_xyz"]) and np.array_equal(rgba, fx[f"{scene}_ply_rgba"])
    assert np.all(xyz >= fx[f"{scene}_ply_min"]) and np.all(xyz <= fx[f"{scene}_ply_max"])
    # what write_points_ply emits for these vertices is, past the comment line, the same file
    out = tmp_path / "rewritten.ply"
    sfm_io.write_points_ply(str(out), xyz, rgba)
    mine = out.read_bytes()
    strip = lambda b: b"\n".join(ln for ln in b.split(b"end_header\n")[0].split(b"\n") if not ln.startswith(b"comment"))  # noqa: E731
    assert strip(mine) == strip(small)
    assert mine.split(b"end_header\n", 1)[1] == path.read_bytes().split(b"end_header\n", 1)[1]

    for kind in ("estimated", "gt"):
        if f"{scene}_{kind}_json" not in fx:
            continue
        jp = tmp_path / f"{kind}.json"
        jp.write_bytes(fx[f"{scene}_{kind}_json"].tobytes())
        names, RT, K = sfm_io.read_camera_parameters(str(jp))
        assert names == list(fx[f"{scene}_{kind}_names"])
        assert RT.shape == (3, 3, 4) and np.array_equal(RT, fx[f"{scene}_{kind}_RT"])
        if f"{scene}_{kind}_K" in fx:
            assert np.array_equal(K, fx[f"{scene}_{kind}_K"])
        else:
            assert K is None
