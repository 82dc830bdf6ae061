"""Builds tests/golden/stage1_io.npz from the output files the reference has committed under
/root/reference/data/stage1/{box,boot}/ (data only: bytes of the files, no code).

Per scene: the PLY header as written by the reference's run (trimesh), the raw bytes of the first 32
vertex records and of the last one, the vertex count; from estimated_camera_parameters.json and
gt_camera_parameters.json the raw text of the first three "extrinsics" entries re-wrapped as a
document of the same shape, and the intrinsics of the ground-truth file.  The expected values are
decoded here with struct / json only, independently of sfm_amd.sfm_io, which the test then checks.

    python tests/golden/make_stage1_io_fixture.py
"""
import json
import os
import struct

import numpy as np

REF = "/root/reference/data/stage1"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage1_io.npz")

doc = {}
for scene in ("box", "boot"):
    raw = open(os.path.join(REF, scene, "estimated_points.ply"), "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end]
    n = int([ln for ln in header.decode("ascii").splitlines() if ln.startswith("element vertex")][0].split()[-1])
    body = raw[end:]
    assert len(body) == 16 * n, (len(body), n)
    first = body[: 16 * 32]
    last = body[16 * (n - 1):]
    dec = [struct.unpack("<fffBBBB", body[16 * i: 16 * i + 16]) for i in list(range(32)) + [n - 1]]
    doc[f"{scene}_ply_header"] = np.frombuffer(header, dtype=np.uint8)
    doc[f"{scene}_ply_first32"] = np.frombuffer(first, dtype=np.uint8)
    doc[f"{scene}_ply_last"] = np.frombuffer(last, dtype=np.uint8)
    doc[f"{scene}_ply_n"] = np.int64(n)
    doc[f"{scene}_ply_xyz"] = np.asarray([d[:3] for d in dec], dtype=np.float64)
    doc[f"{scene}_ply_rgba"] = np.asarray([d[3:] for d in dec], dtype=np.uint8)
    # bounding box and centroid of the whole cloud, decoded with struct
    allv = np.asarray([struct.unpack("<fff", body[16 * i: 16 * i + 12]) for i in range(n)], dtype=np.float64)
    doc[f"{scene}_ply_min"] = allv.min(0)
    doc[f"{scene}_ply_max"] = allv.max(0)
    doc[f"{scene}_ply_mean"] = allv.mean(0)
    for kind in ("estimated", "gt"):
        path = os.path.join(REF, scene, f"{kind}_camera_parameters.json")
        if not os.path.exists(path):
            continue
        j = json.load(open(path))
        names = sorted(j["extrinsics"])
        small = {"extrinsics": {k: j["extrinsics"][k] for k in names[:3]}}
        if "intrinsics" in j:
            small["intrinsics"] = j["intrinsics"]
        doc[f"{scene}_{kind}_json"] = np.frombuffer(json.dumps(small).encode("ascii"), dtype=np.uint8)
        doc[f"{scene}_{kind}_n_cameras"] = np.int64(len(names))
        doc[f"{scene}_{kind}_names"] = np.asarray(names[:3])
        doc[f"{scene}_{kind}_shape"] = np.asarray(np.asarray(j["extrinsics"][names[0]]).shape)
        doc[f"{scene}_{kind}_RT"] = np.asarray([np.asarray(j["extrinsics"][k], dtype=np.float64)[:3, :4] for k in names[:3]])
        if "intrinsics" in j:
            doc[f"{scene}_{kind}_K"] = np.asarray(j["intrinsics"], dtype=np.float64)
np.savez_compressed(OUT, **doc)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
