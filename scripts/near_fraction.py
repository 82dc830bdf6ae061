"""Share of the points (and observations) of the golden fixtures that the plan keeps near (fused
point-major assembly) and far (camera-pair kernels):  python scripts/near_fraction.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sfm_amd  # noqa: F401,E402
from sfm_amd.device import Plan  # noqa: E402

for name in ("ba_4x16_bigrot", "ba_8x400", "ba_20x3000", "ba_6x3000_long", "real_box", "real_boot"):
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.exists(path):
        continue
    d = np.load(path)
    if "cam_idx" in d.files:
        cam, pt = d["cam_idx"], d["pt_idx"]
    else:
        cam, pt = d["viewpoint_indices"], d["point_indices"]
    pl = Plan(cam.astype(np.int32), pt.astype(np.int32))
    n_obs_far = pl.n_far_obs
    cls = "; ".join(f"{c['n_near']} points in {c['n_keys']} windows of {c['slots']} cameras" for c in pl.classes if c["n_runs"])
    print(f"{name}: {pl.m} cameras, {pl.n} points, {pl.n_obs} observations: near {pl.n_near} points "
          f"({100.0 * pl.n_near / pl.n:.1f} %), far {pl.n_far} points with {n_obs_far} observations "
          f"({100.0 * n_obs_far / pl.n_obs:.1f} %) | {cls if cls else 'no runs: camera-pair path'}")
