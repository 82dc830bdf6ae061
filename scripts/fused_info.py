import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import sfm_amd
from sfm_amd import synthetic
from sfm_amd.device import Plan
for w in ("c4_500x300k", "venice1800"):
    p = synthetic.make_workload(w, visibility="banded", seed=0)
    pl = Plan(p.cam_idx, p.pt_idx)
    rr = pl.read("rng_run"); rp = pl.read("run_pt0")
    print(w, "fused", pl.fused, "classes", pl.classes, "far points", pl.n_far, "far observations", pl.n_far_obs)
    d = np.diff(rr); print(" runs per range: mean %.2f max %d" % (d.mean(), d.max()), "points per run: mean %.1f min %d" % (np.diff(rp).mean(), np.diff(rp).min()))
    print(" hist runs/range", np.bincount(d)[:12])
