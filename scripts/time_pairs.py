"""Time of the camera-pair assembly from the state alone (no solve): python scripts/time_pairs.py [visibility] [workload]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd  # noqa: F401
from sfm_amd import _lib, synthetic
from sfm_amd.device import DeviceProblem

vis = sys.argv[1] if len(sys.argv) > 1 else "uniform"
wl = sys.argv[2] if len(sys.argv) > 2 else "c4_500x300k"
q = synthetic.make_workload(wl, visibility=vis, seed=0)
pr = DeviceProblem(q.K, q.R, q.T, q.X, q.cam_idx, q.pt_idx, q.x_true, n_cams=synthetic.WORKLOADS[wl][0])
pr.camera_prepare()
if pr.plan.fused:
    fn = lambda: pr.assemble_state(0.5, want_cost=True)  # noqa: E731
else:
    pr.jacobian(want_B=False)
    fn = lambda: pr.assemble_pairs_state(0.5, want_cost=True)  # noqa: E731
for _ in range(3):
    fn()
torch.cuda.synchronize()
_lib.profile_read()
_lib.profile_enable(True)
for _ in range(5):
    fn()
torch.cuda.synchronize()
_lib.profile_enable(False)
plan = pr.plan.far if pr.plan.far is not None else pr.plan
print(vis, wl, "items", plan.n_items, "entries", plan.n_entries, "blocks", plan.n_blocks,
      {k: round(ms / cnt * 1e3, 1) for k, (ms, cnt) in _lib.profile_read().items()})
