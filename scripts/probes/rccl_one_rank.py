"""One-rank RCCL group on the one GPU of the test box: the sharded flow (reduce, then damp) with the
collectives going through RCCL on device buffers, against the plain single-process flow."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sfm_amd  # noqa: E402,F401
from sfm_amd import synthetic  # noqa: E402
from sfm_amd.device import DeviceProblem  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    for vis, m in (("banded", 60), ("uniform", 14)):
        p = synthetic.make_problem(m, 6000, k=5, visibility=vis, seed=4)
        a = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
        b = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, group=dist.group.WORLD, sharded=True)
        ca = a.run(max_iter=3, threshold=0.0)
        cb = b.run(max_iter=3, threshold=0.0)
        torch.cuda.synchronize()
        dist.barrier()
        rel = np.max(np.abs(np.asarray(ca) - np.asarray(cb)) / np.asarray(ca))
        print(vis, "band exchange" if b.band_sharded else "triangle exchange", "rel cost diff", rel, flush=True)
        assert rel < 1e-10
finally:
    dist.destroy_process_group()
print("rccl one-rank ok")
