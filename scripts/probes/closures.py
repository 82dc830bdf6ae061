import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
import sfm_amd
from sfm_amd import synthetic
from sfm_amd.device import DeviceProblem
for name in ("c4_500x300k","venice1800"):
    p=synthetic.make_workload(name, visibility="closures", seed=0)
    d=DeviceProblem(p.K,p.R,p.T,p.X,p.cam_idx,p.pt_idx,p.x_true)
    print(name, "hb", d.half_bandwidth, "fused", d.plan.fused, flush=True)
    c=d.run(max_iter=3, threshold=0.0)
    torch.cuda.synchronize()
    t=time.perf_counter(); c=d.run(max_iter=10, threshold=0.0); torch.cuda.synchronize()
    print(name, "ms/iter", (time.perf_counter()-t)/10*1e3, c[-1], flush=True)
