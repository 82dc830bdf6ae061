"""Host-side cost of handing a problem over as numpy arrays: plan build + uploads, then 10 iterations.
python scripts/time_setup.py [workload] [visibility]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sfm_amd
from sfm_amd import synthetic
from sfm_amd.device import DeviceProblem

wl = sys.argv[1] if len(sys.argv) > 1 else "c4_500x300k"
vis = sys.argv[2] if len(sys.argv) > 2 else "banded"
p = synthetic.make_workload(wl, visibility=vis, seed=0)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
from sfm_amd.device import Plan
cam_d, pt_d = torch.from_numpy(p.cam_idx).cuda(), torch.from_numpy(p.pt_idx).cuda()
for where, a, b in (("host", p.cam_idx, p.pt_idx), ("device", cam_d, pt_d)):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl = Plan(a, b, where=where)
        torch.cuda.synchronize()
        print(f"{wl}/{vis} plan built on the {where}, rep {rep}: {1e3*(time.perf_counter()-t0):.2f} ms")
        del pl
for where in ("host", "device", "device"):
  for rep in range(1):
    t0 = time.perf_counter()
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, plan_where=where)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        prob.iterate_async()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    R, T, C, P = prob.cameras(); X = prob.points()
    t3 = time.perf_counter()
    rep = where
    host_bytes = sum(a.nbytes for a in (p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true))
    print(f"{wl}/{vis} plan on the {rep}: setup (checks + plan build + uploads) {1e3*(t1-t0):.1f} ms for {host_bytes/1e6:.1f} MB of caller arrays "
          f"(+ {prob.plan.device_bytes/1e6:.1f} MB of plan), 10 iterations {1e3*(t2-t1):.2f} ms, read-back {1e3*(t3-t2):.2f} ms; "
          f"PCIe-inclusive rate {10*p.n_obs/(t3-t0):.3e} obs/s vs resident {10*p.n_obs/(t2-t1):.3e} obs/s")
    del prob
