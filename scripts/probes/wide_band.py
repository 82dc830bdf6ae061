"""Time of the reduced solve for bands wider than 64 (general path): python scripts/probes/wide_band.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import sfm_amd
from sfm_amd import synthetic, _lib
from sfm_amd.device import DeviceProblem

for m, n, k in ((500, 100_000, 12), (500, 100_000, 30), (1800, 300_000, 12), (1800, 300_000, 30)):
    p = synthetic.make_problem(m, n, k, visibility="banded", seed=0)
    prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    prob.camera_prepare(); prob.jacobian(want_B=False)
    for _ in range(2):
        prob.assemble(0.5, by_products=False, B=False); prob.solve()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        prob.assemble(0.5, by_products=False, B=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prob.solve(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    n7 = 7 * m
    S = None
    prob.check_flags()
    print(f"m={m} views={k} half bandwidth {prob.half_bandwidth} of {n7}: solve {min(ts)*1e3:.3f} ms "
          f"(dense flops {n7**3/3/1e9:.1f} G, band flops {n7*(prob.half_bandwidth+1)**2/1e9:.2f} G)")
