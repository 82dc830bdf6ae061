"""Time of the reduced solve for bands wider than 64: python scripts/probes/wide_band.py
(SFM_BA_WIDE_BAND=0 in the environment: the general path instead of the reduction on superblocks.)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import sfm_amd
from sfm_amd import synthetic, _lib
from sfm_amd.device import DeviceProblem

CASES = ((500, 100_000, 12), (500, 100_000, 18), (500, 100_000, 30), (500, 50_000, 60), (1800, 300_000, 12),
         (1800, 300_000, 18), (1800, 100_000, 30), (1800, 50_000, 60))
if len(sys.argv) > 1:
    CASES = [CASES[int(a)] for a in sys.argv[1:]]
for m, n, k in CASES:
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
