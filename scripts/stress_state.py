"""Random banded problems: the loop with the rows of A formed in the kernels against the camera-pair
path (two iterations each): python scripts/stress_state.py [n_cases]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import sfm_amd
from sfm_amd import synthetic
from sfm_amd.device import DeviceProblem

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for case in range(n_cases):
    m = int(rng.integers(9, 90))
    n = int(rng.integers(300, 30000))
    klo = int(rng.integers(2, 8))
    khi = int(rng.integers(klo, 9))
    p = synthetic.make_problem(m, n, (klo, khi) if khi > klo else klo, visibility="banded", seed=int(rng.integers(1 << 30)))
    a = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, state=True)
    b = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, state=False)
    if not a.plan.fused:
        print(f"case {case}: m={m} n={n} k=({klo},{khi}) not fused, skipped")
        continue
    assert a.from_state
    b.plan.set_fused(0)
    errs = []
    for it in range(2):
        a.iterate_async(); b.iterate_async()
        ca, cb = float(a.cost_buf.item()), float(b.cost_buf.item())
        a.check_flags(); b.check_flags()
        errs.append(abs(ca - cb) / cb)
        errs.append(float((a.delta_b - b.delta_b).abs().max().item()) / float(b.delta_b.abs().max().item()))
        errs.append(float((a.delta_a - b.delta_a).abs().max().item()) / float(b.delta_a.abs().max().item()))
    worst = max(worst, max(errs))
    print(f"case {case}: m={m} n={n} k=({klo},{khi}) slots={a.plan.fused_slots} runs={a.plan.n_runs} max rel diff {max(errs):.2e}", flush=True)
    assert np.isfinite(max(errs)) and max(errs) < 1e-6, errs
print("worst", worst)
