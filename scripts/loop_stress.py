"""Stress of the Gauss-Newton loop: the same two iterations from the same state, many times, with the GPU left idle
before each run; every run must reproduce the first one's bits (updates, state, cost).
python scripts/loop_stress.py <workload> <visibility> <reps> [idle seconds] [points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sfm_amd
from sfm_amd import synthetic
from sfm_amd.device import DeviceProblem

name, vis, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
idle = float(sys.argv[4]) if len(sys.argv) > 4 else 0.02
pts = int(sys.argv[5]) if len(sys.argv) > 5 else None
p = synthetic.make_workload(name, visibility=vis, points=pts)
prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
state0 = [t.clone() for t in (prob.R, prob.T, prob.C, prob.P, prob.X)]
first = None
bad = 0
for i in range(reps):
    for t, s in zip((prob.R, prob.T, prob.C, prob.P, prob.X), state0):
        t.copy_(s)
    prob._blk_fresh = False
    prob._sys_fresh = False
    torch.cuda.synchronize()
    if idle > 0:
        time.sleep(idle)
    for _ in range(2):
        prob.iterate_async()
    torch.cuda.synchronize()
    out = [t.clone() for t in (prob.delta_a, prob.delta_b, prob.X, prob.R, prob.cost_buf)]
    if first is None:
        first = out
        continue
    same = [bool((a == b).all().item()) for a, b in zip(out, first)]
    if not all(same):
        bad += 1
        if bad <= 5:
            print(f"rep {i}: differs in {[n for n, s in zip(('delta_a', 'delta_b', 'X', 'R', 'cost'), same) if not s]}", flush=True)
print(f"{name}/{vis}: {reps} repetitions, {bad} differ")
