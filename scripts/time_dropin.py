"""Wall time of the drop-in entry point on frames / map objects: python scripts/time_dropin.py [workload]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sfm_amd
from sfm_amd import synthetic, bundle_adjustment as ba
from sfm_amd.frame import Frame, Map

wl = sys.argv[1] if len(sys.argv) > 1 else "ladybug50"
p = synthetic.make_workload(wl, visibility="banded", seed=0)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for rep in range(2):
    frames, map_ = synthetic.to_frames_and_map(p, Frame, Map)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    costs = ba.optimize_using_BA(frames, map_, verbose=False, max_iter=10, threshold=0.0)
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"{wl} rep {rep}: optimize_using_BA, {len(costs)} iterations, {p.n_obs} observations: {dt*1e3:.1f} ms")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:3000])
