"""Does replaying the iteration as a captured graph shorten it?  python scripts/probes/graph_iter.py [workload]"""
import sys, time
sys.path.insert(0, ".")
import torch
import sfm_amd
from sfm_amd import synthetic
from sfm_amd.device import DeviceProblem

w = sys.argv[1] if len(sys.argv) > 1 else "c4_500x300k"
p = synthetic.make_workload(w, visibility="banded", seed=0)
prob = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
for _ in range(3):
    prob.iterate_async()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    prob.iterate_async()
torch.cuda.synchronize()
print("eager ms/iter", (time.perf_counter() - t0) / 20 * 1e3, "cost", float(prob.cost_buf.item()))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    prob.iterate_async()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    prob.iterate_async()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph ms/iter", (time.perf_counter() - t0) / 20 * 1e3, "cost", float(prob.cost_buf.item()))
prob.check_flags()
