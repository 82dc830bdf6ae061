"""Diagnostic: how far apart are LU (numpy), Cholesky (device) and the exact solution of S da = e."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sfm_amd
from sfm_amd import synthetic, device
from oracle import ba_oracle as orc

vis = sys.argv[1] if len(sys.argv) > 1 else "uniform"
p = synthetic.make_workload("ladybug50", visibility=vis, seed=0)
prob = device.DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
prob.camera_prepare(); prob.jacobian(); prob.assemble(0.5)
n7 = prob.n7
S = prob.S[:n7].cpu().numpy().copy(); e = prob.S[n7].cpu().numpy().copy()
prob.solve(); torch.cuda.synchronize()
da_gpu = prob.delta_a.cpu().numpy().reshape(-1)
da_lu = np.linalg.solve(S, e)
# "exact": refine LU solution in extended precision
Sl = S.astype(np.longdouble); el = e.astype(np.longdouble)
x = da_lu.astype(np.longdouble)
for _ in range(6):
    r = el - Sl @ x
    x = x + np.linalg.solve(S, r.astype(np.float64)).astype(np.longdouble)
x = x.astype(np.float64)
nrm = np.linalg.norm(x)
print("cond", np.linalg.cond(S))
print("|gpu-exact|/|x|", np.linalg.norm(da_gpu - x) / nrm, " |lu-exact|/|x|", np.linalg.norm(da_lu - x) / nrm,
      " |gpu-lu|/|x|", np.linalg.norm(da_gpu - da_lu) / nrm)
print("max-norm versions", np.abs(da_gpu - x).max() / np.abs(x).max(), np.abs(da_lu - x).max() / np.abs(x).max())
# one round of refinement of the gpu solution in fp64
r = e - S @ da_gpu
x1 = da_gpu + np.linalg.solve(S, r)
print("gpu + 1 fp64 refinement step: ", np.linalg.norm(x1 - x) / nrm)
