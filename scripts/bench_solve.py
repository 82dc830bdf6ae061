"""Micro-benchmark of the reduced solve on a random SPD system: python scripts/bench_solve.py [n] [reps] [half_bandwidth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sfm_amd
from sfm_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hb = int(sys.argv[3]) if len(sys.argv) > 3 else -1
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
G = torch.randn((n, n + 64), dtype=torch.float64, device="cuda", generator=g)
S0 = G @ G.T + 0.5 * torch.eye(n, dtype=torch.float64, device="cuda")
if hb >= 0:
    # band of a positive definite matrix, made diagonally dominant so that it stays positive definite
    i = torch.arange(n, device="cuda")
    S0 = torch.where((i[:, None] - i[None, :]).abs() <= hb, S0 / n, torch.zeros_like(S0))
    S0 = S0 + torch.diag(S0.abs().sum(1))
e = torch.randn((n,), dtype=torch.float64, device="cuda", generator=g)
buf = torch.empty((n + 1, n), dtype=torch.float64, device="cuda")
da = torch.empty((n,), dtype=torch.float64, device="cuda")
ws = torch.empty((int(lib.sfm_ba_cholesky_workspace_bytes(n)) // 8 + 8,), dtype=torch.float64, device="cuda")
info = torch.zeros((1,), dtype=torch.int32, device="cuda")
p = _lib.ptr
ts = []
for i in range(reps + 2):
    buf[:n].copy_(S0); buf[n].copy_(e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if hb >= 0:
        _lib.check(lib.sfm_ba_solve_reduced_banded(p(buf), n, hb, p(da), p(ws), p(info), None))
    else:
        _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts = ts[2:]
x = torch.linalg.solve(S0, e)
err = float((da - x).norm() / x.norm())
res = float((S0 @ da - e).norm() / e.norm())
fl = n ** 3 / 3
print(f"n={n} hb={hb} solve median {np.median(ts)*1e3:.3f} ms min {min(ts)*1e3:.3f} ms -> {fl/np.median(ts)/1e12:.2f} TFLOP/s; info={int(info.item())} rel diff vs torch.linalg.solve {err:.2e} residual {res:.2e}")
