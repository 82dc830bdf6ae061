import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sfm_amd
from sfm_amd import _lib
lib = _lib.load()
def run(n, hb, general):
    g = torch.Generator(device="cuda").manual_seed(0)
    G = torch.randn((n, n + 64), dtype=torch.float64, device="cuda", generator=g)
    S0 = G @ G.T
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
    for r in range(7):
        buf[:n].copy_(S0); buf[n].copy_(e)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if general: _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
        else: _lib.check(lib.sfm_ba_solve_reduced_banded(p(buf), n, hb, p(da), p(ws), p(info), None))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts[2:]) * 1e3
for n, hb in [(512, 100), (768, 100), (1024, 100), (1536, 100), (1536, 200), (2048, 200), (2560, 200), (3500, 209), (700, 100), (3500, 100)]:
    blk = int(lib.sfm_ba_reduced_solve_block(n, hb))
    print(f"n={n} hb={hb}: banded entry (block {blk}) {run(n, hb, False):.3f} ms, general {run(n, hb, True):.3f} ms")
