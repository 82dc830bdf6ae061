"""Dense solve over sizes around the tile, look-ahead and regime boundaries (1 .. 9217 unknowns), each from a workspace
full of NaN, against an LU solve on the GPU: python scripts/solve_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import _lib
lib = _lib.load(); p = _lib.ptr
worst = 0
for n in [1,2,15,16,17,63,64,65,66,127,128,129,130,191,192,193,255,256,257,258,319,320,321,383,384,385,447,448,449,511,512,513,700,1023,1024,1025,2047,2048,2049,8959,8960,8961,9000,9001,9062,9063,9064,9065,9127,9128,9129,9216,9217]:
    g = torch.Generator(device="cuda").manual_seed(n)
    G = torch.randn((n, n + 16), dtype=torch.float64, device="cuda", generator=g)
    S0 = G @ G.T / n + 0.5 * torch.eye(n, dtype=torch.float64, device="cuda")
    e = torch.randn((n,), dtype=torch.float64, device="cuda", generator=g)
    x_ref = torch.linalg.solve(S0, e)
    buf = torch.empty((n + 1, n), dtype=torch.float64, device="cuda"); buf[:n].copy_(S0); buf[n].copy_(e)
    da = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    ws = torch.full((int(lib.sfm_ba_cholesky_workspace_bytes(n)) // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
    info = torch.zeros((1,), dtype=torch.int32, device="cuda")
    _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
    torch.cuda.synchronize()
    err = float((da - x_ref).norm() / x_ref.norm())
    worst = max(worst, err)
    if not (err < 1e-11) or int(info.item()) != 0:
        print("FAIL n", n, "err", err, "info", int(info.item()))
print("sweep done, worst relative error", worst)
