"""Repeatability of the dense solve: python scripts/solve_repeat.py n reps - every repetition must give the same
bits and no pivot flag."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import _lib
n = int(sys.argv[1]); reps = int(sys.argv[2])
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
G = torch.randn((n, n + 64), dtype=torch.float64, device="cuda", generator=g)
S0 = G @ G.T + 0.5 * torch.eye(n, dtype=torch.float64, device="cuda")
e = torch.randn((n,), dtype=torch.float64, device="cuda", generator=g)
buf = torch.empty((n + 1, n), dtype=torch.float64, device="cuda")
da = torch.empty((n,), dtype=torch.float64, device="cuda")
ws = torch.empty((int(lib.sfm_ba_cholesky_workspace_bytes(n)) // 8 + 8,), dtype=torch.float64, device="cuda")
info = torch.zeros((1,), dtype=torch.int32, device="cuda")
p = _lib.ptr
first = None
bad = 0
for i in range(reps):
    buf[:n].copy_(S0); buf[n].copy_(e); info.zero_()
    # the workspace is scratch: whatever it holds on entry must not matter
    if i > 0: ws.fill_(float('nan'))
    _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
    torch.cuda.synchronize()
    if first is None:
        first = da.clone(); L0 = buf.clone()
    same = bool((da == first).all().item())
    if int(info.item()) != 0 or not same:
        bad += 1
        diff = (buf != L0) & ~(torch.isnan(buf) & torch.isnan(L0))
        dt = diff[:n].reshape(-1)  # tiles that differ, by (column tile, row tile)
        rr, cc = diff[:n].nonzero(as_tuple=True)
        tiles = sorted(set(zip((cc // 64).tolist(), (rr // 64).tolist())))
        print("   differing tiles (col tile, row tile), first 12 of %d:" % len(tiles), tiles[:12], " NaNs in the factor:", int(torch.isnan(torch.tril(buf[:n])).sum().item()))
        t0 = tiles[0]
        sub = diff[t0[1] * 64:t0[1] * 64 + 64, t0[0] * 64:t0[0] * 64 + 64]
        print("   first differing tile: rows with a difference", sub.any(1).nonzero().flatten().tolist()[:8], "cols", sub.any(0).nonzero().flatten().tolist()[:8], "count", int(sub.sum().item()))
        rows = diff.any(1).nonzero().flatten()
        print("rep", i, "info", int(info.item()), "same bits", same, "first differing row of the factor", int(rows[0].item()) if len(rows) else None,
              "cols", diff[rows[0]].nonzero().flatten()[:4].tolist() if len(rows) else None)
res = float((S0 @ first - e).norm() / e.norm())
print(f"n={n}: {reps} repetitions, {bad} differ or carry a pivot flag; residual of the first {res:.2e}")
