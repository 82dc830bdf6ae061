"""Stress of the dense solve: python scripts/solve_stress.py n reps - every repetition starts from a workspace full of
NaN and must reproduce the first one's bits; prints how many did not, and where the factor first differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import _lib
import time
n = int(sys.argv[1]); reps = int(sys.argv[2])
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0  # seconds of idle GPU before every solve
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
G = torch.randn((n, n + 64), dtype=torch.float64, device="cuda", generator=g)
S0 = G @ G.T + 0.5 * torch.eye(n, dtype=torch.float64, device="cuda")
e = torch.randn((n,), dtype=torch.float64, device="cuda", generator=g)
del G
buf = torch.empty((n + 1, n), dtype=torch.float64, device="cuda")
da = torch.empty((n,), dtype=torch.float64, device="cuda")
ws = torch.empty((int(lib.sfm_ba_cholesky_workspace_bytes(n)) // 8 + 8,), dtype=torch.float64, device="cuda")
info = torch.zeros((1,), dtype=torch.int32, device="cuda")
junk = torch.empty((64 * 1024 * 1024,), dtype=torch.float64, device="cuda")  # 512 MB: sweeps the caches between repetitions
p = _lib.ptr
first = None; L0 = None
bad = []
for i in range(reps):
    buf[:n].copy_(S0); buf[n].copy_(e); info.zero_(); ws.fill_(float("nan"))
    if i % 2: junk.fill_(float(i))
    if idle > 0:
        torch.cuda.synchronize()
        time.sleep(idle)
    _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
    if first is None:
        torch.cuda.synchronize()
        first = da.clone(); L0 = buf.clone()
        np_ = (n + 63) // 64
        inv0 = ws[:np_ * 4096].clone()
        continue
    same = bool((da == first).all().item())
    if not same or int(info.item()) != 0:
        diff = (buf != L0) & ~(torch.isnan(buf) & torch.isnan(L0))
        rr, cc = diff[:n].nonzero(as_tuple=True)
        tiles = sorted(set(zip((cc // 64).tolist(), (rr // 64).tolist())))
        t0 = tiles[0] if tiles else None
        msg = f"rep {i}: info {int(info.item())}, {len(tiles)} tiles differ, first (col tile, row tile) {t0}"
        if t0:
            sub = diff[t0[1] * 64:t0[1] * 64 + 64, t0[0] * 64:t0[0] * 64 + 64]
            msg += f", in it rows {sub.any(1).nonzero().flatten().tolist()[:6]}.. ({int(sub.any(1).sum())} rows) cols {sub.any(0).nonzero().flatten().tolist()[:6]}.. ({int(sub.any(0).sum())} cols)"
            msg += f", NaN in the factor {int(torch.isnan(torch.tril(buf[:n])).sum().item())}"
        inv = ws[:np_ * 4096]
        dinv = ((inv != inv0) & ~(torch.isnan(inv) & torch.isnan(inv0))).reshape(np_, 64, 64)
        tl = dinv.reshape(np_, -1).any(1).nonzero().flatten().tolist()
        msg += f"; inverse diagonal factors that differ: {tl[:5]}.. ({len(tl)})"
        if tl:
            d0 = dinv[tl[0]]
            msg += f", in the first: rows {d0.any(1).nonzero().flatten().tolist()[:5]}.. ({int(d0.any(1).sum())}) cols {d0.any(0).nonzero().flatten().tolist()[:5]}.. ({int(d0.any(0).sum())})"
            r0 = int(d0.any(1).nonzero().flatten()[0]); c0 = int(d0[r0].nonzero().flatten()[0])
            msg += f"; entry ({r0},{c0}): {float(inv.reshape(np_,64,64)[tl[0], r0, c0])!r} instead of {float(inv0.reshape(np_,64,64)[tl[0], r0, c0])!r}"
            Lt = buf[tl[0]*64:tl[0]*64+64, tl[0]*64:tl[0]*64+64]; Lt0 = L0[tl[0]*64:tl[0]*64+64, tl[0]*64:tl[0]*64+64]
            msg += f"; its factor block equal: {bool((torch.tril(Lt) == torch.tril(Lt0)).all().item())}"
        bad.append(msg)
        if len(bad) <= 6: print(msg, flush=True)
print(f"n={n}: {reps} repetitions, {len(bad)} bad")
