"""Reduced solve time against the number of unknowns for two bands (100: superblocks of 128, 200: of
256) on diagonally dominant band matrices: python scripts/probes/wide_crossover.py
Run with SFM_BA_WIDE_MIN_BLOCKS=2 (reduction wherever the band fits) and with SFM_BA_WIDE_BAND=0
(general path) to see where the two cross."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import sfm_amd  # noqa: F401
from sfm_amd import _lib

lib = _lib.load()
p = _lib.ptr
rng = np.random.default_rng(0)
for hb in (100, 200):
    for n in (512, 640, 768, 896, 1024, 1280, 1536, 2048, 2560, 3072, 3584):
        r = np.arange(n)
        band = np.abs(r[:, None] - r[None, :]) <= hb
        S = np.where(band, rng.standard_normal((n, n)) * 0.01, 0.0)
        S = S + S.T + np.diag(np.full(n, 5.0))
        e = rng.standard_normal(n)
        src = torch.from_numpy(np.vstack([S, e[None, :]])).cuda()
        buf = src.clone()
        da = torch.zeros(n, dtype=torch.float64, device="cuda")
        ws = torch.empty(int(lib.sfm_ba_cholesky_workspace_bytes(n)) // 8 + 8, dtype=torch.float64, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        ts = []
        for it in range(6):
            buf.copy_(src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(lib.sfm_ba_solve_reduced_banded(p(buf), n, hb, p(da), p(ws), p(info), None))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        x = np.linalg.solve(S, e)
        err = np.linalg.norm(da.cpu().numpy() - x) / np.linalg.norm(x)
        print(f"hb {hb} n {n:5d} block {int(lib.sfm_ba_reduced_solve_block(n, hb)):4d}  {min(ts[1:]) * 1e3:7.3f} ms  err {err:.1e}",
              flush=True)
