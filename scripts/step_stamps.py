"""Times inside launch 40 of the dense factorisation (diagnostic build with -DSFMBA_STEP_STAMPS):
SFM_BA_VARIANT=stamps python scripts/step_stamps.py [n]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
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
for i in range(3):
    buf[:n].copy_(S0); buf[n].copy_(e)
    _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
    torch.cuda.synchronize()
st = (ctypes.c_longlong * 32)()
raw = ctypes.CDLL(lib._name)
raw.sfmba_debug_step_stamps(st)
c = [st[i] - st[0] for i in range(8)]
l = [st[i] - st[16] for i in range(16, 20)]
print("chain (cycles from its start): loads+sync %d, operands in LDS %d, solve %d, P in LDS %d, update %d, parked %d, factored %d" % tuple(c[1:8]))
print("look-ahead (cycles from its start): start %d, first product done %d, before X %d, end %d" % tuple(l))
print("row-tile workgroup (cycles from its start): operands ready %d, end %d" % (st[9] - st[8], st[10] - st[8]))
print("bulk tile (1, 0): accumulators requested %d, products done %d, stored %d" % (st[21] - st[20], st[22] - st[20], st[23] - st[20]))
import numpy as np
w = (ctypes.c_longlong * 4096)()
raw.sfmba_debug_step_wall(w)
w = np.array(w[:], dtype=np.int64).reshape(2048, 2)
t0 = w[0, 0]
on = (w[:, 1] > w[:, 0]) & (w[:, 0] >= t0 - 10**7)
idx = np.nonzero(on)[0]
st_us = (w[idx, 0] - t0) / 100.0
en_us = (w[idx, 1] - t0) / 100.0
print("workgroups with both stamps: %d; chain %.1f .. %.1f us" % (len(idx), st_us[0], en_us[0]))
for lo, hi in ((1, 2), (2, 60), (60, 200), (200, 330), (330, 460)):
    m = (idx >= lo) & (idx < hi)
    if m.any():
        print("workgroups %4d..%4d: start %.1f .. %.1f us, end %.1f .. %.1f us, mean duration %.1f us" % (lo, hi - 1, st_us[m].min(), st_us[m].max(), en_us[m].min(), en_us[m].max(), (en_us[m] - st_us[m]).mean()))
hw = (ctypes.c_int * 4096)()
raw.sfmba_debug_step_hw(hw)
hw = np.array(hw[:], dtype=np.int64).reshape(2048, 2) & 0xffffffff
# HW_ID: cu_id bits 8..11, sh_id bit 12, se_id bits 13..15 (gfx9); XCC_ID register: bits 0..3
cu = (hw[:, 0] >> 8) & 0xf; sh = (hw[:, 0] >> 12) & 1; se = (hw[:, 0] >> 13) & 7; xcc = hw[:, 1] & 0xf
key = [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in idx]
from collections import defaultdict
d = defaultdict(list)
for i, kk in zip(idx, key): d[kk].append(int(i))
print("distinct (xcc, se, sh, cu): %d for %d workgroups; the chain's CU %s holds workgroups %s" % (len(d), len(idx), key[0], d[key[0]]))
print("first CUs:", [(kk, v) for kk, v in list(d.items())[:6]])
part = sorted(v for v in d.values() if len(v) > 1)
print("CUs with more than one workgroup: %d, e.g. %s" % (len(part), part[:12]))
