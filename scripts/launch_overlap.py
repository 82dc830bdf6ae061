"""Do consecutive launches of the dense factorisation overlap?  (diagnostic build with -DSFMBA_STEP_STAMPS)
SFM_BA_VARIANT=stamps python scripts/launch_overlap.py n"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_amd import _lib
n = int(sys.argv[1])
lib = _lib.load()
raw = ctypes.CDLL(lib._name)
g = torch.Generator(device="cuda").manual_seed(0)
G = torch.randn((n, n + 64), dtype=torch.float64, device="cuda", generator=g)
S0 = G @ G.T + 0.5 * torch.eye(n, dtype=torch.float64, device="cuda")
e = torch.randn((n,), dtype=torch.float64, device="cuda", generator=g)
buf = torch.empty((n + 1, n), dtype=torch.float64, device="cuda")
da = torch.empty((n,), dtype=torch.float64, device="cuda")
ws = torch.empty((int(lib.sfm_ba_cholesky_workspace_bytes(n)) // 8 + 8,), dtype=torch.float64, device="cuda")
info = torch.zeros((1,), dtype=torch.int32, device="cuda")
p = _lib.ptr
span = (ctypes.c_ulonglong * 1024)()
worst = None
for rep in range(6):
    buf[:n].copy_(S0); buf[n].copy_(e)
    torch.cuda.synchronize()
    raw.sfmba_debug_launch_span(span, 1)
    _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
    torch.cuda.synchronize()
    raw.sfmba_debug_launch_span(span, 0)
    a = np.array(span[:], dtype=np.uint64).reshape(512, 2).astype(np.int64)
    np_ = (n + 63) // 64
    st, en = a[:np_, 0], a[:np_, 1]
    gap = (st[1:] - en[:-1]) / 100.0  # us; negative: the next launch started before the last workgroup of this one ended
    k = int(np.argmin(gap))
    print(f"rep {rep}: launches {np_}, smallest gap between the last end of launch k and the first start of launch k+1: {gap.min():.2f} us at k={k}; "
          f"negative gaps: {(gap < 0).sum()}")
