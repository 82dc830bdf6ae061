"""Compare one GPU Gauss-Newton step against a golden fixture, stage by stage (prints relative errors)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_amd
from sfm_amd import sparse_ba
from tests.helpers import load, relerr
case = sys.argv[1] if len(sys.argv) > 1 else "ba_8x400"
d = load(case)
opt = sparse_ba.SBA(list(d["step_vi"]), list(d["step_pi"]))
out = opt.compute_all(d["step_x_t"], d["step_x_p"], d["step_A"], d["step_B"], None, 0.5)
for k, g in [("U", "step_U"), ("V_inv", "step_V_inv"), ("S", "step_S"), ("e", "step_e"), ("delta_a", "step_delta_a"), ("delta_b", "step_delta_b")]:
    print(k, relerr(out[k], d[g]))
S, Sr = out["S"], d["step_S"]
m = S.shape[0] // 7
bad = [(j, k, float(np.abs(S[7*j:7*j+7, 7*k:7*k+7] - Sr[7*j:7*j+7, 7*k:7*k+7]).max())) for j in range(m) for k in range(m)]
bad = [b for b in bad if b[2] > 1e-6 * np.abs(Sr).max()]
print("bad blocks", bad[:10], len(bad))
if bad:
    j, k, _ = bad[0]
    np.set_printoptions(precision=4, linewidth=200)
    print(S[7*j:7*j+7, 7*k:7*k+7]); print(Sr[7*j:7*j+7, 7*k:7*k+7])
