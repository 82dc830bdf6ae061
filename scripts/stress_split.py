"""Random closed-loop / two-lap problems: SplitProblem and DeviceProblem(reorder=True) against the
plain DeviceProblem in acquisition order (3 iterations each).  python scripts/stress_split.py [cases]"""
import sys

sys.path.insert(0, ".")
import numpy as np

import sfm_amd  # noqa: F401
from sfm_amd import synthetic
from sfm_amd.device import DeviceProblem
from sfm_amd.split import SplitProblem

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(123)
worst = 0.0
for c in range(cases):
    vis = "loop" if c % 3 else "closures"
    kmax = int(rng.integers(3, 9))
    kmin = int(rng.integers(2, kmax + 1))
    m = int(rng.integers(4 * kmax + 2, 700))
    n = int(rng.integers(30 * m, 120 * m))
    p = synthetic.make_problem(m, n, (kmin, kmax), visibility=vis, seed=int(rng.integers(1 << 30)))
    a = DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true)
    ca = np.asarray(a.run(max_iter=3, threshold=0.0))
    out = [f"case {c}: {vis} m={m} n={n} k={kmin}..{kmax} hb {a.half_bandwidth}"]
    for name in ("split", "folded"):
        try:
            b = (SplitProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true) if name == "split" else
                 DeviceProblem(p.K, p.R, p.T, p.X, p.cam_idx, p.pt_idx, p.x_true, reorder=True))
        except ValueError as e:
            out.append(f"{name}: refused ({e})")
            continue
        cb = np.asarray(b.run(max_iter=3, threshold=0.0))
        rel = float(np.max(np.abs(ca - cb) / ca))
        dx = float(np.max(np.abs(a.points() - b.points())) / np.max(np.abs(a.points())))
        dr = float(np.max(np.abs(a.cameras()[0] - b.cameras()[0])))
        worst = max(worst, rel)
        out.append(f"{name}: band {b.half_bandwidth} cost {rel:.1e} X {dx:.1e} R {dr:.1e}")
        assert rel < 1e-8 and dx < 1e-7 and dr < 1e-7, out
    print("; ".join(out), flush=True)
print(f"worst relative cost difference {worst:.2e}")
