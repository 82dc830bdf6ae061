"""Golden vectors for the two-view triangulation, from the reference's own `triangulate_pts`.

Runs ONLY in the build container (reference at /root/reference).  Three camera pairs looking at a
point cloud, correspondences as the pipeline holds them (float32 homogeneous pixels, with and
without pixel noise), plus the two projection matrices and points of the reference's
test/test_triangulation.py.  Stores inputs and the reference's outputs as numbers.

    python tests/golden/make_triangulation.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import sfm_amd  # noqa: E402,F401
from sfm_amd import synthetic  # noqa: E402
from src.triangulation import triangulate_pts  # noqa: E402
from src.utils import project_3D_to_2D  # noqa: E402


def main():
    rng = np.random.default_rng(11)
    K = synthetic.K_DEFAULT
    data = {}
    n = 400
    X = np.concatenate([rng.uniform(-2, 2, size=(n, 3)) + np.array([0, 0, 8.0]), np.ones((n, 1))], axis=1)
    for c, (noise, baseline) in enumerate([(0.0, 0.6), (0.5, 0.6), (1.0, 0.15)]):
        R1 = synthetic.rodrigues(rng.normal(0, 0.05, size=3))
        R2 = synthetic.rodrigues(rng.normal(0, 0.05, size=3))
        T1 = rng.normal(0, 0.1, size=3)
        T2 = T1 + np.array([baseline, 0.02, -0.03])
        P1 = K @ np.hstack([R1, T1[:, None]])
        P2 = K @ np.hstack([R2, T2[:, None]])
        x1 = project_3D_to_2D(X, P1)
        x2 = project_3D_to_2D(X, P2)
        x1[:, :2] += rng.normal(0, noise, size=(n, 2))
        x2[:, :2] += rng.normal(0, noise, size=(n, 2))
        x1, x2 = x1.astype(np.float32), x2.astype(np.float32)  # keypoints are float32 (src/utils.py:22)
        data[f"c{c}_x1"], data[f"c{c}_x2"], data[f"c{c}_P1"], data[f"c{c}_P2"] = x1, x2, P1, P2
        data[f"c{c}_Xh"] = triangulate_pts(x1, x2, P1, P2)
    # the reference's crafted case (test/test_triangulation.py:6-29)
    P1 = np.array([[700.0, 320.0, 0.0, -95.0], [0.0, 240.0002, -933.3334, 173.3333], [0.0, 1.0, 0.0, -0.25]])
    P2 = np.array([[700.0, 320.0, 0.0, -255.0], [0.0, 240.0002, -933.3334, 173.3333], [0.0, 1.0, 0.0, -0.25]])
    Xc = np.array([[0, 5, 2, 1], [2, 5, 2, 1]])
    x1, x2 = project_3D_to_2D(Xc, P1), project_3D_to_2D(Xc, P2)
    data.update(crafted_P1=P1, crafted_P2=P2, crafted_X=Xc, crafted_x1=x1, crafted_x2=x2,
                crafted_Xh=triangulate_pts(x1, x2, P1, P2))
    data["n_cases"] = np.int64(3)
    path = os.path.join(HERE, "triangulation.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes; dtype", data["c0_Xh"].dtype)


if __name__ == "__main__":
    main()
