"""Shared helpers for the tests: rebuild frames / map objects from a golden fixture."""
import os

import numpy as np

import sfm_amd
from sfm_amd.frame import Frame, Map

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def frames_from_fixture(d, prefix=""):
    """Frames and map holding the fixture's initial state (prefix "" or "noisy_")."""
    K = d[prefix + "K"]
    R0, T0, X0 = d[prefix + "R0"], d[prefix + "T0"], d[prefix + "X0"]
    frames = []
    for j in range(R0.shape[0]):
        fr = Frame(np.zeros((1080, 1920, 3), dtype=np.uint8), j, K.copy(), correspondence=True)
        fr.keypoints = d[f"{prefix}kp_{j}"]
        fr.triangulated_idx = [int(v) for v in d[f"{prefix}tri_{j}"]]
        fr.index_kp_3d = [int(v) for v in d[f"{prefix}idx3d_{j}"]]
        fr.R = R0[j].copy()
        fr.T = T0[j].reshape(3, 1).copy()
        fr.compute_projection_matrix()
        frames.append(fr)
    map_ = Map()
    Xh = np.ones((X0.shape[0], 4))
    Xh[:, :3] = X0
    map_.X = Xh
    return frames, map_


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
