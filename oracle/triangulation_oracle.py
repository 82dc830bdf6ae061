"""CPU restatement of the reference's two-view triangulation.  TEST INFRASTRUCTURE ONLY: imported by
tests/, never by the product path.

Follows /root/reference/src/triangulation.py:10-32 (`triangulate_pts`) and src/utils.py:4-5
(`Vec2Skew`), batched: one numpy SVD call over all correspondences instead of the reference's
per-point loop.  Pinned by tests/golden/triangulation.npz (outputs of the reference function) and by
the two known-answer cases of the reference's own test/test_triangulation.py.
"""
import numpy as np


def skew(v):
    """(n,3) vectors -> (n,3,3) cross-product matrices, src/utils.py:4-5."""
    v = np.asarray(v, dtype=np.float64)
    z = np.zeros(v.shape[0])
    return np.stack([np.stack([z, -v[:, 2], v[:, 1]], -1), np.stack([v[:, 2], z, -v[:, 0]], -1),
                     np.stack([-v[:, 1], v[:, 0], z], -1)], -2)


def triangulate_pts(x1, x2, P1, P2):
    """src/triangulation.py:10-32: A = [[x1]_x P1; [x2]_x P2], last right singular vector, divided
    by its last entry, float32."""
    assert len(x1) == len(x2)
    if len(x1) == 0:
        return np.zeros((0, 4), dtype=np.float32)
    A = np.concatenate([skew(x1) @ np.asarray(P1, dtype=np.float64), skew(x2) @ np.asarray(P2, dtype=np.float64)], axis=1)
    _, _, vt = np.linalg.svd(A)
    X = vt[:, -1, :]
    return (X / X[:, -1:]).astype(np.float32)
