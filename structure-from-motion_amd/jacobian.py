"""Per-observation analytic Jacobians, computed by libsfm_ba.so.

Mirrors ``compute_X_jacobian_mat`` and ``compute_pose_jacobian_mat`` of
``/root/reference/src/jacobian.py:75-126, 215-220`` (one frame, one point) and offers
the batched form the BA loop uses.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .device import F64, I32, _dev, _stream


def jacobian_blocks(K, R, C, X, cam_idx, pt_idx, P=None):
    """A (N,2,7), B (N,2,3) [and x_pred (N,2) when P (m,3,4) is given] for the
    observations (cam_idx[i], pt_idx[i]).  K (3,3) or (m,3,3); R (m,3,3); C (m,3); X (n,3)."""
    lib, d, p = _lib.load(), _dev(), _lib.ptr
    R = np.asarray(R, dtype=np.float64).reshape(-1, 9)
    m = R.shape[0]
    K = np.asarray(K, dtype=np.float64)
    k_stride = 0 if K.ndim == 2 else 9
    Kd = torch.from_numpy(K.reshape(-1, 9).copy()).to(d)
    Rd = torch.from_numpy(R.copy()).to(d)
    Cd = torch.from_numpy(np.asarray(C, dtype=np.float64).reshape(-1, 3).copy()).to(d)
    Xd = torch.from_numpy(np.asarray(X, dtype=np.float64).reshape(-1, 3).copy()).to(d)
    cam = torch.from_numpy(np.ascontiguousarray(cam_idx, dtype=np.int32)).to(d)
    pt = torch.from_numpy(np.ascontiguousarray(pt_idx, dtype=np.int32)).to(d)
    N = int(cam.shape[0])
    blk = torch.empty((m, _lib.CAM_BLK), dtype=F64, device=d)
    flags = torch.zeros((m,), dtype=I32, device=d)
    _lib.check(lib.sfm_ba_camera_prepare(p(Rd), p(Cd), p(Kd), k_stride, m, p(blk), p(flags), _stream()))
    A = torch.empty((N, 2, 7), dtype=F64, device=d)
    B = torch.empty((N, 2, 3), dtype=F64, device=d)
    Pd = xp = None
    if P is not None:
        Pd = torch.from_numpy(np.asarray(P, dtype=np.float64).reshape(-1, 12).copy()).to(d)
        xp = torch.empty((N, 2), dtype=F64, device=d)
    _lib.check(lib.sfm_ba_jacobian(p(blk), p(Pd), p(Xd), p(cam), p(pt), N, p(A), p(B), p(xp), _stream()))
    # rotation_matrix_to_quaternion asserts det(R) ~ 1 for every frame it is handed (src/utils.py:74-76)
    used = torch.zeros((m,), dtype=torch.bool, device=d)
    if N:
        used[cam.long()] = True
    assert not bool(((flags & _lib.FLAG_DET_NOT_ONE) != 0)[used].any().item()), \
        "Input matrix must have determinant 1 (be a rotation matrix)"
    out = (A.cpu().numpy(), B.cpu().numpy())
    return out + (xp.cpu().numpy(),) if xp is not None else out


def compute_X_jacobian_mat(frame, X):
    """dF/dX (2,3) for one frame and one point X (3,1)."""
    _, B = jacobian_blocks(frame.K, frame.R[None], np.asarray(frame.C).reshape(1, 3),
                           np.asarray(X, dtype=np.float64).reshape(1, 3), [0], [0])
    return B[0]


def compute_pose_jacobian_mat(frame, X):
    """[dF/dQ | dF/dC] (2,7) for one frame and one point X (3,1)."""
    A, _ = jacobian_blocks(frame.K, frame.R[None], np.asarray(frame.C).reshape(1, 3),
                           np.asarray(X, dtype=np.float64).reshape(1, 3), [0], [0])
    return A[0]
