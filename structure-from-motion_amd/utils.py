"""Projection and quaternion helpers of the BA path, computed by libsfm_ba.so.

Mirrors ``project_3D_to_2D``, ``quaternion_to_rotation_matrix`` and
``rotation_matrix_to_quaternion`` of ``/root/reference/src/utils.py:8-11, 25-59, 62-105``
(quaternion order [x, y, z, w], as the code there unpacks it).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .device import F64, I32, _dev, _stream


def project_3D_to_2D(X, P):
    """X (k,4) homogeneous points, P (3,4): returns (k,3) rows [x, y, 1]."""
    X = np.asarray(X, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    k = X.shape[0]
    out = np.ones((k, 3))
    if k == 0:
        return out
    lib, d, p = _lib.load(), _dev(), _lib.ptr
    # homogeneous scale of the input is honoured by dividing first (last column is 1 in the pipeline)
    Xe = torch.from_numpy(np.ascontiguousarray(X[:, :3] / X[:, 3:4])).to(d)
    Pd = torch.from_numpy(P.reshape(1, 12).copy()).to(d)
    blk = torch.zeros((1, _lib.CAM_BLK), dtype=F64, device=d)
    cam = torch.zeros((k,), dtype=I32, device=d)
    pt = torch.arange(k, dtype=I32, device=d)
    xp = torch.empty((k, 2), dtype=F64, device=d)
    _lib.check(lib.sfm_ba_jacobian(p(blk), p(Pd), p(Xe), p(cam), p(pt), k, None, None, p(xp), _stream()))
    out[:, :2] = xp.cpu().numpy()
    return out


def rotation_matrix_to_quaternion(R):
    """(3,3) rotation -> [x, y, z, w]; AssertionError unless det(R) is close to 1."""
    R = np.asarray(R, dtype=np.float64)
    assert R.shape == (3, 3), "Input matrix must be 3x3"
    return rotation_matrices_to_quaternions(R[None])[0]


def rotation_matrices_to_quaternions(R):
    """Batched form: (m,3,3) -> (m,4)."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 9)
    m = R.shape[0]
    lib, d, p = _lib.load(), _dev(), _lib.ptr
    Rd = torch.from_numpy(R.copy()).to(d)
    Cd = torch.zeros((m, 3), dtype=F64, device=d)
    Kd = torch.eye(3, dtype=F64, device=d).reshape(1, 9)
    blk = torch.empty((m, _lib.CAM_BLK), dtype=F64, device=d)
    flags = torch.zeros((m,), dtype=I32, device=d)
    _lib.check(lib.sfm_ba_camera_prepare(p(Rd), p(Cd), p(Kd), 0, m, p(blk), p(flags), _stream()))
    assert not bool((flags & _lib.FLAG_DET_NOT_ONE).any().item()), \
        "Input matrix must have determinant 1 (be a rotation matrix)"
    return blk[:, 48:52].cpu().numpy()


def quaternion_to_rotation_matrix(q):
    """Unit quaternion [x, y, z, w] -> (3,3); AssertionError unless |q| is close to 1."""
    q = np.asarray(q, dtype=np.float64)
    assert len(q) == 4, "Input quaternion must have 4 elements"
    assert np.isclose(np.linalg.norm(q), 1.0), "Input quaternion must be unit quaternion"
    lib, d, p = _lib.load(), _dev(), _lib.ptr
    # the camera-update kernel converts R -> Q, adds q_step * delta, normalises and converts back:
    # start from the identity (Q = [0,0,0,1]) with delta = q - [0,0,0,1]
    Rd = torch.eye(3, dtype=F64, device=d).reshape(1, 9).clone()
    Td = torch.zeros((1, 3), dtype=F64, device=d)
    Cd = torch.zeros((1, 3), dtype=F64, device=d)
    Kd = torch.eye(3, dtype=F64, device=d).reshape(1, 9)
    da = torch.zeros((1, 7), dtype=F64, device=d)
    da[0, :4] = torch.from_numpy(q - np.array([0.0, 0.0, 0.0, 1.0])).to(d)
    flags = torch.zeros((1,), dtype=I32, device=d)
    _lib.check(lib.sfm_ba_update_cameras(p(Rd), p(Td), p(Cd), None, p(Kd), 0, p(da), 1, 0, 1.0, 0.0,
                                         p(flags), _stream()))
    return Rd.cpu().numpy().reshape(3, 3)
