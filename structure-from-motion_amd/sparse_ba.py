"""Sparse Schur bundle-adjustment step, computed by libsfm_ba.so.

Mirrors ``SBA`` of ``/root/reference/src/sparse_ba.py:293-406`` (constructor checks of
``Indices``, :58-97; ``check_args``, :30-49; one damped Gauss-Newton step, :336-406) with
the same signatures, argument meaning and exceptions.  numpy in, numpy out.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .device import F64, I32, Plan, _dev, _stream


def can_run_ba(n_viewpoints, n_points, n_visible, n_pose_params, n_point_params):
    """src/sparse_ba.py:21-27."""
    return 2 * n_visible >= n_pose_params * n_viewpoints + n_point_params * n_points


class SBA(object):
    """``SBA(viewpoint_indices, point_indices, do_check_args=True)``.

    The constructor validates the visibility lists (``ValueError("Found non-unique (i, j)
    pair")``, ``AssertionError`` for missing ids or unequal lengths) and builds the
    device-side index plan once; ``compute`` may then be called repeatedly.
    """

    def __init__(self, viewpoint_indices, point_indices, do_check_args=True):
        assert len(viewpoint_indices) == len(point_indices)
        self.viewpoint_indices = np.ascontiguousarray(viewpoint_indices, dtype=np.int32)
        self.point_indices = np.ascontiguousarray(point_indices, dtype=np.int32)
        self.plan = Plan(self.viewpoint_indices, self.point_indices, check=True)
        self.do_check_args = do_check_args
        self.n_visible = int(self.viewpoint_indices.shape[0])
        self.n_viewpoints = self.plan.m
        self.n_points = self.plan.n

    def _check_args(self, x_true, x_pred, A, B, weights, mu):
        """src/sparse_ba.py:30-49."""
        assert A.shape[0] == B.shape[0] == self.n_visible
        assert x_true.shape[0] == x_pred.shape[0] == self.n_visible
        assert A.shape[1] == B.shape[1] == 2
        assert mu >= 0
        if not can_run_ba(self.n_viewpoints, self.n_points, self.n_visible, A.shape[2], B.shape[2]):
            raise ValueError("n_rows(J) must be greater than n_cols(J)")
        if weights is not None:
            assert weights.shape[1:3] == (2, 2)
            if not np.allclose(weights[:, 0, 1], weights[:, 1, 0]):
                raise ValueError("All weights must be symmetric")

    def compute(self, x_true, x_pred, A, B, weights=None, mu=0.0):
        """One Gauss-Newton update: returns (delta_a (m,7), delta_b (n,3))."""
        out = self.compute_all(x_true, x_pred, A, B, weights, mu, want=())
        return out["delta_a"], out["delta_b"]

    def compute_all(self, x_true, x_pred, A, B, weights=None, mu=0.0,
                    want=("U", "V_inv", "W", "Y", "S", "epsilon_a", "epsilon_b", "e"), solve=True):
        """``compute`` that also returns the intermediates named in ``want``.  ``solve=False`` stops
        after the assembly (no ``delta_a`` / ``delta_b``): with ``mu=0`` the reduced system of a free
        network is singular, the reference's LU then returns numbers without meaning and the
        Cholesky factorisation here raises ``LinAlgError``, but the assembled blocks are defined -
        V_inv is the pseudo-inverse for a rank-deficient V_i, as ``numpy.linalg.pinv`` gives it
        (``/root/reference/src/sparse_ba.py:201-212``)."""
        x_true = np.asarray(x_true, dtype=np.float64)
        x_pred = np.asarray(x_pred, dtype=np.float64)
        A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        if weights is not None:
            weights = np.asarray(weights, dtype=np.float64)
        if self.do_check_args:
            self._check_args(x_true, x_pred, A, B, weights, mu)
        if A.shape[2] != 7 or B.shape[2] != 3:
            raise NotImplementedError("the kernels are specialised for 7 pose and 3 point parameters")
        lib, d, p = _lib.load(), _dev(), _lib.ptr
        m, n, N = self.n_viewpoints, self.n_points, self.n_visible
        n7 = 7 * m
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)  # noqa: E731
        Ad, Bd, xt, xp = up(A.reshape(N, 14)), up(B.reshape(N, 6)), up(x_true[:, :2]), up(x_pred[:, :2])
        wd = up(weights.reshape(N, 4)) if weights is not None else None
        S = torch.zeros((n7 + 1, n7), dtype=F64, device=d)
        e = torch.empty((m, 7), dtype=F64, device=d)
        U = torch.empty((m, 49), dtype=F64, device=d)
        eps_a = torch.empty((m, 7), dtype=F64, device=d)
        V_inv = torch.empty((n, 9), dtype=F64, device=d)
        eps_b = torch.empty((n, 3), dtype=F64, device=d)
        ws = torch.empty((max(self.plan.workspace_bytes, 256) // 8,), dtype=F64, device=d)
        _lib.check(lib.sfm_ba_schur_assemble(self.plan.handle, p(Ad), p(Bd), p(xt), p(xp), p(wd), float(mu),
                                             float(mu), p(S), p(e), p(U), p(eps_a), p(V_inv), p(eps_b), None,
                                             p(ws), _stream()))
        out = {}
        if "S" in want:
            out["S"] = S[:n7].cpu().numpy()
        if "W" in want or "Y" in want:
            W = torch.empty((N, 21), dtype=F64, device=d)
            Y = torch.empty((N, 21), dtype=F64, device=d)
            pt = up(self.point_indices)
            _lib.check(lib.sfm_ba_calc_W_Y(self.plan.handle, p(Ad), p(Bd), p(wd), p(V_inv), p(pt), p(W), p(Y),
                                           _stream()))
            out["W"] = W.cpu().numpy().reshape(N, 7, 3)
            out["Y"] = Y.cpu().numpy().reshape(N, 7, 3)
        if not solve:
            return self._collect(out, want, m, n, U, V_inv, eps_a, eps_b, e)
        delta_a = torch.empty((m, 7), dtype=F64, device=d)
        delta_b = torch.empty((n, 3), dtype=F64, device=d)
        chol_ws = torch.empty((int(lib.sfm_ba_cholesky_workspace_bytes(n7)) // 8 + 8,), dtype=F64, device=d)
        info = torch.zeros((1,), dtype=I32, device=d)
        _lib.check(lib.sfm_ba_solve_reduced(p(S), n7, p(delta_a), p(chol_ws), p(info), _stream()))
        _lib.check(lib.sfm_ba_back_substitute(self.plan.handle, p(Ad), p(Bd), p(wd), p(V_inv), p(eps_b),
                                              p(delta_a), p(delta_b), None, 0.0, _stream()))
        if int(info.item()) != 0:
            raise np.linalg.LinAlgError("Singular matrix")
        out["delta_a"] = delta_a.cpu().numpy()
        out["delta_b"] = delta_b.cpu().numpy()
        return self._collect(out, want, m, n, U, V_inv, eps_a, eps_b, e)

    @staticmethod
    def _collect(out, want, m, n, U, V_inv, eps_a, eps_b, e):
        if "U" in want:
            out["U"] = U.cpu().numpy().reshape(m, 7, 7)
        if "V_inv" in want:
            out["V_inv"] = V_inv.cpu().numpy().reshape(n, 3, 3)
        if "epsilon_a" in want:
            out["epsilon_a"] = eps_a.cpu().numpy()
        if "epsilon_b" in want:
            out["epsilon_b"] = eps_b.cpu().numpy()
        if "e" in want:
            out["e"] = e.cpu().numpy()
        return out
