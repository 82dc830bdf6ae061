"""Sequences that revisit places, assembled in acquisition order and solved in the folded order.

Almost every point of such a sequence is seen by a few consecutive cameras; the few points seen on
two passes through the same place (``far`` points: cameras more than a window apart) are what
puts blocks far from the diagonal of S.  ``SplitProblem`` keeps the two kinds apart:

* the near points form a problem of their own whose plan has runs - Jacobian rows formed inside
  the fused point-major assembly, as for an open sequence;
* the far points form a second, small problem on the same cameras, assembled by the camera-pair
  kernels without damping; its blocks are added into the first problem's S;
* the sum is renumbered into the folded camera order (``sfm_ba_permute_system``), where it is a
  band, solved there, and delta_a comes back in acquisition order;
* each problem back-substitutes its own points; the cameras (shared device arrays) are updated
  once, and the two per-camera residual sums are added for the cost.

This is the point-sharded flow of ``/root/reference``'s loop (``src/bundle_adjustment.py:75-101``
over the sums of ``src/sparse_ba.py:147-258``, which are additive over points) run on one GPU with
the shards chosen by the shape of the points' camera sets.  Results equal ``DeviceProblem``'s to
rounding and come back in the caller's order.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ordering
from .device import F64, I32, DeviceProblem, _stream

MAX_WINDOW = 8  # cameras a fused window can hold


def point_spans(cam_idx, pt_idx, n_points: int):
    """Per point: (first camera, last camera) over its observations, (-1, -1) when unobserved."""
    cam = np.asarray(cam_idx, dtype=np.int64)
    pt = np.asarray(pt_idx, dtype=np.int64)
    order = np.argsort(cam, kind="stable")
    hi = np.full(n_points, -1, dtype=np.int64)
    lo = np.full(n_points, -1, dtype=np.int64)
    hi[pt[order]] = cam[order]
    lo[pt[order[::-1]]] = cam[order[::-1]]
    return lo, hi


class SplitProblem:
    """Same constructor arguments and results as ``DeviceProblem`` (one GPU, no process group).

    Raises ValueError when the split does not apply: no far points, too many of them (more than
    a tenth of the observations), a camera without near observations, or no camera order that
    brings the band down to 256."""

    def __init__(self, K, R, T, X, cam_idx, pt_idx, x_true, n_cams=None, check=True, device=None,
                 plan_where="device"):
        self.lib = _lib.load()
        R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
        T = np.asarray(T, dtype=np.float64).reshape(-1, 3)
        X = np.asarray(X, dtype=np.float64).reshape(-1, 3)
        cam_idx = np.ascontiguousarray(cam_idx, dtype=np.int32)
        pt_idx = np.ascontiguousarray(pt_idx, dtype=np.int32)
        x_true = np.ascontiguousarray(x_true, dtype=np.float64).reshape(-1, 2)
        m = int(R.shape[0]) if n_cams is None else int(n_cams)
        n = int(X.shape[0])
        self.m, self.n, self.N = m, n, int(cam_idx.shape[0])
        if check:
            # what the reference's Indices asserts, on the unsplit lists
            if self.N == 0 or np.unique(cam_idx).size != m or np.unique(pt_idx).size != n:
                raise AssertionError("index lists do not cover every camera / point")
        lo, hi = point_spans(cam_idx, pt_idx, n)
        far_pt = (hi - lo + 1) > MAX_WINDOW
        far_obs = far_pt[pt_idx]
        if not far_pt.any():
            raise ValueError("no point is seen by cameras more than a window apart: nothing to split")
        if far_obs.sum() * 10 > self.N:
            raise ValueError("more than a tenth of the observations belong to far points")
        if np.unique(cam_idx[~far_obs]).size != m:
            raise ValueError("a camera has no near observation")
        hb_nat = ordering.half_bandwidth(np.arange(m), cam_idx, pt_idx, n)
        hb_new, pos, order = ordering.narrowest_order(cam_idx, pt_idx, m, n, hb_nat)
        if pos is None or hb_new > 256:
            raise ValueError("no camera order brings the band of the reduced system down to 256")
        self.natural_half_bandwidth, self.half_bandwidth = int(hb_nat), int(hb_new)
        self.cam_pos, self.cam_order = pos, order

        self.near_ids = np.flatnonzero(~far_pt)
        self.far_ids = np.flatnonzero(far_pt)
        renum = np.empty(n, dtype=np.int32)
        renum[self.near_ids] = np.arange(self.near_ids.size, dtype=np.int32)
        renum[self.far_ids] = np.arange(self.far_ids.size, dtype=np.int32)
        near = DeviceProblem(K, R, T, X[self.near_ids], cam_idx[~far_obs], renum[pt_idx[~far_obs]], x_true[~far_obs],
                             n_cams=m, check=True, device=device, plan_where=plan_where)
        if not (near.from_state and near.plan.fused):
            raise ValueError("the near points do not give a plan with runs")
        # far points: size hints, since they need not touch every camera
        far = DeviceProblem(K, R, T, X[self.far_ids], cam_idx[far_obs], renum[pt_idx[far_obs]], x_true[far_obs],
                            n_cams=m, check=False, device=device, plan_where=plan_where, state=False)
        # one set of cameras, one delta_a, one set of status words
        for name in ("K", "R", "T", "C", "P", "cam_blk", "flags", "upd_flags", "delta_a", "info"):
            setattr(far, name, getattr(near, name))
        self.near, self.far = near, far
        self.device = near.device
        d = self.device
        n7 = 7 * m
        self.n7 = n7
        # entries of the far problem's blocks (lower triangle of S, block (j, k) with j >= k)
        from scipy.sparse import csr_matrix

        fc, fp = cam_idx[far_obs].astype(np.int64), renum[pt_idx[far_obs]].astype(np.int64)
        V = csr_matrix((np.ones(fc.size, dtype=np.float32), (fp, fc)), shape=(self.far_ids.size, m))
        G = (V.T @ V).tocoo()
        keep = G.row >= G.col
        bj, bk = G.row[keep].astype(np.int64), G.col[keep].astype(np.int64)
        a = np.arange(7, dtype=np.int64)
        rows = (7 * bj[:, None, None] + a[None, :, None])
        cols = (7 * bk[:, None, None] + a[None, None, :])
        flat = (rows * n7 + cols).reshape(-1)
        self.far_blocks = int(bj.size)
        # blocks of S only the far points fill (near points never couple cameras a window apart)
        self.n_blocks = int(near.plan.n_blocks + np.count_nonzero(bj - bk >= MAX_WINDOW))
        self.far_entries = torch.from_numpy(flat).to(d)
        self.order_dev = torch.from_numpy(order.astype(np.int32)).to(d)
        self.pos_dev = torch.from_numpy(pos.astype(np.int64)).to(d)
        self.S_folded = torch.empty(((n7 + 1) * n7,), dtype=F64, device=d)
        self.da_folded = torch.zeros((m, 7), dtype=F64, device=d)
        # a band path reads whole blocks: the diagonal ones and the couplings below them, i.e. entries
        # up to two block edges from the diagonal (zeros beyond the band, but they must be there); the
        # general path wants the whole lower triangle
        self.solve_block = int(self.lib.sfm_ba_reduced_solve_block(n7, self.half_bandwidth))
        self.copy_band = 2 * self.solve_block - 1 if self.solve_block else n7
        self.cost_buf, self.status_buf = near.cost_buf, near.status_buf
        self.MU = near.MU

    # -- one Gauss-Newton iteration -------------------------------------------------------------
    def iterate_async(self, mu=None):
        near, far, lib, p = self.near, self.far, self.lib, _lib.ptr
        mu = self.MU if mu is None else mu
        if not mu > 0:
            raise ValueError("the split flow needs damping (mu > 0), as the fused assembly does")
        n7 = self.n7
        if not near._blk_fresh:
            near.camera_prepare()
        s_near = near.sys[: (n7 + 1) * n7]
        s_far = far.sys[: (n7 + 1) * n7]
        # blocks only the far points touch are not written by the near assembly: clear last iteration's
        s_near.index_fill_(0, self.far_entries, 0.0)
        near.assemble_state(mu)
        far.jacobian(want_B=False)
        _lib.check(lib.sfm_ba_schur_assemble(
            far.plan.handle, p(far.A), None, p(far.x_true), p(far.x_pred), None, float(mu), 0.0, p(far.S), p(far.e),
            None, None, p(far.V_inv), p(far.eps_b), None, p(far.workspace), _stream()))
        s_near.index_add_(0, self.far_entries, s_far.index_select(0, self.far_entries))
        near.S[n7].add_(far.S[n7])
        _lib.check(lib.sfm_ba_permute_system(p(near.S), n7, p(self.order_dev), self.copy_band, p(self.S_folded),
                                             _stream()))
        _lib.check(lib.sfm_ba_solve_reduced_banded(p(self.S_folded), n7, self.half_bandwidth, p(self.da_folded),
                                                   p(near.chol_ws), p(near.info), _stream()))
        torch.index_select(self.da_folded, 0, self.pos_dev, out=near.delta_a)
        near.back_substitute_state(update_points=True)
        far.back_substitute(update_points=True, B=False)
        near.update_cameras(prepare_next=True)
        self.cost_async()

    def cost_async(self):
        near, far, lib, p = self.near, self.far, self.lib, _lib.ptr
        for q in (near, far):
            _lib.check(lib.sfm_ba_cost_partial(q.plan.handle, p(q.P), p(q.X), p(q.cam_idx), p(q.pt_idx), p(q.x_true),
                                               p(q.ssq), p(q.workspace), _stream()))
        near.ssq.add_(far.ssq)
        _lib.check(lib.sfm_ba_cost_finish(p(near.ssq), self.m, p(near.cost_buf), _stream()))

    def set_state(self, C=None, P=None):
        """Override the derived camera centre / projection (shared by both problems)."""
        self.near.set_state(C=C, P=P)

    def cost(self) -> float:
        self.cost_async()
        return float(self.cost_buf.item())

    def status(self):
        return self.near.status()

    def clear_flags(self):
        self.near.clear_flags()

    def check_flags(self, status=None):
        self.near.check_flags(status)

    run = DeviceProblem.run

    # -- results back to the host ------------------------------------------------------------------
    def cameras(self):
        return self.near.cameras()

    def points(self):
        X = np.empty((self.n, 3), dtype=np.float64)
        X[self.near_ids] = self.near.points()
        X[self.far_ids] = self.far.points()
        return X

    @property
    def delta_a(self):
        return self.near.delta_a
