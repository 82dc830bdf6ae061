"""Point-ID sharding of a bundle-adjustment problem over the GPUs of one node.

A rank owns a contiguous range of point ids and every observation of those points; cameras
are replicated.  Per-point quantities (V_i, its inverse, eps_b_i, the point update) stay on
the owning rank; the camera-side sums (U_j, eps_a_j, the Y W^T blocks of S and the Y eps_b
term of e - sums over points in ``/root/reference/src/sparse_ba.py:189-198, 151-160, 237-271``)
are formed per rank and added with one all-reduce of the packed system per iteration
(``DeviceProblem.assemble``); the damping mu enters U exactly once, after the reduction.
"""
from __future__ import annotations

import numpy as np

from .synthetic import BAProblem


def shard_bounds(pt_idx: np.ndarray, n_points: int, world: int) -> np.ndarray:
    """Point-id boundaries (world+1,) of contiguous ranges with near-equal observation counts."""
    counts = np.bincount(np.asarray(pt_idx, dtype=np.int64), minlength=n_points)
    csum = np.concatenate([[0], np.cumsum(counts)])
    total = csum[-1]
    targets = total * np.arange(world + 1) / world
    bounds = np.searchsorted(csum, targets, side="left")
    bounds[0], bounds[-1] = 0, n_points
    return np.maximum.accumulate(bounds).astype(np.int64)


def local_problem(problem: BAProblem, rank: int, world: int) -> tuple[BAProblem, tuple[int, int]]:
    """The shard of ``problem`` owned by ``rank``: all cameras, points [p0, p1) re-indexed from 0,
    their observations in the original (frame-major) relative order."""
    bounds = shard_bounds(problem.pt_idx, problem.n, world)
    p0, p1 = int(bounds[rank]), int(bounds[rank + 1])
    keep = (problem.pt_idx >= p0) & (problem.pt_idx < p1)
    local = BAProblem(
        K=problem.K, R=problem.R, T=problem.T, X=problem.X[p0:p1].copy(),
        cam_idx=problem.cam_idx[keep].copy(), pt_idx=(problem.pt_idx[keep] - p0).astype(np.int32),
        x_true=problem.x_true[keep].copy(), visibility=problem.visibility, meta=dict(problem.meta),
    )
    local.meta.update(rank=rank, world=world, point_range=(p0, p1))
    return local, (p0, p1)


def packed_system_elements(m: int, half_bandwidth: int | None = None) -> int:
    """fp64 elements of the per-iteration all-reduces: the camera system and the per-camera cost sums
    (m).  Dense system: S (7m x 7m) and e (7m).  With a half bandwidth of at most 64 the ranks exchange
    the block-tridiagonal form: per 64-row block the lower triangle of the 64 x 64 diagonal block,
    the entries of the 64 x 64 coupling that the band leaves non-zero, 64 right-hand-side entries."""
    n7 = 7 * m
    if half_bandwidth is not None and half_bandwidth <= 64:
        np_ = (n7 + 63) // 64
        w = half_bandwidth  # 64 - (64 - hb): coupling entries S(k1 + r, k0 + c) with c >= r + 64 - hb
        return np_ * (64 * 65 // 2 + w * (w + 1) // 2 + 64) + m
    return (n7 + 1) * n7 + m
