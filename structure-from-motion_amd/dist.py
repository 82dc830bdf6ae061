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
    """Point-id boundaries (world+1,) of contiguous ranges with near-equal observation counts.
    Every range holds at least one point (a rank without observations could not build a plan and
    its peers would wait for it in the first collective): ``ValueError`` on every rank alike when
    there are fewer points than ranks."""
    if world < 1 or n_points < world:
        raise ValueError(f"cannot shard {n_points} points over {world} ranks: every rank needs at least one point")
    counts = np.bincount(np.asarray(pt_idx, dtype=np.int64), minlength=n_points)
    csum = np.concatenate([[0], np.cumsum(counts)])
    total = csum[-1]
    targets = total * np.arange(world + 1) / world
    bounds = np.searchsorted(csum, targets, side="left").astype(np.int64)
    bounds[0], bounds[-1] = 0, n_points
    bounds = np.maximum.accumulate(bounds)
    # skewed counts can put several targets on one point: move boundaries apart, first from the
    # left (each at least one past its predecessor), then from the right (room for the successors)
    for r in range(1, world):
        bounds[r] = max(bounds[r], bounds[r - 1] + 1)
    for r in range(world - 1, 0, -1):
        bounds[r] = min(bounds[r], bounds[r + 1] - 1)
    return bounds


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


def pack_lower(S: np.ndarray, e: np.ndarray) -> np.ndarray:
    """Host statement of the dense exchange layout (``sfm_ba_lower_pack``): row i of the lower
    triangle of S at offset i (i + 1) / 2, then e."""
    n7 = S.shape[0]
    return np.concatenate([S[np.tril_indices(n7)], np.asarray(e).reshape(-1)])


def unpack_lower(buf: np.ndarray, n7: int) -> tuple[np.ndarray, np.ndarray]:
    """(S symmetric from its lower triangle, e) out of a buffer laid out by ``pack_lower``."""
    nt = n7 * (n7 + 1) // 2
    S = np.zeros((n7, n7))
    S[np.tril_indices(n7)] = buf[:nt]
    S = S + np.tril(S, -1).T
    return S, np.array(buf[nt: nt + n7])


def packed_system_elements(m: int, half_bandwidth: int | None = None) -> int:
    """fp64 elements of the per-iteration all-reduces: the camera system and the per-camera cost sums
    (m).  Dense system: the lower triangle of S (7m (7m + 1) / 2) and e (7m).  With a half bandwidth of at most 64 the ranks exchange
    the block-tridiagonal form: per 64-row block the lower triangle of the 64 x 64 diagonal block,
    the entries of the 64 x 64 coupling that the band leaves non-zero, 64 right-hand-side entries."""
    n7 = 7 * m
    if half_bandwidth is not None and half_bandwidth <= 64:
        np_ = (n7 + 63) // 64
        w = half_bandwidth  # 64 - (64 - hb): coupling entries S(k1 + r, k0 + c) with c >= r + 64 - hb
        return np_ * (64 * 65 // 2 + w * (w + 1) // 2 + 64) + m
    return n7 * (n7 + 1) // 2 + n7 + m
