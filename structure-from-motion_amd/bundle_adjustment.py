"""Bundle-adjustment harness on ``frames`` / ``map_`` objects, computed by libsfm_ba.so.

Mirrors ``compute_reprojection_error_total``, ``compute_A_B``, ``get_indices`` and
``optimize_using_BA`` of ``/root/reference/src/bundle_adjustment.py:9-105``: same
signatures, same in-place effects on ``frames`` and ``map_``.  The two plotting calls of
the reference's loop (:71, :105) are not part of the computation and are not made.
"""
from __future__ import annotations

import numpy as np

from .device import DeviceProblem
from .jacobian import jacobian_blocks
from .sparse_ba import can_run_ba


def _pack_frames(frames):
    K = np.stack([np.asarray(f.K, dtype=np.float64) for f in frames])
    R = np.stack([np.asarray(f.R, dtype=np.float64) for f in frames])
    T = np.stack([np.asarray(f.T, dtype=np.float64).reshape(3) for f in frames])
    C = np.stack([np.asarray(f.C, dtype=np.float64).reshape(3) for f in frames])
    P = np.stack([np.asarray(f.P, dtype=np.float64) for f in frames])
    return K, R, T, C, P


def _flatten(frames):
    """Frame-major observation lists (src/bundle_adjustment.py:50-63) as arrays: viewpoint index
    and point index (int32) and the observed pixels (N, 2).  The frames' index lists are converted
    once each; nothing is built element by element."""
    for frame in frames:
        assert len(frame.triangulated_idx) == len(frame.index_kp_3d)
    counts = np.fromiter((len(f.index_kp_3d) for f in frames), dtype=np.int64, count=len(frames))
    vi = np.repeat(np.arange(len(frames), dtype=np.int32), counts)
    pi = [np.zeros(0, dtype=np.int32)]
    xt = [np.zeros((0, 2))]
    for frame in frames:
        pi.append(np.asarray(frame.index_kp_3d, dtype=np.int32).reshape(-1))
        idx = np.asarray(frame.triangulated_idx, dtype=np.intp).reshape(-1)
        xt.append(np.asarray(frame.keypoints)[idx][:, :2])
    return vi, np.concatenate(pi), np.vstack(xt)


def _problem_from(frames, map_, check=True, reorder=False):
    K, R, T, C, P = _pack_frames(frames)
    vi, pi, x_t = _flatten(frames)
    # (reorder: sequences that revisit places are kept on the device in the camera order that narrows
    # the band of the reduced system)
    prob = DeviceProblem(K, R, T, np.asarray(map_.X)[:, :3], vi, pi, x_t, n_cams=len(frames), check=check,
                         reorder=reorder)
    prob.set_state(C=C, P=P)
    return prob, vi, pi, x_t


def compute_reprojection_error_total(frames, map_):
    """Mean over frames of the Frobenius norm of the frame's residuals."""
    prob, _, _, _ = _problem_from(frames, map_, check=False)
    return prob.cost()


def get_indices(frames, map_):
    """Returns (viewpoints_indices, point_indices, x_t (N,2), x_p (N,2))."""
    K, R, T, C, P = _pack_frames(frames)
    vi, pi, x_t = _flatten(frames)
    if len(vi) == 0:
        return [], [], x_t, np.zeros((0, 2))
    _, _, x_p = jacobian_blocks(K, R, C, np.asarray(map_.X)[:, :3], vi, pi, P=P)
    return vi.tolist(), pi.tolist(), x_t, x_p  # the reference returns lists


def compute_A_B(frames, map_, viewpoints_indices, point_indices):
    """Jacobian blocks A (N,2,7) and B (N,2,3) of every observation."""
    K, R, T, C, P = _pack_frames(frames)
    A, B = jacobian_blocks(K, R, C, np.asarray(map_.X)[:, :3], viewpoints_indices, point_indices)
    return A, B


def optimize_using_BA(frames, map_, verbose=True, max_iter=50, threshold=100.0, reorder=False):
    """Iterate damped Gauss-Newton steps (mu = 0.5) while the cost exceeds ``threshold`` and
    fewer than ``max_iter`` iterations ran; updates ``frame.R/T/C/RT/P`` of every frame but
    the first and ``map_.X[:, :3]`` in place.  Returns the per-iteration costs (the
    reference returns None and prints them).  ``reorder`` (not in the reference): for sequences
    that revisit places, solve with the cameras in the order that narrows the band of the reduced
    system (``sfm_amd.ordering``); results are the same to rounding and come back per frame."""
    prob, vi, pi, _ = _problem_from(frames, map_, check=True, reorder=reorder)
    # check_args of the step (src/sparse_ba.py:39-46)
    if not can_run_ba(prob.m, prob.n, prob.N, 7, 3):
        raise ValueError("n_rows(J) must be greater than n_cols(J)")

    def report(n_iter, err):
        if verbose:
            print(f"Iteration_{n_iter}: Error: {err}")

    costs = prob.run(max_iter=max_iter, threshold=threshold, callback=report)
    R, T, C, P = prob.cameras()
    for f, frame in enumerate(frames):
        if f == 0:
            continue
        frame.R = R[f].copy()
        frame.T = T[f].reshape(3, 1).copy()
        frame.C = C[f].reshape(3, 1).copy()
        frame.RT = np.hstack((frame.R, frame.T))
        frame.P = P[f].copy()
    map_.X[:, :3] = prob.points()
    return costs
