"""Entry point the incremental pipeline calls after registering a frame.

Mirrors ``optimize_pose_and_map`` of ``/root/reference/src/optimize.py:9-20`` without its two
blocking 3D-viewer calls.
"""
from __future__ import annotations

from .bundle_adjustment import compute_reprojection_error_total, optimize_using_BA


def optimize_pose_and_map(frames, map_, verbose=True):
    avg_error = compute_reprojection_error_total(frames, map_)
    if verbose:
        print("Raw Reprojection Error:", avg_error)
    optimize_using_BA(frames, map_, verbose=verbose)
    avg_error = compute_reprojection_error_total(frames, map_)
    if verbose:
        print("Optimized Reprojection Error:", avg_error)
    return avg_error
