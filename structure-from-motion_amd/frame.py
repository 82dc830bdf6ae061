"""State containers read and written by the bundle-adjustment path.

Mirrors of the attributes of ``/root/reference/src/frame.py:6-53`` (``Frame``) and
``/root/reference/src/map.py:4-15`` (``Map``) that the BA path touches.  Any object
with these attributes works (the reference's own classes do); feature detection
and the correspondence bookkeeping of the reference's classes are not part of
this path and are not provided.
"""
from __future__ import annotations

import numpy as np


class Frame:
    """Camera frame: intrinsics K (3,3), pose R (3,3) / T (3,1) with x_cam = R X + T,
    centre C (3,1), RT (3,4), P = K RT (3,4); ``keypoints`` (k,3) float32
    homogeneous pixels; ``triangulated_idx`` positions into ``keypoints`` and
    ``index_kp_3d`` rows of ``Map.X``, same length and order."""

    def __init__(self, image=None, img_id=0, K=None, correspondence=True):
        if not correspondence:
            raise NotImplementedError("feature detection is outside the bundle-adjustment path")
        self.image = image
        self.img_id = img_id
        self.keypoints = None
        self.triangulated_idx = []
        self.index_kp_3d = []
        self.K = K
        self.R = None
        self.T = None
        self.C = None
        self.RT = None
        self.P = None
        self.correspondence = correspondence

    def compute_projection_matrix(self):
        """C = -R^T T, RT = [R|T], P = K RT (``/root/reference/src/frame.py:50-53``)."""
        self.C = -np.dot(self.R.T, self.T)
        self.RT = np.hstack((self.R, self.T))
        self.P = np.dot(self.K, self.RT)


class Map:
    """3D points, homogeneous: ``X`` (n,4) float64 with last column 1."""

    def __init__(self):
        self.X = np.zeros((0, 4))
        self.color = []

    def update_map(self, X_new, color):
        self.X = np.vstack((self.X, X_new))
        self.color.extend(color)

    def update_coordinates(self, X):
        self.X = X
