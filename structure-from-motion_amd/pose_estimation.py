"""Mirror of the linear-PnP functions of the reference's ``src/pose_estimation.py`` on the GPU.

Same names, arguments and return values as ``/root/reference/src/pose_estimation.py:21-152``
(``cleanup_RT_mat``, ``decompose_projection_mat``, ``estimate_pose_Linear_PnP``,
``estimate_pose_Linear_PnP_n``, ``estimate_pose_Linear_PnP_RANSAC``,
``estimate_pose_3d_2d_mapping``).  The stacked-row SVD, the rotation clean-up and the inlier
counting run in ``libsfm_ba.so``; the 3x3 inverse of K and the random sampling stay on the host
(``np.linalg.inv`` and ``np.random.choice``, the calls the reference makes, so that a seeded run
draws the same samples).  ``estimate_pose_Essential_Matrix`` belongs to the two-view
initialisation and is not provided.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _device():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("sfm_amd needs a GPU: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _up(a, shape, dtype=np.float64):
    import torch

    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype).reshape(shape))).to(_device())


def _stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


def _check_flags(flags):
    if bool((flags != 0).any().item()):
        # cleanup_RT_mat assigns its result only when det(U V^T) rounds to +1 or -1 (:24-30)
        raise UnboundLocalError("local variable 'R_cleaned' referenced before assignment")


class Correspondences:
    """3D-2D correspondences resident on the GPU: ``x`` (n,3) homogeneous pixels, ``X`` (n,4)
    homogeneous points.  ``solve`` runs linear PnP for many subsets at once."""

    def __init__(self, x, X):
        x = np.asarray(x, dtype=np.float64)
        X = np.asarray(X, dtype=np.float64)
        assert len(x) == len(X)
        self.n = len(x)
        self.x = _up(x, (self.n, 3))
        self.X = _up(X, (self.n, 4))

    def solve(self, idx=None, Kinv_pre=None, Kinv_post=None):
        """Poses (R (h,9), T (h,3), flags (h,)) as device tensors, one per row of ``idx`` ((h,s)
        int: the correspondences each hypothesis uses); ``idx=None``: one pose from all of them."""
        import torch

        lib, p, dev = _lib.load(), _lib.ptr, _device()
        if idx is None:
            n_hyp, s, d_idx = 1, self.n, None
        else:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            n_hyp, s = idx.shape
            assert idx.min() >= 0 and idx.max() < self.n
            d_idx = torch.from_numpy(idx).to(dev)
        ws = torch.empty((int(lib.sfm_ba_pnp_workspace_bytes(n_hyp, s)) // 8 + 1,), dtype=torch.float64, device=dev)
        R = torch.empty((n_hyp, 9), dtype=torch.float64, device=dev)
        T = torch.empty((n_hyp, 3), dtype=torch.float64, device=dev)
        flags = torch.zeros((n_hyp,), dtype=torch.int32, device=dev)
        pre = _up(Kinv_pre, (9,)) if Kinv_pre is not None else None
        post = _up(Kinv_post, (9,)) if Kinv_post is not None else None
        _lib.check(lib.sfm_ba_pnp_solve(p(self.x), p(self.X), p(pre), p(post), p(d_idx), n_hyp, s, p(ws), p(R), p(T),
                                        p(flags), _stream()))
        return R, T, flags

    def inliers(self, K, R, T, threshold, want_mask=False):
        """Per pose, the number of correspondences that reproject within ``threshold`` (and the
        (h,n) byte mask when asked for)."""
        import torch

        lib, p, dev = _lib.load(), _lib.ptr, _device()
        n_hyp = R.shape[0]
        counts = torch.empty((n_hyp,), dtype=torch.int32, device=dev)
        mask = torch.empty((n_hyp, self.n), dtype=torch.uint8, device=dev) if want_mask else None
        _lib.check(lib.sfm_ba_pnp_inliers(p(self.x), p(self.X), p(_up(K, (9,))), p(R), p(T), n_hyp, self.n,
                                          float(threshold), p(counts), p(mask), _stream()))
        return counts, mask


def _cleanup(RT, Kinv_post=None):
    import torch

    lib, p, dev = _lib.load(), _lib.ptr, _device()
    d_RT = _up(RT, (1, 12))
    R = torch.empty((1, 9), dtype=torch.float64, device=dev)
    T = torch.empty((1, 3), dtype=torch.float64, device=dev)
    flags = torch.zeros((1,), dtype=torch.int32, device=dev)
    post = _up(Kinv_post, (9,)) if Kinv_post is not None else None
    _lib.check(lib.sfm_ba_pnp_cleanup(p(d_RT), p(post), 1, p(R), p(T), p(flags), _stream()))
    _check_flags(flags)
    return R[0].cpu().numpy().reshape(3, 3), T[0].cpu().numpy()


def cleanup_RT_mat(R, T):
    """Nearest rotation to ``R`` (U V^T, determinant +1) and ``T`` divided by the largest singular
    value of ``R``, both negated when det(U V^T) = -1 (``/root/reference/src/pose_estimation.py:21-30``).
    ``T`` keeps the shape it came with."""
    T = np.asarray(T, dtype=np.float64)
    RT = np.zeros((3, 4))
    RT[:, :3] = np.asarray(R, dtype=np.float64)
    RT[:, 3] = T.reshape(3)
    Rc, Tc = _cleanup(RT)
    return Rc, Tc.reshape(T.shape)


def decompose_projection_mat(P, K):
    """R, T (3,1) of ``K^-1 P`` after the clean-up (``/root/reference/src/pose_estimation.py:33-38``)."""
    Rc, Tc = _cleanup(np.asarray(P, dtype=np.float64), Kinv_post=np.linalg.inv(K))
    return Rc, Tc.reshape(3, 1)


def _single(x, X, **kw):
    R, T, flags = Correspondences(x, X).solve(**kw)
    _check_flags(flags)
    return R[0].cpu().numpy().reshape(3, 3), T[0].cpu().numpy().reshape(3, 1)


def estimate_pose_Linear_PnP(x, X, K):
    """Linear PnP through the projection matrix (``/root/reference/src/pose_estimation.py:41-70``).

    Args: x (n,3) homogeneous pixels, X (n,4) homogeneous points, K (3,3).  Returns R (3,3), T (3,1)."""
    assert len(x) == len(X)
    return _single(x, X, Kinv_post=np.linalg.inv(K))


def estimate_pose_Linear_PnP_n(x, X, K):
    """Linear PnP on normalised pixels, [R|T] directly
    (``/root/reference/src/pose_estimation.py:73-105``).  Same arguments and results."""
    assert len(x) == len(X)
    return _single(x, X, Kinv_pre=np.linalg.inv(K))


def estimate_pose_Linear_PnP_RANSAC(x, X, K, threshold=2000):
    """200 minimal samples of 6 correspondences, the pose of each, its inliers, then a refit on the
    largest inlier set (``/root/reference/src/pose_estimation.py:108-144``).  The samples are drawn
    with ``np.random.choice`` in the reference's order; all 200 poses and inlier counts are one
    launch each.  Returns R, T, number of inliers."""
    x = np.asarray(x, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    idx = np.stack([np.random.choice(len(x), 6, replace=False) for _ in range(200)])
    corr = Correspondences(x, X)
    Kinv = np.linalg.inv(K)
    R, T, flags = corr.solve(idx=idx, Kinv_pre=Kinv)
    _check_flags(flags)
    counts, _ = corr.inliers(K, R, T, threshold)
    counts = counts.cpu().numpy()
    best = int(np.argmax(counts))  # the first of the largest, as the reference's strict '>' keeps it
    if counts[best] == 0:
        best_inliers = np.zeros(0, dtype=np.int64)
    else:
        _, mask = corr.inliers(K, R[best:best + 1], T[best:best + 1], threshold, want_mask=True)
        best_inliers = np.nonzero(mask[0].cpu().numpy())[0]
    Rr, Tr = estimate_pose_Linear_PnP_n(x[best_inliers], X[best_inliers], K)
    return Rr, Tr, len(best_inliers)


def estimate_pose_3d_2d_mapping(map_, frame_curr):
    """``/root/reference/src/pose_estimation.py:147-152``"""
    idx_3d = frame_curr.index_kp_3d
    X = map_.X[idx_3d]
    x = frame_curr.keypoints[np.array(frame_curr.matched_idx)[frame_curr.intersect_idx]]
    return estimate_pose_Linear_PnP_n(x, X, frame_curr.K)
