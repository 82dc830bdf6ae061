"""Camera ordering for the reduced camera system.

S has a non-zero 7 x 7 block (j, k) exactly when cameras j and k share a point, so the order of
the cameras decides its band.  An incremental sequence gives a narrow band in acquisition order;
a sequence that returns to a place it has seen (a loop closure) adds blocks far from the diagonal
in that order, but few cameras apart along the folded loop.  The breadth-first order of the
co-visibility graph from camera 0 is that folded order; camera 0 keeps position 0, which is the
camera the loop holds fixed (``/root/reference/src/bundle_adjustment.py:84``: updates start at
frame 1).  The reference solves S with a dense LU in whatever order the frames come
(``/root/reference/src/sparse_ba.py:274-276``); the solution does not depend on the order beyond
rounding.
"""
from __future__ import annotations

import numpy as np


def half_bandwidth(cam_pos, cam_idx, pt_idx, n_points: int) -> int:
    """Half bandwidth of S (in unknowns) when camera j sits at position ``cam_pos[j]``: 7 times the
    widest span of positions among the cameras of one point, plus 6."""
    pos = np.asarray(cam_pos, dtype=np.int64)[np.asarray(cam_idx, dtype=np.int64)]
    pt = np.asarray(pt_idx, dtype=np.int64)
    lo = np.full(n_points, np.iinfo(np.int64).max, dtype=np.int64)
    hi = np.full(n_points, -1, dtype=np.int64)
    order = np.argsort(pos, kind="stable")
    # positions ascending: the last write per point is the largest, and in reverse the smallest
    hi[pt[order]] = pos[order]
    lo[pt[order[::-1]]] = pos[order[::-1]]
    seen = hi >= 0
    if not seen.any():
        return 0
    return int(7 * np.max(hi[seen] - lo[seen]) + 6)


def _scipy():
    """scipy.sparse, needed only by the camera orders of this module (``reorder=True``): a clear error
    where it is missing, instead of an ImportError from the middle of a run."""
    try:
        import scipy.sparse  # noqa: F401
        import scipy.sparse.csgraph  # noqa: F401
        import scipy.sparse.linalg  # noqa: F401
    except ImportError as exc:  # pragma: no cover - scipy is part of the image
        raise ImportError("camera reordering (reorder=True, ordering.narrowest_order) needs scipy: "
                          "scipy.sparse, scipy.sparse.csgraph, scipy.sparse.linalg.eigsh") from exc


def covisibility_graph(cam_idx, pt_idx, n_cams: int, n_points: int):
    """Sparse (n_cams, n_cams) matrix with a non-zero (j, k) where cameras j and k share a point
    (the block pattern of S, diagonal included)."""
    _scipy()
    from scipy.sparse import csr_matrix

    cam = np.asarray(cam_idx, dtype=np.int64)
    pt = np.asarray(pt_idx, dtype=np.int64)
    V = csr_matrix((np.ones(cam.shape[0], dtype=np.float32), (pt, cam)), shape=(n_points, n_cams))
    return (V.T @ V).tocsr()


def breadth_first_positions(cam_idx, pt_idx, n_cams: int, n_points: int, graph=None):
    """(position of every camera, camera at every position) for the breadth-first order of the
    co-visibility graph from camera 0; cameras that no chain of shared points links to camera 0
    follow, component by component."""
    from scipy.sparse.csgraph import breadth_first_order

    G = covisibility_graph(cam_idx, pt_idx, n_cams, n_points) if graph is None else graph
    seen = np.zeros(n_cams, dtype=bool)
    parts = []
    for root in range(n_cams):
        if seen[root]:
            continue
        o = breadth_first_order(G, root, directed=False, return_predecessors=False)
        seen[o] = True
        parts.append(np.asarray(o, dtype=np.int64))
    order = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
    pos = np.empty(n_cams, dtype=np.int64)
    pos[order] = np.arange(n_cams)
    return pos, order


def reorder_observations(cam_pos, cam_idx, pt_idx, x_true):
    """Observation lists with the cameras renumbered by ``cam_pos`` and put back in frame-major
    order (points ascending inside a frame).  Returns (cam, pt, x_true, order) with ``order`` the
    index of every new observation in the old lists."""
    new_cam = np.asarray(cam_pos, dtype=np.int64)[np.asarray(cam_idx, dtype=np.int64)]
    order = np.lexsort((np.asarray(pt_idx), new_cam))
    return (new_cam[order].astype(np.int32), np.asarray(pt_idx)[order].astype(np.int32),
            np.asarray(x_true)[order], order)


def spectral_positions(cam_idx, pt_idx, n_cams: int, n_points: int, graph=None):
    """(position of every camera, camera at every position) by the second eigenvector of the
    Laplacian of the co-visibility graph (unit weights): along a sequence that folds back on
    itself the eigenvector varies with the place, not with the time of acquisition, so sorting
    by it interleaves the passes through the same place.  None when the graph is not connected,
    when it is so dense that no order gives a band (more than 128 neighbours per camera on
    average), or when the eigen-solver does not converge."""
    _scipy()
    from scipy.sparse import diags
    from scipy.sparse.csgraph import connected_components
    from scipy.sparse.linalg import ArpackError, ArpackNoConvergence, eigsh

    if n_cams < 3:
        return None
    G = (covisibility_graph(cam_idx, pt_idx, n_cams, n_points) if graph is None else graph).copy()
    G.data[:] = 1.0
    G.setdiag(0.0)
    G.eliminate_zeros()
    if G.nnz > 128 * n_cams:
        return None  # cameras share points with most other cameras: no band in any order
    if connected_components(G, directed=False, return_labels=False) != 1:
        return None
    deg = np.asarray(G.sum(axis=1)).reshape(-1)
    L = (diags(deg) - G).astype(np.float64).tocsc()
    try:
        # shift-invert just below zero: the two smallest eigenvalues (0 and the one wanted)
        vals, vecs = eigsh(L, k=2, sigma=-1e-3, which="LM", v0=np.linspace(-1.0, 1.0, n_cams))
    except (ArpackNoConvergence, ArpackError, RuntimeError):
        # (no convergence, or a singular shifted matrix: the breadth-first order remains)
        return None
    f = vecs[:, np.argmax(vals)]
    order = np.argsort(f, kind="stable").astype(np.int64)
    pos = np.empty(n_cams, dtype=np.int64)
    pos[order] = np.arange(n_cams)
    return pos, order


def narrowest_order(cam_idx, pt_idx, n_cams: int, n_points: int, natural_half_bandwidth=None):
    """The narrowest of the natural, breadth-first and spectral camera orders:
    (half bandwidth, position of every camera, camera at every position); positions are None when
    the natural order is the narrowest."""
    hb = natural_half_bandwidth
    if hb is None:
        hb = half_bandwidth(np.arange(n_cams), cam_idx, pt_idx, n_points)
    best = (int(hb), None, None)
    G = covisibility_graph(cam_idx, pt_idx, n_cams, n_points)
    for cand in (breadth_first_positions(cam_idx, pt_idx, n_cams, n_points, G),
                 spectral_positions(cam_idx, pt_idx, n_cams, n_points, G)):
        if cand is None:
            continue
        h = half_bandwidth(cand[0], cam_idx, pt_idx, n_points)
        if h < best[0]:
            best = (h, cand[0], cand[1])
    return best
