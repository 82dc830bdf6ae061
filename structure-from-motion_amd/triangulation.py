"""Mirror of the reference's ``src/triangulation.py`` on the GPU."""
from __future__ import annotations

import numpy as np

from . import _lib


def triangulate_pts(x1, x2, P1, P2):
    """Two-view linear triangulation (``/root/reference/src/triangulation.py:10-32``).

    Args:
        x1, x2 (n,3): homogeneous pixel coordinates in the two images
        P1, P2 (3,4): projection matrices of the two frames
    Returns:
        X_h (n,4) float32: triangulated homogeneous points, last column 1 (the reference stacks its
        results into a float32 array)
    """
    import torch

    assert len(x1) == len(x2)
    n = len(x1)
    if n == 0:
        return np.zeros((0, 4), dtype=np.float32)
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("sfm_amd needs a GPU: there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    up = lambda a, shape: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))).to(dev)  # noqa: E731
    d1, d2 = up(x1, (n, 3)), up(x2, (n, 3))
    p1, p2 = up(P1, (12,)), up(P2, (12,))
    out = torch.empty((n, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    p = _lib.ptr
    _lib.check(lib.sfm_ba_triangulate(p(d1), p(d2), p(p1), p(p2), n, p(out), stream))
    return out.cpu().numpy()
