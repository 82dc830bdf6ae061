"""Seeded synthetic bundle-adjustment problems.

The reference ships no BA fixture and its datasets' images/correspondences are
absent, so every test and benchmark problem is generated here: a cloud of points
in front of a ring of perturbed cameras, k views per point, pixel noise on the
observations and a perturbed initial state.  The observation list is emitted in
the order the reference's ``get_indices`` produces (frame-major, i.e. sorted by
camera; ``/root/reference/src/bundle_adjustment.py:44-67``), with keypoints stored
as float32 homogeneous pixels exactly as the reference's correspondence reader
does (``/root/reference/src/utils.py:14-22``).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

K_DEFAULT = np.array([[930.0, 0.0, 960.0], [0.0, 930.0, 540.0], [0.0, 0.0, 1.0]])


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """Rotation matrices (..., 3, 3) from rotation vectors (..., 3)."""
    rvec = np.asarray(rvec, dtype=np.float64)
    theta = np.linalg.norm(rvec, axis=-1, keepdims=True)
    safe = np.where(theta > 0, theta, 1.0)
    k = rvec / safe
    kx, ky, kz = k[..., 0], k[..., 1], k[..., 2]
    zero = np.zeros_like(kx)
    Kx = np.stack(
        [np.stack([zero, -kz, ky], -1), np.stack([kz, zero, -kx], -1), np.stack([-ky, kx, zero], -1)],
        -2,
    )
    th = theta[..., None]
    eye = np.broadcast_to(np.eye(3), Kx.shape)
    return eye + np.sin(th) * Kx + (1.0 - np.cos(th)) * (Kx @ Kx)


@dataclass
class BAProblem:
    """Structure-of-arrays BA problem (initial state + observations)."""

    K: np.ndarray  # (3, 3)
    R: np.ndarray  # (m, 3, 3) initial rotations
    T: np.ndarray  # (m, 3)   initial translations (x_cam = R X + T)
    X: np.ndarray  # (n, 3)   initial points
    cam_idx: np.ndarray  # (N,) int32, ascending (frame-major order)
    pt_idx: np.ndarray  # (N,) int32
    x_true: np.ndarray  # (N, 2) float64 holding float32-representable pixel values
    visibility: str = "uniform"
    meta: dict = field(default_factory=dict)

    @property
    def m(self) -> int:
        return int(self.R.shape[0])

    @property
    def n(self) -> int:
        return int(self.X.shape[0])

    @property
    def n_obs(self) -> int:
        return int(self.cam_idx.shape[0])


def _views(rng, m: int, n: int, k, visibility: str) -> tuple[np.ndarray, np.ndarray]:
    """Per-point camera sets.  Returns (pt_idx, cam_idx) of all observations, unsorted."""
    if visibility == "tracks":
        # feature tracks: a point is followed over consecutive frames for a geometrically distributed
        # length (2 + Geometric, mean k, at most 30 and at most m frames), point ids in the order of
        # the first observing frame - the order in which an incremental pipeline triangulates them
        mean = float(k if np.isscalar(k) else k[0])
        kk = np.minimum(1 + rng.geometric(1.0 / max(mean - 1.0, 1.0), size=n), min(30, m)).astype(np.int64)
    elif np.isscalar(k):
        kk = np.full(n, int(k), dtype=np.int64)
    else:
        lo, hi = int(k[0]), int(k[1])
        kk = rng.integers(lo, hi + 1, size=n)
    kmax = int(kk.max())
    if kmax > m:
        raise ValueError("views per point exceeds number of cameras")
    if visibility == "loop":
        # one lap that ends where it started: k consecutive cameras on a ring, the last cameras
        # see the points of the first ones.  In acquisition order S has blocks in its far corners;
        # ordered along the folded ring it is a band twice as wide as the open sequence's.
        if m < 4 * kmax:
            raise ValueError("a loop needs at least four times as many cameras as views per point")
        start = np.sort(rng.integers(0, m, size=n), kind="stable")
        cams = (start[:, None] + np.arange(kmax)[None, :]) % m
    elif visibility in ("banded", "closures", "tracks"):
        # k consecutive cameras from a random start (no wrap-around: S is block-banded).  Point
        # ids ascend with the first observing camera, as they do in the incremental pipeline
        # that feeds the reference's BA: new points are appended to the map when a frame is
        # registered (/root/reference/src/map.py:17-37, src/pipeline.py:121-122).
        start = rng.integers(0, m - kk + 1)
        order = np.argsort(start, kind="stable")
        start, kk = start[order], kk[order]
        offs = np.arange(kmax)[None, :]
        cams = start[:, None] + offs
        if visibility == "closures":
            # two laps over the same places: one point in a thousand keeps the
            # first half of its views and takes the other half from cameras half a sequence later,
            # so S gains blocks far from the diagonal and is no longer block-tridiagonal
            if m < 4 * kmax:
                raise ValueError("closures need at least four times as many cameras as views per point")
            # (drawn among the points of the first half of the sequence, whose cameras have a
            # counterpart m // 2 further on: the second pass sees what the first pass saw)
            early = np.flatnonzero(start + m // 2 + kmax <= m)
            far = np.zeros(n, dtype=bool)
            far[rng.choice(early, size=max(1, n // 1000), replace=False)] = True
            half = (kk + 1) // 2
            moved = far[:, None] & (offs >= half[:, None])
            cams = np.where(moved, cams + m // 2, cams)
    elif visibility == "uniform":
        # random k-subset of the m cameras per point: draw k cameras, redraw the
        # rows that hold a duplicate until none is left (k << m in every workload)
        cams = rng.integers(0, m, size=(n, kmax))
        todo = np.arange(n)
        while todo.size:
            srt = np.sort(cams[todo], axis=1)
            dup = (srt[:, 1:] == srt[:, :-1]).any(axis=1)
            todo = todo[dup]
            if todo.size:
                cams[todo] = rng.integers(0, m, size=(todo.size, kmax))
    else:
        raise ValueError(f"unknown visibility mode {visibility!r}")
    valid = np.arange(kmax)[None, :] < kk[:, None]
    pt = np.broadcast_to(np.arange(n)[:, None], cams.shape)[valid]
    return pt.astype(np.int64), cams[valid].astype(np.int64)


def make_problem(
    m: int,
    n: int,
    k=8,
    visibility: str = "uniform",
    seed: int = 0,
    pixel_noise: float = 0.5,
    point_perturb: float = 0.01,
    rot_perturb: float = 0.002,
    trans_perturb: float = 0.005,
    K: np.ndarray | None = None,
    base_rotations: np.ndarray | None = None,
    base_centres: np.ndarray | None = None,
    shuffle_ids: bool = False,
) -> BAProblem:
    """Generate a seeded problem.

    Points are uniform in [-2,2]^3 + (0,0,8); camera rotations are Rodrigues of
    N(0,0.08^2) rad and centres N(0,0.5^2), unless ``base_rotations`` /
    ``base_centres`` (m,3,3)/(m,3) are given (used to exercise every branch of the
    rotation->quaternion conversion).  Observations are exact projections plus
    N(0, pixel_noise^2) px, rounded to float32.  The initial state is the ground
    truth perturbed by N(0, point_perturb^2) on points and, for cameras >= 1, a
    left-multiplied Rodrigues N(0, rot_perturb^2) and N(0, trans_perturb^2) on T.
    ``shuffle_ids``: the point ids are a random relabelling of the order the visibility mode
    produces (the reference's ``Indices`` takes any numbering, ``/root/reference/src/sparse_ba.py:58-97``).
    """
    rng = np.random.default_rng(seed)
    K = K_DEFAULT.copy() if K is None else np.asarray(K, dtype=np.float64)

    X_gt = rng.uniform(-2.0, 2.0, size=(n, 3)) + np.array([0.0, 0.0, 8.0])
    R_small = rodrigues(rng.normal(0.0, 0.08, size=(m, 3)))
    C_small = rng.normal(0.0, 0.5, size=(m, 3))
    if base_rotations is not None:
        base_rotations = np.asarray(base_rotations, dtype=np.float64)
        R_gt = R_small @ base_rotations
    else:
        R_gt = R_small
    C_gt = C_small if base_centres is None else C_small * 0.1 + np.asarray(base_centres, dtype=np.float64)
    T_gt = -np.einsum("mij,mj->mi", R_gt, C_gt)

    pt, cam = _views(rng, m, n, k, visibility)
    if shuffle_ids:
        pt = np.random.default_rng(seed + 7919).permutation(n)[pt]
    p = observe_and_perturb(rng, K, R_gt, T_gt, X_gt, pt, cam, pixel_noise, point_perturb, rot_perturb,
                            trans_perturb)
    p.visibility = visibility
    p.meta = dict(m=m, n=n, k=k, seed=seed, pixel_noise=pixel_noise, shuffle_ids=bool(shuffle_ids))
    return p


def observe_and_perturb(rng, K, R_gt, T_gt, X_gt, pt, cam, pixel_noise=0.5, point_perturb=0.01,
                        rot_perturb=0.002, trans_perturb=0.005) -> BAProblem:
    """Observations and initial state for a given ground-truth geometry and visibility list
    (``pt[i]`` is seen by ``cam[i]``): frame-major order, exact projections plus pixel noise
    rounded to float32, ground truth perturbed as described in :func:`make_problem`."""
    m, n = R_gt.shape[0], X_gt.shape[0]
    order = np.lexsort((pt, cam))  # frame-major, points ascending inside a frame
    pt, cam = pt[order], cam[order]

    xc = np.einsum("nij,nj->ni", R_gt[cam], X_gt[pt]) + T_gt[cam]
    if not np.all(xc[:, 2] > 0.1):
        raise ValueError("synthetic geometry put a point behind a camera")
    uvw = xc @ K.T
    x_exact = uvw[:, :2] / uvw[:, 2:3]
    x_obs = x_exact + rng.normal(0.0, pixel_noise, size=x_exact.shape)
    x_true = x_obs.astype(np.float32).astype(np.float64)

    X0 = X_gt + rng.normal(0.0, point_perturb, size=X_gt.shape)
    dR = rodrigues(rng.normal(0.0, rot_perturb, size=(m, 3)))
    dT = rng.normal(0.0, trans_perturb, size=(m, 3))
    dR[0] = np.eye(3)
    dT[0] = 0.0
    R0 = dR @ R_gt
    T0 = T_gt + dT

    return BAProblem(
        K=K,
        R=R0,
        T=T0,
        X=X0,
        cam_idx=cam.astype(np.int32),
        pt_idx=pt.astype(np.int32),
        x_true=x_true,
    )


# --------------------------------------------------------------------------- #
# Named workloads of the benchmark (sizes from the build's measurement plan)
# --------------------------------------------------------------------------- #
WORKLOADS = {
    # name: (m cameras, n points, views per point)
    "ladybug50": (50, 20_000, 8),
    "c4_500x300k": (500, 300_000, (6, 7)),
    "venice1800": (1800, 1_000_000, 5),
}


def make_workload(name: str, visibility: str = "banded", seed: int = 0, points: int | None = None,
                  shuffle_ids: bool = False) -> BAProblem:
    m, n, k = WORKLOADS[name]
    if points is not None:
        n = int(points)
    if visibility == "tracks":
        k = 6  # mean track length
    p = make_problem(m, n, k, visibility=visibility, seed=seed, shuffle_ids=shuffle_ids)
    p.meta["workload"] = name
    return p


# --------------------------------------------------------------------------- #
# Frame / Map views of a problem (the reference's data model, duck-typed)
# --------------------------------------------------------------------------- #
def to_frames_and_map(problem: BAProblem, frame_cls, map_cls, extra_keypoints: int = 0, seed: int = 1):
    """Materialise ``frames`` and ``map_`` objects carrying the attributes the BA
    path reads (``/root/reference/src/frame.py:6-53``, ``/root/reference/src/map.py:4-15``).

    ``frame_cls(image, img_id, K, correspondence=True)`` and ``map_cls()`` may be
    the reference's own classes or this package's mirrors.  ``extra_keypoints``
    unobserved keypoints are mixed into every frame and ``triangulated_idx`` is
    shuffled so that the keypoint indirection is exercised.
    """
    rng = np.random.default_rng(seed)
    frames = []
    m = problem.m
    bounds = np.searchsorted(problem.cam_idx, np.arange(m + 1))
    for j in range(m):
        s, e = int(bounds[j]), int(bounds[j + 1])
        cnt = e - s
        total = cnt + extra_keypoints
        slots = rng.permutation(total)[:cnt] if extra_keypoints else np.arange(cnt)
        kp = np.ones((total, 3), dtype=np.float32)
        kp[:, :2] = rng.uniform(0, 1000, size=(total, 2)).astype(np.float32)
        kp[slots, :2] = problem.x_true[s:e].astype(np.float32)
        image = np.zeros((1080, 1920, 3), dtype=np.uint8)
        fr = frame_cls(image, j, problem.K.copy(), correspondence=True)
        fr.keypoints = kp
        fr.triangulated_idx = [int(v) for v in slots]
        fr.index_kp_3d = [int(v) for v in problem.pt_idx[s:e]]
        fr.R = problem.R[j].copy()
        fr.T = problem.T[j].reshape(3, 1).copy()
        fr.compute_projection_matrix()
        frames.append(fr)
    map_ = map_cls()
    Xh = np.ones((problem.n, 4))
    Xh[:, :3] = problem.X
    map_.X = Xh
    map_.color = [(0, 0, 0)] * problem.n
    return frames, map_
