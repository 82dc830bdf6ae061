"""CPU oracle for linear PnP - TEST INFRASTRUCTURE ONLY (imported by tests/ and the cpu_baseline leg
of bench.py; the product path never imports it).

numpy restatement of /root/reference/src/pose_estimation.py:21-144 and of the two helpers it uses,
`Vec2Skew` and `project_3D_to_2D` (/root/reference/src/utils.py:4-11).  Pinned by the reference's own
known-answer cases (/root/reference/test/test_linear_pnp.py: the `cleanup_RT_mat` matrix and the
crafted six-point pose), which tests/test_pnp.py restates as data.  The reference could not be run
for this part of the path, so beyond those cases parity is unpinned: the restatement follows the
source line by line and uses the same numpy calls (`np.linalg.svd`, `np.linalg.inv`,
`np.random.choice`), so it draws the same random samples from the same seed.
"""
import numpy as np


def project_3D_to_2D(X, P):
    """src/utils.py:8-11"""
    x = np.dot(P, np.asarray(X).T)
    return (x / x[-1]).T


def cleanup_RT_mat(R, T):
    """src/pose_estimation.py:21-30: nearest rotation U V^T with det +1, T scaled by 1/d[0]."""
    u, d, v = np.linalg.svd(R)
    det = np.round(np.linalg.det(np.dot(u, v)))
    if det == 1:
        R_cleaned = np.dot(u, v)
        T_cleaned = T / d[0]
    elif det == -1:
        R_cleaned = -np.dot(u, v)
        T_cleaned = -T / d[0]
    return R_cleaned, T_cleaned


def decompose_projection_mat(P, K):
    """src/pose_estimation.py:33-38"""
    RT = np.dot(np.linalg.inv(K), P)
    return cleanup_RT_mat(RT[:, 0:3], RT[:, -1].reshape(3, 1))


def _rows(x, X):
    """The stacked rows skew(x_i) [X_i 0 0; 0 X_i 0; 0 0 X_i] of src/pose_estimation.py:53-65 / :86-98,
    built without the per-point vstack."""
    x = np.asarray(x, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    n = x.shape[0]
    A = np.zeros((n, 3, 12))
    z = np.zeros_like(x[:, 0])
    skew = np.stack([np.stack([z, -x[:, 2], x[:, 1]], 1), np.stack([x[:, 2], z, -x[:, 0]], 1),
                     np.stack([-x[:, 1], x[:, 0], z], 1)], 1)  # (n,3,3), src/utils.py:4-5
    for b in range(3):
        A[:, :, 4 * b:4 * b + 4] = skew[:, :, b:b + 1] * X[:, None, :]
    return A.reshape(3 * n, 12)


def estimate_pose_Linear_PnP(x, X, K):
    """src/pose_estimation.py:41-70: projection matrix first, then K^-1 P."""
    assert len(x) == len(X)
    _, _, v = np.linalg.svd(_rows(x, X))
    return decompose_projection_mat(v.T[:, -1].reshape(3, 4), K)


def estimate_pose_Linear_PnP_n(x, X, K):
    """src/pose_estimation.py:73-105: normalised pixels, [R|T] directly."""
    assert len(x) == len(X)
    x_normalized = np.dot(np.linalg.inv(K), np.array(x).T).T
    _, _, v = np.linalg.svd(_rows(x_normalized, X))
    RT = v.T[:, -1].reshape(3, 4)
    return cleanup_RT_mat(RT[:, 0:3], RT[:, -1].reshape(3, 1))


def estimate_pose_Linear_PnP_RANSAC(x, X, K, threshold=2000):
    """src/pose_estimation.py:108-144: 200 minimal samples of 6 from numpy's global generator."""
    x = np.asarray(x)
    X = np.asarray(X)
    best_inliers = []
    for _ in range(200):
        sample_indices = np.random.choice(len(x), 6, replace=False)
        R, T = estimate_pose_Linear_PnP_n(x[sample_indices], X[sample_indices], K)
        P = np.dot(K, np.hstack((R, T)))
        err = np.linalg.norm(project_3D_to_2D(X, P) - x, axis=1)
        inliers = np.nonzero(err < threshold)[0].tolist()
        if len(inliers) > len(best_inliers):
            best_inliers = inliers
    R, T = estimate_pose_Linear_PnP_n(x[best_inliers], X[best_inliers], K)
    return R, T, len(best_inliers)
