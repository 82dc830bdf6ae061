"""The plan built on the device equals the plan built on the host, array for array, and reports
the same errors (the checks of Indices.__init__, /root/reference/src/sparse_ba.py:58-97)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _views(rng, m, n, kmax, banded):
    cam, pt = [], []
    for i in range(n):
        k = int(rng.integers(2, kmax + 1))
        if banded:
            c0 = i * m // n
            cs = np.unique(np.clip(np.arange(c0 - k // 2, c0 - k // 2 + k), 0, m - 1))
        else:
            cs = rng.choice(m, size=min(k, m), replace=False)
        cam.extend(int(c) for c in cs)
        pt.extend([i] * len(cs))
    return np.asarray(cam, np.int32), np.asarray(pt, np.int32)


def _cover(cam, pt, m, n):
    """make every camera appear at least once (the reference requires contiguous ids)"""
    missing = np.setdiff1d(np.arange(m), cam)
    have = set(zip(cam.tolist(), pt.tolist()))
    extra_c, extra_p = [], []
    for c in missing:
        extra_c.append(c)
        extra_p.append(int(c) % n)
        assert (int(c), int(c) % n) not in have
    return np.concatenate([cam, np.asarray(extra_c, np.int32)]), np.concatenate([pt, np.asarray(extra_p, np.int32)])


CASES = [
    # m, n, kmax, banded, order
    (8, 300, 4, False, "frame"),
    (20, 3000, 6, True, "frame"),
    (20, 3000, 6, True, "shuffled"),
    (37, 5000, 9, False, "point"),
    (120, 40000, 11, True, "frame"),     # split blocks, several per-camera items
    (300, 2000, 200, False, "shuffled"),  # many views per point: small group windows
]


@pytest.mark.parametrize("m,n,kmax,banded,order", CASES)
def test_same_arrays(m, n, kmax, banded, order):
    from sfm_amd.device import Plan

    rng = np.random.default_rng(m * 1000 + n)
    cam, pt = _views(rng, m, n, kmax, banded)
    cam, pt = _cover(cam, pt, m, n)
    if order == "frame":
        o = np.lexsort((pt, cam))
    elif order == "point":
        o = np.lexsort((cam, pt))
    else:
        o = rng.permutation(cam.shape[0])
    cam, pt = np.ascontiguousarray(cam[o]), np.ascontiguousarray(pt[o])
    host = Plan(cam, pt, check=True, where="host")
    dev = Plan(cam, pt, check=True, where="device")
    for name in ("m", "n", "n_obs", "n_blocks", "n_entries", "n_pair_slots", "n_items", "workspace_bytes",
                 "half_bandwidth", "device_bytes"):
        assert getattr(host, name) == getattr(dev, name), name
    for f in Plan.FIELDS:
        a, b = host.read(f), dev.read(f)
        assert a.shape == b.shape, f
        assert np.array_equal(a, b), f
    # groups of the per-point kernels: at most 256 observations and 256 points each
    g, pp = dev.read("grp_pt"), dev.read("pt_ptr")
    assert g[0] == 0 and g[-1] == n and np.all(np.diff(g) > 0)
    assert np.max(pp[g[1:]] - pp[g[:-1]]) <= 256 and np.max(np.diff(g)) <= 256


def test_hints_and_unobserved_points():
    """a shard passes global counts: cameras and points it does not observe are allowed"""
    from sfm_amd.device import Plan

    rng = np.random.default_rng(5)
    cam, pt = _views(rng, 12, 900, 5, True)
    keep = (pt % 7 != 3) & (cam != 4)  # points and one camera without observations
    cam, pt = np.ascontiguousarray(cam[keep]), np.ascontiguousarray(pt[keep])
    host = Plan(cam, pt, n_cams_hint=14, n_pts_hint=1000, check=True, where="host")
    dev = Plan(cam, pt, n_cams_hint=14, n_pts_hint=1000, check=True, where="device")
    assert (dev.m, dev.n) == (14, 1000)
    for f in Plan.FIELDS:
        assert np.array_equal(host.read(f), dev.read(f)), f


@pytest.mark.parametrize("where", ["host", "device"])
def test_errors(where):
    from sfm_amd import _lib
    from sfm_amd.device import Plan

    cam = np.asarray([0, 0, 1, 1, 2, 2], np.int32)
    pt = np.asarray([0, 1, 0, 1, 0, 1], np.int32)
    Plan(cam, pt, where=where)
    dup_c, dup_p = np.append(cam, 1).astype(np.int32), np.append(pt, 1).astype(np.int32)
    with pytest.raises(ValueError, match="non-unique"):
        Plan(dup_c, dup_p, where=where)
    Plan(dup_c, dup_p, check=False, where=where)
    with pytest.raises(AssertionError):
        Plan(cam, np.asarray([0, 2, 0, 2, 0, 2], np.int32), where=where)  # point 1 missing
    with pytest.raises(AssertionError):
        Plan(np.asarray([0, 0, 3, 3, 2, 2], np.int32), pt, where=where)  # camera 1 missing
    with pytest.raises(_lib.SfmBaError):
        Plan(np.asarray([0, -1, 1, 1, 2, 2], np.int32), pt, where=where)
    with pytest.raises(_lib.SfmBaError):
        Plan(cam, pt, n_cams_hint=2, n_pts_hint=2, where=where)  # id beyond the stated count
    # more views of one point than a workgroup of the per-point kernels holds: a range of its own
    big_c = np.arange(300, dtype=np.int32)
    big = Plan(big_c, np.zeros(300, np.int32), where=where)
    assert list(big.read("grp_pt")) == [0, 1]
