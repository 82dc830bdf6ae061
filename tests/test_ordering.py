"""Host logic of the camera ordering (sfm_amd/ordering.py): band of S for a given order, the
breadth-first and spectral orders on sequences that close a loop."""
import numpy as np
import pytest

import sfm_amd  # noqa: F401
from sfm_amd import ordering, synthetic


def test_half_bandwidth_by_hand():
    # point 0: cameras 0, 1; point 1: cameras 1, 4; point 2: camera 3 only
    cam = np.array([0, 1, 1, 3, 4])
    pt = np.array([0, 0, 1, 2, 1])
    assert ordering.half_bandwidth(np.arange(5), cam, pt, 3) == 7 * 3 + 6
    # cameras 1 and 4 made neighbours
    pos = np.array([0, 1, 3, 4, 2])
    assert ordering.half_bandwidth(pos, cam, pt, 3) == 7 * 1 + 6
    # a point nobody observes does not count; no observations at all
    assert ordering.half_bandwidth(np.arange(5), cam, pt, 5) == 7 * 3 + 6
    assert ordering.half_bandwidth(np.arange(5), cam[:0], pt[:0], 3) == 0


@pytest.mark.parametrize("m,n,k", [(60, 4000, 5), (200, 20000, (6, 7))])
def test_loop_is_folded(m, n, k):
    p = synthetic.make_problem(m, n, k, visibility="loop", seed=2)
    kmax = k if np.isscalar(k) else k[1]
    natural = ordering.half_bandwidth(np.arange(m), p.cam_idx, p.pt_idx, p.n)
    assert natural == 7 * (m - 1) + 6  # first and last camera share points
    pos, order = ordering.breadth_first_positions(p.cam_idx, p.pt_idx, m, p.n)
    assert pos[0] == 0 and order[0] == 0
    assert np.array_equal(np.sort(order), np.arange(m)) and np.array_equal(pos[order], np.arange(m))
    hb, best_pos, best_order = ordering.narrowest_order(p.cam_idx, p.pt_idx, m, p.n)
    assert best_pos is not None and np.array_equal(best_pos[best_order], np.arange(m))
    assert hb == ordering.half_bandwidth(best_pos, p.cam_idx, p.pt_idx, p.n)
    # the folded ring: two cameras per place, so about twice the open sequence's band
    assert hb <= 7 * (3 * kmax) + 6
    cam2, pt2, xt2, perm = ordering.reorder_observations(best_pos, p.cam_idx, p.pt_idx, p.x_true)
    assert np.all(np.diff(cam2) >= 0)
    assert np.array_equal(cam2, best_pos[p.cam_idx[perm]]) and np.array_equal(pt2, p.pt_idx[perm])
    assert np.array_equal(xt2, p.x_true[perm])
    same_cam = np.diff(cam2) == 0
    assert np.all(np.diff(pt2)[same_cam] > 0)


def test_band_keeps_its_order():
    p = synthetic.make_problem(50, 3000, 5, visibility="banded", seed=3)
    hb, pos, order = ordering.narrowest_order(p.cam_idx, p.pt_idx, p.m, p.n)
    assert pos is None and hb == ordering.half_bandwidth(np.arange(p.m), p.cam_idx, p.pt_idx, p.n)


def test_disconnected_graph():
    # two sequences that share nothing: breadth-first order lists them one after the other,
    # the spectral order does not apply
    a = synthetic.make_problem(10, 200, 3, visibility="banded", seed=4)
    cam = np.concatenate([a.cam_idx, a.cam_idx + 10])
    pt = np.concatenate([a.pt_idx, a.pt_idx + 200])
    pos, order = ordering.breadth_first_positions(cam, pt, 20, 400)
    assert np.array_equal(np.sort(order), np.arange(20)) and pos[0] == 0
    assert set(order[:10]) == set(range(10))
    assert ordering.spectral_positions(cam, pt, 20, 400) is None


def test_loop_visibility():
    p = synthetic.make_problem(40, 3000, (4, 6), visibility="loop", seed=5)
    key = p.pt_idx.astype(np.int64) * p.m + p.cam_idx
    assert np.unique(key).size == p.n_obs and np.all(np.diff(p.cam_idx) >= 0)
    k = np.bincount(p.pt_idx, minlength=p.n)
    assert k.min() >= 4 and k.max() <= 6
    with pytest.raises(ValueError):
        synthetic.make_problem(12, 100, 4, visibility="loop")
