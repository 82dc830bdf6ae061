"""Host visibility-plan builder checked without a GPU: tests/native/plan_check.cpp compiles the
builder with device allocation replaced by host memory and verifies the invariants the kernels
rely on (pair slots <-> entries bijection, work items tile the entry list and stay inside their
camera-pair block, partial results of split blocks contiguous and complete, valid padding; near
points tiled by runs with every observation at the slot of its camera, far points and their plan,
the finishing kernel's entries covering every (block, window) a near point contributes to)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("plan") / "plan_check")
    src = os.path.join(ROOT, "tests", "native", "plan_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", exe,
                    "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


@pytest.mark.parametrize("args", [
    (200, 20000, 8, 0, 1, 0),       # uniform visibility: camera sets do not repeat, every point is far, no fused path
    (50, 20000, 8, 0, 1),           # uniform over fewer cameras: some points fit windows of 16 consecutive cameras
    (20, 2000, 6, 0, 1),            # few cameras, uniform: some camera sets repeat often enough for windows of their own
    (20, 2000, 6, 1, 1, 1, 100),    # banded around a moving centre
    (20, 2000, 6, 2, 1, 1, 100),    # first camera ascends: windows of 6 cameras
    (60, 30000, 8, 2, 5, 1, 100),   # windows of 8 cameras
    (90, 30000, 7, 2, 6, 1, 100),   # windows of 7
    (12, 5000, 3, 2, 7, 1, 100),    # windows of 4, several first-cameras per window
    (200, 100000, 9, 1, 2, 1, 100), # split blocks; points with 9 cameras go to the class with windows of 16
    (60, 30000, 14, 2, 5, 1, 100),  # up to 14 consecutive cameras: both classes
    (3, 50, 3, 0, 3),
    (40, 400, 30, 0, 4, 0),         # many views per point
    (90, 30000, 7, 3, 6, 1, 100),   # banded with shuffled point ids
    (200, 100000, 8, 4, 2, 1, 90),  # tracks of geometric length up to 30, ids in random order: both classes and far points
    (100, 50000, 6, 5, 3, 1, 99),   # distant camera sets shared by many points: windows of their own
])
def test_plan_invariants(checker, args):
    out = subprocess.run([checker] + [str(a) for a in args], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_closures_visibility():
    """Banded visibility with loop closures: same observation count per point as the banded mode,
    every point seen once per camera, and one point in a thousand spans about half the sequence."""
    import numpy as np

    import sfm_amd  # noqa: F401
    from sfm_amd import synthetic

    p = synthetic.make_problem(80, 6000, (4, 6), visibility="closures", seed=3)
    key = p.pt_idx.astype(np.int64) * p.m + p.cam_idx
    assert np.unique(key).size == p.n_obs
    assert np.all(np.diff(p.cam_idx) >= 0)
    order = np.lexsort((p.cam_idx, p.pt_idx))
    pt, cam = p.pt_idx[order], p.cam_idx[order]
    b = np.searchsorted(pt, np.arange(p.n + 1))
    k = np.diff(b)
    assert k.min() >= 4 and k.max() <= 6
    span = np.maximum.reduceat(cam, b[:-1]) - np.minimum.reduceat(cam, b[:-1]) + 1
    far = span > 6
    assert far.sum() == p.n // 1000
    assert np.all(span[far] >= 80 // 2 - 6)
    with pytest.raises(ValueError):
        synthetic.make_problem(12, 100, 4, visibility="closures")
