"""Host visibility-plan builder checked without a GPU: tests/native/plan_check.cpp compiles the
builder with device allocation replaced by host memory and verifies the invariants the kernels
rely on (pair slots <-> entries bijection, work items tile the entry list and stay inside their
camera-pair block, partial results of split blocks contiguous and complete, valid padding)."""
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
    (20, 2000, 6, 0, 1, 0),   # uniform visibility: every camera pair is a block, no runs
    (20, 2000, 6, 1, 1, 0),   # banded around a moving centre: first cameras alternate, no runs
    (20, 2000, 6, 2, 1, 1),   # first camera ascends: runs of the fused assembly in windows of 6 cameras
    (60, 30000, 8, 2, 5, 1),  # windows of 8 cameras
    (90, 30000, 7, 2, 6, 1),  # windows of 7
    (12, 5000, 3, 2, 7, 1),   # two first-cameras per window of 4
    (200, 100000, 9, 1, 2),   # split blocks
    (3, 50, 3, 0, 3),
    (40, 400, 30, 0, 4),      # many views per point
])
def test_plan_invariants(checker, args):
    out = subprocess.run([checker] + [str(a) for a in args], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
