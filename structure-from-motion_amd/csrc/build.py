"""Build libsfm_ba.so (gfx950) in place with hipcc.  `python build.py [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ba_kernels.hip", "ba_fused.hip", "ba_cholesky.hip", "ba_triangulate.hip", "ba_pnp.hip", "ba_plan.cpp", "ba_plan_device.hip", "ba_capi.cpp", "ba_profile.cpp"]
HEADERS = ["ba_common.h", "ba_bcr_wide.inc", "ba_fused_step_a.inc", "ba_fused_step_b.inc", "ba_fused_flush.inc",
           os.path.join("..", "..", "include", "sfm_ba.h")]
LIB = os.path.join(HERE, "libsfm_ba.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ["build.py"])


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-x", "hip",
           "-Wall", "-Wno-unused-result"]
    cmd += os.environ.get("SFM_BA_EXTRA_FLAGS", "").split()  # diagnostic builds, e.g. -DSFMBA_FZ_STAMPS
    cmd += [os.path.join(HERE, s) for s in SOURCES]
    cmd += ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
