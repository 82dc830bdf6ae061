"""Build libsfm_ba.so (gfx950) in place with hipcc.  `python build.py [--force]`.

Every source is compiled to an object of its own under `obj/` (up to four at a time), and only the
sources that are newer than their object - or older than a header - are compiled again; the objects
are then linked into the shared library next to this file."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ba_kernels.hip", "ba_fused.hip", "ba_cholesky.hip", "ba_triangulate.hip", "ba_pnp.hip", "ba_plan.cpp", "ba_plan_device.hip", "ba_capi.cpp", "ba_profile.cpp"]
HEADERS = ["ba_common.h", "ba_rows.h", "ba_bcr_wide.inc", "ba_chol_step.inc", "ba_fused_step_a.inc", "ba_fused_step_b.inc", "ba_fused_flush.inc",
           os.path.join("..", "..", "include", "sfm_ba.h")]
# SFM_BA_VARIANT=<name>: a diagnostic build beside the library (libsfm_ba_<name>.so, objects of its own),
# which _lib.py loads when the same variable is set
_VARIANT = os.environ.get("SFM_BA_VARIANT", "")
LIB = os.path.join(HERE, f"libsfm_ba_{_VARIANT}.so" if _VARIANT else "libsfm_ba.so")
OBJ = os.path.join(HERE, f"obj_{_VARIANT}" if _VARIANT else "obj")


def _flags():
    return os.environ.get("SFM_BA_EXTRA_FLAGS", "").split()  # diagnostic builds, e.g. -DSFMBA_FZ_STAMPS


def _stamp():
    """What the objects were compiled with: a change of the extra flags compiles everything again."""
    return " ".join(_flags())


def _stale_sources(force):
    newest_header = max(os.path.getmtime(os.path.join(HERE, f)) for f in HEADERS + ["build.py"])
    stamp_file = os.path.join(OBJ, "flags.txt")
    same_flags = os.path.exists(stamp_file) and open(stamp_file).read() == _stamp()
    out = []
    for s in SOURCES:
        o = os.path.join(OBJ, s + ".o")
        if force or not same_flags or not os.path.exists(o) or os.path.getmtime(o) < max(
                os.path.getmtime(os.path.join(HERE, s)), newest_header):
            out.append(s)
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    stamp_file = os.path.join(OBJ, "flags.txt")
    if os.path.exists(stamp_file) and open(stamp_file).read() != _stamp():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ["build.py"])


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    base = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-x", "hip", "-Wall", "-Wno-unused-result"] + _flags()

    def compile_one(s):
        cmd = base + ["-c", os.path.join(HERE, s), "-o", os.path.join(OBJ, s + ".o")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)

    stale = _stale_sources(force)
    with ThreadPoolExecutor(max_workers=int(os.environ.get("SFM_BA_BUILD_JOBS", "4"))) as pool:
        list(pool.map(compile_one, stale))
    with open(os.path.join(OBJ, "flags.txt"), "w") as f:
        f.write(_stamp())
    link = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950"] + [os.path.join(OBJ, s + ".o") for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
