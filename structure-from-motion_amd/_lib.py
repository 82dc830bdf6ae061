"""ctypes binding of ``csrc/libsfm_ba.so`` (the C ABI declared in ``include/sfm_ba.h``).

The library is the only compute path of this package: if it is missing the import
of any compute module fails with an explicit error - there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SFM_BA_VARIANT=<name> loads a diagnostic build csrc/libsfm_ba_<name>.so made by csrc/build.py instead)
_VARIANT = os.environ.get("SFM_BA_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "csrc", f"libsfm_ba_{_VARIANT}.so" if _VARIANT else "libsfm_ba.so")

OK = 0
EINVAL, EHIP, ENONUNIQUE, EMISSING_POINT, EMISSING_VIEW, ENOTPD, ETOOBIG = -1, -2, -3, -4, -5, -6, -7
CAM_BLK = 56
CAM_RED = 43
FLAG_DET_NOT_ONE, FLAG_ZERO_QUAT, FLAG_NONFINITE = 1, 2, 4

# every symbol include/sfm_ba.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _f64, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_int
SYMBOLS = {
    "sfm_ba_version": (_int, []),
    "sfm_ba_error_string": (C.c_char_p, [_int]),
    "sfm_ba_last_hip_error": (C.c_char_p, []),
    "sfm_ba_plan_create": (_int, [_vp, _vp, _i64, _i32, _i32, _int, C.POINTER(_vp)]),
    "sfm_ba_plan_destroy": (None, [_vp]),
    "sfm_ba_plan_create_device": (_int, [_vp, _vp, _i64, _i32, _i32, _int, _vp, C.POINTER(_vp)]),
    "sfm_ba_plan_read": (_int, [_vp, _int, _vp, _i64, C.POINTER(_i64)]),
    "sfm_ba_plan_sizes": (_int, [_vp, C.POINTER(_i64)]),
    "sfm_ba_plan_workspace_bytes": (_i64, [_vp]),
    "sfm_ba_plan_fused_info": (_int, [_vp, C.POINTER(_i32)]),
    "sfm_ba_plan_set_fused": (_int, [_vp, _int]),
    "sfm_ba_camera_prepare": (_int, [_vp, _vp, _vp, _int, _i32, _vp, _vp, _vp]),
    "sfm_ba_jacobian": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "sfm_ba_schur_assemble": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_ba_plan_far": (_vp, [_vp]),
    "sfm_ba_plan_state_path": (_int, [_vp]),
    "sfm_ba_back_substitute_state": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _f64, _vp, _vp]),
    "sfm_ba_schur_assemble_pairs_state": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_ba_schur_assemble_state": (_int, [_vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_ba_calc_W_Y": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_ba_add_diagonal": (_int, [_vp, _i32, _f64, _vp]),
    "sfm_ba_iteration_status": (_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "sfm_ba_lower_elements": (_i64, [_i32]),
    "sfm_ba_lower_pack": (_int, [_vp, _i32, _vp, _vp]),
    "sfm_ba_lower_unpack": (_int, [_vp, _i32, _f64, _vp, _vp]),
    "sfm_ba_cholesky_workspace_bytes": (_i64, [_i32]),
    "sfm_ba_solve_reduced": (_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "sfm_ba_solve_reduced_banded": (_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sfm_ba_reduced_solve_block": (_i32, [_i32, _i32]),
    "sfm_ba_permute_system": (_int, [_vp, _i32, _vp, _i32, _vp, _vp]),
    "sfm_ba_plan_half_bandwidth": (_i32, [_vp]),
    "sfm_ba_triangulate": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "sfm_ba_pnp_workspace_bytes": (_i64, [_i32, _i32]),
    "sfm_ba_pnp_solve": (_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sfm_ba_pnp_cleanup": (_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "sfm_ba_pnp_inliers": (_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, C.c_double, _vp, _vp, _vp]),
    "sfm_ba_band_packed_elements": (_i64, [_i32]),
    "sfm_ba_band_pack": (_int, [_vp, _i32, _vp, _vp]),
    "sfm_ba_band_compact_elements": (_i64, [_i32, _i32]),
    "sfm_ba_band_compact": (_int, [_vp, _i32, _i32, _vp, _vp]),
    "sfm_ba_band_expand": (_int, [_vp, _i32, _i32, _vp, _vp]),
    "sfm_ba_solve_band_packed": (_int, [_vp, _i32, _f64, _vp, _vp, _vp, _vp]),
    "sfm_ba_back_substitute": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp]),
    "sfm_ba_update_cameras": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _i32, _i32, _f64, _f64, _vp, _vp]),
    "sfm_ba_cost_partial": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_ba_update_cameras_prepare": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _i32, _i32, _f64, _f64, _vp, _vp, _vp,
                                             _vp]),
    "sfm_ba_update_cameras_held": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _i32, _i32, _f64, _f64, _vp, _vp, _vp,
                                             _vp]),
    "sfm_ba_cost": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_ba_cost_finish": (_int, [_vp, _i32, _vp, _vp]),
    "sfm_ba_cost_noalloc": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_ba_profile_enable": (_int, [_int]),
    "sfm_ba_profile_select": (_int, [C.c_uint64]),
    "sfm_ba_profile_kernel_count": (_int, []),
    "sfm_ba_profile_kernel_name": (C.c_char_p, [_int]),
    "sfm_ba_profile_read": (_int, [C.POINTER(_f64), C.POINTER(_i64)]),
}

_lib = None


class SfmBaError(RuntimeError):
    """A failure reported by libsfm_ba.so that has no counterpart in the reference."""


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.sfm_ba_version() != 1:
        raise ImportError("libsfm_ba.so version mismatch")
    _lib = lib
    return lib


def error_string(code: int) -> str:
    return load().sfm_ba_error_string(code).decode()


def check(rc: int):
    """Translate a status code into the exception the reference raises in the same situation."""
    if rc == OK:
        return
    if rc == ENONUNIQUE:
        raise ValueError("Found non-unique (i, j) pair")  # src/sparse_ba.py:62
    if rc in (EMISSING_POINT, EMISSING_VIEW):
        raise AssertionError(error_string(rc))  # src/sparse_ba.py:89,91
    if rc == ENOTPD:
        raise np.linalg.LinAlgError("Singular matrix")  # numpy.linalg.solve at src/sparse_ba.py:275
    msg = error_string(rc)
    if rc == EHIP:
        msg += ": " + load().sfm_ba_last_hip_error().decode()
    raise SfmBaError(msg)


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def profile_enable(on, chained=False, kernels=None):
    """Per-kernel HIP-event timing on/off.  ``chained``: consecutive timed launches share an event
    (only valid when nothing else is enqueued on the stream between the library's calls).
    ``kernels``: names of the kernels to time (default: all)."""
    lib = load()
    mask = (1 << 64) - 1
    if kernels is not None:
        names = [lib.sfm_ba_profile_kernel_name(i).decode() for i in range(lib.sfm_ba_profile_kernel_count())]
        mask = 0
        for k in kernels:
            mask |= 1 << names.index(k)
    check(lib.sfm_ba_profile_select(mask))
    check(lib.sfm_ba_profile_enable((2 if chained else 1) if on else 0))


def profile_read():
    """{kernel name: (total milliseconds, launches)} since the last read."""
    lib = load()
    k = lib.sfm_ba_profile_kernel_count()
    ms = (C.c_double * k)()
    cnt = (C.c_int64 * k)()
    check(lib.sfm_ba_profile_read(ms, cnt))
    return {lib.sfm_ba_profile_kernel_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(k) if cnt[i]}
