"""MI355X-native bundle adjustment: the Gauss-Newton inner loop of an incremental
structure-from-motion pipeline as hand-written HIP kernels behind a C-ABI library
(``csrc/``, ``include/sfm_ba.h``), with a Python host layer that keeps the
reference's function signatures and numpy-in / numpy-out contract.

Import as ``sfm_amd`` (see ``sfm_amd.py`` at the repository root).  Modules that
compute (``device``, ``sparse_ba``, ``bundle_adjustment`` ...) load
``csrc/libsfm_ba.so`` on first use and raise if it is missing.
"""
from . import synthetic  # noqa: F401
from .frame import Frame, Map  # noqa: F401

__all__ = ["synthetic", "Frame", "Map"]
