"""Importable alias for the ``structure-from-motion_amd/`` package directory.

The directory name carries a hyphen and cannot be written in an ``import``
statement; ``import sfm_amd`` loads that directory as the package ``sfm_amd``.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "structure-from-motion_amd")
_spec = importlib.util.spec_from_file_location(
    "sfm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sfm_amd"] = _mod
_spec.loader.exec_module(_mod)
