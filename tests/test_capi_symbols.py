"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/sfm_ba.h declares (no compute calls here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfm_ba.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfm_ba_[a-z_0-9A-Z]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    import ctypes

    import sfm_amd  # noqa: F401
    from sfm_amd import _lib

    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in sfm_ba.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "python binding table and header disagree"
    assert lib.sfm_ba_version() == 1
    assert _lib.error_string(0) == "ok"
    assert _lib.error_string(-3) == "Found non-unique (i, j) pair"
    assert lib.sfm_ba_cholesky_workspace_bytes(350) > 0


def test_product_path_does_not_import_oracle():
    """No module of the package may reference the oracle: it is test infrastructure only."""
    pkg = os.path.join(ROOT, "structure-from-motion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"
    # nor may the helper scripts: only tests/, the smoke check and bench.py's CPU-baseline leg do
    import re

    for f in os.listdir(os.path.join(ROOT, "scripts")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "scripts", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"scripts/{f} imports the oracle"


def test_status_codes_map_to_reference_exceptions():
    import numpy as np
    import pytest

    import sfm_amd  # noqa: F401
    from sfm_amd import _lib

    with pytest.raises(ValueError, match="Found non-unique"):
        _lib.check(_lib.ENONUNIQUE)
    with pytest.raises(AssertionError):
        _lib.check(_lib.EMISSING_POINT)
    with pytest.raises(AssertionError):
        _lib.check(_lib.EMISSING_VIEW)
    with pytest.raises(np.linalg.LinAlgError):
        _lib.check(_lib.ENOTPD)
    with pytest.raises(_lib.SfmBaError):
        _lib.check(_lib.EINVAL)
