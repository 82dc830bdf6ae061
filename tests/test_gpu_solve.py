"""GPU tests of the reduced camera solve through the C ABI: the paths (general blocked Cholesky,
single-workgroup band chain, block cyclic reduction on 64-blocks and on superblocks for wider
bands) against numpy on the same system.

The reference solves S delta_a = e with numpy.linalg.solve (src/sparse_ba.py:274-276); the
systems here are well conditioned, so every path must agree with it to ~1e-13.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _system(n, hb, seed):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n + 16))
    S = G @ G.T / n
    if hb is not None:
        i = np.arange(n)
        S = np.where(np.abs(i[:, None] - i[None, :]) <= hb, S, 0.0)
    S = S + np.diag(np.abs(S).sum(1) + 0.5)  # diagonally dominant: positive definite
    e = rng.standard_normal(n)
    return S, e


def _solve(S, e, hb, banded_entry):
    import torch
    import sfm_amd  # noqa: F401
    from sfm_amd import _lib

    lib = _lib.load()
    n = S.shape[0]
    buf = torch.from_numpy(np.vstack([S, e[None, :]])).cuda()
    da = torch.zeros((n,), dtype=torch.float64, device="cuda")
    ws = torch.empty((int(lib.sfm_ba_cholesky_workspace_bytes(n)) // 8 + 8,), dtype=torch.float64, device="cuda")
    info = torch.zeros((1,), dtype=torch.int32, device="cuda")
    p = _lib.ptr
    if banded_entry:
        _lib.check(lib.sfm_ba_solve_reduced_banded(p(buf), n, int(hb), p(da), p(ws), p(info), None))
    else:
        _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
    torch.cuda.synchronize()
    return da.cpu().numpy(), int(info.item()), buf[:n].cpu().numpy()


# (n, half bandwidth): one block; chain of 2 and 3 blocks (partial last block); cyclic reduction with
# 4, 5, 8, 11 and 37 blocks, full and partial last block, narrow and full-width bands
BAND_CASES = [(50, 10), (64, 63), (100, 48), (130, 55), (192, 64), (256, 48), (300, 55), (512, 64), (700, 20),
              (2333, 48)]


@pytest.mark.parametrize("n,hb", BAND_CASES)
def test_banded_paths_match_numpy(n, hb):
    S, e = _system(n, hb, seed=n)
    x_ref = np.linalg.solve(S, e)
    x, info, S_after = _solve(S, e, hb, banded_entry=True)
    assert info == 0
    assert np.linalg.norm(x - x_ref) <= 1e-12 * np.linalg.norm(x_ref)
    # the narrow-band paths only read the matrix
    assert np.array_equal(S_after, S)
    # same system through the general path
    x2, info2, _ = _solve(S, e, hb, banded_entry=False)
    assert info2 == 0
    assert np.linalg.norm(x2 - x_ref) <= 1e-12 * np.linalg.norm(x_ref)


# bands wider than one 64-block: cyclic reduction on superblocks of 2 or 4 tiles (128, 256 unknowns)
# with 9, 9 (partial last one), 33, 5 and 14, 13, 6 superblocks; (2500, 300) has too wide a band and
# (900, 200) too few superblocks for it: general path
WIDE_CASES = [(1100, 128, 128), (1030, 70, 128), (4200, 65, 128), (600, 100, 128), (3400, 130, 256), (3300, 200, 256),
              (1300, 200, 256), (2500, 300, 0), (900, 200, 0)]


@pytest.mark.parametrize("n,hb,expected_block", WIDE_CASES)
def test_wide_band_reduction_matches_numpy(n, hb, expected_block):
    S, e = _system(n, hb, seed=n)
    x_ref = np.linalg.solve(S, e)
    x, info, S_after = _solve(S, e, hb, banded_entry=True)
    assert info == 0
    assert np.linalg.norm(x - x_ref) <= 1e-12 * np.linalg.norm(x_ref)
    # the reduction only reads the matrix; the general path factors it in place
    import sfm_amd  # noqa: F401
    from sfm_amd import _lib

    block = int(_lib.load().sfm_ba_reduced_solve_block(n, hb))
    assert block == expected_block
    assert np.array_equal(S_after, S) == (block != 0)


def test_wide_band_takes_general_path():
    S, e = _system(400, 150, seed=3)
    x_ref = np.linalg.solve(S, e)
    x, info, _ = _solve(S, e, 150, banded_entry=True)
    assert info == 0
    assert np.linalg.norm(x - x_ref) <= 1e-12 * np.linalg.norm(x_ref)


@pytest.mark.parametrize("n,hb,bad_row", [(100, 48, 70), (640, 48, 200), (640, 48, 0), (640, None, 333), (1100, 100, 500),
                                           (1100, 100, 1099)])
def test_not_positive_definite_is_reported(n, hb, bad_row):
    """A non-positive pivot is what numpy's LU would not notice but a Cholesky must: the library
    reports 1 + the index of a failing pivot (status SFM_BA_ENOTPD at the Python layer)."""
    S, e = _system(n, hb, seed=7)
    S[bad_row, bad_row] = -1.0
    _, info, _ = _solve(S, e, hb if hb is not None else 0, banded_entry=hb is not None)
    assert info != 0


# dense systems through the general path (one launch per panel): one panel, two, the look-ahead with and without
# a partial last panel, n a multiple of 64 (the right-hand side in a row tile of its own), the size of the
# benchmark's dense reduced system, and one above the size from which the trailing matrix takes two panels per pass
@pytest.mark.parametrize("n", [40, 65, 130, 200, 640, 1000, 3500, 3520, 9100])
def test_dense_general_path(n):
    """Against an LU solve of the same system (GPU, fp64) at 1e-11; the factor left in S reproduces the
    matrix (sizes up to 1000); the workspace is scratch - filled with NaN, then with huge numbers, the result
    must not change by a bit; no pivot flag."""
    import torch
    import sfm_amd  # noqa: F401
    from sfm_amd import _lib

    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(n)
    G = torch.randn((n, n + 16), dtype=torch.float64, device="cuda", generator=g)
    S0 = G @ G.T / n + 0.5 * torch.eye(n, dtype=torch.float64, device="cuda")
    e = torch.randn((n,), dtype=torch.float64, device="cuda", generator=g)
    x_ref = torch.linalg.solve(S0, e)
    buf = torch.empty((n + 1, n), dtype=torch.float64, device="cuda")
    da = torch.empty((n,), dtype=torch.float64, device="cuda")
    ws = torch.empty((int(lib.sfm_ba_cholesky_workspace_bytes(n)) // 8 + 8,), dtype=torch.float64, device="cuda")
    info = torch.zeros((1,), dtype=torch.int32, device="cuda")
    p = _lib.ptr
    first = None
    for fill in (float("nan"), 1e300, 0.0):
        buf[:n].copy_(S0)
        buf[n].copy_(e)
        ws.fill_(fill)
        da.fill_(float("nan"))
        _lib.check(lib.sfm_ba_solve_reduced(p(buf), n, p(da), p(ws), p(info), None))
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        if first is None:
            first = da.clone()
            assert float((da - x_ref).norm() / x_ref.norm()) <= 1e-11
            if n <= 1000:
                L = torch.tril(buf[:n])
                assert float((L @ L.T - S0).abs().max() / S0.abs().max()) <= 1e-13
        else:
            assert bool((da == first).all().item())
