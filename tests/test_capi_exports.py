"""The C-ABI library loads and exports every symbol declared in include/tangram_hip.h (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "tangram_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z_]+)\s*\(", txt)))


def test_header_declares_the_expected_entry_points():
    from tangram_amd import _capi
    assert set(_capi.EXPORTS) == set(_declared())


def test_hip_library_exports_every_declared_symbol():
    from tangram_amd import _build
    path = _build.build()
    lib = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.tg_abi_version.restype = ctypes.c_int
    assert lib.tg_abi_version() == 1


def test_argument_errors_are_reported_not_aborted():
    """tg_query_sizes validates its configuration on the host (no GPU involved)."""
    from tangram_amd import _build, _capi
    lib = _capi._declare(ctypes.CDLL(_build.build()))
    cfg = _capi.TgConfig()
    sizes = _capi.TgSizes()
    assert lib.tg_query_sizes(ctypes.byref(cfg), ctypes.byref(sizes)) == -1          # abi_version 0
    cfg.abi_version = 1
    cfg.n_cells, cfg.n_genes, cfg.n_spots = 10, 5, 7
    cfg.lambda_g1 = 0.0
    assert lib.tg_query_sizes(ctypes.byref(cfg), ctypes.byref(sizes)) == -1          # lambda_g1 cannot be 0
    assert b"lambda_g1" in lib.tg_last_error()
    cfg.lambda_g1 = 1.0
    assert lib.tg_query_sizes(ctypes.byref(cfg), ctypes.byref(sizes)) == 0
    assert sizes.m_pitch == 64 and sizes.state_bytes > 0 and sizes.workspace_bytes > 0


def test_product_path_has_no_cpu_fallback():
    import numpy as np
    from tangram_amd import _capi
    from tangram_amd.mapping_optimizer import Mapper
    assert not _capi.is_emulated()
    with pytest.raises(RuntimeError):
        Mapper(np.ones((4, 3), np.float32), np.ones((5, 3), np.float32), device="cpu")
