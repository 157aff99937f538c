"""CPU: the C-ABI library loads and exports every symbol include/spurfies_hip.h declares; the ctypes
binding covers exactly that set; argument validation fails loudly without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "spurfies_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(spf_[a-z0-9_]+)\s*\(", src))


@pytest.fixture(scope="module")
def lib():
    from spurfies_amd import _lib, build

    build.build(verbose=False)
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from spurfies_amd import _lib

    decl = declared_functions()
    assert len(decl) >= 20
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert set(_lib.SIGNATURES) == decl, (set(_lib.SIGNATURES) ^ decl)
    assert lib.spf_abi_version() == 6
    assert not [n for n in decl if n.endswith("_set_mode") or n.endswith("_get_mode")], "no process-global modes in the ABI"
    assert lib.spf_geo_packed_floats() > 0 and lib.spf_color_packed_floats() > 0


def test_argument_validation_without_gpu(lib):
    from spurfies_amd import _lib

    h = ctypes.c_void_p()
    cfg = _lib.GridConfig()
    cfg.voxel_size[:] = (0.025, 0.025, 0.025)
    cfg.voxel_scale[:] = (3, 3, 3)
    cfg.kernel_size[:] = (2, 3, 3)            # even kernel size is refused
    cfg.ranges[:] = (-1, -1, -1, 1, 1, 1)
    assert lib.spf_grid_create(ctypes.byref(cfg), ctypes.byref(h)) == -22
    assert b"kernel_size" in lib.spf_last_error()
    cfg.kernel_size[:] = (3, 3, 3)
    assert lib.spf_grid_create(ctypes.byref(cfg), ctypes.byref(h)) == 0 and h.value
    assert lib.spf_grid_query(h, None, 4, 3, 9, 2.0, 2, None, None, None, None, None, None) == -22   # k > SPF_KMAX
    assert lib.spf_geo_forward(None, None, None, None, None, None, None, 8, 64, 9, None, None, None, 45.0, None, None, None, None, None, 0, None) == -22
    # the arithmetic is a per-call argument (no process-wide mode in the library): unknown values are refused
    assert lib.spf_geo_forward(None, None, None, None, None, None, None, 8, 64, 8, None, None, None, 45.0, None, None, None, None, None, 7, None) == -22
    assert b"arith" in lib.spf_last_error()
    assert lib.spf_wgrad(None, None, 256, 256, None, 16, None, 256, None, None, 0, 2, 0, 0, None) == -22 and b"arith" in lib.spf_last_error()
    assert lib.spf_wgrad(None, None, 104, 104, None, 16, None, 103, None, None, 0, 0, 39, 200, None) == -22 and b"col_rot" in lib.spf_last_error()
    assert lib.spf_render_forward(None, None, None, None, None, None, 4, 1000, None, None, None, None, None, None, None, None, None, None, None, 1.0, None) == -22
    lib.spf_grid_destroy(h)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from spurfies_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SpurfiesHipError, match="no fallback"):
        _lib.lib()
