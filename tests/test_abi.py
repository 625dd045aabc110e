"""CPU checks of the C-ABI boundary: the library is built, loads, and exports exactly what
include/gridhip.h declares; without a GPU every compute path refuses loudly (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "gridhip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gridhip_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for need in ("gridhip_grid", "gridhip_convgrid", "gridhip_convgrid2", "gridhip_degrid2",
                 "gridhip_convgrid2_dev", "gridhip_create", "gridhip_last_error"):
        assert need in syms


def test_library_exports_every_declared_symbol():
    from gridhip import _lib
    assert os.path.exists(_lib.LIB_PATH), "libgridhip.so missing: run __graft_entry__.build()"
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in gridhip.h but not exported: {missing}"
    # and the Python prototypes cover the header too
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_library_is_gfx950_code_object():
    from gridhip import _lib
    out = subprocess.run(["strings", "-a", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_version_and_strerror():
    from gridhip import _lib
    lib = _lib.load()
    assert lib.gridhip_version() >= 100
    assert lib.gridhip_strerror(0) == b"ok"
    assert b"argument" in lib.gridhip_strerror(-1)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present; covered by the gpu tests")
    import gridhip
    with pytest.raises(gridhip.GridHipError) as ei:
        gridhip.Context(0)
    assert ei.value.code == gridhip._lib.ENODEV


def test_null_context_is_rejected():
    from gridhip import _lib
    lib = _lib.load()
    assert lib.gridhip_synchronize(None) == _lib.EINVAL
    assert lib.gridhip_set_option(None, b"tile", 64) == _lib.EINVAL
    assert lib.gridhip_last_error(None) == b"null context"
