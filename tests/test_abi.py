"""-m "not gpu": the C-ABI library loads and exports every symbol include/defensegan_hip.h declares."""
import os
import re

from defensegan_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "defensegan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(n for n, _, _ in _native.SYMBOLS)


def test_library_loads_and_exports_everything():
    lib = _native.load()
    for name in declared_symbols():
        assert getattr(lib, name) is not None
    assert lib.dg_version() == _native.ABI_VERSION


def test_no_gpu_calls_fail_cleanly():
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        return
    lib = _native.load()
    assert lib.dg_device_count() < 0
    h = C.c_void_p()
    assert lib.dg_create(0, 128, 64, 0, 0, C.byref(h)) != 0
    assert lib.dg_last_error()
    assert lib.dg_create(7, 128, 64, 0, 0, C.byref(h)) == -1          # DG_E_INVALID before any HIP call
    assert lib.dg_create(0, 100, 64, 0, 0, C.byref(h)) == -1
