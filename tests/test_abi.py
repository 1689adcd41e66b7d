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


def test_measurement_build_is_a_separate_library_with_the_same_abi():
    """Same sources with -DDG_MEASURE (traces, phase-removal switches, superseded cross-check kernels): its own file, never
    the one the product path loads; same symbols."""
    lib, mlib = _native.load(), _native.load(measure=True)
    assert _native.MEASURE_LIB_PATH != _native.LIB_PATH and mlib is not lib
    for name in declared_symbols():
        assert getattr(mlib, name) is not None
    assert mlib.dg_version() == _native.ABI_VERSION


def test_product_library_holds_no_measurement_kernels():
    """The kernels the verdict called dead-by-default must be absent from the product library's code objects and present
    in the measurement build."""
    def blob(path):
        with open(path, "rb") as f:
            return f.read()
    prod, meas = blob(_native.LIB_PATH), blob(_native.MEASURE_LIB_PATH)
    for sym in (b"27celeba_tail_fwd_mfma_kernelILi", b"27celeba_tail_bwd_mfma_kernelILi", b"celeba_tail_fwd16_kernelILi64ELb1E"):   # mangled
        assert sym not in prod, sym
        assert sym in meas, sym
    for sym in (b"celeba_tail_fwd16_kernel", b"celeba_tail_bwd_persist_kernel", b"mnist_tail_pipe_kernel", b"gemm_batched_kernel"):
        assert sym in prod, sym


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


def test_python_mirror_reports_native_errors_without_a_gpu():
    """The host class turns a non-zero status into NativeError carrying the library's message (and raises, never falls back,
    when no GPU is visible)."""
    import numpy as np
    import pytest
    import torch
    from defensegan_amd.gan import MnistDefenseGAN
    g = MnistDefenseGAN(cfg={"USE_BN": False}, test_mode=True)
    with pytest.raises(_native.NativeError, match="defensegan_hip error -1"):
        g._check(-1)
    g._check(0)
    if not torch.cuda.is_available():
        with pytest.raises(_native.NativeError, match="no CPU fallback"):
            g.reconstruct(np.zeros((1, 28, 28, 1), np.float32))
        with pytest.raises(_native.NativeError, match="no CPU fallback"):
            g.prepare(4)
