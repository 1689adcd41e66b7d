"""ctypes binding of include/defensegan_hip.h (the C-ABI drop-in boundary).

The product path has NO fallback: if the HIP library is missing or does not load, importing a
compute entry point raises.  PyTorch-ROCm is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdefensegan_hip.so")
# same sources built with -DDG_MEASURE (in-kernel traces, phase-removal switches, superseded cross-check kernels): loaded only
# by tools/ and by the variant tests that ask for it, never by the product path
MEASURE_LIB_PATH = os.path.join(_HERE, "lib", "libdefensegan_hip_measure.so")

DG_OK = 0
ABI_VERSION = 1

# every symbol include/defensegan_hip.h declares: (name, restype, argtypes)
_vp, _i, _i64, _u64, _f, _cp = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_char_p
SYMBOLS = [
    ("dg_version", _i, []),
    ("dg_last_error", _cp, []),
    ("dg_device_count", _i, []),
    ("dg_device_info", _i, [_i, _cp, _i, C.POINTER(_i), C.POINTER(_i64)]),
    ("dg_create", _i, [_i, _i, _i, _i, _i, C.POINTER(_vp)]),
    ("dg_destroy", _i, [_vp]),
    ("dg_set_weights", _i, [_vp, _cp, _vp, C.POINTER(_i64), _i, _i]),
    ("dg_weights_complete", _i, [_vp]),
    ("dg_reconstruct", _i, [_vp, _vp, _vp, _u64, _i64, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    ("dg_prepare", _i, [_vp, _i, _i, _vp]),
    ("dg_call_row_groups", _i, [_vp, _i, _i]),
    ("dg_generate", _i, [_vp, _vp, _i, _vp, _vp]),
    ("dg_loss_grad", _i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    ("dg_init_latents", _i, [_vp, _vp, _i64, _u64, _i64, _f, _vp]),
    ("dg_profile_enable", _i, [_vp, _i]),
    ("dg_profile_count", _i, [_vp]),
    ("dg_profile_read", _i, [_vp, _i, _cp, _i, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("dg_profile_reset", _i, [_vp]),
    ("dg_debug_read", _i64, [_vp, _cp, _vp, _i64]),
    ("dg_set_option", _i, [_vp, _cp, _cp]),
    ("dg_export_tuning", _i64, [_vp, _vp, _i64]),
    ("dg_import_tuning", _i, [_vp, _cp]),
    ("dg_clf_create", _i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    ("dg_clf_destroy", _i, [_vp]),
    ("dg_clf_add_layer", _i, [_vp, _i, _i, _i, _i, _i, _i, _i]),
    ("dg_clf_output_width", _i, [_vp]),
    ("dg_clf_set_weights", _i, [_vp, _i, _vp, C.POINTER(_i64), _i, _vp, _i64, _i]),
    ("dg_clf_forward", _i, [_vp, _vp, _i, _vp, _vp, _vp]),
    ("dg_eval_batch", _i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("dg_clf_input_gradient", _i, [_vp, _vp, _vp, _i, _vp, _vp]),
    ("dg_fgsm", _i, [_vp, _vp, _vp, _i, _f, _f, _f, _vp, _vp]),
    ("dg_comm_unique_id", _i, [_vp]),
    ("dg_comm_create", _i, [_i, _vp, _i, _i, C.POINTER(_vp)]),
    ("dg_comm_destroy", _i, [_vp]),
    ("dg_gather_eval", _i, [_vp, _vp, _vp, _i64, _vp]),
]

_libs = {}


class NativeError(RuntimeError):
    pass


def load(measure: bool = False) -> C.CDLL:
    """Loads the in-tree HIP library (``measure``: its -DDG_MEASURE build); raises (never falls back) when it is missing."""
    if measure in _libs:
        return _libs[measure]
    path = MEASURE_LIB_PATH if measure else LIB_PATH
    if not os.path.exists(path):
        raise NativeError(
            "HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`python defensegan_amd/build.py`); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.dg_version()
    if v != ABI_VERSION:
        raise NativeError("ABI version mismatch: library %d, binding %d" % (v, ABI_VERSION))
    _libs[measure] = lib
    return lib


def check(rc: int, lib: Optional[C.CDLL] = None) -> None:
    if rc != DG_OK:
        msg = (lib or load()).dg_last_error()
        raise NativeError("defensegan_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
