"""Pickles of NumPy arrays that the Python-2 reference can read.

The reference loads its reconstruction cache and feature files with ``cPickle.load`` under Python 2 and a
NumPy <= 1.16 (/root/reference/models/gan.py:489, 527).  Protocol 2 is necessary but not sufficient: NumPy >= 2
pickles an ndarray through ``numpy._core.multiarray._reconstruct``, a module path that does not exist in the
NumPy of that era (only ``numpy.core``), so ``cPickle.load`` raises ImportError.  This pickler writes every
``numpy._core*`` global under its historical ``numpy.core*`` name; current NumPy still resolves those names, so the
files read back here as well.
"""
from __future__ import annotations

import pickle
import types

_NEW, _OLD = "numpy._core", "numpy.core"


class _Py2NumpyPickler(pickle._Pickler):          # the pure-Python pickler: its save_global can be overridden
    def save_global(self, obj, name=None):
        mod = getattr(obj, "__module__", None) or ""
        if mod == _NEW or mod.startswith(_NEW + "."):
            leaf = name or getattr(obj, "__qualname__", None) or obj.__name__
            self.write(pickle.GLOBAL + (_OLD + mod[len(_NEW):]).encode("ascii") + b"\n" + leaf.encode("ascii") + b"\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)

    # C functions (multiarray._reconstruct, multiarray.scalar) reach save_global through their __reduce_ex__ string;
    # Python-level functions through the dispatch table, which holds the base class's function object
    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[types.FunctionType] = save_global


def dump(obj, fileobj) -> None:
    """``pickle.dump(obj, fileobj, protocol=2)`` with Python-2-era NumPy module paths."""
    _Py2NumpyPickler(fileobj, protocol=2).dump(obj)


def dumps(obj) -> bytes:
    import io
    buf = io.BytesIO()
    dump(obj, buf)
    return buf.getvalue()
