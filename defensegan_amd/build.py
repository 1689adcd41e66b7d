"""Builds the HIP shared library in-tree (defensegan_amd/lib/libdefensegan_hip.so) for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdefensegan_hip.so")
# the same sources with -DDG_MEASURE: in-kernel phase traces, phase-removal switches and the superseded tail kernels kept as
# cross-checks (tools/, tests/test_gpu_variants.py).  Never loaded by the product path.
LIB_MEASURE = os.path.join(LIBDIR, "libdefensegan_hip_measure.so")
SOURCES = ["dg_engine.cpp", "dg_engine_lists.cpp", "dg_engine_prof.cpp", "dg_engine_opts.cpp", "dg_comm.cpp", "dg_plan.cpp", "dg_gemm.hip", "dg_fgemm.hip", "dg_linear.hip", "dg_turn.hip", "dg_tail_mnist.hip", "dg_tail_celeba.hip", "dg_bn.hip", "dg_small.hip", "dg_clf.hip"]
HEADERS = ["dg_engine.h", "dg_kernels.h", "dg_tail_common.h", "dg_device.h", "dg_plan.h", "dg_types.h", os.path.join("..", "..", "include", "defensegan_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _digest() -> str:
    """SHA-256 over the raw bytes of every source and header: the rebuild stamp (any edit rebuilds)."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _strip_comments(text: str) -> str:
    """C/C++ source without its comments, whitespace runs collapsed (string and character literals are left alone)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":                                      # literal: copy through the closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def code_digest() -> str:
    """Identity of the CODE the library is built from (bench.build_id(), profiles/): as _digest() but over the sources with
    comments and layout removed, so that correcting a comment does not orphan the measurements taken on that code."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "r", encoding="utf-8") as fh:
            h.update(_strip_comments(fh.read()).encode("utf-8"))
        h.update(b"\0")
    return h.hexdigest()


def _build_one(lib: str, objdir: str, defines, verbose: bool) -> None:
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".", "_") + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj, "-I", CSRC] + list(defines)
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _fresh(lib: str, stamp: str, dig: str) -> bool:
    return os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == dig


def build(force: bool = False, verbose: bool = True, measure: bool = True) -> str:
    """Builds the product library and (``measure``) the measurement library (-DDG_MEASURE) from the same sources; returns the
    product's path.  Each library has its own stamp, written as soon as that library is linked: a failing measurement build
    no longer leaves the product library looking stale."""
    os.makedirs(LIBDIR, exist_ok=True)
    dig = _digest()
    stamp = os.path.join(LIBDIR, "build.stamp")
    stamp_m = os.path.join(LIBDIR, "build_measure.stamp")
    if force or not _fresh(LIB, stamp, dig):
        _build_one(LIB, os.path.join(LIBDIR, "obj"), [], verbose)
        with open(stamp, "w") as fh:
            fh.write(dig)
    if measure and (force or not _fresh(LIB_MEASURE, stamp_m, dig)):
        _build_one(LIB_MEASURE, os.path.join(LIBDIR, "obj_measure"), ["-DDG_MEASURE"], verbose)
        with open(stamp_m, "w") as fh:
            fh.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, measure="--no-measure" not in sys.argv)
    print(LIB)
