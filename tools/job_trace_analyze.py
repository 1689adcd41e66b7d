#!/usr/bin/env python
"""Offline reading of the per-workgroup records tools/job_trace.py dumps (dump=<file>.npz): the job list is rebuilt on the host
from the tuning record stored with the trace (tests/support/plan_host_exec.cpp + dg_plan.cpp, no GPU), joined with the records by
workgroup index, and printed as
  * per (K chunks, tile shape): count, duration, start window, microseconds per chunk, cycles per MFMA issue slot of the job's waves
    (64 = a wave alone on its SIMD at full rate, 128 = two sharing it, ...), split into jobs of the first dispatch round and later ones;
  * the chip's matrix rate over time (every job's FLOP spread evenly over its life), resident jobs, CUs that hold work;
  * which sets of jobs the CUs started with and when those were done.
    python tools/job_trace_analyze.py gpurun_out/<dir>/trace_*.npz"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "defensegan_amd", "csrc")
SUP = os.path.join(ROOT, "tests", "support")
SO = os.path.join(SUP, "_build", "libdgplan_test.so")

LAYERS = {   # (arch, layer) -> planner call (no Batchnorm)
    ("mnist", "F2"): ("deconv_fwd", (4, 4, 7, 7, 256, 128, 128)), ("mnist", "F3"): ("deconv_fwd", (7, 7, 14, 14, 128, 64, 64)),
    ("mnist", "B2"): ("deconv_bwd", (4, 4, 7, 7, 256, 128, 256)), ("mnist", "B3"): ("deconv_bwd", (7, 7, 14, 14, 128, 64, 128)),
    ("celeba", "F2"): ("deconv_fwd", (4, 4, 8, 8, 256, 128, 128)), ("celeba", "F3"): ("deconv_fwd", (8, 8, 16, 16, 128, 64, 64)),
    ("celeba", "F5"): ("deconv_fwd", (16, 16, 32, 32, 64, 64, 64)), ("celeba", "B2"): ("deconv_bwd", (4, 4, 8, 8, 256, 128, 256)),
    ("celeba", "B3"): ("deconv_bwd", (8, 8, 16, 16, 128, 64, 128)), ("celeba", "B5"): ("deconv_bwd", (16, 16, 32, 32, 64, 64, 64)),
}
SLOTS = [2, 3, 5]
BM = [[128, 64, 64], [256, 128, 64]]
BN = [[128, 128, 64], [64, 64, 64]]


def _lib():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(SUP, "plan_host_exec.cpp"), os.path.join(CSRC, "dg_plan.cpp")]
    deps = srcs + [os.path.join(CSRC, "dg_plan.h"), os.path.join(CSRC, "dg_types.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", SO] + srcs)
    l = C.CDLL(SO)
    l.dgp_build.restype = C.c_void_p
    l.dgp_build.argtypes = [C.c_char_p] + [C.c_int] * 7
    l.dgp2_build.restype = C.c_void_p
    l.dgp2_build.argtypes = [C.c_void_p]
    l.dgp2_classes.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp2_jobs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp2_job_pairs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp2_make_recorded.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                     C.c_double, C.c_char_p, C.c_int]
    return l


def load(path):
    d = np.load(path, allow_pickle=True)
    arch, op, rows = str(d["arch"]), str(d["op"]), int(d["rows"])
    rec = [ln for ln in str(d["tuning"]).splitlines() if ln.startswith("%s %d " % (op, rows))][0]
    f = rec.split()
    lvl, slack, snake, xhead, n_jobs = int(f[2]), float(f[3]), int(f[4]), float(f[6]), int(f[7])
    taper = float(f[9]) if len(f) > 9 else 0.0
    l = _lib()
    kind, p = LAYERS[(arch, op)]
    h2 = l.dgp2_build(l.dgp_build(kind.encode(), *(list(p) + [0] * (7 - len(p)))))
    line = C.create_string_buffer(256)
    n = l.dgp2_make_recorded(h2, op.encode(), rows, 256, SLOTS[lvl], lvl, slack, snake, xhead, taper, line, 256)
    assert n == n_jobs, (n, n_jobs)
    jobs = (C.c_int * (6 * n))()
    l.dgp2_jobs(h2, jobs)
    jobs = np.array(jobs).reshape(n, 6)
    own = (C.c_int * (4 * n))()
    l.dgp2_job_pairs(h2, own)                      # a job's OWN K chunks (K-pair jobs cover half of their class's taps), pair id, role
    own = np.array(own).reshape(n, 4)
    blk = d["block"]
    t0 = d["start"].min()
    hw = d["hwid"]
    cu = (blk % 8) * 256 + (((hw >> 8) & 15) | (((hw >> 13) & 7) << 4))
    J = jobs[blk]
    assert (own[blk, 0] == d["chunks"]).all(), "the rebuilt list is not the traced one"
    return dict(arch=arch, op=op, rows=rows, fam=0 if p[-1] % 128 == 0 else 1, lvl=lvl, rec=rec, shape=J[:, 1], chunks=d["chunks"], paired=own[blk, 1] != 0,
                s=(d["start"] - t0) / 100.0, e=(d["end"] - t0) / 100.0, cu=cu)


def report(path):
    T = load(path)
    fam, shape, ch, s, e, cu = T["fam"], T["shape"], T["chunks"], T["s"], T["e"], T["cu"]
    d = e - s
    flop = 2.0 * ch * 32 * np.array([BM[fam][q] * BN[fam][q] for q in shape])
    mf = np.array([BM[fam][q] * BN[fam][q] // 256 for q in shape])           # MFMAs per wave and K chunk
    span = e.max()
    print("== %s %s, %d rows: list '%s' (%d resident per CU), %d jobs (%d of them K-pair halves), span %.1f us under the trace, %.1f TFLOP/s over the span" % (
        T["arch"], T["op"], T["rows"], T["rec"], SLOTS[T["lvl"]], len(s), int(T["paired"].sum()), span, flop.sum() / span / 1e6))
    for c in np.unique(ch)[::-1]:
        for q in np.unique(shape):
            for nm, m in (("first round", (ch == c) & (shape == q) & (s < 5)), ("later      ", (ch == c) & (shape == q) & (s >= 5))):
                if m.sum():
                    print("  %3d chunks %3dx%-3d %s: %4d jobs, start %5.0f..%-5.0f duration %6.1f (%5.1f..%5.1f) us = %.2f us/chunk, %3.0f cycles per MFMA slot" % (
                        c, BM[fam][q], BN[fam][q], nm, m.sum(), s[m].min(), s[m].max(), d[m].mean(), d[m].min(), d[m].max(), (d[m] / c).mean(),
                        (d[m] * 2400 / (c * mf[m])).mean()))
    print("  matrix rate over time (TFLOP/s; jobs resident; CUs holding work):")
    step = 10.0 if span < 400 else 20.0
    row = []
    for t0 in np.arange(0, span, step):
        ov = np.clip(np.minimum(e, t0 + step) - np.maximum(s, t0), 0, None)
        live = (s < t0 + step) & (e > t0)
        row.append("%3.0f:%5.1f/%d/%d" % (t0, (flop * ov / d).sum() / step / 1e6, live.sum(), len(np.unique(cu[live]))))
    for i in range(0, len(row), 6):
        print("    " + "  ".join(row[i:i + 6]))
    fin = np.array([e[cu == k].max() for k in np.unique(cu)])
    print("  last job of a CU ends at %.1f us on average (min %.1f, max %.1f = span): %.1f %% of CU-time idle at the end" % (
        fin.mean(), fin.min(), fin.max(), 100.0 * (1.0 - fin.mean() / fin.max())))
    first = {}
    for k in np.unique(cu):
        m = (cu == k) & (s < 5)
        key = tuple(sorted(("%dx%d" % (BM[fam][q], BN[fam][q]), int(c)) for c, q in zip(ch[m], shape[m])))
        first.setdefault(key, []).append(e[m].max())
    for key, v in sorted(first.items(), key=lambda kv: -len(kv[1]))[:6]:
        print("  %3d CUs started with %s: all of them done at %.0f us" % (len(v), " + ".join("%s x %d chunks" % kq for kq in key), np.mean(v)))


if __name__ == "__main__":
    for a in sys.argv[1:]:
        report(a)
        print()
