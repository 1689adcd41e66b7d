#!/usr/bin/env python
"""HBM-side bytes per launch and per kernel symbol from two rocprofv3 PMC passes (rocpd sqlite), as bench.py's
`roofline.traffic` reads them.

    rocprofv3 --pmc FETCH_SIZE -d out -o fetch -- python bench.py --steps 1 --warmup 1 --rec_iters 4 --no-cpu-baseline --no-profile
    rocprofv3 --pmc WRITE_SIZE -d out -o write -- python bench.py ... (same command)
    python tools/pmc_traffic.py mnist out/fetch_results.db out/write_results.db >> json

Only the dispatches of the LAST projection call are averaged (everything after the second-to-last select_kernel): the warm-up
call also holds the launches that time the candidate job lists.

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide
coalesced stream).  Infinity-Cache hits are included in FETCH_SIZE (the counter sits at the L2/fabric boundary)."""
import collections
import json
import re
import sqlite3
import sys


def short_name(k):
    k = k.replace("(anonymous namespace)::", "")
    return re.sub(r"^void ", "", k.split("(")[0]).replace("dg::", "")


def last_call_start(c):
    """dispatch id after which only the last dg_reconstruct's kernels follow."""
    ids = sorted(set(d for d, in c.execute("select dispatch_id from counters_collection where kernel_name like '%select_kernel%'")))
    return ids[-2] if len(ids) >= 2 else -1


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    first = last_call_start(c)
    cur = c.execute("select kernel_name, dispatch_id, value from counters_collection where counter_name = ? and dispatch_id > ?",
                    (counter, first))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, d, v in cur:
        k = short_name(k)
        acc[k][d] += v
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def layer_labels(names, workload):
    """Layer of every dispatch of one projection call from the FIXED launch order of a GD iteration (dg_engine.cpp
    enqueue_steps): Linear + deconv forwards, the tail, the backwards in reverse, the momentum update.  One kernel symbol serves
    several layers, and which symbol a layer uses is a timed choice -- the position in the sequence is not."""
    fwd = ["F1", "F2", "F3"] + (["F5"] if workload == "celeba" else [])
    bwd = (["B5"] if workload == "celeba" else []) + ["B3", "B2", "B1"]
    out, nf, nb, seen_tail = [], 0, 0, False
    for k in names:
        if "gemm_batched_kernel" in k or "lin_stationary_kernel" in k:
            if not seen_tail:
                out.append(fwd[nf] if nf < len(fwd) else "F?"); nf += 1
            else:
                out.append(bwd[nb] if nb < len(bwd) else "B?"); nb += 1
        elif "tail" in k:
            seen_tail = True
            out.append({"mnist_tail_pipe": "T5fb", "mnist_tail_mfma_kernel": "T5", "celeba_tail_fwd": "T6f", "celeba_tail_bwd": "T6b"}.get(
                next((p for p in ("mnist_tail_pipe", "mnist_tail_mfma_kernel", "celeba_tail_fwd", "celeba_tail_bwd") if p in k), ""), "T?"))
        elif "momentum_update_kernel" in k:
            out.append("UPD")
            nf, nb, seen_tail = 0, 0, False
        else:
            out.append("")
    return out


def per_layer(path, counter, workload):
    c = sqlite3.connect(path)
    first = last_call_start(c)
    rows = list(c.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ? and dispatch_id > ? "
                          "order by dispatch_id", (counter, first)))
    per_dispatch = collections.OrderedDict()
    for d, k, v in rows:                                  # one row per (dispatch, XCD / instance): sum them
        e = per_dispatch.setdefault(d, [short_name(k), 0.0])
        e[1] += v
    names = [e[0] for e in per_dispatch.values()]
    labels = layer_labels(names, workload)
    acc = collections.defaultdict(list)
    sym = {}
    for (name, v), lab in zip(per_dispatch.values(), labels):
        if lab:
            acc[lab].append(v)
            sym.setdefault(lab, set()).add(name)
    return {l: sum(v) / len(v) for l, v in acc.items()}, {l: len(v) for l, v in acc.items()}, {l: sorted(s) for l, s in sym.items()}


def main():
    workload, fdb, wdb = sys.argv[1:4]
    build = sys.argv[4] if len(sys.argv) > 4 else None
    tuning = sys.argv[5] if len(sys.argv) > 5 else None
    f, nf = per_kernel(fdb, "FETCH_SIZE")
    w, _ = per_kernel(wdb, "WRITE_SIZE")
    out = {}
    for k in sorted(f, key=lambda k: -f[k]):
        if f[k] + w.get(k, 0.0) < 1024:          # < 1 MB per launch: not worth a line
            continue
        out[k] = {"launches_profiled": nf[k], "fetch_kb_raw": round(f[k], 1), "write_kb": round(w.get(k, 0.0), 1),
                  "bytes_per_launch": int((2.0 * f[k] + w.get(k, 0.0)) * 1024)}
    lf, ln, lsym = per_layer(fdb, "FETCH_SIZE", workload)
    lw, _, _ = per_layer(wdb, "WRITE_SIZE", workload)
    by_layer = {}
    for l in sorted(lf, key=lambda l: -lf[l]):
        by_layer[l] = {"kernel": " | ".join(lsym[l]), "launches_profiled": ln[l], "fetch_kb_raw": round(lf[l], 1),
                       "write_kb": round(lw.get(l, 0.0), 1), "bytes_per_launch": int((2.0 * lf[l] + lw.get(l, 0.0)) * 1024)}
    doc = {workload: {"by_layer": by_layer, "by_symbol": out}}
    if build:
        doc["builds"] = {workload: build}
    if tuning:
        doc["tuning_ids"] = {workload: tuning}
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
