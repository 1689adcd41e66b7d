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


def main():
    workload, fdb, wdb = sys.argv[1:4]
    build = sys.argv[4] if len(sys.argv) > 4 else None
    f, nf = per_kernel(fdb, "FETCH_SIZE")
    w, _ = per_kernel(wdb, "WRITE_SIZE")
    out = {}
    for k in sorted(f, key=lambda k: -f[k]):
        if f[k] + w.get(k, 0.0) < 1024:          # < 1 MB per launch: not worth a line
            continue
        out[k] = {"launches_profiled": nf[k], "fetch_kb_raw": round(f[k], 1), "write_kb": round(w.get(k, 0.0), 1),
                  "bytes_per_launch": int((2.0 * f[k] + w.get(k, 0.0)) * 1024)}
    doc = {workload: out}
    if build:
        doc["build"] = build
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
