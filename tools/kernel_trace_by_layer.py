#!/usr/bin/env python
"""Per-LAYER kernel statistics from a rocprofv3 --kernel-trace csv.

rocprofv3 --stats aggregates by kernel SYMBOL, and one symbol serves several layers here (gemm_batched_kernel<0, 2, 1> is both
the Linear forward, 33 us, and Generator.2's forward, 277 us), so a symbol's average means nothing; the launches that time
candidate job lists (dg_prepare) are mixed into every symbol's Calls as well.  A layer's launches all have the same
(symbol, grid size, LDS size) -- its job list -- so grouping by that triple separates the layers; groups with fewer than
`min_calls` launches (the candidate lists that were not kept) are folded into one "other" row per symbol.

    python tools/kernel_trace_by_layer.py <kernel_trace.csv> [bench.json] [min_calls] > per_layer.csv

With the JSON line bench.py printed in the same run, each group is labelled with the layer whose hipEvent average (bench.py's
"kernels" rows, same symbol) is closest to the group's average.

Row groups on several streams (CelebA's default, option two_streams): the same launch runs OVERLAPPED with the other group's
kernels in the timed region and ALONE in bench.py's marked step.  A launch counts as overlapped when another kernel of the
trace runs during more than 10 % of its duration; the two kinds get separate rows ("<layer> [overlapped]": its duration is not
a rate) -- bench.py's roofline is checked against the rows of the launches that ran alone.
"""
import collections
import csv
import json
import math
import re
import sys


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("dg::", "")
    return re.sub(r"^void ", "", k.split("(")[0])


def same_kernel(label, symbol):
    """bench.py's label of a launch against the trace's symbol: equal, the bare name, or the symbol with its defaulted trailing
    template argument (lin_stationary_kernel<4, 2> is lin_stationary_kernel<4, 2, false>)."""
    return label == symbol or label == symbol.split("<")[0] or label == symbol.replace(", false>", ">")


def main():
    path = sys.argv[1]
    rest = sys.argv[2:]
    layers = []
    if rest and not rest[0].isdigit():
        try:
            with open(rest[0]) as f:
                layers = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1]).get("kernels", [])
        except Exception:
            layers = []
        rest = rest[1:]
    min_calls = int(rest[0]) if rest else 64

    def layer_of(name, mean_ns):
        cands = [k for k in layers if same_kernel(k["kernel"], name)]
        if not cands:
            return ""
        best = min(cands, key=lambda k: abs(k["avg_us"] * 1e3 - mean_ns))
        if len(cands) == 1:
            return best["name"]                  # the symbol serves ONE layer: its launches are that layer's whatever the marker overhead was
        # (under the profiler the hipEvent markers add 3-6 us to a launch -- 20+ us to the update kernel in one traced run: relative
        # tolerance for the long launches, absolute for the short ones)
        return best["name"] if abs(best["avg_us"] * 1e3 - mean_ns) <= max(0.25 * mean_ns, 7000.0) else ""
    launches = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            grid = int(row["Grid_Size_X"]) // max(1, int(row["Workgroup_Size_X"]))
            launches.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), (short(row["Kernel_Name"]), grid, int(row["LDS_Block_Size"]))))
    launches.sort()
    # time each launch shares with other kernels: sweep in start order, `active` = launches that have not ended yet
    shared = [0] * len(launches)
    active = []
    for i, (st, en, _) in enumerate(launches):
        active = [j for j in active if launches[j][1] > st]
        for j in active:
            ov = min(en, launches[j][1]) - st
            if ov > 0:
                shared[i] += ov
                shared[j] += ov
        active.append(i)
    groups = collections.defaultdict(list)
    for (st, en, key), sh in zip(launches, shared):
        groups[key + (sh > 0.10 * max(1, en - st),)].append(en - st)
    folded = collections.defaultdict(list)
    rows = []
    alone_mean = {}
    for (name, grid, lds, ovl), d in groups.items():
        if not ovl:
            alone_mean[(name, grid, lds)] = sum(d) / len(d)
    for (name, grid, lds, ovl), d in groups.items():
        if len(d) < min_calls and "gemm_batched_kernel" in name:
            folded[name].extend(d)
        elif ovl:
            # labelled through the same launch's ALONE rows (its own average says nothing about which layer it is)
            lay = layer_of(name, alone_mean[(name, grid, lds)]) if (name, grid, lds) in alone_mean and grid else ""
            rows.append((name, grid, lds, d, (lay + " [overlapped]").strip()))
        else:
            rows.append((name, grid, lds, d, None))
    for name, d in folded.items():
        rows.append((name + " [candidate job lists, not kept]", 0, 0, d, None))
    total = sum(sum(r[3]) for r in rows)
    w = csv.writer(sys.stdout)
    w.writerow(["Layer", "Kernel", "Workgroups", "LDS_bytes", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, grid, lds, d, label in sorted(rows, key=lambda r: -sum(r[3])):
        n, s = len(d), sum(d)
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in d) / n)
        lay = label if label is not None else (layer_of(name, mean) if grid else "")
        w.writerow([lay, name, grid, lds, n, s, round(mean, 1), round(100.0 * s / total, 2), min(d), max(d), round(sd, 1)])


if __name__ == "__main__":
    main()
