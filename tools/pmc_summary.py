#!/usr/bin/env python
"""Per-kernel average of rocprofv3 --pmc counters from a rocpd sqlite db.
    python tools/pmc_summary.py gpurun_out/pmc/sq_results.db [more.db ...]
"""
import collections
import sqlite3
import sys


def summarize(path):
    c = sqlite3.connect(path)
    # only the last projection call (after the second-to-last select_kernel): the first one also times candidate job lists
    ids = sorted(set(d for d, in c.execute("select dispatch_id from counters_collection where kernel_name like '%select_kernel%'")))
    first = ids[-2] if len(ids) >= 2 else -1
    cur = c.execute("select kernel_name, dispatch_id, counter_name, value, duration, grid_size, lds_block_size, "
                    "vgpr_count, accum_vgpr_count, sgpr_count from counters_collection where dispatch_id > ?", (first,))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(dict)
    meta = {}
    for k, d, name, v, du, grid, lds, vg, ag, sg in cur:
        k = k.replace("(anonymous namespace)::", "").split("(")[0][-48:] + " #%d" % grid      # one layer = one (symbol, grid)
        acc[k][name] += v
        cnt[k].add(d)
        dur[k][d] = du
        meta[k] = (grid, lds, vg, ag, sg)
    print("# %s" % path)
    for k in sorted(acc, key=lambda k: -sum(dur[k].values())):
        n = len(cnt[k])
        if sum(dur[k].values()) < 50000:
            continue
        print("%-50s n=%d avg_dur_us=%.1f grid=%s lds=%s vgpr=%s agpr=%s sgpr=%s" % ((k, n, sum(dur[k].values()) / n / 1e3) + meta[k]))
        for name in sorted(acc[k]):
            print("    %-34s %16.1f" % (name, acc[k][name] / n))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarize(p)
