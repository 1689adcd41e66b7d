#!/usr/bin/env python
"""Turns a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into the text summary kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    rows = cur.fetchall()
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# %-74s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        if pct < 0.01:
            continue
        name = name if len(name) <= 74 else name[:71] + "..."
        print("%-76s %8d %14.1f %12.3f %7.2f" % (name, calls, tot, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1])
