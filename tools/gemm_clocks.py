#!/usr/bin/env python
"""Effective shader clock of every kernel of a tools/pmc_summary.py listing (profiles/rNN_pmc_sq.txt): GRBM_GUI_ACTIVE is summed over
the 8 XCDs, so clock = GRBM_GUI_ACTIVE / 8 / duration; and the matrix pipe's busy share, SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
GRBM_GUI_ACTIVE / 8).  (Durations under counter collection are longer than production ones; the clock is the ratio.)
    python tools/gemm_clocks.py profiles/r04_pmc_sq.txt profiles/r05_pmc_sq.txt"""
import re
import sys

for path in sys.argv[1:]:
    print("== " + path)
    name = dur = grid = None
    vals = {}

    def flush():
        if name and "GRBM_GUI_ACTIVE" in vals and dur:
            clk = vals["GRBM_GUI_ACTIVE"] / 8.0 / dur / 1e3
            busy = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * vals["GRBM_GUI_ACTIVE"] / 8.0)
            print("  %-58s grid %-8s %7.1f us  %.3f GHz  MFMA pipe busy %5.1f %%" % (name[:58], grid, dur, clk, 100.0 * busy))
    for line in open(path):
        m = re.search(r"(dg::)?([A-Za-z_0-9]+(<[^>]*>)?)\s+#(\d+)\s+n=\d+\s+avg_dur_us=([0-9.]+)", line)
        if m:
            flush()
            name, grid, dur, vals = m.group(2), m.group(4), float(m.group(5)), {}
            continue
        m = re.match(r"\s+([A-Z_0-9]+)\s+([0-9.]+)\s*$", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    flush()
