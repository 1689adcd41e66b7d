#!/usr/bin/env python
"""Event-driven model of how the GEMM workgroups of one layer fill the chip: in-order dispatch into free
workgroup slots, processor sharing of each CU's MFMA pipes among its resident workgroups.
Used to reason about tile shapes / orderings (not a measurement)."""
import heapq
import sys

import numpy as np


def layer_tiles(kind, n_rows, bm, bn):
    """returns list of chunk counts per tile in dispatch order (sorted by K desc, position-major) and MFMAs/chunk/wave"""
    def taps1d_fwd(h_in, e):
        out = []
        for i in range(e):
            c = 0
            for kh in range(5):
                t = i + 1 - kh
                if t >= 0 and t % 2 == 0 and t // 2 < h_in:
                    c += 1
            out.append(c)
        return out

    def taps1d_bwd(h_in, e):
        return [sum(1 for kh in range(5) if 0 <= 2 * o + kh - 1 < e) for o in range(h_in)]
    spec = {"F2": ("f", 4, 7, 256, 128), "F3": ("f", 7, 14, 128, 64), "B3": ("b", 7, 14, 128, 64), "B2": ("b", 4, 7, 256, 128)}[kind]
    d, h_in, e, cin, cout = spec
    if d == "f":
        t1 = taps1d_fwd(h_in, e); kch = cin; ncols = cout
    else:
        t1 = taps1d_bwd(h_in, e); kch = cout; ncols = cin
    pos = sorted([a * b for a in t1 for b in t1], reverse=True)
    n_mt = (n_rows + bm - 1) // bm
    tiles = []
    for p in pos:
        for _ in range(ncols // bn):
            tiles += [p * kch // 32] * n_mt
    mf = (bm // 64) * (bn // 64) * 16     # MFMAs per chunk per wave
    return tiles, mf


def simulate(tiles, mf, slots_per_cu, n_cu=256, rate_tf=136.0, overhead_chunks=0.0):
    """processor sharing: a CU with k>=1 resident WGs splits its MFMA throughput equally. Work unit = MFMA-wave-slots."""
    # time unit: cycles of one SIMD at full MFMA rate; a tile needs chunks*mf*64 cycles on each of 4 SIMDs (4 waves)
    eff = rate_tf / 157.3
    work = [(c + overhead_chunks) * mf * 64.0 / eff for c in tiles]
    resident = [dict() for _ in range(n_cu)]   # cu -> {tile: remaining}
    t = 0.0
    nxt = 0
    n = len(work)
    done = 0
    # fill
    order = list(range(n_cu)) * slots_per_cu
    for cu in order:
        if nxt < n:
            resident[cu][nxt] = work[nxt]; nxt += 1
    while done < n:
        # next completion
        best = None
        for cu in range(n_cu):
            k = len(resident[cu])
            if k == 0:
                continue
            m = min(resident[cu].values())
            dt = m * k
            if best is None or dt < best[0]:
                best = (dt, cu)
        dt, _ = best
        t += dt
        for cu in range(n_cu):
            k = len(resident[cu])
            if k == 0:
                continue
            dec = dt / k
            fin = []
            for tid in resident[cu]:
                resident[cu][tid] -= dec
                if resident[cu][tid] <= 1e-6:
                    fin.append(tid)
            for tid in fin:
                del resident[cu][tid]
                done += 1
                if nxt < n:
                    resident[cu][nxt] = work[nxt]; nxt += 1
    return t / 2.4e3   # us at 2.4 GHz


if __name__ == "__main__":
    N = 2560
    for kind in ["F2", "F3", "B3", "B2"]:
        for (bm, bn, slots) in [(64, 128, 3), (128, 128, 2), (128, 64, 3), (64, 64, 4), (32, 128, 4)]:
            if kind in ("F3",) and bn == 128:
                continue
            tiles, mf = layer_tiles(kind, N, bm, bn)
            ideal = sum(tiles) * mf * 64.0 / (256 * 2.4e3) / (136 / 157.3)
            ms = simulate(tiles, mf, slots)
            print("%s %3dx%-3d slots=%d tiles=%5d  ideal(136TF)=%6.1f us  simulated=%6.1f us  eff=%.3f" % (kind, bm, bn, slots, len(tiles), ideal, ms, ideal / ms))
