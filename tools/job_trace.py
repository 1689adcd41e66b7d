#!/usr/bin/env python
"""Timeline of one GEMM launch (engine option job_trace): start / end (100 MHz ticks) and HW_ID of every workgroup (= job).
Prints how the launch's span splits into ramp, full occupancy and tail, and the busy fraction of the CU slots.
    python tools/job_trace.py F2 [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import archs, synth
from defensegan_amd.gan import dataset_gan_dict

op = sys.argv[1] if len(sys.argv) > 1 else "F2"
arch, B, R = "mnist", 256, 10
opts = {}
tuning = dump = None
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "arch": arch = v
    elif k == "B": B = int(v)
    elif k == "tuning": tuning = v          # a dg_export_tuning text (profiles/rNN_tuning_<arch>.txt): trace THOSE lists
    elif k == "dump": dump = v              # .npz of the raw records (block, start, end, hw id, chunks) + the tuning text
    else: opts[k] = v
a = archs.make_arch(arch)
gan = dataset_gan_dict[arch](cfg={"USE_BN": False}, test_mode=True, measure=True, rec_rr=R, rec_iters=3, device=0)
gan.set_weights(synth.make_weights(arch, seed=1234, gain=2.0))
for k, v in opts.items():
    gan.set_option(k, v)
if tuning:
    gan.import_tuning(open(tuning).read())
x = gan.generate(gan.init_latents(B, seed=1))
x = torch.clamp(x + 0.3 * torch.sign(torch.randn_like(x)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)                       # builds / tunes the job lists
gan.set_option("job_trace", op)
gan.reconstruct(x, seed=2)
t = gan.debug_read("job_trace", 65536 * 4 * 2).cpu().numpy().view(np.int64).reshape(-1, 4)
block = np.nonzero(t[:, 1] > 0)[0]
t = t[t[:, 1] > 0]
start, end, hwid, chunks = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
if dump:
    np.savez_compressed(dump, block=block, start=start, end=end, hwid=hwid, chunks=chunks, op=op, arch=arch, rows=B * R,
                        tuning=gan.export_tuning())
t0 = start.min()
s_us, e_us = (start - t0) / 100.0, (end - t0) / 100.0
span = e_us.max()
print("%s: %d jobs, span %.1f us, sum of job times %.0f us" % (op, len(t), span, (e_us - s_us).sum()))
grid = np.arange(0.0, span, 1.0)
active = ((s_us[None, :] <= grid[:, None]) & (e_us[None, :] > grid[:, None])).sum(axis=1)
peak = active.max()
print("resident workgroups: peak %d; mean over the span %.0f (%.1f %% of peak)" % (peak, active.mean(), 100.0 * active.mean() / peak))
first_full = grid[np.argmax(active >= 0.95 * peak)]
last_full = grid[len(active) - 1 - np.argmax(active[::-1] >= 0.95 * peak)]
print("ramp until %.1f us, >= 95 %% occupancy until %.1f us, tail %.1f us" % (first_full, last_full, span - last_full))
for lo in range(0, int(span) + 1, max(1, int(span) // 12)):
    print("  t = %4d us: %4d resident" % (lo, active[min(lo, len(active) - 1)]))
d = e_us - s_us
for c in np.unique(chunks):
    m = chunks == c
    print("  chunks %3d: %5d jobs, duration mean %.1f us (min %.1f, max %.1f), first start %.1f, last end %.1f"
          % (c, m.sum(), d[m].mean(), d[m].min(), d[m].max(), s_us[m].min(), e_us[m].max()))
# HW_ID fields (gfx9 layout): wave slot 3:0, SIMD 5:4, CU 11:8, SE 15:13, workgroup slot on the CU (TG_ID) 19:16
tg = (hwid >> 16) & 15
cu = ((hwid >> 8) & 15) | (((hwid >> 13) & 7) << 4)
print("workgroup slots (HW_ID.TG_ID) seen: %s" % dict(zip(*np.unique(tg, return_counts=True))))
prio = ((hwid >> 16) & 3)
for pr in np.unique(prio):
    m = prio == pr
    print("  slot & 3 = %d: %5d jobs, mean duration per chunk %.3f us" % (pr, m.sum(), (d[m] / np.maximum(chunks[m], 1)).mean()))

# ---- how level is the END of the launch?  per-CU time of its last job's end: a CU that finishes early idles until the launch ends
xcd = block % 8
cu_key = xcd * 256 + cu
fin = np.array([e_us[cu_key == k].max() for k in np.unique(cu_key)])
busy = np.array([(e_us[cu_key == k] - s_us[cu_key == k]).sum() for k in np.unique(cu_key)])
print("CUs seen %d; last-job end per CU: mean %.1f us, p10 %.1f, min %.1f, max %.1f (= span)  -> %.1f %% of CU-time idle at the end"
      % (len(fin), fin.mean(), np.percentile(fin, 10), fin.min(), fin.max(), 100.0 * (1.0 - fin.mean() / fin.max())))
print("jobs per CU: min %d max %d; sum of job durations per CU: mean %.0f us (p10 %.0f, p90 %.0f)" % (
    min((cu_key == k).sum() for k in np.unique(cu_key)), max((cu_key == k).sum() for k in np.unique(cu_key)), busy.mean(), np.percentile(busy, 10), np.percentile(busy, 90)))
# residency-weighted: time during which a CU holds at least 1 / at least 2 jobs
for need in (1, 2):
    tot = 0.0
    for k in np.unique(cu_key):
        m = cu_key == k
        ev = sorted([(a, 1) for a in s_us[m]] + [(b, -1) for b in e_us[m]])
        c, last, acc = 0, 0.0, 0.0
        for tt, d in ev:
            if c >= need:
                acc += tt - last
            c += d
            last = tt
        tot += acc
    print("mean time a CU holds >= %d workgroup(s): %.1f us of %.1f" % (need, tot / len(np.unique(cu_key)), span))
