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
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "arch": arch = v
    elif k == "B": B = int(v)
    else: opts[k] = v
a = archs.make_arch(arch)
gan = dataset_gan_dict[arch](cfg={"USE_BN": False}, test_mode=True, measure=True, rec_rr=R, rec_iters=3, device=0)
gan.set_weights(synth.make_weights(arch, seed=1234, gain=2.0))
for k, v in opts.items():
    gan.set_option(k, v)
x = gan.generate(gan.init_latents(B, seed=1))
x = torch.clamp(x + 0.3 * torch.sign(torch.randn_like(x)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)                       # builds / tunes the job lists
gan.set_option("job_trace", op)
gan.reconstruct(x, seed=2)
t = gan.debug_read("job_trace", 65536 * 4 * 2).cpu().numpy().view(np.int64).reshape(-1, 4)
t = t[t[:, 1] > 0]
start, end, hwid, chunks = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
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
