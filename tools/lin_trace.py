#!/usr/bin/env python
"""Phase stamps of the weight-stationary Linear kernels (dg_linear.hip, measurement build; engine option job_trace = F1 / B1):
per workgroup the shader-clock time from its start to (first block staged + weights loaded), (first block multiplied),
(first block stored), (end).    python tools/lin_trace.py F1 [B=256] [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import archs, synth
from defensegan_amd.gan import dataset_gan_dict

op = sys.argv[1] if len(sys.argv) > 1 else "F1"
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
gan.set_option("graph_max_rows", 0)
for k, v in opts.items():
    gan.set_option(k, v)
x = gan.generate(gan.init_latents(B, seed=1))
x = torch.clamp(x + 0.3 * torch.sign(torch.randn_like(x)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)
gan.set_option("job_trace", op)
gan.reconstruct(x, seed=2)
t = gan.debug_read("job_trace", 65536 * 4 * 2).cpu().numpy().view(np.int64).reshape(-1, 8)
t = t[(t[:, 4] > 0) & (t[:, 0] > 0)]
t0 = t[:, 0].min()
us = lambda c: c / 2400.0
print("%s: %d workgroups, blocks per workgroup %s" % (op, len(t), dict(zip(*np.unique(t[:, 6], return_counts=True)))))
print("  start spread %.2f us; kernel span (first start -> last end) %.2f us" % (us(t[:, 0].max() - t0), us(t[:, 4].max() - t0)))
nb = t[:, 6].astype(np.float64)
rows = [("start -> first block ready (weights + DMA + barrier)", 0, 1, nb > 0), ("block 0 (nothing to write out yet)", 1, 2, nb > 0),
        ("block 1 (block 0's write-out rides inside)", 2, 3, nb > 1), ("whole workgroup", 0, 4, nb > 0)]
for name, i, j, m in rows:
    if not m.any():
        continue
    d = us(t[m, j] - t[m, i])
    print("  %-56s mean %7.2f us  p10 %7.2f  p90 %7.2f  max %7.2f" % (name, d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max()))
m = nb > 2
if m.any():
    print("  blocks 2.. + the last write-out, per block: mean %.2f us" % (us(t[m, 4] - t[m, 3]) / (nb[m] - 2)).mean())
cu = ((t[:, 5] >> 8) & 15) | (((t[:, 5] >> 13) & 7) << 4) | (t[:, 7] << 8)
print("  CUs by number of workgroups they ran {n: CUs}: %s" % dict(zip(*np.unique(np.unique(cu, return_counts=True)[1], return_counts=True))))
