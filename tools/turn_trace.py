#!/usr/bin/env python
"""Phase stamps of the fused latent turn (dg_turn.hip, measurement build; engine option job_trace = TURN): per workgroup the
shader-clock time of every phase of one launch.    python tools/turn_trace.py [arch=mnist] [B=256] [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import archs
from defensegan_amd import synth
from defensegan_amd.gan import dataset_gan_dict

arch, B, R = "mnist", 256, 10
opts = {}
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "arch": arch = v
    elif k == "B": B = int(v)
    else: opts[k] = v
a = archs.make_arch(arch)
gan = dataset_gan_dict[arch](cfg={"USE_BN": False}, test_mode=True, measure=True, rec_rr=R, rec_iters=4, device=0)
gan.set_weights(synth.make_weights(arch, seed=1234, gain=2.0))
gan.set_option("graph_max_rows", 0)
gan.set_option("turn_fused", 1)
gan.set_option("two_streams", 0)
for k, v in opts.items():
    gan.set_option(k, v)
x = gan.generate(gan.init_latents(B, seed=1))
x = torch.clamp(x + 0.3 * torch.sign(torch.randn_like(x)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)
gan.set_option("job_trace", "TURN")
gan.reconstruct(x, seed=2)
t = gan.debug_read("job_trace", 65536 * 4 * 2).cpu().numpy().view(np.int64).reshape(-1, 8)
t = t[(t[:, 7] > 0) & (t[:, 0] > 0)]
t0 = t[:, 0].min()
us = lambda c: c / 100.0          # s_memrealtime: 100 MHz, one time base for all XCDs
print("latent turn (%s, %d rows): %d workgroups" % (arch, B * R, len(t)))
print("  start spread %.2f us; kernel span (first start -> last end) %.2f us" % (us(t[:, 0].max() - t0), us(t[:, 7].max() - t0)))
names = ["start -> backward ready (weights + first block)", "backward multiplies", "last write-out + drain of the partials", "barrier 1 (arrive, poll)",
         "update (loads, fmas, stores) + drain of z", "barrier 2 (forward weights arrive meanwhile)", "z staged + forward multiplies + write-out"]
for k, nm in enumerate(names):
    d = us(t[:, k + 1] - t[:, k])
    print("  %-52s mean %7.2f us  p10 %7.2f  p90 %7.2f  max %7.2f" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max()))
for k, nm in enumerate(["start", "backward ready", "backward multiplied", "partials drained", "past barrier 1", "z drained", "past barrier 2", "end"]):
    d = us(t[:, k] - t0)
    print("  at %-20s mean %7.2f us  min %7.2f  max %7.2f" % (nm, d.mean(), d.min(), d.max()))
# within a row group (16 / 32 consecutive workgroups): how far apart do its workgroups start / reach the barriers?
ns = int(opts.get("nsplit", 32 if arch == "celeba" else 16))
if len(t) % ns == 0:
    g = t.reshape(-1, ns, 8)
    for k, nm in [(0, "start"), (2, "backward multiplied"), (3, "partials drained"), (5, "z drained")]:
        sp = us(g[:, :, k].max(axis=1) - g[:, :, k].min(axis=1))
        print("  spread inside a row group at %-20s mean %6.2f us  max %6.2f" % (nm, sp.mean(), sp.max()))
    # who drains slowly?  by K slice (= blockIdx % nsplit) and by XCD (= blockIdx % 8)
    d = us(g[:, :, 3] - g[:, :, 2])
    print("  write-out + drain by K slice: " + " ".join("%.1f" % v for v in d.mean(axis=0)))
    print("  barrier 1 by K slice:         " + " ".join("%.1f" % v for v in us(g[:, :, 4] - g[:, :, 3]).mean(axis=0)))
    print("  share of row groups whose slowest drain is slice k: " + " ".join("%d" % v for v in np.bincount(d.argmax(axis=1), minlength=ns)))
    print("  write-out + drain by row group: " + " ".join("%.1f" % v for v in d.max(axis=1)))
