#!/usr/bin/env python
"""Phase timing of the persistent CelebA backward tail (engine option tail_trace): per-workgroup cycle totals of
the last launch, wave 0 of each workgroup.   python tools/tail_trace.py [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import archs, synth
from defensegan_amd.gan import dataset_gan_dict

B, R, L = 128, 10, 4
a = archs.make_arch("celeba")
gan = dataset_gan_dict["celeba"](cfg={"USE_BN": False}, test_mode=True, rec_rr=R, rec_iters=L, device=0)
gan.set_weights(synth.make_weights("celeba", seed=1234, gain=2.0))
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    gan.set_option(k, v)
x = torch.clamp(gan.generate(gan.init_latents(B, seed=1)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)
gan.set_option("tail_trace", "1")
gan.reconstruct(x, seed=1)
t = gan.debug_read("tail_trace", 4096 * 16).cpu().numpy().view(np.int64).reshape(-1, 8)
t = t[t[:, 5] > 0]
names = ["fetch", "gv+mfma", "stores", "wait+park", "barrier"]
it = t[:, 5].astype(float)
print("workgroups", len(t), "items/wg", it.mean())
for q, nm in enumerate(names):
    v = t[:, q] / it
    print("%-10s cycles/item mean %.0f  p10 %.0f p90 %.0f" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
print("sum/item %.0f ; kernel cycles per workgroup mean %.0f ; wall 100MHz ticks mean %.0f -> shader clock %.3f GHz" % ((t[:, :5].sum(1) / it).mean(), t[:, 6].mean(), t[:, 7].mean(), t[:, 6].mean() / t[:, 7].mean() * 0.1))

beg = t[:, 4].astype(np.int64); end = beg + t[:, 7]
k0 = beg.min()
print("start spread (us): p50 %.1f p90 %.1f max %.1f ; end (us after first start): min %.1f p50 %.1f max %.1f ; lifetime us min %.1f max %.1f" % (
    np.percentile(beg - k0, 50) / 100, np.percentile(beg - k0, 90) / 100, (beg - k0).max() / 100,
    (end - k0).min() / 100, np.percentile(end - k0, 50) / 100, (end - k0).max() / 100, t[:, 7].min() / 100, t[:, 7].max() / 100))
