#!/usr/bin/env python
"""Phase timing of the CelebA tails (engine option tail_trace): per-workgroup cycle totals of the last launch, wave 0 of
each workgroup.   python tools/tail_trace.py [fwd] [key=value ...]
  (default) the persistent backward tail;  fwd: the forward tail (tail_dbg = 8), workgroups 3072 .. 6143 of the launch"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import archs, synth
from defensegan_amd.gan import dataset_gan_dict

B, R, L = 128, 10, 4
a = archs.make_arch("celeba")
gan = dataset_gan_dict["celeba"](cfg={"USE_BN": False}, test_mode=True, measure=True, rec_rr=R, rec_iters=L, device=0)
gan.set_weights(synth.make_weights("celeba", seed=1234, gain=2.0))
FWD = "fwd" in sys.argv[1:]
for kv in sys.argv[1:]:
    if kv == "fwd":
        continue
    k, v = kv.split("=")
    gan.set_option(k, v)
if FWD:
    gan.set_option("tail_dbg", "8")
x = torch.clamp(gan.generate(gan.init_latents(B, seed=1)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)
gan.set_option("tail_trace", "1")
gan.reconstruct(x, seed=1)
t = gan.debug_read("tail_trace", 4096 * 16).cpu().numpy().view(np.int64).reshape(-1, 8)
if FWD:
    t = t[1024:]
    t = t[t.sum(axis=1) > 0]
    names = ["issue x/DMA/filter loads", "wait for the staged rows", "fragment reads", "barrier 1", "GEMM (5 units)", "barrier 2",
             "gather + tanh + stores", "loss reduce + exit"]
    print("workgroups traced: %d; cycles per phase (wave 0): mean / p10 / p90, share of the workgroup's lifetime" % len(t))
    tot = t.sum(axis=1).mean()
    for q, nm in enumerate(names):
        v = t[:, q]
        print("  %-28s %7.0f %7.0f %7.0f   %4.1f %%" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90), 100.0 * v.mean() / tot))
    print("  lifetime %.0f cycles (= %.2f us at 2.3 GHz); MFMA issue floor of the GEMM phase: 160 x 32 = 5120 cycles" % (tot, tot / 2300.0))
    sys.exit(0)
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
