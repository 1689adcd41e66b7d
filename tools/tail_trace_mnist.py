#!/usr/bin/env python
"""Phase timing of the pipelined MNIST tail (engine option tail_trace): per-workgroup cycle totals of the last launch
(wave 0 = an MFMA wave, wave 8 = a gather wave), steady-state steps only.   python tools/tail_trace_mnist.py [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import archs, synth
from defensegan_amd.gan import dataset_gan_dict

B, R, L = 256, 10, 4
a = archs.make_arch("mnist")
gan = dataset_gan_dict["mnist"](cfg={"USE_BN": False}, test_mode=True, measure=True, rec_rr=R, rec_iters=L, device=0)
gan.set_weights(synth.make_weights("mnist", seed=1234, gain=2.0))
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    gan.set_option(k, v)
x = torch.clamp(gan.generate(gan.init_latents(B, seed=1)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)
gan.set_option("tail_trace", "1")
gan.reconstruct(x, seed=1)
t = gan.debug_read("tail_trace", 2048 * 32).cpu().numpy().view(np.int64).reshape(-1, 16)
t = t[t[:, 5] > 0]
it = t[:, 5].astype(float)
print("workgroups", len(t), "steady steps/wg", it.mean(), "clock GHz %.3f" % (t[:, 6].mean() / t[:, 7].mean() * 0.1))
for q, nm in enumerate(["M read_frags", "M stage issue", "M bwd", "M fwd", "M barrier"]):
    v = t[:, q] / it
    print("%-14s cycles/step mean %6.0f  p10 %6.0f p90 %6.0f" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
for q, nm in ((8, "G gather"), (9, "G barrier")):
    v = t[:, q] / it
    print("%-14s cycles/step mean %6.0f  p10 %6.0f p90 %6.0f" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
print("M sum/step %.0f ; workgroup lifetime us mean %.1f" % ((t[:, :5].sum(1) / it).mean(), t[:, 7].mean() / 100))
