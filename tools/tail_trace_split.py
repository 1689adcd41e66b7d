#!/usr/bin/env python
"""Phase timing of the role-split CelebA forward tail (measurement build, option tail_trace): cycles per step of M wave 0 and
G wave 4 of every workgroup, last launch.   python tools/tail_trace_split.py [workgroups]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from defensegan_amd import archs, synth
from defensegan_amd.gan import dataset_gan_dict
B, R, L = 128, 10, 4
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
a = archs.make_arch("celeba")
gan = dataset_gan_dict["celeba"](cfg={"USE_BN": False}, test_mode=True, measure=True, rec_rr=R, rec_iters=L, device=0)
gan.set_weights(synth.make_weights("celeba", seed=1234, gain=2.0))
gan.set_option("tail_fwd_split", W)
x = torch.clamp(gan.generate(gan.init_latents(B, seed=1)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)
gan.set_option("tail_trace", "1")
gan.reconstruct(x, seed=1)
t = gan.debug_read("tail_trace", 4096 * 16).cpu().numpy().view(np.int64).reshape(-1, 16)[:W & 0xffff]
steps = np.maximum(t[:, 4], 1)[:, None]
m = t[:, :4] / steps
g = t[:, 8:11] / steps
names_m = ["wait for the staged rows", "fragment reads + DMA issue + 80 MFMAs", "barrier A (G done with P)", "P stores + barrier B"]
names_g = ["x load + gather + tanh + stores", "barrier A", "barrier B"]
print("cycles per step, mean / p10 / p90 over %d workgroups (steps per workgroup %d)" % (len(t), int(steps.mean())))
for i, nm in enumerate(names_m):
    print("  M wave 0: %-40s %7.0f %7.0f %7.0f" % (nm, m[:, i].mean(), np.percentile(m[:, i], 10), np.percentile(m[:, i], 90)))
print("  M total %.0f" % m.sum(axis=1).mean())
for i, nm in enumerate(names_g):
    print("  G wave 4: %-40s %7.0f %7.0f %7.0f" % (nm, g[:, i].mean(), np.percentile(g[:, i], 10), np.percentile(g[:, i], 90)))
print("  G total %.0f" % g.sum(axis=1).mean())
