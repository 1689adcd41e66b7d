#!/usr/bin/env python
"""Phase timing of mnist_tail_pipe3_kernel (measurement build, engine option tail_trace): per role (wave 0 forward GEMM, wave 6
backward GEMM, wave 12 gather) the shader cycles per steady-state step spent between the barriers (work) and inside them (wait).
    python tools/tail_trace_mnist3.py [B=256] [use_bn=1] [key=value ...]       (use_bn=1: the kernel's Batchnorm form, bn_fused = 2)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import archs, synth
from defensegan_amd.gan import dataset_gan_dict

B, R, L = 256, 10, 4
use_bn = "use_bn=1" in sys.argv[1:]
sys.argv = [v for v in sys.argv if not v.startswith("use_bn=")]
a = archs.make_arch("mnist")
gan = dataset_gan_dict["mnist"](cfg={"USE_BN": use_bn}, test_mode=True, measure=True, rec_rr=R, rec_iters=L, device=0)
gan.set_weights(synth.make_weights("mnist", seed=1234, gain=2.0, use_bn=use_bn, bn_jitter=0.2 if use_bn else 0.0))
gan.set_option("tail_pipe_version", 3)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "B":
        B = int(v)
    else:
        gan.set_option(k, v)
x = torch.clamp(gan.generate(gan.init_latents(B, seed=1)), a.in_lo, a.in_hi)
gan.reconstruct(x, seed=1)
gan.set_option("tail_trace", "1")
gan.reconstruct(x, seed=1)
t = gan.debug_read("tail_trace", 2048 * 32).cpu().numpy().view(np.int64).reshape(-1, 16)
t = t[t[:, 2] > 0]
print("workgroups %d, steady steps per workgroup %.1f (B = %d: %d rows per workgroup)" % (len(t), t[:, 2].mean(), B, B * R // 256))
for slot, nm in enumerate(["forward wave 0 (+ 192..195 forward)", "backward wave 6", "gather wave 12", "forward wave 3 (+ 192..195 backward)"]):
    n = t[:, slot * 3 + 2].astype(float)
    w, b = t[:, slot * 3] / n, t[:, slot * 3 + 1] / n
    print("%-38s work %6.0f cycles/step (p10 %6.0f p90 %6.0f)   barrier wait %6.0f   step %6.0f" % (nm, w.mean(), np.percentile(w, 10), np.percentile(w, 90), b.mean(), (w + b).mean()))
