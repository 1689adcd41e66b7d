#!/usr/bin/env python
"""Long-horizon sensitivity of the projection loop: device fp32 vs the torch-CPU restatement in fp32 and fp64 on the same
inputs.  Shows which (arch, gain, lr) regimes are contractive (all three agree after L = 200) and which are chaotic
(fp32 vs fp64 of the SAME algorithm already differ by tens of percent), cf. SURVEY.md section 8c.
    python tools/diag_long_horizon.py celeba 2.0 10 3 1"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from defensegan_amd import synth
from tests.helpers import clean_targets, make_gan
from oracle import torch_ref as T

arch = sys.argv[1] if len(sys.argv) > 1 else "celeba"
gain = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
lrs = [float(v) for v in sys.argv[3:]] or [10.0]
R, nb = 10, 1
gan, p = make_gan(arch, wseed=1234, gain=gain, bias_range=0.0, rec_rr=R, rec_iters=200)
x, _ = clean_targets(p, arch, 4, seed=41)
z0 = synth.make_z(4 * R, 128, seed=42)
for lr in lrs:
    gan.rec_lr = lr
    for L in (50, 200):
        gan.rec_iters = L
        out = gan.reconstruct(x[:nb], z_init_val=z0[:nb * R], return_details=True)
        t32 = T.reconstruct(p, x[:nb], z0[:nb * R], R, L, lr=lr, arch=arch)
        t64 = T.reconstruct(p, x[:nb].astype(np.float64), z0[:nb * R].astype(np.float64), R, L, lr=lr, arch=arch, dtype=torch.float64)
        s = t64["loss"].max()
        print("lr %5.1f L %3d  loss gpu mean %.3e min %.3e | t64 mean %.3e min %.3e | max|gpu-t64|/max %.2e  max|t32-t64|/max %.2e  idx %s %s %s"
              % (lr, L, out["loss"].mean(), out["loss"].min(), t64["loss"].mean(), t64["loss"].min(),
                 np.abs(out["loss"] - t64["loss"]).max() / s, np.abs(t32["loss"] - t64["loss"]).max() / s,
                 out["idx"], t32["idx"], t64["idx"]))
