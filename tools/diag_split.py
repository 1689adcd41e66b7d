#!/usr/bin/env python
"""Diagnostic: role-split CelebA forward tail against the band kernel -- y, loss, dz of one loop body, per image row."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from defensegan_amd import archs, synth
from tests.test_gpu_variants import _make
B, R = 5, 10
a = archs.make_arch("celeba")
gan, p = _make("celeba", R=R, L=1)
rs = np.random.RandomState(17)
x = np.asarray(gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32)))
x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=18)
z = (rs.standard_normal((B * R, 128)) * 0.15).astype(np.float32)
y0, l0, d0 = gan.loss_grad(x, z)
g2, _ = _make("celeba", R=R, L=1)
g2.set_option("tail_fwd_split", 512)
y1, l1, d1 = g2.loss_grad(x, z)
dy = np.abs(y1 - y0)
print("y   max abs diff %.3e, unequal elements %d of %d" % (dy.max(), (y1 != y0).sum(), y0.size))
bad = np.argwhere(y1 != y0)
if len(bad):
    print("first unequal (row, i, j, co):", bad[:12].tolist())
    rows = np.bincount(bad[:, 1], minlength=64)
    print("unequal per output row i:", rows.tolist())
    cols = np.bincount(bad[:, 2], minlength=64)
    print("unequal per output col j:", cols.tolist())
print("loss rel diff %.3e" % (np.abs(l1 - l0) / np.abs(l0)).max())
print("dz  max abs diff %.3e (max |dz| %.3e), unequal %d" % (np.abs(d1 - d0).max(), np.abs(d0).max(), (d1 != d0).sum()))
