"""Synthetic generator weights and projection inputs (no datasets / checkpoints on the GPU box).

Follows SURVEY.md section 8(d): tflib initialisers drawn from ``np.random.RandomState(seed)`` in
layer order, multiplied by a gain so that G(z) has dynamic range.

* Linear: glorot-uniform  U(+-sqrt(3)*sqrt(2/(in+out)))   -- tflib/ops/linear.py:55-60
* Deconv2D: he-uniform    U(+-sqrt(3)*sqrt(4/(fan_in+fan_out))), fan_in = Cin*25/4,
  fan_out = Cout*25, shape [5,5,Cout,Cin]                  -- tflib/ops/deconv2d.py:46-76
* biases: reference inits zero; ``bias_range`` > 0 draws U(+-bias_range) so the bias path is exercised.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .archs import Arch, make_arch, weight_shapes


def _uniform(rs: np.random.RandomState, stdev: float, shape) -> np.ndarray:
    lim = stdev * np.sqrt(3.0)
    return rs.uniform(low=-lim, high=lim, size=shape).astype(np.float32)


def make_weights(arch="mnist", seed: int = 1234, gain: float = 2.0, bias_range: float = 0.0,
                 use_bn: bool = False, latent_dim: int = 128, net_dim: int = 64,
                 bn_jitter: float = 0.0) -> Dict[str, np.ndarray]:
    a = arch if isinstance(arch, Arch) else make_arch(arch, latent_dim, net_dim)
    rs = np.random.RandomState(seed)
    w: Dict[str, np.ndarray] = {}
    w["Generator.Input.W"] = _uniform(
        rs, np.sqrt(2.0 / (a.latent_dim + a.lin_out)), (a.latent_dim, a.lin_out)) * np.float32(gain)
    for d in a.deconvs:
        fan_in = d.cin * 25 / 4.0
        fan_out = d.cout * 25.0
        w[d.name + ".Filters"] = _uniform(
            rs, np.sqrt(4.0 / (fan_in + fan_out)), (5, 5, d.cout, d.cin)) * np.float32(gain)
    # biases after all filters so that the filter stream does not depend on bias_range
    shapes = weight_shapes(a, use_bn)
    for name, shp in shapes.items():
        if name.endswith(".b") or name.endswith(".Biases"):
            if bias_range > 0:
                w[name] = rs.uniform(-bias_range, bias_range, size=shp).astype(np.float32)
            else:
                w[name] = np.zeros(shp, np.float32)
    if use_bn:
        for name, shp in shapes.items():
            if name.endswith(".scale"):
                w[name] = (np.ones(shp, np.float32) +
                           (rs.uniform(-bn_jitter, bn_jitter, size=shp).astype(np.float32)
                            if bn_jitter > 0 else 0)).astype(np.float32)
            elif name.endswith(".offset"):
                w[name] = (rs.uniform(-bn_jitter, bn_jitter, size=shp).astype(np.float32)
                           if bn_jitter > 0 else np.zeros(shp, np.float32))
    return w


def make_z(n_rows: int, latent_dim: int = 128, seed: int = 0, std: Optional[float] = None,
           first_row: int = 0) -> np.ndarray:
    """z rows ~ N(0, 1/latent_dim) (gan.py:370-375), keyed by the GLOBAL row index so the draw is
    independent of how the image list is sharded over GPUs (row = image*R + restart)."""
    if std is None:
        std = float(np.sqrt(1.0 / latent_dim))
    out = np.empty((n_rows, latent_dim), np.float32)
    # one small RandomState per block of 64 rows keeps this O(n) and shard-independent
    blk = 64
    r = first_row
    end = first_row + n_rows
    while r < end:
        b0 = (r // blk) * blk
        rs = np.random.RandomState([seed & 0x7FFFFFFF, b0 & 0x7FFFFFFF, b0 >> 31])
        blockvals = rs.standard_normal((blk, latent_dim)).astype(np.float32) * np.float32(std)
        lo = r - b0
        hi = min(blk, end - b0)
        out[r - first_row: r - first_row + (hi - lo)] = blockvals[lo:hi]
        r = b0 + hi
    return out


def adversarial(x: np.ndarray, eps: float, lo: float, hi: float, seed: int = 7) -> np.ndarray:
    """clip(x + eps*sign(N(0,1)), lo, hi): stand-in for FGSM eps=0.3 inputs
    (whitebox.py:199, blackbox.py:523-528)."""
    rs = np.random.RandomState(seed)
    s = np.sign(rs.standard_normal(x.shape)).astype(np.float32)
    return np.clip(x + np.float32(eps) * s, lo, hi).astype(np.float32)
