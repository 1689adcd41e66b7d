import os

import numpy as np

from defensegan_amd import archs, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        d = {k: f[k] for k in f.files}
    for k in ("arch",):
        d[k] = str(d[k])
    for k in ("wseed", "R", "L"):
        d[k] = int(d[k])
    for k in ("gain", "bias_range", "lr", "momentum"):
        d[k] = float(d[k])
    return d


def make_gan(arch="mnist", wseed=1234, gain=2.0, bias_range=0.1, rec_rr=10, rec_iters=200, rec_lr=10.0,
             use_bn=False, **kw):
    from defensegan_amd.gan import dataset_gan_dict
    a = archs.make_arch(arch)
    cls = dataset_gan_dict[{"fmnist": "f-mnist"}.get(arch, arch)]
    gan = cls(cfg={"USE_BN": bool(use_bn), "LATENT_DIM": a.latent_dim, "NET_DIM": a.net_dim}, test_mode=True,
              rec_rr=rec_rr, rec_iters=rec_iters, rec_lr=rec_lr, **kw)
    p = synth.make_weights(arch, seed=wseed, gain=gain, bias_range=bias_range, use_bn=use_bn,
                           bn_jitter=0.2 if use_bn else 0.0)
    assert gan.set_weights(p) == []
    return gan, p


def clean_targets(p, arch, B, seed):
    """x = G(z_true), z_true ~ N(0, 1/latent): clean, in-range targets."""
    from oracle import defensegan_oracle as O
    a = archs.make_arch(arch)
    rs = np.random.RandomState(seed)
    zt = (rs.standard_normal((B, a.latent_dim)) * np.sqrt(1.0 / a.latent_dim)).astype(np.float32)
    x, _ = O.generator_forward(p, zt, arch)
    return x.astype(np.float32), zt


# ---- the distributional parity tier shared by tests/test_gpu_parity_tiers.py and tests/test_gpu_config4.py ----
def _paired_permutation_p(a, b, rs):
    """P(mean|a| - mean|b| >= observed) when each image's two deviations are exchangeable (one-sided, paired): exact for
    <= 12 images, 20 000 random label swaps otherwise."""
    a, b = np.abs(a), np.abs(b)
    obs = a.mean() - b.mean()
    n = len(a)
    if n <= 12:
        swaps = ((np.arange(1 << n)[:, None] >> np.arange(n)) & 1).astype(bool)
    else:
        swaps = rs.rand(20000, n) < 0.5
    d = np.where(swaps, b - a, a - b).mean(axis=1)
    return float((d >= obs - 1e-18).mean())


def _bootstrap_se(stat, bdev, b64, rs, n_boot=2000):
    """Standard error of stat(bdev) - stat(b64) over resampled images (pairs kept together)."""
    n = len(b64)
    idx = rs.randint(0, n, size=(n_boot, n))
    return float(np.std([stat(bdev[i]) - stat(b64[i]) for i in idx]))


def decidable(l32, l64):
    """Images whose selected restart float32 rounding cannot change: the float64 top-2 gap exceeds twice the image's largest
    per-restart |f32 - f64| difference, and the float32 run selects the same restart as the float64 run."""
    srt = np.sort(l64, axis=1)
    spread = np.abs(l32 - l64).max(axis=1)
    return ((srt[:, 1] - srt[:, 0]) > 2.0 * spread) & (l32.argmin(axis=1) == l64.argmin(axis=1))


def distributional_tier(l32, l64, ldev, idx_dev, seed=0):
    """Per-restart losses [n_images, R] of the torch-float32 run, the torch-float64 run and the device, and the restart the
    device selected.  Asserts the SIZE / BIAS / SELECTION statements of the distributional tier and returns
    (message, decided mask):

    * SIZE: the device's per-image deviation of the best-restart loss from float64 is not significantly larger than
      torch-float32's (paired permutation test on mean|.|, one-sided, p >= 0.005) and never above 2.5 x in the mean;
    * BIAS: mean / p50 / p90 of the best-restart loss equal float64's within max(2 x torch-f32's own deviation of that
      statistic, 3 bootstrap standard errors of device - float64);
    * SELECTION: the device selects float64's restart on every decidable image."""
    l32, l64, ldev = np.asarray(l32, np.float64), np.asarray(l64, np.float64), np.asarray(ldev, np.float64)
    b32, b64, bdev = l32.min(axis=1), l64.min(axis=1), ldev.min(axis=1)
    d32, ddev = b32 - b64, bdev - b64
    rs = np.random.RandomState(seed)
    stat_fns = (np.mean, lambda v: np.percentile(v, 50), lambda v: np.percentile(v, 90))
    st = lambda v: np.array([f(v) for f in stat_fns])
    s32, s64, sdev = st(b32), st(b64), st(bdev)
    se = np.array([_bootstrap_se(f, bdev, b64, rs) for f in stat_fns])
    floor = 1e-6 * b64.mean()
    tol = np.maximum(np.maximum(2.0 * np.abs(s32 - s64), 3.0 * se), floor)
    p_size = _paired_permutation_p(ddev, d32, rs)
    dec = decidable(l32, l64)
    msg = ("best-restart loss  [mean, p50, p90]\n  f64 %s\n  f32 %s\n  dev %s\n  tol %s\n  mean|d32| %.3e  mean|ddev| %.3e  "
           "permutation p %.4f  decidable %d of %d" % (s64, s32, sdev, tol, np.abs(d32).mean(), np.abs(ddev).mean(), p_size,
                                                       dec.sum(), len(dec)))
    assert np.isfinite(ldev).all(), msg
    assert np.abs(ddev).mean() <= 2.5 * np.abs(d32).mean() + floor, msg
    assert p_size >= 0.005 or np.abs(ddev).mean() <= np.abs(d32).mean() + floor, msg
    assert (np.abs(sdev - s64) <= tol).all(), msg
    assert (np.asarray(idx_dev)[dec] == l64.argmin(axis=1)[dec]).all(), (msg, idx_dev, l64.argmin(axis=1), dec)
    return msg, dec
