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


# ---- oracle runs as fixtures ------------------------------------------------------------------------------------------------
# The long-horizon parity tests compare the device with torch-CPU float32 / float64 runs of oracle/torch_ref.py over L = 200
# steps: minutes of host time per test, inside the GPU suite's time limit (round 5: 775 s of 1200).  Those oracle OUTPUTS are
# data: they are generated once (tools/make_parity_fixtures.sh: this test suite with DG_WRITE_GOLDEN set, on a GPU box because
# some inputs are drawn by the device's Philox generator) and committed under tests/golden/oracle_*.npz together with the exact
# inputs they were computed from.  A test hands its inputs over; when the fixture's inputs match them (to 1e-6: a device-generated
# input may move in its last bit with a kernel change) the fixture's outputs AND ITS EXACT INPUTS are returned -- the device is
# then run on those very inputs --, otherwise the oracle is simply computed as before (slow, never wrong).
def oracle_fixture(name, inputs, compute):
    """(outputs, inputs to use).  ``inputs``: dict of arrays the oracle run depends on; ``compute(inputs) -> dict of arrays``."""
    path = os.path.join(GOLDEN, "oracle_" + name + ".npz")
    inputs = {k: np.ascontiguousarray(v) for k, v in inputs.items()}
    if os.path.exists(path) and not os.environ.get("DG_REGEN_GOLDEN"):
        with np.load(path) as f:
            d = {k: f[k] for k in f.files}
        ok = all(("in_" + k) in d and d["in_" + k].shape == v.shape and d["in_" + k].dtype == v.dtype and
                 float(np.abs(d["in_" + k].astype(np.float64) - v).max()) <= 1e-6 for k, v in inputs.items())
        if ok:
            return ({k: v for k, v in d.items() if not k.startswith("in_")}, {k: d["in_" + k] for k in inputs})
        print("[oracle_fixture] %s: the committed fixture was made from other inputs -- recomputing the oracle" % name)
    out = {k: np.asarray(v) for k, v in compute(inputs).items()}
    dst = os.environ.get("DG_WRITE_GOLDEN")
    if dst:
        os.makedirs(dst, exist_ok=True)
        np.savez(os.path.join(dst, "oracle_" + name + ".npz"), **{"in_" + k: v for k, v in inputs.items()}, **out)
    return out, inputs


def torch_runs(p, arch, R, L, lr, horizons=(), want32=True, want_rec=False):
    """compute() for oracle_fixture: the torch restatement in float64 (and float32) from inputs {x, z0}; per-restart losses at
    every horizon as ``l64_at_<L>`` / ``l32_at_<L>``, final ones as ``l64`` / ``l32``, float64 selection ``idx64`` (and ``rec64``)."""
    def compute(inp):
        import torch
        from oracle import torch_ref as T
        torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
        x, z0 = inp["x"], inp["z0"]
        out = {}
        t64 = T.reconstruct(p, x.astype(np.float64), z0.astype(np.float64), R, L, lr=lr, momentum=0.7, arch=arch, dtype=torch.float64,
                            loss_at=tuple(horizons))
        out["l64"] = np.asarray(t64["loss"], np.float64)
        out["idx64"] = np.asarray(t64["idx"], np.int64)
        if want_rec:
            out["rec64"] = np.asarray(t64["rec"], np.float32)
        for h in horizons:
            out["l64_at_%d" % h] = np.asarray(t64["loss_at"][h], np.float64)
        if want32:
            t32 = T.reconstruct(p, x, z0, R, L, lr=lr, momentum=0.7, arch=arch, loss_at=tuple(horizons))
            out["l32"] = np.asarray(t32["loss"], np.float64)
            for h in horizons:
                out["l32_at_%d" % h] = np.asarray(t32["loss_at"][h], np.float64)
        return out
    return compute
