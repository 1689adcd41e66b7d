"""The hand-over kit for the parity pin that cannot be made in the build container (tools/make_tf_fixtures.py).

The reference is Python 2 + TensorFlow 1.7 and holds no tests or golden vectors; neither interpreter nor TF exists here, so
the oracle is "parity unpinned" (oracle/defensegan_oracle.py header).  tools/make_tf_fixtures.py runs the REFERENCE's own
``reconstruct`` on this repository's synthetic weights / inputs on a TF-1 machine and writes ``tests/golden/tf_<case>.npz``
(+ one Saver checkpoint per case under ``tests/golden/tf_ckpt/``).  The tests below

* always (CPU): check that the weight / latent generator embedded in that script still equals defensegan_amd.synth, i.e.
  that a maintainer who runs it computes on the arrays these tests expect;
* when the fixtures are present: pin the float64 oracle (CPU) and the HIP engine (-m gpu) to TensorFlow's outputs, and the
  TensorFlow-free checkpoint reader to a real ``tf.train.Saver`` file.  Absent fixtures skip -- with the reason.
"""
import glob
import importlib.util
import os

import numpy as np
import pytest

from defensegan_amd import archs, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _kit():
    spec = importlib.util.spec_from_file_location("make_tf_fixtures", os.path.join(ROOT, "tools", "make_tf_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_the_kit_generates_exactly_the_arrays_of_synth():
    kit = _kit()
    for name, arch, wseed, gain, bias_range, B, R, adv, zseed in kit.CASES:
        w, names = kit.make_weights(arch, wseed, gain, bias_range)
        want = synth.make_weights(arch, seed=wseed, gain=gain, bias_range=bias_range)
        assert sorted(names) == sorted(want), name
        for k in names:
            assert w[k].dtype == np.float32 and np.array_equal(w[k], want[k]), (name, k)
        a = archs.make_arch(arch)
        assert list(a.image_dim) == kit.image_dim(arch) and a.latent_dim == kit.LATENT and a.net_dim == kit.NET_DIM
        assert [(d.name, d.cin, d.cout) for d in a.deconvs] == kit.deconvs(arch)
        zt, z0 = kit.make_latents(zseed, B, R)
        assert zt.shape == (B, 128) and z0.shape == (B * R, 128) and B % R == 0      # the reference needs batch_size % rec_rr == 0
        if R >= 3:
            assert np.array_equal(z0[1], z0[2])                                      # the tie-break row
        if adv:
            x = synth.adversarial(np.full([B] + kit.image_dim(arch), 0.5, np.float32), 0.3, 0.0, 1.0, seed=zseed + 1)
            assert np.array_equal(np.clip(0.5 + np.float32(0.3) * kit.sign_noise(zseed, x.shape), 0.0, 1.0).astype(np.float32), x)
    chk = kit.self_check()
    assert set(chk) == {c[0] for c in kit.CASES} and all(len(v["weights"]) == 64 for v in chk.values())


def _fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "tf_*.npz")))


def _case(path):
    kit = _kit()
    f = np.load(path)
    name = os.path.basename(path)[3:-4]
    spec = [c for c in kit.CASES if c[0] == name][0]
    w, _ = kit.make_weights(spec[1], spec[2], spec[3], spec[4])
    return f, spec, w


NO_FIXTURES = ("no TensorFlow fixtures committed: run `python2 tools/make_tf_fixtures.py --reference <checkout of kabkabm/defensegan> "
               "--out tests/golden` on a Python-2.7 / TensorFlow-1.7 machine (the build image has neither)")


def _compare(f, spec, reconstruct, tol_scale):
    """reconstruct(x, z0, R, L) -> dict(rec [B..], idx [B], loss [B*R], rows [B*R..]) against the reference's outputs."""
    name, arch, wseed, gain, bias_range, B, R, adv, zseed = spec
    x, z0 = f["x"], f["z0"]
    for L in [int(v) for v in f["iters"]]:
        got = reconstruct(x, z0, R, L)
        rows_tf = f["rows_%d" % L].reshape(B * R, -1)
        # TF float32 against float64 / the device's float32: single-step agreement is ~1e-6; over L steps a ReLU gate within
        # rounding of zero may flip in a row, so the gate is on the bulk of the rows and a looser one on all
        err = np.abs(got["rows"].reshape(B * R, -1) - rows_tf).max(axis=1)
        assert np.median(err) < 2e-5 * tol_scale and (err < 2e-5 * tol_scale).mean() >= 0.9, (name, L, err)
        assert err.max() < 5e-2, (name, L, err)
        loss_tf = f["loss_%d" % L]
        assert np.allclose(got["loss"], loss_tf, rtol=2e-3 * tol_scale, atol=1e-7), (name, L)
        # selection: wherever TF's own top-2 gap is far above the disagreement in the losses
        srt = np.sort(loss_tf.reshape(B, R), axis=1)
        gap = srt[:, 1] - srt[:, 0] if R > 1 else np.ones(B)
        dec = gap > 4.0 * np.abs(got["loss"] - loss_tf).reshape(B, R).max(axis=1)
        assert (np.asarray(got["idx"])[dec] == f["idx_%d" % L][dec]).all(), (name, L)
        rec_tf = f["rec_%d" % L].reshape(B, -1)
        e2 = np.abs(got["rec"].reshape(B, -1) - rec_tf).max(axis=1)
        assert (e2[dec] < 5e-2).all() and np.median(e2[dec]) < 2e-5 * tol_scale if dec.any() else True, (name, L, e2)


@pytest.mark.parametrize("path", _fixtures() or [None])
def test_oracle_against_the_reference_run_under_tensorflow(path):
    if path is None:
        pytest.skip(NO_FIXTURES)
    from oracle import defensegan_oracle as O
    f, spec, w = _case(path)

    def rec(x, z0, R, L):
        out = O.reconstruct(w, x, z0, R, L, lr=10.0, momentum=0.7, arch=spec[1], dtype=np.float64)
        # every restart's G(z_{L-1}): the R = 1 call on the tiled images
        tiled = O.reconstruct(w, np.repeat(x, R, axis=0), z0, 1, L, lr=10.0, momentum=0.7, arch=spec[1], dtype=np.float64)
        return {"rec": out["rec"], "idx": out["idx"], "loss": out["loss"], "rows": tiled["rec"]}
    _compare(f, spec, rec, tol_scale=1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("path", _fixtures() or [None])
def test_engine_against_the_reference_run_under_tensorflow(path):
    if path is None:
        pytest.skip(NO_FIXTURES)
    from defensegan_amd.gan import dataset_gan_dict
    f, spec, w = _case(path)
    name, arch, wseed, gain, bias_range, B, R, adv, zseed = spec

    def rec(x, z0, R_, L):
        g = dataset_gan_dict[arch](cfg={"USE_BN": False, "LATENT_DIM": 128, "NET_DIM": 64}, test_mode=True, rec_rr=R_, rec_iters=L, rec_lr=10.0)
        assert g.set_weights(w) == []
        d = g.reconstruct(x, z_init_val=z0, return_details=True)
        g1 = dataset_gan_dict[arch](cfg={"USE_BN": False, "LATENT_DIM": 128, "NET_DIM": 64}, test_mode=True, rec_rr=1, rec_iters=L, rec_lr=10.0)
        assert g1.set_weights(w) == []
        rows = g1.reconstruct(np.repeat(x, R_, axis=0), z_init_val=z0)
        return {"rec": np.asarray(d["rec"]), "idx": np.asarray(d["idx"]), "loss": np.asarray(d["loss"]), "rows": np.asarray(rows)}
    _compare(f, spec, rec, tol_scale=1.5)


@pytest.mark.parametrize("ckpt", sorted(glob.glob(os.path.join(GOLDEN, "tf_ckpt", "*"))) or [None])
def test_checkpoint_reader_against_a_real_saver_file(ckpt):
    if ckpt is None:
        pytest.skip(NO_FIXTURES)
    from defensegan_amd import tf_checkpoint
    with np.load(os.path.join(ckpt, "weights.npz")) as f:
        want = {k: f[k] for k in f.files}
    got = tf_checkpoint.generator_weights(ckpt, want.keys())
    assert sorted(got) == sorted(want)
    for k in want:
        assert np.array_equal(np.asarray(got[k]).reshape(want[k].shape), want[k]), k
