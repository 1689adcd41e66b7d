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
    assert {c[0] for c in kit.CASES if c[9]} == {"mnist_bn", "celeba_bn"} and any(c[1] == "celeba" and not c[9] for c in kit.CASES)
    for name, arch, wseed, gain, bias_range, B, R, adv, zseed, use_bn in kit.CASES:
        w, names = kit.make_weights(arch, wseed, gain, bias_range, use_bn)
        want = synth.make_weights(arch, seed=wseed, gain=gain, bias_range=bias_range, use_bn=use_bn, bn_jitter=kit.BN_JITTER if use_bn else 0.0)
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
    w, _ = kit.make_weights(spec[1], spec[2], spec[3], spec[4], spec[9])
    return f, spec, w


NO_FIXTURES = ("no TensorFlow fixtures committed: run `python2 tools/make_tf_fixtures.py --reference <checkout of kabkabm/defensegan> "
               "--out tests/golden` on a Python-2.7 / TensorFlow-1.7 machine (the build image has neither); one run pins the three "
               "regimes -- no Batchnorm, USE_BN True, CelebA -- and the checkpoint reader")


def _compare(f, spec, reconstruct, tol_scale):
    """reconstruct(x, z0, R, L) -> dict(rec [B..], idx [B], loss [B*R], rows [B*R..]) against the reference's outputs."""
    name, arch, wseed, gain, bias_range, B, R, adv, zseed, use_bn = spec
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
        out = O.reconstruct(w, x, z0, R, L, lr=10.0, momentum=0.7, arch=spec[1], dtype=np.float64, use_bn=bool(spec[9]))
        # every restart's G(z_{L-1}): the R = 1 call on the tiled images (the same B * R rows: the same batch statistics)
        tiled = O.reconstruct(w, np.repeat(x, R, axis=0), z0, 1, L, lr=10.0, momentum=0.7, arch=spec[1], dtype=np.float64, use_bn=bool(spec[9]))
        return {"rec": out["rec"], "idx": out["idx"], "loss": out["loss"], "rows": tiled["rec"]}
    _compare(f, spec, rec, tol_scale=1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("path", _fixtures() or [None])
def test_engine_against_the_reference_run_under_tensorflow(path):
    if path is None:
        pytest.skip(NO_FIXTURES)
    from defensegan_amd.gan import dataset_gan_dict
    f, spec, w = _case(path)
    name, arch, wseed, gain, bias_range, B, R, adv, zseed, use_bn = spec

    def rec(x, z0, R_, L):
        g = dataset_gan_dict[arch](cfg={"USE_BN": bool(use_bn), "LATENT_DIM": 128, "NET_DIM": 64}, test_mode=True, rec_rr=R_, rec_iters=L, rec_lr=10.0)
        assert g.set_weights(w) == []
        d = g.reconstruct(x, z_init_val=z0, return_details=True)
        g1 = dataset_gan_dict[arch](cfg={"USE_BN": bool(use_bn), "LATENT_DIM": 128, "NET_DIM": 64}, test_mode=True, rec_rr=1, rec_iters=L, rec_lr=10.0)
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


# ---- a committed fixture must be USABLE: a present-but-malformed file fails (it must not look like "no fixtures: skipped") ----
def validate_fixture(path):
    """Raises AssertionError naming what is wrong with a tests/golden/tf_<case>.npz: unknown case name, missing arrays, shapes
    that do not fit the case, or inputs that are not the ones the kit generates for that case (stale kit / edited file)."""
    kit = _kit()
    name = os.path.basename(path)[3:-4]
    specs = [c for c in kit.CASES if c[0] == name]
    assert specs, "%s: no case '%s' in tools/make_tf_fixtures.py CASES" % (path, name)
    _, arch, wseed, gain, bias_range, B, R, adv, zseed, use_bn = specs[0]
    try:
        f = np.load(path)
        files = set(f.files)
    except Exception as e:                                                   # not an npz at all
        raise AssertionError("%s: unreadable (%s)" % (path, e))
    for k in ("x", "z0", "iters"):
        assert k in files, "%s: array '%s' missing" % (path, k)
    dim = kit.image_dim(arch)
    assert list(f["x"].shape) == [B] + dim, "%s: x has shape %r, case %s needs %r" % (path, f["x"].shape, name, [B] + dim)
    assert f["z0"].shape == (B * R, kit.LATENT), "%s: z0 has shape %r" % (path, f["z0"].shape)
    zt, z0 = kit.make_latents(zseed, B, R)
    assert np.array_equal(f["z0"], z0), "%s: z0 is not the kit's draw for this case (stale kit or edited fixture)" % path
    iters = [int(v) for v in f["iters"]]
    assert iters and all(v >= 1 for v in iters), "%s: iters = %r" % (path, iters)
    P = int(np.prod(dim))
    for L in iters:
        for k, shape in (("rec_%d" % L, B * P), ("rows_%d" % L, B * R * P), ("loss_%d" % L, B * R), ("idx_%d" % L, B)):
            assert k in files, "%s: array '%s' missing" % (path, k)
            assert f[k].size == shape, "%s: '%s' has %d values, expected %d" % (path, k, f[k].size, shape)
        assert np.isfinite(f["rows_%d" % L]).all() and np.isfinite(f["loss_%d" % L]).all(), "%s: non-finite outputs at L = %d" % (path, L)
        idx = np.asarray(f["idx_%d" % L]).reshape(B)
        assert ((idx >= 0) & (idx < R)).all(), "%s: idx_%d outside [0, R)" % (path, L)


@pytest.mark.parametrize("path", _fixtures() or [None])
def test_committed_tf_fixtures_are_well_formed(path):
    if path is None:
        pytest.skip(NO_FIXTURES)
    validate_fixture(path)


def test_a_malformed_fixture_fails_instead_of_skipping(tmp_path):
    """The validator the committed fixtures go through rejects: an unknown case name, a truncated file, missing arrays, a z0
    that is not the kit's draw -- each with an AssertionError (a failure), never a skip."""
    kit = _kit()
    name, arch, wseed, gain, bias_range, B, R, adv, zseed, use_bn = kit.CASES[0]
    zt, z0 = kit.make_latents(zseed, B, R)
    P = int(np.prod(kit.image_dim(arch)))
    good = {"x": np.zeros([B] + kit.image_dim(arch), np.float32), "z0": z0, "iters": np.array([1]),
            "rec_1": np.zeros((B, P), np.float32), "rows_1": np.zeros((B * R, P), np.float32), "loss_1": np.zeros(B * R, np.float32),
            "idx_1": np.zeros(B, np.int64)}
    ok = str(tmp_path / ("tf_%s.npz" % name))
    np.savez(ok, **good)
    validate_fixture(ok)                                                     # the well-formed one passes
    bad = []
    p1 = str(tmp_path / "tf_no_such_case.npz"); np.savez(p1, **good); bad.append(p1)
    p2 = str(tmp_path / "a" / ("tf_%s.npz" % name)); os.makedirs(os.path.dirname(p2)); open(p2, "wb").write(open(ok, "rb").read()[:100]); bad.append(p2)
    p3 = str(tmp_path / "b" / ("tf_%s.npz" % name)); os.makedirs(os.path.dirname(p3)); np.savez(p3, **{k: v for k, v in good.items() if k != "rows_1"}); bad.append(p3)
    p4 = str(tmp_path / "c" / ("tf_%s.npz" % name)); os.makedirs(os.path.dirname(p4)); np.savez(p4, **dict(good, z0=z0 + 1)); bad.append(p4)
    p5 = str(tmp_path / "d" / ("tf_%s.npz" % name)); os.makedirs(os.path.dirname(p5)); np.savez(p5, **dict(good, idx_1=np.full(B, R))); bad.append(p5)
    for q in bad:
        with pytest.raises(AssertionError):
            validate_fixture(q)


def test_the_kit_is_python_2_7_syntax():
    """tools/make_tf_fixtures.py must run UNMODIFIED under Python 2.7 (the reference's interpreter; not installed here, so it
    cannot be py_compiled by one): its syntax tree holds none of the constructs Python 2.7 rejects, it imports print_function,
    and it compiles here with every warning an error."""
    import ast
    import tokenize
    import warnings
    path = os.path.join(ROOT, "tools", "make_tf_fixtures.py")
    src = open(path).read()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tree = ast.parse(src, path)
        compile(src, path, "exec")
    py3_only = (ast.JoinedStr, ast.AnnAssign, ast.Nonlocal, ast.YieldFrom, ast.AsyncFunctionDef, ast.AsyncFor, ast.AsyncWith,
                ast.Await, ast.MatMult, ast.NamedExpr)
    bad = []
    future_print = False
    for node in ast.walk(tree):
        if isinstance(node, py3_only):
            bad.append((type(node).__name__, node.lineno))
        if isinstance(node, (ast.FunctionDef, ast.Lambda)):
            a = node.args
            if a.kwonlyargs or getattr(a, "posonlyargs", None):
                bad.append(("keyword-only / positional-only arguments", node.lineno))
            if isinstance(node, ast.FunctionDef) and (node.returns is not None or any(x.annotation is not None for x in a.args)):
                bad.append(("annotations", node.lineno))
        if isinstance(node, ast.Raise) and node.cause is not None:
            bad.append(("raise ... from", node.lineno))
        if isinstance(node, ast.Call):
            if isinstance(node.func, ast.Name) and node.func.id == "super" and not node.args:
                bad.append(("zero-argument super()", node.lineno))
            if sum(isinstance(x, ast.Starred) for x in node.args) > 1 or sum(k.arg is None for k in node.keywords) > 1:
                bad.append(("PEP 448 unpacking in a call", node.lineno))
            if isinstance(node.func, ast.Name) and node.func.id == "open" and any(k.arg in ("encoding", "newline") for k in node.keywords):
                bad.append(("open(encoding=)", node.lineno))
        if isinstance(node, ast.Dict) and any(k is None for k in node.keys):
            bad.append(("{**d}", node.lineno))
        if isinstance(node, (ast.List, ast.Tuple, ast.Set)) and isinstance(getattr(node, "ctx", None), ast.Load) and any(isinstance(x, ast.Starred) for x in node.elts):
            bad.append(("[*a]", node.lineno))
        if isinstance(node, ast.Assign) and any(isinstance(t, (ast.Tuple, ast.List)) and any(isinstance(e, ast.Starred) for e in t.elts) for t in node.targets):
            bad.append(("a, *b = ...", node.lineno))
        if isinstance(node, ast.ClassDef) and node.keywords:
            bad.append(("class keywords (metaclass=)", node.lineno))
        if isinstance(node, ast.ImportFrom) and node.module == "__future__" and any(n.name == "print_function" for n in node.names):
            future_print = True
        if isinstance(node, ast.Name) and node.id in ("FileNotFoundError", "PermissionError", "ModuleNotFoundError", "nonlocal"):
            bad.append((node.id, node.lineno))
    assert future_print, "from __future__ import print_function is missing"
    with open(path, "rb") as fh:
        for tok in tokenize.tokenize(fh.readline):
            if tok.type == tokenize.NUMBER and "_" in tok.string:
                bad.append(("1_000 literal", tok.start[0]))
            if tok.type == tokenize.STRING and tok.string[:2].lower() in ("f'", 'f"', "rb", "br") and tok.string[0].lower() == "f":
                bad.append(("f-string", tok.start[0]))
            if tok.type == tokenize.OP and tok.string in ("->", ":=", "@="):
                bad.append((tok.string, tok.start[0]))
    assert not bad, bad


def test_the_batchnorm_case_compares_like_the_others_on_a_stand_in_fixture():
    """The USE_BN cases of the kit exercise two things the others do not: Batchnorm parameters in the weight stream, and the fact
    that the R = 1 call on the R-times tiled images (``rows_L``) sees the SAME B * R rows -- hence the same batch statistics
    (tflib/ops/batchnorm.py:80-93) -- as the rec_rr = R call.  No TensorFlow here, so the comparison code is run on a STAND-IN
    fixture made by the float32 oracle in the kit's format (this checks the plumbing, not the reference: parity stays unpinned)."""
    from oracle import defensegan_oracle as O
    kit = _kit()
    spec = [c for c in kit.CASES if c[0] == "mnist_bn"][0]
    name, arch, wseed, gain, bias_range, B, R, adv, zseed, use_bn = spec
    w, _ = kit.make_weights(arch, wseed, gain, bias_range, use_bn)
    zt, z0 = kit.make_latents(zseed, B, R)
    x = O.generator_forward(w, np.repeat(zt, R, axis=0), arch, True)[0][::R].astype(np.float32)      # any in-range images do
    f = {"x": x, "z0": z0, "iters": np.array([1, 3])}
    for L in (1, 3):
        o = O.reconstruct(w, x, z0, R, L, lr=10.0, momentum=0.7, arch=arch, dtype=np.float32, use_bn=True)
        t = O.reconstruct(w, np.repeat(x, R, axis=0), z0, 1, L, lr=10.0, momentum=0.7, arch=arch, dtype=np.float32, use_bn=True)
        rows = t["rec"].reshape(B * R, -1).astype(np.float64)
        loss = ((rows - np.repeat(x.reshape(B, -1).astype(np.float64), R, axis=0)) ** 2).mean(axis=1)
        f.update({"rec_%d" % L: o["rec"], "rows_%d" % L: t["rec"], "loss_%d" % L: loss, "idx_%d" % L: loss.reshape(B, R).argmin(axis=1)})
        assert np.allclose(o["loss"], loss, rtol=1e-4)            # the tiled call's rows ARE the restarts of the rec_rr = R call

    def rec(x_, z0_, R_, L):
        out = O.reconstruct(w, x_, z0_, R_, L, lr=10.0, momentum=0.7, arch=arch, dtype=np.float64, use_bn=True)
        tiled = O.reconstruct(w, np.repeat(x_, R_, axis=0), z0_, 1, L, lr=10.0, momentum=0.7, arch=arch, dtype=np.float64, use_bn=True)
        return {"rec": out["rec"], "idx": out["idx"], "loss": out["loss"], "rows": tiled["rec"]}
    _compare(f, spec, rec, tol_scale=1.0)
