"""SURVEY.md 8f-N2, the step after the projection: classifier forward + model_eval_gan's per-batch reduction.
CPU: the oracle's conv semantics pinned against torch conv2d with explicit asymmetric SAME padding and hand-computed
shapes; layer plumbing of the host mirror.  -m gpu: the HIP path (through the C ABI) against the oracle."""
import numpy as np
import pytest

from defensegan_amd import network_builder as nb
from oracle import classifier_oracle as CO


def _layers_of(model):
    out = []
    for l in model.layers:
        if isinstance(l, nb.Conv2D):
            out.append(("conv", l.output_channels, l.kernel_shape, l.strides, l.padding))
        elif isinstance(l, nb.Linear):
            out.append(("linear", l.num_hid))
        else:
            out.append((l.__class__.__name__.lower(),))
    return out


def test_same_padding_known_answers():
    # tf.nn.conv2d SAME: out = ceil(in / s), pad_total = max((out-1)*s + k - in, 0), pad_before = pad_total // 2
    assert CO.same_padding(28, 8, 2) == (14, 3, 3)       # model B/F first layer
    assert CO.same_padding(28, 5, 1) == (28, 2, 2)       # model A first layer
    assert CO.same_padding(28, 5, 2) == (14, 1, 2)       # odd total: the extra row goes AFTER
    assert CO.same_padding(7, 3, 2) == (4, 1, 1)
    assert CO.same_padding(5, 1, 3) == (2, 0, 0)


@pytest.mark.parametrize("H,k,s,pad", [(28, 8, 2, "SAME"), (28, 5, 2, "SAME"), (14, 6, 2, "VALID"), (9, 3, 1, "SAME"), (11, 5, 3, "VALID")])
def test_oracle_conv_matches_torch(H, k, s, pad):
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(H * 10 + k)
    x = rs.standard_normal((3, H, H + 1, 4))
    K = rs.standard_normal((k, k, 4, 5))
    b = rs.standard_normal(5)
    y = CO.conv2d(x, K, b, (s, s), pad)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    if pad == "SAME":
        _, pt, pb = CO.same_padding(H, k, s)
        _, pl, pr = CO.same_padding(H + 1, k, s)
        xt = F.pad(xt, (pl, pr, pt, pb))
    yt = F.conv2d(xt, torch.from_numpy(K).permute(3, 2, 0, 1), torch.from_numpy(b), stride=s).permute(0, 2, 3, 1).numpy()
    assert y.shape == yt.shape
    np.testing.assert_allclose(y, yt, rtol=1e-12, atol=1e-12)


def test_model_zoo_shapes_and_names():
    # network_builder.py:333-521: flatten widths of the MNIST models
    widths = {}
    for name, fn in nb.MODELS.items():
        m = fn()
        shape = (28, 28, 1)
        for l in m.layers:
            if isinstance(l, nb.Conv2D):
                shape = nb.conv_output_shape(shape, l)
        widths[name] = int(np.prod(shape))
        assert m.layer_names[-1] == "probs" and m.layer_names[-2] == "logits"
    assert widths["A"] == 12 * 12 * 64 and widths["B"] == 1 * 1 * 128 and widths["F"] == 128 and widths["C"] == 12 * 12 * 64
    assert widths["D"] == 784 and widths["E"] == 784


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(nb.MODELS))
def test_classifier_forward_vs_oracle(name):
    m = nb.MODELS[name]()
    params = m.init_like_reference(seed=ord(name))
    rs = np.random.RandomState(3)
    params = [(W, rs.uniform(-0.2, 0.2, size=b.shape).astype(np.float32)) for W, b in params]     # exercise the bias path
    m.set_weights(params)
    x = rs.uniform(0, 1, size=(7, 28, 28, 1)).astype(np.float32)
    out = m.fprop(x)
    lo, po = CO.forward(_layers_of(m), [(W.astype(np.float64), b.astype(np.float64)) for W, b in params], x.astype(np.float64))
    assert out["logits"].shape == (7, 10)
    np.testing.assert_allclose(out["logits"], lo, rtol=0, atol=2e-5 * max(1.0, np.abs(lo).max()))
    np.testing.assert_allclose(out["probs"], po, rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["probs"].sum(axis=1), 1.0, atol=1e-6)
    assert np.array_equal(m(x), out["probs"]) and np.array_equal(m.get_logits(x), out["logits"])


@pytest.mark.gpu
def test_color_input_and_eval_batch_vs_oracle():
    m = nb.model_a(nb_filters=16, nb_classes=7, input_shape=(None, 64, 64, 3))
    params = m.init_like_reference(seed=5)
    rs = np.random.RandomState(9)
    B = 37
    rec = rs.uniform(-1, 1, size=(B, 64, 64, 3)).astype(np.float32)
    orig = np.clip(rec + 0.1 * rs.standard_normal(rec.shape), -1, 1).astype(np.float32)
    _, po = CO.forward(_layers_of(m), [(W.astype(np.float64), b.astype(np.float64)) for W, b in params], rec.astype(np.float64))
    labels = po.argmax(axis=1).astype(np.int32)
    labels[::5] = (labels[::5] + 1) % 7                      # some wrong on purpose
    n_ok, preds, diffs = m.eval_batch(rec, orig, labels)
    want_ok, want_preds, want_diffs = CO.eval_batch(po, labels, rec.astype(np.float64), orig.astype(np.float64))
    assert n_ok == want_ok and np.array_equal(preds.cpu().numpy(), want_preds)
    np.testing.assert_allclose(diffs.cpu().numpy(), want_diffs, rtol=2e-6)
    n2, p2, d2 = m.eval_batch(rec)                           # no labels / originals
    assert n2 == 0 and d2 is None and np.array_equal(p2.cpu().numpy(), want_preds)


@pytest.mark.gpu
def test_defended_evaluation_end_to_end_on_device():
    """cfg-5 pipeline shape on one rank: x -> Defense-GAN projection -> classifier -> (accuracy, roc_info), ragged batches,
    through the same model_eval_gan harness the reference drives (gan_defense.py:113-179)."""
    from defensegan_amd import gan_defense
    from tests.helpers import clean_targets, make_gan
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.0, rec_rr=3, rec_iters=20)
    x, _ = clean_targets(p, "mnist", 23, seed=4)
    clf = nb.model_f(nb_filters=8)
    clf.init_like_reference(seed=1)
    labels = clf(x).argmax(axis=1)                             # the undefended predictions as "ground truth"
    correct, n, roc = gan_defense.model_eval_gan(gan.reconstruct, clf, x, labels, batch_size=10, rec_rr=3, seed=2)
    assert n == 23 and roc[0].shape == roc[1].shape == roc[2].shape == (23,)
    assert correct == int((roc[1] == labels).sum()) and (roc[2] >= 0).all() and np.isfinite(roc[2]).all()
    # clean in-range targets are reconstructed almost exactly, so the defended classifier agrees with the undefended one
    assert correct >= 21
    # the classifier wrapped around the projection (whitebox.py:185 add_rec_model) gives the same predictions
    clf.add_rec_model(gan, None, 23)
    out = clf.fprop(x)
    assert "reconstruction" in out and out["reconstruction"].shape == x.shape
    assert (out["probs"].argmax(axis=1) == labels).mean() > 0.9


# ------------------------------------------------------------------------------------------------ N3: FGSM
def test_oracle_input_gradient_finite_differences():
    m = nb.MLP([nb.Conv2D(3, (3, 3), (2, 2), "SAME"), nb.ReLU(), nb.Conv2D(4, (2, 2), (1, 1), "VALID"), nb.ReLU(), nb.Flatten(),
                nb.Linear(6), nb.ReLU(), nb.Dropout(0.5), nb.Linear(5), nb.Softmax()], (None, 7, 6, 2))
    layers = _layers_of(m)
    rs = np.random.RandomState(0)
    params = [(rs.standard_normal((3, 3, 2, 3)), rs.standard_normal(3)), (rs.standard_normal((2, 2, 3, 4)), rs.standard_normal(4)),
              (rs.standard_normal((4 * 3 * 2, 6)) if False else rs.standard_normal((24, 6)), rs.standard_normal(6)),
              (rs.standard_normal((6, 5)), rs.standard_normal(5))]
    x = rs.standard_normal((2, 7, 6, 2))
    lo, _ = CO.forward(layers, params, x)
    assert lo.shape == (2, 5)
    labels = np.array([1, 3])

    def loss(xx):
        l, p = CO.forward(layers, params, xx)
        return -np.log(p[np.arange(2), labels]).sum()
    g = CO.input_gradient(layers, params, x, labels)
    num = np.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        d = np.zeros_like(x); d[idx] = 1e-6
        num[idx] = (loss(x + d) - loss(x - d)) / 2e-6
    np.testing.assert_allclose(g, num, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["A", "B", "E"])
def test_input_gradient_and_fgsm_vs_oracle(name):
    m = nb.MODELS[name]()
    params = m.init_like_reference(seed=11)
    rs = np.random.RandomState(12)
    params = [(W, rs.uniform(-0.1, 0.1, size=b.shape).astype(np.float32)) for W, b in params]
    m.set_weights(params)
    x = rs.uniform(0, 1, size=(6, 28, 28, 1)).astype(np.float32)
    labels = rs.randint(0, 10, size=6).astype(np.int32)
    p64 = [(W.astype(np.float64), b.astype(np.float64)) for W, b in params]
    for lab in (labels, None):
        g = m.input_gradient(x, lab)
        xo, go = CO.fgsm(_layers_of(m), p64, x.astype(np.float64), 0.3, 0.0, 1.0, lab)
        np.testing.assert_allclose(g, go, rtol=0, atol=2e-5 * np.abs(go).max())
        xa = nb.FastGradientMethod(m).generate(x, eps=0.3, clip_min=0.0, clip_max=1.0, y=lab)
        assert xa.min() >= 0.0 and xa.max() <= 1.0 and np.abs(xa - x).max() <= 0.3 + 1e-6
        decided = np.abs(go) > 1e-4 * np.abs(go).max()           # sign() is only comparable away from 0
        assert decided.mean() > 0.5
        np.testing.assert_allclose(xa[decided], xo[decided], rtol=0, atol=1e-6)
    # one-hot labels as cleverhans takes them
    onehot = np.eye(10, dtype=np.float32)[labels]
    assert np.array_equal(nb.FastGradientMethod(m).generate(x, eps=0.3, clip_min=0.0, clip_max=1.0, y=onehot),
                          nb.FastGradientMethod(m).generate(x, eps=0.3, clip_min=0.0, clip_max=1.0, y=labels))


@pytest.mark.gpu
def test_attack_then_defend_pipeline_runs_on_device():
    """blackbox.py:521-575 in miniature: FGSM inputs from a substitute, projection, black-box classifier, reduction."""
    from defensegan_amd import gan_defense
    from tests.helpers import clean_targets, make_gan
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.0, rec_rr=2, rec_iters=10)
    x, _ = clean_targets(p, "mnist", 12, seed=6)
    sub, bbox = nb.model_e(), nb.model_b(nb_filters=8)
    sub.init_like_reference(seed=3)
    bbox.init_like_reference(seed=4)
    x_adv = nb.FastGradientMethod(sub).generate(x, eps=0.3, clip_min=0.0, clip_max=1.0)
    assert x_adv.shape == x.shape and 0.05 < np.abs(x_adv - x).max() <= 0.3 + 1e-6
    labels = bbox(x).argmax(axis=1)
    c, n, roc = gan_defense.model_eval_gan(gan.reconstruct, bbox, x_adv, labels, batch_size=5, rec_rr=2, seed=1)
    assert n == 12 and 0 <= c <= 12 and (roc[2] > 0).all()       # adversarial inputs are off the generator's range


@pytest.mark.gpu
def test_classifier_errors_are_reported_through_dg_last_error():
    from defensegan_amd import _native
    m = nb.model_e()
    with pytest.raises(_native.NativeError):
        m(np.zeros((2, 28, 28, 1), np.float32))                     # weights not set
    with pytest.raises(_native.NativeError, match=r"Linear W must be \[784,200\]"):
        m.set_weights([(np.zeros((10, 200), np.float32), np.zeros(200, np.float32))] * 3)
    bad = nb.MLP([nb.Linear(10)], (None, 28, 28, 1))               # Linear without Flatten
    with pytest.raises(_native.NativeError, match="Flatten"):
        bad._ensure()


@pytest.mark.gpu
def test_whitebox_fgsm_through_the_defense_reproduces_the_reference():
    """whitebox.py:185-223 in miniature (BASELINE configs[4]): the reconstruction layer is attached FIRST, then FGSM is built
    on that model.  The reference's gradient through ReconstructionLayer is identically zero, so
    ``adv_x == clip(x)``, ``diff_op = mean((adv_x - x)^2) == 0`` for in-range inputs, and the evaluated predictions are those
    of classifier(reconstruct(x)).  The bare-classifier attack stays available through no_rec / building it earlier."""
    from defensegan_amd import gan_defense
    from tests.helpers import clean_targets, make_gan
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.0, rec_rr=2, rec_iters=6)
    x, _ = clean_targets(p, "mnist", 10, seed=16)
    x[0, :2, :2, 0] = [[1.4, -0.3], [0.5, 2.0]]                    # out-of-range pixels: only the clip acts on them
    model = nb.model_a(nb_filters=8)
    model.init_like_reference(seed=5)
    bare_attack = nb.FastGradientMethod(model).generate(x, eps=0.3, clip_min=0.0, clip_max=1.0)
    model.add_rec_model(gan, None, 5)
    with pytest.warns(UserWarning, match="whitebox.py:185-214"):
        adv = nb.FastGradientMethod(model).generate(x, eps=0.3, ord=np.inf, clip_min=0.0, clip_max=1.0)
    assert np.array_equal(adv, np.clip(x, 0.0, 1.0))
    assert np.abs(bare_attack - np.clip(x, 0, 1)).max() > 0.25       # the undefended model does yield a real attack
    with pytest.warns(UserWarning):
        assert not np.asarray(model.input_gradient(x)).any()
    assert np.asarray(model.input_gradient(x, no_rec=True)).any()
    # evaluation as whitebox.py:214-223: predictions of the defended model on adv_x, diffs = mean((adv_x - x)^2)
    labels = model.fprop(adv, no_rec=True)["logits"].argmax(axis=1)
    clf = lambda im: model.fprop(im, no_rec=True)["probs"]           # the harness applies the projection itself
    c, n, roc = gan_defense.model_eval_gan(gan.reconstruct, clf, adv, labels, batch_size=5, rec_rr=2, seed=3)
    assert n == 10 and 0 <= c <= 10
    diff_op = ((adv - x) ** 2).reshape(10, -1).mean(axis=1)
    assert (diff_op[1:] == 0).all() and diff_op[0] > 0


def test_rand_fgsm_prestep_is_the_reference_expression():
    """whitebox.py:191-195: clip(x + alpha * sign(randn), min_val, 1) and eps - alpha, from the same generator state."""
    from defensegan_amd import network_builder as nb
    x = np.random.RandomState(0).uniform(-1, 1, size=(5, 8, 8, 3)).astype(np.float32)
    got, eps = nb.rand_fgsm_prestep(x, eps=0.3, alpha=0.05, min_val=-1.0, rng=np.random.RandomState([11, 24, 1990]))
    rs = np.random.RandomState([11, 24, 1990])
    want = np.clip(x + 0.05 * np.sign(rs.randn(*x.shape)), -1.0, 1.0)
    assert abs(eps - 0.25) < 1e-12 and got.dtype == np.float32 and np.allclose(got, want, atol=1e-7)
    assert np.abs(got - x).max() <= 0.05 + 1e-6 and got.min() >= -1.0 and got.max() <= 1.0
