"""-m gpu: the callers' batches (the reference's BATCH_SIZE 50, experiments/cfgs/gans/default.yml:2) coalesced into large engine
calls by ``model_eval_gan`` / ``reconstruct_dataset`` (/root/reference/utils/gan_defense.py:113-162, models/gan.py:504-557)
against one engine call per caller batch: same reconstructions, predictions, differences and cache files, bit for bit."""
import os

import numpy as np
import pytest

from defensegan_amd import gan_defense as gd, network_builder as nb, synth
from tests.helpers import clean_targets, make_gan

pytestmark = pytest.mark.gpu


def test_coalesced_evaluation_and_cache_equal_the_per_batch_loop(tmp_path, monkeypatch):
    R, L, N, BS = 10, 4, 230, 50                    # 5 caller batches, the last one ragged (30 images)
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=L)
    x, _ = clean_targets(p, "mnist", N, seed=71)
    x = synth.adversarial(x, 0.3, 0.0, 1.0, seed=72)
    y = (np.arange(N) * 3 % 10).astype(np.int64)
    clf = nb.model_a()
    clf.init_like_reference(seed=5)
    sizes = []
    real = gan.reconstruct

    class Spy(object):                               # a bound method of an engine model, as the callers pass it
        use_bn, rec_rr = gan.use_bn, gan.rec_rr

        def reconstruct(self, images, **kw):
            sizes.append(len(images))
            return real(images, **kw)
    spy = Spy()
    assert gd.rows_are_independent(gan.reconstruct) and gd.rows_are_independent(spy.reconstruct)
    per = gd.model_eval_gan(spy.reconstruct, clf, x, y, batch_size=BS, rec_rr=R, seed=5, first_image=1000, coalesce=False)
    assert sizes == [50, 50, 50, 50, 30]
    del sizes[:]
    co = gd.model_eval_gan(spy.reconstruct, clf, x, y, batch_size=BS, rec_rr=R, seed=5, first_image=1000)
    assert sizes == [230]
    del sizes[:]
    some = gd.model_eval_gan(spy.reconstruct, clf, x, y, batch_size=BS, rec_rr=R, seed=5, first_image=1000, coalesce=120)
    assert sizes == [100, 100, 30]
    for got in (co, some):
        assert got[0] == per[0] and got[1] == per[1]
        for a, b in zip(got[2], per[2]):
            assert np.array_equal(a, b)
    # --same_init: every caller batch restarts from the head of one z block, inside a coalesced call too
    zi = synth.make_z(BS * R, 128, seed=9)
    a = gd.model_eval_gan(gan.reconstruct, clf, x, y, batch_size=BS, rec_rr=R, same_init_z=zi, coalesce=False)
    b = gd.model_eval_gan(gan.reconstruct, clf, x, y, batch_size=BS, rec_rr=R, same_init_z=zi)
    assert a[0] == b[0] and all(np.array_equal(u, v) for u, v in zip(a[2], b[2]))

    # the reconstruction cache: runs of caller batches against one engine call per batch -- the same bytes in every pickle
    targets = y
    out = {}
    for name, rows in (("per_batch", 1), ("coalesced", gd.COALESCE_ROWS)):
        monkeypatch.setattr(gd, "COALESCE_ROWS", rows)
        ck = str(tmp_path / name)
        rec = gan.reconstruct_dataset({"test": (x, targets)}, ck, batch_size=BS, seed=77)["test"][0]
        d = os.path.join(ck, os.listdir(ck)[0], "test", "pickles")
        out[name] = (rec, {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))})
    assert np.array_equal(out["per_batch"][0], out["coalesced"][0])
    assert len(out["coalesced"][1]) == N and out["per_batch"][1] == out["coalesced"][1]


def test_command_line_and_defended_classifier_coalesce_like_the_evaluation(tmp_path):
    """``python -m defensegan_amd`` and ``MLP.model_eval`` (the evaluation path of a classifier with ``add_rec_model``) hand runs of
    whole --batch_size batches to the engine; ``--no_coalesce`` / ``coalesce=False`` is the reference's loop (one call per batch,
    whitebox.py / blackbox.py feeding BATCH_SIZE = 50 images per session.run): the same bits, ``--same_init`` included."""
    from defensegan_amd import __main__ as cli
    R, L, N, BS = 10, 3, 130, 50
    p = synth.make_weights("mnist", seed=1234, gain=2.0, bias_range=0.1)
    x, _ = clean_targets(p, "mnist", N, seed=81)
    x = synth.adversarial(x, 0.3, 0.0, 1.0, seed=82)
    pack, inp = str(tmp_path / "generator.npz"), str(tmp_path / "x.npy")
    np.savez(pack, **p)
    np.save(inp, x)
    outs = {}
    for name, extra in (("co", []), ("per", ["--no_coalesce"]), ("co_same", ["--same_init"]), ("per_same", ["--same_init", "--no_coalesce"])):
        out = str(tmp_path / (name + ".npy"))
        assert cli.main(["--cfg", "mnist", "--init_path", pack, "--input", inp, "--output", out, "--rec_rr", str(R), "--rec_iters", str(L),
                         "--batch_size", str(BS), "--seed", "5"] + extra) == 0
        outs[name] = np.load(out)
    assert np.isfinite(outs["co"]).all() and np.abs(outs["co"] - x).mean() < 0.3
    assert np.array_equal(outs["co"], outs["per"]) and np.array_equal(outs["co_same"], outs["per_same"])
    assert not np.array_equal(outs["co"], outs["co_same"])

    # a defended classifier: model_eval == the loop over fprop-sized batches
    gan, _ = make_gan("mnist", gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=L)
    y = (np.arange(N) * 7 % 10).astype(np.int64)
    clf = nb.model_a()
    clf.init_like_reference(seed=5)
    zi = synth.make_z(BS * R, 128, seed=9)
    clf.add_rec_model(gan, zi, BS)
    got = clf.model_eval(x, y, BS)
    preds = []
    for b0 in range(0, N, BS):                      # the reference's loop: one fprop (one session.run) per batch
        preds.append(np.asarray(clf.get_probs(x[b0:b0 + BS])).argmax(axis=1))
    preds = np.concatenate(preds)
    assert np.array_equal(got[2][1], preds) and got[0] == int((preds == y).sum()) and got[1] == N
