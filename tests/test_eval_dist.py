"""Host logic of the eval harness and of the batch-sharded multi-process driver (gloo, world_size 2, CPU).
The reconstruct callable is a stand-in here (no GPU): what is under test is batching, sharding, the global
row index handed to the engine, and the single all_gather."""
import os
import socket

import numpy as np
import pytest

from defensegan_amd import config as cfgmod
from defensegan_amd import gan_defense as gd


def fake_reconstruct(calls):
    def f(x, seed=None, first_row=None, z_init_val=None):
        calls.append((len(x), first_row))
        return np.clip(np.asarray(x) * 0.5, 0, 1)
    return f


def classifier(x):
    s = np.asarray(x).reshape(len(x), -1).sum(axis=1)
    logits = np.stack([s, 40.0 - s], axis=1)
    return logits


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 10000, 10007):
        for w in (1, 2, 3, 8):
            r = [gd.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert gd.shard_range(10000, 3, 8) == (3750, 5000)        # cfg 5: 8 x 1250 images


def test_model_eval_gan_batches_and_counts():
    rs = np.random.RandomState(0)
    x = rs.rand(23, 4, 4, 1).astype(np.float32) * 5
    y = rs.randint(0, 2, 23)
    calls = []
    c, n, roc = gd.model_eval_gan(fake_reconstruct(calls), classifier, x, y, batch_size=10, rec_rr=3)
    assert n == 23 and [k for k, _ in calls] == [10, 10, 3]          # ragged last batch
    assert [fr for _, fr in calls] == [0, 30, 60]                    # global row = image * R
    preds = classifier(np.clip(x * 0.5, 0, 1)).argmax(1)
    assert c == int((preds == y).sum())
    np.testing.assert_array_equal(roc[1], preds)
    np.testing.assert_allclose(roc[2], ((x - np.clip(x * .5, 0, 1)) ** 2).reshape(23, -1).mean(1), rtol=1e-6)
    onehot = np.eye(2)[y]
    c2, _, _ = gd.model_eval_gan(fake_reconstruct([]), classifier, x, onehot, batch_size=7, rec_rr=3)
    assert c2 == c


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(0)
        x = rs.rand(37, 4, 4, 1).astype(np.float32) * 5
        y = rs.randint(0, 2, 37)
        calls = []
        acc, roc = gd.model_eval_gan_sharded(fake_reconstruct(calls), classifier, x, y, batch_size=8, rec_rr=2)
        # the CLI's gather of the reconstructions themselves: contiguous shards -> the whole array on every rank
        s0, e0 = gd.shard_range(len(x), rank, world)
        whole = gd.gather_shards(x[s0:e0] * 0.5, len(x))
        assert whole.shape == x.shape and np.array_equal(whole, x * 0.5)
        q.put((rank, acc, roc[0].tolist(), roc[1].tolist(), roc[2].tolist(), calls))
    finally:
        dist.destroy_process_group()


def test_sharded_eval_two_ranks_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rs = np.random.RandomState(0)
    x = rs.rand(37, 4, 4, 1).astype(np.float32) * 5
    y = rs.randint(0, 2, 37)
    c, n, roc = gd.model_eval_gan(fake_reconstruct([]), classifier, x, y, batch_size=8, rec_rr=2)
    for rank, acc, labels, preds, diffs, calls in res:
        assert abs(acc - c / n) < 1e-12
        assert labels == roc[0].tolist() and preds == roc[1].tolist()
        np.testing.assert_allclose(diffs, roc[2], rtol=1e-6)
    # rank 0 got images [0,19), rank 1 [19,37): first_row is the GLOBAL row index
    assert [fr for _, fr in res[0][5]] == [0, 16, 32]
    assert [fr for _, fr in res[1][5]] == [38, 54, 70]


def test_config_surface():
    cfg = cfgmod.load_config(cfgmod.builtin_cfg("mnist"))
    assert cfg["REC_ITERS"] == 200 and cfg["REC_RR"] == 10 and cfg["REC_LR"] == 10.0
    assert cfg["LATENT_DIM"] == 128 and cfg["USE_BN"] is False and cfg["BATCH_SIZE"] == 50
    assert cfgmod.load_config(cfgmod.builtin_cfg("celeba"))["REC_RR"] == 2
    import argparse
    ap = cfgmod.add_rec_flags(argparse.ArgumentParser())
    a = ap.parse_args(["--rec_iters", "10", "--rec_rr", "1", "--batch_size", "50"])
    r = cfgmod.resolve_rec_params(cfg, a)
    assert r == {"rec_rr": 1, "rec_lr": 10.0, "rec_iters": 10, "batch_size": 50}
    # a rec_path wins over the CLI unless --override (whitebox.py:246-264)
    name = cfgmod.rec_dir_name(5, 1.0, 100)
    assert name == "recs_rr5_lr1.00000_iters100"
    a = ap.parse_args(["--rec_iters", "10", "--rec_path", "out/" + name])
    assert cfgmod.resolve_rec_params(cfg, a)["rec_iters"] == 100
    a = ap.parse_args(["--rec_iters", "10", "--rec_path", "out/" + name, "--override"])
    assert cfgmod.resolve_rec_params(cfg, a)["rec_iters"] == 10


def test_gan_object_surface_without_gpu():
    """Attribute surface of DefenseGANBase (gan.py:41-68); compute must fail loudly without a GPU."""
    import torch
    from defensegan_amd import _native
    from defensegan_amd.gan import dataset_gan_dict, gan_from_config
    gan = gan_from_config(cfgmod.builtin_cfg("mnist"))
    assert (gan.rec_iters, gan.rec_rr, gan.rec_lr, gan.latent_dim, gan.net_dim) == (200, 10, 10.0, 128, 64)
    assert gan.use_bn is False and gan.batch_size == 50 and gan.test_batch_size == 50
    assert gan.dataset_name == "mnist" and list(gan.image_dim) == [28, 28, 1]
    assert set(dataset_gan_dict) == {"mnist", "f-mnist", "celeba"}
    from defensegan_amd import synth
    assert gan.set_weights(synth.make_weights("mnist")) == [] and gan.initialized
    if not torch.cuda.is_available():
        with pytest.raises(_native.NativeError):
            gan.reconstruct(np.zeros((1, 28, 28, 1), np.float32))


def _assert_python2_era_numpy_can_unpickle(raw, n_objects):
    """The reference reads these files with cPickle under Python 2 and NumPy <= 1.16 (gan.py:489, 527): protocol <= 2 and
    every global must exist there -- ``numpy.core.multiarray._reconstruct`` / ``numpy.ndarray`` / ``numpy.dtype`` /
    ``_codecs.encode``; NumPy >= 2's own pickles name ``numpy._core.multiarray``, which that NumPy cannot import."""
    import io
    import pickle
    import pickletools
    assert b"numpy._core" not in raw and b"numpy.core.multiarray" in raw
    allowed = {("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"), ("numpy", "ndarray"),
               ("numpy", "dtype"), ("_codecs", "encode")}
    f = io.BytesIO(raw)
    for _ in range(n_objects):
        ops = list(pickletools.genops(f))
        assert ops[0][0].name == "PROTO" and ops[0][1] == 2
        for op, arg, _pos in ops:
            assert op.proto <= 2, op.name
            if op.name == "GLOBAL":
                assert tuple(arg.split(" ")) in allowed, arg
    assert f.read() == b""


def test_reconstruct_dataset_cache_layout(tmp_path):
    """reconstruct_dataset writes the reference's cache layout (gan.py:467-478, 504-557) and reloads it."""
    import pickle
    from defensegan_amd.gan import MnistDefenseGAN, ReconstructionLayer
    gan = MnistDefenseGAN(cfg={"USE_BN": False}, test_mode=True, rec_rr=2, rec_iters=5, rec_lr=10.0)
    calls = []

    def fake(images, batch_size=None, back_prop=True, reconstructor_id=0, z_init_val=None, seed=None, first_row=0, **kw):
        calls.append((len(images), first_row, reconstructor_id))
        return np.asarray(images) * 0.5
    gan.reconstruct = fake
    x = np.random.RandomState(0).rand(7, 28, 28, 1).astype(np.float32)
    y = np.arange(7) % 3
    out = gan.reconstruct_dataset({"test": (x, y)}, str(tmp_path), batch_size=3)
    d = tmp_path / "recs_rr2_lr10.00000_iters5" / "test" / "pickles"
    assert sorted(p.name for p in d.iterdir())[0] == "rec_0000000_l0.pkl" and len(list(d.iterdir())) == 7
    np.testing.assert_allclose(pickle.load(open(d / "rec_0000004_l1.pkl", "rb")), x[4] * 0.5)
    _assert_python2_era_numpy_can_unpickle((d / "rec_0000004_l1.pkl").read_bytes(), n_objects=1)
    # caller batches of 3 decide what is cached; what has to be computed goes to the engine in runs of consecutive batches
    # (rows are independent without Batchnorm): here all seven images in one call, global row 0
    assert [c[:2] for c in calls] == [(7, 0)]
    np.testing.assert_allclose(out["test"][0], x * 0.5)
    n_before = len(calls)
    again = gan.reconstruct_dataset({"test": (x, y)}, str(tmp_path), batch_size=3)       # served from the cache
    assert len(calls) == n_before
    np.testing.assert_allclose(again["test"][0], x * 0.5)
    # one missing pickle: exactly its caller batch (images 3..5, global row 6) is recomputed, the cached neighbours are not
    (d / "rec_0000004_l1.pkl").unlink()
    gan.reconstruct_dataset({"test": (x, y)}, str(tmp_path), batch_size=3)
    assert [c[:2] for c in calls[n_before:]] == [(3, 6)]
    # two neighbouring batches missing: one run; with Batchnorm the batch is the unit of the statistics: one call per batch
    (d / "rec_0000004_l1.pkl").unlink(); (d / "rec_0000006_l0.pkl").unlink()
    n_mid = len(calls)
    gan.reconstruct_dataset({"test": (x, y)}, str(tmp_path), batch_size=3)
    assert [c[:2] for c in calls[n_mid:]] == [(4, 6)]
    gan.use_bn = True
    n_mid = len(calls)
    gan.reconstruct_dataset({"test": (x, y)}, str(tmp_path), batch_size=3, test_again=True)
    assert [c[:2] for c in calls[n_mid:]] == [(3, 0), (3, 6), (1, 12)]
    gan.use_bn = False
    n_before = len(calls)
    # a whole-split feats.pkl (gan.py:484-496) takes precedence over the per-image pickles
    feats = tmp_path / "recs_rr2_lr10.00000_iters5" / "test" / "feats.pkl"
    with open(feats, "wb") as f:
        pickle.dump(x * 0.25, f, protocol=2)
    np.testing.assert_allclose(gan.reconstruct_dataset({"test": (x, y)}, str(tmp_path), batch_size=3)["test"][0], x * 0.25)
    assert len(calls) == n_before
    feats.unlink()
    # save_ds (gan.py:604-646): transformed images + targets as two consecutive pickles
    raw = (x * 255.0).astype(np.float32)
    paths = gan.save_ds({"dev": (raw, y)}, root=str(tmp_path / "cache"))
    assert paths["dev"].endswith("mnist_pkl/dev/feats.pkl")
    with open(paths["dev"], "rb") as f:
        a, b = pickle.load(f), pickle.load(f)
    np.testing.assert_allclose(a, x, rtol=1e-6)
    assert a.shape == (7, 28, 28, 1) and np.array_equal(b, y)
    _assert_python2_era_numpy_can_unpickle(open(paths["dev"], "rb").read(), n_objects=2)
    # the classifier-side wrapper forwards to reconstruct with the reference's arguments
    layer = ReconstructionLayer(gan, None, [None, 28, 28, 1], 3)
    np.testing.assert_allclose(layer.fprop(x[:2]), x[:2] * 0.5)
    assert calls[-1][2] == 123


def test_idx_ubyte_reader_and_split(tmp_path):
    from defensegan_amd import datasets
    rs = np.random.RandomState(0)
    for name, n in (("train", 60), ("t10k", 12)):
        img = rs.randint(0, 256, size=(n, 28, 28), dtype=np.uint8)
        lab = rs.randint(0, 10, size=n).astype(np.uint8)
        hdr_i = np.array([0, 0, 8, 3, 0, 0, 0, n, 0, 0, 0, 28, 0, 0, 0, 28], np.uint8)
        hdr_l = np.array([0, 0, 8, 1, 0, 0, 0, n], np.uint8)
        (tmp_path / ("%s-images-idx3-ubyte" % name)).write_bytes(hdr_i.tobytes() + img.tobytes())
        (tmp_path / ("%s-labels-idx1-ubyte" % name)).write_bytes(hdr_l.tobytes() + lab.tobytes())
        if name == "t10k":
            test_img, test_lab = img, lab
    x, y = datasets.load_mnist_split(str(tmp_path), "test")
    assert x.shape == (12, 28, 28, 1) and x.dtype == np.float32
    np.testing.assert_array_equal(x[..., 0], test_img)
    np.testing.assert_array_equal(y, test_lab)
    xt, _ = datasets.load_mnist_split(str(tmp_path), "train")
    xv, _ = datasets.load_mnist_split(str(tmp_path), "val")
    assert len(xt) == 50 and len(xv) == 10                      # 5/6 : 1/6 like 50 000 : 10 000
    g = datasets.to_generator_range(x, "mnist")
    assert g.min() >= 0 and g.max() <= 1
    c = datasets.to_generator_range(x, "celeba")
    assert c.min() >= -1 and c.max() <= 1


def test_command_line_surface():
    """python -m defensegan_amd keeps the reference's reconstruction flags (whitebox.py:358-395)."""
    from defensegan_amd import __main__ as cli
    a = cli.build_parser().parse_args(["--cfg", "celeba", "--init_path", "w.npz", "--input", "x.npy", "--output", "r.npy",
                                       "--rec_iters", "50", "--rec_rr", "4", "--same_init"])
    cfg = cfgmod.load_config(cli.resolve_cfg(a.cfg))
    assert cfg["DATASET_NAME"] == "celeba" and cfg["IMAGE_DIM"] == [64, 64, 3]
    assert cfgmod.resolve_rec_params(cfg, a) == {"rec_rr": 4, "rec_lr": 10.0, "rec_iters": 50, "batch_size": 50}
    assert a.same_init and not a.raw and a.seed == 11241990


def test_celeba_lazy_loader_crop_bytescale_resize(tmp_path):
    """datasets/celeba.py + LazyDataset: 1-based {i:06d}.jpg files, split index ranges, centre crop 108 -> scipy.misc.imresize
    (byte-scale over the crop's own min / max, PIL bilinear) -> 64x64x3, 'male' column as the label."""
    PIL = pytest.importorskip("PIL.Image")
    from defensegan_amd import datasets
    assert datasets.CELEBA_SPLITS["train"] == (1, 162770) and datasets.CELEBA_SPLITS["test"] == (182638, 202599)
    rs = np.random.RandomState(0)
    # a smooth 218 x 178 image (CelebA's size) whose crop does NOT span 0..255: the byte-scaling must stretch it
    yy, xx = np.mgrid[0:218, 0:178]
    img = np.stack([60 + 0.5 * yy, 80 + 0.4 * xx, 100 + 0.2 * (xx + yy)], axis=-1).astype(np.float64)
    got = datasets.prepare_celeba_image(img)
    assert got.shape == (64, 64, 3) and got.dtype == np.float32 and got.min() >= 0 and got.max() <= 255
    crop = img[55:163, 35:143]                                            # round((218-108)/2) = 55, round((178-108)/2) = 35
    assert got.max() > 250 and got.min() < 5                               # stretched over the crop's own range
    scaled = (crop - crop.min()) * 255.0 / (crop.max() - crop.min())
    # bilinear 108 -> 64 with PIL's area-scaled triangle filter stays within a pixel value of simple block means of a ramp
    centers = (np.arange(64) + 0.5) * 108 / 64 - 0.5
    ref = np.stack([np.stack([scaled[int(round(cy)), int(round(cx))] for cx in centers]) for cy in centers])
    assert np.abs(got - ref).max() <= 3.0
    # file access: three tiny "dataset" images + attribute file, 1-based names
    d = tmp_path / "celebA"
    d.mkdir()
    for i in (182638, 182639, 182640):
        arr = rs.randint(0, 256, size=(218, 178, 3)).astype(np.uint8)
        PIL.fromarray(arr).save(str(d / ("%06d.jpg" % i)), quality=95)
    lazy, labels = datasets.load_celeba_split(str(d), "test")
    assert labels is None and len(lazy) == 202599 - 182638 + 1 and lazy.shape == (None, 64, 64, 3)
    batch = lazy[0:3]
    assert batch.shape == (3, 64, 64, 3) and np.array_equal(batch[1], lazy[1]) and np.array_equal(lazy[[2, 0]][0], batch[2])
    g = datasets.to_generator_range(batch, "celeba")
    assert g.min() >= -1 and g.max() <= 1
    with open(str(d / "list_attr_celeba.txt"), "w") as f:
        f.write("202599\n5_o_Clock_Shadow Male Young\n")
        for i in range(1, 202600):
            f.write("%06d.jpg %d %d %d\n" % (i, -1, 1 if i % 3 == 0 else -1, 1))
    _, lab = datasets.load_celeba_split(str(d), "val", attribute="gender")
    assert lab.shape == (182637 - 162771 + 1,) and lab[0] == (1 if 162771 % 3 == 0 else 0) and set(lab.tolist()) == {0, 1}
    with pytest.raises(ValueError):
        datasets.load_celeba_split(str(d), "all")


class _FakeModel(object):
    """Stand-in with the attributes ``rows_are_independent`` looks at; its output depends on the image and on the z0 row the
    harness names for it (first_row / z_init_val), never on the batch it arrives in -- like the engine without Batchnorm."""

    def __init__(self, use_bn=False, rec_rr=3):
        self.use_bn, self.rec_rr, self.calls = use_bn, rec_rr, []

    def reconstruct(self, x, seed=None, first_row=None, z_init_val=None):
        x = np.asarray(x)
        self.calls.append((len(x), first_row))
        rows = first_row + np.arange(len(x)) * self.rec_rr                 # the global latent row of each image's first restart
        z = np.zeros(len(x)) if z_init_val is None else np.asarray(z_init_val).reshape(len(x), self.rec_rr, -1)[:, 0, 0]
        return np.clip(x * 0.5 + (rows % 7)[:, None, None, None] * 0.01 + z[:, None, None, None], 0, 1).astype(np.float32)


def test_engine_batch_images_takes_whole_caller_batches():
    assert gd.engine_batch_images(50, 10, 10000) == 1250           # the reference's batch: 25 caller batches = 12 500 rows
    assert gd.engine_batch_images(50, 10, 120) == 120              # never more than there is
    assert gd.engine_batch_images(256, 10, 10000) == 1280
    assert gd.engine_batch_images(2000, 10, 10000) == 2000         # a caller batch above the target stays whole
    assert gd.engine_batch_images(7, 1, 10000, target_rows=100) == 98


def test_model_eval_gan_coalesces_caller_batches_without_changing_the_result():
    rs = np.random.RandomState(1)
    x = rs.rand(130, 4, 4, 1).astype(np.float32) * 2
    y = rs.randint(0, 2, 130)
    per = _FakeModel()
    c0, n0, roc0 = gd.model_eval_gan(per.reconstruct, classifier, x, y, batch_size=10, rec_rr=3, coalesce=False)
    assert [k for k, _ in per.calls] == [10] * 13 and [fr for _, fr in per.calls] == [30 * i for i in range(13)]
    auto = _FakeModel()
    c1, n1, roc1 = gd.model_eval_gan(auto.reconstruct, classifier, x, y, batch_size=10, rec_rr=3)
    assert auto.calls == [(130, 0)]                                  # one engine call: 390 rows fit the target
    assert (c1, n1) == (c0, n0) and all(np.array_equal(a, b) for a, b in zip(roc0, roc1))
    some = _FakeModel()
    c2, _, roc2 = gd.model_eval_gan(some.reconstruct, classifier, x, y, batch_size=10, rec_rr=3, coalesce=45, first_image=20)
    assert [k for k, _ in some.calls] == [40, 40, 40, 10] and [fr for _, fr in some.calls] == [60, 180, 300, 420]
    # --same_init: every CALLER batch restarts from the head of the block, also inside a coalesced call (ragged tail included)
    zi = rs.rand(10 * 3, 2).astype(np.float32) * 0.05
    a, b = _FakeModel(), _FakeModel()
    ra = gd.model_eval_gan(a.reconstruct, classifier, x[:25], y[:25], batch_size=10, rec_rr=3, same_init_z=zi, coalesce=False)
    rb = gd.model_eval_gan(b.reconstruct, classifier, x[:25], y[:25], batch_size=10, rec_rr=3, same_init_z=zi)
    assert b.calls == [(25, 0)] and ra[0] == rb[0] and all(np.array_equal(p, q) for p, q in zip(ra[2], rb[2]))
    # Batchnorm couples the rows of a batch, a plain function says nothing about its rows: the caller's batches are kept
    bn = _FakeModel(use_bn=True)
    gd.model_eval_gan(bn.reconstruct, classifier, x, y, batch_size=10, rec_rr=3)
    assert [k for k, _ in bn.calls] == [10] * 13
    calls = []
    gd.model_eval_gan(fake_reconstruct(calls), classifier, x, y, batch_size=10, rec_rr=3)
    assert [k for k, _ in calls] == [10] * 13
