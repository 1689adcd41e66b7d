"""-m gpu: SURVEY 8f-N4 through the REAL engine -- ``reconstruct_dataset`` (/root/reference/models/gan.py:451-587) on the HIP
path: batches of 16 over 40 images (ragged tail), per-image pickles under the reference's directory template, their contents
against the float64 oracle at L = 5, and the cache-hit path (gan.py:504-557) returning the stored arrays without recomputing."""
import os
import pickle

import numpy as np
import pytest

from defensegan_amd import config as cfgmod, synth
from tests.helpers import clean_targets, make_gan

pytestmark = pytest.mark.gpu


def test_reconstruct_dataset_through_the_engine_matches_the_oracle_and_its_cache(tmp_path):
    from oracle import defensegan_oracle as O
    R, L, N, BS = 3, 5, 40, 16
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=L)
    x, _ = clean_targets(p, "mnist", N, seed=61)
    x = synth.adversarial(x, 0.3, 0.0, 1.0, seed=62)
    y = (np.arange(N) * 7 % 10).astype(np.int64)
    ck = str(tmp_path / "ckpt")
    rets = gan.reconstruct_dataset({"test": (x, y)}, ck, batch_size=BS, seed=99)
    recs, targets, orig = rets["test"]
    assert recs.shape == x.shape and recs.dtype == np.float32 and np.array_equal(targets, y) and np.array_equal(orig, x)
    # the reference's cache layout (gan.py:470-480, 504-513): recs_rr{R}_lr{lr:.5f}_iters{L}/<split>/pickles/rec_{i:07d}_l{label}.pkl
    d = os.path.join(ck, cfgmod.rec_dir_name(R, 10.0, L), "test", "pickles")
    assert os.path.basename(os.path.dirname(os.path.dirname(d))) == "recs_rr3_lr10.00000_iters5"
    names = sorted(os.listdir(d))
    assert names == ["rec_{:07d}_l{}.pkl".format(i, y[i]) for i in range(N)]
    for i in (0, 15, 16, 39):
        with open(os.path.join(d, names[i]), "rb") as f:
            blob = f.read()
        assert blob[:2] == b"\x80\x02"                                  # protocol 2: loadable by the Python-2 reference
        a = pickle.loads(blob, encoding="latin1")
        assert a.dtype == np.float32 and a.shape == (28, 28, 1) and np.array_equal(a, recs[i])
    # contents: z0 rows are keyed by (seed, global row = image * R + r), whatever the batching -- the float64 oracle from the
    # same latents reproduces every image (a ReLU gate within float32 rounding of zero may move a row: 1 of 40 tolerated)
    z0 = gan.init_latents(N * R, seed=99, first_row=0).cpu().numpy()
    ref = O.reconstruct(p, x, z0, R, L, lr=10.0, momentum=0.7, arch="mnist", dtype=np.float64)
    err = np.abs(recs.astype(np.float64) - ref["rec"]).reshape(N, -1).max(axis=1)
    assert (err <= 2e-5).sum() >= N - 1 and np.median(err) <= 2e-6 and err.max() <= 2e-2, np.sort(err)[-3:]
    # one unbatched call gives the same arrays bit for bit (batch-composition independence through the cache writer)
    whole = gan.reconstruct(x, seed=99, first_row=0)
    assert np.array_equal(np.asarray(whole), recs)
    # cache hit: every pickle exists -> nothing is recomputed.  Proof: the engine now has other weights, the arrays do not move.
    p2 = synth.make_weights("mnist", seed=77, gain=2.0, bias_range=0.1)
    assert gan.set_weights(p2) == []
    hit = gan.reconstruct_dataset({"test": (x, y)}, ck, batch_size=BS, seed=99)["test"][0]
    assert np.array_equal(hit, recs)
    # a partially filled cache recomputes only the batch with a missing pickle (here: with the new weights) ...
    os.remove(os.path.join(d, names[20]))
    part = gan.reconstruct_dataset({"test": (x, y)}, ck, batch_size=BS, seed=99)["test"][0]
    assert np.array_equal(part[:16], recs[:16]) and np.array_equal(part[32:], recs[32:])
    assert not np.array_equal(part[16:32], recs[16:32])
    # ... test_again recomputes everything, and max_num gets its own directory (gan.py:472-474)
    again = gan.reconstruct_dataset({"test": (x, y)}, ck, batch_size=BS, seed=99, test_again=True)["test"][0]
    assert np.array_equal(again[16:32], part[16:32]) and not np.array_equal(again[:16], recs[:16])
    few = gan.reconstruct_dataset({"test": (x, y)}, ck, batch_size=BS, max_num=10, seed=99)["test"]
    assert few[0].shape[0] == 10 and os.path.isdir(os.path.join(ck, cfgmod.rec_dir_name(R, 10.0, L) + "_num10", "test", "pickles"))
    assert np.array_equal(few[0], again[:10])
