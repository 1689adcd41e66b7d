"""-m gpu: BASELINE configs[4] at the REAL per-GPU shard -- MNIST white-box FGSM eps = 0.3, the 10 000-image test list sharded
over 8 GPUs = 1250 images per rank, R = 10, L = 200 -- through model_eval_gan_sharded over a one-rank `nccl` (RCCL) group,
which is exactly what rank 3 of 8 executes (images 3750 .. 4999 of the list, z0 rows keyed by the global image index).

Inputs are what the reference feeds the defense (whitebox.py:198-210): FastGradientMethod(classifier).generate(x, eps = 0.3,
clip 0..1) on the BARE classifier (the attack is built before the reconstruction layer is attached, so the gradient is the
classifier's, N3), labels = the classifier's predictions on the clean images.

Checks: size-independent properties on all 1250 images (selection = first argmin, diff_op = loss of the selected restart,
predictions = classifier(rec), batch-composition independence of a ragged re-batching, bit for bit), and the DISTRIBUTIONAL
parity tier (tests.helpers.distributional_tier) of the first 16 images against torch-float32 / torch-float64."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.test_gpu_dist import ROOT, _env

pytestmark = pytest.mark.gpu

_WORKER = r"""
import json, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from defensegan_amd import gan_defense as gd, network_builder as nb, synth, archs
from tests.helpers import make_gan, distributional_tier
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
try:
    N_TOTAL, WORLD, RANK, R, L, NB = 10000, 8, 3, 10, 200, 16
    s0, e0 = gd.shard_range(N_TOTAL, RANK, WORLD)
    n = e0 - s0
    assert (s0, n) == (3750, 1250)
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.0, rec_rr=R, rec_iters=L)
    a = archs.make_arch("mnist")
    clean = gan.generate(gan.init_latents(n, seed=1000, first_row=s0))          # this rank's images of the global list
    clf = nb.model_a()
    clf.init_like_reference(seed=5)
    labels = clf.fprop(clean)["logits"].argmax(dim=1)
    fgsm = nb.FastGradientMethod(clf)
    x = fgsm.generate(clean, eps=0.3, y=labels, clip_min=0.0, clip_max=1.0)   # whitebox.py:198-210 on the bare classifier
    moved = float((x - clean).abs().max().item())
    frac_eps = float(((x - clean).abs() > 0.29).float().mean().item())
    labels_np = labels.cpu().numpy()
    details = {}
    def rec_fn(xb, **kw):
        d = gan.reconstruct(xb, return_details=True, **kw)
        details.setdefault("loss", []).append(d["loss"].cpu().numpy())
        details.setdefault("idx", []).append(d["idx"].cpu().numpy())
        details.setdefault("rec", []).append(d["rec"])
        return d["rec"]
    gan.prepare(n)
    # rank 3's call of the sharded evaluation: it holds only its shard; the group has one rank here, so the shard bounds are
    # passed through first_image (what shard_range gives rank 3 of 8)
    c, nn, roc = gd.model_eval_gan(rec_fn, clf, x, labels_np, batch_size=n, rec_rr=R, seed=2024, first_image=s0)
    loss = np.concatenate(details["loss"]).reshape(n, R)
    idx = np.concatenate(details["idx"])
    rec = torch.cat(details["rec"])
    # the same shard through the collective path (one-rank nccl group: all_gather over RCCL) with ragged batches of 500
    details.clear()
    acc2, roc2 = gd.model_eval_gan_sharded(lambda xb, **kw: rec_fn(xb, **dict(kw, first_row=kw["first_row"] + s0 * R)), clf, x, labels_np,
                                           batch_size=500, rec_rr=R, seed=2024)
    loss2 = np.concatenate(details["loss"]).reshape(n, R)
    preds_direct = clf.fprop(rec)["logits"].argmax(dim=1).cpu().numpy()
    gan.rec_iters = 1
    loss_start = gan.reconstruct(x, seed=2024, first_row=s0 * R, return_details=True)["loss"].cpu().numpy().reshape(n, R)
    gan.rec_iters = L
    mse_sel = ((rec - x) ** 2).flatten(1).mean(dim=1).cpu().numpy()
    # oracle subset: the first NB images of the shard, torch float32 and float64
    from tests.helpers import oracle_fixture, torch_runs
    xs = x[:NB].cpu().numpy()
    zs = gan.init_latents(NB * R, seed=2024, first_row=s0 * R).cpu().numpy()
    fx, fin = oracle_fixture("config4_fgsm_%%d" %% NB, {"x": xs, "z0": zs}, torch_runs(p, "mnist", R, L, 10.0))
    sub = gan.reconstruct(torch.from_numpy(fin["x"]).to(x.device), z_init_val=torch.from_numpy(fin["z0"]).to(x.device), return_details=True)
    if np.array_equal(fin["x"], xs) and np.array_equal(fin["z0"], zs):
        assert np.array_equal(sub["loss"].cpu().numpy().reshape(NB, R), loss[:NB])        # rows do not depend on their batch
    msg, dec = distributional_tier(fx["l32"].reshape(NB, R), fx["l64"].reshape(NB, R), sub["loss"].cpu().numpy().reshape(NB, R),
                                   sub["idx"].cpu().numpy())
    print("RESULT " + json.dumps({
        "backend": dist.get_backend(), "n": int(nn), "acc": c / nn, "acc2": acc2,
        "fgsm_max_move": moved, "fgsm_frac_at_eps": frac_eps, "in_range": bool(x.min().item() >= 0.0 and x.max().item() <= 1.0),
        "finite": bool(np.isfinite(loss).all()), "argmin_ok": bool((idx == loss.argmin(axis=1)).all()),
        "diff_is_best_loss": float(np.abs(roc[2] - loss.min(axis=1)).max() / loss.min(axis=1).max()),
        "diff_is_mse": float(np.abs(roc[2] - mse_sel).max() / mse_sel.max()),
        "preds_ok": bool((roc[1] == preds_direct).all()), "labels_ok": bool((roc[0] == labels_np).all()),
        "rebatched_loss_equal": bool(np.array_equal(loss, loss2)), "rebatched_preds_equal": bool((roc[1] == roc2[1]).all()),
        "rebatched_diffs_equal": bool(np.array_equal(roc[2], roc2[2])),
        "descended": float((loss.min(axis=1) < loss_start.min(axis=1)).mean()),
        "decidable": int(dec.sum()), "tier": msg}))
finally:
    dist.destroy_process_group()
"""


def test_config4_rank_shard_of_1250_fgsm_images_L200_over_rccl():
    r = subprocess.run([sys.executable, "-c", _WORKER % {"root": ROOT}], env=_env(), cwd=ROOT, capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    print(res["tier"])
    assert res["backend"] == "nccl" and res["n"] == 1250
    # the inputs really are FGSM at eps = 0.3 inside [0, 1]
    assert 0.299 <= res["fgsm_max_move"] <= 0.3 + 1e-6 and res["fgsm_frac_at_eps"] > 0.3 and res["in_range"]
    assert res["finite"] and res["argmin_ok"] and res["preds_ok"] and res["labels_ok"]
    assert res["diff_is_best_loss"] < 2e-4 and res["diff_is_mse"] < 2e-4
    # 1250-image batch == ragged batches of 500 over the RCCL path, bit for bit (rows are independent of their batch)
    assert res["rebatched_loss_equal"] and res["rebatched_preds_equal"] and res["rebatched_diffs_equal"]
    assert abs(res["acc"] - res["acc2"]) < 1e-12
    assert res["descended"] > 0.99
