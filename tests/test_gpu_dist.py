"""-m gpu: the multi-GPU code path with the REAL backend (`nccl` = RCCL on ROCm) and the REAL projection, on the one GPU a
test box has: a one-rank process group.  World-size-2 logic is covered on CPU by tests/test_eval_dist.py (gloo)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env(**extra):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "WORLD_SIZE": "1",
                "LOCAL_RANK": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra)
    return env


_WORKER = r"""
import json, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from defensegan_amd import gan_defense as gd, network_builder as nb, synth
from tests.helpers import make_gan
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
try:
    R = 3
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.0, rec_rr=R, rec_iters=8)
    x = np.asarray(gan.generate(synth.make_z(64, 128, seed=4)))
    x = synth.adversarial(x, 0.3, 0.0, 1.0, seed=5)
    clf = nb.model_a(nb_filters=8)
    clf.init_like_reference(seed=6)
    labels = clf.fprop(x)["logits"].argmax(axis=1)
    acc, roc = gd.model_eval_gan_sharded(gan.reconstruct, clf, x, labels, batch_size=25, rec_rr=R, seed=77)
    c, n, roc1 = gd.model_eval_gan(gan.reconstruct, clf, x, labels, batch_size=64, rec_rr=R, seed=77)
    whole = gd.gather_shards(x, len(x))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    print("RESULT " + json.dumps({"backend": dist.get_backend(), "acc": acc, "acc1": c / n,
                                  "preds_equal": bool((roc[1] == roc1[1]).all()), "labels_equal": bool((roc[0] == roc1[0]).all()),
                                  "diffs_equal": bool(np.array_equal(roc[2], roc1[2])), "n": int(len(roc[1])),
                                  "gather_equal": bool(np.array_equal(whole, x)), "allreduce": float(t.sum().item())}))
finally:
    dist.destroy_process_group()
"""


def test_sharded_evaluation_over_rccl_equals_the_unsharded_call():
    """model_eval_gan_sharded with backend nccl, the engine's reconstruct and the device classifier: batch size 25 (ragged)
    sharded == batch size 64 unsharded, bit for bit (z0 rows are keyed by the global image index)."""
    r = subprocess.run([sys.executable, "-c", _WORKER % {"root": ROOT}], env=_env(), cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["backend"] == "nccl" and res["n"] == 64 and res["allreduce"] == 4.0
    assert res["preds_equal"] and res["labels_equal"] and res["diffs_equal"] and res["gather_equal"]
    assert abs(res["acc"] - res["acc1"]) < 1e-12


@pytest.mark.parametrize("extra", [[], ["--strong", "--images", "230", "--batch", "100"]])
def test_bench_runs_its_rccl_path_with_one_rank(extra):
    """bench.py under DG_BENCH_FORCE_DIST=1: process group over nccl, barrier + all_gather / the sharded evaluation inside
    the timed region, ONE JSON line with the contract's keys."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--rec_iters", "4",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=_env(DG_BENCH_FORCE_DIST="1"), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "build"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["dtype"] == "f32"
    if extra:
        assert res["scaling"] == "strong" and 0.0 <= res["accuracy"] <= 1.0 and "configs[4]" in res["config"]["workload"]
    else:
        assert res["scaling"] == "weak" and res["roofline"]["kernel"] and res["roofline"]["frac"] > 0
        # the untimed event pass: the kernels of one step cannot take longer than a timed step plus launch slack
        assert res["roofline"]["sum_kernel_ms_per_step"] <= 1.10 * res["ms_per_step"] + 2.0
