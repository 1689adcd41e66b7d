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


@pytest.mark.parametrize("extra", [[], ["--strong", "--images", "230", "--batch", "100"]])
def test_bench_two_rank_flow_on_one_gpu_over_gloo(extra):
    """The driver's N > 1 command line -- torch.distributed.run starting two ranks of bench.py --gpus 2 -- on the one GPU of a test
    box: DG_BENCH_BACKEND=gloo moves the control tensors of the collectives to the CPU (RCCL refuses two ranks on one device),
    everything else is the code an 8-GPU run executes: rank 0 installs / times the job lists and broadcasts them, rank 1 imports
    them, barrier, the timed steps on two engines, the gather, the max over ranks, ONE line from rank 0.  The line says what it is
    (n_gpus = 1) and carries both ranks' times and job-list ids."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--rec_iters", "4", "--no-cpu-baseline"] + extra
    env = dict(os.environ)
    env.update({"DG_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["value"] > 0                       # two ranks shared one GPU: never reported as two
    assert len(res["ms_per_step_per_rank"]) == 2 and res["ms_per_step"] == pytest.approx(max(res["ms_per_step_per_rank"]), abs=1e-3)
    assert len(res["tuning_id_per_rank"]) == 2 and len(set(res["tuning_id_per_rank"])) == 1    # rank 1 runs rank 0's job lists
    assert res["config"]["parallelism"] == "shard2"
    if extra:
        assert res["scaling"] == "strong" and 0.0 <= res["accuracy"] <= 1.0


_WORKER2 = r"""
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from defensegan_amd import gan_defense as gd, network_builder as nb, synth
from tests.helpers import make_gan
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("DG_TEST_BACKEND", "nccl")
gpu = 0 if os.environ.get("DG_TEST_SAME_DEVICE") == "1" else rank
torch.cuda.set_device(gpu)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
else:
    dist.init_process_group(backend)
try:
    R, N = 3, 90
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.0, rec_rr=R, rec_iters=6, device=gpu)
    # rank 0 times the job lists, the others install its choices: identical lists everywhere
    if rank == 0:
        gan.prepare(32)
        text = [gan.export_tuning()]
    else:
        text = [None]
    dist.broadcast_object_list(text, src=0)
    if rank != 0:
        gan.import_tuning(text[0])
        gan.prepare(32)
    ids = [None] * world
    dist.all_gather_object(ids, gan.tuning_id())
    x = np.asarray(gan.generate(synth.make_z(N, 128, seed=4)))
    x = synth.adversarial(x, 0.3, 0.0, 1.0, seed=5)
    clf = nb.model_a(nb_filters=8)
    clf._device = gpu
    clf.init_like_reference(seed=6)
    labels = np.asarray(clf.fprop(x)["logits"].argmax(axis=1))
    acc, roc = gd.model_eval_gan_sharded(gan.reconstruct, clf, x, labels, batch_size=32, rec_rr=R, seed=77)
    s0, e0 = gd.shard_range(N, rank, world)
    rec = gan.reconstruct(x[s0:e0], seed=77, first_row=s0 * R)
    whole = gd.gather_shards(np.asarray(rec.cpu() if hasattr(rec, "cpu") else rec), N)
    out = {"rank": rank, "backend": dist.get_backend(), "world": dist.get_world_size(), "acc": acc, "ids": ids,
           "preds": roc[1].tolist(), "diffs": [float(v) for v in roc[2]], "rec_sum": float(np.abs(whole).sum())}
    if rank == 0:
        c, n, roc1 = gd.model_eval_gan(gan.reconstruct, clf, x, labels, batch_size=N, rec_rr=R, seed=77)
        rec1 = np.asarray(gan.reconstruct(x, seed=77, first_row=0))
        out.update({"acc1": c / n, "preds_equal": bool((roc[1] == roc1[1]).all()), "diffs_equal": bool(np.array_equal(roc[2], roc1[2])),
                    "gather_equal": bool(np.array_equal(whole, rec1))})
    print("RESULT " + json.dumps(out))
finally:
    dist.destroy_process_group()
"""


def _run_two_ranks(**extra):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = _env(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_PORT=str(port), **extra)
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER2 % {"root": ROOT}], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, out[-2000:] + err[-4000:]
        res.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):]))
    res.sort(key=lambda r: r["rank"])
    assert [r["world"] for r in res] == [2, 2]
    assert res[0]["ids"] == res[1]["ids"] and len(set(res[0]["ids"])) == 1            # the same job lists on both ranks
    assert res[0]["preds"] == res[1]["preds"] and res[0]["diffs"] == res[1]["diffs"] and res[0]["rec_sum"] == res[1]["rec_sum"]
    assert res[0]["preds_equal"] and res[0]["diffs_equal"] and res[0]["gather_equal"]
    assert abs(res[0]["acc"] - res[0]["acc1"]) < 1e-12
    return res


def test_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """World size 2 on real GPUs (skipped on the 1-GPU test boxes): one process per GPU, nccl = RCCL; rank 0's timed job lists
    are installed on rank 1 (same tuning id); every rank evaluates its contiguous image shard with its own engine, ONE
    all_gather assembles (labels, preds, diffs) -- identical on both ranks and bit-identical to one GPU doing everything."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (device_count = %d)" % torch.cuda.device_count())
    res = _run_two_ranks()
    assert [r["backend"] for r in res] == ["nccl", "nccl"]


def test_two_rank_worker_on_one_gpu_over_gloo():
    """The same two-rank worker with both ranks on GPU 0 and a gloo group (RCCL refuses two ranks on one device): everything
    but the transport -- the tuning hand-over between processes, two engines evaluating their shards, the device-built int32
    message and its single all_gather -- runs on the 1-GPU test box."""
    res = _run_two_ranks(DG_TEST_BACKEND="gloo", DG_TEST_SAME_DEVICE="1")
    assert [r["backend"] for r in res] == ["gloo", "gloo"]


def test_the_collective_behind_the_c_abi_without_torch_distributed():
    """dg_comm_* / dg_gather_eval (include/defensegan_hip.h): RCCL bound by the library itself.  A one-rank communicator here (one
    GPU per box): the gathered message is the message, and model_eval_gan_sharded through it returns what the plain evaluation
    returns; no torch.distributed group exists in this process."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from defensegan_amd import gan_defense as gd, network_builder as nb, synth
    from tests.helpers import clean_targets, make_gan
    assert not dist.is_initialized()
    uid = gd.EvalComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = gd.EvalComm(1, uid, 0, device=0)
    msg = torch.arange(1000, dtype=torch.int32, device="cuda:0")
    got = comm.all_gather_i32(msg)
    torch.cuda.synchronize()
    assert torch.equal(got, msg)
    R, L, N, BS = 4, 2, 70, 25
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.1, rec_rr=R, rec_iters=L)
    x, _ = clean_targets(p, "mnist", N, seed=91)
    y = (np.arange(N) * 3 % 10).astype(np.int64)
    clf = nb.model_a()
    clf.init_like_reference(seed=5)
    acc, roc = gd.model_eval_gan_sharded(gan.reconstruct, clf, x, y, batch_size=BS, rec_rr=R, seed=3, comm=comm)
    c1, n1, roc1 = gd.model_eval_gan(gan.reconstruct, clf, x, y, batch_size=BS, rec_rr=R, seed=3)
    assert abs(acc - c1 / n1) < 1e-12 and all(np.array_equal(a, b) for a, b in zip(roc, roc1))
    comm.close()
