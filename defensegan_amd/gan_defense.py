"""Evaluation harness around the projection path: counterpart of ``model_eval_gan``
(/root/reference/utils/gan_defense.py:32-179) plus the batch-sharded multi-GPU driver.

``model_eval_gan`` in the reference feeds ``test_images`` batch by batch, re-initialises the latent
variables before every batch (gan_defense.py:119), accumulates the number of correct predictions and
returns ``accuracy, roc_info=[labels, preds, diffs]`` (gan_defense.py:166-179).  Here the classifier is
any callable ``images -> logits/probabilities`` (torch or NumPy) and ``reconstruct`` is the engine's
``DefenseGANBase.reconstruct``; arrays replace the TF placeholders.

Multi-GPU: images are independent (use_bn=False), so the image list is sharded contiguously over the
ranks of a ``torch.distributed`` group (one process per GPU, RCCL over xGMI on ROCm; gloo on CPU) with
no collective on the data path and ONE all_gather of (labels, preds, diffs) at the end.  z0 rows are
keyed by the GLOBAL image index, so results do not depend on the GPU count.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [start, end) of ``n_items`` for ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _np(a):
    if isinstance(a, np.ndarray):
        return a
    return a.detach().cpu().numpy()


# Latent rows one engine call is filled up to when caller batches are coalesced: 12 500-row launches (the per-GPU shard of
# BASELINE configs[4] in one call) run the loop at 0.852 of the fp32 peak, the reference's default 500 rows (BATCH_SIZE 50,
# experiments/cfgs/gans/default.yml:2) at 0.656 (profiles/r04_exp_batch_sweep.txt, r04_bench_b50.json).
COALESCE_ROWS = 12800


def engine_batch_images(batch_size: int, rec_rr: int, n: int, target_rows: Optional[int] = None) -> int:
    """Images per engine call when the caller's batches of ``batch_size`` images are coalesced: as many WHOLE caller batches as
    fit ``target_rows`` (default ``COALESCE_ROWS``) latent rows (at least one), at most the ``n`` images there are."""
    per_call = max(1, int(COALESCE_ROWS if target_rows is None else target_rows) // max(1, int(rec_rr)))
    whole = max(1, per_call // max(1, int(batch_size)))
    return max(1, min(int(n), whole * int(batch_size)))


def rows_are_independent(reconstruct) -> bool:
    """True when ``reconstruct`` is the bound ``reconstruct`` of an engine model without Batchnorm: every (image, restart) row
    is then computed independently of the batch it is in, so ANY regrouping of the images into engine calls returns the same bits
    (tests/test_gpu_mnist.py test_rows_are_independent_of_batching_and_deterministic).  The invariant behind it: every output
    element is a FIXED tree of k-ordered fp32 chains -- tiles are cut along M / N freely, but along K only in the two fixed halves
    of the K-pair classes (dg_plan.h class_is_paired / pair_first_taps: functions of the layer plan alone, never of the row
    count or the job list; tests/test_plan.py pins that) whose sum commutes -- and z0 is keyed by the global row.  With USE_BN
    the rows of a batch share its statistics (tflib/ops/batchnorm.py:80-93) and the caller's batch boundaries are part of the
    result."""
    owner = getattr(reconstruct, "__self__", None)
    return owner is not None and hasattr(owner, "rec_rr") and hasattr(owner, "use_bn") and not bool(owner.use_bn)


def engine_step(reconstruct, batch_size: int, rec_rr: int, n: int, coalesce=None) -> int:
    """Images per engine call for a caller that feeds batches of ``batch_size``: ``coalesce`` None = automatic (runs of whole
    caller batches up to ``COALESCE_ROWS`` latent rows when ``rows_are_independent``), 0 / False = one call per caller batch (the
    reference's loops), an integer = that many images per call (rounded down to whole caller batches)."""
    if coalesce is None:
        coalesce = rows_are_independent(reconstruct)
    if coalesce is True:
        return engine_batch_images(batch_size, rec_rr, max(n, 1))
    if coalesce:
        return max(1, int(coalesce) // int(batch_size)) * int(batch_size)
    return int(batch_size)


def same_init_rows(same_init_z, start: int, end: int, batch_size: int, rec_rr: int):
    """--same_init (whitebox.py:181-183) for the images [start, end) of an engine call that spans several CALLER batches: every
    caller batch starts from the first rows of the same z block."""
    parts = [same_init_z[: (min(end, b0 + batch_size) - b0) * rec_rr] for b0 in range(start, end, batch_size)]
    return parts[0] if len(parts) == 1 else _cat(parts)


def project_in_batches(reconstruct: Callable, images, batch_size: int, rec_rr: int, seed: int = 11241990, first_image: int = 0,
                       same_init_z=None, coalesce=None) -> np.ndarray:
    """``reconstruct`` over ``images`` the way a per-batch caller would (the command line's loop, ``ReconstructionLayer`` users:
    latents of image i = rows ``(first_image + i) * rec_rr + r`` of the seeded stream, ``same_init_z`` restarting at every caller
    batch), with the caller batches coalesced into engine calls as in ``model_eval_gan``.  Returns the reconstructions [n, ...]."""
    n = len(images)
    step = engine_step(reconstruct, batch_size, rec_rr, n, coalesce)
    out = None
    for start in range(0, n, step):
        end = min(n, start + step)
        kw = {}
        if same_init_z is not None:
            kw["z_init_val"] = same_init_rows(same_init_z, start, end, batch_size, rec_rr)
        rec = _np(reconstruct(images[start:end], seed=seed, first_row=(first_image + start) * rec_rr, **kw))
        if out is None:
            out = np.empty((n,) + tuple(rec.shape[1:]), rec.dtype)
        out[start:end] = rec
    return out if out is not None else np.zeros((0,) + tuple(np.shape(images)[1:]), np.float32)


def model_eval_gan(reconstruct: Optional[Callable], classifier: Callable, test_images, test_labels,
                   batch_size: int, rec_rr: int = 1, compute_diffs: bool = True, seed: int = 11241990,
                   first_image: int = 0, same_init_z: Optional[np.ndarray] = None,
                   verbose: bool = False, as_tensors: bool = False, coalesce=None):
    """Accuracy of ``classifier(reconstruct(x))`` over ``test_images`` + reconstruction errors.

    ``batch_size`` is the caller's (the reference's ``BATCH_SIZE`` 50, default.yml:2).  It fixes what a batch MEANS -- the
    latents of image i are row ``(first_image + i) * rec_rr + r`` of the seeded stream, ``same_init_z`` restarts at every batch
    boundary -- but not how many images go to the engine at once: when the rows are independent (``rows_are_independent``:
    an engine model without Batchnorm) consecutive caller batches are COALESCED into engine calls of up to ``COALESCE_ROWS``
    latent rows, which returns the same bits several times faster (500-row launches run at 0.66 of the peak, 12 500-row ones at
    0.85).  ``coalesce``: None = automatic as described, 0 / False = one engine call per caller batch (the reference's loop,
    gan_defense.py:113-162), an integer = that many images per engine call (rounded down to whole caller batches).

    reconstruct : ``f(images, z_init_val=None, seed=..., first_row=...) -> reconstructions`` or None
                  (no defense: the classifier sees the inputs).
    test_labels : class indices [n] or one-hot [n, classes] (the reference takes argmax, gan_defense.py:91-99)
    Returns ``(correct_count, n, roc_info)`` with ``roc_info = [labels, preds, diffs]``;
    ``diffs[i] = mean((x_i - rec_i)^2)`` (diff_op of whitebox.py:218 / blackbox.py:571-572).
    Accuracy = correct_count / n (gan_defense.py:166) -- kept as a count so shards can be summed.

    With a device classifier (network_builder.MLP.eval_batch) the per-batch predictions, differences and correct counts stay
    on the device and no batch waits for the stream: the host enqueues the whole evaluation and reads the results back once
    at the end (the reference fetches ``acc_value / cur_preds / diff_op`` every batch, gan_defense.py:132-160).  ``as_tensors``
    returns ``roc_info`` as tensors (labels / preds int32, diffs float32, on the classifier's device) for the sharded driver.
    """
    n = len(test_images)
    labels = _np(test_labels)
    if labels.ndim > 1:
        labels = labels.argmax(axis=-1)
    # images per engine call: whole caller batches (the z_init / seed bookkeeping below is per caller batch either way)
    step = engine_step(reconstruct, batch_size, rec_rr, n, coalesce)
    nb_batches = int(math.ceil(float(n) / step))
    on_device = hasattr(classifier, "eval_batch")
    preds: List = []
    diffs: List = []
    counts: List = []
    correct = 0
    labels_dev = None
    if on_device and n:
        import torch
        classifier._ensure()
        labels_dev = torch.from_numpy(np.ascontiguousarray(labels.astype(np.int32))).to(torch.device("cuda", classifier._device))
    for batch in range(nb_batches):
        start = batch * step
        end = min(n, start + step)                # last batch may be smaller (gan_defense.py:124-130)
        x = test_images[start:end]
        if reconstruct is not None:
            kw = {}
            if same_init_z is not None:
                # --same_init (whitebox.py:181-183): every CALLER batch starts from the first rows of the same block
                kw["z_init_val"] = same_init_rows(same_init_z, start, end, batch_size, rec_rr)
            rec = reconstruct(x, seed=seed, first_row=(first_image + start) * rec_rr, **kw)
        else:
            rec = x
        if on_device:
            # network_builder.MLP: classifier forward, argmax, correct count and diff_op in one device pass (dg_eval_batch);
            # [B] predictions / differences and the count stay on the device
            c_dev, p_dev, d_dev = classifier.eval_batch(rec, x if compute_diffs else None, labels_dev[start:end], sync=False)
            counts.append(c_dev)
            preds.append(p_dev)
            if compute_diffs:
                diffs.append(d_dev)
        else:
            out = _np(classifier(rec))
            p = out.argmax(axis=-1) if out.ndim > 1 else out.astype(np.int64)
            preds.append(p.astype(np.int64))
            correct += int((p == labels[start:end]).sum())
            if compute_diffs:
                xr, rr = _np(x).reshape(end - start, -1), _np(rec).reshape(end - start, -1)
                diffs.append(((xr - rr) ** 2).mean(axis=1).astype(np.float32))
        if verbose:
            print("[#] Eval batch {}/{}".format(batch, nb_batches))
    if on_device and n:
        import torch
        preds_t = torch.cat(preds)
        diffs_t = torch.cat(diffs) if diffs else torch.zeros(0, dtype=torch.float32, device=preds_t.device)
        correct = int(torch.cat(counts).sum().item())          # the one wait of the evaluation
        if as_tensors:
            return correct, n, [labels_dev, preds_t, diffs_t]
        return correct, n, [labels.astype(np.int64), preds_t.cpu().numpy().astype(np.int64), diffs_t.cpu().numpy().astype(np.float32)]
    preds_all = np.concatenate(preds) if preds else np.zeros(0, np.int64)
    diffs_all = np.concatenate(diffs) if diffs else np.zeros(0, np.float32)
    if as_tensors:
        import torch
        return correct, n, [torch.from_numpy(labels.astype(np.int32)), torch.from_numpy(preds_all.astype(np.int32)),
                            torch.from_numpy(diffs_all.astype(np.float32))]
    return correct, n, [labels.astype(np.int64), preds_all, diffs_all]


def _cat(parts):
    if isinstance(parts[0], np.ndarray):
        return np.concatenate(parts)
    import torch
    return torch.cat(list(parts))


def gather_shards(local_rows: np.ndarray, n_total: int, group=None, device=None) -> np.ndarray:
    """Assembles the per-rank contiguous shards (``shard_range``) of an [n_total, ...] float32 array on every rank with
    ONE tensor all_gather (padded to the largest shard) -- RCCL over xGMI when the group's backend is nccl."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    assert len(local_rows) == sizes[rank], (len(local_rows), sizes[rank])
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    tail = tuple(local_rows.shape[1:])
    buf = torch.zeros((max(sizes),) + tail, dtype=torch.float32, device=device)
    buf[:sizes[rank]] = torch.from_numpy(np.ascontiguousarray(local_rows, np.float32))
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return np.concatenate([t[:k].cpu().numpy() for t, k in zip(out, sizes)])


class EvalComm(object):
    """An RCCL communicator behind the C ABI (include/defensegan_hip.h dg_comm_*): the evaluation's one all_gather WITHOUT
    torch.distributed -- what a caller that is not a torch program (the reference is TensorFlow) binds.  One process per GPU:

        uid = EvalComm.unique_id() on rank 0, carried to the other ranks by the caller (file, MPI, environment ...)
        comm = EvalComm(nranks, uid, rank, device)
        acc, roc = model_eval_gan_sharded(gan.reconstruct, clf, x, y, 50, rec_rr=10, comm=comm)
    """

    def __init__(self, nranks: int, unique_id: bytes, rank: int, device: int = 0):
        import ctypes as C
        from . import _native
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        self._lib = _native.load()
        self._id = C.create_string_buffer(bytes(unique_id), 128)
        h = C.c_void_p()
        _native.check(self._lib.dg_comm_create(int(nranks), C.cast(self._id, C.c_void_p), int(rank), int(device), C.byref(h)))
        self._h, self.rank, self.world, self.device = h, int(rank), int(nranks), int(device)

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _native
        buf = C.create_string_buffer(128)
        _native.check(_native.load().dg_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    def all_gather_i32(self, buf):
        """[count] int32 device tensor of every rank -> [world * count] on every rank (asynchronous on torch's current stream)."""
        import torch
        from . import _native
        assert buf.dtype == torch.int32 and buf.is_cuda and buf.is_contiguous()
        out = torch.empty(self.world * buf.numel(), dtype=torch.int32, device=buf.device)
        stream = torch.cuda.current_stream(buf.device).cuda_stream
        _native.check(self._lib.dg_gather_eval(self._h, buf.data_ptr(), out.data_ptr(), buf.numel(), stream))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def model_eval_gan_sharded(reconstruct, classifier, test_images, test_labels, batch_size: int, rec_rr: int = 1,
                           group=None, device=None, n_total: Optional[int] = None, comm: Optional["EvalComm"] = None, **kw):
    """Batch-sharded evaluation over a torch.distributed group: every rank evaluates its contiguous shard,
    then ONE all_gather (padded to the largest shard) assembles ``roc_info`` in global image order.
    Returns ``(accuracy, roc_info)`` identically on every rank.

    ``test_images`` / ``test_labels`` hold the WHOLE list (every rank slices its ``shard_range``), or -- when ``n_total`` is
    given -- only this rank's shard of a list of ``n_total`` images (nothing but the shard needs to exist on a rank).
    ``comm``: an ``EvalComm`` (RCCL through the C ABI) instead of a torch.distributed group."""
    import torch
    import torch.distributed as dist

    if comm is None and (not dist.is_available() or not dist.is_initialized()):
        c, n, roc = model_eval_gan(reconstruct, classifier, test_images, test_labels, batch_size, rec_rr, **kw)
        return c / max(n, 1), roc
    rank, world = (comm.rank, comm.world) if comm is not None else (dist.get_rank(group), dist.get_world_size(group))
    presharded = n_total is not None
    if not presharded:
        n_total = len(test_images)
    s, e = shard_range(n_total, rank, world)
    if presharded:
        if len(test_images) != e - s:
            raise ValueError("rank %d holds %d images, its shard of %d is [%d,%d)" % (rank, len(test_images), n_total, s, e))
        shard_x, shard_y = test_images, _np(test_labels)
    else:
        shard_x, shard_y = test_images[s:e], _np(test_labels)[s:e]
    c, n, roc = model_eval_gan(reconstruct, classifier, shard_x, shard_y, batch_size, rec_rr, first_image=s, as_tensors=True, **kw)
    cap = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    if device is None:
        device = (torch.device("cuda", comm.device) if comm is not None else
                  torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu")
    # ONE message per rank, built where the results already are (the device, with an nccl group): int32
    # [count | labels (cap) | preds (cap) | diffs (cap, float32 bits)] -- exact, 12 bytes per image (15 KB per rank for the
    # 1250-image shards of BASELINE configs[4]), one all_gather, one copy back to the host
    buf = torch.zeros(1 + 3 * cap, dtype=torch.int32, device=device)
    buf[0] = n
    buf[1:1 + n] = roc[0].to(device=device, dtype=torch.int32)
    buf[1 + cap:1 + cap + n] = roc[1].to(device=device, dtype=torch.int32)
    if len(roc[2]):
        buf[1 + 2 * cap:1 + 2 * cap + n] = roc[2].to(device=device, dtype=torch.float32).view(torch.int32)
    if comm is not None:
        out = comm.all_gather_i32(buf)
    else:
        out = torch.empty(world * (1 + 3 * cap), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(out, buf, group=group)
    t = out.view(world, 1 + 3 * cap).cpu().numpy()
    labels, preds, diffs = [], [], []
    for row in t:
        k = int(row[0])
        labels.append(row[1:1 + k].astype(np.int64))
        preds.append(row[1 + cap:1 + cap + k].astype(np.int64))
        diffs.append(np.ascontiguousarray(row[1 + 2 * cap:1 + 2 * cap + k]).view(np.float32).copy())
    labels, preds, diffs = np.concatenate(labels), np.concatenate(preds), np.concatenate(diffs)
    acc = float((labels == preds).sum()) / max(n_total, 1)
    return acc, [labels, preds, diffs]
