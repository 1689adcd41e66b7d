"""Host-side mirror of the reference's GAN model object for the latent-projection path.

Keeps the surface of ``DefenseGANBase`` (/root/reference/models/gan.py:39-647) that the callers of
the hot path use -- ``GAN(cfg, test_mode=True)``, ``.load_generator()``, ``.reconstruct(images,
batch_size, back_prop, reconstructor_id, z_init_val)``, attributes ``rec_iters / rec_rr / rec_lr /
latent_dim / net_dim / use_bn / image_dim / batch_size / dataset_name`` -- and forwards the work to
the hand-written HIP engine through the C ABI (include/defensegan_hip.h).  Arrays replace TF tensors:
NumPy in -> NumPy out, torch in -> torch (on the engine's device) out.

Differences from the reference that are deliberate:
* stateless per call: fresh z0 / zero momentum every batch (what ``model_eval_gan`` arranges with
  ``tf.local_variables_initializer()``, utils/gan_defense.py:119); no warm start across batches;
* ragged last batches are accepted (the reference's static graph needs exactly ``batch_size`` images);
* the learning rate is the constant ``rec_lr`` -- what the reference executes, because the decay's
  step variable is never advanced (gan.py:362-386, base_model.py:188-192).
There is no CPU fallback: without the HIP library / a GPU every compute method raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _native, archs, config as _config


def _torch():
    import torch
    return torch


class DefenseGANBase(object):
    arch_name = None          # set by subclasses

    _default_attributes = ['dataset_name', 'batch_size', 'use_bn', 'test_batch_size', 'latent_dim',
                           'net_dim', 'rec_iters', 'image_dim', 'rec_rr', 'rec_lr', 'debug']

    def __init__(self, cfg=None, test_mode=False, verbose=False, device=None, measure=False, **args):
        # defaults of DefenseGANBase.__init__ (gan.py:50-68)
        self.dataset_name = None
        self.batch_size = 32
        self.use_bn = True
        self.test_batch_size = 20
        self.latent_dim = None
        self.net_dim = None
        self.debug = False
        self.rec_iters = 200
        self.image_dim = [None, None, None]
        self.rec_rr = 10
        self.rec_lr = 10.0
        self.rec_momentum = 0.7            # hard-coded in the reference (gan.py:390)
        # "constant": lr == rec_lr throughout = what the reference executes (its decay never fires, gan.py:362-386);
        # "intended": the x0.1 staircase at ceil(0.8 * rec_iters) its code asks for.  Off by default.
        self.rec_lr_schedule = args.get("rec_lr_schedule", "constant")
        self.test_mode = test_mode
        self.verbose = verbose
        self.cfg = dict(cfg) if cfg else {}
        self.initialized = False
        # attribute resolution order of AbstractModel._set_attr (base_model.py:120-148):
        # explicit keyword > cfg[UPPER] > cfg[lower] > class default
        for name in self._default_attributes:
            val = args.get(name)
            if val is None:
                if name.upper() in self.cfg:
                    val = self.cfg[name.upper()]
                elif name.lower() in self.cfg:
                    val = self.cfg[name.lower()]
            if val is not None:
                setattr(self, name, val)
        if self.latent_dim is None:
            self.latent_dim = 128
        if self.net_dim is None:
            self.net_dim = 64
        self._arch = archs.make_arch(self.arch_name or self.dataset_name, int(self.latent_dim), int(self.net_dim))
        if self.image_dim is None or self.image_dim[0] is None:
            self.image_dim = list(self._arch.image_dim)
        if list(self.image_dim) != list(self._arch.image_dim):
            raise ValueError("image_dim %r does not match the %s generator (%r)" %
                             (self.image_dim, self._arch.name, self._arch.image_dim))
        if test_mode:
            self.test_batch_size = self.batch_size        # gan.py:105
        self._device = device
        self._measure = bool(measure)       # True: the -DDG_MEASURE build of the library (tools/, cross-check tests)
        self._handle = None
        self._weights: Dict[str, np.ndarray] = {}
        self._default_seed = 11241990                     # whitebox.py:143 / blackbox.py:464
        self._calls = 0

    # ------------------------------------------------------------------ engine plumbing
    def _lib(self):
        return _native.load(self._measure)

    def _check(self, rc):
        _native.check(rc, self._lib())

    def _ensure_handle(self):
        if self._handle is not None:
            return self._handle
        torch = _torch()
        lib = self._lib()
        if not torch.cuda.is_available():
            raise _native.NativeError("no GPU visible: the projection engine has no CPU fallback")
        dev = self._device
        if dev is None:
            dev = torch.cuda.current_device()
        if isinstance(dev, str):
            dev = torch.device(dev)
        if hasattr(dev, "index"):
            dev = dev.index if dev.index is not None else torch.cuda.current_device()
        self._device = int(dev)
        h = C.c_void_p()
        self._check(lib.dg_create(self._arch.arch_id, int(self.latent_dim), int(self.net_dim),
                                    1 if self.use_bn else 0, self._device, C.byref(h)))
        self._handle = h
        for k, v in self._weights.items():
            self._push_weight(k, v)
        self._lr_schedule_set = "constant"
        return h

    def _push_weight(self, name: str, arr: np.ndarray):
        lib = self._lib()
        a = np.ascontiguousarray(arr, dtype=np.float32)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        self._check(lib.dg_set_weights(self._handle, name.encode(), a.ctypes.data_as(C.c_void_p), shape,
                                         a.ndim, 0))

    def close(self):
        if self._handle is not None:
            self._lib().dg_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def set_weights(self, weights: Dict[str, np.ndarray]):
        """weights: tflib parameter name -> array in the reference layout
        (tflib/__init__.py:9-33; names listed in include/defensegan_hip.h)."""
        want = archs.weight_shapes(self._arch, bool(self.use_bn))
        for k, v in weights.items():
            if k not in want:
                continue
            v = np.asarray(v)
            if k.endswith(".scale") or k.endswith(".offset"):
                # tflib Batchnorm stores these with the keep_dims shape of the moments: [1,4096] for BN1,
                # [1,1,1,C] for BN2/BN3 (tflib/ops/batchnorm.py:83-89); any shape with C values is accepted
                if v.size != int(np.prod(want[k])):
                    raise ValueError("%s: shape %r, expected %d values" % (k, tuple(v.shape), int(np.prod(want[k]))))
                v = v.reshape(want[k])
            elif tuple(v.shape) != tuple(want[k]):
                raise ValueError("%s: shape %r, expected %r" % (k, tuple(v.shape), want[k]))
            self._weights[k] = np.ascontiguousarray(v, np.float32)
            if self._handle is not None:
                self._push_weight(k, self._weights[k])
        missing = [k for k in want if k not in self._weights]
        self.initialized = not missing
        return missing

    def load_generator(self, ckpt_path: Optional[str] = None):
        """Counterpart of ``load_generator`` (gan.py:85-87 -> base_model.py:294-335).  ``ckpt_path`` is

        * an ``.npz`` weight pack whose keys are the tflib parameter names (or a directory holding ``generator.npz``), or
        * a TensorFlow checkpoint written by the reference: a directory with a ``checkpoint`` state file (resolved like
          ``tf.train.get_checkpoint_state``), a checkpoint prefix, or its ``.index`` file -- read without TensorFlow by
          ``tf_checkpoint`` (V2 tensor-bundle format; variables are matched by the last component of their name, so
          ``Generator.Input/Generator.Input.W`` and optimizer slots in the same file are handled).
        """
        if ckpt_path is None:
            ckpt_path = getattr(self, "checkpoint_dir", None)
        if ckpt_path is None:
            raise ValueError("load_generator needs the path of an .npz weight pack or of a TensorFlow checkpoint")
        if os.path.isdir(ckpt_path) and os.path.exists(os.path.join(ckpt_path, "generator.npz")):
            ckpt_path = os.path.join(ckpt_path, "generator.npz")
        if ckpt_path.endswith(".npz"):
            with np.load(ckpt_path) as f:
                weights = {k: f[k] for k in f.files}
        else:
            from . import tf_checkpoint
            expected = archs.weight_shapes(self._arch, bool(self.use_bn))
            weights = tf_checkpoint.generator_weights(ckpt_path, expected.keys())
        missing = self.set_weights(weights)
        if missing:
            raise ValueError("%s lacks %s" % (ckpt_path, ", ".join(missing)))
        return True

    def save_ds(self, splits, root: str = "data/cache", test_again: bool = False):
        """Counterpart of ``save_ds`` (gan.py:604-646): ``<root>/<dataset>_pkl/<split>/feats.pkl`` holding two consecutive
        pickles, the transformed images ``[n,H,W,C]`` (``input_transform`` of the raw [0,255] data) and the targets.
        ``splits``: {'train'|'dev'|'test': (raw images, targets)}.  Written by ``py2pickle`` (protocol 2, Python-2-era
        NumPy module paths) so the Python-2 reference can read it."""
        from . import py2pickle
        out = {}
        for split, (images, targets) in splits.items():
            out_dir = os.path.join(root, "{}_pkl".format(self.dataset_name or self.arch_name), split)
            os.makedirs(out_dir, exist_ok=True)
            path = os.path.join(out_dir, "feats.pkl")
            out[split] = path
            if os.path.exists(path) and not test_again:
                continue
            x = self.input_transform(np.asarray(images, np.float32)).reshape([-1] + list(self.image_dim)).astype(np.float32)
            with open(path, "wb") as f:
                py2pickle.dump(x, f)
                py2pickle.dump(np.asarray(targets), f)
        return out

    # ------------------------------------------------------------------ the hot path
    def input_transform(self, X):
        """[0,255] -> generator range: /255 for MNIST / F-MNIST (gan.py:684-685, 697-698),
        2*(x/255 - .5) for CelebA (gan.py:764-765)."""
        if self._arch.arch_id == archs.ARCH_CELEBA:
            return 2.0 * (X / 255.0 - 0.5)
        return X / 255.0

    def _to_device(self, a, dtype=None):
        torch = _torch()
        dev = torch.device("cuda", self._device)
        if isinstance(a, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(a))
        else:
            t = a
        t = t.to(device=dev, dtype=dtype or torch.float32)
        return t.contiguous()

    def reconstruct(self, images, batch_size=None, back_prop=True, reconstructor_id=0, z_init_val=None,
                    seed=None, first_row=0, return_details=False):
        """Defense-GAN projection of ``images`` [B,H,W,C] (already in generator range).

        Same arguments as the reference (gan.py:333-335).  ``batch_size`` only validates the leading
        dimension when given (the reference needs it for its static graph); ``back_prop`` is accepted and
        ignored -- the reference's gradient through this op is identically zero;
        ``reconstructor_id`` only named TF variables.  ``z_init_val`` [B*rec_rr, latent] fixes z0
        (row b*rec_rr + r, gan.py:348-359, 395-397); otherwise rows are drawn N(0, 1/latent) on the
        device from (seed, first_row + row).  Returns the reconstructions [B,H,W,C]; with
        ``return_details`` a dict with rec, idx [B], loss [B*R], z [B*R, latent].
        """
        self._ensure_handle()
        if not self.initialized:
            raise _native.NativeError("generator weights not loaded (load_generator / set_weights)")
        torch = _torch()
        lib = self._lib()
        was_numpy = isinstance(images, np.ndarray)
        x = self._to_device(images)
        H, W, Cc = self._arch.image_dim
        if x.dim() == 2 and x.shape[1] == H * W * Cc:
            x = x.view(-1, H, W, Cc)
        if x.dim() != 4 or tuple(x.shape[1:]) != (H, W, Cc):
            raise ValueError("images must be [B,%d,%d,%d], got %r" % (H, W, Cc, tuple(x.shape)))
        B = int(x.shape[0])
        if batch_size is not None and B > int(batch_size):
            raise ValueError("got %d images for batch_size=%d" % (B, batch_size))
        R, L = int(self.rec_rr), int(self.rec_iters)
        n_rows = B * R
        if self.rec_lr_schedule != self._lr_schedule_set:
            self._check(lib.dg_set_option(self._handle, b"lr_schedule", str(self.rec_lr_schedule).encode()))
            self._lr_schedule_set = self.rec_lr_schedule
        z0 = None
        if z_init_val is not None:
            z0 = self._to_device(z_init_val)
            if tuple(z0.shape) != (n_rows, int(self.latent_dim)):
                raise ValueError("z_init_val must be [%d,%d], got %r" % (n_rows, self.latent_dim, tuple(z0.shape)))
        if seed is None:
            seed = self._default_seed + self._calls
        self._calls += 1
        dev = x.device
        rec = torch.empty_like(x)
        idx = torch.empty(B, dtype=torch.int32, device=dev)
        loss = torch.empty(n_rows, dtype=torch.float32, device=dev)
        zout = torch.empty(n_rows, int(self.latent_dim), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            self._check(lib.dg_reconstruct(
                self._handle, x.data_ptr(), z0.data_ptr() if z0 is not None else None,
                int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_row), B, R, L, float(self.rec_lr), float(self.rec_momentum),
                rec.data_ptr(), idx.data_ptr(), loss.data_ptr(), zout.data_ptr(), stream))
        if return_details:
            out = {"rec": rec, "idx": idx, "loss": loss, "z": zout}
            if was_numpy:
                out = {k: v.cpu().numpy() for k, v in out.items()}
            return out
        return rec.cpu().numpy() if was_numpy else rec

    def row_groups(self, batch_size=None) -> int:
        """Number of row groups a ``reconstruct`` of ``batch_size`` images runs as (dg_call_row_groups; option ``two_streams``)."""
        h = self._ensure_handle()
        if not self.initialized:
            raise _native.NativeError("generator weights not loaded (load_generator / set_weights)")
        lib = self._lib()
        n = lib.dg_call_row_groups(h, int(batch_size or self.test_batch_size), int(self.rec_rr))
        if n < 0:
            raise _native.NativeError(lib.dg_last_error().decode())
        return int(n)

    def prepare(self, batch_size=None):
        """Builds everything ``reconstruct`` needs for batches of ``batch_size`` images (workspace, the per-layer job lists
        chosen by timing) ahead of the first call -- the counterpart of the reference building its static graph for
        ``batch_size * rec_rr`` rows (gan.py:345-377).  Optional: an unprepared ``reconstruct`` prepares itself, blocking
        once per new batch size; after ``prepare`` the call only enqueues work on the stream (dg_prepare)."""
        self._ensure_handle()
        if not self.initialized:
            raise _native.NativeError("generator weights not loaded (load_generator / set_weights)")
        torch = _torch()
        dev = torch.device("cuda", self._device)
        B = int(batch_size or self.test_batch_size)
        # DG_TUNING_CACHE=<file>: job-list choices made by an earlier process (the bench, before a profiler pass) are installed
        # instead of being timed again, and this process's own choices are written back -- both runs then launch the same lists
        cache = os.environ.get("DG_TUNING_CACHE")
        if cache and os.path.exists(cache) and not getattr(self, "_tuning_cache_loaded", False):
            self._tuning_cache_loaded = True
            try:
                with open(cache) as fh:
                    self.import_tuning(fh.read())
            except (_native.NativeError, OSError) as e:          # another configuration / device: tune afresh
                if self.verbose:
                    print("[defensegan_amd] %s ignored: %s" % (cache, e))
        with torch.cuda.device(dev):
            self._check(self._lib().dg_prepare(self._handle, B, int(self.rec_rr),
                                                    torch.cuda.current_stream(dev).cuda_stream))
        if cache:
            text = self.export_tuning()
            try:
                old = open(cache).read() if os.path.exists(cache) else None
                if old != text:
                    tmp = "%s.%d.tmp" % (cache, os.getpid())
                    with open(tmp, "w") as fh:
                        fh.write(text)
                    os.replace(tmp, cache)
            except OSError:
                pass

    # ------------------------------------------------------------------ tuning hand-over (dg_export_tuning / dg_import_tuning)
    def export_tuning(self) -> str:
        """The job-list choices of this handle as text (one line per GEMM layer and row count)."""
        self._ensure_handle()
        lib = self._lib()
        need = int(lib.dg_export_tuning(self._handle, None, 0))
        if need < 0:
            self._check(need)
        buf = C.create_string_buffer(need)
        got = int(lib.dg_export_tuning(self._handle, C.cast(buf, C.c_void_p), need))
        if got < 0:
            self._check(got)
        return buf.value.decode()

    def import_tuning(self, text: str) -> int:
        """Installs the choices of another handle's ``export_tuning`` (same architecture, options and CU count) without timing;
        returns the number of lists installed."""
        self._ensure_handle()
        n = int(self._lib().dg_import_tuning(self._handle, text.encode()))
        if n < 0:
            self._check(n)
        return n

    def tuning_id(self) -> str:
        """12 hex digits identifying WHICH job lists are installed (the measured durations are left out): two processes that
        print the same id launch the same lists for every layer."""
        return tuning_text_id(self.export_tuning())

    def reconstruct_dataset(self, splits, checkpoint_dir, batch_size=None, max_num=-1, test_again=False, seed=None):
        """Counterpart of ``reconstruct_dataset`` (gan.py:451-587) for in-memory splits.

        ``splits``: {'train'|'dev'|'test': (images [n,H,W,C] already in generator range, targets [n])}.
        Reconstructions are cached exactly where the reference caches them, so its readers find them:
        ``<checkpoint_dir>/recs_rr{R}_lr{lr:.5f}_iters{L}[_num{max_num}]/<split>/pickles/rec_{i:07d}_l{label}.pkl``
        (one array per image, gan.py:504-557; directory name parsed back by whitebox.py:252-257).  Pickles are
        written by ``py2pickle`` (protocol 2 + ``numpy.core`` module paths) so that the Python-2 reference can load them.  A batch whose pickles all exist is
        loaded instead of recomputed unless ``test_again``.  Returns {split: [recs, targets, originals]} (gan.py:585).
        """
        import pickle
        from . import py2pickle
        bs = int(batch_size or self.test_batch_size)
        if seed is None:
            seed = self._default_seed        # one seed per call: z0 rows are keyed by (seed, global row), not by how many
                                             # batches happened to be computed rather than loaded from the cache
        name = _config.rec_dir_name(int(self.rec_rr), float(self.rec_lr), int(self.rec_iters))
        if max_num > 0:
            name += "_num{:d}".format(max_num)
        rets = {}
        for split, (images, targets) in splits.items():
            out_dir = os.path.join(checkpoint_dir, name, split)
            pk_dir = os.path.join(out_dir, "pickles")
            os.makedirs(pk_dir, exist_ok=True)
            n = len(images) if max_num <= 0 else min(len(images), max_num)
            recs = []
            # a whole-split ``feats.pkl`` (gan.py:484-496) short-cuts the per-image pickles when present
            feats_path = os.path.join(out_dir, "feats.pkl")
            if os.path.exists(feats_path) and not test_again:
                try:
                    with open(feats_path, "rb") as f:
                        allr = np.asarray(pickle.load(f, encoding="latin1"), np.float32).reshape([-1] + list(self.image_dim))
                    if len(allr) >= n:
                        rets[split] = [allr[:n], np.asarray(targets[:n]), np.asarray(images[:n]).reshape([-1] + list(self.image_dim))]
                        continue
                except Exception:
                    pass
            # Caller batches of ``bs`` images decide what is cached (a batch whose pickles all exist is loaded, gan.py:504-557);
            # the batches that have to be computed are handed to the engine in runs of consecutive batches of up to
            # gan_defense.COALESCE_ROWS latent rows -- rows are independent without Batchnorm and z0 is keyed by (seed, global
            # row), so the pickles hold the same bits as one engine call per batch would write, several times sooner at the
            # reference's batch of 50 (tests/test_gpu_cache.py).  With USE_BN a batch is the unit of the statistics: no runs.
            from . import gan_defense as _gd
            per_call = bs if self.use_bn else _gd.engine_batch_images(bs, int(self.rec_rr), max(n, 1))
            path_of = lambda i: os.path.join(pk_dir, "rec_{:07d}_l{}.pkl".format(i, targets[i]))
            loaded = {}
            for start in range(0, n, bs):
                paths = [path_of(i) for i in range(start, min(n, start + bs))]
                if not test_again and all(os.path.exists(q) for q in paths):
                    try:
                        loaded[start] = np.stack([pickle.load(open(q, "rb"), encoding="latin1") for q in paths])
                    except Exception:
                        pass
            start = 0
            while start < n:
                if start in loaded:
                    recs.append(loaded[start])
                    start = min(n, start + bs)
                    continue
                end = min(n, start + bs)
                while end < n and end not in loaded and end - start + min(bs, n - end) <= per_call:
                    end = min(n, end + bs)
                batch = self.reconstruct(np.asarray(images[start:end], np.float32),
                                         seed=seed, first_row=start * int(self.rec_rr))
                batch = np.asarray(batch.cpu().numpy() if hasattr(batch, "cpu") else batch, np.float32)
                for i, r in zip(range(start, end), batch):
                    with open(path_of(i), "wb") as f:
                        py2pickle.dump(r, f)
                recs.append(batch)
                start = end
            all_recs = np.concatenate(recs).reshape([-1] + list(self.image_dim)) if recs else np.zeros([0] + list(self.image_dim), np.float32)
            rets[split] = [all_recs, np.asarray(targets[:n]), np.asarray(images[:n]).reshape([-1] + list(self.image_dim))]
        return rets

    def generate(self, z):
        """G(z): generator_fn(z, is_training=False) (gan.py:399)."""
        self._ensure_handle()
        torch = _torch()
        lib = self._lib()
        was_numpy = isinstance(z, np.ndarray)
        zz = self._to_device(z)
        N = int(zz.shape[0])
        H, W, Cc = self._arch.image_dim
        y = torch.empty(N, H, W, Cc, dtype=torch.float32, device=zz.device)
        with torch.cuda.device(zz.device):
            for r0 in range(0, N, 1 << 20):                   # dg_generate takes at most 2^20 rows per call
                n = min(1 << 20, N - r0)
                self._check(lib.dg_generate(self._handle, zz[r0:r0 + n].data_ptr(), n, y[r0:r0 + n].data_ptr(),
                                              torch.cuda.current_stream(zz.device).cuda_stream))
        return y.cpu().numpy() if was_numpy else y

    def loss_grad(self, images, z):
        """One loop body without the update: (y [B*R,H,W,C], loss [B*R], dz [B*R, latent])
        for images [B,...] and z [B*R, latent] (gan.py:409-417)."""
        self._ensure_handle()
        torch = _torch()
        lib = self._lib()
        was_numpy = isinstance(images, np.ndarray)
        x = self._to_device(images)
        zz = self._to_device(z)
        B = int(x.shape[0])
        R = int(zz.shape[0]) // B
        if R * B != int(zz.shape[0]):
            raise ValueError("z rows must be a multiple of the image count")
        H, W, Cc = self._arch.image_dim
        y = torch.empty(B * R, H, W, Cc, dtype=torch.float32, device=x.device)
        loss = torch.empty(B * R, dtype=torch.float32, device=x.device)
        dz = torch.empty_like(zz)
        with torch.cuda.device(x.device):
            self._check(lib.dg_loss_grad(self._handle, x.data_ptr(), zz.data_ptr(), B, R, y.data_ptr(),
                                           loss.data_ptr(), dz.data_ptr(),
                                           torch.cuda.current_stream(x.device).cuda_stream))
        if was_numpy:
            return y.cpu().numpy(), loss.cpu().numpy(), dz.cpu().numpy()
        return y, loss, dz

    def init_latents(self, n_rows, seed=0, first_row=0, std=None):
        self._ensure_handle()
        torch = _torch()
        lib = self._lib()
        dev = torch.device("cuda", self._device)
        z = torch.empty(int(n_rows), int(self.latent_dim), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            self._check(lib.dg_init_latents(self._handle, z.data_ptr(), int(n_rows), int(seed), int(first_row),
                                              float(std) if std else 0.0,
                                              torch.cuda.current_stream(dev).cuda_stream))
        return z

    # ------------------------------------------------------------------ measurement hooks
    def set_option(self, key: str, value) -> None:
        self._ensure_handle()
        self._check(self._lib().dg_set_option(self._handle, key.encode(), str(value).encode()))

    def profile_enable(self, stride: int) -> None:
        self._ensure_handle()
        self._check(self._lib().dg_profile_enable(self._handle, int(stride)))

    def profile_reset(self) -> None:
        self._ensure_handle()
        self._check(self._lib().dg_profile_reset(self._handle))

    def profile_read(self):
        """[{name, launches, ms, flops}] per kernel family since the last reset."""
        self._ensure_handle()
        lib = self._lib()
        n = lib.dg_profile_count(self._handle)
        out = []
        for i in range(max(n, 0)):
            name = C.create_string_buffer(64)
            launches = C.c_int64()
            ms = C.c_double()
            fl = C.c_double()
            self._check(lib.dg_profile_read(self._handle, i, name, 64, C.byref(launches), C.byref(ms), C.byref(fl)))
            out.append({"name": name.value.decode(), "launches": launches.value, "ms": ms.value, "flops": fl.value})
        return out

    def debug_read(self, what: str, n: int):
        self._ensure_handle()
        torch = _torch()
        dev = torch.device("cuda", self._device)
        torch.cuda.synchronize(dev)
        t = torch.empty(int(n), dtype=torch.float32, device=dev)
        got = self._lib().dg_debug_read(self._handle, what.encode(), t.data_ptr(), int(n))
        if got < 0:
            self._check(int(got))
        return t[:got]


def tuning_text_id(text: str) -> str:
    """12 hex digits naming the job lists a dg_export_tuning text describes.  A record is
    ``op n_rows min_level slack snake xcd_order xcd_head n_jobs measured_us [taper [prio [pair_kernel]]]`` (dg_plan.cpp format_tune_record):
    every field but ``measured_us`` (index 8, informational: two timings of the same list differ) enters the id, the taper
    and the priority mode -- which do change the list -- included; fields an older text lacks count as 0."""
    import hashlib
    lines = text.strip().splitlines()
    keys = []
    for ln in lines[1:]:
        f = ln.split()
        if len(f) < 9:
            keys.append(" ".join(f))         # not a record this build writes: hashed as it is
            continue
        taper = f[9] if len(f) > 9 else "0"
        prio = f[10] if len(f) > 10 else "0"                  # eleventh field (round 5): wave priorities by job length
        pairk = f[11] if len(f) > 11 else "0"                 # twelfth: the list runs the PAIR instantiation of the kernel
        keys.append(" ".join(f[:8] + ["%.17g" % float(taper), str(int(prio)), str(int(pairk))]))
    return hashlib.sha256("\n".join(sorted(keys) if keys else lines).encode()).hexdigest()[:12]


class ReconstructionLayer(object):
    """The classifier-side wrapper of the reference (utils/network_builder.py:239-271): a layer whose forward pass
    is the Defense-GAN projection.  ``model`` is a DefenseGANBase; ``z_init`` an optional fixed [B*R, latent] init
    (whitebox.py --same_init); the reference's ``reconstructor_id=123`` only named TF variables."""

    def __init__(self, model, z_init, input_shape, batch_size):
        self.z_init = z_init
        self.rec_model = model
        self.input_shape = input_shape
        self.batch_size = batch_size
        self.rec = None

    def set_input_shape(self, shape):
        self.input_shape = shape
        self.output_shape = shape

    def get_output_shape(self):
        return self.output_shape

    def fprop(self, x):
        z = self.z_init
        if z is not None:
            z = z[: len(x) * int(self.rec_model.rec_rr)]
        self.rec = self.rec_model.reconstruct(x, batch_size=self.batch_size, back_prop=True, z_init_val=z,
                                              reconstructor_id=123)
        return self.rec


class MnistDefenseGAN(DefenseGANBase):          # gan.py:649-685
    arch_name = "mnist"


class FmnistDefenseDefenseGAN(MnistDefenseGAN):  # gan.py:688-698 (same generator, other weights; name sic)
    arch_name = "f-mnist"


class CelebADefenseGAN(DefenseGANBase):         # gan.py:717-765
    arch_name = "celeba"


# whitebox.py / blackbox.py pick the class by FLAGS.dataset_name
dataset_gan_dict = {
    "mnist": MnistDefenseGAN,
    "f-mnist": FmnistDefenseDefenseGAN,
    "celeba": CelebADefenseGAN,
}


def gan_from_config(cfg_path: str, test_mode: bool = True, **overrides) -> DefenseGANBase:
    """``GAN(cfg=load_config(path), test_mode=True)`` as whitebox.py:239-244 does."""
    cfg = _config.load_config(cfg_path)
    cls = dataset_gan_dict[str(cfg.get("DATASET_NAME", overrides.get("dataset_name", "mnist"))).lower()]
    return cls(cfg=cfg, test_mode=test_mode, **overrides)
