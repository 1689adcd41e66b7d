"""Host-side mirror of the reference's classifier container (/root/reference/utils/network_builder.py:129-331, models
A-F :333-521) for the step AFTER the projection -- SURVEY.md section 8f row N2.  The layers only describe the network;
the arithmetic runs in the HIP library (dg_clf_* in include/defensegan_hip.h), never on the host.

    model = model_a()                      # same builders, same argument names as the reference
    model.set_weights([(kernels, b), ...]) # one (W, b) per Conv2D / Linear layer, reference layouts
    probs = model(x)                       # = get_probs(x); x NumPy or torch [B,H,W,C]; returns the same kind
    model.add_rec_model(gan, z_init, batch_size)   # prepend the Defense-GAN projection (network_builder.py:179-183)

Differences from the reference that follow from having no TF graph: ``fprop`` exposes only 'logits' and 'probs' (and
'reconstruction' after add_rec_model), not every hidden layer; Dropout is the identity (evaluation phase)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native

KIND = {"Conv2D": 0, "ReLU": 1, "Linear": 2, "Flatten": 3, "Softmax": 4, "Dropout": 5}


class Layer(object):
    has_params = False

    def spec(self) -> Tuple[int, ...]:
        return (KIND[self.__class__.__name__], 0, 0, 0, 0, 0, 0)


class Conv2D(Layer):
    """network_builder.py:206-236: tf.nn.conv2d(x, kernels[kh,kw,cin,cout], (1,)+strides+(1,), padding) + b."""
    has_params = True

    def __init__(self, output_channels, kernel_shape, strides, padding):
        assert padding in ("SAME", "VALID")
        self.output_channels, self.kernel_shape, self.strides, self.padding = int(output_channels), tuple(kernel_shape), tuple(strides), padding

    def spec(self):
        return (KIND["Conv2D"], self.output_channels, self.kernel_shape[0], self.kernel_shape[1], self.strides[0], self.strides[1],
                1 if self.padding == "SAME" else 0)


class Linear(Layer):
    """network_builder.py:190-203: tf.matmul(x, W[in,out]) + b."""
    has_params = True

    def __init__(self, num_hid):
        self.num_hid = int(num_hid)

    def spec(self):
        return (KIND["Linear"], self.num_hid, 0, 0, 0, 0, 0)


class ReLU(Layer):
    pass


class Flatten(Layer):
    pass


class Softmax(Layer):
    pass


class Dropout(Layer):
    """Identity at evaluation (tf.cond(K.learning_phase(), ...), network_builder.py:296-297)."""

    def __init__(self, prob):
        self.prob = prob


class MLP(object):
    """network_builder.py:129-183."""

    def __init__(self, layers: Sequence[Layer], input_shape=(None, 28, 28, 1), rec_model=None, device: int = 0):
        self.layers = list(layers)
        self.input_shape = tuple(input_shape)
        self.rec_model = rec_model
        self.rec_layer = None
        self._device = int(device)
        self._handle = None
        self.layer_names: List[str] = []
        for i, layer in enumerate(self.layers):
            self.layer_names.append(layer.__class__.__name__ + str(i))
        if isinstance(self.layers[-1], Softmax):
            self.layer_names[-1], self.layer_names[-2] = "probs", "logits"
        else:
            self.layer_names[-1] = "logits"
        self._param_layers = [i for i, l in enumerate(self.layers) if l.has_params]
        self._weights_set = False

    # ------------------------------------------------------------------ native handle
    def _ensure(self):
        if self._handle is not None:
            return
        lib = _native.load()
        _, H, W, Cc = self.input_shape
        h = C.c_void_p()
        _native.check(lib.dg_clf_create(self._device, int(H), int(W), int(Cc), C.byref(h)))
        self._handle = h
        self._native_index = []
        for layer in self.layers:
            rc = lib.dg_clf_add_layer(h, *layer.spec())
            if rc < 0:
                _native.check(rc)
            self._native_index.append(rc)
        self.nb_classes = int(lib.dg_clf_output_width(h))

    def close(self):
        if self._handle is not None:
            _native.load().dg_clf_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_layer_names(self):
        return (["reconstruction"] if self.rec_layer is not None else []) + self.layer_names

    def set_weights(self, params: Sequence[Tuple[np.ndarray, np.ndarray]]) -> None:
        """One (W, b) per Conv2D / Linear layer, in order; kernels [kh,kw,cin,cout], W [in,out] (reference layouts)."""
        self._ensure()
        if len(params) != len(self._param_layers):
            raise ValueError("expected %d (W, b) pairs, got %d" % (len(self._param_layers), len(params)))
        lib = _native.load()
        for li, (W, b) in zip(self._param_layers, params):
            W = np.ascontiguousarray(W, np.float32)
            b = np.ascontiguousarray(b, np.float32)
            shp = (C.c_int64 * W.ndim)(*W.shape)
            _native.check(lib.dg_clf_set_weights(self._handle, self._native_index[li], W.ctypes.data_as(C.c_void_p), shp, W.ndim,
                                                 b.ctypes.data_as(C.c_void_p), b.size, 0))
        self._weights_set = True

    def init_like_reference(self, seed: int = 0) -> List[Tuple[np.ndarray, np.ndarray]]:
        """The reference's initialisers (normal, normalised per output unit, zero bias: network_builder.py:196-203, 217-224)
        with a NumPy stream; returns and installs the parameters.  For tests and synthetic benchmarks."""
        rs = np.random.RandomState(seed)
        _, H, W, Cc = self.input_shape
        shape, flat, params = (H, W, Cc), None, []
        for layer in self.layers:
            if isinstance(layer, Conv2D):
                kh, kw = layer.kernel_shape
                k = rs.standard_normal((kh, kw, shape[2], layer.output_channels)).astype(np.float32)
                k = k / np.sqrt(1e-7 + np.square(k).sum(axis=(0, 1, 2)))
                params.append((k.astype(np.float32), np.zeros(layer.output_channels, np.float32)))
                shape = conv_output_shape(shape, layer)
            elif isinstance(layer, Flatten):
                flat = int(np.prod(shape))
            elif isinstance(layer, Linear):
                w = rs.standard_normal((flat, layer.num_hid)).astype(np.float32)
                w = w / np.sqrt(1e-7 + np.square(w).sum(axis=0, keepdims=True))
                params.append((w.astype(np.float32), np.zeros(layer.num_hid, np.float32)))
                flat = layer.num_hid
        self.set_weights(params)
        return params

    # ------------------------------------------------------------------ forward
    def add_rec_model(self, model, z_init, batch_size):
        """network_builder.py:179-183: prepend the Defense-GAN projection."""
        from .gan import ReconstructionLayer
        self.rec_layer = ReconstructionLayer(model, z_init, self.input_shape, batch_size)

    def model_eval(self, test_images, test_labels, batch_size: int, **kw):
        """Accuracy of this classifier over ``test_images`` -- behind the Defense-GAN projection when ``add_rec_model`` installed
        one -- as ``(correct, n, roc_info)``: ``gan_defense.model_eval_gan`` with the layer's model, ``z_init`` and ``rec_rr``.  THE
        way to evaluate a defended classifier here: ``fprop`` / ``get_probs`` project whatever single batch they are handed in ONE
        engine call and cannot coalesce across calls (the reference's ``ReconstructionLayer.fprop`` sits inside one session.run per
        BATCH_SIZE = 50 images, utils/network_builder.py:266-271: 500 latent rows, where the loop runs at 0.67 of the peak), while
        this routes the same per-batch semantics through runs of whole batches (0.85)."""
        from . import gan_defense
        rl = self.rec_layer
        if rl is None:
            return gan_defense.model_eval_gan(None, self, test_images, test_labels, batch_size, **kw)
        kw.setdefault("same_init_z", rl.z_init)
        return gan_defense.model_eval_gan(rl.rec_model.reconstruct, self, test_images, test_labels, batch_size,
                                          rec_rr=int(rl.rec_model.rec_rr), **kw)

    def _forward(self, x, no_rec=False):
        import torch
        self._ensure()
        if not self._weights_set:
            raise _native.NativeError("classifier weights not set")
        was_numpy = isinstance(x, np.ndarray)
        dev = torch.device("cuda", self._device)
        t = torch.from_numpy(np.ascontiguousarray(x, np.float32)) if was_numpy else x
        t = t.to(device=dev, dtype=torch.float32).contiguous()
        rec = None
        if self.rec_layer is not None and not no_rec:
            rec = self.rec_layer.fprop(t)
            t = rec if not isinstance(rec, np.ndarray) else torch.from_numpy(rec).to(dev)
            t = t.to(device=dev, dtype=torch.float32).contiguous()
        B = int(t.shape[0])
        logits = torch.empty(B, self.nb_classes, dtype=torch.float32, device=dev)
        probs = torch.empty_like(logits)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _native.check(_native.load().dg_clf_forward(self._handle, t.data_ptr(), B, logits.data_ptr(), probs.data_ptr(), stream))
        out = {"logits": logits, "probs": probs}
        if rec is not None:
            out["reconstruction"] = t
        if was_numpy:
            torch.cuda.synchronize(dev)
            out = {k: v.cpu().numpy() for k, v in out.items()}
        return out

    def fprop(self, x, set_ref=False, no_rec=False):
        return self._forward(x, no_rec=no_rec)

    def get_logits(self, x):
        return self._forward(x)["logits"]

    def get_probs(self, x):
        return self._forward(x)["probs"]

    def __call__(self, x):
        return self.get_probs(x)

    def eval_batch(self, rec, orig=None, labels=None, sync=True):
        """One batch of model_eval_gan on the device (gan_defense.py:113-179): returns (n_correct, preds [B], diffs [B] or None);
        with ``sync=False`` n_correct stays a device tensor [1] (int32) and the call does not wait for the stream.
        ``rec`` = the classifier's input (reconstructions), ``orig`` = the images they are compared with (diff_op,
        blackbox.py:569-572), ``labels`` int class indices."""
        import torch
        self._ensure()
        dev = torch.device("cuda", self._device)
        to = lambda a, dt: (torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a).to(device=dev, dtype=dt).contiguous()
        r = to(rec, torch.float32)
        B = int(r.shape[0])
        o = to(orig, torch.float32) if orig is not None else None
        lab = to(labels, torch.int32) if labels is not None else None
        preds = torch.empty(B, dtype=torch.int32, device=dev)
        diffs = torch.empty(B, dtype=torch.float32, device=dev) if o is not None else None
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _native.check(_native.load().dg_eval_batch(
                self._handle, r.data_ptr(), o.data_ptr() if o is not None else None, lab.data_ptr() if lab is not None else None, B,
                preds.data_ptr(), diffs.data_ptr() if diffs is not None else None, cnt.data_ptr(), stream))
        if sync:
            return int(cnt.item()), preds, diffs
        return cnt, preds, diffs          # device tensors: nothing waits for the stream (gan_defense.model_eval_gan sums at the end)


def conv_output_shape(shape, layer: Conv2D):
    H, W, _ = shape
    (kh, kw), (sh, sw) = layer.kernel_shape, layer.strides
    if layer.padding == "SAME":
        return (-(-H // sh), -(-W // sw), layer.output_channels)
    return ((H - kh) // sh + 1, (W - kw) // sw + 1, layer.output_channels)


# ---------------------------------------------------------------------- the reference's model zoo (network_builder.py:333-521)
# Architectures are data: (channels, kernel, stride, padding) per Conv2D+ReLU stage, hidden widths per Linear+ReLU(+Dropout)
# stage; the builders keep the reference's names and signatures.
def _conv_zoo(convs, hidden, nb_classes, input_shape, rec_model, drop_after_flatten=None, drop_input=None, drop_before_flatten=None,
              hidden_dropout=True):
    layers: List[Layer] = []
    if drop_input is not None:
        layers.append(Dropout(drop_input))
    for (ch, k, s, pad) in convs:
        layers += [Conv2D(ch, (k, k), (s, s), pad), ReLU()]
    if drop_before_flatten is not None:
        layers.append(Dropout(drop_before_flatten))
    layers.append(Flatten())
    if drop_after_flatten is not None:
        layers.append(Dropout(drop_after_flatten))
    for h in hidden:
        layers += [Linear(h), ReLU()] + ([Dropout(0.5)] if hidden_dropout else [])
    layers += [Linear(nb_classes), Softmax()]
    return MLP(layers, input_shape, rec_model=rec_model)


_BF_CONVS = lambda f: [(f, 8, 2, "SAME"), (2 * f, 6, 2, "VALID"), (2 * f, 5, 1, "VALID")]


def model_f(nb_filters=64, nb_classes=10, input_shape=(None, 28, 28, 1), rec_model=None):
    return _conv_zoo(_BF_CONVS(nb_filters), [], nb_classes, input_shape, rec_model)


def model_b(nb_filters=64, nb_classes=10, input_shape=(None, 28, 28, 1), rec_model=None):
    return _conv_zoo(_BF_CONVS(nb_filters), [], nb_classes, input_shape, rec_model, drop_input=0.2, drop_before_flatten=0.5)


def model_e(input_shape=(None, 28, 28, 1), nb_classes=10):
    return _conv_zoo([], [200, 200], nb_classes, input_shape, None, hidden_dropout=False)


def model_d(input_shape=(None, 28, 28, 1), nb_classes=10):
    m = _conv_zoo([], [200, 200], nb_classes, input_shape, None, hidden_dropout=False)
    layers = list(m.layers)
    layers.insert(3, Dropout(0.5))          # Flatten, Linear(200), ReLU, Dropout(0.5), Linear(200), ReLU, Linear, Softmax
    return MLP(layers, input_shape)


def model_a(nb_filters=64, nb_classes=10, input_shape=(None, 28, 28, 1), rec_model=None):
    return _conv_zoo([(nb_filters, 5, 1, "SAME"), (nb_filters, 5, 2, "VALID")], [128], nb_classes, input_shape, rec_model, 0.25)


def model_c(nb_filters=64, nb_classes=10, input_shape=(None, 28, 28, 1), rec_model=None):
    return _conv_zoo([(nb_filters * 2, 3, 1, "SAME"), (nb_filters, 5, 2, "VALID")], [128], nb_classes, input_shape, rec_model, 0.25)


def model_y(nb_filters=64, nb_classes=10, input_shape=(None, 28, 28, 1), rec_model=None):
    return _conv_zoo([(nb_filters, 3, 1, "SAME"), (nb_filters, 3, 2, "VALID"), (2 * nb_filters, 3, 2, "VALID"),
                      (2 * nb_filters, 3, 2, "VALID")], [256, 256], nb_classes, input_shape, rec_model)


def model_q(nb_filters=32, nb_classes=10, input_shape=(None, 28, 28, 1), rec_model=None):
    return _conv_zoo([(nb_filters, 3, 1, "SAME"), (nb_filters, 3, 2, "VALID"), (2 * nb_filters, 3, 1, "VALID"),
                      (2 * nb_filters, 3, 2, "VALID")], [256, 256], nb_classes, input_shape, rec_model)


def model_z(nb_filters=32, nb_classes=10, input_shape=(None, 28, 28, 1), rec_model=None):
    return _conv_zoo([(nb_filters, 3, 1, "SAME"), (nb_filters, 3, 2, "VALID"), (2 * nb_filters, 3, 1, "VALID"),
                      (2 * nb_filters, 3, 2, "VALID"), (4 * nb_filters, 3, 1, "VALID"), (4 * nb_filters, 3, 2, "VALID")],
                     [600, 600], nb_classes, input_shape, rec_model)


MODELS = {"A": model_a, "B": model_b, "C": model_c, "D": model_d, "E": model_e, "F": model_f, "Y": model_y, "Q": model_q,
          "Z": model_z}


# ---------------------------------------------------------------------- the step BEFORE the path: FGSM
_REC_GRADIENT_NOTE = (
    "the model has the Defense-GAN reconstruction layer attached (add_rec_model): the reference differentiates "
    "THROUGH ReconstructionLayer (whitebox.py:185-214, network_builder.py:266-271), whose gradient w.r.t. its input is "
    "identically zero (the projected latents live in variables updated by ApplyMomentum; the selected restart comes "
    "from an argmin) -- the input gradient is 0 and FGSM returns clip(x).  Build the attack before add_rec_model (or "
    "pass no_rec=True) to differentiate the bare classifier.")


def _mlp_input_gradient(self, x, labels=None, no_rec=False):
    """d(sum_b CE(softmax(logits_b), y_b))/dx on the device; y = ``labels`` (class indices) or the model's own prediction.
    With the reconstruction layer attached the result is the reference's: zeros (see _REC_GRADIENT_NOTE)."""
    import torch
    if self.rec_layer is not None and not no_rec:
        import warnings
        warnings.warn(_REC_GRADIENT_NOTE, stacklevel=2)
        return np.zeros_like(x) if isinstance(x, np.ndarray) else torch.zeros_like(x)
    self._ensure()
    if not self._weights_set:
        raise _native.NativeError("classifier weights not set")
    was_numpy = isinstance(x, np.ndarray)
    dev = torch.device("cuda", self._device)
    t = (torch.from_numpy(np.ascontiguousarray(x, np.float32)) if was_numpy else x).to(device=dev, dtype=torch.float32).contiguous()
    lab = None
    if labels is not None:
        lab = (torch.from_numpy(np.ascontiguousarray(labels)) if isinstance(labels, np.ndarray) else labels).to(device=dev, dtype=torch.int32).contiguous()
    g = torch.empty_like(t)
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        _native.check(_native.load().dg_clf_input_gradient(self._handle, t.data_ptr(), lab.data_ptr() if lab is not None else None,
                                                           int(t.shape[0]), g.data_ptr(), stream))
    return g.cpu().numpy() if was_numpy else g


MLP.input_gradient = _mlp_input_gradient


class FastGradientMethod(object):
    """The cleverhans attack object the reference instantiates (whitebox.py:198-200, blackbox.py:530-534), ord = inf only:
    ``adv = clip(x + eps * sign(grad_x CE(model(x), y)), clip_min, clip_max)``; without ``y`` the model's own prediction is
    the label (cleverhans' default).  ``sess`` / ``back`` are accepted for signature compatibility and ignored.

    White-box use on a DEFENDED model (whitebox.py:185-200 attaches the reconstruction layer first, then builds the attack
    on that model): the gradient through the projection is identically zero in the reference, so ``generate`` returns
    ``clip(x, clip_min, clip_max)`` -- reproduced here, with a warning."""

    def __init__(self, model: MLP, back="tf", sess=None):
        self.model = model

    def generate(self, x, eps=0.3, ord=np.inf, y=None, clip_min=None, clip_max=None, **kwargs):
        import torch
        if ord not in (np.inf, "inf", float("inf")):
            raise NotImplementedError("only ord = inf (the reference's setting) is implemented")
        m = self.model
        m._ensure()
        was_numpy = isinstance(x, np.ndarray)
        dev = torch.device("cuda", m._device)
        t = (torch.from_numpy(np.ascontiguousarray(x, np.float32)) if was_numpy else x).to(device=dev, dtype=torch.float32).contiguous()
        lo = float("-inf") if clip_min is None else float(clip_min)
        hi = float("inf") if clip_max is None else float(clip_max)
        if m.rec_layer is not None:
            import warnings
            warnings.warn(_REC_GRADIENT_NOTE, stacklevel=2)
            out = torch.clamp(t, lo, hi)                 # x + eps * sign(0)
            return out.cpu().numpy() if was_numpy else out
        lab = None
        if y is not None:
            yy = y if isinstance(y, np.ndarray) else y.detach().cpu().numpy()
            if yy.ndim > 1:
                yy = yy.argmax(axis=-1)                            # one-hot labels as cleverhans takes them
            lab = torch.from_numpy(np.ascontiguousarray(yy)).to(device=dev, dtype=torch.int32)
        out = torch.empty_like(t)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _native.check(_native.load().dg_fgsm(m._handle, t.data_ptr(), lab.data_ptr() if lab is not None else None,
                                                 int(t.shape[0]), float(eps), lo, hi, out.data_ptr(), stream))
        return out.cpu().numpy() if was_numpy else out


def rand_fgsm_prestep(test_images, eps: float, alpha: float, min_val: float = 0.0, max_val: float = 1.0, rng=None):
    """The ``rand`` half of ``--attack_type rand+fgsm`` (/root/reference/whitebox.py:191-195): one random sign step of size
    ``alpha`` before the FGSM step, whose budget shrinks by it:

        x' = clip(x + alpha * sign(N(0, 1)), min_val, 1),   eps' = eps - alpha

    Returns ``(x', eps')``; feed both to ``FastGradientMethod.generate``.  ``rng``: a ``numpy.random.RandomState`` (the reference
    draws from the global NumPy generator it seeded with [11, 24, 1990], whitebox.py:143-144)."""
    rng = np.random if rng is None else rng
    x = np.asarray(test_images, np.float32)
    out = np.clip(x + np.float32(alpha) * np.sign(rng.randn(*x.shape)).astype(np.float32), min_val, max_val).astype(np.float32)
    return out, float(eps) - float(alpha)
