"""CPU restatement of the reference classifier forward -- TEST INFRASTRUCTURE ONLY, like the rest of
oracle/: only tests/ may import it; the product path runs the HIP kernels of defensegan_amd/csrc/dg_clf.hip.

Follows /root/reference/utils/network_builder.py:190-331 (Linear: x @ W + b; Conv2D: tf.nn.conv2d(x, kernels[kh,kw,cin,
cout], strides, padding) + b; ReLU; Flatten = NHWC row-major reshape; Softmax; Dropout = identity at evaluation) and the
evaluation reduction of /root/reference/utils/gan_defense.py:91-179 (argmax preds, correct count) with diff_op of
/root/reference/blackbox.py:569-572.  The arithmetic lives in TensorFlow 1.7 (absent, unpinned): tf.nn.conv2d is a
cross-correlation with SAME = (out = ceil(in/stride), pad_before = pad_total // 2) -- "parity unpinned", pinned here only
against torch.nn.functional.conv2d with explicit asymmetric padding and hand-computed shapes (tests/test_classifier.py)."""
import numpy as np


def same_padding(n_in: int, k: int, s: int):
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return out, total // 2, total - total // 2


def conv2d(x: np.ndarray, K: np.ndarray, b: np.ndarray, strides, padding: str) -> np.ndarray:
    """x [B,H,W,Cin], K [kh,kw,Cin,Cout] -> [B,Ho,Wo,Cout]."""
    B, H, W, Cin = x.shape
    kh, kw, cin, cout = K.shape
    assert cin == Cin
    sh, sw = strides
    if padding == "SAME":
        Ho, pt, pb = same_padding(H, kh, sh)
        Wo, pl, pr = same_padding(W, kw, sw)
    else:
        Ho, Wo, pt, pb, pl, pr = (H - kh) // sh + 1, (W - kw) // sw + 1, 0, 0, 0, 0
    xp = np.zeros((B, H + pt + pb, W + pl + pr, Cin), x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    y = np.zeros((B, Ho, Wo, cout), x.dtype)
    for a in range(kh):
        for c in range(kw):
            patch = xp[:, a:a + (Ho - 1) * sh + 1:sh, c:c + (Wo - 1) * sw + 1:sw, :]
            y += patch @ K[a, c].astype(x.dtype)
    return y + b.astype(x.dtype)


def forward(layers, params, x: np.ndarray):
    """layers: list of ("conv", cout, (kh,kw), (sh,sw), padding) | ("relu",) | ("flatten",) | ("linear", n) | ("softmax",) |
    ("dropout",); params: one (W, b) per conv/linear.  Returns (logits, probs)."""
    h = x
    it = iter(params)
    logits = None
    for L in layers:
        kind = L[0]
        if kind == "conv":
            W, b = next(it)
            h = conv2d(h, W, b, L[3], L[4])
        elif kind == "linear":
            W, b = next(it)
            h = h @ W.astype(h.dtype) + b.astype(h.dtype)
        elif kind == "relu":
            h = np.maximum(h, 0)
        elif kind == "flatten":
            h = h.reshape(len(h), -1)
        elif kind == "softmax":
            logits = h
            e = np.exp(h - h.max(axis=1, keepdims=True))
            h = e / e.sum(axis=1, keepdims=True)
        elif kind == "dropout":
            pass
        else:
            raise ValueError(kind)
    return (logits if logits is not None else h), h


def eval_batch(probs: np.ndarray, labels: np.ndarray, rec: np.ndarray, orig: np.ndarray):
    preds = probs.argmax(axis=1)
    diffs = ((orig - rec) ** 2).reshape(len(rec), -1).mean(axis=1)
    return int((preds == labels).sum()), preds, diffs


# ---------------------------------------------------------------------------------------------- input gradient / FGSM
def conv2d_backward_input(g: np.ndarray, K: np.ndarray, x_shape, strides, padding: str) -> np.ndarray:
    """d/dx of sum(g * conv2d(x, K)): scatter form of the same index map as conv2d above."""
    B, H, W, Cin = x_shape
    kh, kw, _, cout = K.shape
    sh, sw = strides
    if padding == "SAME":
        Ho, pt, pb = same_padding(H, kh, sh)
        Wo, pl, pr = same_padding(W, kw, sw)
    else:
        Ho, Wo, pt, pb, pl, pr = (H - kh) // sh + 1, (W - kw) // sw + 1, 0, 0, 0, 0
    dxp = np.zeros((B, H + pt + pb, W + pl + pr, Cin), g.dtype)
    for a in range(kh):
        for c in range(kw):
            dxp[:, a:a + (Ho - 1) * sh + 1:sh, c:c + (Wo - 1) * sw + 1:sw, :] += g @ K[a, c].astype(g.dtype).T
    return dxp[:, pt:pt + H, pl:pl + W, :]


def input_gradient(layers, params, x: np.ndarray, labels=None):
    """d(sum_b CE(softmax(logits_b), y_b))/dx; y = labels or the model's own first argmax (cleverhans fgm default)."""
    acts = [x]
    it = iter(params)
    used = []
    for L in layers:
        h = acts[-1]
        kind = L[0]
        if kind == "conv":
            W, b = next(it); used.append((W, b)); h = conv2d(h, W, b, L[3], L[4])
        elif kind == "linear":
            W, b = next(it); used.append((W, b)); h = h @ W.astype(h.dtype) + b.astype(h.dtype)
        elif kind == "relu":
            h = np.maximum(h, 0)
        elif kind == "flatten":
            h = h.reshape(len(h), -1)
        elif kind == "softmax":
            break
        acts.append(h)
    logits = acts[-1]
    y = np.asarray(labels) if labels is not None else logits.argmax(axis=1)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    g = e / e.sum(axis=1, keepdims=True)
    g[np.arange(len(g)), y] -= 1.0
    body = [L for L in layers if L[0] != "softmax"]
    pi = len(used)
    for li in range(len(body) - 1, -1, -1):
        L, xin, out = body[li], acts[li], acts[li + 1]
        kind = L[0]
        if kind == "conv":
            pi -= 1
            g = conv2d_backward_input(g, used[pi][0], xin.shape, L[3], L[4])
        elif kind == "linear":
            pi -= 1
            g = g @ used[pi][0].astype(g.dtype).T
        elif kind == "relu":
            g = g * (out > 0)
        elif kind == "flatten":
            g = g.reshape(xin.shape)
    return g


def fgsm(layers, params, x, eps, clip_min, clip_max, labels=None):
    g = input_gradient(layers, params, x, labels)
    return np.clip(x + eps * np.sign(g), clip_min, clip_max), g
