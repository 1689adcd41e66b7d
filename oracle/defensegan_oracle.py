"""CPU ORACLE for the Defense-GAN latent-projection path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference algorithm.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; the product
path (``defensegan_amd``) never imports it and fails loudly when the HIP library is missing.

PARITY UNPINNED.  The reference (/root/reference, Python 2 + TensorFlow 1.7) holds no tests, golden
vectors or fixtures for this path, and neither Python 2 nor TensorFlow exists in the build container,
so the reference itself cannot be run to generate vectors.  The arithmetic lives in TensorFlow 1.7
(not vendored; un-pinned in /root/reference/requirements.txt): tf.matmul, tf.nn.conv2d_transpose,
tf.nn.moments / tf.nn.batch_normalization, tf.train.MomentumOptimizer, tf.while_loop, tf.argmin.
Their published semantics are restated here and anchored on the reference's call sites:

  reconstruct()            /root/reference/models/gan.py:333-449
  learning-rate schedule   /root/reference/models/base_model.py:153-194 (constant in effect: the
                           step variable ``rec_iter_const`` is never assigned, gan.py:362-368)
  mnist_generator          /root/reference/models/dataset_models.py:36-71
  celeba_generator         /root/reference/models/dataset_models.py:127-165
  Linear                   /root/reference/tflib/ops/linear.py:129-142
  Deconv2D                 /root/reference/tflib/ops/deconv2d.py:100-117
  Batchnorm (else-branch)  /root/reference/tflib/ops/batchnorm.py:80-93

The restatement is made trustworthy by (tests/test_oracle.py): a literal pure-Python loop version
of every primitive on tiny shapes, the adjoint identity <conv_same_s2(x), y> = <x, deconv(y)> against
TF's documented SAME-padding rule, finite differences of dL/dz in float64, and an independent
PyTorch-autograd formulation (oracle/torch_ref.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

KS = 5


# --------------------------------------------------------------------------------------------
# layer primitives
# --------------------------------------------------------------------------------------------
def linear(z: np.ndarray, W: np.ndarray, b: np.ndarray) -> np.ndarray:
    """x @ W[in,out] + b   (tflib/ops/linear.py:129-142)."""
    return z @ W + b


def _tap_slices(k: int, h_in: int, h_out: int):
    """For kernel index k: input range [o0,o1) and output start i0 such that i = 2*o + k - 1 is
    inside [0, h_out).  TF SAME, stride 2, k=5: pad_total = 3, pad_before = 1."""
    o0 = 0
    while 2 * o0 + k - 1 < 0:
        o0 += 1
    o1 = h_in
    while o1 > o0 and 2 * (o1 - 1) + k - 1 >= h_out:
        o1 -= 1
    return o0, o1, 2 * o0 + k - 1


def deconv2d(x: np.ndarray, F: np.ndarray, b: Optional[np.ndarray], h_out: Optional[int] = None) -> np.ndarray:
    """tf.nn.conv2d_transpose(x, F[5,5,Cout,Cin], [N,2h,2w,Cout], strides 2, 'SAME') + bias
    (tflib/ops/deconv2d.py:100-117):
        y[n,i,j,co] = b[co] + sum x[n,oh,ow,ci] * F[kh,kw,co,ci],  i = 2*oh+kh-1, j = 2*ow+kw-1
    (no kernel flip: conv2d_transpose is the gradient of conv2d w.r.t. its input).
    ``h_out`` < 2h computes only the leading h_out x h_out outputs (the MNIST 7x7 crop)."""
    n, h, w, cin = x.shape
    cout = F.shape[2]
    ho = 2 * h if h_out is None else h_out
    wo = 2 * w if h_out is None else h_out
    y = np.zeros((n, ho, wo, cout), dtype=x.dtype)
    for kh in range(KS):
        a0, a1, i0 = _tap_slices(kh, h, ho)
        if a1 <= a0:
            continue
        for kw in range(KS):
            c0, c1, j0 = _tap_slices(kw, w, wo)
            if c1 <= c0:
                continue
            contrib = x[:, a0:a1, c0:c1, :] @ F[kh, kw].T.astype(x.dtype)      # [.., Cout]
            y[:, i0:i0 + 2 * (a1 - a0):2, j0:j0 + 2 * (c1 - c0):2, :] += contrib
    if b is not None:
        y += b
    return y


def deconv2d_backward_input(dy: np.ndarray, F: np.ndarray, h_in: int) -> np.ndarray:
    """Gradient of deconv2d w.r.t. x = a stride-2 SAME forward conv (TF op Conv2D):
        dx[n,oh,ow,ci] = sum dy[n,2*oh+kh-1,2*ow+kw-1,co] * F[kh,kw,co,ci]   over valid indices.
    dy may be the cropped (h_out_used) map; rows/cols beyond it are treated as zero."""
    n, ho, wo, cout = dy.shape
    cin = F.shape[3]
    dx = np.zeros((n, h_in, h_in, cin), dtype=dy.dtype)
    for kh in range(KS):
        a0, a1, i0 = _tap_slices(kh, h_in, ho)
        if a1 <= a0:
            continue
        for kw in range(KS):
            c0, c1, j0 = _tap_slices(kw, h_in, wo)
            if c1 <= c0:
                continue
            g = dy[:, i0:i0 + 2 * (a1 - a0):2, j0:j0 + 2 * (c1 - c0):2, :]
            dx[:, a0:a1, c0:c1, :] += g @ F[kh, kw].astype(dy.dtype)            # [Cout,Cin]
    return dx


BN_EPS = 1e-5


def batchnorm_fwd(a: np.ndarray, scale: np.ndarray, offset: np.ndarray, axes):
    """tflib/ops/batchnorm.py:80-93 (the branch the generator takes): batch statistics always,
    biased variance, eps 1e-5."""
    mean = a.mean(axis=axes, keepdims=True)
    var = ((a - mean) ** 2).mean(axis=axes, keepdims=True)
    rstd = 1.0 / np.sqrt(var + a.dtype.type(BN_EPS))
    xhat = (a - mean) * rstd
    return xhat * scale + offset, (xhat, rstd)


def batchnorm_bwd(dy: np.ndarray, scale: np.ndarray, cache, axes) -> np.ndarray:
    xhat, rstd = cache
    m1 = dy.mean(axis=axes, keepdims=True)
    m2 = (dy * xhat).mean(axis=axes, keepdims=True)
    return (scale * rstd) * (dy - m1 - xhat * m2)


# --------------------------------------------------------------------------------------------
# generators
# --------------------------------------------------------------------------------------------
def _arch_layers(arch: str):
    """(name, h_in, h_out_used, act, bn_name) per Deconv2D."""
    if arch in ("mnist", "mnist28", "f-mnist", "fmnist"):
        return [("Generator.2", 4, 7, "relu", "Generator.BN2"),
                ("Generator.3", 7, 14, "relu", "Generator.BN3"),
                ("Generator.5", 14, 28, "sigmoid", "")]
    if arch in ("celeba", "celeba64"):
        return [("Generator.2", 4, 8, "relu", "Generator.BN2"),
                ("Generator.3", 8, 16, "relu", "Generator.BN3"),
                ("Generator.5", 16, 32, "none", ""),
                ("Generator.6", 32, 64, "tanh", "")]
    raise ValueError(arch)


def generator_forward(p: Dict[str, np.ndarray], z: np.ndarray, arch: str = "mnist",
                      use_bn: bool = False):
    """Returns (y[N,H,W,C], cache).  dataset_models.py:36-71 / 127-165."""
    dt = z.dtype
    cache: Dict[str, object] = {}
    W = p["Generator.Input.W"].astype(dt)
    a = linear(z, W, p["Generator.Input.b"].astype(dt))
    if use_bn:
        a, cache["bn1"] = batchnorm_fwd(a, p["Generator.BN1.scale"].astype(dt),
                                        p["Generator.BN1.offset"].astype(dt), (0,))
    pre = [a]                                     # pre-activations (tests use them to spot ReLU kinks)
    h = np.maximum(a, 0)
    cin0 = W.shape[1] // 16
    h = h.reshape(-1, 4, 4, cin0)                 # feature f = (oh*4+ow)*C + c   (NHWC reshape)
    acts = [h]
    for (name, h_in, h_used, act, bn) in _arch_layers(arch):
        F = p[name + ".Filters"].astype(dt)
        b = p[name + ".Biases"].astype(dt)
        x_in = acts[-1]
        bn_on = use_bn and bn != ""
        if bn_on:
            # BN statistics are over the FULL 2h x 2h map, the crop comes after ReLU
            a = deconv2d(x_in, F, b, None)
            a, cache["bn_" + name] = batchnorm_fwd(a, p[bn + ".scale"].astype(dt),
                                                   p[bn + ".offset"].astype(dt), (0, 1, 2))
        else:
            a = deconv2d(x_in, F, b, h_used)       # crop-aware: cropped outputs never influence anything
        pre.append(a[:, :h_used, :h_used, :])
        if act == "relu":
            hh = np.maximum(a, 0)[:, :h_used, :h_used, :]
        elif act == "none":
            hh = a
        elif act == "sigmoid":
            hh = 1.0 / (1.0 + np.exp(-a))
        elif act == "tanh":
            hh = np.tanh(a)
        else:
            raise ValueError(act)
        acts.append(hh)
    cache["acts"] = acts
    cache["pre"] = pre
    return acts[-1], cache


def generator_backward(p: Dict[str, np.ndarray], cache, dy: np.ndarray, arch: str = "mnist",
                       use_bn: bool = False) -> np.ndarray:
    """dL/dz given dL/dy (what TF autodiff yields for var_list=[z_hat], gan.py:416-417)."""
    dt = dy.dtype
    acts = cache["acts"]
    layers = _arch_layers(arch)
    g = dy
    for li in range(len(layers) - 1, -1, -1):
        name, h_in, h_used, act, bn = layers[li]
        out = acts[li + 1]
        if act == "sigmoid":
            g = g * out * (1 - out)
        elif act == "tanh":
            g = g * (1 - out * out)
        elif act == "relu":
            g = g * (out > 0)                      # ReluGrad masks on the output; 0 at exactly 0
        F = p[name + ".Filters"].astype(dt)
        if use_bn and bn != "":
            full = 2 * h_in
            gp = np.zeros((g.shape[0], full, full, g.shape[3]), dt)
            gp[:, :h_used, :h_used, :] = g         # gradient of the crop = zero padding
            g = batchnorm_bwd(gp, p[bn + ".scale"].astype(dt), cache["bn_" + name], (0, 1, 2))
        g = deconv2d_backward_input(g, F, h_in)
    g = g.reshape(g.shape[0], -1)
    h1 = acts[0].reshape(g.shape[0], -1)
    g = g * (h1 > 0)
    if use_bn:
        g = batchnorm_bwd(g, p["Generator.BN1.scale"].astype(dt), cache["bn1"], (0,))
    return g @ p["Generator.Input.W"].astype(dt).T


# --------------------------------------------------------------------------------------------
# the projection loop
# --------------------------------------------------------------------------------------------
def reconstruct(p: Dict[str, np.ndarray], x: np.ndarray, z0: np.ndarray, R: int, L: int,
                lr: float = 10.0, momentum: float = 0.7, arch: str = "mnist",
                use_bn: bool = False, dtype=np.float32, trace: bool = False, lr_schedule: str = "constant"):
    """DefenseGANBase.reconstruct (gan.py:333-449).

    lr_schedule: "constant" = what the reference EXECUTES: the decay's step variable `rec_iter_const` is never advanced
    (gan.py:362-386, 416-417), so exponential_decay always sees step 0 and lr == rec_lr.  "intended" = what the code asks
    for (gan.py:380-386 -> base_model.py:186-192): tf.train.exponential_decay(rec_lr, step, ceil(0.8 * rec_iters), 0.1,
    staircase=True) with step = the loop counter, i.e. iteration k uses rec_lr * 0.1 ** floor(k / ceil(0.8 * L)).

    x  [B,H,W,C]; z0 [B*R, latent] with row j = b*R + r (gan.py:348-359).
    Schedule (gan.py:409-437): for k in 0..L-1: y_k = G(z_k), loss_k, then the momentum update
    -> z_{k+1}.  The loop returns y_{L-1}, loss_{L-1}; the L-th update is dead work.  L == 0
    returns G(z_0) and its loss (the pre-loop graph, gan.py:399-406).
    Selection (gan.py:438-449): first argmin over the R restarts of each image.

    Returns dict(rec[B,H,W,C], idx[B] (restart index 0..R-1), loss[B*R], z[B*R,latent] = z_{L-1},
    y[B*R,H,W,C]) (+ per-step traces).
    """
    dt = np.dtype(dtype).type
    B = x.shape[0]
    P = int(np.prod(x.shape[1:]))
    xt = np.repeat(x.astype(dtype), R, axis=0)          # tile: row b*R+r holds image b
    z = z0.astype(dtype).copy()
    m = np.zeros_like(z)                                # momentum slot, zero-initialised
    lr_t, mom_t = dt(lr), dt(momentum)
    if lr_schedule not in ("constant", "intended"):
        raise ValueError("lr_schedule must be 'constant' or 'intended'")
    decay_iter = int(np.ceil(L * 0.8)) if L > 0 else 1
    zs: List[np.ndarray] = []
    losses: List[np.ndarray] = []
    steps = max(L, 1)
    for k in range(steps):
        y, cache = generator_forward(p, z, arch, use_bn)
        d = y - xt
        loss = (d * d).reshape(d.shape[0], -1).mean(axis=1)      # image_rec_loss, gan.py:411-413
        if trace:
            zs.append(z.copy())
            losses.append(loss.copy())
        if k == steps - 1:
            break                                               # last update is discarded
        dy = (dt(2.0) / dt(P)) * d                               # d(sum_j mean_pix)/dy
        g = generator_backward(p, cache, dy, arch, use_bn)
        m = mom_t * m + g                                        # ApplyMomentum (non-Nesterov)
        lr_k = lr_t if lr_schedule == "constant" else dt(np.float32(lr) * np.float32(0.1) ** np.float32(k // decay_iter))
        z = z - lr_k * m
    idx = np.empty(B, np.int32)
    for b in range(B):
        idx[b] = int(np.argmin(loss[b * R:(b + 1) * R]))        # first minimum
    rows = np.arange(B) * R + idx
    out = {"rec": y[rows].reshape(x.shape), "idx": idx, "loss": loss, "z": z, "y": y}
    if trace:
        out["z_trace"] = zs
        out["loss_trace"] = losses
    return out


# --------------------------------------------------------------------------------------------
# literal loop versions (tiny shapes only) used to pin the vectorised primitives
# --------------------------------------------------------------------------------------------
def deconv2d_literal(x, F, b, h_out=None):
    n, h, w, cin = x.shape
    cout = F.shape[2]
    ho = 2 * h if h_out is None else h_out
    y = np.zeros((n, ho, ho, cout), np.float64)
    for nn in range(n):
        for oh in range(h):
            for ow in range(w):
                for kh in range(KS):
                    for kw in range(KS):
                        i, j = 2 * oh + kh - 1, 2 * ow + kw - 1
                        if 0 <= i < ho and 0 <= j < ho:
                            for co in range(cout):
                                for ci in range(cin):
                                    y[nn, i, j, co] += float(x[nn, oh, ow, ci]) * float(F[kh, kw, co, ci])
    if b is not None:
        y += np.asarray(b, np.float64)
    return y


def conv2d_same_s2_literal(y, F, h_in):
    """TF Conv2D, stride 2, SAME, on input y[N,Ho,Wo,Cout] with filter viewed as
    [kh,kw,in=Cout... the forward conv whose input-gradient conv2d_transpose is defined as.
    For input extent Ho=2*h_in, k=5, s=2: out = ceil(Ho/2) = h_in, pad_total = max((h_in-1)*2+5-Ho,0)=3,
    pad_before = pad_total//2 = 1  =>  out[o] = sum_k in[2*o + k - 1] * w[k]."""
    n, ho, wo, cout = y.shape
    cin = F.shape[3]
    out = np.zeros((n, h_in, h_in, cin), np.float64)
    for nn in range(n):
        for oh in range(h_in):
            for ow in range(h_in):
                for kh in range(KS):
                    for kw in range(KS):
                        i, j = 2 * oh + kh - 1, 2 * ow + kw - 1
                        if 0 <= i < ho and 0 <= j < wo:
                            for co in range(cout):
                                for ci in range(cin):
                                    out[nn, oh, ow, ci] += float(y[nn, i, j, co]) * float(F[kh, kw, co, ci])
    return out
