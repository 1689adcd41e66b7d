"""Independent PyTorch-CPU (autograd) formulation of the projection path -- TEST INFRASTRUCTURE ONLY.

Second, differently-built restatement used (a) to cross-check oracle/defensegan_oracle.py and (b) as
the ``cpu_baseline`` ("port") timed by bench.py on the GPU box's host cores.  It mirrors the
reference's TF graph op for op (MatMul+BiasAdd, Conv2DBackpropInput, Relu, StridedSlice, Sigmoid/Tanh,
autodiff to z only, ApplyMomentum) -- /root/reference/models/gan.py:333-449,
/root/reference/models/dataset_models.py:36-71,127-165 -- but through a different code path than the
NumPy oracle: ``conv_transpose2d(stride=2, padding=1)`` (which yields 2h+1 outputs with
i = 2*o + k - 1) cropped to 2h, NCHW tensors, and ``torch.autograd.grad`` instead of hand-written
backward formulas.  PARITY UNPINNED (see defensegan_oracle.py header).  Never imported by the product.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as Fn


def _layers(arch: str):
    if arch in ("mnist", "mnist28", "f-mnist", "fmnist"):
        return [("Generator.2", 7, "relu", "Generator.BN2"), ("Generator.3", 14, "relu", "Generator.BN3"),
                ("Generator.5", 28, "sigmoid", "")]
    return [("Generator.2", 8, "relu", "Generator.BN2"), ("Generator.3", 16, "relu", "Generator.BN3"),
            ("Generator.5", 32, "none", ""), ("Generator.6", 64, "tanh", "")]


class TorchGenerator:
    def __init__(self, params: Dict[str, np.ndarray], arch: str = "mnist", use_bn: bool = False,
                 dtype=torch.float32):
        self.arch, self.use_bn, self.dtype = arch, use_bn, dtype
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
        self.W = t(params["Generator.Input.W"])
        self.b = t(params["Generator.Input.b"])
        self.layers = []
        for name, used, act, bn in _layers(arch):
            F = t(params[name + ".Filters"])                  # [kh,kw,Cout,Cin]
            w = F.permute(3, 2, 0, 1).contiguous()            # torch conv_transpose2d: [Cin,Cout,kh,kw]
            self.layers.append((w, t(params[name + ".Biases"]), used, act, bn))
        self.bn = {k: t(v) for k, v in params.items() if ".BN" in k}

    def _bn(self, a, name, dims):
        mean = a.mean(dim=dims, keepdim=True)
        var = ((a - mean) ** 2).mean(dim=dims, keepdim=True)
        shape = [1, -1] + [1] * (a.dim() - 2)
        return (a - mean) * torch.rsqrt(var + 1e-5) * self.bn[name + ".scale"].view(shape) + \
            self.bn[name + ".offset"].view(shape)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z [N,latent] -> y [N,H,W,C] (NHWC, like the reference)."""
        a = z @ self.W + self.b
        if self.use_bn:
            a = self._bn(a, "Generator.BN1", (0,))
        h = torch.relu(a)
        c0 = self.W.shape[1] // 16
        h = h.view(-1, 4, 4, c0).permute(0, 3, 1, 2)          # NHWC reshape, then to NCHW
        for (w, b, used, act, bn) in self.layers:
            hin = h.shape[-1]
            a = Fn.conv_transpose2d(h, w, bias=b, stride=2, padding=1)[..., :2 * hin, :2 * hin]
            if self.use_bn and bn:
                a = self._bn(a, bn, (0, 2, 3))
            if act == "relu":
                h = torch.relu(a)[..., :used, :used]
            elif act == "none":
                h = a
            elif act == "sigmoid":
                h = torch.sigmoid(a)
            else:
                h = torch.tanh(a)
        return h.permute(0, 2, 3, 1)


def reconstruct(params, x: np.ndarray, z0: np.ndarray, R: int, L: int, lr: float = 10.0,
                momentum: float = 0.7, arch: str = "mnist", use_bn: bool = False,
                dtype=torch.float32, gen: TorchGenerator = None, loss_at=None):
    """Same contract as defensegan_oracle.reconstruct (gan.py:333-449).

    loss_at: optional horizons L' <= L; the result then also holds "loss_at" = {L': image_rec_loss of every restart as a run
    with rec_iters = L' would return it} (the loss of forward k = L' - 1 does not depend on what follows it)."""
    g = gen or TorchGenerator(params, arch, use_bn, dtype)
    xt = torch.from_numpy(np.ascontiguousarray(x)).to(dtype).repeat_interleave(R, dim=0)
    z = torch.from_numpy(np.ascontiguousarray(z0)).to(dtype).clone()
    m = torch.zeros_like(z)
    steps = max(L, 1)
    at = {}
    for k in range(steps):
        z.requires_grad_(True)
        y = g.forward(z)
        loss_rows = ((y - xt) ** 2).flatten(1).mean(dim=1)      # reduce_mean over H,W,C
        if loss_at and (k + 1) in loss_at:
            at[k + 1] = loss_rows.detach().numpy().copy()
        if k == steps - 1:
            z = z.detach()
            break
        (grad,) = torch.autograd.grad(loss_rows.sum(), z)       # rec_loss = reduce_sum(image_rec_loss)
        with torch.no_grad():
            m = momentum * m + grad
            z = (z - lr * m).detach()
    loss = loss_rows.detach()
    B = x.shape[0]
    idx = loss.view(B, R).argmin(dim=1)                          # first minimum
    rows = torch.arange(B) * R + idx
    y = y.detach()
    out = {"rec": y[rows].numpy().reshape(x.shape), "idx": idx.numpy().astype(np.int32),
           "loss": loss.numpy(), "z": z.numpy(), "y": y.numpy()}
    if loss_at:
        out["loss_at"] = at
    return out
