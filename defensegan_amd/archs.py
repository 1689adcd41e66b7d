"""Generator architectures of the Defense-GAN projection path.

Pure shape bookkeeping (no arithmetic): layer lists, weight names/shapes in the
reference's own layouts, MAC counts.  Mirrors

* ``mnist_generator``  -- /root/reference/models/dataset_models.py:36-71
  (F-MNIST reuses it, /root/reference/models/gan.py:688)
* ``celeba_generator`` -- /root/reference/models/dataset_models.py:127-165

Weight layouts are the reference's: ``Linear`` W is ``[in, out]``
(tflib/ops/linear.py:129-142), ``Deconv2D`` filters are ``[5, 5, Cout, Cin]``
(tflib/ops/deconv2d.py:66-74), activations NHWC.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

ARCH_MNIST = 0    # also F-MNIST
ARCH_CELEBA = 1

ARCH_IDS = {"mnist": ARCH_MNIST, "f-mnist": ARCH_MNIST, "fmnist": ARCH_MNIST,
            "mnist28": ARCH_MNIST, "celeba": ARCH_CELEBA, "celeba64": ARCH_CELEBA}

KSIZE = 5   # every Deconv2D in the generators is 5x5, stride 2, SAME


@dataclass(frozen=True)
class Deconv:
    name: str          # reference layer name, e.g. "Generator.2"
    cin: int
    cout: int
    h_in: int          # input spatial extent (square)
    h_out_used: int    # output extent that is consumed downstream (7 after the MNIST crop)
    act: str           # "relu" | "none" | "sigmoid" | "tanh"
    bn: str            # BN layer name applied to this layer's output when use_bn, or ""

    @property
    def h_out(self) -> int:      # extent TF materialises
        return 2 * self.h_in


@dataclass(frozen=True)
class Arch:
    arch_id: int
    name: str
    latent_dim: int
    net_dim: int
    image_dim: Tuple[int, int, int]     # H, W, C
    deconvs: Tuple[Deconv, ...]
    out_act: str                        # "sigmoid" | "tanh"
    in_lo: float                        # generator output range (gan.py:684-685, 764-765)
    in_hi: float

    @property
    def lin_out(self) -> int:
        return 4 * 4 * 4 * self.net_dim

    @property
    def pixels(self) -> int:
        h, w, c = self.image_dim
        return h * w * c


def make_arch(arch, latent_dim: int = 128, net_dim: int = 64) -> Arch:
    if isinstance(arch, str):
        arch = ARCH_IDS[arch.lower()]
    nd = net_dim
    if arch == ARCH_MNIST:
        # dataset_models.py:41-69
        return Arch(ARCH_MNIST, "mnist28", latent_dim, nd, (28, 28, 1), (
            Deconv("Generator.2", 4 * nd, 2 * nd, 4, 7, "relu", "Generator.BN2"),
            Deconv("Generator.3", 2 * nd, nd, 7, 14, "relu", "Generator.BN3"),
            Deconv("Generator.5", nd, 1, 14, 28, "sigmoid", ""),
        ), "sigmoid", 0.0, 1.0)
    if arch == ARCH_CELEBA:
        # dataset_models.py:133-163 ; Generator.5 has no nonlinearity and no BN
        return Arch(ARCH_CELEBA, "celeba64", latent_dim, nd, (64, 64, 3), (
            Deconv("Generator.2", 4 * nd, 2 * nd, 4, 8, "relu", "Generator.BN2"),
            Deconv("Generator.3", 2 * nd, nd, 8, 16, "relu", "Generator.BN3"),
            Deconv("Generator.5", nd, nd, 16, 32, "none", ""),
            Deconv("Generator.6", nd, 3, 32, 64, "tanh", ""),
        ), "tanh", -1.0, 1.0)
    raise ValueError("unknown arch %r" % (arch,))


def weight_shapes(a: Arch, use_bn: bool = False) -> Dict[str, Tuple[int, ...]]:
    """name -> shape in the reference's layouts (names follow the tflib param registry,
    /root/reference/tflib/__init__.py:9-33)."""
    s: Dict[str, Tuple[int, ...]] = {
        "Generator.Input.W": (a.latent_dim, a.lin_out),
        "Generator.Input.b": (a.lin_out,),
    }
    for d in a.deconvs:
        s[d.name + ".Filters"] = (KSIZE, KSIZE, d.cout, d.cin)
        s[d.name + ".Biases"] = (d.cout,)
    if use_bn:
        s["Generator.BN1.scale"] = (a.lin_out,)
        s["Generator.BN1.offset"] = (a.lin_out,)
        for d in a.deconvs:
            if d.bn:
                s[d.bn + ".scale"] = (d.cout,)
                s[d.bn + ".offset"] = (d.cout,)
    return s


def valid_taps_1d(h_in: int, h_out_used: int) -> int:
    """Number of (o, k) pairs with 0 <= 2*o + k - 1 < h_out_used, 0 <= o < h_in, 0 <= k < 5."""
    n = 0
    for o in range(h_in):
        for k in range(KSIZE):
            i = 2 * o + k - 1
            if 0 <= i < h_out_used:
                n += 1
    return n


def fwd_macs_per_row(a: Arch) -> int:
    """Crop-aware forward MACs per latent row: 16 572 992 for MNIST,
    50 226 880 for CelebA."""
    m = a.latent_dim * a.lin_out
    for d in a.deconvs:
        t = valid_taps_1d(d.h_in, d.h_out_used)
        m += t * t * d.cin * d.cout
    return m


def flop_per_image(a: Arch, R: int, L: int) -> float:
    """Algorithmic FLOP per projected image: L forwards, L-1 useful backwards
    (the L-th update is dead work in the reference, gan.py:409-445)."""
    return float(R) * (2 * L - 1) * 2.0 * fwd_macs_per_row(a)
