"""Generates the golden fixtures under tests/golden/ from the CPU oracle (float64 arithmetic, stored
as float32 inputs / float64 expected outputs).

The reference itself (Python 2 + TF 1.7) cannot run in the build container, so these vectors pin the
HIP path to the ORACLE's restatement of /root/reference/models/gan.py:333-449 -- "parity unpinned"
with respect to the reference's own binaries (see oracle/defensegan_oracle.py header).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from defensegan_amd import archs, synth          # noqa: E402
from oracle import defensegan_oracle as O        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, arch, wseed, gain, bias_range, B, R, L, lr, adversarial, zseed):
    a = archs.make_arch(arch)
    p = synth.make_weights(arch, seed=wseed, gain=gain, bias_range=bias_range)
    rs = np.random.RandomState(zseed)
    zt = (rs.standard_normal((B, a.latent_dim)) * np.sqrt(1.0 / a.latent_dim)).astype(np.float32)
    x, _ = O.generator_forward(p, zt.astype(np.float64), arch)
    x = x.astype(np.float32)
    if adversarial:
        x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=zseed + 1)
    z0 = (rs.standard_normal((B * R, a.latent_dim)) * np.sqrt(1.0 / a.latent_dim)).astype(np.float32)
    if R >= 3:
        z0[2] = z0[1]            # duplicated restart: exercises the first-minimum tie-break
    out = O.reconstruct(p, x, z0, R, L, lr=lr, momentum=0.7, arch=arch, dtype=np.float64, trace=True)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        arch=arch, wseed=wseed, gain=gain, bias_range=bias_range, R=R, L=L, lr=lr, momentum=0.7,
        x=x, z0=z0, rec=out["rec"], idx=out["idx"], loss=out["loss"], z=out["z"],
        loss_trace=np.stack(out["loss_trace"]))
    print(name, "loss first/last", out["loss_trace"][0][:3], out["loss"][:3], "idx", out["idx"])


if __name__ == "__main__":
    make("mnist_clean_L5", "mnist", 1234, 2.0, 0.1, B=4, R=3, L=5, lr=10.0, adversarial=False, zseed=11)
    make("mnist_adv_L3", "mnist", 1234, 3.0, 0.1, B=3, R=2, L=3, lr=10.0, adversarial=True, zseed=12)
    make("fmnist_clean_L4", "f-mnist", 4321, 2.0, 0.0, B=2, R=4, L=4, lr=10.0, adversarial=False, zseed=13)
    if "--celeba" in sys.argv:
        make("celeba_clean_L3", "celeba", 1234, 2.0, 0.1, B=2, R=2, L=3, lr=10.0, adversarial=False, zseed=14)
