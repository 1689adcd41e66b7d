"""-m "not gpu": the oracle (float32 and float64) reproduces the committed golden vectors, and the
independent torch formulation agrees with them too."""
import numpy as np
import pytest

from defensegan_amd import synth
from oracle import defensegan_oracle as O
from tests.helpers import load_golden


@pytest.mark.parametrize("name", ["mnist_clean_L5", "mnist_adv_L3", "fmnist_clean_L4", "celeba_clean_L3"])
def test_oracle_reproduces_golden(name):
    g = load_golden(name)
    p = synth.make_weights(g["arch"], seed=g["wseed"], gain=g["gain"], bias_range=g["bias_range"])
    o64 = O.reconstruct(p, g["x"], g["z0"], g["R"], g["L"], lr=g["lr"], momentum=g["momentum"], arch=g["arch"],
                        dtype=np.float64)
    np.testing.assert_allclose(o64["rec"], g["rec"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(o64["loss"], g["loss"], rtol=1e-12)
    assert (o64["idx"] == g["idx"]).all()
    o32 = O.reconstruct(p, g["x"], g["z0"], g["R"], g["L"], lr=g["lr"], momentum=g["momentum"], arch=g["arch"],
                        dtype=np.float32)
    np.testing.assert_allclose(o32["rec"], g["rec"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(o32["loss"], g["loss"], rtol=3e-4)
    assert (o32["idx"] == g["idx"]).all()


def test_torch_formulation_reproduces_golden():
    import torch
    from oracle import torch_ref as T
    g = load_golden("mnist_clean_L5")
    p = synth.make_weights(g["arch"], seed=g["wseed"], gain=g["gain"], bias_range=g["bias_range"])
    t = T.reconstruct(p, g["x"], g["z0"], g["R"], g["L"], lr=g["lr"], momentum=g["momentum"], arch=g["arch"],
                      dtype=torch.float64)
    np.testing.assert_allclose(t["rec"], g["rec"], rtol=0, atol=1e-10)
    assert (t["idx"] == g["idx"]).all()
