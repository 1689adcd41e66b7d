"""-m gpu: the fragment-order forward path (dg_fgemm.hip, option frag_path; default on) against the position-batched
kernels of dg_gemm.hip (frag_path = 0) and the float64 oracle.

The two paths compute the same products; the fragment-order kernel splits the K axis of a tile over up to four waves (a fixed
tree of k-ordered chains per tap class), so the results agree to float32 rounding, not bit for bit; each path is deterministic and
independent of the batch a row is in (bit-identical across row counts)."""
import numpy as np
import pytest

from defensegan_amd import archs, synth

pytestmark = pytest.mark.gpu


FRAG_MODES = (1, 2)        # 1: one workgroup per job (K split over waves), 2: persistent waves walking their own tile lists (no K split)


def _make(arch, frag, R=2, L=3, gain=2.0, seed=1234, lr=10.0):
    from defensegan_amd.gan import dataset_gan_dict
    gan = dataset_gan_dict[arch](cfg={"USE_BN": False, "LATENT_DIM": 128, "NET_DIM": 64}, test_mode=True,
                                 rec_rr=R, rec_iters=L, rec_lr=lr)
    p = synth.make_weights(arch, seed=seed, gain=gain, bias_range=0.1)
    assert gan.set_weights(p) == []
    gan.set_option("frag_path", frag)
    return gan, p


def _np(v):
    return np.asarray(v.cpu().numpy() if hasattr(v, "cpu") else v)


@pytest.mark.parametrize("frag", FRAG_MODES)
@pytest.mark.parametrize("arch", ["mnist", "celeba"])
@pytest.mark.parametrize("n", [1, 37, 64, 500])
def test_activations_loss_and_gradient_match_the_position_batched_path(arch, n, frag):
    a = archs.make_arch(arch)
    g1, p = _make(arch, frag)
    g0, _ = _make(arch, 0)
    rs = np.random.RandomState(n)
    z = (rs.standard_normal((n, 128)) * 0.15).astype(np.float32)
    x = rs.uniform(a.in_lo, a.in_hi, size=(n,) + tuple(a.image_dim)).astype(np.float32)
    y1, y0 = _np(g1.generate(z)), _np(g0.generate(z))
    sizes = {"mnist": [4096, 7 * 7 * 128, 14 * 14 * 64], "celeba": [4096, 8 * 8 * 128, 16 * 16 * 64, 32 * 32 * 64]}[arch]
    for d, sz in enumerate(sizes):
        a1 = _np(g1.debug_read("act%d" % d, n * sz)).reshape(n, sz)
        a0 = _np(g0.debug_read("act%d" % d, n * sz)).reshape(n, sz)
        scale = max(1.0, float(np.abs(a0).max()))
        assert np.abs(a1 - a0).max() <= 2e-6 * scale, (d, np.abs(a1 - a0).max(), scale)
        assert ((a1 > 0) == (a0 > 0)).mean() > 0.9999
    np.testing.assert_allclose(y1, y0, rtol=0, atol=2e-5)            # (pre-activations of up to 1600 products, another summation tree)
    got = [_np(v) for v in g1.loss_grad(x, z)]
    ref = [_np(v) for v in g0.loss_grad(x, z)]
    np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-5)
    scale = np.abs(ref[2]).max()
    # rows whose ReLU gates differ between two equally valid summation orders are allowed to depart (a pre-activation within
    # rounding of zero); everything else agrees to rounding
    err = np.abs(got[2] - ref[2]).max(axis=1) / scale
    assert (err < 1e-5).mean() >= 0.9, err
    assert (err < 0.05).all(), err.max()


@pytest.mark.parametrize("frag", FRAG_MODES)
@pytest.mark.parametrize("arch,B,R", [("mnist", 50, 10), ("celeba", 13, 10)])
def test_projection_matches_and_rows_are_batch_independent(arch, B, R, frag):
    a = archs.make_arch(arch)
    # (CelebA at the reference's lr = 10 is chaotic from the first steps on -- DESIGN section 2 --: compared where the loop contracts)
    lr = 10.0 if arch == "mnist" else 2.0
    g1, p = _make(arch, frag, R=R, L=3, lr=lr)
    g0, _ = _make(arch, 0, R=R, L=3, lr=lr)
    rs = np.random.RandomState(5)
    x = _np(g0.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32)))
    x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=6)
    z0 = synth.make_z(B * R, 128, seed=7)
    d1 = {k: _np(v) for k, v in g1.reconstruct(x, z_init_val=z0, return_details=True).items()}
    d0 = {k: _np(v) for k, v in g0.reconstruct(x, z_init_val=z0, return_details=True).items()}
    np.testing.assert_allclose(d1["loss"], d0["loss"], rtol=2e-4)
    # (three GD steps on adversarial targets amplify the rounding; CelebA's tanh images span [-1, 1] and its loop is the touchier one)
    assert np.abs(d1["rec"] - d0["rec"]).max() < (3e-3 if arch == "mnist" else 3e-2)
    assert (d1["idx"] == d0["idx"]).mean() >= 0.9
    # the same images as part of a smaller call: bit-identical rows (the K split is a constant of the tap class)
    h = B // 2
    d2 = {k: _np(v) for k, v in g1.reconstruct(x[:h], z_init_val=z0[:h * R], return_details=True).items()}
    for k in ("rec", "idx"):
        assert np.array_equal(d2[k], d1[k][:h]), k
    assert np.array_equal(d2["loss"], d1["loss"][:h * R])
    assert np.array_equal(d2["z"], d1["z"][:h * R])
