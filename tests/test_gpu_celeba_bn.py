"""-m gpu: parity of the CelebA generator path and of the use_bn=True variants with the CPU oracle."""
import numpy as np
import pytest

from defensegan_amd import archs, synth
from tests.helpers import load_golden, make_gan

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import defensegan_oracle as O
    return O


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _targets(arch, B, seed):
    a = archs.make_arch(arch)
    rs = np.random.RandomState(seed)
    return (rs.rand(B, *a.image_dim) * (a.in_hi - a.in_lo) + a.in_lo).astype(np.float32)


@pytest.mark.parametrize("N", [3, 70])
def test_celeba_generate_layers_vs_oracle(N):
    O = _oracle()
    gan, p = make_gan("celeba", bias_range=0.1)
    rs = np.random.RandomState(N)
    z = (rs.standard_normal((N, 128)) * 0.3).astype(np.float32)
    y = gan.generate(z)
    yo, cache = O.generator_forward(p, z.astype(np.float64), "celeba")
    acts = cache["acts"]
    for d in range(4):
        got = gan.debug_read("act%d" % d, acts[d].size).cpu().numpy().reshape(acts[d].shape)
        assert _rel(got, acts[d]) < 3e-6, (d, _rel(got, acts[d]))
    assert y.shape == (N, 64, 64, 3)
    np.testing.assert_allclose(y, yo, rtol=0, atol=1e-5)


@pytest.mark.parametrize("B,R", [(2, 2), (5, 3)])
def test_celeba_loss_and_gradient_vs_oracle(B, R):
    O = _oracle()
    gan, p = make_gan("celeba", gain=2.0, bias_range=0.1)
    x = _targets("celeba", B, 5)
    rs = np.random.RandomState(B * 10 + R)
    z = (rs.standard_normal((B * R, 128)) * 0.2).astype(np.float32)
    y, loss, dz = gan.loss_grad(x, z)
    yo, cache = O.generator_forward(p, z.astype(np.float64), "celeba")
    xt = np.repeat(x.astype(np.float64), R, axis=0)
    lo = ((yo - xt) ** 2).reshape(B * R, -1).mean(axis=1)
    go = O.generator_backward(p, cache, 2.0 / 12288 * (yo - xt), "celeba")
    np.testing.assert_allclose(y, yo, rtol=0, atol=1e-5)
    np.testing.assert_allclose(loss, lo, rtol=1e-5)
    kink = np.zeros(B * R, bool)
    for a_pre in cache["pre"][:3]:
        kink |= (np.abs(a_pre).reshape(B * R, -1).min(axis=1) < 1e-6)
    err = np.abs(dz - go).max(axis=1) / np.abs(go).max()
    assert (err[~kink] < 2e-5).all(), err[~kink].max()
    assert (err < 0.2).all()


def test_celeba_reconstruct_matches_golden():
    g = load_golden("celeba_clean_L3")
    gan, p = make_gan("celeba", wseed=g["wseed"], gain=g["gain"], bias_range=g["bias_range"],
                      rec_rr=g["R"], rec_iters=g["L"], rec_lr=g["lr"])
    out = gan.reconstruct(g["x"], z_init_val=g["z0"], return_details=True)
    np.testing.assert_allclose(out["loss"], g["loss"], rtol=3e-4)
    np.testing.assert_allclose(out["z"], g["z"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(out["rec"], g["rec"], rtol=0, atol=5e-5)
    assert (out["idx"] == g["idx"]).all()


@pytest.mark.parametrize("arch,B,R", [("mnist", 6, 3), ("celeba", 6, 3), ("mnist", 64, 3)])
def test_bn_forward_backward_vs_oracle(arch, B, R):
    """use_bn=True: batch statistics at inference (batchnorm.py:80-93) couple all B*R rows.  The 192-row case has more
    than 32 * 1024 statistics rows in the last BN layer (37 632): row blocks longer than the minimum, 1018 partial sums
    per channel in the finalize pass."""
    O = _oracle()
    gan, p = make_gan(arch, gain=2.0, bias_range=0.1, use_bn=True)
    a = archs.make_arch(arch)
    x = _targets(arch, B, 3)
    rs = np.random.RandomState(9)
    z = (rs.standard_normal((B * R, 128)) * 0.3).astype(np.float32)
    y, loss, dz = gan.loss_grad(x, z)
    yo, cache = O.generator_forward(p, z.astype(np.float64), arch, True)
    xt = np.repeat(x.astype(np.float64), R, axis=0)
    lo = ((yo - xt) ** 2).reshape(B * R, -1).mean(axis=1)
    go = O.generator_backward(p, cache, 2.0 / a.pixels * (yo - xt), arch, True)
    np.testing.assert_allclose(y, yo, rtol=0, atol=2e-5)
    np.testing.assert_allclose(loss, lo, rtol=5e-5)
    # a ReLU kink anywhere perturbs every row through the statistics: compare in aggregate
    row_err = np.abs(dz - go).max(axis=1) / np.abs(go).max()
    if B * R <= 32:
        assert _rel(dz, go) < 5e-3, _rel(dz, go)
        assert np.median(row_err) < 5e-5, np.median(row_err)
    else:
        # 192 rows x 50 176 gated activations: some gate sits within float32 rounding of zero.  The float32 run of the
        # NumPy oracle on this input differs from its float64 run by 1.3e-2 in one row (a flipped gate), by <= 1.3e-4 in
        # every other row (all rows move a little: the flip shifts the batch statistics), median 8e-5
        assert np.median(row_err) < 3e-4, np.median(row_err)
        assert (row_err < 1e-3).mean() >= 0.98 and row_err.max() < 5e-2, np.sort(row_err)[-4:]


def test_bn_reconstruct_short_horizon_vs_oracle():
    O = _oracle()
    B, R, L = 4, 2, 3
    gan, p = make_gan("mnist", gain=2.0, bias_range=0.1, use_bn=True, rec_rr=R, rec_iters=L, rec_lr=1.0)
    x = _targets("mnist", B, 4)
    z0 = synth.make_z(B * R, 128, seed=8)
    out = gan.reconstruct(x, z_init_val=z0, return_details=True)
    ref = O.reconstruct(p, x, z0, R, L, lr=1.0, momentum=0.7, arch="mnist", use_bn=True, dtype=np.float64)
    np.testing.assert_allclose(out["loss"], ref["loss"], rtol=2e-3)
    np.testing.assert_allclose(out["rec"], ref["rec"], rtol=0, atol=2e-3)
    assert (out["idx"] == ref["idx"]).all()


def test_bn_mnist_crop_positions_are_zeroed():
    """MNIST + BN keeps the 8x8 map (statistics cover the cropped row/column); the gradient of the crop is zero
    padding, which the backward writes explicitly."""
    B, R = 3, 2
    gan, p = make_gan("mnist", use_bn=True)
    x = _targets("mnist", B, 1)
    z = synth.make_z(B * R, 128, seed=2)
    gan.loss_grad(x, z)
    act1 = gan.debug_read("act1", B * R * 8 * 8 * 128).cpu().numpy().reshape(B * R, 8, 8, 128)
    assert np.isfinite(act1).all()


@pytest.mark.parametrize("arch,B,R", [("mnist", 64, 3), ("celeba", 30, 3), ("mnist", 7, 2)])
def test_bn_statistics_from_the_gemm_epilogue(arch, B, R):
    """Round 5: with USE_BN the forward statistics come from the producing GEMM's epilogue (per-32-row-block column sums of the
    pre-activations, EPI_BIAS_STATS) instead of a pass over them.  (i) Against the separate pass (option bn_fused = 0, float64
    sums from the first add): the statistics differ only by the float32 rounding of a 32-row block sum, so one loop body agrees
    to 1e-5 of its scale; (ii) a block is a FIXED set of 32 rows of the class's M axis summed in a fixed order, so the result does
    not depend on the job list that ran: other starting levels / cutting thresholds give the same bits."""
    a = archs.make_arch(arch)
    x = _targets(arch, B, 13)
    rs = np.random.RandomState(14)
    z = (rs.standard_normal((B * R, 128)) * 0.3).astype(np.float32)

    def run(opts):
        gan, p = make_gan(arch, gain=2.0, bias_range=0.1, use_bn=True)
        for k, v in opts.items():
            gan.set_option(k, v)
        return gan.loss_grad(x, z)
    y1, l1, d1 = run({})
    y0, l0, d0 = run({"bn_fused": 0})
    dy = np.abs(np.asarray(y1) - np.asarray(y0))
    assert (dy <= 1e-5).mean() >= 0.9999 and dy.max() < 2e-3, (float((dy > 1e-5).mean()), float(dy.max()))    # (measured: 6 of 1.1 M pixels at 1.2e-5)
    np.testing.assert_allclose(l1, l0, rtol=1e-5)
    err = np.abs(d1 - d0).max(axis=1) / np.abs(d0).max()
    assert np.median(err) < 1e-5 and (err < 1e-4).mean() >= 0.95, np.sort(err)[-3:]       # (a ReLU gate at rounding distance may move a row)
    # round 6: the default (bn_fused = 2) also takes the BACKWARD sums (dy, dy * xhat) of the layers normalised over rows x positions
    # in the epilogue of the GEMM that writes dy (EPI_MASK_STATS; the gate is re-formed from the pre-activations).  Against
    # bn_fused = 1 (backward sums by a float64 pass): the same forward bits, the gradient to the rounding of 32-row float32 sums
    yf, lf, df = run({"bn_fused": 1})
    assert np.array_equal(yf, y1) and np.array_equal(lf, l1)
    errf = np.abs(d1 - df).max(axis=1) / np.abs(df).max()
    assert np.median(errf) < 1e-5 and errf.max() < 1e-4, np.sort(errf)[-3:]
    for opts in ({"jobs.min_level": 1, "jobs.tune": 0}, {"jobs.slack": 1e30, "jobs.min_level": 2}, {"jobs.slack": 0.01, "jobs.min_level": 0}):
        y2, l2, d2 = run(opts)
        assert np.array_equal(y2, y1) and np.array_equal(l2, l1) and np.array_equal(d2, d1), opts


@pytest.mark.parametrize("B,R,pipe", [(4, 2, 4), (7, 3, 10), (64, 10, 256)])
def test_bn_mnist_tail_in_its_batchnorm_form(B, R, pipe):
    """Round 6 (bn_fused = 2): inside the projection loop the MNIST tail reads the PRE-ACTIVATIONS of Generator.3's Batchnorm layer,
    applies relu(bn(.)) itself (the activation image is never written) and leaves that layer's backward sums (dg_tail_mnist.hip
    mnist_tail_pipe3_kernel<64, true>; it runs when a launch has >= 2 rows per workgroup of the persistent kernel: `tail_pipe`).
    Against bn_fused = 1 (forward apply pass + float64 backward statistics pass) on the same inputs: the forward bits are the same
    expression, the sums differ by float32 rounding -- three loop steps agree to 1e-5 of z's scale; and against the float64 oracle
    on the smallest case."""
    x = _targets("mnist", B, 21)
    z0 = synth.make_z(B * R, 128, seed=22)

    def run(fused):
        gan, p = make_gan("mnist", gain=2.0, bias_range=0.1, use_bn=True, rec_rr=R, rec_iters=3, rec_lr=1.0)
        gan.set_option("tail_pipe", pipe)
        gan.set_option("bn_fused", fused)
        return gan.reconstruct(x, z_init_val=z0, return_details=True), p
    o2, p = run(2)
    o1, _ = run(1)
    # (batch statistics couple all rows and a ReLU gate at rounding distance flips: most of the batch agrees to 1e-5, a few rows move)
    np.testing.assert_allclose(o2["loss"], o1["loss"], rtol=5e-3)
    assert np.median(np.abs(np.asarray(o2["loss"]) / np.asarray(o1["loss"]) - 1.0)) < 2e-5
    dr = np.abs(np.asarray(o2["rec"]) - np.asarray(o1["rec"]))
    assert (dr <= 1e-4).mean() >= 0.97 and dr.max() < 1e-2, (float((dr > 1e-4).mean()), float(dr.max()))
    dz = np.abs(np.asarray(o2["z"]) - np.asarray(o1["z"])).max(axis=1) / np.abs(np.asarray(o1["z"])).max()
    assert np.median(dz) < 1e-5 and (dz < 1e-3).mean() >= 0.95, np.sort(dz)[-3:]
    if B * R <= 8:
        O = _oracle()
        ref = O.reconstruct(p, x, z0, R, 3, lr=1.0, momentum=0.7, arch="mnist", use_bn=True, dtype=np.float64)
        np.testing.assert_allclose(o2["loss"], ref["loss"], rtol=2e-3)
        np.testing.assert_allclose(o2["rec"], ref["rec"], rtol=0, atol=2e-3)
