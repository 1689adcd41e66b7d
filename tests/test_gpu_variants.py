"""-m gpu: the template instantiations and engine modes the default-configuration parity tests do not reach.

* NET_DIM = 128 / LATENT_DIM = 64 (the C = 128 instantiations of every tail kernel, other K extents) against the
  float64 oracle;
* launch-shape variants of the same arithmetic (GEMM job lists: whole tiles only, everything cut to halves / quarters,
  model-chosen instead of timed; concurrent row groups; non-persistent tails): each output element is the same fixed-order
  sum whatever the launch shape, so these must reproduce the default engine BIT FOR BIT;
* formulation variants whose summation order differs (CelebA 32-wide forward tail, split-K count): tolerance.
"""
import numpy as np
import pytest

from defensegan_amd import archs, synth

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import defensegan_oracle as O
    return O


# options whose kernels exist only in the measurement build of the library (-DDG_MEASURE, lib/libdefensegan_hip_measure.so):
# the superseded tail formulations, kept as cross-checks of the product kernels
MEASURE_ONLY = ("tail_fwd16", "tail_bwd_bands", "tail_dbg", "tail_prio", "tail_trace", "job_trace")


def _needs_measure(opts):
    return any(k in MEASURE_ONLY for k in opts) or str(opts.get("tail_bwd_persist", 1)) == "0"


def _make(arch, net_dim=64, latent_dim=128, gain=2.0, R=2, L=3, seed=1234, measure=False):
    from defensegan_amd.gan import dataset_gan_dict
    gan = dataset_gan_dict[arch](cfg={"USE_BN": False, "LATENT_DIM": latent_dim, "NET_DIM": net_dim}, test_mode=True,
                                 rec_rr=R, rec_iters=L, rec_lr=10.0, measure=measure)
    p = synth.make_weights(arch, seed=seed, gain=gain, bias_range=0.1, latent_dim=latent_dim, net_dim=net_dim)
    assert gan.set_weights(p) == []
    return gan, p


@pytest.mark.parametrize("arch,net_dim,latent_dim,B,R", [
    ("mnist", 128, 128, 3, 2), ("mnist", 128, 128, 35, 2), ("mnist", 64, 64, 4, 3),
    ("celeba", 128, 128, 2, 2), ("celeba", 128, 128, 9, 8), ("celeba", 64, 192, 3, 2)])
def test_other_widths_vs_oracle(arch, net_dim, latent_dim, B, R):
    O = _oracle()
    gan, p = _make(arch, net_dim, latent_dim, gain=2.0, R=R)
    a = archs.make_arch(arch, latent_dim, net_dim)
    P = int(np.prod(a.image_dim))
    rs = np.random.RandomState(B * 31 + R)
    zt = (rs.standard_normal((B, latent_dim)) * np.sqrt(1.0 / latent_dim)).astype(np.float32)
    x = O.generator_forward(p, zt, arch)[0].astype(np.float32)
    x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=6)
    z = (rs.standard_normal((B * R, latent_dim)) * 0.15).astype(np.float32)
    y, loss, dz = gan.loss_grad(x, z)
    yo, cache = O.generator_forward(p, z.astype(np.float64), arch)
    xt = np.repeat(x.astype(np.float64), R, axis=0)
    lo = ((yo - xt) ** 2).reshape(B * R, -1).mean(axis=1)
    go = O.generator_backward(p, cache, 2.0 / P * (yo - xt), arch)
    # fp32 vs float64: the output pre-activation is a sum of up to 1600 products; the gate scales with its magnitude
    atol = 1e-6 * max(5.0, float(np.abs(cache["pre"][-1]).max()))
    np.testing.assert_allclose(y, yo, rtol=0, atol=atol)
    np.testing.assert_allclose(loss, lo, rtol=1e-5)
    kink = np.zeros(B * R, bool)
    for a_pre, (_, _, _, act, _) in zip(cache["pre"][1:], O._arch_layers(arch)):
        if act == "relu":
            kink |= (np.abs(a_pre).reshape(B * R, -1).min(axis=1) < 1e-6)
    kink |= (np.abs(cache["pre"][0]).reshape(B * R, -1).min(axis=1) < 1e-6)
    scale = np.abs(go).max()
    err = np.abs(dz - go).max(axis=1) / scale
    assert kink.sum() <= max(2, 0.3 * B * R)
    assert (err[~kink] < 1e-5).all(), err[~kink].max()
    assert (err < 0.2).all(), err.max()


def _run(gan, x, z0):
    d = gan.reconstruct(x, z_init_val=z0, return_details=True)
    return {k: (v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)) for k, v in d.items()}


BITWISE = {
    "mnist": [{"tail_pipe": 0}, {"tail_pipe": 100}, {"tail_pipe_version": 1}, {"tail_pipe_version": 1, "tail_pipe": 100},
              {"tail_pipe_version": 3}, {"tail_pipe_version": 3, "tail_pipe": 100}, {"tail_pipe_version": 3, "tail_pipe": 37},
              {"tail_pipe": 37}, {"two_streams": 2, "two_stream_min_rows": 64},
              {"jobs.slack": 1e30, "jobs.min_level": 0}, {"jobs.slack": 0.01, "jobs.min_level": 0}, {"jobs.min_level": 1},
              {"jobs.slack": 1e30, "jobs.min_level": 2}, {"jobs.tune": 0}, {"jobs.tune": 0, "jobs.slots0": 5, "jobs.rate2": 300},
              # wave priorities by predicted job length: on every list / never (the default offers them to the timing)
              {"jobs.prio": 2}, {"jobs.prio": 2, "jobs.min_level": 1, "jobs.tune": 0}, {"jobs.prio": 0},
              # without the single-round candidates (balance_order, jobs_balanced) of the timing
              {"jobs.balance": 0},
              # every list without pairs on the PAIR instantiation of the kernel / never (the default times both forms)
              {"jobs.pair_kernel": 2}, {"jobs.pair_kernel": 0},
              # the latent turn: position-batched kernel instead of the weight-stationary ones; other workgroup counts
              {"latent_turn": 0}, {"lin_groups_fwd": 3, "lin_groups_bwd": 5}, {"lin_groups_fwd": 64, "lin_groups_bwd": 64},
              # the momentum update folded into the Linear backward launch (last-arriving K slice), alone / with other group counts /
              # with two row groups on two streams
              {"update_fold": 1}, {"update_fold": 1, "lin_groups_bwd": 5}, {"update_fold": 1, "lin_groups_bwd": 64},
              {"update_fold": 1, "two_streams": 2, "two_stream_min_rows": 64},
              # the whole latent turn as one launch (dg_turn.hip): alone / other group counts / two and three row groups on streams
              {"turn_fused": 1}, {"turn_fused": 1, "lin_groups_bwd": 5}, {"turn_fused": 1, "lin_groups_bwd": 64},
              {"turn_fused": 1, "two_streams": 2, "two_stream_min_rows": 64}, {"turn_fused": 1, "two_streams": 3, "two_stream_min_rows": 64}],
    "celeba": [{"jobs.slack": 1e30, "jobs.min_level": 0}, {"jobs.slack": 0.01}, {"jobs.min_level": 1, "jobs.tune": 0},
               {"jobs.prio": 2}, {"jobs.prio": 0}, {"jobs.pair_kernel": 2},
               {"tail_bwd_persist": 0}, {"tail_bwd_persist": 0, "tail_bwd_bands": 2}, {"tail_bwd_persist": 300},
               {"latent_turn": 0}, {"lin_groups_fwd": 7, "lin_groups_bwd": 1}, {"update_fold": 1}, {"update_fold": 1, "lin_groups_bwd": 3},
               # row groups on streams (CelebA's default for calls of >= 1024 rows), 2 and 3 of them, unequal halves
               {"two_streams": 2, "two_stream_min_rows": 64}, {"two_streams": 3, "two_stream_min_rows": 64},
               {"two_streams": 2, "two_stream_min_rows": 64, "two_stream_split": 60},
               {"turn_fused": 1}, {"turn_fused": 1, "lin_groups_bwd": 3}, {"turn_fused": 1, "two_streams": 2, "two_stream_min_rows": 64}],
}


@pytest.mark.parametrize("arch,B,R", [("mnist", 140, 10), ("celeba", 70, 10)])
def test_launch_shape_variants_are_bit_identical(arch, B, R):
    """1400 / 700 rows: more jobs than resident slots in every list, and both tail item loops iterate."""
    a = archs.make_arch(arch)
    gan, p = _make(arch, R=R, L=3)
    rs = np.random.RandomState(7)
    x = gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32))
    x = np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x, np.float32)
    x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=8)
    z0 = synth.make_z(B * R, 128, seed=9)
    ref = _run(gan, x, z0)
    assert np.isfinite(ref["loss"]).all()
    for opts in BITWISE[arch]:
        g2, _ = _make(arch, R=R, L=3, measure=_needs_measure(opts))
        for k, v in opts.items():
            g2.set_option(k, v)
        got = _run(g2, x, z0)
        for k in ("rec", "idx", "loss", "z"):
            assert np.array_equal(got[k], ref[k]), (opts, k, np.abs(got[k].astype(np.float64) - ref[k]).max())


def test_celeba_calls_of_1024_rows_or_more_run_as_two_row_groups_with_the_same_bits():
    """Round 6: CelebA calls of >= 1024 latent rows run as two row groups on two streams where timing the call shape finds that
    faster (option two_streams = "auto", dg_call_row_groups; +2-3 % at configs[3]'s 1280 rows, profiles/r06_ab_celeba_row_groups.txt);
    MNIST and use_bn keep one.  The split is by whole images and
    rows are independent: the same bits as one group -- also while the per-launch profile is on, when the groups run one after
    the other on the caller's stream (bench.py's marked step), with every layer then sampled once per group."""
    a = archs.make_arch("celeba")
    B, R = 104, 10
    gan, p = _make("celeba", R=R, L=2)
    # default "auto": one or two groups is timed when the shape is prepared (one group until then, always one below 1024 rows)
    assert gan.row_groups(B) == 1 and gan.row_groups(102) == 1
    gan.prepare(B)
    assert gan.row_groups(B) in (1, 2) and gan.row_groups(102) == 1
    gan.set_option("two_streams", 2)                                  # from here on: two groups, not a timed choice
    assert gan.row_groups(B) == 2 and gan.row_groups(128) == 2 and gan.row_groups(102) == 1
    gm, _ = _make("mnist", R=R, L=2)
    gm.prepare(256)
    assert gm.row_groups(256) == 1
    rs = np.random.RandomState(31)
    x = rs.uniform(a.in_lo, a.in_hi, size=(B,) + tuple(a.image_dim)).astype(np.float32)
    z0 = synth.make_z(B * R, 128, seed=32)
    ref = _run(gan, x, z0)
    g1, _ = _make("celeba", R=R, L=2)
    g1.set_option("two_streams", 0)
    assert g1.row_groups(B) == 1
    one = _run(g1, x, z0)
    gan.profile_reset()
    gan.profile_enable(1)
    marked = _run(gan, x, z0)
    gan.profile_enable(0)
    prof = {q["name"].partition("@")[0]: q["launches"] for q in gan.profile_read()}
    assert prof["F5"] == 2 * 2 and prof["B5"] == 2 * 1, prof          # L = 2: two forwards, one backward, per group
    for k in ("rec", "idx", "loss", "z"):
        assert np.array_equal(one[k], ref[k]) and np.array_equal(marked[k], ref[k]), k


@pytest.mark.parametrize("arch,B,R", [("mnist", 50, 10), ("celeba", 30, 10)])
def test_poisoned_pair_counters_cannot_reach_the_next_call(arch, B, R):
    """K-pair hand-off (dg_gemm.hip): a launch that dies between the two arrivals of a pair leaves its counter at 1; the first
    arriver of the next call would then add a STALE accumulator image and run the epilogue.  Every call clears the counters of the
    lists it launches (clear_pair_counters, dg_engine.cpp): with every counter of every list poisoned (debug hook) the next
    projection -- and the next loop body -- still reproduce the clean run bit for bit.  The hand-off itself is stressed too: the
    same call 12 times, one after the other, bit-identical every time (a stale or torn image read across XCDs would show)."""
    a = archs.make_arch(arch)
    gan, p = _make(arch, R=R, L=4)
    rs = np.random.RandomState(23)
    x = np.asarray(gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32)))
    x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=24)
    z0 = synth.make_z(B * R, 128, seed=25)
    ref = _run(gan, x, z0)
    lg_ref = [np.asarray(v) for v in gan.loss_grad(x, z0)]
    assert np.isfinite(ref["loss"]).all()
    for value in (1, 1, 7):
        gan.set_option("debug.poison_pair_counters", value)
        got = _run(gan, x, z0)
        for k in ("rec", "idx", "loss", "z"):
            assert np.array_equal(got[k], ref[k]), (value, k, np.abs(got[k].astype(np.float64) - ref[k]).max())
        gan.set_option("debug.poison_pair_counters", value)
        for u, v in zip([np.asarray(v) for v in gan.loss_grad(x, z0)], lg_ref):
            assert np.array_equal(u, v), value
    for rep in range(12):
        got = _run(gan, x, z0)
        for k in ("rec", "idx", "loss", "z"):
            assert np.array_equal(got[k], ref[k]), (rep, k)


@pytest.mark.parametrize("opts", [{}, {"update_fold": 1}, {"turn_fused": 1}])
def test_handoffs_under_uneven_load_are_bit_identical(opts):
    """Every in-launch hand-off of the loop (K-pair accumulator images of dg_gemm.hip; with the options, the folded update's and the
    fused latent turn's partials and latents) read back by another workgroup while a SECOND stream streams 256-MB copies through the
    same L2s and fabric: workgroups then start and finish unevenly and the readers' caches are warm with other lines -- the
    conditions under which a missing drain or a cached stale line shows.  Six loaded calls of 2560 rows x 12 steps each reproduce
    the quiet run bit for bit."""
    import torch
    a = archs.make_arch("mnist")
    B, R, L = 256, 10, 12
    gan, p = _make("mnist", R=R, L=L)
    for k, v in opts.items():
        gan.set_option(k, v)
    rs = np.random.RandomState(31)
    x = np.asarray(gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32)))
    x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=32)
    z0 = synth.make_z(B * R, 128, seed=33)
    quiet, _ = _make("mnist", R=R, L=L)
    ref = _run(quiet, x, z0)
    assert np.isfinite(ref["loss"]).all()
    assert all(np.array_equal(_run(gan, x, z0)[k], ref[k]) for k in ("rec", "idx", "loss", "z"))      # (quiet, with the options)
    side = torch.cuda.Stream()
    src = torch.empty(64 << 20, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(120 + 40 * rep):                      # ~0.1 ms each: the copies outlast the 15-ms call
                dst.copy_(src, non_blocking=True)
        got = _run(gan, x, z0)
        assert not side.query() or rep == 0, "the background copies ended before the call: no load"
        side.synchronize()
        for k in ("rec", "idx", "loss", "z"):
            assert np.array_equal(got[k], ref[k]), (opts, rep, k, np.abs(got[k].astype(np.float64) - ref[k]).max())


@pytest.mark.parametrize("B,R,L", [(256, 10, 60), (121, 10, 40), (3, 1, 25)])
def test_folded_update_is_bit_identical_over_many_steps(B, R, L):
    """Option update_fold: the workgroup that delivers the last K slice of a 32-row block of dz applies the momentum update
    inside the Linear backward launch (write-through partials, one arrival counter per block).  Every step reads the partials
    other workgroups -- on other XCDs -- wrote into the SAME buffer a step earlier, so one stale word would change z for good:
    after L steps z, the losses and the selection are bit-identical to the separate momentum_update_kernel's, twice in a row
    (the counters are back at zero), at full (2560), ragged (1210) and tiny (3) row counts."""
    a = archs.make_arch("mnist")
    gan, p = _make("mnist", R=R, L=L)
    rs = np.random.RandomState(17)
    x = gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32))
    x = np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x, np.float32)
    x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=18)
    z0 = synth.make_z(B * R, 128, seed=19)
    ref = _run(gan, x, z0)
    g2, _ = _make("mnist", R=R, L=L)
    g2.set_option("update_fold", 1)
    for rep in range(2):
        got = _run(g2, x, z0)
        for k in ("rec", "idx", "loss", "z"):
            assert np.array_equal(got[k], ref[k]), (rep, k, np.abs(got[k].astype(np.float64) - ref[k]).max())


@pytest.mark.parametrize("arch,B,R,L", [("mnist", 256, 10, 60), ("mnist", 121, 10, 40), ("mnist", 3, 1, 25), ("mnist", 50, 10, 30),
                                        ("celeba", 64, 10, 20), ("celeba", 7, 3, 12)])
def test_fused_latent_turn_is_bit_identical_over_many_steps(arch, B, R, L):
    """Option turn_fused (dg_turn.hip): Linear backward, momentum update and the next step's Linear forward as ONE launch, the
    workgroups of a row group meeting at two barriers of their own.  Every step reads partials and latents other workgroups --
    on other XCDs -- wrote into the SAME buffers a step earlier (and, for z, a phase earlier), so one stale word would change z
    for good: after L steps z, the losses and the selection are bit-identical to the three separate launches', twice in a row
    (the barrier counters restart at zero with every call), at full, ragged and tiny row counts of both architectures."""
    a = archs.make_arch(arch)
    gan, p = _make(arch, R=R, L=L)
    rs = np.random.RandomState(23)
    x = gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32))
    x = np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x, np.float32)
    x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=24)
    z0 = synth.make_z(B * R, 128, seed=25)
    ref = _run(gan, x, z0)
    assert np.isfinite(ref["loss"]).all()
    g2, _ = _make(arch, R=R, L=L)
    g2.set_option("turn_fused", 1)
    for rep in range(2):
        got = _run(g2, x, z0)
        for k in ("rec", "idx", "loss", "z"):
            assert np.array_equal(got[k], ref[k]), (rep, k, np.abs(got[k].astype(np.float64) - ref[k]).max())


@pytest.mark.parametrize("arch,latent_dim,net_dim", [("mnist", 128, 64), ("mnist", 64, 64), ("celeba", 192, 64), ("mnist", 128, 128)])
def test_latent_turn_kernels_reproduce_the_generic_gemm(arch, latent_dim, net_dim):
    """dg_linear.hip (weights stationary in registers, rows streamed in 32-row blocks) against the position-batched kernel
    (option latent_turn = 0) on ragged row counts: the same fma chains and epilogue expressions, so G(z), the loss and dL/dz
    are BIT-identical.  latent_dim 64 / 192 exercise the 2- and 6-chunk forward instantiations (their backward and the
    NET_DIM 128 backward stay on the position-batched kernel)."""
    a = archs.make_arch(arch, latent_dim, net_dim)
    g1, p = _make(arch, net_dim, latent_dim, R=1, L=1)
    g0, _ = _make(arch, net_dim, latent_dim, R=1, L=1)
    g0.set_option("latent_turn", 0)
    rs = np.random.RandomState(3)
    for n in (1, 5, 31, 32, 33, 97, 130, 517):
        z = (rs.standard_normal((n, latent_dim)) * 0.15).astype(np.float32)
        x = rs.uniform(a.in_lo, a.in_hi, size=(n,) + tuple(a.image_dim)).astype(np.float32)
        got = [np.asarray(v) for v in g1.loss_grad(x, z)]
        ref = [np.asarray(v) for v in g0.loss_grad(x, z)]
        assert np.isfinite(ref[2]).all() and np.abs(ref[2]).max() > 0
        for name, u, v in zip(("y", "loss", "dz"), got, ref):
            assert np.array_equal(u, v), (n, name, np.abs(u.astype(np.float64) - v).max())


@pytest.mark.parametrize("B,wgs", [(70, 512), (3, 512), (40, 100)])
def test_celeba_role_split_forward_tail_reproduces_the_band_kernel(B, wgs):
    """celeba_tail_fwd_split_kernel (persistent 8-wave workgroups, GEMM waves and gather waves on half-bands; the default for
    64 channels, option tail_fwd_split = workgroups) against celeba_tail_fwd16_kernel (tail_fwd_split = 0): every P entry is the same k-ordered MFMA chain and the
    taps are added in the same order, so y and da6 -- hence the reconstructions and the latents after L steps -- are BIT-identical;
    the per-row loss is a sum of the same squares in another grouping (64 partials per row instead of 32): float32 rounding."""
    a = archs.make_arch("celeba")
    R = 10
    gan, p = _make("celeba", R=R, L=3)
    rs = np.random.RandomState(17)
    x = np.asarray(gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32)))
    x = synth.adversarial(x, 0.3, a.in_lo, a.in_hi, seed=18)
    z0 = synth.make_z(B * R, 128, seed=19)
    gan.set_option("tail_fwd_split", 0)                 # the per-band kernel is the reference here
    ref = _run(gan, x, z0)
    g2, _ = _make("celeba", R=R, L=3)
    g2.set_option("tail_fwd_split", wgs)
    got = _run(g2, x, z0)
    assert np.array_equal(got["z"], ref["z"]) and np.array_equal(got["rec"], ref["rec"])
    np.testing.assert_allclose(got["loss"], ref["loss"], rtol=2e-6)
    assert np.array_equal(got["idx"], ref["idx"]) or np.allclose(np.sort(ref["loss"].reshape(B, R), axis=1)[:, 0],
                                                                 np.sort(ref["loss"].reshape(B, R), axis=1)[:, 1], rtol=1e-5)
    # one loop body: y, loss, dz
    z = (rs.standard_normal((B * R, 128)) * 0.15).astype(np.float32)
    y0, l0, d0 = gan.loss_grad(x, z)
    y1, l1, d1 = g2.loss_grad(x, z)
    assert np.array_equal(y1, y0) and np.array_equal(d1, d0)
    np.testing.assert_allclose(l1, l0, rtol=2e-6)


def test_measurement_options_are_refused_by_the_product_library():
    """The product library carries no measurement kernels: their options fail loudly instead of being ignored."""
    from defensegan_amd import _native
    gan, _ = _make("celeba", R=2, L=1)
    for k, v in (("tail_fwd16", 0), ("tail_bwd_persist", 0), ("tail_dbg", 1), ("tail_trace", 1), ("job_trace", "F2")):
        with pytest.raises(_native.NativeError, match="measurement build"):
            gan.set_option(k, v)
    gan.set_option("tail_bwd_persist", 300)          # the product kernel's own knob stays
    gm, _ = _make("celeba", R=2, L=1, measure=True)
    gm.set_option("tail_fwd16", 0)


@pytest.mark.parametrize("arch,opts", [("celeba", {"tail_fwd16": 0}), ("mnist", {"nsplit": 4}), ("mnist", {"nsplit": 8})])
def test_reordered_formulations_agree_to_rounding(arch, opts):
    a = archs.make_arch(arch)
    B, R = 6, 3
    gan, p = _make(arch, R=R, L=1)
    rs = np.random.RandomState(11)
    x = gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32))
    x = np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x, np.float32)
    z = (rs.standard_normal((B * R, 128)) * 0.15).astype(np.float32)
    y0, l0, g0 = gan.loss_grad(x, z)
    g2, _ = _make(arch, R=R, L=1, measure=_needs_measure(opts))
    for k, v in opts.items():
        g2.set_option(k, v)
    y1, l1, g1 = g2.loss_grad(x, z)
    np.testing.assert_allclose(y1, y0, rtol=0, atol=2e-6)
    np.testing.assert_allclose(l1, l0, rtol=2e-6)
    np.testing.assert_allclose(g1, g0, rtol=0, atol=2e-5 * np.abs(g0).max())


def test_load_generator_from_tensorflow_checkpoint(tmp_path):
    """a reference-style checkpoint directory (scoped names, optimizer slots, other networks) loads through
    ``load_generator`` and gives the generator the same weights as ``set_weights``."""
    from defensegan_amd import tf_checkpoint
    from defensegan_amd.gan import dataset_gan_dict
    gan, p = _make("mnist", R=2, L=2)
    t = {}
    for name, a in p.items():
        scope = name.rsplit(".", 1)[0]
        t["%s/%s" % (scope, name)] = a
        t["%s/%s/Adam" % (scope, name)] = np.zeros_like(a)
    t["Discriminator.1/Discriminator.1.Filters"] = np.ones((5, 5, 1, 64), np.float32)
    t["global_step"] = np.array(7, np.int64)
    tf_checkpoint.write_checkpoint(str(tmp_path / "GAN.model-7"), t)
    g2 = dataset_gan_dict["mnist"](cfg={"USE_BN": False}, test_mode=True, rec_rr=2, rec_iters=2)
    assert not g2.initialized
    assert g2.load_generator(str(tmp_path)) is True
    z = synth.make_z(5, 128, seed=2)
    assert np.array_equal(np.asarray(g2.generate(z)), np.asarray(gan.generate(z)))
    bad = dict(t)
    del bad["Generator.3/Generator.3.Filters"]
    tf_checkpoint.write_checkpoint(str(tmp_path / "bad" / "m"), bad)
    g3 = dataset_gan_dict["mnist"](cfg={"USE_BN": False}, test_mode=True)
    with pytest.raises(ValueError, match="Generator.3.Filters"):
        g3.load_generator(str(tmp_path / "bad"))


def test_command_line_projects_a_file(tmp_path):
    """python -m defensegan_amd: weight pack + image stack in, reconstructions out, ragged batches; equals the API call."""
    from defensegan_amd import __main__ as cli
    gan, p = _make("mnist", R=3, L=4)
    np.savez(str(tmp_path / "generator.npz"), **p)
    x = np.asarray(gan.generate(synth.make_z(11, 128, seed=3)))
    np.save(str(tmp_path / "x.npy"), x)
    rc = cli.main(["--cfg", "mnist", "--init_path", str(tmp_path / "generator.npz"), "--input", str(tmp_path / "x.npy"),
                   "--output", str(tmp_path / "rec.npy"), "--rec_rr", "3", "--rec_iters", "4", "--batch_size", "4", "--seed", "5"])
    assert rc == 0
    rec = np.load(str(tmp_path / "rec.npy"))
    want = np.asarray(gan.reconstruct(x, seed=5, first_row=0))       # batch-composition independent, z0 keyed by global row
    assert rec.shape == x.shape and np.array_equal(rec, want)


@pytest.mark.parametrize("arch", ["mnist", "celeba"])
def test_divergent_runs_return_and_never_select_a_nan(arch):
    """A step size far beyond stability: latents overflow, losses become inf/NaN.  The call must still return, indices
    must be valid, and a restart with a NaN loss must never be selected while a finite one exists (strict "<" keeps the
    first minimum; all-NaN images select restart 0)."""
    a = archs.make_arch(arch)
    B, R = 6, 4
    gan, p = _make(arch, gain=3.0, R=R, L=6)
    rs = np.random.RandomState(1)
    x = np.asarray(gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32)))
    z0 = synth.make_z(B * R, 128, seed=2)
    z0[1::R] *= 0.0                                              # one restart per image starts at z = 0 and ...
    gan.rec_lr = 1e9
    out = gan.reconstruct(x, z_init_val=z0, return_details=True)
    loss = out["loss"].reshape(B, R)
    assert out["idx"].min() >= 0 and out["idx"].max() < R
    for b in range(B):
        finite = np.isfinite(loss[b])
        if finite.any():
            assert finite[out["idx"][b]] and loss[b, out["idx"][b]] == loss[b][finite].min()
        else:
            assert out["idx"][b] == 0
    gan.rec_lr = 10.0
    ok = gan.reconstruct(x, z_init_val=z0, return_details=True)    # the handle is still usable afterwards
    assert np.isfinite(ok["loss"]).all()


def test_loss_grad_optional_outputs_through_the_c_abi():
    """include/defensegan_hip.h lets out_y / out_loss / out_dz of dg_loss_grad be NULL independently.  At >= 512 rows the MNIST
    tail would run the third-generation pipelined kernel, which writes neither the per-row loss nor y: a call that asks for y (or
    the loss) must fall back to the kernel that does -- every combination returns what the all-outputs call returns."""
    import ctypes as C
    import torch
    B, R = 60, 10                                   # 600 rows: above the pipelined tail's threshold (2 x 256)
    gan, p = _make("mnist", R=R, L=3)
    a = archs.make_arch("mnist")
    rs = np.random.RandomState(3)
    x = gan.generate((rs.standard_normal((B, 128)) * 0.09).astype(np.float32))
    x = torch.as_tensor(synth.adversarial(np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x, np.float32), 0.3, a.in_lo, a.in_hi, seed=4)).cuda()
    z = torch.as_tensor(synth.make_z(B * R, 128, seed=5)).cuda()
    y_ref, loss_ref, dz_ref = gan.loss_grad(x, z)
    lib, h = gan._lib(), gan._handle
    stream = torch.cuda.current_stream().cuda_stream
    for want_y, want_loss, want_dz in [(1, 0, 1), (1, 0, 0), (0, 1, 1), (0, 0, 1), (1, 1, 0), (0, 1, 0)]:
        y = torch.full_like(y_ref, float("nan"))
        loss = torch.full_like(loss_ref, float("nan"))
        dz = torch.full_like(dz_ref, float("nan"))
        gan._check(lib.dg_loss_grad(h, x.data_ptr(), z.data_ptr(), B, R, y.data_ptr() if want_y else None,
                                    loss.data_ptr() if want_loss else None, dz.data_ptr() if want_dz else None, stream))
        torch.cuda.synchronize()
        if want_y:
            assert torch.equal(y, y_ref), (want_y, want_loss, want_dz)
        if want_loss:
            assert torch.equal(loss, loss_ref), (want_y, want_loss, want_dz)
        if want_dz:
            assert torch.equal(dz, dz_ref), (want_y, want_loss, want_dz)
