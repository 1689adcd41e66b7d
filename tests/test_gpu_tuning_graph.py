"""-m gpu: the hand-over of timed job-list choices between handles / processes (dg_export_tuning, dg_import_tuning), the
replayed graph of the projection loop for small call shapes (option graph_max_rows), and the sampling stride of the per-kernel
event profile."""
import os
import subprocess
import sys

import numpy as np
import pytest

from defensegan_amd import _native
from defensegan_amd.gan import tuning_text_id
from tests.helpers import make_gan

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np(d):
    return {k: (v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)) for k, v in d.items()}


def test_exported_tuning_reproduces_the_lists_in_another_handle():
    B, R = 140, 10
    g1, p = make_gan("mnist", rec_rr=R, rec_iters=3)
    g1.set_option("latent_turn", 0)                   # all eight GEMM layers carry job lists
    g1.prepare(B)
    text = g1.export_tuning()
    lines = text.strip().splitlines()
    assert lines[0].startswith("dgtune 1 arch 0 latent 128 net_dim 64 use_bn 0 nsplit 16 cus ")
    assert sorted(l.split()[0] for l in lines[1:]) == ["B1", "B2", "B3", "F1", "F2", "F3"]
    assert all(int(l.split()[1]) == B * R for l in lines[1:])
    g2, _ = make_gan("mnist", rec_rr=R, rec_iters=3)
    g2.set_option("latent_turn", 0)
    assert g2.import_tuning(text) == 6
    g2.prepare(B)                                     # finds every list: allocates the workspace, times nothing
    assert g2.tuning_id() == g1.tuning_id() == tuning_text_id(text)
    # ... not even the measured durations changed: nothing was timed again
    assert sorted(g2.export_tuning().strip().splitlines()) == sorted(lines)
    # an untuned third handle is free to choose otherwise, yet computes the same bits
    rs = np.random.RandomState(1)
    x = rs.uniform(0, 1, size=(B, 28, 28, 1)).astype(np.float32)
    z0 = (rs.standard_normal((B * R, 128)) * 0.09).astype(np.float32)
    a = _np(g1.reconstruct(x, z_init_val=z0, return_details=True))
    b = _np(g2.reconstruct(x, z_init_val=z0, return_details=True))
    for k in ("rec", "idx", "loss", "z"):
        assert np.array_equal(a[k], b[k]), k
    # the default configuration runs the Linear layers on the weight-stationary kernels: no list for them
    g3, _ = make_gan("mnist", rec_rr=R, rec_iters=3)
    g3.prepare(B)
    assert sorted(l.split()[0] for l in g3.export_tuning().strip().splitlines()[1:]) == ["B2", "B3", "F2", "F3"]
    # another configuration refuses the text; garbage is reported
    gc, _ = make_gan("celeba", rec_rr=2, rec_iters=2)
    with pytest.raises(_native.NativeError, match="another configuration"):
        gc.import_tuning(text)
    with pytest.raises(_native.NativeError, match="malformed"):
        g3.import_tuning(lines[0] + "\nF2 12 nonsense\n")
    with pytest.raises(_native.NativeError, match="unknown layer"):
        g3.import_tuning(lines[0] + "\nF9 2560 1 1 0 0 0 10 0\n")


def test_tuning_cache_file_hands_the_choice_to_the_next_process(tmp_path):
    """DG_TUNING_CACHE: the second process installs the first one's lists (same tuning id) instead of timing its own."""
    cache = str(tmp_path / "tuning.txt")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.helpers import make_gan\n"
            "g, _ = make_gan('mnist', rec_rr=10, rec_iters=2)\n"
            "g.prepare(64)\n"
            "print('TID', g.tuning_id())\n" % ROOT)
    ids = []
    for _ in range(2):
        env = dict(os.environ, DG_TUNING_CACHE=cache)
        r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        ids.append([l for l in r.stdout.splitlines() if l.startswith("TID ")][-1].split()[1])
    assert os.path.exists(cache) and open(cache).read().startswith("dgtune 1 ")
    assert ids[0] == ids[1] == tuning_text_id(open(cache).read())


def test_the_row_group_choice_of_a_call_shape_travels_with_the_tuning_text():
    """CelebA's default two_streams = "auto": whether a shape runs as one row group or two is timed when the shape is prepared and
    written into the tuning text ("groups B R n"); a handle that imports the text runs the same form without timing -- the bench
    line, the rocprofv3 trace and the PMC passes of tools/collect_profiles.sh (where counters serialise the kernels, so that two
    groups could never win a timing) then launch the same kernels.  A record for the other form is honoured as well."""
    g1, p = make_gan("celeba", rec_rr=10, rec_iters=2)
    g1.prepare(104)
    n1 = g1.row_groups(104)
    text = g1.export_tuning()
    assert ("groups 104 10 %d" % n1) in text.splitlines()
    other = 2 if n1 == 1 else 1
    forced = "\n".join(("groups 104 10 %d" % other) if l.startswith("groups 104 10 ") else l for l in text.splitlines()) + "\n"
    assert tuning_text_id(forced) != tuning_text_id(text)
    g2, _ = make_gan("celeba", rec_rr=10, rec_iters=2)
    assert g2.import_tuning(forced) > 0
    assert g2.row_groups(104) == other
    g2.prepare(104)                                            # nothing is timed again: the installed choice stands
    assert g2.row_groups(104) == other and g2.tuning_id() == tuning_text_id(forced)


@pytest.mark.parametrize("arch,B,R,L,turn", [("mnist", 24, 5, 6, 0), ("mnist", 50, 10, 4, 0), ("celeba", 6, 10, 3, 0), ("mnist", 50, 10, 7, 1)])
def test_replayed_loop_graph_reproduces_the_enqueued_launches(arch, B, R, L, turn):
    """(Opt-in.)  Call shapes of at most graph_max_rows latent rows replay a captured graph of the L-step loop (images staged into the
    engine's own buffer): same kernels, same arguments -> bit-identical to enqueuing them, for new images through the same
    graph, for seeded latents, and after an option change rebuilt the graph."""
    g1, p = make_gan(arch, rec_rr=R, rec_iters=L)
    g1.set_option("graph_max_rows", 1024)                      # opt-in (off by default: include/defensegan_hip.h)
    if turn:                                                   # the captured loop holds the fused latent turn (dg_turn.hip): its barrier
        g1.set_option("turn_fused", 1)                         # counters are cleared in front of every replay, outside the graph
    g0, _ = make_gan(arch, rec_rr=R, rec_iters=L)
    for seed in (3, 5):
        x = g1.generate(g1.init_latents(B, seed=seed)).contiguous()
        z0 = g1.init_latents(B * R, seed=seed + 1)
        a, b = _np(g1.reconstruct(x, z_init_val=z0, return_details=True)), _np(g0.reconstruct(x, z_init_val=z0, return_details=True))
        for k in ("rec", "idx", "loss", "z"):
            assert np.array_equal(a[k], b[k]), (seed, k)
        a, b = _np(g1.reconstruct(x, seed=11, first_row=70, return_details=True)), _np(g0.reconstruct(x, seed=11, first_row=70, return_details=True))
        for k in ("rec", "idx", "loss", "z"):
            assert np.array_equal(a[k], b[k]), (seed, k, "seeded")
    # other hyper-parameters -> another graph; an option change invalidates the captured ones
    g1.rec_lr, g0.rec_lr = 3.0, 3.0
    g1.rec_lr_schedule = g0.rec_lr_schedule = "intended"
    g1.set_option("latent_turn", 0)
    a, b = _np(g1.reconstruct(x, z_init_val=z0, return_details=True)), _np(g0.reconstruct(x, z_init_val=z0, return_details=True))
    for k in ("rec", "idx", "loss", "z"):
        assert np.array_equal(a[k], b[k]), k
    # a ragged smaller batch on the same handle (own graph), then the first shape again
    xs, zs = x[: B - 1].contiguous(), z0[: (B - 1) * R].contiguous()
    a, b = _np(g1.reconstruct(xs, z_init_val=zs, return_details=True)), _np(g0.reconstruct(xs, z_init_val=zs, return_details=True))
    assert np.array_equal(a["rec"], b["rec"]) and np.array_equal(a["z"], b["z"])
    a, b = _np(g1.reconstruct(x, z_init_val=z0, return_details=True)), _np(g0.reconstruct(x, z_init_val=z0, return_details=True))
    assert np.array_equal(a["rec"], b["rec"]) and np.array_equal(a["z"], b["z"])


def test_profile_stride_charges_a_launch_with_its_own_time_only():
    """dg_profile_enable(stride > 1): steps that are not sampled break the marker chain, so the first sampled launch of a
    step (F1) is not charged with the skipped steps before it."""
    B, R, L = 100, 10, 9
    gan, p = make_gan("mnist", rec_rr=R, rec_iters=L)
    x = gan.generate(gan.init_latents(B, seed=3)).contiguous()
    gan.prepare(B)
    gan.reconstruct(x, seed=1)

    def f1_avg(stride):
        gan.profile_reset()
        gan.profile_enable(stride)
        gan.reconstruct(x, seed=1)
        gan.profile_enable(0)
        prof = {q["name"].split("@")[0]: q for q in gan.profile_read()}
        n = prof["F1"]["launches"]
        return prof["F1"]["ms"] / n, n, sum(q["ms"] for q in prof.values())

    a1, n1, tot1 = f1_avg(1)
    a2, n2, tot2 = f1_avg(2)
    a4, n4, tot4 = f1_avg(4)
    assert (n1, n2, n4) == (9, 5, 3)
    assert a2 < 2.0 * a1 and a4 < 2.0 * a1, (a1, a2, a4)         # was ~ (stride - 1) whole steps too long
    assert tot2 < 0.75 * tot1 and tot4 < 0.5 * tot1, (tot1, tot2, tot4)
