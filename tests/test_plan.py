"""CPU checks of the host planner (defensegan_amd/csrc/dg_plan.cpp): the per-position tap tables are
executed with plain host loops (tests/support/plan_host_exec.cpp, test-only) and compared with the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import defensegan_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "defensegan_amd", "csrc")
SUP = os.path.join(ROOT, "tests", "support")
SO = os.path.join(SUP, "_build", "libdgplan_test.so")


@pytest.fixture(scope="module")
def lib():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(SUP, "plan_host_exec.cpp"), os.path.join(CSRC, "dg_plan.cpp")]
    deps = srcs + [os.path.join(CSRC, "dg_plan.h"), os.path.join(CSRC, "dg_types.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", SO] + srcs)
    l = C.CDLL(SO)
    l.dgp_build.restype = C.c_void_p
    l.dgp_build.argtypes = [C.c_char_p] + [C.c_int] * 7
    l.dgp_free.argtypes = [C.c_void_p]
    l.dgp_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    l.dgp_tap_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp_apply.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_int]
    l.dgp2_build.restype = C.c_void_p
    l.dgp2_build.argtypes = [C.c_void_p]
    l.dgp2_free.argtypes = [C.c_void_p]
    l.dgp2_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    l.dgp2_classes.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp2_class_grids.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp2_make_jobs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    l.dgp2_predicted_us.restype = C.c_double
    l.dgp2_predicted_us.argtypes = [C.c_void_p]
    l.dgp2_jobs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp2_apply.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
    l.dgp2_make_recorded.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                     C.c_double, C.c_char_p, C.c_int]
    l.dgp2_rebuild_matches.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    l.dgp2_job_pairs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp2_frag_apply.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.POINTER(C.c_double)]
    l.dgp2_make_balanced.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    l.dgp2_assign_priorities.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    l.dgp2_format_with_prio.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    return l


def build(lib, kind, *p):
    p = list(p) + [0] * (7 - len(p))
    h = lib.dgp_build(kind.encode(), *p)
    assert h
    info = (C.c_longlong * 8)()
    lib.dgp_info(h, info)
    return h, dict(zip(["n_pos", "n_taps", "a_rowstride", "out_rowstride", "w_rowstride", "kch", "ncols", "macs"], info))


def apply(lib, h, A, W, bias, out, n, mode):
    A = np.ascontiguousarray(A, np.float64); W = np.ascontiguousarray(W, np.float64)
    b = np.ascontiguousarray(bias, np.float64) if bias is not None else np.zeros(1)
    assert out.dtype == np.float64 and out.flags.c_contiguous
    lib.dgp_apply(h, A.ctypes.data, W.ctypes.data, b.ctypes.data, out.ctypes.data, n, mode)


def test_mac_counts_full_size(lib):
    # crop-aware MACs per row: G.2 7 372 800, G.3 8 388 608; CelebA 9 469 952 / 11 214 848 / 24 285 184
    for (h_in, e, cin, cout, want) in [(4, 7, 256, 128, 7_372_800), (7, 14, 128, 64, 8_388_608),
                                       (4, 8, 256, 128, 9_469_952), (8, 16, 128, 64, 11_214_848),
                                       (16, 32, 64, 64, 24_285_184)]:
        pitch_in = h_in if h_in != 7 else 7
        hf, f = build(lib, "deconv_fwd", h_in, pitch_in, e, e, cin, cout, 64)
        hb, b = build(lib, "deconv_bwd", h_in, pitch_in, e, e, cin, cout, 64)
        assert f["macs"] == want and b["macs"] == want
        assert f["n_pos"] == e * e * (cout // 64) and b["n_pos"] == h_in * h_in * (cin // 64)
        cnt = (C.c_int * f["n_pos"])()
        lib.dgp_tap_counts(hf, cnt)
        assert list(cnt) == sorted(cnt, reverse=True)          # longest tiles first
        lib.dgp_free(hf); lib.dgp_free(hb)


@pytest.mark.parametrize("h_in,e,pitch_out", [(4, 7, 7), (4, 8, 8), (7, 14, 14), (4, 7, 8)])
def test_deconv_fwd_plan_vs_oracle(lib, h_in, e, pitch_out):
    rs = np.random.RandomState(0)
    cin, cout, N, bn = 64, 128, 3, 64
    x = rs.randn(N, h_in, h_in, cin); F = rs.randn(5, 5, cout, cin); bias = rs.randn(cout)
    h, info = build(lib, "deconv_fwd", h_in, h_in, e, pitch_out, cin, cout, bn)
    out = np.full((N, pitch_out, pitch_out, cout), 777.0)
    apply(lib, h, x, F, bias, out, N, 2)
    want = np.maximum(O.deconv2d(x, F, bias, e), 0)
    np.testing.assert_allclose(out[:, :e, :e], want, rtol=1e-12, atol=1e-12)
    if pitch_out > e:
        assert (out[:, e:] == 777.0).all() and (out[:, :, e:] == 777.0).all()   # untouched
    lib.dgp_free(h)


@pytest.mark.parametrize("h_in,e", [(4, 7), (4, 8), (7, 14)])
def test_deconv_bwd_plan_vs_oracle(lib, h_in, e):
    rs = np.random.RandomState(1)
    cin, cout, N, bn = 128, 64, 2, 64
    dy = rs.randn(N, e, e, cout); F = rs.randn(5, 5, cout, cin)
    Ft = np.ascontiguousarray(F.transpose(0, 1, 3, 2))              # [kh,kw,cin,cout]
    hact = rs.randn(N, h_in, h_in, cin)                             # activation that gets masked in place
    h, info = build(lib, "deconv_bwd", h_in, h_in, e, e, cin, cout, bn)
    out = hact.copy()
    apply(lib, h, dy, Ft, None, out, N, 3)
    want = O.deconv2d_backward_input(dy, F, h_in) * (hact > 0)
    np.testing.assert_allclose(out, want, rtol=1e-12, atol=1e-12)
    lib.dgp_free(h)


def test_linear_plans_vs_oracle(lib):
    rs = np.random.RandomState(2)
    latent, feat, N = 64, 256, 5
    z = rs.randn(N, latent); W = rs.randn(latent, feat); b = rs.randn(feat)
    h, info = build(lib, "linear_fwd", latent, feat, 64)
    out = np.zeros((N, feat))
    apply(lib, h, z, np.ascontiguousarray(W.T), b, out, N, 1)
    np.testing.assert_allclose(out, z @ W + b, rtol=1e-12, atol=1e-12)
    lib.dgp_free(h)
    da = rs.randn(N, feat)
    nsplit = 4
    h, info = build(lib, "linear_bwd", latent, feat, nsplit, 64)
    part = np.zeros((N, nsplit, latent))
    apply(lib, h, da, W, None, part, N, 0)
    np.testing.assert_allclose(part.sum(axis=1), da @ W.T, rtol=1e-11, atol=1e-11)
    lib.dgp_free(h)


# ---------------------------------------------------------------------------------- position-batched form (dg_gemm.hip)
def batched(lib, kind, *p):
    h1, info = build(lib, kind, *p)
    h2 = lib.dgp2_build(h1)
    i2 = (C.c_longlong * 4)()
    lib.dgp2_info(h2, i2)
    cls = (C.c_int * (2 * i2[0]))()
    lib.dgp2_classes(h2, cls)
    return h1, h2, info, {"n_classes": i2[0], "n_pos": i2[1], "n_taps": i2[2], "family": i2[3],
                          "classes": [(cls[2 * i], cls[2 * i + 1]) for i in range(i2[0])]}


def jobs_of(lib, h2, n_rows, slots, alpha):
    n = lib.dgp2_make_jobs(h2, n_rows, slots, alpha)
    buf = (C.c_int * (6 * n))()
    lib.dgp2_jobs(h2, buf)
    return np.array(buf, np.int64).reshape(n, 6)


def apply_jobs(lib, h2, A, W, bias, out, mode):
    A = np.ascontiguousarray(A, np.float64); W = np.ascontiguousarray(W, np.float64)
    b = np.ascontiguousarray(bias, np.float64) if bias is not None else np.zeros(1)
    touched = np.zeros(out.shape, np.int32)
    lib.dgp2_apply(h2, A.ctypes.data, W.ctypes.data, b.ctypes.data, out.ctypes.data, touched.ctypes.data, mode)
    return touched


def test_classes_of_the_mnist_layers(lib):
    """Generator.2 forward (4x4 -> 7x7 used): per dimension the outputs fall in 4 relative-tap classes (1, 3, 1, 2 positions with
    1, 2, 2, 3 taps) -> 16 classes; Generator.3 forward (7 -> 14): 5 per dimension -> 25; the stride-2 conv backward of
    Generator.3 (7x7 grid reading 14x14): 3 per dimension -> 9, the interior class holds 25 positions x 25 taps."""
    for kind, p, want_classes, biggest in [("deconv_fwd", (4, 4, 7, 7, 256, 128, 128), 16, (4, 9 * 8)),
                                           ("deconv_fwd", (7, 7, 14, 14, 128, 64, 64), 25, (25, 9 * 4)),
                                           ("deconv_bwd", (7, 7, 14, 14, 128, 64, 128), 9, (25, 25 * 2))]:
        h1, h2, info, b = batched(lib, kind, *p)
        assert b["n_classes"] == want_classes and b["classes"][0] == biggest
        assert [c[1] for c in b["classes"]] == sorted([c[1] for c in b["classes"]], reverse=True)
        assert sum(s * k for s, k in b["classes"]) * 32 * info["ncols"] == info["macs"]     # valid taps only, nothing added
        lib.dgp2_free(h2); lib.dgp_free(h1)


@pytest.mark.parametrize("kind,p,n_rows,alpha", [
    ("deconv_fwd", (4, 4, 7, 7, 64, 128, 128), 37, 1e30),      # ragged M tiles, one job per tile
    ("deconv_fwd", (4, 4, 7, 8, 64, 128, 128), 40, 0.05),      # everything cut to quarters, pitch > used extent
    ("deconv_fwd", (7, 7, 14, 14, 32, 64, 64), 9, 0.0),        # 64-column family, cutting chosen by simulated makespan
    ("deconv_bwd", (4, 4, 7, 7, 128, 64, 128), 50, 0.3),
    ("deconv_bwd", (7, 7, 14, 14, 64, 32, 64), 21, 0.0),
])
def test_job_lists_cover_every_output_once_and_match_the_oracle(lib, kind, p, n_rows, alpha):
    rs = np.random.RandomState(5)
    h1, h2, info, b = batched(lib, kind, *p)
    jobs = jobs_of(lib, h2, n_rows, 16, alpha)
    assert len(jobs) > 0 and (jobs[:, 5] >= 1).all()
    if kind == "deconv_fwd":
        h_in, pitch_in, e, pitch_out, cin, cout, _ = p
        x = rs.randn(n_rows, pitch_in, pitch_in, cin); F = rs.randn(5, 5, cout, cin); bias = rs.randn(cout)
        out = np.full((n_rows, pitch_out, pitch_out, cout), 777.0)
        touched = apply_jobs(lib, h2, x, F, bias, out, 2)
        want = np.maximum(O.deconv2d(x[:, :h_in, :h_in], F, bias, e), 0)
        np.testing.assert_allclose(out[:, :e, :e], want, rtol=1e-12, atol=1e-12)
        assert (touched[:, :e, :e] == 1).all() and touched.sum() == n_rows * e * e * cout
    else:
        h_in, pitch_out, e, a_pitch, cin, cout, _ = p
        dy = rs.randn(n_rows, a_pitch, a_pitch, cout); F = rs.randn(5, 5, cout, cin)
        Ft = np.ascontiguousarray(F.transpose(0, 1, 3, 2))
        hact = rs.randn(n_rows, pitch_out, pitch_out, cin)
        out = hact.copy()
        touched = apply_jobs(lib, h2, dy, Ft, None, out, 3)
        want = O.deconv2d_backward_input(dy[:, :e, :e], F, h_in) * (hact[:, :h_in, :h_in] > 0)
        np.testing.assert_allclose(out[:, :h_in, :h_in], want, rtol=1e-12, atol=1e-12)
        assert (touched[:, :h_in, :h_in] == 1).all() and touched.sum() == n_rows * h_in * h_in * cin
    lib.dgp2_free(h2); lib.dgp_free(h1)


def test_linear_layers_as_batched_jobs(lib):
    rs = np.random.RandomState(6)
    latent, feat, N = 64, 256, 70
    z = rs.randn(N, latent); W = rs.randn(latent, feat); bvec = rs.randn(feat)
    h1, h2, info, b = batched(lib, "linear_fwd", latent, feat, feat)
    assert b["n_classes"] == 1 and b["family"] == 0
    jobs_of(lib, h2, N, 8, 0.0)
    out = np.zeros((N, feat))
    assert (apply_jobs(lib, h2, z, np.ascontiguousarray(W.T), bvec, out, 1) == 1).all()
    np.testing.assert_allclose(out, z @ W + bvec, rtol=1e-12, atol=1e-12)
    lib.dgp2_free(h2); lib.dgp_free(h1)
    da = rs.randn(N, feat)
    nsplit = 4
    h1, h2, info, b = batched(lib, "linear_bwd", latent, feat, nsplit, latent)
    assert b["n_classes"] == nsplit and b["family"] == 1
    jobs_of(lib, h2, N, 8, 0.0)
    part = np.zeros((N, nsplit, latent))
    assert (apply_jobs(lib, h2, da, W, None, part, 0) == 1).all()
    np.testing.assert_allclose(part.sum(axis=1), da @ W.T, rtol=1e-11, atol=1e-11)
    lib.dgp2_free(h2); lib.dgp_free(h1)


def test_job_cutting_levels_the_end_of_a_launch(lib):
    """BASELINE configs[1] row count (2560): Generator.2 forward as 128x128 tiles alone leaves 980 jobs of very different
    length for 512 slots; cutting the late jobs along M / N (never K) brings the simulated makespan within 8 % of the ideal."""
    h1, h2, info, b = batched(lib, "deconv_fwd", 4, 4, 7, 7, 256, 128, 128)
    plain = jobs_of(lib, h2, 2560, 512, 1e30)
    t_plain = lib.dgp2_predicted_us(h2)
    auto = jobs_of(lib, h2, 2560, 512, 0.0)
    t_auto = lib.dgp2_predicted_us(h2)
    # 980 whole tiles (no class of this layer reaches kPairMinChunks = 80 K chunks: its longest, the 9-tap class, has 72)
    n_paired_tiles = sum(int(np.ceil(2560 * s / 128.0)) for s, k in b["classes"] if k >= 80)
    assert n_paired_tiles == 0 and len(plain) == 980 + n_paired_tiles and (plain[:, 1] == 0).all() and len(auto) > len(plain)
    ideal = 2.0 * info["macs"] * 2560 / 141.5e6               # microseconds at the full-tile rate of the cost model
    assert t_auto < t_plain and t_auto < 1.08 * ideal + 8.0, (t_plain, t_auto, ideal)
    pr = (C.c_int * (4 * len(auto)))(); lib.dgp2_job_pairs(h2, pr)
    pr = np.array(pr).reshape(-1, 4)
    # (the two jobs of a pair sit next to each other and are ordered by the longer half, the second one)
    order_cost = [int(k) * (128 if s == 0 else 64) * (64 if s == 2 else 128) for k, s, (_, pid, role, _) in zip(pr[:, 0], auto[:, 1], pr.tolist())
                  if not (pid and role == 0)]
    assert order_cost[:400] == sorted(order_cost[:400], reverse=True) and auto[0, 1] == 0      # whole long tiles first
    assert set(auto[-64:, 1]) <= {1, 2}                                                        # small pieces last
    lib.dgp2_free(h2); lib.dgp_free(h1)


def test_xcd_locality_order_is_a_permutation_that_gives_each_xcd_one_row_range(lib):
    """dg_plan.h order_for_xcd on CelebA's Generator.5 forward at BASELINE configs[3]'s 1280 rows (5500 jobs, ten dispatch
    rounds): the same jobs (so the same results: every output element is still written exactly once by the same job), slot i
    of the head holds a job of the latent row range of XCD i mod 8 in ascending row order, the tail keeps its longest-first
    order, and the simulated makespan does not move.  With 2-3 rounds (MNIST Generator.2 at 2560 rows) giving up
    longest-first costs tens of percent in the simulation -- which is why the engine only offers the order to its timing when
    the simulation stays within 3 %."""
    import ctypes as C
    lib.dgp2_order_for_xcd.restype = C.c_double
    lib.dgp2_order_for_xcd.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int]

    def read(h2, n):
        buf = (C.c_int * (6 * n))()
        lib.dgp2_jobs(h2, buf)
        return np.array(buf, np.int64).reshape(n, 6)

    N, SLOTS = 1280, 512
    h1, h2, info, b = batched(lib, "deconv_fwd", 16, 16, 32, 32, 64, 64, 64)
    before = jobs_of(lib, h2, N, SLOTS, 0.0)
    t_before = lib.dgp2_predicted_us(h2)
    t_after = lib.dgp2_order_for_xcd(h2, N, 0.75, 8, SLOTS)
    after = read(h2, len(before))
    key = lambda j: tuple(map(tuple, j[np.lexsort(j.T[::-1])]))
    assert key(before) == key(after) and not np.array_equal(before, after)          # a permutation, and not the identity
    head = int(len(before) * 0.75) // 8 * 8
    assert np.array_equal(before[head:], after[head:])                              # the tail is untouched
    xcd_of_rows = np.minimum(after[:head, 3].astype(np.int64) * 8 // N, 7)          # column 3 = n_first
    slot_xcd = np.arange(head) % 8
    assert (xcd_of_rows == slot_xcd).mean() > 0.8                                   # (the row ranges do not hold equally many jobs)
    for x in range(8):
        rows = after[:head][(slot_xcd == x) & (xcd_of_rows == x), 3]
        assert (np.diff(rows) >= 0).all()                                          # each XCD sweeps its range once, ascending
    assert t_after <= 1.01 * t_before, (t_before, t_after)
    lib.dgp2_free(h2); lib.dgp_free(h1)
    # MNIST Generator.2 forward, 2560 rows: 2.7 rounds -- the simulation shows what the order would cost there
    h1, h2, info, b = batched(lib, "deconv_fwd", 4, 4, 7, 7, 256, 128, 128)
    jobs_of(lib, h2, 2560, SLOTS, 0.0)
    t0 = lib.dgp2_predicted_us(h2)
    assert lib.dgp2_order_for_xcd(h2, 2560, 0.75, 8, SLOTS) > 1.2 * t0
    lib.dgp2_free(h2); lib.dgp_free(h1)
    # a small problem end to end: the permuted list still covers every output once with the oracle's values
    rs = np.random.RandomState(4)
    n_rows = 64
    h1, h2, info, b = batched(lib, "deconv_fwd", 4, 4, 7, 7, 64, 128, 128)
    jobs_of(lib, h2, n_rows, 8, 0.0)
    lib.dgp2_order_for_xcd(h2, n_rows, 0.75, 8, 8)
    x = rs.randn(n_rows, 4, 4, 64); F = rs.randn(5, 5, 128, 64); bias = rs.randn(128)
    out = np.full((n_rows, 7, 7, 128), 777.0)
    touched = apply_jobs(lib, h2, x, F, bias, out, 2)
    np.testing.assert_allclose(out, np.maximum(O.deconv2d(x, F, bias, 7), 0), rtol=1e-12, atol=1e-12)
    assert (touched == 1).all()
    lib.dgp2_free(h2); lib.dgp_free(h1)


def test_tuning_records_round_trip_to_the_same_job_lists(lib):
    """dg_export_tuning / dg_import_tuning (include/defensegan_hip.h): a job list is a pure function of the few numbers of its
    record (row count, starting level, cutting threshold, order variant), so the text one process exports rebuilds, in another,
    exactly the list the first one timed and kept -- byte for byte, for every variant the timing offers."""
    h, _ = build(lib, "deconv_bwd", 7, 7, 14, 14, 128, 64, 128)          # MNIST Generator.3 backward
    b = lib.dgp2_build(h)
    lines = []
    variants = [(2560, 0, 1e30, 0, 0.0, 0.0), (2560, 1, 0.97, 0, 0.0, 0.0), (2560, 1, 1.04, 1, 0.0, 0.0), (500, 2, 0.85, 1, 0.0, 0.0),
                (2560, 0, 0.0, 0, 0.0, 0.0), (1280, 1, 1.0, 0, 0.5, 0.0), (37, 2, 1.1, 0, 0.0, 0.0),
                (2560, 1, 1e30, 0, 0.0, 0.65), (2560, 0, 1.0, 0, 0.0, 0.5)]
    slots = {0: 2, 1: 3, 2: 5}
    for n_rows, lvl, slack, snake, xhead, taper in variants:
        line = C.create_string_buffer(256)
        n = lib.dgp2_make_recorded(b, b"B3", n_rows, 256, slots[lvl], lvl, slack, snake, xhead, taper, line, 256)
        assert n > 0
        text = line.value.decode()
        f = text.split()
        assert f[0] == "B3" and int(f[1]) == n_rows and int(f[2]) == lvl and float(f[3]) == slack and int(f[7]) == n and float(f[9]) == taper
        lines.append(text)
        # the record alone rebuilds the list that is current
        assert lib.dgp2_rebuild_matches(b, text.encode(), 0, 256, slots[lvl]) == 1
        # ... and the order variant is part of it (the record is not vacuous): the same record with the snake flag flipped
        # describes another list whenever more than one round of 256 jobs exists
        if n > 256:
            f2 = list(f)
            f2[4] = str(1 - snake)
            assert lib.dgp2_rebuild_matches(b, (" ".join(f2) + "\n").encode(), 0, 256, slots[lvl]) == 0
    # a text of several records: each is found by position; garbage is reported, not skipped
    text = "".join(lines)
    n_rows, lvl, slack, snake, xhead, taper = variants[-1]
    assert lib.dgp2_rebuild_matches(b, text.encode(), len(variants) - 1, 256, slots[lvl]) == 1
    assert lib.dgp2_rebuild_matches(b, (text + "B3 what\n").encode(), len(variants), 256, 2) == -1
    assert lib.dgp2_rebuild_matches(b, text.encode(), len(variants), 256, 2) == -1
    # a text without the tenth field (exported before the taper existed) still parses: taper 0
    n_rows, lvl, slack, snake, xhead, taper = variants[1]
    line = C.create_string_buffer(256)
    lib.dgp2_make_recorded(b, b"B3", n_rows, 256, slots[lvl], lvl, slack, snake, xhead, 0.0, line, 256)
    old = " ".join(line.value.decode().split()[:9]) + "\n"
    assert lib.dgp2_rebuild_matches(b, old.encode(), 0, 256, slots[lvl]) == 1
    # ... and one without the eleventh (before the wave priorities): priority mode 0
    ten = " ".join(line.value.decode().split()[:10]) + "\n"
    assert len(line.value.decode().split()) == 12 and lib.dgp2_rebuild_matches(b, ten.encode(), 0, 256, slots[lvl]) == 1
    # ... and one without the twelfth (before the kernel instantiation became a timed choice): the plain instantiation
    eleven = " ".join(line.value.decode().split()[:11]) + "\n"
    assert lib.dgp2_rebuild_matches(b, eleven.encode(), 0, 256, slots[lvl]) == 1
    lib.dgp2_free(b)
    lib.dgp_free(h)


def test_wave_priorities_follow_the_predicted_job_length_and_change_nothing_else(lib):
    """JobDesc::prio (s_setprio of the job's workgroup): 3 for jobs predicted within a quarter of the longest job of the list, 2, 1, 0
    below; same jobs, same order; recorded as the eleventh field of a tuning record and rebuilt from it; mode 0 clears them."""
    h, info = build(lib, "deconv_bwd", 4, 4, 8, 8, 256, 128, 256)            # CelebA's 4x4 <- 8x8 backward: 36 .. 100 K chunks per job
    b = lib.dgp2_build(h)
    line = C.create_string_buffer(256)
    n = lib.dgp2_make_recorded(b, b"B2", 1280, 256, 3, 1, 1.04, 0, 0.0, 0.0, line, 256)
    before = (C.c_int * (6 * n))(); lib.dgp2_jobs(b, before)
    pr = (C.c_int * n)()
    lib.dgp2_assign_priorities(b, 256, 3, 1, pr)
    after = (C.c_int * (6 * n))(); lib.dgp2_jobs(b, after)
    assert list(before) == list(after)                                        # cls, shape, n0, n_first, j_first, m_valid untouched
    jobs = np.array(after).reshape(n, 6)
    own = (C.c_int * (4 * n))(); lib.dgp2_job_pairs(b, own)
    chunks = np.array(own).reshape(n, 4)[:, 0]                                # the job's OWN K chunks (K-pair jobs: half of the class's)
    work = chunks * np.array([128 * 128, 64 * 128, 64 * 64])[jobs[:, 1]]      # MFMA work of the job's tile x K
    pr = np.array(pr)
    assert set(pr) <= {0, 1, 2, 3} and pr.max() == 3 and pr.min() < 3
    assert pr[np.argmax(work)] == 3
    order = np.argsort(work, kind="stable")
    assert (np.diff(pr[order]) >= 0).all()                                    # monotone in the job's work
    # the record carries the mode: rebuilt with it -> the list that is current; without it -> another one (the prio fields differ)
    with_prio = C.create_string_buffer(256)
    assert lib.dgp2_format_with_prio(b, line.value, 1, with_prio, 256) == 0
    assert with_prio.value.decode().split()[10] == "1"
    assert lib.dgp2_rebuild_matches(b, with_prio.value, 0, 256, 3) == 1
    assert lib.dgp2_rebuild_matches(b, line.value, 0, 256, 3) == 0
    pr0 = (C.c_int * n)()
    lib.dgp2_assign_priorities(b, 256, 3, 0, pr0)
    assert not np.array(pr0).any() and lib.dgp2_rebuild_matches(b, line.value, 0, 256, 3) == 1
    lib.dgp2_free(b)
    lib.dgp_free(h)


def test_tuning_text_id_ignores_order_and_measured_durations():
    from defensegan_amd.gan import tuning_text_id
    head = "dgtune 1 arch 0 latent 128 net_dim 64 use_bn 0 nsplit 16 cus 256\n"
    a = head + "F2 2560 1 0.97 0 0 0 2024 293.500\nB2 2560 1 1 0 0 0 1344 285.600\n"
    b = head + "B2 2560 1 1 0 0 0 1344 281.000\nF2 2560 1 0.97 0 0 0 2024 299.500\n"
    c = head + "B2 2560 1 1 1 0 0 1344 281.000\nF2 2560 1 0.97 0 0 0 2024 299.500\n"
    assert tuning_text_id(a) == tuning_text_id(b) != tuning_text_id(c)
    # the records this build writes have a tenth field, the taper (it changes the list; measured_us, the ninth, does not)
    a10 = head + "F2 2560 1 1e+30 0 0 0 2384 266.517 0.65000000000000002\nB2 2560 0 1 0 0 0 1472 275.302 0.5\n"
    b10 = head + "B2 2560 0 1 0 0 0 1472 279.000 0.5\nF2 2560 1 1e+30 0 0 0 2384 270.100 0.65000000000000002\n"
    c10 = head + "B2 2560 0 1 0 0 0 1472 275.302 0.65\nF2 2560 1 1e+30 0 0 0 2384 266.517 0.65000000000000002\n"
    assert tuning_text_id(a10) == tuning_text_id(b10) != tuning_text_id(c10)
    # a nine-field record of a round-4a text is a taper of 0
    assert tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 293.500\n") == tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 1.0 0\n")
    # the eleventh field, the priority mode, names another list; absent = 0
    assert tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 1.0 0 0\n") == tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 2.0 0\n")
    assert tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 1.0 0 1\n") != tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 1.0 0 0\n")
    # the twelfth, the kernel instantiation the list runs on; absent = 0
    assert tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 1.0 0 0 0\n") == tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 1.0 0 0\n")
    assert tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 1.0 0 0 1\n") != tuning_text_id(head + "F2 2560 1 0.97 0 0 0 2024 1.0 0 0 0\n")


def test_tapered_lists_end_on_small_jobs_and_still_cover_everything(lib):
    """JobModel::taper: pieces may only start before taper x the ideal makespan at full size (level 1: before the midpoint to the
    end); the list is the same work (every output element exactly once, same arithmetic) cut differently -- checked like any list --
    and its last jobs are small."""
    rs = np.random.RandomState(5)
    h, info = build(lib, "deconv_bwd", 4, 4, 7, 7, 128, 32, 128)         # 128 output columns (family 0), K = 32 per tap
    b = lib.dgp2_build(h)
    line = C.create_string_buffer(256)
    n_rows = 700
    n_plain = lib.dgp2_make_recorded(b, b"B2", n_rows, 16, 3, 1, 1e30, 0, 0.0, 0.0, line, 256)
    n_taper = lib.dgp2_make_recorded(b, b"B2", n_rows, 16, 3, 1, 1e30, 0, 0.0, 0.5, line, 256)
    assert n_taper > n_plain                                   # late pieces were cut
    jobs = (C.c_int * (6 * n_taper))()
    lib.dgp2_jobs(b, jobs)
    shapes = np.asarray(jobs).reshape(-1, 6)[:, 1]
    assert shapes[: n_taper // 4].min() == 1 and shapes[-max(4, n_taper // 10):].min() == 2      # starts whole (level 1), ends on quarters
    A = rs.standard_normal((n_rows, info["a_rowstride"]))
    W = rs.standard_normal(25 * 128 * 32)
    out = np.zeros((n_rows, info["out_rowstride"]))
    touched = np.zeros(out.shape, np.int32)
    lib.dgp2_apply(b, A.ctypes.data, W.ctypes.data, np.zeros(1).ctypes.data, out.ctypes.data, touched.ctypes.data, 0)
    ref = np.zeros_like(out)
    apply(lib, h, A, W, None, ref, n_rows, 0)
    assert (touched == 1).all() and np.allclose(out, ref, rtol=1e-12, atol=1e-12)
    lib.dgp2_free(b)
    lib.dgp_free(h)


def test_committed_tuning_files_describe_lists_this_planner_builds(lib):
    """profiles/<round>_tuning_{mnist,celeba}.txt (bench.TUNING_FILE) are what bench.py installs by default (dg_import_tuning) so that its launches are the
    ones the committed rocprofv3 / PMC evidence was collected on.  The engine refuses a record whose job count this build's
    planner does not reproduce (bench.py then times afresh and `roofline.traffic` goes null): a change to dg_plan.cpp that
    orphans the committed files must show up here, on the CPU."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    layers = {       # the engine's build_plans() for NET_DIM 64 without Batchnorm: (kind, h_in, in_pitch, e_out, out_pitch, cin, cout, bn)
        "mnist": {"F2": ("deconv_fwd", 4, 4, 7, 7, 256, 128, 128), "F3": ("deconv_fwd", 7, 7, 14, 14, 128, 64, 64),
                  "B2": ("deconv_bwd", 4, 4, 7, 7, 256, 128, 256), "B3": ("deconv_bwd", 7, 7, 14, 14, 128, 64, 128)},
        "celeba": {"F2": ("deconv_fwd", 4, 4, 8, 8, 256, 128, 128), "F3": ("deconv_fwd", 8, 8, 16, 16, 128, 64, 64),
                   "F5": ("deconv_fwd", 16, 16, 32, 32, 64, 64, 64), "B2": ("deconv_bwd", 4, 4, 8, 8, 256, 128, 256),
                   "B3": ("deconv_bwd", 8, 8, 16, 16, 128, 64, 128), "B5": ("deconv_bwd", 16, 16, 32, 32, 64, 64, 64)},
    }
    slots = {0: 2, 1: 3, 2: 5}                       # dg_handle::job_slots_per_cu
    checked = 0
    import bench
    for wl, table in [(w, t) for w, t in layers.items() for _ in (0,)]:
        path = os.path.join(root, "profiles", bench.TUNING_FILE % wl)      # the file bench.py installs (this round's)
        if not os.path.exists(path):
            continue
        lines = open(path).read().splitlines()
        head = lines[0].split()
        assert head[:2] == ["dgtune", "1"] and head[head.index("net_dim") + 1] == "64" and head[head.index("use_bn") + 1] == "0"
        cus = int(head[head.index("cus") + 1])
        seen = set()
        for line in lines[1:]:
            f = line.split()
            if f[0] not in table:
                continue                              # Linear layers: on the weight-stationary kernels, no job list in use
            h, _ = build(lib, *table[f[0]])
            b = lib.dgp2_build(h)
            rc = lib.dgp2_rebuild_matches(b, (line + "\n").encode(), 0, cus, slots[int(f[2])])
            assert rc >= 0, (wl, line, rc)            # -2: the planner makes another number of jobs, -1: malformed
            lib.dgp2_free(b)
            lib.dgp_free(h)
            seen.add(f[0])
            checked += 1
        assert seen == set(table), (wl, seen)
    if not checked:
        pytest.skip("no committed tuning files")


def test_every_class_with_taps_is_a_rectangular_grid(lib):
    """The device computes a position's offsets from the job record (ClassDesc wc / a_base / a_rs / a_cs / o_*, dg_types.h) instead
    of reading the position tables: holds for every class of every GEMM layer of both generators (with and without Batchnorm's
    padded maps) -- the formula reproduces the tables exactly; a class that is not a grid (none with taps) keeps wc = 0."""
    layers = [("deconv_fwd", (4, 4, 7, 7, 256, 128, 128)), ("deconv_fwd", (7, 7, 14, 14, 128, 64, 64)),
              ("deconv_bwd", (4, 4, 7, 7, 256, 128, 256)), ("deconv_bwd", (7, 7, 14, 14, 128, 64, 128)),
              ("deconv_fwd", (4, 4, 8, 8, 256, 128, 128)), ("deconv_fwd", (8, 8, 16, 16, 128, 64, 64)),
              ("deconv_fwd", (16, 16, 32, 32, 64, 64, 64)), ("deconv_bwd", (4, 4, 8, 8, 256, 128, 256)),
              ("deconv_bwd", (8, 8, 16, 16, 128, 64, 128)), ("deconv_bwd", (16, 16, 32, 32, 64, 64, 64)),
              ("deconv_fwd", (4, 4, 8, 8, 256, 128, 128)), ("deconv_bwd", (7, 8, 14, 14, 128, 64, 128)),
              ("linear_fwd", (128, 4096, 4096)), ("linear_bwd", (128, 4096, 16, 128))]
    for kind, p in layers:
        h1, h2, info, b = batched(lib, kind, *p)
        g = (C.c_int * (2 * b["n_classes"]))()
        lib.dgp2_class_grids(h2, g)
        for i, (s, k) in enumerate(b["classes"]):
            wc, same = g[2 * i], g[2 * i + 1]
            assert same == 1
            if k > 0:
                assert wc >= 1 and s % wc == 0, (kind, p, i, s, wc)
        lib.dgp2_free(h2); lib.dgp_free(h1)


def test_balance_order_gives_every_cu_the_same_work(lib):
    """A list that fits the resident slots is dispatched workgroup i -> CU i mod 256 and a CU shares its matrix pipes among its
    resident jobs: balance_order (TuneRecord.snake = 2) permutes the list so that positions b, b + 256, b + 512, ... -- CU b's jobs --
    carry nearly the same predicted work for every b (plain longest-first order hands CU 0 the longest job of every round)."""
    h, info = build(lib, "deconv_fwd", 7, 7, 14, 14, 128, 64, 64)            # MNIST Generator.3 forward at 400 rows
    b = lib.dgp2_build(h)
    cl = (C.c_int * 128)(); lib.dgp2_classes(b, cl)
    chunks_of = np.array(cl).reshape(-1, 2)[:, 1]
    area = np.array([256 * 64, 128 * 64, 64 * 64])

    def per_cu(order_flag):
        line = C.create_string_buffer(256)
        n = lib.dgp2_make_recorded(b, b"F3", 400, 256, 5, 2, 1e30, order_flag, 0.0, 0.0, line, 256)      # everything in quarters: <= 5 per CU
        assert 1024 < n <= 1280 and lib.dgp2_rebuild_matches(b, line.value, 0, 256, 5) == 1               # the record rebuilds the order
        j = (C.c_int * (6 * n))(); lib.dgp2_jobs(b, j)
        j = np.array(j).reshape(n, 6)
        work = chunks_of[j[:, 0]] * area[j[:, 1]]
        key = sorted(map(tuple, j.tolist()))
        return np.array([work[c::256].sum() for c in range(256)]), key
    plain, k0 = per_cu(0)
    snake, k1 = per_cu(1)
    bal, k2 = per_cu(2)
    assert k0 == k1 == k2                                                     # permutations of one list
    spread = lambda w: (w.max() - w.min()) / w.mean()
    print("per-CU work spread (max - min) / mean: plain %.3f, snake %.3f, balanced %.3f" % (spread(plain), spread(snake), spread(bal)))
    assert spread(bal) < 0.6 * spread(plain) and spread(bal) <= spread(snake)
    # a list with more jobs than resident slots is left in its order
    line = C.create_string_buffer(256)
    n = lib.dgp2_make_recorded(b, b"F3", 2560, 256, 3, 1, 1e30, 2, 0.0, 0.0, line, 256)
    j2 = (C.c_int * (6 * n))(); lib.dgp2_jobs(b, j2)
    n0 = lib.dgp2_make_recorded(b, b"F3", 2560, 256, 3, 1, 1e30, 0, 0.0, 0.0, line, 256)
    j0 = (C.c_int * (6 * n0))(); lib.dgp2_jobs(b, j0)
    assert n == n0 > 768 and list(j2) == list(j0)
    lib.dgp2_free(b)
    lib.dgp_free(h)


def test_spread_order_mixes_lengths_in_the_first_round(lib):
    """spread_order (TuneRecord.snake = 3) on a list of several dispatch rounds: a permutation; the first round of 768 jobs is no
    longer ONE length (longest-first order: Generator.3's backward at 2560 rows starts with 768 jobs of 50 chunks, which end
    together and whose successors start together); half of it still is the head of the list, and everything behind the first
    round is in its old (longest-first) order."""
    h, info = build(lib, "deconv_bwd", 7, 7, 14, 14, 128, 64, 128)           # MNIST Generator.3 backward
    b = lib.dgp2_build(h)
    cl = (C.c_int * 128)(); lib.dgp2_classes(b, cl)
    chunks_of = np.array(cl).reshape(-1, 2)[:, 1]
    area = np.array([128 * 128, 64 * 128, 64 * 64])

    def jobs_of(flag):
        line = C.create_string_buffer(256)
        n = lib.dgp2_make_recorded(b, b"B3", 2560, 256, 3, 1, 1e30, flag, 0.0, 0.5, line, 256)
        assert lib.dgp2_rebuild_matches(b, line.value, 0, 256, 3) == 1
        j = (C.c_int * (6 * n))(); lib.dgp2_jobs(b, j)
        return np.array(j).reshape(n, 6)
    plain, spread = jobs_of(0), jobs_of(3)
    assert sorted(map(tuple, plain.tolist())) == sorted(map(tuple, spread.tolist()))
    work = lambda j: chunks_of[j[:, 0]] * area[j[:, 1]]
    wp, ws = work(plain), work(spread)
    assert len(np.unique(wp[:768])) == 1 and len(np.unique(ws[:768])) >= 4   # one length -> a mix
    assert (ws[:768] == ws.max()).sum() >= 384                                # at least half of the round is still the longest jobs
    assert (np.diff(ws[768:]) <= 0).all()                                     # behind the first round: longest first as before
    lib.dgp2_free(b)
    lib.dgp_free(h)


def test_k_pair_jobs_split_the_taps_of_the_long_classes_in_two(lib):
    """Classes of >= 80 K chunks (dg_plan.h kPairMinChunks) are computed by K-pair jobs: every tile of such a class appears as TWO
    jobs of the same shape and extent, next to each other in the list, whose tap ranges partition the class's taps (first half
    floor(taps / 2)); each pair owns two disjoint accumulator images in the scratch; classes below the threshold are untouched; the
    host executor (which adds the halves) still writes every output element exactly once with the right value."""
    rs = np.random.RandomState(11)
    n_rows = 70
    h1, h2, info, b = batched(lib, "deconv_bwd", 4, 4, 7, 7, 256, 128, 256)          # MNIST Generator.2 backward: 100 / 80 / 64 / 40 / 32 / 16 chunks
    jobs = jobs_of(lib, h2, n_rows, 512, 0.0)
    pr = (C.c_int * (4 * len(jobs)))(); lib.dgp2_job_pairs(h2, pr)
    pr = np.array(pr).reshape(-1, 4)
    class_chunks = np.array([k for _, k in b["classes"]])[jobs[:, 0]]
    paired = pr[:, 1] != 0
    assert (paired == (class_chunks >= 80)).all() and paired.any() and (~paired).any() and 64 in set(class_chunks[~paired])
    assert (pr[~paired, 0] == class_chunks[~paired]).all()
    idx = np.nonzero(paired)[0]
    assert len(idx) % 2 == 0
    area = np.array([128 * 128, 64 * 128, 64 * 64])
    spans = []
    for a, c in zip(idx[0::2], idx[1::2]):
        assert c == a + 1 and (jobs[a] == jobs[c]).all()                             # same class, shape, columns, rows
        assert pr[a, 1] == pr[c, 1] and (pr[a, 2], pr[c, 2]) == (0, 1) and pr[a, 3] == pr[c, 3]
        taps = class_chunks[a] // 4                                                  # kch = 128: 4 chunks per tap
        assert pr[a, 0] == (taps // 2) * 4 and pr[a, 0] + pr[c, 0] == class_chunks[a]
        spans.append((256 * pr[a, 3], 256 * pr[a, 3] + 2 * area[jobs[a, 1]]))     # pair_off is in units of 256 floats
    spans.sort()
    assert all(e0 <= s1 for (_, e0), (s1, _) in zip(spans, spans[1:]))               # the pairs' images do not overlap
    assert sorted(set(pr[paired, 1])) == list(range(1, len(idx) // 2 + 1))           # counters 0 .. pairs - 1
    dy = rs.randn(n_rows, 7, 7, 128); Ft = rs.randn(25, 256, 128)
    out = rs.rand(n_rows, 4, 4, 256) - 0.3
    gate = out > 0
    touched = apply_jobs(lib, h2, dy, Ft, None, out, 3)
    assert (touched == 1).all()
    ref = np.zeros((n_rows, 4, 4, 256))
    for oh in range(4):
        for ow in range(4):
            for kh in range(5):
                for kw in range(5):
                    i, j = 2 * oh + kh - 1, 2 * ow + kw - 1
                    if 0 <= i < 7 and 0 <= j < 7:
                        ref[:, oh, ow, :] += dy[:, i, j, :] @ Ft[kh * 5 + kw].T
    np.testing.assert_allclose(out, np.where(gate, ref, 0.0), rtol=1e-10, atol=1e-10)
    lib.dgp2_free(h2); lib.dgp_free(h1)


def test_which_classes_are_paired_depends_on_the_layer_plan_only(lib):
    """The invariant behind "rows are independent of their batch" now that tiles ARE cut along K (gan_defense.rows_are_independent):
    which tap classes are computed by K-pair jobs, and where their taps are split, is a function of the layer plan alone -- the
    same for every row count, cutting threshold and slot count -- so an output element's summation tree never depends on the
    call it is part of."""
    for layer in (("deconv_bwd", 4, 4, 7, 7, 256, 128, 256), ("deconv_bwd", 4, 4, 8, 8, 256, 128, 256), ("deconv_bwd", 7, 7, 14, 14, 128, 64, 128),
                  ("deconv_fwd", 4, 4, 7, 7, 256, 128, 128)):
        h1, h2, info, b = batched(lib, *layer)
        seen = {}
        for n_rows, slots, slack in ((3, 512, 0.0), (70, 512, 0.0), (500, 768, 1e30), (500, 1280, 0.01), (2560, 512, 0.0), (2560, 768, 0.97)):
            jobs = jobs_of(lib, h2, n_rows, slots, slack)
            pr = (C.c_int * (4 * len(jobs)))(); lib.dgp2_job_pairs(h2, pr)
            pr = np.array(pr).reshape(-1, 4)
            for (cls, *_), (own, pid, role, _) in zip(jobs.tolist(), pr.tolist()):
                key = (cls, role if pid else -1)
                assert seen.setdefault(key, own) == own, (layer, n_rows, key)            # the same K extent per (class, half) everywhere
            paired_now = {c for (c, r) in seen if r >= 0}
            assert all((c, -1) not in seen for c in paired_now), (layer, n_rows)          # a class is paired in every list or in none
        lib.dgp2_free(h2); lib.dgp_free(h1)


def test_jobs_balanced_builds_a_single_round_list_of_equal_work_per_cu(lib):
    """dg_plan.h jobs_balanced (TuneRecord.snake = 4): a list for ONE dispatch round -- at most cus * slots_per_cu jobs, CU b's jobs at
    the positions b, b + 256, ... -- whose tiles are cut (along M / N only) until the heaviest CU carries at most a few percent more
    than the mean.  CelebA's 4x4 <- 8x8 backward at 1280 rows as whole tiles is 480 jobs for 256 CUs: 32 CUs would hold one job.
    Every output element is still written exactly once with the right value; the record rebuilds the list; a layer that does not
    fit one round gives no list."""
    rs = np.random.RandomState(12)
    h1, h2, info, b = batched(lib, "deconv_bwd", 4, 4, 8, 8, 256, 128, 256)
    n_rows, cus, spc = 1280, 256, 2
    n = lib.dgp2_make_balanced(h2, n_rows, cus, spc, 0)
    assert 480 < n <= cus * spc
    jobs = (C.c_int * (6 * n))(); lib.dgp2_jobs(h2, jobs)
    jobs = np.array(jobs).reshape(n, 6)
    pr = (C.c_int * (4 * n))(); lib.dgp2_job_pairs(h2, pr)
    own = np.array(pr).reshape(n, 4)[:, 0]
    work = own * np.array([128 * 128, 64 * 128, 64 * 64])[jobs[:, 1]]
    per_cu = np.array([work[c::cus].sum() for c in range(cus)])               # (rounds are prefixes: position r * 256 + b is CU b's r-th job)
    assert (np.bincount(np.arange(n) % cus, minlength=cus) <= spc).all()
    over = lambda w: (w.max() - w.mean()) / w.mean()
    print("heaviest CU above the mean: balanced %.3f" % over(per_cu))
    # whole tiles in longest-first order: the same measure
    plain = jobs_of(lib, h2, n_rows, cus * spc, 1e30)
    pr0 = (C.c_int * (4 * len(plain)))(); lib.dgp2_job_pairs(h2, pr0)
    w0 = np.array(pr0).reshape(-1, 4)[:, 0] * np.array([128 * 128, 64 * 128, 64 * 64])[plain[:, 1]]
    per_cu0 = np.array([w0[c::cus].sum() for c in range(cus)])
    print("heaviest CU above the mean: whole tiles, longest first %.3f" % over(per_cu0))
    # (two slots per CU at this level and 480 whole tiles: only 32 cuts fit the round -- the balance improves, it cannot become flat)
    assert over(per_cu) < 0.75 * over(per_cu0)
    # with room to cut (quarters, five slots per CU) the heaviest CU ends within a few percent of the mean
    n5 = lib.dgp2_make_balanced(h2, 500, cus, 5, 2)
    j5 = (C.c_int * (6 * n5))(); lib.dgp2_jobs(h2, j5); j5 = np.array(j5).reshape(n5, 6)
    p5 = (C.c_int * (4 * n5))(); lib.dgp2_job_pairs(h2, p5)
    w5 = np.array(p5).reshape(n5, 4)[:, 0] * np.array([128 * 128, 64 * 128, 64 * 64])[j5[:, 1]]
    cu5 = np.array([w5[c::cus].sum() for c in range(cus)])
    print("500 rows, quarters: %d jobs, heaviest CU above the mean %.3f" % (n5, over(cu5)))
    assert 0 < n5 <= cus * 5 and over(cu5) < 0.12
    # correctness of the cut list (small row count so that the host executor is quick)
    n_small = lib.dgp2_make_balanced(h2, 60, 16, 5, 1)
    assert n_small > 0
    dy = rs.randn(60, 8, 8, 128); Ft = rs.randn(25, 256, 128)
    out = rs.rand(60, 4, 4, 256) - 0.3
    gate = out > 0
    touched = apply_jobs(lib, h2, dy, Ft, None, out, 3)
    assert (touched == 1).all()
    ref = np.zeros((60, 4, 4, 256))
    for oh in range(4):
        for ow in range(4):
            for kh in range(5):
                for kw in range(5):
                    i, j = 2 * oh + kh - 1, 2 * ow + kw - 1
                    if 0 <= i < 8 and 0 <= j < 8:
                        ref[:, oh, ow, :] += dy[:, i, j, :] @ Ft[kh * 5 + kw].T
    np.testing.assert_allclose(out, np.where(gate, ref, 0.0), rtol=1e-10, atol=1e-10)
    # the tuning record (order code 4) rebuilds the list; a layer too big for one round has no such list
    line = C.create_string_buffer(256)
    n2 = lib.dgp2_make_recorded(h2, b"B2", n_rows, cus, spc, 0, 0.0, 4, 0.0, 0.0, line, 256)
    assert n2 == n and lib.dgp2_rebuild_matches(h2, line.value, 0, cus, spc) == 1
    assert lib.dgp2_make_balanced(h2, 12500, cus, spc, 0) == 0
    lib.dgp2_free(h2); lib.dgp_free(h1)


def test_random_sweep_of_list_variants_covers_every_output_once_with_the_right_value(lib):
    """Every way the engine can build a job list (dg_plan.h jobs_from_record: starting level, cutting threshold, boustrophedon /
    balance_order / spread_order, XCD order, taper; jobs_balanced) over the geometries of both generators (with and without K-pair
    classes), at seeded random row counts and CU counts small enough that lists span several dispatch rounds: the host executor
    writes every output element exactly once and reproduces the dense oracle.  36 seeded draws."""
    rs = np.random.RandomState(2025)
    geoms = [("deconv_fwd", 4, 7), ("deconv_fwd", 4, 8), ("deconv_fwd", 7, 14), ("deconv_fwd", 8, 16),
             ("deconv_bwd", 4, 7), ("deconv_bwd", 4, 8), ("deconv_bwd", 7, 14), ("deconv_bwd", 8, 16)]
    slots = {0: 2, 1: 3, 2: 5}
    seen_pairs = seen_balanced = 0
    for draw in range(36):
        kind, h_in, e = geoms[draw % len(geoms)]
        fwd = kind == "deconv_fwd"
        cin, cout = (int(rs.choice([64, 128])), int(rs.choice([64, 128]))) if draw % 3 else (256, 128)
        if h_in >= 7:                                     # the big grids: keep the executor quick
            cin, cout = min(cin, 128), 64
        bn = cout if fwd else cin                         # as the engine builds its plans: one column block = all output channels
        n_rows = int(rs.randint(1, 70))
        cus = int(rs.choice([4, 8, 16]))
        balanced = rs.rand() < 0.3
        if balanced:                                      # a single-round list: more pieces than CUs, at most cus * slots jobs
            n_rows = int(rs.randint(1, 24))
        lvl = int(rs.randint(0, 3))
        h1, h2, info, b = batched(lib, kind, h_in, h_in, e, e, cin, cout, bn)
        if balanced:
            n = 0
            for cus in (4, 8, 16, 32, 64, 128, 256):      # the first CU count at which the layer fits one round
                n = lib.dgp2_make_balanced(h2, n_rows, cus, slots[lvl], lvl)
                if n:
                    assert cus < n <= cus * slots[lvl]
                    seen_balanced += 1
                    break
            if n == 0:                                    # no such CU count (fewer pieces than CUs everywhere): the plain list instead
                n = lib.dgp2_make_jobs(h2, n_rows, cus * slots[lvl], 0.0)
        else:
            slack = float(rs.choice([1e30, 1.04, 0.9, 0.0]))
            snake = int(rs.choice([0, 1, 2, 3]))
            xhead = float(rs.choice([0.0, 0.0, 0.5]))
            taper = float(rs.choice([0.0, 0.0, 0.65]))
            line = C.create_string_buffer(256)
            n = lib.dgp2_make_recorded(h2, b"XX", n_rows, cus, slots[lvl], lvl, slack, snake, xhead, taper, line, 256)
            assert lib.dgp2_rebuild_matches(h2, line.value, 0, cus, slots[lvl]) == 1
        assert n > 0
        pr = (C.c_int * (4 * n))(); lib.dgp2_job_pairs(h2, pr)
        seen_pairs += int((np.array(pr).reshape(n, 4)[:, 1] != 0).any())
        if fwd:
            x = rs.randn(n_rows, h_in, h_in, cin); F = rs.randn(5, 5, cout, cin); bias = rs.randn(cout)
            out = np.full((n_rows, e, e, cout), 777.0)
            touched = apply_jobs(lib, h2, x, F, bias, out, 2)
            want = np.maximum(O.deconv2d(x, F, bias, e), 0)
        else:
            dy = rs.randn(n_rows, e, e, cout); F = rs.randn(5, 5, cout, cin)
            hact = rs.rand(n_rows, h_in, h_in, cin) - 0.3
            out = hact.copy()
            touched = apply_jobs(lib, h2, dy, np.ascontiguousarray(F.transpose(0, 1, 3, 2)), None, out, 3)
            want = O.deconv2d_backward_input(dy, F, h_in) * (hact > 0)
        assert (touched == 1).all(), (draw, kind, h_in, e, cin, cout, bn, n_rows, cus, lvl)
        np.testing.assert_allclose(out, want, rtol=1e-10, atol=1e-10, err_msg=str((draw, kind, h_in, e, cin, cout, bn, n_rows, cus, lvl)))
        lib.dgp2_free(h2); lib.dgp_free(h1)
    assert seen_pairs >= 2 and seen_balanced >= 3         # the sweep met K-pair lists and single-round balanced lists


# ---------------------------------------------------------------------------------- fragment-order lists (dg_fgemm.hip)
def apply_frag(lib, h2, n_rows, n_wgs, A, W, bias, out, mode):
    A = np.ascontiguousarray(A, np.float64); W = np.ascontiguousarray(W, np.float64)
    b = np.ascontiguousarray(bias, np.float64) if bias is not None else np.zeros(1)
    touched = np.zeros(out.shape, np.int32)
    info = (C.c_double * 5)()
    rc = lib.dgp2_frag_apply(h2, n_rows, n_wgs, A.ctypes.data, W.ctypes.data, b.ctypes.data, out.ctypes.data, touched.ctypes.data, mode, info)
    return rc, touched, {"n_jobs": int(info[0]), "supported": bool(info[1]), "ksplit": int(info[2]), "heaviest": info[3], "mean": info[4]}


@pytest.mark.parametrize("n_wgs", [0, 6, 512])
@pytest.mark.parametrize("kind,p,n_rows,ksplit", [
    ("deconv_fwd", (4, 4, 7, 7, 256, 128, 128), 37, 4),        # Generator.2 forward: 9 taps x 8 chunks = 72; ragged last row block
    ("deconv_fwd", (7, 7, 14, 14, 128, 64, 64), 64, 2),        # Generator.3 forward: 9 x 4 = 36 chunks
    ("deconv_fwd", (8, 8, 16, 16, 128, 64, 64), 5, 2),         # CelebA's: fewer rows than one row block
    ("deconv_bwd", (7, 7, 14, 14, 128, 64, 128), 33, 2),       # backward of Generator.3: 25 x 2 = 50 chunks, but K extent 64 = 8
                                                               # k8-steps per tap caps the split at 2 (whole ring turns per wave)
    ("deconv_bwd", (4, 4, 7, 7, 256, 128, 256), 70, 4),        # backward of Generator.2: 25 x 4 = 100 chunks
])
def test_fragment_order_lists_cover_every_output_once_and_match_the_oracle(lib, kind, p, n_rows, ksplit, n_wgs):
    """dg_plan.cpp build_frag_jobs (n_wgs = 0) and build_frag_tiles (persistent waves): the records' own magic multipliers and
    affine tap grids, executed on the host, write every used output element exactly once with the layer's value; the persistent
    form hands every wave a contiguous range, and the heaviest wave is within one tile of the mean."""
    rs = np.random.RandomState(11)
    h1, h2, info, b = batched(lib, kind, *p)
    if kind == "deconv_fwd":
        h_in, pitch_in, e, pitch_out, cin, cout, _ = p
        x = rs.randn(n_rows, pitch_in, pitch_in, cin); F = rs.randn(5, 5, cout, cin); bias = rs.randn(cout)
        out = np.full((n_rows, pitch_out, pitch_out, cout), 777.0)
        rc, touched, fi = apply_frag(lib, h2, n_rows, n_wgs, x, F, bias, out, 2)
        assert fi["supported"] and rc == fi["n_jobs"] > 0
        want = np.maximum(O.deconv2d(x[:, :h_in, :h_in], F, bias, e), 0)
        np.testing.assert_allclose(out[:, :e, :e], want, rtol=1e-12, atol=1e-12)
        assert (touched[:, :e, :e] == 1).all() and touched.sum() == n_rows * e * e * cout
    else:
        h_in, pitch_out, e, a_pitch, cin, cout, _ = p
        dy = rs.randn(n_rows, a_pitch, a_pitch, cout); F = rs.randn(5, 5, cout, cin)
        Ft = np.ascontiguousarray(F.transpose(0, 1, 3, 2))
        hact = rs.randn(n_rows, pitch_out, pitch_out, cin)
        out = hact.copy()
        rc, touched, fi = apply_frag(lib, h2, n_rows, n_wgs, dy, Ft, None, out, 3)
        assert fi["supported"] and rc == fi["n_jobs"] > 0
        want = O.deconv2d_backward_input(dy[:, :e, :e], F, h_in) * (hact[:, :h_in, :h_in] > 0)
        np.testing.assert_allclose(out[:, :h_in, :h_in], want, rtol=1e-12, atol=1e-12)
        assert (touched[:, :h_in, :h_in] == 1).all() and touched.sum() == n_rows * h_in * h_in * cin
    if n_wgs == 0:
        assert fi["ksplit"] == ksplit
    else:
        assert fi["ksplit"] == 1                                   # a persistent wave owns its tile's whole K axis
        biggest = max(k for _, k in b["classes"]) + 1.5
        assert fi["heaviest"] <= fi["mean"] + biggest + 1e-9
    lib.dgp2_free(h2); lib.dgp_free(h1)


def test_fragment_order_path_refuses_layers_it_cannot_address(lib):
    """K extent 32 (not 64 / 128 / 256) -> frag_supported is false and the engine keeps dg_gemm.hip (dg_engine.cpp frag_shapes_ok)."""
    h1, h2, info, b = batched(lib, "deconv_fwd", 4, 4, 7, 7, 32, 64, 64)
    out = np.zeros((4, 7, 7, 64))
    rc, touched, fi = apply_frag(lib, h2, 4, 0, np.zeros((4, 4, 4, 32)), np.zeros((5, 5, 64, 32)), np.zeros(64), out, 2)
    assert rc == 0 and not fi["supported"] and touched.sum() == 0
    lib.dgp2_free(h2); lib.dgp_free(h1)
