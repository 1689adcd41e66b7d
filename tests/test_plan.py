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
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-I", CSRC, "-o", SO] + srcs)
    l = C.CDLL(SO)
    l.dgp_build.restype = C.c_void_p
    l.dgp_build.argtypes = [C.c_char_p] + [C.c_int] * 7
    l.dgp_free.argtypes = [C.c_void_p]
    l.dgp_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    l.dgp_tap_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.dgp_apply.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_int]
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
