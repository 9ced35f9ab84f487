"""The per-lane device code (dsp_amd/csrc/dsp_math.h, sync_lane.h) compiled for the host and
checked against the golden vectors / the oracle -- the same source the HIP kernels run, tested
without a GPU.  Gardner is plain IEEE arithmetic and must be bit-exact; the FLL goes through
the library's own sincos / atan2 kernels (abs. error ~2e-16)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import golden, ROOT

HM_DIR = os.path.join(ROOT, "tests", "hostmath")


@pytest.fixture(scope="module")
def hm():
    subprocess.check_call(["make", "-C", HM_DIR, "-s"])
    L = C.CDLL(os.path.join(HM_DIR, "libhostmath.so"))
    L.hm_fll.restype = C.c_double
    L.hm_gardner2.restype = C.c_int64
    L.hm_gardner_ted.restype = C.c_int64
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def hm_g2(L, x, sps, alpha, mu0=0.0, h=0, scale=1.0):
    x = np.ascontiguousarray(x, np.complex128)
    cap = x.size // sps + 8
    rec = np.empty(cap, np.complex128); err = np.empty(cap); mu = np.empty(cap)
    n = L.hm_gardner2(_p(x), C.c_int64(x.size), C.c_int(sps), C.c_double(alpha), C.c_double(mu0),
                      C.c_int(h), C.c_double(scale), _p(rec), _p(err), _p(mu), C.c_int64(cap))
    return rec[:n], err[:n - 1], mu[:n - 1]


def hm_ted(L, x, sps, dmu, h=0):
    x = np.ascontiguousarray(x, np.complex128)
    cap = max(x.size // sps + 8, 8)
    y = np.empty(cap, np.complex128); err = np.empty(cap); mu = np.empty(cap)
    n = L.hm_gardner_ted(_p(x), C.c_int64(x.size), C.c_int(sps), C.c_double(dmu), C.c_int(h),
                         _p(y), _p(err), _p(mu), C.c_int64(cap))
    return y[:n], err[:n], mu[:n]


def hm_fll(L, x, Bn, y, i0=0, state=None):
    x = np.ascontiguousarray(x, np.complex128)
    M = x.size
    kp = 4 * Bn / 1000 / (y + 1 / (4 * y))
    ki = 4 * (Bn / 1000 / (y + 1 / (4 * y))) ** 2
    out = np.empty(M, np.complex128); fr = np.empty(M); ef = np.empty(M); q = np.empty(M, np.uint8)
    cf = L.hm_fll(_p(x), C.c_int64(M), C.c_double(kp), C.c_double(ki), C.c_int64(i0), _p(state),
                  _p(out), _p(fr), _p(ef), _p(q))
    return out, cf, fr, ef, q


def test_sincos_atan2_accuracy(hm):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-1000, 1000, 200000), rng.uniform(-1e7, 1e7, 100000),
                        rng.uniform(-1, 1, 100000), np.arange(-50, 50) * np.pi / 4])
    s = np.empty_like(x); c = np.empty_like(x)
    hm.hm_sincos(_p(x), C.c_int64(x.size), _p(s), _p(c))
    assert np.max(np.abs(s - np.sin(x))) < 4e-16 and np.max(np.abs(c - np.cos(x))) < 4e-16
    y = rng.standard_normal(400000); xx = rng.standard_normal(400000)
    y[:10] = [0, 0, 1, -1, 0.0, -0.0, 1e-300, 5, -5, 1]
    xx[:10] = [1, -1, 0, 0, 0.0, -0.0, 1, 5, 5, -1]
    y[10:20000] *= 1e-3
    o = np.empty_like(y)
    hm.hm_atan2(_p(y), _p(xx), C.c_int64(y.size), _p(o))
    ref = np.arctan2(y, xx)
    assert np.max(np.abs(o - ref)) < 1e-15
    assert np.array_equal(o[:10], ref[:10])      # axes and signed zeros exactly


def test_gardner_bit_exact(hm):
    g = golden("g_config1")
    rec, err, mu = hm_g2(hm, g["y"], 4, 0.007)
    assert np.array_equal(rec, g["rec"]) and np.array_equal(err, g["terr"]) and np.array_equal(mu, g["mu"])
    g = golden("g_gardner2_demo")
    rec, err, mu = hm_g2(hm, g["x"], 8, 0.07)
    assert np.array_equal(rec, g["rec"]) and np.array_equal(mu, g["mu"])
    g = golden("g_edges")
    for nm in ("pos", "mid", "negmid"):
        rec, err, mu = hm_g2(hm, g["wrap_x"], 4, float(g[f"g2_wrap_{nm}_alpha"]))
        assert np.array_equal(rec, g[f"g2_wrap_{nm}_rec"]) and np.array_equal(mu, g[f"g2_wrap_{nm}_mu"])
    rec, err, mu = hm_g2(hm, g["wrap_x"], 4, 0.007, 3.7)
    assert np.array_equal(rec, g["g2_mu0_rec"])
    y, e, m = hm_ted(hm, g["wrap_x"], 4, 0.8)
    assert np.array_equal(y, g["ted_clip_y"]) and np.array_equal(m, g["ted_clip_mu"])
    g = golden("g_hw1_gardner")
    y, e, m = hm_ted(hm, g["baseband"], 2, 0.002)
    assert np.array_equal(y, g["y_f64"]) and np.array_equal(m, g["mu_f64"])


def test_fll_vs_reference(hm):
    g = golden("g_config1")
    out, cf, fr, ef, q = hm_fll(hm, g["rec"], 0.03, 1)
    H = 3300
    assert np.max(np.abs(fr[:H] - g["fll_freq"][:H])) < 1e-15
    assert np.array_equal(q[:H], g["dec"][:H])
    g2 = golden("g_config2")
    out, cf, fr, ef, q = hm_fll(hm, g2["y"][0::4], 1e-6, 1)
    assert np.max(np.abs(fr - g2["fllA_freq"])) < 1e-20
    assert np.max(np.abs(out - g2["fllA_out"])) < 1e-13
    gs = golden("g_synth_bursts")
    for b in range(6):
        n = int(gs["n_recB"][b])
        out, cf, fr, ef, q = hm_fll(hm, gs["recB"][b, :n], 0.03, 1)
        assert np.max(np.abs(fr - gs["fllB_freq"][b, :n])) < 1e-16
        assert np.array_equal(q, gs["decB"][b, :n])
    ge = golden("g_edges")
    out, cf, fr, ef, q = hm_fll(hm, ge["fll_axis_x"], 50.0, 1)
    assert np.max(np.abs(fr - ge["fll_axis_freq"])) < 1e-15
    assert np.max(np.abs(out - ge["fll_axis_out"])) < 1e-13


def hm_fll_fast(L, x, Bn, y, i0=0, blk=8):
    x = np.ascontiguousarray(x, np.complex128)
    M = x.size
    kp = 4 * Bn / 1000 / (y + 1 / (4 * y))
    ki = 4 * (Bn / 1000 / (y + 1 / (4 * y))) ** 2
    out = np.empty(M, np.complex128); fr = np.empty(M); ef = np.empty(M); q = np.empty(M, np.uint8)
    cf = C.c_double(0.0)
    L.hm_fll_fast.restype = C.c_int64
    redone = L.hm_fll_fast(_p(x), C.c_int64(M), C.c_double(kp), C.c_double(ki), C.c_int64(i0), C.c_int64(blk),
                           _p(out), _p(fr), _p(ef), _p(q), C.byref(cf))
    return out, cf.value, fr, ef, q, redone


def test_fll_fast_step_vs_reference(hm):
    """fll_step_fast (the restated iteration the two-wave kernel runs on blocks whose lanes are all live)
    against the reference's vectors with the bars of fll_step, and against fll_step itself."""
    g = golden("g_config1")
    out, cf, fr, ef, q, redone = hm_fll_fast(hm, g["rec"], 0.03, 1)
    H = 3300
    assert redone == 0
    assert np.max(np.abs(fr[:H] - g["fll_freq"][:H])) < 1e-15
    assert np.array_equal(q[:H], g["dec"][:H])
    g2 = golden("g_config2")
    out, cf, fr, ef, q, redone = hm_fll_fast(hm, g2["y"][0::4], 1e-6, 1)
    assert np.max(np.abs(fr - g2["fllA_freq"])) < 1e-20
    assert np.max(np.abs(out - g2["fllA_out"])) < 1e-13
    gs = golden("g_synth_bursts")
    for b in range(6):
        n = int(gs["n_recB"][b])
        out, cf, fr, ef, q, redone = hm_fll_fast(hm, gs["recB"][b, :n], 0.03, 1, blk=3 + b)
        assert np.max(np.abs(fr - gs["fllB_freq"][b, :n])) < 1e-16
        assert np.array_equal(q, gs["decB"][b, :n])
        o0, c0, f0, e0, q0 = hm_fll(hm, gs["recB"][b, :n], 0.03, 1)
        assert np.max(np.abs(out - o0)) < 1e-15 * np.max(np.abs(o0)) and np.max(np.abs(ef - e0)) < 1e-19
    # samples on the axes (a component exactly zero: np.sign(0) = 0): such blocks are run again
    # through fll_step, so the result is fll_step's
    ge = golden("g_edges")
    out, cf, fr, ef, q, redone = hm_fll_fast(hm, ge["fll_axis_x"], 50.0, 1, blk=4)
    assert redone > 0
    assert np.max(np.abs(fr - ge["fll_axis_freq"])) < 1e-15
    assert np.max(np.abs(out - ge["fll_axis_out"])) < 1e-13
    # large absolute indices (inside the loop's stability horizon 1/(pi*kp) for this Bn)
    rng = np.random.default_rng(3)
    x = np.exp(2j * np.pi * (1e-7 * np.arange(4000) + rng.uniform(0, 1))) * (1 + 0.1 * rng.standard_normal(4000))
    a = hm_fll_fast(hm, x, 1e-5, 1, i0=2_000_000, blk=6)
    b = hm_fll(hm, x, 1e-5, 1, i0=2_000_000)
    assert a[5] == 0 and np.max(np.abs(a[2] - b[2])) < 1e-18 and np.array_equal(a[4], b[4])
    assert np.max(np.abs(a[0] - b[0])) < 1e-15 * 1.5


def test_lagrange_vs_reference(hm):
    g = golden("g_lagrange")
    x = np.ascontiguousarray(g["clean"], np.complex128)
    pos = np.ascontiguousarray(g["pos"], np.float64)
    out = np.empty(pos.size, np.complex128)
    hm.hm_lagrange(_p(x), C.c_int64(x.size), _p(pos), C.c_int64(pos.size), C.c_int(4), _p(out))
    assert np.max(np.abs(out - g["arb"])) < 1e-15
    pos = np.arange(x.size, dtype=np.float64) + 0.35
    out = np.empty(pos.size, np.complex128)
    hm.hm_lagrange(_p(x), C.c_int64(x.size), _p(pos), C.c_int64(pos.size), C.c_int(6), _p(out))
    assert np.max(np.abs(out - g["mis"])) < 1e-15
