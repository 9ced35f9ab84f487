"""The oracle (oracle/dsp_oracle.c) against the golden vectors produced by the reference
(tests/golden/make_golden.py).  This is what pins the oracle; it runs on the CPU.

Bit-exact (np.array_equal) wherever the reference's arithmetic is plain IEEE double
operations; ulp-level tolerances where the reference goes through code whose summation
order / libm differs between machines (np.convolve via BLAS, np.arctan2 via SVML).
"""
import numpy as np
import pytest

from conftest import golden, GOLDEN
import os


def test_rrc_filter(oracle):
    g = golden("g_rrc_taps")
    for i, (sps, span, al) in enumerate(g["params"]):
        h = oracle.rrc_filter(int(sps), int(span), float(al))
        ref = g[f"h{i}"]
        assert h.shape == ref.shape
        if h.size <= 128:
            assert np.array_equal(h, ref), (sps, span, al)
        else:   # 1001 taps: normalisation sum differs in the last bit
            assert np.max(np.abs(h - ref)) < 1e-17
    h = oracle.rrc_filter(4, 10, 0.35)
    assert h.size == 41 and abs(h[20] - 0.547856) < 1e-6 and abs(h.sum() - 1.995634) < 1e-6


def test_config1_chain(oracle):
    g = golden("g_config1")
    raw = np.fromfile(os.path.join(GOLDEN, "data", str(g["datafile"])), dtype=np.float32)
    x = oracle.read_iq_f32_words(raw)
    assert x.size == 41999 and x.dtype == np.complex64
    y = oracle.convolve_same(x, g["h"])
    scale = np.max(np.abs(g["conv"]))
    assert np.max(np.abs(y - g["conv"])) / scale < 1e-14
    assert oracle.np_std(g["conv"]) == float(g["std"])                  # np.std bit for bit
    yy = g["conv"].copy()
    oracle.lib().orc_scale_complex(yy.ctypes.data, yy.size, float(g["std"]))
    assert np.array_equal(yy, g["y"])                                    # y / std bit for bit
    rec, te, mu = oracle.gardner_timing_recovery(g["y"], 4, 0.007)
    assert np.array_equal(rec, g["rec"]) and np.array_equal(te, g["terr"]) and np.array_equal(mu, g["mu"])
    assert len(rec) == 10499 and mu[-1] == -0.05891794872812668
    rec2, te2, mu2 = oracle.gardner_timing_recovery(g["yd"], 4, 0.04)
    assert np.array_equal(rec2, g["rec2"]) and np.array_equal(mu2, g["mu2"])
    out, cf, freq, ef = oracle.fll_func(g["rec"], 0.03, 1)
    # np.arctan2 runs through SVML on AVX512 hosts: last-ulp differences in the error
    # (and beyond i ~ 3316 the loop is chaotic, see test_fll_chaos_beyond_horizon: only the
    # stable horizon is asserted; on the build container the whole run happens to agree)
    H = 3300
    assert np.max(np.abs(freq[:H] - g["fll_freq"][:H])) < 1e-17
    assert np.max(np.abs(out[:H] - g["fll_out"][:H])) < 1e-13
    assert float(g["fll_cf"]) == 0.044408380459126276
    assert np.array_equal(oracle.quadrants(out)[:H], g["dec"][:H])


def test_config2_chain(oracle):
    g = golden("g_config2")
    y = oracle.convolve_same(g["derot"], oracle.rrc_filter(4, 10, 0.35))
    assert np.max(np.abs(y - g["conv"])) / np.max(np.abs(g["conv"])) < 1e-14
    out, cf, freq, ef = oracle.fll_func(g["y"][0::4], 0.000001, 1)
    assert abs(cf - (-3.118829464601185e-07)) < 1e-21
    assert np.max(np.abs(freq - g["fllA_freq"])) < 1e-21
    rec, te, mu = oracle.gardner_timing_recovery(g["y"], 4, 0.007)
    assert len(rec) == 10470 and mu[-1] == 0.3285688473607562
    assert np.array_equal(rec, g["rec"]) and np.array_equal(mu, g["mu"])
    out, cf, freq, ef = oracle.fll_func(rec, 0.03, 1)
    assert np.max(np.abs(freq[:3300] - g["fllB_freq"][:3300])) < 1e-17
    assert np.array_equal(oracle.quadrants(out)[:3300], g["dec"][:3300])


def test_gardner_ted_hw1(oracle):
    g = golden("g_hw1_gardner")
    y, e, m = oracle.gardner_ted(g["baseband"], 2)
    assert len(y) == 49998
    assert np.array_equal(y, g["y_f64"]) and np.array_equal(e, g["err_f64"]) and np.array_equal(m, g["mu_f64"])
    y2, e2, m2 = oracle.gardner_ted(g["baseband"][:12000], 2, 0.005)
    assert np.array_equal(y2, g["y2_f64"]) and np.array_equal(m2, g["mu2_f64"])
    # the reference's own script run (complex64 input => float32 arithmetic under NEP 50)
    assert np.max(np.abs(m - g["mu"])) / 2 < 1e-3


def test_gardner2_demo(oracle):
    g = golden("g_gardner2_demo")
    rec, te, mu = oracle.gardner_timing_recovery(g["x"], 8, 0.07, 0.0)
    assert np.array_equal(rec, g["rec"]) and np.array_equal(te, g["terr"]) and np.array_equal(mu, g["mu"])


def test_lagrange(oracle):
    g = golden("g_lagrange")
    clean = g["clean"]
    pos = np.arange(len(clean), dtype=float) + 0.35
    mis = oracle.lagrange_resample(clean, pos, 6)
    assert np.max(np.abs(mis - g["mis"])) < 1e-15
    cor = oracle.lagrange_resample(g["mis"], np.arange(len(clean), dtype=float) - 0.35, 6)
    assert np.max(np.abs(cor - g["cor"])) < 1e-15
    arb = oracle.lagrange_resample(clean, g["pos"], 4)
    assert np.max(np.abs(arb - g["arb"])) < 1e-15
    assert np.array_equal(arb[:4], g["arb"][:4])          # exact-node shortcut


def test_rrc_cascade_apply_filter(oracle):
    """hw1/test_rrc_cascade.py: full convolution cropped by len(h)//2 with the 1002-tap filter;
    the printed pass criterion is max symbol error < 0.02 after RRC x RRC."""
    g = golden("g_rrc_cascade")
    h = g["taps"]
    x = g["impulses"]
    full = np.convolve(x, h, mode="full")
    d = len(h) // 2
    assert np.max(np.abs(full[d:d + len(x)] - g["after1"])) < 1e-13
    rec = g["after2"][::100][:40]
    assert np.max(np.abs(rec[6:34] - g["symbols"][6:34])) < 0.02     # away from the burst edges


def test_synth_bursts_chain(oracle):
    g = golden("g_synth_bursts")
    h = oracle.rrc_filter(4, 10, 0.35)
    for b in range(32):
        y = oracle.filter_samples(g["x"][b], h)
        o = oracle.sync_chain_c64(y.astype(np.complex64))
        n = o["n"]
        assert n == g["n_recB"][b]
        assert np.array_equal(o["dec"], g["decB"][b, :n])
        assert o["mu"][-1] == g["muB_last"][b]
        assert abs(o["freq"][-1] - g["freqB_last"][b]) < 1e-17
        if b < 6:
            assert np.max(np.abs(y - g["y"][b])) < 1e-14
            assert np.array_equal(o["rec"], g["recB"][b, :n])
            assert np.array_equal(o["mu"], g["muB"][b, :n - 1])
            assert np.max(np.abs(o["freq"] - g["fllB_freq"][b, :n])) < 1e-17


def test_edges(oracle):
    g = golden("g_edges")
    rng = np.random.default_rng(99)
    z = rng.standard_normal(64) + 1j * rng.standard_normal(64)
    for L in (1, 3, 4, 7, 8, 9, 12, 13):
        rec, e, m = oracle.gardner_timing_recovery(z[:L], 4, 0.05)
        assert np.array_equal(rec, g[f"g2_len{L}_rec"]) and np.array_equal(m, g[f"g2_len{L}_mu"])
        y, e, m = oracle.gardner_ted(z[:L], 4)
        assert np.array_equal(y, g[f"ted_len{L}_y"]) and np.array_equal(m, g[f"ted_len{L}_mu"])
    with pytest.raises(IndexError):
        oracle.gardner_timing_recovery(np.zeros(0, complex), 4)
    for nm in ("pos", "mid", "negmid"):
        rec, e, m = oracle.gardner_timing_recovery(g["wrap_x"], 4, float(g[f"g2_wrap_{nm}_alpha"]))
        assert np.array_equal(rec, g[f"g2_wrap_{nm}_rec"]) and np.array_equal(m, g[f"g2_wrap_{nm}_mu"])
    assert set(g["g2_wrap_nonterminating"]) == {"neg", "big"}
    rec, e, m = oracle.gardner_timing_recovery(g["wrap_x"], 4, 0.007, 3.7)
    assert np.array_equal(rec, g["g2_mu0_rec"]) and np.array_equal(m, g["g2_mu0_mu"])
    y, e, m = oracle.gardner_ted(g["wrap_x"], 4, 0.8)
    assert np.array_equal(y, g["ted_clip_y"]) and np.array_equal(m, g["ted_clip_mu"])
    out, cf, freq, ef = oracle.fll_func(g["fll_axis_x"], 50.0, 1)
    assert np.max(np.abs(freq - g["fll_axis_freq"])) < 1e-15
    out, cf, freq, ef = oracle.fll_func(g["wrap_x"][::4], 0.5, 0.707)
    assert np.max(np.abs(freq[:150] - g["fll_y707_freq"][:150])) < 1e-15
    out, cf, freq, ef = oracle.fll_func(np.zeros(0, complex), 0.03, 1)
    assert cf == 0.0 and float(g["fll_empty_cf"]) == 0.0


def test_read_iq_int16(oracle):
    g = golden("g_read_iq")
    for k in ("odd", "zeros", "minmax", "rand"):
        assert np.array_equal(oracle.read_iq_int16_words(g[k + "_raw"], True), g[k + "_norm"]), k
        assert np.array_equal(oracle.read_iq_int16_words(g[k + "_raw"], False), g[k + "_nonorm"]), k
    assert bool(g["short_raises"])
    with pytest.raises(ValueError):
        oracle.read_iq_int16_words(np.array([5], np.int16))
    h = golden("g_hw1_gardner")
    raw = np.fromfile(os.path.join(GOLDEN, "data", "iq_int16.pcm"), dtype=np.int16)
    assert np.array_equal(oracle.read_iq_int16_words(raw, True), h["iq"])
    assert np.array_equal(oracle.read_iq_int16_words(raw, False), h["iq_raw"])


def test_fll_chaos_beyond_horizon(oracle):
    """Documents why FLL parity is asserted on a bounded horizon for Bn = 0.03: the loop gain
    2*pi*kp*i exceeds 2 at i = 1/(pi*kp) ~ 3316; beyond it a 1e-12 relative input perturbation
    changes the oracle's own decisions wholesale, before it none."""
    g = golden("g_config1")
    rec = g["rec"]
    rng = np.random.default_rng(0)
    a = oracle.fll_func(rec, 0.03, 1)
    b = oracle.fll_func(rec * (1 + 1e-12 * rng.standard_normal(rec.size)), 0.03, 1)
    qa, qb = oracle.quadrants(a[0]), oracle.quadrants(b[0])
    assert np.array_equal(qa[:3300], qb[:3300])
    assert np.max(np.abs(a[2][:3300] - b[2][:3300])) < 1e-12
    assert np.sum(qa[5000:] != qb[5000:]) > 1000


def test_py_loops_match_reference_vectors():
    """oracle/py_loops.py (the interpreter-bound speed proxies bench.py times) against the
    reference's own outputs on a golden synthetic burst: Gardner bit-exact, FLL to 1e-15."""
    from oracle import py_loops as P
    g = golden("g_synth_bursts")
    y = g["y"][0]
    n = int(g["n_rec"][0])
    for gard, fll in ((P.gardner_np, P.fll_np), (P.gardner, P.fll)):
        rec, err, mu = gard(y if gard is P.gardner_np else y.tolist(), 4, 0.007, 0.0)
        out, f, fr, ef = fll(rec, 0.03, 1)              # lists stay lists for the plain-Python pair
        rec, mu = np.asarray(rec), np.asarray(mu)
        assert len(rec) == n
        assert np.array_equal(rec, g["rec"][0][:n]) and np.array_equal(mu, g["mu"][0][:n - 1])
        assert np.abs(np.asarray(out) - g["fll_out"][0][:n]).max() < 1e-14
        assert np.abs(np.asarray(fr) - g["fll_freq"][0][:n]).max() < 1e-15
