"""Coarse frequency estimators, derotation and the burst cutter:
diff_method.py:267-298, worker.py:284-304, estimate_freq.py:4-58.

Golden values come from the reference itself (tests/golden/make_golden.py -> g_config2.npz f_rel /
derot, make_golden_cfo.py -> g_cfo.npz, make_golden_preamble.py -> g_worker.npz).  The device sums
are tree-ordered where numpy's are sequential / pairwise and the line fit is the closed form of
np.polyfit's least squares: estimates agree to 1e-12 cycles/sample, derotated samples to 1e-12.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
F_TOL = 1e-12


def _load(name):
    return np.load(os.path.join(G, name))


# ---------------------------------------------------------------- oracle vs golden (CPU)
def test_oracle_preamble_cfo_config2():
    g2, gd = _load("g_config2.npz"), _load("g_diffcorr.npz")
    pre, al = gd["pre"], g2["aligned"]
    f = O.preamble_cfo(al[::4][:len(pre)], pre, 4)
    assert np.abs(f - g2["f_rel"]).max() < 1e-15
    assert np.abs(O.derotate(al, g2["f_rel"][1]) - g2["derot"]).max() < 1e-14


def test_oracle_fourth_power():
    gc = _load("g_cfo.npz")
    got = np.array([O.estimate_initial_frequency(gc["fp_x"], int(n)) for n in gc["fp_n"]])
    assert np.abs(got - gc["fp_f"]).max() < 1e-15
    got = np.array([O.estimate_initial_frequency(r, 300) for r in gc["fp_rows"]])
    assert np.abs(got - gc["fp_rows_f"]).max() < 1e-15


def test_oracle_unwrap_matches_numpy():
    rng = np.random.default_rng(3)
    ph = np.angle(np.exp(1j * np.cumsum(rng.uniform(-3.5, 3.5, 3000))))
    assert np.array_equal(O.unwrap(ph), np.unwrap(ph))
    ph = np.array([0.0, np.pi, 0.0, -np.pi, 2 * np.pi, -2 * np.pi, 0.5])      # exact +-pi jumps
    assert np.array_equal(O.unwrap(ph), np.unwrap(ph))


def test_oracle_worker_f_rel():
    gw, gc = _load("g_worker.npz"), _load("g_cfo.npz")
    sig, pre = gw["w_samples"], gw["w_pre"]
    offs = [int(p) // 2 for p in gw["w_peaks_n"]]
    got = np.array([O.preamble_cfo(sig[o:o + 1600:4], pre, 4)[0] for o in offs])
    assert np.abs(got - gc["w_frel_n"]).max() < 1e-15


# ---------------------------------------------------------------- device vs golden / oracle
@pytest.mark.gpu
def test_preamble_cfo_config2():
    import dsp_amd
    g2, gd = _load("g_config2.npz"), _load("g_diffcorr.npz")
    f = dsp_amd.preamble_cfo(g2["aligned"], gd["pre"], 4)
    assert f.shape == (3,)
    assert np.abs(f - g2["f_rel"]).max() < F_TOL


@pytest.mark.gpu
def test_preamble_cfo_batch_offsets_and_c64():
    import dsp_amd
    rng = np.random.default_rng(5)
    L, sps, n = 150, 4, 40000                      # three wave-groups of preamble symbols, ragged tail
    pre = np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, L)))
    x = 0.2 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    offs = np.sort(rng.choice(n - L * sps, 37, replace=False)).astype(np.int64)
    offs[-1] = n - ((L - 1) * sps + 1)             # the last burst ends exactly at the end of the signal
    fs = rng.uniform(-0.02, 0.02, offs.size)       # up to +-0.08 cycles/symbol: the phases wrap many times
    for o, f in zip(offs, fs):
        x[o:o + L * sps:sps] += pre * np.exp(1j * (2 * np.pi * f * sps * np.arange(L) + 1.0))
    got = dsp_amd.preamble_cfo(x, pre, sps, offs)
    want = np.array([O.preamble_cfo(x[o::sps][:L], pre, sps) for o in offs])
    assert got.shape == (offs.size, 3)
    assert np.abs(got - want).max() < F_TOL
    # complex64 samples: float64 arithmetic on the upcast samples
    x32 = x.astype(np.complex64)
    got = dsp_amd.preamble_cfo(x32, pre, sps, offs)
    want = np.array([O.preamble_cfo(x32[o::sps][:L].astype(np.complex128), pre, sps) for o in offs])
    assert np.abs(got - want).max() < F_TOL
    with pytest.raises(Exception):
        dsp_amd.preamble_cfo(x, pre, sps, [n - (L - 1) * sps])        # one sample short
    with pytest.raises(ValueError):
        dsp_amd.preamble_cfo(x, pre[:1], sps)


@pytest.mark.gpu
def test_fourth_power_golden():
    import dsp_amd
    gc = _load("g_cfo.npz")
    got = np.array([dsp_amd.estimate_initial_frequency(gc["fp_x"], int(n)) for n in gc["fp_n"]])
    assert np.abs(got - gc["fp_f"]).max() < F_TOL
    assert got[4] == 0.0 and got[5] == 0.0                       # fewer than two samples (:33-34)
    got = dsp_amd.estimate_initial_frequency_batch(gc["fp_rows"], 300)
    assert np.abs(got - gc["fp_rows_f"]).max() < F_TOL
    # ragged rows
    lens = np.array([300, 1, 0, 2, 299, 150], np.int32)
    got = dsp_amd.estimate_initial_frequency_batch(gc["fp_rows"], 300, lens=lens)
    want = np.array([O.estimate_initial_frequency(r[:l], 300) for r, l in zip(gc["fp_rows"], lens)])
    assert np.abs(got - want).max() < F_TOL
    assert dsp_amd.estimate_initial_frequency(np.zeros(0, np.complex128), 10) == 0.0


@pytest.mark.gpu
def test_derotate_config2():
    import dsp_amd
    g2 = _load("g_config2.npz")
    d = dsp_amd.derotate(g2["aligned"], float(g2["f_rel"][1]))
    assert d.dtype == np.complex128
    assert np.abs(d - g2["derot"]).max() < 1e-12
    d = dsp_amd.derotate(g2["aligned"][:5000], 3.1e-3, offset=123456)
    assert np.abs(d - O.derotate(g2["aligned"][:5000], 3.1e-3, 123456)).max() < 1e-12


@pytest.mark.gpu
def test_rx_bursts_matches_worker_golden():
    """dsp_rx_bursts on the seeded worker.Generation signal: f_rel per burst as the reference printed
    it, FLL output as the reference returned it (tolerances of tests/test_preamble.py)."""
    import dsp_amd
    gw, gc = _load("g_worker.npz"), _load("g_cfo.npz")
    sig, pre = gw["w_samples"], gw["w_pre"]
    offs = [int(p) // 2 for p in gw["w_peaks_n"]]
    h = dsp_amd.rrc_filter(4, 10, 0.35)
    out = dsp_amd.rx_bursts(sig, offs, 1600, pre, h, 4, 0.007, 0.0, 0.03, 1, cfo_method=1, outputs=("sig", "dec"))
    assert np.abs(out["f_rel"] - gc["w_frel_n"]).max() < F_TOL
    for b in range(len(offs)):
        want = gw[f"w_found_n{b}"]
        assert int(out["n_out"][b]) == len(want)
        assert np.abs(out["sig"][b, :len(want)] - want).max() < 1e-6
        assert np.array_equal(out["dec"][b, :len(want)], O.quadrants(want))
    # methods 2 and 3, and a given f_rel, against the oracle chain on the same cuts
    for method in (2, 3, 0):
        kw = dict(f_rel=[1e-3, -2e-3, 0.0]) if method == 0 else dict(cfo_method=method)
        out = dsp_amd.rx_bursts(sig, offs, 1600, pre, h, 4, 0.007, 0.0, 0.03, 1, outputs=("sig", "dec"), **kw)
        for b, o in enumerate(offs):
            cut = sig[o:o + 1600]
            f = kw["f_rel"][b] if method == 0 else O.preamble_cfo(cut[::4][:len(pre)], pre, 4)[method - 1]
            assert abs(out["f_rel"][b] - f) < F_TOL
            y = O.filter_samples(O.derotate(cut, f, o), h)
            rec, _, _ = O.gardner_timing_recovery(y, 4, 0.007, 0.0)
            s, _, _, _ = O.fll_func(rec, 0.03, 1)
            assert int(out["n_out"][b]) == len(s)
            assert np.abs(out["sig"][b, :len(s)] - s).max() < 1e-6
            assert np.array_equal(out["dec"][b, :len(s)], O.quadrants(s))
    with pytest.raises(Exception):
        dsp_amd.rx_bursts(sig, [len(sig) - 1599], 1600, pre, h)


@pytest.mark.gpu
def test_receiver_records_f_rel():
    from dsp_amd.receiver import Processing
    gw, gc = _load("g_worker.npz"), _load("g_cfo.npz")
    for tag, interp in (("n", False), ("i", True)):
        pr = Processing(signal=gw["w_samples"], preamble=gw["w_pre"], packet_symb=gw["w_symbols"], sps=4,
                        constellation="qpsk", interp=interp)
        peaks, found = pr.process()
        assert np.array_equal(peaks, gw[f"w_peaks_{tag}"])
        assert np.abs(pr.f_rel - gc[f"w_frel_{tag}"]).max() < 1e-10
        assert len(found) == int(gw[f"w_nfound_{tag}"])


@pytest.mark.gpu
def test_front_end_argument_errors():
    """Bad arguments come back as exceptions with a message, never as a fault on the device."""
    import dsp_amd
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4000) + 1j * rng.standard_normal(4000))
    pre = np.exp(1j * np.pi / 2 * rng.integers(0, 4, 32))
    h = dsp_amd.rrc_filter(4, 10, 0.35)
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.rx_bursts(x, [-1], 400, pre, h)                       # cut starts before the signal
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.rx_bursts(x, [3700], 400, pre, h)                     # ... ends after it
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.rx_bursts(x, [0], 40, pre, h)                         # packet shorter than the filter
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.rx_bursts(x, [0], 100, pre, h)                        # ... than the preamble
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.rx_bursts(x, [0], 400, pre, h, cfo_method=7)
    out = dsp_amd.rx_bursts(x, [], 400, pre, h)                       # no bursts: empty result
    assert out["n_out"].size == 0 and out["f_rel"].size == 0
    assert dsp_amd.derotate(np.zeros(0, np.complex128), 0.1).size == 0
    assert dsp_amd.estimate_initial_frequency_batch(np.zeros((3, 0), np.complex128), 5).tolist() == [0.0, 0.0, 0.0]
    assert dsp_amd.preamble_cfo(x, pre, 4, []).shape == (0, 3)
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.preamble_cfo(x, pre, 4, [4000 - 31 * 4])              # last preamble symbol one sample out
    o = dsp_amd.fll_slicer_batch(np.zeros((0, 5), np.complex128), 0.01, 1, "8psk")
    assert o["sig"].shape == (0, 5)
