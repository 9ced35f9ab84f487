"""The constellation switch of worker.Processing.fll_func / decision_direct (worker.py:18-170;
SURVEY.md 8f rank 4) against golden vectors produced by the reference itself
(tests/golden/make_golden_slicers.py): loop outputs to 1e-9 (700 symbols, inside the loop's stable
horizon), decided symbols exactly, match percentages exactly."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "g_slicers.npz"))
CONS = ("bpsk", "qpsk", "8psk", "16psk", "16apsk")


def _oracle_points(c, dec, sl):
    if c == "bpsk":
        return np.array([1, -1, 0], np.complex64)[dec]
    if c == "qpsk":
        return ((np.sign(sl.real) + 1j * np.sign(sl.imag)) / np.sqrt(2)).astype(np.complex64)
    return O.constellation_points(c)[dec].astype(np.complex64)


@pytest.mark.parametrize("c", CONS)
def test_oracle_matches_reference(c):
    sl, cf, fr, ef, dec = O.fll_func_constellation(G[c + "_x"], float(G[c + "_bn"]), 1, c)
    assert np.abs(sl - G[c + "_sig"]).max() < 1e-14
    assert abs(cf - float(G[c + "_cf"])) < 1e-16
    assert np.abs(fr - G[c + "_freq"]).max() < 1e-16 and np.abs(ef - G[c + "_ef"]).max() < 1e-16
    assert np.array_equal(_oracle_points(c, dec, sl), G[c + "_dec"])


@pytest.mark.gpu
@pytest.mark.parametrize("c", CONS)
def test_fll_func_constellation_golden(c):
    import dsp_amd
    from dsp_amd.receiver import Processing
    pr = Processing(packet_symb=G[c + "_sym"].astype(np.complex64), sps=4, constellation=c)
    sl, cf, fr, ef = pr.fll_func(G[c + "_x"], Bn=float(G[c + "_bn"]), y=1)
    assert sl.dtype == np.complex128 and isinstance(cf, float)
    assert np.abs(sl - G[c + "_sig"]).max() < 1e-9
    assert abs(cf - float(G[c + "_cf"])) < 1e-12
    assert np.abs(fr - G[c + "_freq"]).max() < 1e-12 and np.abs(ef - G[c + "_ef"]).max() < 1e-12
    dec = pr.decision_direct(G[c + "_sig"])
    assert dec.dtype == np.complex64
    assert np.array_equal(dec, G[c + "_dec"])
    rates = np.array([r for _, r in pr.match_rates[1:]])
    assert np.array_equal(rates, G[c + "_rates"])
    # decision indices against the restated slicer
    _, idx = dsp_amd.decide_symbols(G[c + "_sig"], c)
    want = O.fll_func_constellation(G[c + "_sig"], 0.0, 1, c)[4]
    assert np.array_equal(idx, want)


@pytest.mark.gpu
def test_fll_slicer_batch_ragged_and_c64():
    import dsp_amd
    rng = np.random.default_rng(8)
    B, M = 70, 300                                           # two tiles, the second partly filled
    pts = O.constellation_points("16apsk")
    x = pts[rng.integers(0, 16, (B, M))] * np.exp(1j * 2 * np.pi * rng.uniform(-3e-4, 3e-4, (B, 1)) * np.arange(M))
    x = x + 0.01 * (rng.standard_normal((B, M)) + 1j * rng.standard_normal((B, M)))
    lens = rng.integers(0, M + 1, B).astype(np.int32)
    lens[:3] = (0, 1, M)
    o = dsp_amd.fll_slicer_batch(x, 0.02, 1, "16apsk", lens=lens)
    for b in (0, 1, 2, 5, 63, 64, 69):
        sl, cf, fr, ef, dec = O.fll_func_constellation(x[b, :lens[b]], 0.02, 1, "16apsk")
        n = lens[b]
        assert np.abs(o["sig"][b, :n] - sl).max(initial=0) < 1e-9
        assert np.abs(o["freq"][b, :n] - fr).max(initial=0) < 1e-12
        assert np.array_equal(o["dec"][b, :n], dec)
        assert abs(o["curr_freq"][b] - cf) < 1e-12
        assert not o["sig"][b, n:].any() and not o["dec"][b, n:].any()
    # complex64 in -> complex64 / float32 out, float64 arithmetic inside
    o32 = dsp_amd.fll_slicer_batch(x.astype(np.complex64), 0.02, 1, "16apsk", lens=lens, outputs=("sig", "dec"))
    assert o32["sig"].dtype == np.complex64
    b = 2
    sl, _, _, _, dec = O.fll_func_constellation(x[b].astype(np.complex64), 0.02, 1, "16apsk")
    assert np.abs(o32["sig"][b] - sl).max() < 1e-6
    assert np.mean(o32["dec"][b] == dec) > 0.99
    with pytest.raises(ValueError):
        dsp_amd.fll_slicer_batch(x, 0.02, 1, "64qam")


@pytest.mark.gpu
@pytest.mark.parametrize("c", ["8psk", "16psk", "16apsk"])
def test_nearest_point_ties_as_numpy(c):
    """The table slicers decide by squared distances and send only near-ties through numpy's own np.abs
    expression (nco_kernels.hip: nearest_point).  Samples ON the decision boundaries -- midpoints of every pair
    of points, the origin (every point of a PSK ring equally far), points themselves, the same scaled and
    nudged by an ulp or 1e-13 -- must decide as np.argmin(np.abs(points - sample)) does: the first of the
    nearest in the reference's order, whatever the last bits of the two np.abs values say."""
    import dsp_amd
    rng = np.random.default_rng(31)
    pts = O.constellation_points(c)
    mids = np.array([(p + q) / 2 for i, p in enumerate(pts) for q in pts[i + 1:]])
    base = np.concatenate([mids, [0j], pts, 0.5 * mids, 2.0 * mids, rng.standard_normal(64) + 1j * rng.standard_normal(64)])
    xs = [base]
    for eps in (1e-16, 1e-15, 1e-13, 1e-11, 1e-9):
        xs.append(base * (1 + eps))
        xs.append(base + eps * (rng.standard_normal(base.size) + 1j * rng.standard_normal(base.size)))
    nz = base[(base.real != 0) & (base.imag != 0)]           # (an ulp off ZERO is a subnormal: outside the loops' range of
    xs.append(np.nextafter(nz.real, 9.0) + 1j * nz.imag)     #  sample magnitudes, dsp_math.h div_fast)
    xs.append(nz.real + 1j * np.nextafter(nz.imag, -9.0))
    x = np.concatenate(xs)
    want = np.array([np.argmin(np.abs(pts - v)) for v in x], dtype=np.uint8)
    _, idx = dsp_amd.decide_symbols(x, c)
    assert np.array_equal(idx, want), np.flatnonzero(idx != want)[:10]
    # the same samples spread over a full tile (no padding lanes), one burst per lane
    n = x.size // 64 * 64
    o = dsp_amd.fll_slicer_batch(x[:n].reshape(64, -1), 0.0, 1, c, outputs=("dec",))
    assert np.array_equal(o["dec"].reshape(-1), want[:n])


@pytest.mark.gpu
def test_receiver_8psk_end_to_end():
    """worker.Processing.process with a table slicer (8psk) on a seeded worker.Generation signal:
    peaks as the reference found them, FLL outputs as it returned them."""
    from dsp_amd.receiver import Processing
    pr = Processing(signal=G["w8_samples"], preamble=G["w8_pre"], packet_symb=G["w8_symbols"], sps=4,
                    constellation="8psk", interp=False)
    peaks, found = pr.process()
    assert np.array_equal(peaks, G["w8_peaks"])
    assert len(found) == int(G["w8_nfound"]) and len(found) > 0
    for j, f in enumerate(found):
        want = G[f"w8_found{j}"]
        assert f.shape == want.shape
        assert np.abs(f - want).max() < 1e-6
    assert len(pr.decisions) == len(found) and pr.decisions[0].dtype == np.complex64
