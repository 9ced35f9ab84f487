"""a10: facade for the reference's class FrequencyLockedLoop.  PARITY UNPINNED -- fll.py is absent
from the reference; the device loop is checked against the CPU restatement of
the library's own definition, and on function: it must find and remove a carrier offset."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def qpsk_stream(n_sym, cfo, seed=3):
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 4, n_sym)
    sym = ((1 - 2 * (q & 1)) + 1j * (1 - 2 * (q >> 1))) / np.sqrt(2)
    up = np.zeros(n_sym * 4, complex)
    up[::4] = sym
    h = O.rrc_filter(4, 10, 0.35)
    x = np.convolve(np.convolve(up, h, "same"), h, "same")
    x = x / np.std(x)
    t = np.arange(x.size)
    x = x * np.exp(1j * (2 * np.pi * cfo * t + 0.3))
    return x + 0.03 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))


@pytest.mark.parametrize("method", ["cross_product", "atan2", "decision_directed"])
def test_fll_class_matches_own_definition(oracle, ctx, method):
    import dsp_amd
    x = qpsk_stream(1500, 2.0e-3)
    fll = dsp_amd.FrequencyLockedLoop(detector_method=method, Kp=0.001, Ki=0.00005, freq_limit=0.05, ctx=ctx)
    y, fe = fll.process_signal(x)
    ry, rfe, rerr, rfr, rph = oracle.fll_class(x, method, 0.001, 0.00005, 0.05)
    assert y.shape == x.shape and set(fll.history) == {"errors", "freq_estimates", "phases"}
    assert np.max(np.abs(fll.history["freq_estimates"] - rfr)) < 1e-9
    assert np.max(np.abs(fll.history["errors"] - rerr)) < 1e-7
    assert np.max(np.abs(y - ry)) < 1e-6
    assert abs(fe - rfe) < 1e-9


def test_fll_class_removes_cfo(ctx):
    import dsp_amd
    cfo = 1.0e-3
    x = qpsk_stream(6000, cfo)
    # symbol-spaced samples (decision-directed detector wants symbol instants)
    xs = x[::4]
    fll = dsp_amd.FrequencyLockedLoop("decision_directed", Kp=0.02, Ki=0.0005, freq_limit=0.05, ctx=ctx)
    y, fe = fll.process_signal(xs)
    assert abs(fe - 4 * cfo) < 2e-4                        # cycles per (symbol-spaced) sample
    tail = y[-2000:]
    assert np.mean(np.abs(np.abs(tail.real) - np.abs(tail.imag))) < 0.15   # constellation on the diagonals
    fll32 = dsp_amd.FrequencyLockedLoop("atan2", ctx=ctx)
    y32, fe32 = fll32.process_signal(xs.astype(np.complex64))
    assert y32.dtype == np.complex64 and fll32.history["errors"].dtype == np.float32
