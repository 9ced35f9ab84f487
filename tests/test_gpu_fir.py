"""GPU parity tests of the loaders, the matched filter + np.std normalisation and the full
receive chain from raw samples (through the C ABI)."""
import os

import numpy as np
import pytest

from conftest import golden, GOLDEN

pytestmark = pytest.mark.gpu


def test_read_iq_int16_bit_exact(ctx):
    import dsp_amd
    g = golden("g_read_iq")
    for k in ("odd", "zeros", "minmax", "rand"):
        assert np.array_equal(dsp_amd.read_iq_int16_words(g[k + "_raw"], True), g[k + "_norm"]), k
        assert np.array_equal(dsp_amd.read_iq_int16_words(g[k + "_raw"], False), g[k + "_nonorm"]), k
    with pytest.raises(ValueError):
        dsp_amd.read_iq_int16_words(np.array([5], np.int16))
    h = golden("g_hw1_gardner")
    p = os.path.join(GOLDEN, "data", "iq_int16.pcm")
    iq = dsp_amd.read_iq_int16(p)
    assert iq.dtype == np.complex64 and np.array_equal(iq, h["iq"])            # 100 000 samples
    assert np.array_equal(dsp_amd.read_iq_int16(p, normalize=False), h["iq_raw"])
    x = dsp_amd.read_iq_float32(os.path.join(GOLDEN, "data", "qpsk_high_snr_sps_4_float32.pcm"))
    assert x.size == 41999 and x.dtype == np.complex64 and abs(np.std(x) - 832.864) < 1e-2


def test_filter_samples_config1(ctx):
    """worker.py:173-178 on config 1: complex128 path vs the reference's np.convolve + np.std."""
    import dsp_amd
    g = golden("g_config1")
    x = dsp_amd.read_iq_float32(os.path.join(GOLDEN, "data", str(g["datafile"])))
    h = dsp_amd.rrc_filter(4, 10, 0.35)
    y, std = dsp_amd.fir_batch(x.astype(np.complex128)[None, :], h, normalize=False, want_std=True)
    scale = np.max(np.abs(g["conv"]))
    assert np.max(np.abs(y[0] - g["conv"])) / scale < 1e-14
    assert abs(std[0] - float(g["std"])) / float(g["std"]) < 1e-14
    yn = dsp_amd.filter_samples(x)
    assert yn.dtype == np.complex128
    assert np.max(np.abs(yn - g["y"])) < 1e-13
    # complex64 in -> complex64 out: float64 accumulation, one rounding
    y32 = dsp_amd.fir_batch(x[None, :], h, normalize=True)[0]
    assert y32.dtype == np.complex64
    ref32 = g["y"].astype(np.complex64)
    ulp = np.spacing(np.maximum(np.abs(ref32.real), np.abs(ref32.imag)))
    assert np.max(np.abs(y32 - ref32) / ulp) <= 1.0
    assert np.mean(y32 == ref32) > 0.999


def test_fir_batch_ragged_vs_oracle(ctx, oracle):
    import dsp_amd
    rng = np.random.default_rng(3)
    B, N = 70, 1000
    X = (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N)) + (0.3 - 0.2j)).astype(np.complex64)
    lens = rng.integers(81, N + 1, B).astype(np.int32)
    lens[:3] = [81, 82, N]
    for taps in (dsp_amd.rrc_filter(4, 10, 0.35), dsp_amd.rrc_filter(8, 10, 0.35), rng.standard_normal(40)):
        Y, std = dsp_amd.fir_batch(X, taps, normalize=True, lens=lens, want_std=True)
        for b in range(B):
            L = int(lens[b])
            yo = oracle.convolve_same(X[b, :L], taps)
            so = oracle.np_std(yo)
            assert abs(std[b] - so) / so < 1e-13
            ref = (yo * (1.0 / so)).astype(np.complex64)
            ulp = np.spacing(np.maximum(np.abs(ref.real), np.abs(ref.imag)))
            assert np.max(np.abs(Y[b, :L] - ref) / ulp) <= 1.0
            assert np.all(Y[b, L:] == 0)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_fir_batch_full_tiles_vs_oracle(ctx, oracle, dtype):
    """Full tiles of equal-length bursts: interior chunks take the staging path without bounds logic,
    the first and last chunks (and the 6 bursts of the third tile) the general one; a DC offset 50
    times the spread checks the one-pass statistics."""
    import dsp_amd
    rng = np.random.default_rng(11)
    B, N = 134, 1500
    X = (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N))).astype(dtype)
    X[5] += 50.0 - 20.0j
    for taps in (dsp_amd.rrc_filter(4, 10, 0.35), rng.standard_normal(17), rng.standard_normal(75)):
        Y, std = dsp_amd.fir_batch(X, taps, normalize=True, want_std=True)
        for b in list(range(0, B, 7)) + [5, 63, 64, 127, 128, B - 1]:
            yo = oracle.convolve_same(X[b], taps)
            so = oracle.np_std(yo)
            assert abs(std[b] - so) / so < 1e-13, b
            ref = (yo * (1.0 / so)).astype(dtype)
            if dtype == np.complex64:
                ulp = np.spacing(np.maximum(np.abs(ref.real), np.abs(ref.imag)))
                assert np.max(np.abs(Y[b] - ref) / ulp) <= 1.0, b
            else:
                assert np.max(np.abs(Y[b] - ref)) < 1e-12 * np.max(np.abs(ref)), b


def test_rrc_cascade_long_filter(ctx):
    """hw1/test_rrc_cascade.py: 1002-tap filter, full convolution cropped by len(h)//2 (the long
    filter takes the direct-form kernel)."""
    import dsp_amd
    g = golden("g_rrc_cascade")
    h = g["taps"]
    a1 = dsp_amd.fir_batch(g["impulses"][None, :], h, normalize=False, start=len(h) // 2)[0]
    assert np.max(np.abs(a1 - g["after1"])) < 1e-13
    a2 = dsp_amd.fir_batch(a1[None, :], h, normalize=False, start=len(h) // 2)[0]
    assert np.max(np.abs(a2 - g["after2"])) < 1e-13
    rec = a2[::100][:40]
    assert np.max(np.abs(rec[6:34] - g["symbols"][6:34])) < 0.02          # the script's criterion


def test_rx_chain_from_raw_bursts(ctx, oracle):
    """Raw synthetic bursts -> RRC -> /std -> Gardner -> FLL: decisions bit-exact vs the reference
    (golden chain A, all float64) and vs the oracle run with the device's own rounding points."""
    import dsp_amd
    g = golden("g_synth_bursts")
    h = dsp_amd.rrc_filter(4, 10, 0.35)
    X = g["x"]
    r = dsp_amd.rx_chain_batch(X, h, 4, 0.007, 0.0, 0.03, 1)
    assert np.max(np.abs(r["std"] - g["std"]) / g["std"]) < 1e-13
    assert np.array_equal(r["n_out"], g["n_rec"])
    assert np.array_equal(r["dec"][:, :2048], g["dec"])                  # reference decisions, 32 bursts
    for b in range(6):
        n = int(g["n_rec"][b])
        assert np.max(np.abs(r["mu"][b, :n - 1] - g["mu"][b, :n - 1])) < 1e-5      # << 1e-3 symbol
        assert np.max(np.abs(r["freq"][b, :n] - g["fll_freq"][b, :n])) < 1e-7      # << 1e-4 rad/sample
    # oracle with the same rounding points: y32 = fl32(conv), x' = y32 * (1/std)
    for b in range(8):
        yo = oracle.convolve_same(X[b], h)
        so = oracle.np_std(yo)
        o = oracle.sync_chain_c64(yo.astype(np.complex64), scale=1.0 / so)
        n = o["n"]
        assert np.array_equal(r["dec"][b, :n], o["dec"])
        assert np.max(np.abs(r["mu"][b, :n - 1] - o["mu"])) < 1e-6
    # float64 all the way (complex128 in)
    r64 = dsp_amd.rx_chain_batch(X[:8].astype(np.complex128), h, 4, 0.007, 0.0, 0.03, 1)
    for b in range(6):
        n = int(g["n_rec"][b])
        assert np.array_equal(r64["dec"][b, :n], g["dec"][b, :n])
        assert np.max(np.abs(r64["mu"][b, :n - 1] - g["mu"][b, :n - 1])) < 1e-11
        assert np.max(np.abs(r64["freq"][b, :n] - g["fll_freq"][b, :n])) < 1e-12


def test_rx_chain_int16_capture(ctx, oracle):
    """int16 capture cut into bursts: peak normalisation (one peak for the block) -> RRC -> chain.
    The oracle is composed with the device's rounding points -- samples scaled by 1/peak in float32
    (read_iq_int16), y32 = fl32(conv in float64), x' = float64(y32) * (1/std) -- and everything
    downstream is then held to the bits: symbol counts, mu (float64 trajectory rounded to the
    float32 output), and the decisions of all 2047 symbols (inside the FLL horizon of 3316)."""
    import dsp_amd
    raw = np.fromfile(os.path.join(GOLDEN, "data", "iq_int16.pcm"), dtype=np.int16)
    B, N = 12, 4096
    blk = raw[:B * N * 2].reshape(B, N * 2)
    h = dsp_amd.rrc_filter(2, 10, 0.35)
    r = dsp_amd.rx_chain_batch(blk, h, 2, 0.002, 0.0, 0.03, 1, variant=dsp_amd.DSP_GARDNER_TED)
    iq = oracle.read_iq_int16_words(raw[:B * N * 2], True).reshape(B, N)
    for b in range(B):
        yo = oracle.convolve_same(iq[b], h)
        so = oracle.np_std(yo)
        assert abs(r["std"][b] - so) / so < 1e-13
        # 1/std as the device forms it (Chan-merged statistics differ from numpy's pairwise sum in
        # the last bits, and the last bit of the scale moves every product)
        x = (yo.astype(np.complex64).astype(np.complex128)) * (1.0 / r["std"][b])
        y, e, m = oracle.gardner_ted(x, 2, 0.002)
        assert r["n_out"][b] == len(y)
        assert np.array_equal(r["mu"][b, :len(m)], m.astype(np.float32)), b
        assert np.array_equal(r["rec"][b, :len(y)], y.astype(np.complex64)), b
        out = oracle.fll_func(y, 0.03, 1)
        q = oracle.quadrants(out[0])
        assert np.array_equal(r["dec"][b, :len(q)], q), b
        assert np.max(np.abs(r["freq"][b, :len(q)] - out[2])) < 1e-9
