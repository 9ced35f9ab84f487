"""Differential-correlation preamble search: find_preamble_offset,
diff_method.py:92-162.  The golden vectors come from the reference itself
(tests/golden/make_golden_preamble.py): config 2's capture (burst at symbol 518, phase 1) and a
synthetic complex64 case with a dead stretch (windows whose std is 0)."""
import os

import numpy as np
import pytest

from conftest import golden, GOLDEN


def test_oracle_find_preamble_offset(oracle):
    g = golden("g_diffcorr")
    off, aligned, conv, cmax, ph = oracle.find_preamble_offset(g["sig"], g["pre"], 4)
    assert (off, ph) == (518, 1) == (int(g["offset"]), int(g["phase"]))
    assert len(aligned) == int(g["n_aligned"])
    for p in range(4):
        assert np.array_equal(conv[p], g[f"conv{p}"])
    off, aligned, conv, cmax, ph = oracle.find_preamble_offset(g["syn_x"], g["syn_pre"], 2)
    assert (off, ph) == (int(g["syn_offset"]), int(g["syn_phase"]))
    assert np.array_equal(conv[0], g["syn_conv0"]) and np.array_equal(conv[1], g["syn_conv1"])


@pytest.mark.gpu
def test_gpu_find_preamble_offset_config2(ctx):
    import dsp_amd
    g = golden("g_diffcorr")
    off, aligned, conv, cmax, ph = dsp_amd.find_preamble_offset(g["sig"], g["pre"], 4, ctx=ctx)
    assert (off, ph) == (518, 1)
    assert len(aligned) == int(g["n_aligned"]) and np.array_equal(aligned, g["sig"][518 * 4 + 1:])
    for p in range(4):
        ref = g[f"conv{p}"]
        assert conv[p].shape == ref.shape and conv[p].dtype == np.complex64
        assert np.max(np.abs(conv[p] - ref)) <= 2e-6 * np.max(np.abs(ref))     # float64 sums, complex64 store
        assert int(cmax[p][0]) == int(g["conv_max"][p, 0])
        assert abs(cmax[p][1] - g["conv_max"][p, 1]) < 1e-5 * g["conv_max"][p, 1]
    bp, corr = dsp_amd.find_optimal_sampling_phase(aligned, g["pre"], 4)
    assert int(bp) == int(g["best_phase"]) and np.allclose(corr, g["phase_corr"], rtol=1e-12)


@pytest.mark.gpu
def test_gpu_find_preamble_offset_full_capture(ctx):
    """diff_method.py:164-190 on the whole 43 956-sample capture."""
    import dsp_amd
    from scipy.signal import firwin, lfilter
    g = golden("g_diffcorr")
    raw = np.fromfile(os.path.join(GOLDEN, "data", "sig_symb_x4_ncr_77671952566_logon_id_1_tx_id_7616.pcm"),
                      dtype=np.float32)
    sig = raw[::2] + 1j * raw[1::2]
    sig = sig * np.exp(-1j * 2 * np.pi * 0.0105 * np.arange(len(sig)))
    sig = lfilter(firwin(101, cutoff=0.14, window="hamming", fs=1.0), 1.0, sig)
    sig = sig / np.std(sig)
    assert len(sig) == int(g["n_full"])
    off, aligned, conv, cmax, ph = dsp_amd.find_preamble_offset(sig, g["pre"], 4, ctx=ctx)
    assert (off, ph) == (int(g["offset_full"]), int(g["phase_full"])) == (518, 1)
    for p in range(4):
        assert int(cmax[p][0]) == int(g["conv_max_full"][p, 0])
        assert abs(cmax[p][1] - g["conv_max_full"][p, 1]) < 1e-5 * g["conv_max_full"][p, 1]
    assert len(aligned) == 41883


@pytest.mark.gpu
def test_gpu_find_preamble_offset_complex64_and_dead_windows(oracle, ctx):
    import dsp_amd
    g = golden("g_diffcorr")
    x, pre = g["syn_x"], g["syn_pre"]
    assert x.dtype == np.complex64
    off, aligned, conv, cmax, ph = dsp_amd.find_preamble_offset(x, pre, 2, ctx=ctx)
    assert (off, ph) == (int(g["syn_offset"]), int(g["syn_phase"])) == (100, 0)
    for p in range(2):
        ref = g[f"syn_conv{p}"]                       # the reference ran this one in float32 (complex64 in)
        assert conv[p].shape == ref.shape
        assert np.max(np.abs(conv[p] - ref)) <= 1e-4 * np.max(np.abs(ref))
        assert np.array_equal(conv[p] == 0, ref == 0)          # dead windows and the tail stay 0
    # float64 definition on the upcast samples = the reference on x.astype(complex128)
    o2 = oracle.find_preamble_offset(x.astype(np.complex128), pre, 2)
    for p in range(2):
        assert np.max(np.abs(conv[p] - o2[2][p])) <= 2e-6 * np.max(np.abs(o2[2][p]))
    with pytest.raises(ValueError):
        dsp_amd.diff_corr(x, pre[:1], 2, ctx=ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("interp", [False, True])
def test_gpu_worker_processing(ctx, interp):
    """worker.Processing.process (worker.py:276-337) end to end on the seeded worker.Generation signal
    the reference itself processed: same peaks, same number of bursts, same derotated symbols."""
    from dsp_amd.receiver import Processing
    g = golden("g_worker")
    tag = "i" if interp else "n"
    pr = Processing(signal=g["w_samples"], preamble=g["w_pre"], packet_symb=g["w_symbols"], sps=4,
                    constellation="qpsk", interp=interp, ctx=ctx)
    peaks, found = pr.process()
    assert np.array_equal(np.asarray(peaks), g[f"w_peaks_{tag}"])
    assert len(found) == int(g[f"w_nfound_{tag}"])
    for j, s in enumerate(found):
        ref = g[f"w_found_{tag}{j}"]
        assert s.shape == ref.shape
        assert np.max(np.abs(s - ref)) < 1e-9
        quad = lambda z: (z.real < 0).astype(np.uint8) + 2 * (z.imag < 0)      # noqa: E731
        assert np.array_equal(quad(s), quad(ref))


@pytest.mark.gpu
def test_generation_reproduces_reference(tmp_path):
    """worker.Generation (worker.py:339-427) with the same seed: identical symbols (same numpy draws),
    samples equal to the reference's up to the device filter's rounding; the PCM writer round-trips."""
    import dsp_amd
    from dsp_amd.generation import Generation, save_as_pcm_float32_iq
    gw = np.load(os.path.join(GOLDEN, "g_worker.npz"))
    np.random.seed(20251114)
    gen = Generation(constellation="qpsk", preamble_length=64, signal_length=400, sps=4, Frel=0.0004,
                     Noise_power=0.02, N=3, Delay=30)
    pre, samples, symbols = gen.generate()
    assert np.array_equal(pre, gw["w_pre"]) and np.array_equal(symbols, gw["w_symbols"])
    assert samples.dtype == np.complex128 and samples.shape == gw["w_samples"].shape
    assert np.abs(samples - gw["w_samples"]).max() < 1e-12
    for c in ("bpsk", "8psk", "16psk", "16apsk"):
        p, s, y = Generation(c, 16, 40, 4, 1e-4, 0.01, 2, 5).generate()
        assert len(p) == 16 and len(y) == 40 and len(s) == (5 + 2 * 45) * 4
    path = str(tmp_path / "x.pcm")
    save_as_pcm_float32_iq(samples, path)
    back = dsp_amd.read_iq_float32(path)
    assert np.array_equal(back, samples.astype(np.complex64))


@pytest.mark.gpu
@pytest.mark.parametrize("n_pre,sps,nsym", [(300, 32, 340), (130, 3, 400), (2, 1, 50)])
def test_diff_corr_shapes(oracle, n_pre, sps, nsym):
    """Preambles long enough to leave the LDS-staged kernel for the direct one ((L-1)*sps > ~9000),
    an odd sps with a ragged tail, and the shortest preamble at one sample per symbol: all against
    the restated loop."""
    import dsp_amd
    rng = np.random.default_rng(n_pre)
    n = nsym * sps + (sps // 2)
    pre = np.exp(1j * np.pi / 2 * rng.integers(0, 4, n_pre))
    x = 0.5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x[7 * sps + 1:7 * sps + 1 + n_pre * sps:sps] += pre * np.exp(0.3j)
    got = dsp_amd.diff_corr(x, pre, sps)
    _, _, want, _, _ = oracle.find_preamble_offset(x, pre, sps)
    assert len(got) == sps
    for p in range(sps):
        assert got[p].shape == want[p].shape
        assert np.abs(got[p] - want[p]).max() <= 2e-6 * max(1.0, np.abs(want[p]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("n_pre,sps,nsym,c64", [
    (64, 4, 11000, True), (64, 4, 3000, False), (2, 1, 50, False), (3, 1, 5000, True), (8, 2, 777, False),
    (9, 3, 4001, True), (200, 7, 2500, False), (15, 5, 300, True), (300, 32, 340, False), (33, 64, 200, True),
    (130, 64, 180, False), (1000, 2, 9000, True),
])
def test_diff_corr_window_kernel_same_bits(ctx, n_pre, sps, nsym, c64):
    """The register-window kernel (a thread's outputs share their LDS reads) against round 1's kernel:
    the same sums in the same order, so the same bits -- lengths that end inside a workgroup's block,
    taps that are not a multiple of the window, one to sixty-four phases, both sample types.  (The last
    but one shape is too long for the window kernel's LDS rows and runs the older kernels either way.)"""
    import dsp_amd
    rng = np.random.default_rng(n_pre * 131 + sps)
    n = nsym * sps + (sps // 2)
    pre = np.exp(1j * np.pi / 2 * rng.integers(0, 4, n_pre))
    x = 0.5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x[7 * sps + 1:7 * sps + 1 + n_pre * sps:sps] += pre[:len(x[7 * sps + 1:7 * sps + 1 + n_pre * sps:sps])] * np.exp(0.3j)
    if c64:
        x = x.astype(np.complex64)
    got = {}
    try:
        for w in (0, 1):
            ctx.set_option("corr_window", w)
            got[w] = dsp_amd.diff_corr(x, pre, sps, ctx=ctx)
    finally:
        ctx.set_option("corr_window", 1)
    assert len(got[0]) == len(got[1]) == sps
    for p in range(sps):
        assert got[0][p].shape == got[1][p].shape
        assert np.array_equal(got[0][p].view(np.uint32), got[1][p].view(np.uint32)), p
    assert n_pre == 2 or any(np.abs(g).max(initial=0) > 0 for g in got[1])      # one difference: its std is 0, all zeros


@pytest.mark.gpu
def test_demo_worker_script():
    """tools/demo_worker.py = the reference's test_worker.py flow (interp=True): five noise-free packets
    found half a sample apart on the doubled grid, every packet received and decided correctly."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("demo_worker", os.path.join(os.path.dirname(GOLDEN), "..", "tools",
                                                                              "demo_worker.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    peaks, found, f_rel, best, quad = mod.main(True)
    assert list(map(int, peaks)) == [240, 16480, 32720, 48960, 65200]
    assert len(found) == 5 and all(len(f) == 2000 for f in found)
    assert np.abs(f_rel).max() < 1e-4
    assert quad > 0.97          # the loops acquire over the first symbols of a packet
    peaks, found, f_rel, best, quad = mod.main(False)
    assert list(map(int, peaks)) == [120, 8240, 16360, 24480, 32600]      # the peaks SURVEY.md 8c quotes
