"""Segment-parallel mode (north-star (c)): one long capture cut into segments
with warm-up overlap and stitching.  The reference has no counterpart; parity is defined as
(1) the oracle's segmented restatement agrees with the SEQUENTIAL oracle Gardner loop over the
whole stream after warm-up (timing within 1e-3 symbol, same symbol count, decisions equal to the
transmitted symbols), and (2) the device path equals the oracle's segmented restatement bit for
bit on the same filtered input."""
import numpy as np
import pytest

SPS = 4


def make_stream(n_sym, seed=7, cfo=1.5e-4, tau=0.3, esn0_db=20.0, drift=2.0e-6):
    """Continuous QPSK at 4 sps: RRC(0.35) TX, fractional delay with slow drift, CFO, AWGN."""
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 4, n_sym)
    sym = ((1 - 2 * (q & 1)) + 1j * (1 - 2 * (q >> 1))) / np.sqrt(2)
    up = np.zeros(n_sym * SPS, complex)
    up[::SPS] = sym
    h = O.rrc_filter(SPS, 10, 0.35)
    tx = np.convolve(up, h, "same")
    t = np.arange(tx.size, dtype=float)
    d = tau + drift * t                                   # slowly drifting timing offset (samples)
    txd = np.interp(t - d, t, tx.real) + 1j * np.interp(t - d, t, tx.imag)
    x = txd * np.exp(1j * (2 * np.pi * cfo * t + 0.4))
    sigma = np.sqrt(np.mean(np.abs(x) ** 2) * SPS / (2 * 10 ** (esn0_db / 10)))
    x = x + sigma * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))
    return q.astype(np.uint8), x, h


def best_rotation_match(dec, q):
    turn = np.array([1, 3, 0, 2], np.uint8)
    best = 0.0
    d = dec.copy()
    for _ in range(4):
        best = max(best, float(np.mean(d == q)))
        d = turn[d]
    return best


@pytest.fixture(scope="module")
def stream(oracle):
    q, x, h = make_stream(40000)
    y = oracle.filter_samples(x, h).astype(np.complex64)
    return q, x, h, y


def test_oracle_segmented_vs_sequential(oracle, stream):
    q, x, h, y = stream
    seg = oracle.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1.0, seg_len=16384, warmup=4096, acquire=1)
    rec, err, mu = oracle.gardner_timing_recovery(y, SPS, 0.007)
    assert seg["n"] == len(rec)                                   # same symbols, no gap, no duplicate
    S = seg["seg_info"].shape[0]
    assert S == 10 and (seg["seg_info"][1:, 0] > 4000).all()      # warm-up rows are discarded
    # timing: mu of the stitched stream vs the sequential loop.  The timing loop forgets its start
    # with a time constant of ~480 symbols at alpha = 0.007: 4096 warm-up symbols leave < 1e-3 symbol
    # (2048 leave ~5e-3, 1024 ~3e-2: measured on this stream)
    mu_seq = np.concatenate(([0.0], mu))
    dm = np.abs(seg["mu"][2000:] - mu_seq[2000:len(seg["mu"])])
    assert np.max(dm) / SPS < 1e-3, np.max(dm)                    # 1e-3 symbol (north star)
    # decisions: the transmitted symbols up to ONE global rotation (delay of the two RRCs is 0
    # in 'same' mode; the first symbol of the stream is sample 0)
    n = min(seg["n"], q.size)
    assert best_rotation_match(seg["dec"][500:n], q[500:n]) > 0.999


@pytest.mark.parametrize("which", ["stream", "stream_half"])
@pytest.mark.parametrize("warm", [2048, 1536])
def test_oracle_segmented_fractional_start(oracle, request, which, warm):
    """acquire = 2 (the default): every window starts at the fractional phase of the symbol-rate line of
    |y|^2 over its first 512 symbols -- 0.01 sample rms from where a sequential loop stands at that point,
    where the sample grid (acquire = 1) is up to half a sample off -- so the same 1e-3-symbol bar against ONE
    sequential loop, started at the first window's phase, holds after 1536 warm-up symbols (measured 4.6e-4
    and 6.3e-4 on these streams; 2048, the default, leaves 2e-4: what 4096 leave from the sample grid)."""
    q, x, h, y = request.getfixturevalue(which)
    m0 = oracle.acquire_phase_frac(y[:16384 + 64], SPS)
    assert 0.0 <= m0 < SPS and m0 * 1024 == int(m0 * 1024)
    assert abs(m0 - oracle.acquire_phase(y[:4096], SPS)) < 0.6
    seg = oracle.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1.0, seg_len=16384, warmup=warm)
    rec, err, mu = oracle.gardner_timing_recovery(y, SPS, 0.007, m0)
    assert seg["n"] == len(rec)
    mu_seq = np.concatenate(([m0], mu))
    dm = np.abs(seg["mu"][2000:] - mu_seq[2000:len(seg["mu"])])
    assert np.max(dm) / SPS < 1e-3, np.max(dm)
    n = min(seg["n"], q.size)
    assert best_rotation_match(seg["dec"][500:n], q[500:n]) > 0.999


def test_oracle_fractional_phase_definition(oracle):
    """The estimate on a noiseless line: energies A + B cos(2 pi (p - tau) / sps) give tau back (to the 1/1024
    rounding) at any sps >= 3 (at 2 samples per symbol the line sits at Nyquist and only its sign is left:
    the sample grid's answer); an all-zero window gives 0."""
    w2 = np.sqrt(2.0 + np.cos(np.pi * (np.arange(128) - 0.37))).astype(np.complex64)
    assert oracle.acquire_phase_frac(w2, 2) == 0.0 and oracle.acquire_phase_frac(w2[1:], 2) == 1.0
    for sps in (3, 4, 5, 8, 16):
        for tau in (0.0, 0.37, sps / 2.0, sps - 0.2):
            p = np.arange(sps * 64) % sps
            w = np.sqrt(2.0 + np.cos(2 * np.pi * (p - tau) / sps)).astype(np.complex64)
            got = oracle.acquire_phase_frac(w, sps)
            d = (got - tau + sps / 2) % sps - sps / 2
            assert abs(d) < 1.0 / 1024 + 1e-6, (sps, tau, got)
    assert oracle.acquire_phase_frac(np.zeros(100, np.complex64), 4) == 0.0


@pytest.fixture(scope="module")
def stream_half(oracle):
    """True timing half a symbol from mu = 0: the worst cold start of a Gardner loop."""
    q, x, h = make_stream(40000, seed=9, tau=2.05)
    y = oracle.filter_samples(x, h).astype(np.complex64)
    return q, x, h, y


def test_oracle_segmented_half_symbol_offset(oracle, stream_half):
    """Windows started cold at mu = 0 hang on the detector's unstable zero and the stitched stream is
    useless; started at the strobe-energy phase (acquire, the default) they lock at once and agree
    with a sequential loop started at the same lock point to < 1e-3 symbol."""
    q, x, h, y = stream_half
    assert oracle.acquire_phase(y[:4096], SPS) == 2.0
    cold = oracle.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1.0, seg_len=16384, warmup=4096, acquire=False)
    n = min(cold["n"], q.size)
    assert best_rotation_match(cold["dec"][12000:n], q[12000:n]) < 0.6
    seg = oracle.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1.0, seg_len=16384, warmup=4096, acquire=1)
    n = min(seg["n"], q.size)
    assert best_rotation_match(seg["dec"][2000:n], q[2000:n]) > 0.999
    rec, err, mu = oracle.gardner_timing_recovery(y, SPS, 0.007, 2.0)
    assert abs(seg["n"] - len(rec)) <= 1
    mu_seq = np.concatenate(([2.0], mu))
    m = min(len(mu_seq), len(seg["mu"]))
    dm = np.abs(seg["mu"][2000:m] - mu_seq[2000:m])
    assert np.max(dm) / SPS < 1e-3, np.max(dm)


def test_oracle_segmented_single_segment_is_sequential(oracle, stream):
    q, x, h, y = stream
    seg = oracle.sync_capture_segmented(y[:20000], SPS, 0.007, 0.0, 0.003, 1.0, seg_len=32768, warmup=512)
    o = oracle.sync_chain_c64(y[:20000], sps=SPS, alpha=0.007, Bn=0.003)
    assert seg["n"] == o["n"] and np.array_equal(seg["dec"], o["dec"])


@pytest.mark.gpu
def test_gpu_segmented_matches_oracle(oracle, stream, ctx):
    import dsp_amd
    q, x, h, y = stream
    for seg_len, warm in ((16384, 1024), (8192, 512), (65536, 1024)):
        ref = oracle.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1.0, seg_len=seg_len, warmup=warm)
        got = dsp_amd.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1, seg_len=seg_len, warmup=warm, ctx=ctx)
        assert got["n"] == ref["n"]
        assert np.array_equal(got["seg_info"], ref["seg_info"])
        assert np.array_equal(got["dec"], ref["dec"])                        # bit-exact decisions
        assert np.max(np.abs(got["mu"] - ref["mu"])) < 1e-6
        assert np.max(np.abs(got["freq"] - ref["freq"])) < 1e-9              # 1e-4 rad/sample >> this


@pytest.mark.gpu
def test_gpu_segmented_half_symbol_offset(oracle, stream_half, ctx):
    """Per-window phase acquisition on the device: same start phases, shares, rotations and decisions
    as the restatement, with and without it."""
    import dsp_amd
    q, x, h, y = stream_half
    for acq in (True, False):
        ref = oracle.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1.0, seg_len=16384, warmup=2048, acquire=acq)
        got = dsp_amd.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1, seg_len=16384, warmup=2048, acquire=acq,
                                             ctx=ctx)
        assert got["n"] == ref["n"]
        assert np.array_equal(got["seg_info"], ref["seg_info"])
        assert np.array_equal(got["dec"], ref["dec"])
        assert np.max(np.abs(got["mu"] - ref["mu"])) < 1e-6
    n = min(got["n"], q.size)
    assert best_rotation_match(dsp_amd.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1, seg_len=16384, warmup=2048,
                                                              ctx=ctx)["dec"][2000:n], q[2000:n]) > 0.999


@pytest.mark.gpu
def test_gpu_segmented_int16_with_matched_filter(oracle, stream, ctx):
    import dsp_amd
    q, x, h, y = stream
    pk = np.max(np.abs(np.concatenate((x.real, x.imag))))
    w = np.empty(2 * x.size, np.int16)
    w[0::2] = np.round(x.real / pk * 20000)
    w[1::2] = np.round(x.imag / pk * 20000)
    got = dsp_amd.sync_capture_segmented(w, SPS, 0.007, 0.0, 0.003, 1, taps=h, seg_len=16384, warmup=1024, ctx=ctx)
    xc = oracle.read_iq_int16_words(w, True)
    yf = oracle.convolve_same(xc, h)                      # unscaled, held as complex64 on the device
    ref = oracle.sync_capture_segmented(yf.astype(np.complex64), SPS, 0.007, 0.0, 0.003, 1.0, seg_len=16384,
                                        warmup=1024, scale=1.0 / oracle.np_std(yf))
    assert got["n"] == ref["n"]
    # the device filter accumulates in another order: last-ulp differences of the complex64 input
    assert np.mean(got["dec"] == ref["dec"]) > 0.9999
    assert np.max(np.abs(got["mu"] - ref["mu"])) < 1e-3
    n = min(got["n"], q.size)
    assert best_rotation_match(got["dec"][500:n], q[500:n]) > 0.999


@pytest.mark.gpu
def test_gpu_capture_pipeline_device_generated(ctx):
    """2^24-sample int16 capture generated on the device (config 5 scaled down): symbol count and
    decisions against the transmitted symbols, aligned by the stitched positions; a drift large
    enough to make the timing loop slip symbols."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    from bench_capture import check_decisions, SEED
    from dsp_amd.pipeline import CapturePipeline
    n, sps = 1 << 24, 4
    for drift in (0.0, 1.0e-6):                 # 1e-6 * 2^24 = 16.8 samples = 4 symbol slips
        pipe = CapturePipeline(ctx, n, sps, seg_len=65536, warmup=4096)
        pipe.generate(SEED, tau0=0.3, drift=drift)
        nsym = pipe.run()
        rate, nslip = check_decisions(pipe, n // sps, 0.3, drift, sps, 0, nsym - 8)
        assert abs(nsym - int(round(n * (1.0 + drift) / sps))) <= 4
        assert (nslip >= 3) == (drift > 0)
        assert rate > 0.999, (drift, rate)
        info = pipe.info
        assert (info[1:, 0] > 4000).all() and (info[:, 1] > info[:, 0]).all()
        pipe.close()


@pytest.mark.gpu
def test_gpu_segmented_edge_shapes(oracle, stream, ctx):
    """Ragged ends: capture shorter than one segment, a last segment of a few symbols, no warm-up,
    a capture length that is not a multiple of sps, and bad arguments."""
    import dsp_amd
    q, x, h, y = stream
    cases = [(y[:5000], 8192, 256),          # one segment: the plain sequential chain
             (y[:16384 + 40], 16384, 512),   # second segment of 10 symbols
             (y[:50001], 4096, 0),           # odd length, no warm-up (shares still tile the stream)
             (y[:30000], 4096, 2048)]        # warm-up longer than the segment
    for xs, seg_len, warm in cases:
        ref = oracle.sync_capture_segmented(xs, SPS, 0.007, 0.0, 0.003, 1.0, seg_len=seg_len, warmup=warm)
        got = dsp_amd.sync_capture_segmented(xs, SPS, 0.007, 0.0, 0.003, 1, seg_len=seg_len, warmup=warm, ctx=ctx)
        assert got["n"] == ref["n"], (len(xs), seg_len, warm)
        assert np.array_equal(got["seg_info"], ref["seg_info"])
        assert np.array_equal(got["dec"], ref["dec"])
    o = oracle.sync_chain_c64(y[:5000], sps=SPS, alpha=0.007, Bn=0.003)
    got = dsp_amd.sync_capture_segmented(y[:5000], SPS, 0.007, 0.0, 0.003, 1, seg_len=8192, warmup=256, ctx=ctx)
    assert got["n"] == o["n"] and np.array_equal(got["dec"], o["dec"])
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.sync_capture_segmented(y[:5000], SPS, seg_len=4098, ctx=ctx)      # not a multiple of sps
    with pytest.raises(IndexError):
        dsp_amd.sync_capture_segmented(np.zeros(0, np.complex64), SPS, ctx=ctx)


def test_auto_seg_len_rule():
    """auto_seg_len: a multiple of sps the C-ABI accepts, the warm-up shareable (lead <= seg_len)
    from 2^28 samples on, 512 tiles aimed at for long captures."""
    from dsp_amd.api import auto_seg_len
    for sps in (2, 4, 5, 8):
        for warm in (0, 100, 1024, 4096):
            for lg in range(10, 34, 3):
                s = auto_seg_len(1 << lg, sps, warm)
                assert s % sps == 0 and s >= 16 * sps
                if lg >= 28:
                    assert s >= warm * sps
    assert auto_seg_len(1 << 30, 4, 4096) == 32768
    assert auto_seg_len(1 << 28, 4, 4096) == 16384
    assert auto_seg_len(1 << 24, 4, 4096) == 4096
    # rounded up: never a 513th tile of windows while one warm-up per segment still fits, whatever the length
    rng = np.random.default_rng(3)
    for n in [int(v) for v in rng.integers(1 << 29, 1 << 31, 50)] + [(1 << 30) - 1000, (1 << 30) + 4, 10 ** 9]:
        for sps in (2, 4, 8):
            s = auto_seg_len(n, sps, 2048)
            assert s % sps == 0 and s % 4 == 0
            if s >= 2048 * sps and s < max(2 * 2048, 8192) * sps:
                assert -(-n // s) <= 512 * 64, (n, sps, s)


def test_seam_frequency_steps():
    """|freq| steps across the seams of a stitched capture, from the shares in seg_info (empty shares skipped)."""
    from dsp_amd.api import seam_frequency_steps
    info = np.array([[0, 3, 0, 5], [2, 2, 0, 4], [1, 5, 0, 6], [0, 2, 0, 2]], np.int32)    # shares 3, 0, 4, 2
    freq = np.array([1.0, 1.0, 1.5, 2.0, 2.0, 2.0, 2.5, 1.0, 1.0], np.float32)
    d = seam_frequency_steps(freq, info)
    assert d.dtype == np.float64 and np.allclose(d, [0.5, 1.5])                           # seams at symbols 3 and 7
    assert seam_frequency_steps(freq[:3], info[:1]).size == 0


@pytest.mark.gpu
def test_capture_buffers_kept_and_trimmed(ctx, oracle, stream):
    """The context keeps the mode's work buffers between calls: a larger call after a smaller one,
    a smaller one after it, and a call after trim() all give the oracle's result; seg_len=None is
    auto_seg_len."""
    import dsp_amd
    y = stream[3]
    for n, seg_len in ((20000, 4096), (y.size, 8192), (9000, 2048)):
        ref = oracle.sync_capture_segmented(y[:n], SPS, 0.007, 0.0, 0.003, 1.0, seg_len=seg_len, warmup=512)
        got = dsp_amd.sync_capture_segmented(y[:n], SPS, 0.007, 0.0, 0.003, 1, seg_len=seg_len, warmup=512, ctx=ctx)
        assert got["n"] == ref["n"] and np.array_equal(got["dec"], ref["dec"])
        assert np.array_equal(got["seg_info"], ref["seg_info"])
        if n == y.size:
            ctx.trim()
    auto = dsp_amd.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1, warmup=512, ctx=ctx)
    ref = oracle.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1.0, warmup=512,
                                        seg_len=dsp_amd.api.auto_seg_len(y.size, SPS, 512))
    assert auto["n"] == ref["n"] and np.array_equal(auto["dec"], ref["dec"])


@pytest.mark.gpu
def test_capture_int16_pointer_alignment(ctx):
    """The device-pointer entry point takes any IQ-pair-aligned (4-byte) capture pointer: the peak
    search reads 16-byte words between a head and a tail, the filter stages from arbitrary window
    starts.  Same capture at byte offsets 0, 4, 8, 12 and with lengths that leave 0..3 samples after
    the last whole word -> the same decisions; the peak sample sits in the head / tail once each."""
    from dsp_amd.pipeline import CapturePipeline
    rng = np.random.default_rng(99)
    n = 60000 + 3
    q, x, h = make_stream(n // SPS + 1, seed=5)
    x = x[:n]
    iq = np.empty(2 * n, np.int16)
    iq[0::2] = np.clip(np.rint(x.real * 4000), -32767, 32767)
    iq[1::2] = np.clip(np.rint(x.imag * 4000), -32767, 32767)
    ref = {}
    for cut in (0, 1, 2, 3):
        for where in ("none", "head", "tail"):
            m = n - cut
            w = iq[:2 * m].copy()
            if where == "head":
                w[0] = 32767
            elif where == "tail":
                w[2 * m - 1] = -32768
            for off in (0, 1, 2, 3):
                pipe = CapturePipeline(ctx, m + 4, SPS, seg_len=8192, warmup=512, want_freq=True)
                pipe.n = m
                pipe.S = -(-m // 8192)
                pipe.info = np.zeros((pipe.S, 4), np.int32)
                pipe._args.seg_info = pipe.info.ctypes.data
                pipe._args.n_samples = m
                pipe._args.x = pipe.d_x + 4 * off
                ctx.h2d(pipe.d_x + 4 * off, w)
                ns = pipe.run()
                got = (ns, pipe.fetch("dec").tobytes(), pipe.fetch("mu").tobytes(), pipe.fetch("freq").tobytes())
                pipe.close()
                if off == 0:
                    ref[(cut, where)] = got
                assert got == ref[(cut, where)], (cut, where, off)
    assert ref[(0, "none")][1] != ref[(0, "head")][1] or ref[(0, "none")][2] != ref[(0, "head")][2]


@pytest.mark.gpu
@pytest.mark.parametrize("n_keep", [100 * 2048, 130 * 2048 - 700, 128 * 2048, 192 * 2048])
def test_gpu_shared_warmup_rows_bit_identical(oracle, ctx, n_keep):
    """More than 64 windows with the matched filter: a window's warm-up rows come from its
    predecessor (stored by the predecessor's filter workgroup, or copied afterwards) instead of being
    filtered again.  All three ways give the same bits; the result follows the oracle as closely as
    the per-window filter does (test_gpu_segmented_int16_with_matched_filter).
    A whole number of 64-window tiles of whole segments (the last two sizes) takes, with lead_copy 0,
    the segment-wise pass of the persistent filter kernel (fir_strip_kernel STREAM: every stream sample
    filtered once, stored into its own window and into the neighbours' warm-up / tail rows) -- the
    same bits again."""
    import dsp_amd
    q, x, h = make_stream(n_keep // SPS + 64, seed=23)
    x = x[:n_keep]
    pk = np.max(np.abs(np.concatenate((x.real, x.imag))))
    w = np.empty(2 * x.size, np.int16)
    w[0::2] = np.round(x.real / pk * 20000)
    w[1::2] = np.round(x.imag / pk * 20000)
    outs = []
    try:
        for mode in (2, 1, 0):
            ctx.set_option("lead_copy", mode)
            outs.append(dsp_amd.sync_capture_segmented(w, SPS, 0.007, 0.0, 0.003, 1, taps=h, seg_len=2048, warmup=256,
                                                       ctx=ctx))
    finally:
        ctx.set_option("lead_copy", 0)
    assert outs[0]["seg_info"].shape[0] > 64
    for o in outs[1:]:
        assert o["n"] == outs[0]["n"]
        for k in ("dec", "mu", "freq", "seg_info"):
            assert np.array_equal(o[k], outs[0][k]), k
    xc = oracle.read_iq_int16_words(w, True)
    yf = oracle.convolve_same(xc, h)
    ref = oracle.sync_capture_segmented(yf.astype(np.complex64), SPS, 0.007, 0.0, 0.003, 1.0, seg_len=2048, warmup=256,
                                        scale=1.0 / oracle.np_std(yf))
    assert outs[0]["n"] == ref["n"]
    assert np.mean(outs[0]["dec"] == ref["dec"]) > 0.9999
    n = min(ref["n"], q.size)
    assert best_rotation_match(outs[0]["dec"][500:n], q[500:n]) > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("nseg,seg_len,warm,kind,acquire,trim", [
    (64, 4096, 1024, "i16", True, 0),       # one tile: every window is an edge-tile window
    (128, 2048, 512, "c64", True, 0),       # warm-up == seg_len / sps: the longest lead the segment tiles take
    (128, 2048, 0, "i16", True, 0),         # no warm-up at all
    (192, 2048, 256, "c64", False, 0),      # fixed start value instead of the energy rule
    (256, 1024, 128, "i16", True, 0),
    (128, 2048, 256, "i16", True, 1000),    # the stream ends inside its last segment
    (130, 2048, 256, "i16", True, 0),       # 62 empty segments behind the stream's end (lane-major: lanes 43.. of tile 1, ...)
    (150, 2048, 512, "c64", True, 1234),    # both
    (70, 4096, 1024, "i16", True, 1332),
])
def test_gpu_segment_tiles_match_window_tiles(oracle, ctx, nseg, seg_len, warm, kind, acquire, trim):
    """Whole tiles of whole segments with the matched filter: the stream is filtered once into lane-major
    segment tiles and the windows are row ranges of that buffer.  Every output --
    decisions, mu, frequency, shares, rotations -- has the bits of the pass over the overlapping windows
    (lead_copy 2 filters every window on its own), for int16 and complex64 input; decisions follow the
    oracle's segmented definition as closely as that pass does."""
    import dsp_amd
    n = nseg * seg_len - trim
    q, x, h = make_stream(n // SPS + 64, seed=31 + nseg)
    x = x[:n]
    if kind == "i16":
        pk = np.max(np.abs(np.concatenate((x.real, x.imag))))
        w = np.empty(2 * x.size, np.int16)
        w[0::2] = np.round(x.real / pk * 20000)
        w[1::2] = np.round(x.imag / pk * 20000)
        xin, xc = w, oracle.read_iq_int16_words(w, True)
    else:
        xin = x.astype(np.complex64)
        xc = xin
    outs = []
    try:
        for mode in (2, 0):
            ctx.set_option("lead_copy", mode)
            outs.append(dsp_amd.sync_capture_segmented(xin, SPS, 0.007, 0.0, 0.003, 1, taps=h, seg_len=seg_len,
                                                       warmup=warm, acquire=acquire, ctx=ctx))
    finally:
        ctx.set_option("lead_copy", 0)
    a, b = outs
    assert a["seg_info"].shape[0] == nseg and a["n"] == b["n"]
    for k in ("dec", "mu", "freq", "seg_info"):
        assert np.array_equal(a[k], b[k]), k
    yf = oracle.convolve_same(xc, h)
    ref = oracle.sync_capture_segmented(yf.astype(np.complex64), SPS, 0.007, 0.0, 0.003, 1.0, seg_len=seg_len,
                                        warmup=warm, scale=1.0 / oracle.np_std(yf), acquire=acquire)
    assert b["n"] == ref["n"]
    assert np.mean(b["dec"] == ref["dec"]) > 0.9999


@pytest.mark.gpu
@pytest.mark.parametrize("lead_copy", [0, 2])
def test_gpu_more_windows_than_resident_tiles(ctx, lead_copy):
    """100 032 windows = 1563 tiles, more than six per CU: the recurrence goes out as two launches of
    whole tiles (option batch_split), over the lane-major segment tiles (lead_copy 0: the windows are
    overlapping row ranges of one buffer and tile 0's warm-up rows live elsewhere) and over separate
    window tiles (2).  Same bits as the single launch with the surplus tiles queued on the CUs."""
    import dsp_amd
    nseg, seg_len, warm = 100032, 256, 32
    q, x, h = make_stream(400000, seed=5)
    reps = nseg * seg_len // x.size + 1
    x = np.tile(x.astype(np.complex64), reps)[:nseg * seg_len]       # period 6250 segments: not a whole number of tiles
    outs = []
    try:
        ctx.set_option("lead_copy", lead_copy)
        for split in (1, 0):
            ctx.set_option("batch_split", split)
            outs.append(dsp_amd.sync_capture_segmented(x, SPS, 0.007, 0.0, 0.003, 1, taps=h, seg_len=seg_len,
                                                       warmup=warm, ctx=ctx))
    finally:
        ctx.set_option("lead_copy", 0)
        ctx.set_option("batch_split", 1)
    a, b = outs
    assert a["seg_info"].shape[0] == nseg and a["n"] == b["n"] and a["n"] > 0.99 * (x.size // SPS)
    for k in ("dec", "mu", "freq", "seg_info"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
def test_gpu_segmented_frequency_across_seams(oracle, stream, ctx):
    """What a caller may assume about `freq` across seams (SURVEY 8d asks for the seam steps to be
    reported): it is a per-window estimate.  The reported steps are those of the oracle's definition;
    after a 4096-symbol warm-up they stay within a few standard deviations of the loop's own estimate
    noise, with a 512-symbol warm-up (shorter than the loop's settling time at Bn = 0.003) they are
    several times larger -- while inside a window the estimate moves by < 1e-5 per symbol."""
    import dsp_amd
    from dsp_amd.api import seam_frequency_steps
    q, x, h, y = stream
    steps = {}
    for seg_len, warm in ((16384, 4096), (8192, 512)):
        ref = oracle.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1.0, seg_len=seg_len, warmup=warm)
        got = dsp_amd.sync_capture_segmented(y, SPS, 0.007, 0.0, 0.003, 1, seg_len=seg_len, warmup=warm, ctx=ctx)
        want = seam_frequency_steps(ref["freq"], ref["seg_info"])
        assert got["seam_dfreq"].shape == want.shape == (ref["seg_info"].shape[0] - 1,)
        assert np.max(np.abs(got["seam_dfreq"] - want)) < 1e-9
        f = got["freq"].astype(np.float64)
        inner = np.abs(np.diff(f))
        share = np.cumsum(got["seg_info"][:, 1] - got["seg_info"][:, 0])[:-1]
        inner[share - 1] = 0.0
        assert inner.max() < 1e-5
        steps[warm] = (got["seam_dfreq"].max(), f[4096:].std())
    assert steps[4096][0] < 5.0 * steps[4096][1]          # a few sigma of the estimate's own noise
    assert steps[512][0] > 3.0 * steps[4096][0]           # an unsettled loop: not a continuous quantity
