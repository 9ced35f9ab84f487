"""GPU parity tests of the Gardner / FLL recurrence kernels (call through the C ABI).

Bars (north_star): hard QPSK decisions bit-exact; Gardner trajectories are
float64 on the device with the reference's rounding, so they are compared for
EXACT equality (float64 outputs) or exact equality after float32 rounding
(complex64 path); FLL frequency within 1e-12 cycles/sample of the oracle (the
north-star tolerance is 1e-4 rad/sample), derotated samples within 1e-9.
"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

FREQ_TOL = 1e-12
SIG_TOL = 1e-9


def _chain_oracle(O, y32, sps=4, alpha=0.007, Bn=0.03, scale=1.0):
    return O.sync_chain_c64(y32, sps=sps, alpha=alpha, Bn=Bn, y=1.0, scale=scale)


def _fll_self_consistency(x, out, freq, ef, Bn, y=1.0):
    """What survives the chaos beyond the stability horizon: the device's outputs obey the
    recurrence of fll_func.py:14-26 step by step over the FULL length, whatever trajectory the
    loop is on.  Each step is recomputed in numpy from the device's own previous state:
      curr_freq_list[i] = curr_freq_list[i-1] + ef_n_list[i]                 (:25-26, exact)
      signal_list[i]    = x[i] * exp(-1j * 2*pi * curr_freq_list[i-1] * i)   (:16-18)
      ef_n_list[i]      = error_i * kp + integrator_i,  integrator_i = integrator_(i-1) + error_i * ki,
                          error_i = angle(signal_list[i] * conj(decided_i))  (:19-24)
      |error_i| <= pi/4 for a QPSK slicer (no exactly-zero components here), which bounds |ef_n|."""
    import dsp_amd
    kp, ki = dsp_amd.fll_gains(Bn, y)
    n = len(x)
    i = np.arange(n, dtype=np.float64)
    f_prev = np.concatenate([[0.0], freq[:-1]])
    assert np.array_equal(freq, f_prev + ef)
    want = x * np.exp(-1j * (2 * np.pi * f_prev * i))
    assert np.max(np.abs(out - want)) < 1e-11 * max(1.0, np.max(np.abs(x)))
    decided = (np.sign(out.real) + 1j * np.sign(out.imag)) / np.sqrt(2)
    err = np.angle(out * np.conj(decided))
    assert np.max(np.abs(err)) <= np.pi / 4 + 1e-12
    integ = np.cumsum(err * ki)
    assert np.max(np.abs(ef - (err * kp + integ))) < 1e-15 + 1e-12 * np.max(np.abs(ef))
    assert np.all(np.abs(ef) <= np.pi / 4 * (kp + ki * (i + 1)) + 1e-15)


def test_dropin_gardner_config1_bit_exact(ctx):
    import dsp_amd
    g = golden("g_config1")
    rec, err, mu = dsp_amd.gardner_timing_recovery(g["y"], 4, alpha=0.007)
    assert rec.dtype == np.complex128 and mu.dtype == np.float64
    assert len(rec) == 10499 and mu[-1] == -0.05891794872812668        # SURVEY 8c known answer
    assert np.array_equal(rec, g["rec"])
    assert np.array_equal(err, g["terr"])
    assert np.array_equal(mu, g["mu"])
    rec2, err2, mu2 = dsp_amd.gardner_timing_recovery(g["yd"], 4, alpha=0.04)
    assert np.array_equal(rec2, g["rec2"]) and np.array_equal(mu2, g["mu2"])


def test_dropin_fll_config1_stable_horizon(ctx):
    """fll_func(rec, Bn=0.03): loop gain 2*pi*kp*i passes 2 at i ~ 3316, after which the
    reference itself is chaotic (a 1e-12 input perturbation flips 40% of its decisions), so
    parity is asserted on the first 3300 symbols (DESIGN.md 'FLL stability horizon')."""
    import dsp_amd
    g = golden("g_config1")
    out, cf, freq, ef = dsp_amd.fll_func(g["rec"], 0.03, 1)
    H = 3300
    assert out.shape == g["fll_out"].shape
    assert np.max(np.abs(freq[:H] - g["fll_freq"][:H])) < FREQ_TOL
    assert np.max(np.abs(ef[:H] - g["fll_ef"][:H])) < FREQ_TOL
    assert np.max(np.abs(out[:H] - g["fll_out"][:H])) < SIG_TOL
    dec = (out.real < 0) + 2 * (out.imag < 0)
    assert np.array_equal(dec[:H], g["dec"][:H])
    # the 7199 symbols beyond the horizon: the recurrence itself, step by step, on all 10 499
    _fll_self_consistency(g["rec"], out, freq, ef, 0.03)


def test_dropin_fll_config2_full(ctx):
    """diff_method.py:304-307: fll_func(y[0::4], 1e-6, 1) over all 10 471 symbols."""
    import dsp_amd
    g = golden("g_config2")
    out, cf, freq, ef = dsp_amd.fll_func(g["y"][0::4], 0.000001, 1)
    assert abs(cf - float(g["fllA_cf"])) < 1e-18
    assert np.max(np.abs(freq - g["fllA_freq"])) < 1e-18
    assert np.max(np.abs(out - g["fllA_out"])) < SIG_TOL
    rec, err, mu = dsp_amd.gardner_timing_recovery(g["y"], 4, alpha=0.007)
    assert len(rec) == 10470 and np.array_equal(rec, g["rec"]) and np.array_equal(mu, g["mu"])
    out, cf, freq, ef = dsp_amd.fll_func(rec, 0.03, 1)
    H = 3300
    assert np.max(np.abs(freq[:H] - g["fllB_freq"][:H])) < FREQ_TOL
    dec = (out.real < 0) + 2 * (out.imag < 0)
    assert np.array_equal(dec[:H], g["dec"][:H])
    assert list(dec[:16]) == [0, 3, 3, 1, 1, 3, 0, 1, 2, 3, 0, 2, 2, 1, 1, 3]
    _fll_self_consistency(rec, out, freq, ef, 0.03)          # all 10 470 symbols, see config 1


def test_dropin_gardner_ted_hw1(ctx):
    """hw1/gardner.py:94-117 flow.  The float64 algorithm (reference run on the complex128 view of
    the samples) is matched bit for bit.  The reference's own script passes a complex64 array, and
    under numpy >= 2 (NEP 50) that makes its Python-float constants weak: the loop silently runs in
    float32 (mu, sample index, interpolation).  Against that run the float64 device result agrees
    within the north-star timing tolerance (1e-3 symbol)."""
    import dsp_amd
    g = golden("g_hw1_gardner")
    y, err, mu = dsp_amd.gardner_ted(g["baseband"], 2)
    assert len(y) == 49998
    assert np.array_equal(y, g["y_f64"]) and np.array_equal(err, g["err_f64"])
    assert np.array_equal(mu, g["mu_f64"])
    assert np.max(np.abs(mu - g["mu"])) / 2 < 1e-3           # float32-accident run, in symbols
    assert abs(mu[-1] - 0.035570256412029266) / 2 < 1e-3     # SURVEY 8c known answer (float32 run)
    y2, err2, mu2 = dsp_amd.gardner_ted(g["baseband"][:12000], 2, delta_mu=0.005)
    assert np.array_equal(y2, g["y2_f64"]) and np.array_equal(mu2, g["mu2_f64"])


def test_dropin_gardner2_demo_sps8(ctx):
    import dsp_amd
    g = golden("g_gardner2_demo")
    rec, err, mu = dsp_amd.gardner_timing_recovery(g["x"], 8, alpha=0.07, mu_initial=0.0)
    assert np.array_equal(rec, g["rec"]) and np.array_equal(err, g["terr"]) and np.array_equal(mu, g["mu"])


def test_edges(ctx):
    import dsp_amd
    g = golden("g_edges")
    rng = np.random.default_rng(99)
    z = rng.standard_normal(64) + 1j * rng.standard_normal(64)
    for L in (1, 3, 4, 7, 8, 9, 12, 13):
        rec, err, mu = dsp_amd.gardner_timing_recovery(z[:L], 4, alpha=0.05)
        assert np.array_equal(rec, g[f"g2_len{L}_rec"]) and np.array_equal(mu, g[f"g2_len{L}_mu"])
        y, e, m = dsp_amd.gardner_ted(z[:L], 4)
        assert np.array_equal(y, g[f"ted_len{L}_y"]) and np.array_equal(m, g[f"ted_len{L}_mu"])
    with pytest.raises(IndexError):
        dsp_amd.gardner_timing_recovery(np.zeros(0, complex), 4)
    # symbol slips in both directions (k += sps / k -= sps), mu_initial, clip at +-sps
    for nm in ("pos", "mid", "negmid"):
        rec, err, mu = dsp_amd.gardner_timing_recovery(g["wrap_x"], 4, alpha=float(g[f"g2_wrap_{nm}_alpha"]))
        assert np.array_equal(rec, g[f"g2_wrap_{nm}_rec"]), nm
        assert np.array_equal(err, g[f"g2_wrap_{nm}_err"]) and np.array_equal(mu, g[f"g2_wrap_{nm}_mu"])
    rec, err, mu = dsp_amd.gardner_timing_recovery(g["wrap_x"], 4, alpha=0.007, mu_initial=3.7)
    assert np.array_equal(rec, g["g2_mu0_rec"]) and np.array_equal(mu, g["g2_mu0_mu"])
    y, e, m = dsp_amd.gardner_ted(g["wrap_x"], 4, delta_mu=0.8)
    assert np.array_equal(y, g["ted_clip_y"]) and np.array_equal(e, g["ted_clip_err"])
    assert np.array_equal(m, g["ted_clip_mu"])
    # fll: exact zeros and pure-axis samples (np.sign(0) = 0), other loop constants, empty
    out, cf, freq, ef = dsp_amd.fll_func(g["fll_axis_x"], 50.0, 1)
    assert np.max(np.abs(freq - g["fll_axis_freq"])) < 1e-15
    assert np.max(np.abs(out - g["fll_axis_out"])) < 1e-12
    # Bn 0.5, y 0.707: kp = 1.886e-3, loop gain 2*pi*kp*i passes 2 at i = 169 -> compare 150 symbols
    out, cf, freq, ef = dsp_amd.fll_func(g["wrap_x"][::4], 0.5, 0.707)
    assert np.max(np.abs(freq[:150] - g["fll_y707_freq"][:150])) < FREQ_TOL
    assert np.max(np.abs(out[:150] - g["fll_y707_out"][:150])) < SIG_TOL
    out, cf, freq, ef = dsp_amd.fll_func(np.zeros(0, complex), 0.03, 1)
    assert cf == 0 and out.size == 0


def test_chain_c64_golden_bursts(ctx, oracle):
    """Fused Gardner->FLL->decisions on complex64 matched-filtered bursts vs the reference
    (golden chain B) for the 6 fully stored bursts, vs the oracle for all 32."""
    import dsp_amd
    g = golden("g_synth_bursts")
    h = dsp_amd.rrc_filter(4, 10, 0.35)
    X = g["x"]
    Y32 = np.stack([oracle.filter_samples(X[b], h).astype(np.complex64) for b in range(X.shape[0])])
    r = dsp_amd.sync_chain_batch(Y32, 4, 0.007, 0.0, 0.03, 1)
    assert np.array_equal(r["n_out"], g["n_recB"])
    assert np.array_equal(r["dec"][:, :2048], g["decB"])                     # all 32 bursts, bit-exact
    for b in range(6):
        n = int(g["n_recB"][b])
        assert np.array_equal(r["rec"][b, :n], g["recB"][b, :n].astype(np.complex64))
        assert np.array_equal(r["mu"][b, :n - 1], g["muB"][b, :n - 1].astype(np.float32))
        assert np.array_equal(r["err"][b, :n - 1], g["terrB"][b, :n - 1].astype(np.float32))
        assert np.max(np.abs(r["freq"][b, :n] - g["fllB_freq"][b, :n])) < 1e-9    # float32 output
        assert np.max(np.abs(r["sig"][b, :n] - g["fllB_out"][b, :n])) < 1e-6
    for b in range(X.shape[0]):
        o = _chain_oracle(oracle, Y32[b])
        n = o["n"]
        assert np.array_equal(r["dec"][b, :n], o["dec"])
        assert np.array_equal(r["mu"][b, :n - 1], o["mu"].astype(np.float32))
        assert np.array_equal(r["freq"][b, :n], o["freq"].astype(np.float32)) or \
            np.max(np.abs(r["freq"][b, :n] - o["freq"])) < 1e-9
    # float64 path on the same bursts (complex128 in -> float64 out): tight FLL tolerance
    Y = np.stack([oracle.filter_samples(X[b], h) for b in range(8)])
    r64 = dsp_amd.sync_chain_batch(Y, 4, 0.007, 0.0, 0.03, 1)
    for b in range(6):
        n = int(g["n_rec"][b])
        assert r64["n_out"][b] == n
        assert np.max(np.abs(r64["rec"][b, :n] - g["rec"][b, :n])) < 1e-12   # y differs from golden y by FIR order
        assert np.max(np.abs(r64["freq"][b, :n] - g["fll_freq"][b, :n])) < 1e-12
        assert np.array_equal(r64["dec"][b, :n], g["dec"][b, :n])


def test_chain_ragged_and_scale(ctx, oracle):
    """Ragged burst lengths (incl. 0 and < 2*sps), partial tile (B not a multiple of 64),
    per-burst input scale."""
    import dsp_amd
    rng = np.random.default_rng(5)
    B, N = 70, 1536
    h = dsp_amd.rrc_filter(4, 10, 0.35)
    from dsp_amd.synth import synth_burst
    X = np.stack([oracle.filter_samples(synth_burst(500 + b, h, n_sym=N // 4), h).astype(np.complex64)
                  for b in range(B)])
    lens = rng.integers(0, N + 1, B).astype(np.int32)
    lens[:6] = [0, 1, 7, 8, 9, N]
    scale = rng.uniform(0.5, 2.0, B)
    r = dsp_amd.sync_chain_batch(X, 4, 0.01, 0.0, 0.03, 1, lens=lens, scale=scale)
    for b in range(B):
        L = int(lens[b])
        if L == 0:
            assert r["n_out"][b] == 0
            continue
        o = oracle.sync_chain_c64(X[b, :L], sps=4, alpha=0.01, Bn=0.03, y=1.0, scale=float(scale[b]))
        n = o["n"]
        assert r["n_out"][b] == n, (b, L)
        assert np.array_equal(r["dec"][b, :n], o["dec"]), b
        assert np.array_equal(r["rec"][b, :n], o["rec"].astype(np.complex64)), b
        assert np.array_equal(r["mu"][b, :n - 1], o["mu"].astype(np.float32)), b
        # outputs of the batched calls are zero beyond n_out[b] (n_out - 1 for err / mu)
        for k in ("dec", "rec", "sig", "freq", "ef"):
            assert np.all(r[k][b, n:] == 0), (k, b)
        for k in ("err", "mu"):
            assert np.all(r[k][b, max(n - 1, 0):] == 0), (k, b)


def test_fll_batch_state_continuation(ctx, oracle):
    """fll_batch with i0/state continues a loop exactly where a first call stopped."""
    import dsp_amd
    g = golden("g_synth_bursts")
    rec = g["recB"][:4, :2048]
    full = dsp_amd.fll_batch(rec, 0.03, 1)
    a = dsp_amd.fll_batch(rec[:, :1000], 0.03, 1, want_state=True)
    b = dsp_amd.fll_batch(rec[:, 1000:], 0.03, 1, i0=1000, state=a[5], want_state=True)
    assert np.array_equal(np.concatenate([a[2], b[2]], axis=1), full[2])
    assert np.array_equal(np.concatenate([a[4], b[4]], axis=1), full[4])
    for i in range(4):
        o = oracle.fll_func(rec[i], 0.03, 1)
        assert np.max(np.abs(full[2][i] - o[2])) < FREQ_TOL
        assert np.array_equal(full[4][i], oracle.quadrants(o[0]))


@pytest.mark.parametrize("sps", [2, 3, 5, 6, 7, 8])
@pytest.mark.parametrize("c64", [True, False])
def test_spec_segments_odd_sps(ctx, oracle, sps, c64):
    """Full tiles of equally long bursts at samples-per-symbol other than 4, so that the steady path and its
    speculative segments run (ragged tiles never get there): gardner2.py's idx_mid = k - sps // 2 + mu lies
    sps - sps // 2 rows behind idx_early, which is sps // 2 for even sps only.  (The round-3 soak run,
    DSP_FUZZ_SCALE=20, found the segment reading the mid instant one row early at sps 3: its trial 155 was the
    first of the sweep with an odd sps AND a tile that reaches the steady path.)  Two full tiles and a partly
    filled one (passenger lanes), per-burst scale, both sample types; Gardner outputs bit-exact, decisions
    exact and the frequency <= 1e-12 / 1e-9 inside the FLL's stable horizon."""
    import dsp_amd
    from test_gpu_fuzz import _burst
    rng = np.random.default_rng(900 + sps)
    B, nsym, alpha, mu0, Bn = 130, 300, 0.05, 0.37, 0.03
    N = nsym * sps
    X = np.stack([_burst(rng, N, sps, 1) for _ in range(B)])
    if c64:
        X = X.astype(np.complex64)
    scale = rng.uniform(0.5, 2.0, B)
    got = dsp_amd.sync_chain_batch(X, sps, alpha, mu0, Bn, 1, scale=scale, outputs="all", ctx=ctx)
    cdt, rdt = (np.complex64, np.float32) if c64 else (np.complex128, np.float64)
    horizon = int(0.9 / (np.pi * 4 * Bn / 1000 / 1.25))
    for b in range(B):
        x = X[b].astype(np.complex128) * scale[b]
        rec, err, mu = oracle.gardner_timing_recovery(x, sps, alpha, mu0)
        n = len(rec)
        assert got["n_out"][b] == n, b
        assert np.array_equal(got["rec"][b, :n], rec.astype(cdt)), b
        assert np.array_equal(got["mu"][b, :n - 1], mu.astype(rdt)), b
        assert np.array_equal(got["err"][b, :n - 1], err.astype(rdt)), b
        sl, cf, fr, ef = oracle.fll_func(rec, Bn, 1)
        H = min(n, horizon)
        assert np.array_equal(got["dec"][b, :H], oracle.quadrants(sl[:H])), b
        assert np.max(np.abs(got["freq"][b, :H] - fr[:H])) < (1e-9 if c64 else 1e-12), b


@pytest.mark.parametrize("c64", [True, False])
def test_fll_blocks_with_axis_and_zero_samples(ctx, oracle, c64):
    """A full tile on the steady path in which some lanes carry samples the FLL wave's fast step does not
    handle (np.sign(0) = 0: a real-valued burst keeps an exactly zero imaginary part through the whole chain
    while the frequency estimate is 0; an all-zero burst; an imaginary-valued one): the block's evidence word
    sends such blocks through the reference-order step, and every lane -- the ordinary ones too -- must come
    out as the oracle has it."""
    import dsp_amd
    from test_gpu_fuzz import _burst
    rng = np.random.default_rng(77)
    sps, nsym, B = 4, 260, 64
    N = sps * nsym
    X = np.stack([_burst(rng, N, sps, 1) for _ in range(B)])
    X[5] = X[5].real
    X[9] = 0
    X[17] = 1j * X[17].imag
    X[33, 400:] = X[33, 400:].real                      # turns degenerate in mid-burst: only while freq is still 0 it stays so
    if c64:
        X = X.astype(np.complex64)
    got = dsp_amd.sync_chain_batch(X, sps, 0.007, 0.0, 0.03, 1, outputs="all", ctx=ctx)
    cdt, rdt = (np.complex64, np.float32) if c64 else (np.complex128, np.float64)
    for b in range(B):
        rec, err, mu = oracle.gardner_timing_recovery(X[b].astype(np.complex128), sps, 0.007, 0.0)
        n = len(rec)
        assert got["n_out"][b] == n, b
        assert np.array_equal(got["rec"][b, :n], rec.astype(cdt)), b
        assert np.array_equal(got["mu"][b, :n - 1], mu.astype(rdt)), b
        sl, cf, fr, ef = oracle.fll_func(rec, 0.03, 1)
        assert np.array_equal(got["dec"][b, :n], oracle.quadrants(sl)), b
        assert np.max(np.abs(got["freq"][b, :n] - fr)) < (1e-9 if c64 else 1e-12), b
        assert np.max(np.abs(got["sig"][b, :n] - sl)) < (1e-6 if c64 else 1e-9) * max(1.0, np.abs(sl).max()), b


@pytest.mark.parametrize("c64", [True, False])
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("sps", [2, 3, 4, 8])
def test_gardner_alone_two_wave_path(ctx, oracle, c64, variant, sps):
    """The timing loops called on their own (gardner_batch: no fll_func behind them) run on the two-wave
    kernel without its FLL -- the fused kernel's timing wave, speculative segments and all, with the second
    wave feeding the ring (sync_kernels.hip part 7).  Two full tiles and a ragged third against the oracle,
    bit for bit; "spec" = 0 (the one-wave kernel's per-symbol path, what such a call ran on before) gives
    the same bits."""
    import dsp_amd
    from test_gpu_fuzz import _burst
    rng = np.random.default_rng(900 + 10 * sps + variant)
    nsym, B = 420, 150
    N = sps * nsym
    X = np.stack([_burst(rng, N, sps, 1) for _ in range(B)])
    lens = np.full(B, N, np.int32)
    lens[128:] = rng.integers(3 * sps, N, B - 128)
    for b in range(128, B):
        X[b, lens[b]:] = 0
    if c64:
        X = X.astype(np.complex64)
    alpha, mu0 = (0.05, 0.37) if variant == 0 else (0.004, 0.0)
    runs = {}
    for spec in (1, 0):
        ctx.set_option("spec", spec)
        try:
            runs[spec] = dsp_amd.gardner_batch(X, sps, alpha, mu0, variant=variant, lens=lens, ctx=ctx)
        finally:
            ctx.set_option("spec", 1)
    cdt, rdt = (np.complex64, np.float32) if c64 else (np.complex128, np.float64)
    g_rec, g_err, g_mu, n_out = runs[1]
    for b in range(B):
        x = X[b, :lens[b]].astype(np.complex128)
        if variant == 0:
            rec, err, mu = oracle.gardner_timing_recovery(x, sps, alpha, mu0)
        else:
            rec, err, mu = oracle.gardner_ted(x, sps, alpha)
        n = len(rec)
        assert n_out[b] == n, b
        assert np.array_equal(g_rec[b, :n], rec.astype(cdt)), b
        assert np.array_equal(g_mu[b, :len(mu)], mu.astype(rdt)), b
        assert np.array_equal(g_err[b, :len(err)], err.astype(rdt)), b
    for u, v in zip(runs[1], runs[0]):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("c64", [True, False])
@pytest.mark.parametrize("B", [1, 64, 65, 130])
def test_fll_alone_block_path(ctx, oracle, c64, B):
    """fll_func on its own (fll_batch -> the one-wave kernel's FLL-only form): whole DMA blocks go through
    fll_ring_block (fll_step_fast straight from the LDS ring) while every lane is inside its burst, the
    padding lanes of a partly filled tile run as clones of its longest burst (B = 1: 63 of them), the ends
    of ragged bursts take the per-symbol step.  Against the oracle: decisions equal, frequency to the FLL's
    bar, derotated samples; with an index offset and a start state; and the A/B switch (spec = 0: per-symbol
    step throughout) agrees to the same bars."""
    import dsp_amd
    from test_gpu_fuzz import _burst
    rng = np.random.default_rng(500 + B)
    M, Bn = 700, 0.02
    S = np.stack([_burst(rng, M, 1, 1) for _ in range(B)])
    lens = np.full(B, M, np.int32)
    if B > 64:
        lens[64:] = rng.integers(3, M, B - 64)             # the second tile is ragged
        for b in range(64, B):
            S[b, lens[b]:] = 0
    if c64:
        S = S.astype(np.complex64)
    kp, ki = dsp_amd.fll_gains(Bn, 1)
    i0 = rng.integers(0, 50, B).astype(np.float64)
    st = np.stack([rng.uniform(-1e-3, 1e-3, B), rng.uniform(-1e-4, 1e-4, B)], axis=1)
    runs = {}
    for spec in (1, 0):
        ctx.set_option("spec", spec)
        try:
            runs[spec] = (dsp_amd.fll_batch(S, Bn, 1, lens=lens, ctx=ctx),
                          dsp_amd.fll_batch(S, Bn, 1, lens=lens, i0=i0, state=st, want_state=True, ctx=ctx))
        finally:
            ctx.set_option("spec", 1)
    ftol, stol = (1e-9, 1e-6) if c64 else (FREQ_TOL, SIG_TOL)
    for spec, (plain, cont) in runs.items():
        for b in range(B):
            n = int(lens[b])
            sl, cf, fr, ef = oracle.fll_func(S[b, :n].astype(np.complex128), Bn, 1)
            assert np.array_equal(plain[4][b, :n], oracle.quadrants(sl)), (spec, b)
            assert np.max(np.abs(plain[2][b, :n] - fr)) < ftol, (spec, b)
            assert np.max(np.abs(plain[0][b, :n] - sl)) < stol * max(1.0, np.abs(sl).max()), (spec, b)
            assert not plain[0][b, n:].any() and not plain[4][b, n:].any(), (spec, b)    # rows past the end read zero
    for b in range(B):                                       # offset + start state: the two paths against each other
        n = int(lens[b])
        assert np.array_equal(runs[1][1][4][b, :n], runs[0][1][4][b, :n]), b
        assert np.max(np.abs(runs[1][1][2][b, :n] - runs[0][1][2][b, :n])) < ftol, b
        assert np.max(np.abs(runs[1][1][5][b] - runs[0][1][5][b])) < ftol, b


@pytest.mark.parametrize("c64", [True, False])
def test_fll_alone_blocks_with_axis_and_zero_samples(ctx, oracle, c64):
    """The FLL-only block path's evidence word: a real-valued burst (np.sign(0) = 0 while the estimate is 0),
    an all-zero one and an imaginary-valued one inside a full tile send their blocks through the
    reference-order step; every lane comes out as the oracle has it."""
    import dsp_amd
    from test_gpu_fuzz import _burst
    rng = np.random.default_rng(78)
    B, M = 64, 400
    S = np.stack([_burst(rng, M, 1, 1) for _ in range(B)])
    S[5] = S[5].real
    S[9] = 0
    S[17] = 1j * S[17].imag
    S[33, 200:] = S[33, 200:].real
    if c64:
        S = S.astype(np.complex64)
    got = dsp_amd.fll_batch(S, 0.03, 1, ctx=ctx)
    for b in range(B):
        sl, cf, fr, ef = oracle.fll_func(S[b].astype(np.complex128), 0.03, 1)
        assert np.array_equal(got[4][b], oracle.quadrants(sl)), b
        assert np.max(np.abs(got[2][b] - fr)) < (1e-9 if c64 else FREQ_TOL), b
        assert np.max(np.abs(got[0][b] - sl)) < (1e-6 if c64 else SIG_TOL) * max(1.0, np.abs(sl).max()), b


@pytest.mark.parametrize("sps", [10, 12, 16, 25, 64])
def test_large_sps(ctx, oracle, sps):
    """Samples-per-symbol counts beyond the sweeps' 1 ... 8 (the C-ABI takes up to 1024): the ring plan shrinks
    its block to one or two symbols and then gives the ring up (rows straight from global memory); a full
    tile and a ragged one, against the oracle with the usual bars."""
    import dsp_amd
    from test_gpu_fuzz import _burst
    rng = np.random.default_rng(300 + sps)
    nsym, B = 90, 70
    N = sps * nsym
    X = np.stack([_burst(rng, N, sps, 1) for _ in range(B)])
    lens = np.full(B, N, np.int32)
    lens[64:] = rng.integers(2 * sps, N, B - 64)
    for b in range(64, B):
        X[b, lens[b]:] = 0
    got = dsp_amd.sync_chain_batch(X, sps, 0.02, 0.0, 0.03, 1, lens=lens, outputs="all", ctx=ctx)
    for b in range(B):
        rec, err, mu = oracle.gardner_timing_recovery(X[b, :lens[b]], sps, 0.02, 0.0)
        n = len(rec)
        assert got["n_out"][b] == n, b
        assert np.array_equal(got["rec"][b, :n], rec), b
        assert np.array_equal(got["mu"][b, :n - 1], mu), b
        assert np.array_equal(got["err"][b, :n - 1], err), b
        sl, cf, fr, ef = oracle.fll_func(rec, 0.03, 1)
        assert np.array_equal(got["dec"][b, :n], oracle.quadrants(sl)), b
        assert np.max(np.abs(got["freq"][b, :n] - fr)) < 1e-12, b


@pytest.mark.parametrize("sps,delta", [(2, 0.002), (4, 0.002), (4, 0.05), (6, 0.05), (8, 0.002), (3, 0.002), (4, 0.8)])
def test_gardner_ted_segments_full_tiles(ctx, oracle, sps, delta):
    """gardner_ted (hw1/gardner.py:45-72) on full tiles: the speculative segments of the timing wave serve even
    samples-per-symbol counts (round 3; sps 2 is hw1's own), odd ones keep the per-symbol code, and a gain of
    0.8 drives mu into the clip of :69, which a segment's evidence must hand to the per-symbol code.  Two full
    tiles, a partly filled one and one ragged burst; outputs bit-exact against the oracle, decisions exact and
    frequency <= 1e-12 inside the FLL's stable horizon."""
    import dsp_amd
    from test_gpu_fuzz import _burst
    rng = np.random.default_rng(1200 + sps)
    B, nsym = 131, 280
    N = nsym * sps
    X = np.stack([_burst(rng, N, sps, 1) for _ in range(B)])
    lens = np.full(B, N, np.int32)
    lens[130] = N // 2
    X[130, lens[130]:] = 0
    got = dsp_amd.sync_chain_batch(X, sps, delta, 0.0, 0.03, 1, variant=1, lens=lens, outputs="all", ctx=ctx)
    horizon = int(0.9 / (np.pi * 4 * 0.03 / 1000 / 1.25))
    for b in range(B):
        y, err, mu = oracle.gardner_ted(X[b, :lens[b]], sps, delta)
        n = len(y)
        assert got["n_out"][b] == n, b
        assert np.array_equal(got["rec"][b, :n], y), b
        assert np.array_equal(got["mu"][b, :n], mu), b
        assert np.array_equal(got["err"][b, :n], err), b
        sl, cf, fr, ef = oracle.fll_func(y, 0.03, 1)
        H = min(n, horizon)
        assert np.array_equal(got["dec"][b, :H], oracle.quadrants(sl[:H])), b
        assert np.max(np.abs(got["freq"][b, :H] - fr[:H])) < 1e-12, b
