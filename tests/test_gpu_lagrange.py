"""a9: lagrange_resample (hw1/test_lagrange_resampling.py:34-72) on the device, and the timing
loop with the Lagrange interpolator in place of the linear one (interp_half_len > 0, an
extension: the reference's Gardner only ever interpolates linearly)."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_lagrange_resample_golden(ctx):
    import dsp_amd
    g = golden("g_lagrange")
    clean = g["clean"]
    n = len(clean)
    mis = dsp_amd.lagrange_resample(clean, np.arange(n, dtype=float) + 0.35, 6, ctx=ctx)
    assert mis.dtype == clean.dtype and np.max(np.abs(mis - g["mis"])) < 1e-15
    cor = dsp_amd.apply_fractional_shift(g["mis"], -0.35, 6, ctx=ctx)
    assert np.max(np.abs(cor - g["cor"])) < 1e-15
    arb = dsp_amd.lagrange_resample(clean, g["pos"], 4, ctx=ctx)
    assert np.max(np.abs(arb - g["arb"])) < 1e-15
    assert np.array_equal(arb[:4], g["arb"][:4])            # exact-node shortcut


def test_lagrange_resample_matches_oracle(oracle, ctx):
    import dsp_amd
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(3000) + 1j * rng.standard_normal(3000))
    pos = rng.uniform(-3, 3003, 5000)                       # includes clipped edges
    pos[:10] = np.arange(10)                                # exact nodes
    for h in (1, 2, 4, 8):
        ref = oracle.lagrange_resample(x, pos, h)
        got = dsp_amd.lagrange_resample(x, pos, h, ctx=ctx)
        assert np.max(np.abs(got - ref)) < 1e-13 * max(1.0, np.max(np.abs(ref)))
    x32 = x.astype(np.complex64)
    got = dsp_amd.lagrange_resample(x32, pos, 4, ctx=ctx)
    assert got.dtype == np.complex64
    assert np.max(np.abs(got - oracle.lagrange_resample(x32, pos, 4))) < 1e-6
    xr = x.real.copy()
    got = dsp_amd.lagrange_resample(xr, pos, 4, ctx=ctx)
    assert got.dtype == np.float64 and np.max(np.abs(got - oracle.lagrange_resample(xr, pos, 4))) < 1e-12


def gardner2_lagrange_restated(oracle, x, sps, alpha, h):
    """gardner2.py:52-100 with interpolate() replaced by the (2h+1)-point Lagrange interpolator
    (same bounds rule: 0.0 outside [0, len))."""
    x = np.asarray(x, np.complex128)
    n = len(x)

    def interp(idx):
        ii = int(idx)
        if ii < 0 or ii >= n:
            return 0.0
        return oracle.lagrange_resample(x, np.array([idx]), h)[0]

    rec, err, mus = [x[0]], [], []
    mu, k = 0.0, sps
    while k <= n - sps:
        while mu >= sps:
            mu -= sps; k += sps
        while mu < -sps:
            mu += sps; k -= sps
        ie, im, il = k - sps + mu, k - sps // 2 + mu, k + mu
        if ie < 0 or il + 1 >= n:
            break
        xe, xm, xl = interp(ie), interp(im), interp(il)
        e = float(np.real(xm * np.conj(xl - xe)))
        mu -= alpha * e
        rec.append(interp(k + mu)); err.append(e); mus.append(mu)
        k += sps
    return np.array(rec), np.array(err), np.array(mus)


def test_gardner_with_lagrange_interpolator(oracle, ctx):
    import dsp_amd
    g = golden("g_synth_bursts")
    h = oracle.rrc_filter(4, 10, 0.35)
    y = oracle.filter_samples(g["x"][0], h)[:1600]
    for half in (1, 2, 4):
        rec, err, mu = gardner2_lagrange_restated(oracle, y, 4, 0.007, half)
        g_rec, g_err, g_mu, n_out = dsp_amd.gardner_batch(y[None, :], 4, 0.007, 0.0, interp_half_len=half, ctx=ctx)
        n = int(n_out[0])
        assert n == len(rec)
        assert np.max(np.abs(g_mu[0, :n - 1] - mu)) < 1e-12
        assert np.max(np.abs(g_err[0, :n - 1] - err)) < 1e-12
        assert np.max(np.abs(g_rec[0, :n] - rec)) < 1e-12
    # and through the fused chain: decisions with the higher-order interpolator stay the same
    # symbols as with the linear one on a clean burst
    a = dsp_amd.sync_chain_batch(y[None, :], 4, 0.007, 0.0, 0.03, 1, interp_half_len=2, ctx=ctx)
    b = dsp_amd.sync_chain_batch(y[None, :], 4, 0.007, 0.0, 0.03, 1, ctx=ctx)
    na = int(a["n_out"][0])
    assert na == int(b["n_out"][0])
    assert np.mean(a["dec"][0, 50:na] == b["dec"][0, 50:na]) > 0.99


@pytest.mark.parametrize("halves,picks", [((1, 2, 4), list(range(0, 70, 9)) + [1, 2, 3, 63, 64, 69]),
                                          ((3, 6, 8), [0, 1, 2, 3, 36, 63, 64, 69]),
                                          ((5, 7), [0, 2, 3, 37, 63, 64, 69])])
def test_lagrange_two_wave_kernel_full_tiles_ragged(oracle, ctx, halves, picks):
    """interp_half_len 1, 2, 4 (round 2) and 3, 5, 6, 7, 8 (round 3: 6 is the order of the reference's own worked
    Lagrange scenario, hw1/test_lagrange_resampling.py:157) through the fused chain (both loops -> the two-wave kernel, whose timing
    wave interpolates out of the LDS ring with product-form Lagrange basis values): full tiles of
    bursts with ragged lengths, every burst against the restated loop to 1e-12 -- the per-lane code
    (barycentric form) serves burst starts, ends and idle lanes, the straight-line code the rest --
    and the one-wave kernel (split = 0) within 1e-12 of it too."""
    import dsp_amd
    g = golden("g_synth_bursts")
    h = oracle.rrc_filter(4, 10, 0.35)
    Y = np.stack([oracle.filter_samples(g["x"][b % 32], h) for b in range(70)])
    Y = np.roll(Y, 3, axis=1) * (1.0 + 0.01 * np.arange(70))[:, None]       # 70 distinct bursts
    N = 2400
    Y = Y[:, :N]
    rng = np.random.default_rng(41)
    lens = rng.integers(1200, N + 1, 70).astype(np.int32)
    lens[:4] = [N, N - 1, 1200, 1777]
    for half in halves:
        r = dsp_amd.sync_chain_batch(Y, 4, 0.007, 0.0, 0.03, 1, interp_half_len=half, lens=lens,
                                     outputs=("rec", "err", "mu", "dec", "freq"), ctx=ctx)
        ctx.set_option("split", 0)
        try:
            r1 = dsp_amd.sync_chain_batch(Y, 4, 0.007, 0.0, 0.03, 1, interp_half_len=half, lens=lens,
                                          outputs=("rec", "err", "mu", "dec", "freq"), ctx=ctx)
        finally:
            ctx.set_option("split", -1)
        assert np.array_equal(r["n_out"], r1["n_out"])
        for b in picks:
            L = int(lens[b])
            rec, err, mu = gardner2_lagrange_restated(oracle, Y[b, :L], 4, 0.007, half)
            n = int(r["n_out"][b])
            assert n == len(rec), (half, b)
            assert np.max(np.abs(r["mu"][b, :n - 1] - mu)) < 1e-12, (half, b)
            assert np.max(np.abs(r["err"][b, :n - 1] - err)) < 1e-12, (half, b)
            assert np.max(np.abs(r["rec"][b, :n] - rec)) < 1e-12, (half, b)
            assert np.max(np.abs(r1["mu"][b, :n - 1] - mu)) < 1e-12, (half, b)
            out = oracle.fll_func(rec, 0.03, 1)
            assert np.array_equal(r["dec"][b, :n], oracle.quadrants(out[0])), (half, b)
            assert np.max(np.abs(r["freq"][b, :n] - out[2])) < 1e-12
            assert not r["rec"][b, n:].any()


def gardner_ted_lagrange_restated(oracle, x, sps, delta_mu, h):
    """hw1/gardner.py:45-72 with interpolate_sample replaced by the (2h+1)-point Lagrange interpolator
    (lagrange_resample's own clipping at the ends)."""
    x = np.asarray(x, np.complex128)
    n = len(x)

    def interp(idx):
        return oracle.lagrange_resample(x, np.array([idx]), h)[0]

    ys, errs, mus = [], [], []
    mu = 0.0
    for k in range((n - 2 * sps) // sps):
        i_k, i_n, i_m = k * sps + mu, (k + 1) * sps + mu, k * sps + mu + sps / 2.0
        if i_n + 1 >= n or i_m + 1 >= n:
            break
        xk, xn, xm = interp(i_k), interp(i_n), interp(i_m)
        e = float(np.real(xm * np.conj(xn - xk)))
        ys.append(xk); errs.append(e); mus.append(mu)
        mu = float(np.clip(mu + delta_mu * e, -sps, sps))
    return np.array(ys), np.array(errs), np.array(mus)


def test_gardner_ted_with_lagrange_on_the_two_wave_kernel(oracle, ctx):
    """The gardner_ted variant with the Lagrange interpolator and the FLL behind it: two waves per tile since
    round 3 (timing wave: per-lane ted_iter + lagrange_at; FLL wave in parallel), every order the
    two-wave kernel is built for, against the restated loop and the one-wave kernel."""
    import dsp_amd
    g = golden("g_synth_bursts")
    h = oracle.rrc_filter(4, 10, 0.35)
    Y = np.stack([oracle.filter_samples(g["x"][b % 32], h)[:1500] for b in range(66)])
    Y = Y * (1.0 + 0.02 * np.arange(66))[:, None]
    lens = np.full(66, 1500, np.int32)
    lens[:3] = [1500, 1203, 900]
    for half in (1, 2, 3, 4, 5, 6, 7, 8):
        r = dsp_amd.sync_chain_batch(Y, 4, 0.002, 0.0, 0.03, 1, variant=1, interp_half_len=half,
                                     lens=lens, outputs=("rec", "err", "mu", "dec", "freq"), ctx=ctx)
        ctx.set_option("split", 0)
        try:
            r1 = dsp_amd.sync_chain_batch(Y, 4, 0.002, 0.0, 0.03, 1, variant=1, interp_half_len=half,
                                          lens=lens, outputs=("rec", "err", "mu", "dec", "freq"), ctx=ctx)
        finally:
            ctx.set_option("split", -1)
        for k in ("n_out", "rec", "err", "mu", "dec"):
            assert np.array_equal(r[k], r1[k]), (half, k)          # same per-lane code on either kernel
        for b in (0, 1, 2, 65):
            y, err, mu = gardner_ted_lagrange_restated(oracle, Y[b, :int(lens[b])], 4, 0.002, half)
            n = int(r["n_out"][b])
            assert n == len(y), (half, b)
            assert np.max(np.abs(r["mu"][b, :n] - mu)) < 1e-12 and np.max(np.abs(r["err"][b, :n] - err)) < 1e-12
            assert np.max(np.abs(r["rec"][b, :n] - y)) < 1e-12
            out = oracle.fll_func(y, 0.03, 1)
            assert np.array_equal(r["dec"][b, :n], oracle.quadrants(out[0])), (half, b)
