"""Randomised parity sweep of the fused Gardner + FLL kernels against the oracle (fixed seed): every
trial draws the variant, samples per symbol, burst count (partly filled tiles), burst lengths
(ragged, zero, shorter than two symbols), loop constants, start value, per-burst scale, output
subset and which kernel layout runs (one wave or two per tile), on complex128 input so that the
Gardner outputs must be bit-exact and the FLL frequency within 1e-12 inside its stable horizon."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FREQ_TOL = 1e-12


def _burst(rng, n, sps, kind):
    if kind == 0 or n < 4 * sps:                      # noise only
        return rng.standard_normal(n) + 1j * rng.standard_normal(n)
    nsym = n // sps + 2
    sym = np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, nsym)))
    t = np.arange(n) / sps + rng.uniform(-0.5, 0.5) / sps * sps * rng.uniform(0, 1)
    # smooth pulse shaping by linear interpolation of a raised-cosine-ish pulse train
    k = np.floor(t).astype(int)
    f = t - k
    x = (sym[np.clip(k, 0, nsym - 1)] * (0.5 + 0.5 * np.cos(2 * np.pi * (f - 0.5)))
         + 0.3 * sym[np.clip(k + 1, 0, nsym - 1)] * f)
    x = x * np.exp(1j * (2 * np.pi * rng.uniform(-1e-3, 1e-3) * np.arange(n) + rng.uniform(0, 6.28)))
    return x + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))


@pytest.mark.parametrize("trial", range(64))
def test_fuzz_chain(ctx, oracle, trial):
    import dsp_amd
    rng = np.random.default_rng(1000 + trial)
    sps = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8]))
    variant = int(rng.integers(0, 2))                       # 0 gardner2, 1 gardner_ted
    B = int(rng.choice([1, 3, 63, 64, 65, 130]))
    N = int(rng.choice([0, 1, 2 * sps - 1, 2 * sps, 37, 400, 1501]))
    if N == 0:
        N = 5
    # (gains near 0.3 can make gardner2.py's own loop run forever: k -= sps undoes k += sps; the
    #  kernels then report DSP_E_CAPACITY, covered by test_edges)
    alpha = float(rng.choice([0.007, 0.002, 0.05, 0.1]))
    mu0 = float(rng.choice([0.0, 0.0, 0.37, -0.9, 1.5])) if variant == 0 else 0.0
    Bn = float(rng.choice([0.03, 0.001, 0.2]))
    lens = rng.integers(0, N + 1, B).astype(np.int32)
    lens[rng.integers(0, B)] = N
    if B > 2:
        lens[1] = 0
        lens[2] = min(N, 1)
    X = np.zeros((B, N), np.complex128)
    for b in range(B):
        X[b, :lens[b]] = _burst(rng, int(lens[b]), sps, int(rng.integers(0, 3)))
    c64 = bool(rng.integers(0, 2))                          # complex64 in -> float32 outputs, float64 arithmetic
    if c64:
        X = X.astype(np.complex64)
    scale = rng.uniform(0.5, 2.0, B) if rng.integers(0, 2) else None
    split = int(rng.choice([-1, 0, 1]))
    outs = "all" if rng.integers(0, 2) else ("dec", "mu", "freq", "rec")
    ctx.set_option("split", split)
    try:
        got = dsp_amd.sync_chain_batch(X, sps, alpha, mu0, Bn, 1, variant=variant, lens=lens, scale=scale,
                                       outputs=outs, ctx=ctx)
    finally:
        ctx.set_option("split", -1)
    kp = 4 * Bn / 1000 / 1.25
    horizon = int(0.9 / (np.pi * kp))
    for b in list(range(min(B, 6))) + ([B - 1] if B > 6 else []) + ([64] if B > 64 else []):
        x = X[b, :lens[b]].astype(np.complex128) * (scale[b] if scale is not None else 1.0)
        if variant == 0:
            if lens[b] == 0:
                assert got["n_out"][b] == 0
                continue
            rec, err, mu = oracle.gardner_timing_recovery(x, sps, alpha, mu0)
            n, nm = len(rec), len(rec) - 1
        else:
            rec, err, mu = oracle.gardner_ted(x, sps, alpha) if lens[b] else (np.zeros(0, complex), np.zeros(0), np.zeros(0))
            n = nm = len(rec)
        assert got["n_out"][b] == n, (b, got["n_out"][b], n)
        cdt, rdt = (np.complex64, np.float32) if c64 else (np.complex128, np.float64)
        assert got["rec"].dtype == cdt and got["mu"].dtype == rdt
        assert np.array_equal(got["rec"][b, :n], rec.astype(cdt))          # float64 loop, one rounding on store
        assert np.array_equal(got["mu"][b, :nm], mu.astype(rdt))
        assert not got["rec"][b, n:].any()
        if n:
            sl, cf, fr, ef = oracle.fll_func(rec, Bn, 1)                   # the loop is fed the float64 symbols
            H = min(n, horizon)
            assert np.max(np.abs(got["freq"][b, :H] - fr[:H])) < (1e-9 if c64 else FREQ_TOL)
            assert np.array_equal(got["dec"][b, :H], oracle.quadrants(sl[:H]))
            if outs == "all":
                assert np.array_equal(got["err"][b, :nm], err.astype(rdt))
                assert np.max(np.abs(got["sig"][b, :H] - sl[:H])) < (1e-6 if c64 else 1e-9) * max(1.0, np.abs(sl[:H]).max())
