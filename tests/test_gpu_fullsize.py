"""Full-size runs of the device pipeline (BASELINE.json configs 3 and 4: 4096 and 65536
synthetic bursts): oracle parity on sampled tiles plus size-independent properties on every
burst (symbol counts, decisions == transmitted symbols up to the QPSK phase ambiguity,
idempotence of a second launch)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 424242
ROT = np.array([1, 3, 0, 2])      # quadrant index after a +90 degree turn


def _match_rates(dec, seed, b0, skip=400):
    from dsp_amd.synth import device_symbols
    out = np.empty(dec.shape[0])
    for b in range(dec.shape[0]):
        tx = device_symbols(seed, b0 + b, dec.shape[1])
        best, d = 0.0, dec[b]
        for _ in range(4):
            best = max(best, float(np.mean(d[skip:] == tx[skip:])))
            d = ROT[d]
        out[b] = best
    return out


@pytest.mark.parametrize("B,split", [(4096, 1), (4096, 0), (32768, -1), (65536, -1)])
def test_pipeline_fullsize(ctx, oracle, B, split):
    from dsp_amd.pipeline import BurstPipeline
    ctx.set_option("split", split)
    try:
        pipe = BurstPipeline(ctx, B, 2048, 4)
        pipe.generate(SEED, b0=0)
        pipe.matched_filter(free_raw=True)
        pipe.run()
        n_out = pipe.n_out()
        assert (n_out == 2048).all()
        # properties on every burst
        step = 8192
        worst = 1.0
        for lo in range(0, B, step):
            dec = pipe.fetch("dec", lo, min(B, lo + step))[:, :2048]
            worst = min(worst, _match_rates(dec[::(1 if B <= 4096 else 16)], SEED, lo)[...].min()
                        if B <= 4096 else
                        min(_match_rates(dec[i:i + 1], SEED, lo + i)[0] for i in range(0, dec.shape[0], 16)))
        assert worst > 0.995
        # oracle parity on the first, a middle and the last tile
        for t in (0, (B // 64) // 2, B // 64 - 1):
            X = pipe.fetch_input(t * 64, t * 64 + 64)
            inv = pipe.fetch_inv_std(t * 64, t * 64 + 64)
            dec = pipe.fetch("dec", t * 64, t * 64 + 64)
            mu = pipe.fetch("mu", t * 64, t * 64 + 64)
            freq = pipe.fetch("freq", t * 64, t * 64 + 64)
            rec = pipe.fetch("rec", t * 64, t * 64 + 64)
            for b in range(0, 64, 3):
                o = oracle.sync_chain_c64(X[b], scale=inv[b])
                n = o["n"]
                assert n == 2048
                assert np.array_equal(dec[b, :n], o["dec"])
                assert np.array_equal(mu[b, :n - 1], o["mu"].astype(np.float32))
                assert np.array_equal(rec[b, :n], o["rec"].astype(np.complex64))
                assert np.max(np.abs(freq[b, :n] - o["freq"])) < 1e-9
        # idempotence: a second launch over the same input rewrites identical outputs
        d1 = pipe.fetch("dec", 0, 64)[:, :2048]             # (rows past the 2048 symbols are scratch)
        f1 = pipe.fetch("freq", 0, 64)[:, :2048]
        pipe.run()
        assert np.array_equal(pipe.fetch("dec", 0, 64)[:, :2048], d1)
        assert np.array_equal(pipe.fetch("freq", 0, 64)[:, :2048], f1)
        pipe.close()
    finally:
        ctx.set_option("split", -1)


def test_every_burst_of_the_bench_batch_against_the_oracle(ctx, oracle):
    """BASELINE.json config 4 on one GPU, the bench line's workload: ALL 65 536 bursts (not a sample of
    tiles) through the fused kernel and through the CPU oracle on the same matched-filtered complex64
    input and the same 1/std -- decisions, symbol counts, the timing trajectory and the recovered
    samples bit-exact (after the device's float32 rounding of the outputs), the frequency trajectory
    <= 1e-9.  The oracle runs one burst per call on a pool of host threads (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    from dsp_amd.pipeline import BurstPipeline
    B, step = 65536, 4096
    pipe = BurstPipeline(ctx, B, 2048, 4)
    pipe.generate(SEED, b0=0)
    pipe.matched_filter(free_raw=True)
    pipe.run()
    n_out = pipe.n_out()
    try:
        workers = min(16, len(os.sched_getaffinity(0)))
    except AttributeError:
        workers = min(16, os.cpu_count() or 1)
    bad = []
    worst_f = 0.0

    with ThreadPoolExecutor(workers) as ex:
        for lo in range(0, B, step):
            X = pipe.fetch_input(lo, lo + step)
            inv = pipe.fetch_inv_std(lo, lo + step)
            dec = pipe.fetch("dec", lo, lo + step)
            mu = pipe.fetch("mu", lo, lo + step)
            rec = pipe.fetch("rec", lo, lo + step)
            freq = pipe.fetch("freq", lo, lo + step)

            def check(b):
                o = oracle.sync_chain_c64(X[b], scale=inv[b])
                n = o["n"]
                ok = (n == n_out[lo + b] and np.array_equal(dec[b, :n], o["dec"])
                      and np.array_equal(mu[b, :n - 1], o["mu"].astype(np.float32))
                      and np.array_equal(rec[b, :n], o["rec"].astype(np.complex64)))
                return ok, float(np.max(np.abs(freq[b, :n] - o["freq"])))

            for b, (ok, df) in enumerate(ex.map(check, range(step))):
                worst_f = max(worst_f, df)
                if not ok:
                    bad.append(lo + b)
    pipe.close()
    assert not bad, (len(bad), bad[:8])
    assert worst_f < 1e-9, worst_f


def test_device_generator_statistics(ctx):
    """The device generator follows the 8(d) recipe: unit symbol energy, Es/N0 = 20 dB, CFO and
    delay inside their ranges, reproducible from (seed, burst)."""
    import dsp_amd
    from dsp_amd.pipeline import BurstPipeline
    from dsp_amd.synth import device_params
    pipe = BurstPipeline(ctx, 128, 2048, 4)
    pipe.generate(SEED, b0=7)
    raw = np.empty((128, 8192), np.complex64)
    ctx.d2h(raw, pipe.d_raw)
    p = np.mean(np.abs(raw) ** 2)
    assert abs(p - (0.25 + 0.01)) < 0.01            # Es/sps + N0
    pipe.generate(SEED, b0=7)
    raw2 = np.empty_like(raw)
    ctx.d2h(raw2, pipe.d_raw)
    assert np.array_equal(raw, raw2)
    tau, f, phi = device_params(SEED, 7)
    assert -0.5 <= tau <= 0.5 and abs(f) <= 5e-4 and 0 <= phi <= 2 * np.pi
    pipe.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_spare_wave_launch_bit_identical(ctx, oracle, variant):
    """Two workgroups per CU (257..512 tiles): the two-wave kernel is launched with an idle middle
    wave.  Same bits as the plain two-wave launch for every burst and output, both timing-loop variants,
    and oracle parity on sampled bursts (gardner2)."""
    from dsp_amd.pipeline import BurstPipeline
    B = 20032                                   # 313 tiles
    pipe = BurstPipeline(ctx, B, 512, 4, variant=variant)
    pipe.generate(SEED, b0=0)
    pipe.matched_filter(free_raw=True)
    got = {}
    try:
        for sw in (0, 1, -1):                   # plain launch, one idle wave, two idle waves + CU ticket (auto)
            ctx.set_option("spare_wave", sw)
            pipe.run()
            got[sw] = {k: pipe.fetch(k, 0, B)[:, :512 - (k == "mu")] for k in ("dec", "mu", "freq", "rec")}
            got[sw]["n"] = pipe.n_out().copy()
    finally:
        ctx.set_option("spare_wave", -1)
    for sw in (1, -1):
        assert np.array_equal(got[0]["n"], got[sw]["n"])
        for k in ("dec", "mu", "freq", "rec"):
            assert np.array_equal(got[0][k], got[sw][k]), (sw, k)
    if variant == 0:
        for b in (0, 63, 64, 9999, B - 1):
            t0 = b // 64 * 64
            X = pipe.fetch_input(t0, t0 + 64)
            o = oracle.sync_chain_c64(X[b - t0], scale=pipe.fetch_inv_std(b, b + 1)[0])
            assert np.array_equal(got[-1]["dec"][b, :o["n"]], o["dec"])
    pipe.close()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("c64", [True, False])
def test_ring_feeder_wave_launch_bit_identical(ctx, oracle, variant, c64):
    """At most one tile per CU: the fused kernel is launched with THREE live waves per tile -- timing, FLL and a
    third that only feeds the LDS ring (spare_wave 3, automatic there).  Same bits as the plain two-wave launch
    for every burst and output: full tiles, a ragged one and a partly filled one, both timing loops, both sample
    types (complex128 tiles have the 128 KiB ring there; "big_ring" = 0, the 64 KiB one, gives the same bits
    too), and oracle parity on sampled bursts."""
    import dsp_amd
    from test_gpu_fuzz import _burst
    rng = np.random.default_rng(4100 + variant)
    sps, nsym, B = 4, 300, 64 * 3 + 17
    N = sps * nsym
    X = np.stack([_burst(rng, N, sps, 1) for _ in range(B)])
    lens = np.full(B, N, np.int32)
    lens[128:192] = rng.integers(2 * sps, N, 64)               # third tile ragged, fourth partly filled
    for b in range(128, 192):
        X[b, lens[b]:] = 0
    if c64:
        X = X.astype(np.complex64)
    alpha = 0.02 if variant == 0 else 0.004
    scale = rng.uniform(0.5, 2.0, B)
    got = {}
    try:
        for name, sw, br in (("two", 0, 1), ("feeder", -1, 1), ("feeder_small_ring", 3, 0)):
            ctx.set_option("spare_wave", sw)
            ctx.set_option("big_ring", br)
            got[name] = dsp_amd.sync_chain_batch(X, sps, alpha, 0.0, 0.03, 1, variant=variant, lens=lens, scale=scale,
                                                 outputs="all", ctx=ctx)
    finally:
        ctx.set_option("spare_wave", -1)
        ctx.set_option("big_ring", 1)
    for k, v in got["two"].items():
        assert np.array_equal(v, got["feeder"][k]), k
    # another ring size moves the block boundaries, and with them the symbols that take the FLL's restated step
    # (fll_step_fast) and the reference-order one (the blocks around a burst's ends): the timing loop's outputs and
    # the decisions keep their bits, the FLL's float64 outputs agree to its bar
    for k, v in got["two"].items():
        w = got["feeder_small_ring"][k]
        if k in ("sig", "freq", "ef"):
            assert np.max(np.abs(v - w)) < 1e-12 * max(1.0, float(np.max(np.abs(v)))), k
        else:
            assert np.array_equal(v, w), k
    if variant == 0:
        cdt = np.complex64 if c64 else np.complex128
        for b in (0, 63, 130, 191, B - 1):
            x = X[b, :lens[b]].astype(np.complex128) * scale[b]
            rec, err, mu = oracle.gardner_timing_recovery(x, sps, alpha, 0.0)
            n = len(rec)
            assert got["feeder"]["n_out"][b] == n
            assert np.array_equal(got["feeder"]["rec"][b, :n], rec.astype(cdt)), b
            sl, cf, fr, ef = oracle.fll_func(rec, 0.03, 1)
            assert np.array_equal(got["feeder"]["dec"][b, :n], oracle.quadrants(sl)), b


def test_priority_rotation_bit_identical(ctx):
    """Three tiles per CU and more: the two-wave kernel rotates its waves' priorities (prio_period,
    automatic from there on).  Scheduling only -- every output has the bits of the launch without it."""
    from dsp_amd.pipeline import BurstPipeline
    B = 53000                                   # 829 tiles, the last one partly filled: four per CU
    pipe = BurstPipeline(ctx, B, 512, 4)
    pipe.generate(SEED, b0=0)
    pipe.matched_filter(free_raw=True)
    got = {}
    try:
        for pp in (0, 2, -1):
            ctx.set_option("prio_period", pp)
            pipe.run()
            got[pp] = {k: pipe.fetch(k, 0, B)[:, :512 - (k == "mu")] for k in ("dec", "mu", "freq", "rec", "sig")}
            got[pp]["n"] = pipe.n_out().copy()
    finally:
        ctx.set_option("prio_period", -1)
    n_min = int(got[0]["n"].min())             # rows past a burst's n_out are scratch
    assert n_min > 500
    for pp in (2, -1):
        assert np.array_equal(got[0]["n"], got[pp]["n"])
        for k in ("dec", "mu", "freq", "rec", "sig"):
            assert np.array_equal(got[0][k][:, :n_min - 1], got[pp][k][:, :n_min - 1]), (pp, k)
    pipe.close()


def test_spare_wave_complex128_host_path(ctx):
    """The complex128 form of the kernel through the host-pointer entry point at two workgroups per CU."""
    import dsp_amd
    rng = np.random.default_rng(5)
    B, N = 20032, 256
    q = rng.integers(0, 4, (B, N // 4))
    sym = ((1 - 2 * (q & 1)) + 1j * (1 - 2 * (q >> 1))) / np.sqrt(2)
    x = np.repeat(sym, 4, axis=1) + 0.05 * (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N)))
    h = dsp_amd.rrc_filter(4, 10, 0.35)
    out = {}
    try:
        for sw in (0, -1):
            ctx.set_option("spare_wave", sw)
            out[sw] = dsp_amd.rx_chain_batch(x, h, 4, 0.007, 0.0, 0.03, 1, outputs=("sig", "dec", "mu", "freq"), ctx=ctx)
    finally:
        ctx.set_option("spare_wave", -1)
    for k in ("sig", "dec", "mu", "freq", "n_out"):
        assert np.array_equal(out[0][k], out[-1][k]), k


def test_more_tiles_than_lds_residency_allows(ctx):
    """B = 200 000 bursts = 3125 tiles, 13 per CU: more than the per-CU LDS can hold at once.  The
    launch must not be refused: the host issues such a batch as several launches of whole tiles
    (option batch_split), and every tile must produce the bits of a 64-burst call.  Two different
    tiles (samples, ragged lengths and per-burst scales all differ) alternate without a period, so a
    launch that read or wrote another launch's tiles, lengths or scales would show."""
    import dsp_amd
    n_cu, lds = ctx.device_info()
    assert n_cu == 256 and lds == 160 * 1024                      # MI355X; the host code reads both
    rng = np.random.default_rng(17)
    n_sym = 24

    def tile64():
        q = rng.integers(0, 4, (64, n_sym))
        sym = ((1 - 2 * (q & 1)) + 1j * (1 - 2 * (q >> 1))) / np.sqrt(2)
        x = (np.repeat(sym, 4, axis=1) + 0.05 * (rng.standard_normal((64, 4 * n_sym))
                                                 + 1j * rng.standard_normal((64, 4 * n_sym)))).astype(np.complex64)
        lens = rng.integers(4 * n_sym - 20, 4 * n_sym + 1, 64)
        scale = rng.uniform(0.5, 2.0, 64)
        return x, lens, scale

    pats = [tile64(), tile64()]
    names = ("rec", "err", "mu", "sig", "freq", "ef", "dec")

    def check(B, cdt, outs, split):
        nt = B // 64
        sel = (rng.random(nt) < 0.5).astype(np.int64)
        refs = [dsp_amd.sync_chain_batch(x.astype(cdt), 4, 0.007, 0.0, 0.03, 1, lens=l, scale=sc, outputs=outs, ctx=ctx)
                for x, l, sc in pats]
        X = np.stack([pats[0][0], pats[1][0]])[sel].reshape(B, -1).astype(cdt)
        L = np.stack([pats[0][1], pats[1][1]])[sel].reshape(B)
        S = np.stack([pats[0][2], pats[1][2]])[sel].reshape(B)
        ctx.set_option("batch_split", split)
        try:
            got = dsp_amd.sync_chain_batch(X, 4, 0.007, 0.0, 0.03, 1, lens=L, scale=S, outputs=outs, ctx=ctx)
        finally:
            ctx.set_option("batch_split", 1)
        n_ref = np.stack([refs[0]["n_out"], refs[1]["n_out"]])[sel].reshape(B)
        assert (got["n_out"] == n_ref).all()
        for k in outs:
            g = got[k].reshape(nt, 64, -1)
            r = np.stack([refs[0][k], refs[1][k]])[sel]
            n = n_ref.reshape(nt, 64, 1) - (1 if k in ("err", "mu") else 0)
            live = np.arange(g.shape[2])[None, None, :] < n          # rows past n_out are scratch
            assert (np.where(live, g, 0) == np.where(live, r, 0)).all(), (k, split)

    check(200000, np.complex64, names, 1)                         # three launches of 1280 / 1280 / 565 tiles
    check(200000, np.complex64, ("rec", "dec"), 0)                # one launch, the surplus tiles queue on the CUs
    # the complex128 form reaches the limit at half as many tiles per CU
    check(128000, np.complex128, ("rec", "dec", "freq"), 1)       # 2000 tiles: two launches of 1024 / 976
    check(96000, np.complex128, ("rec", "dec", "freq"), 1)        # 1500 tiles: one launch


def test_one_burst_beyond_4gib_of_output_rows(ctx, oracle):
    """A single complex128 burst of 2^22 + 3000 symbols: the `recovered` tile passes 4 GiB at row 2^22
    (64 lanes x 16 bytes per row), where the steady path's 32-bit store offsets would wrap.  Bit-exact
    against the oracle over the whole length (Bn = 1e-6: the FLL horizon is ~1e8 symbols)."""
    import dsp_amd
    rng = np.random.default_rng(23)
    n_sym = (1 << 22) + 3000
    q = rng.integers(0, 4, n_sym)
    sym = ((1 - 2 * (q & 1)) + 1j * (1 - 2 * (q >> 1))) / np.sqrt(2)
    x = np.repeat(sym, 4)
    x += 0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))
    r = dsp_amd.sync_chain_batch(x[None, :], 4, 0.007, 0.0, 1e-6, 1, outputs=("rec", "dec"), ctx=ctx)
    rec, err, mu = oracle.gardner_timing_recovery(x, 4, 0.007, 0.0)
    n = len(rec)
    assert r["n_out"][0] == n and n > (1 << 22)
    assert np.array_equal(r["rec"][0, :n], rec)
    out = oracle.fll_func(rec, 1e-6, 1)
    assert np.array_equal(r["dec"][0, :n], oracle.quadrants(out[0]))


def test_four_wave_launch_places_live_waves_on_distinct_simds(ctx):
    """Two workgroups per CU (512 tiles): the two-wave kernel is launched with four waves per tile, one on
    each SIMD, and a per-CU ticket decides which SIMD pair keeps its waves (DESIGN.md 3.1 finding 5).  The
    kernel reports where each tile's live waves run (option "place_ptr"): on every CU that holds two tiles
    the four live waves sit on four different SIMDs, and a tile's two waves never share one."""
    from dsp_amd.pipeline import BurstPipeline
    B = 32768
    pipe = BurstPipeline(ctx, B, 256, 4)
    pipe.generate(SEED, b0=0)
    pipe.matched_filter(free_raw=True)
    nt = B // 64
    d = ctx.malloc(nt * 2 * 4)
    try:
        ctx.memset(d, 0, nt * 2 * 4)
        ctx.set_option("place_ptr", d)
        pipe.run()
        ctx.synchronize()
        ctx.set_option("place_ptr", 0)
        w = np.empty((nt, 2), np.uint32)
        ctx.d2h(w, d)
    finally:
        ctx.set_option("place_ptr", 0)
        ctx.free(d)
        pipe.close()
    assert (w >> 31).all(), "every tile's two live waves reported (the four-wave launch was used)"
    hw = w & 0xffff
    simd = (hw >> 4) & 3
    cu = ((w >> 16) & 15).astype(np.int64) * 4096 + (hw >> 8).astype(np.int64)      # XCC | SE, SH, CU
    assert (cu[:, 0] == cu[:, 1]).all()
    assert (simd[:, 0] != simd[:, 1]).all()
    per_cu = {}
    for t in range(nt):
        per_cu.setdefault(int(cu[t, 0]), []).extend([int(simd[t, 0]), int(simd[t, 1])])
    two = [v for v in per_cu.values() if len(v) == 4]
    assert len(two) >= len(per_cu) // 2, "most CUs hold two tiles at 512 tiles"
    assert all(len(set(v)) == 4 for v in two)


def test_partly_filled_tile_ignores_its_idle_columns(ctx, oracle):
    """A partly filled last tile: its idle lanes run as passenger lanes (clones of the tile's longest
    burst, so that the tile keeps the steady path) and never read their own input columns -- with those
    columns overwritten by NaNs every output of every burst has the same bits, and the oracle's."""
    from dsp_amd.pipeline import BurstPipeline
    B = 64 * 5 + 23
    pipe = BurstPipeline(ctx, B, 512, 4)
    try:
        pipe.generate(SEED, b0=0)
        pipe.matched_filter(free_raw=True)
        pipe.run()
        names = ("dec", "mu", "freq", "rec", "err")
        before = {k: pipe.fetch(k, 0, B) for k in names}
        n_before = pipe.n_out().copy()
        # the last tile's rows [rows][64]: poison lanes 23..63
        nbytes = pipe.rows * 64 * 8
        tile = np.empty(pipe.rows * 64, np.complex64)
        last = pipe.d_tiles + 5 * nbytes
        ctx.d2h(tile, last)
        t2 = tile.reshape(pipe.rows, 64)
        t2[:, 23:] = np.nan + 1j * np.nan
        ctx.h2d(last, tile)
        pipe.run()
        assert np.array_equal(pipe.n_out(), n_before) and (n_before == 512).all()
        for k in names:                          # bits, not values: rows past n_out hold whatever was there (NaNs too)
            assert pipe.fetch(k, 0, B).tobytes() == before[k].tobytes(), k
        X = pipe.fetch_input(320, B)
        inv = pipe.fetch_inv_std(320, B)
        for b in (0, 11, 22):
            o = oracle.sync_chain_c64(X[b], scale=inv[b])
            assert np.array_equal(before["dec"][320 + b, :o["n"]], o["dec"])
            assert np.array_equal(before["mu"][320 + b, :o["n"] - 1], o["mu"].astype(np.float32))
    finally:
        pipe.close()
