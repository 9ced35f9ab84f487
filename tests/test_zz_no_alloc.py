"""dsp_fir_tiles_static_dev: the matched filter entry point that allocates nothing (static device
scratch instead of a caller workspace).  Through the C ABI, against dsp_fir_tiles_dev on the same
device-resident bursts, bit for bit -- filtered tiles, std and 1/std -- on two calls in a row, the
second with a smaller input (the array keeps the first call's records), then with ragged lengths, a
request beyond the scratch size (refused, nothing launched) and the first call repeated."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(ctx, L, entry, X, taps, lens=None, want_std=True):
    """One call of `entry` (a dsp_fir_tiles*_dev) on bursts X [B, N] complex64 -> (tiles, std, inv)."""
    from dsp_amd._lib import FirArgs, check
    B, N = X.shape
    rows = int(L.dsp_tile_rows(N))
    ntiles = (B + 63) // 64
    bufs = []

    def dev(nbytes):
        p = ctx.malloc(max(nbytes, 16))
        bufs.append(p)
        return p

    try:
        d_x, d_h = dev(X.nbytes), dev(taps.nbytes)
        d_t, d_std, d_inv = dev(ntiles * rows * 64 * 8), dev(B * 8), dev(B * 8)
        ctx.h2d(d_x, X)
        ctx.h2d(d_h, taps)
        ctx.memset(d_t, 0xFF, ntiles * rows * 64 * 8)       # every row a call must write is written
        f = FirArgs()
        f.src = d_x; f.in_kind = 0; f.start = -1; f.B = B; f.N = N; f.src_stride = N
        f.h = d_h; f.n_taps = taps.size; f.rows = rows; f.tiles = d_t
        if lens is not None:
            d_l = dev(B * 4)
            ctx.h2d(d_l, lens)
            f.lens = d_l
        if want_std:
            f.std_out = d_std; f.inv_std_out = d_inv
        if entry is L.dsp_fir_tiles_dev and want_std:
            ws = int(L.dsp_fir_workspace_bytes(0, B, N, taps.size))
            f.workspace = dev(ws); f.workspace_bytes = ws
        check(entry(ctx.handle, C.byref(f)))
        ctx.synchronize()
        t = np.empty(ntiles * rows * 64, np.complex64)
        ctx.d2h(t, d_t)
        std, inv = np.zeros(B), np.zeros(B)
        if want_std:
            ctx.d2h(std, d_std)
            ctx.d2h(inv, d_inv)
        return t.view(np.uint64), std, inv
    finally:
        for p in bufs:
            ctx.free(p)


def _free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_static_entry_point_leaves_free_memory_unchanged(ctx):
    """No hipMalloc / hipFree inside the call: the device's free memory is the same before and after
    (every buffer is the test's own, allocated before the first reading)."""
    import dsp_amd
    from dsp_amd._lib import FirArgs, check
    L = dsp_amd.lib()
    rng = np.random.default_rng(5)
    B, N = 256, 2048
    X = (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N))).astype(np.complex64)
    taps = dsp_amd.rrc_filter(4, 10, 0.35)
    rows = int(L.dsp_tile_rows(N))
    bufs = [ctx.malloc(n) for n in (X.nbytes, taps.nbytes, (B // 64) * rows * 64 * 8, B * 8, B * 8)]
    try:
        ctx.h2d(bufs[0], X)
        ctx.h2d(bufs[1], taps)
        f = FirArgs()
        f.src = bufs[0]; f.in_kind = 0; f.start = -1; f.B = B; f.N = N; f.src_stride = N
        f.h = bufs[1]; f.n_taps = taps.size; f.rows = rows; f.tiles = bufs[2]
        f.std_out = bufs[3]; f.inv_std_out = bufs[4]
        check(L.dsp_fir_tiles_static_dev(ctx.handle, C.byref(f)))      # first call: code objects loaded
        ctx.synchronize()
        before = _free_bytes()
        for _ in range(3):
            check(L.dsp_fir_tiles_static_dev(ctx.handle, C.byref(f)))
            assert _free_bytes() == before                             # while the call's work is in flight
        ctx.synchronize()
        assert _free_bytes() == before
    finally:
        for p in bufs:
            ctx.free(p)


def test_static_scratch_entry_point_matches(ctx):
    import dsp_amd
    L = dsp_amd.lib()
    assert L.dsp_static_scratch_bytes() == 16 << 20
    rng = np.random.default_rng(77)
    taps = dsp_amd.rrc_filter(4, 10, 0.35)

    def bursts(B, N):
        return (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N)) + (0.2 - 0.1j)).astype(np.complex64)

    X1, X2 = bursts(700, 4096), bursts(70, 1000)              # strips + ends + a partial tile; then chunk per workgroup
    for X, lens in ((X1, None), (X2, None), (X2, rng.integers(100, 1001, 70).astype(np.int32)), (X1, None)):
        want = _run(ctx, L, L.dsp_fir_tiles_dev, X, taps, lens)
        got = _run(ctx, L, L.dsp_fir_tiles_static_dev, X, taps, lens)
        for w, g in zip(want, got):
            assert np.array_equal(w, g)
        assert np.isfinite(got[1]).all() and (got[1] > 0).all()
    # without statistics nothing but the taps is kept in the array
    want = _run(ctx, L, L.dsp_fir_tiles_dev, X2, taps, None, want_std=False)
    got = _run(ctx, L, L.dsp_fir_tiles_static_dev, X2, taps, None, want_std=False)
    assert np.array_equal(want[0], got[0])
    # a request beyond the array is refused with a clear message
    from dsp_amd._lib import FirArgs
    f = FirArgs()
    d = ctx.malloc(64)
    try:
        f.src = d; f.h = d; f.tiles = None; f.in_kind = 0; f.start = -1; f.B = 65536; f.N = 8192; f.src_stride = 8192
        f.n_taps = 41; f.rows = 8192; f.std_out = d; f.inv_std_out = d
        assert L.dsp_fir_tiles_static_dev(ctx.handle, C.byref(f)) == -1
        assert b"static scratch" in L.dsp_last_error()
    finally:
        ctx.free(d)
