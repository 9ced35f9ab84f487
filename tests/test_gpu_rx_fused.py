"""Cut + derotation fused into the matched filter's staging (fir_kernels.hip, FirParams::derot_f;
worker.py:287-299): the fused pass must give THE SAME BITS as the derotation pass followed by the
complex128 matched filter, which tests/test_cfo.py pins against the reference's golden vectors.

Covered: complex64 and complex128 signals, offsets at odd samples (16-byte loads that start on an
8-byte boundary), overlapping bursts, full tiles + a partly filled one (strips + edge kernel), fewer
than 64 bursts, an odd packet length and a tap count with an odd start (both leave the strips to the
chunk-per-workgroup kernel), the three estimators and a given f_rel, and the device-resident entry
point dsp_rx_front_dev against dsp_cfo_derotate_dev + dsp_fir_tiles_dev.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _signal(rng, n, offs, pre, sps, dtype):
    x = 0.3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    L = len(pre)
    for o in offs:
        f = rng.uniform(-4e-3, 4e-3)
        x[o:o + L * sps:sps] += 2.0 * pre * np.exp(1j * (2 * np.pi * f * sps * np.arange(L) + 0.3))
    return x.astype(dtype)


CASES = [
    # B, n_pkt, n_taps, cfo_method, sps
    (200, 1792, 41, 1, 4),   # 3 full tiles through the strips + 8 bursts through the chunk kernel
    (130, 2048, 41, 3, 4),
    (70, 1001, 41, 2, 4),    # odd packet length: no two-sample pieces for complex64
    (5, 1600, 33, 1, 4),     # fewer bursts than lanes
    (128, 1200, 31, 0, 4),   # start = 15: odd, the strips' pieces would straddle the burst's first sample
    (64, 960, 21, 0, 4),
    (130, 1200, 31, 1, 3),   # odd samples-per-symbol counts: the whole flow, and full tiles on the recurrence's
    (128, 1500, 41, 2, 5),   #   steady path behind it (RRC span 10 at 3 sps = 31 taps, span 8 at 5 sps = 41)
]


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("B,n_pkt,n_taps,method,sps", CASES)
def test_fused_front_end_is_bit_identical(dtype, B, n_pkt, n_taps, method, sps):
    import dsp_amd
    rng = np.random.default_rng(B * 7 + n_pkt)
    L = 48
    n = 60000 + n_pkt
    pre = np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, L)))
    offs = np.sort(rng.integers(0, n - n_pkt, B)).astype(np.int64)
    offs[::3] |= 1                                   # odd starts
    offs = np.minimum(offs, n - n_pkt)
    offs[-1] = n - n_pkt                             # the last burst ends with the signal
    offs[0] = 0
    x = _signal(rng, n, offs, pre, sps, dtype)
    full = dsp_amd.rrc_filter(sps, 8 if sps == 5 else 10, 0.35)
    h = np.ascontiguousarray(full[(len(full) - n_taps) // 2:][:n_taps])
    assert len(h) == n_taps
    kw = dict(f_rel=rng.uniform(-3e-3, 3e-3, B)) if method == 0 else dict(cfo_method=method)
    ctx = dsp_amd.Context(0)
    res = []
    for fuse in (0, 1):
        ctx.set_option("fuse_derot", fuse)
        res.append(dsp_amd.rx_bursts(x, offs, n_pkt, pre, h, sps, 0.007, 0.0, 1e-3, 1, ctx=ctx, **kw))
    ctx.close()
    two, one = res
    for k in two:
        assert np.array_equal(two[k], one[k], equal_nan=True), k
    assert two["n_out"].min() > 0
    # ... and a burst against the oracle chain (the tolerances of tests/test_cfo.py)
    for b in (0, B - 1):
        o = int(offs[b])
        cut = x[o:o + n_pkt].astype(np.complex128)
        der = O.derotate(cut, one["f_rel"][b], o)
        y = O.filter_samples(der, h)
        sd = np.std(np.convolve(der, h, mode="same"))
        rec, _, _ = O.gardner_timing_recovery(y, sps, 0.007, 0.0)
        s, _, _, _ = O.fll_func(rec, 1e-3, 1)
        assert int(one["n_out"][b]) == len(s)
        assert np.abs(one["rec"][b, :len(s)] - rec).max() < 1e-9
        assert np.abs(one["sig"][b, :len(s)] - s).max() < 1e-6
        assert np.abs(one["std"][b] - sd) < 1e-12 * sd


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_rx_front_dev_matches_the_two_calls(dtype):
    """dsp_rx_front_dev: tiles, std and 1/std are the bits of dsp_cfo_derotate_dev + dsp_fir_tiles_dev."""
    import dsp_amd
    from dsp_amd._lib import FirArgs, check
    rng = np.random.default_rng(11)
    B, n_pkt, sps, L = 192, 1344, 4, 32
    n = 50001
    pre = np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, L)))
    offs = np.sort(rng.integers(0, n - n_pkt, B)).astype(np.int64)
    offs[1::2] |= 1
    offs = np.minimum(offs, n - n_pkt)
    x = _signal(rng, n, offs, pre, sps, dtype)
    h = np.ascontiguousarray(dsp_amd.rrc_filter(sps, 10, 0.35), dtype=np.float64)
    ctx = dsp_amd.Context(0)
    lib = dsp_amd.lib()
    vp = C.c_void_p
    dt = 1 if dtype == np.complex128 else 0
    rows = lib.dsp_tile_rows(n_pkt)
    ntiles = (B + 63) // 64
    tile_bytes = ntiles * rows * 64 * 16
    wsb = max(lib.dsp_fir_workspace_bytes(1, B, n_pkt, len(h)), lib.dsp_rx_front_workspace_bytes(B, n_pkt))
    assert lib.dsp_rx_front_workspace_bytes(B, n_pkt) > 0
    d = {k: ctx.malloc(v) for k, v in dict(x=x.nbytes, off=offs.nbytes, pre=pre.nbytes, f=B * 24, cut=B * n_pkt * 16,
                                           h=h.nbytes, t1=tile_bytes, t2=tile_bytes, s1=B * 8, s2=B * 8, i1=B * 8,
                                           i2=B * 8, ws=wsb).items()}
    ctx.h2d(d["x"], x); ctx.h2d(d["off"], offs); ctx.h2d(d["pre"], pre); ctx.h2d(d["h"], h)
    zeros = np.zeros(tile_bytes // 16, np.complex128)
    ctx.h2d(d["t1"], zeros); ctx.h2d(d["t2"], zeros)
    got = {}
    for method in (2, 0):
        if method == 0:
            ctx.h2d(d["f"], rng.uniform(-2e-3, 2e-3, B))
        check(lib.dsp_cfo_derotate_dev(ctx.handle, vp(d["x"]), dt, n, vp(d["off"]), B, n_pkt, sps, vp(d["pre"]), L, method,
                                       vp(d["f"]), vp(d["cut"])))
        a = FirArgs()
        a.src = d["cut"]; a.in_kind = 1; a.start = -1; a.B = B; a.N = n_pkt; a.src_stride = n_pkt
        a.h = d["h"]; a.n_taps = len(h); a.tiles = d["t1"]; a.rows = rows
        a.std_out = d["s1"]; a.inv_std_out = d["i1"]; a.workspace = d["ws"]; a.workspace_bytes = wsb
        check(lib.dsp_fir_tiles_dev(ctx.handle, C.byref(a)))
        ctx.synchronize()
        check(lib.dsp_rx_front_dev(ctx.handle, vp(d["x"]), dt, n, vp(d["off"]), B, n_pkt, sps, vp(d["pre"]), L, method,
                                   vp(d["f"]), vp(d["h"]), len(h), vp(d["t2"]), rows, vp(d["s2"]), vp(d["i2"]), vp(d["ws"]),
                                   wsb))
        ctx.synchronize()
        for a_, b_, shape, tp in (("t1", "t2", tile_bytes // 16, np.complex128), ("s1", "s2", B, np.float64),
                                  ("i1", "i2", B, np.float64)):
            u, v = np.zeros(shape, tp), np.zeros(shape, tp)
            ctx.d2h(u, d[a_]); ctx.d2h(v, d[b_])
            assert np.array_equal(u, v), (method, a_)
            got[a_] = u
        assert np.all(np.isfinite(got["s1"])) and got["s1"].min() > 0
    # argument errors come back as codes
    assert lib.dsp_rx_front_dev(ctx.handle, vp(d["x"]), dt, n, vp(d["off"]), B, n_pkt, sps, vp(d["pre"]), L, 1, vp(d["f"]),
                                vp(d["h"]), 43, vp(d["t2"]), rows, None, None, None, 0) != 0       # too many taps
    assert lib.dsp_rx_front_dev(ctx.handle, vp(d["x"]), dt, n, vp(d["off"]), B, n_pkt, sps, vp(d["pre"]), L, 1, vp(d["f"]),
                                vp(d["h"]), len(h), vp(d["t2"]), rows, vp(d["s2"]), None, vp(d["ws"]), 64) != 0  # workspace
    for v in d.values():
        ctx.free(v)
    ctx.close()
