#!/usr/bin/env python3
"""Times the burst cutter's front end on device-resident data: preamble-aided
coarse CFO per burst + cut + derotation (dsp_cfo_derotate_dev), the complex128 matched filter that
follows it (dsp_fir_tiles_dev), and the two as ONE pass (dsp_rx_front_dev: derotation inside the
filter's staging loads), on B synthetic bursts laid back to back as ONE complex64 signal.

    python tools/bench_rx_front.py [log2_bursts=14]

Prints one JSON line: ms and GB/s (algorithmic bytes: 8 B read + 16 B written per sample for the
derotation, 16 + 16 for the filter, 8 + 16 for the fused pass; the estimator touches n_pre samples
per burst), checks a few bursts against numpy and the fused tiles against the two-pass ones, bit for bit.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from dsp_amd._lib import FirArgs, check  # noqa: E402
from dsp_amd.pipeline import BurstPipeline  # noqa: E402


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    B, N, SPS, L = 1 << lb, 8192, 4, 64
    ctx = dsp_amd.Context(0)
    lib = dsp_amd.lib()
    pipe = BurstPipeline(ctx, B, N // SPS, SPS)
    pipe.generate()
    rng = np.random.default_rng(1)
    pre = np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, L)))
    offs = np.arange(B, dtype=np.int64) * N
    d_pre, d_off = ctx.malloc(pre.nbytes), ctx.malloc(offs.nbytes)
    ctx.h2d(d_pre, pre); ctx.h2d(d_off, offs)
    d_f, d_cut = ctx.malloc(B * 24), ctx.malloc(B * N * 16)
    vp = C.c_void_p

    def run(method):
        check(lib.dsp_cfo_derotate_dev(ctx.handle, vp(pipe.d_raw), 0, B * N, vp(d_off), B, N, SPS, vp(d_pre), L,
                                       method, vp(d_f), vp(d_cut)))

    res = {}
    for name, method in (("estimate+derotate", 1), ("derotate_only", 0)):
        if method == 0:
            ctx.h2d(d_f, np.full(B, 1e-4))
        for _ in range(2):
            run(method)
        ctx.synchronize(); ctx.timer_start()
        for _ in range(5):
            run(method)
        ms = ctx.timer_stop() / 5
        res[name] = {"ms": ms, "GBps": B * N * 24 / (ms * 1e-3) / 1e9}
    # sanity: first and last burst against numpy
    run(1)
    f = np.zeros((B, 3)); ctx.d2h(f, d_f)
    for b in (0, B - 1):
        raw = np.zeros(N, np.complex64); cut = np.zeros(N, np.complex128)
        check(lib.dsp_d2h(ctx.handle, raw.ctypes.data_as(vp), vp(pipe.d_raw + b * N * 8), raw.nbytes))
        check(lib.dsp_d2h(ctx.handle, cut.ctypes.data_as(vp), vp(d_cut + b * N * 16), cut.nbytes))
        p = raw[::SPS][:L].astype(np.complex128) * np.conj(pre)
        f1 = np.mean(np.diff(np.unwrap(np.angle(p)))) / (2 * np.pi * SPS)
        assert abs(f1 - f[b, 0]) < 1e-12, (f1, f[b])
        want = raw * np.exp(-1j * 2 * np.pi * f1 * (np.arange(N) + offs[b]))
        assert np.abs(cut - want).max() < 1e-9 * max(1.0, np.abs(raw).max()), np.abs(cut - want).max()
    # the matched filter behind it, and the fused pass
    h = np.ascontiguousarray(dsp_amd.rrc_filter(SPS, 10, 0.35), dtype=np.float64)
    rows = lib.dsp_tile_rows(N)
    tile_bytes = (B // 64) * rows * 64 * 16
    wsb = max(lib.dsp_fir_workspace_bytes(1, B, N, len(h)), lib.dsp_rx_front_workspace_bytes(B, N))
    d_h, d_t1, d_t2, d_ws = ctx.malloc(h.nbytes), ctx.malloc(tile_bytes), ctx.malloc(tile_bytes), ctx.malloc(wsb)
    d_s, d_i = ctx.malloc(B * 8), ctx.malloc(B * 8)
    ctx.h2d(d_h, h)
    a = FirArgs()
    a.src = d_cut; a.in_kind = 1; a.start = -1; a.B = B; a.N = N; a.src_stride = N
    a.h = d_h; a.n_taps = len(h); a.tiles = d_t1; a.rows = rows
    a.std_out = d_s; a.inv_std_out = d_i; a.workspace = d_ws; a.workspace_bytes = wsb

    def filt():
        check(lib.dsp_fir_tiles_dev(ctx.handle, C.byref(a)))

    def fused(method=0):
        check(lib.dsp_rx_front_dev(ctx.handle, vp(pipe.d_raw), 0, B * N, vp(d_off), B, N, SPS, vp(d_pre), L, method,
                                   vp(d_f), vp(d_h), len(h), vp(d_t2), rows, vp(d_s), vp(d_i), vp(d_ws), wsb))

    ctx.h2d(d_f, np.full(B, 1e-4))
    run(0)
    for name, fn, bps in (("filter_c128", filt, 32), ("fused_derotate_filter", fused, 24)):
        for _ in range(2):
            fn()
        ctx.synchronize(); ctx.timer_start()
        for _ in range(5):
            fn()
        ms = ctx.timer_stop() / 5
        res[name] = {"ms": ms, "GBps": B * N * bps / (ms * 1e-3) / 1e9}
    res["two_pass_ms"] = res["derotate_only"]["ms"] + res["filter_c128"]["ms"]
    # the fused tiles are the two-pass tiles, bit for bit (a slice of every 97th tile keeps the copy small)
    same = True
    for t in range(0, B // 64, 97):
        u, v = np.zeros(rows * 64, np.complex128), np.zeros(rows * 64, np.complex128)
        check(lib.dsp_d2h(ctx.handle, u.ctypes.data_as(vp), vp(d_t1 + t * rows * 64 * 16), u.nbytes))
        check(lib.dsp_d2h(ctx.handle, v.ctypes.data_as(vp), vp(d_t2 + t * rows * 64 * 16), v.nbytes))
        same = same and np.array_equal(u, v) and bool(np.any(u != 0))
    assert same, "fused tiles differ from derotate + filter"
    res["fused_bits_equal_two_pass"] = True
    for dd in (d_h, d_t1, d_t2, d_ws, d_s, d_i):
        ctx.free(dd)
    res.update(workload=f"{B} bursts x {N} complex64 samples back to back, preamble {L} symbols, sps {SPS}",
               bytes_per_sample=24)
    print(json.dumps(res))
    for d in (d_pre, d_off, d_f, d_cut):
        ctx.free(d)
    pipe.close(); ctx.close()


if __name__ == "__main__":
    main()
