#!/usr/bin/env python3
"""Times the HOST-pointer entry points of the boundary, PCIe both ways included (bench.py's `value` is
the device-resident rate; this one is what a caller sees who hands over numpy arrays, as the reference's
own functions do).

    python tools/bench_host_boundary.py [log2_bursts=13]

  * dsp_sync_chain  (fll_func.py:3-29 + gardner2.py:6-100 on B matched-filtered complex64 bursts, all
    seven outputs): 8 B in + 8.25 B out per sample over the link;
  * dsp_rx_chain    (int16 IQ words in -> loader, matched filter + np.std, loops, decisions + mu out):
    4 B in + 1.25 B out per sample.
Prints one JSON line: ms per call, Msamples/s, and the GB/s the call moved over the link.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from dsp_amd import _lib  # noqa: E402
from dsp_amd.api import _vp  # noqa: E402
from dsp_amd._lib import check  # noqa: E402


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    B, N, SPS = 1 << lb, 8192, 4
    ctx = dsp_amd.Context(0)
    L = _lib.lib()
    rng = np.random.default_rng(3)
    # QPSK-like bursts (the loops only need something to lock on; the timing does not depend on the values)
    sym = (rng.integers(0, 2, (B, N // SPS)) * 2 - 1 + 1j * (rng.integers(0, 2, (B, N // SPS)) * 2 - 1)) / np.sqrt(2)
    x = np.repeat(sym, SPS, axis=1).astype(np.complex64)
    x += (rng.standard_normal(x.shape, dtype=np.float32) * 0.05).astype(np.complex64)
    lens = np.full(B, N, dtype=np.int32)
    cap = N // SPS + 8
    rec = np.zeros((B, cap), np.complex64); sig = np.zeros((B, cap), np.complex64)
    err = np.zeros((B, cap), np.float32); mu = np.zeros((B, cap), np.float32)
    freq = np.zeros((B, cap), np.float32); ef = np.zeros((B, cap), np.float32)
    dec = np.zeros((B, cap), np.uint8); n_out = np.zeros(B, np.int32)

    def chain():
        check(L.dsp_sync_chain(ctx.handle, _vp(x), 0, B, N, N, _vp(lens), None, SPS, 0.007, 0.0, 0, 0, 0.03, 1.0,
                               _vp(rec), _vp(err), _vp(mu), _vp(sig), _vp(freq), _vp(ef), _vp(dec), _vp(n_out), cap))

    res = {}
    reps = 3
    link = x.nbytes + sum(a.nbytes for a in (rec, sig, err, mu, freq, ef, dec, n_out))

    def timed(fn):
        for _ in range(2):
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    for mode, threads in (("runtime_pageable_copy", 0), ("staged_4_threads", 4)):
        ctx.set_option("link_threads", threads)
        dt = timed(chain)
        res["dsp_sync_chain/" + mode] = {"ms": dt * 1e3, "Msamples_per_s": B * N / dt / 1e6,
                                         "link_GBps": link / dt / 1e9, "n_out0": int(n_out[0])}
    # the same call on page-locked arrays (dsp_host_alloc): no staging at all
    keep = (x, rec, sig, err, mu, freq, ef, dec)
    x, rec, sig, err, mu, freq, ef, dec = (dsp_amd.pinned_empty(a.shape, a.dtype) for a in keep)
    x[...] = keep[0]
    dt = timed(chain)
    res["dsp_sync_chain/page_locked_arrays"] = {"ms": dt * 1e3, "Msamples_per_s": B * N / dt / 1e6,
                                                "link_GBps": link / dt / 1e9, "n_out0": int(n_out[0])}
    x, rec, sig, err, mu, freq, ef, dec = keep
    # two contexts on two host threads, half the bursts each: one half's upload runs beside the other's download
    import threading
    ctx2 = dsp_amd.Context(0)
    ctx.set_option("link_threads", 4); ctx2.set_option("link_threads", 4)
    h = B // 2

    def half(cx, lo, hi):
        check(L.dsp_sync_chain(cx.handle, _vp(x[lo:hi]), 0, hi - lo, N, N, _vp(lens[lo:hi]), None, SPS, 0.007, 0.0, 0, 0,
                               0.03, 1.0, _vp(rec[lo:hi]), _vp(err[lo:hi]), _vp(mu[lo:hi]), _vp(sig[lo:hi]),
                               _vp(freq[lo:hi]), _vp(ef[lo:hi]), _vp(dec[lo:hi]), _vp(n_out[lo:hi]), cap))

    def both():
        ts = [threading.Thread(target=half, args=(ctx, 0, h)), threading.Thread(target=half, args=(ctx2, h, B))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    dt = timed(both)
    res["dsp_sync_chain/two_contexts_two_threads"] = {"ms": dt * 1e3, "Msamples_per_s": B * N / dt / 1e6,
                                                      "link_GBps": link / dt / 1e9, "n_out0": int(n_out[0])}
    ctx2.close()
    # int16 IQ words in, decisions + mu out: the lean flow of worker.py:299-304 on raw bursts
    xi = np.empty((B, 2 * N), np.int16)
    xi[:, 0::2] = np.clip(x.real * 8000, -32768, 32767).astype(np.int16)
    xi[:, 1::2] = np.clip(x.imag * 8000, -32768, 32767).astype(np.int16)
    h = np.ascontiguousarray(dsp_amd.rrc_filter(SPS, 10, 0.35), dtype=np.float64)
    std = np.zeros(B, np.float64)

    def rx():
        check(L.dsp_rx_chain(ctx.handle, _vp(xi), _lib.DSP_I16, B, N, N, _vp(lens), 1, _vp(h), h.size, 1, SPS, 0.007,
                             0.0, 0, 0, 0.03, 1.0, None, None, _vp(mu), None, None, None, _vp(dec), _vp(n_out), cap,
                             _vp(std)))

    link_rx = xi.nbytes + mu.nbytes + dec.nbytes + n_out.nbytes + std.nbytes
    for mode, threads in (("runtime_pageable_copy", 0), ("staged_4_threads", 4)):
        ctx.set_option("link_threads", threads)
        dt = timed(rx)
        res["dsp_rx_chain_i16_lean/" + mode] = {"ms": dt * 1e3, "Msamples_per_s": B * N / dt / 1e6,
                                                "link_GBps": link_rx / dt / 1e9, "n_out0": int(n_out[0]),
                                                "link_bytes_per_sample": link_rx / (B * N)}
    res.update(workload=f"{B} bursts x {N} complex64 samples in pageable numpy arrays, all seven outputs back",
               link_bytes_per_sample=link / (B * N))
    print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
