#!/usr/bin/env python3
"""The fused Gardner + FLL kernel on complex128 tiles (the drop-in path's sample type: 16 B per sample in,
float64 outputs, 32.25 B per sample): the matched-filtered complex64 tiles of the bench workload widened
on the host and copied back.

    python tools/bench_sync_c128.py [log2_bursts=14] [option=value ...]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd                                                # noqa: E402
from dsp_amd._lib import SyncArgs, check                      # noqa: E402
from dsp_amd.pipeline import BurstPipeline                    # noqa: E402


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    B = 1 << lb
    ctx = dsp_amd.Context(0)
    for opt in sys.argv[2:]:                                  # dsp_set_option names, e.g. big_ring=0
        k, v = opt.split("=")
        ctx.set_option(k, int(v))
    L = dsp_amd.lib()
    nb = min(B, 4096)
    pipe = BurstPipeline(ctx, nb, 2048, 4, outputs=("dec",))
    pipe.generate()
    pipe.matched_filter(free_raw=True)
    ctx.synchronize()
    rows, ntl = pipe.rows, nb // 64
    t64 = np.empty(ntl * rows * 64, np.complex64)
    ctx.d2h(t64, pipe.d_tiles)
    inv = pipe.fetch_inv_std(0, nb)
    t128 = t64.astype(np.complex128)
    ntiles = B // 64
    d_x = ctx.malloc(ntiles * rows * 64 * 16)
    d_inv = ctx.malloc(B * 8)
    for i in range(0, B, nb):                                 # the same 4096 bursts, repeated
        ctx.h2d(d_x + (i // 64) * rows * 64 * 16, t128)
        ctx.h2d(d_inv + i * 8, inv)
    pipe.close()
    out_rows = 2048 + 8
    elem = {"rec": 16, "err": 8, "mu": 8, "sig": 16, "freq": 8, "ef": 8, "dec": 1}
    a = SyncArgs()
    a.x = d_x; a.dtype = 1; a.variant = 0; a.B = B; a.N = 8192; a.in_rows = rows; a.sps = 4
    a.alpha = 0.007; a.mu0 = 0.0; a.do_fll = 1; a.Bn = 0.03; a.y = 1.0; a.scale = d_inv
    for k, e in elem.items():
        setattr(a, k, ctx.malloc(ntiles * out_rows * 64 * e))
    a.out_rows = out_rows; a.n_out = ctx.malloc(B * 4); a.status = ctx.malloc(B * 4)
    for _ in range(3):
        check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(a)))
    ctx.synchronize(); ctx.timer_start()
    for _ in range(10):
        check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(a)))
    ms = ctx.timer_stop() / 10
    n = np.empty(B, np.int32)
    ctx.d2h(n, a.n_out)
    print({"bursts": B, "ms_per_launch": ms, "Gsamples_per_s": B * 8192 / ms / 1e6,
           "GBps_32.25B_per_sample": B * 8192 * 32.25 / ms / 1e6, "symbols_ok": bool((n == 2048).all())}, flush=True)


if __name__ == "__main__":
    main()
