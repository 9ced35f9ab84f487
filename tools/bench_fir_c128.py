#!/usr/bin/env python3
"""One pass of the matched filter on complex128 bursts (the drop-in path's sample type): device-resident
[B][N] complex128 -> complex128 tiles + np.std statistics, 41 taps.

    python tools/bench_fir_c128.py [log2_bursts=13] [N=8192]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd                                                # noqa: E402
from dsp_amd._lib import FirArgs, check                       # noqa: E402


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    B = 1 << lb
    ctx = dsp_amd.Context(0)
    L = dsp_amd.lib()
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((256, N)) + 1j * rng.standard_normal((256, N))).astype(np.complex128)
    rows = int(L.dsp_tile_rows(N))
    ntiles = (B + 63) // 64
    d_x = ctx.malloc(B * N * 16)
    for i in range(0, B, 256):
        ctx.h2d(d_x + i * N * 16, x[:min(256, B - i)])
    taps = dsp_amd.rrc_filter(4, 10, 0.35)
    d_h = ctx.malloc(taps.nbytes)
    ctx.h2d(d_h, taps)
    d_t = ctx.malloc(ntiles * rows * 64 * 16)
    d_std, d_inv = ctx.malloc(B * 8), ctx.malloc(B * 8)
    ws = int(L.dsp_fir_workspace_bytes(1, B, N, taps.size))
    d_ws = ctx.malloc(ws)
    f = FirArgs()
    f.src = d_x; f.in_kind = 1; f.start = -1; f.B = B; f.N = N; f.src_stride = N
    f.h = d_h; f.n_taps = taps.size; f.rows = rows; f.tiles = d_t
    f.std_out = d_std; f.inv_std_out = d_inv; f.workspace = d_ws; f.workspace_bytes = ws
    for _ in range(2):
        check(L.dsp_fir_tiles_dev(ctx.handle, C.byref(f)))
    ctx.synchronize(); ctx.timer_start()
    for _ in range(5):
        check(L.dsp_fir_tiles_dev(ctx.handle, C.byref(f)))
    ms = ctx.timer_stop() / 5
    print({"bursts": B, "N": N, "ms_per_pass": ms, "Gsamples_per_s": B * N / ms / 1e6,
           "GBps_32B_per_sample": B * N * 32 / ms / 1e6}, flush=True)


if __name__ == "__main__":
    main()
