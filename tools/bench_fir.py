#!/usr/bin/env python3
"""Times one pass of the RRC matched-filter kernel (fir_tile_kernel) on device-resident bursts:

    python tools/bench_fir.py [log2_bursts=16] [both]

statistics only / output only / both, complex64 in and out, 41 taps, 8192 samples per burst.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from dsp_amd._lib import FirArgs, check  # noqa: E402
from dsp_amd.pipeline import BurstPipeline  # noqa: E402


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ctx = dsp_amd.Context(0)
    pipe = BurstPipeline(ctx, 1 << lb, 2048, 4)
    pipe.generate()
    L = dsp_amd.lib()

    def run(stats, write):
        f = FirArgs()
        f.src = pipe.d_raw; f.in_kind = 0; f.start = -1; f.B = pipe.B; f.N = pipe.N; f.src_stride = pipe.N
        f.h = pipe.d_taps; f.n_taps = pipe.taps.size; f.rows = pipe.rows
        f.workspace = pipe.d_ws; f.workspace_bytes = pipe.ws_bytes
        f.tiles = pipe.d_tiles if write else None
        if stats:
            f.std_out = pipe.d_std; f.inv_std_out = pipe.d_inv
        for _ in range(2):
            check(L.dsp_fir_tiles_dev(ctx.handle, C.byref(f)))
        ctx.synchronize(); ctx.timer_start()
        for _ in range(5):
            check(L.dsp_fir_tiles_dev(ctx.handle, C.byref(f)))
        return ctx.timer_stop() / 5

    if len(sys.argv) > 2 and sys.argv[2] == "both":       # one mode only (profiling)
        print("bursts %d: both %.2f ms" % (pipe.B, run(1, 1)))
    else:
        print("bursts %d: stats only %.2f  write only %.2f  both %.2f ms" % (pipe.B, run(1, 0), run(0, 1), run(1, 1)))
    pipe.close(); ctx.close()


if __name__ == "__main__":
    main()
