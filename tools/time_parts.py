#!/usr/bin/env python3
"""time_parts.py B -- the reference's two functions called on their own, device-resident: gardner_timing_recovery
alone (dsp_sync_tiles_dev with do_fll = 0: rec, err, mu out) and fll_func alone on the recovered symbols
(variant 2: sig, freq, ef, dec out), beside the fused launch that bench.py times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd                                               # noqa: E402
from dsp_amd._lib import SyncArgs, check                     # noqa: E402
from dsp_amd.pipeline import BurstPipeline                   # noqa: E402
import ctypes as C                                           # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = dsp_amd.Context(0)
for a in sys.argv[2:]:
    k, v = a.split("=")
    ctx.set_option(k, int(v))
L = dsp_amd.lib()
pipe = BurstPipeline(ctx, B, 2048, 4)
pipe.generate()
pipe.matched_filter(free_raw=True)


def timed(args, reps=20):
    for _ in range(3):
        check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(args)))
    ctx.synchronize(); ctx.timer_start()
    for _ in range(reps):
        check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(args)))
    return ctx.timer_stop() / reps


fused = pipe._sync_args
t_fused = timed(fused)
g = SyncArgs()
C.memmove(C.byref(g), C.byref(fused), C.sizeof(SyncArgs))
g.do_fll = 0; g.sig = None; g.freq = None; g.ef = None; g.dec = None
t_g = timed(g)
f = SyncArgs()
C.memmove(C.byref(f), C.byref(fused), C.sizeof(SyncArgs))
f.variant = 2; f.x = pipe.d_out["rec"]; f.in_rows = pipe.out_rows; f.N = pipe.out_rows - 8; f.lens = pipe.d_nout
f.scale = None; f.rec = None; f.err = None; f.mu = None; f.do_fll = 1
d_n2 = ctx.malloc(B * 4); f.n_out = d_n2
t_f = timed(f)
print(f"B {B}: fused {t_fused:.3f} ms | gardner alone {t_g:.3f} ms | fll alone {t_f:.3f} ms | sum {t_g + t_f:.3f}", flush=True)
ctx.free(d_n2)
pipe.close(); ctx.close()
