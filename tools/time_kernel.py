#!/usr/bin/env python3
"""time_kernel.py B [opt=value ...] -- HIP-event time of the fused Gardner+FLL launch over B device-generated
bursts (no result checks: also for experiment builds whose arithmetic is deliberately cut).  Options are
dsp_set_option names (ring_rows, queue_block, spec, split, spare_wave, prio_period ...); half=H, drop=outputs,
variant=1 (gardner_ted in place of gardner2)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd                                               # noqa: E402
from dsp_amd.pipeline import BurstPipeline                   # noqa: E402

B = int(sys.argv[1])
ctx = dsp_amd.Context(0)
drop = []
half = 0
variant = 0
for a in sys.argv[2:]:
    k, v = a.split("=")
    if k == "half":                      # half=H : (2H+1)-point Lagrange interpolator in the timing loop
        half = int(v)
    elif k == "variant":
        variant = int(v)
    elif k == "drop":                      # drop=rec,err,mu,sig,freq,ef,dec : outputs not asked for
        drop = v.split(",")
    else:
        ctx.set_option(k, int(v))
pipe = BurstPipeline(ctx, B, 2048, 4, interp_half_len=half, variant=variant)
pipe.generate()
pipe.matched_filter(free_raw=True)
for k in drop:
    setattr(pipe._sync_args, k, None)
for _ in range(5):
    pipe.run()
ctx.synchronize()
ctx.timer_start()
for _ in range(30):
    pipe.run()
ms = ctx.timer_stop() / 30
print(os.path.basename(os.environ.get("DSP_HIP_LIB", "product")), "B", B, " ".join(sys.argv[2:]), "ms %.4f" % ms, flush=True)
pipe.close()
ctx.close()
