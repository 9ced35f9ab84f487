#!/usr/bin/env python3
"""Diagnostic: per-wave start/end stamps of the split recurrence kernel (trace build only).

    make -C dsp_amd/csrc trace
    DSP_HIP_LIB=dsp_amd/libdsp_hip_trace.so python tools/wave_timeline.py [bursts] [--check]

Prints how the tiles were spread over XCDs / CUs / SIMDs and how long the waves lived,
which is what decides whether a launch is balanced.  Never used by the product path.

History: the first run of this build in round 2 faulted the GPU box.  Cause (round 3, found in the ISA on
the CPU): the stamps build's register pressure made the compiler restore a spilled store base with
v_readlane directly in front of an inline-asm `global_store ... s[n:n+1]`, which needs 5 wait states after
a VALU write of an SGPR; the stores' asm text now carries its own guard and `make trace` / `make stamps`
link only after tools/isa_lint.py passed over their assembly.
"""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd                                               # noqa: E402
from dsp_amd.pipeline import BurstPipeline                   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
ctx = dsp_amd.Context(0)
pipe = BurstPipeline(ctx, B, 2048, 4, variant=int(os.environ.get("VARIANT", "0")))     # VARIANT=1: gardner_ted
pipe.generate()
pipe.matched_filter(free_raw=True)
nt = pipe.ntiles
d_tr = ctx.malloc(nt * 2 * 20 * 8)
ctx.memset(d_tr, 0, nt * 2 * 20 * 8)
ctx.set_option("trace_ptr", d_tr)
for opt in ("prio_period", "queue_block", "ring_rows", "trace_stamps", "split"):
    if os.environ.get(opt.upper()):
        ctx.set_option(opt, int(os.environ[opt.upper()]))
for _ in range(3):
    pipe.run()
ctx.synchronize()
ctx.timer_start()
pipe.run()
ms = ctx.timer_stop()
tr = np.empty((nt, 2, 20), np.uint64)
ctx.d2h(tr, d_tr)
t0 = tr[..., 0].astype(np.int64)
t1 = tr[..., 1].astype(np.int64)
hw = tr[..., 2].astype(np.int64)
xcc = tr[..., 3].astype(np.int64) & 0xF
# HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] (se is 3 bits on gfx94x)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
print(f"B={B} tiles={nt} kernel {ms:.3f} ms")
base = t0.min()
life = (t1 - t0)
print("wave lifetime cycles  G: min %d med %d max %d   F: min %d med %d max %d" % (
    life[:, 0].min(), np.median(life[:, 0]), life[:, 0].max(),
    life[:, 1].min(), np.median(life[:, 1]), life[:, 1].max()))
q = [0, 5, 25, 50, 75, 95, 100]
print("lifetime percentiles", q, np.percentile(life, q).astype(np.int64).tolist())
# (the s_memtime counters of different CUs are not synchronised: only differences inside a wave)
cuid = xcc[:, 0] * 1000 + se[:, 0] * 100 + sh[:, 0] * 16 + cu[:, 0]
cnt = collections.Counter(cuid.tolist())
print("CUs used:", len(cnt), " tiles per CU histogram:", sorted(collections.Counter(cnt.values()).items()))
same = (simd[:, 0] == simd[:, 1]).mean()
print(f"G and F wave of a tile on the same SIMD: {same * 100:.1f}%")
sc = collections.Counter((cuid * 4 + simd[:, 0]).tolist())
sc.update((cuid * 4 + simd[:, 1]).tolist())
print("waves per SIMD histogram:", sorted(collections.Counter(sc.values()).items()))
if "--check" in sys.argv:
    # self-check of the idle-wave launches (DESIGN.md 3.1 finding 5): with at most two tiles on a CU
    # (four live waves) no SIMD may carry two of them
    if max(cnt.values()) <= 2:
        worst = max(sc.values())
        print("SIMD placement check:", "ok" if worst == 1 else f"FAILED ({worst} live waves on one SIMD)")
        if worst != 1:
            sys.exit(1)
    else:
        print("SIMD placement check: skipped (more than two tiles per CU)")
# lifetime as a function of how many tiles shared the CU
for n in sorted(set(cnt.values())):
    m = np.array([cnt[c] == n for c in cuid.tolist()])
    print(f"  CUs with {n} tiles: median wave life {int(np.median(life[m]))} cycles")
if os.environ.get("TRACE_STAMPS"):
    # segment sums (ticks) per wave, averaged over the tiles, per symbol
    seg = tr[..., 4:].astype(np.float64).mean(axis=0) / 2048.0
    names_g = ["loop/advance overhead", "index+vote", "LDS reads a,b,c", "lerp+TED+mu+out index", "LDS reads out",
               "out lerp + commit", "cvt + stores", "queue write", "barrier", "DMA wait+issue"]
    names_f = ["barrier -> more flag", "queue read", "phase+sincos", "rotate+sign+vote", "atan2", "PI + stores + loop"]
    print("G wave, ticks per symbol (total %.0f):" % seg[0].sum())
    for i, n in enumerate(names_g):
        print(f"   {n:26s} {seg[0, i]:8.1f}  {100 * seg[0, i] / seg[0].sum():5.1f}%")
    print("F wave, ticks per symbol (total %.0f):" % seg[1].sum())
    for i, n in enumerate(names_f):
        print(f"   {n:26s} {seg[1, i]:8.1f}  {100 * seg[1, i] / seg[1].sum():5.1f}%")
if not os.environ.get("TRACE_STAMPS"):
    # trace build without stamps: four coarse register timers per wave (sync_kernels.hip DSP_TT)
    seg = tr[..., 4:10].astype(np.float64).mean(axis=0) / 2048.0
    tot = life.astype(np.float64).mean(axis=0) / 2048.0
    print("ticks per symbol        DMA wait+issue   segments/blocks   barrier   rest   head->segment   segment->hand-over  (lifetime)")
    for w, name in enumerate(("timing wave", "FLL wave")):
        print(f"  {name:14s} {seg[w, 0]:14.0f} {seg[w, 1]:17.0f} {seg[w, 2]:9.0f} {seg[w, 3]:6.0f} {seg[w, 4]:14.0f} {seg[w, 5]:18.0f}     {tot[w]:.0f}")
if os.environ.get("TRACE_STAMPS"):
    # round 3 (speculative segments): coarse stamps only -- per symbol, all 16 slots of both waves
    print("G raw ticks/symbol:", " ".join(f"{i}:{v:.0f}" for i, v in enumerate(seg[0])))
    print("   9 = DMA wait + issue, 10 = block bookkeeping in front of a segment, 11 = inside speculative segments, 8 = hand-over + barrier")
    print("F raw ticks/symbol:", " ".join(f"{i}:{v:.0f}" for i, v in enumerate(seg[1])))
    print("   0 = barrier wait, 12 = inside speculative blocks (flag read + dispatch + loop + vote), 5 = loop tail")
ctx.free(d_tr)
pipe.close()
ctx.close()
