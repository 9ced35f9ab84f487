#!/usr/bin/env python3
"""First GPU run of the four-wave (CU ticket) launch of the two-wave Lagrange kernels, smallest size:
128 bursts x 64 symbols, interp_half_len 1 / 2 / 4, complex64 and complex128, bits against the plain
two-wave launch.  Run alone under its own timeout (round 2's fault: a v_readlane-restored store base
in front of an asm store, see sync_kernels.hip `sst`)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402

ctx = dsp_amd.Context(0)
rng = np.random.default_rng(7)
B, N = 128, 64 * 4 + 64
sym = (rng.integers(0, 2, (B, N // 4)) * 2 - 1 + 1j * (rng.integers(0, 2, (B, N // 4)) * 2 - 1)) / np.sqrt(2)
x = np.repeat(sym, 4, axis=1) + 0.05 * (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N)))
outs = ("rec", "err", "mu", "sig", "freq", "ef", "dec")
for dt in (np.complex64, np.complex128):
    for half in (1, 2, 4):
        xs = x.astype(dt)
        print("launch", dt.__name__, "half_len", half, "spare_wave 0", flush=True)
        ctx.set_option("spare_wave", 0)
        a = dsp_amd.sync_chain_batch(xs, 4, 0.007, 0.0, 0.03, 1, interp_half_len=half, outputs=outs, ctx=ctx)
        print("launch", dt.__name__, "half_len", half, "spare_wave 2 (four waves, two stay)", flush=True)
        ctx.set_option("spare_wave", 2)
        b = dsp_amd.sync_chain_batch(xs, 4, 0.007, 0.0, 0.03, 1, interp_half_len=half, outputs=outs, ctx=ctx)
        for k in outs + ("n_out",):
            assert np.array_equal(a[k], b[k]), (dt.__name__, half, k)
        assert int(a["n_out"].min()) >= 60
        print("   same bits,", int(a["n_out"][0]), "symbols", flush=True)
ctx.set_option("spare_wave", -1)
ctx.close()
print("probe_fourwave_lagr: ok")
