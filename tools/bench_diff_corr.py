#!/usr/bin/env python3
"""Kernel time of the preamble search (dsp_diff_corr) on a long signal; run under
rocprofv3 --kernel-trace --stats to read the kernel's own duration (the call includes H2D / D2H).

    python tools/bench_diff_corr.py [log2_samples=24] [preamble_symbols=64] [corr_window=1]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if len(sys.argv) > 3:
    dsp_amd.default_context().set_option("corr_window", int(sys.argv[3]))
rng = np.random.default_rng(0)
n = 1 << lg
x = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32)).astype(np.complex64)
pre = np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, L)))
x[4 * 1000:4 * (1000 + L):4] += 3 * pre.astype(np.complex64)
dsp_amd.diff_corr(x[:4096], pre, 4)
t = time.perf_counter()
c = dsp_amd.diff_corr(x, pre, 4)
dt = time.perf_counter() - t
print("n", n, "L", L, "call %.1f ms" % (dt * 1e3), "argmax", int(np.argmax(np.abs(c[0]))), "(planted at 1000)")
