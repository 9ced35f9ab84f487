#!/usr/bin/env python3
"""fuzz_case.py TRIAL [opt=value ...] -- one trial of tests/test_gpu_fuzz.py::test_fuzz_chain outside pytest: where
the device and the oracle differ (burst, row, values), under any dsp_set_option (spec=0, split=0 ...)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import dsp_amd                                               # noqa: E402
from oracle import oracle                                    # noqa: E402
from test_gpu_fuzz import _burst                             # noqa: E402

trial = int(sys.argv[1])
ctx = dsp_amd.Context(0)
rng = np.random.default_rng(1000 + trial)
sps = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8]))
variant = int(rng.integers(0, 2))
B = int(rng.choice([1, 3, 63, 64, 65, 130]))
N = int(rng.choice([0, 1, 2 * sps - 1, 2 * sps, 37, 400, 1501]))
if N == 0:
    N = 5
alpha = float(rng.choice([0.007, 0.002, 0.05, 0.1]))
mu0 = float(rng.choice([0.0, 0.0, 0.37, -0.9, 1.5])) if variant == 0 else 0.0
Bn = float(rng.choice([0.03, 0.001, 0.2]))
lens = rng.integers(0, N + 1, B).astype(np.int32)
lens[rng.integers(0, B)] = N
if B > 2:
    lens[1] = 0
    lens[2] = min(N, 1)
X = np.zeros((B, N), np.complex128)
for b in range(B):
    X[b, :lens[b]] = _burst(rng, int(lens[b]), sps, int(rng.integers(0, 3)))
c64 = bool(rng.integers(0, 2))
if c64:
    X = X.astype(np.complex64)
scale = rng.uniform(0.5, 2.0, B) if rng.integers(0, 2) else None
split = int(rng.choice([-1, 0, 1]))
outs = "all" if rng.integers(0, 2) else ("dec", "mu", "freq", "rec")
print(dict(sps=sps, variant=variant, B=B, N=N, alpha=alpha, mu0=mu0, Bn=Bn, c64=c64, scaled=scale is not None, split=split, outs=outs))
ctx.set_option("split", split)
for a in sys.argv[2:]:
    k, v = a.split("=")
    ctx.set_option(k, int(v))
got = dsp_amd.sync_chain_batch(X, sps, alpha, mu0, Bn, 1, variant=variant, lens=lens, scale=scale, outputs=outs, ctx=ctx)
nbad = 0
for b in range(B):
    if lens[b] == 0:
        continue
    x = X[b, :lens[b]].astype(np.complex128) * (scale[b] if scale is not None else 1.0)
    try:
        rec, err, mu = oracle.gardner_timing_recovery(x, sps, alpha, mu0) if variant == 0 else oracle.gardner_ted(x, sps, alpha)
    except RuntimeError as e:
        print("burst", b, "oracle:", e)
        continue
    n = len(rec)
    nm = n - 1 if variant == 0 else n
    cdt, rdt = (np.complex64, np.float32) if c64 else (np.complex128, np.float64)
    if got["n_out"][b] != n:
        print("burst", b, "len", lens[b], "n_out", got["n_out"][b], "oracle", n)
        nbad += 1
        continue
    dr = np.nonzero(got["rec"][b, :n] != rec.astype(cdt))[0]
    dm = np.nonzero(got["mu"][b, :nm] != mu.astype(rdt))[0]
    if dr.size or dm.size:
        nbad += 1
        print("burst", b, "len", lens[b], "n", n, "rec rows", dr[:8], "of", dr.size, "mu rows", dm[:8], "of", dm.size)
        if dr.size:
            r = dr[0]
            print("   rec dev", got["rec"][b, r], "oracle", rec[r], " mu around", mu[max(0, r - 2):r + 1])
        if dm.size:
            r = dm[0]
            print("   mu dev", got["mu"][b, r], "oracle", mu[r])
print("bursts that differ:", nbad)
ctx.close()
