#!/usr/bin/env python3
"""CPU baseline (1) of SURVEY.md 8(d): the REFERENCE's own Python path -- rrc_filter + np.convolve
'same' + / np.std + gardner2.gardner_timing_recovery + fll_func.fll_func -- timed in the build
container (the reference never travels to the GPU box): 64 synthetic bursts of 8192 samples (the 32
golden bursts of tests/golden/g_synth_bursts.npz, twice), one process and an 8-process pool,
median of 3 passes.  Prints one JSON line; the numbers are quoted in DESIGN.md section 5.

    python tools/time_reference_cpu.py          (needs /root/reference; not part of any GPU run)
"""
import contextlib
import io
import json
import multiprocessing as mp
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _load():
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use = lambda *a, **k: None
    with contextlib.redirect_stdout(io.StringIO()):
        import fll_func as ref_fll
        import gardner2 as ref_g2
        import read_file as ref_rf
    return ref_rf.rrc_filter, ref_g2.gardner_timing_recovery, ref_fll.fll_func


_F = None


def chain(x):
    global _F
    if _F is None:
        cwd = os.getcwd(); os.chdir(REF)
        try:
            _F = _load()
        finally:
            os.chdir(cwd)
    rrc, gardner, fll = _F
    with contextlib.redirect_stdout(io.StringIO()):
        h = rrc(4, 10, 0.35)
        y = np.convolve(x, h, mode="same")
        y = y / np.std(y)
        rec, _, _ = gardner(y, 4, alpha=0.007, mu_initial=0.0)
        sl, cf, _, _ = fll(rec, 0.03, 1)
    return len(sl)


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g_synth_bursts.npz"))
    X = [np.asarray(r, dtype=np.complex128) for r in g["x"]] * 2          # 64 bursts
    n = sum(len(x) for x in X)
    chain(X[0])
    t1 = []
    for _ in range(3):
        t = time.perf_counter()
        for x in X:
            chain(x)
        t1.append(time.perf_counter() - t)
    with mp.Pool(8) as pool:
        pool.map(chain, X[:8])
        t8 = []
        for _ in range(3):
            t = time.perf_counter()
            pool.map(chain, X, chunksize=2)
            t8.append(time.perf_counter() - t)
    m1, m8 = statistics.median(t1), statistics.median(t8)
    print(json.dumps({"bursts": len(X), "samples": n, "one_process_s": m1, "one_process_Msamples_per_s": n / m1 / 1e6,
                      "pool8_s": m8, "pool8_Msamples_per_s": n / m8 / 1e6, "cpus": os.cpu_count()}))


if __name__ == "__main__":
    main()
