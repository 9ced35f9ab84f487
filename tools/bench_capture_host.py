#!/usr/bin/env python3
"""The segment-parallel capture through its HOST-pointer entry point (dsp_sync_capture_segmented): an int16 IQ
capture in a numpy array in, decisions + mu + freq in numpy arrays out -- upload and download included
(tools/bench_capture.py is the device-resident form that bench.py reports).

    python tools/bench_capture_host.py [log2_samples=28]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    n, sps = 1 << lg, 4
    rng = np.random.default_rng(2)
    h = dsp_amd.rrc_filter(sps, 10, 0.35)
    # a periodic QPSK stream (one 2^20-sample period repeated: the timing does not depend on the symbols)
    m = 1 << 20
    q = rng.integers(0, 4, m // sps)
    sym = ((1 - 2 * (q & 1)) + 1j * (1 - 2 * (q >> 1))) / np.sqrt(2)
    up = np.zeros(m, complex); up[::sps] = sym
    tx = np.convolve(up, h, "same")
    tx = tx + 0.05 * (rng.standard_normal(m) + 1j * rng.standard_normal(m))
    w1 = np.empty(2 * m, np.int16)
    pk = np.max(np.abs(np.concatenate((tx.real, tx.imag))))
    w1[0::2] = np.round(tx.real / pk * 6000); w1[1::2] = np.round(tx.imag / pk * 6000)
    raw = np.tile(w1, n // m)
    res = {}
    for threads in (0, 4):
        dsp_amd.default_context().set_option("link_threads", threads)
        dsp_amd.sync_capture_segmented(raw, sps, 0.007, 0.0, 0.003, 1, taps=h)
        t0 = time.perf_counter()
        o = dsp_amd.sync_capture_segmented(raw, sps, 0.007, 0.0, 0.003, 1, taps=h)
        dt = time.perf_counter() - t0
        res["link_threads_%d" % threads] = {"ms": dt * 1e3, "Msamples_per_s": n / dt / 1e6, "symbols": int(o["n"])}
    res["workload"] = f"one int16 IQ capture of 2^{lg} samples in a numpy array, decisions + mu + freq back"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
