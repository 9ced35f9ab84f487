#!/usr/bin/env python3
"""Config 5 of BASELINE.json: one continuous int16 IQ capture, segment-parallel receiver.

    python tools/bench_capture.py [log2_samples=28] [seg_len=auto] [warmup=2048] [trim=0]

(trim: the capture is 2^log2_samples - trim samples long, for sizes that are not whole segments)

Generates the capture on the device, runs the whole segmented receiver (peak, RRC + std of the
whole stream, windows, Gardner + FLL, stitching), checks the decisions against the transmitted
symbols (position-aligned, up to the one global QPSK rotation) and prints one JSON line.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd                                               # noqa: E402
from dsp_amd.pipeline import CapturePipeline                 # noqa: E402
from dsp_amd.synth import stream_symbols                     # noqa: E402

SEED = 20251114


def check_decisions(pipe, n_sym, tau0, drift, sps, lo, count):
    """Symbols [lo, lo+count) of the stitched stream against the transmitted ones."""
    dec = pipe.fetch("dec", lo, lo + count)
    mu = pipe.fetch("mu", 0, lo + count).astype(np.float64)
    # position of stitched symbol r: sps * (r + slips) + mu (row 0 of the stream is sample 0)
    d = np.diff(mu[1:])
    step = np.zeros(mu.size, np.int64)
    step[2:][d < -0.5 * sps] = 1
    step[2:][d > 0.5 * sps] = -1
    r = np.arange(mu.size)
    pos = sps * (r + np.cumsum(step)) + mu
    pos[0] = 0.0
    idx = np.rint((pos[lo:] * (1.0 + drift) + tau0) / sps).astype(np.int64)
    ok = (idx >= 0) & (idx < n_sym)
    tx = stream_symbols(SEED, int(idx[ok].min()), int(idx[ok].max() - idx[ok].min() + 1))[idx[ok] - idx[ok].min()]
    turn = np.array([1, 3, 0, 2], np.uint8)
    best, dd = 0.0, dec[ok].copy()
    for _ in range(4):
        best = max(best, float(np.mean(dd == tx)))
        dd = turn[dd]
    return best, int(np.sum(step != 0))


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    seg_len = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    trim = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    n, sps = (1 << lg) - trim, 4
    tau0, drift = 0.3, 2.0e-8
    ctx = dsp_amd.Context(0)
    for kv in os.environ.get("DSP_CAPTURE_OPTS", "").split(","):      # e.g. ring_rows=256,queue_block=32
        if kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    pipe = CapturePipeline(ctx, n, sps, seg_len=seg_len, warmup=warmup)
    pipe.generate(SEED, tau0=tau0, drift=drift)
    ctx.synchronize()
    pipe.run()                                  # warm-up (allocations, code objects)
    t0 = time.perf_counter()
    ctx.timer_start()
    nsym = pipe.run()
    ms = ctx.timer_stop()
    wall = time.perf_counter() - t0
    count = min(nsym - 8, 1 << 22)
    rate, nslip = check_decisions(pipe, n // sps, tau0, drift, sps, 0, count)
    expect = int(round(n * (1.0 + drift) / sps))       # a drifting symbol clock changes the symbol count
    assert abs(nsym - expect) <= 4, (nsym, expect)
    assert rate > 0.999, rate
    line = {"workload": f"one int16 IQ capture of {'2^%d' % lg if not trim else n} samples (4 sps QPSK, RRC 0.35, CFO 1.5e-4, drifting timing, "
                        f"Es/N0 20 dB), segment-parallel: seg_len {pipe.seg_len}, warm-up {warmup} symbols",
            "segments": pipe.S, "symbols": nsym, "ms": ms, "wall_ms": wall * 1e3,
            "Msamples_per_s": n / ms / 1e3, "symbol_match_rate": rate,
            "lean_bytes_per_sample": 4.25, "lean_GBps": n * 4.25 / ms / 1e6,
            "stages": "peak + RRC (int16 -> complex64 window tiles + whole-stream std) + Gardner/FLL over windows + stitch"}
    print(json.dumps(line), flush=True)
    pipe.close()
    ctx.close()


if __name__ == "__main__":
    main()
