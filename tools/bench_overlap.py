#!/usr/bin/env python3
"""Experiment: the matched filter of batch j + 1 on one HIP stream while the fused Gardner + FLL kernel of
batch j runs on another (two full-size batches in flight, events between the streams).  The filter is
fp64-issue bound and the recurrence HBM / latency bound, so together they may fill a CU better than one
after the other.  Prints ms per batch for both schedules.

    python tools/bench_overlap.py [log2_bursts=16] [steps=20]
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                  # noqa: E402  (streams and events only)
import dsp_amd                                                # noqa: E402
from dsp_amd._lib import check                                # noqa: E402
from dsp_amd.pipeline import BurstPipeline                    # noqa: E402


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    B = 1 << lb
    L = dsp_amd.lib()
    torch.cuda.set_device(0)
    ctx_f, ctx_s = dsp_amd.Context(0), dsp_amd.Context(0)
    s_f, s_s = torch.cuda.Stream(), torch.cuda.Stream()
    ctx_f.set_stream(s_f.cuda_stream)
    ctx_s.set_stream(s_s.cuda_stream)
    pipes = [BurstPipeline(ctx_f, B, 2048, 4) for _ in range(2)]
    for i, p in enumerate(pipes):
        p.generate(20251114, b0=i * B)
        p.matched_filter()
    torch.cuda.synchronize()

    def sync_on(p, ctx):
        check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(p._sync_args)))

    def sequential(n):
        for j in range(n):
            p = pipes[j % 2]
            p.matched_filter()
            sync_on(p, ctx_f)

    def overlapped(n):
        fir_done = [torch.cuda.Event() for _ in pipes]
        sync_done = [torch.cuda.Event() for _ in pipes]
        for e in sync_done:
            e.record(s_s)
        for j in range(n):
            k = j % 2
            s_f.wait_event(sync_done[k])          # the tiles of batch k are free again
            pipes[k].matched_filter()
            fir_done[k].record(s_f)
            s_s.wait_event(fir_done[k])
            sync_on(pipes[k], ctx_s)
            sync_done[k].record(s_s)

    out = {}
    for name, fn in (("sequential", sequential), ("overlapped", overlapped), ("sequential_again", sequential)):
        fn(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) * 1e3 / steps
    n_ok = all((p.n_out() == 2048).all() for p in pipes)
    print({"bursts": B, "steps": steps, "ms_per_batch": out, "all_bursts_2048_symbols": bool(n_ok)}, flush=True)
    for p in pipes:
        p.close()


if __name__ == "__main__":
    main()
