#!/usr/bin/env python3
"""bench.py -- throughput of the fused Gardner+FLL kernel on batched QPSK bursts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bursts B]

Metric (BASELINE.json): Msamples/s of FLL+Gardner on batched synthetic QPSK
bursts (4 sps, 8192 samples = 2048 symbols per burst), whole job.  One "step" =
one pass of the fused recurrence kernel over the rank's B matched-filtered
complex64 bursts, resident in HBM in the tiled layout, writing ALL reference
outputs (recovered, timing error, mu, derotated symbols, frequency, loop-filter
output, decisions).  Weak scaling: every rank owns B bursts; no collective is
on the data path (torch.distributed only synchronises the clocks).

The line also carries the roofline of the kernel (algorithmic bytes: 16.25
B/sample = 8 read + 33 B per symbol written, SURVEY.md 8d, divided by the HIP
event time of the launch on the library's stream, against 8 TB/s HBM3E) and a
CPU baseline: the oracle (fp64 C port of the reference loops) on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SYM, SPS = 2048, 4
N = N_SYM * SPS
ALPHA, BN, Y = 0.007, 0.03, 1.0
A_FULL = 16.25          # algorithmic bytes per input sample, all outputs
HBM_PEAK = 8.0e12       # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
SEED = 20251114


def cpu_baseline(X, seconds=12.0):
    """Oracle chain (gardner -> fll -> decisions) on the host cores, bounded sample."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))

    def work(i):
        t_end = time.perf_counter() + seconds
        n = 0
        while time.perf_counter() < t_end:
            O.sync_chain_c64(X[(i * 17 + n) % X.shape[0]], sps=SPS, alpha=ALPHA, Bn=BN, y=Y, want="dec")
            n += 1
        return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        counts = list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    bursts = sum(counts)
    # the reference's own speed class: the same two loops written in its idiom (numpy scalars and
    # ufunc calls per sample, oracle/py_loops.py) on one core -- the reference itself cannot travel
    from oracle import py_loops as P
    nb = min(6, X.shape[0])
    t1 = time.perf_counter()
    for b in range(nb):
        rec, _, _ = P.gardner_np(X[b].astype(np.complex128), SPS, ALPHA, 0.0)
        P.fll_np(rec, BN, Y)
    dp = time.perf_counter() - t1
    return {"value": bursts * N / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{bursts} bursts x {N} samples (oracle Gardner+FLL+decisions, fp64 C, "
                      f"{cores} threads, {dt:.1f} s)",
            "reference_idiom_python": {"value": nb * N / dp / 1e6, "unit": "Msamples/s", "cores": 1,
                                       "sample": f"{nb} bursts, numpy-scalar loops as the reference writes them"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bursts", type=int, default=65536, help="bursts per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--lean", action="store_true",
                    help="also time the launch with decisions as the only output (informational; off by "
                         "default so that every launch of a profiled run is the measured one)")
    ap.add_argument("--ring-rows", type=int, default=0)
    ap.add_argument("--lds-pad", type=int, default=-1)
    ap.add_argument("--split", type=int, default=-1, help="-1 auto, 0 one wave per tile, 1 two")
    ap.add_argument("--prio-period", type=int, default=0, help="priority rotation period (DMA blocks)")
    ap.add_argument("--queue-block", type=int, default=0, help="split kernel hand-off block (0 auto)")
    ap.add_argument("--spare-wave", type=int, default=-1, help="split kernel: idle middle wave per tile (1, default) or not (0)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # rehearsal on a one-GPU box: DSP_BENCH_BACKEND=gloo DSP_BENCH_DEVICE=0 runs every rank on GPU 0
    backend = os.environ.get("DSP_BENCH_BACKEND", "nccl")
    if "DSP_BENCH_DEVICE" in os.environ:
        local = int(os.environ["DSP_BENCH_DEVICE"])
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import dsp_amd
    from dsp_amd.pipeline import BurstPipeline, shard_range
    from dsp_amd.synth import device_symbols

    ctx = dsp_amd.Context(local)
    if args.ring_rows:
        ctx.set_option("ring_rows", args.ring_rows)
    if args.lds_pad >= 0:
        ctx.set_option("lds_pad", args.lds_pad)
    ctx.set_option("split", args.split)
    if args.prio_period:
        ctx.set_option("prio_period", args.prio_period)
    if args.queue_block:
        ctx.set_option("queue_block", args.queue_block)
    if args.spare_wave >= 0:
        ctx.set_option("spare_wave", args.spare_wave)
    B = args.bursts
    b_lo, b_hi = shard_range(B * world, rank, world)      # this rank's bursts of the whole job

    # ---- synthetic bursts generated on the device, matched-filtered, resident in HBM ----
    pipe = BurstPipeline(ctx, B, N_SYM, SPS, alpha=ALPHA, Bn=BN, y=Y)
    pipe.generate(SEED, b0=b_lo)
    ctx.synchronize()
    ctx.timer_start()
    pipe.matched_filter(free_raw=True)
    fir_ms = ctx.timer_stop()

    def step():
        pipe.run()

    def barrier():
        ctx.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    # same-box reference: a plain float4 copy of the input tiles (1:1 read/write stream)
    copy_gbps = None
    nb = pipe.ntiles * pipe.rows * 64 * 8
    d_tmp = ctx.malloc(nb)
    try:
        ctx.stream_copy(d_tmp, pipe.d_tiles, nb)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(5):
            ctx.stream_copy(d_tmp, pipe.d_tiles, nb)
        copy_gbps = 2.0 * nb * 5 / (ctx.timer_stop() * 1e-3) / 1e9
    finally:
        ctx.free(d_tmp)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop()                       # HIP events on the launch stream
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        tw = torch.tensor([wall, ev_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tw[0]), float(tw[1])

    # ---- informational: the same launch with decisions as the only output (A_lean = 8.25 B/sample,
    #      SURVEY.md 8d); not part of `value`
    lean_ms = None
    if rank == 0 and args.lean:
        a = pipe._sync_args
        saved = {k: getattr(a, k) for k in ("rec", "err", "mu", "sig", "freq", "ef")}
        for k in saved:
            setattr(a, k, None)
        for _ in range(2):
            step()
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(5):
            step()
        lean_ms = ctx.timer_stop() / 5
        for k, v in saved.items():
            setattr(a, k, v)
        step()                                  # restore every output for the checks below
        ctx.synchronize()

    # ---- sanity (not timed, no oracle): every burst produced 2048 symbols, and the hard
    #      decisions of the first 64 bursts equal the transmitted symbols up to the QPSK
    #      phase ambiguity (the check of worker.py:136-170) once the loops have locked
    n_out = pipe.n_out()
    assert (n_out == N_SYM).all(), "unexpected symbol counts"
    dec = pipe.fetch("dec", 0, 64)[:, :N_SYM]
    rot = np.array([1, 3, 0, 2])          # quadrant index (Re<0)+2(Im<0) after a +90 degree turn
    worst = 1.0
    for b in range(64):
        tx = device_symbols(SEED, b_lo + b, N_SYM)
        best, d = 0.0, dec[b].copy()
        for _ in range(4):
            best = max(best, float(np.mean(d[400:] == tx[400:])))
            d = rot[d]
        worst = min(worst, best)
    assert worst > 0.995, f"decisions do not match the transmitted symbols ({worst})"
    samples = float(B) * N * world
    ms_per_step = wall * 1e3 / args.steps
    kern_ms = ev_ms / args.steps
    value = samples / (wall / args.steps) / 1e6
    achieved = B * N * A_FULL / (kern_ms * 1e-3) / 1e9
    traffic = None        # HBM bytes per launch from the committed PMC profile of this workload
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj["bursts_per_gpu"] == B and tj["samples_per_burst"] == N:
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    line = {
        "metric": "Msamples/s (node) FLL+Gardner on batched QPSK bursts",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} synthetic QPSK bursts per GPU x {N} samples (4 sps, 2048 symbols, "
                               "RRC 0.35, random CFO/timing offset, Es/N0 20 dB), matched-filtered "
                               "complex64 in HBM -> fused Gardner(alpha .007)+FLL(Bn .03)+decisions, "
                               "all 7 outputs", "bursts_per_gpu": B, "samples_per_burst": N,
                   "generator": "on-device, counter-based RNG", "min_symbol_match_rate": worst,
                   "matched_filter_ms": fir_ms,
                   "parallelism": f"bursts sharded x{world}, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic,
                     "kernel": "sync_split_kernel<c64,gardner2>" if args.split != 0 else "sync_kernel<c64,gardner2,fll>", "kernel_ms": kern_ms,
                     "bytes_per_sample": A_FULL,
                     "same_box_float4_copy_GBps": copy_gbps,
                     "decisions_only": None if lean_ms is None else
                     {"kernel_ms": lean_ms, "bytes_per_sample": 8.25,
                      "achieved": B * N * 8.25 / (lean_ms * 1e-3) / 1e9}},
    }
    if rank == 0:
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(pipe.fetch_input(0, 128))
        print(json.dumps(line), flush=True)
    pipe.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
