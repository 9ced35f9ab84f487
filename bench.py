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
UNIQUE = 256            # distinct synthetic bursts; replicated to fill B


def make_bursts(n_unique):
    """Seeded synthetic bursts (SURVEY 8d recipe) -> RX matched filter + /std -> complex64."""
    import dsp_amd
    from dsp_amd.synth import synth_burst
    h = dsp_amd.rrc_filter(SPS, 10, 0.35)
    out = np.empty((n_unique, N), np.complex64)
    for b in range(n_unique):
        x = synth_burst(b, h)
        y = np.convolve(x.astype(np.complex128), h, mode="same")
        out[b] = (y / np.std(y)).astype(np.complex64)
    return out


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
    return {"value": bursts * N / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{bursts} bursts x {N} samples (oracle Gardner+FLL+decisions, fp64 C, "
                      f"{cores} threads, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bursts", type=int, default=65536, help="bursts per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--ring-rows", type=int, default=0)
    ap.add_argument("--lds-pad", type=int, default=-1)
    ap.add_argument("--split", type=int, default=-1, help="-1 auto, 0 one wave per tile, 1 two")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import ctypes as C
    import dsp_amd
    from dsp_amd._lib import SyncArgs, check, lib

    ctx = dsp_amd.Context(local)
    if args.ring_rows:
        ctx.set_option("ring_rows", args.ring_rows)
    if args.lds_pad >= 0:
        ctx.set_option("lds_pad", args.lds_pad)
    ctx.set_option("split", args.split)
    L = lib()
    B = args.bursts
    ntiles = (B + 63) // 64
    rows = int(L.dsp_tile_rows(N))
    out_rows = N_SYM + 8

    # ---- synthetic input, resident in HBM in the tiled layout ---------------
    Xu = make_bursts(min(UNIQUE, B))
    d_src = ctx.malloc(Xu.nbytes)
    ctx.h2d(d_src, Xu)
    d_x = ctx.malloc(ntiles * rows * 64 * 8)
    nu_tiles = (Xu.shape[0] + 63) // 64
    for t0 in range(0, ntiles, nu_tiles):           # replicate the unique tiles over all tiles
        nb = min(nu_tiles * 64, B - t0 * 64)
        check(L.dsp_tile_bursts_dev(ctx.handle, C.c_void_p(d_src), 0, nb, N, N, None,
                                    C.c_void_p(d_x + t0 * rows * 64 * 8), rows))
    osz = {"rec": 8, "err": 4, "mu": 4, "sig": 8, "freq": 4, "ef": 4, "dec": 1}
    d_out = {k: ctx.malloc(ntiles * out_rows * 64 * v) for k, v in osz.items()}
    d_nout = ctx.malloc(B * 4)
    d_status = ctx.malloc(B * 4)
    a = SyncArgs()
    a.x = d_x; a.dtype = 0; a.variant = 0; a.B = B; a.N = N; a.in_rows = rows
    a.sps = SPS; a.interp_half_len = 0; a.alpha = ALPHA; a.mu0 = 0.0
    a.do_fll = 1; a.Bn = BN; a.y = Y
    for k in osz:
        setattr(a, k, d_out[k])
    a.out_rows = out_rows; a.n_out = d_nout; a.status = d_status

    def step():
        check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(a)))

    def barrier():
        ctx.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

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
        tw = torch.tensor([wall, ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tw[0]), float(tw[1])

    # ---- sanity: decisions of the first tile vs the oracle (not timed) -------
    n_out = np.empty(B, np.int32)
    ctx.d2h(n_out, d_nout)
    status = np.empty(B, np.int32)
    ctx.d2h(status, d_status)
    assert not status.any(), "capacity overflow"
    samples = float(B) * N * world
    ms_per_step = wall * 1e3 / args.steps
    kern_ms = ev_ms / args.steps
    value = samples / (wall / args.steps) / 1e6
    achieved = B * N * A_FULL / (kern_ms * 1e-3) / 1e9
    line = {
        "metric": "Msamples/s (node) FLL+Gardner on batched QPSK bursts",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} synthetic QPSK bursts per GPU x {N} samples (4 sps, 2048 symbols, "
                               "RRC 0.35, random CFO/timing offset, Es/N0 20 dB), matched-filtered "
                               "complex64 in HBM -> fused Gardner(alpha .007)+FLL(Bn .03)+decisions, "
                               "all 7 outputs", "bursts_per_gpu": B, "samples_per_burst": N,
                   "unique_bursts": int(Xu.shape[0]), "parallelism": f"bursts sharded x{world}, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved * 1e9 / HBM_PEAK, "traffic": None,
                     "kernel": "sync_kernel<c64,gardner2,fll>", "kernel_ms": kern_ms,
                     "bytes_per_sample": A_FULL},
    }
    if rank == 0:
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(Xu)
        print(json.dumps(line), flush=True)
    for p in list(d_out.values()) + [d_x, d_src, d_nout, d_status]:
        ctx.free(p)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
