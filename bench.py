#!/usr/bin/env python3
"""bench.py -- throughput of the fused Gardner+FLL kernel on batched QPSK bursts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bursts B] [--scaling weak|strong]

Metric (BASELINE.json): Msamples/s of FLL+Gardner on batched synthetic QPSK
bursts (4 sps, 8192 samples = 2048 symbols per burst), whole job.  One "step" =
one pass of the fused recurrence kernel over the rank's matched-filtered
complex64 bursts, resident in HBM in the tiled layout, writing ALL reference
outputs (recovered, timing error, mu, derotated symbols, frequency, loop-filter
output, decisions).

Multi-GPU.  Bursts are independent: every rank owns a contiguous block of the
job's bursts, no collective is on the data path (torch.distributed only aligns
the clocks).  Ranks come from torchrun (RANK / WORLD_SIZE / LOCAL_RANK in the
environment) or, when `--gpus N` is given without them, from this script's own
launcher: the parent starts N fresh child processes BEFORE it imports anything
that could touch a GPU, waits for them and prints rank 0's line.
  --scaling weak    (default) 65536 bursts per GPU;
  --scaling strong  BASELINE config 4 as written: 65536 bursts in total, split N ways.
With N > 1 the line of either mode also carries the other mode's timing
(`config.strong_scaling` / `config.weak_scaling`), measured right after the
main region on the same resident bursts.

The line carries the roofline of the recurrence kernel (algorithmic bytes 16.25
B/sample = 8 read + 33 B per symbol written, SURVEY.md 8d, over the HIP-event
time of the launch on the library's stream, against 8 TB/s), `roofline.chain`
(the same bytes over matched filter + recurrence: RRC -> /std -> Gardner -> FLL,
worker.py:299-304) and a CPU baseline: the oracle on the host cores.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

N_SYM, SPS = 2048, 4
N = N_SYM * SPS
ALPHA, BN, Y = 0.007, 0.03, 1.0
A_FULL = 16.25          # algorithmic bytes per input sample, all outputs
A_FIR = 16.0            # matched filter alone: complex64 in + complex64 out
HBM_PEAK = 8.0e12       # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
SEED = 20251114
JOB_BURSTS = 65536      # BASELINE config 4


# ---------------------------------------------------------------------------
# launcher (parent process; standard library only, nothing here touches a GPU)
def child_envs(n, port, base=None):
    """Environment of each of the n ranks this script starts by itself (what torchrun would set)."""
    base = dict(os.environ if base is None else base)
    envs = []
    for r in range(n):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DSP_BENCH_CHILD="1")
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(e)
    return envs


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(n, argv, timeout=900, script=None):
    """Start n rank processes of this script (or `script`), wait, return (exit code, rank 0's stdout)."""
    envs = child_envs(n, free_port())
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, rc = b"", 0
    t_end = time.time() + timeout
    try:
        out, _ = procs[0].communicate(timeout=timeout)
        for p in procs:
            p.wait(timeout=max(1.0, t_end - time.time()))
            rc = rc or p.returncode
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc, out.decode("utf-8", "replace")


# ---------------------------------------------------------------------------
def capture_mode(ctx, log2_samples=30, warmup=2048):
    """Informational (BASELINE.json configs[5]): ONE int16 IQ capture of 2^30 samples, generated on the
    device, through the segment-parallel receiver (peak, RRC + whole-stream std, Gardner + FLL over the
    windows, stitching); the second call is timed with HIP events, the decisions of the first 2^18 symbols
    are checked against the transmitted ones.  tools/bench_capture.py is the same run as a program."""
    import importlib.util
    from dsp_amd.pipeline import CapturePipeline
    spec = importlib.util.spec_from_file_location(
        "bench_capture", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "bench_capture.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)
    n, tau0, drift = 1 << log2_samples, 0.3, 2.0e-8
    pipe = CapturePipeline(ctx, n, SPS, seg_len=None, warmup=warmup)
    try:
        pipe.generate(bc.SEED, tau0=tau0, drift=drift)
        ctx.synchronize()
        pipe.run()                              # allocations, code objects
        ctx.timer_start()
        nsym = pipe.run()
        ms = ctx.timer_stop()
        rate, _ = bc.check_decisions(pipe, n // SPS, tau0, drift, SPS, 0, min(nsym - 8, 1 << 18))
        assert abs(nsym - int(round(n * (1.0 + drift) / SPS))) <= 4 and rate > 0.999, (nsym, rate)
        return {"workload": f"one int16 IQ capture of 2^{log2_samples} samples (4 sps QPSK, RRC 0.35, CFO 1.5e-4, "
                            f"drifting timing, Es/N0 20 dB), segment-parallel: seg_len {pipe.seg_len}, "
                            f"warm-up {warmup} symbols from a fractional feed-forward start, decisions + mu out",
                "segments": pipe.S, "symbols": nsym, "ms": ms, "Msamples_per_s": n / ms / 1e3,
                "symbol_match_rate": rate}
    finally:
        pipe.close()


def cpu_baseline(X, inv_std, seconds=12.0):
    """Oracle chain (gardner -> fll -> decisions) on the host cores, bounded sample."""
    import numpy as np
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
            j = (i * 17 + n) % X.shape[0]
            O.sync_chain_c64(X[j], sps=SPS, alpha=ALPHA, Bn=BN, y=Y, scale=inv_std[j], want="dec")
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
        rec, _, _ = P.gardner_np(X[b].astype(np.complex128) * inv_std[b], SPS, ALPHA, 0.0)
        P.fll_np(rec, BN, Y)
    dp = time.perf_counter() - t1
    return {"value": bursts * N / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{bursts} bursts x {N} samples (oracle Gardner+FLL+decisions, fp64 C, "
                      f"{cores} threads, {dt:.1f} s)",
            "reference_idiom_python": {"value": nb * N / dp / 1e6, "unit": "Msamples/s", "cores": 1,
                                       "sample": f"{nb} bursts, numpy-scalar loops as the reference writes them"}}


def kernel_source_hash():
    """sha256 over the sources of the recurrence kernel: a committed PMC figure is only reported
    for the kernel it was measured on."""
    h = hashlib.sha256()
    for f in ("sync_kernels.hip", "sync_lane.h", "dsp_math.h"):
        with open(os.path.join(ROOT, "dsp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_traffic(B):
    """HBM bytes per launch from the committed PMC profile, if it is of this workload AND of this
    kernel source; None otherwise (a stale figure is dropped, not reported)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if (tj["bursts_per_gpu"] == B and tj["samples_per_burst"] == N
                and tj.get("kernel_source_sha256") == kernel_source_hash()):
            return tj["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def functions_alone(ctx, pipe, B, reps=5):
    """HIP-event times of dsp_sync_tiles_dev with do_fll = 0 (the timing loop alone) and with variant 2 on the
    recovered symbols (fll_func alone); same buffers as the fused launch, whose outputs they reproduce."""
    import ctypes as C
    import dsp_amd
    from dsp_amd._lib import SyncArgs, check
    L = dsp_amd.lib()
    fused = pipe._sync_args

    def timed(a):
        for _ in range(2):
            check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(a)))
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(a)))
        return ctx.timer_stop() / reps

    g = SyncArgs()
    C.memmove(C.byref(g), C.byref(fused), C.sizeof(SyncArgs))
    g.do_fll = 0
    g.sig = None; g.freq = None; g.ef = None; g.dec = None
    g_ms = timed(g)
    f = SyncArgs()
    C.memmove(C.byref(f), C.byref(fused), C.sizeof(SyncArgs))
    f.variant = 2; f.x = pipe.d_out["rec"]; f.in_rows = pipe.out_rows; f.N = pipe.n_sym; f.lens = None
    f.scale = None; f.rec = None; f.err = None; f.mu = None; f.do_fll = 1
    f_ms = timed(f)
    check(L.dsp_sync_tiles_dev(ctx.handle, C.byref(fused)))       # leave the fused outputs behind
    ctx.synchronize()
    g_bytes, f_bytes = B * N * 12.0, B * N_SYM * 25.0
    return {"gardner_timing_recovery": {"kernel": "sync_split_kernel<c64,gardner2,no FLL>", "kernel_ms": g_ms,
                                        "bytes_per_sample": 12.0, "achieved": g_bytes / (g_ms * 1e-3) / 1e9,
                                        "frac": g_bytes / (g_ms * 1e-3) / HBM_PEAK},
            "fll_func": {"kernel": "sync_kernel<c64,fll only>", "kernel_ms": f_ms, "bytes_per_symbol": 25.0,
                         "achieved": f_bytes / (f_ms * 1e-3) / 1e9, "frac": f_bytes / (f_ms * 1e-3) / HBM_PEAK}}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --bursts per GPU; strong: --bursts in total, split over the GPUs")
    ap.add_argument("--bursts", type=int, default=JOB_BURSTS, help="bursts per GPU (weak) / in total (strong)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-other-mode", action="store_true", help="N > 1: do not also time the other scaling mode")
    ap.add_argument("--fir-reps", type=int, default=10, help="timed matched-filter passes for roofline.chain")
    ap.add_argument("--no-parts", action="store_true", help="skip the timing of the two functions on their own")
    ap.add_argument("--no-capture", action="store_true",
                    help="skip the informational capture-mode record (config.capture_2p30: one 2^30-sample int16 "
                         "capture through the segment-parallel receiver, timed after everything else on rank 0)")
    ap.add_argument("--lean", action="store_true",
                    help="also time the launch with decisions as the only output (informational; off by "
                         "default so that every launch of a profiled run is the measured one)")
    ap.add_argument("--interp-half-len", type=int, default=0,
                    help="(2h+1)-point Lagrange interpolator inside the timing loop instead of the reference's "
                         "linear one (informational runs; 0 = the metric's configuration)")
    ap.add_argument("--ring-rows", type=int, default=0)
    ap.add_argument("--lds-pad", type=int, default=-1)
    ap.add_argument("--split", type=int, default=-1, help="-1 auto, 0 one wave per tile, 1 two")
    ap.add_argument("--prio-period", type=int, default=-1, help="priority rotation period (DMA blocks; -1 auto, 0 off)")
    ap.add_argument("--queue-block", type=int, default=0, help="split kernel hand-off block (0 auto)")
    ap.add_argument("--spec", type=int, default=1,
                    help="split kernel: speculative branch-free segments (1, default) or per-symbol votes (0)")
    ap.add_argument("--spare-wave", type=int, default=-1,
                    help="split kernel, idle waves for SIMD balance: -1 auto (default), 0 never, 1 three waves, 2 four")
    return ap.parse_args(argv)


def rank_main(args):
    # Exactly one line on stdout: libraries write there too (RCCL prints a five-line version banner when
    # its first communicator is made, gloo its connection notes), so file descriptor 1 is pointed at
    # stderr for the whole run and the JSON line goes out through a duplicate of the real stdout.
    sys.stdout.flush()
    real_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    sys.path.insert(0, ROOT)
    from dsp_amd import sharding

    rank, world, local = sharding.rank_env()
    # several ranks on one device (rehearsal on a one-GPU box) cannot form an RCCL group
    backend = os.environ.get("DSP_BENCH_BACKEND", "gloo" if "DSP_BENCH_DEVICE" in os.environ else "nccl")
    try:
        dist = sharding.init_clock_group(world, local, backend)
        if dist is not None:
            dist.barrier()
    except Exception as e:                      # RCCL would not come up: the clocks can be aligned over gloo as well
        if backend != "nccl":
            raise
        print(f"bench.py: RCCL process group failed ({type(e).__name__}: {e}); aligning clocks over gloo",
              file=sys.stderr, flush=True)
        try:
            import torch.distributed as _d
            if _d.is_initialized():
                _d.destroy_process_group()
        except Exception:
            pass
        backend = "gloo"
        dist = sharding.init_clock_group(world, local, backend)
    red_dev = "cuda" if (dist is not None and backend == "nccl") else "cpu"

    import dsp_amd
    from dsp_amd.pipeline import BurstPipeline
    from dsp_amd.synth import device_symbols

    ctx = dsp_amd.Context(local)
    if args.ring_rows:
        ctx.set_option("ring_rows", args.ring_rows)
    if args.lds_pad >= 0:
        ctx.set_option("lds_pad", args.lds_pad)
    ctx.set_option("split", args.split)
    if args.prio_period >= 0:
        ctx.set_option("prio_period", args.prio_period)
    if args.queue_block:
        ctx.set_option("queue_block", args.queue_block)
    if args.spare_wave >= 0:
        ctx.set_option("spare_wave", args.spare_wave)
    ctx.set_option("spec", args.spec)

    # ---- this rank's bursts of the whole job ---------------------------------------------
    total = args.bursts * world if args.scaling == "weak" else args.bursts
    b_lo, b_hi = sharding.shard_range(total, rank, world)
    B = b_hi - b_lo
    # the other mode runs on a prefix (strong from weak) of the same resident bursts; from strong,
    # weak needs `bursts` per GPU, so the rank holds that many and the main region uses a prefix
    other = None if (world == 1 or args.no_other_mode) else ("strong" if args.scaling == "weak" else "weak")
    B_other = 0
    if other == "strong":
        lo2, hi2 = sharding.shard_range(args.bursts, rank, world)
        B_other = hi2 - lo2
    elif other == "weak":
        B_other = args.bursts
    B_res = max(B, B_other)

    # ---- synthetic bursts generated on the device, matched-filtered, resident in HBM ----
    pipe = BurstPipeline(ctx, B_res, N_SYM, SPS, alpha=ALPHA, Bn=BN, y=Y, interp_half_len=args.interp_half_len)
    pipe.generate(SEED, b0=b_lo)
    for _ in range(max(1, args.warmup)):       # untimed passes, as many as the fused kernel gets
        pipe.matched_filter()
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(max(1, args.fir_reps)):
        pipe.matched_filter()
    fir_ms = ctx.timer_stop() / max(1, args.fir_reps) * (B / B_res)     # per pass over this rank's B bursts
    pipe.free_raw()

    def set_bursts(nb):
        pipe._sync_args.B = nb

    def step():
        pipe.run()

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

    timer = (ctx.timer_start, ctx.timer_stop)     # HIP events on the launch stream
    set_bursts(B)
    wall, ev_ms = sharding.timed_region(step, ctx.synchronize, args.steps, args.warmup, dist, red_dev, timer)
    other_rec = None
    if other:
        set_bursts(B_other)
        w2, e2 = sharding.timed_region(step, ctx.synchronize, args.steps, args.warmup, dist, red_dev, timer)
        tot2 = args.bursts if other == "strong" else args.bursts * world
        other_rec = {"scaling": other, "bursts_total": tot2, "ms_per_step": w2 * 1e3 / args.steps,
                     "kernel_ms": e2 / args.steps, "value": tot2 * float(N) / (w2 / args.steps) / 1e6,
                     "unit": "Msamples/s"}
        set_bursts(B)
        step()
        ctx.synchronize()

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
    n_out = pipe.n_out()[:B]
    assert (n_out == N_SYM).all(), "unexpected symbol counts"
    nchk = min(64, B)
    dec = pipe.fetch("dec", 0, nchk)[:, :N_SYM]
    rot = np.array([1, 3, 0, 2])          # quadrant index (Re<0)+2(Im<0) after a +90 degree turn
    worst = 1.0
    for b in range(nchk):
        tx = device_symbols(SEED, b_lo + b, N_SYM)
        best, d = 0.0, dec[b].copy()
        for _ in range(4):
            best = max(best, float(np.mean(d[400:] == tx[400:])))
            d = rot[d]
        worst = min(worst, best)
    assert worst > 0.995, f"decisions do not match the transmitted symbols ({worst})"

    samples = float(total) * N
    ms_per_step = wall * 1e3 / args.steps
    kern_ms = ev_ms / args.steps
    value = samples / (wall / args.steps) / 1e6
    # roofline of the slowest rank's launch (ev_ms is the MAX over ranks; shards differ by <= 1 burst)
    achieved = B * N * A_FULL / (kern_ms * 1e-3) / 1e9
    chain_ms = fir_ms + kern_ms
    chain = B * N * A_FULL / (chain_ms * 1e-3) / 1e9
    mode_cfg = {"scaling": args.scaling, "bursts_total": total, "ms_per_step": ms_per_step,
                "kernel_ms": kern_ms, "value": value, "unit": "Msamples/s"}
    cfg = {"workload": f"{B} synthetic QPSK bursts per GPU x {N} samples (4 sps, 2048 symbols, "
                       "RRC 0.35, random CFO/timing offset, Es/N0 20 dB), matched-filtered "
                       "complex64 fl32(y) in HBM, 1/std folded into the loads -> fused "
                       "Gardner(alpha .007)+FLL(Bn .03)+decisions, all 7 outputs",
           "bursts_per_gpu": B, "bursts_total": total, "samples_per_burst": N,
           "generator": "on-device, counter-based RNG", "min_symbol_match_rate": worst,
           "matched_filter_ms": fir_ms, "interp_half_len": args.interp_half_len,
           "parallelism": f"bursts sharded x{world}, no collective" + (f" (clocks aligned over {backend})" if world > 1 else ""),
           args.scaling + "_scaling": mode_cfg}
    if other_rec:
        cfg[other + "_scaling"] = other_rec
    line = {
        "metric": "Msamples/s (node) FLL+Gardner on batched QPSK bursts",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": cfg,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved * 1e9 / HBM_PEAK, "traffic": committed_traffic(B),
                     "kernel": "sync_split_kernel<c64,gardner2>" if args.split != 0 else "sync_kernel<c64,gardner2,fll>",
                     "kernel_ms": kern_ms, "bytes_per_sample": A_FULL,
                     "same_box_float4_copy_GBps": copy_gbps,
                     "chain": {"what": "matched filter (RRC 'same' + np.std statistics, one pass) + fused "
                                       "Gardner+FLL+decisions, same algorithmic bytes (the filtered "
                                       "intermediate is not credited)",
                               "fir_ms": fir_ms, "sync_ms": kern_ms, "achieved": chain,
                               "frac": chain * 1e9 / HBM_PEAK, "bytes_per_sample": A_FULL,
                               "fir_kernel": "fir_strip_kernel<c64> (+ fir_stats_finish)",
                               "fir_achieved": B * N * A_FIR / (fir_ms * 1e-3) / 1e9,
                               "fir_bytes_per_sample": A_FIR},
                     "decisions_only": None if lean_ms is None else
                     {"kernel_ms": lean_ms, "bytes_per_sample": 8.25,
                      "achieved": B * N * 8.25 / (lean_ms * 1e-3) / 1e9}},
    }
    # ---- informational: the reference's two functions called on their own, on the same resident bursts
    #      (gardner_timing_recovery: 8 B read per sample + rec, err, mu = 16 B per symbol -> 12 B/sample;
    #      fll_func on the recovered symbols: 8 B read + sig, freq, ef, dec = 17 B written per symbol)
    if rank == 0 and world == 1 and not args.no_parts:
        line["roofline"]["functions_alone"] = functions_alone(ctx, pipe, B)
    if rank == 0:
        if not args.no_cpu:
            # N > 1: a 3-second sample (the other ranks wait at the closing barrier meanwhile)
            nb = min(128, B // 64 * 64) or B
            line["cpu_baseline"] = cpu_baseline(pipe.fetch_input(0, nb), pipe.fetch_inv_std(0, nb),
                                                seconds=12.0 if world == 1 else 3.0)
    pipe.close()
    if rank == 0:
        if not args.no_capture:
            line["config"]["capture_2p30"] = capture_mode(ctx)     # rank 0's GPU alone, whatever N is
        real_out.write(json.dumps(line) + "\n")
        real_out.flush()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher gave us ranks: be the launcher.  Nothing above has imported torch, dsp_amd
        # or any HIP-using module, and this process never will.
        rc, out = launch(args.gpus, argv)
        sys.stdout.write(out)
        sys.stdout.flush()
        sys.exit(rc)
    rank_main(args)


if __name__ == "__main__":
    main()
