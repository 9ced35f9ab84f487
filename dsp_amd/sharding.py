"""Multi-GPU jobs: independent bursts sharded over ranks, one process and one dsp_ctx per GPU.

The path has no exchange step: rank r of W owns the contiguous block
shard_range(total, r, W) of the job's bursts, runs the same device pipeline on it, and the
results are concatenated by whoever collects them.  torch.distributed is used for ONE thing
only: aligning the clocks of a timed region (barrier before and after, MAX over ranks of the
elapsed time).  No collective ever touches samples or decisions.

    rank_env()      rank / world / local device from the launcher's environment
    timed_region()  warm-up, barrier, K timed steps, barrier, MAX over ranks  (bench.py)
    run_rank()      shard -> make the shard's runner -> timed_region -> per-rank record
    collect()       concatenate per-rank blocks in rank order and check that they tile the job

The rank body takes the shard runner as a callable, so the same code drives the HIP pipeline
(bench.py, tests/test_gpu_sharding.py) and a CPU stand-in under two gloo ranks
(tests/test_sharding_gloo.py).
"""
from __future__ import annotations

import os
import time


def shard_range(total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of `total` independent bursts owned by `rank` of `world`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_env(env=None):
    """(rank, world, local_device) as torchrun / bench.py's own launcher export them.
    DSP_BENCH_DEVICE pins every rank to one device (rehearsal of N ranks on a one-GPU box)."""
    env = os.environ if env is None else env
    rank = int(env.get("RANK", "0"))
    world = int(env.get("WORLD_SIZE", "1"))
    local = int(env.get("LOCAL_RANK", "0"))
    if "DSP_BENCH_DEVICE" in env:
        local = int(env["DSP_BENCH_DEVICE"])
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad RANK/WORLD_SIZE: {rank}/{world}")
    return rank, world, local


def init_clock_group(world: int, local: int, backend: str = "nccl", force: bool = False):
    """torch.distributed process group used for barriers and the MAX of the timings, or None
    for a single rank.  backend "nccl" is RCCL on ROCm (one GPU per rank); "gloo" works on the
    CPU and when several ranks rehearse on one device.  force: form the group for a single rank
    too (tests/test_gpu_sharding.py runs the RCCL branch that way on a one-GPU box)."""
    if world <= 1 and not force:
        return None
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return dist


def _max_over_ranks(dist, values, device):
    if dist is None:
        return [float(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def timed_region(step, sync, steps: int, warmup: int, dist=None, device="cpu", timer=None):
    """W untimed warm-up steps, then exactly K steps bracketed by (device sync + barrier) on both
    sides.  Returns (wall_s, timer_ms): the MAX over ranks of the wall time of the K steps and of
    the optional device timer (timer = (start, stop) callables around the same K launches, e.g.
    HIP events on the launch stream; stop returns milliseconds)."""
    def fence():
        sync()
        if dist is not None:
            dist.barrier()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    if timer:
        timer[0]()
    for _ in range(steps):
        step()
    ev_ms = timer[1]() if timer else 0.0
    fence()
    wall = time.perf_counter() - t0
    wall, ev_ms = _max_over_ranks(dist, [wall, ev_ms], device)
    return wall, ev_ms


def run_rank(total: int, rank: int, world: int, make_runner, steps: int = 1, warmup: int = 0,
             dist=None, device="cpu"):
    """One rank of a sharded job.  make_runner(lo, hi) builds the shard's work for bursts
    [lo, hi) of the job and returns an object with step(), sync() and result() (and optionally
    timer_start() / timer_stop() -> ms, close()).  Returns a record with the shard's range, the
    job-wide timings (MAX over ranks) and the shard's result."""
    lo, hi = shard_range(total, rank, world)
    runner = make_runner(lo, hi)
    timer = None
    if hasattr(runner, "timer_start") and hasattr(runner, "timer_stop"):
        timer = (runner.timer_start, runner.timer_stop)
    try:
        wall, ev_ms = timed_region(runner.step, runner.sync, steps, warmup, dist, device, timer)
        res = runner.result()
    finally:
        if hasattr(runner, "close"):
            runner.close()
    return {"rank": rank, "world": world, "lo": lo, "hi": hi, "wall_s": wall, "timer_ms": ev_ms,
            "result": res}


def collect(records, total: int):
    """Rank records -> list of results in burst order; raises unless the shards tile [0, total)."""
    recs = sorted(records, key=lambda r: r["lo"])
    pos = 0
    for r in recs:
        if r["lo"] != pos:
            raise ValueError(f"shards do not tile the job: gap or overlap at burst {pos}")
        pos = r["hi"]
    if pos != total:
        raise ValueError(f"shards cover {pos} of {total} bursts")
    return [r["result"] for r in recs]


class PipelineRunner:
    """The HIP pipeline as a shard runner: bursts [lo, hi) of the job generated on the device
    (burst identity = global index, so the concatenated job does not depend on the sharding),
    matched filter, then step() = one fused Gardner+FLL+decision launch over the shard."""

    def __init__(self, ctx, lo, hi, n_sym=2048, sps=4, seed=20251114, want=("dec",), **kw):
        from .pipeline import BurstPipeline
        self.ctx, self.lo, self.hi, self.want = ctx, lo, hi, tuple(want)
        self.n_sym = int(n_sym)
        self.pipe = BurstPipeline(ctx, hi - lo, n_sym, sps, **kw) if hi > lo else None
        if self.pipe:
            self.pipe.generate(seed, b0=lo)
            self.pipe.matched_filter(free_raw=True)

    def step(self):
        if self.pipe:
            self.pipe.run()

    def sync(self):
        self.ctx.synchronize()

    def timer_start(self):
        self.ctx.timer_start()

    def timer_stop(self):
        return self.ctx.timer_stop()

    def result(self):
        if not self.pipe:
            return {}
        # the rows a burst produced: n_out[b] symbols (mu / err: one per iteration, n_out[b] - 1; a timing
        # slip makes that n_sym +- 1); what lies beyond them in the device tiles is scratch, not part of
        # the result, and is returned as zeros
        import numpy as np
        n = self.pipe.n_out()
        width = max(int(n.max(initial=0)), self.n_sym)
        out = {}
        for k in self.want:
            a = self.pipe.fetch(k, 0, self.hi - self.lo)[:, :width]
            live = np.arange(a.shape[1])[None, :] < (n - (1 if k in ("mu", "err") else 0))[:, None]
            out[k] = np.where(live, a, np.zeros((), a.dtype))
        out["n_out"] = n
        return out

    def close(self):
        if self.pipe:
            self.pipe.close()
            self.pipe = None
