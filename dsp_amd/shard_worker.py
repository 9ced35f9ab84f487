"""One rank of a sharded burst job as a process of its own:

    RANK=r WORLD_SIZE=w LOCAL_RANK=d python -m dsp_amd.shard_worker --total T --out PREFIX

runs bursts shard_range(T, r, w) of the job through the device pipeline (sharding.PipelineRunner:
generate -> matched filter -> fused Gardner+FLL) and writes PREFIX.rank<r>.npz (lo, hi, decisions,
mu, symbol counts, the job-wide step time).  Whoever started the ranks concatenates the files
(sharding.collect).  tests/test_gpu_sharding.py starts two of these on one GPU and compares the
concatenation with a single-process run bit for bit."""
import argparse
import os
import sys

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--total", type=int, required=True)
    ap.add_argument("--n-sym", type=int, default=256)
    ap.add_argument("--sps", type=int, default=4)
    ap.add_argument("--seed", type=int, default=20251114)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)

    from . import sharding
    rank, world, local = sharding.rank_env()
    backend = os.environ.get("DSP_BENCH_BACKEND", "gloo" if "DSP_BENCH_DEVICE" in os.environ else "nccl")
    dist = sharding.init_clock_group(world, local, backend)
    from ._lib import Context
    ctx = Context(local)
    rec = sharding.run_rank(
        a.total, rank, world,
        lambda lo, hi: sharding.PipelineRunner(ctx, lo, hi, a.n_sym, a.sps, a.seed, want=("dec", "mu")),
        steps=a.steps, warmup=a.warmup, dist=dist, device="cuda" if (dist and backend == "nccl") else "cpu")
    res = rec["result"]
    np.savez(f"{a.out}.rank{rank}.npz", lo=rec["lo"], hi=rec["hi"], wall_s=rec["wall_s"],
             dec=res.get("dec", np.zeros((0, 0), np.uint8)), mu=res.get("mu", np.zeros((0, 0), np.float32)),
             n_out=res.get("n_out", np.zeros(0, np.int32)))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
