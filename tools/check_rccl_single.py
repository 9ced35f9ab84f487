#!/usr/bin/env python3
"""The RCCL branch of dsp_amd.sharding with a world of ONE rank (a one-GPU box cannot hold two RCCL
ranks): init_clock_group("nccl") as bench.py calls it for N > 1, barrier, float64 MAX all-reduce on the
device, timed_region with its fences.  tests/test_gpu_sharding.py runs this as a child process, so that
the first 8-GPU bench is not the first run of this code."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
import torch                                   # noqa: E402
from dsp_amd import sharding                   # noqa: E402

rank, world, local = sharding.rank_env()
assert (rank, world) == (0, 1)
dist = sharding.init_clock_group(world, local, "nccl", force=True)
assert dist is not None and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.barrier()
dev = torch.device("cuda", local)
got = sharding._max_over_ranks(dist, [1.5, 2.5], dev)
assert got == [1.5, 2.5], got
print("max over ranks:", got)
wall, ev = sharding.timed_region(lambda: None, torch.cuda.synchronize, 3, 1, dist, dev)
assert wall >= 0.0
print("timed_region ok", wall, ev)
dist.barrier()
dist.destroy_process_group()
print("ok")
