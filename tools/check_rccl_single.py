#!/usr/bin/env python3
"""Sanity check of the calls dsp_amd.sharding makes on the RCCL backend, with a world of ONE rank (a
one-GPU box cannot hold two RCCL ranks): process group with device_id, barrier, float64 MAX all-reduce."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402
from dsp_amd import sharding                   # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dist.barrier()
print("max over ranks:", sharding._max_over_ranks(dist, [1.5, 2.5], torch.device("cuda", 0)))
wall, ev = sharding.timed_region(lambda: None, torch.cuda.synchronize, 3, 1, dist, torch.device("cuda", 0))
print("timed_region ok", wall >= 0.0, ev)
dist.barrier()
dist.destroy_process_group()
print("ok")
