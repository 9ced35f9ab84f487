"""Multi-GPU path on the CPU: two gloo ranks shard a set of bursts with shard_range (no
data-path collective), each processes its block (the oracle stands in for the GPU kernel here),
and the concatenated decisions equal the single-process result.  Also checks the bench's timing
reduction (barrier + MAX over ranks)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dsp_amd.pipeline import shard_range
    from dsp_amd.synth import synth_burst
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    h = O.rrc_filter(4, 10, 0.35)
    dec = np.zeros((hi - lo, 256), np.uint8)
    for b in range(lo, hi):
        y = O.filter_samples(synth_burst(b, h, n_sym=256), h).astype(np.complex64)
        dec[b - lo] = O.sync_chain_c64(y, want="dec")["dec"]
    dist.barrier()
    t = torch.tensor([0.25 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, dec))
    if rank == 0:
        q.put((float(t[0]), gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from dsp_amd.pipeline import shard_range
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_two_rank_gloo_sharding(oracle):
    import torch.multiprocessing as mp
    total, world = 10, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    tmax, gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 1.25                                        # MAX over ranks
    gathered.sort(key=lambda g: g[0])
    assert gathered[0][0] == 0 and gathered[0][1] == gathered[1][0] and gathered[1][1] == total
    dec = np.concatenate([g[2] for g in gathered])
    from dsp_amd.synth import synth_burst
    h = oracle.rrc_filter(4, 10, 0.35)
    for b in range(total):
        y = oracle.filter_samples(synth_burst(b, h, n_sym=256), h).astype(np.complex64)
        assert np.array_equal(dec[b], oracle.sync_chain_c64(y, want="dec")["dec"])
