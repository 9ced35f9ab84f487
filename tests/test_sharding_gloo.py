"""Multi-GPU path on the CPU.  Two gloo ranks run the PRODUCT's rank body (dsp_amd.sharding.run_rank:
shard_range -> shard runner -> warm-up, barrier, K timed steps, barrier, MAX over ranks) with a CPU
stand-in for the device pipeline as the injected runner; the concatenated decisions equal the
single-process result and the timing reduction is the MAX.  Also: bench.py's own launcher wires the
rank environment the way torchrun does and starts N fresh processes without touching a GPU."""
import json
import os
import socket
import subprocess
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


class _StubRunner:
    """Stands in for sharding.PipelineRunner on a machine without a GPU: same interface, the
    oracle computes the shard's decisions (test infrastructure; the product runner is HIP)."""

    def __init__(self, lo, hi, rank):
        self.lo, self.hi, self.rank, self.steps, self.t = lo, hi, rank, 0, 0.0

    def step(self):
        self.steps += 1

    def sync(self):
        pass

    def timer_start(self):
        self.t = 0.0

    def timer_stop(self):
        return 0.25 + self.rank            # "device milliseconds": rank 1 is the slower one

    def result(self):
        from dsp_amd.synth import synth_burst
        from oracle import oracle as O
        h = O.rrc_filter(4, 10, 0.35)
        dec = np.zeros((self.hi - self.lo, 256), np.uint8)
        for b in range(self.lo, self.hi):
            y = O.filter_samples(synth_burst(b, h, n_sym=256), h).astype(np.complex64)
            dec[b - self.lo] = O.sync_chain_c64(y, want="dec")["dec"]
        return {"dec": dec, "steps": self.steps}


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from dsp_amd import sharding
    r, w, local = sharding.rank_env()
    assert (r, w, local) == (rank, world, rank)
    dist = sharding.init_clock_group(w, local, "gloo")
    rec = sharding.run_rank(total, r, w, lambda lo, hi: _StubRunner(lo, hi, r), steps=3, warmup=2, dist=dist)
    gathered = [None] * w
    dist.all_gather_object(gathered, rec)       # test-side collection only: the job itself exchanges nothing
    if r == 0:
        q.put(gathered)
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


def test_two_rank_gloo_rank_body(oracle):
    import torch.multiprocessing as mp
    from dsp_amd import sharding
    total, world = 10, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    recs = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r["timer_ms"] == 1.25 for r in recs)                 # MAX over ranks, known to both
    assert all(r["wall_s"] == recs[0]["wall_s"] for r in recs)
    assert all(r["result"]["steps"] == 5 for r in recs)             # 2 warm-up + 3 timed, every rank
    parts = sharding.collect(recs, total)
    dec = np.concatenate([p["dec"] for p in parts])
    from dsp_amd.synth import synth_burst
    h = oracle.rrc_filter(4, 10, 0.35)
    for b in range(total):
        y = oracle.filter_samples(synth_burst(b, h, n_sym=256), h).astype(np.complex64)
        assert np.array_equal(dec[b], oracle.sync_chain_c64(y, want="dec")["dec"])
    with pytest.raises(ValueError):
        sharding.collect(recs[:1], total)


def test_single_rank_body_without_dist():
    from dsp_amd import sharding

    class R:
        n = 0
        def step(self): R.n += 1
        def sync(self): pass
        def result(self): return R.n
    rec = sharding.run_rank(7, 0, 1, lambda lo, hi: R(), steps=4, warmup=1)
    assert (rec["lo"], rec["hi"], rec["result"]) == (0, 7, 5) and rec["wall_s"] >= 0.0


def test_rank_env():
    from dsp_amd import sharding
    assert sharding.rank_env({}) == (0, 1, 0)
    assert sharding.rank_env({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"}) == (3, 8, 3)
    assert sharding.rank_env({"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1", "DSP_BENCH_DEVICE": "0"}) == (1, 2, 0)
    with pytest.raises(ValueError):
        sharding.rank_env({"RANK": "2", "WORLD_SIZE": "2"})


def test_bench_launcher_env_wiring():
    import bench
    envs = bench.child_envs(4, 29511, {"PATH": "/bin", "DSP_BENCH_DEVICE": "0"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"
               for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["DSP_BENCH_DEVICE"] == "0" and e["PATH"] == "/bin"
               for e in envs)
    a = bench.parse_args(["--gpus", "8", "--scaling", "strong"])
    assert a.gpus == 8 and a.scaling == "strong" and a.bursts == 65536


def test_bench_launcher_starts_fresh_processes(tmp_path):
    """bench.launch() with a stand-in rank script: N processes, each with its own rank, and the
    parent returns rank 0's stdout.  The parent itself must not have imported torch or dsp_amd."""
    script = tmp_path / "rank.py"
    script.write_text(
        "import json, os, sys\n"
        "open(sys.argv[1] + '.' + os.environ['RANK'], 'w').write(os.environ['WORLD_SIZE'])\n"
        "print(json.dumps({'rank': int(os.environ['RANK']), 'n_gpus': int(os.environ['WORLD_SIZE']), 'argv': sys.argv[2:]}))\n")
    code = (
        "import sys, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        f"rc, out = bench.launch(3, [{str(tmp_path / 'mark')!r}, '--gpus', '3'], script={str(script)!r})\n"
        "assert 'torch' not in sys.modules and 'dsp_amd' not in sys.modules and 'numpy' not in sys.modules\n"
        "print(json.dumps({'rc': rc, 'out': out}))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["rc"] == 0
    line = json.loads(res["out"])
    assert line == {"rank": 0, "n_gpus": 3, "argv": ["--gpus", "3"]}
    assert sorted(p.name for p in tmp_path.glob("mark.*")) == ["mark.0", "mark.1", "mark.2"]
