"""Two ranks of the product's sharded job on ONE GPU (two child processes, device 0): the
concatenated decisions / mu / symbol counts equal a single-process run of the whole job bit for
bit, and `bench.py --gpus 2` launches its own ranks and reports n_gpus 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_one_gpu_bit_identical(ctx, tmp_path):
    import bench
    from dsp_amd import sharding
    total, n_sym = 200, 256                      # 100 bursts per rank: partly filled tiles on both
    prefix = str(tmp_path / "shard")
    envs = bench.child_envs(2, bench.free_port(), dict(os.environ, DSP_BENCH_DEVICE="0", PYTHONPATH=ROOT))
    cmd = [sys.executable, "-m", "dsp_amd.shard_worker", "--total", str(total), "--n-sym", str(n_sym),
           "--out", prefix]
    procs = [subprocess.Popen(cmd, env=e, cwd=ROOT) for e in envs]
    for p in procs:
        assert p.wait(timeout=300) == 0
    recs = []
    for r in range(2):
        z = np.load(f"{prefix}.rank{r}.npz")
        recs.append({"lo": int(z["lo"]), "hi": int(z["hi"]),
                     "result": {"dec": z["dec"], "mu": z["mu"], "n_out": z["n_out"]}, "wall_s": float(z["wall_s"])})
    assert recs[0]["wall_s"] == recs[1]["wall_s"]                 # MAX over ranks, known to both
    parts = sharding.collect(recs, total)
    one = sharding.run_rank(total, 0, 1, lambda lo, hi: sharding.PipelineRunner(ctx, lo, hi, n_sym, 4,
                                                                               want=("dec", "mu")))["result"]
    for k in ("dec", "mu", "n_out"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), one[k]), k
    assert (one["n_out"] == n_sym).all()


def test_bench_gpus_2_launches_its_own_ranks():
    env = dict(os.environ, DSP_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bursts", "4096",
                        "--steps", "3", "--warmup", "1", "--no-capture", "--fir-reps", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # stdout is the JSON line and nothing else (the process-group libraries print to stdout too)
    assert len(r.stdout.splitlines()) == 1, r.stdout[:500]
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["bursts_total"] == 8192 and cfg["bursts_per_gpu"] == 4096
    assert cfg["strong_scaling"]["bursts_total"] == 4096 and cfg["strong_scaling"]["value"] > 0
    assert cfg["weak_scaling"]["value"] == line["value"]
    assert line["roofline"]["chain"]["frac"] < line["roofline"]["frac"]
    # the N > 1 line is self-contained: the CPU baseline (a 3-second sample) rides on rank 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"


def test_rccl_clock_group_with_one_rank():
    """The `nccl` (= RCCL) branch of sharding.init_clock_group, which bench.py takes for N > 1 on a
    multi-GPU node: group creation with device_id, barrier, MAX all-reduce of float64 on the device and
    timed_region's fences -- with the one rank a one-GPU box can hold, in a child process."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(__import__("bench").free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rccl_single.py")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert r.stdout.strip().endswith("ok")
