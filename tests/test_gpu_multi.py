"""The N > 1 path tests ITSELF whenever more than one GPU is visible (``-m gpu``): ``bench.py --gpus N`` - one process per GPU
under ``torch.distributed.run``, RCCL (``nccl`` backend) over xGMI, heliostat i -> rank i mod N, all-reduce of the per-target flux,
all-gather of the control-point gradients - started as a fresh CHILD process (this process has initialised the GPU: a child,
never an exec), the way the reference leaves the launch to torchrun (artist/util/env.py:33-93,
tutorials/02_heliostat_raytracing_distributed_tutorial.py:60-75).  On a one-GPU box the test is skipped and says so; the
many-ranks-on-one-GPU rehearsal of the same code over gloo is tests/test_gpu_parity.py::test_two_processes_share_the_field."""
import json
import os
import pathlib
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_bench_shards_over_the_visible_gpus():
    count = torch.cuda.device_count()
    if count < 2:
        pytest.skip(f"{count} GPU visible: the RCCL path needs at least two (bench.py --gpus N runs it on the driver's 8-GPU node)")
    # at most four ranks: a GPU box allows six of this job's processes on its cards at once, and this process is one of them
    N = min(count, int(os.environ.get("ARTIST_TEST_MAX_RANKS", "4")))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("ARTIST_BENCH_BACKEND", None)                                  # RCCL, not the gloo rehearsal
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(N), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join((res.stdout + "\n" + res.stderr).splitlines()[-40:])
    assert res.returncode == 0, f"bench.py --gpus {N} failed (rc {res.returncode}):\n{tail}"
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}:\n{tail}"
    d = json.loads(lines[0])
    assert d["n_gpus"] == N and d["config"]["parallelism"] == f"heliostat-sharded dp{N}", d["config"]
    assert d["sharding_check"]["ok"] is True, d["sharding_check"]         # reduced flux == the field traced by ONE rank
    assert d["check"]["ray_counters_equal"] and d["check"]["flux_rel_l2"] < 1e-5, d["check"]
    # every rank reported its own device
    ranks = dict(re.findall(r"\[bench\] rank (\d+)/\d+ on (cuda:\d+)", res.stderr))
    assert len(ranks) == N and len(set(ranks.values())) == N, (ranks, tail)
    assert all(re.search(r"backend nccl", ln) for ln in res.stderr.splitlines() if ln.startswith("[bench] rank")), tail
