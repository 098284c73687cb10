"""bench.py run live on the MI355X: the line it prints is held to the contract, for one rank and for the self-launched
two-rank form (``python bench.py --gpus 2`` starts its own ranks; on a one-GPU box both share the card)."""
import json
import os
import subprocess
import sys

import pytest

from test_bench_contract import check_bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
LEGS_OFF = ["--no-distill-mix", "--no-ddim", "--no-unfrozen", "--no-compos", "--no-zs-frontend", "--no-clock-probe"]


def run_bench(*args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_line_live_one_gpu():
    d = run_bench("--steps", "3", "--warmup", "2", "--no-cpu-k8", *LEGS_OFF)
    check_bench_line(d, n_gpus=1)
    assert "roofline" in d and "cpu_baseline" in d
    assert d["steps"] == 3 and d["warmup"] == 2 and d["value"] > 20
    assert d["cpu_baseline"]["cpu_model"] and len(d["cpu_baseline"]["timed_seconds"]) >= 3


def test_bench_spawns_its_own_two_ranks():
    d = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", *LEGS_OFF)
    check_bench_line(d, n_gpus=2)
    assert d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
    assert d["config"]["grad_allreduce_bytes"] > 5e8 and d["config"]["dist_backend"] in ("nccl", "gloo")
