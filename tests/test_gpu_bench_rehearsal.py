"""The N > 1 flow of bench.py with REAL engines on a one-GPU box: `bench.py --gpus 2 --rehearse-one-gpu` starts two
ranks (gloo for the barrier / reductions, both engines on device 0), each walks its own herd, and rank 0 prints ONE
line that sums the jumps, takes the slowest rank's time and carries the self-check of both ranks.  No scaling number
is read from this (two ranks share one GPU); it proves that the multi-rank path executes end to end.  The reference's
counterpart is one GPU thread per -gpuId entry (core/kangaroo.cpp:780-786)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_print_one_summed_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    steps = 3
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--grid", "128,128",
                        "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--check", "256"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    n = 128 * 128 * 128
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["kangaroos_per_gpu"] == n
    jumps = 2 * n * 256 * steps  # both ranks' herds
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * steps - jumps) < 1e-6 * jumps
    sc = r["config"]["self_check"]
    assert sc["bad"] == 0 and sc["sampled_per_gpu"] == 256 and sc["ranks_checked"] == 2
    assert len(r["roofline"]["board_power_median_W_per_rank"]) == 2
    assert "cpu_baseline" not in r
