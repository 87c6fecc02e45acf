"""bench.py --gpus N must start N ranks by itself when no launcher did (the driver may call it either way), label the
line with the number of ranks that really ran, sum the jumps over them and take the slowest rank's time.  CPU test
of that plumbing (--plumbing-only: no engine, gloo); the measured path is the same code with the engine in it."""
import json
import os
import subprocess
import sys

from conftest import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}


def test_self_launch_two_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "--grid", "4,4", "--steps", "3",
                        "--nb-run", "256"], env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1  # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["valid"] is False
    assert r["jumps"] == 2 * (128 * 4 * 4) * 256 * 3  # summed over both ranks
    assert r["elapsed_s"] >= 0.02  # rank 1 sleeps 20 ms: the max over ranks sets the clock


def test_under_a_launcher_world_must_match_gpus():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], env=env, capture_output=True, text=True,
                       timeout=280)
    assert p.returncode == 2 and "--gpus 2 but 1 rank" in p.stderr and p.stdout.strip() == ""


def test_torchrun_style_environment_two_ranks():
    """As the driver launches it: one process per rank with RANK / WORLD_SIZE / MASTER_* set."""
    for attempt in range(3):  # the port is picked by bind-and-close: another process may take it before rank 0 binds it
        env = dict(_clean_env(), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "--grid", "2,2", "--steps", "5"],
                                  env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                  text=True) for r in range(2)]
        outs = [p.communicate(timeout=280) for p in procs]
        if all(p.returncode == 0 for p in procs) or not any("address already in use" in e.lower() or "EADDRINUSE" in e for _, e in outs):
            break
    assert all(p.returncode == 0 for p in procs), outs
    json_lines = [l for o, _ in outs for l in o.splitlines() if l.startswith("{")]
    assert len(json_lines) == 1
    assert json.loads(json_lines[0])["jumps"] == 2 * (128 * 2 * 2) * 256 * 5
