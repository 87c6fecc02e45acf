"""The multi-rank solve loop (solve.solve_key_multi) on the CPU: two gloo ranks, each walking its own herd with the
oracle stand-in engine, the DP stream all-gathered and the DP table sharded over the ranks by x.  Also the
world_size-1 path of the same code, and the item exchange on ragged and empty inputs."""
import json
import os
import subprocess
import sys

from conftest import free_port
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIV = 0xB5A7C3D1F  # 36-bit key
RANGE_END = (1 << 36) - 1

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path[:0] = [%r, %r, %r]
    import numpy as np
    from kangaroo_amd import dist, host, solve
    from kangaroo_amd.engine import ITEM_DTYPE
    from cpu_engine import OracleEngine
    rank, local_rank, world = dist.init(backend="gloo")
    # 1. the exchange itself: ragged lengths, one rank empty
    mine = np.zeros(3 * rank, dtype=ITEM_DTYPE)          # rank 0 sends nothing, rank 1 three items
    mine["kIdx"] = 100 * rank + np.arange(len(mine))
    mine["x"][:, 0] = 7 + np.arange(len(mine))
    got = dist.all_gather_items(mine)
    none = dist.all_gather_items(np.zeros(0, dtype=ITEM_DTYPE))
    ctrl = dist.all_gather_array(np.array([rank, 10 + rank], dtype=np.uint64))
    # 2. a solve
    pub = host.ec_mul_g(%d)
    key, st = solve.solve_key_multi(0, %d, pub, dp_bits=3, seed=5, nb_run=64, max_launches=200,
                                    engine_factory=lambda lr: OracleEngine(1024))
    print("RESULT " + json.dumps({"rank": rank, "gathered_kidx": got["kIdx"].tolist(), "gathered_x0": got["x"][:, 0].tolist(),
                                  "none": len(none), "ctrl": ctrl.tolist(), "key": key, "launches": st["launches"],
                                  "table": st["table_this_rank"], "dps": st["dps"], "world": st["world"]}), flush=True)
""") % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), PRIV, RANGE_END)


def test_two_rank_solve_with_sharded_table(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, err = p.communicate(timeout=280)
        assert p.returncode == 0, err[-3000:]
        outs.append(o)
    res = sorted((json.loads(l.split("RESULT ", 1)[1]) for o in outs for l in o.splitlines() if l.startswith("RESULT ")),
                 key=lambda r: r["rank"])
    assert [r["rank"] for r in res] == [0, 1]
    for r in res:
        assert r["gathered_kidx"] == [100, 101, 102] and r["gathered_x0"] == [7, 8, 9] and r["none"] == 0
        assert r["ctrl"] == [[0, 10], [1, 11]]
        assert r["key"] == PRIV and r["world"] == 2
    # same loop on both ranks: same number of launches and DPs seen; the table is split between them
    assert res[0]["launches"] == res[1]["launches"] and res[0]["dps"] == res[1]["dps"]
    assert res[0]["table"] > 0 and res[1]["table"] > 0
    assert abs(res[0]["table"] - res[1]["table"]) < 0.2 * (res[0]["table"] + res[1]["table"])


def test_single_rank_runs_the_same_loop():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_engine import OracleEngine
    from kangaroo_amd import host, solve

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        assert k not in os.environ or os.environ[k] in ("0", "1")
    pub = host.ec_mul_g(PRIV)
    key, st = solve.solve_key_multi(0, RANGE_END, pub, dp_bits=3, seed=5, nb_run=64, max_launches=400,
                                    engine_factory=lambda lr: OracleEngine(1024))
    assert key == PRIV and st["world"] == 1 and st["table_this_rank"] > 0


FAIL_WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [%r, %r, %r]
    from kangaroo_amd import dist, host, solve
    from cpu_engine import OracleEngine
    rank, local_rank, world = dist.init(backend="gloo")

    class Flaky(OracleEngine):
        def Launch(self, spinWait=False, copy=True):
            if rank == 1 and self.launches == 2:
                raise RuntimeError("engine failure injected on rank 1")
            return super().Launch(spinWait)

    pub = host.ec_mul_g(%d)
    try:
        solve.solve_key_multi(0, (1 << 60) - 1, pub, dp_bits=3, seed=5, nb_run=16, max_launches=50,
                              engine_factory=lambda lr: Flaky(256))
    except RuntimeError as ex:
        print("RAISED " + str(ex), flush=True)
        sys.exit(3)
""") % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), PRIV)


def test_failing_rank_stops_the_others_instead_of_hanging(tmp_path):
    """A rank whose engine fails publishes that in the control exchange before raising; the other rank raises too
    (it would otherwise wait for ever in the next all-gather)."""
    script = tmp_path / "fail_worker.py"
    script.write_text(FAIL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [3, 3], outs
    assert "rank(s) [1] failed" in outs[0][0]
    assert "engine failure injected on rank 1" in outs[1][0]
