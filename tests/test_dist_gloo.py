"""world_size-2 gloo test of the multi-GPU plumbing (CPU): disjoint herd shards, config broadcast,
max-over-ranks timing reduction."""
import os
import subprocess
import sys

from conftest import free_port
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    from kangaroo_amd import dist, host
    rank, local_rank, world = dist.init(backend="gloo")
    cfg = np.arange(10, dtype=np.uint64) * (rank + 1)      # only rank 0's content must survive
    cfg = dist.broadcast_config(cfg)
    d0, stride = dist.herd_shard(rank, world, 1000)
    px, py, d = host.synthetic_herd(1000, d0=d0, stride=stride)
    dist.barrier()
    mx = dist.max_over_ranks(1.0 + rank)
    sm = dist.sum_over_ranks(10.0 * (rank + 1))
    first = int(d[0][0]) | (int(d[0][1]) << 64) | (int(d[0][2]) << 128) | (int(d[0][3]) << 192)
    last = int(d[-1][0]) | (int(d[-1][1]) << 64) | (int(d[-1][2]) << 128) | (int(d[-1][3]) << 192)
    print("RESULT " + json.dumps({"rank": rank, "world": world, "cfg": cfg.tolist(), "max": mx, "sum": sm,
                                  "first": str(first), "last": str(last), "x0": int(px[0][0])}), flush=True)
""") % ROOT


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        o, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        outs.append(o)
    import json

    res = sorted((json.loads(l.split("RESULT ", 1)[1]) for o in outs for l in o.splitlines() if l.startswith("RESULT ")),
                 key=lambda r: r["rank"])
    assert [r["rank"] for r in res] == [0, 1] and all(r["world"] == 2 for r in res)
    assert res[0]["cfg"] == res[1]["cfg"] == list(range(10))
    assert res[0]["max"] == res[1]["max"] == 2.0
    assert res[0]["sum"] == res[1]["sum"] == 30.0
    # disjoint, adjacent progressions: rank 1 starts one stride after rank 0's last distance
    N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    stride = 0x9E3779B97F4A7C15F39CC0605CEDC835
    assert int(res[1]["first"]) == (int(res[0]["last"]) + stride) % N
    assert res[0]["x0"] != res[1]["x0"]
