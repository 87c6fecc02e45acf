"""The reference's multi-GPU solve shape (core/kangaroo.cpp:780-786, :475): one host thread + one engine per GPU,
ONE shared DP table, no collective; the key loop of Kangaroo::Run (:767-791); and -w save / -i resume
(storage/backup.cpp:101-150, 327-428; core/kangaroo.cpp:493-500).  CPU tests: the engines are the oracle stand-in
(tests/cpu_engine.py), everything else is the product's host code.  The same loops run on the GPU in
tests/test_gpu_configs.py."""
import os

import numpy as np

from cpu_engine import OracleEngine
from kangaroo_amd import host, solve

PRIVS = [0xB5A7C3D1F, 0x3C0FFEE42, 0x9A1B2C3D4]  # 36-bit keys
RANGE_END = (1 << 36) - 1


def test_two_threads_share_one_table():
    pub = host.ec_mul_g(PRIVS[0])
    key, st = solve.solve_key_threads(0, RANGE_END, pub, gpu_ids=(0, 1), dp_bits=3, seed=5, nb_run=64, max_launches=400,
                                      engine_factory=lambda g: OracleEngine(1024))
    assert key == PRIVS[0]
    assert st["gpus"] == [0, 1] and len(st["per_thread"]) == 2
    # both threads walked and fed the ONE table
    assert all(t["launches"] > 0 and t["dps"] > 0 for t in st["per_thread"])
    assert st["table"] + st["dead"] <= st["dps"] and st["table"] > 0
    assert st["jumps"] == sum(t["kangaroos"] * 64 * t["launches"] for t in st["per_thread"])


def test_key_loop_fresh_engine_per_key():
    made = []

    def factory(g):
        e = OracleEngine(1024)
        made.append(e)
        return e

    pubs = [host.ec_mul_g(p) for p in PRIVS]
    res = solve.solve_keys(0, RANGE_END, pubs, gpu_ids=(0,), dp_bits=3, seed=9, nb_run=64, max_launches=600,
                           engine_factory=factory)
    assert [r[0] for r in res] == PRIVS
    walked = [e for e in made if e.launches]
    assert len(walked) == len(PRIVS)  # one engine per key (the geometry probes never launch)
    assert all(r[1]["table"] > 0 for r in res)


def test_save_interrupt_and_resume(tmp_path):
    """An interrupted run leaves a work file with the table and the herd; the resumed run loads both and finds the
    key; the file's header carries the counters (Kangaroo::SaveWork / LoadWork)."""
    path = str(tmp_path / "work.kg")
    pub = host.ec_mul_g(PRIVS[1])
    # far too few launches to solve: stop after the first save
    key, st = solve.solve_key_threads(0, (1 << 44) - 1, host.ec_mul_g(0xABCDEF12345), gpu_ids=(0, 0), dp_bits=4,
                                      seed=7, nb_run=32, work_file=path, save_period=0.05, stop_after_saves=1,
                                      engine_factory=lambda g: OracleEngine(512), max_launches=100000)
    assert key is None and st["saves"] == 1 and os.path.exists(path)
    info = host.work_info(path)
    assert info["dp_bits"] == 4 and info["kangaroos"] == 1024 and info["items"] > 0
    assert info["total_count"] > 0 and info["total_count"] % (512 * 32) == 0

    # resume a solvable case: run 1 interrupted, run 2 resumes from its file
    path2 = str(tmp_path / "work2.kg")
    key, st1 = solve.solve_key_threads(0, RANGE_END, pub, gpu_ids=(0,), dp_bits=3, seed=11, nb_run=16, work_file=path2,
                                       save_period=0.0, stop_after_saves=1, engine_factory=lambda g: OracleEngine(1024))
    assert key is None and st1["saves"] == 1
    i2 = host.work_info(path2)
    assert i2["kangaroos"] == 1024 and i2["items"] == st1["table"]
    # the saved herd is on its track: P = d*G (+ key for wild)
    sv = host.Solver(0, RANGE_END, pub)
    px, py, d = sv.load_work(path2, 1024)
    assert sv.table_size == st1["table"] and len(px) == 1024
    hx, hy = host.ec_mul_g_batch(np.ascontiguousarray(d))
    for i in range(0, 1024, 37):
        pt = (host.to_int(hx[i]), host.to_int(hy[i]))
        if i & 1:
            pt = host.ec_add(pt, sv.key)
        assert pt == (host.to_int(px[i]), host.to_int(py[i]))
    key, st2 = solve.solve_key_threads(0, RANGE_END, pub, gpu_ids=(0,), dp_bits=3, seed=12, nb_run=64, resume_from=path2,
                                       max_launches=600, engine_factory=lambda g: OracleEngine(1024))
    assert key == PRIVS[1]
    assert st2["table"] >= st1["table"]  # the loaded table was kept
