"""The C++ drop-in class itself (kangaroo_amd/dropin/GPUEngine_amd.cpp: the reference's `class GPUEngine`,
math/gpu/engine.h:13-54, on top of the C ABI), RUN on the GPU by tests/dropin_host.cpp the way SolveKeyGPU drives
it (core/kangaroo.cpp:408-500), and compared with the oracle: Int packing, Launch's item conversion, SetKangaroo
between launches, GetKangaroos with a launch in flight.  The binary is built by __graft_entry__.build()
(make dropin_test) against the stand-in headers of tests/dropin_stub — the reference tree is absent on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

import kgoracle as K
from gpu_common import make_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "dropin_host")


def test_cpp_class_matches_oracle(tmp_path):
    assert os.path.exists(BIN), "tests/_build/dropin_host missing: run __graft_entry__.build()"
    groups, per_group, launches, poke_after, poke_idx = 2, 32, 3, 0, 77
    n = 128 * groups * per_group
    case = make_case(64, n, seed=4711, dp_bits=10)
    nx, ny, nd = K.create_herd(case["rng"], 64, case["width"], case["key"], 1, poke_idx % 2)
    u64 = lambda *v: np.array(v, dtype=np.uint64)
    limbs = lambda v: u64(*[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])
    blob = np.concatenate([u64(groups, per_group, 131072, case["dp_mask"], launches, poke_after, poke_idx),
                           limbs(case["width"] >> 1), case["jd"].reshape(-1), case["jx"].reshape(-1), case["jy"].reshape(-1),
                           case["px"].reshape(-1), case["py"].reshape(-1), case["d"].reshape(-1),
                           nx.reshape(-1), ny.reshape(-1), nd.reshape(-1)]).astype(np.uint64)
    fin, fout = tmp_path / "case.bin", tmp_path / "out.bin"
    blob.tofile(fin)
    r = subprocess.run([BIN, str(fin), str(fout)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = np.fromfile(fout, dtype=np.uint64)
    assert int(out[0]) == n and int(out[1]) > n * 96 and int(out[2]) == groups * per_group and int(out[3]) == 128
    pos = 4
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    to_int = lambda a: sum(int(a[i]) << (64 * i) for i in range(4))
    walk = lambda: sorted((k, x, dd) for _, k, x, dd in
                          K.walk(px, py, d, 256, case["dp_mask"], case["jd"], case["jx"], case["jy"]))
    # Launch call l returns the DPs of kernel launch l and queues launch l + 1; SetKangaroo issued after call
    # `poke_after` therefore lands behind launch poke_after + 1
    for l in range(launches):
        cnt = int(out[pos])
        rec = out[pos + 1:pos + 1 + 9 * cnt].reshape(cnt, 9)
        pos += 1 + 9 * cnt
        got = sorted((int(r_[8]), to_int(r_[0:4]), to_int(r_[4:8])) for r_ in rec)
        ref = walk()
        assert len(ref) > 100 and got == ref, f"launch {l}"
        if l == poke_after + 1:
            px[poke_idx], py[poke_idx], d[poke_idx] = nx[0], ny[0], nd[0]
    assert poke_after + 1 < launches
    walk()  # the launch that was still in flight at GetKangaroos
    herd = out[pos:pos + 12 * n].reshape(3, n, 4)
    assert pos + 12 * n == len(out)
    assert np.array_equal(herd[0], px) and np.array_equal(herd[1], py) and np.array_equal(herd[2], d)
