"""CPU tests of the product's host-side helpers (kangaroo_amd/csrc/host_secp.cpp) and of the C-ABI
library itself: symbols exported, set-up code equal to the reference's golden vectors, DP table and
collision logic reproducing the table the reference's own SolveKeyCPU run left behind."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kgoracle as K
from kangaroo_amd import engine as E
from kangaroo_amd import host as Hh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = lambda s: int(s, 16)


def test_library_exports_every_declared_symbol():
    L = C.CDLL(E.lib_path())
    declared = []
    for hdr in ("kangaroo_amd.h", "kangaroo_amd_host.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared += re.findall(r"\b(kgh?_[a-z0-9_]+)\s*\(", text)
    declared = sorted(set(declared))
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert sorted(set(E.ABI_SYMBOLS + Hh.HOST_SYMBOLS)) == declared


def test_no_gpu_calls_fail_loudly_without_device():
    """On a machine without a gfx950 the engine refuses to construct (there is no CPU fallback)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.KgError):
        E.GPUEngine(2, 64, 0, 1024)


def test_ec_and_pubkey_parsing(golden):
    for v in golden["ec"]:
        assert Hh.ec_mul_g(H(v["k"])) == (H(v["x"]), H(v["y"]))
        assert Hh.parse_pubkey(v["pub"]) == (H(v["x"]), H(v["y"]))
        assert Hh.parse_pubkey("04" + v["x"] + v["y"]) == (H(v["x"]), H(v["y"]))
    ks = K.ints_to_array([H(v["k"]) for v in golden["ec"]])
    bx, by = Hh.ec_mul_g_batch(ks)
    assert K.array_to_ints(bx) == [H(v["x"]) for v in golden["ec"]]
    assert K.array_to_ints(by) == [H(v["y"]) for v in golden["ec"]]
    for v in golden["ecadd"]:
        assert Hh.ec_add((H(v["x1"]), H(v["y1"])), (H(v["x2"]), H(v["y2"]))) == (H(v["rx"]), H(v["ry"]))
    with pytest.raises(ValueError):
        Hh.parse_pubkey("02" + "00" * 31 + "05")  # x = 5 is not on the curve
    with pytest.raises(ValueError):
        Hh.parse_pubkey("05" + "11" * 32)
    assert Hh.ec_mul_g(0) is None


def test_jump_tables_equal_reference(golden):
    for t in golden["jump_tables"]:
        jd, jx, jy, _ = Hh.create_jump_table(t["rangePower"])
        assert K.array_to_ints(jd) == [H(x) for x in t["jD"]]
        assert K.array_to_ints(jx) == [H(x) for x in t["jPx"]]
        assert K.array_to_ints(jy) == [H(x) for x in t["jPy"]]


def test_rng_is_the_reference_stream(golden):
    r = Hh.Rng(golden["rng"]["seed"])
    assert [r.next() for _ in range(8)] == golden["rng"]["draws"]


def test_synthetic_herd_points_are_distance_times_g():
    px, py, d = Hh.synthetic_herd(10000, d0=12345, stride=(1 << 100) + 77)
    ds = K.array_to_ints(d)
    assert ds[:3] == [12345, 12345 + (1 << 100) + 77, 12345 + 2 * ((1 << 100) + 77)]
    for i in (0, 1, 4095, 4096, 4097, 8191, 8192, 9999):
        assert ds[i] == (12345 + i * ((1 << 100) + 77)) % K.N
        assert (K.from_limbs(px[i]), K.from_limbs(py[i])) == K.ec_mul_g(ds[i])
    assert len({K.from_limbs(r) for r in px}) == 10000


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_product_solver_reproduces_reference_table(golden, idx):
    """Product herd creation + product DP table / collision logic, fed by the oracle walk (the checker
    stands in for the GPU here), must rebuild the reference's table and key exactly."""
    s = golden["solves"][idx]
    sv = Hh.Solver(H(s["rangeStart"]), H(s["rangeEnd"]), (H(s["pubx"]), H(s["puby"])))
    assert sv.range_power == s["rangePower"] and sv.key == (H(s["keyx"]), H(s["keyy"]))
    assert sv.wild_offset == (H(s["rangeEnd"]) - H(s["rangeStart"])) >> 1
    rng = Hh.Rng(s["seed"])
    px, py, d = Hh.create_herd(rng, sv.range_power, sv.range_width, sv.key, s["grp"], 0)
    assert [[K.from_limbs(px[i]), K.from_limbs(py[i]), K.from_limbs(d[i])] for i in range(s["grp"])] == \
        [[H(a), H(b), H(c)] for a, b, c in s["herd"]]
    jd, jx, jy, _ = Hh.create_jump_table(sv.range_power)
    dp_mask = H(s["dpMask"])
    n = s["grp"]
    while sv.found is None:
        found, buf = K.walk_raw(px, py, d, 1, dp_mask, jd, jx, jy, n)
        if not found:
            continue
        items = np.zeros(found, dtype=E.ITEM_DTYPE)
        for i in range(found):
            items[i]["x"] = buf[i].x[:]
            items[i]["d"] = buf[i].d[:]
            items[i]["kIdx"] = buf[i].kIdx
        rc, dead = sv.add_items(items)
        for di in dead:
            kidx = int(items[di]["kIdx"])
            nx, ny, nd = Hh.create_herd(rng, sv.range_power, sv.range_width, sv.key, 1, kidx % 2)
            px[kidx], py[kidx], d[kidx] = nx[0], ny[0], nd[0]
    assert sv.found == H(s["priv"])
    assert sv.dead == s["dead"]
    assert sv.dump() == sorted((h, H(x), H(dd)) for h, x, dd in s["table"])
