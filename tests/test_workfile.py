"""Work file in the reference's HEADW layout (storage/backup.cpp:256-428, memory/hashtable.cpp:13-21,262-324).

PINNED (round 3): tests/golden/ref_work_r32_dp8.bin.gz is a work file written by the reference's own
Kangaroo::SaveWork — oracle/ref_harness.cpp --savework, the reference's CPU translation units compiled where they lie
(oracle/Makefile, `make -C oracle workfile`) — after a real SolveKeyCPU run, and ref_work_r32_dp8.json is the same state
as the reference holds it in memory.  This repository's reader must read that file, its writer must reproduce it byte
for byte, and — in the container that has the reference, where oracle/_ref/ref_harness exists — the reference's own
LoadWork + FetchWalks must read a file written here.  The reference's tests hold no work file of their own."""
import gzip
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import kgoracle as K
from kangaroo_amd import host
from kangaroo_amd.engine import ITEM_DTYPE

H = lambda s: int(s, 16)


def _filled_solver(golden, idx=0):
    s = golden["solves"][idx]
    sv = host.Solver(H(s["rangeStart"]), H(s["rangeEnd"]), (H(s["pubx"]), H(s["puby"])))
    # feed the reference's own final table back as DPs: (h, x, d) with d in the packed form
    for h, x, dd in s["table"]:
        D = H(dd)
        typ = (D >> 126) & 1
        sign = (D >> 127) & 1
        dist = D & ((1 << 126) - 1)
        if sign:
            dist = (-dist) % K.N
        assert sv.add(H(x), dist, typ) == 1
    return s, sv


def test_layout_field_by_field(golden, tmp_path):
    s, sv = _filled_solver(golden)
    path = str(tmp_path / "work.bin")
    n = 64
    px, py, d = host.create_herd(host.Rng(1), sv.range_power, sv.range_width, sv.key, n, 0)
    sv.save_work(path, 6, total_count=123456789, total_time=42.5, herd=(px, py, d))
    raw = open(path, "rb").read()
    magic, version, dp = struct.unpack_from("<III", raw, 0)
    assert magic == 0xFA6A8001 and version == 0 and dp == 6               # constants.h:43, backup.cpp:256-271
    ints = [int.from_bytes(raw[12 + 32 * i: 44 + 32 * i], "little") for i in range(4)]
    assert ints == [H(s["rangeStart"]), H(s["rangeEnd"]), H(s["pubx"]), H(s["puby"])]
    total_count, total_time = struct.unpack_from("<Qd", raw, 140)
    assert total_count == 123456789 and total_time == 42.5
    off = 156
    seen = []
    for h in range(1 << 18):                                              # hashtable.cpp:262-285
        nb, cap = struct.unpack_from("<II", raw, off)
        off += 8
        assert cap >= nb and (nb == 0) == (cap == 0)
        prev = -1
        for _ in range(nb):
            x = int.from_bytes(raw[off: off + 16], "little")
            D = int.from_bytes(raw[off + 16: off + 32], "little")
            off += 32
            assert x > prev                                               # ascending by (i64[1], i64[0])
            prev = x
            seen.append((h, x, D))
    ref = sorted((h, H(x) & ((1 << 128) - 1), H(dd) & ((1 << 128) - 1)) for h, x, dd in s["table"])
    assert sorted(seen) == ref                                            # the reference's own table, entry for entry
    (walks,) = struct.unpack_from("<Q", raw, off)
    off += 8
    assert walks == n and len(raw) == off + n * 96                        # backup.cpp:389-410
    first = [int.from_bytes(raw[off + 32 * i: off + 32 * i + 32], "little") for i in range(3)]
    assert first == [host.to_int(px[0]), host.to_int(py[0]), host.to_int(d[0])]


def test_save_info_load_round_trip(golden, tmp_path):
    s, sv = _filled_solver(golden, 1)
    path = str(tmp_path / "work2.bin")
    n = 33
    px, py, d = host.create_herd(host.Rng(2), sv.range_power, sv.range_width, sv.key, n, 1)
    sv.save_work(path, 6, total_count=7, total_time=1.25, herd=(px, py, d))
    info = host.work_info(path)
    assert info == {"dp_bits": 6, "range_start": H(s["rangeStart"]), "range_end": H(s["rangeEnd"]),
                    "key": (H(s["pubx"]), H(s["puby"])), "total_count": 7, "total_time": 1.25,
                    "items": len(s["table"]), "kangaroos": n}
    sv2 = host.Solver(H(s["rangeStart"]), H(s["rangeEnd"]), (H(s["pubx"]), H(s["puby"])))
    lx, ly, ld = sv2.load_work(path, max_kangaroos=100)
    assert np.array_equal(lx, px) and np.array_equal(ly, py) and np.array_equal(ld, d)
    low = lambda t: sorted((h, x & ((1 << 128) - 1), dd & ((1 << 128) - 1)) for h, x, dd in t)
    assert low(sv2.dump()) == low(sv.dump()) and sv2.table_size == len(s["table"])
    # a loaded table keeps working: the DP that solved the reference run solves it again
    other = host.Solver(0, 1, (H(s["pubx"]), H(s["puby"])))
    try:
        other.load_work(path)
        assert False, "a file for another range must be refused"
    except RuntimeError as e:
        assert "-5" in str(e)
    # empty file without kangaroos
    sv.save_work(path, 6)
    assert host.work_info(path)["kangaroos"] == 0


GOLD = os.path.join(os.path.dirname(__file__), "golden")
HARNESS = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "ref_harness")
LOW = (1 << 128) - 1


def _ref_file(tmp_path):
    raw = gzip.open(os.path.join(GOLD, "ref_work_r32_dp8.bin.gz")).read()
    path = str(tmp_path / "ref_work.bin")
    open(path, "wb").write(raw)
    return path, raw, json.load(open(os.path.join(GOLD, "ref_work_r32_dp8.json")))


def test_reads_the_file_the_reference_wrote(tmp_path):
    path, raw, j = _ref_file(tmp_path)
    n = len(j["herd"])
    assert host.work_info(path) == {
        "dp_bits": j["dp"], "range_start": H(j["rangeStart"]), "range_end": H(j["rangeEnd"]),
        "key": (H(j["pubx"]), H(j["puby"])), "total_count": j["totalCount"], "total_time": j["totalTime"],
        "items": len(j["table"]), "kangaroos": n}
    sv = host.Solver(H(j["rangeStart"]), H(j["rangeEnd"]), (H(j["pubx"]), H(j["puby"])))
    px, py, d = sv.load_work(path, max_kangaroos=n + 7)
    assert len(px) == n
    got = [(host.to_int(px[i]), host.to_int(py[i]), host.to_int(d[i])) for i in range(n)]
    assert got == [(H(x), H(y), H(dd)) for x, y, dd in j["herd"]]
    # the file keeps 128 bits of x and the packed 128-bit distance of every entry (hashtable.cpp:13-21)
    assert [(h, x & LOW, dd & LOW) for h, x, dd in sv.dump()] == sorted((h, H(x) & LOW, H(dd) & LOW) for h, x, dd in j["table"])
    # the kangaroos of the file are on their tracks: P = d*G for the tame ones, d*G + (pub - rangeStart*G) for the wild ones
    sx, sy = host.ec_mul_g(H(j["rangeStart"]))
    key = host.ec_add((H(j["pubx"]), H(j["puby"])), (sx, K.P - sy))
    for i, (x, y, dd) in enumerate(got):
        pt = host.ec_mul_g(dd)
        assert (x, y) == (pt if i % 2 == 0 else host.ec_add(pt, key))


def test_writes_the_file_the_reference_wrote_byte_for_byte(tmp_path):
    path, raw, j = _ref_file(tmp_path)
    n = len(j["herd"])
    sv = host.Solver(H(j["rangeStart"]), H(j["rangeEnd"]), (H(j["pubx"]), H(j["puby"])))
    herd = sv.load_work(path, max_kangaroos=n)
    out = str(tmp_path / "ours.bin")
    sv.save_work(out, j["dp"], total_count=j["totalCount"], total_time=j["totalTime"], herd=herd)
    assert open(out, "rb").read() == raw
    # and from a table built DP by DP instead of loaded: same bytes (bucket capacities follow the reference's growth)
    sv2 = host.Solver(H(j["rangeStart"]), H(j["rangeEnd"]), (H(j["pubx"]), H(j["puby"])))
    for h, x, dd in j["table"]:
        D = H(dd)
        dist = D & ((1 << 126) - 1)
        if (D >> 127) & 1:
            dist = (-dist) % K.N
        assert sv2.add(H(x), dist, (D >> 126) & 1) == 1
    sv2.save_work(out, j["dp"], total_count=j["totalCount"], total_time=j["totalTime"], herd=herd)
    assert open(out, "rb").read() == raw


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness is built only where the reference is")
def test_the_reference_reads_the_file_written_here(golden, tmp_path):
    s, sv = _filled_solver(golden, 1)                                     # range that starts at 2^44
    n = 40
    px, py, d = host.create_herd(host.Rng(5), sv.range_power, sv.range_width, sv.key, n, 0)
    path, out = str(tmp_path / "ours.bin"), str(tmp_path / "ref_read.json")
    sv.save_work(path, 6, total_count=987654321, total_time=3.5, herd=(px, py, d))
    subprocess.run([HARNESS, "--loadwork", path, out], check=True, stdout=subprocess.DEVNULL, timeout=120)
    j = json.load(open(out))
    assert (j["dp"], H(j["rangeStart"]), H(j["rangeEnd"]), H(j["pubx"]), H(j["puby"]), j["totalCount"], j["totalTime"], j["left"]) == \
        (6, H(s["rangeStart"]), H(s["rangeEnd"]), H(s["pubx"]), H(s["puby"]), 987654321, 3.5, 0)
    assert [(H(x), H(y), H(dd)) for x, y, dd in j["herd"]] == \
        [(host.to_int(px[i]), host.to_int(py[i]), host.to_int(d[i])) for i in range(n)]
    assert sorted((h, H(x) & LOW, H(dd) & LOW) for h, x, dd in j["table"]) == \
        sorted((h, H(x) & LOW, H(dd) & LOW) for h, x, dd in s["table"])


def test_large_table_round_trip_through_the_worker_pool(tmp_path):
    """120 000 random DPs (every shard, every partition of the file): saved and loaded on the solver's worker pool, the
    loaded table equals the saved one entry for entry, and saving it again gives the same bytes."""
    from kangaroo_amd.engine import ITEM_DTYPE

    n = 120_000
    rnd = np.random.default_rng(77)
    it = np.zeros(n, dtype=ITEM_DTYPE)
    it["x"] = rnd.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    it["d"][:, :2] = rnd.integers(0, 1 << 62, size=(n, 2), dtype=np.uint64)
    it["d"][::3, 3] = np.uint64(0xFFFFFFFFFFFFFFFF)          # some negative distances (mod n), HashTable::Convert's sign bit
    it["d"][::3, 2] = np.uint64(0xFFFFFFFFFFFFFFFE)
    it["kIdx"] = rnd.integers(0, 1 << 20, size=n, dtype=np.uint64)
    pub = K.ec_mul_g(0x1234567)
    sv = host.Solver(0, (1 << 64) - 1, pub)
    sv.set_threads(8)
    rc, dead = sv.add_items(it)
    assert rc == 0 and len(dead) == 0 and sv.table_size == n
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    sv.save_work(a, 9, total_count=5, total_time=0.5)
    assert host.work_info(a)["items"] == n
    sv2 = host.Solver(0, (1 << 64) - 1, pub)
    sv2.set_threads(8)
    sv2.load_work(a)
    assert sv2.table_size == n
    low = lambda t: sorted((h, x & LOW, dd & LOW) for h, x, dd in t)  # the file keeps the low 128 bits of x
    assert low(sv2.dump()) == low(sv.dump())
    sv2.save_work(b, 9, total_count=5, total_time=0.5)
    assert open(a, "rb").read() == open(b, "rb").read()
    rc, dead = sv2.add_items(it)                              # every DP is found again in the loaded table
    assert len(dead) == n and sv2.table_size == n
