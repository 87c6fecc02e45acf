"""Work file in the reference's HEADW layout (storage/backup.cpp:256-428, memory/hashtable.cpp:13-21,262-324).

PARITY UNPINNED for the file as a whole: the reference's tests hold no work file and binaries built from the
reference are not run in this project, so the layout is checked field by field against the offsets and sizes read
from the reference's source.  The 32-byte entries themselves (HashTable::Convert packing) are pinned: the table
saved here is the one that reproduces the reference's DP table in tests/test_host_lib.py."""
import struct

import numpy as np

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
