"""Host-side logic of the reference's CPU solve loop, restated around the oracle walk.

Test helper (imports the oracle).  Follows core/kangaroo.cpp:326-379 (DP processing in g
order, reset of a kangaroo after a same-herd collision / duplicate), core/kangaroo.cpp:155-240
(CheckKey / CollisionCheck / AddToTable) and memory/hashtable.cpp:65-89,167-226 (Convert, Add,
128-bit compare).
"""
import numpy as np

import kgoracle as K

TAME, WILD = 0, 1
MASK128 = (1 << 128) - 1


class HashTable:
    """memory/hashtable.cpp: buckets keyed by x.bits64[2] & 0x3FFFF, entries compared on the
    low 128 bits of x only (hashtable.cpp:214-226)."""

    def __init__(self):
        self.entries = {}  # (h, x & MASK128) -> (X, D)

    def add(self, x, d, ktype):
        h, X, D = K.ht_convert(x, d, ktype)
        key = (h, X & MASK128)
        if key in self.entries:
            _, D0 = self.entries[key]
            if D0 == D:
                return "dup", None
            return "col", self.calc_dist_and_type(D0)
        self.entries[key] = (X, D)
        return "ok", None

    @staticmethod
    def calc_dist_and_type(D):
        # hashtable.cpp:181-191 (bits 62/63 of the SECOND 64-bit word carry type and sign)
        w1 = (D >> 64) & 0xFFFFFFFFFFFFFFFF
        ktype = 1 if (w1 & 0x4000000000000000) else 0
        sign = 1 if (w1 & 0x8000000000000000) else 0
        w1 &= 0x3FFFFFFFFFFFFFFF
        dist = (D & 0xFFFFFFFFFFFFFFFF) | (w1 << 64) | (D >> 128 << 128)
        if sign:
            dist = K.order_neg(dist)
        return dist, ktype

    def dump(self):
        return sorted((h, X, D) for (h, _), (X, D) in self.entries.items())


class Solver:
    def __init__(self, range_start, range_end, pub, dp_bits):
        self.range_start = range_start
        self.range_width = range_end - range_start
        self.range_power = self.range_width.bit_length()  # InitRange, core/kangaroo.cpp:649-660
        self.pub = pub
        # InitSearchKey (core/kangaroo.cpp:662-674): key - rangeStart*G
        if range_start != 0:
            rs = K.ec_mul_g(range_start)
            self.key = K.ec_add(pub, (rs[0], K.modneg(rs[1])))
        else:
            self.key = pub
        self.key_neg = (self.key[0], K.modneg(self.key[1]))
        self.dp_mask = 0 if dp_bits == 0 else (~((1 << (64 - min(dp_bits, 64))) - 1)) & 0xFFFFFFFFFFFFFFFF
        self.jd, self.jx, self.jy, _ = K.create_jump_table(self.range_power)
        self.table = HashTable()
        self.found = None
        self.dead = 0

    def check_key(self, d1, d2, t):
        # core/kangaroo.cpp:155-181 ; Point::equals compares y only (math/point.cpp:54-56)
        if t & 1:
            d1 = K.order_neg(d1)
        if t & 2:
            d2 = K.order_neg(d2)
        pk = K.order_add(d1, d2)
        P = K.ec_mul_g(pk)
        if P is None:
            return False
        if P[1] == self.key[1] or P[1] == self.key_neg[1]:
            pk = K.order_add(pk, self.range_start)
            PR = K.ec_mul_g(pk)
            if PR is not None and PR[1] == self.pub[1]:
                self.found = pk
                return True
        return False

    def collision_check(self, d1, t1, d2, t2):
        # core/kangaroo.cpp:183-230
        if t1 == t2:
            return False
        td, wd = (d1, d2) if t1 == TAME else (d2, d1)
        return any(self.check_key(td, wd, t) for t in range(4))

    def add_to_table(self, x, d, ktype):
        st, info = self.table.add(x, d, ktype)
        if st == "col":
            return self.collision_check(info[0], info[1], d, ktype)
        return st == "ok"

    def process_dps(self, dps, reset_cb):
        """dps: iterable of (kIdx, x, d) in emission order.  reset_cb(kIdx, ktype) re-seeds one
        kangaroo.  Returns True when the key has been found."""
        for kidx, x, d in dps:
            if self.found is not None:
                break
            ktype = kidx % 2
            if not self.add_to_table(x, d, ktype):
                reset_cb(kidx, ktype)
                self.dead += 1
        return self.found is not None


def solve_cpu(solver, px, py, d, rng, max_steps=10**9):
    """Kangaroo::SolveKeyCPU with the oracle walk: one step for the whole group, then DPs in g order."""
    n = px.shape[0]
    steps = 0

    def reset(kidx, ktype):
        # CreateHerd(1, ..., g % 2, false): core/kangaroo.cpp:369
        nx, ny, nd = K.create_herd(rng, solver.range_power, solver.range_width, solver.key, 1, ktype)
        px[kidx], py[kidx], d[kidx] = nx[0], ny[0], nd[0]

    while solver.found is None and steps < max_steps:
        found, buf = K.walk_raw(px, py, d, 1, solver.dp_mask, solver.jd, solver.jx, solver.jy, n)
        steps += 1
        if found:
            dps = []
            for i in range(found):
                e = buf[i]
                dps.append((int(e.kIdx), K.from_limbs(np.array(e.x[:], dtype=np.uint64)),
                            K.from_limbs(np.array(e.d[:], dtype=np.uint64))))
            solver.process_dps(dps, reset)
    return steps
