"""Python binding of the host-side helpers (include/kangaroo_amd_host.h): curve set-up and the DP
table / collision logic the reference runs on the CPU around the GPU engine."""
import ctypes as C

import numpy as np

from .engine import ITEM_DTYPE, load_library

_u64p = C.POINTER(C.c_uint64)
_ready = False

HOST_SYMBOLS = [
    "kgh_ec_mul_g", "kgh_ec_mul_g_batch", "kgh_ec_add", "kgh_parse_pubkey", "kgh_create_jump_table", "kgh_rng_new",
    "kgh_rng_free", "kgh_rng_next", "kgh_create_herd", "kgh_herd_distances", "kgh_comb_table", "kgh_synthetic_herd", "kgh_solver_new", "kgh_solver_free", "kgh_solver_set_threads", "kgh_work_save", "kgh_work_info", "kgh_work_load",
    "kgh_solver_range_power", "kgh_solver_wild_offset", "kgh_solver_range_width", "kgh_solver_search_key",
    "kgh_solver_add", "kgh_solver_add_items", "kgh_solver_found", "kgh_solver_table_size", "kgh_solver_dead_count", "kgh_solver_unstored_count",
    "kgh_solver_dump",
]


def _lib():
    global _ready
    L = load_library()
    if not _ready:
        V = C.c_void_p
        sig = {
            "kgh_ec_mul_g": ([_u64p, _u64p, _u64p], C.c_int),
            "kgh_ec_mul_g_batch": ([_u64p, _u64p, _u64p, C.c_size_t], C.c_int),
            "kgh_ec_add": ([_u64p] * 6, C.c_int),
            "kgh_parse_pubkey": ([C.c_char_p, _u64p, _u64p], C.c_int),
            "kgh_create_jump_table": ([C.c_int, _u64p, _u64p, _u64p], C.c_int),
            "kgh_rng_new": ([C.c_uint32], V),
            "kgh_rng_free": ([V], None),
            "kgh_rng_next": ([V], C.c_uint32),
            "kgh_create_herd": ([V, C.c_int, _u64p, _u64p, _u64p, C.c_int, C.c_size_t, _u64p, _u64p, _u64p], C.c_int),
            "kgh_herd_distances": ([V, C.c_int, _u64p, C.c_int, C.c_size_t, _u64p], C.c_int),
            "kgh_comb_table": ([C.POINTER(C.c_uint32), _u64p, _u64p], C.c_int),
            "kgh_solver_set_threads": ([V, C.c_int], None),
            "kgh_work_save": ([V, C.c_char_p, C.c_uint32, C.c_uint64, C.c_double, _u64p, _u64p, _u64p, C.c_uint64], C.c_int),
            "kgh_work_info": ([C.c_char_p, C.POINTER(C.c_uint32), _u64p, _u64p, _u64p, _u64p, C.POINTER(C.c_uint64),
                               C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)], C.c_int),
            "kgh_work_load": ([V, C.c_char_p, _u64p, _u64p, _u64p, C.c_uint64, C.POINTER(C.c_uint64)], C.c_int),
            "kgh_synthetic_herd": ([_u64p, _u64p, C.c_size_t, _u64p, _u64p, _u64p], C.c_int),
            "kgh_solver_new": ([_u64p, _u64p, _u64p, _u64p], V),
            "kgh_solver_free": ([V], None),
            "kgh_solver_range_power": ([V], C.c_int),
            "kgh_solver_wild_offset": ([V, _u64p], None),
            "kgh_solver_range_width": ([V, _u64p], None),
            "kgh_solver_search_key": ([V, _u64p, _u64p], None),
            "kgh_solver_add": ([V, _u64p, _u64p, C.c_uint64], C.c_int),
            "kgh_solver_add_items": ([V, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)], C.c_int),
            "kgh_solver_found": ([V, _u64p], C.c_int),
            "kgh_solver_table_size": ([V], C.c_uint64),
            "kgh_solver_dead_count": ([V], C.c_uint64),
            "kgh_solver_unstored_count": ([V], C.c_uint64),
            "kgh_solver_dump": ([V, _u64p, C.c_uint64], C.c_uint64),
        }
        for name, (a, r) in sig.items():
            fn = getattr(L, name)
            fn.argtypes = a
            fn.restype = r
        _ready = True
    return L


def limbs(v):
    v &= (1 << 256) - 1
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def to_int(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(4))


def _p(a):
    return a.ctypes.data_as(_u64p)


def ec_mul_g(k):
    x, y = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    ok = _lib().kgh_ec_mul_g(_p(x), _p(y), _p(limbs(k)))
    return (to_int(x), to_int(y)) if ok else None


def ec_mul_g_batch(ks):
    ks = np.ascontiguousarray(ks, dtype=np.uint64)
    x, y = np.zeros_like(ks), np.zeros_like(ks)
    _lib().kgh_ec_mul_g_batch(_p(x), _p(y), _p(ks), ks.shape[0])
    return x, y


def ec_add(p1, p2):
    rx, ry = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    rc = _lib().kgh_ec_add(_p(rx), _p(ry), _p(limbs(p1[0])), _p(limbs(p1[1])), _p(limbs(p2[0])), _p(limbs(p2[1])))
    return None if rc else (to_int(rx), to_int(ry))


def parse_pubkey(hexstr):
    x, y = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    rc = _lib().kgh_parse_pubkey(hexstr.encode(), _p(x), _p(y))
    if rc:
        raise ValueError(f"bad public key ({rc})")
    return to_int(x), to_int(y)


def create_jump_table(range_power):
    jd, jx, jy = (np.zeros((64, 4), np.uint64) for _ in range(3))
    tries = _lib().kgh_create_jump_table(range_power, _p(jd), _p(jx), _p(jy))
    return jd, jx, jy, tries


class Rng:
    def __init__(self, seed):
        self._h = _lib().kgh_rng_new(seed & 0xFFFFFFFF)

    def next(self):
        return _lib().kgh_rng_next(self._h)

    def __del__(self):
        try:
            _lib().kgh_rng_free(self._h)
        except Exception:
            pass


def create_herd(rng, range_power, range_width, key, n, first_type=0):
    px, py, d = (np.zeros((n, 4), np.uint64) for _ in range(3))
    rc = _lib().kgh_create_herd(rng._h, range_power, _p(limbs(range_width)), _p(limbs(key[0])), _p(limbs(key[1])),
                                first_type, n, _p(px), _p(py), _p(d))
    if rc:
        raise RuntimeError("kgh_create_herd failed")
    return px, py, d


def herd_distances(rng, range_power, range_width, n, first_type=0):
    d = np.zeros((n, 4), np.uint64)
    rc = _lib().kgh_herd_distances(rng._h, range_power, _p(limbs(range_width)), first_type, n, _p(d))
    if rc:
        raise RuntimeError("kgh_herd_distances failed")
    return d


def synthetic_herd(n, d0=0x1234567, stride=0x9E3779B97F4A7C15F39CC0605CEDC835):
    px, py, d = (np.zeros((n, 4), np.uint64) for _ in range(3))
    _lib().kgh_synthetic_herd(_p(limbs(d0)), _p(limbs(stride)), n, _p(px), _p(py), _p(d))
    return px, py, d


def work_info(path):
    """Kangaroo::WorkInfo: header of a work file."""
    dp, tc, tt, ni, nk = C.c_uint32(0), C.c_uint64(0), C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    rs, re, kx, ky = (np.zeros(4, np.uint64) for _ in range(4))
    rc = _lib().kgh_work_info(path.encode(), C.byref(dp), _p(rs), _p(re), _p(kx), _p(ky), C.byref(tc), C.byref(tt),
                              C.byref(ni), C.byref(nk))
    if rc:
        raise RuntimeError(f"kgh_work_info failed ({rc})")
    return {"dp_bits": dp.value, "range_start": to_int(rs), "range_end": to_int(re), "key": (to_int(kx), to_int(ky)),
            "total_count": tc.value, "total_time": tt.value, "items": ni.value, "kangaroos": nk.value}


class Solver:
    """DP table + collision resolution (HashTable / AddToTable / CollisionCheck / CheckKey)."""

    def __init__(self, range_start, range_end, pub):
        L = _lib()
        self._h = L.kgh_solver_new(_p(limbs(range_start)), _p(limbs(range_end)), _p(limbs(pub[0])), _p(limbs(pub[1])))
        self.range_power = L.kgh_solver_range_power(self._h)
        a = np.zeros(4, np.uint64)
        L.kgh_solver_wild_offset(self._h, _p(a))
        self.wild_offset = to_int(a)
        L.kgh_solver_range_width(self._h, _p(a))
        self.range_width = to_int(a)
        x, y = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
        L.kgh_solver_search_key(self._h, _p(x), _p(y))
        self.key = (to_int(x), to_int(y))

    def set_threads(self, threads):
        _lib().kgh_solver_set_threads(self._h, threads)

    def add(self, x, d, k_idx):
        return _lib().kgh_solver_add(self._h, _p(limbs(x)), _p(limbs(d)), k_idx)

    def add_items(self, items):
        """items: structured array (ITEM_DTYPE).  Returns (status, indices needing a reset)."""
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n = len(items)
        dead = np.zeros(max(n, 1), dtype=np.uint32)
        nd = C.c_uint32(0)
        rc = _lib().kgh_solver_add_items(self._h, items.ctypes.data, n, dead.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         C.byref(nd))
        return rc, dead[: nd.value]

    @property
    def found(self):
        a = np.zeros(4, np.uint64)
        return to_int(a) if _lib().kgh_solver_found(self._h, _p(a)) else None

    @property
    def table_size(self):
        return int(_lib().kgh_solver_table_size(self._h))

    @property
    def dead(self):
        return int(_lib().kgh_solver_dead_count(self._h))

    @property
    def unstored(self):
        """Items handed in that are neither in the table nor dead (the DP that solved the key and what arrived after
        it): table_size + dead + unstored == items handed in."""
        return int(_lib().kgh_solver_unstored_count(self._h))

    def save_work(self, path, dp_bits, total_count=0, total_time=0.0, herd=None):
        """Kangaroo::SaveWork: header + DP table (+ kangaroos if herd = (px, py, d) is given)."""
        if herd is None:
            rc = _lib().kgh_work_save(self._h, path.encode(), dp_bits, total_count, total_time, None, None, None, 0)
        else:
            px, py, d = (np.ascontiguousarray(a, dtype=np.uint64) for a in herd)
            rc = _lib().kgh_work_save(self._h, path.encode(), dp_bits, total_count, total_time, _p(px), _p(py), _p(d), len(px))
        if rc:
            raise RuntimeError(f"kgh_work_save failed ({rc})")

    def load_work(self, path, max_kangaroos=0):
        """Kangaroo::LoadWork: table into this solver; returns (px, py, d) of the saved kangaroos (up to max)."""
        px, py, d = (np.zeros((max(max_kangaroos, 1), 4), np.uint64) for _ in range(3))
        got = C.c_uint64(0)
        rc = _lib().kgh_work_load(self._h, path.encode(), _p(px), _p(py), _p(d), max_kangaroos, C.byref(got))
        if rc:
            raise RuntimeError(f"kgh_work_load failed ({rc})")
        return px[: got.value], py[: got.value], d[: got.value]

    def dump(self):
        n = self.table_size
        out = np.zeros((max(n, 1), 9), dtype=np.uint64)
        m = int(_lib().kgh_solver_dump(self._h, _p(out), n))
        return sorted((int(r[0]), to_int(r[1:5]), to_int(r[5:9])) for r in out[:m])

    def __del__(self):
        try:
            _lib().kgh_solver_free(self._h)
        except Exception:
            pass
