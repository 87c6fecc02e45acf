"""ctypes binding for the CPU oracle (oracle/kg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (kangaroo_amd/) never imports it.

All 256-bit values cross this boundary as numpy uint64 arrays of shape (..., 4),
little-endian limbs (the reference's Int::bits64[0..3]).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libkgoracle.so")

NB_JUMP = 64
P = (1 << 256) - 0x1000003D1
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def build(force=False):
    src = os.path.join(_HERE, "kg_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB


_lib = None
_u64p = C.POINTER(C.c_uint64)


class Rng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


class DP(C.Structure):
    _fields_ = [("x", C.c_uint64 * 4), ("d", C.c_uint64 * 4), ("kIdx", C.c_uint64), ("step", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        for name in ("kgo_modmul", "kgo_modsub", "kgo_order_add", "kgo_order_sub"):
            getattr(L, name).argtypes = [_u64p, _u64p, _u64p]
            getattr(L, name).restype = None
        for name in ("kgo_modsqr", "kgo_modinv", "kgo_modneg", "kgo_order_neg"):
            getattr(L, name).argtypes = [_u64p, _u64p]
            getattr(L, name).restype = None
        L.kgo_group_modinv.argtypes = [_u64p, C.c_size_t, _u64p]
        L.kgo_group_modinv.restype = None
        L.kgo_ec_mul_g.argtypes = [_u64p, _u64p, _u64p]
        L.kgo_ec_mul_g.restype = C.c_int
        L.kgo_ec_add.argtypes = [_u64p] * 6
        L.kgo_ec_add.restype = None
        L.kgo_ec_mul_g_batch.argtypes = [_u64p, _u64p, _u64p, C.c_size_t]
        L.kgo_ec_mul_g_batch.restype = None
        L.kgo_ec_decompress.argtypes = [_u64p, _u64p, C.c_int]
        L.kgo_ec_decompress.restype = C.c_int
        L.kgo_rseed.argtypes = [C.POINTER(Rng), C.c_uint32]
        L.kgo_rseed.restype = None
        L.kgo_rndl.argtypes = [C.POINTER(Rng)]
        L.kgo_rndl.restype = C.c_uint32
        L.kgo_rand_bits.argtypes = [C.POINTER(Rng), _u64p, C.c_int]
        L.kgo_rand_bits.restype = None
        L.kgo_create_jump_table.argtypes = [C.c_int, _u64p, _u64p, _u64p]
        L.kgo_create_jump_table.restype = C.c_int
        L.kgo_create_herd.argtypes = [C.POINTER(Rng), C.c_int, _u64p, _u64p, _u64p, C.c_int, C.c_size_t,
                                      _u64p, _u64p, _u64p]
        L.kgo_create_herd.restype = None
        L.kgo_walk.argtypes = [_u64p, _u64p, _u64p, C.c_size_t, C.c_uint64, C.c_uint64, _u64p, _u64p, _u64p,
                               C.POINTER(DP), C.c_size_t]
        L.kgo_walk.restype = C.c_size_t
        L.kgo_ht_convert.argtypes = [_u64p, _u64p, C.c_uint32, _u64p, _u64p, _u64p]
        L.kgo_ht_convert.restype = None
        _lib = L
    return _lib


# ---- 256-bit <-> limb helpers -------------------------------------------------------
def to_limbs(v):
    """python int -> uint64[4] (value taken mod 2^256)."""
    v &= (1 << 256) - 1
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def from_limbs(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(4))


def ints_to_array(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = to_limbs(v)
    return out


def array_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [from_limbs(r) for r in arr]


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def _bin(fn, a, b):
    r = np.zeros(4, dtype=np.uint64)
    fn(_p(r), _p(to_limbs(a)), _p(to_limbs(b)))
    return from_limbs(r)


def _un(fn, a):
    r = np.zeros(4, dtype=np.uint64)
    fn(_p(r), _p(to_limbs(a)))
    return from_limbs(r)


def modmul(a, b):
    return _bin(lib().kgo_modmul, a, b)


def modsqr(a):
    return _un(lib().kgo_modsqr, a)


def modsub(a, b):
    return _bin(lib().kgo_modsub, a, b)


def modinv(a):
    return _un(lib().kgo_modinv, a)


def modneg(a):
    return _un(lib().kgo_modneg, a)


def order_add(a, b):
    return _bin(lib().kgo_order_add, a, b)


def order_sub(a, b):
    return _bin(lib().kgo_order_sub, a, b)


def order_neg(a):
    return _un(lib().kgo_order_neg, a)


def group_modinv(vals):
    arr = ints_to_array(vals)
    scr = np.zeros_like(arr)
    lib().kgo_group_modinv(_p(arr), len(vals), _p(scr))
    return array_to_ints(arr)


def ec_mul_g(k):
    x = np.zeros(4, dtype=np.uint64)
    y = np.zeros(4, dtype=np.uint64)
    ok = lib().kgo_ec_mul_g(_p(x), _p(y), _p(to_limbs(k)))
    return (from_limbs(x), from_limbs(y)) if ok else None


def ec_mul_g_batch(ks):
    """ks: uint64 (n,4) array -> (x, y) uint64 (n,4) arrays."""
    ks = np.ascontiguousarray(ks, dtype=np.uint64)
    x = np.zeros_like(ks)
    y = np.zeros_like(ks)
    lib().kgo_ec_mul_g_batch(_p(x), _p(y), _p(ks), ks.shape[0])
    return x, y


def ec_add(p1, p2):
    rx = np.zeros(4, dtype=np.uint64)
    ry = np.zeros(4, dtype=np.uint64)
    lib().kgo_ec_add(_p(rx), _p(ry), _p(to_limbs(p1[0])), _p(to_limbs(p1[1])), _p(to_limbs(p2[0])),
                     _p(to_limbs(p2[1])))
    return from_limbs(rx), from_limbs(ry)


def ec_decompress(x, y_is_odd):
    y = np.zeros(4, dtype=np.uint64)
    ok = lib().kgo_ec_decompress(_p(y), _p(to_limbs(x)), int(bool(y_is_odd)))
    return from_limbs(y) if ok else None


def parse_pubkey(hexstr):
    """02/03/04-prefixed public key -> (x, y)  (math/secp256k1.cpp:109-168)."""
    b = bytes.fromhex(hexstr)
    if b[0] in (2, 3):
        x = int.from_bytes(b[1:33], "big")
        y = ec_decompress(x, b[0] == 3)
        if y is None:
            raise ValueError("not on curve")
        return x, y
    if b[0] == 4:
        return int.from_bytes(b[1:33], "big"), int.from_bytes(b[33:65], "big")
    raise ValueError("bad prefix")


class RNG:
    def __init__(self, seed):
        self.state = Rng()
        lib().kgo_rseed(C.byref(self.state), seed & 0xFFFFFFFF)

    def rndl(self):
        return lib().kgo_rndl(C.byref(self.state))

    def rand_bits(self, nbit):
        r = np.zeros(4, dtype=np.uint64)
        lib().kgo_rand_bits(C.byref(self.state), _p(r), nbit)
        return from_limbs(r)


def create_jump_table(range_power):
    jd = np.zeros((NB_JUMP, 4), dtype=np.uint64)
    jx = np.zeros((NB_JUMP, 4), dtype=np.uint64)
    jy = np.zeros((NB_JUMP, 4), dtype=np.uint64)
    tries = lib().kgo_create_jump_table(range_power, _p(jd), _p(jx), _p(jy))
    return jd, jx, jy, tries


def create_herd(rng, range_power, range_width, key, n, first_type=0):
    """Returns px, py, d as uint64 (n,4).  key=(x,y) is the (range-start-shifted) search key."""
    px = np.zeros((n, 4), dtype=np.uint64)
    py = np.zeros((n, 4), dtype=np.uint64)
    d = np.zeros((n, 4), dtype=np.uint64)
    w8 = to_limbs(range_width >> 3)
    lib().kgo_create_herd(C.byref(rng.state), range_power, _p(w8), _p(to_limbs(key[0])), _p(to_limbs(key[1])),
                          first_type, n, _p(px), _p(py), _p(d))
    return px, py, d


def walk(px, py, d, nsteps, dp_mask, jd, jx, jy, cap=None):
    """In-place walk of all kangaroos; returns list of (step, kIdx, x, d) in (step,g) order."""
    n = px.shape[0]
    if cap is None:
        cap = 1 << 16
    while True:
        buf = (DP * cap)()
        # work on copies so that a too-small cap can be retried
        cx, cy, cd = px.copy(), py.copy(), d.copy()
        found = lib().kgo_walk(_p(cx), _p(cy), _p(cd), n, nsteps, dp_mask, _p(jd), _p(jx), _p(jy), buf, cap)
        if found <= cap:
            px[:], py[:], d[:] = cx, cy, cd
            break
        cap = int(found)
    out = []
    for i in range(found):
        e = buf[i]
        out.append((int(e.step), int(e.kIdx), from_limbs(np.array(e.x[:], dtype=np.uint64)),
                    from_limbs(np.array(e.d[:], dtype=np.uint64))))
    return out


def walk_raw(px, py, d, nsteps, dp_mask, jd, jx, jy, cap):
    """Like walk() but returns (found, ctypes DP buffer) without Python conversion (bulk use)."""
    buf = (DP * max(cap, 1))()
    found = lib().kgo_walk(_p(px), _p(py), _p(d), px.shape[0], nsteps, dp_mask, _p(jd), _p(jx), _p(jy), buf, cap)
    return found, buf


def ht_convert(x, d, ktype):
    h = C.c_uint64(0)
    X = np.zeros(4, dtype=np.uint64)
    D = np.zeros(4, dtype=np.uint64)
    lib().kgo_ht_convert(_p(to_limbs(x)), _p(to_limbs(d)), ktype, C.byref(h), _p(X), _p(D))
    return h.value, from_limbs(X), from_limbs(D)
