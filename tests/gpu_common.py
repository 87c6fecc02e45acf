"""Shared helpers for the GPU parity tests (the checker side uses the oracle)."""
import numpy as np

import kgoracle as K


def make_case(range_power, n, seed, dp_bits, priv=None):
    """Seeded herd + jump table exactly as the reference builds them (oracle restatement)."""
    width = (1 << range_power) - 1
    if priv is None:
        priv = (0xA5B3C1D2E7F9 * 0x10001 + seed) % width
    key = K.ec_mul_g(priv)
    jd, jx, jy, _ = K.create_jump_table(range_power)
    rng = K.RNG(seed)
    px, py, d = K.create_herd(rng, range_power, width, key, n, 0)
    dp_mask = 0 if dp_bits == 0 else (~((1 << (64 - dp_bits)) - 1)) & 0xFFFFFFFFFFFFFFFF
    return dict(range_power=range_power, width=width, priv=priv, key=key, jd=jd, jx=jx, jy=jy, px=px, py=py, d=d,
                dp_mask=dp_mask, rng=rng)


def dp_multiset(items):
    """structured items -> sorted list of (kIdx, x, d) python ints."""
    out = []
    for it in items:
        x = sum(int(it["x"][i]) << (64 * i) for i in range(4))
        d = sum(int(it["d"][i]) << (64 * i) for i in range(4))
        out.append((int(it["kIdx"]), x, d))
    return sorted(out)


def oracle_dps(case, px, py, d, nsteps):
    """Advance (px,py,d) in place by nsteps with the oracle; returns sorted (kIdx, x, d)."""
    dps = K.walk(px, py, d, nsteps, case["dp_mask"], case["jd"], case["jx"], case["jy"], cap=1 << 16)
    return sorted((k, x, dd) for _, k, x, dd in dps)
