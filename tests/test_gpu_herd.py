"""GPU herd creation (kg_herd_from_distances) against the host CreateHerd restatement, which is
itself pinned to the reference's herds (tests/test_host_lib.py), and against the oracle's scalar multiplication."""
import numpy as np
import pytest

import kgoracle as K
from kangaroo_amd import GPUEngine, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("groups,per_group,range_power", [(1, 32, 64), (2, 64, 125), (3, 40, 250)])
def test_herd_points_equal_host_create_herd(groups, per_group, range_power):
    priv = 0xC0FFEE1234567890ABCDEF % (1 << (range_power - 1))
    key = K.ec_mul_g(priv)
    width = (1 << range_power) - 1
    e = GPUEngine(groups, per_group, 0, 65536)
    n = e.nbKangaroo
    e.SetWildOffset(width >> 1)
    px, py, d = host.create_herd(host.Rng(99), range_power, width, key, n, 0)
    d2 = host.herd_distances(host.Rng(99), range_power, width, n, 0)
    assert np.array_equal(d, d2)
    e.HerdFromDistances(d2, key)
    gx, gy, gd = e.GetKangaroos()
    assert np.array_equal(gx, px) and np.array_equal(gy, py) and np.array_equal(gd, d)
    e.close()


def test_scalar_edge_cases_against_oracle():
    """Plain d*G (zero key) for scalars that stress the comb: tiny, single-window, all-ones, n-1, > 2^255."""
    e = GPUEngine(1, 2, 0, 65536)  # 256 kangaroos
    n = e.nbKangaroo
    rnd = np.random.default_rng(5)
    ks = [1, 2, 3, 255, 256, (1 << 64) - 1, 1 << 64, (1 << 248) | 1, K.N - 1, K.N - 2, (1 << 255) + 12345,
          int.from_bytes(b"\xff" * 31, "little")]
    while len(ks) < n:
        ks.append(int.from_bytes(rnd.bytes(32), "little") % K.N or 1)
    d = np.array([[(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for k in ks], dtype=np.uint64)
    e.SetWildOffset(0)
    e.HerdFromDistances(d)
    gx, gy, _ = e.GetKangaroos()
    for i in list(range(16)) + [n - 1]:
        x, y = K.ec_mul_g(ks[i])
        assert host.to_int(gx[i]) == x and host.to_int(gy[i]) == y, i
    ex, ey = host.ec_mul_g_batch(d)
    assert np.array_equal(gx, ex) and np.array_equal(gy, ey)
    e.close()


def test_full_size_herd_sampled_against_host():
    """The bench herd (512 x 128 grid, 8 388 608 kangaroos, 125-bit range): 4096 sampled start points and the edges
    of the lane/slot layout equal d*G (+ key for the wild, odd-indexed half) computed by the host code."""
    range_power = 125
    width = (1 << range_power) - 1
    key = host.ec_mul_g(0x0A5B3C1D2E7F9081726354453627180F % width)
    e = GPUEngine(512, 128, 0, 65536)
    n = e.nbKangaroo
    assert n == 8388608
    e.SetWildOffset(width >> 1)
    d = host.herd_distances(host.Rng(7), range_power, width, n, 0)
    e.HerdFromDistances(d, key)
    gx, gy, gd = e.GetKangaroos()
    e.close()
    assert np.array_equal(gd, d)
    rng = np.random.default_rng(3)
    sample = np.unique(np.concatenate([np.array([0, 1, 63, 64, 255, 256, 65535, 65536, 131071, 131072, n // 2, n - 2, n - 1]),
                                       rng.integers(0, n, 4096)]))
    hx, hy = host.ec_mul_g_batch(np.ascontiguousarray(d[sample]))
    for j, i in enumerate(sample):
        pt = (host.to_int(hx[j]), host.to_int(hy[j]))
        if i & 1:
            pt = host.ec_add(pt, key)
        assert pt == (host.to_int(gx[i]), host.to_int(gy[i])), int(i)
