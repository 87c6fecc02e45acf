"""Known-key solves on the GPU through the C ABI (BASELINE config 2 and smaller): the engine, driven by the
product's host loop (kangaroo_amd/solve.py, the restatement of Kangaroo::SolveKeyGPU core/kangaroo.cpp:397-508),
must recover the private key.  The expected keys are fixed here; the public keys are derived with the oracle
so that the product's own EC code is not its own judge."""
import pytest

import kgoracle as K
from kangaroo_amd import solve

pytestmark = pytest.mark.gpu

CASES = [
    # (name, range_start, range_end, private key, grid, dp_bits)
    ("40bit", 0, (1 << 40) - 1, 0xA5B3C1D2E7, (8, 32), 8),
    ("48bit_offset", 1 << 70, (1 << 70) + (1 << 48) - 1, (1 << 70) + 0xB3C1D2E7F0A1, (16, 32), 8),
    ("64bit_dp12", 0, (1 << 64) - 1, 0xC0FFEE1234567890, (32, 128), 12),
    # BASELINE config 2 as written: 64-bit range, DP = 16, the engine's default grid (8.4 M kangaroos): about 2^33 jumps
    # for the collision plus 2^23 * 2^16 = 2^39 of DP overhead (every kangaroo has to reach a DP twice), ~30 s
    ("config2_64bit_dp16", 0, (1 << 64) - 1, 0xC0FFEE1234567890, (0, 0), 16),
]


@pytest.mark.parametrize("name,lo,hi,priv,grid,dp", CASES, ids=[c[0] for c in CASES])
def test_known_key_is_recovered(name, lo, hi, priv, grid, dp):
    pub = K.ec_mul_g(priv)
    found, st = solve.solve_key(lo, hi, pub, grid=grid, dp_bits=dp, seed=3, max_launches=2000)
    print(name, st)
    assert found == priv
    # sanity of the accounting the reference prints (core/kangaroo.cpp:459): jumps = kangaroos * NB_RUN * launches
    assert st["jumps"] == st["kangaroos"] * 256 * st["launches"]
    # Pollard kangaroo expects about 2 * sqrt(range) jumps (+ DP overhead); 64x that means something is broken
    assert st["jumps"] < 64 * 2 * (1 << (st["range_power"] // 2)) + st["kangaroos"] * max(256 * 8, 4 << dp)
    assert st["table"] + st["dead"] + st["unstored"] == st["dps"]


def test_multi_rank_loop_on_one_gpu():
    """solve_key_multi with world_size 1 (no torch.distributed): the same loop the ranks of a multi-GPU run execute,
    on the real engine; the two-rank exchange itself is covered on the CPU (tests/test_solve_multi.py)."""
    priv = (1 << 70) + 0xB3C1D2E7F0A1
    pub = K.ec_mul_g(priv)
    found, st = solve.solve_key_multi(1 << 70, (1 << 70) + (1 << 48) - 1, pub, grid=(16, 32), dp_bits=8, seed=3,
                                      max_launches=2000)
    assert found == priv and st["world"] == 1
    assert st["jumps"] == st["kangaroos"] * 256 * st["launches"]
