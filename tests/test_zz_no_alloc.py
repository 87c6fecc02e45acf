"""kg_kick_static: the entry point that allocates nothing (workspace in the library's static device array).  Its
output must equal kg_kick's, bit for bit, on two calls in a row, the second with a smaller input; sizes beyond the
static workspace are refused with a clear error."""
import numpy as np
import pytest

from gpu_common import dp_multiset, make_case, oracle_dps
from kangaroo_amd import GPUEngine, KgError

pytestmark = pytest.mark.gpu


def _pair(groups, per_group, seed, nb_run):
    case = make_case(64, 128 * groups * per_group, seed=seed, dp_bits=5)
    engines = []
    for _ in range(2):
        e = GPUEngine(groups, per_group, 0, 65536 * 2)
        e.SetWildOffset(case["width"] >> 1)
        e.SetParams(case["dp_mask"], case["jd"], case["jx"], case["jy"])
        e.SetNbRun(nb_run)
        e.SetKangaroos(case["px"], case["py"], case["d"])
        engines.append(e)
    return case, engines[0], engines[1]


def test_static_kick_equals_kick_twice_second_smaller():
    # first call: 16 384 kangaroos x 48 steps; second call: 2 560 kangaroos x 33 steps (smaller workspace: whatever
    # the first call left in the array beyond it must not matter)
    for groups, per_group, seed, nb_run in ((4, 32, 71, 48), (5, 4, 72, 33)):
        case, a, b = _pair(groups, per_group, seed, nb_run)
        assert "stream" in a.KernelName()
        for launch in range(2):  # two launches each: state carried over, both pass directions
            a.callKernel()
            ra = a.Collect()
            b.callKernelStatic()
            rb = b.Collect()
            assert len(ra) > 100 and dp_multiset(ra) == dp_multiset(rb), (groups, launch)
        sa, sb = a.GetKangaroos(), b.GetKangaroos()
        assert all(np.array_equal(x, y) for x, y in zip(sa, sb))
        # and both equal the oracle
        px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
        oracle_dps(case, px, py, d, 2 * nb_run)
        assert np.array_equal(sb[0], px) and np.array_equal(sb[1], py) and np.array_equal(sb[2], d)
        a.close()
        b.close()


def test_static_kick_rejects_what_does_not_fit():
    e = GPUEngine(64, 128, 0, 65536 * 2)  # 1 048 576 kangaroos: 113 MB of slots alone
    case = make_case(64, 128, seed=1, dp_bits=5)
    e.SetWildOffset(case["width"] >> 1)
    e.SetParams(0xFFFFFFFF00000000, case["jd"], case["jx"], case["jy"])
    e.SetNbRun(8)
    e.HerdFromDistances(np.ones((e.nbKangaroo, 4), dtype=np.uint64))
    with pytest.raises(KgError) as ei:
        e.callKernelStatic()
    assert "static workspace holds" in str(ei.value)
    e.callKernel()  # the ordinary entry point still works on the same engine
    e.Collect()
    e.close()
    r = GPUEngine(1, 64, 0, 65536, mode="reg")
    r.SetWildOffset(case["width"] >> 1)
    r.SetParams(0xFFFFFFFF00000000, case["jd"], case["jx"], case["jy"])
    r.HerdFromDistances(np.ones((r.nbKangaroo, 4), dtype=np.uint64))
    with pytest.raises(KgError):
        r.callKernelStatic()
    r.close()
