"""Error behaviour of the C ABI on a real device: calls out of order and bad arguments return an error code and a
message, never crash, and leave the engine usable (the reference prints "GPUEngine: ..." and returns false,
math/gpu/engine.cu:427-440, 486-500)."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import dp_multiset, make_case, oracle_dps
from kangaroo_amd import GPUEngine, KgError
from kangaroo_amd.engine import load_library

pytestmark = pytest.mark.gpu


def test_calls_out_of_order_and_bad_arguments():
    L = load_library()
    case = make_case(64, 128 * 1 * 8, seed=99, dp_bits=4)
    e = GPUEngine(1, 8, 0, 1024)
    with pytest.raises(KgError) as ei:  # no params, no herd
        e.callKernel()
    assert ei.value.code == -3 and "SetParams/SetKangaroos" in str(ei.value)
    with pytest.raises(KgError) as ei:  # nothing in flight
        e.Collect()
    assert ei.value.code == -3
    e.SetWildOffset(case["width"] >> 1)
    e.SetParams(case["dp_mask"], case["jd"], case["jx"], case["jy"])
    with pytest.raises(ValueError):  # wrong herd size is caught by the Python mirror ...
        e.SetKangaroos(case["px"][:-1], case["py"][:-1], case["d"][:-1])
    short = np.ascontiguousarray(case["px"][:-1])
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
    assert L.kg_set_kangaroos(e._h, p(short), p(short), p(short), len(short)) == -1  # ... and by the C ABI
    assert b"n must equal" in L.kg_last_error()
    with pytest.raises(KgError) as ei:
        e.SetKangaroo(10 ** 9, 1, 2, 3)
    assert ei.value.code == -1
    with pytest.raises(KgError):
        e.SetNbRun(0)
    for bad in ((0, 0, 8), (0, 1, 0), (999, 1, 8)):  # gpu id, groups, per group
        with pytest.raises(KgError):
            GPUEngine(bad[1], bad[2], bad[0], 1024)
    with pytest.raises(KgError) as ei:
        GPUEngine(1, 8, 0, 1024, mode="reg", k_per_lane=5)
    assert "16 kangaroos per lane" in str(ei.value)
    # the engine still works after all that
    e.SetNbRun(16)
    e.SetKangaroos(case["px"], case["py"], case["d"])
    e.callKernel()
    with pytest.raises(KgError) as ei:  # a launch is already in flight
        e.callKernel()
    assert ei.value.code == -3
    px, py, d = case["px"].copy(), case["py"].copy(), case["d"].copy()
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 16)
    e.callKernelAndWait()  # kick + wait: DPs stay queued for the next collect
    assert dp_multiset(e.Collect()) == oracle_dps(case, px, py, d, 16)
    assert e.GetMemory() > 0 and e.GetNbThread() == 8 and e.GetGroupSize() == 128
    e.close()
    e.close()  # idempotent
