"""Test stand-in for GPUEngine on a machine without a GPU: the member names the solve loops use, the jumps done by
the oracle.  Test infrastructure only (it imports the oracle); nothing in the product refers to it."""
import numpy as np

import kgoracle as K
from kangaroo_amd import host
from kangaroo_amd.engine import ITEM_DTYPE


class OracleEngine:
    deviceName = "oracle walk on the CPU (test stand-in)"

    def __init__(self, n_kangaroo):
        assert n_kangaroo % 2 == 0
        self.nbKangaroo = n_kangaroo
        self.nb_run = 256
        self.launches = 0

    def SetWildOffset(self, offset):
        pass  # the oracle carries true distances, as the engine's interface does

    def SetParams(self, dp_mask, jd, jx, jy):
        self.dp_mask, self.jd, self.jx, self.jy = dp_mask, jd, jx, jy

    def SetNbRun(self, nb_run):
        self.nb_run = nb_run

    def HerdFromDistances(self, d, key=None):
        self.d = np.ascontiguousarray(d, dtype=np.uint64).copy()
        self.px, self.py = host.ec_mul_g_batch(self.d)
        if key is not None:
            for i in range(1, self.nbKangaroo, 2):
                x, y = host.ec_add((host.to_int(self.px[i]), host.to_int(self.py[i])), key)
                self.px[i], self.py[i] = host.limbs(x), host.limbs(y)

    def SetKangaroos(self, px, py, d):
        self.px, self.py, self.d = (np.ascontiguousarray(a, dtype=np.uint64).copy() for a in (px, py, d))

    def GetKangaroos(self):
        return self.px.copy(), self.py.copy(), self.d.copy()

    def SetKangaroo(self, k_idx, x, y, d):
        self.px[k_idx], self.py[k_idx], self.d[k_idx] = x, y, d

    def callKernel(self):
        pass

    def Launch(self, spinWait=False, copy=True):
        dps = K.walk(self.px, self.py, self.d, self.nb_run, self.dp_mask, self.jd, self.jx, self.jy)
        self.launches += 1
        items = np.zeros(len(dps), dtype=ITEM_DTYPE)
        for i, (_, k, x, dd) in enumerate(dps):
            items["x"][i], items["d"][i], items["kIdx"][i] = host.limbs(x), host.limbs(dd), k
        return items

    def Collect(self, copy=True):
        return np.zeros(0, dtype=ITEM_DTYPE)

    def close(self):
        pass
