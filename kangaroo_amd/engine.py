"""Python mirror of the reference's `class GPUEngine` (math/gpu/engine.h:13-54) over the C ABI.

Same member names and argument meaning as the C++ class; 256-bit values are numpy uint64 arrays of
shape (..., 4) (Int::bits64[0..3]).  Errors raise KgError with the message the C++ class would print.
There is no CPU fallback: if libkangaroo_amd.so is missing or no gfx950 device is present the
constructor raises.
"""
import ctypes as C
import os
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)

Item = namedtuple("Item", "x d kIdx")  # math/cpu/int.h:339-343


class KgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class _CItem(C.Structure):
    _fields_ = [("x", C.c_uint64 * 4), ("d", C.c_uint64 * 4), ("kIdx", C.c_uint64)]


class _CTuning(C.Structure):
    _fields_ = [("mode", C.c_int), ("k_per_lane", C.c_int), ("ring_max_items", C.c_uint32), ("ws_limit_bytes", C.c_uint64)]


MODE_AUTO, MODE_STREAM, MODE_REG = 0, 1, 2
STATIC_WORKSPACE_BYTES = 32 << 20  # KG_STATIC_WORKSPACE_BYTES

ITEM_DTYPE = np.dtype([("x", "<u8", (4,)), ("d", "<u8", (4,)), ("kIdx", "<u8")])
assert ITEM_DTYPE.itemsize == C.sizeof(_CItem) == 72

KG_ERR_LOST = -4
_lib = None


def lib_path():
    # KANGAROO_AMD_LIB: alternative build of the same library (A/B measurements)
    return os.environ.get("KANGAROO_AMD_LIB") or os.path.join(_HERE, "libkangaroo_amd.so")


def load_library():
    """Loads the HIP engine library.  Raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise KgError(-100, f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(path)
    H = C.c_void_p
    sig = {
        "kg_device_count": ([C.POINTER(C.c_int)], C.c_int),
        "kg_default_grid": ([C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "kg_create": ([C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(H)], C.c_int),
        "kg_create_tuned": ([C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(_CTuning), C.POINTER(H)], C.c_int),
        "kg_destroy": ([H], C.c_int),
        "kg_device_name": ([H, C.c_char_p, C.c_size_t], C.c_int),
        "kg_memory_bytes": ([H, _u64p], C.c_int),
        "kg_nb_thread": ([H], C.c_int),
        "kg_group_size": ([H], C.c_int),
        "kg_nb_kangaroo": ([H], C.c_uint64),
        "kg_set_wild_offset": ([H, _u64p], C.c_int),
        "kg_set_params": ([H, C.c_uint64, _u64p, _u64p, _u64p], C.c_int),
        "kg_set_kangaroos": ([H, _u64p, _u64p, _u64p, C.c_uint64], C.c_int),
        "kg_get_kangaroos": ([H, _u64p, _u64p, _u64p, C.c_uint64], C.c_int),
        "kg_herd_from_distances": ([H, _u64p, C.c_uint64, _u64p, _u64p], C.c_int),
        "kg_set_kangaroo": ([H, C.c_uint64, _u64p, _u64p, _u64p], C.c_int),
        "kg_set_nb_run": ([H, C.c_uint32], C.c_int),
        "kg_kick": ([H], C.c_int),
        "kg_kick_static": ([H], C.c_int),
        "kg_collect": ([H, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)], C.c_int),
        "kg_launch": ([H, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)], C.c_int),
        "kg_kick_and_wait": ([H], C.c_int),
        "kg_dp_capacity": ([H, C.POINTER(C.c_uint32)], C.c_int),
        "kg_last_kernel_ms": ([H, C.POINTER(C.c_float)], C.c_int),
        "kg_last_sum_kernel_ms": ([H, C.POINTER(C.c_float)], C.c_int),
        "kg_last_kernel_clock": ([H, _u64p, _u64p], C.c_int),
        "kg_kernel_info": ([H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
        "kg_kernel_name": ([H, C.c_char_p, C.c_size_t], C.c_int),
        "kg_field_kat": ([C.c_int, C.c_int, _u32p, _u32p, _u32p, _u32p, C.c_uint32], C.c_int),
        "kg_alloc_pinned": ([C.c_size_t], C.c_void_p),
        "kg_free_pinned": ([C.c_void_p], None),
        "kg_print_device_info": ([], C.c_int),
        "kg_device_pci_bus_id": ([C.c_int, C.c_char_p, C.c_size_t], C.c_int),
        "kg_last_error": ([], C.c_char_p),
        "kg_version": ([], C.c_char_p),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export the ABI
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


ABI_SYMBOLS = [
    "kg_device_count", "kg_default_grid", "kg_create", "kg_create_tuned", "kg_destroy", "kg_device_name", "kg_memory_bytes",
    "kg_nb_thread", "kg_group_size", "kg_nb_kangaroo", "kg_set_wild_offset", "kg_set_params", "kg_set_kangaroos",
    "kg_get_kangaroos", "kg_herd_from_distances", "kg_set_kangaroo", "kg_set_nb_run", "kg_kick", "kg_kick_static", "kg_collect", "kg_launch",
    "kg_kick_and_wait", "kg_dp_capacity", "kg_last_kernel_ms", "kg_last_sum_kernel_ms", "kg_last_kernel_clock", "kg_kernel_info", "kg_kernel_name", "kg_field_kat", "kg_alloc_pinned", "kg_free_pinned",
    "kg_print_device_info", "kg_device_pci_bus_id", "kg_last_error", "kg_version",
]


def _chk(rc):
    if rc != 0:
        raise KgError(rc, load_library().kg_last_error().decode())


def _arr(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if n is not None and a.size != n * 4:
        raise ValueError(f"expected {n} x 4 limbs, got shape {a.shape}")
    return a


def _p(a):
    return a.ctypes.data_as(_u64p)


def _limbs(v):
    if isinstance(v, int):
        return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    return _arr(v, 1)


class GPUEngine:
    """GPUEngine(nbThreadGroup, nbThreadPerGroup, gpuId, maxFound) — math/gpu/engine.cu:67-170."""

    def __init__(self, nbThreadGroup, nbThreadPerGroup, gpuId, maxFound, mode=None, k_per_lane=0, ring_max=0, ws_limit=0):
        """mode / k_per_lane / ring_max / ws_limit: kg_tuning (kg_create_tuned); mode is None, "stream" or "reg"; ring_max
        caps the DP ring (items per launch), ws_limit the workspace bytes of one kernel launch (0 = the engine's defaults)."""
        self._L = load_library()
        self._h = C.c_void_p()
        if mode is None and not k_per_lane and not ring_max and not ws_limit:
            _chk(self._L.kg_create(gpuId, nbThreadGroup, nbThreadPerGroup, maxFound, C.byref(self._h)))
        else:
            t = _CTuning({None: MODE_AUTO, "stream": MODE_STREAM, "reg": MODE_REG}[mode], int(k_per_lane or 0), int(ring_max or 0), int(ws_limit or 0))
            _chk(self._L.kg_create_tuned(gpuId, nbThreadGroup, nbThreadPerGroup, maxFound, C.byref(t), C.byref(self._h)))
        buf = C.create_string_buffer(256)
        _chk(self._L.kg_device_name(self._h, buf, 256))
        self.deviceName = buf.value.decode()
        self.nbKangaroo = int(self._L.kg_nb_kangaroo(self._h))
        self._cap = 0
        self._items = None  # staging array for kg_collect / kg_launch, sized from kg_dp_capacity when first needed
        self.lost = 0
        self.nb_run = 256  # NB_RUN, constants.h:21

    def close(self):
        if self._h:
            self._L.kg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- static helpers (math/gpu/engine.cu:191-275)
    @staticmethod
    def GetGridSize(gpuId, x=0, y=0):
        L = load_library()
        cx, cy = C.c_int(x), C.c_int(y)
        _chk(L.kg_default_grid(gpuId, C.byref(cx), C.byref(cy)))
        return cx.value, cy.value

    @staticmethod
    def PrintCudaInfo():
        _chk(load_library().kg_print_device_info())

    @staticmethod
    def PciBusId(gpuId):
        """PCI address of the device ("0000:c1:00.0"), kg_device_pci_bus_id."""
        buf = C.create_string_buffer(32)
        _chk(load_library().kg_device_pci_bus_id(gpuId, buf, 32))
        return buf.value.decode().lower()

    @staticmethod
    def DeviceCount():
        c = C.c_int(0)
        _chk(load_library().kg_device_count(C.byref(c)))
        return c.value

    # -- accessors
    def GetNbThread(self):
        return self._L.kg_nb_thread(self._h)

    def GetGroupSize(self):
        return self._L.kg_group_size(self._h)

    def GetMemory(self):
        b = C.c_uint64(0)
        _chk(self._L.kg_memory_bytes(self._h, C.byref(b)))
        return b.value

    def KernelInfo(self):
        k, dw, lds, grid = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _chk(self._L.kg_kernel_info(self._h, C.byref(k), C.byref(dw), C.byref(lds), C.byref(grid)))
        return {"k_per_lane": k.value, "delta_words": dw.value, "lds_bytes": lds.value, "grid": grid.value}

    def KernelName(self):
        buf = C.create_string_buffer(128)
        _chk(self._L.kg_kernel_name(self._h, buf, C.c_size_t(len(buf))))
        return buf.value.decode()

    def LastKernelMs(self):
        ms = C.c_float(0)
        _chk(self._L.kg_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def LastSumKernelMs(self):
        """Duration of the last completed launch's history kernel (kg_last_sum_kernel_ms)."""
        ms = C.c_float(0)
        _chk(self._L.kg_last_sum_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def LastKernelClockMHz(self):
        """Shader clock of the last streaming launch (kg_last_kernel_clock), 0.0 if unknown."""
        c, r = C.c_uint64(0), C.c_uint64(0)
        _chk(self._L.kg_last_kernel_clock(self._h, C.byref(c), C.byref(r)))
        return 100.0 * c.value / r.value if r.value else 0.0

    # -- configuration
    def SetWildOffset(self, offset):
        _chk(self._L.kg_set_wild_offset(self._h, _p(_limbs(offset))))

    def SetParams(self, dpMask, distance, px, py):
        jd, jx, jy = _arr(distance, 64), _arr(px, 64), _arr(py, 64)
        _chk(self._L.kg_set_params(self._h, dpMask, _p(jd), _p(jx), _p(jy)))

    def SetNbRun(self, nbRun):
        _chk(self._L.kg_set_nb_run(self._h, nbRun))
        self.nb_run = int(nbRun)

    def SetKangaroos(self, px, py, d):
        n = self.nbKangaroo
        a, b, c = _arr(px, n), _arr(py, n), _arr(d, n)
        _chk(self._L.kg_set_kangaroos(self._h, _p(a), _p(b), _p(c), n))

    def HerdFromDistances(self, d, key=(0, 0)):
        """Points computed on the GPU: d_i*G (+ key for odd i), then as SetKangaroos (kg_herd_from_distances)."""
        n = self.nbKangaroo
        dd = _arr(d, n)
        _chk(self._L.kg_herd_from_distances(self._h, _p(dd), n, _p(_limbs(key[0])), _p(_limbs(key[1]))))

    def GetKangaroos(self):
        n = self.nbKangaroo
        px = np.zeros((n, 4), dtype=np.uint64)
        py = np.zeros((n, 4), dtype=np.uint64)
        d = np.zeros((n, 4), dtype=np.uint64)
        _chk(self._L.kg_get_kangaroos(self._h, _p(px), _p(py), _p(d), n))
        return px, py, d

    def SetKangaroo(self, kIdx, px, py, d):
        _chk(self._L.kg_set_kangaroo(self._h, kIdx, _p(_limbs(px)), _p(_limbs(py)), _p(_limbs(d))))

    # -- launches
    def callKernel(self):
        _chk(self._L.kg_kick(self._h))
        return True

    def callKernelStatic(self):
        """kg_kick_static: callKernel without any allocation (workspace in the library's static device array)."""
        _chk(self._L.kg_kick_static(self._h))
        return True

    def callKernelAndWait(self):
        _chk(self._L.kg_kick_and_wait(self._h))
        return True

    def DpCapacity(self):
        """Items one launch can return under the current configuration (kg_dp_capacity)."""
        c = C.c_uint32(0)
        _chk(self._L.kg_dp_capacity(self._h, C.byref(c)))
        return c.value

    def _finish(self, fn, copy):
        cap = self.DpCapacity()  # follows SetParams / SetNbRun
        if self._items is None or cap > self._cap:
            self._cap = cap
            self._items = np.empty(cap, dtype=ITEM_DTYPE)
        n, lost = C.c_uint32(0), C.c_uint32(0)
        rc = fn(self._h, self._items.ctypes.data, self._cap, C.byref(n), C.byref(lost))
        self.lost = lost.value
        if rc != 0 and rc != KG_ERR_LOST:
            _chk(rc)
        if rc == KG_ERR_LOST:
            # the reference prints this and exits (engine.cu:506-510)
            raise KgError(rc, self._L.kg_last_error().decode())
        out = self._items[: n.value]
        return out.copy() if copy else out  # the staging array is reused by the next call

    def Launch(self, spinWait=False, copy=True):
        """Wait for the launch in flight, return its DPs (structured array x,d,kIdx), start the next.
        copy=False returns a view of the engine's staging array, valid until the next Launch / Collect: for callers
        that consume the items at once (at DP = 8 a launch of the default herd returns 600 MB of them).
        spinWait: the reference polls the event in a loop instead of blocking on it (engine.cu:486-500); this engine
        always blocks on its stream's event, the flag is accepted and has no effect."""
        return self._finish(self._L.kg_launch, copy)

    def Collect(self, copy=True):
        """Wait for the launch in flight and return its DPs without starting another."""
        return self._finish(self._L.kg_collect, copy)


def items_to_tuples(items):
    """structured array -> list of Item(x:int, d:int, kIdx:int)."""
    out = []
    for it in items:
        x = sum(int(it["x"][i]) << (64 * i) for i in range(4))
        d = sum(int(it["d"][i]) << (64 * i) for i in range(4))
        out.append(Item(x, d, int(it["kIdx"])))
    return out


KAT_OPS = {"mul": 0, "sqr": 1, "sub": 2, "sub2": 3, "submul": 4, "canon": 5, "inv": 6, "weakmul": 7, "subweak": 8,
           "settle01": 9, "settle2": 10}


def field_kat_limbs(op, limbs, gpu_id=0):
    """settle01 / settle2 on raw 29-bit limb vectors (lists of nine ints each); returns the canonical values."""
    a = [sum(l[k] << (32 * k) for k in range(8)) for l in limbs]
    b = [l[8] for l in limbs]
    return field_kat(op, a, b, b, gpu_id)


def field_kat(op, a, b=None, c=None, gpu_id=0):
    """kg_field_kat: run the DEVICE field arithmetic on lists of Python ints (< 2^256); returns a list of ints."""
    n = len(a)
    b = a if b is None else b
    c = a if c is None else c
    assert len(b) == n and len(c) == n

    def words(vs):
        out = np.zeros((n, 8), dtype=np.uint32)
        for i, v in enumerate(vs):
            for k in range(8):
                out[i, k] = (v >> (32 * k)) & 0xFFFFFFFF
        return out

    wa, wb, wc = words(a), words(b), words(c)
    r = np.zeros((n, 8), dtype=np.uint32)
    q = lambda x: x.ctypes.data_as(_u32p)
    _chk(load_library().kg_field_kat(gpu_id, KAT_OPS[op], q(wa), q(wb), q(wc), q(r), n))
    return [sum(int(r[i, k]) << (32 * k) for k in range(8)) for i in range(n)]
