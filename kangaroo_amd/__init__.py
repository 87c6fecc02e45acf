"""kangaroo_amd — MI355X (gfx950) Pollard-kangaroo jump engine.

The product is the C-ABI shared library `libkangaroo_amd.so` (include/kangaroo_amd.h) built from the
HIP sources under kangaroo_amd/csrc.  This package is the thin Python mirror of the reference's
GPUEngine interface used by the tests and by bench.py; it never falls back to a CPU path.
"""
from .engine import GPUEngine, KgError, Item, lib_path, load_library  # noqa: F401

GPU_GRP_SIZE = 128  # constants.h:15
NB_JUMP = 64  # constants.h:12
NB_RUN = 256  # constants.h:21
