"""jm_amd -- MI355X (gfx950) implementation of JM lencod's data-parallel inner loop.

The product is the C-ABI library `libjmhip.so` (include/jmhip.h, sources in jm_amd/csrc);
this package is its Python host-side mirror (ctypes) used by the tests and by bench.py.
There is no CPU fallback: importing jm_amd.lib without a built library, or creating a
context without a gfx950 device, raises.
"""
from .lib import JmHip, JmHipError, PARTITIONS, ME_JOB, ME_RESULT, ME_BEST, SUBPEL_JOB, CAND, TQ_OUT, DB_MB, DB_MOTION, load_library  # noqa: F401
