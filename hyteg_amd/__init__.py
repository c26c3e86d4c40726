"""hyteg_amd -- MI355X-native kernels for HyTeG's matrix-free P1 multigrid hot path.

The product is the C-ABI shared library ``hyteg_amd/lib/libhyteg_hip.so`` (include/hyteg_hip.h) and the
C++ host layer under ``hyteg_amd/host`` that mirrors HyTeG's Operator<>/Function<> interface.
``hyteg_amd.capi`` is a ctypes binding of that C-ABI used by the tests, bench.py and the Python-side
multi-GPU driver.  There is no CPU fallback anywhere in this package: loading fails loudly when the
HIP library has not been built.
"""
from .capi import HytegHipError, lib, lib_path  # noqa: F401

__all__ = ["HytegHipError", "lib", "lib_path"]
