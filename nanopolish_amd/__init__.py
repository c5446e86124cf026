"""nanopolish_amd -- MI355X-native (gfx950) implementation of nanopolish's signal-level HMM hot path.

The product is the C-ABI shared library `libnp_hip.so` (include/np_hmm.h), hand-written HIP for CDNA4.
This package is the thin Python host side used by the tests and bench.py: ctypes bindings (`lib`) and a
mirror of the reference's entry points (`api`).  There is no CPU fallback: importing `lib` raises if the
library has not been built, and creating a context raises if no gfx950 device is present.
"""
from .lib import load_library, library_path, build_library  # noqa: F401

__all__ = ["load_library", "library_path", "build_library"]
