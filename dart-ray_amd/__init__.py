"""dart-ray_amd: MI355X-native implementation of DART-Ray's ray-cell radiative-transfer hot path.

The directory name is the one the build contract asks for; it is imported under the module name
`dartray_amd` (see tests/conftest.py:load_pkg and __graft_entry__.py). The product is the C-ABI
library `dart-ray_amd/csrc/libdartray_rt.so` (hand-written HIP for gfx950, include/dartray_rt.h) and
the Fortran facade in `dart-ray_amd/fortran/`; the Python here is the thin host binding used by the
tests and the benchmark, plus the synthetic-grid generators.
"""
from . import grids, io  # noqa: F401
from . import host  # noqa: F401
from .host import DartRayGPU, DartRayError  # noqa: F401

__all__ = ["grids", "host", "DartRayGPU", "DartRayError"]
