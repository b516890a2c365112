"""MI355X-native point-set-registration core: the package that ships the path.

Layout
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/mi_slam.h) -> libmislam.so
  host/      C++ host side mirroring the reference's registration interface (Common::SlamFunc,
             GetCudaIcpTransformationMatrix / GetCudaCpdTransformationMatrix, config + OBJ input)
  capi.py    ctypes pass-through of the C ABI for this repository's Python callers (tests, bench.py)

The directory name contains a hyphen, so it is imported by path (see `load_package()` in __graft_entry__.py).
"""
from . import capi  # noqa: F401
