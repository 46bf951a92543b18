"""MI355X-native exact Gaussian-process regression (fit / predict hot path).

Everything numerical lives in ``csrc/libgpx.so`` (HIP, gfx950) behind the C ABI of
``include/gpx.h``; this package is the thin Python host.  Importing the package does
not load the library; constructing a :class:`GP` does, and raises if it is missing.
"""
from .gp import GP  # noqa: F401
from ._abi import GpxError  # noqa: F401

__all__ = ["GP", "GpxError"]
__version__ = "0.1.0"
