"""pymodem_amd -- MI355X (gfx950) implementation of pymodem's demod_chain sample-processing path.

Host side stays Python (like the reference); all sample processing runs in hand-written HIP kernels
behind the C ABI of libpymodem_amd.so (include/pymodem_amd.h), reached through ctypes.
There is no CPU fallback: without the built library or without a GPU the stage objects raise.
"""
from ._native import NativeError, lib, library_path  # noqa: F401
from .device import Context, DeviceBuffer  # noqa: F401

__version__ = "0.1.0"
