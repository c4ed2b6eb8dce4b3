"""Stream stage: self-synchronising descrambler / NRZI decoder (LFSR lfsr.py:10-52 of the reference), native
C++ behind pm_lfsr_unscramble.  poly 0x1 = pass-through, 0x3 = NRZI, 0x63003 = G3RUH + NRZI."""
import ctypes

import numpy as np

from ._native import check, lib
from .data_classes import AddressedArray
from .string_ops import check_boolean


class LFSR:
    def __init__(self, **kwargs):
        self.polynomial = kwargs.get('poly', 0x1)
        self.invert = kwargs.get('invert', False)
        self.shift_register = 0

    def StringOptionsRetune(self, options):   # lfsr.py:18-20
        self.polynomial = int(options.get('poly', 0x1), 16) if isinstance(options.get('poly', 0x1), str) else int(options.get('poly', 0x1))
        self.invert = check_boolean(options.get('invert', "false"))

    def stream_unscramble_8bit(self, data):
        """One output byte per input byte, addresses passed through.  Accepts list[AddressedData] or AddressedArray."""
        src = AddressedArray.coerce(data)
        out = np.empty_like(src.data)
        sr = ctypes.c_uint64(self.shift_register)
        check(lib().pm_lfsr_unscramble(src.data.ctypes.data_as(ctypes.c_void_p), len(src), ctypes.c_uint64(self.polynomial),
                                       int(bool(self.invert)), ctypes.byref(sr), out.ctypes.data_as(ctypes.c_void_p)))
        self.shift_register = sr.value
        return AddressedArray(out, src.address)
