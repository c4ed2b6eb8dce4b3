"""Codec stage objects: AX25Codec (ax25.py:11-93) and IL2PCodec (il2p.py:110-519 with rs_functions.py,
gf_functions.py, lfsr.py:54-92) of the reference, as native C++ state machines behind pm_codec_*."""
import ctypes

import numpy as np

from ._native import check, lib, packet_dtype, quick
from .data_classes import AddressedArray
from .packet_meta import rows_to_packets
from .string_ops import check_boolean


class _NativeCodec:
    _kind = 0
    _h = None

    def _handle(self):
        if self._h is None:
            h = ctypes.c_void_p()
            check(quick().pm_codec_create(self._kind, int(self.collect_trailing_crc), int(self.disable_rs), int(self.min_distance),
                                        int(self.sync_tolerance), 0, ctypes.byref(h)))
            self._h = h
        return self._h

    def decode_pending(self, data):
        """Run the decoder over `data`; the packets stay queued inside the native codec.  -> how many are waiting."""
        src = AddressedArray.coerce(data)
        pending = ctypes.c_int64()
        check(lib().pm_codec_decode(self._handle(), src.data.ctypes.data_as(ctypes.c_void_p),
                                    src.address.ctypes.data_as(ctypes.c_void_p), len(src), ctypes.byref(pending)))
        return pending.value

    def fetch_into(self, rows):
        """Move queued packets into `rows` (a C-contiguous pm_packet array, may be uninitialised: every byte is written)."""
        if len(rows):
            assert rows.flags.c_contiguous and rows.dtype == packet_dtype()
            got = ctypes.c_int64()
            check(lib().pm_codec_fetch(self._handle(), rows.ctypes.data_as(ctypes.c_void_p), len(rows), ctypes.byref(got)))
            assert got.value == len(rows)
        return rows

    def decode_rows(self, data):
        """Like decode(), but the packets stay rows of a NumPy structured array with the pm_packet layout
        (pymodem_amd._native.packet_dtype): CRC and header validity already filled by the native codec."""
        return self.fetch_into(np.empty(self.decode_pending(data), dtype=packet_dtype()))

    def decode(self, data):
        """list[AddressedData] | AddressedArray -> list[PacketMeta] (data, streamaddress, SourceDecoder, BytesCorrected)."""
        return rows_to_packets(self.decode_rows(data), self.identifier)

    def __del__(self):
        try:
            if self._h is not None:
                quick().pm_codec_destroy(self._h)
        except Exception:
            pass


class AX25Codec(_NativeCodec):
    _kind = 0

    def __init__(self, **kwargs):
        self.min_packet_length = kwargs.get('min_packet_length', 18)
        self.max_packet_length = kwargs.get('max_packet_length', 1023)
        self.identifier = kwargs.get('ident', 1)
        self.collect_trailing_crc, self.disable_rs, self.min_distance, self.sync_tolerance = False, False, 0, 0


class IL2PCodec(_NativeCodec):
    _kind = 1

    def __init__(self, **kwargs):
        self.collect_trailing_crc = kwargs.get('crc', True)
        self.identifier = kwargs.get('ident', 1)
        self.min_distance = kwargs.get('min_dist', 0)
        self.disable_rs = kwargs.get('disable_rs', False)
        self.sync_tolerance = kwargs.get('sync_tol', 0)

    def StringOptionsRetune(self, options):   # il2p.py:140-144
        self.collect_trailing_crc = check_boolean(options.get('crc', 'yes'))
        self.disable_rs = check_boolean(options.get('disable_rs', 'no'))
        self.min_distance = int(options.get('min_dist', self.min_distance))
        self.sync_tolerance = int(options.get('sync_tol', self.sync_tolerance))
