"""Packet records, validation and the cross-chain de-dup (PacketMeta packet_meta.py:178-208,
PacketMetaArray packet_meta.py:210-271,283-305 of the reference).  CRC, header validation and Correlate run in
the native library (pm_crc16_ccitt, pm_correlate); the text reports of the reference are out of scope."""
import ctypes
from collections import Counter

from ._native import Packet, check, lib


class ReportStyle:
    def __init__(self, options):
        self.destination = options.get('destination', 'std_out')
        self.style = options.get('style', 'raw')


class PacketMeta:
    def __init__(self):
        self.data = []
        self.streamaddress = 0
        self.source_sample_rate = 0.0
        self.CalculatedCRC = 0
        self.CarriedCRC = 0
        self.ValidCRC = False
        self.ValidHeader = False
        self.SourceDecoder = 0
        self.BytesCorrected = 0
        self.CorrelatedDecoders = []
        self.SlicedIQSamples = []

    @classmethod
    def from_native(cls, rec, decoder_name):
        p = cls()
        p.data = list(bytes(rec.data[:rec.len]))
        p.streamaddress = int(rec.streamaddress)
        p.SourceDecoder = decoder_name
        p.BytesCorrected = int(rec.bytes_corrected)
        return p

    def _native(self, source_index):
        r = Packet()
        r.streamaddress = int(self.streamaddress)
        n = min(len(self.data), len(r.data))
        r.len = n
        ctypes.memmove(r.data, bytes(bytearray(int(b) & 0xFF for b in self.data[:n])), n)
        r.bytes_corrected = int(self.BytesCorrected)
        r.calculated_crc, r.carried_crc = int(self.CalculatedCRC), int(self.CarriedCRC)
        r.valid_crc, r.valid_header = int(bool(self.ValidCRC)), int(bool(self.ValidHeader))
        r.source_decoder = source_index
        return r

    def CalcCRC(self):                        # packet_meta.py:197-203, crc_functions.py:9-61
        raw = bytes(bytearray(int(b) & 0xFF for b in self.data))
        self.CarriedCRC = int((raw[-1] * 256) + raw[-2])
        self.CalculatedCRC = lib().pm_crc16_ccitt(raw, len(raw) - 2)
        self.ValidCRC = self.CarriedCRC == self.CalculatedCRC
        return self.ValidCRC

    def Validate(self):                       # packet_meta.py:205-208 with ValidateHeader :21-41
        d = self.data
        ok = len(d) > 15
        if ok:
            for b in d[:7]:                   # the reference's sub-field index never resets: bytes 0..6 only
                ch = int(b) >> 1
                if (ch < 32 or ch > 126) and ch != 0:
                    ok = False
        self.ValidHeader = ok


class PacketMetaArray:
    def __init__(self):
        self.raw_packet_arrays = []
        self.unique_packet_array = []

    def add(self, array):
        self.raw_packet_arrays.append(array)

    def CalcCRCs(self):
        for array in self.raw_packet_arrays:
            for packet in array:
                packet.CalcCRC()
                packet.Validate()

    def Correlate(self, **kwargs):
        """packet_meta.py:230-271, evaluated by pm_correlate.  Decoders are identified by their SourceDecoder value."""
        self.address_distance = kwargs.get('address_distance', 1000)
        names, index = [], {}
        flat, counts = [], (ctypes.c_int64 * max(len(self.raw_packet_arrays), 1))()
        for c, array in enumerate(self.raw_packet_arrays):
            counts[c] = len(array)
            for p in array:
                key = p.SourceDecoder
                if key not in index:
                    index[key] = len(names)
                    names.append(key)
                flat.append(p)
        recs = (Packet * max(len(flat), 1))()
        for k, p in enumerate(flat):
            recs[k] = p._native(index[p.SourceDecoder])
        uniq = (ctypes.c_int64 * max(len(flat), 1))()
        corr = (ctypes.c_int32 * max(4 * len(flat), 1))()
        n = check(lib().pm_correlate(recs, counts, len(self.raw_packet_arrays), float(self.address_distance), uniq, corr, len(corr)))
        self.unique_packet_array = []
        w = 0
        for k in range(n):
            p = flat[uniq[k]]
            cnt = recs[uniq[k]].correlated_count
            p.CorrelatedDecoders = [names[corr[w + j]] for j in range(cnt)]
            w += cnt
            self.unique_packet_array.append(p)
        decoder_list = [d for p in self.unique_packet_array for d in p.CorrelatedDecoders]
        self.DecoderUniqueHistogram = Counter(p.SourceDecoder for p in self.unique_packet_array if len(p.CorrelatedDecoders) == 1)
        self.DecoderHistogram = Counter(decoder_list)

    def CountBad(self):
        self.bad_count = sum(1 for arr in self.raw_packet_arrays for p in arr if (p.ValidCRC is False) or (p.ValidHeader is False))
        return self.bad_count

    def CountGood(self):
        self.good_count = sum(1 for p in self.unique_packet_array if p.ValidCRC and p.ValidHeader)
        return self.good_count
