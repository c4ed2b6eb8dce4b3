"""Packet records, validation and the cross-chain de-dup (PacketMeta packet_meta.py:178-208,
PacketMetaArray packet_meta.py:210-271,283-305 of the reference).  CRC, header validation and Correlate run in
the native library (pm_crc16_ccitt, pm_correlate); the text reports of the reference are out of scope."""
import ctypes
from collections import Counter

import numpy as np

from ._native import PKT_MAX, Packet, check, lib, packet_dtype


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
        self._raw = None          # bytes twin of .data for the native paths (no per-byte Python loops)

    @classmethod
    def from_bytes(cls, raw, streamaddress, decoder_name, bytes_corrected):
        p = cls()
        p._raw = bytes(raw)
        p.data = list(p._raw)
        p.streamaddress = int(streamaddress)
        p.SourceDecoder = decoder_name
        p.BytesCorrected = int(bytes_corrected)
        return p

    @classmethod
    def from_native(cls, rec, decoder_name):
        return cls.from_bytes(ctypes.string_at(ctypes.addressof(rec) + Packet.data.offset, rec.len), rec.streamaddress,
                              decoder_name, rec.bytes_corrected)

    def raw(self):
        """The packet as bytes; rebuilt from .data if a caller changed the list's length."""
        if self._raw is None or len(self._raw) != len(self.data):
            self._raw = bytes(bytearray(int(b) & 0xFF for b in self.data))
        return self._raw

    def _native(self, source_index):
        raw = self.raw()
        r = Packet()
        r.streamaddress = int(self.streamaddress)
        n = min(len(raw), PKT_MAX)
        r.len = n
        ctypes.memmove(ctypes.addressof(r) + Packet.data.offset, raw, n)
        r.bytes_corrected = int(self.BytesCorrected)
        r.calculated_crc, r.carried_crc = int(self.CalculatedCRC), int(self.CarriedCRC)
        r.valid_crc, r.valid_header = int(bool(self.ValidCRC)), int(bool(self.ValidHeader))
        r.source_decoder = source_index
        return r

    def CalcCRC(self):                        # packet_meta.py:197-203, crc_functions.py:9-61
        raw = self.raw()
        self.CarriedCRC = int((raw[-1] * 256) + raw[-2])
        self.CalculatedCRC = lib().pm_crc16_ccitt(raw, len(raw) - 2)
        self.ValidCRC = self.CarriedCRC == self.CalculatedCRC
        return self.ValidCRC

    def Validate(self):                       # packet_meta.py:205-208 with ValidateHeader :21-41
        raw = self.raw()
        ok = len(raw) > 15
        if ok:
            for b in raw[:7]:                 # the reference's sub-field index never resets: bytes 0..6 only
                ch = b >> 1
                if (ch < 32 or ch > 126) and ch != 0:
                    ok = False
        self.ValidHeader = ok


def rows_to_packets(rows, decoder_name):
    """pm_packet rows -> list[PacketMeta] (CRC fields are left for CalcCRC, like the reference's codecs leave them)."""
    lens, addrs, corr = rows["len"].tolist(), rows["streamaddress"].tolist(), rows["bytes_corrected"].tolist()
    data = rows["data"]
    return [PacketMeta.from_bytes(data[k, :lens[k]].tobytes(), addrs[k], decoder_name, corr[k]) for k in range(len(rows))]


class PacketTable:
    """All chains' packets as one array of pm_packet rows (chain c's rows are contiguous, chains in config order,
    source_decoder = chain index).  The fast path of the group executor and of the multi-GPU gather: CRC/header validity
    come from the native codec, Correlate runs natively on the rows, PacketMeta objects exist only if asked for."""

    def __init__(self, rows_by_chain, names):
        self.names = list(names)
        self.counts = [len(rows_by_chain.get(c, ())) for c in range(len(names))]
        parts = []
        for c in range(len(names)):
            r = rows_by_chain.get(c)
            if r is not None and len(r):
                r["source_decoder"] = c          # in place: the rows are the executor's own, fresh from the codec
                parts.append(r)
        self.rows = self._stack(parts)
        self.unique_idx = None

    @staticmethod
    def _stack(parts):
        """Chain-ordered row blocks -> one array.  No copy when they are consecutive slices of one array, which is how the group
        executor delivers them (chain_execute._host_rows)."""
        if not parts:
            return np.zeros(0, dtype=packet_dtype())
        if len(parts) == 1:
            return parts[0]
        base = parts[0].base
        if isinstance(base, np.ndarray) and base.dtype == parts[0].dtype and all(p.base is base for p in parts):
            addr = parts[0].ctypes.data
            for p in parts:
                if p.ctypes.data != addr or not p.flags.c_contiguous:
                    break
                addr += p.nbytes
            else:
                start = (parts[0].ctypes.data - base.ctypes.data) // base.dtype.itemsize
                return base[start:start + sum(len(p) for p in parts)]
        return np.concatenate(parts)

    def correlate(self, address_distance):
        """packet_meta.py:230-271 on the rows.  Sets unique_idx (rows of the unique packets, by stream address) and
        correlated decoders per unique packet."""
        n = len(self.rows)
        counts = (ctypes.c_int64 * max(len(self.counts), 1))(*self.counts)
        uniq = np.zeros(max(n, 1), dtype=np.int64)
        corr = np.zeros(max(4 * n, 1), dtype=np.int32)
        k = check(lib().pm_correlate(self.rows.ctypes.data_as(ctypes.c_void_p), counts, len(self.counts), float(address_distance),
                                     uniq.ctypes.data_as(ctypes.c_void_p), corr.ctypes.data_as(ctypes.c_void_p), len(corr))) if n else 0
        self.unique_idx = uniq[:k]
        cc = self.rows["correlated_count"][self.unique_idx]
        ends = np.cumsum(cc)
        self.unique_decoders = [[self.names[d] for d in corr[e - c:e]] for c, e in zip(cc.tolist(), ends.tolist())]
        return self

    def CountGood(self):
        return int(len(self.unique_idx))

    def CountBad(self):
        return int(np.count_nonzero((self.rows["valid_crc"] == 0) | (self.rows["valid_header"] == 0)))

    def packets(self, chain):
        """Materialise chain `chain`'s packets as PacketMeta objects."""
        lo = sum(self.counts[:chain])
        return rows_to_packets(self.rows[lo:lo + self.counts[chain]], self.names[chain])

    def unique_packets(self):
        out = []
        for i, decs in zip(self.unique_idx.tolist(), self.unique_decoders):
            r = self.rows[i]
            p = PacketMeta.from_bytes(r["data"][:int(r["len"])].tobytes(), r["streamaddress"], self.names[int(r["source_decoder"])], r["bytes_corrected"])
            p.CalculatedCRC, p.CarriedCRC = int(r["calculated_crc"]), int(r["carried_crc"])
            p.ValidCRC, p.ValidHeader = bool(r["valid_crc"]), bool(r["valid_header"])
            p.CorrelatedDecoders = list(decs)
            out.append(p)
        return out


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
        nf = len(flat)
        recs = np.zeros(max(nf, 1), dtype=packet_dtype())
        for k, p in enumerate(flat):
            raw = p.raw()[:PKT_MAX]
            recs[k]["data"][:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
        recs["streamaddress"][:nf] = [int(p.streamaddress) for p in flat]
        recs["len"][:nf] = [min(len(p.data), PKT_MAX) for p in flat]
        recs["calculated_crc"][:nf] = [int(p.CalculatedCRC) for p in flat]
        recs["valid_crc"][:nf] = [int(bool(p.ValidCRC)) for p in flat]
        recs["valid_header"][:nf] = [int(bool(p.ValidHeader)) for p in flat]
        recs["source_decoder"][:nf] = [index[p.SourceDecoder] for p in flat]
        uniq = np.zeros(max(nf, 1), dtype=np.int64)
        corr = np.zeros(max(4 * nf, 1), dtype=np.int32)
        n = check(lib().pm_correlate(recs.ctypes.data_as(ctypes.c_void_p), counts, len(self.raw_packet_arrays), float(self.address_distance),
                                     uniq.ctypes.data_as(ctypes.c_void_p), corr.ctypes.data_as(ctypes.c_void_p), len(corr)))
        self.unique_packet_array = []
        w = 0
        for k in range(n):
            p = flat[int(uniq[k])]
            cnt = int(recs[int(uniq[k])]["correlated_count"])
            p.CorrelatedDecoders = [names[int(corr[w + j])] for j in range(cnt)]
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
