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
        whole = getattr(self, "_frame_crc", None)
        if whole is not None:
            self.CalculatedCRC, self.CarriedCRC = whole
            self.ValidCRC = self.CarriedCRC == self.CalculatedCRC
            return self.ValidCRC
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
    out = [PacketMeta.from_bytes(data[k, :lens[k]].tobytes(), addrs[k], decoder_name, corr[k]) for k in range(len(rows))]
    from ._native import PKT_MAX
    for k, n in enumerate(lens):
        if n >= PKT_MAX:        # possibly the head of a longer AX.25 frame (pymodem_amd.h, PM_PKT_MAX): the codec's CRC is the whole frame's
            out[k]._frame_crc = (int(rows["calculated_crc"][k]), int(rows["carried_crc"][k]))
    return out


def _stamp(rows, c):
    """rows['source_decoder'] = c unless the codec has written it already (pm_codec_set_source; one chain's rows come from one codec,
    so the first and the last say it all): a strided pass over rows that another core has just written costs ~1 ms per recording."""
    sd = rows["source_decoder"]
    if len(sd) and (sd[0] != c or sd[-1] != c):
        sd[...] = c


class PacketTable:
    """All chains' packets in config order (chain c's records are contiguous, source_decoder = chain index).  The fast path of
    the group executor and of the multi-GPU gather: CRC/header validity come from the native codec, Correlate runs natively,
    PacketMeta objects exist only if asked for.

    Two storage forms behind one interface.  `heads` always exists: one record per packet with the pm_packet_head fields
    (streamaddress, len, CRCs, validity, source_decoder, correlated_count) -- everything the de-dup and the counters read.
      full     heads IS the array of pm_packet rows (1.3 KB apart), as the codecs wrote them (single rank);
      compact  heads is a dense 40-byte array indexed out of the gathered wire streams; the payloads stay where the exchange
               left them and `rows` expands them only if somebody asks (rank 0 of a multi-GPU job never does)."""

    def __init__(self, rows_by_chain, names):
        self.names = list(names)
        self.counts = [len(rows_by_chain.get(c, ())) for c in range(len(names))]
        parts = []
        for c in range(len(names)):
            r = rows_by_chain.get(c)
            if r is not None and len(r):
                _stamp(r, c)                     # in place: the rows are the executor's own, fresh from the codec
                parts.append(r)
        self._rows = self._stack(parts)
        self.heads = self._rows
        self._streams = None
        self.unique_idx = None

    @classmethod
    def from_array(cls, rows, counts, names):
        """All chains' rows already in one array, chain by chain in config order (counts[c] rows of chain c)."""
        self = cls.__new__(cls)
        self.names, self.counts, self._rows, self.unique_idx = list(names), [int(c) for c in counts], rows, None
        self.heads, self._streams = rows, None
        assert sum(self.counts) == len(rows) and len(self.counts) == len(self.names)
        return self

    @classmethod
    def from_streams(cls, streams, counts, names):
        """Compact form from wire streams (pm_packets_pack output, one uint8 array per contributing rank, in rank order).
        Records are put in chain order if the ranks did not own consecutive chain blocks."""
        from ._native import head_dtype
        self = cls.__new__(cls)
        self.names, self.counts, self._rows, self.unique_idx = list(names), [int(c) for c in counts], None, None
        total = sum(self.counts)
        heads = np.empty(total, dtype=head_dtype())
        where = np.empty(total, dtype=np.int64)         # payload offset inside its stream
        which = np.empty(total, dtype=np.int32)         # stream index
        at = 0
        for si, buf in enumerate(streams):
            if len(buf) == 0:
                continue
            buf = np.ascontiguousarray(buf)
            streams[si] = buf
            k = check(lib().pm_packets_index(buf.ctypes.data_as(ctypes.c_void_p), len(buf), heads[at:].ctypes.data_as(ctypes.c_void_p),
                                             where[at:].ctypes.data_as(ctypes.c_void_p), total - at))
            which[at:at + k] = si
            at += k
        assert at == total
        src = heads["source_decoder"]
        if total and np.any(src[1:] < src[:-1]):
            order = np.argsort(src, kind="stable")
            heads, where, which = heads[order], where[order], which[order]
        self.heads, self._streams, self._where, self._which = heads, list(streams), where, which
        return self

    @property
    def rows(self):
        """Full pm_packet rows (expanded on first use in the compact form)."""
        if self._rows is None:
            rows = np.zeros(len(self.heads), dtype=packet_dtype())
            for name in self.heads.dtype.names:
                rows[name] = self.heads[name]
            data = rows["data"]
            for k in range(len(rows)):
                n, o = int(self.heads["len"][k]), int(self._where[k])
                data[k, :n] = self._streams[self._which[k]][o:o + n]
            self._rows = rows
        elif self._streams is not None:
            self._rows["correlated_count"] = self.heads["correlated_count"]
        return self._rows

    def payload(self, i):
        """Packet i's bytes."""
        n = int(self.heads["len"][i])
        if self._streams is not None:
            o = int(self._where[i])
            return self._streams[self._which[i]][o:o + n].tobytes()
        return self._rows["data"][i, :n].tobytes()

    @staticmethod
    def _stack(parts):
        """Chain-ordered row blocks -> one array.  No copy when they are consecutive slices of one array, which is how the group
        executor delivers them (chain_execute._host_rows)."""
        if not parts:
            return np.zeros(0, dtype=packet_dtype())
        if len(parts) == 1:
            return parts[0]
        base = parts[0].base
        if isinstance(base, np.ndarray) and base.flags.c_contiguous and all(p.base is base for p in parts):
            addr = parts[0].ctypes.data
            for p in parts:
                if p.ctypes.data != addr or not p.flags.c_contiguous:
                    break
                addr += p.nbytes
            else:
                total = sum(len(p) for p in parts)
                if base.dtype == parts[0].dtype:
                    start = (parts[0].ctypes.data - base.ctypes.data) // base.dtype.itemsize
                    return base[start:start + total]
                # (the executor inside the library hands its rows out as views of a byte array over its own memory)
                return np.ndarray((total,), dtype=parts[0].dtype, buffer=base, offset=parts[0].ctypes.data - base.ctypes.data)
        return np.concatenate(parts)

    def correlate(self, address_distance):
        """packet_meta.py:230-271 on the record heads.  Sets unique_idx (records of the unique packets, by stream address) and
        the correlated decoders per unique packet."""
        n = len(self.heads)
        counts = (ctypes.c_int64 * max(len(self.counts), 1))(*self.counts)
        uniq = np.empty(max(n, 1), dtype=np.int64)
        corr = np.empty(max(n, 1), dtype=np.int32)          # every valid packet names its decoder exactly once
        heads = self.heads
        assert n == 0 or heads.flags.c_contiguous
        k = check(lib().pm_correlate_strided(heads.ctypes.data_as(ctypes.c_void_p), heads.strides[0] if n else 40, counts, len(self.counts),
                                             float(address_distance), uniq.ctypes.data_as(ctypes.c_void_p),
                                             corr.ctypes.data_as(ctypes.c_void_p), len(corr))) if n else 0
        self.unique_idx = uniq[:k]
        cc = heads["correlated_count"][self.unique_idx]
        self._corr, self._corr_ends = corr, np.cumsum(cc)
        self._unique_decoders = None
        return self

    @property
    def unique_decoders(self):
        """Names of the decoders that produced each unique packet (built on first use)."""
        if self._unique_decoders is None:
            cc = self.heads["correlated_count"][self.unique_idx]
            self._unique_decoders = [[self.names[d] for d in self._corr[e - c:e]] for c, e in zip(cc.tolist(), self._corr_ends.tolist())]
        return self._unique_decoders

    def CountGood(self):
        return int(len(self.unique_idx))

    def CountBad(self):
        return int(np.count_nonzero((self.heads["valid_crc"] == 0) | (self.heads["valid_header"] == 0)))

    def packets(self, chain):
        """Materialise chain `chain`'s packets as PacketMeta objects."""
        lo = sum(self.counts[:chain])
        return rows_to_packets(self.rows[lo:lo + self.counts[chain]], self.names[chain])

    def unique_packets(self):
        out = []
        h = self.heads
        for i, decs in zip(self.unique_idx.tolist(), self.unique_decoders):
            p = PacketMeta.from_bytes(self.payload(i), h["streamaddress"][i], self.names[int(h["source_decoder"][i])], h["bytes_corrected"][i])
            p.CalculatedCRC, p.CarriedCRC = int(h["calculated_crc"][i]), int(h["carried_crc"][i])
            p.ValidCRC, p.ValidHeader = bool(h["valid_crc"][i]), bool(h["valid_header"][i])
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
