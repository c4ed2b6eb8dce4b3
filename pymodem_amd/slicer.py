"""Slicer stage objects (BinarySlicer slicer.py:9-107, QuadratureSlicer slicer.py:109-242 of the reference).
`slice()` runs on the GPU: sign bitmap (pm_signs_f64) -> chunk-parallel timing recovery (pm_slice_*).
FourLevelSlicer is not provided: it raises NameError in the reference (slicer.py:312,432)."""
import ctypes

import numpy as np

from ._native import SliceJob, SlicerParams, SlicerState, check, lib
from .data_classes import AddressedArray, DeviceIQ, IQData, SignBits
from .device import Context, DeviceBuffer, NativeError


_PARAMS_CACHE = {}          # (sps, lock, bits, mask, demap) -> (SlicerParams, its bytes): shared, read-only


class _SlicerBase:
    _ctx = None
    last_stats = None
    own_key = None          # stable key given by the group executor (else tied to this object's lifetime)

    def _own_key(self):
        return self.own_key if self.own_key is not None else self._ctx.owner_key(self)

    def retune(self, **kwargs):
        self.symbol_rate = kwargs.get('symbol_rate', self.symbol_rate)
        self.lock_rate = kwargs.get('lock_rate', self.lock_rate)
        self.sample_rate = kwargs.get('sample_rate', self.sample_rate)
        self.tune()

    def StringOptionsRetune(self, options):   # slicer.py:43-47,187-191
        self.symbol_rate = float(options.get('symbol_rate', self.symbol_rate))
        self.sample_rate = float(options.get('sample_rate', self.sample_rate))
        self.lock_rate = float(options.get('lock_rate', self.lock_rate))
        self.tune()

    def tune(self):                           # slicer.py:49-56,193-202
        self.phase_clock = 0.0
        self.samples_per_symbol = self.sample_rate / self.symbol_rate
        self.rollover_threshold = (self.samples_per_symbol / 2.0) - 0.5
        self.streamaddress = 0
        self._state = SlicerState()        # phase clock, last sample sign(s), partial byte, address: carried from slice() to slice()

    def _params_entry(self):
        # built once per set of values (a batch asks every slicer twice; the executor makes eight slicers per recording)
        key = (self.samples_per_symbol, self.lock_rate, self.bits_per_symbol, self.state_mask, tuple(self.demap))
        hit = _PARAMS_CACHE.get(key)
        if hit is None:
            p = SlicerParams()
            p.samples_per_symbol = self.samples_per_symbol
            p.lock_rate = self.lock_rate
            p.bits_per_symbol = self.bits_per_symbol
            p.state_mask = self.state_mask
            for k, v in enumerate(self.demap):
                p.demap[k] = v
            if len(_PARAMS_CACHE) > 256:
                _PARAMS_CACHE.clear()
            hit = _PARAMS_CACHE[key] = (p, bytes(p))
        return hit

    def _params(self):
        return self._params_entry()[0]

    def _params_bytes(self):
        return self._params_entry()[1]

    def _bits(self, ctx, x, tag):
        if not isinstance(x, DeviceBuffer):
            x = ctx.upload(np.ascontiguousarray(x, dtype=np.float64))
        assert x.dtype == np.dtype(np.float64)
        bits = ctx.scratch((self._own_key(), tag), (x.n + 63) // 64 + 1, np.uint64)
        check(lib().pm_signs_f64(ctx.handle, x.ptr, x.n, bits.ptr))
        return bits, x.n

    def sign_bitmaps(self, samples):
        """Stage 1 of slice(): the (x >= 0) bitmap(s) of the demodulated stream -> (bits_i, bits_q | None, n)."""
        ctx = self._ctx = self._ctx or Context.default()
        if isinstance(samples, SignBits):
            return samples.bits_i, samples.bits_q, samples.n
        if isinstance(samples, (IQData, DeviceIQ)):
            bi, n = self._bits(ctx, samples.i_data, "bits_i")
            bq, nq = self._bits(ctx, samples.q_data, "bits_q")
            assert n == nq
            return bi, bq, n
        bi, n = self._bits(ctx, samples, "bits_i")
        return bi, None, n

    def _run(self, ctx, bits_i, bits_q, n):
        return slice_batch([self], [(bits_i, bits_q, n)])[0]


def slice_batch(slicers, bitmaps, ctx=None, defer=False, reserve=1.0, out_tag=None, compact=False):
    """Stage 2 of slice() for many independent streams in ONE pm_slice_batch call (shared iteration launches).
    bitmaps[k] = (bits_i, bits_q | None, n) from slicers[k].sign_bitmaps().  Returns one AddressedArray per stream.
    `ctx`: the context (stream) to run on; the bitmaps must be complete (their producer stream synchronised) if it is not the
    one that made them.  defer=True: the slicers have run (states updated) but their output is still in device memory; returns
    fetch(copy_ctx) -> the list, which copies it to the host on copy_ctx's stream (the pipelined executor fetches on another
    thread while this context already slices the next batch).  compact=True (with defer): the output crosses to the host in
    pm_slice_compact's form -- exact counts, 16-bit address steps -- a quarter of the bytes; the arrays returned are the same."""
    ctx = ctx or slicers[0]._ctx or Context.default()
    # Slicers of the same kind and state on the same bitmap(s) (chains that differ only after the slicer) are one job: run the first,
    # hand its result and end state to the others.
    first, dup_of = {}, {}
    for k, (sl, bm) in enumerate(zip(slicers, bitmaps)):
        key = (bm[0].ptr.value if bm[0] is not None else None, bm[1].ptr.value if bm[1] is not None else None, bm[2],
               sl._params_bytes(), bytes(sl._state))
        if key in first:
            dup_of[k] = first[key]
        else:
            first[key] = k
    if dup_of:
        uniq = [k for k in range(len(slicers)) if k not in dup_of]
        inner = slice_batch([slicers[k] for k in uniq], [bitmaps[k] for k in uniq], ctx, defer=True, reserve=reserve, out_tag=out_tag, compact=compact)

        def fetch_all(copy_ctx=None):
            got = dict(zip(uniq, inner(copy_ctx)))
            return [got[dup_of.get(k, k)] for k in range(len(slicers))]
        fetch_all.fetchers = getattr(inner, "fetchers", [])
        for k, src in dup_of.items():
            ctypes.memmove(ctypes.byref(slicers[k]._state), ctypes.byref(slicers[src]._state), ctypes.sizeof(SlicerState))
            slicers[k].last_stats = slicers[src].last_stats
            slicers[k].phase_clock, slicers[k].streamaddress = slicers[src].phase_clock, slicers[src].streamaddress
            slicers[k]._ctx = slicers[k]._ctx or ctx
        return fetch_all if defer else fetch_all()
    out = [None] * len(slicers)
    fetchers = []
    for base in range(0, len(slicers), 64):
        group = list(range(base, min(base + 64, len(slicers))))
        saved = [SlicerState.from_buffer_copy(slicers[k]._state) for k in group]
        try:
            fetchers.append(_slice_group(ctx, slicers, bitmaps, group, out, tight=True, reserve=reserve, out_tag=out_tag, compact=compact and defer))
        except NativeError as e:                      # a stream produced more than twice its nominal symbol count: full-size buffers
            if "capacity" not in str(e):
                raise
            for k, st in zip(group, saved):
                ctypes.memmove(ctypes.byref(slicers[k]._state), ctypes.byref(st), ctypes.sizeof(SlicerState))
            fetchers.append(_slice_group(ctx, slicers, bitmaps, group, out, tight=False, reserve=reserve, out_tag=out_tag, compact=compact and defer))

    def fetch(copy_ctx=None):
        for f in fetchers:
            f(copy_ctx)
        return out
    fetch.fetchers = fetchers
    return fetch if defer else fetch()


_TIGHT_FACTOR = 1.5
_COMPACT_HEAD = 80        # PM_COMPACT_HEAD: first and last address, 64 flag bytes


def _slice_group(ctx, slicers, bitmaps, group, out, tight, reserve=1.0, out_tag=None, compact=False):
    jobs = (SliceJob * len(group))()
    # One device block for the whole batch's output (addresses first, then bytes) and ONE device-to-host copy per batch.  The
    # hard bound is one symbol per sample; the clock can at most double its nominal rate (every crossing pulls it towards zero,
    # from where half a symbol period remains); real streams stay within a few percent of nominal, so 1.5x the nominal count is
    # tried first and the full bound only after a capacity error.
    caps, a_off, d_off, at = [], [], [], 0
    for k in group:
        n, sl = bitmaps[k][2], slicers[k]
        sl._ctx = sl._ctx or ctx                  # bitmaps made elsewhere (another stream, a caller's own kernel): adopt this context
        cap = n * sl.bits_per_symbol // 8 + 5
        if tight:
            cap = min(cap, int(n * sl.bits_per_symbol / (8.0 * sl.samples_per_symbol) * _TIGHT_FACTOR) + 64)
        caps.append(cap)
        a_off.append(at)
        at += cap * 8
    for cap in caps:
        d_off.append(at)
        at += (cap + 4 + 7) // 8 * 8
    # `reserve`: ask for that many times the bytes (a caller whose batches vary in size asks for the largest once: a growth later is a
    # free and a malloc in the middle of the pipeline); `out_tag`: the caller's own key for the block (it must stay untouched until
    # fetched: the pipelined executor rotates sixteen per slicer stream)
    block = ctx.scratch((out_tag if out_tag is not None else slicers[group[0]]._own_key(), "slice_out"), int(at * max(1.0, reserve)), np.uint8)
    for j, k in enumerate(group):
        sl, (bi, bq, n) = slicers[k], bitmaps[k]
        sl._ctx = sl._ctx or ctx
        jobs[j].d_bits_i = bi.ptr if bi is not None else None
        jobs[j].d_bits_q = bq.ptr if bq is not None else None
        jobs[j].n = n
        jobs[j].params = sl._params()
        jobs[j].d_data, jobs[j].d_addr, jobs[j].cap = block.ptr.value + d_off[j], block.ptr.value + a_off[j], caps[j]
        jobs[j].h_state = ctypes.pointer(sl._state)
    check(lib().pm_slice_batch(ctx.handle, jobs, len(group)))
    it, cl, nc = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    lib().pm_slicer_stats(ctx.handle, ctypes.byref(it), ctypes.byref(cl), ctypes.byref(nc))
    counts = [jobs[j].count for j in range(len(group))]
    for j, k in enumerate(group):
        slicers[k].last_stats = {"iterations": it.value, "chunk_len": cl.value, "chunks": nc.value}
        slicers[k].phase_clock, slicers[k].streamaddress = slicers[k]._state.phase_clock, slicers[k]._state.streamaddress

    if compact:
        # exact counts and 16-bit address steps, packed by a small kernel behind the batch: what crosses to the host is 3 bytes per
        # data byte instead of 9 and no unused capacity; the dense block is sized for the capacity once (times `reserve`)
        dense_cap = sum(_COMPACT_HEAD + (2 * c + 7) // 8 * 8 + (c + 7) // 8 * 8 for c in caps)
        dense = ctx.scratch((out_tag if out_tag is not None else slicers[group[0]]._own_key(), "slice_dense"), int(dense_cap * max(1.0, reserve)), np.uint8)
        offs, used = (ctypes.c_int64 * len(group))(), ctypes.c_size_t()
        check(lib().pm_slice_compact(ctx.handle, jobs, len(group), dense.ptr, dense.n, offs, ctypes.byref(used)))

        def fetch_compact(copy_ctx=None):
            host = dense.download(used.value, recycle=True, ctx=copy_ctx, room=dense.n)
            for j, k in enumerate(group):
                c, o = counts[j], offs[j]
                first = int(host[o:o + 8].view(np.int64)[0])
                steps = host[o + _COMPACT_HEAD:o + _COMPACT_HEAD + 2 * c].view(np.uint16)
                data = host[o + _COMPACT_HEAD + (2 * c + 7) // 8 * 8:o + _COMPACT_HEAD + (2 * c + 7) // 8 * 8 + c]
                if c and host[o + 16:o + _COMPACT_HEAD].any():               # a step beyond 16 bits: this stream's addresses in full
                    full = block.view(a_off[j], c * 8).download(ctx=copy_ctx).view(np.int64)
                    out[k] = AddressedArray(data, full)
                else:
                    out[k] = AddressedArray.from_steps(data, steps, first)
        fetch_compact.room = dense.n
        return fetch_compact

    def fetch(copy_ctx=None):
        host = block.download(at, recycle=True, ctx=copy_ctx)
        for j, k in enumerate(group):
            # views into this call's own download (a fresh array every call): no second copy
            out[k] = AddressedArray(host[d_off[j]:d_off[j] + counts[j]], host[a_off[j]:a_off[j] + counts[j] * 8].view(np.int64))
    return fetch


class BinarySlicer(_SlicerBase):
    bits_per_symbol, state_mask, demap = 1, 0x3, [0, 0, 1, 1]

    def __init__(self, **kwargs):
        self.definition = kwargs.get('config', '1200')
        self.sample_rate = kwargs.get('sample_rate', '8000')
        self.symbol_rate, self.lock_rate = {'300': (300, 0.75), '9600': (9600, 0.88), '4800': (4800, 0.88)}.get(
            self.definition, (1200, 0.75))        # slicer.py:22-33
        self.tune()

    def slice(self, samples):
        """-> AddressedArray (list[AddressedData]-compatible).  `samples`: host float64 sequence or DeviceBuffer."""
        ctx = self._ctx = self._ctx or Context.default()
        n = samples.n if isinstance(samples, (DeviceBuffer, SignBits)) else len(samples)
        if n == 0:
            return AddressedArray(np.zeros(0, np.uint8), np.zeros(0, np.int64))
        bits, _, n = self.sign_bitmaps(samples)
        return self._run(ctx, bits, None, n)


class QuadratureSlicer(_SlicerBase):
    _QPSK = [3, 1, 2, 0, 2, 3, 0, 1, 1, 0, 3, 2, 0, 2, 1, 3]
    _PRESETS = {   # slicer.py:124-165: (state_mask, bits_per_symbol, demap, symbol_rate, lock_rate)
        'qpsk_600': (0xF, 2, _QPSK, 300, 0.815), 'bpsk_300': (0x3, 1, [0, 0, 1, 1], 300, 0.815),
        'bpsk_1200': (0x3, 1, [0, 0, 1, 1], 1200, 0.9), 'qpsk_2400': (0xF, 2, _QPSK, 1200, 0.9),
        'qpsk_4800': (0xF, 2, _QPSK, 2400, 0.99), 'qpsk_3600': (0xF, 2, _QPSK, 1800, 0.99),
    }

    def __init__(self, **kwargs):
        self.sample_rate = kwargs.get('sample_rate', '8000')
        self.definition = kwargs.get('config', '600')
        self.state_mask, self.bits_per_symbol, self.demap, self.symbol_rate, self.lock_rate = self._PRESETS.get(
            self.definition, (0xF, 2, self._QPSK, 1200, 0.9))
        self.tune()

    def slice(self, iq_samples):
        ctx = self._ctx = self._ctx or Context.default()
        if isinstance(iq_samples, SignBits):
            n = iq_samples.n
        else:
            n = iq_samples.i_data.n if isinstance(iq_samples.i_data, DeviceBuffer) else len(iq_samples.i_data)
        if n == 0:
            return AddressedArray(np.zeros(0, np.uint8), np.zeros(0, np.int64))
        bi, bq, n = self.sign_bitmaps(iq_samples)
        return self._run(ctx, bi, bq, n)
