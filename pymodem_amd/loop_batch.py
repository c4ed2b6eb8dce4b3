"""Many recordings through carrier-loop chains at once (pm_lbatch, csrc/pm_loopbatch.hip).

A BPSK / MPSK / QPSK / AFSK-PLL chain spends its time in one sequential carrier loop (psk.py:173-189, psk.py:734-747,
psk.py:434-467, afsk_pll.py:153-165): one GPU lane steps it slower than one host core does, and nothing makes a single loop faster
(DESIGN.md 4.5).  The GPU's answer is width: a loop needs one lane, so the loops of hundreds of recordings x chains run in the time
of one.  `LoopBatch` is the object that puts them side by side: R recordings x the C chains of a group that shares its front end
(band-pass, AGC, Hilbert pair; the chains of configs/qpsk_2400.json differ in carrier_freq only) go through ONE engine run, in time
chunks, every loop of every recording in one launch per chunk.  `process_recordings_device` is chain_execute.process_chains_device
for a list of recordings: engine -> all slicers in batches -> LFSR + codec per recording.

Results are those of process_chain on every recording, bit for bit (tests/test_gpu_loopbatch.py).
"""
import ctypes

import numpy as np

from . import _native as N
from ._native import check, lib
from .data_classes import SignBits
from .device import Context, DeviceBuffer
from .modems import AFSKPLLModem, BPSKModem, MPSKModem, QPSKModem

_MODEM_ID = {BPSKModem: N.MODEM_BPSK, MPSKModem: N.MODEM_MPSK, AFSKPLLModem: N.MODEM_AFSK_PLL, QPSKModem: N.MODEM_QPSK}


def _output_taps(modem):
    return modem.output_lpf if isinstance(modem, AFSKPLLModem) else modem.rrc_taps


def group_key(modem):
    """Chains whose modems have equal keys can share one engine run: same kind, same front end, same output filter, same tables;
    what is left to differ is the loop (carrier_freq, loop filter, PI constants)."""
    if type(modem) not in _MODEM_ID:
        return None
    fe = modem.front_end_key() if hasattr(modem, "front_end_key") else (
        "qpsk", float(modem.sample_rate), modem.input_bpf.tobytes(),
        (modem.AGC.attack_rate, modem.AGC.decay_rate, modem.AGC.sustain_time, modem.AGC.target_amplitude))
    extra = modem.phase_error_table.tobytes() if isinstance(modem, MPSKModem) else b""
    return (type(modem).__name__, fe, _output_taps(modem).tobytes(), np.asarray(modem.wavetable).tobytes(), extra)


class LoopBatch:
    """One pm_lbatch engine: `recordings` recordings (at most) x the chains of `modems` per run.  `modems`: the modem objects of ONE
    recording's chains (equal group_key); their loop parameters and initial states are what every recording starts from."""

    def __init__(self, modems, recordings, ctx=None, chunk=0):
        keys = {group_key(m) for m in modems}
        if len(keys) != 1 or None in keys:
            raise ValueError("LoopBatch: the chains of a run must be carrier-loop modems that share front end, output filter and tables")
        self.ctx = ctx or Context.default()
        self.recordings, self.chains = int(recordings), len(modems)
        lead = modems[0]
        self.kind = type(lead)
        self.quadrature = isinstance(lead, (MPSKModem, QPSKModem))
        d = N.LBatchDesc()
        keep = []

        def vec(x, dtype=np.float64):
            a = np.ascontiguousarray(x, dtype=dtype)
            keep.append(a)
            return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double if dtype == np.float64 else ctypes.c_int32)), len(a)
        d.modem, d.recordings, d.chains, d.chunk = _MODEM_ID[self.kind], self.recordings, self.chains, int(chunk)
        d.input_fir, d.n_input_fir = vec(lead.input_bpf)
        if isinstance(lead, MPSKModem):
            d.hilbert, d.n_hilbert = vec(lead.hilbert_taps)
            d.hilbert_delay = lead.hilbert_delay
            d.pd_table = vec(lead.phase_error_table.reshape(-1), np.int32)[0]
        d.output_fir, d.n_output_fir = vec(_output_taps(lead))
        a = lead.AGC
        d.agc = N.AGCParams(a.attack_rate, a.decay_rate, a.sustain_time, a.sample_rate, a.target_amplitude)
        loops = (N.Loop * self.chains)()
        for c, m in enumerate(modems):
            ctypes.memmove(ctypes.byref(loops[c]), m._loop0, ctypes.sizeof(N.Loop))
        d.loops = loops
        d.wavetable = vec(lead.wavetable)[0]
        self._h = ctypes.c_void_p()
        check(lib().pm_lbatch_create(self.ctx.handle, ctypes.byref(d), ctypes.byref(self._h)))
        self._front = None

    def geometry(self, n):
        """(samples per demodulated stream, final-filter outputs per chunk, chunks) for recordings of n samples."""
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(lib().pm_lbatch_geometry(self._h, int(n), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    @property
    def front(self):
        """The engine's own context (band-pass, AGC, Hilbert pair of the next chunk) as a borrowed Context, for profile_read()."""
        if self._front is None:
            f = Context.__new__(Context)
            f._h, f.device = ctypes.c_void_p(lib().pm_lbatch_front_ctx(self._h)), self.ctx.device
            f.close = lambda: None                          # the engine owns it
            self._front = f
        return self._front

    @property
    def tail(self):
        """The engine's third context (the matched filters of chunk t - 1 beside the loops of chunk t), borrowed, for profile_read()."""
        if getattr(self, "_tail", None) is None:
            self._tail = Context.borrowed(lib().pm_lbatch_tail_ctx(self._h), self.ctx.device)
        return self._tail

    @property
    def loop(self):
        """The engine's loop context when the carrier loops have compute units of their own (csrc/pm_loopbatch.hip), borrowed, for
        profile_read(); None when they run on the caller's context."""
        if getattr(self, "_loop", None) is None:
            h = lib().pm_lbatch_loop_ctx(self._h)
            self._loop = Context.borrowed(h, self.ctx.device) if h else False
        return self._loop or None

    @property
    def slicing(self):
        """The engine's slicer context (a sliced run's row slicers, beside the matched filters of the next chunk), borrowed, for profile_read()."""
        if getattr(self, "_slicing", None) is None:
            self._slicing = Context.borrowed(lib().pm_lbatch_slice_ctx(self._h), self.ctx.device)
        return self._slicing

    def _bits(self, r, nout, slot):
        stride = ((nout + 63) // 64 + 1 + 7) // 8 * 8
        streams = r * self.chains
        bits_i = self.ctx.scratch(("lbatch", id(self), slot, "i"), streams * stride, np.uint64)
        bits_q = self.ctx.scratch(("lbatch", id(self), slot, "q"), streams * stride, np.uint64) if self.quadrature else None
        return stride, bits_i, bits_q

    def reserve(self, recordings, n, slot=0):
        """Allocate the bitmap set of `slot` for `recordings` recordings of n samples now (a warm-up's job: the first run of a size
        would otherwise do it, with the allocator's wait inside the run)."""
        nout = self.geometry(n)[0]
        if nout >= 1:
            self._bits(int(recordings), nout, slot)

    def run(self, audios, slot=0):
        """audios: DeviceBuffers of int16 recordings of equal length (they may be the same buffer).  Enqueues the whole run and
        returns [[SignBits of chain 0, chain 1, ...] per recording]; the bitmaps are complete when this context's stream gets there
        (slice on it, or behind an event recorded on it).  `slot`: which of the caller's bitmap sets to write (a set stays untouched
        until the same slot is used again)."""
        r = len(audios)
        if not 1 <= r <= self.recordings:
            raise ValueError(f"LoopBatch.run: {r} recordings, the engine was made for {self.recordings}")
        n = audios[0].n
        for a in audios:
            if not isinstance(a, DeviceBuffer) or a.dtype != np.dtype(np.int16) or a.n != n:
                raise ValueError("LoopBatch.run: recordings must be int16 DeviceBuffers of equal length")
        nout = self.geometry(n)[0]
        if nout < 1:
            raise ValueError(f"input of {n} samples is shorter than the filters of the chain")
        stride, bits_i, bits_q = self._bits(r, nout, slot)
        ptrs = (ctypes.c_void_p * r)(*[a.ptr.value for a in audios])
        got = ctypes.c_int64()
        check(lib().pm_lbatch_run(self._h, ptrs, r, n, bits_i.ptr, bits_q.ptr if bits_q is not None else None, stride, ctypes.byref(got)))
        out = []
        for k in range(r):
            row = []
            for c in range(self.chains):
                s = k * self.chains + c
                row.append(SignBits(bits_i.view(s * stride, stride), bits_q.view(s * stride, stride) if bits_q is not None else None, got.value))
            out.append(row)
        return out

    def _sliced_room(self, r, nout, slicers, slot):
        # room per stream: 1.5 x the nominal byte count (a clock pulled by its crossings runs at most twice nominal; real streams stay
        # within a few percent); a row that needs more says so in its record and the caller slices that run the other way
        nominal = max(nout * sl.bits_per_symbol / (8.0 * sl.samples_per_symbol) for sl in slicers)
        cap = (int(nominal * 1.5) + 64 + 7) // 8 * 8
        rows = r * self.chains
        data = self.ctx.scratch(("lbatch", id(self), slot, "sliced-data"), rows * cap, np.uint8)
        steps = self.ctx.scratch(("lbatch", id(self), slot, "sliced-steps"), rows * cap, np.uint16)
        recs = self.ctx.scratch(("lbatch", id(self), slot, "sliced-recs"), rows * ctypes.sizeof(N.RowSliceRec), np.uint8)
        return cap, rows, data, steps, recs

    def reserve_sliced(self, recordings, n, slicers, slot=0):
        """reserve() for run_sliced(): the rows' output blocks for `recordings` recordings of n samples, now."""
        nout = self.geometry(n)[0]
        if nout >= 1:
            self._sliced_room(int(recordings), nout, slicers, slot)

    def run_sliced(self, audios, slicers, slot=0):
        """run() with the slicers inside the engine (pm_lbatch_run_sliced): every stream is sliced chunk by chunk behind its matched
        filter by a lane of its own, no sign bitmap of a whole recording is kept and nothing is left to slice when the run ends.
        `slicers`: one slicer object per chain -- its parameters; every stream starts from the just-tuned state (slicer.py:49-56).
        -> SlicedRun (complete when this context's stream gets there)."""
        r = len(audios)
        if not 1 <= r <= self.recordings:
            raise ValueError(f"LoopBatch.run_sliced: {r} recordings, the engine was made for {self.recordings}")
        if len(slicers) != self.chains:
            raise ValueError("LoopBatch.run_sliced: one slicer per chain")
        n = audios[0].n
        for a in audios:
            if not isinstance(a, DeviceBuffer) or a.dtype != np.dtype(np.int16) or a.n != n:
                raise ValueError("LoopBatch.run_sliced: recordings must be int16 DeviceBuffers of equal length")
        nout = self.geometry(n)[0]
        if nout < 1:
            raise ValueError(f"input of {n} samples is shorter than the filters of the chain")
        params = (N.SlicerParams * self.chains)(*[sl._params() for sl in slicers])
        cap, rows, data, steps, recs = self._sliced_room(r, nout, slicers, slot)
        ptrs = (ctypes.c_void_p * r)(*[a.ptr.value for a in audios])
        got = ctypes.c_int64()
        check(lib().pm_lbatch_run_sliced(self._h, ptrs, r, n, params, self.chains, data.ptr, steps.ptr, cap, recs.ptr, ctypes.byref(got)))
        return SlicedRun(self.ctx, data, steps, recs, cap, rows, got.value)

    def close(self):
        if self._h:
            lib().pm_lbatch_destroy(self._h)                 # waits for both of the engine's streams
            self._h = ctypes.c_void_p()
            pool = self.ctx.__dict__.get("_pool", {})
            for tag in [t for t in pool if isinstance(t, tuple) and t[:2] == ("lbatch", id(self))]:
                pool.pop(tag).free()                         # the bitmap sets
            self.ctx.__dict__.get("_pool_views", {}).clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlicedRun:
    """What a sliced engine run left in device memory: per stream (row = recording * chains + chain) its data bytes, address steps and
    record.  records() and fetch() want the run complete (synchronise the context it ran on first)."""

    def __init__(self, ctx, data, steps, recs, cap, rows, nout):
        self.ctx, self.data, self.steps, self.recs, self.cap, self.rows, self.nout = ctx, data, steps, recs, cap, rows, nout
        self._records = None

    def records(self, copy_ctx=None):
        if self._records is None:
            self._records = self.recs.download(self.rows * ctypes.sizeof(N.RowSliceRec), ctx=copy_ctx).view(N.rowslice_dtype())
        return self._records

    def ok(self):
        """False if a row overflowed its room or took an address step beyond 16 bits: slice that run with slice_batch instead."""
        return not self.records()["flags"].any()

    def state_into(self, row, slicer):
        """The end state of the row's slicer, as slice() would have left it in the object."""
        rec, st = self.records()[row], slicer._state
        st.phase_clock, st.last_i_negative, st.last_q_negative = float(rec["clk"]), int(rec["li_neg"]), int(rec["lq_neg"])
        st.working_byte, st.working_bits, st.state_register, st.streamaddress = int(rec["wbyte"]), int(rec["wbits"]), int(rec["sreg"]), int(rec["seen"])
        slicer.phase_clock, slicer.streamaddress = st.phase_clock, st.streamaddress

    def fetch(self, row0, nrows, copy_ctx=None, tag=None):
        """-> [AddressedArray of row row0, row0 + 1, ...]: one gather launch and ONE copy to the host for the lot (on copy_ctx's stream)."""
        from .data_classes import AddressedArray
        ctx = copy_ctx or self.ctx
        recs = self.records()[row0:row0 + nrows]
        counts = np.minimum(recs["count"], self.cap)
        sizes = (2 * counts + 7) // 8 * 8 + (counts + 7) // 8 * 8
        offs = np.concatenate(([0], np.cumsum(sizes)))
        used = int(offs[-1])
        out = []
        if used:
            # (host blocks in 8 MB size classes: streams run within a few percent of their nominal counts, so the groups of a run ask
            # for the same class and the page-locked blocks go round; the worst case, 1.5 x that, would be filled and pinned for nothing)
            room = (used + (8 << 20) - 1) // (8 << 20) * (8 << 20)
            block = ctx.scratch((tag if tag is not None else ("sliced-run", id(self)), "dense"), room, np.uint8)
            check(lib().pm_rows_gather(ctx.handle, self.recs.ptr, self.data.ptr, self.steps.ptr, self.cap, row0, nrows, block.ptr, block.n))
            host = block.download(used, recycle=True, ctx=ctx, room=room)
        for k in range(nrows):
            c, o = int(counts[k]), int(offs[k])
            if c == 0:
                out.append(AddressedArray(np.zeros(0, np.uint8), np.zeros(0, np.int64)))
                continue
            sw = (2 * c + 7) // 8 * 8
            out.append(AddressedArray.from_steps(host[o + sw:o + sw + c], host[o:o + 2 * c].view(np.uint16), int(recs["first_addr"][k])))
        return out


_ENGINES = {}


def engine_for(modems, recordings, ctx=None, chunk=0):
    """The cached engine for this group of modems on this context (made for at least `recordings` recordings)."""
    ctx = ctx or Context.default()
    key = (id(ctx), group_key(modems[0]), tuple(bytes(m._loop0) for m in modems), int(chunk))
    hit = _ENGINES.get(key)
    if hit is None or hit.recordings < recordings:
        if hit is not None:
            hit.close()
        hit = _ENGINES[key] = LoopBatch(modems, recordings, ctx, chunk)
    return hit


def close_engines():
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()


def process_recordings_device(chain_sets, audios, ctx=None, chunk=0, rows=False, chain_ids=None, stages=None, slot=0, defer=False):
    """chain_sets[k] = the chains [name, modem, slicer, stream, codec] of recording k (every recording brings the same group of
    chains, as a service decoding successive recordings with one config does), audios[k] its int16 samples (host array or
    DeviceBuffer; equal lengths).  -> [[packets of chain 0, ...] per recording], identical to chain_execute.process_chain on each
    (rows=True: pm_packet rows instead of PacketMeta lists).  All carrier loops of all recordings advance together.
    defer=True: returns when the engine's run is complete, with a function that does the rest (the rows to the host, LFSR + codec) and
    returns the result -- a service calls it on another thread while this one starts the next batch's run (with the other `slot`): the
    host's work on batch k then lies beside the GPU's on batch k + 1 instead of behind it."""
    from .chain_execute import _host_rows, _host_stages, _pool
    from .slicer import slice_batch
    ctx = ctx or Context.default()
    r = len(chain_sets)
    if r == 0:
        return (lambda: []) if defer else []
    dev = []
    for a in audios:
        if isinstance(a, DeviceBuffer):
            dev.append(a)
        else:
            a = np.asarray(a)
            if a.dtype != np.int16:
                raise ValueError("process_recordings_device takes int16 recordings")
            dev.append(ctx.upload(a))
    nchains = len(chain_sets[0])
    # groups of chains that can share an engine, by position in the chain list (the same for every recording)
    groups = {}
    for c, ch in enumerate(chain_sets[0]):
        k = group_key(ch[1])
        if k is None:
            raise ValueError(f"chain {ch[0]!r} is not a carrier-loop chain: use chain_execute.process_chains_device")
        groups.setdefault(k, []).append(c)
    for cs in chain_sets:
        if len(cs) != nchains or any(group_key(cs[c][1]) != k for k, members in groups.items() for c in members):
            raise ValueError("every recording must bring the same group of chains")
    import os
    import time
    t0 = time.perf_counter()
    # One group of chains, every slicer just tuned and the same from recording to recording (a service's chains are made from one config
    # per recording): the slicers run INSIDE the engine, a lane per stream (pm_lbatch_run_sliced).  PYMODEM_AMD_LOOP_FUSED_SLICERS=0, or
    # anything else about the slicers: the engine leaves sign bitmaps and pm_slice_batch slices them afterwards, as before.
    from ._native import SlicerState
    fresh = bytes(SlicerState())
    if (len(groups) == 1 and os.environ.get("PYMODEM_AMD_LOOP_FUSED_SLICERS", "1") != "0"
            and all(hasattr(ch[2], "_params_bytes") and bytes(ch[2]._state) == fresh for cs in chain_sets for ch in cs)
            and all(cs[c][2]._params_bytes() == chain_sets[0][c][2]._params_bytes() for cs in chain_sets for c in range(nchains))):
        eng = engine_for([ch[1] for ch in chain_sets[0]], r, ctx, chunk)
        run = eng.run_sliced(dev, [ch[2] for ch in chain_sets[0]], slot=(slot, 0))
        ctx.sync_relaxed()
        t1 = time.perf_counter()
        run.records(Context.side(ctx.device, 399))          # (not on the context's own stream: the next run may be queued on it by now)
        if run.ok():
            def rest():
                for rec in range(r):
                    for c in range(nchains):
                        sl = chain_sets[rec][c][2]
                        sl._ctx = sl._ctx or ctx
                        run.state_into(rec * nchains + c, sl)
                # to the host a few hundred streams at a time (one gather, one copy), several copies in flight on streams of their own, and a
                # recording whose streams are there goes to the host stage at once, beside the copies still to come
                parts = max(1, min(int(os.environ.get("PYMODEM_AMD_LOOP_SLICE_STREAMS", "8")), r))
                per = max(1, int(os.environ.get("PYMODEM_AMD_LOOP_FETCH_ROWS", "256")) // nchains)
                cuts = [r * p // parts for p in range(parts + 1)]
                sides = [Context.side(ctx.device, 400 + p) for p in range(parts)]
                early = {}

                def part(p):
                    out = []
                    for lo in range(cuts[p], cuts[p + 1], per):
                        hi = min(lo + per, cuts[p + 1])
                        got = run.fetch(lo * nchains, (hi - lo) * nchains, sides[p], tag=("loop-sliced", p))
                        out += got
                        for rec in range(lo, hi):
                            if rows:
                                early[rec] = _pool().submit(_host_rows, chain_sets[rec], got[(rec - lo) * nchains:(rec - lo + 1) * nchains], chain_ids)
                    return out
                futs = [_pool().submit(part, p) for p in range(parts)]
                sliced = [x for f in futs for x in f.result()]
                t2 = time.perf_counter()
                if stages is not None:
                    stages["sliced"] = [sliced[rec * nchains:(rec + 1) * nchains] for rec in range(r)]
                    stages["seconds"] = {"engine": t1 - t0, "slicers": t2 - t1}
                    stages["fused_slicers"] = True
                if rows:
                    out = [early[rec].result() for rec in range(r)]
                    if stages is not None:
                        stages["seconds"]["host"] = time.perf_counter() - t2
                    return out
                futs = [[_pool().submit(_host_stages, ch, sl) for ch, sl in zip(chain_sets[rec], sliced[rec * nchains:(rec + 1) * nchains])] for rec in range(r)]
                return [[f.result() for f in futs[rec]] for rec in range(r)]
            return rest if defer else rest()
        t0 = time.perf_counter()                # (a row outgrew its room or its 16-bit steps: the whole run again, the other way)
    bitmaps = [[None] * nchains for _ in range(r)]
    for gi, (k, members) in enumerate(groups.items()):
        eng = engine_for([chain_sets[0][c][1] for c in members], r, ctx, chunk)
        got = eng.run(dev, slot=(0 if defer else slot, gi))    # (this way everything is over when the call returns: one bitmap set will do)
        for rec in range(r):
            for j, c in enumerate(members):
                sl = chain_sets[rec][c][2]
                sl._ctx = sl._ctx or ctx
                bitmaps[rec][c] = sl.sign_bitmaps(got[rec][j])
    ctx.sync_relaxed()                 # seconds: sleep through them instead of spinning in the slicer's first stream wait
    t1 = time.perf_counter()
    flat_slicers = [chain_sets[rec][c][2] for rec in range(r) for c in range(nchains)]
    flat_bits = [bitmaps[rec][c] for rec in range(r) for c in range(nchains)]
    if len(flat_slicers) >= 256:
        # Thousands of streams go through pm_slice_batch 64 at a time, each call a sequence of lockstep launches that ends in a wait.
        # Here the parallelism is in the streams, not inside one: long chunks (1024 words instead of 384: N (1 + m/L) lane-steps with m/L
        # a few percent instead of 0.6) and eight calls in flight on eight streams of their own (eight threads) -- qpsk_2400, 2048 x 8
        # quadrature streams: 2.4 ms per recording with two calls of short chunks in flight
        import os
        parts = max(1, min(int(os.environ.get("PYMODEM_AMD_LOOP_SLICE_STREAMS", "8")), r))
        words = int(os.environ.get("PYMODEM_AMD_LOOP_SLICE_CHUNK_WORDS", "1024"))
        cuts = [(r * p // parts) * nchains for p in range(parts + 1)]
        sides = [Context.side(ctx.device, 400 + p) for p in range(parts)]
        for sc in sides:
            if getattr(sc, "_loop_slice_words", None) != words:
                check(lib().pm_slicer_limits(sc.handle, words))
                sc._loop_slice_words = words
        # ... and the output crosses to the host in pm_slice_compact's form (exact counts, 16-bit address steps: 3 bytes per data byte
        # instead of 9 x the 1.5x capacity -- 18 GB instead of 80 for that run)
        # -- one group of 64 streams at a time, fetched before the next: the device blocks of a stream's context are the same for
        # every group (a key per group kept 80 GB of them alive beside the bitmaps)
        # ... and a recording whose streams are all through goes to the host stage at once (LFSR + codec on the library's threads), beside
        # the slicer groups still to come: the two stages of a run overlap instead of following each other (qpsk_2400: 0.56 + 0.75 ms per
        # recording one after the other)
        early = {}

        def part(p):
            out, rec = [], cuts[p] // nchains
            for lo in range(cuts[p], cuts[p + 1], 64):
                hi = min(lo + 64, cuts[p + 1])
                out += slice_batch(flat_slicers[lo:hi], flat_bits[lo:hi], sides[p], defer=True, compact=True, out_tag=("loop-slice", p))(sides[p])
                while rows and (rec + 1) * nchains - cuts[p] <= len(out):
                    at = rec * nchains - cuts[p]
                    early[rec] = _pool().submit(_host_rows, chain_sets[rec], out[at:at + nchains], chain_ids)
                    rec += 1
            return out
        futs = [_pool().submit(part, p) for p in range(parts)]
        sliced = [x for f in futs for x in f.result()]
    else:
        sliced, early = slice_batch(flat_slicers, flat_bits, ctx), {}
    t2 = time.perf_counter()
    if stages is not None:
        stages["sliced"] = [sliced[rec * nchains:(rec + 1) * nchains] for rec in range(r)]
        stages["seconds"] = {"engine": t1 - t0, "slicers": t2 - t1}
    out = []
    if rows:
        futs = [early[rec] if rec in early else _pool().submit(_host_rows, chain_sets[rec], sliced[rec * nchains:(rec + 1) * nchains], chain_ids)
                for rec in range(r)]
        out = [f.result() for f in futs]
        if stages is not None:
            stages["seconds"]["host"] = time.perf_counter() - t2
        return (lambda: out) if defer else out
    futs = [[_pool().submit(_host_stages, ch, sl) for ch, sl in zip(chain_sets[rec], sliced[rec * nchains:(rec + 1) * nchains])] for rec in range(r)]
    for rec in range(r):
        out.append([f.result() for f in futs[rec]])
    return (lambda: out) if defer else out
