"""Chain execution (chain_execute.py:6-52 of the reference): demod -> slice -> stream -> codec.

`process_chain` / `multiprocess_chain` keep the reference's signatures and stage-by-stage hand-offs.
`process_chain_device` keeps the demodulated stream in HBM between modem and slicer.
`process_chains_device` runs a whole group of independent chains over one recording the way the hardware wants it:
chains with the same front end share it (every chain of afsk_1200_ax25_super_opt.json has the same input band-pass;
the qpsk_2400.json chains share band-pass, AGC and Hilbert pair), MPSK carrier loops of such a group run in one
launch (one lane each), AFSK correlator banks that share their mark filters run as one pm_afsk_correlate_group launch, all
slicers run in one pm_slice_batch call, and the host-integer stages (LFSR, codec) of the chains run concurrently in a
small thread pool (the native calls release the GIL).
`RecordingPipeline` overlaps those stages ACROSS successive recordings (separate HIP streams and threads).
`NativeChain` is the Python face of the whole-chain C entry points (pm_chain_create / pm_chain_run).
"""
import ctypes
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from ._native import Loop, check, lib, quick
from .device import Context, DeviceBuffer
from .modems import AFSKModem, AFSKPLLModem, BPSKModem, MPSKModem, QPSKModem
from .slicer import slice_batch


def process_chain(chain, input_audio):
    demod_audio = chain[1].demod(input_audio)
    sliced_data = chain[2].slice(demod_audio)
    descrambled_data = chain[3].stream_unscramble_8bit(sliced_data)
    return chain[4].decode(descrambled_data)


def multiprocess_chain(chain, input_audio, queue):
    queue.put(process_chain(chain, input_audio))


def process_chain_device(chain, input_audio, stages=None):
    """Same result as process_chain; `input_audio` may already be a DeviceBuffer.  If `stages` is a dict it
    receives the slicer output and the descrambled stream (for parity checks)."""
    demod_audio = chain[1].demod_signs(input_audio)          # last FIR writes the sign bitmap only
    sliced_data = chain[2].slice(demod_audio)
    descrambled_data = chain[3].stream_unscramble_8bit(sliced_data)
    if stages is not None:
        stages["sliced"], stages["descrambled"] = sliced_data, descrambled_data
    return chain[4].decode(descrambled_data)


def _host_stages(chain, sliced):
    return chain[4].decode(chain[3].stream_unscramble_8bit(sliced))


def _host_pending(chain, sliced):
    return chain[4].decode_pending(chain[3].stream_unscramble_8bit(sliced))


def _host_threads():
    import os
    return max(1, min(int(os.environ.get("PYMODEM_AMD_HOST_THREADS", "16")), (os.cpu_count() or 2)))


def _host_rows(chains, sliced, chain_ids=None):
    """LFSR + codec of every chain -> one pm_packet row block per chain, all of them consecutive slices of ONE array the codecs
    wrote straight into (PacketTable then takes the whole array without a copy).  Two native calls per recording
    (pm_host_decode_batch, pm_codec_fetch_batch): the library's own threads take one chain each, the interpreter lock is
    released throughout."""
    import ctypes
    from ._native import HostJob, check, lib, packet_dtype, quick
    from .data_classes import AddressedArray
    from .lfsr import LFSR
    n = len(chains)
    if not all(isinstance(ch[3], LFSR) and hasattr(ch[4], "_handle") for ch in chains):      # foreign stream / codec objects
        pool = _pool()
        counts = [f.result() for f in [pool.submit(_host_pending, ch, sl) for ch, sl in zip(chains, sliced)]]
        block = np.empty(sum(counts), dtype=packet_dtype())
        views, at = [], 0
        for c in counts:
            views.append(block[at:at + c])
            at += c
        for f in [pool.submit(ch[4].fetch_into, v) for ch, v in zip(chains, views)]:
            f.result()
        return views
    jobs = (HostJob * n)()
    keep = []
    for j, (ch, sl) in enumerate(zip(chains, sliced)):
        src = AddressedArray.coerce(sl)
        keep.append(src)
        jobs[j].codec = ch[4]._handle()
        if chain_ids is not None:               # the codec stamps its packets with the chain's place in the config as it writes them
            quick().pm_codec_set_source(jobs[j].codec, int(chain_ids[j]))
        steps = src.address_steps
        if steps is not None:                   # straight from the slicer's compact block: expanded inside the native call
            jobs[j].h_data, jobs[j].h_addr, jobs[j].n = src.data.ctypes.data, None, len(src)
            jobs[j].h_addr_delta, jobs[j].addr_first = steps[0].ctypes.data, steps[1]
        else:
            jobs[j].h_data, jobs[j].h_addr, jobs[j].n = src.data.ctypes.data, src.address.ctypes.data, len(src)
        jobs[j].lfsr_poly, jobs[j].lfsr_state, jobs[j].lfsr_invert = ch[3].polynomial, ch[3].shift_register, int(bool(ch[3].invert))
    threads = _host_threads()
    rc = lib().pm_host_decode_batch(jobs, n, threads)
    for j, ch in enumerate(chains):             # the registers move on even when a later job failed
        ch[3].shift_register = jobs[j].lfsr_state
    check(rc)
    counts = (ctypes.c_int64 * n)(*[jobs[j].pending for j in range(n)])
    handles = (ctypes.c_void_p * n)(*[jobs[j].codec for j in range(n)])
    block = np.empty(sum(counts), dtype=packet_dtype())
    check(lib().pm_codec_fetch_batch(handles, counts, n, block.ctypes.data_as(ctypes.c_void_p), threads))
    views, at = [], 0
    for c in counts:
        views.append(block[at:at + c])
        at += c
    return views


_POOL = None
_USE_SWEEP = True          # pm_afsk_sweep_signs for gain sweeps (tests switch it off to compare against the exact group path)
_GROUP_RUN_QUICK = [__import__("os").environ.get("PYMODEM_AMD_GROUP_RUN_QUICK", "0") != "0"]     # measured: submit 0.16 ms instead of 0.4, the step unchanged
_USE_GROUP_NATIVE = True   # band-pass + all sweeps of a group in one native call (tests switch it off to compare with the separate calls)


def _pool():
    global _POOL
    if _POOL is None:
        import os
        _POOL = ThreadPoolExecutor(max_workers=max(2, min(int(os.environ.get("PYMODEM_AMD_HOST_THREADS", "16")), (os.cpu_count() or 4) // 2)))
    return _POOL


def process_chains_table(chains, input_audio, chain_ids=None, names=None):
    """Same work as process_chains_device, but the packets come back as pm_packet rows keyed by (global) chain index:
    {chain id: rows}.  Feed them to PacketTable / dist.gather_rows; no per-packet Python objects are made."""
    rows = process_chains_device(chains, input_audio, _rows=True)
    ids = list(range(len(chains))) if chain_ids is None else list(chain_ids)
    return dict(zip(ids, rows))


def process_chains_split(chains, input_audio):
    """The GPU half of process_chains_table, returning a zero-argument callable for the host half: demod + slice run (and
    finish) now; calling the result runs LFSR + codec for every chain and returns their pm_packet rows.  Lets the host half
    of one recording overlap the GPU half of the next (bench.py --overlap)."""
    sliced = process_chains_device(chains, input_audio, _sliced_only=True)

    return lambda: _host_rows(chains, sliced)


class _BatchFetch:
    """The deferred device-to-host copy of one slicer batch, done once by the first taker."""

    def __init__(self, fetch, ctx, lock):
        import threading
        self._fetch, self._ctx, self._copy_lock = fetch, ctx, lock
        self._lock = threading.Lock()
        self._out = None

    def get(self, lo, hi):
        with self._lock:
            if self._out is None:
                with self._copy_lock:
                    self._out = self._fetch(self._ctx)
                self._fetch = None
        return self._out[lo:hi]


class RecordingPipeline:
    """Successive recordings through chain groups with the stages of the path overlapped, each on its own resource:

      demod   FIR / correlator / loop kernels (vector-f64 ALU and HBM)                 default stream, caller's thread
      slice   lockstep walkers (pm_slice_batch): a batch of up to four recordings costs   `slice_workers` high-priority side streams,
              about what one costs (dependent-latency bound), so a worker waits for       one thread each
              four finished demods before it starts one (one worker collects at a time); the batch's bytes and address
              steps come back in compact form on the same stream (pm_slice_compact)
      host    LFSR + codec (native, GIL released)                                       three to nine threads (recordings), sized to the chain
                                                                                        group; chains on library threads
      finish  the caller's `finish(rows per chain)`: the packet exchange                 one thread, submission order (collectives)
      post    the caller's `post(...)`: rank 0's payload copy, indexing and de-dup       three threads

    While recordings k-1..k-3 are being sliced, recording k+1 is demodulated and k-4 finished.  The only GPU buffers that cross
    stages are the sign bitmaps (one bit per sample), kept in sixteen rotating slots (the submitter runs ahead until they are
    all in flight); a GPU event, not a host wait, orders slicer after demod.  drain() waits for everything submitted, the executor
    stays usable (bench.py keeps one across warm-up and timed steps); close() ends its threads.  Results are identical to
    process_chains_table on each recording (tests/test_gpu_chains.py)."""

    def __init__(self, slice_workers=2, slice_group=4, slots=None):
        from collections import deque
        import os
        import queue
        import threading
        self._workers = max(1, int(slice_workers))
        # ONE demod stream.  (Rounds 1-2 could alternate recordings on two: it never gained anything measurable, showed an intermittent
        # GPU memory fault in round 2 that could not be reproduced into a cause, and is gone -- parameter, code path and test case.)
        self._group = max(1, min(int(os.environ.get("PYMODEM_AMD_SLICE_GROUP", slice_group)), 8))
        self._host_blocks_warm = False
        self._collect_lock = None if os.environ.get("PYMODEM_AMD_SLICE_COLLECT") == "free" else threading.Lock()
        self._fetch_inline = os.environ.get("PYMODEM_AMD_FETCH", "worker") != "copy"
        self._min_group = max(1, min(int(os.environ.get("PYMODEM_AMD_SLICE_MIN_GROUP", 4)), self._group))
        # LFSR + codec of several recordings at a time, each on up to one library thread per chain: about two dozen native threads in
        # all is where it stops paying (8-chain AFSK group: 3 recordings 1.21 ms per step, 5 recordings 1.38-1.41, medians of
        # interleaved runs; a 3-chain IL2P group at 4-5 ms per chain takes eight).  Made at the first submit, when the group is known.
        self._host = None
        self._finish = ThreadPoolExecutor(max_workers=1)
        self._post = ThreadPoolExecutor(max_workers=int(os.environ.get("PYMODEM_AMD_POST_THREADS", 3)))        # whatever follows the ordered step (rank 0's payload copy, indexing, de-dup)
        self._inflight = deque()
        self._tails = deque()                                 # the last stage's future of every recording not yet through
        self._failed = []
        self._n = 0
        self._upload = ThreadPoolExecutor(max_workers=1)      # host -> HBM copies of the NEXT recording, on a stream of their own
        self._uploads = 0
        self._upload_guard = {}                               # upload slot -> event after which its buffer may be overwritten
        self._upload_user = {}                                # upload slot -> "sliced" future of the recording that read it last
        self._upload_done = {}                                # upload slot -> event marking the end of its copy (re-used)
        # The slicer stage: a batch takes ~3 ms of a few hundred long-lived waves however many streams are in it (pm_slice_batch:
        # one walker per 32 k samples), so a worker takes EVERY recording whose demod has been submitted when it becomes free (up to
        # `slice_group`, 64 streams per batch): the batch size settles where the slicers keep up with the demod stream, with two or
        # three streams (HIP maps streams onto a handful of hardware queues; more slicer streams than that end up sharing a queue
        # with the demod stream and stall it).
        self._pending = queue.Queue()
        self._batch_seq = {}
        self._copy_ctx = Context.side(index=201, high_priority=False)     # a copy stream of its own (prefetch() uploads on 200)
        self._copy_lock = threading.Lock()
        self._slots = int(slots or os.environ.get("PYMODEM_AMD_BITMAP_SLOTS", 0) or 16)
        self._events = [None] * self._slots                   # bitmaps: one set per recording between "demod submitted" and "sliced"
        self.stage_seconds = {"demod": 0.0, "slice": 0.0, "host": 0.0, "finish": 0.0}   # busy time per stage, summed over recordings
        self.slice_batches = 0
        self.slice_log = []
        self.timeline = []                                    # per recording: host clock at the stage boundaries (diagnostics)
        self._watch = None
        if os.environ.get("PYMODEM_AMD_PIPE_WATCH"):          # diagnostic: stamp the moment each recording's demod finished on the GPU
            self._watch = deque()

            def watch():
                import time
                while self._watch is not None:
                    while self._watch and Context.event_done(self._watch[0][1]):
                        self._watch.popleft()[0]["demod_done"] = time.perf_counter()
                    time.sleep(0.0001)
            threading.Thread(target=watch, daemon=True).start()
        self._slice_threads = [threading.Thread(target=self._slice_loop, args=(Context.side(index=i),), daemon=True) for i in range(self._workers)]
        for th in self._slice_threads:
            th.start()

    def _collect_batch(self, taken=None):
        """-> the next batch of recordings (None: the pipeline is closing).  `taken` receives every item as soon as it is off the
        queue, so that a caller can fail them if one of the waits below raises."""
        import queue
        import time
        item = self._pending.get()
        if item is None:
            self._pending.put(None)                            # one marker ends every worker
            return None
        items = taken if taken is not None else []
        items.append(item)
        # demod runs in submission order: wait (on the host, holding nothing) for this recording's bitmaps, then take along
        # every later recording whose bitmaps are complete as well
        Context.event_sync(item[2])
        item[6]["ready"] = time.perf_counter()
        while len(items) < self._group:
            try:
                nxt = self._pending.get_nowait()
            except queue.Empty:
                break
            if nxt is None:
                self._pending.put(None)
                break
            items.append(nxt)
            if not Context.event_done(nxt[2]):
                if len(items) <= self._min_group:              # its demod is already queued on the GPU (<= 1 ms): a batch of three costs
                    Context.event_sync(nxt[2])                 # what a batch of one costs, so a short wait here saves whole batches
                else:
                    items.pop()
                    with self._pending.mutex:                  # not ready yet: back to the FRONT of the queue
                        self._pending.queue.appendleft(nxt)
                        self._pending.not_empty.notify()
                    break
        return items

    def _slice_loop(self, side):
        import os
        import queue
        import time
        while True:
            # ONE worker at a time puts a batch together, so that a batch is CONSECUTIVE recordings and starts when its third demod is
            # done; three workers collecting at once took every third recording each, and all three batches started only when the
            # ninth demod was done (seen in the per-recording timeline).  PYMODEM_AMD_SLICE_COLLECT=free: the old behaviour.
            taken = []
            try:
                if self._collect_lock is not None:
                    with self._collect_lock:
                        items = self._collect_batch(taken)
                else:
                    items = self._collect_batch(taken)
            except BaseException as e:                         # noqa: BLE001
                # an event wait failed (HIP error, a demod that faulted): the recordings already taken off the queue get the error --
                # their host stages would wait for ever otherwise -- and the worker lives on
                for it in taken:
                    if not it[3].done():
                        it[3].set_exception(e)
                continue
            if items is None:
                return
            t = time.perf_counter()
            try:
                slicers, bitmaps = [], []
                for bi, (chains, bm, ready, _, sweeps, audio, rec) in enumerate(items):
                    rec["slice0"] = t
                    if sweeps:                                 # finished (their event has): overflowed ones are redone exactly, here
                        resolve_sweeps(chains, bm, sweeps, audio, side, tag=bi)
                    slicers += [ch[2] for ch in chains]
                    bitmaps += bm
                # (copied to the host by this worker before its next batch: ONE output block per worker will do -- keyed by the first
                # recording's slot there would be sixteen of them, 1.9 GB per worker, allocated as the slots come round for the first time)
                t_a = time.perf_counter()
                fetch = slice_batch(slicers, bitmaps, side, defer=True, reserve=self._group / len(items), compact=True,
                                    out_tag=("pipeline-out",) if self._fetch_inline else None)
                t_b = time.perf_counter()
                if not getattr(side, "_slicer_block_reserved", False):
                    # the stream's work block (checkpoints, symbol bitmaps, lists: ~150 MB per recording) sized for a batch of four from
                    # the first batch on: growing it later is a free + malloc in the middle of the pipeline (10 ms measured)
                    side._slicer_block_reserved = True        # once per context (contexts outlive pipelines)
                    have = ctypes.c_size_t()
                    check(lib().pm_ctx_scratch(side.handle, 0, ctypes.byref(have)))
                    check(lib().pm_ctx_scratch(side.handle, int(have.value * self._group / len(items)), None))
                    # ... and the page-locked host blocks the compact output is copied into: a block stays referenced until the host
                    # stages of all its recordings are through, and with sixteen recordings in flight that can be more than two batches
                    # per worker; one made in front of a copy costs 8-20 ms there (touch, pin, under load)
                    rooms = [f.room for f in getattr(fetch, "fetchers", []) if getattr(f, "room", 0)]
                    if rooms and self._fetch_inline and not self._host_blocks_warm:
                        from .device import DeviceBuffer
                        self._host_blocks_warm = True         # the pool is the process's: once, by whichever worker has the first batch
                        DeviceBuffer.prewarm_host_blocks(side, max(rooms), 4 * self._workers)
                # The slicers' bytes and addresses are still in device memory: whichever host-stage thread needs them first copies
                # the whole batch over on the copy stream (this worker's stream is already slicing the next batch).
                if self._fetch_inline:
                    # the compact output of a batch is a few megabytes: copied here, on the stream that made it and is idle at this
                    # point, it is on the host a third of a millisecond later; on the copy stream it waits for whatever kernel the
                    # hardware queue that stream shares is running (1-2 ms measured)
                    t_c = time.perf_counter()
                    done = fetch(side)
                    fetch = lambda _ctx, _done=done: _done
                    if time.perf_counter() - t > 0.008 and os.environ.get("BENCH_TIMELINE"):
                        import sys
                        print("[slow batch] %d recordings: before slice_batch %.2f, slice_batch %.2f, between %.2f, fetch %.2f ms" % (
                            len(items), (t_a - t) * 1e3, (t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (time.perf_counter() - t_c) * 1e3), file=sys.stderr)
                shared = _BatchFetch(fetch, self._copy_ctx, self._copy_lock)
                at = 0
                for chains, _, _, fut, _, _, rec in items:
                    rec["slice1"] = time.perf_counter()
                    fut.set_result((shared, at, at + len(chains)))
                    at += len(chains)
            except BaseException as e:                         # noqa: BLE001
                for it in items:
                    if not it[3].done():
                        it[3].set_exception(e)
            dt = time.perf_counter() - t
            self.stage_seconds["slice"] += dt
            self.slice_batches += 1
            self.slice_log.append((len(items), t, dt))

    def prefetch(self, host_audio):
        """Start copying a recording (host int16 / float64 array) into HBM on a copy stream; returns a handle for submit().  Call
        it one recording ahead and the copy runs while the previous recording is demodulated (as many rotating device buffers as
        bitmap slots; a buffer is overwritten only after the demod that read it has finished, which the copy stream waits for on
        the GPU, and after its recording has left the slicer stage)."""
        a = np.asarray(host_audio)
        a = np.ascontiguousarray(a if a.dtype == np.int16 else a.astype(np.float64))
        k = self._uploads % self._slots
        self._uploads += 1
        guard = self._upload_guard.get(k)
        user = self._upload_user.pop(k, None)
        cctx = Context.side(index=200, high_priority=False)

        def copy():
            # The buffer's last reader may be the slicer stage: a certified sweep whose list overflowed is demodulated again from the
            # audio there (resolve_sweeps), long after the demod stream's event.  As many upload buffers as bitmap slots, and the
            # recording that used this one `slots` uploads ago has left the slicer stage (submit() waits for exactly that before it
            # hands a slot out again), so this never waits in practice.
            if user is not None:
                try:
                    user.result()
                except Exception:                              # noqa: BLE001  (its own future reports it)
                    pass
            if guard is not None:
                cctx.wait_event(guard)
            buf = cctx.scratch(("upload", k), a.size, a.dtype)
            check(lib().pm_h2d(cctx.handle, buf.ptr, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
            self._upload_done[k] = done = cctx.record_event(self._upload_done.get(k))
            return buf, done, k
        return self._upload.submit(copy)

    def submit(self, chains, input_audio, finish=None, post=None, prepare=None, chain_ids=None, unordered=False):
        """Start one recording; returns a Future of post(finish(rows per chain)) (either may be None = identity).  `finish` calls
        run one at a time in submission order (the place for collectives); `post` calls run in parallel with later recordings.
        If finish returns a future itself, post receives its result (and waits for it: see flush_finish).  `prepare(rows)`, if
        given, runs at the end of the host stage (several recordings at a time) and its result is what finish receives.  `chain_ids`:
        each chain's place in the config (its key in the packet table); the codecs then stamp their packets with it as they write them.
        `unordered`: finish does not depend on the recordings' order (no collective behind it): finish and post run at the end of the
        host stage, on its thread."""
        import time
        acc = self.stage_seconds
        self._forget_finished()
        slots = self._slots
        slot = self._n % slots
        self._n += 1
        while len(self._inflight) >= slots - 1:               # the slicer that read this slot `slots` recordings ago is done
            self._inflight.popleft().result()
        t0 = time.perf_counter()
        import os
        if int(os.environ.get("PYMODEM_AMD_CU_SPLIT", "0")) > 0:  # demod on the CUs the slicer streams do not use
            dctx = Context.side(index=101, high_priority=False)
        else:
            dctx = Context.default()
        upload_slot = None
        if hasattr(input_audio, "result"):                    # a prefetch() handle: the demod stream waits for the copy on the GPU
            input_audio, copied, upload_slot = input_audio.result()
            dctx.wait_event(copied)
        st = {}
        bitmaps = process_chains_device(chains, input_audio, stages=st, _bitmaps_only=True, _slot=slot, _ctx=dctx)
        sweeps = st.get("sweeps") or []
        self._events[slot] = ready = dctx.record_event(self._events[slot])   # bitmaps complete at this point of the stream
        if upload_slot is not None:
            self._upload_guard[upload_slot] = dctx.record_event(self._upload_guard.get(upload_slot))   # its own event, re-used per slot
        acc["demod"] += time.perf_counter() - t0

        from concurrent.futures import Future
        f_sliced, f_fetched = Future(), Future()
        rec = {"submit0": t0, "submit1": time.perf_counter()}
        self.timeline.append(rec)
        if self._watch is not None:
            self._watch.append((rec, ready))
        if upload_slot is not None:
            self._upload_user[upload_slot] = f_fetched
        self._pending.put((chains, bitmaps, ready, f_sliced, sweeps, input_audio, rec))
        # the bitmap slot (and the slicers' output block keyed by it) is free again once the slicers' output is on the host
        self._inflight.append(f_fetched)

        def host_stage():
            try:
                shared, lo, hi = f_sliced.result()
                sliced = shared.get(lo, hi)
                rec["fetched"] = time.perf_counter()
            finally:
                f_fetched.set_result(None)
            t = rec["host0"] = time.perf_counter()
            rows = _host_rows(chains, sliced, chain_ids)
            if prepare is not None:                            # e.g. dist.Exchanger.prepare: packing for the wire, off the ordered thread
                rows = prepare(rows)
            rec["host1"] = time.perf_counter()
            acc["host"] += rec["host1"] - t
            if unordered:                                      # nothing needs the recordings' order: finish and post right here,
                x = rows if finish is None else finish(rows)   # two thread hand-overs (and their waits for the interpreter lock) less
                rec["finish1"] = time.perf_counter()
                acc["finish"] += rec["finish1"] - rec["host1"]
                if post is not None:
                    if hasattr(x, "result") and hasattr(x, "done"):
                        x = x.result()
                    x = post(x)
                    rec["post1"] = time.perf_counter()
                    acc["post"] = acc.get("post", 0.0) + rec["post1"] - rec["finish1"]
                return x
            return rows
        if self._host is None:
            import os
            # (one more when finish and post run on these threads too)
            n_host = int(os.environ.get("PYMODEM_AMD_HOST_STAGE_THREADS", 0)) or max(2, min(8, 24 // max(len(chains), 1))) + (1 if unordered else 0)
            self._host = ThreadPoolExecutor(max_workers=n_host)
        f_rows = self._host.submit(host_stage)
        if unordered:
            self._tails.append(f_rows)
            return f_rows

        def finish_stage():
            rows = f_rows.result()
            if finish is None:
                return rows
            t = time.perf_counter()
            out = finish(rows)
            rec["finish1"] = time.perf_counter()
            acc["finish"] += rec["finish1"] - t
            return out
        f_fin = self._finish.submit(finish_stage)
        if post is None:
            self._tails.append(f_fin)
            return f_fin

        def post_stage():
            x = f_fin.result()
            if hasattr(x, "result") and hasattr(x, "done"):    # finish handed out a future of its own (dist.Exchanger: resolved a step later)
                x = x.result()
            t = time.perf_counter()
            out = post(x)
            rec["post1"] = time.perf_counter()
            acc["post"] = acc.get("post", 0.0) + rec["post1"] - t
            return out
        f_post = self._post.submit(post_stage)
        self._tails.append(f_post)
        return f_post

    def drain(self):
        """Waits until every recording submitted so far has left its last stage (the pipeline stays usable); re-raises the first
        stage error."""
        while self._tails:
            self._tails.popleft().result()
        if self._failed:
            f, self._failed = self._failed[0], []
            f.result()

    def _forget_finished(self):
        # a finished recording's future holds its result (megabytes of packet rows): let go of it as soon as it is through, keeping
        # only the ones that failed for drain() to report
        while self._tails and self._tails[0].done():
            f = self._tails.popleft()
            if f.exception() is not None:
                self._failed.append(f)

    def reset_stats(self):
        for k in list(self.stage_seconds):
            self.stage_seconds[k] = 0.0
        self.slice_batches = 0
        self.slice_log = []
        self.timeline = []

    def flush_finish(self, fn):
        """Run fn() on the finish thread after every finish submitted so far (e.g. dist.Exchanger.flush, which resolves the last
        recording's exchange); returns its Future."""
        return self._finish.submit(fn)

    def close(self):
        """Waits for everything submitted."""
        self._upload.shutdown(wait=True)
        w, self._watch = self._watch, None
        for f in list(self._inflight):
            try:
                f.result()
            except Exception:                                  # noqa: BLE001  (surfaces through the stage futures)
                pass
        self._pending.put(None)
        for th in self._slice_threads:
            th.join()
        if self._host is not None:
            self._host.shutdown(wait=True)
        self._finish.shutdown(wait=True)
        self._post.shutdown(wait=True)


def _afsk_group_native(ctx, chains, planned, audio, group_key, front, bitmaps, sweeps):
    """Band-pass of the group's shared front end and every planned certified sweep on it in ONE native call (pm_afsk_group_run): what
    modem.front_end() + AFSKModem.sweep_signs() per sweep launch, without ten trips through the C boundary per recording -- the
    submitting thread of the pipelined executor waits for the interpreter lock after each one."""
    from ._native import AfskSweepDesc
    from .data_classes import SignBits
    lead = planned[0][1][0]
    lead.scratch_key = (group_key, "front", len(front))
    mb = len(lead.input_bpf)
    if audio.n < mb:
        raise ValueError(f"input of {audio.n} samples is shorter than the {mb}-tap filter input_bpf")
    nb = audio.n - mb + 1
    taps = lead._const("input_bpf", lead.input_bpf)
    bpf = ctx.scratch((lead._key(), "input_bpf"), nb, np.float64)           # the buffer front_end() would have used
    front[lead.front_end_key()] = bpf
    bound = getattr(lead, "_bpf_bound", None)
    if bound is None or bound[0] is not lead.input_bpf:
        bound = lead._bpf_bound = (lead.input_bpf, float(np.abs(lead.input_bpf).sum()) * 32768.0)
    descs = (AfskSweepDesc * len(planned))()
    keep, outs = [], []
    for j, (part, mods) in enumerate(planned):
        prep = AFSKModem._sweep_prepare(mods)
        mc, ml, g = len(mods[0].mark_correlator_i), len(mods[0].output_lpf), len(mods)
        if nb < mc + ml - 1:
            raise ValueError("input shorter than the correlators and the output filter")
        nout = nb - mc - ml + 2
        ptrs = (ctypes.c_void_p * g)()
        bits = []
        for i, md in enumerate(mods):
            md._context()
            b = ctx.scratch((md._own_key(), "signs", "output_lpf"), (nout + 63) // 64 + 1, np.uint64)
            bits.append(b)
            ptrs[i] = b.ptr.value
        k, d = prep["consts"], descs[j]
        d.d_mark_i, d.d_mark_q, d.d_unit_i, d.d_unit_q, d.d_space = (k[i].ptr.value for i in range(5))
        d.h_gains, d.groups, d.m = ctypes.addressof(prep["gains"]), g, mc
        d.d_lpf, d.ml, d.lpf_abs_sum = k[5].ptr.value, ml, prep["lpf_abs"]
        d.h_bits = ctypes.addressof(ptrs)
        d.h_tones = ctypes.addressof(prep["tones"]) if prep["tones"] is not None else None
        keep.append(ptrs)
        outs.append((part, bits, nout))
    tickets = (ctypes.c_int64 * len(planned))()
    # launches only (115 us): without dropping the interpreter lock when the pipelined executor asks for it -- getting it back
    # behind a dozen threads cost the submitting thread 0.5 ms per recording
    native = quick() if _GROUP_RUN_QUICK[0] else lib()
    check(native.pm_afsk_group_run(ctx.handle, audio.ptr, audio.n, taps.ptr, mb, bpf.ptr, bound[1], descs, len(planned), tickets))
    AFSKModem.sweeps_run += len(planned)
    for j, (part, bits, nout) in enumerate(outs):
        sweeps.append(((ctx, int(tickets[j])), list(part)))
        for k, b in zip(part, bits):
            bitmaps[k] = chains[k][2].sign_bitmaps(SignBits(b, None, nout))


def resolve_sweeps(chains, bitmaps, sweeps, audio, ctx, tag=0):
    """The certified gain sweeps of one recording have FINISHED: any whose list of uncertain samples overflowed (digital silence,
    audio far below the stated bound -- never on a signal) did not leave valid bitmaps; its chains are demodulated again with the
    exact kernels on `ctx` (BPF, correlators, low-pass + sign), which is what the in-call fallback of pm_afsk_sweep_signs would have
    done.  `tag`: distinguishes the recordings whose bitmaps must exist side by side (the recordings of one slicer batch).  Returns
    the number of sweeps redone."""
    redone = 0
    # one copy per producing context for all of the recording's sweeps (pm_afsk_sweep_results)
    over = {}
    by_ctx = {}
    for sweep, _ in sweeps:
        by_ctx.setdefault(id(sweep[0]), (sweep[0], []))[1].append(sweep[1])
    for sctx, tickets in by_ctx.values():
        t = (ctypes.c_int64 * len(tickets))(*tickets)
        n, cap = (ctypes.c_int64 * len(tickets))(), ctypes.c_int64()
        check(lib().pm_afsk_sweep_results(sctx.handle, t, len(tickets), ctx.handle, n, ctypes.byref(cap)))
        for tk, v in zip(tickets, n):
            over[(id(sctx), tk)] = v > cap.value
    for sweep, ks in sweeps:
        if not over[(id(sweep[0]), sweep[1])]:
            continue
        redone += 1
        for k in ks:
            # On a PRIVATE copy of the modem: this runs on a slicer worker's thread up to sixteen recordings after the demod, and the
            # caller may hand the same modem objects to every recording (bench.py does) -- the submitting thread can be inside
            # process_chains_device with them right now.  The copy shares the taps and the uploaded constants, nothing else;
            # context, work-buffer keys and FIR history are its own.
            import copy
            modem = copy.copy(chains[k][1])
            modem.use_context(ctx)
            modem.scratch_key = ("chain-group", "sweep-fallback")
            modem.own_key = ("chain-group", "sweep-fallback", "own", tag, k)
            modem._hist = None
            chains[k][2]._ctx = chains[k][2]._ctx or ctx
            sb = modem.demod_signs(audio)
            # the slicers of this batch read these bitmaps after the call returns: they must not share storage with the next fallback
            bitmaps[k] = chains[k][2].sign_bitmaps(sb)
    return redone


def _plan_sweeps(chains, afsk_groups):
    """-> [(chain indices, their modems)] per certified gain sweep: chains of one mark-filter group that differ in space_gain only,
    eight at most per sweep; a lone chain only if its templates are tones (the sliding sums pay for themselves there)."""
    planned = []
    for key, members in afsk_groups.items():
        sweep_sets = {}
        for k in members:
            sk = chains[k][1].sweep_key()
            if sk is not None:
                sweep_sets.setdefault(sk, []).append(k)
        for part_all in sweep_sets.values():
            for base in range(0, len(part_all), 8):
                part = part_all[base:base + 8]
                mods = [chains[k][1] for k in part]
                if len(part) < 2 and not (AFSKModem.sliding_sums and AFSKModem._sweep_prepare(mods)["tones"] is not None):
                    continue                 # one chain whose templates are not tones: the exact kernels are cheaper
                planned.append((part, mods))
    return planned


def process_chains_device(chains, input_audio, stages=None, _rows=False, _sliced_only=False, _bitmaps_only=False, _slot=0, _ctx=None):
    """[chain, ...] -> [packets of chain 0, packets of chain 1, ...] (config order), identical to running
    process_chain on each.  See the module docstring for what is shared and batched."""
    ctx = _ctx or Context.default()
    if _ctx is not None:
        for ch in chains:
            ch[1].use_context(ctx)
            ch[2]._ctx = ctx
    audio = input_audio
    if not isinstance(audio, DeviceBuffer):
        a = np.asarray(audio)
        audio = ctx.upload(a if a.dtype == np.int16 else np.ascontiguousarray(a, dtype=np.float64))
    n_chains = len(chains)
    bitmaps = [None] * n_chains
    group_key = "chain-group"          # work buffers are reused from one group run to the next
    for k, ch in enumerate(chains):    # stable per-chain keys: a new set of stage objects reuses the previous run's buffers
        ch[1].own_key = (group_key, "modem", _slot, k)      # the sign bitmaps: double-buffered by slot for RecordingPipeline
        ch[2].own_key = (group_key, "slicer", _slot, k)

    # ---- shared front ends ------------------------------------------------------------------------------------
    front = {}

    def shared_front(modem):
        key = modem.front_end_key()
        if key not in front:
            modem.scratch_key = (group_key, "front", len(front))
            front[key] = modem.front_end(audio)
        return front[key]

    # ---- MPSK groups: one carrier-loop launch per shared front end ----------------------------------------------
    mpsk_groups = {}
    for k, ch in enumerate(chains):
        if isinstance(ch[1], MPSKModem):
            mpsk_groups.setdefault(ch[1].front_end_key(), []).append(k)
    for gi, (key, members) in enumerate(mpsk_groups.items()):
        lead = chains[members[0]][1]
        real, imag = shared_front(lead)
        n = imag.n
        g = len(members)
        loops = (Loop * g)()
        for j, k in enumerate(members):
            ctypes.memmove(ctypes.byref(loops[j]), ctypes.byref(chains[k][1]._loop), ctypes.sizeof(Loop))
        i_mix = ctx.scratch((group_key, "i_mix", gi), n * g, np.float64)
        q_mix = ctx.scratch((group_key, "q_mix", gi), n * g, np.float64)
        check(lib().pm_mpsk_loop(ctx.handle, loops, g, lead._const("wavetable", lead.wavetable).ptr,
                                 lead._const("pd", lead.phase_error_table.reshape(-1), np.int32).ptr,
                                 real.ptr, imag.ptr, 0, n, i_mix.ptr, q_mix.ptr, n))
        for j, k in enumerate(members):
            modem = chains[k][1]
            ctypes.memmove(ctypes.byref(modem._loop), ctypes.byref(loops[j]), ctypes.sizeof(Loop))
            modem.scratch_key = (group_key, "mpsk_back")
            out = modem.back_end(i_mix.view(j * n, n), q_mix.view(j * n, n), signs=True)
            bitmaps[k] = chains[k][2].sign_bitmaps(out)

    # ---- BPSK Costas / AFSK PLL groups: same idea, one real input stream -----------------------------------------------
    loop_groups = {}
    for k, ch in enumerate(chains):
        if isinstance(ch[1], (BPSKModem, AFSKPLLModem)):
            loop_groups.setdefault(ch[1].front_end_key(), []).append(k)
    for gi, (key, members) in enumerate(loop_groups.items()):
        lead = chains[members[0]][1]
        x = shared_front(lead)
        n, g = x.n, len(members)
        loops = (Loop * g)()
        for j, k in enumerate(members):
            ctypes.memmove(ctypes.byref(loops[j]), ctypes.byref(chains[k][1]._loop), ctypes.sizeof(Loop))
        mix = ctx.scratch((group_key, "loop_out", gi), n * g, np.float64)
        check(getattr(lib(), lead.loop_entry)(ctx.handle, loops, g, lead._const("wavetable", lead.wavetable).ptr, x.ptr, 0, n, mix.ptr, n))
        for j, k in enumerate(members):
            modem = chains[k][1]
            ctypes.memmove(ctypes.byref(modem._loop), ctypes.byref(loops[j]), ctypes.sizeof(Loop))
            modem.scratch_key = (group_key, "loop_back")
            bitmaps[k] = chains[k][2].sign_bitmaps(modem.back_end(mix.view(j * n, n), signs=True))

    # ---- AFSK: correlator banks that share their mark filters run as one launch per group of up to 8, the rest one by one;
    # then the output low-passes of all chains with equal taps run as one batched sign-only launch ----------------------------
    afsk_groups, corr = {}, {}
    sweeps = []                        # certified sweeps run: ((ctx, ticket), chain indices), resolved once they have finished
    int16_audio = audio.dtype == np.dtype(np.int16)
    for k, ch in enumerate(chains):
        if isinstance(ch[1], AFSKModem) and not ch[1].carry_history:      # a modem that carries FIR history runs on its own input
            ch[1].use_context(ctx)
            afsk_groups.setdefault(ch[1].mark_key(), []).append(k)
    gi = 0
    # a gain sweep (members differ in space_gain only, int16 audio so that |band-passed| <= sum|bpf| * 32768): certified sign
    # bitmaps from two correlator pairs and two low-passes for the whole sweep (pm_afsk_sweep_signs)
    planned = _plan_sweeps(chains, afsk_groups) if int16_audio and _USE_SWEEP else []    # (chain indices, their modems) per certified sweep
    if planned:
        fe = {mods[0].front_end_key() for _, mods in planned}
        if _USE_GROUP_NATIVE and len(fe) == 1 and next(iter(fe)) not in front:
            # the usual case (every sweep of the group on ONE band-passed stream): band-pass and all sweeps in one native call
            _afsk_group_native(ctx, chains, planned, audio, group_key, front, bitmaps, sweeps)
        else:                          # the overflow fallback of these sweeps is ours (resolve_sweeps), not three gated launches each
            check(lib().pm_afsk_sweep_mode(ctx.handle, 1))
            try:
                for part, mods in planned:
                    bpf = shared_front(mods[0])
                    got = AFSKModem.sweep_signs(mods, bpf, float(np.abs(mods[0].input_bpf).sum()) * 32768.0)
                    sweeps.append((got[0].sweep, list(part)))
                    for k, sb in zip(part, got):
                        bitmaps[k] = chains[k][2].sign_bitmaps(sb)
            finally:
                check(lib().pm_afsk_sweep_mode(ctx.handle, 0))
    for key, members in afsk_groups.items():
        members = [k for k in members if bitmaps[k] is None]
        for base in range(0, len(members), 8):
            part = members[base:base + 8]
            mods = [chains[k][1] for k in part]
            bpf = shared_front(mods[0])
            if len(part) >= 2:
                streams = AFSKModem.correlate_group(mods, bpf, (group_key, "afsk_corr_group", gi))
                gi += 1
            else:
                streams = [mods[0].correlate(bpf, (group_key, "afsk_corr_single", part[0]))]
            for k, c in zip(part, streams):
                corr[k] = c
    lpf_groups = {}
    for k in corr:
        lpf_groups.setdefault(chains[k][1].output_lpf.tobytes(), []).append(k)
    for members in lpf_groups.values():
        for k, sb in zip(members, AFSKModem.lpf_signs_batch([chains[k][1] for k in members], [corr[k] for k in members])):
            bitmaps[k] = chains[k][2].sign_bitmaps(sb)

    # ---- everything else, chain by chain (work buffers shared across the group); FSK modems that are the same filter (equal taps
    # and polarity: the chains of fsk_9600.json differ in stream and codec only) share ONE sign bitmap --------------------------------
    from .modems import FSKModem
    fsk_done = {}
    for k, ch in enumerate(chains):
        if bitmaps[k] is not None:
            continue
        modem = ch[1]
        modem.scratch_key = (group_key, type(modem).__name__)
        if isinstance(modem, FSKModem) and not modem.carry_history:
            key = modem.front_end_key()
            if key not in fsk_done:
                fsk_done[key] = modem.demod_signs(audio)
            bitmaps[k] = ch[2].sign_bitmaps(fsk_done[key])
            continue
        bitmaps[k] = ch[2].sign_bitmaps(modem.demod_signs(audio))

    # ---- all slicers in one batch, host stages in parallel ---------------------------------------------------------
    if _bitmaps_only:
        if stages is not None:
            stages["sweeps"] = sweeps
        return bitmaps
    if sweeps:
        ctx.sync()                     # the sweeps have finished: look at their counters
        resolve_sweeps(chains, bitmaps, sweeps, audio, ctx)
    sliced = slice_batch([ch[2] for ch in chains], bitmaps)
    if _sliced_only:
        return sliced
    if _rows:
        packets = _host_rows(chains, sliced)
    else:
        packets = [f.result() for f in [_pool().submit(_host_stages, ch, sl) for ch, sl in zip(chains, sliced)]]
    if stages is not None:
        stages["sliced"] = sliced
    return packets


class _PipeRows:
    """One finished recording's packet rows inside the library (pm_pipe_wait): numpy arrays made from it keep it alive, and the
    memory goes back (pm_pipe_release) when the last of them is gone."""

    def __init__(self, pipe, ticket, ptr, nbytes):
        self._pipe, self._ticket = pipe, ticket
        # read-only: the rows are the library's (a pooled block it keeps zero wherever no packet has written); a consumer that wants to
        # change a row copies it
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, True), "version": 3}
        pipe._live += 1

    def __del__(self):
        try:
            pipe = self._pipe
            if pipe is not None and pipe._h is not None:
                lib().pm_pipe_release(pipe._h, self._ticket)
                pipe._live -= 1
                if pipe._closed and pipe._live == 0:
                    pipe._destroy()
        except Exception:                                      # noqa: BLE001  (interpreter shutdown: the library may be gone)
            pass


class NativePipeline:
    """RecordingPipeline for an AFSK chain group with every stage inside the library (pm_pipe_*, csrc/pm_pipe.hip): submit() is one
    native call that launches the recording's demod and returns; slicer batches, LFSR + codec and the cross-chain de-dup run on the
    library's own threads; table() hands back the finished recording as a PacketTable (correlate() already done).  The chains are
    the TEMPLATE of every recording: each recording is decoded by fresh slicer / stream / codec states, as chain_builder.py makes
    them per run of the reference (pymodem.py:140-163).

    Accepts what the certified-sweep path of process_chains_device accepts -- AFSK modems on one shared band-pass whose every chain
    belongs to a gain sweep (or is a lone tone-template chain), binary slicers, LFSR streams, native codecs, int16 audio -- and
    raises ValueError for anything else (use RecordingPipeline there)."""

    def __init__(self, chains, max_samples, address_distance, ctx=None, names=None, chain_ids=None, slots=0, slice_workers=0, slice_group=0,
                 host_threads=0, decode_threads=0, slice_min_group=0, demod_streams=0, keep_slices=False):
        import os
        demod_streams = demod_streams or int(os.environ.get("PYMODEM_AMD_PIPE_DEMOD_STREAMS", 0))
        slots = slots or int(os.environ.get("PYMODEM_AMD_PIPE_SLOTS", 0))              # tuning knobs
        slice_workers = int(os.environ.get("PYMODEM_AMD_PIPE_WORKERS", 0)) or slice_workers
        slice_group = slice_group or int(os.environ.get("PYMODEM_AMD_PIPE_GROUP", 0))
        slice_min_group = slice_min_group or int(os.environ.get("PYMODEM_AMD_PIPE_MIN_GROUP", 0))
        host_threads = host_threads or int(os.environ.get("PYMODEM_AMD_PIPE_HOST_THREADS", 0))
        from ._native import AfskSweepDesc, PipeChain, PipeDesc, PipeFir
        from .codecs import _NativeCodec
        from .modems import FSKModem
        from .lfsr import LFSR
        from .slicer import BinarySlicer
        self._h = None
        self._closed, self._live = False, 0
        ctx = self._ctx = ctx or Context.default()
        n = len(chains)
        ids = list(range(n)) if chain_ids is None else [int(c) for c in chain_ids]
        self.names = list(names) if names is not None else [ch[0] for ch in chains]
        if not n or not all(isinstance(ch[1], (AFSKModem, FSKModem)) and not ch[1].carry_history and isinstance(ch[2], BinarySlicer)
                            and isinstance(ch[3], LFSR) and isinstance(ch[4], _NativeCodec) for ch in chains):
            raise ValueError("NativePipeline: AFSK / FSK modem + binary slicer + LFSR + AX25/IL2P codec chains only")
        groups, fsk_groups = {}, {}
        for k, ch in enumerate(chains):
            ch[1].use_context(ctx)
            if isinstance(ch[1], FSKModem):                   # one sign FIR per distinct modem (fsk.py:149-159), its chains slice that bitmap
                fsk_groups.setdefault(ch[1].front_end_key(), []).append(k)
            else:
                groups.setdefault(ch[1].mark_key(), []).append(k)
        planned = _plan_sweeps(chains, groups) if groups else []
        covered = sorted([k for part, _ in planned for k in part] + [k for part in fsk_groups.values() for k in part])
        if covered != list(range(n)) or len({mods[0].front_end_key() for _, mods in planned}) > 1:
            raise ValueError("NativePipeline: every AFSK chain must belong to a certified sweep on one shared band-pass")
        lead = planned[0][1][0] if planned else None
        taps = lead._const("input_bpf", lead.input_bpf) if planned else None
        descs = (AfskSweepDesc * max(len(planned), 1))()
        pchains = (PipeChain * n)()
        self._keep = [taps, chains]

        def stages(pc, ch, c):
            pc.slicer = ch[2]._params()
            pc.lfsr_poly, pc.lfsr_invert = int(ch[3].polynomial), int(bool(ch[3].invert))
            pc.codec_kind, pc.crc, pc.disable_rs = int(ch[4]._kind), int(ch[4].collect_trailing_crc), int(ch[4].disable_rs)
            pc.min_dist, pc.sync_tol, pc.source_decoder = int(ch[4].min_distance), int(ch[4].sync_tolerance), ids[c]
        firs = (PipeFir * max(len(fsk_groups), 1))()
        for f, part in enumerate(fsk_groups.values()):
            md = chains[part[0]][1]
            ft = md._const("input_lpf", md.input_lpf)
            self._keep.append(ft)
            firs[f].d_taps, firs[f].m, firs[f].flags = ft.ptr.value, len(md.input_lpf), 1 if md.invert else 0
            for c in part:
                pchains[c].sweep, pchains[c].slot = -(f + 1), 0
                stages(pchains[c], chains[c], c)
        for j, (part, mods) in enumerate(planned):
            prep = AFSKModem._sweep_prepare(mods)
            k, d = prep["consts"], descs[j]
            d.d_mark_i, d.d_mark_q, d.d_unit_i, d.d_unit_q, d.d_space = (k[i].ptr.value for i in range(5))
            d.h_gains, d.groups, d.m = ctypes.addressof(prep["gains"]), len(mods), len(mods[0].mark_correlator_i)
            d.d_lpf, d.ml, d.lpf_abs_sum = k[5].ptr.value, len(mods[0].output_lpf), prep["lpf_abs"]
            d.h_bits = None
            d.h_tones = ctypes.addressof(prep["tones"]) if prep["tones"] is not None else None
            self._keep.append(prep)
            for place, c in enumerate(part):
                pchains[c].sweep, pchains[c].slot = j, place
                stages(pchains[c], chains[c], c)
        ctx.sync()                                          # the constants are in place before another stream reads them
        desc = PipeDesc()
        if planned:
            desc.d_bpf, desc.mb, desc.nsweeps = taps.ptr.value, len(lead.input_bpf), len(planned)
            desc.x_bound = float(np.abs(lead.input_bpf).sum()) * 32768.0
        desc.sweeps, desc.chains, desc.nchains = descs, pchains, n
        desc.firs, desc.nfirs = firs, len(fsk_groups)
        desc.slots, desc.slice_workers, desc.slice_group, desc.slice_min_group = int(slots), int(slice_workers), int(slice_group), int(slice_min_group)
        desc.host_threads, desc.decode_threads, desc.demod_streams = int(host_threads), int(decode_threads), int(demod_streams)
        desc.address_distance, desc.max_samples = float(address_distance), int(max_samples)
        desc.keep_slices = int(bool(keep_slices))
        h = ctypes.c_void_p()
        check(lib().pm_pipe_create(ctx.handle, ctypes.byref(desc), ctypes.byref(h)))
        self._h = h
        self.nchains = n
        self.done_at_ms = {}                                # ticket -> when it left the last stage (host clock since the pipeline was made)
        self.slots = int(lib().pm_pipe_slots(h))            # as pm_pipe_create settled it

    def prefetch(self, host_audio):
        """Start copying a recording (host int16 array) into HBM on a copy stream; returns a handle for submit().  Called one
        recording ahead, the copy runs while the previous recording is demodulated.  The device buffers rotate: two more than the
        pipeline has bitmap slots, so a buffer comes round again only after the recording that used it has left the slicer stage
        (submit() blocks on exactly that before it reuses a slot)."""
        from concurrent.futures import ThreadPoolExecutor
        a = np.ascontiguousarray(np.asarray(host_audio))
        if a.dtype != np.int16:
            raise ValueError("NativePipeline.prefetch: int16 audio")
        if getattr(self, "_up_pool", None) is None:
            self._up_pool, self._up_n, self._up_done = ThreadPoolExecutor(max_workers=1), 0, {}
        k = self._up_n % (self.slots + 2)
        self._up_n += 1
        cctx = Context.side(index=200, high_priority=False)

        def copy():
            buf = cctx.scratch(("native-pipe-upload", id(self), k), a.size, a.dtype)
            check(lib().pm_h2d(cctx.handle, buf.ptr, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
            self._up_done[k] = done = cctx.record_event(self._up_done.get(k))
            return buf.view(0, a.size), done
        return self._up_pool.submit(copy)

    def submit(self, audio):
        """Start one recording (int16 DeviceBuffer, resident until the recording has been sliced; or a prefetch() handle) -> ticket."""
        if hasattr(audio, "result"):
            audio, copied = audio.result()
            self._ctx.wait_event(copied)
        if not isinstance(audio, DeviceBuffer) or audio.dtype != np.dtype(np.int16):
            raise ValueError("NativePipeline.submit: an int16 DeviceBuffer")
        t = ctypes.c_int64(-1)
        rc = lib().pm_pipe_submit(self._h, audio.ptr, audio.n, ctypes.byref(t))
        try:
            check(rc)
        finally:
            if rc and t.value >= 0:                          # the launches failed: the ticket is finished with that error, nobody waits for it
                lib().pm_pipe_release(self._h, t.value)
        self._next_ticket = t.value + 1
        return t.value

    def submit_many(self, audios):
        """Resident recordings, in order, from ONE library call on a thread of its own (pm_pipe_submit_many: the interpreter lock is not
        needed between recordings) -> (first ticket, join).  Tickets first .. first + len(audios) - 1 may be waited for at once;
        join() returns when the last recording is submitted and raises what the submission raised.  No other submit meanwhile."""
        import threading
        audios = list(audios)
        for a in audios:
            if not isinstance(a, DeviceBuffer) or a.dtype != np.dtype(np.int16):
                raise ValueError("NativePipeline.submit_many: int16 DeviceBuffers")
        k = len(audios)
        announced = ctypes.c_int64(-1)
        check(lib().pm_pipe_promise(self._h, k, ctypes.byref(announced)))      # waits on the new tickets may start before the thread has
        first = announced.value
        self._next_ticket = first + k
        ptrs = (ctypes.c_void_p * k)(*[a.ptr.value for a in audios])
        ns = (ctypes.c_int64 * k)(*[a.n for a in audios])
        got, err = ctypes.c_int64(-1), []

        def run():
            try:
                check(lib().pm_pipe_submit_many(self._h, ptrs, ns, k, ctypes.byref(got)))
                if got.value != first:
                    raise RuntimeError(f"submit_many: tickets start at {got.value}, expected {first}")
            except BaseException as e:                        # noqa: BLE001
                err.append(e)
        th = threading.Thread(target=run)
        th.start()

        def join():
            th.join()
            del audios[:]
            if err:
                raise err[0]
        return first, join

    def _wait(self, ticket):
        from ._native import PipeResult
        res = PipeResult()
        check(lib().pm_pipe_wait(self._h, int(ticket), ctypes.byref(res)))
        return res

    def unique(self, ticket, release=True):
        """Waits for the recording -> (unique packets, packets of all chains); by default its rows are given back at once."""
        res = self._wait(ticket)
        out = (int(res.unique), int(res.rows))
        self.done_at_ms[int(ticket)] = res.done_at_ms
        if release:
            check(lib().pm_pipe_release(self._h, int(ticket)))
        return out

    def rows(self, ticket):
        """Waits for the recording -> [pm_packet rows of chain 0, chain 1, ...]: views of the library's memory (no copy), which goes
        back when the last of them is gone.  For pipelines that de-duplicate elsewhere (address_distance < 0: this rank holds a part
        of the config and its rows go to dist.Exchanger)."""
        from ._native import packet_dtype
        res = self._wait(ticket)
        dt = packet_dtype()
        counts = [res.h_counts[c] for c in range(self.nchains)]
        if res.rows:
            block = np.asarray(_PipeRows(self, int(ticket), res.h_rows, res.rows * dt.itemsize)).view(dt)
        else:
            block = np.zeros(0, dtype=dt)
            check(lib().pm_pipe_release(self._h, int(ticket)))
        out, at = [], 0
        for c in counts:
            out.append(block[at:at + c])
            at += c
        return out

    def table(self, ticket):
        """Waits for the recording -> PacketTable over the library's rows (no copy), correlate() done.  The rows go back to the
        library when the table and every array taken from it are gone."""
        from ._native import packet_dtype
        from .packet_meta import PacketTable
        res = self._wait(ticket)
        dt = packet_dtype()
        counts = [res.h_counts[c] for c in range(self.nchains)]
        k = int(res.unique)
        uniq = np.ctypeslib.as_array(ctypes.cast(res.h_unique_idx, ctypes.POINTER(ctypes.c_int64)), (max(k, 1),))[:k].copy()
        corr = np.ctypeslib.as_array(ctypes.cast(res.h_corr_decoders, ctypes.POINTER(ctypes.c_int32)), (max(int(res.rows), 1),)).copy()
        if res.rows:
            rows = np.asarray(_PipeRows(self, int(ticket), res.h_rows, res.rows * dt.itemsize)).view(dt)
        else:
            rows = np.zeros(0, dtype=dt)
            check(lib().pm_pipe_release(self._h, int(ticket)))
        table = PacketTable.from_array(rows, counts, self.names)
        table.unique_idx, table._corr = uniq, corr
        table._corr_ends = np.cumsum(rows["correlated_count"][table.unique_idx])
        table._unique_decoders = None
        table.latency_ms = {"demod_done": res.ms_to_demod_done, "sliced": res.ms_to_sliced, "done": res.ms_to_done}
        self.done_at_ms[int(ticket)] = res.done_at_ms
        return table

    def slices(self, ticket, chain):
        """A finished recording's bitstream for one chain (pipelines made with keep_slices=True): what chain[2].slice returned --
        bytes and stream addresses (slicer.py:59-107) -- and what chain[3].stream_unscramble_8bit made of the bytes (lfsr.py:22-52),
        as copies: (AddressedArray, uint8 array).  Call before the recording's rows are released."""
        from .data_classes import AddressedArray
        self._wait(ticket)
        d, a, q, k = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        check(lib().pm_pipe_slices(self._h, int(ticket), int(chain), ctypes.byref(d), ctypes.byref(a), ctypes.byref(q), ctypes.byref(k)))
        n = k.value
        if not n:
            return AddressedArray(np.zeros(0, np.uint8), np.zeros(0, np.int64)), np.zeros(0, np.uint8)
        data = np.ctypeslib.as_array(ctypes.cast(d, ctypes.POINTER(ctypes.c_uint8)), (n,)).copy()
        addr = np.ctypeslib.as_array(ctypes.cast(a, ctypes.POINTER(ctypes.c_int64)), (n,)).copy()
        plain = np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(ctypes.c_uint8)), (n,)).copy()
        return AddressedArray(data, addr), plain

    def bitmap(self, ticket, chain, nout):
        """A finished recording's sign bitmap for one chain as a bool array of nout entries: bit k = (demod output k >= 0), what the
        chain's slicer read (pipelines made with keep_slices=True, before `slots` further recordings have been submitted)."""
        self._wait(ticket)
        words = (int(nout) + 63) // 64
        buf = np.zeros(max(words, 1), dtype=np.uint64)
        check(lib().pm_pipe_bitmap(self._h, int(ticket), int(chain), buf.ctypes.data_as(ctypes.c_void_p), max(words, 1)))
        return np.unpackbits(buf.view(np.uint8), bitorder="little")[: int(nout)].astype(bool)

    def release(self, ticket):
        check(lib().pm_pipe_release(self._h, int(ticket)))

    def drain(self):
        check(lib().pm_pipe_drain(self._h))

    def stats(self):
        b, r = ctypes.c_int64(), ctypes.c_int64()
        s, h = ctypes.c_double(), ctypes.c_double()
        check(lib().pm_pipe_stats(self._h, ctypes.byref(b), ctypes.byref(r), ctypes.byref(s), ctypes.byref(h)))
        return {"slice_batches": b.value, "recordings": r.value, "slice_busy_ms": s.value, "host_busy_ms": h.value}

    def side_contexts(self):
        """The contexts the library owns -- the slicer workers' and the demod streams beyond the caller's: for Context.profile /
        profile_read / sync."""
        out = []
        for entry, first in ((lib().pm_pipe_side_ctx, 0), (lib().pm_pipe_demod_ctx, 1)):
            i = first
            while True:
                c = entry(self._h, i)
                if not c:
                    break
                out.append(Context.borrowed(c, self._ctx.device))
                i += 1
        return out

    def _destroy(self):
        h, self._h = self._h, None
        if h is not None:
            lib().pm_pipe_destroy(h)

    def close(self):
        """Waits for everything submitted; the library's side goes when the last table made from it has gone."""
        if self._h is None or self._closed:
            return
        if getattr(self, "_up_pool", None) is not None:
            self._up_pool.shutdown(wait=True)
            self._up_pool = None
        check(lib().pm_pipe_drain(self._h))
        self._closed = True
        if self._live == 0:
            self._destroy()

    def __del__(self):
        try:
            self.close()
        except Exception:                                      # noqa: BLE001
            pass


class NativeChain:
    """One demod_chain's modem + slicer behind the whole-chain C entry points (pm_chain_create / pm_chain_run): what a C or C++
    host would call.  Built from the same stage objects chain_builder makes, so tap design stays on the host in one place.
    `run(audio)` -> the slicer's AddressedArray; feed it to stream / codec as usual."""

    def __init__(self, modem, slicer, ctx=None):
        from . import _native as N
        from .modems import FSKModem
        self._ctx = ctx or Context.default()
        d = N.ChainDesc()
        keep = []

        def vec(x, dtype=np.float64):
            a = np.ascontiguousarray(x, dtype=dtype)
            keep.append(a)
            return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double if dtype == np.float64 else ctypes.c_int32)), len(a)
        if isinstance(modem, AFSKModem):
            d.modem = N.MODEM_AFSK
            d.flags = N.CHAIN_CARRY_HISTORY if modem.carry_history else 0
            d.input_fir, d.n_input_fir = vec(modem.input_bpf)
            d.mark_i, d.n_corr = vec(modem.mark_correlator_i)
            d.mark_q, d.space_i, d.space_q = vec(modem.mark_correlator_q)[0], vec(modem.space_correlator_i)[0], vec(modem.space_correlator_q)[0]
            d.output_fir, d.n_output_fir = vec(modem.output_lpf)
        elif isinstance(modem, FSKModem):
            d.modem, d.flags = N.MODEM_FSK, (N.CHAIN_INVERT if modem.invert else 0) | (N.CHAIN_CARRY_HISTORY if modem.carry_history else 0)
            d.input_fir, d.n_input_fir = vec(modem.input_lpf)
        else:
            d.modem = {BPSKModem: N.MODEM_BPSK, MPSKModem: N.MODEM_MPSK, AFSKPLLModem: N.MODEM_AFSK_PLL, QPSKModem: N.MODEM_QPSK}[type(modem)]
            d.flags = N.CHAIN_CARRY_HISTORY if modem.carry_history else 0
            d.input_fir, d.n_input_fir = vec(modem.input_bpf)
            a = modem.AGC
            d.use_agc, d.agc = 1, N.AGCParams(a.attack_rate, a.decay_rate, a.sustain_time, a.sample_rate, a.target_amplitude)
            ctypes.memmove(ctypes.byref(d.loop), modem._loop0, ctypes.sizeof(Loop))
            d.wavetable = vec(modem.wavetable)[0]
            if isinstance(modem, MPSKModem):
                d.hilbert, d.n_hilbert = vec(modem.hilbert_taps)
                d.hilbert_delay = modem.hilbert_delay
                d.pd_table = vec(modem.phase_error_table.reshape(-1), np.int32)[0]
                d.output_fir, d.n_output_fir = vec(modem.rrc_taps)
            elif isinstance(modem, (BPSKModem, QPSKModem)):
                d.output_fir, d.n_output_fir = vec(modem.rrc_taps)
            else:
                d.output_fir, d.n_output_fir = vec(modem.output_lpf)
        d.quadrature = int(isinstance(modem, (MPSKModem, QPSKModem)))
        d.slicer = slicer._params()
        self._h = ctypes.c_void_p()
        check(lib().pm_chain_create(self._ctx.handle, ctypes.byref(d), ctypes.byref(self._h)))
        self._bps = slicer.bits_per_symbol

    def run(self, audio):
        from .data_classes import AddressedArray
        if isinstance(audio, DeviceBuffer):
            assert audio.dtype == np.dtype(np.int16)
            ptr, n, on_dev = audio.ptr, audio.n, 1
        else:
            a = np.ascontiguousarray(audio, dtype=np.int16)
            ptr, n, on_dev = a.ctypes.data_as(ctypes.c_void_p), len(a), 0
        cap = n * self._bps // 8 + 8
        data, addr = np.empty(cap, np.uint8), np.empty(cap, np.int64)
        count = ctypes.c_int64()
        check(lib().pm_chain_run(self._h, ptr, n, on_dev, data.ctypes.data_as(ctypes.c_void_p), addr.ctypes.data_as(ctypes.c_void_p), cap,
                                 ctypes.byref(count)))
        return AddressedArray(data[:count.value].copy(), addr[:count.value].copy())

    def reset(self):
        check(lib().pm_chain_reset(self._h))

    def close(self):
        if self._h:
            lib().pm_chain_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
