"""Device context and buffers on top of the C ABI (pm_ctx_*, pm_malloc, pm_h2d, pm_d2h)."""
import ctypes
import os
import threading
import weakref

import numpy as np

from ._native import NativeError, check, lib, quick


class Context:
    """One HIP stream on one GPU.  Created lazily by the stage objects inside demod()/slice(), i.e. in the
    process that runs the chain (fork-safe, see pymodem.py:144-151)."""
    _default = {}
    _side = {}

    def __init__(self, device=0, high_priority=False, cu_mask=None):
        self._h = ctypes.c_void_p()
        self.device = device
        if cu_mask is not None:
            words = (ctypes.c_uint32 * len(cu_mask))(*cu_mask)
            check(lib().pm_ctx_create_cumask(device, words, len(cu_mask), ctypes.byref(self._h)))
        else:
            check(lib().pm_ctx_create_prio(device, int(bool(high_priority)), ctypes.byref(self._h)))

    @classmethod
    def borrowed(cls, handle, device=0):
        """A view of a context the library owns (a pipeline's slicer stream): for profile()/sync(), never destroyed from here."""
        self = cls.__new__(cls)
        self._h, self.device, self._borrowed = ctypes.c_void_p(handle), device, True
        return self

    @staticmethod
    def cu_split(device, per_xcd):
        """(slicer mask, demod mask): the first `per_xcd` CUs of each of the 8 XCDs for the slicers' streams, the rest for the FIR
        kernels.  Bit i of a CU mask belongs to XCD i mod 8 (pm_ctx_create_cumask)."""
        cus = lib().pm_device_cus(device) or 256
        nwords = (cus + 31) // 32
        low = [0] * nwords
        for i in range(min(per_xcd * 8, cus)):
            low[i // 32] |= 1 << (i % 32)
        high = [0] * nwords
        for i in range(cus):
            if not (low[i // 32] >> (i % 32)) & 1:
                high[i // 32] |= 1 << (i % 32)
        return low, high

    @classmethod
    def side(cls, device=None, index=0, high_priority=True):
        """Additional streams on the default context's GPU (one per process, device and index).  The pipelined executor runs
        slicers on high-priority ones while the next recordings' FIR/correlator kernels run on the default stream and on one more
        normal-priority stream."""
        import os
        main = cls.default(device)
        key = (os.getpid(), main.device, int(index))
        if key not in cls._side:
            if "PYMODEM_AMD_SIDE_PRIORITY" in os.environ:                     # tuning knobs
                high_priority = os.environ["PYMODEM_AMD_SIDE_PRIORITY"] != "0"
            split = int(os.environ.get("PYMODEM_AMD_CU_SPLIT", "0"))
            if split > 0 and index < 200:                                     # slicer streams (< 100) on their own CUs, demod streams (100..) on the rest
                low, high = cls.cu_split(main.device, split)
                cls._side[key] = cls(main.device, cu_mask=low if index < 100 else high)
            else:
                cls._side[key] = cls(main.device, high_priority=high_priority)
            # fewer, longer chunks while other streams share the CUs
            check(lib().pm_slicer_tune(cls._side[key]._h, int(os.environ.get("PYMODEM_AMD_SIDE_LANES", "16384"))))
        return cls._side[key]

    @classmethod
    def default(cls, device=None):
        import os
        if device is None:
            device = int(os.environ.get("PYMODEM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = lib().pm_device_count()
            if n > 0:
                device %= n
        key = (os.getpid(), device)
        if key not in cls._default:
            cls._default[key] = cls(device)
        return cls._default[key]

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(lib().pm_ctx_sync(self._h))

    def record_event(self, event=None):
        """Mark the point this context's stream has reached; returns the (re-usable) event handle."""
        ev = event if event is not None else ctypes.c_void_p()
        check(quick().pm_event_record(self._h, ctypes.byref(ev)))
        return ev

    def wait_event(self, event):
        """Work submitted to this context from now on waits (on the GPU) for `event`."""
        check(quick().pm_event_wait(self._h, event))

    @staticmethod
    def event_done(event):
        """True once everything submitted before the event's record has finished (no waiting)."""
        r = quick().pm_event_query(event)
        if r < 0:
            check(r)
        return r == 1

    @staticmethod
    def event_sync(event):
        check(lib().pm_event_sync(event))

    def sync_relaxed(self, poll_us=200):
        """Wait for everything submitted to this context so far without spinning on it (a run of the carrier-loop engine takes
        seconds)."""
        ev = self.__dict__.get("_relaxed_event")
        ev = self.__dict__["_relaxed_event"] = self.record_event(ev)
        check(lib().pm_event_sync_relaxed(ev, int(poll_us)))

    def empty(self, n, dtype):
        return DeviceBuffer(self, int(n), np.dtype(dtype))

    def upload(self, host):
        host = np.ascontiguousarray(host)
        buf = DeviceBuffer(self, host.size, host.dtype)
        check(lib().pm_h2d(self._h, buf.ptr, host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
        return buf

    def owner_key(self, obj):
        """A pool key tied to `obj`'s lifetime: its work buffers are released when the object is garbage collected
        (stage objects are single-use per recording; without this every new object would leave its buffers behind)."""
        import weakref
        key = ("obj", id(obj))
        owned = self.__dict__.setdefault("_owned", set())
        if key not in owned:
            owned.add(key)
            weakref.finalize(obj, self._release, key)
        return key

    def _release(self, key):
        self.__dict__.get("_pool_views", {}).clear()
        pool = self.__dict__.get("_pool", {})
        for tag in [t for t in pool if isinstance(t, tuple) and t and t[0] == key]:
            try:
                pool.pop(tag).free()
            except Exception:
                pass
        self.__dict__.get("_owned", set()).discard(key)

    def scratch(self, tag, n, dtype):
        """A persistent work buffer of at least n elements for `tag`; contents are undefined and the same
        storage is handed out again on the next call with the same tag (no hipMalloc in steady state)."""
        dtype = np.dtype(dtype)
        pool = self.__dict__.setdefault("_pool", {})
        need = max(int(n), 1) * dtype.itemsize
        raw = pool.get(tag)
        if raw is None or raw.n < need:
            if raw is not None:
                raw.free()
            want = need + need // 2 + 256                                    # big steps: a growth frees, and a free waits for the stream
            raw = self._arena_alloc(want) or DeviceBuffer(self, want, np.uint8)
            pool[tag] = raw
        views = self.__dict__.setdefault("_pool_views", {})
        vk = (tag, int(n), dtype)
        out = views.get(vk)
        if out is None or out._parent is not raw:           # the same borrowed view for the same request (ten per recording)
            out = DeviceBuffer(self, int(n), dtype, ptr=raw.ptr.value)
            out._parent = raw
            if len(views) > 4096:
                views.clear()
            views[vk] = out
        return out

    # Work buffers come out of large device blocks (2 GB; PYMODEM_AMD_ARENA_MB, 0 = a hipMalloc per buffer): the pipelined executor
    # touches a new set of per-slot buffers with each of its first sixteen recordings, and a hipMalloc in the middle of a full GPU
    # can hold every queue for milliseconds (seen as an 8 ms hole in the demod stream and two slicer batches at once).  A released
    # buffer (its owner object gone) is handed out again to the next request of the same size.
    def _arena_alloc(self, nbytes):
        mb = int(os.environ.get("PYMODEM_AMD_ARENA_MB", "2048"))
        nb = (int(nbytes) + 255) & ~255
        if mb <= 0 or nb > (mb << 20) // 4:
            return None
        with self.__dict__.setdefault("_arena_lock", threading.Lock()):
            free = self.__dict__.setdefault("_arena_free", {})
            if free.get(nb):
                return _ArenaBuffer(self, nb, free[nb].pop())
            chunks = self.__dict__.setdefault("_arena_chunks", [])
            if not chunks or chunks[-1][1] + nb > chunks[-1][0].n:
                chunks.append([DeviceBuffer(self, mb << 20, np.uint8), 0])
            chunk = chunks[-1]
            ptr = chunk[0].ptr.value + chunk[1]
            chunk[1] += nb
            return _ArenaBuffer(self, nb, ptr)

    def _arena_release(self, buf):
        if getattr(buf, "_gen", 0) != self.__dict__.get("_arena_gen", 0):
            return                                         # its block went back to the device meanwhile (drop_scratch)
        self.sync()                                        # like pm_free: what this stream still does with it finishes first
        with self.__dict__.setdefault("_arena_lock", threading.Lock()):
            self.__dict__.setdefault("_arena_free", {}).setdefault(buf.n, []).append(buf.ptr.value)

    def tune(self, **switches):
        """Diagnostic switches of this context's launchers (pm_ctx_tune; README.md lists them): ctx.tune(afsk_unfused=1).  A context
        reads the PM_* environment variables of the same names once, when it is made; no launch looks at the environment."""
        for name, value in switches.items():
            check(lib().pm_ctx_tune(self.handle, name.encode(), int(value)))
        return self

    def profile(self, on=True):
        check(lib().pm_prof_enable(self._h, int(bool(on))))

    def profile_read(self):
        """{kernel class: (total_ms, launches)} accumulated since profile(True)."""
        from ._native import KERNEL_CLASSES
        out = {}
        for k, name in enumerate(KERNEL_CLASSES):
            ms, n = ctypes.c_double(), ctypes.c_int64()
            check(lib().pm_prof_read(self._h, k, ctypes.byref(ms), ctypes.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def profile_intervals(self, name):
        """(start, end) in ms of every tracked launch of one kernel class since the device's reference event -> float64 array [n, 2].
        Intervals of several contexts of the device share the time base."""
        from ._native import KERNEL_CLASSES
        k = KERNEL_CLASSES.index(name)
        n = ctypes.c_int64()
        check(lib().pm_prof_intervals(self._h, k, None, None, 0, ctypes.byref(n)))
        out = np.empty((2, max(n.value, 1)), dtype=np.float64)
        check(lib().pm_prof_intervals(self._h, k, out[0].ctypes.data_as(ctypes.c_void_p), out[1].ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n)))
        return out[:, :n.value].T.copy()

    def profile_work(self):
        """{kernel class: (algorithmic HBM bytes, f64 flops)} of the launches profile_read() timed."""
        from ._native import KERNEL_CLASSES
        out = {}
        for k, name in enumerate(KERNEL_CLASSES):
            b, f = ctypes.c_double(), ctypes.c_double()
            check(lib().pm_prof_work(self._h, k, ctypes.byref(b), ctypes.byref(f)))
            out[name] = (b.value, f.value)
        return out

    def timer_start(self):
        check(lib().pm_timer_start(self._h))

    def timer_stop(self):
        ms = ctypes.c_float()
        check(lib().pm_timer_stop(self._h, ctypes.byref(ms)))
        return ms.value

    def drop_scratch(self):
        """Give every work buffer of this context (scratch() pool and the arena blocks behind it) back to the device.  For a process
        that moves on to a workload of a different shape: a run over thousands of streams leaves tens of gigabytes of slicer output
        blocks in the pool.  Nothing obtained from scratch() may be in use or be used afterwards."""
        self.sync()
        self.__dict__.get("_pool_views", {}).clear()
        pool = self.__dict__.get("_pool", {})
        for tag in list(pool):
            buf = pool.pop(tag)
            if not isinstance(buf, _ArenaBuffer):
                buf.free()
        with self.__dict__.setdefault("_arena_lock", threading.Lock()):
            for chunk in self.__dict__.pop("_arena_chunks", []):
                chunk[0].free()
            self.__dict__.pop("_arena_free", None)
            self._arena_gen = self.__dict__.get("_arena_gen", 0) + 1      # ranges handed out before this are nobody's any more

    def close(self):
        if getattr(self, "_borrowed", False):
            self._h = ctypes.c_void_p()
            return
        if self._h:
            for chunk in self.__dict__.pop("_arena_chunks", []):
                chunk[0].free()
            self.__dict__.pop("_arena_free", None)
            lib().pm_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()


_HOST_BLOCKS, _HOST_BLOCKS_LOCK, _HOST_BLOCKS_MAX = {False: [], True: []}, threading.Lock(), 32


def _host_block(nbytes, ctx=None):
    """A uint8 host array of at least nbytes that nobody else refers to (see DeviceBuffer.download).  With `ctx` a new block is
    page-locked (pm_host_pin) for as long as it lives: the pool's blocks take a copy per recording, and a copy into pageable
    memory goes through the runtime's bounce buffers at a third of the link's rate.  Page-locked and plain blocks are pooled
    apart: a stream of small plain blocks (the packet exchange's) must not push the large page-locked ones out -- making and
    releasing one of those costs 8 ms (fill, pin, unpin)."""
    import sys
    pinned = ctx is not None and os.environ.get("PYMODEM_AMD_PIN_HOST", "1") != "0"
    with _HOST_BLOCKS_LOCK:
        pool = _HOST_BLOCKS[pinned]
        for blk in pool:
            # references: the pool's list, the loop variable, getrefcount's argument
            if blk.nbytes >= nbytes and blk.nbytes <= 2 * nbytes + (1 << 20) and sys.getrefcount(blk) == 3:
                return blk
        blk = np.empty(max(nbytes, 1), dtype=np.uint8)
        if pinned:
            blk.fill(0)                                        # touch the pages here rather than inside the pin
            at = blk.ctypes.data
            if lib().pm_host_pin(ctx.handle, ctypes.c_void_p(at), blk.nbytes) == 0:
                weakref.finalize(blk, lib().pm_host_unpin, ctypes.c_void_p(at))   # runs before the array's memory is released
        if len(pool) >= _HOST_BLOCKS_MAX:
            for i, old in enumerate(pool):                     # drop an idle block rather than grow without bound
                if sys.getrefcount(old) == 3:
                    del pool[i]
                    break
        if len(pool) < _HOST_BLOCKS_MAX:
            pool.append(blk)
        return blk


class DeviceBuffer:
    """A typed device allocation (or a borrowed device pointer, e.g. torch.Tensor.data_ptr())."""

    def __init__(self, ctx, n, dtype, ptr=None):
        self.ctx, self.n, self.dtype = ctx, int(n), np.dtype(dtype)
        self._owned = ptr is None
        if ptr is None:
            p = ctypes.c_void_p()
            check(lib().pm_malloc(ctx.handle, self.n * self.dtype.itemsize, ctypes.byref(p)))
            self.ptr = p
        else:
            self.ptr = ctypes.c_void_p(int(ptr))

    @classmethod
    def borrow(cls, ctx, ptr, n, dtype):
        return cls(ctx, n, dtype, ptr=ptr)

    def view(self, offset, n):
        """Sub-range [offset, offset+n) as a borrowed buffer."""
        assert 0 <= offset and offset + n <= self.n
        out = DeviceBuffer(self.ctx, n, self.dtype, ptr=(self.ptr.value or 0) + offset * self.dtype.itemsize)
        out._parent = self
        return out

    @staticmethod
    def prewarm_host_blocks(ctx, nbytes, count):
        """Make sure the pool holds `count` idle page-locked blocks of nbytes: making one costs ~8 ms (touch, pin), which belongs in a
        warm-up, not in front of a batch's copy."""
        held = [_host_block(nbytes, ctx) for _ in range(count)]      # each call returns a different block while the others are held
        del held

    def download(self, n=None, recycle=False, ctx=None, room=None):
        """-> a host array of the first n elements.  recycle=True takes the memory from a small pool of host blocks that are handed
        out again once nothing refers to them any more (views keep their block alive): a fresh 10-40 MB allocation per call costs
        more in first-touch page faults than the copy itself."""
        n = self.n if n is None else int(n)
        # `room`: ask the pool for a block of that many elements (callers whose sizes vary name their largest: one size class)
        out = _host_block(max(n, int(room or 0)) * self.dtype.itemsize, ctx or self.ctx).view(self.dtype)[:n] if recycle else np.empty(n, dtype=self.dtype)
        if n:      # `ctx`: copy on that context's stream instead (the data must be complete: the caller synchronised its producer)
            check(lib().pm_d2h((ctx or self.ctx).handle, out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self._owned and self.ptr:
            lib().pm_free(self.ctx.handle, self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _ArenaBuffer(DeviceBuffer):
    """A range of one of a context's large blocks (Context._arena_alloc)."""

    def __init__(self, ctx, nbytes, ptr):
        super().__init__(ctx, nbytes, np.uint8, ptr=ptr)
        self._gen = ctx.__dict__.get("_arena_gen", 0)

    def free(self):
        if self.ptr:
            try:
                self.ctx._arena_release(self)
            finally:
                self.ptr = ctypes.c_void_p()


__all__ = ["Context", "DeviceBuffer", "NativeError"]
