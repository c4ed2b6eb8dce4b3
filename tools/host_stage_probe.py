#!/usr/bin/env python3
"""The host half of one recording of the headline config on ONE core, no GPU: a line bit stream like the bench buffer's (690 AX.25 UI
frames with noise bits between, NRZI) through pm_host_decode_batch (LFSR + AX.25 for eight chains), pm_codec_fetch_batch (rows with CRC
and header checks) and pm_correlate -- what pm_pipe's host workers do per recording -- and the same fetch into a row block that is kept
zero-tailed (pm_codec_fetch_batch_clean, what pm_pipe.hip uses since round 4)."""
import sys, time, ctypes
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymodem_amd import siggen
from pymodem_amd._native import lib, check, HostJob, packet_dtype
rng = np.random.default_rng(1)
# a line bit stream like the headline buffer's: ~690 AX.25 UI frames with noise bits between, NRZI-scrambled (poly 0x3, invert)
bits = []
for k in range(690):
    info = [int(c) for c in rng.integers(32, 127, int(rng.integers(20, 80)))]
    frame = siggen.ax25_ui_frame("CQ", f"N0CAL{k%10}", info)
    bits.append(np.array(siggen.ax25_hdlc_bits(frame), dtype=np.uint8))
    bits.append(rng.integers(0, 2, int(rng.integers(200, 400)), dtype=np.uint8))
b = np.concatenate(bits)
line = np.array(siggen.lfsr_scramble(b.tolist(), 0x3, True), dtype=np.uint8)
line = line[:len(line)//8*8]
data = np.packbits(line)
n = len(data)
print("bytes", n, "expected ~90000")
addr = np.arange(n, dtype=np.int64) * 320 + 100
L = lib()
def one(nch=8, threads=8, reps=20):
    t0 = time.perf_counter()
    for _ in range(reps):
        codecs = []
        jobs = (HostJob * nch)()
        for c in range(nch):
            h = ctypes.c_void_p()
            check(L.pm_codec_create(0, 1, 0, 0, 0, c, ctypes.byref(h)))
            codecs.append(h)
            jobs[c].codec = h; jobs[c].h_data = data.ctypes.data; jobs[c].h_addr = addr.ctypes.data; jobs[c].n = n
            jobs[c].lfsr_poly = 0x3; jobs[c].lfsr_invert = 1
        check(L.pm_host_decode_batch(jobs, nch, threads))
        pend = [jobs[c].pending for c in range(nch)]
        for h in codecs: L.pm_codec_destroy(h)
    dt = (time.perf_counter() - t0) / reps
    return dt, pend
for nch, th in ((1,1),(8,1),(8,8)):
    dt, pend = one(nch, th)
    print(f"{nch} chains, {th} threads: {dt*1e3:.3f} ms per recording, {dt/nch/n*1e9:.2f} ns per byte and chain (1 thread) packets {pend[:2]}")
# fetch (CRC + header checks) and correlate
import numpy as np
def full(nch=8, threads=1, reps=10):
    dt_fetch = dt_corr = dt_dec = 0.0
    for _ in range(reps):
        codecs = (ctypes.c_void_p * nch)()
        jobs = (HostJob * nch)()
        for c in range(nch):
            h = ctypes.c_void_p()
            check(L.pm_codec_create(0, 1, 0, 0, 0, c, ctypes.byref(h)))
            codecs[c] = h
            jobs[c].codec = h; jobs[c].h_data = data.ctypes.data; jobs[c].h_addr = addr.ctypes.data; jobs[c].n = n
            jobs[c].lfsr_poly = 0x3; jobs[c].lfsr_invert = 1
        t0 = time.perf_counter()
        check(L.pm_host_decode_batch(jobs, nch, threads))
        t1 = time.perf_counter()
        counts = (ctypes.c_int64 * nch)(*[jobs[c].pending for c in range(nch)])
        total = sum(counts)
        rows = np.empty(total, dtype=packet_dtype())
        check(L.pm_codec_fetch_batch(codecs, counts, nch, rows.ctypes.data_as(ctypes.c_void_p), threads))
        t2 = time.perf_counter()
        uniq = np.empty(total, np.int64); corr = np.empty(total, np.int32)
        k = L.pm_correlate(rows.ctypes.data_as(ctypes.c_void_p), counts, nch, 1200.0, uniq.ctypes.data_as(ctypes.c_void_p), corr.ctypes.data_as(ctypes.c_void_p), total)
        t3 = time.perf_counter()
        dt_dec += t1 - t0; dt_fetch += t2 - t1; dt_corr += t3 - t2
        for c in range(nch): L.pm_codec_destroy(codecs[c])
    print(f"decode {dt_dec/reps*1e3:.3f} ms, fetch {dt_fetch/reps*1e3:.3f} ms, correlate {dt_corr/reps*1e3:.3f} ms (unique {k} of {total}), rows {rows.nbytes/1e6:.1f} MB")
full()
# clean fetch into a zero block that is scrubbed and reused
def full_clean(nch=8, threads=1, reps=10):
    dt_fetch = 0.0
    Lc = L.pm_codec_fetch_batch_clean
    Lc.argtypes = L.pm_codec_fetch_batch.argtypes; Lc.restype = ctypes.c_int
    rows = np.zeros(8000, dtype=packet_dtype())
    for _ in range(reps):
        codecs = (ctypes.c_void_p * nch)()
        jobs = (HostJob * nch)()
        for c in range(nch):
            h = ctypes.c_void_p()
            check(L.pm_codec_create(0, 1, 0, 0, 0, c, ctypes.byref(h)))
            codecs[c] = h
            jobs[c].codec = h; jobs[c].h_data = data.ctypes.data; jobs[c].h_addr = addr.ctypes.data; jobs[c].n = n
            jobs[c].lfsr_poly = 0x3; jobs[c].lfsr_invert = 1
        check(L.pm_host_decode_batch(jobs, nch, threads))
        counts = (ctypes.c_int64 * nch)(*[jobs[c].pending for c in range(nch)])
        total = sum(counts)
        t1 = time.perf_counter()
        check(Lc(codecs, counts, nch, rows.ctypes.data_as(ctypes.c_void_p), threads))
        t2 = time.perf_counter()
        dt_fetch += t2 - t1
        rows[:total] = 0
        for c in range(nch): L.pm_codec_destroy(codecs[c])
    print(f"clean fetch {dt_fetch/reps*1e3:.3f} ms")
full_clean()
