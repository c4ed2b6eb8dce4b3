"""Modem stage objects: same constructor kwargs, `StringOptionsRetune`, `retune`, `tune`, `demod` and
attribute names as the reference's classes, so they drop in under chain_builder / chain_execute.
`demod()` runs entirely on the GPU through the C ABI (pm_fir_valid_*, pm_afsk_correlate, pm_agc_apply,
pm_costas_bpsk, pm_mpsk_loop, pm_pll_afsk); there is no CPU path.

Reference: AFSKModem afsk.py:13-167, FSKModem fsk.py:15-159, BPSKModem psk.py:20-195,
QPSKModem psk.py:197-476, MPSKModem psk.py:479-773, AFSKPLLModem afsk_pll.py:16-170.

Inputs: a host ndarray (int16 as scipy.io.wavfile returns it, or float64) or a DeviceBuffer already in HBM.
Outputs: host float64 ndarray / IQData by default (the reference's contract); with `device_out=True`
the result stays in HBM as DeviceBuffer / DeviceIQ for the slicer.
"""
import ctypes
import math

import numpy as np

from . import taps as T
from ._native import AGCParams, Loop, check, lib
from .data_classes import DeviceIQ, IQData, SignBits
from .device import Context, DeviceBuffer
from .string_ops import check_boolean


class _DeviceStage:
    """Lazy device state: nothing touches HIP until demod() runs (the reference forks after construction)."""
    _ctx = None
    scratch_key = None      # chains run one after another on one stream may share work buffers: same key, same storage
    own_key = None          # stable per-chain key given by the group executor (else tied to this object's lifetime)

    def _key(self):
        return self.scratch_key if self.scratch_key is not None else self._ctx.owner_key(self)

    def _own_key(self):
        """Key for buffers that must stay distinct per chain inside a group (the sign bitmaps the batched slicer reads)."""
        return self.own_key if self.own_key is not None else self._ctx.owner_key(self)

    def _context(self):
        if self._ctx is None:
            self._ctx = Context.default()
        if not hasattr(self, "_dev"):
            self._dev = {}
        return self._ctx

    def use_context(self, ctx):
        """Run this stage's next calls on `ctx` (its stream and work-buffer pool).  Uploaded constants stay valid: device memory is
        shared by all contexts of a GPU."""
        self._ctx = ctx
        if not hasattr(self, "_dev"):
            self._dev = {}

    def _const(self, name, host, dtype=np.float64):
        """Upload a constant array once per (name, contents)."""
        host = np.ascontiguousarray(host, dtype=dtype)
        key = (name, host.tobytes())
        hit = self._dev.get(name)
        if hit is None or hit[0] != key[1]:
            self._dev[name] = (key[1], self._ctx.upload(host))
            # once per constant: the copy is complete before anybody can use it -- stage objects move between contexts (the pipelined
            # executor's demod stream, a slicer worker's stream for the deferred sweep fallback), and a kernel on another stream is
            # not ordered behind a copy on this one
            self._ctx.sync()
        return self._dev[name][1]

    def _input(self, audio):
        """-> (DeviceBuffer, is_int16).  Host arrays are uploaded; anything not int16 is taken as float64,
        which is what numpy.convolve's type promotion does to it."""
        ctx = self._context()
        if isinstance(audio, DeviceBuffer):
            assert audio.dtype in (np.dtype(np.int16), np.dtype(np.float64))
            return audio, audio.dtype == np.dtype(np.int16)
        a = np.asarray(audio)
        if a.ndim != 1:
            raise ValueError("demod expects a 1-D mono sample array")
        if a.dtype == np.int16:
            return ctx.upload(a), True
        return ctx.upload(np.ascontiguousarray(a, dtype=np.float64)), False

    # ---- seamless chunked input, opt-in --------------------------------------------------------------
    # The reference's demod() is numpy.convolve(..., 'valid') per call (afsk.py:151-166, fsk.py:151): every call loses the first
    # M - 1 outputs of every filter, so a recording fed in pieces is NOT the recording fed at once.  With carry_history = True a
    # modem whose demod() is a cascade of FIRs and pointwise operations (AFSK correlator, FSK) keeps the last H = sum(M_i - 1)
    # input samples of a call and puts them in front of the next call's input: the cascade's 'valid' output over [tail | new] is
    # exactly the continuation of the previous call's output, so the pieces give the slicer the stream -- and, with the slicer's
    # own carried state, the bytes, addresses and packets -- of the single call on the concatenation.  The first call has no tail
    # and is the reference's call.  Default off: the reference's per-call behaviour.
    carry_history = False

    def _history_len(self):
        return 0

    def _with_history(self, x, is_i16):
        """x (DeviceBuffer of this call's samples) -> the buffer demod() runs on: [carried tail | x]; remembers the new tail."""
        if not self.carry_history:
            return x
        h = self._history_len()
        ctx = self._ctx
        tail = getattr(self, "_hist", None)
        if tail is not None and tail.dtype != x.dtype:
            raise ValueError("carry_history: the sample type changed between calls")
        nt = 0 if tail is None else len(tail)
        if nt:
            buf = ctx.scratch((self._own_key(), "hist_in"), nt + x.n, x.dtype)
            check(lib().pm_h2d(ctx.handle, buf.ptr, tail.ctypes.data_as(ctypes.c_void_p), tail.nbytes))
            check(lib().pm_d2d(ctx.handle, buf.ptr.value + tail.nbytes, x.ptr, x.n * x.dtype.itemsize))
        else:
            buf = x
        # the new tail: the last h samples of [tail | x]
        take = min(h, x.n)
        fresh = x.view(x.n - take, take).download() if take else np.zeros(0, x.dtype)
        keep = h - take
        self._hist = np.concatenate([tail[nt - min(keep, nt):], fresh]) if (nt and keep) else fresh
        return buf

    # ---- the same for modems with a carrier loop (round 3) --------------------------------------------------------------------------
    # BPSK / MPSK / QPSK / AFSK-PLL modems are FIR -> AGC -> (FIR) -> loop -> FIR.  With carry_history every FIR of the cascade keeps the
    # last M - 1 samples of ITS input stream and puts them in front of the next call's: band-pass over [audio tail | audio], Hilbert pair
    # over [AGC'd tail | new AGC'd samples], matched / output filter over [loop-output tail | new loop outputs].  AGC envelope and loop
    # registers are carried anyway (reset() clears them).  What stays per call is `normal = max(buffer)` of AGC.apply (agc.py:67): the
    # reference normalises by the maximum of the band-passed samples of each call, and so do we -- so pieces are NOT the whole here,
    # they are what the reference's primitives give when they are fed this way (tests/golden/make_goldens.py, psk_history).
    def _carry(self, name, x, h):
        """One FIR stage's input with carried history: x = this call's new samples of the stage's input stream (DeviceBuffer) ->
        [carried tail | x] as a DeviceBuffer, remembering the last h = taps - 1 samples of it."""
        if not self.carry_history or h <= 0:
            return x
        ctx = self._ctx
        tails = self.__dict__.setdefault("_tails", {})
        tail = tails.get(name)
        nt = 0 if tail is None else len(tail)
        if nt:
            buf = ctx.scratch((self._own_key(), "carry", name), nt + x.n, x.dtype)
            check(lib().pm_h2d(ctx.handle, buf.ptr, tail.ctypes.data_as(ctypes.c_void_p), tail.nbytes))
            if x.n:
                check(lib().pm_d2d(ctx.handle, buf.ptr.value + tail.nbytes, x.ptr, x.n * x.dtype.itemsize))
        else:
            buf = x
        take = min(h, x.n)
        fresh = x.view(x.n - take, take).download() if take else np.zeros(0, x.dtype)      # (waits for the stream: the tail is final)
        keep = h - take
        tails[name] = np.concatenate([tail[nt - min(keep, nt):], fresh]) if (nt and keep) else fresh
        return buf

    def _empty(self, tag):
        return self._ctx.scratch((self._key(), "empty", tag), 1, np.float64).view(0, 0)

    def _fir_carried(self, x, is_i16, taps_name, taps, tag=None, signs=False):
        """_fir / _fir_signs for a stage whose input may still be shorter than its filter (carry_history: the samples wait in the
        tail): an empty stream then."""
        if x.n < len(taps):
            if not self.carry_history:
                raise ValueError(f"input of {x.n} samples is shorter than the {len(taps)}-tap filter {taps_name}")
            if signs:
                return self._ctx.scratch((self._own_key(), "signs", tag or taps_name), 2, np.uint64), 0
            return self._empty(tag or taps_name)
        if signs:
            return self._fir_signs(x, is_i16, taps_name, taps, tag=tag)
        return self._fir(x, is_i16, taps_name, taps, tag=tag)

    def _starved(self, x, signs, device_out):
        """carry_history: [tail | x] is still shorter than the cascade needs for one output -- nothing comes out of this call (the
        samples wait in the tail).  None when there is enough."""
        if not self.carry_history or x.n > self._history_len():
            return None
        if signs:
            return SignBits(self._ctx.scratch((self._own_key(), "signs", "starved"), 2, np.uint64), None, 0)
        y = self._ctx.scratch((self._key(), "starved"), 1, np.float64).view(0, 0)
        return y if device_out else np.zeros(0, np.float64)

    def _fir(self, x, is_i16, taps_name, taps, flags=0, tag=None):
        ctx = self._ctx
        m = len(taps)
        if x.n < m:
            raise ValueError(f"input of {x.n} samples is shorter than the {m}-tap filter {taps_name}")
        y = ctx.scratch((self._key(), tag or taps_name), x.n - m + 1, np.float64)
        fn = lib().pm_fir_valid_i16 if is_i16 else lib().pm_fir_valid_f64
        check(fn(ctx.handle, x.ptr, x.n, self._const(taps_name, taps).ptr, m, y.ptr, flags))
        return y

    def _fir_signs(self, x, is_i16, taps_name, taps, flags=0, tag=None):
        """The chain's last FIR, writing only the sign bitmap of its output (pm_fir_signs_*)."""
        ctx = self._ctx
        m = len(taps)
        if x.n < m:
            raise ValueError(f"input of {x.n} samples is shorter than the {m}-tap filter {taps_name}")
        nout = x.n - m + 1
        bits = ctx.scratch((self._own_key(), "signs", tag or taps_name), (nout + 63) // 64 + 1, np.uint64)
        fn = lib().pm_fir_signs_i16 if is_i16 else lib().pm_fir_signs_f64
        check(fn(ctx.handle, x.ptr, x.n, self._const(taps_name, taps).ptr, m, bits.ptr, flags))
        return bits, nout

    def _agc(self, buf):
        if buf.n == 0:
            return
        if not hasattr(self, "_agc_state"):
            self._agc_state = (ctypes.c_double * 2)(0.0, 0.0)
        a = self.AGC
        p = AGCParams(a.attack_rate, a.decay_rate, a.sustain_time, a.sample_rate, a.target_amplitude)
        check(lib().pm_agc_apply(self._ctx.handle, buf.ptr, buf.n, ctypes.byref(p), self._agc_state))

    def reset(self):
        """Back to the just-constructed state (AGC envelope, loop phase/filter/integral) without redesigning taps, so one
        object can process another recording.  The reference's objects are single-use; this is what bench.py uses."""
        if hasattr(self, "_agc_state"):
            self._agc_state[0] = self._agc_state[1] = 0.0
        if hasattr(self, "_loop0"):
            ctypes.memmove(ctypes.byref(self._loop), self._loop0, ctypes.sizeof(Loop))
        self._hist = None                              # carry_history: the next call starts a new stream
        self._tails = {}

    def _finish(self, y, device_out):
        return y if device_out else y.download()


class _AGCSettings:
    """The parameters of agc.py:7-24 (the envelope state lives with the modem that applies it)."""
    def __init__(self, sample_rate, attack_rate, sustain_time, decay_rate, target_amplitude):
        self.sample_rate, self.attack_rate, self.sustain_time = sample_rate, attack_rate, sustain_time
        self.decay_rate, self.target_amplitude = decay_rate, target_amplitude


class _LoopFilterSettings:
    """IIR_1 coefficients (iir.py:15-29), fixed at construction with the constructor's sample rate."""
    def __init__(self, sample_rate, cutoff, gain):
        self.sample_rate, self.cutoff_freq, self.gain = sample_rate, cutoff, gain
        b0, b1, a1 = T.one_pole_lowpass(sample_rate, cutoff, gain)
        self.b_coefs, self.a_coefs = [b0, b1], [0.0, a1]


class _PISettings:
    """PI_control parameters and integral (pi_control.py:8-13)."""
    def __init__(self, p, i, i_limit, gain):
        self.p_rate, self.i_rate, self.i_limit, self.gain = p, i, i_limit, gain
        self.integral = 0.0
        self.proportional = 0.0


def _make_loop(sample_rate, carrier, lpf, pi):
    L = Loop()
    L.phase_scaling = 2.0 * math.pi / sample_rate          # nco.py:31
    L.index_scaling = 256 / (2.0 * math.pi)                # nco.py:27
    L.set_frequency = carrier
    L.b0, L.b1, L.a1 = lpf.b_coefs[0], lpf.b_coefs[1], lpf.a_coefs[1]
    L.p_rate, L.i_rate, L.i_limit, L.gain = pi.p_rate, pi.i_rate, pi.i_limit, pi.gain
    L.integral, L.proportional = pi.integral, pi.proportional
    return L


def _snapshot(loop):
    return ctypes.string_at(ctypes.byref(loop), ctypes.sizeof(Loop))


# =============================================================================================
class AFSKModem(_DeviceStage):
    def __init__(self, **kwargs):
        self.definition = kwargs.get('config', '1200')
        self.sample_rate = kwargs.get('sample_rate', 8000)
        if self.definition == '300':          # afsk.py:19-42
            self.symbol_rate = 300.0
            self.input_bpf_low_cutoff = 1500.0
            self.input_bpf_high_cutoff = 1900.0
            self.input_bpf_span = 7
            self.mark_freq = 1695.0
            self.space_freq = 1705.0
            self.space_gain = 1.0
            self.output_lpf_cutoff = 240.0
            self.output_lpf_span = 2.5
            self.correlator_span = 0.3
            self.correlator_offset = 0.0
        else:                                 # afsk.py:43-66
            self.symbol_rate = 1200.0
            self.input_bpf_low_cutoff = 900.0
            self.input_bpf_high_cutoff = 2500.0
            self.input_bpf_span = 3.7
            self.mark_freq = 1200.0
            self.space_freq = 2200.0
            self.space_gain = 1.0
            self.output_lpf_cutoff = 1400.0
            self.output_lpf_span = 2.5
            self.correlator_span = 1.0
            self.correlator_offset = 0.0
        self.output_oversample = 1.0
        self.tune()

    _KEYS = ('symbol_rate', 'input_bpf_low_cutoff', 'input_bpf_high_cutoff', 'input_bpf_span', 'mark_freq', 'space_freq',
             'space_gain', 'output_lpf_cutoff', 'output_lpf_span', 'correlator_span', 'correlator_offset', 'sample_rate')

    def retune(self, **kwargs):
        for k in self._KEYS:
            setattr(self, k, kwargs.get(k, getattr(self, k)))
        self.tune()

    def StringOptionsRetune(self, options):   # afsk.py:87-100; unknown keys are ignored like there
        for k in self._KEYS:
            setattr(self, k, float(options.get(k, getattr(self, k))))
        self.tune()

    def tune(self):                           # afsk.py:102-146
        self.input_bpf_tap_count = round(self.sample_rate * self.input_bpf_span / self.symbol_rate)
        self.output_lpf_tap_count = round(self.sample_rate * self.output_lpf_span / self.symbol_rate)
        self.input_bpf = T.windowed_sinc(self.input_bpf_tap_count, [self.input_bpf_low_cutoff, self.input_bpf_high_cutoff],
                                         self.sample_rate, pass_zero=False)
        self.output_lpf = T.windowed_sinc(self.output_lpf_tap_count, self.output_lpf_cutoff, self.sample_rate, pass_zero=True)
        (self.mark_correlator_i, self.mark_correlator_q, self.space_correlator_i, self.space_correlator_q) = \
            T.afsk_tone_correlators(self.sample_rate, self.symbol_rate, self.mark_freq, self.space_freq, self.space_gain,
                                    self.correlator_span, self.correlator_offset)
        self.output_sample_rate = self.output_oversample * self.sample_rate

    def front_end(self, input_audio):
        """Input band-pass only (afsk.py:151).  Chains with the same BPF (every chain of afsk_1200_ax25_super_opt.json)
        can share its output: chain_execute.process_chains_device does."""
        x, is_i16 = self._input(input_audio)
        return self._fir(x, is_i16, "input_bpf", self.input_bpf)

    def _history_len(self):
        return len(self.input_bpf) - 1 + len(self.mark_correlator_i) - 1 + len(self.output_lpf) - 1

    def front_end_key(self):
        return ("afsk", float(self.sample_rate), self.input_bpf.tobytes())

    def demod(self, input_audio, device_out=False):   # afsk.py:148-167
        if self.carry_history:
            x, is_i16 = self._input(input_audio)
            x = self._with_history(x, is_i16)
            empty = self._starved(x, False, device_out)
            if empty is not None:
                return empty
            return self.back_end(self._fir(x, is_i16, "input_bpf", self.input_bpf), device_out)
        return self.back_end(self.front_end(input_audio), device_out)

    def demod_signs(self, input_audio):
        """demod() for a slicer: the output low-pass writes only the sign bitmap.  -> SignBits"""
        if self.carry_history:
            x, is_i16 = self._input(input_audio)
            x = self._with_history(x, is_i16)
            empty = self._starved(x, True, True)
            if empty is not None:
                return empty
            return self.back_end(self._fir(x, is_i16, "input_bpf", self.input_bpf), signs=True)
        return self.back_end(self.front_end(input_audio), signs=True)

    def mark_key(self):
        """Modems with equal keys see the same band-passed stream through the same mark correlators: their correlator banks
        can run as one pm_afsk_correlate_group launch (chain_execute.process_chains_device)."""
        return (self.front_end_key(), self.mark_correlator_i.tobytes(), self.mark_correlator_q.tobytes())

    @staticmethod
    def correlate_group(modems, a, out_key):
        """Correlator banks of `modems` (equal mark_key) over the band-passed stream `a` in one launch -> one stream per modem."""
        lead = modems[0]
        ctx = lead._context()
        m, g = len(lead.mark_correlator_i), len(modems)
        if a.n < m:
            raise ValueError("input shorter than the correlators")
        nout = a.n - m + 1
        stride = (nout + 63) // 64 * 64
        space = np.stack([np.stack([md.space_correlator_i, md.space_correlator_q]) for md in modems])
        out = ctx.scratch(out_key, stride * g, np.float64)
        check(lib().pm_afsk_correlate_group(ctx.handle, a.ptr, a.n, lead._const("mi", lead.mark_correlator_i).ptr,
                                            lead._const("mq", lead.mark_correlator_q).ptr, lead._const("space_group", space.reshape(-1)).ptr,
                                            g, m, out.ptr, stride))
        return [out.view(j * stride, nout) for j in range(g)]

    def correlate(self, a, out_key=None):
        """This modem's correlator bank over the band-passed stream `a` (afsk.py:153-162) -> DeviceBuffer."""
        ctx = self._context()
        m = len(self.mark_correlator_i)
        if a.n < m:
            raise ValueError("input shorter than the correlators")
        c = ctx.scratch(out_key or (self._key(), "corr"), a.n - m + 1, np.float64)
        check(lib().pm_afsk_correlate(ctx.handle, a.ptr, a.n, self._const("mi", self.mark_correlator_i).ptr,
                                      self._const("mq", self.mark_correlator_q).ptr, self._const("si", self.space_correlator_i).ptr,
                                      self._const("sq", self.space_correlator_q).ptr, m, c.ptr))
        return c

    @staticmethod
    def lpf_signs_batch(modems, corrs):
        """Output low-pass (afsk.py:166) of several modems with EQUAL output_lpf taps over their correlator streams, sign bitmaps
        only, in one launch per 16 streams (pm_fir_signs_f64_batch) -> one SignBits per modem."""
        lead = modems[0]
        ctx = lead._context()
        m = len(lead.output_lpf)
        taps = lead._const("output_lpf", lead.output_lpf)
        out = []
        for base in range(0, len(modems), 16):
            mods, cs = modems[base:base + 16], corrs[base:base + 16]
            g = len(mods)
            xs, ns, bs = (ctypes.c_void_p * g)(), (ctypes.c_int64 * g)(), (ctypes.c_void_p * g)()
            bufs = []
            for j, (md, c) in enumerate(zip(mods, cs)):
                if c.n < m:
                    raise ValueError("input shorter than the output filter")
                md._context()
                nout = c.n - m + 1
                bits = ctx.scratch((md._own_key(), "signs", "output_lpf"), (nout + 63) // 64 + 1, np.uint64)
                xs[j], ns[j], bs[j] = c.ptr.value, c.n, bits.ptr.value
                bufs.append((bits, nout))
            check(lib().pm_fir_signs_f64_batch(ctx.handle, g, xs, ns, taps.ptr, m, bs, 0))
            out += [SignBits(b, None, n) for b, n in bufs]
        return out

    def unit_space_correlators(self):
        """The space correlators this modem would have with space_gain 1.0 (afsk.py:144-145 without the factor)."""
        args = (self.sample_rate, self.symbol_rate, self.mark_freq, self.space_freq, self.correlator_span, self.correlator_offset)
        hit = getattr(self, "_unit_memo", None)
        if hit is None or hit[0] != args:
            _, _, ui, uq = T.afsk_tone_correlators(args[0], args[1], args[2], args[3], 1.0, args[4], args[5])
            hit = self._unit_memo = (args, ui, uq)
        return hit[1], hit[2]

    def sweep_key(self):
        """Modems with equal keys differ in space_gain only (same band-pass, tones, span, output low-pass) AND their space taps are
        exactly gain * unit taps: they can take pm_afsk_sweep_signs together.  None if this modem's taps are not of that form.
        (Remembered for as long as the tap arrays are the same objects and the gain is the same.)"""
        deps = (self.input_bpf, self.mark_correlator_i, self.mark_correlator_q, self.space_correlator_i, self.space_correlator_q,
                self.output_lpf)
        hit = getattr(self, "_sweep_memo", None)
        if hit is not None and hit[1] == self.space_gain and len(hit[0]) == len(deps) and all(a is b for a, b in zip(hit[0], deps)):
            return hit[2]
        ui, uq = self.unit_space_correlators()
        key = None
        if np.array_equal(self.space_correlator_i, self.space_gain * ui) and np.array_equal(self.space_correlator_q, self.space_gain * uq):
            key = (self.mark_key(), ui.tobytes(), uq.tobytes(), self.output_lpf.tobytes())
        self._sweep_memo = (deps, self.space_gain, key)
        return key

    @staticmethod
    def _sweep_prepare(modems):
        """Everything of a certified sweep that depends on the modems' taps only, prepared once per set of modems (as long as the tap
        arrays are the same objects and the gains the same): the submitting thread of a pipelined host comes through here twice per
        recording."""
        lead = modems[0]
        g = len(modems)
        deps = [lead.mark_correlator_i, lead.mark_correlator_q, lead.output_lpf] + [t for md in modems for t in (md.space_correlator_i, md.space_correlator_q)]
        sig = (tuple(id(md) for md in modems), tuple(float(md.space_gain) for md in modems), AFSKModem.sliding_sums)
        memo = getattr(lead, "_sweep_prep", None)
        if memo is None or memo[0] != sig or len(memo[1]) != len(deps) or not all(x is y for x, y in zip(memo[1], deps)):
            ui, uq = lead.unit_space_correlators()
            space = np.stack([np.stack([md.space_correlator_i, md.space_correlator_q]) for md in modems])
            prep = {"gains": (ctypes.c_double * g)(*[float(md.space_gain) for md in modems]),
                    "consts": [lead._const("mi", lead.mark_correlator_i), lead._const("mq", lead.mark_correlator_q), lead._const("unit_i", ui),
                               lead._const("unit_q", uq), lead._const("space_group", space.reshape(-1)), lead._const("output_lpf", lead.output_lpf)],
                    "lpf_abs": float(np.abs(lead.output_lpf).sum()),
                    "tones": lead._tones(ui, uq) if AFSKModem.sliding_sums else None}
            memo = lead._sweep_prep = (sig, deps, prep)
        return memo[2]

    @staticmethod
    def sweep_signs(modems, a, x_bound):
        """Sign bitmaps of a gain sweep (equal sweep_key) over the band-passed stream `a`, |a| <= x_bound guaranteed by the caller
        -> [SignBits per modem].  Never waits for the GPU; `sweep_uncertain(ctx)` tells afterwards how many samples had to be
        recomputed exactly."""
        lead = modems[0]
        ctx = lead._context()
        mc, ml, g = len(lead.mark_correlator_i), len(lead.output_lpf), len(modems)
        if a.n < mc + ml - 1:
            raise ValueError("input shorter than the correlators and the output filter")
        nout = a.n - mc - ml + 2
        prep = AFSKModem._sweep_prepare(modems)
        gains = prep["gains"]
        bits, ptrs = [], (ctypes.c_void_p * g)()
        for j, md in enumerate(modems):
            md._context()
            b = ctx.scratch((md._own_key(), "signs", "output_lpf"), (nout + 63) // 64 + 1, np.uint64)
            bits.append(b)
            ptrs[j] = b.ptr.value
        k = prep["consts"]
        args = (ctx.handle, a.ptr, a.n, float(x_bound), k[0].ptr, k[1].ptr, k[2].ptr, k[3].ptr, k[4].ptr, gains, g, mc, k[5].ptr, ml, prep["lpf_abs"], ptrs)
        tones = prep["tones"]
        if tones is not None:
            check(lib().pm_afsk_sweep_signs_tones(*args, ctypes.byref(tones)))
        else:
            check(lib().pm_afsk_sweep_signs(*args))
        AFSKModem.sweeps_run += 1
        out = [SignBits(b, None, nout) for b in bits]
        ticket = ctypes.c_int64()
        check(lib().pm_afsk_sweep_ticket(ctx.handle, ctypes.byref(ticket)))
        for sb in out:
            sb.sweep = (ctx, ticket.value)             # for sweep_overflowed() (deferred fallback, pm_afsk_sweep_mode)
        return out

    @staticmethod
    def sweep_overflowed(sweep, via=None):
        """(ctx, ticket) of a FINISHED certified sweep -> True if it had more uncertain samples than its list holds: with the fallback
        deferred (pm_afsk_sweep_mode) its bitmaps are then not valid and the exact chain has to run for its modems."""
        ctx, ticket = sweep
        n, cap = ctypes.c_int64(), ctypes.c_int64()
        check(lib().pm_afsk_sweep_result(ctx.handle, ticket, via.handle if via is not None else None, ctypes.byref(n), ctypes.byref(cap)))
        return n.value > cap.value

    @staticmethod
    def sweep_uncertain(ctx):
        """Samples the last pm_afsk_sweep_signs on `ctx` could not certify (recomputed exactly; above 65536 the exact chains ran)."""
        v = ctypes.c_int64()
        check(lib().pm_afsk_sweep_last(ctx.handle, ctypes.byref(v)))
        return v.value

    sweeps_run = 0
    sliding_sums = True        # pm_afsk_sweep_signs_tones where the templates are tones (tests switch it off to compare)

    def _tones(self, ui, uq):
        """pm_afsk_tones of this modem's mark and unit-space templates, or None if they are not the powers of one rotation each
        (remembered for as long as the template arrays are the same objects)."""
        from ._native import AfskTones
        deps = (self.mark_correlator_i, self.mark_correlator_q, ui, uq)
        hit = getattr(self, "_tones_memo", None)
        if hit is not None and all(x is y for x, y in zip(hit[0], deps)):
            return hit[1]
        mk, sp = T.tone_model(deps[0], deps[1]), T.tone_model(ui, uq)
        tones = None
        if mk is not None and sp is not None and max(mk[2], sp[2]) < 1e-9:
            tones = AfskTones()
            tones.mark_rot[:], tones.mark_end[:] = mk[0], mk[1]
            tones.space_rot[:], tones.space_end[:] = sp[0], sp[1]
            tones.tap_dev = max(mk[2], sp[2])
        self._tones_memo = (deps, tones)
        return tones


    def back_end(self, a, device_out=False, signs=False, correlated=None):
        """Correlators + output low-pass on an already band-passed stream (afsk.py:153-166).  `correlated`: this modem's
        correlator output when a group launch already produced it."""
        ctx = self._context()
        m = len(self.mark_correlator_i)
        if a.n < m:
            raise ValueError("input shorter than the correlators")
        c = correlated
        if c is None:
            c = ctx.scratch((self._key(), "corr"), a.n - m + 1, np.float64)
            check(lib().pm_afsk_correlate(ctx.handle, a.ptr, a.n, self._const("mi", self.mark_correlator_i).ptr,
                                          self._const("mq", self.mark_correlator_q).ptr, self._const("si", self.space_correlator_i).ptr,
                                          self._const("sq", self.space_correlator_q).ptr, m, c.ptr))
        if signs:
            bits, nout = self._fir_signs(c, False, "output_lpf", self.output_lpf)
            return SignBits(bits, None, nout)
        y = self._fir(c, False, "output_lpf", self.output_lpf)
        return self._finish(y, device_out)


# =============================================================================================
class FSKModem(_DeviceStage):
    """One FIR (+ optional negate).  The reference builds an AGC but never applies it, and has no
    `output_sample_rate` attribute (the runner falls back to the input rate, pymodem.py:87-90)."""
    _PRESETS = {   # fsk.py:25-103: (symbol_rate, filter type, cutoff, span, rolloff, agc attack/sustain/decay)
        '9600': (9600.0, 'lpf', 6000.0, 1.5, False, (1, 0.1, 1)),
        '4800': (4800.0, 'lpf', 3000.0, 1.5, False, (1, 0.1, 1)),
        '4800-rrc': (4800.0, 'rrc', None, 9, 0.2, (3, 0.1, 3)),
        '9600-rrc': (9600.0, 'rrc', None, 9, 0.2, (50, 0.1, 3)),
        '4800-gauss': (4800.0, 'lpf', 0.9 * 4800.0, 4, False, (50, 0.1, 3)),
        '9600-gauss': (9600.0, 'lpf', 0.9 * 9600.0, 4, False, (50, 0.1, 3)),
    }

    def __init__(self, **kwargs):
        self.definition = kwargs.get('config', '9600')
        self.sample_rate = kwargs.get('sample_rate', 96000)
        p = self._PRESETS.get(self.definition, self._PRESETS['9600'])
        self.symbol_rate, self.input_filter_type, cutoff, self.input_lpf_span, self.rrc_rolloff_rate, agc = p
        if cutoff is not None:
            self.input_lpf_cutoff = cutoff
        self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate = agc
        self.invert = False
        self.tune()

    def StringOptionsRetune(self, options):   # fsk.py:110-113
        self.invert = check_boolean(options.get('invert', "false"))
        self.tune()

    def tune(self):                           # fsk.py:115-147
        self.input_lpf_tap_count = round(self.sample_rate * self.input_lpf_span / self.symbol_rate)
        if self.input_filter_type == 'rrc':
            self.input_lpf = T.root_raised_cosine(self.sample_rate, self.symbol_rate, self.input_lpf_span, self.rrc_rolloff_rate)
        else:
            self.input_lpf = T.windowed_sinc(self.input_lpf_tap_count, [self.input_lpf_cutoff], self.sample_rate, pass_zero=True)

    def _history_len(self):
        return len(self.input_lpf) - 1

    def demod(self, input_audio, device_out=False):   # fsk.py:149-159
        x, is_i16 = self._input(input_audio)
        x = self._with_history(x, is_i16)
        empty = self._starved(x, False, device_out)
        if empty is not None:
            return empty
        y = self._fir(x, is_i16, "input_lpf", self.input_lpf, flags=1 if self.invert else 0)
        return self._finish(y, device_out)

    def demod_signs(self, input_audio):
        x, is_i16 = self._input(input_audio)
        x = self._with_history(x, is_i16)
        empty = self._starved(x, True, True)
        if empty is not None:
            return empty
        bits, nout = self._fir_signs(x, is_i16, "input_lpf", self.input_lpf, flags=1 if self.invert else 0)
        return SignBits(bits, None, nout)

    def front_end_key(self):
        """FSK modems with equal keys are the same filter on the same audio (the three chains of configs/fsk_9600.json:1-3 differ
        in stream and codec only): the group executor computes their sign bitmap once (chain_execute.process_chains_device)."""
        return ("fsk", float(self.sample_rate), self.input_lpf.tobytes(), bool(self.invert))


# =============================================================================================
class BPSKModem(_DeviceStage):
    def __init__(self, **kwargs):
        self.definition = kwargs.get('config', '300')
        self.sample_rate = kwargs.get('sample_rate', 8000.0)
        if self.definition == '300':          # psk.py:26-55
            agc, self.symbol_rate = (500.0, 1.0, 50.0), 300.0
            self.input_bpf_low_cutoff, self.input_bpf_high_cutoff, self.input_bpf_span = 1200.0, 1800.0, 1.5
            self.carrier_freq, self.rrc_rolloff_rate, self.rrc_span = 1500.0, 0.6, 6
            self.max_freq_offset = 25 * 1.25
            lpf, pi_p, pi_gain = (250.0, 1.0), 0.06, 7200
        elif self.definition == '1200':       # psk.py:56-85
            agc, self.symbol_rate = (500.0, 1.0, 50.0), 1200.0
            self.input_bpf_low_cutoff, self.input_bpf_high_cutoff, self.input_bpf_span = 200.0, 2800.0, 4.80
            self.carrier_freq, self.rrc_rolloff_rate, self.rrc_span = 1500.0, 0.9, 6
            self.max_freq_offset = 50 * 1.25
            lpf, pi_p, pi_gain = (250.0, 1.0), 0.4, 1800
        else:
            raise AttributeError(f"BPSKModem has no preset {self.definition!r}")   # the reference fails on a missing attribute
        self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate = agc
        self.Loop_LPF = _LoopFilterSettings(self.sample_rate, lpf[0], lpf[1])
        self.FeedbackController = _PISettings(pi_p, pi_p / 1000, self.max_freq_offset, pi_gain)
        self.oscillator_amplitude = 1.0
        self.tune()

    _KEYS = ('symbol_rate', 'input_bpf_low_cutoff', 'input_bpf_high_cutoff', 'input_bpf_span', 'sample_rate', 'carrier_freq')

    def retune(self, **kwargs):
        for k in self._KEYS:
            setattr(self, k, kwargs.get(k, getattr(self, k)))
        self.tune()

    def StringOptionsRetune(self, options):   # psk.py:102-109
        for k in self._KEYS:
            setattr(self, k, float(options.get(k, getattr(self, k))))
        self.tune()

    def tune(self):                           # psk.py:111-160
        self.input_bpf_tap_count = round(self.sample_rate * self.input_bpf_span / self.symbol_rate)
        self.input_bpf = T.windowed_sinc(self.input_bpf_tap_count, [self.input_bpf_low_cutoff, self.input_bpf_high_cutoff],
                                         self.sample_rate, pass_zero=False)
        self.AGC = _AGCSettings(self.sample_rate, self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate, self.oscillator_amplitude)
        self.wavetable = T.sine_wavetable(self.oscillator_amplitude, 256)
        self.rrc_taps = T.root_raised_cosine(self.sample_rate, self.symbol_rate, self.rrc_span, self.rrc_rolloff_rate)
        self._loop = _make_loop(self.sample_rate, self.carrier_freq, self.Loop_LPF, self.FeedbackController)
        self._loop0 = _snapshot(self._loop)
        self.output_sample_rate = self.sample_rate

    loop_entry = "pm_costas_bpsk"

    def front_end(self, input_audio):
        """Band-pass + AGC (psk.py:165-168): what chains that differ only in carrier_freq share."""
        x, is_i16 = self._input(input_audio)
        x = self._carry("audio", x, len(self.input_bpf) - 1)
        a = self._fir_carried(x, is_i16, "input_bpf", self.input_bpf)
        self._agc(a)
        return a

    def front_end_key(self):
        g = self.AGC
        return ("bpsk", float(self.sample_rate), self.input_bpf.tobytes(), (g.attack_rate, g.decay_rate, g.sustain_time, g.target_amplitude),
                id(self) if self.carry_history else 0)

    def demod(self, input_audio, device_out=False, signs=False):   # psk.py:162-195
        a = self.front_end(input_audio)
        ctx = self._ctx
        d = ctx.scratch((self._key(), "loop"), a.n, np.float64)
        if a.n:
            check(lib().pm_costas_bpsk(ctx.handle, ctypes.byref(self._loop), 1, self._const("wavetable", self.wavetable).ptr,
                                       a.ptr, 0, a.n, d.ptr, a.n))
        return self.back_end(d, device_out, signs)

    def back_end(self, d, device_out=False, signs=False):
        """Matched filter on the loop output (psk.py:193)."""
        self._context()
        d = self._carry("loop", d, len(self.rrc_taps) - 1)
        if signs:
            bits, nout = self._fir_carried(d, False, "rrc", self.rrc_taps, signs=True)
            return SignBits(bits, None, nout)
        y = self._fir_carried(d, False, "rrc", self.rrc_taps)
        return self._finish(y, device_out)

    def demod_signs(self, input_audio):
        return self.demod(input_audio, signs=True)


# =============================================================================================
class QPSKModem(_DeviceStage):
    """psk.py:197-476: band-pass -> AGC -> QPSK Costas loop with low-passed branches (pm_costas_qpsk) -> RRC on both arms.
    chain_builder type 'qpsk'; no bundled config uses it, the generated recordings of tests/golden/qpsk_modem.npz do."""
    _PRESETS = {   # psk.py:203-338
        '600': dict(agc=(500.0, 1.0, 50.0), symbol_rate=300.0, lo=1200.0, hi=1800.0, span=1.5, carrier=1500.0, out_cut=200.0, out_span=1.5,
                    rolloff=0.6, rrc_span=6, max_off=37.5, branch=300.0, loop=100.0, p=0.02, i_div=651, gain=858),
        '3600': dict(agc=(5000.0, 0.1, 50.0), symbol_rate=1800, lo=300.0, hi=3000.0, span=5, carrier=1650.0, out_cut=900.0, out_span=1.5,
                     rolloff=0.3, rrc_span=8, max_off=50, branch=1450.0, loop=200.0, p=0.15, i_div=1000, gain=1350.0),
        '2400': dict(agc=(500.0, 1, 50.0), symbol_rate=1200.0, lo=200.0, hi=2800.0, span=4.8, carrier=1800.0, out_cut=900.0, out_span=1.5,
                     rolloff=0.9, rrc_span=3, max_off=87.5, branch=1200.0, loop=200.0, p=.1, i_div=500, gain=450.0),
    }

    def __init__(self, **kwargs):
        self.definition = kwargs.get('config', '600')
        self.sample_rate = kwargs.get('sample_rate', 44100.0)
        if self.definition not in self._PRESETS:
            raise AttributeError(f"QPSKModem has no preset {self.definition!r}")   # the reference fails on a missing attribute
        p = self._PRESETS[self.definition]
        self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate = p['agc']
        self.symbol_rate = p['symbol_rate']
        self.input_bpf_low_cutoff, self.input_bpf_high_cutoff, self.input_bpf_span = p['lo'], p['hi'], p['span']
        self.carrier_freq = p['carrier']
        self.output_lpf_cutoff, self.output_lpf_span = p['out_cut'], p['out_span']
        self.rrc_rolloff_rate, self.rrc_span, self.max_freq_offset = p['rolloff'], p['rrc_span'], p['max_off']
        # the three IIR_1 filters take the constructor's sample rate (psk.py:223-240), not a later retune's
        self.Cosine_LPF = _LoopFilterSettings(self.sample_rate, p['branch'], 1.0)
        self.Sine_LPF = _LoopFilterSettings(self.sample_rate, p['branch'], 1.0)
        self.Loop_LPF = _LoopFilterSettings(self.sample_rate, p['loop'], 1.0)
        self.FeedbackController = _PISettings(p['p'], p['p'] / p['i_div'], self.max_freq_offset, p['gain'])
        self.oscillator_amplitude = 1.0
        self.tune()

    _KEYS = ('symbol_rate', 'input_bpf_low_cutoff', 'input_bpf_high_cutoff', 'input_bpf_span', 'output_lpf_cutoff', 'output_lpf_span',
             'sample_rate', 'carrier_freq')

    def retune(self, **kwargs):               # psk.py:346-355
        for k in self._KEYS:
            setattr(self, k, kwargs.get(k, getattr(self, k)))
        self.tune()

    def StringOptionsRetune(self, options):   # psk.py:357-366
        for k in self._KEYS:
            setattr(self, k, float(options.get(k, getattr(self, k))))
        self.tune()

    def tune(self):                           # psk.py:368-424
        self.input_bpf_tap_count = round(self.sample_rate * self.input_bpf_span / self.symbol_rate)
        self.output_lpf_tap_count = round(self.sample_rate * self.output_lpf_span / self.symbol_rate)
        self.input_bpf = T.windowed_sinc(self.input_bpf_tap_count, [self.input_bpf_low_cutoff, self.input_bpf_high_cutoff],
                                         self.sample_rate, pass_zero=False)
        self.output_lpf = T.windowed_sinc(self.output_lpf_tap_count, self.output_lpf_cutoff, self.sample_rate, pass_zero=True)   # designed, never applied
        self.AGC = _AGCSettings(self.sample_rate, self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate, self.oscillator_amplitude)
        self.wavetable = T.sine_wavetable(self.oscillator_amplitude, 256)
        self.rrc_taps = T.root_raised_cosine(self.sample_rate, self.symbol_rate, self.rrc_span, self.rrc_rolloff_rate)
        self._loop = _make_loop(self.sample_rate, self.carrier_freq, self.Loop_LPF, self.FeedbackController)
        self._loop.bb0, self._loop.bb1, self._loop.ba1 = self.Cosine_LPF.b_coefs[0], self.Cosine_LPF.b_coefs[1], self.Cosine_LPF.a_coefs[1]
        self._loop0 = _snapshot(self._loop)
        self.output_sample_rate = self.sample_rate

    def front_end(self, input_audio):
        x, is_i16 = self._input(input_audio)
        x = self._carry("audio", x, len(self.input_bpf) - 1)
        a = self._fir_carried(x, is_i16, "input_bpf", self.input_bpf)
        self._agc(a)
        return a

    def demod(self, input_audio, device_out=False, signs=False):   # psk.py:426-476
        a = self.front_end(input_audio)
        ctx = self._ctx
        i_arm = ctx.scratch((self._key(), "i_arm"), a.n, np.float64)
        q_arm = ctx.scratch((self._key(), "q_arm"), a.n, np.float64)
        if a.n:
            check(lib().pm_costas_qpsk(ctx.handle, ctypes.byref(self._loop), 1, self._const("wavetable", self.wavetable).ptr,
                                       a.ptr, 0, a.n, i_arm.ptr, q_arm.ptr, a.n))
        i_arm = self._carry("i_arm", i_arm, len(self.rrc_taps) - 1)
        q_arm = self._carry("q_arm", q_arm, len(self.rrc_taps) - 1)
        if signs:
            bi, nout = self._fir_carried(i_arm, False, "rrc", self.rrc_taps, tag="i", signs=True)
            bq, _ = self._fir_carried(q_arm, False, "rrc", self.rrc_taps, tag="q", signs=True)
            return SignBits(bi, bq, nout)
        i_out = self._fir_carried(i_arm, False, "rrc", self.rrc_taps, tag="i_out")
        q_out = self._fir_carried(q_arm, False, "rrc", self.rrc_taps, tag="q_out")
        if device_out:
            return DeviceIQ(i_out, q_out)
        out = IQData()
        out.i_data, out.q_data = i_out.download(), q_out.download()
        return out

    def demod_signs(self, input_audio):
        return self.demod(input_audio, signs=True)


# =============================================================================================
class MPSKModem(_DeviceStage):
    _PRESETS = {   # psk.py:485-628
        'qpsk_3600': dict(const='qpsk', agc=(5000.0, 0.1, 50.0), symbol_rate=1800, lo=300.0, hi=3000.0, span=2, hilbert=4.5,
                          carrier=1650.0, max_off=12.5 * 1.25, rolloff=0.3, lpf=(250.0, 1), p=0.15, i_div=1000, gain=(14400 / 65536)),
        'qpsk_600': dict(const='qpsk', agc=(500.0, 1, 50.0), symbol_rate=300, lo=1200.0, hi=1800.0, span=4, hilbert=3.4,
                         carrier=1500.0, max_off=25, rolloff=0.6, lpf=(150, 1), p=0.1, i_div=1000, gain=(7200 / 65536)),
        'qpsk_2400': dict(const='qpsk', agc=(500.0, 1, 50.0), symbol_rate=1200, lo=200.0, hi=2800.0, span=2.7, hilbert=3.4,
                          carrier=1500.0, max_off=25 * 1.25, rolloff=0.9, lpf=(250.0, 1), p=0.3, i_div=2000, gain=(14400 / 65536)),
        'bpsk_300': dict(const='bpsk', agc=(500.0, 1, 50.0), symbol_rate=300, lo=1200.0, hi=1800.0, span=2.7, hilbert=2.7,
                         carrier=1500.0, max_off=50, rolloff=0.6, lpf=(250.0, 1.0), p=0.15, i_div=1000, gain=1.5 * (500)),
        'bpsk_1200': dict(const='bpsk', agc=(500.0, 1, 50.0), symbol_rate=1200, lo=200.0, hi=2800.0, span=4.8, hilbert=2,
                          carrier=1500.0, max_off=87.5, rolloff=0.9, lpf=(200.0, 1.0), p=0.15, i_div=1000, gain=5),
    }

    def __init__(self, **kwargs):
        self.definition = kwargs.get('config', 'qpsk_3600')
        self.sample_rate = kwargs.get('sample_rate', 44100.0)
        if self.definition not in self._PRESETS:
            raise AttributeError(f"MPSKModem has no preset {self.definition!r}")
        p = self._PRESETS[self.definition]
        self.constellation_id = p['const']
        self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate = p['agc']
        self.symbol_rate = p['symbol_rate']
        self.input_bpf_low_cutoff, self.input_bpf_high_cutoff = p['lo'], p['hi']
        self.input_bpf_span, self.hilbert_span = p['span'], p['hilbert']      # milliseconds
        self.carrier_freq, self.max_freq_offset = p['carrier'], p['max_off']
        self.rrc_rolloff_rate, self.rrc_span = p['rolloff'], 6
        self.Loop_LPF = _LoopFilterSettings(self.sample_rate, p['lpf'][0], p['lpf'][1])
        self.FeedbackController = _PISettings(p['p'], p['p'] / p['i_div'], self.max_freq_offset, p['gain'])
        self.oscillator_amplitude = 1.0
        self.pd_gain = 32
        self.tune()

    def StringOptionsRetune(self, options):   # psk.py:633-637
        self.symbol_rate = float(options.get('symbol_rate', self.symbol_rate))
        self.sample_rate = float(options.get('sample_rate', self.sample_rate))
        self.carrier_freq = float(options.get('carrier_freq', self.carrier_freq))
        self.tune()

    def tune(self):                           # psk.py:639-703
        self.input_bpf_tap_count = round(self.sample_rate * self.input_bpf_span / 1000)
        self.hilbert_tap_count = round(self.sample_rate * self.hilbert_span / 1000)
        self.input_bpf = T.windowed_sinc(self.input_bpf_tap_count, [self.input_bpf_low_cutoff, self.input_bpf_high_cutoff],
                                         self.sample_rate, pass_zero=False)
        if self.hilbert_tap_count % 2 == 0:
            self.hilbert_tap_count += 1
        self.hilbert_taps, self.hilbert_delay = T.hilbert_transformer(self.hilbert_tap_count)
        self.AGC = _AGCSettings(self.sample_rate, self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate, self.oscillator_amplitude)
        self.wavetable = T.sine_wavetable(self.oscillator_amplitude, 256)
        self.rrc_taps = T.root_raised_cosine(self.sample_rate, self.symbol_rate, self.rrc_span, self.rrc_rolloff_rate)
        self.output_sample_rate = self.sample_rate
        self.FeedbackController.integral = -self.max_freq_offset      # psk.py:703: start at the maximum offset
        self._loop = _make_loop(self.sample_rate, self.carrier_freq, self.Loop_LPF, self.FeedbackController)
        self._loop0 = _snapshot(self._loop)
        self.phase_error_table = T.qpsk_error_table(64, self.pd_gain)

    def front_end(self, input_audio):
        """BPF -> AGC -> Hilbert pair (psk.py:710-716): (real, imag) DeviceBuffers of equal length.  Chains that differ
        only in carrier_freq (configs/qpsk_2400.json) share this part; see chain_execute.run_mpsk_group."""
        x, is_i16 = self._input(input_audio)
        x = self._carry("audio", x, len(self.input_bpf) - 1)
        a = self._fir_carried(x, is_i16, "input_bpf", self.input_bpf)
        self._agc(a)
        a = self._carry("agc", a, len(self.hilbert_taps) - 1)
        imag = self._fir_carried(a, False, "hilbert", self.hilbert_taps)
        # the delay FIR [1,0,...,0] followed by [:-delay] is a pure shift: real[k] = a[k + delay]
        real = a.view(self.hilbert_delay, imag.n) if imag.n else self._empty("real")
        return real, imag

    def front_end_key(self):
        a = self.AGC
        return ("mpsk", float(self.sample_rate), self.input_bpf.tobytes(), self.hilbert_taps.tobytes(),
                (a.attack_rate, a.decay_rate, a.sustain_time, a.target_amplitude), id(self) if self.carry_history else 0)

    def demod(self, input_audio, device_out=False, signs=False):   # psk.py:705-773
        real, imag = self.front_end(input_audio)
        ctx = self._ctx
        n = imag.n
        i_mix = ctx.scratch((self._key(), "i_mix"), n, np.float64)
        q_mix = ctx.scratch((self._key(), "q_mix"), n, np.float64)
        if n:
            check(lib().pm_mpsk_loop(ctx.handle, ctypes.byref(self._loop), 1, self._const("wavetable", self.wavetable).ptr,
                                     self._const("pd", self.phase_error_table.reshape(-1), np.int32).ptr,
                                     real.ptr, imag.ptr, 0, n, i_mix.ptr, q_mix.ptr, n))
        return self.back_end(i_mix, q_mix, device_out, signs)

    def demod_signs(self, input_audio):
        return self.demod(input_audio, signs=True)

    def back_end(self, i_mix, q_mix, device_out=False, signs=False):
        """Matched filter on both arms of the carrier-loop output (psk.py:750-751)."""
        self._context()
        i_mix = self._carry("i_mix", i_mix, len(self.rrc_taps) - 1)
        q_mix = self._carry("q_mix", q_mix, len(self.rrc_taps) - 1)
        if signs:
            bi, nout = self._fir_carried(i_mix, False, "rrc", self.rrc_taps, tag="i", signs=True)
            bq, _ = self._fir_carried(q_mix, False, "rrc", self.rrc_taps, tag="q", signs=True)
            return SignBits(bi, bq, nout)
        i_out = self._fir_carried(i_mix, False, "rrc", self.rrc_taps, tag="i_out")
        q_out = self._fir_carried(q_mix, False, "rrc", self.rrc_taps, tag="q_out")
        if device_out:
            return DeviceIQ(i_out, q_out)
        out = IQData()
        out.i_data, out.q_data = i_out.download(), q_out.download()
        return out


# =============================================================================================
class AFSKPLLModem(_DeviceStage):
    def __init__(self, **kwargs):
        self.definition = kwargs.get('config', '300')
        self.sample_rate = kwargs.get('sample_rate', 8000.0)
        if self.definition != '300':          # afsk_pll.py:22-54 defines this preset only
            raise AttributeError(f"AFSKPLLModem has no preset {self.definition!r}")
        self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate = 500.0, 1.0, 50.0
        self.symbol_rate = 300.0
        self.input_bpf_low_cutoff, self.input_bpf_high_cutoff, self.input_bpf_span = 1500.0, 1900.0, 7.0
        self.carrier_freq = 1700.0
        self.output_lpf_cutoff, self.output_lpf_span = 240.0, 5
        self.max_freq_offset = 50
        self.LoopFilter = _LoopFilterSettings(self.sample_rate, 150.0, 1.0)
        self.FeedbackController = _PISettings(0.6, 0.6 / 6000, self.max_freq_offset, 900)
        self.oscillator_amplitude = 1.0
        self.tune()

    _KEYS = ('symbol_rate', 'input_bpf_low_cutoff', 'input_bpf_high_cutoff', 'input_bpf_span', 'output_lpf_cutoff',
             'output_lpf_span', 'sample_rate', 'carrier_freq')

    def retune(self, **kwargs):
        for k in self._KEYS:
            setattr(self, k, kwargs.get(k, getattr(self, k)))
        self.tune()

    def StringOptionsRetune(self, options):   # afsk_pll.py:73-82
        for k in self._KEYS:
            setattr(self, k, float(options.get(k, getattr(self, k))))
        self.tune()

    def tune(self):                           # afsk_pll.py:84-138
        self.input_bpf_tap_count = round(self.sample_rate * self.input_bpf_span / self.symbol_rate)
        self.output_lpf_tap_count = round(self.sample_rate * self.output_lpf_span / self.symbol_rate)
        self.input_bpf = T.windowed_sinc(self.input_bpf_tap_count, [self.input_bpf_low_cutoff, self.input_bpf_high_cutoff],
                                         self.sample_rate, pass_zero=False)
        self.output_lpf = T.windowed_sinc(self.output_lpf_tap_count, self.output_lpf_cutoff, self.sample_rate, pass_zero=True)
        self.AGC = _AGCSettings(self.sample_rate, self.agc_attack_rate, self.agc_sustain_time, self.agc_decay_rate, self.oscillator_amplitude)
        self.wavetable = T.sine_wavetable(self.oscillator_amplitude, 256)
        self._loop = _make_loop(self.sample_rate, self.carrier_freq, self.LoopFilter, self.FeedbackController)
        self._loop0 = _snapshot(self._loop)
        self.output_sample_rate = self.sample_rate

    loop_entry = "pm_pll_afsk"

    def front_end(self, input_audio):
        """Band-pass + AGC (afsk_pll.py:143-146)."""
        x, is_i16 = self._input(input_audio)
        x = self._carry("audio", x, len(self.input_bpf) - 1)
        a = self._fir_carried(x, is_i16, "input_bpf", self.input_bpf)
        self._agc(a)
        return a

    def front_end_key(self):
        g = self.AGC
        return ("pll", float(self.sample_rate), self.input_bpf.tobytes(), (g.attack_rate, g.decay_rate, g.sustain_time, g.target_amplitude),
                id(self) if self.carry_history else 0)

    def demod(self, input_audio, device_out=False, signs=False):   # afsk_pll.py:140-170
        a = self.front_end(input_audio)
        ctx = self._ctx
        d = ctx.scratch((self._key(), "loop"), a.n, np.float64)
        if a.n:
            check(lib().pm_pll_afsk(ctx.handle, ctypes.byref(self._loop), 1, self._const("wavetable", self.wavetable).ptr,
                                    a.ptr, 0, a.n, d.ptr, a.n))
        return self.back_end(d, device_out, signs)

    def back_end(self, d, device_out=False, signs=False):
        """Output low-pass on the loop's proportional term (afsk_pll.py:168)."""
        self._context()
        d = self._carry("loop", d, len(self.output_lpf) - 1)
        if signs:
            bits, nout = self._fir_carried(d, False, "output_lpf", self.output_lpf, signs=True)
            return SignBits(bits, None, nout)
        y = self._fir_carried(d, False, "output_lpf", self.output_lpf)
        return self._finish(y, device_out)

    def demod_signs(self, input_audio):
        return self.demod(input_audio, signs=True)
