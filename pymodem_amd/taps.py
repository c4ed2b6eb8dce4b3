"""Host-side tap design (float64, once per chain).  The kernels take tap vectors as plain arrays, which keeps
filter design a pure-host concern.  Results are bit-identical to the reference's own
designers on the same NumPy build (tests/test_host_taps.py pins them to tests/golden/taps.npz).

Reference: scipy.signal.firwin call sites afsk.py:112-126, fsk.py:133-138, psk.py:118-124,650-656,
afsk_pll.py:92-108; RRC.tune rrc.py:18-95; Hilbert.__init__ hilbert.py:9-34; AFSK tones afsk.py:134-144;
NCO wavetable nco.py:22-24; IIR_1 coefficients iir.py:15-29; phase-detector table phase_detector.py:36-44.
"""
import math

import numpy as np

_HAMMING_A = (0.54, 1.0 - 0.54)


def hamming_window(count):
    """Symmetric Hamming window, a0 + a1*cos(t), t on a closed grid from -pi to pi."""
    if count == 1:
        return np.ones(1)
    grid = np.linspace(-np.pi, np.pi, count)
    win = np.zeros(count)
    for order, coef in enumerate(_HAMMING_A):
        win += coef * np.cos(order * grid)
    return win


def windowed_sinc(numtaps, cutoff, fs, pass_zero):
    """Hamming-windowed sinc FIR: `cutoff` is a scalar (low-pass) or a sequence of band edges in Hz; with
    pass_zero False the first band starts at the first edge (band-pass).  Unity gain at DC or at the centre
    of the first pass band.  Same arithmetic as scipy.signal.firwin(..., window='hamming', scale=True)."""
    numtaps = int(numtaps)
    edges = np.atleast_1d(np.asarray(cutoff, dtype=np.float64)) / float(0.5 * fs)
    if edges.size == 0 or edges.min() <= 0 or edges.max() >= 1:
        raise ValueError("Invalid cutoff frequency: frequencies must be greater than 0 and less than fs/2.")
    if np.any(np.diff(edges) <= 0):
        raise ValueError("Invalid cutoff frequencies: the frequencies must be strictly increasing.")
    through_nyquist = bool(edges.size & 1) ^ bool(pass_zero)
    if through_nyquist and numtaps % 2 == 0:
        raise ValueError("A filter with an even number of coefficients must have zero response at the Nyquist frequency.")
    full = np.hstack(([0.0] * bool(pass_zero), edges, [1.0] * through_nyquist)).reshape(-1, 2)
    centre = 0.5 * (numtaps - 1)
    m = np.arange(0, numtaps) - centre
    taps = 0
    for lo, hi in full:
        taps += hi * np.sinc(hi * m)
        taps -= lo * np.sinc(lo * m)
    taps = taps * hamming_window(numtaps)
    lo, hi = full[0]
    ref_freq = 0.0 if lo == 0 else (1.0 if hi == 1 else 0.5 * (lo + hi))
    gain = np.sum(taps * np.cos(np.pi * m * ref_freq))
    return taps / gain


_RRC_WINDOWS = {
    "blackmann": (0.355768, 0.487396, 0.144232, 0.012604),
    "blackmann-harris": (0.35875, 0.48829, 0.14128, 0.01168),
    "flattop": (0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368),
}


def _rrc_window(name, count):
    last = count - 1
    if name == "rect":
        return [1] * count
    if name == "hann":
        return [np.power(np.sin(np.pi * k / last), 2) for k in range(count)]
    if name in _RRC_WINDOWS:
        a = _RRC_WINDOWS[name]
        out = []
        for k in range(count):
            v = a[0] - (a[1] * np.cos(2 * np.pi * k / last)) + (a[2] * np.cos(4 * np.pi * k / last)) - (a[3] * np.cos(6 * np.pi * k / last))
            if len(a) > 4:
                v = v + (a[4] * np.cos(8 * np.pi * k / last))
            out.append(v)
        return out
    if name == "tukey":
        a, out, k = 0.25, [], 0
        while k < a * last / 2:
            out.append(0.5 * (1 - np.cos(2 * np.pi * k / (a * last))))
            k += 1
        while k <= last // 2:
            out.append(1)
            k += 1
        while k <= last:
            out.append(out[last - k])
            k += 1
        return out
    raise ValueError(f"unknown RRC window {name!r}")


def root_raised_cosine(sample_rate, symbol_rate, symbol_span, rolloff_rate, window="rect"):
    """Root-raised-cosine taps, L2-normalised, on the reference's time grid (rrc.py:18-48) and window (rrc.py:51-93)."""
    oversample = sample_rate / symbol_rate
    count = int(round(symbol_span * oversample, 0)) + 1
    dt = 1 / sample_rate
    ts = 1 / symbol_rate
    grid = np.arange(0, count * dt, dt) - (count * dt / 2) + (dt / 2)
    count = len(grid)          # float arange may add a point (rrc.py:23-24)
    singular = ts / (4 * rolloff_rate)
    taps = np.empty(count)
    for k, t in enumerate(grid):
        if math.isclose(t, -singular) or math.isclose(t, singular):
            top = rolloff_rate * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * rolloff_rate)) + (1 - (2 / np.pi)) * np.cos(np.pi / (4 * rolloff_rate)))
            taps[k] = top / (ts * pow(2, 0.5))
        else:
            top = np.sin(np.pi * t * (1 - rolloff_rate) / ts) + 4 * rolloff_rate * t * np.cos(np.pi * t * (1 + rolloff_rate) / ts) / ts
            bottom = np.pi * t * (1 - pow(4 * rolloff_rate * t / ts, 2)) / ts
            taps[k] = top / (bottom * ts)
    taps = list(taps) / np.linalg.norm(list(taps))
    return np.multiply(taps, _rrc_window(window, count))


def hilbert_transformer(tap_count):
    """Hann-windowed Hilbert FIR (odd length).  Returns (taps, delay); the matching delay line is a pure shift."""
    delay = tap_count // 2
    ideal = [2 / (math.pi * n) if n % 2 else 0 for n in range(-delay, tap_count - delay)]
    last = tap_count - 1
    return np.array([ideal[k] * (math.sin(math.pi * k / last) ** 2) for k in range(tap_count)], dtype=np.float64), delay


def afsk_tone_correlators(sample_rate, symbol_rate, mark_freq, space_freq, space_gain, correlator_span, correlator_offset):
    """cos/sin templates of the mark and space tones over `correlator_span` symbols (afsk.py:134-144)."""
    n = np.arange(math.ceil(correlator_span * sample_rate / symbol_rate))
    mark = n * (2.0 * np.pi * (mark_freq + correlator_offset) / sample_rate)
    space = n * (2.0 * np.pi * (space_freq + correlator_offset) / sample_rate)
    return np.cos(mark), np.sin(mark), space_gain * np.cos(space), space_gain * np.sin(space)


def tone_model(template_i, template_q):
    """How well a correlator template pair is the powers of one rotation, template[j] = (r^j).real / .imag with
    r = (template_i[1], template_q[1]) -> (rot, end, dev): r, r^m rounded to double, and the largest |template[j] - r^j| over both
    templates, the powers taken in extended precision (pm_afsk_tones; None if the pair is shorter than two taps)."""
    hi, hq = np.asarray(template_i, dtype=np.float64), np.asarray(template_q, dtype=np.float64)
    m = len(hi)
    if m < 2 or len(hq) != m:
        return None
    ld = np.longdouble
    rr, ri = ld(hi[1]), ld(hq[1])
    zr, zi, dev = ld(1.0), ld(0.0), ld(0.0)
    for j in range(m):
        dev = max(dev, abs(zr - ld(hi[j])), abs(zi - ld(hq[j])))
        zr, zi = zr * rr - zi * ri, zr * ri + zi * rr
    # the extended-precision powers themselves are off by a few 2^-64 per step; count that in
    dev = float(dev) + 4.0 * m * float(np.finfo(ld).eps)
    return (float(hi[1]), float(hq[1])), (float(zr), float(zi)), dev


def sine_wavetable(amplitude=1.0, size=256):
    return np.array([amplitude * math.sin(k * 2.0 * math.pi / size) for k in range(size)], dtype=np.float64)


def one_pole_lowpass(sample_rate, cutoff, gain):
    """Bilinear one-pole low-pass (b0, b1, a1), gain folded into b (iir.py:15-29)."""
    warped = 2.0 * sample_rate * math.tan((2.0 * math.pi * cutoff) / (2.0 * sample_rate))
    wt = warped / sample_rate
    b = wt / (2.0 + wt)
    return gain * b, gain * b, (2.0 - wt) / (2.0 + wt)


def qpsk_error_table(granularity=64, gain=32):
    """Integer phase-error table over the first quadrant, zero outside the 15 %..76 % magnitude ring."""
    table = np.zeros((granularity, granularity), dtype=np.int32)
    lo, hi = granularity * .15, granularity * .76
    for re in range(granularity):
        for im in range(granularity):
            if lo <= math.sqrt((re ** 2) + (im ** 2)) <= hi:
                table[re, im] = round(gain * ((math.atan2(im, re) * 180 / math.pi) - 45))
    return table
