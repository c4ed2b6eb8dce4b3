"""Seeded test-signal generator: the inverse of the chain.  The reference is decode-only and 11 of its
12 recordings are missing, so packet-bearing inputs for AFSK-1200, FSK-9600, BPSK and QPSK chains come from here.

    frames (AX.25 UI)  ->  link layer: AX.25 HDLC (flags, bit stuffing) | IL2P (header, scrambling, RS parity, sync, Hamming CRC)
                       ->  inverse of the chain's stream stage (NRZI / G3RUH / inversion = LFSR run backwards)
                       ->  modulator (AFSK | baseband FSK | BPSK | differential QPSK)  ->  + seeded AWGN -> int16

Every encoder here is the inverse of a decoder in the reference: AX25Codec.decode ax25.py:25-93, IL2PCodec.decode
il2p.py:214-518 (header layout :214-290, block sizes :346-358, Hamming CRC :503-518), RS/GF rs_functions.py:9-31,
gf_functions.py:47-74, LFSR.stream_unscramble_8bit lfsr.py:22-52, QuadratureSlicer demap slicer.py:124-165.
Plain NumPy on the host: this is a test fixture tool, not part of the timed path.
"""
import numpy as np

from . import taps as T


# ---- bits / CRC ------------------------------------------------------------------------------------------------
def crc16(data):
    crc = 0xFFFF
    for byte in data:
        byte = int(byte)
        for _ in range(8):
            crc = (crc >> 1) ^ 0x8408 if (crc ^ byte) & 1 else crc >> 1
            byte >>= 1
    return crc ^ 0xFFFF


def ax25_ui_frame(dest, src, info, dest_ssid=0, src_ssid=0, pid=0xF0, command=True):
    """AX.25 UI frame bytes without FCS: shifted callsigns, SSID bytes as the reference's IL2P path rebuilds them
    (il2p.py:292-340), control 0x03, PID, info."""
    def call(c):
        return [(ord(ch) << 1) for ch in c.upper().ljust(6)[:6]]
    out = call(dest) + [(dest_ssid << 1) + 0x60 + (0x80 if command else 0)]
    out += call(src) + [(src_ssid << 1) + 0x60 + (0 if command else 0x80) + 1]
    out += [0x03, pid]
    return out + [int(b) for b in info]


def bytes_to_bits_lsb_first(data):
    return [(b >> i) & 1 for b in data for i in range(8)]


def bytes_to_bits_msb_first(data):
    return [(b >> (7 - i)) & 1 for b in data for i in range(8)]


def ax25_hdlc_bits(frame, pre_flags=20, post_flags=4):
    """Flags + bit-stuffed frame+FCS (bytes LSB first, ax25.py:33-57) + flags."""
    c = crc16(frame)
    body = bytes_to_bits_lsb_first(list(frame) + [c & 0xFF, c >> 8])
    stuffed, ones = [], 0
    for b in body:
        stuffed.append(b)
        ones = ones + 1 if b else 0
        if ones == 5:
            stuffed.append(0)
            ones = 0
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    return flag * pre_flags + stuffed + flag * post_flags


# ---- GF(256) / RS encoder (generator = prod (x + a^i), i = 0..roots-1, field 0x11D) ---------------------------------
def _gf_tables():
    exp, log = [0] * 255, [0] * 256
    reg = 1
    for i in range(254, -1, -1):                     # gf_functions.py:60-64 steps the field from the top down
        fb = reg & 1
        reg >>= 1
        if fb:
            reg ^= 0x11D >> 1
        exp[i], log[reg] = reg, i
    return exp, log


_EXP, _LOG = _gf_tables()


def _gmul(a, b):
    return 0 if a == 0 or b == 0 else _EXP[(_LOG[a] + _LOG[b]) % 255]


def rs_parity(msg, roots):
    g = [_EXP[0], 1]
    for i in range(1, roots):
        f = [_EXP[i], 1]
        r = [0] * (len(g) + 1)
        for a in range(len(g)):
            for b in range(2):
                r[a + b] ^= _gmul(g[a], f[b])
        g = r
    rem = [0] * roots
    for byte in msg:
        fb = byte ^ rem[roots - 1]
        for k in range(roots - 1, 0, -1):
            rem[k] = rem[k - 1] ^ _gmul(fb, g[k])
        rem[0] = _gmul(fb, g[0])
    return rem[::-1]


# ---- LFSR run backwards: the transmit side of the chain's stream stage -------------------------------------------------
def lfsr_scramble(bits, poly, invert, register=0):
    """Bits b such that LFSR(poly, invert).stream_unscramble_8bit(b) == `bits` (lfsr.py:30-51 inverted: the descrambler's
    output bit is in ^ (register & 1) because bit 0 of every polynomial in use is set)."""
    assert poly & 1
    reg, out = register, []
    for want in bits:
        w = want ^ 1 if invert else want
        b = w ^ (reg & 1)
        if b:
            reg ^= poly
        reg >>= 1
        out.append(b)
    return out


# ---- IL2P encoder ---------------------------------------------------------------------------------------------------
_HAMMING_ENCODE = [0x0, 0x71, 0x62, 0x13, 0x54, 0x25, 0x36, 0x47, 0x38, 0x49, 0x5A, 0x2B, 0x6C, 0x1D, 0x0E, 0x7F]
_PID_TO_IL2P = {0x10: 2, 0x01: 3, 0x06: 4, 0x07: 5, 0x08: 6, 0xC3: 7, 0xC4: 8, 0xCA: 9, 0xCB: 10, 0xCC: 11, 0xCD: 12, 0xCE: 13,
                0xCF: 14, 0xF0: 15}


def _il2p_scramble(data):
    bits = lfsr_scramble(bytes_to_bits_msb_first(data), 0x211, False, register=0x1F0)      # il2p.py:160-163
    return [int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8)]


def il2p_frame_bits(dest, src, info, dest_ssid=0, src_ssid=0, pid=0xF0, command=True, trailing_crc=True, preamble=16):
    """IL2P type-1 (translated AX.25 UI) frame as MSB-first bits: preamble 0x55.., sync 0xF15E48, 13+2 header, payload blocks
    (+16 RS each, big blocks first), 4 Hamming(7,4) bytes carrying the CRC of the rebuilt AX.25 frame."""
    count = len(info)
    assert 0 <= count <= 1023
    hdr = [0] * 13
    d, s = dest.upper().ljust(6)[:6], src.upper().ljust(6)[:6]
    for i in range(6):
        hdr[i] |= (ord(d[i]) - 0x20) & 0x3F
        hdr[i + 6] |= (ord(s[i]) - 0x20) & 0x3F
    hdr[12] = (dest_ssid << 4) | src_ssid
    hdr[0] |= 0x40                                   # UI frame
    hdr[1] |= 0x80                                   # header type 1
    for i in range(10):
        if count & (0x200 >> i):
            hdr[i + 2] |= 0x80
    il2p_pid = _PID_TO_IL2P[pid]
    for i in range(4):
        if il2p_pid & (0x8 >> i):
            hdr[i + 1] |= 0x40
    control = (5 << 3) | (0x4 if command else 0)     # UI opcode 5 -> control byte 0x03 (il2p.py:92), C bit
    for i in range(7):
        if control & (0x40 >> i):
            hdr[i + 5] |= 0x40
    sh = _il2p_scramble(hdr)
    out = [0x55] * preamble + [0xF1, 0x5E, 0x48] + sh + rs_parity(sh, 2)
    if count:
        nblocks = -(-count // 239)
        small = count // nblocks
        big = count - nblocks * small
        pos = 0
        for k in range(nblocks):
            size = small + 1 if k < big else small
            blk = _il2p_scramble([int(b) for b in info[pos:pos + size]])
            pos += size
            out += blk + rs_parity(blk, 16)
    if trailing_crc:
        c = crc16(ax25_ui_frame(dest, src, info, dest_ssid, src_ssid, pid, command))
        out += [_HAMMING_ENCODE[(c >> (12 - 4 * i)) & 0xF] for i in range(4)]
    return bytes_to_bits_msb_first(out)


# ---- modulators -----------------------------------------------------------------------------------------------------------
def _symbol_edges(nsym, rate, baud):
    return np.floor(np.arange(nsym + 1) * (rate / baud) + 0.5).astype(np.int64)


def afsk(bits, rate, baud, mark, space, amplitude=8000.0):
    """Phase-continuous AFSK: 1 -> mark tone, 0 -> space tone."""
    edges = _symbol_edges(len(bits), rate, baud)
    freq = np.empty(edges[-1])
    for k, b in enumerate(bits):
        freq[edges[k]:edges[k + 1]] = mark if b else space
    phase = 2.0 * np.pi * np.cumsum(freq) / rate
    return amplitude * np.cos(phase)


def baseband_fsk(bits, rate, baud, amplitude=8000.0, bt_taps=None):
    """Discriminator-output style baseband (what fsk.py expects): +-1 levels through a short raised-cosine smoother."""
    edges = _symbol_edges(len(bits), rate, baud)
    lv = np.empty(edges[-1])
    for k, b in enumerate(bits):
        lv[edges[k]:edges[k + 1]] = 1.0 if b else -1.0
    n = max(3, int(round(rate / baud)) | 1) if bt_taps is None else bt_taps
    w = np.hanning(n + 2)[1:-1]
    return amplitude * np.convolve(lv, w / w.sum(), "same")


def _shaped(symbols, rate, baud, rolloff, span=6):
    sps = rate / baud
    n = int(np.floor(len(symbols) * sps)) + 1
    imp = np.zeros(n)
    idx = np.floor(np.arange(len(symbols)) * sps + 0.5).astype(np.int64)
    imp[idx] = symbols
    h = T.root_raised_cosine(rate, baud, span, rolloff)
    return np.convolve(imp, h / np.max(np.abs(h)), "same")


def bpsk(bits, rate, baud, carrier, rolloff, amplitude=8000.0):
    i = _shaped(np.where(np.asarray(bits) > 0, 1.0, -1.0), rate, baud, rolloff)
    t = np.arange(len(i))
    return amplitude * i * np.cos(2.0 * np.pi * carrier * t / rate)


_QPSK_DEMAP = [3, 1, 2, 0, 2, 3, 0, 1, 1, 0, 3, 2, 0, 2, 1, 3]          # slicer.py:128: demap[(prev << 2) | cur] -> dibit


def qpsk(bits, rate, baud, carrier, rolloff, amplitude=8000.0, conj=False):
    """Differential QPSK matching QuadratureSlicer: quadrant q = (I>=0)<<1 | (Q>=0); each dibit picks the next quadrant so
    that demap[(prev<<2)|cur] returns it.  `baud` is the symbol rate (half the bit rate)."""
    bits = list(bits) + [0] * (len(bits) % 2)
    prev, quads = 0, []
    for k in range(0, len(bits), 2):
        want = (bits[k] << 1) | bits[k + 1]
        cur = next(c for c in range(4) if _QPSK_DEMAP[(prev << 2) | c] == want)
        quads.append(cur)
        prev = cur
    qi = np.array([1.0 if q & 2 else -1.0 for q in quads])
    qq = np.array([1.0 if q & 1 else -1.0 for q in quads])
    i, q = _shaped(qi, rate, baud, rolloff), _shaped(qq, rate, baud, rolloff)
    t = 2.0 * np.pi * carrier * np.arange(len(i)) / rate
    sgn = 1.0 if conj else -1.0
    return amplitude * (i * np.cos(t) + sgn * q * np.sin(t)) / np.sqrt(2.0)


# ---- a whole recording ---------------------------------------------------------------------------------------------------
def recording(mode, rate=48000, packets=6, seed=1, noise_sigma=600.0, gap_s=0.25, payload_len=(20, 120), **kw):
    """Seeded int16 recording with `packets` frames and AWGN.  mode:
       'afsk1200_ax25' | 'afsk1200_il2p' | 'fsk9600_ax25' | 'fsk9600_il2p' | 'bpsk300_il2p' | 'bpsk1200_il2p' | 'qpsk2400_il2p' |
       'qpsk600_il2p' | 'qpsk3600_il2p' | 'afsk300_il2p'.
    Returns (int16 samples, list of AX.25 frames WITHOUT FCS that were sent)."""
    rng = np.random.default_rng(seed)
    frames, parts = [], []
    gap = np.zeros(int(gap_s * rate))
    for k in range(packets):
        n = int(rng.integers(payload_len[0], payload_len[1] + 1))
        info = [int(c) for c in rng.integers(32, 127, n)]
        frame = ax25_ui_frame("CQ", f"N0CAL{k % 10}", info, src_ssid=k % 16)
        frames.append(frame)
        link = mode.split("_")[1]
        if link == "ax25":
            bits = ax25_hdlc_bits(frame)
        else:
            bits = il2p_frame_bits("CQ", f"N0CAL{k % 10}", info, src_ssid=k % 16) + [0, 1] * 8
        if mode.startswith("afsk1200"):
            poly, inv = (0x3, True) if link == "ax25" else (0x1, False)
            x = afsk(lfsr_scramble(bits, poly, inv), rate, 1200, kw.get("mark", 1200.0), kw.get("space", 2200.0))
        elif mode.startswith("afsk300"):
            poly, inv = (0x3, True) if link == "ax25" else (0x1, False)
            x = afsk(lfsr_scramble(bits, poly, inv), rate, 300, kw.get("mark", 1600.0), kw.get("space", 1800.0))
        elif mode.startswith("fsk9600"):
            poly, inv = (0x63003, True) if link == "ax25" else (0x1, False)
            x = baseband_fsk(lfsr_scramble([0, 1] * 40 + bits, poly, inv), rate, 9600)
        elif mode.startswith("bpsk"):
            baud = 300 if mode.startswith("bpsk300") else 1200
            x = bpsk(lfsr_scramble([0, 1] * 60 + bits, 0x3, True), rate, baud, kw.get("carrier", 1500.0), 0.6 if baud == 300 else 0.9)
        elif mode.startswith("qpsk"):
            brate = int(mode[4:].split("_")[0])
            roll = {600: 0.6, 2400: 0.9, 3600: 0.3}[brate]
            x = qpsk([0, 1, 1, 0] * 60 + bits, rate, brate // 2, kw.get("carrier", 1650.0 if brate == 3600 else 1500.0), roll,
                     conj=kw.get("conj", False))
        else:
            raise ValueError(mode)
        parts += [gap, x]
    parts.append(gap)
    sig = np.concatenate(parts)
    sig = sig + rng.standard_normal(len(sig)) * noise_sigma
    return np.clip(np.rint(sig), -32768, 32767).astype(np.int16), frames
