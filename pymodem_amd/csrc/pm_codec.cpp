// Host-integer stages of the demod_chain path, native C++ (no GPU): the byte stream the slicer produces
// is KBs per minute of audio, and these are bit-serial state machines.
//
//   pm_lfsr_unscramble   LFSR.stream_unscramble_8bit          lfsr.py:22-52
//   pm_codec_* (kind 0)  AX25Codec.decode                     ax25.py:25-93
//   pm_codec_* (kind 1)  IL2PCodec.decode + RS + GF + Hamming  il2p.py:110-519, rs_functions.py:33-150, gf_functions.py
//   pm_crc16_ccitt       CheckCRC / AppendCRC                  crc_functions.py:9-76
//   pm_correlate         PacketMetaArray.Correlate             packet_meta.py:230-271
//
// Behaviour follows the reference including its quirks (noted inline), because packet parity is
// "identical bytes, CRCs and stream addresses", not "a better decoder".
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "../../include/pymodem_amd.h"

int pm_set_error(int code, const char *fmt, ...);

namespace {

// ---- CRC-16 (reflected CCITT 0x8408, init 0xFFFF, final xor 0xFFFF) -----------------------------
// crc_functions.py:44-55, bit by bit there; here eight bits per table step (the same reflected polynomial 0x8408).
int crc16(const uint8_t *d, int64_t n)
{
    static const struct Table {
        uint16_t t[256];
        Table()
        {
            for (unsigned v = 0; v < 256; ++v) {
                unsigned c = v;
                for (int i = 0; i < 8; ++i) c = (c & 1) ? (c >> 1) ^ 0x8408 : c >> 1;
                t[v] = (uint16_t)c;
            }
        }
    } table;
    unsigned crc = 0xFFFF;
    for (int64_t k = 0; k < n; ++k) crc = (crc >> 8) ^ table.t[(crc ^ d[k]) & 0xFF];
    return (int)(crc ^ 0xFFFF);
}

// packet_meta.py:21-41.  The loop there never resets its sub-field index, so only bytes 0..6 are examined.
bool valid_header(const uint8_t *d, int len)
{
    if (len <= 15) return false;
    for (int k = 0; k < 7; ++k) {
        const int ch = d[k] >> 1;
        if ((ch < 32 || ch > 126) && ch != 0) return false;
    }
    return true;
}

void finalize(pm_packet &p)
{
    // PacketMeta.CalcCRC / Validate, packet_meta.py:197-208 (needs len >= 2, which every emitter guarantees)
    p.carried_crc = p.len >= 2 ? p.data[p.len - 1] * 256 + p.data[p.len - 2] : 0;
    p.calculated_crc = p.len >= 2 ? crc16(p.data, p.len - 2) : 0;
    p.valid_crc = p.len >= 2 && p.carried_crc == p.calculated_crc;
    p.valid_header = valid_header(p.data, p.len);
    p.correlated_count = 0;
}

// Decoded packets wait here until pm_codec_fetch takes them: their bytes one behind the other in ONE block (a vector per packet was an
// allocation and a free per packet: 5800 of each per ten-minute recording of the headline config)
struct Queued {
    int64_t addr;
    int corrected;
    size_t off, len;                 // bytes[off .. off + len)
};

struct Sink {
    std::vector<Queued> q;
    std::vector<uint8_t> bytes;
    void push(const std::vector<uint8_t> &data, int64_t addr, int corrected, int /*source*/)
    {
        q.push_back(Queued{addr, corrected, bytes.size(), data.size()});
        bytes.insert(bytes.end(), data.begin(), data.end());
    }
    void drop_front(size_t take)
    {
        q.erase(q.begin(), q.begin() + (ptrdiff_t)take);
        if (q.empty()) bytes.clear();                        // (offsets of what is left stay valid otherwise)
    }
};

// ---- GF(2^8) / Reed-Solomon ----------------------------------------------------------------------
struct GF256 {
    int table[255], index[256], inverse[256];
    GF256()
    {
        // gf_functions.py:47-74: a Galois LFSR stepped from a^0, filling the table from the top down
        unsigned reg = 1;
        memset(index, 0, sizeof(index));
        for (int i = 254; i >= 0; --i) {
            const unsigned fb = reg & 1;
            reg >>= 1;
            if (fb) reg ^= 0x11D >> 1;
            table[i] = (int)reg;
            index[reg] = i;
        }
        inverse[0] = 0;
        for (int i = 1; i < 256; ++i) {
            int j = 1;
            while (mul(i, j) != 1) ++j;
            inverse[i] = j;
        }
    }
    int mul(int a, int b) const
    {
        if (a == 0 || b == 0) return 0;
        int r = index[a] + index[b];
        while (r > 254) r -= 255;
        return table[r];
    }
    // row[i][v] = v * table[i] for the 16 generator roots table[0..15]: one lookup per byte in the Horner syndromes
    uint8_t row[16][256];
    void build_rows()
    {
        for (int i = 0; i < 16; ++i)
            for (int v = 0; v < 256; ++v) row[i][v] = (uint8_t)mul(v, table[i]);
    }
};

const GF256 &gf()
{
    static const GF256 g = [] { GF256 t; t.build_rows(); return t; }();
    return g;
}

int wrap255(int x)
{
    while (x > 254) x -= 255;
    return x;
}

// rs_functions.py:33-150 (first_root is 0 for both IL2P codes).  Corrects buf[0..n) in place.
int rs_decode(int num_roots, uint8_t *buf, int n, int min_distance)
{
    const GF256 &g = gf();
    const int first_root = 0, half = num_roots / 2;
    int syn[16];
    auto syndromes = [&]() {
        // Horner in every root at once: byte by byte, the roots' chains side by side (each step of one chain is a table lookup that
        // waits for the one before it: root by root that was 16 x n dependent loads, 2 us for a 100-byte block)
        unsigned v[16] = {0};
        if (num_roots == 16) {
            for (int j = 0; j < n - 1; ++j) {
                const unsigned b = buf[j];
                for (int i = 0; i < 16; ++i) v[i] = g.row[first_root + i][v[i] ^ b];     // row[i]: multiplication by the i-th root
            }
        } else {
            for (int j = 0; j < n - 1; ++j) {
                const unsigned b = buf[j];
                for (int i = 0; i < num_roots; ++i) v[i] = g.row[first_root + i][v[i] ^ b];
            }
        }
        for (int i = 0; i < num_roots; ++i) syn[i] = (int)(v[i] ^ buf[n - 1]);
    };
    syndromes();
    {
        // All syndromes zero: the locator stays {1}, the Chien search finds no root (x = loc[0] = 1), nothing is corrected and
        // the closing check passes -- the reference's algorithm returns 0 with the data untouched.  Skip straight there.
        bool clean = true;
        for (int i = 0; i < num_roots; ++i) clean &= syn[i] == 0;
        if (clean) return 0;
    }
    int loc[17] = {0}, nxt[17] = {0}, corr[18] = {0}, where[17] = {0};
    loc[0] = 1;
    corr[1] = 1;
    int order = 0;
    for (int step = 1; step <= num_roots; ++step) {          // Berlekamp
        const int y = step - 1;
        int e = syn[y];
        for (int i = 1; i <= order; ++i) e ^= g.mul(loc[i], syn[y - i]);
        if (e != 0) {
            for (int i = 0; i <= order; ++i) nxt[i] = loc[i] ^ g.mul(e, corr[i]);
            e = g.inverse[e];
            for (int i = 0; i <= half; ++i) corr[i] = g.mul(loc[i], e);
            for (int i = 0; i <= half; ++i) loc[i] = nxt[i];
        }
        if (2 * order < step) order = step - order;
        for (int i = num_roots; i > 0; --i) corr[i] = corr[i - 1];
        corr[0] = 0;
    }
    int count = 0;
    // Chien search.  The exponent of term i at position j is ((j + 256 - n) i + index[loc[i]]) mod 255: kept per term and advanced by i
    // from one position to the next instead of being reduced from scratch (wrap255 subtracts 255 up to eight times per term)
    int ex[9], ni = 0, which[9];
    for (int i = 1; i <= half; ++i)
        if (loc[i]) {
            ex[ni] = wrap255((256 - n) * i + g.index[loc[i]]);
            which[ni++] = i;
        }
    for (int j = 0; j < n; ++j) {
        int x = 0;
        for (int k = 0; k < ni; ++k) {
            x ^= g.table[ex[k]];
            ex[k] += which[k];
            if (ex[k] > 254) ex[k] -= 255;
        }
        x ^= loc[0];
        if (x == 0) {
            if (count < 17) where[count] = j;
            ++count;
        }
    }
    if (count <= half - min_distance) {                      // Forney
        for (int i = 0; i < count; ++i) {
            corr[i] = syn[first_root + i];
            for (int j = 1; j <= i; ++j) corr[i] ^= g.mul(syn[first_root + i - j], loc[j]);
        }
        for (int i = 0; i < count; ++i) {
            const int e = n - where[i] - 1;
            int z = corr[0];
            for (int j = 1; j < count; ++j) {
                int x = wrap255(e * j);
                x = wrap255(256 - x - 1);
                z ^= g.mul(corr[j], g.table[x]);
            }
            z = g.mul(z, g.table[e]);
            int y = loc[1];
            for (int j = 3; j <= half; j += 2) {
                int x = wrap255(e * (j - 1));
                x = wrap255(256 - x - 1);
                y ^= g.mul(loc[j], g.table[x]);
            }
            y = g.index[y];
            y = 256 - y - 1;
            if (y == 255) y = 0;
            y = g.table[y];
            buf[where[i]] ^= (uint8_t)g.mul(y, z);
        }
    }
    syndromes();
    for (int i = 0; i < num_roots; ++i)
        if (syn[i] != 0) return -1;
    return count;
}

}  // namespace

// ---- codecs ------------------------------------------------------------------------------------------
struct pm_codec {
    Sink sink;
    int source = 0;
    virtual ~pm_codec() {}
    virtual void feed(uint8_t byte, int64_t addr, Sink &sink) = 0;
    virtual void feed_many(const uint8_t *d, const int64_t *a, int64_t n)
    {
        for (int64_t k = 0; k < n; ++k) feed(d[k], a[k], sink);
    }
};

namespace {

// Per (consecutive-ones count on entry, capped at 7, and input byte): what the eight bits do to the decoder, as at most four steps --
// "append these de-stuffed bits", "flag", "k ones past the sixth (abort)" -- in bit order.  It is the bit-serial code below run on
// the part of the state that the bit pattern alone decides (the ones counter); what depends on the rest (bit and byte counters, the
// collected bytes) happens when the steps are carried out.  Streams between packets are far from random bits (long runs of ones on
// the chains with a strong space gain), so "no flag or abort inside this byte" is not the common case it is on a clean signal.
struct Ax25Step {
    uint8_t op, bits;                                    // op: kind << 4 | n; kind 0 append n bits (first appended = bit 0 of `bits`), 1 flag, 2 abort of n ones
};
struct Ax25Entry {
    uint8_t nsteps, ones_out;                            // nsteps 0xFF: more than three steps -> bit-serial
    Ax25Step step[3];
};

const Ax25Entry *ax25_table()
{
    static Ax25Entry t[8][256];                          // 16 KB
    static std::once_flag once;
    std::call_once(once, [] {
        for (int ones0 = 0; ones0 < 8; ++ones0)
            for (int byte = 0; byte < 256; ++byte) {
                Ax25Entry e;
                memset(&e, 0, sizeof(e));
                int ones = ones0, ns = 0, kind[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
                bool over = false;
                auto last = [&](int k) -> int {
                    if (ns && kind[ns - 1] == k && k != 1) return ns - 1;
                    if (ns == 3) { over = true; return -1; }
                    kind[ns] = k;
                    return ns++;
                };
                for (int i = 0; i < 8 && !over; ++i) {
                    const int bit = (byte >> (7 - i)) & 1;
                    if (bit) {
                        if (ones < 7) ++ones;
                        if (ones > 6) {                              // seventh one and beyond: shifted in, counters reset
                            const int q = last(2);
                            if (q >= 0) ++cnt[q];
                        } else {
                            const int q = last(0);
                            if (q >= 0) {
                                e.step[q].bits |= (uint8_t)(1u << cnt[q]);
                                ++cnt[q];
                            }
                        }
                    } else {
                        if (ones < 5) {
                            const int q = last(0);                   // a data zero
                            if (q >= 0) ++cnt[q];
                        } else if (ones == 6) {
                            last(1);                                 // flag
                        }                                            // ones == 5: stuffed zero, dropped; ones > 6: nothing
                        ones = 0;
                    }
                }
                for (int q = 0; q < ns; ++q) e.step[q].op = (uint8_t)(kind[q] << 4 | cnt[q]);
                e.nsteps = over ? 0xFF : (uint8_t)ns;
                e.ones_out = (uint8_t)ones;
                t[ones0][byte] = e;
            }
    });
    return &t[0][0];
}

struct Ax25 : pm_codec {
    unsigned wb = 0;
    int nbytes = 0, ones = 0, nbits = 0;
    std::vector<uint8_t> data;
    static constexpr int kMin = 18, kMax = 1023;           // ax25.py:14-15
    const Ax25Entry *table = ax25_table();
    bool skim_on;
    explicit Ax25(int src)
    {
        source = src;
        static const bool on = [] {
            const char *e = getenv("PM_AX25_SKIM");         // =0: every byte through the table-driven decoder (rounds 2-4), for A/B runs and the tests
            return !(e && e[0] == '0');
        }();
        skim_on = on;
    }

    static int trailing_ones(uint8_t b)
    {
        static const struct Trail {
            uint8_t v[256];
            Trail()
            {
                for (int b = 0; b < 256; ++b) {
                    int t = 0;
                    while (t < 8 && ((b >> t) & 1)) ++t;
                    v[b] = (uint8_t)(t < 7 ? t : 7);
                }
            }
        } trail;
        return trail.v[b];
    }

    // ---- the skim (round 5) ----------------------------------------------------------------------------------------------------------
    // What the decoder does with a stretch of bits is decided by three bit patterns: a zero behind exactly five ones is dropped
    // (ax25.py:69-71), a zero behind exactly six is a flag (:72-89), the seventh one of a run and every one after it clear the bit and byte
    // counters (:36-39) -- and the zero that ends such a run is not appended.  Flags and run ends are the decoder's RESETS; between two of
    // them every bit but the dropped zeros is appended.  So whether a flag closes a frame -- at least 18 bytes and 7 bits counted since the
    // last reset (:74-81) -- follows from positions and a count of dropped zeros, 64 bits at a time, and only a flag that DOES close one
    // (one in fourteen on noise) needs the bytes: the table-driven decoder below is then run from the flag before it, where the registers
    // are known (everything cleared; what the byte register still holds of older bits never reaches a completed byte).  The collected
    // bytes survive run ends (:36-39 clear counters only), which is why that run starts at the last FLAG, not at the last reset.
    // Exact as long as the byte counter does not pass its limit between two resets (it then clears the ones counter in mid-run, :41-50,
    // and the patterns stop telling what the decoder sees): 8000 bits without a reset and the rest of the call goes through the table.
    // One core of the build container, ns per byte: random bits 4.45 -> 2.29, four ones in five (the space-heavy chains between packets:
    // runs end everywhere) 12.3 -> 3.5, the headline's streams (two thirds of their bits inside frames, which need their bytes) 4.4 -> 4.5.
    static uint64_t load_bits(const uint8_t *p, int64_t have)
    {
        uint64_t x = 0;
        memcpy(&x, p, (size_t)(have < 8 ? have : 8));       // (little-endian host: the stream's first byte in bits 0..7)
        // the stream's bits are the bytes' from the top (ax25.py:30-31, :91-92): turned round inside each byte, bit i of x is stream bit i
        x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
        x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
        x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
        return x;
    }

    void reposition(const uint8_t *d, int64_t sb)
    {
        // byte sb holds a flag's closing zero: whatever the registers are in front of it, they are the decoder's own behind it -- and with
        // no byte counted the flag itself closes nothing
        wb = 0;
        nbytes = 0;
        nbits = 0;
        data.clear();
        ones = trailing_ones(d[sb - 1]);
    }

    void feed_many(const uint8_t *d, const int64_t *a, int64_t n) override
    {
        if (!skim_on || n < 24) {
            run(d, a, 0, n);
            return;
        }
        const int k = ones < 8 ? ones : 8;
        uint64_t prev = k >= 7 ? 0xFFull << 56 : (k ? ~0ull << (64 - k) : 0);       // the bits in front of the call: `ones` ones behind a zero
        int64_t R = 0, T0 = (int64_t)nbytes * 8 + nbits;      // R: first bit after the last reset; T0: bits counted before the call
        int64_t cumS = 0, sAtR = 0;                           // dropped zeros before this word / before R
        int64_t LF = -1;                                      // first bit after the last flag of this call
        int64_t cursor = 0;                                   // bytes the registers have taken
        bool first = true;
        auto take_from_last_flag = [&](int64_t upto) {
            if (LF >= 0) {
                const int64_t sb = (LF - 1) >> 3;
                if (sb >= cursor && sb > 0) {
                    reposition(d, sb);
                    cursor = sb;
                }
            }
            run(d, a, cursor, upto);
            cursor = upto;
        };
        for (int64_t w0 = 0; w0 < n; w0 += 8) {
            const int64_t have = n - w0;
            const uint64_t w = load_bits(d + w0, have);
            const uint64_t valid = have >= 8 ? ~0ull : (1ull << (8 * have)) - 1;
            const uint64_t e1 = (w << 1) | (prev >> 63), e2 = (w << 2) | (prev >> 62), e3 = (w << 3) | (prev >> 61), e4 = (w << 4) | (prev >> 60),
                           e5 = (w << 5) | (prev >> 59), e6 = (w << 6) | (prev >> 58), e7 = (w << 7) | (prev >> 57);
            const uint64_t o5 = e1 & e2 & e3 & e4 & e5;
            const uint64_t S = ~w & o5 & ~e6 & valid;           // dropped zeros
            uint64_t ev = ~w & o5 & e6 & valid;                 // zeros behind six ones or more: flags and run ends
            while (ev) {
                const int i = __builtin_ctzll(ev);
                ev &= ev - 1;
                const int64_t P = w0 * 8 + i;
                const int64_t span = T0 + (P - R);
                if (span >= 8000) {                          // the byte counter may have passed its limit: no more skimming in this call
                    take_from_last_flag(n);
                    return;
                }
                const int64_t sHere = cumS + __builtin_popcountll(S & ((1ull << i) - 1));
                if (!((e7 >> i) & 1)) {
                    const int64_t T = span - (sHere - sAtR);  // bits counted at the flag's zero
                    if (first || ((T & 7) == 7 && T >= 8 * kMin + 7)) take_from_last_flag((P >> 3) + 1);
                    first = false;
                    LF = P + 1;
                }
                R = P + 1;
                T0 = 0;
                sAtR = sHere;
            }
            cumS += __builtin_popcountll(S);
            prev = w;
        }
        // the registers as the call leaves them: the bytes since the last flag, the counters since the last reset
        take_from_last_flag(n);
    }

    // bytes [k0, k1) through the registers, eight bits at a time (round 2)
    void run(const uint8_t *d, const int64_t *a, int64_t k0, int64_t k1)
    {
        // the collected bytes as a raw buffer while this call runs (one slot of slack: a byte is stored whether or not it is
        // complete and the length moves on only if it is -- no branch on the data)
        size_t len = data.size();
        auto room = [&](size_t want) {
            if (data.size() < want) data.resize(std::max(want, 2 * data.size() + 64));
        };
        room(len + 2);
        uint8_t *buf = data.data();
        // The ones counter in front of a byte is the run of ones that ends the byte before it (capped at 7, which is all the table
        // distinguishes): a function of that byte alone, so the look-up of byte k + 1 does not wait for the entry of byte k -- the
        // chain through the table was what a byte cost (20 cycles; the other counters are one-cycle additions).  The one exception,
        // a completed byte clearing the counter at the length limit, goes through the bit-serial path, which hands its own count on.
        int ones_in = ones < 7 ? ones : 7;
        for (int64_t k = k0; k < k1; ++k) {
            const uint8_t byte = d[k];
            const Ax25Entry &e = table[ones_in * 256 + byte];
            // near the length limit a completed byte may clear the ones counter in mid-byte (byte_done): bit by bit there
            if (__builtin_expect(e.nsteps > 3 || nbytes >= kMax - 2, 0)) {
                data.resize(len);
                ones = ones_in;
                feed(byte, a[k], sink);
                ones_in = ones < 7 ? ones : 7;
                len = data.size();
                room(len + 2);
                buf = data.data();
                continue;
            }
            ones_in = trailing_ones(byte);
            if (__builtin_expect(e.nsteps == 1 && e.step[0].op < 16, 1)) {           // nine bytes in ten: eight bits' worth of appends
                const unsigned cnt = e.step[0].op;
                const unsigned x = (wb & 0x7F) | ((unsigned)e.step[0].bits << 7);
                const unsigned total = (unsigned)nbits + cnt;
                const unsigned done = total >> 3;
                buf[len] = (uint8_t)(x >> ((7 - nbits) & 7));
                len += done;
                nbytes += (int)done;
                nbits = (int)(total & 7);
                wb = (x >> cnt) & 0x7F;
                if (__builtin_expect(len + 2 > data.size(), 0)) {
                    room(len + 2);
                    buf = data.data();
                }
                continue;
            }
            for (int q = 0; q < e.nsteps; ++q) {
                const unsigned op = e.step[q].op, cnt = op & 15;
                if (op < 16) {
                    // wb holds the last 7 appended bits in bits 6..0 (newest at 6); x extends it with this step's bits
                    const unsigned x = (wb & 0x7F) | ((unsigned)e.step[q].bits << 7);
                    const unsigned total = (unsigned)nbits + cnt;                    // <= 15
                    const unsigned done = total >> 3;                                // 1: this step completes a byte
                    buf[len] = (uint8_t)(x >> ((7 - nbits) & 7));                    // the byte as of its eighth bit (used if done)
                    len += done;
                    nbytes += (int)done;                                             // cannot pass kMax here (guard above)
                    nbits = (int)(total & 7);
                    wb = (x >> cnt) & 0x7F;
                } else if (op < 32) {                                // flag (ax25.py:52-60)
                    if (nbytes >= kMin && nbits == 7) {
                        data.resize(len);
                        sink.push(data, a[k], 0, source);
                    }
                    len = 0;
                    nbytes = 0;
                    nbits = 0;
                } else {                                             // abort: ones shifted in, counters reset, bytes stay (ax25.py:36-39)
                    wb = cnt >= 7 ? 0x7Fu : (((wb & 0x7F) >> cnt) | ((0x7Fu << (7 - cnt)) & 0x7Fu));
                    nbits = 0;
                    nbytes = 0;
                }
            }
            if (__builtin_expect(len + 2 > data.size(), 0)) {
                room(len + 2);
                buf = data.data();
            }
        }
        ones = ones_in;
        data.resize(len);
    }

    void byte_done(bool from_one)
    {
        nbits = 0;
        data.push_back((uint8_t)wb);
        if (++nbytes > kMax) {
            nbytes = 0;
            if (from_one) ones = 0;                          // ax25.py:48-50: only the '1' branch clears it
        }
    }

    void feed(uint8_t byte, int64_t addr, Sink &sink) override
    {
        unsigned b = byte;
        for (int i = 0; i < 8; ++i, b <<= 1) {
            if (b & 0x80) {
                wb |= 0x80;
                ++ones;
                ++nbits;
                if (ones > 6) {                              // abort: counters reset, collected bytes stay (ax25.py:36-39)
                    nbits = 0;
                    nbytes = 0;
                }
                if (nbits == 8) byte_done(true);
                wb >>= 1;
            } else {
                if (ones < 5) {
                    if (++nbits == 8) byte_done(false);
                    wb >>= 1;
                } else if (ones == 6) {                      // flag
                    if (nbytes >= kMin && nbits == 7) sink.push(data, addr, 0, source);
                    data.clear();
                    nbytes = 0;
                    nbits = 0;
                }                                            // ones == 5: stuffed zero dropped; ones > 6: nothing
                ones = 0;
            }
        }
    }
};

const uint8_t kHamming74[128] = {   // il2p.py:23-40
    0x0, 0x0, 0x0, 0x3, 0x0, 0x5, 0xe, 0x7, 0x0, 0x9, 0xe, 0xb, 0xe, 0xd, 0xe, 0xe, 0x0, 0x3, 0x3, 0x3, 0x4, 0xd, 0x6, 0x3,
    0x8, 0xd, 0xa, 0x3, 0xd, 0xd, 0xe, 0xd, 0x0, 0x5, 0x2, 0xb, 0x5, 0x5, 0x6, 0x5, 0x8, 0xb, 0xb, 0xb, 0xc, 0x5, 0xe, 0xb,
    0x8, 0x1, 0x6, 0x3, 0x6, 0x5, 0x6, 0x6, 0x8, 0x8, 0x8, 0xb, 0x8, 0xd, 0x6, 0xf, 0x0, 0x9, 0x2, 0x7, 0x4, 0x7, 0x7, 0x7,
    0x9, 0x9, 0xa, 0x9, 0xc, 0x9, 0xe, 0x7, 0x4, 0x1, 0xa, 0x3, 0x4, 0x4, 0x4, 0x7, 0xa, 0x9, 0xa, 0xa, 0x4, 0xd, 0xa, 0xf,
    0x2, 0x1, 0x2, 0x2, 0xc, 0x5, 0x2, 0x7, 0xc, 0x9, 0x2, 0xb, 0xc, 0xc, 0xc, 0xf, 0x1, 0x1, 0x2, 0x1, 0x4, 0x1, 0x6, 0xf,
    0x8, 0x1, 0xa, 0xf, 0xc, 0xf, 0xf, 0xf};

struct Il2p : pm_codec {
    enum State { kSync, kHeader, kBig, kSmall, kCrc };
    bool want_crc, disable_rs;
    int min_dist, sync_tol;
    State state = kSync;
    uint32_t word = 0xFFFFFF;                               // il2p.py:120
    uint8_t buf[255];
    int nbits = 0, nbuf = 0, block_index = 0, block_count = 0, block_size = 0, big_blocks = 0;
    int corrected = 0;                                      // NOT cleared when a block fails (il2p.py:203-212)
    bool fail = false;
    std::vector<uint8_t> data;
    int sync_run = 0;                                       // whole input bytes taken in sync search since it was (re)entered, saturating

    const uint8_t *feasible = nullptr;

    Il2p(int src, bool crc, bool norx, int md, int tol) : want_crc(crc), disable_rs(norx), min_dist(md), sync_tol(tol)
    {
        feasible = sync_feasible(tol);
        source = src;
        memset(buf, 0, sizeof(buf));
    }

    void feed_many(const uint8_t *d, const int64_t *a, int64_t n) override
    {
        for (int64_t k = 0; k < n; ++k) {
            if (state != kSync) {
                sync_run = 0;
                // inside a packet: an input byte completes exactly one packet byte, nbits bits into it.  Unless that byte ends a
                // block (header, payload block, CRC) nothing else can happen in this input byte: take it in one step.
                const int needed = state == kHeader ? 15 : state == kCrc ? 4 : block_size + 16;
                if (nbuf + 1 < needed) {
                    buf[nbuf++] = (uint8_t)((word << (8 - nbits)) | ((unsigned)d[k] >> nbits));
                    word = d[k];                              // the last eight bits seen
                    continue;
                }
                feed(d[k], a[k], sink);
                continue;
            }
            // Between packets nearly every byte fails the feasibility test below, which reads nothing but the two input bytes in front
            // of it: skip ahead on the input itself -- no shift register carried from byte to byte -- and rebuild the register (the last
            // 32 bits seen) where the run of infeasible bytes ends.  Only once the register IS the last 32 input bits: the reference keeps
            // eight bits of it while it is inside a packet (il2p.py:146-152 with mask 0xFF) and starts with 0xFFFFFF (il2p.py:119), so
            // for the first four bytes of a sync search the register holds zeros (ones) where the input had bits, and a sync word that
            // overlaps the tail of a header or packet must be judged on the register, byte by byte, as below.
            if (feasible && sync_run >= 4 && k >= 4 && k + 1 < n) {
                int64_t j = k;
                while (j < n && !feasible[((unsigned)d[j - 2] << 8) | d[j - 1]]) ++j;
                if (j > k) {
                    nbits += 8 * (int)(j - k);
                    word = ((uint32_t)d[j - 4] << 24) | ((uint32_t)d[j - 3] << 16) | ((uint32_t)d[j - 2] << 8) | d[j - 1];
                    k = j;
                    if (k >= n) break;
                }
            }
            // sync search (il2p.py:367-376): the last 32 bits before each of the byte's 8 bit positions, tested without a
            // per-bit loop; a hit (rare) hands the rest of the byte to the state machine
            const uint64_t win = ((uint64_t)word << 8) | d[k];
            // The two bytes before the current one lie wholly inside all eight candidate windows, so one table lookup on them
            // tells which bit offsets can still be within sync_tol of either pattern -- almost always none
            if (sync_run < 4) ++sync_run;
            if (feasible && !feasible[(win >> 8) & 0xFFFF]) {
                word = (uint32_t)win;
                nbits += 8;
                continue;
            }
            int i = 0;
            bool hit = false;
#define PM_SYNC_AT(S)                                                                                                   \
            if (!hit) {                                                                                                 \
                const uint32_t w = (uint32_t)(win >> (7 - (S)));                                                        \
                if (__builtin_popcount((w & 0xFFFFFF) ^ 0xF15E48) <= sync_tol || __builtin_popcount(w ^ 0x5D57DF7Fu) <= sync_tol) { \
                    hit = true;                                                                                         \
                    i = (S) + 1;                                                                                        \
                    word = w;                                                                                           \
                }                                                                                                       \
            }
            PM_SYNC_AT(0) PM_SYNC_AT(1) PM_SYNC_AT(2) PM_SYNC_AT(3) PM_SYNC_AT(4) PM_SYNC_AT(5) PM_SYNC_AT(6) PM_SYNC_AT(7)
#undef PM_SYNC_AT
            if (!hit) {
                word = (uint32_t)win;
                nbits += 8;
                continue;
            }
            const unsigned b = ((unsigned)d[k] << i) & 0xFF;
            nbits = 0;
            state = kHeader;
            feed_bits(b, 8 - i, a[k], sink);
        }
    }

    // sync_feasible(tol)[v] != 0 iff, with v as the 16 bits before the current byte, some bit offset S lets the 24-bit sync word
    // 0xF15E48 or the 32-bit pattern 0x5D57DF7F end inside the current byte within `tol` mismatches: v supplies window bits
    // S+1 .. S+16 of either pattern, and they alone must not exceed the tolerance.  One 64 KB table per tolerance, built once.
    static const uint8_t *sync_feasible(int tol)
    {
        constexpr int kMaxTol = 8;                       // beyond that nearly every value is feasible: no filter
        if (tol < 0 || tol > kMaxTol) return nullptr;
        static std::once_flag once[kMaxTol + 1];
        static std::vector<uint8_t> table[kMaxTol + 1];
        std::call_once(once[tol], [tol] {
            std::vector<uint8_t> t(65536, 0);
            for (int S = 0; S < 8; ++S) {
                const uint32_t a = (0xF15E48u >> (S + 1)) & 0xFFFF, b = (0x5D57DF7Fu >> (S + 1)) & 0xFFFF;
                for (uint32_t v = 0; v < 65536; ++v)
                    if (__builtin_popcount(v ^ a) <= tol || __builtin_popcount(v ^ b) <= tol) t[v] |= (uint8_t)(1u << S);
            }
            table[tol].swap(t);
        });
        return table[tol].data();
    }

    static void descramble(uint8_t *p, int n)
    {   // il2p.py:160-163 + lfsr.py:54-92: x^9 + x^4 + 1 (0x211), register preset 0x1F0.  The register step is linear over GF(2), so
        // a byte's eight steps are (what the register alone does) XOR (what the byte alone does): four small tables built from the
        // bit-serial loop below, which is the reference's (a 100-byte block took 800 of its iterations).
        struct Tab {
            uint8_t out_reg[512], out_in[256];
            uint16_t nxt_reg[512], nxt_in[256];
            Tab()
            {
                auto run = [](unsigned reg, unsigned b, unsigned &regout) {
                    unsigned w = 0;
                    for (int i = 0; i < 8; ++i) {
                        w = (w << 1) & 0xFE;
                        if (b & 0x80) reg ^= 0x211;
                        w |= reg & 1;
                        b <<= 1;
                        reg >>= 1;
                    }
                    regout = reg;
                    return w;
                };
                for (unsigned r = 0; r < 512; ++r) {
                    unsigned ro;
                    out_reg[r] = (uint8_t)run(r, 0, ro);
                    nxt_reg[r] = (uint16_t)ro;
                }
                for (unsigned b = 0; b < 256; ++b) {
                    unsigned ro;
                    out_in[b] = (uint8_t)run(0, b, ro);
                    nxt_in[b] = (uint16_t)ro;
                }
            }
        };
        static const Tab t;
        unsigned reg = 0x1F0;
        for (int k = 0; k < n; ++k) {
            const unsigned b = p[k];
            p[k] = (uint8_t)(t.out_reg[reg] ^ t.out_in[b]);
            reg = (unsigned)(t.nxt_reg[reg] ^ t.nxt_in[b]);
        }
    }

    void rs(int roots)
    {
        const int r = disable_rs ? 0 : rs_decode(roots, buf, nbuf, min_dist);
        if (r < 0) fail = true;
        else corrected += r;
    }

    void emit(int64_t addr, Sink &sink)
    {
        sink.push(data, addr, corrected, source);
        corrected = 0;
        data.clear();
        state = kSync;
    }

    void finish(int64_t addr, Sink &sink)
    {
        if (want_crc) {
            state = kCrc;
        } else {                                             // il2p.py:427-431: append a computed CRC
            const int c = crc16(data.data(), (int64_t)data.size());
            data.push_back((uint8_t)(c & 0xFF));
            data.push_back((uint8_t)(c >> 8));
            emit(addr, sink);
        }
    }

    // il2p.py:214-344: unpack the 13 header bytes and rebuild the AX.25 header.  Returns the payload byte count.
    int header()
    {
        const uint8_t *b = buf;
        const int type = (b[1] & 0x80) >> 7;
        int count = 0, pid = 0, ctl = 0;
        for (int i = 0; i < 10; ++i)
            if (b[i + 2] & 0x80) count |= 0x200 >> i;
        for (int i = 0; i < 4; ++i)
            if (b[i + 1] & 0x40) pid |= 0x8 >> i;
        for (int i = 0; i < 7; ++i)
            if (b[i + 5] & 0x40) ctl |= 0x40 >> i;
        enum { UI, S, U, I } kind = (b[0] & 0x40) ? UI : (pid == 0 ? S : (pid == 1 ? U : I));
        static const uint8_t pid_table[16] = {0, 0, 0x10, 0x01, 0x06, 0x07, 0x08, 0xC3, 0xC4, 0xCA, 0xCB, 0xCC, 0xCD, 0xCE, 0xCF, 0xF0};
        const bool pf = ctl & 0x40;
        bool cbit = false;
        int nr = 0, ns = 0, op = 0;
        if (kind == I) {
            ns = ctl & 0x7;
            nr = (ctl >> 3) & 0x7;
            cbit = true;
        } else if (kind == S) {
            nr = (ctl >> 3) & 0x7;
            cbit = ctl & 0x4;
            op = ctl & 0x3;
        } else {
            cbit = ctl & 0x4;
            op = (ctl >> 3) & 0x7;
        }
        if (type == 1) {                                     // il2p.py:292-340
            for (int i = 0; i < 6; ++i) data.push_back((uint8_t)(((b[i] & 0x3F) + 0x20) << 1));
            data.push_back((uint8_t)(((b[12] >> 4) << 1) + 0x60 + (cbit ? 0x80 : 0)));
            for (int i = 0; i < 6; ++i) data.push_back((uint8_t)(((b[i + 6] & 0x3F) + 0x20) << 1));
            data.push_back((uint8_t)(((b[12] & 0xF) << 1) + 0x60 + (cbit ? 0 : 0x80) + 1));
            static const uint8_t u_control[8] = {0x2F, 0x43, 0x0F, 0x63, 0x87, 0x03, 0xAF, 0xE3};
            int cb;
            if (kind == U || kind == UI) cb = u_control[op] | (pf ? 0x10 : 0);
            else if (kind == S) cb = 0x1 | (op << 2) | (nr << 5) | (pf ? 0x10 : 0);
            else cb = (ns << 1) | (nr << 5) | (pf ? 0x10 : 0);
            data.push_back((uint8_t)cb);
            if (pid_table[pid] != 0) data.push_back(pid_table[pid]);
        }                                                    // type 0: transparent encapsulation, nothing added
        return count;
    }

    void feed(uint8_t byte, int64_t addr, Sink &sink) override { feed_bits(byte, 8, addr, sink); }

    // `count` bits of b, most significant first (b is pre-shifted so that the next bit is 0x80)
    void feed_bits(unsigned b, int count, int64_t addr, Sink &sink)
    {
        for (int i = 0; i < count; ++i, b <<= 1) {
            const uint32_t mask = state == kSync ? 0xFFFFFFFFu : 0xFFu;
            word = ((word << 1) & mask) | ((b & 0x80) ? 1u : 0u);      // il2p.py:146-152
            ++nbits;
            if (state == kSync) {
                if (__builtin_popcount((word & 0xFFFFFF) ^ 0xF15E48) <= sync_tol ||
                    __builtin_popcount(word ^ 0x5D57DF7Fu) <= sync_tol) {              // il2p.py:367-376
                    nbits = 0;
                    state = kHeader;
                }
                continue;
            }
            if (nbits != 8) continue;
            nbits = 0;
            buf[nbuf++] = (uint8_t)word;
            switch (state) {
            case kHeader: {
                if (nbuf != 15) break;
                rs(2);
                descramble(buf, 13);
                nbuf = 0;
                block_index = 0;
                const int count = header();                  // the header is rebuilt even when RS failed, then dropped
                if (fail) {
                    fail = false;
                    state = kSync;
                    data.clear();
                } else if (count > 0) {                      // il2p.py:346-358
                    block_count = (count + 238) / 239;
                    block_size = count / block_count;
                    big_blocks = count - block_count * block_size;
                    if (big_blocks > 0) {
                        ++block_size;
                        state = kBig;
                    } else {
                        state = kSmall;
                    }
                } else {
                    finish(addr, sink);
                }
                break;
            }
            case kBig:
            case kSmall: {
                if (nbuf != block_size + 16) break;
                rs(16);
                descramble(buf, nbuf);
                data.insert(data.end(), buf, buf + block_size);
                ++block_index;
                nbuf = 0;
                if (fail) {
                    fail = false;
                    data.clear();
                    state = kSync;
                } else if (state == kBig && block_index == big_blocks) {
                    if (block_count > block_index) {
                        --block_size;
                        state = kSmall;
                    } else {
                        finish(addr, sink);
                    }
                } else if (state == kSmall && block_index == block_count) {
                    finish(addr, sink);
                }
                break;
            }
            case kCrc: {
                if (nbuf != 4) break;
                nbuf = 0;
                int c = 0;
                for (int k = 0; k < 4; ++k) c += kHamming74[buf[k] & 0x7F] << (12 - 4 * k);     // il2p.py:509-512
                data.push_back((uint8_t)(c & 0xFF));
                data.push_back((uint8_t)(c >> 8));
                emit(addr, sink);
                break;
            }
            default:
                break;
            }
        }
    }
};

// ---- worker threads of the batched host stage ---------------------------------------------------
// pm_host_decode_batch / pm_codec_fetch_batch run one chain per task.  The workers live as long as the process (started on
// demand, and again in a forked child, which inherits none of them); a batch's caller works through the tasks itself as
// well, so a batch completes even when every worker is busy with other callers' batches.
class HostPool {
public:
    static HostPool &get()
    {
        static std::mutex m;
        static HostPool *pool = nullptr;
        static pid_t owner = 0;
        std::lock_guard<std::mutex> g(m);
        if (!pool || owner != getpid()) {          // first use, or a forked child: the parent's object is left alone
            pool = new HostPool();
            owner = getpid();
        }
        return *pool;
    }

    // fn(k) for k in [0, n), on at most `threads` threads including the caller
    void run(int n, int threads, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        auto st = std::make_shared<Batch>();
        st->n = n;
        st->fn = &fn;
        const int helpers = std::max(0, std::min(std::min(n, threads) - 1, kMaxWorkers));
        if (helpers > 0) {
            {
                std::lock_guard<std::mutex> g(m_);
                for (int h = 0; h < helpers; ++h) q_.push_back(st);
                const int want = std::min(kMaxWorkers - workers_, (int)q_.size() - idle_);
                for (int w = 0; w < want; ++w) {
                    std::thread([this] { loop(); }).detach();
                    ++workers_;
                }
            }
            for (int h = 0; h < helpers; ++h) cv_.notify_one();      // one sleeper per queued share, not the whole pool for a handful of tasks
        }
        work(*st);
        std::unique_lock<std::mutex> g(st->m);
        st->cv.wait(g, [&] { return st->done == st->n; });
    }

private:
    static constexpr int kMaxWorkers = 64;
    struct Batch {
        int n = 0;
        const std::function<void(int)> *fn = nullptr;
        std::atomic<int> next{0};
        int done = 0;
        std::mutex m;
        std::condition_variable cv;
    };
    static void work(Batch &b)
    {
        int mine = 0;
        for (;;) {
            const int k = b.next.fetch_add(1);
            if (k >= b.n) break;
            (*b.fn)(k);                            // run() has not returned while any index is unfinished: fn is alive
            ++mine;
        }
        if (mine) {
            std::lock_guard<std::mutex> g(b.m);
            b.done += mine;
            if (b.done == b.n) b.cv.notify_all();
        }
    }
    void loop()
    {
        (void)pthread_setname_np(pthread_self(), "pm-decode");
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> g(m_);
                ++idle_;
                cv_.wait(g, [&] { return !q_.empty(); });
                --idle_;
                b = q_.front();
                q_.pop_front();
            }
            work(*b);
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Batch>> q_;
    int workers_ = 0, idle_ = 0;
};

}  // namespace

extern "C" {

int pm_lfsr_unscramble(const uint8_t *h_in, int64_t n, uint64_t poly, int invert, uint64_t *h_sr, uint8_t *h_out)
{
    if (n < 0 || (n > 0 && (!h_in || !h_out)) || !h_sr) return pm_set_error(PM_ERR_ARG, "pm_lfsr_unscramble: bad argument");
    // The Galois register of lfsr.py:30-51 is a feed-forward filter over GF(2): with stream bit t (MSB of byte 0 first),
    //     out[t] = XOR over set bits j of poly of in[t - j]   (+ bit t of the incoming register for t < 64)
    // so the stream is processed as big-endian 64-bit words XORed with copies of itself delayed by each tap.
    const uint64_t reg0 = *h_sr;
    if (n == 0) return PM_OK;
    const int64_t nwords = (n + 7) / 8;
    auto load = [&](int64_t w) -> uint64_t {
        if (w < 0) return 0;
        const int64_t base = w * 8;
        if (base + 8 <= n) {                // a whole word: one unaligned load, big-endian (MSB of byte 0 is stream bit 0)
            uint64_t v;
            memcpy(&v, h_in + base, 8);
            return __builtin_bswap64(v);
        }
        uint64_t v = 0;
        const int take = (int)std::max<int64_t>(0, n - base);
        for (int i = 0; i < take; ++i) v |= (uint64_t)h_in[base + i] << (56 - 8 * i);
        return v;
    };
    // the taps as shift amounts, once (the inner loop below runs n/8 times)
    int taps[64], ntaps = 0;
    for (uint64_t p = poly; p; p &= p - 1) taps[ntaps++] = __builtin_ctzll(p);
    uint64_t prev = 0, cur = load(0);
    int64_t w0 = 0;
    if (poly == 1 && n >= 16) {
        // the identity polynomial (configs/fsk_9600.json: "poly": "0x1", with and without inversion): out = in past the first word, which
        // alone sees the incoming register -- a byte loop the compiler vectorises instead of a word at a time through the tap loop
        uint64_t r = reg0, rev = 0;
        for (int i = 0; i < 64; ++i, r >>= 1) rev = (rev << 1) | (r & 1);
        uint64_t o = cur ^ rev;
        if (invert) o = ~o;
        const uint64_t be = __builtin_bswap64(o);
        memcpy(h_out, &be, 8);
        const uint8_t flip = invert ? 0xFF : 0x00;
        for (int64_t i = 8; i < n; ++i) h_out[i] = h_in[i] ^ flip;
        w0 = nwords;
    }
    for (int64_t w = w0; w < nwords; ++w) {
        uint64_t o = 0;
        for (int q = 0; q < ntaps; ++q) {
            const int j = taps[q];
            o ^= j == 0 ? cur : ((cur >> j) | (prev << (64 - j)));
        }
        if (w == 0) {                       // pending contributions of the incoming register: bit t -> stream bit t
            uint64_t r = reg0, rev = 0;
            for (int i = 0; i < 64; ++i, r >>= 1) rev = (rev << 1) | (r & 1);
            o ^= rev;
        }
        if (invert) o = ~o;
        const int64_t base = w * 8;
        if (base + 8 <= n) {
            const uint64_t be = __builtin_bswap64(o);
            memcpy(h_out + base, &be, 8);
        } else {
            const int take = (int)(n - base);
            for (int i = 0; i < take; ++i) h_out[base + i] = (uint8_t)(o >> (56 - 8 * i));
        }
        prev = cur;
        cur = load(w + 1);
    }
    // outgoing register after T = 8n bits: reg0 >> T, plus poly >> (T - t) for every set input bit t among the last 63
    const int64_t T = n * 8;
    uint64_t reg = T < 64 ? (reg0 >> T) : 0;
    for (int64_t t = std::max<int64_t>(0, T - 63); t < T; ++t)
        if ((h_in[t >> 3] >> (7 - (t & 7))) & 1) reg ^= poly >> (T - t);
    *h_sr = reg;
    return PM_OK;
}

int pm_codec_create(int kind, int crc, int disable_rs, int min_dist, int sync_tol, int source_decoder, pm_codec **out)
{
    if (!out || (kind != 0 && kind != 1)) return pm_set_error(PM_ERR_ARG, "pm_codec_create: kind must be 0 (ax25) or 1 (il2p)");
    if (kind == 0) *out = new Ax25(source_decoder);
    else *out = new Il2p(source_decoder, crc != 0, disable_rs != 0, min_dist, sync_tol);
    return PM_OK;
}

int pm_codec_destroy(pm_codec *c)
{
    delete c;
    return PM_OK;
}

int pm_codec_set_source(pm_codec *c, int32_t source_decoder)
{
    if (!c) return pm_set_error(PM_ERR_ARG, "pm_codec_set_source: no codec");
    c->source = source_decoder;
    return PM_OK;
}

int pm_codec_decode(pm_codec *c, const uint8_t *h_data, const int64_t *h_addr, int64_t n, int64_t *h_pending)
{
    if (!c || n < 0 || (n > 0 && (!h_data || !h_addr)) || !h_pending)
        return pm_set_error(PM_ERR_ARG, "pm_codec_decode: bad argument");
    c->feed_many(h_data, h_addr, n);
    *h_pending = (int64_t)c->sink.q.size();
    return PM_OK;
}

// clean: h_out's rows hold zeros wherever this call does not write (a row block kept zero-tailed by its owner, pm_pipe.hip): the
// 1280-byte payload field costs what the packet is long, not what the row is
static int codec_fetch(pm_codec *c, pm_packet *h_out, int64_t cap, int64_t *h_count, bool clean)
{
    if (!c || cap < 0 || (cap > 0 && !h_out) || !h_count) return pm_set_error(PM_ERR_ARG, "pm_codec_fetch: bad argument");
    const int64_t take = std::min<int64_t>(cap, (int64_t)c->sink.q.size());
    for (int64_t k = 0; k < take; ++k) {
        const Queued &src = c->sink.q[(size_t)k];
        const uint8_t *bytes = c->sink.bytes.data() + src.off;
        pm_packet &p = h_out[k];
        memset(&p, 0, offsetof(pm_packet, data));            // h_out may be uninitialised memory: every byte of the row is written
        p.streamaddress = src.addr;
        p.len = (int32_t)std::min<size_t>(src.len, PM_PKT_MAX);
        p.bytes_corrected = src.corrected;
        p.source_decoder = c->source;
        memcpy(p.data, bytes, (size_t)p.len);
        if (!clean) memset(p.data + p.len, 0, sizeof(p.data) - (size_t)p.len);
        finalize(p);
        if (src.len > PM_PKT_MAX) {
            // A frame longer than a row: the reference's AX.25 decoder never drops collected bytes when its byte counter wraps at 1023
            // (ax25.py:41-47), so a flag after a long stretch without one can close a frame of any length.  The row keeps its first
            // PM_PKT_MAX bytes; CRC and validity are those of the WHOLE frame, as PacketMeta.CalcCRC would find them.
            const size_t L = src.len;
            p.carried_crc = bytes[L - 1] * 256 + bytes[L - 2];
            p.calculated_crc = crc16(bytes, (int)(L - 2));
            p.valid_crc = p.carried_crc == p.calculated_crc;
        }
    }
    c->sink.drop_front((size_t)take);
    *h_count = take;
    return PM_OK;
}

int pm_codec_fetch(pm_codec *c, pm_packet *h_out, int64_t cap, int64_t *h_count) { return codec_fetch(c, h_out, cap, h_count, false); }

int pm_host_decode_batch(pm_host_job *jobs, int njobs, int threads)
{
    if (njobs < 0 || (njobs > 0 && !jobs) || threads < 1) return pm_set_error(PM_ERR_ARG, "pm_host_decode_batch: bad argument");
    for (int j = 0; j < njobs; ++j) {
        const pm_host_job &q = jobs[j];
        if (!q.codec || q.n < 0 || (q.n > 0 && (!q.h_data || (!q.h_addr && !q.h_addr_delta))))
            return pm_set_error(PM_ERR_ARG, "pm_host_decode_batch: job %d: bad argument", j);
        for (int i = 0; i < j; ++i)
            if (jobs[i].codec == q.codec) return pm_set_error(PM_ERR_ARG, "pm_host_decode_batch: jobs %d and %d share a codec", i, j);
    }
    HostPool::get().run(njobs, threads, [&](int j) {
        pm_host_job &q = jobs[j];
        thread_local std::vector<uint8_t> plain;
        thread_local std::vector<int64_t> wide;
        if (!q.h_plain && (int64_t)plain.size() < q.n) plain.resize((size_t)q.n);
        const int64_t *addr = q.h_addr;
        if (!addr && q.n > 0) {                 // the compact form of pm_slice_compact: first address + 16-bit steps
            if ((int64_t)wide.size() < q.n) wide.resize((size_t)q.n);
            int64_t a = q.addr_first;
            for (int64_t i = 0; i < q.n; ++i) wide[(size_t)i] = (a += q.h_addr_delta[i]);
            addr = wide.data();
        }
        uint8_t *out = q.h_plain ? q.h_plain : plain.data();
        q.status = pm_lfsr_unscramble(q.h_data, q.n, q.lfsr_poly, q.lfsr_invert, &q.lfsr_state, out);
        if (q.status == PM_OK) q.status = pm_codec_decode(q.codec, out, addr, q.n, &q.pending);
    });
    for (int j = 0; j < njobs; ++j)
        if (jobs[j].status != PM_OK) return jobs[j].status;
    return PM_OK;
}

static int codec_fetch_batch(pm_codec *const *codecs, const int64_t *counts, int n, pm_packet *h_out, int threads, bool clean);
int pm_codec_fetch_batch(pm_codec *const *codecs, const int64_t *counts, int n, pm_packet *h_out, int threads)
{
    return codec_fetch_batch(codecs, counts, n, h_out, threads, false);
}
int pm_codec_fetch_batch_clean(pm_codec *const *codecs, const int64_t *counts, int n, pm_packet *h_out, int threads)
{
    return codec_fetch_batch(codecs, counts, n, h_out, threads, true);
}
static int codec_fetch_batch(pm_codec *const *codecs, const int64_t *counts, int n, pm_packet *h_out, int threads, bool clean)
{
    if (n < 0 || (n > 0 && (!codecs || !counts)) || threads < 1) return pm_set_error(PM_ERR_ARG, "pm_codec_fetch_batch: bad argument");
    std::vector<int64_t> at((size_t)n + 1, 0);
    for (int j = 0; j < n; ++j) {
        if (!codecs[j] || counts[j] < 0 || counts[j] > (int64_t)codecs[j]->sink.q.size())
            return pm_set_error(PM_ERR_ARG, "pm_codec_fetch_batch: codec %d: bad count", j);
        at[(size_t)j + 1] = at[(size_t)j] + counts[j];
    }
    if (at[(size_t)n] > 0 && !h_out) return pm_set_error(PM_ERR_ARG, "pm_codec_fetch_batch: no output");
    std::vector<int> rc((size_t)n, PM_OK);
    HostPool::get().run(n, threads, [&](int j) {
        int64_t got = 0;
        if (counts[j] > 0) rc[(size_t)j] = codec_fetch(codecs[j], h_out + at[(size_t)j], counts[j], &got, clean);
    });
    for (int j = 0; j < n; ++j)
        if (rc[(size_t)j] != PM_OK) return rc[(size_t)j];
    return PM_OK;
}

int pm_crc16_ccitt(const uint8_t *h_data, int64_t n) { return crc16(h_data, n); }

int64_t pm_packets_pack(const pm_packet *rows, int64_t n, uint8_t *out, int64_t cap)
{
    if (n < 0 || (n > 0 && !rows) || cap < 0) return pm_set_error(PM_ERR_ARG, "pm_packets_pack: bad argument");
    constexpr size_t H = offsetof(pm_packet, data);
    int64_t need = 0;
    for (int64_t k = 0; k < n; ++k) {
        if (rows[k].len < 0 || rows[k].len > PM_PKT_MAX) return pm_set_error(PM_ERR_ARG, "pm_packets_pack: row %lld has len %d", (long long)k, rows[k].len);
        need += (int64_t)H + rows[k].len;
    }
    if (need > cap || !out) return need;
    uint8_t *w = out;
    for (int64_t k = 0; k < n; ++k) {
        memcpy(w, &rows[k], H + (size_t)rows[k].len);
        w += H + (size_t)rows[k].len;
    }
    return need;
}

int64_t pm_packets_unpack(const uint8_t *in, int64_t bytes, pm_packet *rows, int64_t cap_rows)
{
    if (bytes < 0 || (bytes > 0 && !in) || cap_rows < 0 || (cap_rows > 0 && !rows)) return pm_set_error(PM_ERR_ARG, "pm_packets_unpack: bad argument");
    constexpr size_t H = offsetof(pm_packet, data);
    int64_t at = 0, k = 0;
    while (at < bytes) {
        if (bytes - at < (int64_t)H || k >= cap_rows) return pm_set_error(PM_ERR_ARG, "pm_packets_unpack: truncated stream or too many rows");
        pm_packet &p = rows[k];
        memcpy(&p, in + at, H);
        if (p.len < 0 || p.len > PM_PKT_MAX || bytes - at - (int64_t)H < p.len) return pm_set_error(PM_ERR_ARG, "pm_packets_unpack: bad length");
        memcpy(p.data, in + at + H, (size_t)p.len);
        memset(p.data + p.len, 0, sizeof(p.data) - (size_t)p.len);
        at += (int64_t)H + p.len;
        ++k;
    }
    return k;
}

int64_t pm_packets_index(const uint8_t *in, int64_t bytes, pm_packet_head *heads, int64_t *payload_at, int64_t cap_rows)
{
    if (bytes < 0 || (bytes > 0 && !in) || cap_rows < 0 || (cap_rows > 0 && (!heads || !payload_at)))
        return pm_set_error(PM_ERR_ARG, "pm_packets_index: bad argument");
    constexpr size_t H = sizeof(pm_packet_head);
    static_assert(sizeof(pm_packet_head) == offsetof(pm_packet, data), "pm_packet_head is the head of pm_packet");
    int64_t at = 0, k = 0;
    while (at < bytes) {
        if (bytes - at < (int64_t)H || k >= cap_rows) return pm_set_error(PM_ERR_ARG, "pm_packets_index: truncated stream or too many records");
        __builtin_prefetch(in + at + 1024);
        memcpy(&heads[k], in + at, H);
        const int32_t len = heads[k].len;
        if (len < 0 || len > PM_PKT_MAX || bytes - at - (int64_t)H < len) return pm_set_error(PM_ERR_ARG, "pm_packets_index: bad length");
        payload_at[k] = at + (int64_t)H;
        at += (int64_t)H + len;
        ++k;
    }
    return k;
}

int64_t pm_correlate(pm_packet *p, const int64_t *counts, int nchains, double address_distance,
                     int64_t *uniq, int32_t *corr_decoders, int64_t corr_cap)
{
    return pm_correlate_strided(p, (int64_t)sizeof(pm_packet), counts, nchains, address_distance, uniq, corr_decoders, corr_cap);
}

int64_t pm_correlate_strided(void *records, int64_t stride, const int64_t *counts, int nchains, double address_distance,
                             int64_t *uniq, int32_t *corr_decoders, int64_t corr_cap)
{
    if (!records || !counts || !uniq || nchains < 0 || stride < (int64_t)sizeof(pm_packet_head))
        return pm_set_error(PM_ERR_ARG, "pm_correlate: bad argument");
    auto rec = [&](int64_t i) -> pm_packet_head & { return *reinterpret_cast<pm_packet_head *>(static_cast<char *>(records) + i * stride); };
    // packet_meta.py:230-271.  Chains in config order; the first chain's valid packets are all unique; a later
    // packet is a duplicate of the FIRST unique packet (insertion order) from another decoder within
    // address_distance and with equal calculated CRC.
    // The candidates for a packet are the unique packets with its CRC: one bucket per CRC value, each in insertion order,
    // with address and decoder kept beside the index (rows are 1.3 KB apart; the scan must not touch them).
    struct Cand { int64_t addr; uint32_t j; int32_t src; };
    std::vector<int64_t> u;
    std::vector<std::vector<int32_t>> decoders;
    std::unordered_map<int32_t, std::vector<Cand>> by_crc;
    // (an 8-GPU run hands rank 0 the records of 64 chains, 44 000 per recording of the headline workload, most of them duplicates of the
    // first chain's ~700: the table is sized for the first chain's packets up front and never rehashes on the way)
    if (nchains > 0) {
        by_crc.reserve((size_t)std::max<int64_t>(64, 2 * counts[0]));
        u.reserve((size_t)std::max<int64_t>(64, 2 * counts[0]));
        decoders.reserve((size_t)std::max<int64_t>(64, 2 * counts[0]));
    }
    int64_t base = 0;
    for (int c = 0; c < nchains; ++c) {
        for (int64_t k = 0; k < counts[c]; ++k) {
            pm_packet_head &r = rec(base + k);
            __builtin_prefetch(static_cast<char *>(records) + (base + k + 8) * stride);
            if (!(r.valid_crc && r.valid_header)) continue;
            std::vector<Cand> &bucket = by_crc[r.calculated_crc];
            bool unique = true;
            if (c > 0) {
                for (const Cand &q : bucket) {
                    if (q.src == r.source_decoder) continue;
                    const int64_t d = r.streamaddress > q.addr ? r.streamaddress - q.addr : q.addr - r.streamaddress;
                    if ((double)d < address_distance) {
                        unique = false;
                        decoders[q.j].push_back(r.source_decoder);
                        break;
                    }
                }
            }
            if (unique) {
                bucket.push_back(Cand{r.streamaddress, (uint32_t)u.size(), r.source_decoder});
                u.push_back(base + k);
                decoders.push_back({r.source_decoder});
            }
        }
        base += counts[c];
    }
    std::vector<size_t> order(u.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rec(u[a]).streamaddress < rec(u[b]).streamaddress; });
    int64_t w = 0;
    for (size_t i = 0; i < order.size(); ++i) {
        const size_t j = order[i];
        uniq[i] = u[j];
        rec(u[j]).correlated_count = (int32_t)decoders[j].size();
        if (corr_decoders)
            for (int32_t d : decoders[j])
                if (w < corr_cap) corr_decoders[w++] = d;
    }
    return (int64_t)u.size();
}

}  // extern "C"
