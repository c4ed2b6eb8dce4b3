// Sequential recurrences of the demod_chain path: AGC (agc.py:26-80) and the three carrier loops
// (BPSK Costas psk.py:173-189, MPSK psk.py:734-747, AFSK PLL afsk_pll.py:153-165).
//
// Each recurrence carries a binary64 state from one sample to the next through quantisers
// (int(), floor(), round()), so it has to be evaluated in the reference's order: one lane owns one
// loop and steps through the samples.  These kernels are dependent-latency-bound, not HBM- or
// ALU-bound; throughput comes from running many loops at once (one lane each, 8 per wave so that
// waves spread over CUs).  What the rest of the wave does: all 64 lanes stream the input tile into LDS
// (coalesced), the owning lanes iterate over it, then all lanes stream the output tile back; for the
// AGC the 64 lanes also do the per-sample division, which is not part of the recurrence.
//
// Built with -ffp-contract=off; no fma is used here because the reference has none.
#include "pm_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr double kTwoPi = 2.0 * 3.141592653589793;

typedef double double2v __attribute__((ext_vector_type(2)));

struct LoopRegs {
    double phase_scaling, index_scaling, set_frequency, b0, b1, a1, p_rate, i_rate, i_limit, gain;
    double phase, control, sine, cosine, x0, x1, y0, integral, proportional;
    double bb0, bb1, ba1, cx0, cx1, cy0, sx0, sx1, sy0;      // QPSK Costas branch filters
};

struct AgcDev {
    double att, dec, sustain_time, sustain_inc, target;
};

// One sample of the envelope follower (agc.py:26-37), branch-free: the statements in the reference's order.
__device__ __forceinline__ void agc_step(double s, double &env, double &sustain, const AgcDev &P)
{
    const double cmp = fabs(s);
    const bool attack = cmp > env;                          // agc.py:28-32
    const double up = fmin(env + P.att, cmp);               // env += att; if env > cmp: env = cmp
    env = attack ? up : env;
    sustain = attack ? 0.0 : sustain;
    const bool decay = sustain >= P.sustain_time;           // agc.py:33-36
    const double dn = env - P.dec;
    env = decay ? (dn < 0 ? 0.0 : dn) : env;
    sustain += P.sustain_inc;                               // agc.py:37
}

__device__ __forceinline__ double iir1(double b0, double b1, double a1, double &x0, double &x1, double &y0, double sample)
{
    x1 = x0;                                                             // iir.py:40-42
    x0 = sample;
    double v = 0.0;
    v += x0 * b0;                                                        // iir.py:45-46
    v += x1 * b1;
    v += y0 * a1;                                                        // iir.py:48-52
    y0 = v;
    return v;
}

// The bodies below are written branch-free: a lone wave pays ~5 cycles per instruction and far more per taken branch, and
// every statement is on the loop-carried path -- ONE dependent chain of ~25 binary64 operations and two LDS reads per sample, at
// ~10 cycles per dependent operation: what bounds these kernels is the LENGTH of that chain, not the number of instructions beside it.
// tab2: 257 (sine, cosine) pairs in LDS, tab2[i] = {table[i], table[(i + 64) & 255]} and tab2[256] = {-, table[64]}: one 16-byte
// read serves both outputs of the NCO.
//
// nco.py:35-40: phase += step; `while p >= 2pi: p -= 2pi`; `while p < 0: p += 2pi`; index = int(p * scale).  For 0 <= p0 < 4pi -- every
// sample of every real run: the step is positive and far below one turn -- the first loop makes at most one trip (p0 - 2pi is exact
// there, Sterbenz, and below 2pi) and the second none.  Both candidates' indices are computed side by side and the compare only
// selects (round 3): the chain no longer runs through compare -> select -> add -> compare -> select before the multiplication,
// ~45 cycles of ~370.  Anything else (negative, a turn or more per sample, NaN) takes the statements as written.
__device__ __forceinline__ void nco_update(LoopRegs &L, const double2v *tab2)
{
    const double ph0 = L.phase + L.phase_scaling * (L.set_frequency + L.control);   // nco.py:35
    const double down = ph0 - kTwoPi;
    const bool wrap = ph0 >= kTwoPi;
    const int i0 = (int)(ph0 * L.index_scaling), i1 = (int)(down * L.index_scaling);  // nco.py:40, int() truncates
    double ph = wrap ? down : ph0;
    int idx = wrap ? i1 : i0;                                               // 0..256 on this path: the table read needs no clamp
    int at = idx;
    if (__builtin_expect(!(ph0 >= 0.0 && ph0 < 2.0 * kTwoPi), 0)) {
        ph = ph0;
        // The reference's two loops, with an end every wave reaches: a phase of 1e300 (or an infinity) is its own predecessor by 2 pi, the
        // reference would spin on it for ever, and on a GPU that is a kernel that never ends -- an input nobody ever produced, unless
        // a launch once reads memory nobody has written.  4096 trips cover 25 000 radians per sample; beyond, the phase stays what it is.
        for (int trip = 0; trip < 4096 && ph >= kTwoPi; ++trip) ph = ph - kTwoPi;      // nco.py:36-37
        for (int trip = 0; trip < 4096 && ph < 0; ++trip) ph = ph + kTwoPi;            // nco.py:38-39 (may round to exactly 2pi, and stays, as there)
        idx = (int)(ph * L.index_scaling);
        at = min(max(idx, 0), 256);
    }
    L.phase = ph;
    double2v sc = tab2[at];
    asm volatile("" : "+v"(sc));                                            // one unconditional ds_read_b128, no branch around it
    L.sine = idx < 256 ? sc.x : L.sine;                                     // nco.py:41-45: index 256 keeps the old value
    L.cosine = sc.y;                                                        // nco.py:46-51
}

__device__ __forceinline__ double iir_update(LoopRegs &L, double sample)
{
    L.x1 = L.x0;                                                         // iir.py:40-42
    L.x0 = sample;
    double v = 0.0;
    v += L.x0 * L.b0;                                                    // iir.py:45-46
    v += L.x1 * L.b1;
    v += L.y0 * L.a1;                                                    // iir.py:48-52 (Y[1] is the previous output)
    L.y0 = v;
    return v;
}

__device__ __forceinline__ double pi_update(LoopRegs &L, double sample)
{
    L.proportional = L.gain * L.p_rate * sample;                         // pi_control.py:26, (gain*p_rate)*sample
    double in = L.integral + L.gain * (L.i_rate * sample);               // pi_control.py:27
    // pi_control.py:28-31: `if I > limit: I = limit` / `if I < -limit: I = -limit` are min and max for every value that is not a
    // NaN (the integral never is: it is a clamped sum of finite products) -- one instruction each instead of compare + two selects,
    // and they sit on the loop-carried path
    in = __builtin_fmin(in, L.i_limit);
    in = __builtin_fmax(in, -L.i_limit);
    L.integral = in;
    return L.proportional + in;                                          // pi_control.py:32
}

__device__ __forceinline__ int pd_lookup(const int32_t *tbl, double re, double im)
{
    // phase_detector.py:124-149, granularity 64: floor(x * 64 * 0.5), clip to +-63, quadrant fold.  Kept in binary64 up to the one
    // conversion of the table index (every value below is a small integer, exact in a double), which takes the integer clamps, the
    // absolute values and the index arithmetic off the dependent chain:
    //   x * 64 * 0.5 == x * 32 bit for bit (powers of two: both products are exact wherever the reference does not overflow);
    //   clip(int(f), -63, 63) == int(clip(f, -63.0, 63.0)) for the integer-valued f = floor(..);
    //   Q1 T[r][i] | Q4 T[-i][r] | Q2 T[i][-r] | Q3 T[-r][-i]: rows and columns swap where the signs differ (0 counts as positive).
    double fr = floor(re * 32.0), fi = floor(im * 32.0);
    const bool swap = (fr >= 0) != (fi >= 0);
    fr = fmin(fmax(fr, -63.0), 63.0);               // >= 64 -> 63, <= -64 -> -63
    fi = fmin(fmax(fi, -63.0), 63.0);
    const double ar = fabs(fr), ai = fabs(fi);
    const double row = swap ? ai : ar, col = swap ? ar : ai;
    return tbl[(int)__builtin_fma(row, 64.0, col)];
}

enum { kCostas = 0, kPll = 1, kMpsk = 2, kQpsk = 3 };

// One sample of one loop: the reference's statements in the reference's order (psk.py:173-189 Costas, afsk_pll.py:153-165 PLL,
// psk.py:434-467 QPSK Costas, psk.py:734-747 MPSK).  s0 (and s1: the Hilbert arm, MPSK) in; o0 (and o1 for the two-output loops) out.
template <int MODE>
__device__ __forceinline__ void loop_step(LoopRegs &L, const double2v *tab2, const int32_t *pdt, double s0, double s1, double &o0, double &o1)
{
    if (MODE == kCostas) {
        const double sm = s0;
        nco_update(L, tab2);
        const double i_mixer = sm * L.cosine;             // psk.py:177
        const double q_mixer = sm * (-L.sine);            // psk.py:182
        const double lp = iir_update(L, i_mixer * q_mixer);
        L.control = pi_update(L, lp);                     // psk.py:187
        o0 = i_mixer;
    } else if (MODE == kPll) {
        nco_update(L, tab2);
        const double mixer = s0 * L.sine;                 // afsk_pll.py:156
        const double lp = iir_update(L, mixer);
        L.control = pi_update(L, lp);                     // afsk_pll.py:160
        o0 = L.proportional;                              // afsk_pll.py:163
    } else if (MODE == kQpsk) {
        const double sm = s0;
        nco_update(L, tab2);
        const double cl = iir1(L.bb0, L.bb1, L.ba1, L.cx0, L.cx1, L.cy0, sm * L.cosine);   // psk.py:438-440
        const double sl = iir1(L.bb0, L.bb1, L.ba1, L.sx0, L.sx1, L.sy0, sm * L.sine);     // psk.py:449-451
        const double a = sl >= 0 ? cl : -cl;              // cos_lp * sgn(sin_lp)            psk.py:455-459
        const double b = cl >= 0 ? sl : -sl;              // sin_lp * sgn(cos_lp)            psk.py:444-447
        const double lp = iir_update(L, a - b);           // psk.py:459-461
        L.control = pi_update(L, lp);                     // psk.py:463
        o0 = sl;                                          // i_data <- Sine_LPF              psk.py:452
        o1 = cl;                                          // q_data <- Cosine_LPF            psk.py:453
    } else {
        const double sr = s0, si = s1;
        nco_update(L, tab2);
        const double ar = L.cosine, ai = -L.sine;         // nco.py:52-53
        const double re = (sr * ar) - (si * ai);          // complexmath.py:16
        const double im = (ar * si) + (sr * ai);          // complexmath.py:17
        const int e = pd_lookup(pdt, re, im);             // psk.py:739
        const double lp = iir_update(L, (double)e);
        L.control = rint(pi_update(L, lp));               // psk.py:740, round() is half-to-even
        o0 = re;
        o1 = im;
    }
}

// A workgroup is TWO waves.  Wave 0 owns the loops (lane l < ng steps loop g0 + l through one tile of kTile samples, reading and
// writing LDS only); wave 1 moves the tiles: while wave 0 steps tile s - 1 it loads tile s from memory into the other input buffer
// and stores tile s - 2 from the other output buffer.  One barrier per tile.  The loop-owning wave therefore never waits for
// memory: with the tile I/O on the stepping wave itself (round 2) a wave whose eight loops read eight different rows -- eight
// recordings of a batch -- spent 50 ns per sample of its 228 waiting for its own loads and stores (tools/loop_scaling.py).
//
// LDS layout (doubles): tab2[257 pairs, 516] | in[2][rows_in][NIN][kPad] | out[2][NOUT][kG][kPad] | pd[4096 int32] (mpsk)
// Input rows: loop l reads row l / per_row (x0 + row * x_stride): per_row = 1 gives every loop its own input, per_row = nloops (with
// any stride) one input for all, and a batch of recordings x chains has per_row = chains (the chains of a recording share its
// front end).  `rows_lds` = the most distinct rows one wave's kG loops can touch (loop_rows_lds), which sizes the LDS image.
//
// Two shapes.  <8, 256>: eight loops per workgroup and tiles of 256 samples -- a launch of up to ~2000 loops spreads over all 256 CUs
// (the stepping wave is bound by instruction issue whatever the number of active lanes, so the fewer loops a wave holds the more
// waves step at once).  <64, 32>: every lane of the stepping wave holds a loop -- for launches with more loops than that, where
// eight-lane waves would queue up behind each other for the CUs (measured, tools/loop_scaling.py: 4096 MPSK loops in eight-lane
// waves take twice as long as 2048; 8192 four times) -- with tiles of 32 samples so that 2 x 128 output rows still fit the LDS.
template <int MODE, int G, int TILE>
__global__ __launch_bounds__(128) void loop_kernel(pm_loop *__restrict__ loops, int nloops, int per_row, int rows_lds,
                                                   const double *__restrict__ table,
                                                   const int32_t *__restrict__ pd, const double *__restrict__ x0,
                                                   const double *__restrict__ x1, int64_t x_stride, int64_t n,
                                                   double *__restrict__ o0, double *__restrict__ o1, int64_t out_stride)
{
    extern __shared__ double lds[];
    constexpr int kG = G, kTile = TILE, kPad = TILE + 1;   // (row pitch kPad: rows of different lanes start on different banks)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = blockIdx.x * kG;
    const int ng = min(kG, nloops - g0);
    const int row0 = g0 / per_row;
    const int rows = (g0 + ng - 1) / per_row - row0 + 1;      // distinct input rows of this workgroup's loops
    const int rows_in = rows_lds;
    constexpr bool kTwoOut = MODE == kMpsk || MODE == kQpsk;
    constexpr int NIN = MODE == kMpsk ? 2 : 1, NOUT = kTwoOut ? 2 : 1;
    double2v *tab2 = reinterpret_cast<double2v *>(lds);
    double *in_base = lds + 516;                           // 257 pairs, rounded up to a multiple of 16 bytes
    const int in_buf = rows_in * NIN * kPad;               // doubles per input buffer: [row][NIN][kPad]
    double *out_base = in_base + 2 * in_buf;
    constexpr int out_buf = NOUT * kG * kPad;              // doubles per output buffer: [NOUT][kG][kPad]
    int32_t *pdt = (int32_t *)(out_base + 2 * out_buf);

    for (int i = threadIdx.x; i < 257; i += 128) tab2[i] = double2v{table[i & 255], table[(i + 64) & 255]};
    if (MODE == kMpsk)
        for (int i = threadIdx.x; i < 4096; i += 128) pdt[i] = pd[i];

    const bool active = wave == 0 && lane < ng;
    // the stepping wave is one dependent chain: every issue slot it loses to the FIR waves that share its SIMD (the engine's matched
    // filters run beside the loops) is time added to the run, while the FIR waves lose nothing they cannot make up
    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    LoopRegs L;
    if (active) {
        const pm_loop &s = loops[g0 + lane];
        L.phase_scaling = s.phase_scaling; L.index_scaling = s.index_scaling; L.set_frequency = s.set_frequency;
        L.b0 = s.b0; L.b1 = s.b1; L.a1 = s.a1;
        L.p_rate = s.p_rate; L.i_rate = s.i_rate; L.i_limit = s.i_limit; L.gain = s.gain;
        L.phase = s.phase; L.control = s.control; L.sine = s.sine; L.cosine = s.cosine;
        L.x0 = s.x0; L.x1 = s.x1; L.y0 = s.y0; L.integral = s.integral; L.proportional = s.proportional;
        if (MODE == kQpsk) {
            L.bb0 = s.bb0; L.bb1 = s.bb1; L.ba1 = s.ba1;
            L.cx0 = s.cx0; L.cx1 = s.cx1; L.cy0 = s.cy0; L.sx0 = s.sx0; L.sx1 = s.sx1; L.sy0 = s.sy0;
        }
    }
    const int my_row = active ? (g0 + lane) / per_row - row0 : 0;

    // step s: wave 1 loads tile s and stores tile s - 2, wave 0 steps tile s - 1; the tables above are complete at the first barrier
    const int64_t ntiles = (n + kTile - 1) / kTile;
    for (int64_t s = 0; s < ntiles + 2; ++s) {
        if (wave == 1) {
            if (s < ntiles) {
                const int64_t tile0 = s * kTile;
                const int len = (int)min((int64_t)kTile, n - tile0);
                double *ib = in_base + (s & 1) * in_buf;
                if (kTile >= 64) {
                    // Two rows at a time, every load of the pair issued before the first LDS write (a load per round trip -- what the
                    // plain loop compiles to -- is 32 dependent memory latencies per tile of eight rows: as long as the tile takes to
                    // step).  Indices are clamped instead of predicated: what lands beyond `len` is never read.
                    constexpr int kPer = kTile >= 64 ? kTile / 64 : 1;
                    for (int r = 0; r < rows; r += 2) {
                        const int r1 = min(r + 1, rows - 1);
                        const int64_t off0 = (int64_t)(row0 + r) * x_stride + tile0, off1 = (int64_t)(row0 + r1) * x_stride + tile0;
                        double va[2][NIN][kPer];
#pragma unroll
                        for (int j = 0; j < kPer; ++j) {
                            const int k = min(lane + 64 * j, len - 1);
                            va[0][0][j] = x0[off0 + k];
                            va[1][0][j] = x0[off1 + k];
                            if (MODE == kMpsk) {
                                va[0][NIN - 1][j] = x1[off0 + k];
                                va[1][NIN - 1][j] = x1[off1 + k];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < kPer; ++j) {
                            const int k = lane + 64 * j;
                            ib[(r * NIN) * kPad + k] = va[0][0][j];
                            ib[(r1 * NIN) * kPad + k] = va[1][0][j];
                            if (MODE == kMpsk) {
                                ib[(r * NIN + 1) * kPad + k] = va[0][NIN - 1][j];
                                ib[(r1 * NIN + 1) * kPad + k] = va[1][NIN - 1][j];
                            }
                        }
                    }
                } else {
                    // short tiles: 64 / kTile rows per pass (a part of the wave each), four passes in flight before the first LDS write
                    constexpr int kRows = 64 / kTile, kFly = 4;
                    const int sub = lane / kTile, k = lane % kTile, kc = min(k, len - 1);
                    for (int r = 0; r < rows; r += kRows * kFly) {
                        double va[kFly][NIN];
#pragma unroll
                        for (int j = 0; j < kFly; ++j) {
                            const int rr = min(r + j * kRows + sub, rows - 1);
                            const int64_t off = (int64_t)(row0 + rr) * x_stride + tile0 + kc;
                            va[j][0] = x0[off];
                            if (MODE == kMpsk) va[j][NIN - 1] = x1[off];
                        }
#pragma unroll
                        for (int j = 0; j < kFly; ++j) {
                            const int rr = min(r + j * kRows + sub, rows - 1);      // (rows past the last repeat it: the same value again)
                            ib[(rr * NIN) * kPad + k] = va[j][0];
                            if (MODE == kMpsk) ib[(rr * NIN + 1) * kPad + k] = va[j][NIN - 1];
                        }
                    }
                }
            }
            if (s >= 2) {
                const int64_t tile0 = (s - 2) * kTile;
                const int len = (int)min((int64_t)kTile, n - tile0);
                const double *ob = out_base + (s & 1) * out_buf;
                if (kTile >= 64) {
                    for (int r = 0; r < ng; ++r) {
                        const int64_t off = (int64_t)(g0 + r) * out_stride + tile0;
                        for (int k = lane; k < len; k += 64) {
                            o0[off + k] = ob[r * kPad + k];
                            if (kTwoOut) o1[off + k] = ob[(kG + r) * kPad + k];
                        }
                    }
                } else {
                    // 64 / kTile loops per pass, each part of the wave writing one loop's run of kTile samples (256 contiguous bytes);
                    // four passes' LDS reads before their stores
                    constexpr int kRows = 64 / kTile, kFly = 4;
                    const int sub = lane / kTile, k = lane % kTile;
                    for (int r = 0; r < ng; r += kRows * kFly) {
                        double va[kFly][NOUT];
#pragma unroll
                        for (int j = 0; j < kFly; ++j) {
                            const int rr = min(r + j * kRows + sub, kG - 1);
                            va[j][0] = ob[rr * kPad + k];
                            if (kTwoOut) va[j][NOUT - 1] = ob[(kG + rr) * kPad + k];
                        }
#pragma unroll
                        for (int j = 0; j < kFly; ++j) {
                            const int rr = r + j * kRows + sub;
                            if (rr < ng && k < len) {
                                const int64_t off = (int64_t)(g0 + rr) * out_stride + tile0 + k;
                                o0[off] = va[j][0];
                                if (kTwoOut) o1[off] = va[j][NOUT - 1];
                            }
                        }
                    }
                }
            }
        } else if (active && s >= 1 && s <= ntiles) {
            const int64_t tile0 = (s - 1) * kTile;
            const int len = (int)min((int64_t)kTile, n - tile0);
            const double *p0 = in_base + ((s - 1) & 1) * in_buf + (my_row * NIN) * kPad;
            const double *p1 = p0 + kPad;
            double *q0 = out_base + ((s - 1) & 1) * out_buf + lane * kPad, *q1 = q0 + kG * kPad;
            for (int k = 0; k < len; ++k) loop_step<MODE>(L, tab2, pdt, p0[k], MODE == kMpsk ? p1[k] : 0.0, q0[k], q1[k]);
        }
        __syncthreads();
    }
    if (active) {
        pm_loop &s = loops[g0 + lane];
        s.phase = L.phase; s.control = L.control; s.sine = L.sine; s.cosine = L.cosine;
        s.x0 = L.x0; s.x1 = L.x1; s.y0 = L.y0; s.integral = L.integral; s.proportional = L.proportional;
        if (MODE == kQpsk) { s.cx0 = L.cx0; s.cx1 = L.cx1; s.cy0 = L.cy0; s.sx0 = L.sx0; s.sx1 = L.sx1; s.sy0 = L.sy0; }
    }
}

// The third shape: one wave per workgroup, a loop on every lane, no second wave and no barrier between waves.  A lane steps eight samples
// of its loop at a time, the next eight already in registers and the eight after them in flight from memory.
// Why no tiles: beside the engine's FIR kernels the tiled shapes run 1.8-2.8x slower than alone -- their LDS traffic (tile reads and
// writes of the stepping wave, the I/O wave's copies) queues behind the filters'.  What is in LDS: the tables the recurrence looks up by
// a computed index (NCO pairs, phase detector) and, round 4, for the two-output loops one 5 KB tile per output through which the wave's
// eight-sample blocks are TRANSPOSED on their way to memory (VEC).  A lane owns a row, so a plain eight-byte store touches 64 cache lines
// per instruction and fills each line with eight separate requests; through the tile a store instruction writes 16 whole lines (four
// lanes per row, 16 bytes each): an eighth of the requests, and fewer memory operations in flight per wave (a wave may have 63).
// Measured (profiles/r04_loop_sweep.txt): qpsk_2400 7.8 against 8.6 ms per step; the BPSK loop, which would also LOAD through a tile
// (a row per loop), ran 20 % slower that way -- LDS operations of a wave complete in order, so the NCO's table read, which is on the
// loop's dependent chain, queues behind the tile's -- and keeps its plain accesses.  Only this wave touches its tiles: no barrier.
// AGC (BPSK with one chain per recording, psk.py:168-189: the AGC's output goes straight into the loop): the row holds the band-passed
// samples and the lane steps the envelope follower too (agc.py:61-80), one block of eight samples AHEAD of the loop: the follower and
// the division depend on nothing the loop computes, so their instructions issue in the shadow of the loop's dependent chain -- and the
// AGC'd stream (230 MB written and read per recording, a kernel of its own on the front stream) never exists.
constexpr int kTileRow = 10;                 // doubles per row of a transposing tile: eight samples + 16 bytes (rows 80 bytes apart: no bank conflicts for 16-byte accesses)

// Between a tile's writes and its reads by other lanes OF THE SAME WAVE: LDS operations of a wave complete in issue order, so nothing has
// to be waited for -- the compiler only must not move them across (a __syncthreads() here also waits for every global load and store
// in flight: the prefetched blocks, i.e. a memory round trip per eight samples on the loop's critical path).
__device__ __forceinline__ void tile_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int MODE, bool AGC = false, bool VEC = false>
__global__ __launch_bounds__(64) void loop_direct_kernel(pm_loop *__restrict__ loops, int nloops, int per_row, const double *__restrict__ table,
                                                         const int32_t *__restrict__ pd, const double *__restrict__ x0,
                                                         const double *__restrict__ x1, int64_t x_stride, int64_t n, double *__restrict__ o0,
                                                         double *__restrict__ o1, int64_t out_stride, const double *__restrict__ agc_consts = nullptr,
                                                         AgcDev P = AgcDev{0, 0, 0, 0, 0}, double2 *__restrict__ agc_state = nullptr)
{
    static_assert(!AGC || MODE == kCostas, "the AGC is stepped in the loop's lane for the BPSK Costas loop only");
    extern __shared__ double lds[];
    constexpr bool kTwoOut = MODE == kMpsk || MODE == kQpsk;
    double2v *tab2 = reinterpret_cast<double2v *>(lds);
    int32_t *pdt = (int32_t *)(lds + 516);
    double *tiles = lds + 516 + (MODE == kMpsk ? 2048 : 0);          // VEC: out0 | out1 (64 rows of kTileRow doubles each)
    for (int i = threadIdx.x; i < 257; i += 64) tab2[i] = double2v{table[i & 255], table[(i + 64) & 255]};
    if (MODE == kMpsk)
        for (int i = threadIdx.x; i < 4096; i += 64) pdt[i] = pd[i];
    __syncthreads();
    const int lane = threadIdx.x, l0 = blockIdx.x * 64;
    const bool alive = l0 + lane < nloops;
    if (!VEC && !alive) return;
    const int l = min(l0 + lane, nloops - 1);                        // (VEC: the lanes past the last loop help moving the tiles and step a copy of it, unsaved)
    LoopRegs L;
    {
        const pm_loop &s = loops[l];
        L.phase_scaling = s.phase_scaling; L.index_scaling = s.index_scaling; L.set_frequency = s.set_frequency;
        L.b0 = s.b0; L.b1 = s.b1; L.a1 = s.a1;
        L.p_rate = s.p_rate; L.i_rate = s.i_rate; L.i_limit = s.i_limit; L.gain = s.gain;
        L.phase = s.phase; L.control = s.control; L.sine = s.sine; L.cosine = s.cosine;
        L.x0 = s.x0; L.x1 = s.x1; L.y0 = s.y0; L.integral = s.integral; L.proportional = s.proportional;
        if (MODE == kQpsk) {
            L.bb0 = s.bb0; L.bb1 = s.bb1; L.ba1 = s.ba1;
            L.cx0 = s.cx0; L.cx1 = s.cx1; L.cy0 = s.cy0; L.sx0 = s.sx0; L.sx1 = s.sx1; L.sy0 = s.sy0;
        }
    }
    const int64_t row = l / per_row;
    const double *p0 = x0 + row * x_stride, *p1 = MODE == kMpsk ? x1 + row * x_stride : nullptr;
    double *q0 = o0 + (int64_t)l * out_stride, *q1 = kTwoOut ? o1 + (int64_t)l * out_stride : nullptr;
    double env = 0.0, sustain = 0.0;
    if (AGC) {
        P.att = agc_consts[4 * row + 1];
        P.dec = agc_consts[4 * row + 2];
        env = agc_state[row].x;
        sustain = agc_state[row].y;
    }
    auto agc = [&](double sv) -> double {                    // agc.py:69-76: the follower's step, then buffer[i] = target * s / env
        agc_step(sv, env, sustain, P);
        return env != 0 ? P.target * sv / env : sv;
    };
    __builtin_amdgcn_s_setprio(3);
    constexpr int B = 8;
    const int64_t full = n / B * B;
    // VEC: lane i moves quarter i % 4 (16 bytes) of the block of loop 16 j + i / 4 in pass j = 0..3: four lanes to a cache line
    const int sub = lane & 3, grp = lane >> 2;
    double *t_o0 = tiles, *t_o1 = tiles + 64 * kTileRow;
    auto fetch = [&](int64_t k, double (&d0)[B], double (&d1)[B]) {          // the block of eight samples at k (clamped into the stream) -> registers
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int64_t kk = min(k + j, n - 1);
            d0[j] = p0[kk];
            d1[j] = MODE == kMpsk ? p1[kk] : 0.0;
        }
    };
    auto put = [&](int64_t k0, const double (&r0)[B], const double (&r1)[B]) {
        if (VEC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<double2v *>(t_o0 + lane * kTileRow + 2 * j) = double2v{r0[2 * j], r0[2 * j + 1]};
                if (kTwoOut) *reinterpret_cast<double2v *>(t_o1 + lane * kTileRow + 2 * j) = double2v{r1[2 * j], r1[2 * j + 1]};
            }
            tile_fence();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ll = l0 + 16 * j + grp;
                const double2v a = *reinterpret_cast<const double2v *>(t_o0 + (16 * j + grp) * kTileRow + 2 * sub);
                const double2v b = kTwoOut ? *reinterpret_cast<const double2v *>(t_o1 + (16 * j + grp) * kTileRow + 2 * sub) : double2v{0.0, 0.0};
                if (ll < nloops) {
                    *reinterpret_cast<double2v *>(o0 + (int64_t)ll * out_stride + k0 + 2 * sub) = a;
                    if (kTwoOut) *reinterpret_cast<double2v *>(o1 + (int64_t)ll * out_stride + k0 + 2 * sub) = b;
                }
            }
            tile_fence();
        } else {
#pragma unroll
            for (int j = 0; j < B; ++j) {
                q0[k0 + j] = r0[j];
                if (kTwoOut) q1[k0 + j] = r1[j];
            }
        }
    };
    // c: stepped now | nx: the block after it (AGC: its AGC steps are taken while c is stepped) | f: the block in flight from memory
    double c0[B] = {0}, c1[B] = {0}, n0[B] = {0}, n1[B] = {0}, f0[B] = {0}, f1[B] = {0};
    fetch(0, c0, c1);
    fetch(B, n0, n1);
    if (AGC && full) {
#pragma unroll
        for (int j = 0; j < B; ++j) c0[j] = agc(c0[j]);
    }
    for (int64_t k0 = 0; k0 < full; k0 += B) {
        fetch(k0 + 2 * B, f0, f1);
        const bool more = k0 + 2 * B <= full;                // the next block is a whole one: (AGC) its samples take their AGC steps now
        double r0[B], r1[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            r1[j] = 0.0;
            if (AGC && more) n0[j] = agc(n0[j]);             // independent of the loop's chain: fills its issue gaps
            loop_step<MODE>(L, tab2, pdt, c0[j], c1[j], r0[j], r1[j]);
        }
        put(k0, r0, r1);
#pragma unroll
        for (int j = 0; j < B; ++j) {
            c0[j] = n0[j];
            c1[j] = n1[j];
            n0[j] = f0[j];
            n1[j] = f1[j];
        }
    }
    if (alive) {
        for (int64_t k = full; k < n; ++k) {
            double a = 0.0, b = 0.0;
            const double sv = AGC ? agc(p0[k]) : p0[k];
            loop_step<MODE>(L, tab2, pdt, sv, MODE == kMpsk ? p1[k] : 0.0, a, b);
            q0[k] = a;
            if (kTwoOut) q1[k] = b;
        }
        pm_loop &s = loops[l];
        s.phase = L.phase; s.control = L.control; s.sine = L.sine; s.cosine = L.cosine;
        s.x0 = L.x0; s.x1 = L.x1; s.y0 = L.y0; s.integral = L.integral; s.proportional = L.proportional;
        if (MODE == kQpsk) { s.cx0 = L.cx0; s.cx1 = L.cx1; s.cy0 = L.cy0; s.sx0 = L.sx0; s.sx1 = L.sx1; s.sy0 = L.sy0; }
        if (AGC) agc_state[row] = make_double2(env, sustain);
    }
}

// the most distinct input rows the g consecutive loops of one workgroup can touch
int loop_rows_lds(int per_row, int nloops, int g)
{
    if (per_row >= nloops) return 1;
    if (per_row % g == 0) return 1;
    return std::min(g, (g - 2) / per_row + 2);
}

size_t loop_lds_bytes(int mode, int rows_in, int g, int tile)
{
    const int nin = mode == kMpsk ? 2 : 1, nout = (mode == kMpsk || mode == kQpsk) ? 2 : 1;
    size_t d = 516 + 2 * (size_t)rows_in * nin * (tile + 1) + 2 * (size_t)nout * g * (tile + 1);
    return d * 8 + (mode == kMpsk ? 4096 * 4 : 0);
}

// The launch itself: d_loops are `nloops` loops in DEVICE memory (parameters and state, read and written); nothing is copied and
// nothing waits.
// every lane a loop once eight-lane waves would outnumber the CUs (PM_LOOP_WIDE=0 / 1 / 2 forces the shape: tests, measurements;
// 1 = the tiled 64-loop shape, 2 = the direct one)
int loop_shape(const pm_ctx *ctx, int nloops)
{
    return ctx->tune.loop_wide >= 0 ? ctx->tune.loop_wide : (nloops > 8 * 256 ? 2 : 0);
}

// the eight-sample blocks of the direct shape as 16-byte pieces through the transposing tiles: every row 16-byte aligned (PM_LOOP_VEC=0: off)
bool loop_io_vec(const pm_ctx *ctx, const double *d_x0, int64_t x_stride, const double *d_o0, const double *d_o1, int64_t out_stride, int64_t n)
{
    return ctx->tune.loop_vec != 0 && d_o1 != nullptr && n >= 8 && (((uintptr_t)d_x0 | (uintptr_t)d_o0 | (uintptr_t)d_o1) & 15) == 0 && x_stride % 2 == 0 && out_stride % 2 == 0;
}

template <int MODE>
int loop_enqueue(pm_ctx *ctx, pm_loop *d_loops, int nloops, int per_row, const double *d_table, const int32_t *d_pd,
                 const double *d_x0, const double *d_x1, int64_t x_stride, int64_t n, double *d_o0, double *d_o1, int64_t out_stride)
{
    const int shape = loop_shape(ctx, nloops);
    if (shape == 2) {
        const bool vec = loop_io_vec(ctx, d_x0, x_stride, d_o0, d_o1, out_stride, n);
        const size_t lds2 = 516 * 8 + (MODE == kMpsk ? 4096 * 4 : 0) + (vec ? 2 * 64 * kTileRow * 8 : 0);
        PmProf prof(ctx, PM_K_LOOP);
        if (vec)
            hipLaunchKernelGGL((loop_direct_kernel<MODE, false, true>), dim3((unsigned)pm_cdiv(nloops, 64)), dim3(64), lds2, ctx->stream, d_loops, nloops, per_row,
                               d_table, d_pd, d_x0, d_x1, x_stride, n, d_o0, d_o1, out_stride);
        else
            hipLaunchKernelGGL((loop_direct_kernel<MODE>), dim3((unsigned)pm_cdiv(nloops, 64)), dim3(64), lds2, ctx->stream, d_loops, nloops, per_row, d_table,
                               d_pd, d_x0, d_x1, x_stride, n, d_o0, d_o1, out_stride);
        PM_HIP(hipGetLastError());
        return PM_OK;
    }
    const bool wide = shape == 1;
    const int g = wide ? 64 : 8, tile = wide ? 32 : 256;
    const int rows_lds = loop_rows_lds(per_row, nloops, g);
    size_t lds = loop_lds_bytes(MODE, rows_lds, g, tile);
    if (ctx->tune.loop_lds_min > 0) lds = std::max(lds, (size_t)ctx->tune.loop_lds_min);     // experiment: one workgroup per CU
    auto go = [&](auto kernel) -> int {
        if (lds > 64 * 1024) PM_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PmProf prof(ctx, PM_K_LOOP);
        hipLaunchKernelGGL(kernel, dim3((unsigned)pm_cdiv(nloops, g)), dim3(128), lds, ctx->stream, d_loops, nloops, per_row, rows_lds, d_table,
                           d_pd, d_x0, d_x1, x_stride, n, d_o0, d_o1, out_stride);
        return PM_OK;
    };
    if (int rc = wide ? go(loop_kernel<MODE, 64, 32>) : go(loop_kernel<MODE, 8, 256>)) return rc;
    PM_HIP(hipGetLastError());
    return PM_OK;
}

template <int MODE>
int loop_launch(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table, const int32_t *d_pd,
                const double *d_x0, const double *d_x1, int64_t x_stride, int64_t n,
                double *d_o0, double *d_o1, int64_t out_stride)
{
    PM_CTX(ctx);
    PM_ARG(h_loops && nloops >= 1 && d_table && n >= 0);
    if (n == 0) return PM_OK;
    PM_ARG(d_x0 && d_o0 && (MODE != kMpsk || (d_x1 && d_o1 && d_pd)) && (MODE != kQpsk || d_o1));
    PM_ARG(nloops == 1 || out_stride >= n);
    const size_t bytes = sizeof(pm_loop) * (size_t)nloops;
    if (int rc = pm_scratch_reserve(ctx, bytes)) return rc;
    pm_loop *d_loops = (pm_loop *)ctx->d_scratch;
    PM_HIP(hipMemcpyAsync(d_loops, h_loops, bytes, hipMemcpyHostToDevice, ctx->stream));
    // x_stride == 0: one input for all loops; otherwise loop l reads x + l * x_stride
    if (int rc = loop_enqueue<MODE>(ctx, d_loops, nloops, x_stride == 0 ? nloops : 1, d_table, d_pd, d_x0, d_x1, x_stride, n, d_o0, d_o1, out_stride)) return rc;
    PM_HIP(hipMemcpyAsync(h_loops, d_loops, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

// ---- AGC ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void max_partial_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[256];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double m = x[0];                                   // max() is order independent; seed with a real element
    for (; i < n; i += stride) {
        const double v = x[i];
        if (v > m) m = v;                              // Python max(): keeps the first maximum, NaN never wins unless first
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---- AGC as a chunk-parallel fixed point -------------------------------------------------------------------------------
// The envelope follower (agc.py:26-37) carries two doubles, (envelope, sustain_count).  Whenever a sample exceeds the
// envelope by less than one attack step the envelope is clamped to |s| and sustain_count restarts, which erases the
// history: two runs that both clamp at the same sample are bit-identical from there on.  So, as in the slicer, the buffer
// is cut into chunks (one lane each), every chunk is run from the end state of its predecessor's previous run, and the
// iteration stops when no start state changes -- at which point the chunk runs are the sequential run.  Worst case (no
// common clamp for a long stretch) degrades to sequential cost.  The converged pass writes the envelope, and the division
// buf[i] = target*s/env (agc.py:75-76), which is not part of the recurrence, runs fully parallel.
struct AgcScale {
    double scaled_attack, scaled_decay;
};

// partial[] -> normal, att, dec (agc.py:29,34,67); one thread
__global__ void agc_prepare_kernel(const double *__restrict__ partial, int npartial, AgcScale sc, double *__restrict__ consts)
{
    double m = partial[0];
    for (int i = 1; i < npartial; ++i)
        if (partial[i] > m) m = partial[i];
    consts[0] = m;
    consts[1] = sc.scaled_attack * m;
    consts[2] = sc.scaled_decay * m;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void agc_iter_kernel(const double *__restrict__ buf, int64_t n, int lc, int64_t nchunks,
                                                       const double *__restrict__ consts, AgcDev P,
                                                       const double2 *__restrict__ s_in, double2 *__restrict__ s_out,
                                                       const uint8_t *__restrict__ d_in, uint8_t *__restrict__ d_out,
                                                       double *__restrict__ env_out, int *__restrict__ changed, int iter)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    if (!EMIT) {
        const bool dirty = c == 0 ? (iter == 0) : (d_in[c] != 0);
        if (!dirty) {
            s_out[c + 1] = s_in[c + 1];
            d_out[c + 1] = 0;
            return;
        }
    }
    P.att = consts[1];
    P.dec = consts[2];
    double env = s_in[c].x, sustain = s_in[c].y;
    const int64_t k0 = c * lc, k1 = min(k0 + (int64_t)lc, n);
    // Eight samples at a time, the next eight in flight while these are stepped (two register blocks taking turns, every load
    // unconditional so that the waits can be counted: the block fetched past the end of the chunk is the chunk's last eight
    // again, and is not used): a load per sample, waited for in front of the ten operations that use it, made the recurrence
    // wait for memory -- every lane reads a cache line of its own -- for several times as long as it computes.
    constexpr int kAhead = 8;
    double cur[kAhead], nxt[kAhead];
    int64_t k = k0;
    if (k + kAhead <= k1) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) cur[u] = buf[k + u];
    }
    for (; k + 2 * kAhead <= k1; k += 2 * kAhead) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) nxt[u] = buf[k + kAhead + u];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            agc_step(cur[u], env, sustain, P);
            if (EMIT) env_out[k + u] = env;
        }
        const int64_t ahead = min(k + 2 * kAhead, k1 - kAhead);
#pragma unroll
        for (int u = 0; u < kAhead; ++u) cur[u] = buf[ahead + u];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            agc_step(nxt[u], env, sustain, P);
            if (EMIT) env_out[k + kAhead + u] = env;
        }
    }
    if (k + kAhead <= k1) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            agc_step(cur[u], env, sustain, P);
            if (EMIT) env_out[k + u] = env;
        }
        k += kAhead;
    }
    for (; k < k1; ++k) {
        agc_step(buf[k], env, sustain, P);
        if (EMIT) env_out[k] = env;
    }
    if (!EMIT) {
        const double2 prev = s_in[c + 1];
        const bool ch = __double_as_longlong(prev.x) != __double_as_longlong(env) ||
                        __double_as_longlong(prev.y) != __double_as_longlong(sustain);
        s_out[c + 1] = make_double2(env, sustain);
        d_out[c + 1] = ch ? 1 : 0;
        if (ch && c + 1 < nchunks) atomicAdd(changed, 1);
    } else if (c == nchunks - 1) {
        s_out[nchunks] = make_double2(env, sustain);        // final state, handed back to the caller
    }
}

__global__ void agc_init_kernel(double2 *sa, double2 *sb, uint8_t *da, uint8_t *db, int64_t nchunks, double env0, double sus0, int *changed)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) *changed = 0;
    if (c > nchunks) return;
    // chunk 0 starts from the caller's state (the true one); every other chunk is guessed "fresh" (0, 0)
    const double2 v = c == 0 ? make_double2(env0, sus0) : make_double2(0.0, 0.0);
    sa[c] = v;
    sb[c] = v;
    da[c] = c < nchunks ? 1 : 0;
    db[c] = 0;
}

__global__ __launch_bounds__(256) void agc_scale_kernel(double *__restrict__ buf, const double *__restrict__ env, int64_t n, double target)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double e = env[i], s = buf[i];
        buf[i] = e != 0 ? target * s / e : s;               // agc.py:75-76
    }
}

// ---- AGC over the rows of a batch, chunk by chunk (pm_lbatch) ------------------------------------------------------------------
// A batch of recordings is processed in time chunks with every sequential state carried on the device, so the envelope follower
// needs no chunk-parallel fixed point here: row r's follower simply continues from d_state[r] (one wave per row, lane 0 owns the
// recurrence over an LDS tile, all lanes stream the tile in, divide -- buf[i] = target*s/env is not part of the recurrence,
// agc.py:75-76 -- and stream it out).  `normal` = max(buffer) (agc.py:67) over the WHOLE band-passed recording is a pass of its own
// before the first chunk (rows_max_kernel over every chunk, folded into d_running).
constexpr int kMaxPart = 32;

__global__ __launch_bounds__(256) void rows_max_kernel(const double *__restrict__ x, int64_t x_stride, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[256];
    const double *xr = x + (int64_t)blockIdx.y * x_stride;
    double m = xr[0];                                  // as max_partial_kernel: seed with a real element, strict >
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = xr[i];
        if (v > m) m = v;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

__global__ void rows_max_fold_kernel(const double *__restrict__ partial, int nparts, double *__restrict__ running, int first, int rows)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double m = partial[(int64_t)r * nparts];
    for (int p = 1; p < nparts; ++p)
        if (partial[(int64_t)r * nparts + p] > m) m = partial[(int64_t)r * nparts + p];
    running[r] = (first || m > running[r]) ? m : running[r];
}

// running[r] -> consts[r] = {normal, scaled attack, scaled decay, -} (agc.py:15-16,29,34,67)
__global__ void agc_rows_prepare_kernel(const double *__restrict__ running, int rows, AgcScale sc, double *__restrict__ consts)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const double m = running[r];
    consts[4 * r] = m;
    consts[4 * r + 1] = sc.scaled_attack * m;
    consts[4 * r + 2] = sc.scaled_decay * m;
    consts[4 * r + 3] = 0.0;
}

// One LANE per row, 64 rows per wave, nothing through LDS: lane l loads eight samples of its row (the next eight already in flight),
// steps the envelope recurrence through them (branch-free, so the lanes do not diverge), divides, stores.  History of this kernel:
// one row per wave with lane 0 stepping and all lanes dividing issued the recurrence's instructions for ONE useful lane -- for
// thousands of rows as much vector issue as the chain's matched filters (15 % / 29 % of the GPU time of a qpsk / bpsk engine run);
// sixteen rows per wave with the tiles in LDS fixed that (bpsk_300 +27 %) but ran five times slower beside the engine's FIR
// kernels than alone (61 against 12.7 ms per 131 072-sample chunk of 8192 rows, tools/agc_rows_probe.py): its LDS traffic queued
// behind theirs.  The rows of a wave are a pitch apart, so a load or store touches 64 cache lines -- at eight bytes per 100 ns and
// lane that is nothing.
__global__ __launch_bounds__(64) void agc_rows_kernel(const double *x, int64_t x_stride, double *y, int64_t y_stride, int rows,
                                                      int64_t n, const double *__restrict__ consts, AgcDev P, double2 *__restrict__ state, int prio)
{
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    const double *xr = x + r * x_stride;
    double *yr = y + r * y_stride;
    P.att = consts[4 * r + 1];
    P.dec = consts[4 * r + 2];
    double env = state[r].x, sustain = state[r].y;
    // one dependent chain per lane: issue slots lost to the FIR waves of the other streams are time -- but the carrier loops' chains are
    // twice as long per sample, so on a SIMD that holds both the loop's wave goes first (priority 3; a kernel ends with its slowest wave)
    if (prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    constexpr int B = 8;
    double cur[B], nxt[B];
    const int64_t full = n / B * B;
#pragma unroll
    for (int j = 0; j < B; ++j) cur[j] = xr[min((int64_t)j, n - 1)];
    for (int64_t k0 = 0; k0 < full; k0 += B) {
#pragma unroll
        for (int j = 0; j < B; ++j) nxt[j] = xr[min(k0 + B + j, n - 1)];
        double out[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            agc_step(cur[j], env, sustain, P);
            out[j] = env != 0 ? P.target * cur[j] / env : cur[j];          // agc.py:75-76
        }
#pragma unroll
        for (int j = 0; j < B; ++j) yr[k0 + j] = out[j];
#pragma unroll
        for (int j = 0; j < B; ++j) cur[j] = nxt[j];
    }
    for (int64_t k = full; k < n; ++k) {
        const double sv = xr[k];
        agc_step(sv, env, sustain, P);
        yr[k] = env != 0 ? P.target * sv / env : sv;
    }
    state[r] = make_double2(env, sustain);
}

}  // namespace

// ---- internal entry points of the batch engine (pm_common.h) ---------------------------------------------------------------------
int pm_loops_rows(pm_ctx *ctx, int modem, pm_loop *d_loops, int nloops, int per_row, const double *d_table, const int32_t *d_pd,
                  const double *d_x0, const double *d_x1, int64_t x_stride, int64_t n, double *d_o0, double *d_o1, int64_t out_stride)
{
    PM_CTX(ctx);
    PM_ARG(d_loops && nloops >= 1 && per_row >= 1 && d_table && d_x0 && d_o0 && n >= 0);
    if (n == 0) return PM_OK;
    switch (modem) {
    case PM_MODEM_BPSK: return loop_enqueue<kCostas>(ctx, d_loops, nloops, per_row, d_table, nullptr, d_x0, nullptr, x_stride, n, d_o0, nullptr, out_stride);
    case PM_MODEM_AFSK_PLL: return loop_enqueue<kPll>(ctx, d_loops, nloops, per_row, d_table, nullptr, d_x0, nullptr, x_stride, n, d_o0, nullptr, out_stride);
    case PM_MODEM_QPSK:
        PM_ARG(d_o1 != nullptr);
        return loop_enqueue<kQpsk>(ctx, d_loops, nloops, per_row, d_table, nullptr, d_x0, nullptr, x_stride, n, d_o0, d_o1, out_stride);
    case PM_MODEM_MPSK:
        PM_ARG(d_x1 && d_o1 && d_pd);
        return loop_enqueue<kMpsk>(ctx, d_loops, nloops, per_row, d_table, d_pd, d_x0, d_x1, x_stride, n, d_o0, d_o1, out_stride);
    default: return pm_set_error(PM_ERR_ARG, "pm_loops_rows: modem %d has no carrier loop", modem);
    }
}

bool pm_loops_rows_take_agc(const pm_ctx *ctx, int modem, int nloops, int per_row)
{
    return ctx && modem == PM_MODEM_BPSK && per_row == 1 && loop_shape(ctx, nloops) == 2 && ctx->tune.loop_agc != 0;
}

int pm_loops_rows_agc(pm_ctx *ctx, pm_loop *d_loops, int nloops, const double *d_table, const double *d_x, int64_t x_stride, int64_t n, double *d_o,
                      int64_t out_stride, const pm_agc_params *hp, const double *d_consts, double *d_state)
{
    PM_CTX(ctx);
    PM_ARG(d_loops && nloops >= 1 && d_table && d_x && d_o && n >= 0 && hp && d_consts && d_state && hp->sample_rate > 0);
    if (n == 0) return PM_OK;
    AgcDev P;
    P.sustain_time = hp->sustain_time;
    P.sustain_inc = 1 / hp->sample_rate;                            // agc.py:17
    P.target = hp->target_amplitude;
    P.att = P.dec = 0;
    PmProf prof(ctx, PM_K_LOOP);
    hipLaunchKernelGGL((loop_direct_kernel<kCostas, true>), dim3((unsigned)pm_cdiv(nloops, 64)), dim3(64), 516 * 8, ctx->stream, d_loops, nloops, 1, d_table,
                       (const int32_t *)nullptr, d_x, (const double *)nullptr, x_stride, n, d_o, (double *)nullptr, out_stride, d_consts, P, (double2 *)d_state);
    PM_HIP(hipGetLastError());
    return PM_OK;
}

int pm_rows_max(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, double *d_partial, double *d_running, int first)
{
    PM_CTX(ctx);
    PM_ARG(d_x && d_partial && d_running && rows >= 1 && rows <= 65535 && n >= 1);
    const int parts = (int)std::min<int64_t>(kMaxPart, pm_cdiv(n, 8192));
    PmProf prof(ctx, PM_K_AGC);
    hipLaunchKernelGGL(rows_max_kernel, dim3((unsigned)parts, (unsigned)rows), dim3(256), 0, ctx->stream, d_x, x_stride, n, d_partial);
    hipLaunchKernelGGL(rows_max_fold_kernel, dim3((unsigned)pm_cdiv(rows, 64)), dim3(64), 0, ctx->stream, d_partial, parts, d_running, first, rows);
    PM_HIP(hipGetLastError());
    return PM_OK;
}

int pm_rows_max_parts(void) { return kMaxPart; }

int pm_agc_rows_prepare(pm_ctx *ctx, const double *d_running, int rows, const pm_agc_params *hp, double *d_consts)
{
    PM_CTX(ctx);
    PM_ARG(d_running && d_consts && hp && rows >= 1 && hp->sample_rate > 0);
    AgcScale sc{hp->attack_rate / hp->sample_rate, hp->decay_rate / hp->sample_rate};      // agc.py:15-16
    hipLaunchKernelGGL(agc_rows_prepare_kernel, dim3((unsigned)pm_cdiv(rows, 64)), dim3(64), 0, ctx->stream, d_running, rows, sc, d_consts);
    PM_HIP(hipGetLastError());
    return PM_OK;
}

int pm_agc_rows(pm_ctx *ctx, const double *d_x, int64_t x_stride, double *d_y, int64_t y_stride, int rows, int64_t n, const pm_agc_params *hp,
                const double *d_consts, double *d_state)
{
    PM_CTX(ctx);
    PM_ARG(d_x && d_y && d_consts && d_state && hp && rows >= 1 && n >= 0 && hp->sample_rate > 0);
    if (n == 0) return PM_OK;
    AgcDev P;
    P.sustain_time = hp->sustain_time;
    P.sustain_inc = 1 / hp->sample_rate;                            // agc.py:17
    P.target = hp->target_amplitude;
    P.att = P.dec = 0;
    PmProf prof(ctx, PM_K_AGC);
    hipLaunchKernelGGL(agc_rows_kernel, dim3((unsigned)pm_cdiv(rows, 64)), dim3(64), 0, ctx->stream, d_x, x_stride, d_y, y_stride, rows, n, d_consts, P,
                       (double2 *)d_state, ctx->tune.agc_rows_prio);
    PM_HIP(hipGetLastError());
    return PM_OK;
}

extern "C" {

int pm_costas_bpsk(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table,
                   const double *d_x, int64_t x_stride, int64_t n, double *d_out, int64_t out_stride)
{
    return loop_launch<kCostas>(ctx, h_loops, nloops, d_table, nullptr, d_x, nullptr, x_stride, n, d_out, nullptr, out_stride);
}

int pm_pll_afsk(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table,
                const double *d_x, int64_t x_stride, int64_t n, double *d_out, int64_t out_stride)
{
    return loop_launch<kPll>(ctx, h_loops, nloops, d_table, nullptr, d_x, nullptr, x_stride, n, d_out, nullptr, out_stride);
}

int pm_mpsk_loop(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table, const int32_t *d_pd_table,
                 const double *d_re, const double *d_im, int64_t x_stride, int64_t n,
                 double *d_i_out, double *d_q_out, int64_t out_stride)
{
    return loop_launch<kMpsk>(ctx, h_loops, nloops, d_table, d_pd_table, d_re, d_im, x_stride, n, d_i_out, d_q_out, out_stride);
}

int pm_costas_qpsk(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table,
                   const double *d_x, int64_t x_stride, int64_t n, double *d_i_out, double *d_q_out, int64_t out_stride)
{
    return loop_launch<kQpsk>(ctx, h_loops, nloops, d_table, nullptr, d_x, nullptr, x_stride, n, d_i_out, d_q_out, out_stride);
}

int pm_agc_rows_apply(pm_ctx *ctx, const double *d_x, int64_t x_stride, double *d_y, int64_t y_stride, int rows, int64_t n,
                      const pm_agc_params *hp, const double *h_normal, double *h_state)
{
    PM_CTX(ctx);
    PM_ARG(d_x && d_y && hp && h_normal && h_state && rows >= 1 && n >= 0);
    // work space: running[rows] | consts[4 rows] | state[2 rows]
    if (int rc = pm_scratch_reserve(ctx, sizeof(double) * 7 * (size_t)rows)) return rc;
    double *d_running = (double *)ctx->d_scratch, *d_consts = d_running + rows, *d_state = d_consts + 4 * (size_t)rows;
    PM_HIP(hipMemcpyAsync(d_running, h_normal, sizeof(double) * (size_t)rows, hipMemcpyHostToDevice, ctx->stream));
    PM_HIP(hipMemcpyAsync(d_state, h_state, sizeof(double) * 2 * (size_t)rows, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = pm_agc_rows_prepare(ctx, d_running, rows, hp, d_consts)) return rc;
    if (int rc = pm_agc_rows(ctx, d_x, x_stride, d_y, y_stride, rows, n, hp, d_consts, d_state)) return rc;
    PM_HIP(hipMemcpyAsync(h_state, d_state, sizeof(double) * 2 * (size_t)rows, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

int pm_rows_max_f64(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, double *h_max)
{
    PM_CTX(ctx);
    PM_ARG(d_x && h_max && rows >= 1 && n >= 1);
    const size_t parts = (size_t)pm_rows_max_parts();
    if (int rc = pm_scratch_reserve(ctx, sizeof(double) * (size_t)rows * (parts + 1))) return rc;
    double *d_partial = (double *)ctx->d_scratch, *d_running = d_partial + (size_t)rows * parts;
    if (int rc = pm_rows_max(ctx, d_x, x_stride, rows, n, d_partial, d_running, 1)) return rc;
    PM_HIP(hipMemcpyAsync(h_max, d_running, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

int pm_agc_apply(pm_ctx *ctx, double *d_buf, int64_t n, const pm_agc_params *hp, double *h_state)
{
    PM_CTX(ctx);
    PM_ARG(ctx && hp && h_state && n >= 0);
    if (n == 0) return PM_OK;
    PM_ARG(d_buf != nullptr && hp->sample_rate > 0);
    const int npartial = (int)std::min<int64_t>(1024, pm_cdiv(n, 256));
    // chunk length: ~64 K lanes at most, never shorter than 2048 samples
    const int lc = (int)std::max<int64_t>(2048, pm_cdiv(n, 65536));
    const int64_t nchunks = pm_cdiv(n, lc);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    const size_t e = (size_t)nchunks + 1;
    const size_t o_part = carve((size_t)npartial * 8), o_const = carve(64), o_sa = carve(e * 16), o_sb = carve(e * 16), o_da = carve(e),
                 o_db = carve(e), o_ch = carve(256), o_env = carve((size_t)n * 8);
    if (int rc = pm_scratch_reserve(ctx, off)) return rc;
    char *base = (char *)ctx->d_scratch;
    double *d_partial = (double *)(base + o_part), *d_const = (double *)(base + o_const), *d_env = (double *)(base + o_env);
    double2 *sa = (double2 *)(base + o_sa), *sb = (double2 *)(base + o_sb);
    uint8_t *da = (uint8_t *)(base + o_da), *db = (uint8_t *)(base + o_db);
    int *changed = (int *)(base + o_ch);

    AgcDev P;
    P.sustain_time = hp->sustain_time;
    P.sustain_inc = 1 / hp->sample_rate;                            // agc.py:17
    P.target = hp->target_amplitude;
    P.att = P.dec = 0;
    AgcScale sc{hp->attack_rate / hp->sample_rate, hp->decay_rate / hp->sample_rate};      // agc.py:15-16

    PmProf prof(ctx, PM_K_AGC);
    hipLaunchKernelGGL(max_partial_kernel, dim3(npartial), dim3(256), 0, ctx->stream, d_buf, n, d_partial);
    hipLaunchKernelGGL(agc_prepare_kernel, dim3(1), dim3(1), 0, ctx->stream, d_partial, npartial, sc, d_const);
    const unsigned grid = (unsigned)pm_cdiv(nchunks, 256);
    hipLaunchKernelGGL(agc_init_kernel, dim3((unsigned)pm_cdiv((int64_t)e, 256)), dim3(256), 0, ctx->stream, sa, sb, da, db, nchunks,
                       h_state[0], h_state[1], changed);
    int *h_flag = (int *)ctx->h_pinned;
    int iters = 0;
    bool converged = nchunks == 1;
    if (converged) {           // one chunk: its start state is the caller's, nothing to iterate
        iters = 0;
    }
    while (!converged) {
        for (int b = 0; b < 4; ++b) {
            hipLaunchKernelGGL((agc_iter_kernel<false>), dim3(grid), dim3(256), 0, ctx->stream, d_buf, n, lc, nchunks, d_const, P,
                               sa, sb, da, db, nullptr, changed, iters);
            std::swap(sa, sb);
            std::swap(da, db);
            ++iters;
        }
        PM_HIP(hipMemcpyAsync(h_flag, changed, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(hipMemsetAsync(changed, 0, sizeof(int), ctx->stream));
        PM_HIP(hipStreamSynchronize(ctx->stream));
        converged = (*h_flag == 0);
        if (ctx->tune.agc_trace) fprintf(stderr, "[agc] after %d iterations: %d start states changed (chunks %lld x %d)\n", iters, *h_flag, (long long)nchunks, lc);
        if (!converged && iters > nchunks + 10)
            return pm_set_error(PM_ERR_NOCONVERGE, "AGC fixed point not reached after %d iterations", iters);
    }
    // converged start states are in sa: emit the envelope, then scale in parallel
    hipLaunchKernelGGL((agc_iter_kernel<true>), dim3(grid), dim3(256), 0, ctx->stream, d_buf, n, lc, nchunks, d_const, P,
                       sa, sb, da, db, d_env, changed, iters);
    hipLaunchKernelGGL(agc_scale_kernel, dim3(2048), dim3(256), 0, ctx->stream, d_buf, d_env, n, hp->target_amplitude);
    PM_HIP(hipGetLastError());
    double2 fin;
    PM_HIP(hipMemcpyAsync(&fin, sb + nchunks, sizeof(fin), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    h_state[0] = fin.x;
    h_state[1] = fin.y;
    ctx->sl_iterations = iters;
    return PM_OK;
}

}  // extern "C"
