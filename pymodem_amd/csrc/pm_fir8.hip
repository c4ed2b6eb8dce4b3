// Long matched filters that feed nothing but a slicer -- BPSKModem's RRC (psk.py:193, 961 taps at 48 kHz) and MPSKModem's pair
// (psk.py:750-751, 241 taps) -- as CERTIFIED SIGNS on the int8 matrix pipe, for the batch engine (pm_loopbatch.hip).
//
// slicer.slice reads `sample >= 0` and nothing else (slicer.py:74-106, :215-231), so what has to be the reference's is the sign of
// the canonical sum  y[k] = sum_t h[m-1-t] x[k+t]  (one fma per tap, ascending input index, from +0: fir_acc_image in pm_fir.hip),
// not its value.  In binary64 on the vector pipe that sum is 961 fma per sample: 55 GFLOP per ten-minute recording, the first
// kernel of configs[1] and the second of configs[4] in round 3's profiles.  Here:
//
//   input   the loop's output x, any magnitude: a workgroup takes a bound on the largest |x| of its window, Xmax <= 2^e, scales by the
//           power of two 2^s2 = 2^(22-e) (exact) and rounds to an integer |X| <= 2^22, written as three balanced base-256 digits
//           X = sum_i x_i 256^i, x_i in [-128, 127] -- the bytes of (X + 0x808080) ^ 0x808080 -- one LDS plane per digit;
//   taps    q[t] = rint(h[t] 2^S), |q| <= 2^22, three balanced digits, laid out once per tap set as the Toeplitz band
//           B[c][j] = d[c - j] of v_mfma_i32_16x16x64_i8 (pm_bpf8.hip has the picture), 64 columns per block, 4 J blocks;
//   sums    W_w[k] = sum_t sum_{i+j=w} q_j[t] x_i[k+t] for w = 2, 3, 4: six exact int32 matrix products per block and 256 outputs
//           (|W_w| < 1024 * 3 * 2^14; the three products of weight 1 and 256 are not computed but bounded); the A operand of
//           (tile q, block kb) is 16 bytes of a plane at 256 q + 64 kb + ..., the same bytes for every pair with 4 q + kb equal:
//           one LDS read serves J blocks;
//   value   y~ = 2^-(S+s2) 65536 (W_2 + 256 W_3 + 65536 W_4): an integer below 2^42, recombined exactly in binary64.
//
// What separates y~ from the canonical sum, for every output of the workgroup (plan constants c1, c2):
//   sum|h - q 2^-S| Xmax  (taps)  +  sum|q| 2^-S * 0.75 2^-s2  (samples: half a unit of rounding, a quarter for passing through
//   binary32)  +  m 128^2 (1 + 2 * 256) 2^-(S+s2)  (the products left out)  +  1.01 (m + 1) u sum|h| Xmax  (the canonical sum's own
//   rounding)   =:  E = c1 Xmax + c2 2^-s2.
// |y~| > E  decides the sign for good.  Every other output is flagged in a mask word (one bit per output, no list, no capacity) and
// fir8_exact_kernel recomputes exactly those with the canonical fma chain from the stored loop output: the bitmap is the exact
// kernel's, bit for bit, whatever the input (tests/test_gpu_fir8.py: noise, signals, silence, denormals, NaN and infinities).
// A window of zeros is +0 everywhere (bits set); a window whose Xmax is not finite, or outside 2^-100 .. 2^100, goes to the exact kernel whole.
#include "pm_common.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

typedef int i4 __attribute__((ext_vector_type(4)));

constexpr int kDig = 3;                                   // digits of the taps and of the samples
constexpr int kAcc = kDig;                                // accumulators: the weights 256^2 .. 256^4 (the products below are bounded, not computed)
constexpr int kT = 4, kRounds = 2, kWaves = 4;            // tiles of 256 outputs per wave and round
constexpr int kWgOut = 256 * kT * kRounds * kWaves;       // 8192 outputs per workgroup
constexpr int kMaxBlocks = 16;                            // 64-column blocks of the band: m + 15 <= 1024

struct Fir8Args {
    const double *x;
    int64_t x_stride, x_room;                // x_room: doubles that may be READ from a row's start (>= n; rows inside a pitched block have slack)
    int aligned16;                           // every row starts on a 16-byte boundary
    uint64_t *bits, *mask;
    int64_t bits_stride, mask_stride;        // 64-bit words
    double c1, c2;                           // E = c1 Xmax + c2 2^-s2; the kernel wants it in units of 2^-(S+s2): c1 and c2 come scaled by 2^S
};

// word g (outputs 64 g .. 64 g + 63 of a tile) from the four ballots: their 16-bit fields at 16 g, side by side -- scalar arithmetic
__device__ __forceinline__ uint64_t tile_word(const uint64_t (&b)[4], int g)
{
    return ((b[0] >> (16 * g)) & 0xFFFFull) | (((b[1] >> (16 * g)) & 0xFFFFull) << 16) | (((b[2] >> (16 * g)) & 0xFFFFull) << 32) |
           (((b[3] >> (16 * g)) & 0xFFFFull) << 48);
}

// J: blocks per residue class (the band has 4 J blocks of 64 columns, zero past the taps)
template <int J>
__global__ __launch_bounds__(256, J == 1 ? 3 : 2) void fir8_kernel(Fir8Args A, int64_t n, int64_t nout, const i4 *__restrict__ btab)
{
    constexpr int NB = 4 * J, W = kWgOut + 64 * NB;       // window bytes per plane (the last 16 are never read)
    __shared__ __attribute__((aligned(16))) unsigned char plane[kDig][W];
    __shared__ uint32_t red[kWaves];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t row = blockIdx.y, wg0 = (int64_t)blockIdx.x * kWgOut;
    const double *x = A.x + row * A.x_stride;
    uint64_t *bits = A.bits + row * A.bits_stride, *mask = A.mask + row * A.mask_stride;

    // ---- the window: four consecutive samples per thread and step, held as binary32 until the scale is known (half the registers;
    // the conversion's 2^-24 |x| is a quarter of a quantum, in c2).  The largest magnitude is the largest of the binary32 bit patterns
    // with the sign removed, compared as integers: NaN and infinities (what binary32 cannot hold included) come out on top.
    constexpr int kGroups = W / 4, kPer = (kGroups + 255) / 256;
    typedef double d2 __attribute__((ext_vector_type(2)));
    float v[kPer][4];
    uint32_t hm = 0;
    const bool inside = wg0 + W <= A.x_room;              // the whole window may be read (what lies past n is anything: it only reaches outputs past nout)
    auto take = [&](int s, int q, double xv) {
        const float f = (float)xv;
        v[s][q] = f;
        hm = max(hm, __float_as_uint(f) & 0x7FFFFFFFu);
    };
    if (inside && A.aligned16) {
#pragma unroll
        for (int s = 0; s < kPer; ++s) {
            const int g = t + 256 * s;
            d2 lo = d2{0.0, 0.0}, hi = d2{0.0, 0.0};
            if (256 * (s + 1) <= kGroups || g < kGroups) {
                const d2 *p = reinterpret_cast<const d2 *>(x + wg0 + 4 * (int64_t)g);
                lo = p[0];
                hi = p[1];
            }
            take(s, 0, lo.x); take(s, 1, lo.y); take(s, 2, hi.x); take(s, 3, hi.y);
        }
    } else {
        // the row's last workgroup (or rows that are not 16-byte aligned): every load unconditional at a clamped index, zeros past the end
        const int64_t lim = inside ? wg0 + W : n;
#pragma unroll
        for (int s = 0; s < kPer; ++s) {
            const int64_t gi = wg0 + 4 * (int64_t)(t + 256 * s);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double xv = x[min(gi + q, lim - 1)];
                take(s, q, gi + q < lim ? xv : 0.0);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) hm = max(hm, (uint32_t)__shfl_xor((int)hm, off));
    if (lane == 0) red[wave] = hm;
    __syncthreads();
    hm = max(max(red[0], red[1]), max(red[2], red[3]));
    const int64_t words = (nout + 63) >> 6;               // the row's bitmap words
    const int64_t w0 = wg0 >> 6;                          // this workgroup's first word (kWgOut / 64 = 128 of them)
    // Outside 2^-100 <= Xmax < 2^100 (binary32's range, with room; NaN and infinities are above, zeros below): no matrix pipe for
    // this workgroup.  All zeros: every sum is +0, bits set inside the stream; anything else: the exact kernel takes every output.
    if (hm >= 0x71800000u || hm < 0x0D800000u) {
        int nonzero = 0;                                   // (from the doubles themselves: tiny values are zeros in binary32)
        const int64_t lim = inside ? wg0 + W : n;
        for (int64_t gi = wg0 + t; gi < min(wg0 + W, lim); gi += 256) nonzero |= !(x[gi] == 0.0);
        const bool zeros = !__syncthreads_or(nonzero);
        for (int w = t; w < kWgOut / 64; w += 256) {
            const int64_t gw = w0 + w;
            if (gw >= words) break;
            const int64_t left = nout - gw * 64;
            const uint64_t in = left >= 64 ? ~0ull : ((1ull << left) - 1);
            bits[gw] = zeros ? in : 0ull;
            mask[gw] = zeros ? 0ull : in;
        }
        return;
    }
    const int e = (int)((hm + 1) >> 23) - 126;             // Xmax <= the binary32 value with pattern hm + 1 <= 2^e
    const double mx = (double)__uint_as_float(hm + 1);     // (a bound on the doubles too: rounding to binary32 is monotone)
    const int s2 = 22 - e;
    const float scale = __uint_as_float((uint32_t)(s2 + 127) << 23);      // 2^s2, s2 in [-79, 123]
    // E = c1 Xmax + c2 2^-s2 in units of 2^-(S+s2), then in the units of the recombined sum below (65536 of them), rounded up
    const double Eint = ceil((A.c1 * (mx * ldexp(1.0, s2)) + A.c2) * (1.0 + 1e-9) * (1.0 / 65536.0)) + 1.0;
#pragma unroll
    for (int s = 0; s < kPer; ++s) {
        const int g = t + 256 * s;
        if (256 * (s + 1) <= kGroups || g < kGroups) {
            uint32_t z[4];
            // x 2^s2 is exact, |..| <= 2^22; adding 1.5 2^23 rounds it to an integer in the low bits of the pattern 0x4B400000 + X
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = (__float_as_uint(__builtin_fmaf(v[s][q], scale, 12582912.0f)) - 0x4B400000u + 0x808080u) ^ 0x808080u;
            const uint32_t a01 = __builtin_amdgcn_perm(z[1], z[0], 0x05010400u), a23 = __builtin_amdgcn_perm(z[3], z[2], 0x05010400u);
            const uint32_t p0 = __builtin_amdgcn_perm(a23, a01, 0x05040100u), p1 = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
            const uint32_t p2 = __builtin_amdgcn_perm(z[1], z[0], 0x0c0c0602u) | __builtin_amdgcn_perm(z[3], z[2], 0x06020c0cu);
            *reinterpret_cast<uint32_t *>(&plane[0][4 * g]) = p0;
            *reinterpret_cast<uint32_t *>(&plane[1][4 * g]) = p1;
            *reinterpret_cast<uint32_t *>(&plane[2][4 * g]) = p2;
        }
    }
    __syncthreads();

    const int r = lane & 15, g4 = lane >> 4;
#pragma unroll 1
    for (int round = 0; round < kRounds; ++round) {
        const int tbase = ((round * kWaves + wave) * kT) * 256;            // this wave's first output of the round, within the workgroup
        if (wg0 + tbase >= nout) break;
        // the products of weight 256^2 and up (six of the nine: the other three are bounded in c2): accumulator w - 2
        i4 acc[kT][kAcc];
#pragma unroll
        for (int q = 0; q < kT; ++q)
#pragma unroll
            for (int w = 0; w < kAcc; ++w) acc[q][w] = i4{0, 0, 0, 0};
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            // the blocks kb = c + 4 j of the band, digit by digit: the same for every tile
            i4 B[J][kDig];
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int d = 0; d < kDig; ++d) B[j][d] = btab[(d * NB + (c + 4 * j)) * 64 + lane];
#pragma unroll
            for (int pp = 0; pp < kT + J - 1; ++pp) {
                // 16 bytes of every plane at position p = c + 4 pp (units of 64 bytes): the A operand of every (tile, block) with 4 q + kb = p
                const int at = tbase + 64 * (c + 4 * pp) + 16 * r + 16 * g4;
                i4 a[kDig];
#pragma unroll
                for (int d = 0; d < kDig; ++d) a[d] = *reinterpret_cast<const i4 *>(&plane[d][at]);
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int q = pp - j;
                    if (q < 0 || q >= kT) continue;
#pragma unroll
                    for (int di = 0; di < kDig; ++di)
#pragma unroll
                        for (int dj = 0; dj < kDig; ++dj)
                            if (di + dj >= kDig - 1)
                                acc[q][di + dj - (kDig - 1)] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[di], B[j][dj], acc[q][di + dj - (kDig - 1)], 0, 0, 0);
                }
            }
        }
        // lane (r, g4) holds outputs 16 (4 g4 + u) + r of each tile, u = 0..3: bit 16 u + r of the tile's word g4
#pragma unroll
        for (int q = 0; q < kT; ++q) {
            const int64_t k0 = wg0 + tbase + 256 * q;
            if (k0 >= nout) break;
            uint64_t pos[4], unsure[4];
            const bool whole = k0 + 256 <= nout;
            uint64_t any = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // sum_w 256^(w-2) W_w: an integer below 2^42, exact in binary64
                double val = (double)acc[q][kAcc - 1][u];
#pragma unroll
                for (int w = kAcc - 2; w >= 0; --w) val = __builtin_fma(val, 256.0, (double)acc[q][w][u]);
                const bool in = whole || k0 + 64 * g4 + 16 * u + r < nout;
                pos[u] = __ballot(in && val > 0.0);
                unsure[u] = __ballot(in && !(fabs(val) > Eint));
                any |= unsure[u];
            }
            const int64_t gw = (k0 >> 6) + lane;
            const bool mine = lane < 4 && gw < words;
            const uint64_t b0 = tile_word(pos, 0), b1 = tile_word(pos, 1), b2 = tile_word(pos, 2), b3 = tile_word(pos, 3);
            if (mine) bits[gw] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;      // (an undecided output: whatever; fir8_exact_kernel writes it)
            if (any == 0) {                                // nearly always: nothing for the exact kernel in this tile
                if (mine) mask[gw] = 0ull;
            } else {
                const uint64_t m0 = tile_word(unsure, 0), m1 = tile_word(unsure, 1), m2 = tile_word(unsure, 2), m3 = tile_word(unsure, 3);
                if (mine) mask[gw] = lane == 0 ? m0 : lane == 1 ? m1 : lane == 2 ? m2 : m3;
            }
        }
    }
}

// The outputs the matrix pipe could not decide, recomputed as the reference's sum: the canonical chain (ascending input index, one fma
// per tap, from +0) for each flagged bit.  A workgroup scans 4096 mask words of a row (sixteen per thread, coalesced), collects the
// flagged ones -- a few dozen, typically -- in LDS and hands them out one per thread, so that the chains run side by side in as few
// waves as possible (a chain is m dependent fma behind m loads: latency, whatever the number of lanes beside it).  hrev[t] =
// h[m - 1 - t] is read with a uniform index: scalar loads.
constexpr int kScanSteps = 16, kListCap = 2048 + 256;

__global__ __launch_bounds__(256) void fir8_exact_kernel(const double *__restrict__ xs, int64_t x_stride, const double *__restrict__ hrev, int m,
                                                         uint64_t *__restrict__ bits_all, int64_t bits_stride, const uint64_t *__restrict__ mask_all,
                                                         int64_t mask_stride, int64_t words, int *__restrict__ count)
{
    __shared__ int lst[kListCap];
    __shared__ int cnt;
    const int64_t row = blockIdx.y, wbase = (int64_t)blockIdx.x * (256 * kScanSteps);
    const uint64_t *mask = mask_all + row * mask_stride;
    uint64_t *bits = bits_all + row * bits_stride;
    const double *x = xs + row * x_stride;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int step = 0; step < kScanSteps; ++step) {
        const int wl = step * 256 + threadIdx.x;
        const int64_t w = wbase + wl;
        if (w < words && mask[w] != 0) lst[atomicAdd(&cnt, 1)] = wl;
        __syncthreads();
        const int have = cnt;                              // the same for every thread
        if (have > kListCap - 256 || (step == kScanSteps - 1 && have > 0)) {
            for (int i = threadIdx.x; i < have; i += 256) {
                const int64_t wi = wbase + lst[i];
                uint64_t mk = mask[wi], word = bits[wi];
                while (mk) {
                    const int b = __ffsll((long long)mk) - 1;
                    mk &= mk - 1;
                    const double *p = x + wi * 64 + b;
                    double acc = 0.0;
                    // thirty-two loads in flight, then their thirty-two dependent fma (a load per fma is a memory round trip per tap)
                    int t0 = 0;
                    for (; t0 + 32 <= m; t0 += 32) {
                        double xv[32];
#pragma unroll
                        for (int j = 0; j < 32; ++j) xv[j] = p[t0 + j];
#pragma unroll
                        for (int j = 0; j < 32; ++j) acc = __builtin_fma(hrev[t0 + j], xv[j], acc);
                    }
                    for (; t0 < m; ++t0) acc = __builtin_fma(hrev[t0], p[t0], acc);
                    const uint64_t bit = 1ull << b;
                    word = acc >= 0.0 ? (word | bit) : (word & ~bit);
                    ++mine;
                }
                bits[wi] = word;
            }
            __syncthreads();
            if (threadIdx.x == 0) cnt = 0;
            __syncthreads();
        }
    }
    if (count && mine) atomicAdd(count, mine);
}

}  // namespace

struct pm_fir8_plan {
    int m = 0, J = 0, S = 0, device = 0;
    double c1 = 0, c2 = 0;
    i4 *d_btab = nullptr;
    double *d_hrev = nullptr;
};

int pm_fir8_plan_create(pm_ctx *ctx, const double *h_taps, int m, pm_fir8_plan **out)
{
    PM_CTX(ctx);
    PM_ARG(h_taps != nullptr && out != nullptr && m >= 1);
    *out = nullptr;
    if (m + 15 > 64 * kMaxBlocks) return pm_set_error(PM_ERR_ARG, "int8 matched filter: %d taps do not fit the %d-column band", m, 64 * kMaxBlocks);
    double hmax = 0.0;
    long double habs = 0.0L;
    for (int t = 0; t < m; ++t) {
        if (!std::isfinite(h_taps[t])) return pm_set_error(PM_ERR_ARG, "int8 matched filter: tap %d is not finite", t);
        hmax = std::max(hmax, std::fabs(h_taps[t]));
        habs += std::fabs((long double)h_taps[t]);
    }
    if (hmax == 0.0) return pm_set_error(PM_ERR_ARG, "int8 matched filter: all taps are zero");
    int e = 0;
    (void)std::frexp(hmax, &e);                              // hmax < 2^e
    pm_fir8_plan *p = new pm_fir8_plan();
    p->m = m;
    p->device = ctx->device;
    p->J = ((m + 15 + 63) / 64 + 3) / 4;
    p->S = 22 - e;                                           // |q| <= 2^22: three balanced digits reach +-(2^23 - 2^15 - 2^7 ...)
    const int NB = 4 * p->J;
    std::vector<int8_t> dig((size_t)kDig * m);
    long double tapq = 0.0L, qabs = 0.0L, d0abs = 0.0L, d1abs = 0.0L;
    for (int t = 0; t < m; ++t) {
        const double scaled = std::ldexp(h_taps[t], p->S);   // exact
        const int64_t q = (int64_t)std::llrint(scaled);
        tapq += std::fabs((long double)scaled - (long double)q);
        qabs += std::fabs((long double)q);
        int64_t v = q;
        for (int b = 0; b < kDig; ++b) {
            const int64_t d = ((v + 128) & 255) - 128;
            dig[(size_t)b * m + t] = (int8_t)d;
            if (b == 0) d0abs += (long double)std::llabs(d);
            if (b == 1) d1abs += (long double)std::llabs(d);
            v = (v - d) / 256;
        }
        if (v != 0) { delete p; return pm_set_error(PM_ERR_ARG, "int8 matched filter: tap %d does not fit three digits", t); }
    }
    const double u = 1.1102230246251565e-16;
    // c1 Xmax: the taps' quantisation and the canonical sum's own rounding; c2 2^-s2: the samples' rounding to integers (half a unit
    // each, and a quarter for their passage through binary32: 2^-24 |x| <= 2^-24 2^e) and the three digit products that are not
    // computed, x_0 q_0 + 256 (x_0 q_1 + x_1 q_0): below 128 (sum|q_0| + 256 (sum|q_0| + sum|q_1|)).  The recombination is exact
    // (integers below 2^53).
    // (both in units of 2^-S: the kernel compares integers of weight 2^-(S+s2))
    p->c1 = (double)((tapq + 1.01L * (m + 1) * u * std::ldexp(habs, p->S)) * 1.000001L);
    p->c2 = (double)((0.75L * qabs + 128.0L * (d0abs + 256.0L * (d0abs + d1abs))) * 1.000001L);
    // B[c][j] = hr[c - j], hr[t] = h[m - 1 - t]: lane (j, g), bytes c = 64 kb + 16 g + 0..15 (as pm_bpf8.hip)
    std::vector<int8_t> tab((size_t)kDig * NB * 64 * 16, 0);
    for (int b = 0; b < kDig; ++b)
        for (int kb = 0; kb < NB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 16; ++i) {
                    const int c = 64 * kb + 16 * (lane >> 4) + i, idx = c - (lane & 15);
                    if (idx >= 0 && idx < m) tab[(((size_t)b * NB + kb) * 64 + lane) * 16 + i] = dig[(size_t)b * m + (m - 1 - idx)];
                }
    std::vector<double> hrev((size_t)m);
    for (int t = 0; t < m; ++t) hrev[(size_t)t] = h_taps[m - 1 - t];
    bool ok = hipSetDevice(ctx->device) == hipSuccess && hipMalloc((void **)&p->d_btab, tab.size()) == hipSuccess &&
              hipMalloc((void **)&p->d_hrev, hrev.size() * sizeof(double)) == hipSuccess;
    ok = ok && hipMemcpy(p->d_btab, tab.data(), tab.size(), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_hrev, hrev.data(), hrev.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        pm_fir8_plan_destroy(p);
        return pm_set_error(PM_ERR_HIP, "int8 matched filter: no device memory for the band table");
    }
    *out = p;
    return PM_OK;
}

void pm_fir8_plan_destroy(pm_fir8_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_btab) (void)hipFree(p->d_btab);
    if (p->d_hrev) (void)hipFree(p->d_hrev);
    delete p;
}

int pm_fir8_digit_pairs(void) { return kDig * (kDig + 1) / 2; }      // fir8_kernel: the products of weight 256^(kDig - 1) and up
int pm_fir8_taps(const pm_fir8_plan *p) { return p ? p->m : 0; }

int pm_fir8_rows_signs(pm_ctx *ctx, const pm_fir8_plan *p, const double *d_x, int64_t x_stride, int rows, int64_t n, uint64_t *d_bits,
                       int64_t bits_stride, int *d_count, int64_t x_room)
{
    PM_CTX(ctx);
    PM_ARG(p != nullptr && d_x != nullptr && d_bits != nullptr && rows >= 1 && n >= p->m && p->device == ctx->device);
    PM_ARG(x_room == 0 || x_room >= n);
    const int64_t nout = n - p->m + 1, words = (nout + 63) / 64, wgs = pm_cdiv(nout, (int64_t)kWgOut);
    PM_ARG(bits_stride >= words && wgs < (1LL << 31));
    // Rows are the grid's y dimension (at most 65535): more rows -- the batch engine sends R x C streams through here, up to 2^20 -- go
    // in batches, one after the other on the stream.  The undecided outputs' mask (one word per bitmap word, in the context's work
    // block: this launch pair is its only user) is a batch's, reused by the next: its size is bounded whatever `rows` is.
    constexpr int kRowsPerLaunch = 65535;
    const int per = std::min(rows, kRowsPerLaunch);
    if (int rc = pm_scratch_reserve(ctx, (size_t)per * (size_t)words * 8)) return rc;
    for (int r0 = 0; r0 < rows; r0 += per) {
        const int nr = std::min(per, rows - r0);
        Fir8Args A;
        A.x = d_x + (int64_t)r0 * x_stride;
        A.x_stride = x_stride;
        A.x_room = x_room ? x_room : n;
        A.aligned16 = (((uintptr_t)d_x) & 15) == 0 && (rows == 1 || x_stride % 2 == 0);
        A.bits = d_bits + (int64_t)r0 * bits_stride;
        A.bits_stride = bits_stride;
        A.mask = (uint64_t *)ctx->d_scratch;
        A.mask_stride = words;
        A.c1 = p->c1;
        A.c2 = p->c2;
        {
            PmProf prof(ctx, PM_K_FIR_F64);
            prof.work((double)nr * ((double)n * 8 + (double)nout / 8), 2.0 * p->m * (double)nout * nr);      // the flops of the sums it stands for
            const dim3 grid((unsigned)wgs, (unsigned)nr);
            switch (p->J) {
            case 1: hipLaunchKernelGGL(fir8_kernel<1>, grid, dim3(256), 0, ctx->stream, A, n, nout, p->d_btab); break;
            case 2: hipLaunchKernelGGL(fir8_kernel<2>, grid, dim3(256), 0, ctx->stream, A, n, nout, p->d_btab); break;
            case 3: hipLaunchKernelGGL(fir8_kernel<3>, grid, dim3(256), 0, ctx->stream, A, n, nout, p->d_btab); break;
            default: hipLaunchKernelGGL(fir8_kernel<4>, grid, dim3(256), 0, ctx->stream, A, n, nout, p->d_btab); break;
            }
            PM_HIP(hipGetLastError());
        }
        {
            PmProf prof(ctx, PM_K_SIGNS);
            hipLaunchKernelGGL(fir8_exact_kernel, dim3((unsigned)pm_cdiv(words, 256 * kScanSteps), (unsigned)nr), dim3(256), 0, ctx->stream, A.x, x_stride, p->d_hrev, p->m,
                               A.bits, bits_stride, A.mask, A.mask_stride, words, d_count);
            PM_HIP(hipGetLastError());
        }
    }
    return PM_OK;
}

extern "C" int pm_fir8_rows_signs_f64(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, const double *h_taps, int m,
                                      uint64_t *d_bits, int64_t bits_stride, int64_t *h_recomputed)
{
    PM_CTX(ctx);
    pm_fir8_plan *p = nullptr;
    if (int rc = pm_fir8_plan_create(ctx, h_taps, m, &p)) return rc;
    int *d_count = nullptr;
    int rc = PM_OK, got = 0;
    if (hipMalloc((void **)&d_count, sizeof(int)) != hipSuccess || hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream) != hipSuccess)
        rc = pm_set_error(PM_ERR_HIP, "int8 matched filter: no memory for the counter");
    if (!rc) rc = pm_fir8_rows_signs(ctx, p, d_x, x_stride, rows, n, d_bits, bits_stride, d_count, 0);
    if (!rc && (hipMemcpyAsync(&got, d_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = pm_set_error(PM_ERR_HIP, "int8 matched filter: the launch failed");
    if (d_count) (void)hipFree(d_count);
    pm_fir8_plan_destroy(p);
    if (h_recomputed) *h_recomputed = got;
    return rc;
}
