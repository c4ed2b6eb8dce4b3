// The shared band-pass of an AFSK chain group (afsk.py:151: int16 audio, ~150 real taps) on the int8 matrix pipe, for consumers that
// need a VALUE WITH A BOUND and not the reference's rounding: the certified gain sweeps (pm_fir.hip).  Every other
// caller keeps fir_valid_kernel<short>, the reference's sum in its canonical order.
//
// Why: in binary64 the band-pass is 148 vector fma per sample, 0.165 ms per recording alone and 29 % of the demod stage's fma issue
// slots; f64 and f32 MFMA were measured earlier and do not run beside vector f64 work.  The int8 MFMA does (tools/ubench/mfma_i8.hip:
// 1.4 P mac/s alone, 0.9 P beside 28 T vector fma/s from the same waves), and integer products need no rounding analysis:
//
//   taps      h[t] ~ q[t] 2^-S,  q[t] a 32-bit integer written in four balanced base-256 digits  q = sum_b d_b 256^b,  d_b in [-128, 127]
//             (round 3: 48 bits in six digits, 36 products per tile; the error this adds to the sweeps' bound is a hundredth of what
//             their own low-pass digits add, profiles/r04_sweep_probe.txt)
//   samples   x = 256 s1 + s0 + 128  with  s1 = x >> 8  and  s0 = (x & 255) - 128,  both in [-128, 127]
//   y[k] = sum_t h[K-1-t] x[k+t]  ~  2^-S ( sum_w 256^w W_w[k] )  +  128 2^-S sum_t q[t],     W_w = sum_t ( d_w s0 + d_(w-1) s1 )[k+t]
//
// The five W_w are exact int32 sums (|W_w| < 148 * 2 * 2^14): 8 int8 products per tap and sample instead of one f64 fma, 24
// v_mfma_i32_16x16x64_i8 per 256 outputs (384 matrix cycles against 2368 vector cycles).  As a matrix product: the tile's outputs
// y[T + 16 i + j] = sum_c A[i][c] B[c][j] with A[i][c] = s[T + 16 i + c] (16 consecutive bytes of a digit plane per lane: one
// ds_read_b128) and the Toeplitz band B[c][j] = d[c - j], c < 192, prepared once per tap set by the host and held in registers.
// What separates the result from the reference's sum:  sum_t |h[t] - q[t] 2^-S| * 32768  (quantisation, ~148 * 2^-33 * 32768)  +  the
// roundings of the recombination  +  the reference's own (K + 1) u sum|h| 32768: pm_bpf8_error() returns the sum, the sweep adds
// sqrt2 m times that to the bound of its sliding magnitudes -- 0.1 % of the certified decision's slack, no sample more goes to the
// exact recomputation, which now starts from the audio (sweep_exact_kernel<true>).
//
// Second consumer (round 4): the carrier-loop engine's max(band-passed recording) (bpf8_max_kernel below) -- up to 241 taps (four
// 64-column blocks of the band instead of three) and two tap digits (its values only choose candidates; the reference's sum decides).
#include "pm_common.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

typedef int i4 __attribute__((ext_vector_type(4)));

constexpr int kMaxDigits = 4, kMaxBlocks = 4, kMaxWeights = kMaxDigits + 1;        // 64 band columns per block: three hold 177 taps, four 241
constexpr int kTilesPerWave = 4, kWaves = 4, kWgOut = 256 * kTilesPerWave * kWaves;      // 4096 outputs per workgroup
template <int KB> struct Geo {
    static constexpr int span = kWgOut + 64 * KB + 240 - 256 + 16;                      // bytes of a digit plane a workgroup reads
    static constexpr int plane = (span + 15) / 16 * 16;
};

struct Scales { double s[kMaxWeights]; double c0; };

// the two digit planes of samples [wg0, wg0 + PLANE) of x into LDS, eight samples per thread and step; -> the OR of the samples' bits
template <int PLANE>
__device__ __forceinline__ uint32_t stage_planes(const int16_t *__restrict__ x, int64_t n, int64_t wg0, int t, unsigned char (*plane)[PLANE])
{
    uint32_t any = 0;
    for (int p = t * 8; p < PLANE; p += 256 * 8) {
        const int64_t gi = wg0 + p;
        uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        if (gi + 8 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(x + gi);
            d0 = v.x; d1 = v.y; d2 = v.z; d3 = v.w;
        } else {
            uint16_t s[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) s[q] = gi + q < n ? (uint16_t)x[gi + q] : (uint16_t)0;
            d0 = s[0] | (uint32_t)s[1] << 16; d1 = s[2] | (uint32_t)s[3] << 16; d2 = s[4] | (uint32_t)s[5] << 16; d3 = s[6] | (uint32_t)s[7] << 16;
        }
        any |= d0 | d1 | d2 | d3;
        const uint32_t lo0 = __builtin_amdgcn_perm(d1, d0, 0x06040200u) ^ 0x80808080u, lo1 = __builtin_amdgcn_perm(d3, d2, 0x06040200u) ^ 0x80808080u;
        const uint32_t hi0 = __builtin_amdgcn_perm(d1, d0, 0x07050301u), hi1 = __builtin_amdgcn_perm(d3, d2, 0x07050301u);
        *reinterpret_cast<uint2 *>(&plane[0][p]) = make_uint2(lo0, lo1);
        *reinterpret_cast<uint2 *>(&plane[1][p]) = make_uint2(hi0, hi1);
    }
    return any;
}

// one tile: lane (r, g) gets outputs tl + 16 (4 g + v) + r, v = 0..3, of the workgroup
template <int KB, int D, int PLANE>
__device__ __forceinline__ void tile_values(const unsigned char (*plane)[PLANE], const i4 (&B)[D][KB], int tl, int lane, const Scales &sc, double (&val)[4])
{
    constexpr int kDigits = D, kWeights = D + 1;
    const int r = lane & 15, g = lane >> 4;
    i4 acc[kWeights];
#pragma unroll
    for (int w = 0; w < kWeights; ++w) acc[w] = i4{0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int at = tl + 16 * r + 64 * kb + 16 * g;
        const i4 a0 = *reinterpret_cast<const i4 *>(&plane[0][at]), a1 = *reinterpret_cast<const i4 *>(&plane[1][at]);
#pragma unroll
        for (int b = 0; b < kDigits; ++b) {
            acc[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, B[b][kb], acc[b], 0, 0, 0);
            acc[b + 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, B[b][kb], acc[b + 1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double x = sc.c0;
#pragma unroll
        for (int w = 0; w < kWeights; ++w) x = __builtin_fma((double)acc[w][v], sc.s[w], x);
        val[v] = x;
    }
}

template <int KB, int D>
__global__ __launch_bounds__(256) void bpf8_kernel(const int16_t *__restrict__ x, int64_t n, const i4 *__restrict__ btab, Scales sc,
                                                   double *__restrict__ y, int64_t nout, int *__restrict__ clear, int nclear)
{
    constexpr int kPlane = Geo<KB>::plane;
    __shared__ __attribute__((aligned(16))) unsigned char plane[2][kPlane];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (blockIdx.x == 0 && t < nclear) clear[t] = 0;         // the recording's sweep counters (pm_sweep_cells)
    const int64_t wg0 = (int64_t)blockIdx.x * kWgOut;
    (void)stage_planes<kPlane>(x, n, wg0, t, plane);
    // the band, digit by digit and block by block: operands of four registers, the same for every tile
    i4 B[D][KB];
#pragma unroll
    for (int b = 0; b < D; ++b)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) B[b][kb] = btab[(b * KB + kb) * 64 + lane];
    __syncthreads();
    const int r = lane & 15, g = lane >> 4;
#pragma unroll 1
    for (int q = 0; q < kTilesPerWave; ++q) {
        const int tl = (wave * kTilesPerWave + q) * 256;               // the tile's first output, within the workgroup
        if (wg0 + tl >= nout) break;
        double val[4];
        tile_values<KB, D, kPlane>(plane, B, tl, lane, sc, val);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t k = wg0 + tl + 16 * (4 * g + v) + r;
            if (k < nout) y[k] = val[v];
        }
    }
}

// ---- max(band-passed recording), the AGC's `normal` (agc.py:67), without the band-passed recording ---------------------------------
// The carrier-loop engine (pm_loopbatch.hip) needs the maximum of every recording's band-pass before its first chunk; it used to run
// the reference's sum over the whole recording for it (a third of the engine's filter work on bpsk_300).  Here the matrix pipe
// produces values y^ with |y^ - y| <= e (e = pm_bpf8_error()), and the maximum of the reference's y comes out EXACTLY:
//   the workgroup's largest y^ is M;  every output whose reference value could be the workgroup's largest has y^ >= M - 2e  (its own
//   y^ >= y - e >= ymax - e >= (M - e) - e);  those few -- usually one -- are recomputed by the reference's chain (one fma per tap,
//   ascending input index, fir_valid_kernel's order) from the int16 audio, and the largest of them is the workgroup's exact maximum.
// Workgroups fold into the row's maximum with one 64-bit atomic max on an order-preserving key; a workgroup whose M + e is below the
// row's maximum so far cannot raise it and recomputes nothing (after the first wave of workgroups that is nearly all of them).  A
// workgroup of digital silence contributes +0 (every fma of the chain returns +0) without any of this: otherwise all 4096 of its
// outputs would tie.  max() of the reference keeps the first of equal values, which for a VALUE is the same thing; no NaN can occur
// (int16 input, finite taps), and -0 cannot either (the chain starts from +0).
__device__ __forceinline__ unsigned long long max_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double max_unkey(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}

template <int KB, int D>
__global__ __launch_bounds__(256) void bpf8_max_kernel(const int16_t *const *__restrict__ rows, int row0, int64_t n, const i4 *__restrict__ btab, Scales sc,
                                                       double err, const double *__restrict__ taps, int m, unsigned long long *__restrict__ keys,
                                                       unsigned long long *__restrict__ redone)
{
    constexpr int kPlane = Geo<KB>::plane;
    __shared__ __attribute__((aligned(16))) unsigned char plane[2][kPlane];
    __shared__ double wmax[kWaves];
    __shared__ double hr[64 * kMaxBlocks];                    // the taps in the order the chain visits them
    __shared__ int nonzero;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // rows are the FAST grid dimension: the workgroups in flight at any time belong to different rows, and a row's later workgroups
    // find the maximum its earlier ones left (with the workgroups of one row side by side, each of them starts from nothing and
    // every one of the first few hundred runs the exact chain)
    const int row = row0 + (int)blockIdx.x;
    const int16_t *__restrict__ x = rows[row];
    const int64_t nout = n - m + 1, wg0 = (int64_t)blockIdx.y * kWgOut;
    if (t == 0) nonzero = 0;
    if (t < m) hr[t] = taps[m - 1 - t];
    __syncthreads();
    const uint32_t any = stage_planes<kPlane>(x, n, wg0, t, plane);
    if (any) nonzero = 1;                                     // (benign race: every writer stores 1)
    i4 B[D][KB];
#pragma unroll
    for (int b = 0; b < D; ++b)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) B[b][kb] = btab[(b * KB + kb) * 64 + lane];
    __syncthreads();
    unsigned long long *key = keys + row;
    if (!nonzero) {
        if (t == 0) atomicMax(key, max_key(0.0));
        return;
    }
    const int r = lane & 15, g = lane >> 4;
    const double ninf = -__builtin_huge_val();
    double val[kTilesPerWave][4];
    double best = ninf;
#pragma unroll
    for (int q = 0; q < kTilesPerWave; ++q) {
        const int tl = (wave * kTilesPerWave + q) * 256;
        if (wg0 + tl < nout) tile_values<KB, D, kPlane>(plane, B, tl, lane, sc, val[q]);      // (uniform per wave)
        if (wg0 + kWgOut > nout) {                           // the row's last workgroup: outputs past the end never count
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (!(wg0 + tl < nout) || wg0 + tl + 16 * (4 * g + v) + r >= nout) val[q][v] = ninf;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) best = val[q][v] > best ? val[q][v] : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    if (lane == 0) wmax[wave] = best;
    __syncthreads();
    double M = wmax[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) M = wmax[w] > M ? wmax[w] : M;
    const double sofar = max_unkey(__atomic_load_n(key, __ATOMIC_RELAXED));      // a reference value of this row, or nothing yet:
    // key 0 decodes to a NaN pattern: nothing yet -> every comparison below is false -> no pruning
    if (M + err < sofar) return;
    double cut = M - 2.0 * err;
    if (sofar - err > cut) cut = sofar - err;                // an output below the row's maximum so far by more than e cannot raise it
    double exact = ninf;
    int mine = 0;
#pragma unroll
    for (int q = 0; q < kTilesPerWave; ++q)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (val[q][v] >= cut) {
                // the samples come back out of the digit planes (x = 256 s1 + s0 + 128, exactly): no trip to memory inside the chain
                const int kl = (wave * kTilesPerWave + q) * 256 + 16 * (4 * g + v) + r;
                const signed char *p0 = reinterpret_cast<const signed char *>(&plane[0][kl]), *p1 = reinterpret_cast<const signed char *>(&plane[1][kl]);
                double acc = 0.0;
#pragma unroll 8
                for (int i = 0; i < m; ++i) acc = __builtin_fma(hr[i], (double)(256 * (int)p1[i] + (int)p0[i] + 128), acc);
                exact = acc > exact ? acc : exact;
                ++mine;
            }
    if (mine) {
        atomicMax(key, max_key(exact));
        if (redone) atomicAdd(redone, (unsigned long long)mine);
    }
}

__global__ void bpf8_max_finish_kernel(const unsigned long long *__restrict__ keys, int rows, double *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) out[r] = max_unkey(keys[r]);
}

}  // namespace

#ifndef PM_BPF8_MAX_DIGITS
#define PM_BPF8_MAX_DIGITS 2
#endif
// Taps of the maximum's plan (pm_bpf8_rows_max) in two digits: its values only have to find the few outputs that could be the largest,
// and a bound 2^16 times wider (about one unit of the 16-bit input) still leaves a handful of candidates per row -- for half the products
constexpr int kMaxDigitsOfMax = PM_BPF8_MAX_DIGITS;
int pm_bpf8_max_digits(void) { return kMaxDigitsOfMax; }

struct pm_bpf8_plan {
    int m = 0, kb = 3, digits = 4;                           // digits: base-256 digits of a quantised tap (4: the sweeps' band-pass; 2: the maximum's)
    // kb: 64-column blocks of the band (3: up to 177 taps, 4: up to 241)
    double err = 0;
    Scales sc{};
    i4 *d_btab = nullptr;
    double *d_taps = nullptr;                                // the taps themselves, for the exact chain of pm_bpf8_rows_max
    int device = 0;
};

int pm_bpf8_plan_create(pm_ctx *ctx, const double *h_taps, int m, pm_bpf8_plan **out, int digits)
{
    PM_CTX(ctx);
    PM_ARG(h_taps != nullptr && out != nullptr && m >= 1 && (digits == 2 || digits == 4));
    const int kDigits = digits, kWeights = digits + 1;
    *out = nullptr;
    if (m + 15 > 64 * kMaxBlocks) return pm_set_error(PM_ERR_ARG, "int8 band-pass: %d taps do not fit the %d-column band", m, 64 * kMaxBlocks);
    const int kBlocks = m + 15 <= 192 ? 3 : 4;
    double hmax = 0.0, habs = 0.0;
    for (int t = 0; t < m; ++t) {
        if (!std::isfinite(h_taps[t])) return pm_set_error(PM_ERR_ARG, "int8 band-pass: tap %d is not finite", t);
        hmax = std::max(hmax, std::fabs(h_taps[t]));
        habs += std::fabs(h_taps[t]);
    }
    if (hmax == 0.0) return pm_set_error(PM_ERR_ARG, "int8 band-pass: all taps are zero");
    int e = 0;
    (void)std::frexp(hmax, &e);                              // hmax = f 2^e, f in [0.5, 1)
    const int S = 8 * kDigits - 2 - e;                       // |q| <= 2^30 (+ 1/2): four balanced digits reach +-2^31 (two: 2^14, +-32639)
    std::vector<int64_t> q(m);
    double quant = 0.0;
    int64_t qsum = 0;
    std::vector<int8_t> dig((size_t)kDigits * m);
    double part = 0.0;                                       // bound on the sum of the magnitudes of the recombination's terms
    for (int t = 0; t < m; ++t) {
        const double scaled = std::ldexp(h_taps[t], S);      // exact
        q[t] = (int64_t)std::llrint(scaled);
        quant += std::fabs(scaled - (double)q[t]);           // both below 2^53: the difference is exact
        qsum += q[t];
        int64_t v = q[t];
        for (int b = 0; b < kDigits; ++b) {
            const int64_t d = ((v + 128) & 255) - 128;
            dig[(size_t)b * m + t] = (int8_t)d;
            v = (v - d) / 256;
        }
        if (v != 0) return pm_set_error(PM_ERR_ARG, "int8 band-pass: tap %d does not fit its digits", t);
    }
    pm_bpf8_plan *p = new pm_bpf8_plan();
    p->m = m;
    p->kb = kBlocks;
    p->digits = digits;
    p->device = ctx->device;
    for (int w = 0; w < kWeights; ++w) {
        p->sc.s[w] = std::ldexp(1.0, 8 * w - S);
        double mag = 0.0;
        for (int t = 0; t < m; ++t) {
            if (w < kDigits) mag += std::fabs((double)dig[(size_t)w * m + t]);
            if (w >= 1) mag += std::fabs((double)dig[(size_t)(w - 1) * m + t]);
        }
        part += mag * 128.0 * p->sc.s[w];
    }
    p->sc.c0 = std::ldexp((double)(128 * qsum), -S);             // |128 sum q| < 2^62; its conversion is one of the nine roundings below
    part += std::fabs(p->sc.c0);
    const double u = 1.1102230246251565e-16;
    // quantisation (|x| <= 32768) + the constant's and the kWeights fma's roundings + the reference's own sum (K + 1) u sum|h| 32768
    p->err = (std::ldexp(quant, -S) * 32768.0 + (kWeights + 2.0) * u * part + (m + 1) * u * habs * 32768.0) * 1.000001;      // (the bound's own roundings)
    // B[c][j] = hr[c - j], hr[t] = h[m - 1 - t] (the reference's sum ascends through the input): lane (j, g), bytes c = 64 kb + 16 g + 0..15
    std::vector<int8_t> tab((size_t)kDigits * kBlocks * 64 * 16, 0);
    for (int b = 0; b < kDigits; ++b)
        for (int kb = 0; kb < kBlocks; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 16; ++i) {
                    const int c = 64 * kb + 16 * (lane >> 4) + i, idx = c - (lane & 15);
                    if (idx >= 0 && idx < m) tab[(((size_t)b * kBlocks + kb) * 64 + lane) * 16 + i] = dig[(size_t)b * m + (m - 1 - idx)];
                }
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void **)&p->d_btab, tab.size()) != hipSuccess) {
        delete p;
        return pm_set_error(PM_ERR_HIP, "int8 band-pass: no device memory for the band table");
    }
    if (hipMemcpy(p->d_btab, tab.data(), tab.size(), hipMemcpyHostToDevice) != hipSuccess || hipMalloc((void **)&p->d_taps, sizeof(double) * (size_t)m) != hipSuccess ||
        hipMemcpy(p->d_taps, h_taps, sizeof(double) * (size_t)m, hipMemcpyHostToDevice) != hipSuccess) {
        pm_bpf8_plan_destroy(p);
        return pm_set_error(PM_ERR_HIP, "int8 band-pass: copying the band table failed");
    }
    *out = p;
    return PM_OK;
}

void pm_bpf8_plan_destroy(pm_bpf8_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_btab) (void)hipFree(p->d_btab);
    if (p->d_taps) (void)hipFree(p->d_taps);
    delete p;
}

double pm_bpf8_error(const pm_bpf8_plan *p) { return p ? p->err : 0.0; }
// what a kernel that runs the band-pass as a stage of its own needs (afsk_fused8_kernel, pm_fir.hip): blocks, band table, scales
int pm_bpf8_plan_view(const pm_bpf8_plan *p, int *kb, const void **d_btab, double *scales6)
{
    if (!p || p->digits != kMaxDigits || !kb || !d_btab || !scales6) return pm_set_error(PM_ERR_ARG, "pm_bpf8_plan_view: not a four-digit plan");
    *kb = p->kb;
    *d_btab = p->d_btab;
    for (int w = 0; w < kMaxWeights; ++w) scales6[w] = p->sc.s[w];
    scales6[kMaxWeights] = p->sc.c0;
    return PM_OK;
}
int pm_bpf8_digit_pairs(void) { return 2 * kMaxDigits; }      // tile_values: two sample digits x the tap digits of the sweeps' plan, every pair computed
int pm_bpf8_taps(const pm_bpf8_plan *p) { return p ? p->m : 0; }

int pm_bpf8_run(pm_ctx *ctx, const pm_bpf8_plan *p, const int16_t *d_audio, int64_t n, double *d_y, int *d_clear, int nclear)
{
    PM_ARG(nclear >= 0 && nclear <= 64 && (nclear == 0 || d_clear != nullptr));
    PM_CTX(ctx);
    PM_ARG(p != nullptr && d_audio != nullptr && d_y != nullptr && n >= p->m && p->device == ctx->device);
    PM_ARG(((uintptr_t)d_audio & 15) == 0 && p->digits == 4);
    const int64_t nout = n - p->m + 1, wgs = pm_cdiv(nout, (int64_t)kWgOut);
    PM_ARG(wgs < (1LL << 31));
    PmProf prof(ctx, PM_K_FIR_I16);
    prof.work((double)n * 2 + (double)nout * 8, 2.0 * p->m * (double)nout);       // the flops of the sum it stands for
    if (p->kb == 3) hipLaunchKernelGGL((bpf8_kernel<3, 4>), dim3((unsigned)wgs), dim3(256), 0, ctx->stream, d_audio, n, p->d_btab, p->sc, d_y, nout, d_clear, nclear);
    else hipLaunchKernelGGL((bpf8_kernel<4, 4>), dim3((unsigned)wgs), dim3(256), 0, ctx->stream, d_audio, n, p->d_btab, p->sc, d_y, nout, d_clear, nclear);
    PM_HIP(hipGetLastError());
    return PM_OK;
}

// d_out[r] = max over the band-pass of row r's n samples, exactly the reference's (see bpf8_max_kernel).  d_keys: rows 64-bit words of
// work space; d_redone (may be null): += the outputs that went through the exact chain.  Rows are 16-byte aligned.
int pm_bpf8_rows_max(pm_ctx *ctx, const pm_bpf8_plan *p, const int16_t *const *d_rows, int rows, int64_t n, unsigned long long *d_keys, double *d_out,
                     unsigned long long *d_redone)
{
    PM_CTX(ctx);
    PM_ARG(p != nullptr && d_rows != nullptr && d_keys != nullptr && d_out != nullptr && rows >= 1 && n >= p->m && p->device == ctx->device);
    PM_ARG(p->digits == kMaxDigitsOfMax);
    const int64_t nout = n - p->m + 1, wgs = pm_cdiv(nout, (int64_t)kWgOut);
    PM_ARG(wgs <= 65535);                                     // (268 M samples per row; checked before anything is enqueued)
    PM_HIP(hipMemsetAsync(d_keys, 0, sizeof(unsigned long long) * (size_t)rows, ctx->stream));       // key 0: below every value
    PmProf prof(ctx, PM_K_FIR_I16);
    prof.work((double)rows * (double)n * 2, 2.0 * p->m * (double)nout * rows);
    for (int r0 = 0; r0 < rows; r0 += 1 << 20) {
        const dim3 grid((unsigned)std::min(1 << 20, rows - r0), (unsigned)wgs);
        if (p->kb == 3) hipLaunchKernelGGL((bpf8_max_kernel<3, kMaxDigitsOfMax>), grid, dim3(256), 0, ctx->stream, d_rows, r0, n, p->d_btab, p->sc, p->err, p->d_taps, p->m, d_keys, d_redone);
        else hipLaunchKernelGGL((bpf8_max_kernel<4, kMaxDigitsOfMax>), grid, dim3(256), 0, ctx->stream, d_rows, r0, n, p->d_btab, p->sc, p->err, p->d_taps, p->m, d_keys, d_redone);
    }
    hipLaunchKernelGGL(bpf8_max_finish_kernel, dim3((unsigned)pm_cdiv(rows, 256)), dim3(256), 0, ctx->stream, d_keys, rows, d_out);
    PM_HIP(hipGetLastError());
    return PM_OK;
}

// ---- the low-pass plan of the certified sweeps (kernel: afsk_slide_lpf8_kernel, pm_fir.hip) ---------------------------------------
int pm_lpf8_plan_create(pm_ctx *ctx, const double *h_taps, int ml, pm_lpf8_plan **out)
{
    PM_CTX(ctx);
    PM_ARG(h_taps != nullptr && out != nullptr && ml >= 1);
    *out = nullptr;
    constexpr int kD = 3, kB = 2;
    if (ml + 15 > 64 * kB) return pm_set_error(PM_ERR_ARG, "int8 low-pass: %d taps do not fit the %d-column band", ml, 64 * kB);
    double hmax = 0.0;
    for (int t = 0; t < ml; ++t) {
        if (!std::isfinite(h_taps[t])) return pm_set_error(PM_ERR_ARG, "int8 low-pass: tap %d is not finite", t);
        hmax = std::max(hmax, std::fabs(h_taps[t]));
    }
    if (hmax == 0.0) return pm_set_error(PM_ERR_ARG, "int8 low-pass: all taps are zero");
    int e = 0;
    (void)std::frexp(hmax, &e);
    pm_lpf8_plan *p = new pm_lpf8_plan();
    p->ml = ml;
    p->S = 22 - e;                                           // |q| <= 2^22: three balanced digits reach beyond +-2^23 - 2^15
    p->hmax = hmax;
    p->device = ctx->device;
    std::vector<int8_t> dig((size_t)kD * ml);
    long double tapq = 0.0L, qabs = 0.0L, d0 = 0.0L, d1 = 0.0L;
    for (int t = 0; t < ml; ++t) {
        const double scaled = std::ldexp(h_taps[t], p->S);   // exact
        const int64_t q = (int64_t)std::llrint(scaled);
        tapq += std::fabs((long double)scaled - (long double)q);
        qabs += std::fabs((long double)q);
        int64_t v = q;
        for (int b = 0; b < kD; ++b) {
            const int64_t d = ((v + 128) & 255) - 128;
            dig[(size_t)b * ml + t] = (int8_t)d;
            if (b == 0) d0 += (long double)std::llabs(d);
            if (b == 1) d1 += (long double)std::llabs(d);
            v = (v - d) / 256;
        }
        if (v != 0) { delete p; return pm_set_error(PM_ERR_ARG, "int8 low-pass: tap %d does not fit three digits", t); }
    }
    p->tapq_int = (double)(tapq * 1.000001L);
    p->qabs = (double)(qabs * 1.000001L);
    p->dlow = (double)(128.0L * d0 * 1.000001L);             // |x_0 q_0| summed over the taps, |x_0| <= 128: the one digit product the kernel leaves out
    (void)d1;
    std::vector<int8_t> tab((size_t)kD * kB * 64 * 16, 0);
    for (int b = 0; b < kD; ++b)
        for (int kb = 0; kb < kB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 16; ++i) {
                    const int c = 64 * kb + 16 * (lane >> 4) + i, idx = c - (lane & 15);
                    if (idx >= 0 && idx < ml) tab[(((size_t)b * kB + kb) * 64 + lane) * 16 + i] = dig[(size_t)b * ml + (ml - 1 - idx)];
                }
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&p->d_btab, tab.size()) != hipSuccess) {
        delete p;
        return pm_set_error(PM_ERR_HIP, "int8 low-pass: no device memory for the band table");
    }
    if (hipMemcpy(p->d_btab, tab.data(), tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->d_btab);
        delete p;
        return pm_set_error(PM_ERR_HIP, "int8 low-pass: copying the band table failed");
    }
    *out = p;
    return PM_OK;
}

void pm_lpf8_plan_destroy(pm_lpf8_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_btab) (void)hipFree(p->d_btab);
    if (p->d_tpl) (void)hipFree(p->d_tpl);
    delete p;
}

// test entry: rows of one int16 buffer, x_stride samples apart (a multiple of 8), -> h_max[rows], *h_redone
extern "C" int pm_bpf8_rows_max_i16(pm_ctx *ctx, const int16_t *d_x, int64_t x_stride, int rows, int64_t n, const double *h_taps, int m, double *h_max,
                                    int64_t *h_redone)
{
    PM_CTX(ctx);
    PM_ARG(d_x != nullptr && h_max != nullptr && rows >= 1 && rows <= (1 << 20) && x_stride % 8 == 0 && ((uintptr_t)d_x & 15) == 0);
    pm_bpf8_plan *p = nullptr;
    if (int rc = pm_bpf8_plan_create(ctx, h_taps, m, &p, kMaxDigitsOfMax)) return rc;
    void *q = nullptr;
    const size_t words = 2 * (size_t)rows + 1;
    int rc = pm_malloc(ctx, words * 8 + sizeof(void *) * (size_t)rows, &q);
    if (!rc) {
        unsigned long long *keys = (unsigned long long *)q, *redone = keys + 2 * rows;
        double *out = (double *)(keys + rows);
        const int16_t **ptrs = (const int16_t **)(keys + words);
        std::vector<const int16_t *> h((size_t)rows);
        for (int r = 0; r < rows; ++r) h[r] = d_x + (int64_t)r * x_stride;
        if (hipMemcpyAsync(ptrs, h.data(), sizeof(void *) * (size_t)rows, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemsetAsync(redone, 0, 8, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = pm_set_error(PM_ERR_HIP, "pm_bpf8_rows_max_i16: setting up failed");
        if (!rc) rc = pm_bpf8_rows_max(ctx, p, ptrs, rows, n, keys, out, redone);
        unsigned long long red = 0;
        if (!rc && (hipMemcpyAsync(h_max, out, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipMemcpyAsync(&red, redone, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = pm_set_error(PM_ERR_HIP, "pm_bpf8_rows_max_i16: the launch failed");
        if (h_redone) *h_redone = (int64_t)red;
        (void)pm_free(ctx, q);
    }
    pm_bpf8_plan_destroy(p);
    return rc;
}

extern "C" int pm_fir_valid_i16_limbs(pm_ctx *ctx, const int16_t *d_x, int64_t n, const double *h_taps, int m, double *d_y, double *h_bound)
{
    pm_bpf8_plan *p = nullptr;
    if (int rc = pm_bpf8_plan_create(ctx, h_taps, m, &p, 4)) return rc;
    if (h_bound) *h_bound = p->err;
    int rc = pm_bpf8_run(ctx, p, d_x, n, d_y);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = pm_set_error(PM_ERR_HIP, "int8 band-pass: the launch failed");
    pm_bpf8_plan_destroy(p);
    return rc;
}
