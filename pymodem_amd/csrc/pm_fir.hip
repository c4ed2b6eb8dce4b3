// FIR stages of the demod_chain path as LDS-staged sliding-window kernels for gfx950.
//
//   pm_fir_valid_{i16,f64}  numpy.convolve(x, h, 'valid')          (19 call sites in the reference)
//   pm_afsk_correlate       4 correlators + magnitudes + difference (afsk.py:153-162)
//   pm_signs_f64            (x >= 0) bitmap, the slicers' only input (slicer.py:85,99-102)
//
// Arithmetic: binary64, one fma per tap, taps visited in ascending input index.  That order is the
// build's canonical order; oracle/pm_oracle.c:pmo_fir_* uses the same one, so GPU and oracle agree
// bit for bit.  Built with -ffp-contract=off: only the fma() written below fuses.
//
// Tiling: a 256-thread workgroup produces T = 256*R consecutive outputs.  The T+m-1 inputs it needs
// are staged once in LDS as f64 (int16 converted on the way in); each thread then owns R consecutive
// outputs and slides a register window over the taps, so one LDS read feeds R fmas.  The LDS image
// carries one spare double after every 8 so that lane t's window starts 9 doubles after lane t-1's
// (18 dwords: conflict-free for ds_read_b64).  Results go back through the same LDS image so that
// the global stores are coalesced (lane-contiguous 8 B).
#include "pm_common.h"
#include "pm_bpf8_dev.h"
#include <chrono>
#include <cmath>
#include <vector>
#include <type_traits>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int kThreads = 256;
constexpr int kGatedGrid = 1024;       // workgroups (per stream) of a gated fallback launch: it walks the tiles if it ever has to run
constexpr int kMaxTaps = 8192;

// one spare double after every R: lane t's window (R consecutive outputs) starts R+1 doubles after lane t-1's,
// an odd number of 8-byte bank pairs, so the 32 lanes of a ds_read_b64 group hit 32 different pairs
template <int R>
__host__ __device__ __forceinline__ int slot(int p) { return p + p / R; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), which would drain the
// next tile's prefetch and this tile's global stores at every barrier; the barriers in these kernels protect nothing
// but the LDS image.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef double double2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));

// Stage inputs [tile0, tile0 + span) of x into the padded LDS image, 16 bytes per global load.  Input pairs (f64) / octets
// (int16) start at even / multiple-of-8 positions, so they never straddle a pad slot: their LDS slots are consecutive.
template <int R>
__device__ __forceinline__ void stage_vec(const double *__restrict__ x, int64_t n, int64_t tile0, int span, int t, double *xs)
{
    for (int p = 2 * t; p < span; p += 2 * kThreads) {
        const int64_t gi = tile0 + p;
        double2v v = {0.0, 0.0};
        if (gi + 1 < n) v = *reinterpret_cast<const double2v *>(x + gi);
        else if (gi < n) v.x = x[gi];
        const int s0 = slot<R>(p);
        xs[s0] = v.x;
        xs[s0 + 1] = v.y;
    }
}

template <int R>
__device__ __forceinline__ void stage_vec(const int16_t *__restrict__ x, int64_t n, int64_t tile0, int span, int t, double *xs)
{
    for (int p = 8 * t; p < span; p += 8 * kThreads) {
        const int64_t gi = tile0 + p;
        short8v v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (gi + 7 < n) v = *reinterpret_cast<const short8v *>(x + gi);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (gi + k < n) v[k] = x[gi + k];
        }
        const int s0 = slot<R>(p);                 // p % 8 == 0 (R == 8): the eight slots are consecutive
#pragma unroll
        for (int k = 0; k < 8; ++k) xs[s0 + k] = (double)v[k];
    }
}

template <int R>
__device__ __forceinline__ void fir_acc_image(const double *__restrict__ xs, const double *__restrict__ h, int m, double (&acc)[R]);

// The sums of one tile: acc[r] = output tile*T + t*R + r of the valid-mode FIR, every sum in ascending input order, one fma per tap.
template <typename InT, int R, bool VEC>
__device__ __forceinline__ void fir_tile_acc(const InT *__restrict__ x, int64_t n, const double *__restrict__ h, int m, int64_t tile,
                                             double (&acc)[R])
{
    extern __shared__ double xs[];
    constexpr int T = kThreads * R;
    const int t = threadIdx.x;
    const int span = T + m - 1;
    const int64_t tile0 = tile * T;
    if (VEC) {
        stage_vec<R>(x, n, tile0, span, t, xs);
    } else {
        for (int idx = t; idx < span; idx += kThreads) {
            int64_t gi = tile0 + idx;
            xs[slot<R>(idx)] = gi < n ? (double)x[gi] : 0.0;
        }
    }
    lds_barrier();
    fir_acc_image<R>(xs, h, m, acc);
}

// The sums over an LDS image of the tile's inputs in the padded layout slot<R>() (thread t's window starts at slot t(R+1)).
template <int R>
__device__ __forceinline__ void fir_acc_image(const double *__restrict__ xs, const double *__restrict__ h, int m, double (&acc)[R])
{
    const int t = threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    // Window registers: two sets of R that alternate between "carry" (last R-1 values of the previous block) and "new".
    // With base = t*R the padded LDS index of input p = base + k is t*(R+1) + k + k/R: the lane part is constant and the
    // tap part is the same for every lane, so each read is lane_base + compile-time offset and only lp advances per block.
    static_assert(R == 8, "the block schedule below is written for 8 outputs per thread and 8 taps per block");
    double s0[R], s1[R];
    const double *lp = xs + t * (R + 1);
#pragma unroll
    for (int j = 0; j < R - 1; ++j) s1[j + 1] = lp[j];
    const double *hp = h + (m - 8);          // hp[7 - b] = h[m - 1 - (i0 + b)]: eight contiguous taps per block
    int i0 = 0;
#define PM_FIR_BLOCK(CARRY, NEW)                                                                   \
    {                                                                                              \
        NEW[0] = lp[7];                                                                            \
        _Pragma("unroll") for (int b = 1; b < 8; ++b) NEW[b] = lp[8 + b];                          \
        _Pragma("unroll") for (int b = 0; b < 8; ++b) {                                            \
            const double g = hp[7 - b];                                                            \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                                          \
                acc[r] = __builtin_fma(g, (r + b < 7) ? CARRY[r + b + 1] : NEW[r + b - 7], acc[r]); \
        }                                                                                          \
        lp += R + 1;                                                                               \
        hp -= 8;                                                                                   \
    }
    for (; i0 + 16 <= m; i0 += 16) {
        PM_FIR_BLOCK(s1, s0)
        PM_FIR_BLOCK(s0, s1)
    }
    if (i0 + 8 <= m) {
        PM_FIR_BLOCK(s1, s0)
        i0 += 8;
    }
#undef PM_FIR_BLOCK
    // m % 8 leftover taps: same register-window block with a compile-time tap count (carry is in s0 after an odd number
    // of blocks, else in s1; both cases are handled by copying the carry into s1 first -- seven moves, once per tile)
    if (i0 < m) {
        if ((i0 >> 3) & 1) {
#pragma unroll
            for (int j = 1; j < R; ++j) s1[j] = s0[j];
        }
        const int left = m - i0;
#define PM_FIR_TAIL(K)                                                                             \
        case K: {                                                                                  \
            s0[0] = lp[7];                                                                         \
            _Pragma("unroll") for (int b = 1; b < K; ++b) s0[b] = lp[8 + b];                       \
            _Pragma("unroll") for (int b = 0; b < K; ++b) {                                        \
                const double g = h[left - 1 - b];                                                  \
                _Pragma("unroll") for (int r = 0; r < R; ++r)                                      \
                    acc[r] = __builtin_fma(g, (r + b < 7) ? s1[r + b + 1] : s0[r + b - 7], acc[r]); \
            }                                                                                      \
        } break;
        switch (left) {
            PM_FIR_TAIL(1) PM_FIR_TAIL(2) PM_FIR_TAIL(3) PM_FIR_TAIL(4) PM_FIR_TAIL(5) PM_FIR_TAIL(6) PM_FIR_TAIL(7)
        default: break;
        }
#undef PM_FIR_TAIL
    }
}

// SIGNS: instead of the float64 outputs, write only their (y >= 0) bitmap -- all a slicer reads of them.
template <typename InT, int R, bool NEG, bool VEC, bool SIGNS>
__device__ __forceinline__ void fir_tile(const InT *__restrict__ x, int64_t n, const double *__restrict__ h, int m,
                                         double *__restrict__ y, int64_t nout, uint64_t *__restrict__ bits, int64_t tile)
{
    extern __shared__ double xs[];
    constexpr int T = kThreads * R;
    const int t = threadIdx.x;
    const int64_t tile0 = tile * T;
    double acc[R];
    fir_tile_acc<InT, R, VEC>(x, n, h, m, tile, acc);
    if (SIGNS) {
        // This thread's 8 consecutive outputs are exactly one byte of the little-endian bitmap (tile0 is a multiple of 2048): no
        // trip through LDS, the 64 lanes of a wave store 64 consecutive bytes.  Bits past nout are written as 0 up to the end of
        // the last 64-bit word.
        static_assert(R == 8, "one bitmap byte per thread");
        const int64_t go = tile0 + (int64_t)t * R;
        unsigned byte = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double v = NEG ? -acc[r] : acc[r];
            byte |= (unsigned)(v >= 0.0 && go + r < nout) << r;
        }
        const int64_t bi = go >> 3;
        if (bi < ((nout + 63) >> 6) * 8) reinterpret_cast<uint8_t *>(bits)[bi] = (uint8_t)byte;
        return;
    }
    lds_barrier();
    {
        double *op = xs + t * (R + 1);     // this thread's R outputs occupy R consecutive slots
#pragma unroll
        for (int r = 0; r < R; ++r) op[r] = NEG ? -acc[r] : acc[r];
    }
    lds_barrier();
    if (VEC) {
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            const int idx = 2 * (r * kThreads + t);            // even: idx and idx+1 sit in adjacent slots
            const int64_t go = tile0 + idx;
            const int s0i = slot<R>(idx);
            const double2v v = {xs[s0i], xs[s0i + 1]};
            if (go + 1 < nout) *reinterpret_cast<double2v *>(y + go) = v;
            else if (go < nout) y[go] = v.x;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int idx = r * kThreads + t;
            const int64_t go = tile0 + idx;
            if (go < nout) y[go] = xs[slot<R>(idx)];
        }
    }
}

template <typename InT, int R, bool NEG, bool VEC, bool SIGNS>
__global__ __launch_bounds__(kThreads) void fir_valid_kernel(const InT *__restrict__ x, int64_t n,
                                                             const double *__restrict__ h, int m,
                                                             double *__restrict__ y, int64_t nout, uint64_t *__restrict__ bits)
{
    fir_tile<InT, R, NEG, VEC, SIGNS>(x, n, h, m, y, nout, bits, (int64_t)blockIdx.x);
}

// Short FIR (M <= 8 taps) on int16 audio with only the sign bitmap kept (fsk.py:149-159 feeding slicer.slice: the fsk_9600 chain):
// the whole window of a lane's 8 outputs is 8 + M - 1 <= 15 samples, i.e. TWO 16-byte loads (its own eight samples and the eight
// after them, which are the next lane's own: the second load is served by the cache), converted once in registers.  No LDS image, no
// barrier: per 8 outputs 2 loads, 15 conversions, 8 M fma and one byte stored -- the kernel is bound by HBM (2 B in and 1/8 B out
// per sample) as long as the conversions keep up.  Sums as everywhere: one fma per tap, ascending input index, from +0.
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

// int16 -> binary64 without v_cvt_f64_i32 (a quarter-rate instruction: fifteen of them cost as much as the 64 fma of the sums):
// u = v + 32768 is the sample with its sign bit flipped, {0x43300000, u} is the double 2^52 + u, and (2^52 + u) - (2^52 + 32768)
// is exact -- the sample itself, +0 for 0.
__device__ __forceinline__ double i16_biased_to_f64(uint32_t u) { return __hiloint2double(0x43300000, (int)u) - 4503599627403264.0; }

constexpr int kShortIter = 8;      // bitmap bytes per lane: a wave that lives for one byte costs more to launch than to run

template <int M, bool NEG>
__global__ __launch_bounds__(kThreads) void fir_short_signs_i16_kernel(const int16_t *__restrict__ x, int64_t n, const double *__restrict__ h,
                                                                       int64_t nout, uint64_t *__restrict__ bits)
{
    const int64_t nbytes = ((nout + 63) >> 6) * 8;                             // the bitmap is written in whole 64-bit words
    double g[M];
#pragma unroll
    for (int j = 0; j < M; ++j) g[j] = h[M - 1 - j];                           // uniform: scalar loads
    // the workgroup's lanes take consecutive bytes (coalesced loads and stores), kShortIter rounds of them
    const int64_t byte0 = (int64_t)blockIdx.x * (kThreads * kShortIter) + threadIdx.x;
    uint4v a, b;
    auto fetch = [&](int64_t by, uint4v &qa, uint4v &qb) {
        const int64_t go = by * 8;                                             // this byte's first output = its first input
        qa = uint4v{0, 0, 0, 0};
        qb = uint4v{0, 0, 0, 0};
        if (by >= nbytes) return;
        if (go + 15 < n) {
            qa = *reinterpret_cast<const uint4v *>(x + go);
            qb = *reinterpret_cast<const uint4v *>(x + go + 8);
        } else {                                                               // the last bytes of the stream: sample by sample
            for (int k = 0; k < 16; ++k) {
                const uint32_t v = go + k < n ? (uint32_t)(uint16_t)x[go + k] : 0u;
                if (k < 8) qa[k >> 1] |= v << ((k & 1) * 16);
                else qb[(k - 8) >> 1] |= v << ((k & 1) * 16);
            }
        }
    };
    fetch(byte0, a, b);
#pragma unroll 1
    for (int it = 0; it < kShortIter; ++it) {
        const int64_t by = byte0 + (int64_t)it * kThreads;
        uint4v na, nb;
        fetch(by + kThreads, na, nb);                                          // the next round's loads fly during this round's sums
        if (by < nbytes) {
            double w[16];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const uint32_t q = (d < 4 ? a[d] : b[d - 4]) ^ 0x80008000u;
                w[2 * d] = i16_biased_to_f64(q & 0xFFFFu);
                w[2 * d + 1] = i16_biased_to_f64(q >> 16);
            }
            uint32_t sign = 0;                                                 // bit r = sign bit of output r
            uint32_t zero = 0;                                                 // bit r = output r is zero (NEG only)
            double acc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = 0.0;
            // tap by tap across the eight outputs: eight independent chains in flight, each in ascending input order
#pragma unroll
            for (int j = 0; j < M; ++j)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_fma(g[j], w[r + j], acc[r]);
#pragma unroll
            for (int r = 7; r >= 0; --r) {
                // a sum is never -0 (it starts from +0), so acc >= 0 is "sign bit clear"; negated (fsk.py:153-154) it is "set, or zero"
                sign = __builtin_amdgcn_alignbit(sign, (uint32_t)__double2hiint(acc[r]), 31);     // (sign << 1) | sign bit
                if (NEG) zero = (zero << 1) | (acc[r] == 0.0 ? 1u : 0u);
            }
            const int64_t left = nout - by * 8;
            const uint32_t valid = left >= 8 ? 0xFFu : left <= 0 ? 0u : (1u << left) - 1u;
            const uint32_t byte = (NEG ? (sign | zero) : ~sign) & valid;
            reinterpret_cast<uint8_t *>(bits)[by] = (uint8_t)byte;
        }
        a = na;
        b = nb;
    }
}

// Several sign-only FIRs with the same taps in ONE launch (blockIdx.y = stream): the output low-passes of a chain group.  One ramp
// and one tail for the whole stage instead of one per chain.
constexpr int kFirBatchMax = 16;
struct FirBatch {
    const double *x[kFirBatchMax];
    uint64_t *bits[kFirBatchMax];
    int64_t n[kFirBatchMax];
};

template <int R, bool NEG, bool VEC>
__global__ __launch_bounds__(kThreads) void fir_signs_batch_kernel(FirBatch B, const double *__restrict__ h, int m,
                                                                   const int *__restrict__ gate = nullptr, int gate_above = 0,
                                                                   int *__restrict__ reset = nullptr)
{
    // the last launch of a certified sweep clears the counter the NEXT sweep on this context will use (pm_ctx::d_sweep is a ring)
    if (reset && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *reset = 0;
    if (gate && *gate <= gate_above) return;                          // a launch that only matters if an earlier kernel said so
    const int s = blockIdx.y;
    const int64_t n = B.n[s], nout = n - m + 1;
    // the grid is sized for the longest stream; a gated launch comes with a small grid and walks the tiles (in the normal case
    // it only has to leave, and leaving costs by the workgroup)
    for (int64_t tile = blockIdx.x; tile * (kThreads * R) < nout; tile += gridDim.x) {
        fir_tile<double, R, NEG, VEC, true>(B.x[s], n, h, m, nullptr, nout, B.bits[s], tile);
        lds_barrier();                                               // the next tile restages the LDS image
    }
}

// The same FIR over `rows` streams of equal length in ONE launch (blockIdx.y = row): the band-pass, Hilbert and matched filters of
// a batch of recordings x chains (pm_lbatch, pm_loopbatch.hip).  A row's input is x + row * x_stride, or x_ptrs[row] + x_off when
// the rows are separate allocations (the recordings of a batch); outputs are rows of one 2-D array.
struct FirRows {
    const void *x;
    int64_t x_stride;
    const void *const *x_ptrs;
    int64_t x_off;
    double *y;
    int64_t y_stride;
    uint64_t *bits;
    int64_t bits_stride;          // 64-bit words
};

template <typename InT, int R, bool NEG, bool VEC, bool SIGNS>
__global__ __launch_bounds__(kThreads) void fir_rows_kernel(FirRows A, int64_t n, const double *__restrict__ h, int m, int64_t nout)
{
    const int64_t r = blockIdx.y;
    const InT *x = A.x_ptrs ? reinterpret_cast<const InT *>(A.x_ptrs[r]) + A.x_off : reinterpret_cast<const InT *>(A.x) + r * A.x_stride;
    fir_tile<InT, R, NEG, VEC, SIGNS>(x, n, h, m, SIGNS ? nullptr : A.y + r * A.y_stride, nout, SIGNS ? A.bits + r * A.bits_stride : nullptr,
                                      (int64_t)blockIdx.x);
}

// Four correlators over one staged window; R outputs x 4 filters = 4R accumulators per thread.
// SPLIT: write the two magnitudes as separate streams (y = mark, y2 = space) instead of their difference (pm_afsk_sweep_signs).
template <int R, bool VEC, bool SPLIT = false>
__global__ __launch_bounds__(kThreads) void afsk_correlate_kernel(const double *__restrict__ x, int64_t n,
                                                                  const double *__restrict__ mi, const double *__restrict__ mq,
                                                                  const double *__restrict__ si, const double *__restrict__ sq,
                                                                  int m, double *__restrict__ y, int64_t nout, double *__restrict__ y2 = nullptr)
{
    extern __shared__ double xs[];
    constexpr int T = kThreads * R;
    const int t = threadIdx.x;
    const int span = T + m - 1;
    const int64_t tile0 = (int64_t)blockIdx.x * T;
    if (VEC) {
        stage_vec<R>(x, n, tile0, span, t, xs);
    } else {
        for (int idx = t; idx < span; idx += kThreads) {
            int64_t gi = tile0 + idx;
            xs[slot<R>(idx)] = gi < n ? x[gi] : 0.0;
        }
    }
    lds_barrier();

    double a[R], b[R], c[R], d[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = b[r] = c[r] = d[r] = 0.0;
    static_assert(R == 4, "the block schedule below is written for 4 outputs per thread and 4 taps per block");
    double s0[R], s1[R];                     // alternating carry / new window sets, as in fir_valid_kernel
    const double *lp = xs + t * (R + 1);
#pragma unroll
    for (int j = 0; j < R - 1; ++j) s1[j + 1] = lp[j];
    const double *pa = mi + (m - 4), *pb = mq + (m - 4), *pc = si + (m - 4), *pd = sq + (m - 4);
    int i0 = 0;
#define PM_CORR_BLOCK(CARRY, NEW)                                                                  \
    {                                                                                              \
        NEW[0] = lp[3];                                                                            \
        _Pragma("unroll") for (int q = 1; q < 4; ++q) NEW[q] = lp[4 + q];                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                            \
            const double ga = pa[3 - q], gb = pb[3 - q], gc = pc[3 - q], gd = pd[3 - q];           \
            _Pragma("unroll") for (int r = 0; r < R; ++r) {                                        \
                const double v = (r + q < 3) ? CARRY[r + q + 1] : NEW[r + q - 3];                  \
                a[r] = __builtin_fma(ga, v, a[r]);                                                 \
                b[r] = __builtin_fma(gb, v, b[r]);                                                 \
                c[r] = __builtin_fma(gc, v, c[r]);                                                 \
                d[r] = __builtin_fma(gd, v, d[r]);                                                 \
            }                                                                                      \
        }                                                                                          \
        lp += R + 1;                                                                               \
        pa -= 4; pb -= 4; pc -= 4; pd -= 4;                                                        \
    }
    for (; i0 + 8 <= m; i0 += 8) {
        PM_CORR_BLOCK(s1, s0)
        PM_CORR_BLOCK(s0, s1)
    }
    if (i0 + 4 <= m) {
        PM_CORR_BLOCK(s1, s0)
        i0 += 4;
    }
#undef PM_CORR_BLOCK
    if (i0 < m) {                            // m % 4 leftover taps, compile-time count, carry moved into s1
        if ((i0 >> 2) & 1) {
#pragma unroll
            for (int j = 1; j < R; ++j) s1[j] = s0[j];
        }
        const int left = m - i0;
#define PM_CORR_TAIL(K)                                                                            \
        case K: {                                                                                  \
            s0[0] = lp[3];                                                                         \
            _Pragma("unroll") for (int q = 1; q < K; ++q) s0[q] = lp[4 + q];                       \
            _Pragma("unroll") for (int q = 0; q < K; ++q) {                                        \
                const int k = left - 1 - q;                                                        \
                const double ga = mi[k], gb = mq[k], gc = si[k], gd = sq[k];                       \
                _Pragma("unroll") for (int r = 0; r < R; ++r) {                                    \
                    const double v = (r + q < 3) ? s1[r + q + 1] : s0[r + q - 3];                  \
                    a[r] = __builtin_fma(ga, v, a[r]);                                             \
                    b[r] = __builtin_fma(gb, v, b[r]);                                             \
                    c[r] = __builtin_fma(gc, v, c[r]);                                             \
                    d[r] = __builtin_fma(gd, v, d[r]);                                             \
                }                                                                                  \
            }                                                                                      \
        } break;
        switch (left) {
            PM_CORR_TAIL(1) PM_CORR_TAIL(2) PM_CORR_TAIL(3)
        default: break;
        }
#undef PM_CORR_TAIL
    }
    double markv[R], spacev[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // afsk.py:153-162: sqrt(i**2 + q**2) with separately rounded squares and sum, then mark - space
        markv[r] = __builtin_sqrt(a[r] * a[r] + b[r] * b[r]);
        spacev[r] = __builtin_sqrt(c[r] * c[r] + d[r] * d[r]);
    }
    // results go back through the LDS image so that the global stores are lane-contiguous
    auto emit = [&](const double (&v)[R], double *__restrict__ dst) {
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R; ++r) xs[t * (R + 1) + r] = v[r];
        lds_barrier();
        if (VEC) {
#pragma unroll
            for (int r = 0; r < R / 2; ++r) {
                const int idx = 2 * (r * kThreads + t);
                const int64_t go = tile0 + idx;
                const int s0i = slot<R>(idx);
                const double2v w = {xs[s0i], xs[s0i + 1]};
                if (go + 1 < nout) *reinterpret_cast<double2v *>(dst + go) = w;
                else if (go < nout) dst[go] = w.x;
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int idx = r * kThreads + t;
                const int64_t go = tile0 + idx;
                if (go < nout) dst[go] = xs[slot<R>(idx)];
            }
        }
    };
    if (SPLIT) {
        emit(markv, y);
        emit(spacev, y2);
    } else {
        double diff[R];
#pragma unroll
        for (int r = 0; r < R; ++r) diff[r] = markv[r] - spacev[r];
        emit(diff, y);
    }
}

// Mark and unit-space magnitudes by a sliding sum (pm_afsk_sweep_signs_tones): the correlator taps of afsk.py:134-144 are the
// powers of one rotation, h[j] = r^j with r = e^{iw} (real part = the cos template, imaginary part = the sin template), so the
// complex correlator sum Z(k) = sum_j r^j x[k + m - 1 - j] obeys
//     Z(k + 1) = x[k + m] + r Z(k) - r^m x[k]
// -- 6 fused operations per tone and sample instead of 2m.  The result is NOT the reference's sum, only within a bound of it
// (L steps of rounding, r^m rounded once, the taps' own deviation from exact powers, measured by the host), which is
// all a certified-sign path needs.  A thread starts its run of L consecutive outputs from the direct sum with the real taps and
// slides from there; runs are short so that the error does not build up and so that the 4m fmas of a start are spread over L
// outputs.  x is staged through LDS with one pad slot per L (lane t's run starts at slot t(L+1)).
struct SlideTones {
    double mr, ms, mer, mes;       // mark:  r = mr + i ms,  r^m = mer + i mes
    double sr, ss, ser, ses;       // space (unit gain)
};
constexpr int kSlideThreads = 128;
template <int L>
__host__ __device__ __forceinline__ int slide_slot(int p) { return p + p / L; }
template <int L>
size_t slide_lds_bytes(int m) { return (size_t)(slide_slot<L>(kSlideThreads * L + m - 1) + 2 + 4 * m) * sizeof(double); }

// sqrt for the sliding sums: reciprocal-square-root seed and ONE coupled Newton step.  With the seed y = (1 + d) / sqrt(v), g = v y and
// h = y / 2 give r = 1/2 - h g = -d - d^2/2 and g (1 + r) = sqrt(v) (1 - 3/2 d^2 + O(d^3)): a seed good to 2^-20 (the ISA manuals give
// V_RSQ_F64 2^29 units in the last place, 2^-23) leaves a relative error below 1.5e-12, which slide_bound() adds to the bound of the
// certified decision (0.1 % of its thousandfold slack) -- the decision needs a value and a bound, not the last bit.  Round 2 took a
// second step (2 units in the last place): three more dependent fma per root, two roots per output, 2 % of the fused kernel.  A v
// below 1e-300 (0 on digital silence, where the seed would be infinite) is raised to that: a root of 1e-150 at most instead of 0.  The IEEE sqrt costs twice the
// instructions again (scaling of subnormal and huge arguments, class checks).
__device__ __forceinline__ double slide_sqrt(double v)
{
    v = __builtin_fmax(v, 1e-300);                            // one instruction where a compare and two selects stood
    const double y = __builtin_amdgcn_rsq(v);
    const double g = v * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
}

// One run: the direct sums of output k0 of the staged tile with the real taps, ascending input index (afsk.py:153-160; every lane
// reads the same four taps per step, an LDS broadcast), then L - 1 sliding steps; the 2 x L magnitudes stay in registers.
template <int L>
__device__ __forceinline__ void slide_run(const double *__restrict__ xs, const double *__restrict__ tp, int run, int m, const SlideTones &T,
                                          double (&mv)[L], double (&sv)[L])
{
    // run starts at input run * L = slot run * (L + 1); input run * L + j sits j + j / L slots further, and j is the same in every lane:
    // the division stays on the scalar unit (as slide_slot(run * L + j) it was a dozen vector instructions per read)
    const double *xr = xs + run * (L + 1);
    double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
    {
        const double2v *tq = reinterpret_cast<const double2v *>(tp);
        int i = 0;
        for (; i + 4 <= m; i += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double v = xr[slide_slot<L>(i + q)];
                const double2v h01 = tq[2 * (i + q)], h23 = tq[2 * (i + q) + 1];
                a = __builtin_fma(h01.x, v, a);
                b = __builtin_fma(h01.y, v, b);
                c = __builtin_fma(h23.x, v, c);
                d = __builtin_fma(h23.y, v, d);
            }
        }
        for (; i < m; ++i) {
            const double v = xr[slide_slot<L>(i)];
            const double2v h01 = tq[2 * i], h23 = tq[2 * i + 1];
            a = __builtin_fma(h01.x, v, a);
            b = __builtin_fma(h01.y, v, b);
            c = __builtin_fma(h23.x, v, c);
            d = __builtin_fma(h23.y, v, d);
        }
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        mv[i] = slide_sqrt(a * a + b * b);                   // afsk.py:157
        sv[i] = slide_sqrt(c * c + d * d);
        if (i + 1 < L) {
            const double xk = xr[i], xn = xr[slide_slot<L>(i + m)];
            const double a2 = __builtin_fma(T.mr, a, __builtin_fma(-T.ms, b, __builtin_fma(-T.mer, xk, xn)));
            const double b2 = __builtin_fma(T.ms, a, __builtin_fma(T.mr, b, -T.mes * xk));
            const double c2 = __builtin_fma(T.sr, c, __builtin_fma(-T.ss, d, __builtin_fma(-T.ser, xk, xn)));
            const double d2 = __builtin_fma(T.ss, c, __builtin_fma(T.sr, d, -T.ses * xk));
            a = a2; b = b2; c = c2; d = d2;
        }
    }
}

// The same run with the two roots in binary32: v_sqrt_f32 of the binary64 sum of squares rounded to binary32 -- for afsk_slide_lpf8_kernel,
// whose magnitudes end as integers of 22 bits anyway.  What that costs in accuracy is a relative 2^-23 (the instruction: within one unit
// in the last place for every one of the 2^24 significands of a binade, checked exhaustively on the device by
// tests/test_gpu_kernels.py::test_v_sqrt_f32_is_within_one_ulp, pm_ubench_sqrt_f32) + 2^-25 (the conversion of the radicand), which the
// kernel adds to its bound in units of its integers; what it saves is the binary64 reciprocal-square-root seed and its Newton step (six
// instructions of which one is quarter-rate) -- twice per sample.  Radicands below binary32's normal range give roots below 2^-63 either
// way: the kernel only scales workgroups whose largest magnitude is above 2^-21, so that is below 2^-20 of a unit.
// (tg: nullptr, or the same four templates -- reversed and interleaved, tg[4 i + f] = h_f[m - 1 - i] -- in DEVICE memory, read through the
// constant address space: uniform addresses, so the loads are scalar loads and the taps reach the fma as scalar operands.  From LDS every
// lane of a wave fetched the same 32 bytes per tap beside its own 8 of the window: 40 bytes per lane and tap against an LDS pipe of 128
// bytes per cycle for the whole CU -- 20 cycles per tap and wave for 16 cycles of fma; the start sums were bound by that, not by the
// vector pipe (taking 15 % of the kernel's vector instructions out of them changed nothing: round 5, gpurun_out r5aa).)
typedef const double __attribute__((address_space(4))) *const_f64_ptr;
template <int L>
__device__ __forceinline__ void slide_run_f32(const double *__restrict__ xs, const double *__restrict__ tp, int run, int m, const SlideTones &T,
                                              float (&mv)[L], float (&sv)[L], const double *tg = nullptr)
{
    const double *xr = xs + run * (L + 1);
    double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
    if (tg) {
        const_f64_ptr tq = (const_f64_ptr)tg;                 // NOLINT: only a C-style cast changes the address space
        const double *xb = xr;
        int i = 0;
        for (; i + L <= m; i += L, xb += L + 1, tq += 4 * L) {
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const double v = xb[j];
                a = __builtin_fma(tq[4 * j + 0], v, a);
                b = __builtin_fma(tq[4 * j + 1], v, b);
                c = __builtin_fma(tq[4 * j + 2], v, c);
                d = __builtin_fma(tq[4 * j + 3], v, d);
            }
        }
        for (int j = 0; i + j < m; ++j) {
            const double v = xb[j];
            a = __builtin_fma(tq[4 * j + 0], v, a);
            b = __builtin_fma(tq[4 * j + 1], v, b);
            c = __builtin_fma(tq[4 * j + 2], v, c);
            d = __builtin_fma(tq[4 * j + 3], v, d);
        }
    } else {
        // The run's window starts on a block boundary of the padded layout (slot(L run) = (L + 1) run), so its taps go in blocks of L
        // whose L values are CONSECUTIVE doubles, one pad apart from block to block: every LDS address of a block is the block's base
        // plus an immediate.  (Round 5: as slide_slot(i + q) per tap -- a division by 12 each -- the start sums spent 20 vector
        // instructions on addresses for every 16 fma, 15 % of the fused kernel's vector instructions.)  Same taps, same order, same sums.
        const double2v *tq = reinterpret_cast<const double2v *>(tp);
        const double *xb = xr;
        int i = 0;
        for (; i + L <= m; i += L, xb += L + 1, tq += 2 * L) {
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const double v = xb[j];
                const double2v h01 = tq[2 * j], h23 = tq[2 * j + 1];
                a = __builtin_fma(h01.x, v, a);
                b = __builtin_fma(h01.y, v, b);
                c = __builtin_fma(h23.x, v, c);
                d = __builtin_fma(h23.y, v, d);
            }
        }
        for (int j = 0; i + j < m; ++j) {                    // fewer than L taps left: inside one block
            const double v = xb[j];
            const double2v h01 = tq[2 * j], h23 = tq[2 * j + 1];
            a = __builtin_fma(h01.x, v, a);
            b = __builtin_fma(h01.y, v, b);
            c = __builtin_fma(h23.x, v, c);
            d = __builtin_fma(h23.y, v, d);
        }
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        mv[i] = __builtin_amdgcn_sqrtf((float)__builtin_fma(a, a, b * b));       // afsk.py:157
        sv[i] = __builtin_amdgcn_sqrtf((float)__builtin_fma(c, c, d * d));
        if (i + 1 < L) {
            const double xk = xr[i], xn = xr[slide_slot<L>(i + m)];
            const double a2 = __builtin_fma(T.mr, a, __builtin_fma(-T.ms, b, __builtin_fma(-T.mer, xk, xn)));
            const double b2 = __builtin_fma(T.ms, a, __builtin_fma(T.mr, b, -T.mes * xk));
            const double c2 = __builtin_fma(T.sr, c, __builtin_fma(-T.ss, d, __builtin_fma(-T.ser, xk, xn)));
            const double d2 = __builtin_fma(T.ss, c, __builtin_fma(T.sr, d, -T.ses * xk));
            a = a2; b = b2; c = c2; d = d2;
        }
    }
}

template <int L>
__global__ __launch_bounds__(kSlideThreads) void afsk_slide_kernel(const double *__restrict__ x, int64_t n, const double *__restrict__ mi,
                                                                   const double *__restrict__ mq, const double *__restrict__ ui,
                                                                   const double *__restrict__ uq, int m, SlideTones T,
                                                                   double *__restrict__ M, double *__restrict__ S, int64_t nout, double gain)
{
    extern __shared__ double xs[];
    constexpr int TILE = kSlideThreads * L;
    const int t = threadIdx.x;
    const int span = TILE + m - 1;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    double *tp = xs + (slide_slot<L>(span) + 2) / 2 * 2;   // the four templates, reversed and interleaved: tp[4i + f] = h_f[m - 1 - i]
    if (((uintptr_t)x & 15) == 0 && tile0 + TILE <= n) {
        // the tile's own TILE inputs: L / 2 independent 16-byte loads per lane, all in flight before the first LDS write
        double2v v[L / 2];
#pragma unroll
        for (int q = 0; q < L / 2; ++q) v[q] = *reinterpret_cast<const double2v *>(x + tile0 + 2 * (q * kSlideThreads + t));
#pragma unroll
        for (int q = 0; q < L / 2; ++q) {
            const int s0 = slide_slot<L>(2 * (q * kSlideThreads + t));     // even position: its pair never straddles a pad slot
            xs[s0] = v[q].x;
            xs[s0 + 1] = v[q].y;
        }
        for (int p = TILE + t; p < span; p += kSlideThreads) {
            const int64_t gi = tile0 + p;
            xs[slide_slot<L>(p)] = gi < n ? x[gi] : 0.0;
        }
    } else {
        for (int p = t; p < span; p += kSlideThreads) {
            const int64_t gi = tile0 + p;
            xs[slide_slot<L>(p)] = gi < n ? x[gi] : 0.0;
        }
    }
    for (int i = t; i < m; i += kSlideThreads) {
        tp[4 * i + 0] = mi[m - 1 - i];
        tp[4 * i + 1] = mq[m - 1 - i];
        tp[4 * i + 2] = ui[m - 1 - i];
        tp[4 * i + 3] = uq[m - 1 - i];
    }
    lds_barrier();
    // The run's 2 x L results stay in registers; once every lane is done with the staged inputs, the LDS image is reused to turn
    // "L consecutive outputs per lane" into coalesced stores, one stream after the other.
    double mv[L], sv[L];
    slide_run<L>(xs, tp, t, m, T, mv, sv);
    if (!S) {                                                // one chain: its mark - space difference (afsk.py:162) in ONE stream
#pragma unroll
        for (int i = 0; i < L; ++i) mv[i] = __builtin_fma(-gain, sv[i], mv[i]);
    }
    const bool full = tile0 + TILE <= nout && ((((uintptr_t)M) | ((uintptr_t)S)) & 15) == 0;      // uniform over the workgroup
    auto emit = [&](const double (&v)[L], double *__restrict__ dst) {
        lds_barrier();
        double *op = xs + t * (L + 1);                       // = slide_slot(k0): this lane's L results in L consecutive slots
#pragma unroll
        for (int i = 0; i < L; ++i) op[i] = v[i];
        lds_barrier();
        if (full) {
#pragma unroll
            for (int q = 0; q < L / 2; ++q) {
                const int p = 2 * (q * kSlideThreads + t);
                const int s0 = slide_slot<L>(p);
                *reinterpret_cast<double2v *>(dst + tile0 + p) = double2v{xs[s0], xs[s0 + 1]};
            }
        } else {
            for (int p = t; p < TILE; p += kSlideThreads)
                if (tile0 + p < nout) dst[tile0 + p] = xs[slide_slot<L>(p)];
        }
    };
    emit(mv, M);
    if (S) emit(sv, S);
}

// G correlator banks that share their mark filters (the chains of afsk_1200_ax25_super_opt.json differ in space gain only):
// F = 2 + 2G filters over one staged window, the mark pair (and its square root) computed once.  w holds the filters interleaved
// and reversed, w[i * F + f] = h_f[m - 1 - i], so that the F coefficients of one tap step are one contiguous scalar load.
// R = 2 outputs per thread: F * R accumulators, one LDS read per F * R fmas, and each lane's results are 16 contiguous bytes,
// so the G output streams are stored straight from registers.
__global__ void pack_group_taps_kernel(const double *__restrict__ mi, const double *__restrict__ mq, const double *__restrict__ sp,
                                       int m, int F, double *__restrict__ w)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * F) return;
    const int i = idx / F, f = idx % F, k = m - 1 - i;
    w[idx] = f == 0 ? mi[k] : f == 1 ? mq[k] : sp[(size_t)(f - 2) * m + k];
}

// tg[4 i + f] = h_f[m - 1 - i], f = mark i, mark q, unit-gain space i, space q: the sliding sums' templates as one table (afsk_fused8_kernel)
__global__ void pack_templates_kernel(const double *__restrict__ mi, const double *__restrict__ mq, const double *__restrict__ ui, const double *__restrict__ uq, int m,
                                      double *__restrict__ tg)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * m) return;
    const int i = idx >> 2, f = idx & 3, k = m - 1 - i;
    tg[idx] = f == 0 ? mi[k] : f == 1 ? mq[k] : f == 2 ? ui[k] : uq[k];
}

template <int G, bool VEC>
__global__ __launch_bounds__(kThreads) void afsk_group_kernel(const double *__restrict__ x, int64_t n, const double *__restrict__ w,
                                                              int m, double *__restrict__ y, int64_t y_stride, int64_t nout,
                                                              const int *__restrict__ gate = nullptr, int gate_above = 0)
{
    extern __shared__ double xs[];
    if (gate && *gate <= gate_above) return;                          // see fir_signs_batch_kernel
    constexpr int R = 2, F = 2 + 2 * G, T = kThreads * R;
    const int t = threadIdx.x;
    const int span = T + m - 1;
    const double *const w_all = w;
    for (int64_t tile = blockIdx.x; tile * T < nout; tile += gridDim.x) {     // one trip, except for a gated launch's small grid
    w = w_all;
    const int64_t tile0 = tile * T;
    if (VEC) {
        stage_vec<R>(x, n, tile0, span, t, xs);
    } else {
        for (int idx = t; idx < span; idx += kThreads) {
            int64_t gi = tile0 + idx;
            xs[slot<R>(idx)] = gi < n ? x[gi] : 0.0;
        }
    }
    lds_barrier();

    double acc[F][R];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[f][r] = 0.0;
    const double *lp = xs + t * (R + 1);       // block b of this lane's window: lp[b * (R + 1) + 0 .. R-1]
    double A[R], B[R];
#pragma unroll
    for (int q = 0; q < R; ++q) A[q] = lp[q];
#define PM_GROUP_BLOCK(CUR, NXT, TAPS)                                                             \
    {                                                                                              \
        _Pragma("unroll") for (int q = 0; q < R; ++q) NXT[q] = lp[(R + 1) + q];                    \
        _Pragma("unroll") for (int q = 0; q < TAPS; ++q) {                                         \
            _Pragma("unroll") for (int f = 0; f < F; ++f) {                                        \
                const double g = w[q * F + f];                                                     \
                _Pragma("unroll") for (int r = 0; r < R; ++r) {                                    \
                    const double v = (q + r < R) ? CUR[q + r] : NXT[q + r - R];                    \
                    acc[f][r] = __builtin_fma(g, v, acc[f][r]);                                    \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        lp += R + 1;                                                                               \
        w += TAPS * F;                                                                             \
    }
    int i0 = 0;
    for (; i0 + 2 * R <= m; i0 += 2 * R) {
        PM_GROUP_BLOCK(A, B, R)
        PM_GROUP_BLOCK(B, A, R)
    }
    if (i0 + R <= m) {
        PM_GROUP_BLOCK(A, B, R)
        i0 += R;
#pragma unroll
        for (int q = 0; q < R; ++q) A[q] = B[q];
    }
    if (i0 < m) PM_GROUP_BLOCK(A, B, 1)       // R = 2: at most one tap left
#undef PM_GROUP_BLOCK
    static_assert(R == 2, "tail and stores are written for two outputs per thread");

    const int64_t go = tile0 + (int64_t)t * R;
    double mark[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mark[r] = __builtin_sqrt(acc[0][r] * acc[0][r] + acc[1][r] * acc[1][r]);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        double o[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double si = acc[2 + 2 * g][r], sq = acc[3 + 2 * g][r];
            o[r] = mark[r] - __builtin_sqrt(si * si + sq * sq);          // afsk.py:153-162
        }
        double *yg = y + (size_t)g * y_stride;
        if (VEC && go + 1 < nout) {
            *reinterpret_cast<double2v *>(yg + go) = double2v{o[0], o[1]};
        } else {
            if (go < nout) yg[go] = o[0];
            if (go + 1 < nout) yg[go + 1] = o[1];
        }
    }
    lds_barrier();                                                   // the next tile restages the LDS image
    }
}

// ---- gain sweep: G AFSK modems that differ in space_gain only, sign bitmaps certified against the exact chain --------------------
// The chains of afsk_1200_ax25_super_opt.json share tones and span and sweep space_gain, which the reference folds into the space
// taps (afsk.py:144-145: taps_g = fl(g * c)).  In exact arithmetic chain g's output is LPF(M) - g * LPF(S), with M the mark
// magnitude and S the space magnitude for the UNIT taps c -- two correlator pairs and two low-passes for the whole sweep instead
// of 2 + 2G pairs and G low-passes.  In binary64 the two routes differ by rounding only, and the slicer reads nothing but the sign:
//     y~_g = fma(-g, B, A),  A = LPF(M), B = LPF(S)   (both by the canonical kernels)
// satisfies |y~_g - y_g| <= E for the bound below, so wherever |y~_g| > E the sign of the exact chain's output y_g is the sign of
// y~_g, and the few samples with |y~_g| <= E are recomputed by the exact chain itself (every fma in its canonical order, one thread
// per sample).  Every bit of every bitmap is therefore the bit the exact kernels write; tests compare them over whole recordings.
//
// Bound.  u = 2^-53, mc / ml = correlator / low-pass taps, X >= max|x|.  Space sums: the taps differ by relative u and each chain of
// mc fmas has relative error <= mc u/(1 - mc u) in sum|t x|, so |sum_g - g sum_1| <= (2 mc + 2) u g mc X; the magnitude sqrt(a^2+b^2)
// is 1-Lipschitz in (a, b) and adds 3 roundings, so |space_g - g S| <= (2 mc + 6) sqrt2 u g mc X; y = M - space adds u |y|.  The
// low-pass is linear up to (ml + 2) u sum|h|(|M| + g|S|) of its own rounding (three canonical sums), and fma(-g, B, A) adds one
// more.  With |M|, |S| <= sqrt2 mc X everything is below  (2 mc + ml + 12) * 2.9 u * sum|h| (1 + g) mc X  ~ 1e-13 * scale; E is
// taken as 1e-10 * sum|h| (1 + g_max) mc sqrt2 X, a thousand times that, which still flags only ~1e-8 of the samples.
constexpr int kSweepMax = 8;
struct SweepArgs {
    double gain[kSweepMax];
    uint64_t *bits[kSweepMax];
};

// The combine step of a thread's R = 8 consecutive outputs: y_g = A - g B (ONE: y = A as it stands), one bitmap byte per modem, and
// the (sample, modem) pairs that cannot be certified (|y| <= E, NaN too) to the list, whose bits sweep_exact_kernel decides afterwards.
// A wave whose outputs all lie inside the stream and are all certified -- all but one in ~1e5 -- spends three vector instructions per
// output and modem: the fma, the sign bit shifted into the byte from the high word (v_alignbit; y != 0 there, so the sign bit is
// `y >= 0` negated) and one compare whose lane mask is folded into a scalar; otherwise the wave goes through its modems once more, lane
// by lane (same bytes for the certified outputs).
template <int R, bool ONE>
__device__ __forceinline__ void sweep_combine(const double (&a)[R], const double (&b)[R], int64_t go, int64_t nout, int G, const SweepArgs &P,
                                              double E, unsigned long long *__restrict__ list, int *__restrict__ count, int cap)
{
    static_assert(R == 8, "one bitmap byte per thread");
    const bool whole = __all(go + R <= nout);
    unsigned long long unsure_any = 0;
    if (whole) {
        for (int g = 0; g < G; ++g) {
            const double mg = -P.gain[g];
            unsigned neg = 0;
#pragma unroll
            for (int r = R - 1; r >= 0; --r) {
                const double y = ONE ? a[r] : __builtin_fma(mg, b[r], a[r]);
                neg = __builtin_amdgcn_alignbit(neg, (unsigned)__double2hiint(y), 31);      // (neg << 1) | sign bit
                unsure_any |= __ballot(!(fabs(y) > E));
            }
            reinterpret_cast<uint8_t *>(P.bits[g])[go >> 3] = (uint8_t)~neg;
        }
        if (unsure_any == 0) return;
    }
    for (int g = 0; g < G; ++g) {                            // the stream's last outputs, or something in this wave is uncertain
        const double mg = -P.gain[g];
        unsigned byte = 0, unsure = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double y = ONE ? a[r] : __builtin_fma(mg, b[r], a[r]);
            const bool in = go + r < nout;
            byte |= (unsigned)(in && y >= 0.0) << r;
            unsure |= (unsigned)(in && !(fabs(y) > E)) << r;                 // cannot be certified (NaN lands here too)
        }
        reinterpret_cast<uint8_t *>(P.bits[g])[go >> 3] = (uint8_t)byte;     // bits past nout: 0 up to the end of the last word
        while (unsure) {
            const int r = __ffs((int)unsure) - 1;
            unsure &= unsure - 1;
            const int idx = atomicAdd(count, 1);
            if (idx < cap) list[idx] = ((unsigned long long)g << 48) | (unsigned long long)(go + r);
        }
    }
}

// The second low-pass of the sweep with the combine step as its epilogue: B = LPF(S) stays in registers, A = LPF(M) is read back
// (the thread's eight consecutive values), and what leaves the kernel is one bitmap byte per modem and thread plus the list of
// samples that could not be certified.
template <int R, bool VEC>
__global__ __launch_bounds__(kThreads) void fir_sweep_kernel(const double *__restrict__ S, int64_t n, const double *__restrict__ h, int m,
                                                             const double *__restrict__ A, int64_t nout, int G, SweepArgs P, double E,
                                                             unsigned long long *__restrict__ list, int *__restrict__ count, int cap)
{
    static_assert(R == 8, "one bitmap byte per thread");
    double b[R];
    fir_tile_acc<double, R, VEC>(S, n, h, m, (int64_t)blockIdx.x, b);
    const int64_t go = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * R;
    if (go >= ((nout + 63) >> 6) * 64) return;
    double a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = A && go + r < nout ? A[go + r] : 0.0;
    if (A) sweep_combine<R, false>(a, b, go, nout, G, P, E, list, count, cap);
    else sweep_combine<R, true>(b, b, go, nout, G, P, E, list, count, cap);      // no A: the input already is mark - gain * space (one chain)
}

// Sliding sums, low-pass(es) and the certified combine in ONE kernel: the magnitude streams never reach memory.  A workgroup owns
// 2048 low-pass outputs; it needs the ml - 1 magnitudes past them too, computes all of them as runs of L = 12 from one staged window of
// x (the first ceil((2047 + ml) / L) lanes do, e.g. 179 of 256 for ml = 100: three waves with short runs beat two with long ones), lays them out as the FIR's padded LDS images -- the
// space image over the window of x, which is dead by then -- and every thread takes its 8 outputs of each low-pass from there.
// ONE: a single chain, its mark - gain * space difference as the only image.  Arithmetic and bound: afsk_slide_kernel + fir_valid_kernel
// + fir_sweep_kernel, value for value.
constexpr int kFuseRun = 12;     // measured (g = 7 / g = 1, 28.8 M samples): runs of 10: 0.416 / 0.289 ms, 12: 0.395 / 0.264, 16: 0.411 / 0.277, 34 (one wave slides): 0.471 / 0.317
inline size_t fuse_region0(int m, int ml, int L = kFuseRun)
{
    const int nmag = kThreads * 8 + ml - 1, nruns = (nmag + L - 1) / L;
    const int p = nruns * L + m - 1;
    const int a = p + p / L + 2, b = slot<8>(nmag) + 2;
    return (size_t)((a > b ? a : b) + 1) / 2 * 2;
}
inline size_t fuse_image(int ml) { return (size_t)(slot<8>(kThreads * 8 + ml - 1) + 3) / 2 * 2; }
inline size_t fuse_lds_bytes(int m, int ml, int L = kFuseRun) { return (fuse_region0(m, ml, L) + fuse_image(ml) + 4 * (size_t)m) * sizeof(double); }

template <bool ONE, int L = kFuseRun>
__global__ __launch_bounds__(kThreads) void afsk_slide_lpf_kernel(const double *__restrict__ x, int64_t n, const double *__restrict__ mi,
                                                                  const double *__restrict__ mq, const double *__restrict__ ui,
                                                                  const double *__restrict__ uq, int m, SlideTones T,
                                                                  const double *__restrict__ h, int ml, int64_t nout, int G, SweepArgs P, double E,
                                                                  unsigned long long *__restrict__ list, int *__restrict__ count, int cap,
                                                                  int region0, int image)
{
    extern __shared__ double xs[];
    constexpr int R = 8, TILE = kThreads * R;
    const int t = threadIdx.x;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const int nmag = TILE + ml - 1, nruns = (nmag + L - 1) / L, xspan = nruns * L + m - 1;
    double *im = xs + region0, *tp = im + image;
    if (((uintptr_t)x & 15) == 0 && tile0 + TILE <= n) {
        double2v v[R / 2];
#pragma unroll
        for (int q = 0; q < R / 2; ++q) v[q] = *reinterpret_cast<const double2v *>(x + tile0 + 2 * (q * kThreads + t));
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const int s0 = slide_slot<L>(2 * (q * kThreads + t));
            xs[s0] = v[q].x;
            xs[s0 + 1] = v[q].y;
        }
        for (int p = TILE + t; p < xspan; p += kThreads) {
            const int64_t gi = tile0 + p;
            xs[slide_slot<L>(p)] = gi < n ? x[gi] : 0.0;
        }
    } else {
        for (int p = t; p < xspan; p += kThreads) {
            const int64_t gi = tile0 + p;
            xs[slide_slot<L>(p)] = gi < n ? x[gi] : 0.0;
        }
    }
    for (int i = t; i < m; i += kThreads) {
        tp[4 * i + 0] = mi[m - 1 - i];
        tp[4 * i + 1] = mq[m - 1 - i];
        tp[4 * i + 2] = ui[m - 1 - i];
        tp[4 * i + 3] = uq[m - 1 - i];
    }
    lds_barrier();
    double mv[L], sv[L];
    if (t < nruns) slide_run<L>(xs, tp, t, m, T, mv, sv);
    lds_barrier();                                           // every lane is done with the window of x
    if (t < nruns) {
        const double g0 = P.gain[0];
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int p = t * L + i;
            if (p < nmag) {
                if (ONE) {
                    im[slot<R>(p)] = __builtin_fma(-g0, sv[i], mv[i]);       // afsk.py:162 on the approximate magnitudes
                } else {
                    im[slot<R>(p)] = mv[i];
                    xs[slot<R>(p)] = sv[i];
                }
            }
        }
    }
    lds_barrier();
    double a[R], b[R];
    fir_acc_image<R>(im, h, ml, a);
    if (!ONE) fir_acc_image<R>(xs, h, ml, b);
    const int64_t go = tile0 + (int64_t)t * R;
    if (go >= ((nout + 63) >> 6) * 64) return;
    sweep_combine<R, ONE>(a, b, go, nout, G, P, E, list, count, cap);
}

// The exact chain for single samples needs to know where the sweep's input came from.  AUDIO (src.audio != nullptr): the band-passed
// stream the sweep saw was itself a value with a bound (pm_bpf8.hip), so the recomputation starts one stage earlier -- the mc + ml - 1
// band-pass outputs under the entry from the int16 audio, the reference's sum in fir_valid_kernel's order.
struct SweepSource {
    const int16_t *audio;            // nullptr: d_x is the reference's band-passed stream
    const double *bpf;
    int mb;
    double e_x;                      // |d_x[k] - reference's band-pass output|
};

// What the fused matrix-pipe kernel needs to decide its own uncertain samples (round 5): the exact chain's operands.
struct SweepTail {
    const double *space;             // the modems' own space taps (gain folded in, afsk.py:144-145): modem g at space + 2 g mc
    const double *lpf;               // the low-pass taps in binary64
    int lds_ok;                      // the exact chain's work space fits the kernel's LDS image: uncertain samples are decided in place
    SweepSource src;
};
constexpr int kTailCap = 48;         // uncertain (sample, modem) pairs a workgroup decides itself (0.06 per workgroup on average); more go to the list

// The exact chain of ONE (sample, modem) pair by a whole workgroup, every sum in the canonical order of fir_valid_kernel /
// afsk_correlate_kernel (= sweep_exact_kernel below, value for value): the mc + ml - 1 band-pass outputs under the entry (one thread
// each, from the audio; or read from x), the ml correlator-bank outputs (one thread each, four sums side by side), the low-pass sum
// (thread 0).  `dd`: (2 ml + 2 mc + 2 mb - 2 + 4 mc + ...) doubles of LDS nobody else uses any more.  Ends with a barrier.
template <int THREADS>
__device__ __forceinline__ void sweep_tail_entry(double *__restrict__ dd, int t, const double *__restrict__ x, const double *__restrict__ mi,
                                                 const double *__restrict__ mq, const double *__restrict__ si, const double *__restrict__ sq, int mc,
                                                 const double *__restrict__ lpf, int ml, const SweepSource &src, int64_t k, unsigned long long *__restrict__ bits)
{
    const int nw = ml + mc - 1, mb = src.audio ? src.mb : 0, na = src.audio ? nw + mb - 1 : 0;
    double *xw = dd + ml, *aw = xw + nw, *tb = aw + na, *tc = tb + mb, *tl = tc + 4 * mc;
    if (src.audio) {
        for (int p = t; p < na; p += THREADS) aw[p] = (double)src.audio[k + p];
        for (int i = t; i < mb; i += THREADS) tb[i] = src.bpf[mb - 1 - i];
    } else {
        for (int p = t; p < nw; p += THREADS) xw[p] = x[k + p];
    }
    for (int i = t; i < mc; i += THREADS) {
        tc[4 * i + 0] = mi[mc - 1 - i];
        tc[4 * i + 1] = mq[mc - 1 - i];
        tc[4 * i + 2] = si[mc - 1 - i];
        tc[4 * i + 3] = sq[mc - 1 - i];
    }
    for (int i = t; i < ml; i += THREADS) tl[i] = lpf[ml - 1 - i];
    __syncthreads();
    if (src.audio) {
        for (int p = t; p < nw; p += THREADS) {
            // (unrolled: the operands of the next fmas are on their way from LDS while the chain waits for its own latency -- a
            // workgroup with an entry holds its slot for as long as this takes, and alone the kernel ends with its last such workgroup)
            double acc = 0.0;
#pragma unroll 8
            for (int i = 0; i < mb; ++i) acc = __builtin_fma(tb[i], aw[p + i], acc);
            xw[p] = acc;
        }
        __syncthreads();
    }
    for (int j = t; j < ml; j += THREADS) {
        double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
#pragma unroll 4
        for (int i = 0; i < mc; ++i) {
            const double v = xw[j + i];
            a = __builtin_fma(tc[4 * i + 0], v, a);
            b = __builtin_fma(tc[4 * i + 1], v, b);
            c = __builtin_fma(tc[4 * i + 2], v, c);
            d = __builtin_fma(tc[4 * i + 3], v, d);
        }
        dd[j] = __builtin_sqrt(a * a + b * b) - __builtin_sqrt(c * c + d * d);
    }
    __syncthreads();
    if (t == 0) {
        double acc = 0.0;
#pragma unroll 8
        for (int j = 0; j < ml; ++j) acc = __builtin_fma(tl[j], dd[j], acc);
        unsigned long long *w = bits + (k >> 6);
        const unsigned long long bit = 1ull << (k & 63);
        if (acc >= 0.0) atomicOr(w, bit); else atomicAnd(w, ~bit);
    }
    __syncthreads();
}
inline size_t sweep_tail_doubles(int mc, int ml, int mb) { return (size_t)2 * ml + (size_t)(ml + mc - 1) * 2 + 2 * (size_t)mb + 4 * (size_t)mc; }

// The same kernel with its low-passes on the int8 matrix pipe (v_mfma_i32_16x16x64_i8; the band-pass went there first: pm_bpf8.hip).
// The low-pass sums are 2/3 of afsk_slide_lpf_kernel's vector instructions and feed nothing but the certified decision, and the int8
// MFMA is the one matrix instruction that was measured to run BESIDE vector f64 work (tools/ubench/mfma_i8.hip).  So: the magnitudes a
// run leaves in registers are rounded to integers |X| <= 2^22 -- scaled by the power of two that fits the WORKGROUP's largest
// magnitude, so a quiet recording keeps its bits -- and written as three planes of signed base-256 digits; the taps come as three
// digits (pm_lpf8_plan: q = rint(h 2^S), |q| <= 2^22); a tile of 256 outputs is  out[16 i + j] = sum_c A[i][c] B[c][j],  A[i][c] = digit
// plane [tile + 16 i + c] (one ds_read_b128 per lane), B = the Toeplitz band of a tap digit -- 2 blocks x the 8 digit pairs of weight
// 256 and up = 16 MFMA per stream and tile, four int32 sums by weight, recombined exactly in binary64 (an integer below 2^50).
// The integer sums are exact, so what separates the value from the reference's low-pass output is  sum|h - q 2^-S| * the largest
// magnitude  +  2^-(S+s2) (sum|q| / 2 + the digit pair left out)  on top of E -- computed by the workgroup from its own scale
// (Ecmp).  Round 3 had 4 x 5 digits, all 20 pairs, and a scale fixed by the caller's bound on the audio: 40 MFMA per stream and tile,
// eight accumulators to clear and recombine, 20 uncertain decisions per recording; now several hundred of 230 M (the exact
// recomputation takes them in its stride) for 40 % of the matrix work and half of the recombination (profiles/r04_sweep_probe.txt;
// with the six pairs of weight 256^2 and up: 2600 uncertain decisions, and the exact kernel behind them cost what the matrix pipe saved).
// Lane (r, g) of a tile holds outputs 64 g + 16 v + r, v = 0..3: sign and bound tests become four ballots per modem, and the tile's
// four bitmap words are put together from their 16-bit pieces.
constexpr int kL8Plane = 2176;           // bytes of a digit plane: 2048 outputs + (ml - 1 <= 112) + what the last tile's band reads beyond
constexpr int kL8Dig = 3;                // digits of a magnitude and of a tap
constexpr int kL8Acc = 4;                // accumulators: the weights 256^1 .. 256^4
struct Lpf8Args {
    int S;                               // taps: q = rint(h 2^S)
    double c_tap;                        // sum |h 2^S - q|: the taps' quantisation, in units of 2^-S
    double c_q;                          // sum |q| / 2 (the magnitudes' rounding) + the digit product that is left out: units of 2^-(S+s2)
    double gfac;                         // 1 + the largest gain (two streams: |a - g b|'s error), 1 for one stream
    double qabs;                         // sum |q|: what a unit of error in every magnitude's integer costs (PM_LPF8_F32MAG: the binary32 roots)
    const int4v *btab;
};

#ifndef PM_LPF8_F32MAG
#define PM_LPF8_F32MAG 1         // the sliding sums' roots in binary32 (slide_run_f32): g = 7 0.203 -> 0.188 ms, g = 1 0.133 -> 0.124, 752 -> 891 / 20 -> 221 uncertain decisions
#endif

#ifndef PM_LPF8_RECOMB32
#define PM_LPF8_RECOMB32 1       // pairs of accumulators recombined as 32-bit integers first (two conversions per output instead of four): 0.206 -> 0.203 ms
#endif
#ifndef PM_LPF8_WAVES
#define PM_LPF8_WAVES 4          // waves per SIMD the fused matrix-pipe kernel is compiled for (-DPM_LPF8_WAVES=5: measured, profiles/r04_lpf8_occupancy.txt)
#endif
// One workgroup's tile of a certified sweep from the band-passed window in LDS (xs, slide_slot layout): sliding sums, digit planes,
// low-pass(es) on the matrix pipe, certified combine -- the body of afsk_slide_lpf8_kernel, and of afsk_fused8_kernel once per sweep.
// planes: kL8Dig digit planes per stream (may lie over xs: nobody reads the window after the sliding sums); tp: 4 m doubles; bl: the
// band operands; wl: the workgroup's list of uncertain (sweep, modem, sample) entries, wl[kTailCap] their count; wmax8: 8 floats.
template <bool ONE>
__device__ __forceinline__ void lpf8_sweep_tile(double *__restrict__ xs, unsigned char *__restrict__ planes, double *__restrict__ tp, int4v *__restrict__ bl,
                                                unsigned *__restrict__ wl, float *__restrict__ wmax8, int sweep, bool lds_ok, int t, int64_t tile0,
                                                const double *__restrict__ mi, const double *__restrict__ mq, const double *__restrict__ ui,
                                                const double *__restrict__ uq, int m, const SlideTones &T, const Lpf8Args &Q, int ml, int64_t nout, int G,
                                                const SweepArgs &P, double E, unsigned long long *__restrict__ list, int *__restrict__ count, int cap,
                                                const double *tg = nullptr)
{
    constexpr int L = kFuseRun, TILE = kThreads * 8;
    const int nmag = TILE + ml - 1, nruns = (nmag + L - 1) / L;
    if (!tg)
        for (int i = t; i < m; i += kThreads) {
            tp[4 * i + 0] = mi[m - 1 - i];
            tp[4 * i + 1] = mq[m - 1 - i];
            tp[4 * i + 2] = ui[m - 1 - i];
            tp[4 * i + 3] = uq[m - 1 - i];
        }
    lds_barrier();
#if PM_LPF8_F32MAG
    float mv[L], sv[L];
    if (t < nruns) slide_run_f32<L>(xs, tp, t, m, T, mv, sv, tg);
    // the workgroup's largest value (what the planes will hold) and, for one stream, the largest mark + gain * space (what its roots'
    // errors scale with: the difference may be far smaller than either)
    float vmaxf = 0.0f, vsumf = 0.0f;
    if (t < nruns) {
        if (ONE) {
            const float g0 = (float)P.gain[0], ag0 = fabsf(g0);
#pragma unroll
            for (int i = 0; i < L; ++i) {
                vsumf = fmaxf(vsumf, fmaf(ag0, sv[i], mv[i]));
                mv[i] = fmaf(-g0, sv[i], mv[i]);             // afsk.py:162 on the approximate magnitudes
            }
        }
#pragma unroll
        for (int i = 0; i < L; ++i) vmaxf = fmaxf(vmaxf, ONE ? fabsf(mv[i]) : fmaxf(mv[i], sv[i]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vmaxf = fmaxf(vmaxf, __shfl_xor(vmaxf, off));
        if (ONE) vsumf = fmaxf(vsumf, __shfl_xor(vsumf, off));
    }
    float (*wmaxf)[kThreads / 64] = reinterpret_cast<float (*)[kThreads / 64]>(wmax8);      // (dynamic block: a static array would move its start)
    if ((t & 63) == 0) {
        wmaxf[0][t >> 6] = vmaxf;
        wmaxf[1][t >> 6] = vsumf;
    }
    lds_barrier();                                           // every lane is done with the window of x: the planes take its place
    static_assert(kThreads == 256, "four waves");
    const double vmax = (double)fmaxf(fmaxf(wmaxf[0][0], wmaxf[0][1]), fmaxf(wmaxf[0][2], wmaxf[0][3]));
    const double vsum = ONE ? (double)fmaxf(fmaxf(wmaxf[1][0], wmaxf[1][1]), fmaxf(wmaxf[1][2], wmaxf[1][3])) : vmax;
    int e2 = 0;
    (void)frexp(vmax, &e2);                                  // vmax < 2^e2
    // (not: NaN, infinities, all zeros, and what binary32 cannot carry: magnitudes below 2^-21 or above 2^60 -- everything goes to the list then)
    const bool scalable = vmax < 1.0e18 && vmax > 4.8e-7 && vsum < 1.0e18;
    const int s2 = scalable ? 22 - e2 : 0;
    const double scale = ldexp(1.0, s2);
    const float scalef = (float)scale;
    // A root is within (2^-23 + 2^-25) of itself of the true one (slide_run_f32), the one-stream difference adds the gain's and its own
    // binary32 roundings (2^-24 each, of mark + gain space at most): in units of the integers, per magnitude; 1e-6: vmax and vsum are
    // themselves rounded values
    const double root_units = (ONE ? 2.13 : 1.13) * 1.1920928955078125e-07 * (vsum * scale) * (1.0 + 1e-6) + 1e-5;
    // |sum h v - 2^-(S+s2) 256 val| <= sum|h - q 2^-S| vmax + 2^-(S+s2) (sum|q| (1/2 + the roots' share) + the product left out), times
    // 1 + gmax for a - g b; plus E, what separates sum h v from the reference's low-pass output: all in the units of `val`, rounded up
    const double Ecmp = scalable ? ceil((ldexp(E, Q.S + s2) + Q.gfac * (Q.c_tap * (vmax * scale) * (1.0 + 1e-6) + Q.c_q + Q.qabs * root_units)) *
                                        (1.0 + 1e-9) * (1.0 / 256.0)) + 2.0
                                 : __builtin_inf();
#else
    double mv[L], sv[L];
    if (t < nruns) slide_run<L>(xs, tp, t, m, T, mv, sv);
    // The largest magnitude of the workgroup (what the planes will hold): the digits are scaled to IT, not to the largest the audio
    // could produce -- a quiet recording keeps its 22 bits (round 4; a fixed scale cost a bit of certainty per halving of the level)
    double vmax = 0.0;
    if (t < nruns) {
        if (ONE) {
            const double g0 = P.gain[0];
#pragma unroll
            for (int i = 0; i < L; ++i) mv[i] = __builtin_fma(-g0, sv[i], mv[i]);       // afsk.py:162 on the approximate magnitudes
        }
#pragma unroll
        for (int i = 0; i < L; ++i) vmax = fmax(vmax, ONE ? fabs(mv[i]) : fmax(fabs(mv[i]), fabs(sv[i])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, off));
    __shared__ double wmax[kThreads / 64];
    if ((t & 63) == 0) wmax[t >> 6] = vmax;
    lds_barrier();                                           // every lane is done with the window of x: the planes take its place
    vmax = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    static_assert(kThreads == 256, "four waves");
    int e2 = 0;
    (void)frexp(vmax, &e2);                                  // vmax < 2^e2 (0 for no signal at all)
    const bool scalable = vmax < 1.0e300 && vmax > 1.0e-280;  // (not: NaN, infinities, all zeros -- everything goes to the list then)
    const int s2 = scalable ? 22 - e2 : 0;
    const double scale = ldexp(1.0, s2);
    // |sum h v - 2^-(S+s2) 256 val| <= sum|h - q 2^-S| vmax + 2^-(S+s2) (sum|q| / 2 + the product left out), times 1 + gmax for
    // a - g b; plus E, what separates sum h v from the reference's low-pass output: all in the units of `val`, rounded up
    const double Ecmp = scalable ? ceil((ldexp(E, Q.S + s2) + Q.gfac * (Q.c_tap * (vmax * scale) + Q.c_q)) * (1.0 + 1e-9) * (1.0 / 256.0)) + 2.0
                                 : __builtin_inf();
#endif
    if (t < nruns) {
        static_assert(L % 4 == 0, "four magnitudes per plane word");
        auto put = [&](const auto (&val)[L], int stream) {
#pragma unroll
            for (int q = 0; q < L / 4; ++q) {
                unsigned w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // v 2^s2 is exact, |..| <= 2^22; adding 1.5 2^52 leaves its nearest integer in the low word (two's complement);
                    // the bytes of (X + 0x808080) ^ 0x808080 are X's three balanced base-256 digits
#if PM_LPF8_F32MAG
                    // (binary32: 1.5 2^23, the integer in the low 23 bits of the significand)
                    const float sc = scalable ? fmaf(val[4 * q + i], scalef, 12582912.0f) : 12582912.0f;
                    w[i] = ((unsigned)((int)__float_as_uint(sc) - 0x4B400000) + 0x808080u) ^ 0x808080u;
#else
                    const double sc = scalable ? __builtin_fma(val[4 * q + i], scale, 6755399441055744.0) : 6755399441055744.0;
                    w[i] = ((unsigned)__double2loint(sc) + 0x808080u) ^ 0x808080u;
#endif
                }
                const unsigned a01 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), a23 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u);
                unsigned char *at = planes + (size_t)stream * kL8Dig * kL8Plane + L * t + 4 * q;
                *reinterpret_cast<unsigned *>(at) = __builtin_amdgcn_perm(a23, a01, 0x05040100u);
                *reinterpret_cast<unsigned *>(at + kL8Plane) = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
                *reinterpret_cast<unsigned *>(at + 2 * kL8Plane) =
                    __builtin_amdgcn_perm(w[1], w[0], 0x0c0c0602u) | __builtin_amdgcn_perm(w[3], w[2], 0x06020c0cu);
            }
        };
        put(mv, 0);
        if (!ONE) put(sv, 1);
    }
    const int lane = t & 63, wave = t >> 6, r = lane & 15, g4 = lane >> 4;
    // the band operands (3 digits x 2 blocks x 64 lanes x 16 bytes) behind the templates
    for (int i = t; i < 2 * kL8Dig * 64; i += kThreads) bl[i] = Q.btab[i];
    lds_barrier();
    const int64_t nout64 = ((nout + 63) >> 6) * 64;
#pragma unroll 1
    for (int q = 0; q < 2; ++q) {
        const int tl = (wave * 2 + q) * 256;
        const int64_t go = tile0 + tl;
        if (go >= nout64) break;
        double a[4], b[4];
#pragma unroll
        for (int stream = 0; stream < (ONE ? 1 : 2); ++stream) {
            // the eight digit products of weight 256 and up (accumulator i + j - 1); the ninth, x_0 q_0, is bounded in c_q
            int4v acc[kL8Acc];
#pragma unroll
            for (int w = 0; w < kL8Acc; ++w) acc[w] = int4v{0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const unsigned char *at = planes + (size_t)stream * kL8Dig * kL8Plane + tl + 16 * r + 64 * kb + 16 * g4;
                int4v d[kL8Dig];
#pragma unroll
                for (int i = 0; i < kL8Dig; ++i) d[i] = *reinterpret_cast<const int4v *>(at + i * kL8Plane);
#pragma unroll
                for (int bb = 0; bb < kL8Dig; ++bb) {
                    const int4v band = bl[(bb * 2 + kb) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < kL8Dig; ++i)
                        if (i + bb >= 1)
                            acc[i + bb - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(d[i], band, acc[i + bb - 1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                // W_1 + 256 W_2 + 256^2 W_3 + 256^3 W_4, an integer below 2^50, exact: the pairs first, as 32-bit integers (ml <= 113
                // taps and at most three digit pairs per weight keep |W_w| below 5.6 M, so W + 256 W' stays inside 2^31) -- two
                // integer-to-double conversions per output instead of four (they are quarter-rate instructions)
#if PM_LPF8_RECOMB32
                const int lo = acc[0][v] + acc[1][v] * 256, hi = acc[2][v] + acc[3][v] * 256;
                const double val = __builtin_fma((double)hi, 65536.0, (double)lo);
#else
                const double val = __builtin_fma(__builtin_fma(__builtin_fma((double)acc[3][v], 256.0, (double)acc[2][v]), 256.0, (double)acc[1][v]), 256.0,
                                                 (double)acc[0][v]);
#endif
                if (stream == 0) a[v] = val; else b[v] = val;
            }
        }
        // lane (r, g4) holds outputs go + 64 g4 + 16 v + r
        const int64_t left = nout - go - (64 * g4 + r);
        const int lim = left > 1024 ? 1024 : (int)left;          // output v is inside the stream iff 16 v < lim
        unsigned long long in[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) in[v] = __ballot(16 * v < lim);
        for (int g = 0; g < G; ++g) {
            const double mg = -P.gain[g];
            double y[4];
            unsigned long long pos[4], uns = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                y[v] = ONE ? a[v] : __builtin_fma(mg, b[v], a[v]);
                pos[v] = __ballot(y[v] >= 0.0) & in[v];
                uns |= ~__ballot(fabs(y[v]) > Ecmp) & in[v];                  // cannot be certified (NaN lands here too)
            }
            // the tile's four bitmap words from the 16-bit pieces of the four ballots: scalar arithmetic (the ballots are uniform), then
            // lane w < 4 picks word w -- with per-lane shifts of 64-bit values this was a dozen vector instructions per modem and tile
            // (putting the words together with scalar arithmetic and a select per lane, as fir8_kernel does for its one bitmap, was measured
            // here and is slower: seven modems' worth of 64-bit scalar shifts per tile, 0.233 against 0.206 ms -- profiles/r04_sweep_probe.txt)
            const int sh = 16 * (lane & 3);
            const unsigned lo = ((unsigned)(pos[0] >> sh) & 0xFFFFu) | ((unsigned)(pos[1] >> sh) << 16);
            const unsigned hi = ((unsigned)(pos[2] >> sh) & 0xFFFFu) | ((unsigned)(pos[3] >> sh) << 16);
            if (lane < 4 && go + 64 * lane < nout64)
                reinterpret_cast<unsigned long long *>(P.bits[g])[(go >> 6) + lane] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            if (uns) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (16 * v < lim && !(fabs(y[v]) > Ecmp)) {
                        const unsigned mine = lds_ok ? atomicAdd(&wl[kTailCap], 1u) : (unsigned)kTailCap;
                        if (mine < (unsigned)kTailCap) {
                            wl[mine] = ((unsigned)sweep << 20) | ((unsigned)g << 16) | (unsigned)(tl + 64 * g4 + 16 * v + r);
                        } else {
                            const int idx = atomicAdd(count, 1);
                            if (idx < cap) list[idx] = ((unsigned long long)g << 48) | (unsigned long long)(go + 64 * g4 + 16 * v + r);
                        }
                    }
            }
        }
    }
}

template <bool ONE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(PM_LPF8_WAVES, PM_LPF8_WAVES))) void afsk_slide_lpf8_kernel(const double *__restrict__ x, int64_t n, const double *__restrict__ mi,
                                                                   const double *__restrict__ mq, const double *__restrict__ ui,
                                                                   const double *__restrict__ uq, int m, SlideTones T, Lpf8Args Q, int ml,
                                                                   int64_t nout, int G, SweepArgs P, double E, unsigned long long *__restrict__ list,
                                                                   int *__restrict__ count, int cap, int region0, SweepTail TL)
{
    extern __shared__ double xs[];
    constexpr int L = kFuseRun, TILE = kThreads * 8;
    const int t = threadIdx.x;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const int nmag = TILE + ml - 1, nruns = (nmag + L - 1) / L, xspan = nruns * L + m - 1;
    double *tp = xs + region0;
    // the workgroup's own list of uncertain (sample, modem) pairs, decided by the exact chain before the workgroup ends (sweep_tail_entry):
    // behind the band operands in the dynamic block -- as a static array it moved the block's start off its 16-byte boundary (232 bytes
    // of static LDS) and every ds_read_b128 of the planes went the slow way: this kernel 0.33 -> 1.4 ms in the pipeline
    int4v *const bl = reinterpret_cast<int4v *>(tp + 4 * m);
    unsigned *const wl = reinterpret_cast<unsigned *>(bl + 2 * kL8Dig * 64);
    float *const wmax8 = reinterpret_cast<float *>(wl + kTailCap + 4);
    if (t == 0) wl[kTailCap] = 0;
    if (((uintptr_t)x & 15) == 0 && tile0 + TILE <= n) {
        double2v v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const double2v *>(x + tile0 + 2 * (q * kThreads + t));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s0 = slide_slot<L>(2 * (q * kThreads + t));
            xs[s0] = v[q].x;
            xs[s0 + 1] = v[q].y;
        }
        for (int p = TILE + t; p < xspan; p += kThreads) {
            const int64_t gi = tile0 + p;
            xs[slide_slot<L>(p)] = gi < n ? x[gi] : 0.0;
        }
    } else {
        for (int p = t; p < xspan; p += kThreads) {
            const int64_t gi = tile0 + p;
            xs[slide_slot<L>(p)] = gi < n ? x[gi] : 0.0;
        }
    }
    lpf8_sweep_tile<ONE>(xs, reinterpret_cast<unsigned char *>(xs), tp, bl, wl, wmax8, 0, TL.lds_ok != 0, t, tile0, mi, mq, ui, uq, m, T, Q, ml, nout, G, P, E,
                         list, count, cap);
    // The workgroup's own uncertain samples, by the reference's chain, here: one in sixteen workgroups has one (750 + 220 per recording of
    // 28 000 workgroups), and it costs that workgroup a few microseconds -- as a launch of its own behind this one the same work sat on the demod
    // stream's critical path, twice per recording, waiting for slots among the other stream's workgroups (86 us per launch against 11 alone).
    __syncthreads();                                         // (vmcnt too: this workgroup's bitmap words are in memory before an atomic touches them)
    const int ne = (int)(wl[kTailCap] < (unsigned)kTailCap ? wl[kTailCap] : (unsigned)kTailCap);
    for (int e = 0; e < ne; ++e) {
        const unsigned ent = wl[e];
        const int g = (int)(ent >> 16) & 15;
        const double *si = TL.space + (size_t)g * 2 * m;
        sweep_tail_entry<kThreads>(xs, t, x, mi, mq, si, si + m, m, TL.lpf, ml, TL.src, tile0 + (int64_t)(ent & 0xFFFFu),
                                   reinterpret_cast<unsigned long long *>(P.bits[g]));
    }
}

// ---- ONE launch per recording for the AFSK stage of a chain group (round 5): band-pass, every sweep, every uncertain sample -------------
// Round 4's stage was three launches and 0.78 GB of traffic per recording: bpf8_kernel wrote the band-passed stream (230 MB of binary64),
// each of the two sweep kernels read it back -- an intermediate SURVEY 8(d) prices at zero.  Here a workgroup owns 2048 outputs of every
// sweep: it stages the int16 audio under them ONCE as digit planes, runs the band-pass on the matrix pipe (bpf8_kernel's arithmetic,
// pm_bpf8_dev.h) for the 2048 + (ml - 1) + (m - 1) values the longest sweep needs -- straight into the sliding sums' LDS window, never into
// memory -- then each sweep's tile from that window (lpf8_sweep_tile: the sweeps differ in tones, span and gains, not in their input),
// then the exact chain for whatever it could not certify.  What crosses HBM: 2 bytes per sample in, one bit per sample and chain out;
// the halo (mb + m + ml - 3 = 305 samples per 2048, 15 %) is band-passed twice, which costs 2 of the 9 + 16 + 8 matrix tiles per workgroup.
struct FusedSweep {
    const double *mi, *mq, *ui, *uq;     // templates (mark pair, unit-gain space pair)
    const double *space, *lpf;           // the exact chain's operands: the modems' own space taps, the low-pass in binary64
    const double *tg;                    // the four templates reversed and interleaved (pm_lpf8_plan::d_tpl): scalar loads in the sliding sums
    int m, ml, G, one;
    SlideTones T;
    Lpf8Args Q;
    SweepArgs P;
    double E;
    unsigned long long *list;
    int *count;
    int64_t nout;
};
struct FusedArgs {
    FusedSweep s[2];
    const int16_t *audio;
    int64_t n, nb;                       // samples; band-pass outputs (n - mb + 1)
    const pm_bpf8_dev::i4 *bp_btab;
    pm_bpf8_dev::Scales sc;
    SweepSource src;
    int xs_span;                         // band-passed values a workgroup needs: the largest runs * L + m - 1 of the sweeps
    int aplane;                          // bytes of an audio digit plane
    int xw_doubles, plane_bytes, mmax;   // LDS layout: window | planes | templates | band operands | list | maxima
    int lds_ok, cap;
};

#ifndef PM_FUSED8_WAVES
#define PM_FUSED8_WAVES 5       // compiled for five waves per SIMD (96 registers): the LDS block admits four workgroups, and four of its waves then leave a SIMD room for a slicer wave (slice_walk_kernel: 88 registers) beside them
#endif
template <int KB, bool ONE0, int NS, bool ONE1>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(PM_FUSED8_WAVES, PM_FUSED8_WAVES))) void afsk_fused8_kernel(FusedArgs A)
{
    extern __shared__ double xs[];
#if defined(PM_FUSED8_PRIO) && PM_FUSED8_PRIO > 0
    __builtin_amdgcn_s_setprio(PM_FUSED8_PRIO);              // (measurement builds: this kernel's waves above the slicers' walkers at the issue port -- profiles/r05_executor_knobs.txt)
#endif
    constexpr int L = kFuseRun, TILE = kThreads * 8;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    unsigned char *const planes = reinterpret_cast<unsigned char *>(xs + A.xw_doubles);
    double *const tp = reinterpret_cast<double *>(planes + A.plane_bytes);
    int4v *const bl = reinterpret_cast<int4v *>(tp + 4 * A.mmax);
    unsigned *const wl = reinterpret_cast<unsigned *>(bl + 2 * kL8Dig * 64);
    float *const wmax8 = reinterpret_cast<float *>(wl + kTailCap + 4);
    if (t == 0) wl[kTailCap] = 0;
    // the audio under the workgroup as two digit planes (in the low-pass planes' place: those come later); the band's operands come from
    // the plan's table block by block (in registers for the kernel's life they took it from 97 to 121: no room left on a SIMD for a
    // slicer wave beside four of these, and the slicers are the other half of the pipeline)
    unsigned char *const ap0 = planes, *const ap1 = planes + A.aplane;
    pm_bpf8_dev::stage_planes_rt(A.audio, A.n, tile0, t, kThreads, ap0, ap1, A.aplane);
    {
        lds_barrier();
        // band-pass tiles of 256 values, waves taking turns, into the window (positions past the stream: 0.0, as the split kernels stage them)
        const int r = lane & 15, g = lane >> 4;
        const int ntiles = (A.xs_span + 255) >> 8;
        for (int q = wave; q < ntiles; q += kThreads / 64) {
            double val[4];
            const pm_bpf8_dev::i4 *bt = A.bp_btab;
            asm volatile("" : "+s"(bt));                     // (opaque per tile: hoisted out of this loop the operands are 48 registers again)
            pm_bpf8_dev::tile_values_tab<KB, 4>(ap0, ap1, bt, q * 256, lane, A.sc, val);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int p = q * 256 + 16 * (4 * g + v) + r;
                if (p < A.xs_span) xs[slide_slot<L>(p)] = tile0 + p < A.nb ? val[v] : 0.0;
            }
        }
    }
    lds_barrier();
    {
        const FusedSweep &S = A.s[0];
        if (tile0 < ((S.nout + 63) >> 6) * 64)               // (uniform: a sweep with a longer correlator has fewer outputs)
            lpf8_sweep_tile<ONE0>(xs, planes, tp, bl, wl, wmax8, 0, A.lds_ok != 0, t, tile0, S.mi, S.mq, S.ui, S.uq, S.m, S.T, S.Q, S.ml, S.nout, S.G, S.P, S.E,
                                  S.list, S.count, A.cap, S.tg);
    }
    if (NS == 2) {
        lds_barrier();                                       // every wave is through with the first sweep's templates, planes and band
        const FusedSweep &S = A.s[1];
        if (tile0 < ((S.nout + 63) >> 6) * 64)
            lpf8_sweep_tile<ONE1>(xs, planes, tp, bl, wl, wmax8, 1, A.lds_ok != 0, t, tile0, S.mi, S.mq, S.ui, S.uq, S.m, S.T, S.Q, S.ml, S.nout, S.G, S.P, S.E,
                                  S.list, S.count, A.cap, S.tg);
    }
    // the workgroup's own uncertain samples, by the reference's chain from the audio (see afsk_slide_lpf8_kernel)
    __syncthreads();
    const int ne = (int)(wl[kTailCap] < (unsigned)kTailCap ? wl[kTailCap] : (unsigned)kTailCap);
    for (int e = 0; e < ne; ++e) {
        const unsigned ent = wl[e];
        const int g = (int)(ent >> 16) & 15;
        const int64_t k = tile0 + (int64_t)(ent & 0xFFFFu);
        if (NS == 1 || (ent >> 20) == 0) {
            const FusedSweep &S = A.s[0];
            const double *si = S.space + (size_t)g * 2 * S.m;
            sweep_tail_entry<kThreads>(xs, t, nullptr, S.mi, S.mq, si, si + S.m, S.m, S.lpf, S.ml, A.src, k, reinterpret_cast<unsigned long long *>(S.P.bits[g]));
        } else {
            const FusedSweep &S = A.s[1];
            const double *si = S.space + (size_t)g * 2 * S.m;
            sweep_tail_entry<kThreads>(xs, t, nullptr, S.mi, S.mq, si, si + S.m, S.m, S.lpf, S.ml, A.src, k, reinterpret_cast<unsigned long long *>(S.P.bits[g]));
        }
    }
}

// every sweep's counter into its page-locked word, and the counters back to zero for the block's next recording (the fused launch has
// no band-pass kernel in front of it to clear them); `keep` holds the counts for whoever works a list off later
__global__ void sweep_mail_reset_kernel(int *__restrict__ count, int *__restrict__ keep, int *__restrict__ mail, int n)
{
    if ((int)threadIdx.x < n) {
        const int c = count[threadIdx.x];
        keep[threadIdx.x] = c;
        mail[threadIdx.x] = c;
        count[threadIdx.x] = 0;
    }
    __threadfence_system();
}

// The exact chain for single samples: correlator bank of modem g at the ml positions the low-pass needs, then the low-pass, every
// sum in the canonical order of afsk_correlate_kernel / fir_valid_kernel.  Runs after fir_sweep_kernel (its bitmap bytes are final).
// AUDIO: the band-passed stream the sweep saw was itself a value with a bound (pm_bpf8.hip), so the recomputation starts one stage
// earlier -- the mc + ml - 1 band-pass outputs under the entry from the int16 audio, the reference's sum in fir_valid_kernel's order.
template <bool AUDIO>
__global__ __launch_bounds__(64) void sweep_exact_kernel(const double *__restrict__ x, const double *__restrict__ mi, const double *__restrict__ mq,
                                                         const double *__restrict__ space, int mc, const double *__restrict__ lpf, int ml,
                                                         SweepArgs P, const unsigned long long *__restrict__ list, const int *__restrict__ count, int cap,
                                                         int *__restrict__ reset, int *__restrict__ mail, SweepSource src)
{
    // deferred fallback (pm_afsk_sweep_mode): this is the sweep's last launch and clears the next sweep's counter (see d_sweep);
    // it also leaves the counter in a page-locked host word, so that the caller who waits for the recording's event anyway reads it
    // without a copy and a stream wait of its own
#ifndef PM_EXACT_PRIO
#define PM_EXACT_PRIO 3
#endif
    // a few hundred lone waves, each a chain of dependent sums, between a recording's two sweeps on the demod stream: every issue slot
    // they lose to the filter and slicer waves beside them is time the whole recording waits (measured in the pipeline: 0.10 ms per
    // launch at the default priority against 0.011 alone)
    __builtin_amdgcn_s_setprio(PM_EXACT_PRIO);
    if (reset && blockIdx.x == 0 && threadIdx.x == 0) *reset = 0;
    if (mail && blockIdx.x == 0 && threadIdx.x == 0) {
        *mail = *count;
        __threadfence_system();
    }
    // One wave per listed sample: the ml correlator-bank outputs the low-pass needs are independent of each other and go to the
    // lanes (each in the canonical tap order); the low-pass sum itself is sequential and stays with lane 0.  (One LANE per sample
    // took 0.25-0.5 ms for a single entry -- 4 mc ml dependent fmas -- and the demod stream waits for it.)
    extern __shared__ double dd[];
    const int lane = threadIdx.x;
    const int cnt = min(*count, cap);
    for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
        const int g = (int)(list[e] >> 48);
        const int64_t k = (int64_t)(list[e] & 0xFFFFFFFFFFFFull);
        if (g >= kSweepMax || P.bits[g] == nullptr) continue;                    // not an entry of this sweep (cannot happen: see sweep_signs)
        const double *si = space + (size_t)g * 2 * mc, *sq = si + mc;
        if (AUDIO) {
            // The audio under the entry and every tap set once, coalesced, into LDS; then the sums from there.  (From global memory --
            // a tap load in front of every fma of three chained sums -- this kernel took 160-200 us for twenty entries, on the demod
            // stream, twice per recording.)
            const int nw = ml + mc - 1, na = nw + src.mb - 1;
            double *xw = dd + ml, *aw = xw + nw, *tb = aw + na, *tc = tb + src.mb, *tl = tc + 4 * mc;
            for (int p = lane; p < na; p += 64) aw[p] = (double)src.audio[k + p];
            for (int t = lane; t < src.mb; t += 64) tb[t] = src.bpf[src.mb - 1 - t];
            for (int t = lane; t < mc; t += 64) {
                tc[4 * t + 0] = mi[mc - 1 - t];
                tc[4 * t + 1] = mq[mc - 1 - t];
                tc[4 * t + 2] = si[mc - 1 - t];
                tc[4 * t + 3] = sq[mc - 1 - t];
            }
            for (int t = lane; t < ml; t += 64) tl[t] = lpf[ml - 1 - t];
            __syncthreads();
            for (int p = lane; p < nw; p += 64) {
                double acc = 0.0;
                for (int t = 0; t < src.mb; ++t) acc = __builtin_fma(tb[t], aw[p + t], acc);
                xw[p] = acc;
            }
            __syncthreads();
            for (int j = lane; j < ml; j += 64) {
                double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
                for (int t = 0; t < mc; ++t) {
                    const double v = xw[j + t];
                    a = __builtin_fma(tc[4 * t + 0], v, a);
                    b = __builtin_fma(tc[4 * t + 1], v, b);
                    c = __builtin_fma(tc[4 * t + 2], v, c);
                    d = __builtin_fma(tc[4 * t + 3], v, d);
                }
                dd[j] = __builtin_sqrt(a * a + b * b) - __builtin_sqrt(c * c + d * d);
            }
            __syncthreads();
            if (lane == 0) {
                double acc = 0.0;
                for (int j = 0; j < ml; ++j) acc = __builtin_fma(tl[j], dd[j], acc);
                unsigned long long *w = reinterpret_cast<unsigned long long *>(P.bits[g]) + (k >> 6);
                const unsigned long long bit = 1ull << (k & 63);
                if (acc >= 0.0) atomicOr(w, bit); else atomicAnd(w, ~bit);
            }
            __syncthreads();
            continue;
        }
        for (int j = lane; j < ml; j += 64) {
            const double *xp = x + k + j;
            double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
            for (int t = 0; t < mc; ++t) {
                const double v = xp[t];
                a = __builtin_fma(mi[mc - 1 - t], v, a);
                b = __builtin_fma(mq[mc - 1 - t], v, b);
                c = __builtin_fma(si[mc - 1 - t], v, c);
                d = __builtin_fma(sq[mc - 1 - t], v, d);
            }
            const double mark = __builtin_sqrt(a * a + b * b);
            const double spc = __builtin_sqrt(c * c + d * d);
            dd[j] = mark - spc;
        }
        __syncthreads();
        if (lane == 0) {
            double acc = 0.0;
            for (int j = 0; j < ml; ++j) acc = __builtin_fma(lpf[ml - 1 - j], dd[j], acc);
            unsigned long long *w = reinterpret_cast<unsigned long long *>(P.bits[g]) + (k >> 6);
            const unsigned long long bit = 1ull << (k & 63);
            if (acc >= 0.0) atomicOr(w, bit); else atomicAnd(w, ~bit);
        }
        __syncthreads();
    }
}

// One 64-bit word per wave per step: lane l tests sample 64*w + l, the ballot is the word.
__global__ __launch_bounds__(kThreads) void signs_kernel(const double *__restrict__ x, int64_t n,
                                                         uint64_t *__restrict__ bits, int64_t nwords)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t w = wave; w < nwords; w += nwaves) {
        const int64_t k = w * 64 + lane;
        const bool p = (k < n) && (x[k] >= 0.0);
        const uint64_t mask = __ballot(p);
        if (lane == 0) bits[w] = mask;
    }
}

template <int R>
size_t lds_bytes(int m) { return (size_t)(slot<R>(kThreads * R + m - 1) + 2) * sizeof(double); }

template <typename K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        PM_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PM_OK;
}

template <typename InT, bool NEG>
int fir_launch2(pm_ctx *ctx, const InT *d_x, int64_t n, const double *d_taps, int m, double *d_y, uint64_t *d_bits)
{
    constexpr int R = 8;
    const int64_t nout = n - m + 1;
    const int64_t ntiles = pm_cdiv(nout, (int64_t)kThreads * R);
    PM_ARG(ntiles < (1LL << 31));
    const size_t lds = lds_bytes<R>(m);
    PmProf prof(ctx, sizeof(InT) == 2 ? PM_K_FIR_I16 : PM_K_FIR_F64);
    prof.work((double)n * sizeof(InT) + (d_bits ? (double)nout / 8 : (double)nout * 8), 2.0 * m * (double)nout);
    const bool vec = (((uintptr_t)d_x | (uintptr_t)d_y) & 15) == 0;          // 16-byte loads and stores
    if (sizeof(InT) == 2 && d_bits && !d_y && vec && m >= 2 && m <= 8 && !ctx->tune.fir_no_short) {
        // the register path for short taps (fir_short_signs_i16_kernel): one lane per bitmap byte
        const int64_t lanes = ((nout + 63) >> 6) * 8;
        const unsigned grid = (unsigned)pm_cdiv(lanes, (int64_t)kThreads * kShortIter);
        const int16_t *xs16 = reinterpret_cast<const int16_t *>(d_x);
#define PM_FIR_SHORT(MM) case MM: hipLaunchKernelGGL((fir_short_signs_i16_kernel<MM, NEG>), dim3(grid), dim3(kThreads), 0, ctx->stream, xs16, n, d_taps, nout, d_bits); break;
        switch (m) { PM_FIR_SHORT(2) PM_FIR_SHORT(3) PM_FIR_SHORT(4) PM_FIR_SHORT(5) PM_FIR_SHORT(6) PM_FIR_SHORT(7) PM_FIR_SHORT(8) }
#undef PM_FIR_SHORT
        PM_HIP(hipGetLastError());
        return PM_OK;
    }
#define PM_FIR_GO(VECF, SIGNF)                                                                                              \
    {                                                                                                                       \
        if (int rc = allow_lds(fir_valid_kernel<InT, R, NEG, VECF, SIGNF>, lds)) return rc;                                  \
        hipLaunchKernelGGL((fir_valid_kernel<InT, R, NEG, VECF, SIGNF>), dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream, \
                           d_x, n, d_taps, m, d_y, nout, d_bits);                                                           \
    }
    if (d_bits) {
        if (vec) PM_FIR_GO(true, true) else PM_FIR_GO(false, true)
    } else {
        if (vec) PM_FIR_GO(true, false) else PM_FIR_GO(false, false)
    }
#undef PM_FIR_GO
    PM_HIP(hipGetLastError());
    return PM_OK;
}

template <typename InT>
int fir_launch(pm_ctx *ctx, const InT *d_x, int64_t n, const double *d_taps, int m, double *d_y, uint64_t *d_bits, int flags)
{
    PM_CTX(ctx);
    PM_ARG(d_x && d_taps && (d_y || d_bits));
    PM_ARG(m >= 1 && m <= kMaxTaps);
    PM_ARG(n >= m);
    return (flags & PM_FIR_NEGATE) ? fir_launch2<InT, true>(ctx, d_x, n, d_taps, m, d_y, d_bits)
                                   : fir_launch2<InT, false>(ctx, d_x, n, d_taps, m, d_y, d_bits);
}

template <typename InT, bool NEG>
int fir_rows_launch2(pm_ctx *ctx, const FirRows &A, bool vec, int rows, int64_t n, const double *d_taps, int m)
{
    constexpr int R = 8;
    const int64_t nout = n - m + 1;
    const int64_t ntiles = pm_cdiv(nout, (int64_t)kThreads * R);
    PM_ARG(ntiles < (1LL << 31) && rows >= 1 && rows <= 65535);
    const size_t lds = lds_bytes<R>(m);
    PmProf prof(ctx, sizeof(InT) == 2 ? PM_K_FIR_I16 : PM_K_FIR_F64);
    prof.work(rows * ((double)n * sizeof(InT) + (A.bits ? (double)nout / 8 : (double)nout * 8)), 2.0 * m * (double)nout * rows);
#define PM_ROWS_GO(VECF, SIGNF)                                                                                                   \
    {                                                                                                                             \
        if (int rc = allow_lds(fir_rows_kernel<InT, R, NEG, VECF, SIGNF>, lds)) return rc;                                         \
        hipLaunchKernelGGL((fir_rows_kernel<InT, R, NEG, VECF, SIGNF>), dim3((unsigned)ntiles, (unsigned)rows), dim3(kThreads), lds, \
                           ctx->stream, A, n, d_taps, m, nout);                                                                   \
    }
    if (A.bits) {
        if (vec) PM_ROWS_GO(true, true) else PM_ROWS_GO(false, true)
    } else {
        if (vec) PM_ROWS_GO(true, false) else PM_ROWS_GO(false, false)
    }
#undef PM_ROWS_GO
    PM_HIP(hipGetLastError());
    return PM_OK;
}

}  // namespace

// Internal (pm_common.h): what pm_fir_rows_* and the batch engine launch.  `x_aligned16`: every input row starts on a 16-byte boundary.
int pm_fir_rows(pm_ctx *ctx, bool i16, const void *d_x, int64_t x_stride, const void *const *d_x_ptrs, int64_t x_off, bool x_aligned16, int rows,
                int64_t n, const double *d_taps, int m, double *d_y, int64_t y_stride, uint64_t *d_bits, int64_t bits_stride, int flags)
{
    PM_CTX(ctx);
    PM_ARG((d_x || d_x_ptrs) && d_taps && ((d_y != nullptr) != (d_bits != nullptr)));
    PM_ARG(m >= 1 && m <= kMaxTaps && n >= m && rows >= 1);
    PM_ARG(d_x_ptrs || rows == 1 || x_stride >= n);
    const int64_t nout = n - m + 1;
    PM_ARG(rows == 1 || (d_y ? y_stride >= nout : bits_stride >= (nout + 63) / 64));
    const bool vec = x_aligned16 && (!d_y || ((((uintptr_t)d_y) & 15) == 0 && y_stride % 2 == 0));
    const bool neg = (flags & PM_FIR_NEGATE) != 0;
    // rows are the grid's y dimension: more than 65535 of them (the batch engine's R x C streams go up to 2^20) in several launches
    constexpr int kRowsPerLaunch = 65535;
    for (int r0 = 0; r0 < rows; r0 += kRowsPerLaunch) {
        const int nr = std::min(kRowsPerLaunch, rows - r0);
        FirRows A{d_x ? (const void *)((const char *)d_x + (size_t)r0 * (size_t)x_stride * (i16 ? 2 : 8)) : nullptr, x_stride, d_x_ptrs ? d_x_ptrs + r0 : nullptr, x_off,
                  d_y ? d_y + (int64_t)r0 * y_stride : nullptr, y_stride, d_bits ? d_bits + (int64_t)r0 * bits_stride : nullptr, bits_stride};
        const int rc = i16 ? (neg ? fir_rows_launch2<int16_t, true>(ctx, A, vec, nr, n, d_taps, m) : fir_rows_launch2<int16_t, false>(ctx, A, vec, nr, n, d_taps, m))
                           : (neg ? fir_rows_launch2<double, true>(ctx, A, vec, nr, n, d_taps, m) : fir_rows_launch2<double, false>(ctx, A, vec, nr, n, d_taps, m));
        if (rc) return rc;
    }
    return PM_OK;
}

template <int G>
static int afsk_group_go(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_w, int m, double *d_y, int64_t y_stride, int64_t nout,
                         const int *gate = nullptr, int gate_above = 0)
{
    constexpr int R = 2;
    const int64_t ntiles = pm_cdiv(nout, (int64_t)kThreads * R);
    PM_ARG(ntiles < (1LL << 31));
    const size_t lds = lds_bytes<R>(m) + 4 * (R + 1) * sizeof(double);        // the last block's look-ahead load
    const bool vec = ((((uintptr_t)d_x | (uintptr_t)d_y) & 15) == 0) && (y_stride % 2 == 0);
    // a gated launch (the certified path's overflow fallback, normally every workgroup leaves at once) is booked with the
    // fallback machinery, not with the correlators
    PmProf prof(ctx, gate ? PM_K_SIGNS : PM_K_AFSK_CORR);
    if (!gate) prof.work((double)n * 8 + (double)G * nout * 8, 2.0 * (2 + 2 * G) * m * (double)nout);
    const unsigned grid = (unsigned)(gate ? std::min<int64_t>(ntiles, kGatedGrid) : ntiles);
    if (vec) {
        if (int rc = allow_lds(afsk_group_kernel<G, true>, lds)) return rc;
        hipLaunchKernelGGL((afsk_group_kernel<G, true>), dim3(grid), dim3(kThreads), lds, ctx->stream, d_x, n, d_w, m, d_y, y_stride, nout,
                           gate, gate_above);
    } else {
        if (int rc = allow_lds(afsk_group_kernel<G, false>, lds)) return rc;
        hipLaunchKernelGGL((afsk_group_kernel<G, false>), dim3(grid), dim3(kThreads), lds, ctx->stream, d_x, n, d_w, m, d_y, y_stride, nout,
                           gate, gate_above);
    }
    PM_HIP(hipGetLastError());
    return PM_OK;
}


extern "C" {

int pm_fir_valid_i16(pm_ctx *ctx, const int16_t *d_x, int64_t n, const double *d_taps, int m, double *d_y, int flags)
{
    return fir_launch<int16_t>(ctx, d_x, n, d_taps, m, d_y, nullptr, flags);
}

int pm_fir_signs_i16(pm_ctx *ctx, const int16_t *d_x, int64_t n, const double *d_taps, int m, uint64_t *d_bits, int flags)
{
    return fir_launch<int16_t>(ctx, d_x, n, d_taps, m, nullptr, d_bits, flags);
}

int pm_fir_signs_f64(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_taps, int m, uint64_t *d_bits, int flags)
{
    return fir_launch<double>(ctx, d_x, n, d_taps, m, nullptr, d_bits, flags);
}

int pm_fir_signs_f64_batch(pm_ctx *ctx, int count, const double *const *h_x, const int64_t *h_n, const double *d_taps, int m,
                           uint64_t *const *h_bits, int flags)
{
    PM_CTX(ctx);
    PM_ARG(count >= 1 && count <= kFirBatchMax && h_x && h_n && d_taps && h_bits);
    PM_ARG(m >= 1 && m <= kMaxTaps);
    constexpr int R = 8;
    FirBatch B;
    memset(&B, 0, sizeof(B));
    int64_t longest = 0;
    bool vec = true;
    double bytes = 0, flops = 0;
    for (int k = 0; k < count; ++k) {
        PM_ARG(h_x[k] && h_bits[k] && h_n[k] >= m);
        B.x[k] = h_x[k];
        B.bits[k] = h_bits[k];
        B.n[k] = h_n[k];
        longest = std::max(longest, h_n[k] - m + 1);
        vec = vec && (((uintptr_t)h_x[k]) & 15) == 0;
        bytes += (double)h_n[k] * 8 + (double)(h_n[k] - m + 1) / 8;
        flops += 2.0 * m * (double)(h_n[k] - m + 1);
    }
    const int64_t ntiles = pm_cdiv(longest, (int64_t)kThreads * R);
    PM_ARG(ntiles < (1LL << 31));
    const size_t lds = lds_bytes<R>(m);
    const bool neg = (flags & PM_FIR_NEGATE) != 0;
    PmProf prof(ctx, PM_K_FIR_F64);
    prof.work(bytes, flops);
#define PM_BATCH_GO(NEGF, VECF)                                                                                             \
    {                                                                                                                       \
        if (int rc = allow_lds(fir_signs_batch_kernel<R, NEGF, VECF>, lds)) return rc;                                       \
        hipLaunchKernelGGL((fir_signs_batch_kernel<R, NEGF, VECF>), dim3((unsigned)ntiles, (unsigned)count), dim3(kThreads), lds, ctx->stream, \
                           B, d_taps, m);                                                                                   \
    }
    if (neg) { if (vec) PM_BATCH_GO(true, true) else PM_BATCH_GO(true, false) }
    else { if (vec) PM_BATCH_GO(false, true) else PM_BATCH_GO(false, false) }
#undef PM_BATCH_GO
    PM_HIP(hipGetLastError());
    return PM_OK;
}

int pm_fir_valid_f64(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_taps, int m, double *d_y, int flags)
{
    return fir_launch<double>(ctx, d_x, n, d_taps, m, d_y, nullptr, flags);
}

static bool rows_aligned16(const void *base, int64_t stride_elems, size_t elem, int rows)
{
    return (((uintptr_t)base) & 15) == 0 && (rows == 1 || (stride_elems * (int64_t)elem) % 16 == 0);
}

int pm_fir_rows_i16(pm_ctx *ctx, const int16_t *d_x, int64_t x_stride, int rows, int64_t n, const double *d_taps, int m, double *d_y,
                    int64_t y_stride, int flags)
{
    PM_ARG(d_x != nullptr);
    return pm_fir_rows(ctx, true, d_x, x_stride, nullptr, 0, rows_aligned16(d_x, x_stride, 2, rows), rows, n, d_taps, m, d_y, y_stride, nullptr, 0, flags);
}

int pm_fir_rows_i16_ptrs(pm_ctx *ctx, const int16_t *const *d_x_ptrs, int64_t x_off, int x_aligned16, int rows, int64_t n, const double *d_taps,
                         int m, double *d_y, int64_t y_stride, int flags)
{
    PM_ARG(d_x_ptrs != nullptr);
    return pm_fir_rows(ctx, true, nullptr, 0, (const void *const *)d_x_ptrs, x_off, x_aligned16 != 0, rows, n, d_taps, m, d_y, y_stride, nullptr, 0, flags);
}

int pm_fir_rows_f64(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, const double *d_taps, int m, double *d_y,
                    int64_t y_stride, int flags)
{
    PM_ARG(d_x != nullptr);
    return pm_fir_rows(ctx, false, d_x, x_stride, nullptr, 0, rows_aligned16(d_x, x_stride, 8, rows), rows, n, d_taps, m, d_y, y_stride, nullptr, 0, flags);
}

int pm_fir_rows_signs_f64(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, const double *d_taps, int m, uint64_t *d_bits,
                          int64_t bits_stride, int flags)
{
    PM_ARG(d_x != nullptr);
    return pm_fir_rows(ctx, false, d_x, x_stride, nullptr, 0, rows_aligned16(d_x, x_stride, 8, rows), rows, n, d_taps, m, nullptr, 0, d_bits, bits_stride, flags);
}

int pm_afsk_correlate(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_mark_i, const double *d_mark_q,
                      const double *d_space_i, const double *d_space_q, int m, double *d_y)
{
    PM_CTX(ctx);
    PM_ARG(ctx && d_x && d_mark_i && d_mark_q && d_space_i && d_space_q && d_y);
    PM_ARG(m >= 1 && m <= kMaxTaps);
    PM_ARG(n >= m);
    constexpr int R = 4;
    const int64_t nout = n - m + 1;
    const int64_t ntiles = pm_cdiv(nout, (int64_t)kThreads * R);
    PM_ARG(ntiles < (1LL << 31));
    const size_t lds = lds_bytes<R>(m);
    PmProf prof(ctx, PM_K_AFSK_CORR);
    prof.work((double)n * 8 + (double)nout * 8, 2.0 * 4 * m * (double)nout);
    const bool vec = (((uintptr_t)d_x | (uintptr_t)d_y) & 15) == 0;
    if (vec) {
        if (int rc = allow_lds(afsk_correlate_kernel<R, true>, lds)) return rc;
        hipLaunchKernelGGL((afsk_correlate_kernel<R, true>), dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream,
                           d_x, n, d_mark_i, d_mark_q, d_space_i, d_space_q, m, d_y, nout);
    } else {
        if (int rc = allow_lds(afsk_correlate_kernel<R, false>, lds)) return rc;
        hipLaunchKernelGGL((afsk_correlate_kernel<R, false>), dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream,
                           d_x, n, d_mark_i, d_mark_q, d_space_i, d_space_q, m, d_y, nout);
    }
    PM_HIP(hipGetLastError());
    return PM_OK;
}

int pm_afsk_correlate_group(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_mark_i, const double *d_mark_q,
                            const double *d_space, int groups, int m, double *d_y, int64_t y_stride)
{
    PM_CTX(ctx);
    PM_ARG(d_x && d_mark_i && d_mark_q && d_space && d_y);
    PM_ARG(groups >= 1 && groups <= PM_AFSK_GROUP_MAX);
    PM_ARG(m >= 1 && m <= kMaxTaps);
    PM_ARG(n >= m);
    const int64_t nout = n - m + 1;
    PM_ARG(groups == 1 || y_stride >= nout);
    const int F = 2 + 2 * groups;
    if (int rc = pm_scratch_reserve(ctx, (size_t)F * m * sizeof(double))) return rc;
    double *d_w = (double *)ctx->d_scratch;
    hipLaunchKernelGGL(pack_group_taps_kernel, dim3((unsigned)pm_cdiv((int64_t)F * m, 256)), dim3(256), 0, ctx->stream,
                       d_mark_i, d_mark_q, d_space, m, F, d_w);
    PM_HIP(hipGetLastError());
    switch (groups) {
    case 1: return afsk_group_go<1>(ctx, d_x, n, d_w, m, d_y, y_stride, nout);
    case 2: return afsk_group_go<2>(ctx, d_x, n, d_w, m, d_y, y_stride, nout);
    case 3: return afsk_group_go<3>(ctx, d_x, n, d_w, m, d_y, y_stride, nout);
    case 4: return afsk_group_go<4>(ctx, d_x, n, d_w, m, d_y, y_stride, nout);
    case 5: return afsk_group_go<5>(ctx, d_x, n, d_w, m, d_y, y_stride, nout);
    case 6: return afsk_group_go<6>(ctx, d_x, n, d_w, m, d_y, y_stride, nout);
    case 7: return afsk_group_go<7>(ctx, d_x, n, d_w, m, d_y, y_stride, nout);
    default: return afsk_group_go<8>(ctx, d_x, n, d_w, m, d_y, y_stride, nout);
    }
}

static int afsk_group_dispatch(pm_ctx *ctx, int groups, const double *d_x, int64_t n, const double *d_w, int m, double *d_y, int64_t y_stride,
                               int64_t nout, const int *gate, int gate_above)
{
    switch (groups) {
    case 1: return afsk_group_go<1>(ctx, d_x, n, d_w, m, d_y, y_stride, nout, gate, gate_above);
    case 2: return afsk_group_go<2>(ctx, d_x, n, d_w, m, d_y, y_stride, nout, gate, gate_above);
    case 3: return afsk_group_go<3>(ctx, d_x, n, d_w, m, d_y, y_stride, nout, gate, gate_above);
    case 4: return afsk_group_go<4>(ctx, d_x, n, d_w, m, d_y, y_stride, nout, gate, gate_above);
    case 5: return afsk_group_go<5>(ctx, d_x, n, d_w, m, d_y, y_stride, nout, gate, gate_above);
    case 6: return afsk_group_go<6>(ctx, d_x, n, d_w, m, d_y, y_stride, nout, gate, gate_above);
    case 7: return afsk_group_go<7>(ctx, d_x, n, d_w, m, d_y, y_stride, nout, gate, gate_above);
    default: return afsk_group_go<8>(ctx, d_x, n, d_w, m, d_y, y_stride, nout, gate, gate_above);
    }
}

// Bound on |sliding magnitude - magnitude of the direct sums| for runs of `steps` (derivation: afsk_magnitudes).
static double slide_bound(const pm_afsk_tones *tones, int m, double x_bound, int steps = 16)
{
    const double u = 1.1102230246251565e-16;
    // (last term: slide_sqrt's one Newton step, relative 1.5e-12 of a magnitude that is at most sqrt2 m x_bound)
    return (16.0 * steps * u * (m + 1) + 3.0 * m * tones->tap_dev + 3.0 * u * m * m + 6.0 * u * m + 1.5e-12 * 1.4143 * m) * x_bound;
}

// M = |mark correlators|, S = |unit-gain space correlators| over x, one stream each (nc = n - m + 1 values): by the sliding sum when
// `tones` describes the templates, else by the direct sums.  *e_slide = bound on |sliding value - direct value| (0 for the direct sums).
static int afsk_magnitudes(pm_ctx *ctx, const double *d_x, int64_t n, double x_bound, const double *d_mark_i, const double *d_mark_q,
                           const double *d_unit_i, const double *d_unit_q, int m, const pm_afsk_tones *tones, double *M, double *S,
                           double *e_slide, double diff_gain = 0.0)
{
    PM_ARG(S || (tones && m >= 2));                          // the one-stream difference exists for the sliding sums only
    const int64_t nc = n - m + 1;
    constexpr int kRun = 16;
    *e_slide = 0.0;
    if (tones && m >= 2) {
        PM_ARG(tones->tap_dev >= 0.0 && tones->tap_dev < 1e-6);
        // |sliding value - the reference's computed sum| at step i of a run that started from the reference's own sum (same taps,
        // same order: no difference at i = 0).  With Z_model the exact sums of the power filter r^j:  the run follows Z_model from a
        // start that is off it by (tap deviation + the start sum's rounding), rotated, and the reference's sum at step i is off
        // Z_model by the same two kinds of term:  2 sqrt2 m tap_dev X  +  2 sqrt2 m^2 u X.  On top, per step, 6 roundings of
        // sums bounded by (m + 2) X and r^m being off by sqrt2 u:  < 16 u (m + 1) X, over at most kRun steps.  The magnitude is
        // 1-Lipschitz in the pair of sums and its own three roundings are the same on both sides up to 3 u m X.
        *e_slide = slide_bound(tones, m, x_bound);
        SlideTones T{tones->mark_rot[0], tones->mark_rot[1], tones->mark_end[0], tones->mark_end[1],
                     tones->space_rot[0], tones->space_rot[1], tones->space_end[0], tones->space_end[1]};
        const int64_t ntiles = pm_cdiv(nc, (int64_t)kSlideThreads * kRun);
        PM_ARG(ntiles < (1LL << 31));
        const size_t lds = slide_lds_bytes<kRun>(m);
        if (lds > 160 * 1024) return pm_set_error(PM_ERR_ARG, "sliding correlator sums: %d taps do not fit the LDS tile", m);
        PmProf prof(ctx, PM_K_AFSK_CORR);
        prof.work((double)n * 8 + (S ? 2.0 : 1.0) * nc * 8, (4.0 * m / kRun + 18.0) * (double)nc);
        if (int rc = allow_lds(afsk_slide_kernel<kRun>, lds)) return rc;
        hipLaunchKernelGGL((afsk_slide_kernel<kRun>), dim3((unsigned)ntiles), dim3(kSlideThreads), lds, ctx->stream, d_x, n, d_mark_i, d_mark_q,
                           d_unit_i, d_unit_q, m, T, M, S, nc, diff_gain);
        PM_HIP(hipGetLastError());
        return PM_OK;
    }
    constexpr int R = 4;
    const int64_t ntiles = pm_cdiv(nc, (int64_t)kThreads * R);
    PM_ARG(ntiles < (1LL << 31));
    const size_t lds = lds_bytes<R>(m);
    PmProf prof(ctx, PM_K_AFSK_CORR);
    prof.work((double)n * 8 + 2.0 * nc * 8, 2.0 * 4 * m * (double)nc);
    if ((((uintptr_t)d_x) & 15) == 0) {
        if (int rc = allow_lds(afsk_correlate_kernel<R, true, true>, lds)) return rc;
        hipLaunchKernelGGL((afsk_correlate_kernel<R, true, true>), dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream, d_x, n, d_mark_i,
                           d_mark_q, d_unit_i, d_unit_q, m, M, nc, S);
    } else {
        if (int rc = allow_lds(afsk_correlate_kernel<R, false, true>, lds)) return rc;
        hipLaunchKernelGGL((afsk_correlate_kernel<R, false, true>), dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream, d_x, n, d_mark_i,
                           d_mark_q, d_unit_i, d_unit_q, m, M, nc, S);
    }
    PM_HIP(hipGetLastError());
    return PM_OK;
}

// one wave per uncertain decision at a time: as many workgroups as a sweep usually has entries (several hundred; an idle one costs a
// dispatch slot for a microsecond), so that the launch lasts one entry's latency and not three
constexpr int kExactGrid = 4096;

static int sweep_signs(pm_ctx *ctx, const double *d_x, int64_t n, double x_bound, const double *d_mark_i, const double *d_mark_q,
                       const double *d_unit_i, const double *d_unit_q, const double *d_space, const double *h_gains, int groups, int m,
                       const double *d_lpf, int ml, double lpf_abs_sum, uint64_t *const *h_bits, const pm_afsk_tones *tones,
                       const SweepSource *src = nullptr, const pm_lpf8_plan *lpf8 = nullptr, int *own_count = nullptr, int *own_mail = nullptr,
                       unsigned long long *own_list = nullptr)
{
    PM_CTX(ctx);
    // own_list (with own_count): the recording's own list of kSweepCap entries -- a sweep on the matrix pipe then decides its uncertain
    // samples inside its workgroups, counts what did not fit (normally nothing) in own_count, NO launch follows it and own_mail is not
    // written: the caller mails the counters of all the recording's sweeps at once (sweep_mail_kernel) and whoever reads the mail runs
    // pm_afsk_sweep_exact_list over a list that is not empty
    PM_ARG(!own_list || own_count);
    // own_count / own_mail: the caller's counter and mailbox word for this sweep (pm_sweep_cells: zeroed by an earlier launch on this
    // stream, read by the caller when the stream has passed this sweep); the context's ring stays where it is
    PM_ARG((own_count == nullptr) == (own_mail == nullptr) && (!own_count || ctx->sweep_deferred));
    PM_ARG(d_x && d_mark_i && d_mark_q && d_unit_i && d_unit_q && d_space && h_gains && d_lpf && h_bits);
    // a band-passed stream that is only near the reference's: certified decisions with the deferred fallback only (the gated exact
    // launches below read d_x), and the exact recomputation goes back to the audio
    PM_ARG(!src || (src->audio && src->bpf && src->mb >= 1 && src->e_x >= 0.0 && src->e_x < 1e-6 * x_bound && ctx->sweep_deferred && tones));
    // everything that can refuse the call is checked before the counter ring moves on: a sweep that takes its slot and then launches
    // nothing leaves the NEXT sweep's slot uncleared (each sweep's last launch clears it), and that sweep would start from whatever
    // count the slot held 64 sweeps ago -- up to 65536 stale list entries to "recompute"
    PM_ARG(!tones || (tones->tap_dev >= 0.0 && tones->tap_dev < 1e-6));
    PM_ARG(groups >= 1 && groups <= kSweepMax && m >= 1 && m <= kMaxTaps && ml >= 1 && ml <= kMaxTaps);
    PM_ARG(x_bound > 0.0 && x_bound < 1e300 && lpf_abs_sum > 0.0 && lpf_abs_sum < 1e300);
    PM_ARG(n >= (int64_t)m + ml - 1);
    const int64_t nc = n - m + 1, nl = nc - ml + 1;
    SweepArgs P;
    memset(&P, 0, sizeof(P));
    double gmax = 0.0;
    for (int g = 0; g < groups; ++g) {
        PM_ARG(h_bits[g] != nullptr && h_gains[g] >= 0.0 && h_gains[g] < 1e100);
        P.gain[g] = h_gains[g];
        P.bits[g] = h_bits[g];
        gmax = std::max(gmax, h_gains[g]);
    }
    const int cap = kSweepCap;                             // more uncertain samples than this: the gated exact path below takes over
    const int F = 2 + 2 * groups;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int64_t stride = (nc + 63) / 64 * 64;
    const size_t b_m = up((size_t)nc * 8), b_a = up((size_t)nl * 8), b_list = up((size_t)cap * 8), b_w = up((size_t)F * m * 8),
                 b_c = up((size_t)stride * groups * 8);
    if (int rc = pm_scratch_reserve(ctx, 2 * b_m + b_a + b_list + 256 + b_w + b_c)) return rc;
    char *base = (char *)ctx->d_scratch;
    double *M = (double *)base, *S = (double *)(base + b_m), *A = (double *)(base + 2 * b_m);
    unsigned long long *list = own_list ? own_list : (unsigned long long *)(base + 2 * b_m + b_a);
    // The counter of uncertain samples lives in a small ring of its own (not in the scratch block, which the next call on this
    // context re-carves and may re-allocate: pm_afsk_sweep_last reads it later).  All slots start at zero; the last launch of a
    // sweep clears the slot the next sweep will use, so there is no memset on the stream.
    if (!own_count && !ctx->d_sweep) {
        PM_HIP(hipMalloc((void **)&ctx->d_sweep, kSweepRing * sizeof(int)));
        PM_HIP(hipMemset(ctx->d_sweep, 0, kSweepRing * sizeof(int)));
        PM_HIP(hipHostMalloc((void **)&ctx->h_sweep, kSweepRing * sizeof(int), hipHostMallocDefault));
        memset(ctx->h_sweep, 0, kSweepRing * sizeof(int));
    }
    int *count = own_count, *count_next = nullptr, *mail = own_mail;
    if (!own_count) {
        count = ctx->d_sweep + (ctx->sweep_seq % kSweepRing);
        count_next = ctx->d_sweep + ((ctx->sweep_seq + 1) % kSweepRing);
        mail = ctx->sweep_deferred ? ctx->h_sweep + (ctx->sweep_seq % kSweepRing) : nullptr;
        ctx->sweep_mail[ctx->sweep_seq % kSweepRing] = mail ? ctx->sweep_seq + 1 : 0;
        ctx->sweep_seq++;
    }
    // a sweep that fails from here on has not run its last launch: the next sweep's counter is cleared by hand
    struct RingGuard {
        hipStream_t st; int *next; bool ok;
        ~RingGuard() { if (!ok && next) (void)hipMemsetAsync(next, 0, sizeof(int), st); }
    } ring{ctx->stream, count_next, false};
    double *d_w = (double *)(base + 2 * b_m + b_a + b_list + 256);
    double *C = (double *)(base + 2 * b_m + b_a + b_list + 256 + b_w);
    if (!own_count) ctx->sweep_count = count;
    double e_slide = 0.0;
    // One chain with tone templates: its mark - gain * space difference leaves the sliding kernel as ONE stream and takes ONE
    // low-pass (the reference's own dataflow, afsk.py:162-166, on approximate magnitudes); a sweep takes two for all its chains.
    const bool one = groups == 1 && tones && m >= 2;
    const double *lp_in = one ? M : S, *lp_a = one ? nullptr : A;
    const int frun = ctx->tune.fuse_run == 16 ? 16 : kFuseRun;      // PM_FUSE_RUN=16: round 1's run length, for comparison
    const bool fused = tones && m >= 2 && (kThreads * 8 + ml - 1 + frun - 1) / frun <= kThreads &&
                       fuse_lds_bytes(m, ml, frun) <= 120 * 1024 && !ctx->tune.afsk_unfused;
    if (fused) {
        // sliding sums, low-pass(es) and combine in one kernel (afsk_slide_lpf_kernel): nothing but the bitmaps is written
        e_slide = slide_bound(tones, m, x_bound, frun);
    } else {
        if (int rc = afsk_magnitudes(ctx, d_x, n, x_bound, d_mark_i, d_mark_q, d_unit_i, d_unit_q, m, tones, M, one ? nullptr : S, &e_slide,
                                     P.gain[0]))
            return rc;
        if (!one)
            if (int rc = fir_launch<double>(ctx, M, nc, d_lpf, ml, A, nullptr, 0)) return rc;
    }
    if (src) {
        // magnitudes are 1-Lipschitz in the pair of correlator sums, each of which moves by at most m e_x; the sliding sums' own
        // bound is stated for inputs up to x_bound, which the approximate stream exceeds by at most e_x
        e_slide = e_slide * (1.0 + src->e_x / x_bound) + 1.4143 * m * src->e_x;
    }
    double E = 1e-10 * lpf_abs_sum * (1.0 + gmax) * (double)m * 1.4143 * x_bound + lpf_abs_sum * (1.0 + gmax) * e_slide;
    // Low-passes on the int8 matrix pipe (afsk_slide_lpf8_kernel): a per-call plan for tests and measurements (PM_AFSK_LPF8=1), the
    // pipeline's own otherwise
    pm_lpf8_plan *own8 = nullptr;
    if (fused && !lpf8 && frun == kFuseRun && ml + 15 <= 128) {
        if (ctx->tune.afsk_lpf8 == 1) {
            std::vector<double> hl((size_t)ml);
            PM_HIP(hipMemcpyAsync(hl.data(), d_lpf, hl.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            PM_HIP(hipStreamSynchronize(ctx->stream));
            if (int rc = pm_lpf8_plan_create(ctx, hl.data(), ml, &own8)) return rc;
            lpf8 = own8;
        }
    }
    struct Own8 { pm_ctx *c; pm_lpf8_plan *p; ~Own8() { if (p) { (void)hipStreamSynchronize(c->stream); pm_lpf8_plan_destroy(p); } } } own8_guard{ctx, own8};
    if (fused && lpf8 && frun == kFuseRun && lpf8->ml == ml && ml + 15 <= 128) {
        PM_ARG(tones->tap_dev >= 0.0 && tones->tap_dev < 1e-6);
        SlideTones T{tones->mark_rot[0], tones->mark_rot[1], tones->mark_end[0], tones->mark_end[1],
                     tones->space_rot[0], tones->space_rot[1], tones->space_end[0], tones->space_end[1]};
        // What the matrix pipe adds to E depends on the workgroup's own largest magnitude (its digits are scaled to it): the kernel
        // computes it from these constants (afsk_slide_lpf8_kernel, Ecmp)
        Lpf8Args Q;
        Q.S = lpf8->S;
        Q.c_tap = lpf8->tapq_int;
        Q.c_q = 0.5 * lpf8->qabs + lpf8->dlow;
        Q.gfac = one ? 1.0 : 1.0 + gmax;
        Q.qabs = lpf8->qabs;
        Q.btab = (const int4v *)lpf8->d_btab;
        const int64_t ntiles = pm_cdiv(nl, (int64_t)kThreads * 8);
        PM_ARG(ntiles < (1LL << 31));
        const int nmag = kThreads * 8 + ml - 1, nruns = (nmag + kFuseRun - 1) / kFuseRun, pspan = nruns * kFuseRun + m - 1;
        const size_t xdoubles = (size_t)pspan + pspan / kFuseRun + 2, pdoubles = (size_t)(one ? 1 : 2) * kL8Dig * kL8Plane / 8;
        const int region0 = (int)((std::max(xdoubles, pdoubles) + 1) / 2 * 2);
        const size_t lds = ((size_t)region0 + 4 * (size_t)m) * sizeof(double) + 2 * kL8Dig * 64 * 16 + (kTailCap + 4) * sizeof(unsigned) + 32;      // x window | planes, templates, band operands, the workgroup's list, its waves' maxima
        PmProf prof(ctx, PM_K_FIR_F64);
        const double nlp = one ? 1.0 : 2.0;
        prof.work((double)n * 8 + (double)groups * nl / 8,
                  (4.0 * m / kFuseRun + 18.0) * (double)nc + nlp * 2.0 * ml * (double)nl + 2.0 * groups * (double)nl);
        SweepTail TL;
        TL.space = d_space;
        TL.lpf = d_lpf;
        TL.src = src ? *src : SweepSource{nullptr, nullptr, 0, 0.0};
        TL.lds_ok = sweep_tail_doubles(m, ml, src ? src->mb : 0) <= (size_t)region0 && !ctx->tune.sweep_no_tail;
        auto go8 = [&](auto kernel) -> int {
            if (int rc = allow_lds(kernel, lds)) return rc;
            hipLaunchKernelGGL(kernel, dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream, d_x, n, d_mark_i, d_mark_q, d_unit_i, d_unit_q, m, T,
                               Q, ml, nl, groups, P, E, list, count, cap, region0, TL);
            return PM_OK;
        };
        if (int rc = one ? go8(afsk_slide_lpf8_kernel<true>) : go8(afsk_slide_lpf8_kernel<false>)) return rc;
        PM_HIP(hipGetLastError());
        if (own_list) { ring.ok = true; return PM_OK; }      // nothing follows on this stream: the caller mails the count
    } else if (fused) {
        PM_ARG(tones->tap_dev >= 0.0 && tones->tap_dev < 1e-6);
        SlideTones T{tones->mark_rot[0], tones->mark_rot[1], tones->mark_end[0], tones->mark_end[1],
                     tones->space_rot[0], tones->space_rot[1], tones->space_end[0], tones->space_end[1]};
        const int64_t ntiles = pm_cdiv(nl, (int64_t)kThreads * 8);
        PM_ARG(ntiles < (1LL << 31));
        const size_t lds = fuse_lds_bytes(m, ml, frun);
        const int region0 = (int)fuse_region0(m, ml, frun), image = (int)fuse_image(ml);
        PmProf prof(ctx, PM_K_FIR_F64);
        const double nlp = one ? 1.0 : 2.0;
        prof.work((double)n * 8 + (double)groups * nl / 8,
                  (4.0 * m / frun + 18.0) * (double)nc + nlp * 2.0 * ml * (double)nl + 2.0 * groups * (double)nl);
        auto go = [&](auto kernel) -> int {
            if (int rc = allow_lds(kernel, lds)) return rc;
            hipLaunchKernelGGL(kernel, dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream, d_x, n, d_mark_i, d_mark_q, d_unit_i, d_unit_q, m, T,
                               d_lpf, ml, nl, groups, P, E, list, count, cap, region0, image);
            return PM_OK;
        };
        int rc;
        switch (frun) {
        case 16: rc = one ? go(afsk_slide_lpf_kernel<true, 16>) : go(afsk_slide_lpf_kernel<false, 16>); break;
        default: rc = one ? go(afsk_slide_lpf_kernel<true, 12>) : go(afsk_slide_lpf_kernel<false, 12>); break;
        }
        if (rc) return rc;
        PM_HIP(hipGetLastError());
    } else {   // B = LPF(S) and the combine step in one pass: B never reaches memory
        constexpr int R = 8;
        const int64_t ntiles = pm_cdiv(nl, (int64_t)kThreads * R);
        PM_ARG(ntiles < (1LL << 31));
        const size_t lds = lds_bytes<R>(ml);
        PmProf prof(ctx, PM_K_FIR_F64);
        prof.work((double)nc * 8 + (one ? 0.0 : (double)nl * 8) + (double)groups * nl / 8, 2.0 * ml * (double)nl + 2.0 * groups * (double)nl);
        if ((((uintptr_t)lp_in) & 15) == 0) {
            if (int rc = allow_lds(fir_sweep_kernel<R, true>, lds)) return rc;
            hipLaunchKernelGGL((fir_sweep_kernel<R, true>), dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream, lp_in, nc, d_lpf, ml, lp_a, nl,
                               groups, P, E, list, count, cap);
        } else {
            if (int rc = allow_lds(fir_sweep_kernel<R, false>, lds)) return rc;
            hipLaunchKernelGGL((fir_sweep_kernel<R, false>), dim3((unsigned)ntiles), dim3(kThreads), lds, ctx->stream, lp_in, nc, d_lpf, ml, lp_a, nl,
                               groups, P, E, list, count, cap);
        }
        PM_HIP(hipGetLastError());
    }
    {
        PmProf prof(ctx, PM_K_SIGNS);
        if (src)
            hipLaunchKernelGGL(sweep_exact_kernel<true>, dim3(kExactGrid), dim3(64), (size_t)(4 * ml + 6 * m + 2 * src->mb - 3) * sizeof(double), ctx->stream, d_x, d_mark_i, d_mark_q,
                               d_space, m, d_lpf, ml, P, list, count, cap, count_next, mail, *src);
        else
            hipLaunchKernelGGL(sweep_exact_kernel<false>, dim3(kExactGrid), dim3(64), (size_t)ml * sizeof(double), ctx->stream, d_x, d_mark_i, d_mark_q, d_space, m,
                               d_lpf, ml, P, list, count, cap, ctx->sweep_deferred ? count_next : nullptr, mail, SweepSource{nullptr, nullptr, 0, 0.0});
    }
    PM_HIP(hipGetLastError());
    // Deferred fallback: the caller looks at the counter once the sweep has finished (pm_afsk_sweep_result) and runs the exact
    // chains itself in the (degenerate) overflow case; the three gated launches below -- which in the normal case only look at
    // the counter and leave, but cost the demod stream three dispatches per sweep -- are not enqueued.
    if (ctx->sweep_deferred) { ring.ok = true; return PM_OK; }
    // More uncertain samples than the list holds (degenerate input: silence, amplitudes far below the caller's bound): the exact
    // chain of every modem runs after all -- the same launches as pm_afsk_correlate_group + pm_fir_signs_f64_batch, each workgroup
    // of which first looks at the counter and leaves at once in the normal case.  No host round trip either way.
    hipLaunchKernelGGL(pack_group_taps_kernel, dim3((unsigned)pm_cdiv((int64_t)F * m, 256)), dim3(256), 0, ctx->stream, d_mark_i, d_mark_q,
                       d_space, m, F, d_w);
    if (int rc = afsk_group_dispatch(ctx, groups, d_x, n, d_w, m, C, stride, nc, count, cap)) return rc;
    {
        constexpr int R = 8;
        FirBatch FB;
        memset(&FB, 0, sizeof(FB));
        bool vec = true;
        for (int g = 0; g < groups; ++g) {
            FB.x[g] = C + (size_t)g * stride;
            FB.bits[g] = h_bits[g];
            FB.n[g] = nc;
            vec = vec && (((uintptr_t)FB.x[g]) & 15) == 0;
        }
        const int64_t ntiles = pm_cdiv(nl, (int64_t)kThreads * R);
        const size_t lds = lds_bytes<R>(ml);
        PmProf prof(ctx, PM_K_SIGNS);                  // gated fallback, see afsk_group_go
        if (vec) {
            if (int rc = allow_lds(fir_signs_batch_kernel<R, false, true>, lds)) return rc;
            hipLaunchKernelGGL((fir_signs_batch_kernel<R, false, true>), dim3((unsigned)std::min<int64_t>(ntiles, kGatedGrid), (unsigned)groups), dim3(kThreads), lds, ctx->stream, FB,
                               d_lpf, ml, count, cap, count_next);
        } else {
            if (int rc = allow_lds(fir_signs_batch_kernel<R, false, false>, lds)) return rc;
            hipLaunchKernelGGL((fir_signs_batch_kernel<R, false, false>), dim3((unsigned)std::min<int64_t>(ntiles, kGatedGrid), (unsigned)groups), dim3(kThreads), lds, ctx->stream, FB,
                               d_lpf, ml, count, cap, count_next);
        }
    }
    PM_HIP(hipGetLastError());
    ring.ok = true;
    return PM_OK;
}

int pm_afsk_sweep_signs(pm_ctx *ctx, const double *d_x, int64_t n, double x_bound, const double *d_mark_i, const double *d_mark_q,
                        const double *d_unit_i, const double *d_unit_q, const double *d_space, const double *h_gains, int groups, int m,
                        const double *d_lpf, int ml, double lpf_abs_sum, uint64_t *const *h_bits)
{
    return sweep_signs(ctx, d_x, n, x_bound, d_mark_i, d_mark_q, d_unit_i, d_unit_q, d_space, h_gains, groups, m, d_lpf, ml, lpf_abs_sum, h_bits,
                       nullptr);
}

int pm_afsk_sweep_signs_tones(pm_ctx *ctx, const double *d_x, int64_t n, double x_bound, const double *d_mark_i, const double *d_mark_q,
                              const double *d_unit_i, const double *d_unit_q, const double *d_space, const double *h_gains, int groups, int m,
                              const double *d_lpf, int ml, double lpf_abs_sum, uint64_t *const *h_bits, const pm_afsk_tones *h_tones)
{
    if (!h_tones) return pm_set_error(PM_ERR_ARG, "pm_afsk_sweep_signs_tones: no tones");
    return sweep_signs(ctx, d_x, n, x_bound, d_mark_i, d_mark_q, d_unit_i, d_unit_q, d_space, h_gains, groups, m, d_lpf, ml, lpf_abs_sum, h_bits,
                       h_tones);
}

int pm_afsk_magnitudes(pm_ctx *ctx, const double *d_x, int64_t n, double x_bound, const double *d_mark_i, const double *d_mark_q,
                       const double *d_space_i, const double *d_space_q, int m, const pm_afsk_tones *h_tones, double *d_mark_mag,
                       double *d_space_mag, double *h_bound)
{
    PM_CTX(ctx);
    PM_ARG(d_x && d_mark_i && d_mark_q && d_space_i && d_space_q && d_mark_mag && d_space_mag);
    PM_ARG(m >= 1 && m <= kMaxTaps && n >= m && x_bound > 0.0 && x_bound < 1e300);
    double e = 0.0;
    if (int rc = afsk_magnitudes(ctx, d_x, n, x_bound, d_mark_i, d_mark_q, d_space_i, d_space_q, m, h_tones, d_mark_mag, d_space_mag, &e)) return rc;
    if (h_bound) *h_bound = e;
    return PM_OK;
}

int pm_afsk_sweep_last(pm_ctx *ctx, int64_t *h_uncertain)
{
    PM_CTX(ctx);
    PM_ARG(h_uncertain != nullptr);
    *h_uncertain = -1;
    if (!ctx->sweep_count) return PM_OK;
    int v = 0;
    PM_HIP(hipMemcpyAsync(&v, ctx->sweep_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    *h_uncertain = v;
    return PM_OK;
}

int pm_afsk_group_run(pm_ctx *ctx, const int16_t *d_audio, int64_t n, const double *d_bpf, int mb, double *d_bpf_out, double x_bound,
                      const pm_afsk_sweep_desc *h_sweeps, int nsweeps, int64_t *h_tickets)
{
    return pm_afsk_group_run_plan(ctx, d_audio, n, d_bpf, mb, d_bpf_out, x_bound, h_sweeps, nsweeps, h_tickets, nullptr, nullptr);
}

}  // extern "C"

// The fused launch of pm_afsk_group_run_plan (afsk_fused8_kernel): what each sweep's certified decision needs is worked out as
// sweep_signs works it out, sweep by sweep.  -> PM_OK and *fused = true when the launch was made; *fused = false (and nothing enqueued) when
// the group does not qualify and the caller takes the split path.
static int afsk_group_run_fused(pm_ctx *ctx, const int16_t *d_audio, int64_t n, const double *d_bpf, int mb, double x_bound, const pm_afsk_sweep_desc *h_sweeps,
                                int nsweeps, const pm_bpf8_plan *plan, const pm_lpf8_plan *const *lpf8, const pm_sweep_cells *cells, bool *fused)
{
    *fused = false;
    if (!plan || !lpf8 || !cells || !cells->d_list || nsweeps < 1 || nsweeps > 2 || ctx->tune.afsk_split || ctx->tune.afsk_unfused || ctx->tune.fuse_run == 16 ||
        ((uintptr_t)d_audio & 15) != 0)
        return PM_OK;
    int kb = 0;
    const void *btab = nullptr;
    double sc6[6];
    if (pm_bpf8_plan_view(plan, &kb, &btab, sc6) != PM_OK || (kb != 3 && kb != 4)) return PM_OK;
    FusedArgs A;
    memset(&A, 0, sizeof(A));
    const int64_t nb = n - mb + 1;
    const double e_x = pm_bpf8_error(plan);
    if (!(e_x >= 0.0 && e_x < 1e-6 * x_bound)) return PM_OK;
    int xs_span = 0, mmax = 0, tail_doubles = 0;
    bool two_streams = false;
    for (int k = 0; k < nsweeps; ++k) {
        const pm_afsk_sweep_desc &w = h_sweeps[k];
        const pm_lpf8_plan *q = lpf8[k];
        if (!q || !w.h_tones || w.m < 2 || w.ml + 15 > 128 || q->ml != w.ml || w.groups < 1 || w.groups > kSweepMax || !w.h_bits || !w.h_gains) return PM_OK;
        if (!(w.h_tones->tap_dev >= 0.0 && w.h_tones->tap_dev < 1e-6) || !(w.lpf_abs_sum > 0.0 && w.lpf_abs_sum < 1e300)) return PM_OK;
        if (nb < (int64_t)w.m + w.ml - 1) return PM_OK;
        FusedSweep &S = A.s[k];
        S.mi = w.d_mark_i; S.mq = w.d_mark_q; S.ui = w.d_unit_i; S.uq = w.d_unit_q;
        S.space = w.d_space; S.lpf = w.d_lpf;
        {
            // the four templates as ONE reversed, interleaved table (pack_group_taps_kernel's layout with F = 4), made the first time the
            // plan meets these templates and kept with it: the kernel reads it through the scalar cache
            pm_lpf8_plan *mq_ = const_cast<pm_lpf8_plan *>(q);
            if (!(mq_->d_tpl && mq_->tpl_m == w.m && mq_->tpl_src[0] == w.d_mark_i && mq_->tpl_src[1] == w.d_mark_q && mq_->tpl_src[2] == w.d_unit_i &&
                  mq_->tpl_src[3] == w.d_unit_q)) {
                if (mq_->d_tpl) { PM_HIP(hipStreamSynchronize(ctx->stream)); (void)hipFree(mq_->d_tpl); mq_->d_tpl = nullptr; }
                PM_HIP(hipMalloc(&mq_->d_tpl, sizeof(double) * 4 * (size_t)w.m + 256));
                hipLaunchKernelGGL(pack_templates_kernel, dim3((unsigned)pm_cdiv(4 * (int64_t)w.m, 256)), dim3(256), 0, ctx->stream, w.d_mark_i, w.d_mark_q, w.d_unit_i,
                                   w.d_unit_q, w.m, (double *)mq_->d_tpl);
                PM_HIP(hipGetLastError());
                PM_HIP(hipStreamSynchronize(ctx->stream));     // once per plan: another context's launch may be the table's next reader
                mq_->tpl_m = w.m;
                mq_->tpl_src[0] = w.d_mark_i; mq_->tpl_src[1] = w.d_mark_q; mq_->tpl_src[2] = w.d_unit_i; mq_->tpl_src[3] = w.d_unit_q;
            }
            S.tg = ctx->tune.sweep_lds_templates ? nullptr : (const double *)mq_->d_tpl;
        }
        S.m = w.m; S.ml = w.ml; S.G = w.groups;
        S.one = w.groups == 1;
        double gmax = 0.0;
        for (int g = 0; g < w.groups; ++g) {
            if (!w.h_bits[g] || !(w.h_gains[g] >= 0.0 && w.h_gains[g] < 1e100)) return PM_OK;
            S.P.gain[g] = w.h_gains[g];
            S.P.bits[g] = w.h_bits[g];
            gmax = std::max(gmax, w.h_gains[g]);
        }
        const pm_afsk_tones *tn = w.h_tones;
        S.T = SlideTones{tn->mark_rot[0], tn->mark_rot[1], tn->mark_end[0], tn->mark_end[1], tn->space_rot[0], tn->space_rot[1], tn->space_end[0], tn->space_end[1]};
        // E: sweep_signs' bound, with the band-passed stream a value within e_x of the reference's (see there)
        double e_slide = slide_bound(tn, w.m, x_bound, kFuseRun);
        e_slide = e_slide * (1.0 + e_x / x_bound) + 1.4143 * w.m * e_x;
        S.E = 1e-10 * w.lpf_abs_sum * (1.0 + gmax) * (double)w.m * 1.4143 * x_bound + w.lpf_abs_sum * (1.0 + gmax) * e_slide;
        S.Q.S = q->S;
        S.Q.c_tap = q->tapq_int;
        S.Q.c_q = 0.5 * q->qabs + q->dlow;
        S.Q.gfac = S.one ? 1.0 : 1.0 + gmax;
        S.Q.qabs = q->qabs;
        S.Q.btab = (const int4v *)q->d_btab;
        S.list = cells->d_list + (size_t)k * kSweepCap;
        S.count = cells->d_count + k;
        S.nout = nb - w.m - w.ml + 2;
        const int nmag = kThreads * 8 + w.ml - 1, nruns = (nmag + kFuseRun - 1) / kFuseRun;
        if (nruns > kThreads) return PM_OK;
        xs_span = std::max(xs_span, nruns * kFuseRun + w.m - 1);
        mmax = std::max(mmax, w.m);
        tail_doubles = std::max(tail_doubles, (int)sweep_tail_doubles(w.m, w.ml, mb));
        two_streams = two_streams || !S.one;
    }
    A.audio = d_audio;
    A.n = n;
    A.nb = nb;
    A.bp_btab = (const pm_bpf8_dev::i4 *)btab;
    for (int w = 0; w < 5; ++w) A.sc.s[w] = sc6[w];
    A.sc.c0 = sc6[5];
    A.src = SweepSource{d_audio, d_bpf, mb, e_x};
    A.xs_span = xs_span;
    const int btiles = (xs_span + 255) / 256;
    A.aplane = (btiles * 256 + 64 * kb + 15) / 16 * 16;      // what the last tile's band reads: 256 (tile) + 64 kb - 16 bytes past its first
    A.xw_doubles = (xs_span + xs_span / kFuseRun + 2 + 1) / 2 * 2;
    A.plane_bytes = std::max((two_streams ? 2 : 1) * kL8Dig * kL8Plane, 2 * A.aplane);
    A.plane_bytes = (A.plane_bytes + 15) / 16 * 16;
    A.mmax = mmax;
    A.cap = kSweepCap;
    A.lds_ok = (size_t)tail_doubles * 8 <= (size_t)A.xw_doubles * 8 + (size_t)A.plane_bytes && !ctx->tune.sweep_no_tail;
    const size_t lds = (size_t)A.xw_doubles * 8 + (size_t)A.plane_bytes + 4 * (size_t)mmax * 8 + 2 * kL8Dig * 64 * 16 + (kTailCap + 4) * sizeof(unsigned) + 32;
    if (lds > 64 * 1024) return PM_OK;
    const size_t lds_launch = lds + (size_t)std::max(0, ctx->tune.fused_lds_pad);      // (PM_FUSED_LDS_PAD: fewer workgroups per CU, to measure what occupancy is worth)
    int64_t tiles = 0;
    double bits_out = 0.0, flops = 2.0 * mb * (double)nb;
    for (int k = 0; k < nsweeps; ++k) {
        tiles = std::max(tiles, pm_cdiv(A.s[k].nout, (int64_t)kThreads * 8));
        bits_out += (double)A.s[k].G * (double)A.s[k].nout / 8;
        flops += (4.0 * A.s[k].m / kFuseRun + 18.0) * (double)(nb - A.s[k].m + 1) + (A.s[k].one ? 1.0 : 2.0) * 2.0 * A.s[k].ml * (double)A.s[k].nout +
                 2.0 * A.s[k].G * (double)A.s[k].nout;
    }
    PM_ARG(tiles >= 1 && tiles < (1LL << 31));
    {
        PmProf prof(ctx, PM_K_FIR_F64);
        prof.work((double)n * 2 + bits_out, flops);          // the recording in, the bitmaps out: nothing else crosses HBM
        auto go = [&](auto kernel) -> int {
            if (int rc = allow_lds(kernel, lds_launch)) return rc;
            hipLaunchKernelGGL(kernel, dim3((unsigned)tiles), dim3(kThreads), lds_launch, ctx->stream, A);
            return PM_OK;
        };
        int rc = PM_OK;
        const bool o0 = A.s[0].one != 0, o1 = nsweeps == 2 && A.s[1].one != 0;
#define PM_FUSED_GO(KB)                                                                                                             \
        rc = nsweeps == 1 ? (o0 ? go(afsk_fused8_kernel<KB, true, 1, false>) : go(afsk_fused8_kernel<KB, false, 1, false>))              \
                          : (o0 ? (o1 ? go(afsk_fused8_kernel<KB, true, 2, true>) : go(afsk_fused8_kernel<KB, true, 2, false>))          \
                                : (o1 ? go(afsk_fused8_kernel<KB, false, 2, true>) : go(afsk_fused8_kernel<KB, false, 2, false>)));
        if (kb == 3) { PM_FUSED_GO(3) } else { PM_FUSED_GO(4) }
#undef PM_FUSED_GO
        if (rc) return rc;
        PM_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(sweep_mail_reset_kernel, dim3(1), dim3(64), 0, ctx->stream, cells->d_count, cells->d_count + nsweeps, cells->h_mail, nsweeps);
    PM_HIP(hipGetLastError());
    *fused = true;
    return PM_OK;
}

int pm_afsk_group_run_plan(pm_ctx *ctx, const int16_t *d_audio, int64_t n, const double *d_bpf, int mb, double *d_bpf_out, double x_bound,
                           const pm_afsk_sweep_desc *h_sweeps, int nsweeps, int64_t *h_tickets, const pm_bpf8_plan *plan,
                           const pm_lpf8_plan *const *lpf8, const pm_sweep_cells *cells)
{
    // The demod stage of a whole AFSK chain group in ONE call: the shared band-pass (afsk.py:151) and every certified sweep on its
    // output (afsk.py:153-166, sign bitmaps only), overflow fallback deferred to the caller (pm_afsk_sweep_results).  The same
    // launches pm_fir_valid_i16 + pm_afsk_sweep_signs[_tones] would make -- a pipelined Python host saves nine boundary crossings
    // per recording, each of which waits for the interpreter lock on the way back.
    PM_CTX(ctx);
    PM_ARG(d_audio && d_bpf && d_bpf_out && h_sweeps && nsweeps >= 1 && nsweeps <= 64 && mb >= 1 && n >= mb);
    static const bool trace = getenv("PM_TRACE_CALLS") != nullptr;
    static double acc_us[3] = {0, 0, 0};
    static long calls = 0;
    const auto t_0 = std::chrono::steady_clock::now();
    SweepSource src{d_audio, d_bpf, mb, plan ? pm_bpf8_error(plan) : 0.0};
    PM_ARG(!cells || (cells->d_count && cells->h_mail));
    if (plan && cells && cells->d_list) {
        // one launch for the whole stage where the group qualifies (all sweeps on the matrix pipe, at most two): afsk_fused8_kernel
        bool fused = false;
        for (int k = 0; k < nsweeps; ++k) PM_ARG(h_sweeps[k].h_tones != nullptr);
        PM_ARG(pm_bpf8_taps(plan) == mb);
        if (int rc = afsk_group_run_fused(ctx, d_audio, n, d_bpf, mb, x_bound, h_sweeps, nsweeps, plan, lpf8, cells, &fused)) return rc;
        if (fused) return PM_OK;
    }
    if (plan) {
        PM_ARG(pm_bpf8_taps(plan) == mb);
        for (int k = 0; k < nsweeps; ++k) PM_ARG(h_sweeps[k].h_tones != nullptr);
        if (int rc = pm_bpf8_run(ctx, plan, d_audio, n, d_bpf_out, cells ? cells->d_count : nullptr, cells ? nsweeps : 0)) return rc;
    } else {
        if (cells) PM_HIP(hipMemsetAsync(cells->d_count, 0, sizeof(int) * (size_t)nsweeps, ctx->stream));
        if (int rc = fir_launch<int16_t>(ctx, d_audio, n, d_bpf, mb, d_bpf_out, nullptr, 0)) return rc;
    }
    const auto t_1 = std::chrono::steady_clock::now();
    const bool was = ctx->sweep_deferred;
    ctx->sweep_deferred = true;
    int rc = PM_OK;
    for (int k = 0; k < nsweeps && rc == PM_OK; ++k) {
        const pm_afsk_sweep_desc &w = h_sweeps[k];
        rc = sweep_signs(ctx, d_bpf_out, n - mb + 1, x_bound, w.d_mark_i, w.d_mark_q, w.d_unit_i, w.d_unit_q, w.d_space, w.h_gains, w.groups, w.m,
                         w.d_lpf, w.ml, w.lpf_abs_sum, w.h_bits, w.h_tones, plan ? &src : nullptr, lpf8 ? lpf8[k] : nullptr,
                         cells ? cells->d_count + k : nullptr, cells ? cells->h_mail + k : nullptr,
                         cells && cells->d_list ? cells->d_list + (size_t)k * kSweepCap : nullptr);
        if (h_tickets && !cells) h_tickets[k] = ctx->sweep_seq - 1;
    }
    ctx->sweep_deferred = was;
    if (rc == PM_OK && cells && cells->d_list) {
        // One wave behind the recording's last sweep leaves every sweep's count in its page-locked word (a sweep that ended with a launch
        // of its own has written the same value there already).  Round 4 had a 4096-workgroup launch between and behind the sweeps for
        // this and the recomputation: 0.23 ms per recording of the demod streams' time.
        hipLaunchKernelGGL(sweep_mail_reset_kernel, dim3(1), dim3(64), 0, ctx->stream, cells->d_count, cells->d_count + nsweeps, cells->h_mail, nsweeps);
        if (hipGetLastError() != hipSuccess) rc = pm_set_error(PM_ERR_HIP, "pm_afsk_group_run: mailing the sweeps' counts failed");
    }
    if (trace) {
        const auto t_2 = std::chrono::steady_clock::now();
        acc_us[0] += std::chrono::duration<double, std::micro>(t_1 - t_0).count();
        acc_us[1] += std::chrono::duration<double, std::micro>(t_2 - t_1).count();
        if (++calls % 100 == 0) {
            fprintf(stderr, "[pm_afsk_group_run] avg over 100 calls: band-pass launch %.1f us, %d sweeps %.1f us\n", acc_us[0] / 100, nsweeps, acc_us[1] / 100);
            acc_us[0] = acc_us[1] = 0;
        }
    }
    return rc;
}

int pm_afsk_sweep_exact_list(pm_ctx *ctx, const int16_t *d_audio, const double *d_bpf, int mb, const pm_afsk_sweep_desc *w, uint64_t *const *h_bits,
                             const unsigned long long *d_list, const int *d_count)
{
    PM_CTX(ctx);
    PM_ARG(d_audio && d_bpf && mb >= 1 && w && h_bits && d_list && d_count && w->groups >= 1 && w->groups <= kSweepMax);
    SweepArgs P;
    memset(&P, 0, sizeof(P));
    for (int g = 0; g < w->groups; ++g) {
        PM_ARG(h_bits[g] != nullptr);
        P.gain[g] = w->h_gains[g];
        P.bits[g] = h_bits[g];
    }
    PmProf prof(ctx, PM_K_SIGNS);
    hipLaunchKernelGGL(sweep_exact_kernel<true>, dim3(kExactGrid), dim3(64), (size_t)(4 * w->ml + 6 * w->m + 2 * mb - 3) * sizeof(double), ctx->stream, (const double *)nullptr,
                       w->d_mark_i, w->d_mark_q, w->d_space, w->m, w->d_lpf, w->ml, P, d_list, d_count, kSweepCap, (int *)nullptr, (int *)nullptr,
                       SweepSource{d_audio, d_bpf, mb, 0.0});
    PM_HIP(hipGetLastError());
    return PM_OK;
}

extern "C" {

int pm_afsk_sweep_mode(pm_ctx *ctx, int deferred)
{
    PM_ARG(ctx != nullptr);
    ctx->sweep_deferred = deferred != 0;
    return PM_OK;
}

int pm_afsk_sweep_ticket(pm_ctx *ctx, int64_t *h_ticket)
{
    PM_ARG(ctx != nullptr && h_ticket != nullptr);
    *h_ticket = ctx->sweep_seq - 1;                    // -1: no sweep yet
    return PM_OK;
}

int pm_afsk_sweep_results(pm_ctx *ctx, const int64_t *tickets, int n, pm_ctx *via, int64_t *h_uncertain, int64_t *h_capacity)
{
    PM_CTX(ctx);
    PM_ARG(h_uncertain != nullptr && tickets != nullptr && n >= 1 && ctx->d_sweep != nullptr);
    for (int k = 0; k < n; ++k) {
        PM_ARG(tickets[k] >= 0 && tickets[k] < ctx->sweep_seq);
        if (ctx->sweep_seq - tickets[k] >= kSweepRing)
            return pm_set_error(PM_ERR_ARG, "pm_afsk_sweep_results: ticket %lld is %lld sweeps old, the ring holds %d", (long long)tickets[k],
                                (long long)(ctx->sweep_seq - tickets[k]), kSweepRing);
    }
    // deferred sweeps have left their counters in the page-locked mailbox (sweep_exact_kernel): the caller knows they have finished
    bool mailed = ctx->h_sweep != nullptr;
    for (int k = 0; k < n && mailed; ++k) mailed = ctx->sweep_mail[tickets[k] % kSweepRing] == tickets[k] + 1;
    if (mailed) {
        for (int k = 0; k < n; ++k) h_uncertain[k] = ((volatile int *)ctx->h_sweep)[tickets[k] % kSweepRing];
        if (h_capacity) *h_capacity = kSweepCap;
        return PM_OK;
    }
    // the caller knows the sweeps have finished; the whole ring (256 bytes) comes over in ONE copy on `via`'s stream (the caller's own:
    // a slicer worker must not queue behind the demod stream's next recordings, nor take the device-wide wait of a synchronous copy)
    pm_ctx *c = via ? via : ctx;
    int *h = (int *)c->h_pinned + 16;                      // past the words other entry points use as flags
    PM_HIP(hipMemcpyAsync(h, ctx->d_sweep, kSweepRing * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PM_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < n; ++k) h_uncertain[k] = h[tickets[k] % kSweepRing];
    if (h_capacity) *h_capacity = kSweepCap;
    return PM_OK;
}

int pm_afsk_sweep_result(pm_ctx *ctx, int64_t ticket, pm_ctx *via, int64_t *h_uncertain, int64_t *h_capacity)
{
    return pm_afsk_sweep_results(ctx, &ticket, 1, via, h_uncertain, h_capacity);
}

int pm_signs_f64(pm_ctx *ctx, const double *d_x, int64_t n, uint64_t *d_bits)
{
    PM_CTX(ctx);
    PM_ARG(ctx && d_bits && n >= 0);
    if (n == 0) return PM_OK;
    PM_ARG(d_x != nullptr);
    const int64_t nwords = pm_cdiv(n, 64);
    int64_t grid = pm_cdiv(nwords, kThreads / 64);
    if (grid > 8192) grid = 8192;
    PmProf prof(ctx, PM_K_SIGNS);
    prof.work((double)n * 8 + (double)n / 8, 0.0);
    hipLaunchKernelGGL(signs_kernel, dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, d_x, n, d_bits, nwords);
    PM_HIP(hipGetLastError());
    return PM_OK;
}

}  // extern "C"

// ---- what slide_run_f32 relies on: v_sqrt_f32 within one unit in the last place, for EVERY significand ---------------------------------
// All 2^23 significands of the binades 2^e and 2^(e+1) (a root's significand depends on the radicand's significand and on the parity of
// its exponent only): the largest |v_sqrt_f32(x) - sqrt(x)| in units of the result's last place, sqrt(x) in binary64 (correctly rounded,
// 29 bits to spare).  *h_worst_ulp_1024 = that, times 1024, rounded up.
namespace {
__global__ __launch_bounds__(256) void sqrt_f32_ulp_kernel(int e, unsigned long long *worst)
{
    const unsigned k = blockIdx.x * 256u + threadIdx.x;       // 2^24 threads: significand k & (2^23 - 1), exponent e + (k >> 23)
    const unsigned bits = ((unsigned)(e + (int)(k >> 23) + 127) << 23) | (k & 0x7FFFFFu);
    const float x = __uint_as_float(bits);
    const float r = __builtin_amdgcn_sqrtf(x);
    const double exact = __builtin_sqrt((double)x);
    int re = 0;
    (void)frexp((double)r, &re);                              // r = f 2^re, f in [0.5, 1): its last place is 2^(re - 24)
    const double ulps = fabs((double)r - exact) * ldexp(1.0, 24 - re);
    unsigned long long mine = (unsigned long long)ceil(ulps * 1024.0);
    if (!(ulps == ulps)) mine = ~0ull;
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(mine, off);
        mine = o > mine ? o : mine;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(worst, mine);
}
}  // namespace

// int8 digit products per tap and output of the matrix-pipe kernels, from the kernels' own constants (bench.py prices the launches with them)
extern "C" int pm_matrix_digit_pairs(int stage)
{
    switch (stage) {
    case 0: return pm_bpf8_digit_pairs();                    // bpf8_kernel: the certified sweeps' band-pass
    case 1: return kL8Dig * kL8Dig - 1;                      // afsk_slide_lpf8_kernel, per low-pass stream: all pairs but x0 q0
    case 2: return pm_fir8_digit_pairs();                    // fir8_kernel: the batch engine's matched filters
    default: return pm_set_error(PM_ERR_ARG, "pm_matrix_digit_pairs: no stage %d", stage);
    }
}

extern "C" int pm_ubench_sqrt_f32(pm_ctx *ctx, int exponent, int64_t *h_worst_ulp_1024)
{
    PM_CTX(ctx);
    PM_ARG(h_worst_ulp_1024 != nullptr && exponent >= -125 && exponent <= 125);
    void *q = nullptr;
    if (int rc = pm_malloc(ctx, 8, &q)) return rc;
    int rc = PM_OK;
    unsigned long long worst = 0;
    if (hipMemsetAsync(q, 0, 8, ctx->stream) != hipSuccess) rc = pm_set_error(PM_ERR_HIP, "pm_ubench_sqrt_f32: memset failed");
    if (!rc) {
        hipLaunchKernelGGL(sqrt_f32_ulp_kernel, dim3(1u << 16), dim3(256), 0, ctx->stream, exponent, (unsigned long long *)q);
        if (hipMemcpyAsync(&worst, q, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = pm_set_error(PM_ERR_HIP, "pm_ubench_sqrt_f32: the launch failed");
    }
    (void)pm_free(ctx, q);
    *h_worst_ulp_1024 = (int64_t)worst;
    return rc;
}
