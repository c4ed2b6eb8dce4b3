// Symbol-timing slicers (BinarySlicer.slice slicer.py:59-107, QuadratureSlicer.slice slicer.py:193-242)
// evaluated chunk-parallel on the sign bitmap of the demodulated stream.
//
// The reference recurrence, per sample k (1-based address k+1):
//     clk += 1.0;  if (clk >= sps/2 - 0.5) { clk -= sps;  take a symbol from sign(x[k]) }
//     if sign(x[k]) != sign(x[k-1]):  clk *= lock_rate
// is sequential in `clk` only.  The stream is cut into chunks of L samples, one lane per chunk.
// Iteration r runs every chunk from the start state handed to it and hands its end state to the next
// chunk; chunk 0 always starts from the true state.  A chunk whose start state did not change is not
// re-run.  When an iteration changes no start state, every chunk started from the end state of its
// predecessor, so by induction from chunk 0 the per-chunk runs ARE the sequential run, bit for bit
// (each lane executes the reference's operations in the reference's order; only the starting value is
// guessed).  Two trajectories that see the same zero crossings contract by lock_rate per crossing, so a
// few iterations suffice on real signals; the worst case (no crossings at all) degrades to nchunks
// iterations, i.e. sequential cost, never to a wrong answer.
//
// After the fixed point: an exclusive scan of per-chunk symbol counts gives every chunk its global
// symbol index (hence byte index and bit phase), and an emit pass re-runs the chunks writing bytes
// (atomicOr of bit fields into a zeroed buffer) and the address of each byte's last symbol.
#include "pm_common.h"
#include <algorithm>
#include <cstring>

namespace {

struct SlicerDev {
    double sps, thr, lock;
    int bps, mask;
    int demap[16];
};

__device__ __forceinline__ uint64_t dbits(double v) { return (uint64_t)__double_as_longlong(v); }
__device__ __forceinline__ double bitsd(uint64_t v) { return __longlong_as_double((long long)v); }

template <bool QUAD>
__global__ __launch_bounds__(64) void slice_iter_kernel(const uint64_t *__restrict__ bi, const uint64_t *__restrict__ bq,
                                                        int64_t n, int lc_words, int64_t nchunks,
                                                        const uint64_t *__restrict__ s_in, uint64_t *__restrict__ s_out,
                                                        const uint8_t *__restrict__ d_in, uint8_t *__restrict__ d_out,
                                                        uint32_t *__restrict__ count, uint8_t *__restrict__ lastsym,
                                                        int *__restrict__ changed, int iter, SlicerDev P)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    // chunk 0 starts from the true state: it runs once.  Nobody writes d_in[0], so it is not consulted.
    const bool dirty = c == 0 ? (iter == 0) : (d_in[c] != 0);
    if (!dirty) {                        // start state unchanged: end state and counts stand
        s_out[c + 1] = s_in[c + 1];
        d_out[c + 1] = 0;
        return;
    }
    double clk = bitsd(s_in[c]);
    const int64_t nwords = (n + 63) >> 6;
    const int64_t w0 = c * lc_words;
    const int64_t w1 = min(w0 + (int64_t)lc_words, nwords);
    // last_sample starts at 0.0, i.e. ">= 0" (slicer.py:55,164-165)
    uint64_t li = w0 == 0 ? 1ull : (bi[w0 - 1] >> 63);
    uint64_t lq = 1ull;
    if (QUAD) lq = w0 == 0 ? 1ull : (bq[w0 - 1] >> 63);
    uint32_t cnt = 0;
    uint32_t ls = 0xFF;
    for (int64_t w = w0; w < w1; ++w) {
        const uint64_t si = bi[w];
        uint64_t zc = si ^ ((si << 1) | li);
        li = si >> 63;
        uint64_t sq = 0;
        if (QUAD) {
            sq = bq[w];
            zc |= sq ^ ((sq << 1) | lq);
            lq = sq >> 63;
        }
        const int64_t left = n - (w << 6);
        const int nb = left < 64 ? (int)left : 64;
        for (int b = 0; b < nb; ++b) {
            clk += 1.0;
            if (clk >= P.thr) {
                clk -= P.sps;
                cnt++;
                if (QUAD) ls = (uint32_t)((((si >> b) & 1) << 1) | ((sq >> b) & 1));
            }
            if ((zc >> b) & 1) clk = clk * P.lock;
        }
    }
    const uint64_t e = dbits(clk);
    const bool ch = e != s_in[c + 1];
    s_out[c + 1] = e;
    d_out[c + 1] = ch ? 1 : 0;
    count[c] = cnt;
    if (QUAD) lastsym[c] = (uint8_t)ls;
    if (ch && c + 1 < nchunks) atomicOr(changed, 1);
}

// Exclusive scan of symbol counts + "last symbol before this chunk" carry.  One workgroup.
__global__ __launch_bounds__(1024) void slice_scan_kernel(const uint32_t *__restrict__ count, const uint8_t *__restrict__ lastsym,
                                                          int64_t nchunks, uint64_t *__restrict__ offset,
                                                          uint8_t *__restrict__ prevsym, int quad, int init_sym)
{
    __shared__ uint64_t sums[1024];
    __shared__ int lasts[1024];
    const int t = threadIdx.x;
    const int64_t per = (nchunks + 1023) / 1024;
    const int64_t c0 = min((int64_t)t * per, nchunks), c1 = min(c0 + per, nchunks);
    uint64_t s = 0;
    int l = -1;
    for (int64_t c = c0; c < c1; ++c) {
        s += count[c];
        if (quad && lastsym[c] != 0xFF) l = lastsym[c];
    }
    sums[t] = s;
    lasts[t] = l;
    __syncthreads();
    if (t == 0) {                       // 1024 partials: a serial pass is ~microseconds
        uint64_t run = 0;
        int carry = init_sym;
        for (int i = 0; i < 1024; ++i) {
            uint64_t v = sums[i];
            int lv = lasts[i];
            sums[i] = run;
            lasts[i] = carry;
            run += v;
            if (lv >= 0) carry = lv;
        }
        offset[nchunks] = run;
    }
    __syncthreads();
    uint64_t run = sums[t];
    int carry = lasts[t];
    for (int64_t c = c0; c < c1; ++c) {
        offset[c] = run;
        if (quad) prevsym[c] = (uint8_t)carry;
        run += count[c];
        if (quad && lastsym[c] != 0xFF) carry = lastsym[c];
    }
}

template <bool QUAD>
__global__ __launch_bounds__(64) void slice_emit_kernel(const uint64_t *__restrict__ bi, const uint64_t *__restrict__ bq,
                                                        int64_t n, int lc_words, int64_t nchunks,
                                                        const uint64_t *__restrict__ start, const uint64_t *__restrict__ offset,
                                                        const uint8_t *__restrict__ prevsym, uint32_t *__restrict__ data32,
                                                        int64_t *__restrict__ addr, int64_t cap, SlicerDev P)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    double clk = bitsd(start[c]);
    uint64_t g = offset[c];
    const uint64_t total = offset[nchunks];
    const int spb = 8 / P.bps;                       // symbols per byte
    const uint64_t nbytes = total / (uint64_t)spb;   // a trailing partial byte is never emitted (slicer.py:94-96)
    uint32_t prev = QUAD ? prevsym[c] : 0;
    const int64_t nwords = (n + 63) >> 6;
    const int64_t w0 = c * lc_words;
    const int64_t w1 = min(w0 + (int64_t)lc_words, nwords);
    uint64_t li = w0 == 0 ? 1ull : (bi[w0 - 1] >> 63);
    uint64_t lq = 1ull;
    if (QUAD) lq = w0 == 0 ? 1ull : (bq[w0 - 1] >> 63);
    uint32_t acc = 0;
    bool pending = false;
    for (int64_t w = w0; w < w1; ++w) {
        const uint64_t si = bi[w];
        uint64_t zc = si ^ ((si << 1) | li);
        li = si >> 63;
        uint64_t sq = 0;
        if (QUAD) {
            sq = bq[w];
            zc |= sq ^ ((sq << 1) | lq);
            lq = sq >> 63;
        }
        const int64_t left = n - (w << 6);
        const int nb = left < 64 ? (int)left : 64;
        for (int b = 0; b < nb; ++b) {
            clk += 1.0;
            if (clk >= P.thr) {
                clk -= P.sps;
                uint32_t v;
                if (QUAD) {
                    const uint32_t cur = (uint32_t)((((si >> b) & 1) << 1) | ((sq >> b) & 1));
                    const uint32_t sreg = ((prev << 2) | cur) & (uint32_t)P.mask;     // slicer.py:210-214
                    v = (uint32_t)P.demap[sreg];
                    prev = cur;
                } else {
                    v = (uint32_t)((si >> b) & 1);                                     // slicer.py:85-90
                }
                const int j = (int)(g % (uint64_t)spb);
                acc |= v << (8 - P.bps * (j + 1));            // MSB-first packing
                pending = true;
                if (j == spb - 1) {
                    const uint64_t idx = g / (uint64_t)spb;
                    if (idx < (uint64_t)cap) {
                        atomicOr(&data32[idx >> 2], (acc & 0xFF) << ((idx & 3) * 8));
                        addr[idx] = (w << 6) + b + 1;          // streamaddress, 1-based
                    }
                    acc = 0;
                    pending = false;
                }
                g++;
            }
            if ((zc >> b) & 1) clk = clk * P.lock;
        }
    }
    if (pending) {                     // head of a byte that a later chunk completes
        const uint64_t idx = g / (uint64_t)spb;
        if (idx < nbytes && idx < (uint64_t)cap) atomicOr(&data32[idx >> 2], (acc & 0xFF) << ((idx & 3) * 8));
    }
}

__global__ void slice_init_kernel(uint64_t *sa, uint64_t *sb, uint8_t *da, uint8_t *db, int64_t nchunks, uint64_t init_clk, int *changed)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) *changed = 0;
    if (c > nchunks) return;
    sa[c] = c == 0 ? init_clk : 0ull;      // cold start: phase_clock = 0.0
    sb[c] = c == 0 ? init_clk : 0ull;
    da[c] = c < nchunks ? 1 : 0;
    db[c] = 0;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <bool QUAD>
int slice_run(pm_ctx *ctx, const uint64_t *d_bi, const uint64_t *d_bq, int64_t n, const pm_slicer_params *hp,
              uint8_t *d_data, int64_t *d_addr, int64_t cap, int64_t *h_count)
{
    PM_ARG(ctx && hp && h_count && n >= 0 && cap >= 0);
    PM_ARG(hp->bits_per_symbol == 1 || hp->bits_per_symbol == 2);
    PM_ARG(hp->samples_per_symbol > 0.0 && hp->lock_rate == hp->lock_rate);
    *h_count = 0;
    ctx->sl_iterations = 0;
    if (n == 0) return PM_OK;
    PM_ARG(d_bi && (!QUAD || d_bq) && (cap == 0 || (d_data && d_addr)));
    PM_ARG(((uintptr_t)d_data & 3) == 0);

    SlicerDev P;
    P.sps = hp->samples_per_symbol;
    P.thr = (hp->samples_per_symbol / 2.0) - 0.5;      // slicer.py:52
    P.lock = hp->lock_rate;
    P.bps = hp->bits_per_symbol;
    P.mask = hp->state_mask;
    for (int i = 0; i < 16; ++i) P.demap[i] = hp->demap[i];

    // chunk length: multiple of 64 samples, ~8192 chunks on long streams, never shorter than 1024 samples
    const int64_t nwords = pm_cdiv(n, 64);
    int64_t lc_words = pm_cdiv(nwords, 8192);
    lc_words = std::max<int64_t>(16, std::min<int64_t>(lc_words, 128));
    const int64_t nchunks = pm_cdiv(nwords, lc_words);
    ctx->sl_chunk_len = (int32_t)(lc_words * 64);
    ctx->sl_chunks = nchunks;

    // carve the scratch
    const size_t e = (size_t)nchunks + 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_sa = carve(e * 8), o_sb = carve(e * 8), o_da = carve(e), o_db = carve(e), o_cnt = carve(e * 4),
                 o_ls = carve(e), o_off = carve(e * 8), o_ps = carve(e), o_ch = carve(256);
    if (int rc = pm_scratch_reserve(ctx, off)) return rc;
    char *base = (char *)ctx->d_scratch;
    uint64_t *sa = (uint64_t *)(base + o_sa), *sb = (uint64_t *)(base + o_sb);
    uint8_t *da = (uint8_t *)(base + o_da), *db = (uint8_t *)(base + o_db);
    uint32_t *cnt = (uint32_t *)(base + o_cnt);
    uint8_t *ls = (uint8_t *)(base + o_ls), *ps = (uint8_t *)(base + o_ps);
    uint64_t *offs = (uint64_t *)(base + o_off);
    int *changed = (int *)(base + o_ch);

    const unsigned grid = (unsigned)pm_cdiv(nchunks, 64);
    const double init_clk = 0.0;
    uint64_t init_bits;
    memcpy(&init_bits, &init_clk, 8);
    hipLaunchKernelGGL(slice_init_kernel, dim3((unsigned)pm_cdiv((int64_t)e, 256)), dim3(256), 0, ctx->stream,
                       sa, sb, da, db, nchunks, init_bits, changed);

    int *h_flag = (int *)ctx->h_pinned;
    int iters = 0;
    const int64_t max_iters = nchunks + 2;
    bool converged = false;
    while (!converged) {
        // a short burst of iterations between host checks keeps the launch queue full
        const int burst = iters < 2 ? 2 : 4;
        for (int b = 0; b < burst; ++b) {
            PmProf prof(ctx, PM_K_SLICE_ITER);
            hipLaunchKernelGGL((slice_iter_kernel<QUAD>), dim3(grid), dim3(64), 0, ctx->stream, d_bi, d_bq, n, (int)lc_words,
                               nchunks, sa, sb, da, db, cnt, ls, changed, iters, P);
            std::swap(sa, sb);
            std::swap(da, db);
            ++iters;
        }
        PM_HIP(hipMemcpyAsync(h_flag, changed, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(hipMemsetAsync(changed, 0, sizeof(int), ctx->stream));
        PM_HIP(hipStreamSynchronize(ctx->stream));
        // `changed` accumulates over the burst; a burst with no change at all means the LAST state is a fixed point
        converged = (*h_flag == 0);
        if (!converged && iters > max_iters + 8)
            return pm_set_error(PM_ERR_NOCONVERGE, "slicer fixed point not reached after %d iterations (%lld chunks)", iters, (long long)nchunks);
    }
    ctx->sl_iterations = iters;

    hipLaunchKernelGGL(slice_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt, ls, nchunks, offs, ps, QUAD ? 1 : 0, 0);
    if (cap > 0) {
        PM_HIP(hipMemsetAsync(d_data, 0, align_up((size_t)cap, 4), ctx->stream));
        PmProf prof(ctx, PM_K_SLICE_EMIT);
        hipLaunchKernelGGL((slice_emit_kernel<QUAD>), dim3(grid), dim3(64), 0, ctx->stream, d_bi, d_bq, n, (int)lc_words, nchunks,
                           sa, offs, ps, (uint32_t *)d_data, d_addr, cap, P);
    }
    uint64_t *h_total = (uint64_t *)ctx->h_pinned;
    PM_HIP(hipMemcpyAsync(h_total, offs + nchunks, 8, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    PM_HIP(hipGetLastError());
    const int64_t nbytes = (int64_t)(*h_total / (uint64_t)(8 / P.bps));
    *h_count = nbytes;
    if (nbytes > cap)
        return pm_set_error(PM_ERR_CAPACITY, "slicer produced %lld bytes, capacity %lld", (long long)nbytes, (long long)cap);
    return PM_OK;
}

}  // namespace

extern "C" {

int pm_slice_binary(pm_ctx *ctx, const uint64_t *d_bits, int64_t n, const pm_slicer_params *h_params,
                    uint8_t *d_data, int64_t *d_addr, int64_t cap, int64_t *h_count)
{
    PM_ARG(h_params && h_params->bits_per_symbol == 1);
    return slice_run<false>(ctx, d_bits, nullptr, n, h_params, d_data, d_addr, cap, h_count);
}

int pm_slice_quadrature(pm_ctx *ctx, const uint64_t *d_bits_i, const uint64_t *d_bits_q, int64_t n,
                        const pm_slicer_params *h_params, uint8_t *d_data, int64_t *d_addr, int64_t cap, int64_t *h_count)
{
    return slice_run<true>(ctx, d_bits_i, d_bits_q, n, h_params, d_data, d_addr, cap, h_count);
}

int pm_slicer_stats(pm_ctx *ctx, int32_t *iterations, int32_t *chunk_len, int64_t *chunks)
{
    PM_ARG(ctx != nullptr);
    if (iterations) *iterations = ctx->sl_iterations;
    if (chunk_len) *chunk_len = ctx->sl_chunk_len;
    if (chunks) *chunks = ctx->sl_chunks;
    return PM_OK;
}

}  // extern "C"
