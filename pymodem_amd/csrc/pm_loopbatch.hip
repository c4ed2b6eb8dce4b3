// Batch engine for the carrier-loop modems: R recordings x C chains through band-pass -> AGC -> (Hilbert pair) -> carrier loop ->
// matched / output filter -> sign bitmap, in TIME CHUNKS with every sequential state carried on the device.
//
// Why: a carrier loop (psk.py:173-189, psk.py:734-747, afsk_pll.py:153-165) is one dependent binary64 chain per sample through
// quantisers; a GPU lane steps it at 170-300 ns per sample, slower than one host core, and nothing makes one loop faster
// (DESIGN.md 4.5).  What the GPU has is room: a loop occupies one lane of one wave, so hundreds of loops -- the chains of MANY
// recordings -- cost what one costs.  This engine is what puts them side by side: one loop launch per chunk steps all R x C loops,
// each from the state the previous chunk left in device memory.  Chunking keeps the float64 intermediates at
// R x C x chunk x 16 B instead of R x C x recording x 16 B, so that the number of recordings in flight is bounded by how many the
// host has, not by HBM.
//
// Per chunk t (chunks are cut in the domain of the LAST filter's output, Lc outputs each, so every chunk's sign bits start on a
// 64-bit word of the bitmaps):
//   front stream:  band-pass of every recording (rows launch)  ->  AGC rows (envelope follower continued from the carried state)
//                  ->  MPSK: Hilbert FIR over [carried history | new] + the delayed real part  ->  loop inputs of set t & 1
//   back stream:   all R x C loops over the chunk (state in d_loops)  ->  matched filter over [carried history | new] writing
//                  sign bits at word t * Lc / 64 of every stream's bitmap
// The front stream works one chunk ahead of the back stream (two input sets, events both ways).  AGC.apply normalises by the
// maximum of the whole band-passed recording (agc.py:67), so a pass of band-pass + row maxima over all chunks precedes chunk 0;
// the band-pass is computed twice rather than stored (57.6 MB read again against 230 MB written and read per recording).
// Every kernel is the one the per-recording path uses (or its rows form: same tile code, same arithmetic), every sequential
// statement executes in the reference's order, chunk boundaries only decide WHEN: results are bit-identical to pm_chain_run and to
// the stage objects' demod() for every chunk length (tests/test_gpu_loopbatch.py).
#include "pm_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

struct pm_lbatch {
    pm_ctx *back = nullptr;                  // the caller's context: loops and output filter
    pm_ctx *front = nullptr;                 // own context: band-pass, AGC, Hilbert pair of the next chunk
    pm_ctx *tail = nullptr;                  // own context: the matched filter(s) of chunk t beside the loops of chunk t + 1
    pm_ctx *loop = nullptr;                  // own context on compute units of its own (or nullptr: the loops run on the caller's stream)
    hipEvent_t loop_go = nullptr, loop_end = nullptr;
    int modem = 0, R = 0, C = 0;
    int mb = 0, mh = 0, mo = 0, delay = 0;
    bool mpsk = false, two_out = false;
    int64_t Lc = 0, pitch = 0;
    int Hh = 0, Ho = 0;                      // history slots in front of the windows (mh - 1 and mo - 1 rounded up to even)
    pm_agc_params agc{};
    std::vector<double> h_taps;
    double *d_taps = nullptr;
    pm_bpf8_plan *bpf8 = nullptr;            // the band-pass on the matrix pipe, for the pass that finds the AGC's `normal` (pm_bpf8.hip; PM_BPF8_MAX=0: off)
    unsigned long long *d_keys = nullptr;
    pm_rowslice *rs = nullptr;               // pm_lbatch_run_sliced: the slicers' parameter table ...
    uint64_t *cbits_i = nullptr, *cbits_q = nullptr;      // ... and two chunks of sign bits per stream in rotation (rows of cstride words; set s at
    int64_t cstride = 0, cset = 0;                        // + s * cset): the slicers of chunk t run on a stream of their own beside the matched
    pm_ctx *slice = nullptr;                              // filters of chunk t + 1
    hipEvent_t bits_done[2] = {nullptr, nullptr}, slice_done[2] = {nullptr, nullptr};
    pm_fir8_plan *fir8 = nullptr;            // the output filter as certified signs on the int8 matrix pipe (pm_fir8.hip; PM_FIR8=0: off)
    size_t o_in = 0, o_hil = 0, o_out = 0, o_wave = 0;
    int32_t *d_pd = nullptr;
    std::vector<pm_loop> h_loops;            // R x C: the C initial loops repeated
    pm_loop *d_loops = nullptr;
    double *tmp = nullptr, *awin = nullptr, *in0[2] = {nullptr, nullptr}, *in1[2] = {nullptr, nullptr};
    double *dwin0[2] = {nullptr, nullptr}, *dwin1[2] = {nullptr, nullptr};    // the loops' outputs, two sets taking turns
    double *d_running = nullptr, *d_partial = nullptr, *d_consts = nullptr, *d_agc_state = nullptr;
    const int16_t **d_audio = nullptr;
    std::vector<const int16_t *> h_audio;
    hipEvent_t front_done[2] = {nullptr, nullptr}, back_done[2] = {nullptr, nullptr}, hist_done[2] = {nullptr, nullptr}, run_done = nullptr;
    bool ran = false;
    int64_t last_chunks = 0;
};

namespace {

size_t put(std::vector<double> &v, const double *src, int n)
{
    const size_t at = v.size();
    if (src && n > 0) v.insert(v.end(), src, src + n);
    while (v.size() % 2) v.push_back(0.0);                 // every vector 16-byte aligned on the device
    return at;
}

template <typename T>
int dev_alloc(pm_ctx *ctx, T *&p, size_t count)
{
    void *q = nullptr;
    if (int rc = pm_malloc(ctx, count * sizeof(T), &q)) return rc;
    p = (T *)q;
    return PM_OK;
}

int round_even(int v) { return (v + 1) & ~1; }

}  // namespace

// Rows of `width` doubles from one pitched block to another (or to another place of the same block: never overlapping).  The
// runtime's rectangle copy does this at a fraction of the memory rate (hipMemcpy2DAsync: 5.9 ms on average for the engine's copies,
// 17 % of the GPU time of a qpsk_2400 run in profiles/r03_qpsk_2400_kernel_stats.csv as first collected); rows are far apart, so
// one grid row per row and consecutive lanes on consecutive doubles.
__global__ __launch_bounds__(256) void rows_copy_kernel(double *__restrict__ dst, int64_t dst_stride, const double *__restrict__ src,
                                                        int64_t src_stride, int64_t width)
{
    const double *s = src + (int64_t)blockIdx.y * src_stride;
    double *d = dst + (int64_t)blockIdx.y * dst_stride;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < width; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}

static int rows_copy(pm_ctx *ctx, double *dst, int64_t dst_stride, const double *src, int64_t src_stride, int64_t width, int rows)
{
    if (width <= 0 || rows <= 0) return PM_OK;
    const unsigned gx = (unsigned)std::min<int64_t>(pm_cdiv(width, 256 * 4), 64);
    for (int r0 = 0; r0 < rows; r0 += 65535) {
        const int nr = std::min(rows - r0, 65535);
        hipLaunchKernelGGL(rows_copy_kernel, dim3(std::max(gx, 1u), (unsigned)nr), dim3(256), 0, ctx->stream, dst + (int64_t)r0 * dst_stride, dst_stride,
                           src + (int64_t)r0 * src_stride, src_stride, width);
    }
    PM_HIP(hipGetLastError());
    return PM_OK;
}

extern "C" {

int pm_lbatch_destroy(pm_lbatch *b)
{
    if (!b) return PM_OK;
    pm_ctx *ctx = b->back;
    if (b->front) (void)pm_ctx_sync(b->front);
    if (b->tail) (void)pm_ctx_sync(b->tail);
    if (b->slice) (void)pm_ctx_sync(b->slice);
    if (ctx) (void)pm_ctx_sync(ctx);
    for (void *p : {(void *)b->d_taps, (void *)b->d_pd, (void *)b->d_loops, (void *)b->tmp, (void *)b->awin, (void *)b->in0[0], (void *)b->in0[1],
                    (void *)b->in1[0], (void *)b->in1[1], (void *)b->dwin0[0], (void *)b->dwin0[1], (void *)b->dwin1[0], (void *)b->dwin1[1], (void *)b->d_running, (void *)b->d_partial,
                    (void *)b->d_consts, (void *)b->d_agc_state, (void *)b->d_audio, (void *)b->d_keys})
        if (p) (void)pm_free(ctx, p);
    for (hipEvent_t e : {b->front_done[0], b->front_done[1], b->back_done[0], b->back_done[1], b->hist_done[0], b->hist_done[1], b->run_done})
        if (e) (void)hipEventDestroy(e);
    pm_fir8_plan_destroy(b->fir8);
    pm_bpf8_plan_destroy(b->bpf8);
    pm_rowslice_destroy(b->rs);
    if (b->cbits_i) (void)pm_free(ctx, b->cbits_i);
    if (b->cbits_q) (void)pm_free(ctx, b->cbits_q);
    if (b->loop) (void)pm_ctx_sync(b->loop);
    for (hipEvent_t e : {b->loop_go, b->loop_end})
        if (e) (void)hipEventDestroy(e);
    if (b->loop) (void)pm_ctx_destroy(b->loop);
    if (b->front) (void)pm_ctx_destroy(b->front);
    if (b->tail) (void)pm_ctx_destroy(b->tail);
    if (b->slice) (void)pm_ctx_destroy(b->slice);
    for (hipEvent_t e : {b->bits_done[0], b->bits_done[1], b->slice_done[0], b->slice_done[1]})
        if (e) (void)hipEventDestroy(e);
    delete b;
    return PM_OK;
}

int pm_lbatch_create(pm_ctx *ctx, const pm_lbatch_desc *desc, pm_lbatch **out)
{
    PM_CTX(ctx);
    PM_ARG(desc != nullptr && out != nullptr);
    const pm_lbatch_desc &d = *desc;
    PM_ARG(d.modem == PM_MODEM_BPSK || d.modem == PM_MODEM_MPSK || d.modem == PM_MODEM_AFSK_PLL || d.modem == PM_MODEM_QPSK);
    PM_ARG(d.recordings >= 1 && d.chains >= 1 && (int64_t)d.recordings * d.chains <= (1 << 20));
    PM_ARG(d.input_fir && d.n_input_fir >= 1 && d.output_fir && d.n_output_fir >= 1 && d.loops && d.wavetable);
    PM_ARG(d.agc.sample_rate > 0);
    if (d.modem == PM_MODEM_MPSK) PM_ARG(d.hilbert && d.n_hilbert >= 1 && d.hilbert_delay >= 0 && d.hilbert_delay < d.n_hilbert && d.pd_table);
    PM_ARG(d.chunk >= 0);
    pm_lbatch *b = new pm_lbatch();
    b->back = ctx;
    b->modem = d.modem;
    b->R = d.recordings;
    b->C = d.chains;
    b->mb = d.n_input_fir;
    b->mo = d.n_output_fir;
    b->mpsk = d.modem == PM_MODEM_MPSK;
    b->mh = b->mpsk ? d.n_hilbert : 1;
    b->delay = b->mpsk ? d.hilbert_delay : 0;
    b->two_out = d.modem == PM_MODEM_MPSK || d.modem == PM_MODEM_QPSK;
    b->agc = d.agc;
    // a chunk is at least twice the longest filter (the history moves below never overlap) and a whole number of FIR tiles
    const int64_t longest = std::max(b->mb, std::max(b->mh, b->mo));
    int64_t lc = d.chunk > 0 ? d.chunk : 262144;
    lc = std::max<int64_t>(lc, 2 * longest);
    b->Lc = (lc + 2047) / 2048 * 2048;
    b->Hh = round_even(b->mh - 1);
    b->Ho = round_even(b->mo - 1);
    b->pitch = (b->Lc + 2 * b->mo + 2 * b->mh + 32 + 7) / 8 * 8;
    int rc = PM_OK;
    do {
        // Compute units for the carrier loops alone.  A loop's wave is one dependent binary64 chain that also fills about half of its
        // SIMD's issue slots (10 cycles per dependent operation, 5.5 to issue it): a filter wave on the same SIMD -- or an AGC wave, another
        // chain -- costs it a third to a half of its pace, and a loop launch ends with its slowest wave (measured: loops of a qpsk_2400
        // chunk 14.2 ms alone, 26-28 ms beside the engine's other streams).  With one wave per SIMD on units the other streams' masks
        // exclude, the loops keep their pace whatever runs elsewhere -- up to what loop waves cost each other on one unit: measured
        // (profiles/r04_loop_sweep.txt), qpsk_2400 with 16 384 loops 8.3 / 7.1 / 6.5-6.8 ms per step on 64 / 96 / 128 units (8.0-8.7 without),
        // bpsk_300 with 16 384 loops 1.30 / 1.48 on 64 / 128 (1.38 without).  Since the pass for the AGC's `normal` left the vector pipe
        // (pm_bpf8.hip) the filters need fewer units and the balance moved: bpsk_300, 16 384 loops, 0.97 / 0.77-0.80 / 0.96 ms per step on
        // 64 / 96 / 128 -- the loops' own time goes 0.73 / 0.53 / 0.50 ms with four / three / two waves on a unit (80 and 88 units still
        // leave four on some), the filters' 0.88 / 0.89 / 1.12; qpsk_2400 5.44-5.48 on 128 and on 160, 6.9-7.1 on 176
        // (profiles/r04_loop_sweep.txt).  So: two-output loops (two rows in, two out, the phase detector's table) two waves per unit, the
        // Costas loop eight on three.  -1: that, at most half the device; 0: no partition.
        int cus = ctx->tune.lbatch_loop_cus;
        const int have = pm_device_cus(ctx->device);
        const int64_t waves = ((int64_t)d.recordings * d.chains + 63) / 64;
        const bool two = d.modem == PM_MODEM_MPSK || d.modem == PM_MODEM_QPSK;
        if (cus < 0) cus = waves >= 64 ? (int)std::min<int64_t>(two ? (waves + 1) / 2 : (3 * waves + 7) / 8, have / 2) : 0;
        cus = std::min(cus, have - 8) / 8 * 8;                // whole rows of the XCDs (bit i of a mask is unit i / 8 of XCD i % 8)
        if (cus >= 8 && have >= 16 && have <= 1024) {
            uint32_t mine[32] = {0}, rest[32] = {0};
            for (int i = 0; i < have; ++i) (i < cus ? mine : rest)[i / 32] |= 1u << (i % 32);
            const int nw = (have + 31) / 32;
            if ((rc = pm_ctx_create_cumask(ctx->device, mine, nw, &b->loop))) break;
            if ((rc = pm_ctx_create_cumask(ctx->device, rest, nw, &b->front))) break;
            if ((rc = pm_ctx_create_cumask(ctx->device, rest, nw, &b->tail))) break;
            // The row slicers share the LOOPS' units: those hold three (or fewer) loop waves on four SIMDs, a slicer wave -- a dependent
            // chain with a word of loads per 64 samples and no LDS -- takes the fourth.  On the filters' units it waited for issue slots
            // behind their waves (0.82 ms of kernel time per qpsk recording, 0.35 here) and took 7 % of that half's capacity, which is
            // the half that sets the pace of a 3072-recording qpsk run: engine 3.47-3.49 -> 3.34-3.44 ms per recording (the loops
            // themselves 2.97 -> 3.16-3.31), bpsk_300 0.611 -> 0.588 (profiles/r04_loop_sweep.txt)
#ifndef PM_LBATCH_SLICE_ON_LOOP_CUS
#define PM_LBATCH_SLICE_ON_LOOP_CUS 1
#endif
            if ((rc = pm_ctx_create_cumask(ctx->device, PM_LBATCH_SLICE_ON_LOOP_CUS ? mine : rest, nw, &b->slice))) break;
            b->loop->tune = ctx->tune;
            hipError_t e1 = hipEventCreateWithFlags(&b->loop_go, hipEventDisableTiming), e2 = hipEventCreateWithFlags(&b->loop_end, hipEventDisableTiming);
            if (e1 != hipSuccess || e2 != hipSuccess) { rc = pm_set_error(PM_ERR_HIP, "hipEventCreate failed"); break; }
        } else {
            if ((rc = pm_ctx_create_prio(ctx->device, 0, &b->front))) break;
            if ((rc = pm_ctx_create_prio(ctx->device, 0, &b->tail))) break;
            if ((rc = pm_ctx_create_prio(ctx->device, 0, &b->slice))) break;
        }
        b->front->tune = ctx->tune;
        b->tail->tune = ctx->tune;
        b->slice->tune = ctx->tune;
        b->o_in = put(b->h_taps, d.input_fir, d.n_input_fir);
        if (b->mpsk) b->o_hil = put(b->h_taps, d.hilbert, d.n_hilbert);
        b->o_out = put(b->h_taps, d.output_fir, d.n_output_fir);
        b->o_wave = put(b->h_taps, d.wavetable, 256);
        // The output filter feeds the slicer's sign test and nothing else (psk.py:193 -> slicer.py:74, psk.py:750-751 -> slicer.py:215):
        // from 64 taps on its sums go to the matrix pipe with certified signs (the short PLL low-pass stays on the vector pipe)
        if (ctx->tune.fir8 && d.n_output_fir >= 64 && d.n_output_fir + 15 <= 1024 && (rc = pm_fir8_plan_create(ctx, d.output_fir, d.n_output_fir, &b->fir8))) break;
        // max(band-passed recording) needs no band-passed recording (pm_bpf8.hip: bpf8_max_kernel); up to 241 taps
        if (ctx->tune.bpf8_max && d.n_input_fir + 15 <= 256) {
            bool finite = true, some = false;
            for (int t = 0; t < d.n_input_fir; ++t) finite = finite && std::isfinite(d.input_fir[t]), some = some || d.input_fir[t] != 0.0;
            if (finite && some && (rc = pm_bpf8_plan_create(ctx, d.input_fir, d.n_input_fir, &b->bpf8, pm_bpf8_max_digits()))) break;
            if (b->bpf8 && (rc = dev_alloc(ctx, b->d_keys, (size_t)b->R))) break;
        }
        if ((rc = dev_alloc(ctx, b->d_taps, b->h_taps.size()))) break;
        if ((rc = pm_h2d(ctx, b->d_taps, b->h_taps.data(), b->h_taps.size() * sizeof(double)))) break;
        if (b->mpsk) {
            if ((rc = dev_alloc(ctx, b->d_pd, 4096))) break;
            if ((rc = pm_h2d(ctx, b->d_pd, d.pd_table, 4096 * sizeof(int32_t)))) break;
        }
        const size_t R = (size_t)b->R, RC = R * (size_t)b->C, P = (size_t)b->pitch;
        b->h_loops.resize(RC);
        for (size_t r = 0; r < R; ++r)
            for (int c = 0; c < b->C; ++c) b->h_loops[r * b->C + c] = d.loops[c];
        if ((rc = dev_alloc(ctx, b->d_loops, RC))) break;
        if ((rc = dev_alloc(ctx, b->tmp, R * P))) break;
        if (b->mpsk && (rc = dev_alloc(ctx, b->awin, R * P))) break;
        for (int s = 0; s < 2 && !rc; ++s) {
            rc = dev_alloc(ctx, b->in0[s], R * P);
            if (!rc && b->mpsk) rc = dev_alloc(ctx, b->in1[s], R * P);
        }
        if (rc) break;
        for (int k = 0; k < 2 && !rc; ++k) {
            rc = dev_alloc(ctx, b->dwin0[k], RC * P);
            if (!rc && b->two_out) rc = dev_alloc(ctx, b->dwin1[k], RC * P);
        }
        if (rc) break;
        if ((rc = dev_alloc(ctx, b->d_running, R))) break;
        if ((rc = dev_alloc(ctx, b->d_partial, R * (size_t)pm_rows_max_parts()))) break;
        if ((rc = dev_alloc(ctx, b->d_consts, 4 * R))) break;
        if ((rc = dev_alloc(ctx, b->d_agc_state, 2 * R))) break;
        if ((rc = dev_alloc(ctx, b->d_audio, R))) break;
        b->h_audio.resize(R);
        hipError_t e = hipSuccess;
        for (hipEvent_t *ev : {&b->front_done[0], &b->front_done[1], &b->back_done[0], &b->back_done[1], &b->hist_done[0], &b->hist_done[1], &b->run_done})
            if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        if (e != hipSuccess) { rc = pm_set_error(PM_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e)); break; }
        rc = pm_ctx_sync(ctx);                                 // the constants are up before the desc's pointers go away
    } while (0);
    if (rc) { pm_lbatch_destroy(b); return rc; }
    *out = b;
    return PM_OK;
}

int pm_lbatch_geometry(pm_lbatch *b, int64_t n, int64_t *h_nout, int64_t *h_chunk, int64_t *h_chunks)
{
    PM_ARG(b != nullptr);
    const int64_t nout = n - (b->mb - 1) - (b->mh - 1) - (b->mo - 1);
    if (h_nout) *h_nout = nout;
    if (h_chunk) *h_chunk = b->Lc;
    if (h_chunks) *h_chunks = nout > 0 ? pm_cdiv(nout, b->Lc) : 0;
    return PM_OK;
}

pm_ctx *pm_lbatch_front_ctx(pm_lbatch *b) { return b ? b->front : nullptr; }
pm_ctx *pm_lbatch_tail_ctx(pm_lbatch *b) { return b ? b->tail : nullptr; }
pm_ctx *pm_lbatch_loop_ctx(pm_lbatch *b) { return b ? b->loop : nullptr; }
pm_ctx *pm_lbatch_slice_ctx(pm_lbatch *b) { return b ? b->slice : nullptr; }

namespace {
struct SliceOut {                      // pm_lbatch_run_sliced: where the rows' bytes, steps and records go
    uint8_t *data;
    uint16_t *steps;
    int64_t cap;
    pm_rowslice_rec *recs;
};
int run_impl(pm_lbatch *b, const int16_t *const *h_d_audio, int recordings, int64_t n, uint64_t *d_bits_i, uint64_t *d_bits_q, int64_t bits_stride,
             int64_t *h_nout, const SliceOut *so);
}  // namespace

int pm_lbatch_run(pm_lbatch *b, const int16_t *const *h_d_audio, int recordings, int64_t n, uint64_t *d_bits_i, uint64_t *d_bits_q,
                  int64_t bits_stride, int64_t *h_nout)
{
    PM_ARG(b != nullptr && h_d_audio != nullptr && d_bits_i != nullptr && h_nout != nullptr);
    return run_impl(b, h_d_audio, recordings, n, d_bits_i, d_bits_q, bits_stride, h_nout, nullptr);
}

int pm_lbatch_run_sliced(pm_lbatch *b, const int16_t *const *h_d_audio, int recordings, int64_t n, const pm_slicer_params *h_params, int nparams,
                         uint8_t *d_data, uint16_t *d_steps, int64_t cap, pm_rowslice_rec *d_recs, int64_t *h_nout)
{
    PM_ARG(b != nullptr && h_d_audio != nullptr && h_params != nullptr && d_data != nullptr && d_steps != nullptr && d_recs != nullptr && h_nout != nullptr);
    PM_ARG(nparams == b->C && cap >= 8 && cap % 8 == 0);
    pm_ctx *ctx = b->back;
    PM_CTX(ctx);
    if (!pm_rowslice_made_for(b->rs, h_params, nparams)) {
        if (b->rs) {                                          // (a run with the old table may still be in flight)
            if (int rc = pm_ctx_sync(b->slice)) return rc;
            pm_rowslice_destroy(b->rs);
            b->rs = nullptr;
        }
        if (int rc = pm_rowslice_create(ctx, h_params, nparams, &b->rs)) return rc;
    }
    if (!b->cbits_i) {
        b->cstride = b->Lc / 64 + 8;
        b->cset = (int64_t)b->R * b->C * b->cstride;
        const size_t words = 2 * (size_t)b->cset;
        for (hipEvent_t *ev : {&b->bits_done[0], &b->bits_done[1], &b->slice_done[0], &b->slice_done[1]})
            if (!*ev && hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) return pm_set_error(PM_ERR_HIP, "hipEventCreate failed");
        // both sets before either is published: a call that fails half way leaves nothing behind and the next one tries again
        void *qi = nullptr, *qq = nullptr;
        if (int rc = pm_malloc(ctx, words * 8, &qi)) return rc;
        if (b->two_out)
            if (int rc = pm_malloc(ctx, words * 8, &qq)) {
                (void)pm_free(ctx, qi);
                return rc;
            }
        b->cbits_i = (uint64_t *)qi;
        b->cbits_q = (uint64_t *)qq;
    }
    const SliceOut so{d_data, d_steps, cap, d_recs};
    return run_impl(b, h_d_audio, recordings, n, b->cbits_i, b->cbits_q, b->cstride, h_nout, &so);
}

namespace {
int run_impl(pm_lbatch *b, const int16_t *const *h_d_audio, int recordings, int64_t n, uint64_t *d_bits_i, uint64_t *d_bits_q, int64_t bits_stride,
             int64_t *h_nout, const SliceOut *so)
{
    const int tail_mode = b->back ? b->back->tune.lbatch_tail : -1;
    // The matched filters of chunk t beside the loops of chunk t + 1 (third stream; PM_LBATCH_TAIL=0: behind them on the caller's stream).
    // With the tiled loop shapes this paid only sometimes -- beside a filter those loops ran 1.8-2.8x slower, their LDS traffic queuing
    // behind the filter's -- with the direct shape (no tiles in LDS) it pays everywhere: same box, same minute, third stream on / off:
    // bpsk_300 (8192 x 1) 14 744 / 12 035 Msamples/s, qpsk_2400 2048 x 8 chains 18 616 / 17 438, 256 x 64 chains 22 062 / 18 865.
    const bool use_tail = tail_mode != 0;
    pm_ctx *B = b->back, *F = b->front, *Tl = use_tail ? b->tail : b->back;
    pm_ctx *Lp = b->loop ? b->loop : B;                       // the loops' stream (the caller's, or the engine's own on its own compute units)
    for (pm_ctx *c : {b->front, b->tail, b->loop, b->slice})  // the caller's switches (pm_ctx_tune) hold for the engine's own contexts too
        if (c) c->tune = B->tune;
    PM_CTX(B);
    PM_ARG(recordings >= 1 && recordings <= b->R);
    PM_ARG(!b->two_out || d_bits_q != nullptr);
    const int R = recordings, C = b->C, RC = R * C;
    const int mb = b->mb, mh = b->mh, mo = b->mo;
    const int64_t P = b->pitch, Lc = b->Lc;
    const int64_t na = n - (mb - 1), nh = na - (mh - 1), nout = nh - (mo - 1);
    if (nout < 1)
        return pm_set_error(PM_ERR_ARG, "pm_lbatch_run: %lld samples are fewer than the filters need for one output (%d + %d + %d taps)",
                            (long long)n, mb, mh, mo);
    PM_ARG(bits_stride >= (std::min(nout, so ? Lc : nout) + 63) / 64);
    *h_nout = nout;
    const double *T = b->d_taps;
    bool aligned = true;
    for (int r = 0; r < R; ++r) {
        PM_ARG(h_d_audio[r] != nullptr);
        b->h_audio[r] = h_d_audio[r];
        aligned = aligned && (((uintptr_t)h_d_audio[r]) & 15) == 0;
    }
    // a run starts when the previous one has left both streams (its buffers and states are re-used)
    if (b->ran) {
        PM_HIP(hipStreamWaitEvent(F->stream, b->run_done, 0));
        PM_HIP(hipStreamWaitEvent(B->stream, b->run_done, 0));
        PM_HIP(hipStreamWaitEvent(Tl->stream, b->run_done, 0));
        PM_HIP(hipStreamWaitEvent(Lp->stream, b->run_done, 0));
    }
    // ---- pass 1: normal = max(band-passed recording) per row (agc.py:67).  On the matrix pipe (pm_bpf8.hip) it needs no band-passed
    // recording, and runs on the caller's stream: nothing else of this run can start before it, and that stream has every compute unit
    // (the engine's own may be masked to their share)
    const bool max8 = b->bpf8 && aligned && B->tune.bpf8_max && pm_cdiv(na, (int64_t)4096) <= 65535;
    if (max8) {
        PM_HIP(hipMemcpyAsync(b->d_audio, b->h_audio.data(), sizeof(void *) * (size_t)R, hipMemcpyHostToDevice, B->stream));
        if (int rc = pm_bpf8_rows_max(B, b->bpf8, b->d_audio, R, n, b->d_keys, b->d_running)) return rc;
    }
    // the front stream is ordered behind whatever the caller has enqueued on its context so far (e.g. the recordings' uploads)
    PM_HIP(hipEventRecord(b->run_done, B->stream));
    PM_HIP(hipStreamWaitEvent(F->stream, b->run_done, 0));
    if (!max8) PM_HIP(hipMemcpyAsync(b->d_audio, b->h_audio.data(), sizeof(void *) * (size_t)R, hipMemcpyHostToDevice, F->stream));
    PM_HIP(hipMemsetAsync(b->d_agc_state, 0, sizeof(double) * 2 * (size_t)R, F->stream));          // fresh AGC objects (agc.py:20-21)
    PM_HIP(hipMemcpyAsync(b->d_loops, b->h_loops.data(), sizeof(pm_loop) * (size_t)RC, hipMemcpyHostToDevice, B->stream));
    if (so) PM_HIP(hipMemsetAsync(so->recs, 0, sizeof(pm_rowslice_rec) * (size_t)RC, B->stream));      // just-tuned slicers (slicer.py:49-56, :193-202)
    if (Lp != B) {                                            // the loops start behind the fresh states
        PM_HIP(hipEventRecord(b->loop_go, B->stream));
        PM_HIP(hipStreamWaitEvent(Lp->stream, b->loop_go, 0));
    }

    // (pass 1 the reference's way: its sums over every chunk, then their maxima)
    for (int64_t at = 0, first = 1; at < na && !max8; at += Lc, first = 0) {
        const int64_t cnt = std::min(Lc, na - at);
        if (int rc = pm_fir_rows(F, true, nullptr, 0, (const void *const *)b->d_audio, at, aligned, R, cnt + mb - 1, T + b->o_in, mb, b->tmp, P, nullptr,
                                 0, 0)) return rc;
        if (int rc = pm_rows_max(F, b->tmp, P, R, cnt, b->d_partial, b->d_running, (int)first)) return rc;
    }
    if (int rc = pm_agc_rows_prepare(F, b->d_running, R, &b->agc, b->d_consts)) return rc;

    // ---- pass 2: chunk by chunk ---------------------------------------------------------------------------------------------
    const int64_t chunks = pm_cdiv(nout, Lc);
    b->last_chunks = chunks;
    int64_t s_agc = 0, s_loop = 0, prev_cnt_l = 0;
    // BPSK, one chain per recording (psk.py:168-189): the AGC's only reader is the loop -- its lane steps the follower too (pm_loops.hip)
    const bool fold_agc = pm_loops_rows_take_agc(Lp, b->modem, RC, C);
    int64_t skip_of[2] = {0, 0};
    for (int64_t t = 0; t < chunks; ++t) {
        const int set = (int)(t & 1);
        const int64_t o0 = t * Lc, o1 = std::min(nout, o0 + Lc);
        const int64_t e_loop = o1 + mo - 1, e_agc = e_loop + (mh - 1);
        const int64_t cnt_a = e_agc - s_agc, cnt_l = e_loop - s_loop;
        const bool more = t + 1 < chunks;
        // -- front: the loop inputs of chunk t into set t & 1
        if (t >= 2) PM_HIP(hipStreamWaitEvent(F->stream, b->back_done[set], 0));            // the loops of chunk t - 2 have read the set
        {
            // band-pass from the 16-byte boundary at or below the first new sample: `skip` outputs are computed again and not used
            const int64_t a0 = s_agc & ~(int64_t)7, skip = s_agc - a0;
            // (with the AGC in the loop's lane the band-passed samples themselves are the chunk's loop input, `skip` samples in)
            if (int rc = pm_fir_rows(F, true, nullptr, 0, (const void *const *)b->d_audio, a0, aligned, R, cnt_a + skip + mb - 1, T + b->o_in, mb,
                                     fold_agc ? b->in0[set] : b->tmp, P, nullptr, 0, 0)) return rc;
            skip_of[set] = skip;
            if (!fold_agc) {
                double *agc_out = b->mpsk ? b->awin + b->Hh : b->in0[set];
                if (int rc = pm_agc_rows(F, b->tmp + skip, P, agc_out, P, R, cnt_a, &b->agc, b->d_consts, b->d_agc_state)) return rc;
            }
            if (b->mpsk) {
                // imag = Hilbert FIR over [history | new]; real[k] = a[k + delay] (the delay FIR [1, 0, ..] and [:-delay], psk.py:714-716)
                const double *hin = t == 0 ? b->awin + b->Hh : b->awin + b->Hh - (mh - 1);
                const int64_t hn = t == 0 ? cnt_a : cnt_a + mh - 1;
                if (int rc = pm_fir_rows(F, false, hin, P, nullptr, 0, (((uintptr_t)hin) & 15) == 0, R, hn, T + b->o_hil, mh, b->in1[set], P, nullptr, 0,
                                         0)) return rc;
                if (int rc = rows_copy(F, b->in0[set], P, hin + b->delay, P, cnt_l, R)) return rc;
                if (more && mh > 1)         // the last mh - 1 AGC'd samples go in front of the next chunk's (cnt_a >= 2 (mh - 1): no overlap)
                    if (int rc = rows_copy(F, b->awin + b->Hh - (mh - 1), P, b->awin + b->Hh + cnt_a - (mh - 1), P, mh - 1, R)) return rc;
            }
            PM_HIP(hipEventRecord(b->front_done[set], F->stream));
        }
        // -- back: every loop over the chunk into output set t & 1.  The set is free once the matched filters of chunk t - 2 have read it
        // and chunk t - 1's history has been taken out of... the OTHER set; both lie behind hist_done of chunk t - 1 on the tail stream.
        PM_HIP(hipStreamWaitEvent(Lp->stream, b->front_done[set], 0));
        if (t >= 1) PM_HIP(hipStreamWaitEvent(Lp->stream, b->hist_done[(t - 1) & 1], 0));
        if (fold_agc) {
            if (int rc = pm_loops_rows_agc(Lp, b->d_loops, RC, T + b->o_wave, b->in0[set] + skip_of[set], P, cnt_l, b->dwin0[set] + b->Ho, P, &b->agc, b->d_consts,
                                           b->d_agc_state)) return rc;
        } else if (int rc = pm_loops_rows(Lp, b->modem, b->d_loops, RC, C, T + b->o_wave, b->d_pd, b->in0[set], b->in1[set], P, cnt_l, b->dwin0[set] + b->Ho,
                                   b->two_out ? b->dwin1[set] + b->Ho : nullptr, P)) return rc;
        PM_HIP(hipEventRecord(b->back_done[set], Lp->stream));
        // -- tail: the output filter's sign bits for chunk t, beside the loops of chunk t + 1.  Its window is [history | new]: the last
        // mo - 1 loop outputs of chunk t - 1 come over from the other set first (the loops of chunk t + 1, which overwrite them, wait
        // for that copy: hist_done).
        PM_HIP(hipStreamWaitEvent(Tl->stream, b->back_done[set], 0));
        {
            const int64_t back = t == 0 ? 0 : mo - 1, fn = cnt_l + back;
            if (t >= 1 && mo > 1) {
                if (int rc = rows_copy(Tl, b->dwin0[set] + b->Ho - (mo - 1), P, b->dwin0[set ^ 1] + b->Ho + prev_cnt_l - (mo - 1), P, mo - 1, RC)) return rc;
                if (b->two_out)
                    if (int rc = rows_copy(Tl, b->dwin1[set] + b->Ho - (mo - 1), P, b->dwin1[set ^ 1] + b->Ho + prev_cnt_l - (mo - 1), P, mo - 1, RC)) return rc;
            }
            PM_HIP(hipEventRecord(b->hist_done[set], Tl->stream));
            const double *f0 = b->dwin0[set] + b->Ho - back;
            // sliced runs keep two chunks of sign bits in rotation, not the recording's: chunk t's go to set t & 1, free once the
            // slicers of chunk t - 2 are through with it
            const int64_t word0 = so ? (t & 1) * b->cset : o0 / 64;
            if (so && t >= 2) PM_HIP(hipStreamWaitEvent(Tl->stream, b->slice_done[t & 1], 0));
            auto signs = [&](const double *f, uint64_t *bits) -> int {
                // (a row of the output windows: Ho history slots, the chunk, slack up to the pitch -- all of it the engine's own memory)
                // (the switch is read per run, like the engine's others: pm_ctx_tune(fir8 = 0) on an engine that exists takes the binary64 rows kernel)
                if (b->fir8 && B->tune.fir8) return pm_fir8_rows_signs(Tl, b->fir8, f, P, RC, fn, bits + word0, bits_stride, nullptr, P - (b->Ho - back));
                return pm_fir_rows(Tl, false, f, P, nullptr, 0, (((uintptr_t)f) & 15) == 0, RC, fn, T + b->o_out, mo, nullptr, 0, bits + word0, bits_stride, 0);
            };
            if (int rc = signs(f0, d_bits_i)) return rc;
            if (b->two_out)
                if (int rc = signs(b->dwin1[set] + b->Ho - back, d_bits_q)) return rc;
            if (so) {
                // ... on a stream of their own: a lane per stream and a dependent chain per sample, nothing a matched filter has to wait behind
                pm_ctx *Sl = b->slice;
                PM_HIP(hipEventRecord(b->bits_done[t & 1], Tl->stream));
                PM_HIP(hipStreamWaitEvent(Sl->stream, b->bits_done[t & 1], 0));
                if (int rc = pm_rowslice_chunk(Sl, b->rs, RC, b->two_out ? 1 : 0, d_bits_i + word0, b->two_out ? d_bits_q + word0 : nullptr, bits_stride, o0,
                                               o1 - o0, so->recs, so->data, so->steps, so->cap)) return rc;
                PM_HIP(hipEventRecord(b->slice_done[t & 1], Sl->stream));
            }
        }
        prev_cnt_l = cnt_l;
        s_agc = e_agc;
        s_loop = e_loop;
    }
    // the tail stream's last filter ends the run: the caller's context waits for it
    PM_HIP(hipEventRecord(b->hist_done[0], Tl->stream));
    PM_HIP(hipStreamWaitEvent(B->stream, b->hist_done[0], 0));
    if (Lp != B) {                                            // (the tail's last filter is behind the last loops; the states they left: this)
        PM_HIP(hipEventRecord(b->loop_end, Lp->stream));
        PM_HIP(hipStreamWaitEvent(B->stream, b->loop_end, 0));
    }
    if (so) PM_HIP(hipStreamWaitEvent(B->stream, b->slice_done[(chunks - 1) & 1], 0));      // the last slicers end a sliced run
    // the run is complete on the caller's context once the back stream gets here; the next run waits for this point on both streams
    PM_HIP(hipEventRecord(b->run_done, B->stream));
    b->ran = true;
    return PM_OK;
}
}  // namespace

}  // extern "C"
