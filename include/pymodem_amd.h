/*
 * pymodem_amd.h -- C ABI of libpymodem_amd.so: the MI355X (gfx950) implementation of pymodem's
 * demod_chain sample-processing path.  Plain pointers and sizes only; no C++ or torch types.
 *
 * The reference (ninocarrillo/pymodem, pure Python) has no FFI: the "binding" a maintainer adds is
 * a ctypes stub inside each stage class.  Every entry point below names the
 * reference code it replaces (file:line relative to the reference checkout).
 *
 * Conventions
 *   - every function returns 0 on success, a negative pm_status on failure; pm_last_error() gives text
 *   - HIP is initialised in pm_ctx_create(), never at dlopen() time (the reference forks one process
 *     per chain after building its stage objects, pymodem.py:144-151)
 *   - `const T *d_*` / `T *d_*` arguments are DEVICE pointers (from pm_malloc or any HIP allocation of
 *     the same process, e.g. a torch tensor's data_ptr()); `h_*` arguments are HOST pointers
 *   - a pm_ctx owns one HIP stream; calls on one ctx are ordered; a ctx is not thread-safe
 *   - all floating point is IEEE binary64, evaluated in the reference's operation order; FIR sums use
 *     the build's canonical order: ascending input index, one fused multiply-add per tap
 */
#ifndef PYMODEM_AMD_H
#define PYMODEM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_VERSION 100          /* 0.1.0 */

typedef enum pm_status {
    PM_OK = 0,
    PM_ERR_HIP = -1,            /* a HIP runtime call failed (text in pm_last_error) */
    PM_ERR_ARG = -2,            /* bad argument (null pointer, n < taps, unsupported size) */
    PM_ERR_NODEV = -3,          /* no gfx950 device visible */
    PM_ERR_CAPACITY = -4,       /* caller's output buffer too small; required size reported */
    PM_ERR_NOCONVERGE = -5      /* internal: slicer walkers still alive after the launch cap (cannot happen: a walker ends with its stream) */
} pm_status;

typedef struct pm_ctx pm_ctx;

/* ---- library / device ------------------------------------------------------------------------ */
int pm_version(void);
int pm_device_count(void);                               /* 0 when no GPU; never throws */
int pm_last_error(char *buf, size_t cap);                /* copies the calling thread's last message */

int pm_ctx_create(int device, pm_ctx **out);
/* Same, with the stream at the device's highest priority when high_priority != 0.  Several ctx may exist on one device: each has
 * its own stream, scratch and profiler, device memory is shared.  The pipelined executor runs the slicer (a few resident waves,
 * dependent-latency bound) on a high-priority ctx while the FIR/correlator kernels of the next recording fill the ALUs. */
int pm_ctx_create_prio(int device, int high_priority, pm_ctx **out);
/* A context whose stream may only use the compute units whose bit is set in cu_mask (nwords x 32 bits; bit i goes to XCD
 * i mod 8, so a run of 8 k consecutive bits takes k CUs from every XCD).  The pipelined executor keeps the slicers' few, long-lived,
 * latency-bound waves on a handful of CUs of their own and the FIR kernels on the rest: a FIR workgroup that shares a SIMD with a
 * slicer wave runs at the pace of its slowest wave (its barriers couple the four SIMDs of the CU). */
int pm_ctx_create_cumask(int device, const uint32_t *cu_mask, int nwords, pm_ctx **out);
int pm_device_cus(int device);      /* compute units of the device (0 if it cannot be queried) */
/* Diagnostic switches of a context's launchers (kernel shapes kept for comparison, traces; README.md lists them).  A context reads
 * its switches from the environment (PM_<NAME>) once, when it is made; this call sets one afterwards.  No switch changes a result. */
int pm_ctx_tune(pm_ctx *ctx, const char *name, int64_t value);
int pm_ctx_destroy(pm_ctx *ctx);
/* Cross-stream ordering without a host wait.  pm_event_record marks the point reached by ctx's stream (creating the event when
 * *event is NULL); pm_event_wait makes everything submitted to ctx AFTER the call wait for that point.  Same device only. */
int pm_event_record(pm_ctx *ctx, void **event);
int pm_event_wait(pm_ctx *ctx, void *event);
int pm_event_query(void *event);     /* 1 = everything before the record has finished, 0 = not yet, < 0 = error */
int pm_event_sync(void *event);      /* host wait for the event */
int pm_event_sync_relaxed(void *event, int poll_us);   /* the same without spinning: polls every poll_us microseconds (waits of seconds) */
int pm_event_destroy(void *event);
int pm_ctx_sync(pm_ctx *ctx);                            /* hipStreamSynchronize on the ctx stream */
void *pm_ctx_stream(pm_ctx *ctx);                        /* the hipStream_t, for interop */

int pm_malloc(pm_ctx *ctx, size_t bytes, void **d_out);
int pm_free(pm_ctx *ctx, void *d_ptr);
int pm_h2d(pm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);   /* async on the ctx stream */
int pm_d2h(pm_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);   /* synchronous */
/* Page-lock a host block the caller owns (and release it again before freeing the block): copies to and from pinned memory go
 * straight over the link instead of through the runtime's bounce buffers.  For blocks that are used many times (the executor's
 * pool of result blocks); pinning costs about as much as one copy of the block. */
int pm_host_pin(pm_ctx *ctx, void *h_block, size_t bytes);
int pm_host_unpin(void *h_block);
/* The context's internal work block (slicer state, sweep intermediates): grow it to at least reserve_bytes now (0: leave it) and
 * report its size.  A host that knows how large its batches will get reserves once instead of paying a free + malloc on the way. */
int pm_ctx_scratch(pm_ctx *ctx, size_t reserve_bytes, size_t *h_bytes);
int pm_d2d(pm_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);        /* device to device, on the context's stream */
int pm_memset(pm_ctx *ctx, void *d_dst, int value, size_t bytes);

/* HIP-event timer on the ctx stream: start, enqueue work, stop -> elapsed milliseconds. */
int pm_timer_start(pm_ctx *ctx);
int pm_timer_stop(pm_ctx *ctx, float *ms);

/* Per-kernel-class timing with HIP events on the ctx stream, for bench.py's roofline line.  While enabled, every
 * launch of a tracked kernel class is bracketed by an event pair; pm_prof_read sums the elapsed times. */
enum { PM_K_FIR_I16 = 0, PM_K_FIR_F64 = 1, PM_K_AFSK_CORR = 2, PM_K_SIGNS = 3, PM_K_SLICE_ITER = 4, PM_K_SLICE_EMIT = 5,
       PM_K_AGC = 6, PM_K_LOOP = 7, PM_K_COUNT = 8 };
int pm_prof_enable(pm_ctx *ctx, int on);                 /* also resets the accumulators */
int pm_prof_read(pm_ctx *ctx, int kernel_class, double *total_ms, int64_t *launches);
/* Algorithmic work of the same launches, summed: compulsory HBM bytes (every input element read once, every output element
 * written once, at their stored width) and f64 flops (2 per fused multiply-add of the FIR sums; epilogues not counted).
 * Zero for the classes that are neither (slicer iterations, carrier loops, AGC). */
int pm_prof_work(pm_ctx *ctx, int kernel_class, double *bytes, double *flops);
/* When each tracked launch of the class began and ended, in milliseconds since one reference event per device (recorded when
 * profiling is first enabled there): launches of several contexts of a device can be laid over each other -- a host that runs
 * the same kernel class on several streams at once learns how many were in flight together and for how long the class kept
 * the GPU busy at all.  *h_n = intervals held (the first min(*h_n, cap) are written). */
int pm_prof_intervals(pm_ctx *ctx, int kernel_class, double *h_start_ms, double *h_end_ms, int64_t cap, int64_t *h_n);

/* ---- FIR stages -------------------------------------------------------------------------------
 * numpy.convolve(x, h, 'valid'): y[k] = sum_j h[j] * x[k+m-1-j], k = 0 .. n-m.  Replaces the 19
 * numpy.convolve call sites (afsk.py:151-166, fsk.py:151, psk.py:165,193,710-751, afsk_pll.py:143,168).
 * d_taps holds h in the reference's order.  flags: PM_FIR_NEGATE writes -y (fsk.py:153-154). */
#define PM_FIR_NEGATE 1
int pm_fir_valid_i16(pm_ctx *ctx, const int16_t *d_x, int64_t n, const double *d_taps, int m, double *d_y, int flags);
/* The same sum for callers that need a value with a bound, not the reference's rounding (the certified AFSK sweeps inside pm_pipe_*):
 * taps quantised to 32-bit integers, samples and taps as signed base-256 digits, exact int32 products on the int8 matrix pipe
 * (v_mfma_i32_16x16x64_i8), recombined in binary64.  |d_y[k] - reference sum| <= *h_bound (~1e-9 of sum|taps| * 32768).  m <= 241,
 * d_x 16-byte aligned, h_taps on the HOST; synchronous (plans its tables per call: the pipeline keeps them).  Test and measurement entry. */
int pm_fir_valid_i16_limbs(pm_ctx *ctx, const int16_t *d_x, int64_t n, const double *h_taps, int m, double *d_y, double *h_bound);
/* What the certified AFSK sweeps' binary32 roots rely on (csrc/pm_fir.hip: slide_run_f32): the largest error of the device's v_sqrt_f32
 * over ALL 2^24 binary32 values of the binades 2^exponent and 2^(exponent + 1), in units of the result's last place x 1024 (rounded up),
 * against the correctly rounded binary64 root.  Test entry. */
int pm_ubench_sqrt_f32(pm_ctx *ctx, int exponent, int64_t *h_worst_ulp_1024);
/* int8 digit products (v_mfma_i32_16x16x64_i8 operand pairs) per tap and output that the matrix-pipe kernels compute: stage 0 = the
 * certified sweeps' band-pass (sample digits x tap digits), 1 = their low-pass, per stream, 2 = the batch engine's matched filters.
 * From the kernels' own constants: what a measurement prices their launches with.  < 0: no such stage. */
int pm_matrix_digit_pairs(int stage);
/* max(numpy.convolve(row, h, 'valid')) per row -- AGC.apply's `normal` (agc.py:67) over the band-passed recording (psk.py:165, :710) --
 * WITHOUT writing the band-passed rows: matrix-pipe values with the bound above pick the outputs that could be the maximum, the
 * reference's own sum (one fma per tap, ascending input index) decides among them; h_max[r] is bit for bit max() of pm_fir_valid_i16.
 * Rows x_stride samples apart (a multiple of 8) from a 16-byte aligned d_x; *h_redone (may be null) = outputs recomputed.  m <= 241.
 * Test and measurement entry: pm_lbatch_* keeps a plan for its band-pass. */
int pm_bpf8_rows_max_i16(pm_ctx *ctx, const int16_t *d_x, int64_t x_stride, int rows, int64_t n, const double *h_taps, int m, double *h_max,
                         int64_t *h_redone);
int pm_fir_valid_f64(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_taps, int m, double *d_y, int flags);

/* The same FIRs fused with the slicer's sign test: only the (y >= 0) bitmap of the n-m+1 outputs is written (bit k of the
 * little-endian uint64 array; (n-m+1+63)/64 words), not the float64 stream.  For the last FIR of a chain, whose output
 * feeds slicer.slice() and nothing else (afsk.py:166, fsk.py:151, psk.py:193,750-751, afsk_pll.py:168). */
int pm_fir_signs_i16(pm_ctx *ctx, const int16_t *d_x, int64_t n, const double *d_taps, int m, uint64_t *d_bits, int flags);
int pm_fir_signs_f64(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_taps, int m, uint64_t *d_bits, int flags);
/* The same for up to 16 streams that share their taps, in ONE launch: the output low-passes of a chain group (every chain of
 * afsk_1200_ax25_super_opt.json has the same output_lpf).  h_x / h_n / h_bits are HOST arrays of `count` device pointers / lengths. */
int pm_fir_signs_f64_batch(pm_ctx *ctx, int count, const double *const *h_x, const int64_t *h_n, const double *d_taps, int m,
                           uint64_t *const *h_bits, int flags);

/* The same FIRs over `rows` streams of equal length in ONE launch (the band-pass, Hilbert and matched filters of a batch of
 * recordings x chains, pm_lbatch below).  Input row r is d_x + r * x_stride (elements), or -- pm_fir_rows_i16_ptrs, rows that are
 * separate allocations -- d_x_ptrs[r] + x_off with d_x_ptrs a DEVICE array of `rows` device pointers (x_aligned16: every row
 * pointer + x_off is 16-byte aligned, which the library cannot see).  Output row r is d_y + r * y_stride (n - m + 1 doubles) or
 * d_bits + r * bits_stride words.  Every row's result is bit-identical to the single-stream call. */
int pm_fir_rows_i16(pm_ctx *ctx, const int16_t *d_x, int64_t x_stride, int rows, int64_t n, const double *d_taps, int m, double *d_y,
                    int64_t y_stride, int flags);
int pm_fir_rows_i16_ptrs(pm_ctx *ctx, const int16_t *const *d_x_ptrs, int64_t x_off, int x_aligned16, int rows, int64_t n, const double *d_taps,
                         int m, double *d_y, int64_t y_stride, int flags);
int pm_fir_rows_f64(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, const double *d_taps, int m, double *d_y,
                    int64_t y_stride, int flags);
int pm_fir_rows_signs_f64(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, const double *d_taps, int m, uint64_t *d_bits,
                          int64_t bits_stride, int flags);
/* The same bitmap for LONG filters whose output feeds only a slicer (BPSK / MPSK matched filters, psk.py:193, :750-751): the sums as
 * int8 digit products on the matrix pipe, each sign certified against a proven bound, the undecided outputs recomputed with the
 * canonical binary64 chain (csrc/pm_fir8.hip; 16 <= m <= 1009).  Bit for bit pm_fir_rows_signs_f64's result for every input; this
 * entry point makes its plan per call (tests, measurements) -- the batch engine keeps one.  *h_recomputed: outputs that took the
 * exact path. */
int pm_fir8_rows_signs_f64(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, const double *h_taps, int m,
                           uint64_t *d_bits, int64_t bits_stride, int64_t *h_recomputed);

/* AFSK mark/space quadrature correlators fused with magnitude and difference (afsk.py:153-162):
 * y[k] = sqrt(mi*x ^2 + mq*x ^2) - sqrt(si*x ^2 + sq*x ^2), each product a 'valid' convolution. */
int pm_afsk_correlate(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_mark_i, const double *d_mark_q,
                      const double *d_space_i, const double *d_space_q, int m, double *d_y);

/* `groups` AFSK modems over the same band-passed stream that share their MARK correlators and differ in the space correlators
 * only (the chains of afsk_1200_ax25_super_opt.json: same tones and span, space_gain 1.25 ... 2.75, afsk.py:144-145): the mark
 * sums and magnitude are computed once.  d_space = [groups][2][m] (space_i then space_q of modem g); output g is written at
 * d_y + g*y_stride (n-m+1 values each).  Every output is bit-identical to pm_afsk_correlate with that modem's four filters. */
#define PM_AFSK_GROUP_MAX 8
int pm_afsk_correlate_group(pm_ctx *ctx, const double *d_x, int64_t n, const double *d_mark_i, const double *d_mark_q,
                            const double *d_space, int groups, int m, double *d_y, int64_t y_stride);

/* Gain sweep: `groups` (<= 8) AFSK modems over the same band-passed stream that share their mark correlators and whose space
 * correlators are ONE unit pair (d_unit_i/q, space_gain 1.0) scaled by each modem's space_gain (afsk.py:144-145), all with the same
 * output low-pass: writes every modem's slicer sign bitmap (what afsk.py:148-167 followed by `>= 0` gives) from two correlator
 * pairs and two low-passes for the whole sweep.  The bitmaps are certified: a sample whose sign the shared computation cannot
 * guarantee is recomputed by the exact chain of that modem (d_space = [groups][2][m], the modems' own space taps), and if there are
 * more such samples than a fixed list holds (degenerate input) the exact chains of all modems run instead, decided on the device:
 * every bit equals pm_afsk_correlate + pm_fir_signs_f64 for that modem, and the call never waits for the GPU.  x_bound: a bound on
 * |d_x| the caller vouches for (sum|input_bpf| * 32768 for int16 audio); lpf_abs_sum = sum|output_lpf|.
 * pm_afsk_sweep_last (diagnostics; waits for the stream): how many samples the last sweep on this ctx could not certify (above
 * 65536 the exact chains ran). */
int pm_afsk_sweep_signs(pm_ctx *ctx, const double *d_x, int64_t n, double x_bound, const double *d_mark_i, const double *d_mark_q,
                        const double *d_unit_i, const double *d_unit_q, const double *d_space, const double *h_gains, int groups, int m,
                        const double *d_lpf, int ml, double lpf_abs_sum, uint64_t *const *h_bits);
/* The same sweep when the correlator templates are what afsk.py:134-144 makes them, the powers of one rotation per tone
 * (template[j] = cos / sin of j*w): the two magnitude streams come from a sliding sum, Z(k+1) = x[k+m] + r Z(k) - r^m x[k], 6 fused
 * operations per tone and sample instead of 2m, restarted from the direct sum every 16 outputs.  Nothing changes in what is
 * guaranteed: the sliding sums only feed the certified decision, whose bound grows by their (small) error, and every bit still
 * equals the exact chain's.  rot = r = (template_i[1], template_q[1]); end = r^m rounded to double; tap_dev = the largest
 * |template[j] - r^j| over both templates of both tones, measured by the host in extended precision (pymodem_amd.taps.tone_model).
 * Fails with PM_ERR_ARG if tap_dev >= 1e-6: such templates are not tones, use pm_afsk_sweep_signs.  groups == 1 is the lone
 * chain: its mark - gain * space difference takes ONE low-pass.  Sliding sums, low-pass(es) and the certified combine run as one
 * kernel (nothing but the bitmaps is written) unless the low-pass is longer than 2049 taps. */
typedef struct pm_afsk_tones {
    double mark_rot[2], mark_end[2];
    double space_rot[2], space_end[2];     /* of the unit-gain space templates */
    double tap_dev;
} pm_afsk_tones;
int pm_afsk_sweep_signs_tones(pm_ctx *ctx, const double *d_x, int64_t n, double x_bound, const double *d_mark_i, const double *d_mark_q,
                              const double *d_unit_i, const double *d_unit_q, const double *d_space, const double *h_gains, int groups,
                              int m, const double *d_lpf, int ml, double lpf_abs_sum, uint64_t *const *h_bits,
                              const pm_afsk_tones *h_tones);
/* The two magnitude streams the sweep works from, on their own (tests, diagnostics): d_mark_mag[k] = sqrt(I^2 + Q^2) of the mark
 * templates at output k, d_space_mag likewise (n - m + 1 values each; afsk.py:153-160 without the subtraction).  h_tones NULL:
 * the direct sums in the reference's order.  h_tones given: the sliding sums; *h_bound receives the bound on their distance
 * from the direct sums that the certified decision uses (0 for the direct sums). */
int pm_afsk_magnitudes(pm_ctx *ctx, const double *d_x, int64_t n, double x_bound, const double *d_mark_i, const double *d_mark_q,
                       const double *d_space_i, const double *d_space_q, int m, const pm_afsk_tones *h_tones, double *d_mark_mag,
                       double *d_space_mag, double *h_bound);
int pm_afsk_sweep_last(pm_ctx *ctx, int64_t *h_uncertain);
/* Deferred fallback for pipelined hosts.  By default every certified sweep ends with three launches that look at its counter of
 * uncertain samples and -- only if the list overflowed (65536: digital silence, input far below the stated bound) -- run the exact
 * chains after all; in the normal case they leave at once but still cost the stream three dispatches per sweep.  With
 * pm_afsk_sweep_mode(ctx, 1) they are not enqueued: take a ticket after the call, and once the sweep has FINISHED (stream or event
 * synchronised) ask pm_afsk_sweep_result; if *h_uncertain > *h_capacity the sweep's bitmaps are not valid and the caller runs the
 * exact path for those modems (pm_afsk_correlate + pm_fir_signs_f64).  Tickets stay valid for 63 further sweeps on the context. */
/* WHO COUNTS WHERE.  The entry points of this block -- pm_afsk_sweep_signs, pm_afsk_sweep_signs_tones, pm_afsk_group_run,
 * pm_afsk_sweep_last / _ticket / _result / _results -- count a sweep's uncertain samples in a ring of 64 counters that belongs to the
 * CONTEXT (a ticket is a sequence number into it).  They are for one thread per context: submitting sweeps on a context from one
 * thread and asking for their results from another is not supported, and a result must be asked for within 63 further sweeps.
 * The pipelined executor (pm_pipe_*) does NOT use the ring: each of its recordings owns its counters, mailbox words and overflow
 * list from submission until its slicer batch has read them (csrc/pm_common.h: pm_sweep_cells), its matrix-pipe sweeps decide their
 * uncertain samples inside the workgroup that found them, and nothing is shared between its threads. */
/* One chain group's demod stage in one call: band-pass on the int16 audio into d_bpf_out (n - mb + 1 doubles), then every sweep of
 * h_sweeps on it (as pm_afsk_sweep_signs / _tones would run them, fallback deferred), h_tickets[k] = the ticket of sweep k. */
typedef struct pm_afsk_sweep_desc {
    const double *d_mark_i, *d_mark_q, *d_unit_i, *d_unit_q, *d_space;
    const double *h_gains;
    int32_t groups, m;
    const double *d_lpf;
    int32_t ml, reserved;
    double lpf_abs_sum;
    uint64_t *const *h_bits;
    const pm_afsk_tones *h_tones;      /* NULL: direct correlator sums */
} pm_afsk_sweep_desc;
int pm_afsk_group_run(pm_ctx *ctx, const int16_t *d_audio, int64_t n, const double *d_bpf, int mb, double *d_bpf_out, double x_bound,
                      const pm_afsk_sweep_desc *h_sweeps, int nsweeps, int64_t *h_tickets);
int pm_afsk_sweep_mode(pm_ctx *ctx, int deferred);
int pm_afsk_sweep_ticket(pm_ctx *ctx, int64_t *h_ticket);
int pm_afsk_sweep_result(pm_ctx *ctx, int64_t ticket, pm_ctx *via, int64_t *h_uncertain, int64_t *h_capacity);
int pm_afsk_sweep_results(pm_ctx *ctx, const int64_t *h_tickets, int n, pm_ctx *via, int64_t *h_uncertain, int64_t *h_capacity);   /* several sweeps of one context, one copy */   /* via: the context whose stream carries the 4-byte copy (NULL: ctx) */

/* Sign bitmap of a float64 stream: bit k of the little-endian uint64 array = (x[k] >= 0), the only
 * property of a sample the slicers read (slicer.py:85,99-102,210-232).  d_bits holds (n+63)/64 words. */
int pm_signs_f64(pm_ctx *ctx, const double *d_x, int64_t n, uint64_t *d_bits);

/* ---- AGC and carrier loops -------------------------------------------------------------------- */
typedef struct pm_agc_params {       /* AGC.__init__, agc.py:7-24 */
    double attack_rate, decay_rate, sustain_time, sample_rate, target_amplitude;
} pm_agc_params;
/* AGC.apply in place (agc.py:61-80): normal = max(buf), envelope follower, buf[i] = target*s/env.
 * h_state[2] = {envelope, sustain_count}, read and written (carried like self.* in the reference). */
int pm_agc_apply(pm_ctx *ctx, double *d_buf, int64_t n, const pm_agc_params *h_params, double *h_state);

/* The envelope follower and normalisation of AGC.apply (agc.py:68-80) for `rows` streams, each CONTINUED from h_state[2r..] =
 * {envelope, sustain_count} over n more samples, out of place (y may be x): what a recording processed in time chunks needs, where
 * `normal` = max(buffer) over the WHOLE buffer (agc.py:67) is known before the first chunk (h_normal[r]; pm_rows_max_f64 gives the
 * maxima of the rows of a chunk).  Pieces of a buffer run through this one after the other give the bits of pm_agc_apply on the
 * whole.  One lane per row (64 rows per wave) steps the recurrence and divides. */
int pm_agc_rows_apply(pm_ctx *ctx, const double *d_x, int64_t x_stride, double *d_y, int64_t y_stride, int rows, int64_t n,
                      const pm_agc_params *h_params, const double *h_normal, double *h_state);
int pm_rows_max_f64(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, double *h_max);

typedef struct pm_loop {             /* one carrier loop = NCO (nco.py) + IIR_1 (iir.py) + PI (pi_control.py) */
    double phase_scaling;            /* 2*pi / sample_rate                       nco.py:31 */
    double index_scaling;            /* 256 / (2*pi)                             nco.py:27 */
    double set_frequency;            /* carrier_freq                             nco.py:13 */
    double b0, b1, a1;               /* gain*b0, gain*b1, a1                     iir.py:15-29 */
    double p_rate, i_rate, i_limit, gain;                                     /* pi_control.py:8-12 */
    /* state, read and written */
    double phase, control, sine, cosine;
    double x0, x1, y0;
    double integral, proportional;
    /* QPSK Costas loop only (psk.py:223-240): the two branch low-pass filters share coefficients; their states */
    double bb0, bb1, ba1;            /* Cosine_LPF / Sine_LPF coefficients       iir.py:15-29 */
    double cx0, cx1, cy0;            /* Cosine_LPF state */
    double sx0, sx1, sy0;            /* Sine_LPF state */
} pm_loop;

/* nloops independent loops over the SAME input (chains that differ only in carrier_freq, e.g.
 * configs/qpsk_2400.json) or, with x_stride != 0, over nloops inputs x + l*x_stride.  One lane per loop.
 * d_table: the 256-entry wavetable amplitude*sin(2*pi*i/256) computed by the caller with libm sin (nco.py:22-24).
 * Outputs are laid out [loop][n] with stride out_stride (elements). */
int pm_costas_bpsk(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table,
                   const double *d_x, int64_t x_stride, int64_t n, double *d_out, int64_t out_stride);      /* psk.py:173-189 */
int pm_pll_afsk(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table,
                const double *d_x, int64_t x_stride, int64_t n, double *d_out, int64_t out_stride);         /* afsk_pll.py:153-165 */
/* d_pd_table: int32[64*64] phase-detector table, row-major [real][imag] (phase_detector.py:36-44). */
int pm_mpsk_loop(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table, const int32_t *d_pd_table,
                 const double *d_re, const double *d_im, int64_t x_stride, int64_t n,
                 double *d_i_out, double *d_q_out, int64_t out_stride);                                     /* psk.py:734-747 */

/* QPSKModem's Costas loop (psk.py:434-467): both mixer products are low-passed (Cosine_LPF, Sine_LPF); the phase detector is
 * cos_lp*sgn(sin_lp) - sin_lp*sgn(cos_lp).  d_i_out receives the SINE branch and d_q_out the COSINE branch, as the reference
 * appends them (psk.py:452-453). */
int pm_costas_qpsk(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table,
                   const double *d_x, int64_t x_stride, int64_t n, double *d_i_out, double *d_q_out, int64_t out_stride);

/* ---- slicers ----------------------------------------------------------------------------------
 * Symbol-timing PLL + bit decision + byte packing.  Input is the sign bitmap(s) of the demodulated
 * stream.  Output: bytes and their 1-based stream addresses (index of the sample that completed the
 * byte), exactly the AddressedData list of the reference.  Evaluated chunk-parallel: one walker per chunk runs the
 * reference's recurrence, leaves a clock checkpoint per 64-sample word and walks on into the next chunks until it meets the
 * trail in front of it; when no walker is left the result is bitwise the sequential run's (pm_slicer.hip). */
typedef struct pm_slicer_params {
    double samples_per_symbol;       /* sample_rate / symbol_rate              slicer.py:51 */
    double lock_rate;                /*                                         slicer.py:22-33,124-165 */
    int32_t bits_per_symbol;         /* 1 (binary, bpsk) or 2 (qpsk)            slicer.py:127-158 */
    int32_t state_mask;              /* 0x3 or 0xF                              slicer.py:126-157 */
    int32_t demap[16];               /* symbol demap table                      slicer.py:128 */
} pm_slicer_params;
/* h_count receives the number of bytes produced; if it exceeds cap the call returns PM_ERR_CAPACITY
 * (the first cap entries are valid).  A safe cap is n*bits_per_symbol/8 + 1. */
int pm_slice_binary(pm_ctx *ctx, const uint64_t *d_bits, int64_t n, const pm_slicer_params *h_params,
                    uint8_t *d_data, int64_t *d_addr, int64_t cap, int64_t *h_count);                       /* slicer.py:59-107 */
int pm_slice_quadrature(pm_ctx *ctx, const uint64_t *d_bits_i, const uint64_t *d_bits_q, int64_t n,
                        const pm_slicer_params *h_params, uint8_t *d_data, int64_t *d_addr, int64_t cap,
                        int64_t *h_count);                                                                  /* slicer.py:193-242 */
/* Many streams in one launch sequence (chains are independent; batching them shares the iteration launches and the
 * host checks).  d_bits_q == NULL selects the binary slicer for that stream.  `count` is written per job. */
/* What a slicer object carries from one slice() call to the next (slicer.py:49-56,193-202): a second call on the same object
 * continues the first, mid-byte if need be.  All zero (last samples counting as >= 0) is the just-tuned state. */
typedef struct pm_slicer_state {
    double phase_clock;
    int32_t last_i_negative;         /* 1 if the previous call's last I (or only) sample was < 0 */
    int32_t last_q_negative;
    int32_t working_byte;            /* the bits shifted in since the last emitted byte (low working_bits bits are meaningful) */
    int32_t working_bits;
    int32_t state_register;          /* quadrature: previous symbol(s), slicer.py:210 */
    int32_t reserved;
    int64_t streamaddress;           /* samples seen so far: the next sample's address is streamaddress + 1 */
} pm_slicer_state;

typedef struct pm_slice_job {
    const uint64_t *d_bits_i;
    const uint64_t *d_bits_q;
    int64_t n;
    pm_slicer_params params;
    uint8_t *d_data;
    int64_t *d_addr;
    int64_t cap;
    int64_t count;                   /* out */
    pm_slicer_state *h_state;        /* in/out; NULL = start from the just-tuned state and do not report the end state */
} pm_slice_job;
int pm_slice_batch(pm_ctx *ctx, pm_slice_job *h_jobs, int njobs);           /* njobs <= 64 */
/* The output of a finished pm_slice_batch in the form that is cheapest to bring to the host (3 bytes per data byte instead of 9, no
 * unused capacity in between): for job j, at d_block + h_offsets[j]: {first address, last address} (2 x int64), 64 flag bytes,
 * count[j] address steps as uint16 (address[i] - address[i-1], the first one 0; padded to a multiple of 8 bytes), count[j] data
 * bytes (padded likewise) -- PM_COMPACT_HEAD = 80 bytes before the steps.  *h_used = bytes written; PM_ERR_CAPACITY (with *h_used
 * set) if block_bytes is less.  A step is eight symbol periods unless the input keeps the clock from its threshold; if any of the
 * flag bytes is non-zero a step of that stream did not fit 16 bits: take its addresses from its d_addr instead.  Runs on the ctx
 * stream, after the batch; h_jobs are the batch's jobs as pm_slice_batch left them (count filled in). */
#define PM_COMPACT_HEAD 80
int pm_slice_compact(pm_ctx *ctx, const pm_slice_job *h_jobs, int njobs, void *d_block, size_t block_bytes, int64_t *h_offsets,
                     size_t *h_used);
/* How many chunks (= walkers) a batch is cut into on this ctx, within 1024..16384 samples per chunk; 0 restores the default 16384.
 * Lane-steps are N (1 + m/L) for merge length m (10-20 k samples) and chunk length L, the depth is (L + longest merge) x the
 * step time: long chunks are cheap, short ones are quick.  Results do not depend on it. */
int pm_slicer_tune(pm_ctx *ctx, int64_t target_lanes);
/* The longest chunk, in 64-sample words (0 restores the default 384 = 24.6 k samples: the optimum when a batch is a few dozen streams
 * and its depth matters, as in the pipelined executor).  A host that slices thousands of streams after a run of the batch engine has
 * its parallelism in the streams and several calls in flight: long chunks (thousands of words) do N (1 + m/L) lane-steps with m/L
 * close to nothing instead of 0.6.  Results do not depend on it. */
int pm_slicer_limits(pm_ctx *ctx, int64_t max_chunk_words);
/* Diagnostics of the last slicer call on this ctx: lockstep launches until no walker was left, chunk length, chunks. */
int pm_slicer_stats(pm_ctx *ctx, int32_t *iterations, int32_t *chunk_len, int64_t *chunks);

/* ---- whole chain: modem + slicer in one call ---------------------------------------------------
 * chain_execute.process_chain up to and including slicer.slice (chain_execute.py:8-14) for one demod_chain: the stage entry points
 * above, strung together in the order of the reference's demod() (afsk.py:148-167, fsk.py:149-159, psk.py:162-195, psk.py:705-773,
 * afsk_pll.py:140-170) with every intermediate kept on the device.  The host designs the taps (firwin / RRC / Hilbert / tones) and
 * hands them over; pointers inside the desc are read during pm_chain_create only.  A chain carries the AGC envelope and the
 * carrier-loop state from one pm_chain_run to the next, like the reference's stage objects; pm_chain_reset returns it to the
 * just-created state.  Output: the slicer's bytes and 1-based stream addresses; when more than `cap` were produced the call
 * returns PM_ERR_CAPACITY with the required size in *h_count -- the run HAS happened (AGC, loop and slicer state have moved on
 * with the stream) and its output is kept on the device: call pm_chain_fetch with buffers of that size, do not run again. */
enum { PM_MODEM_AFSK = 0, PM_MODEM_FSK = 1, PM_MODEM_BPSK = 2, PM_MODEM_MPSK = 3, PM_MODEM_AFSK_PLL = 4, PM_MODEM_QPSK = 5 };
#define PM_CHAIN_INVERT 1       /* FSK: negate the filter output (fsk.py:153-154) */
/* Opt-in: every FIR of the chain continues from the tail of its own input stream instead of starting afresh with every run as the
 * reference's per-call numpy.convolve(..., 'valid') does (afsk.py:151-166, fsk.py:151, psk.py:165,193,710-751, afsk_pll.py:143,168).
 * AFSK / FSK (FIRs and pointwise operations only): the last sum(M - 1) input samples of a run go in front of the next run's input, and
 * a recording fed in pieces gives the bytes, addresses and packets of the single call on the whole.  Carrier-loop
 * modems (BPSK, MPSK, QPSK, AFSK-PLL): band-pass, Hilbert pair and matched / output filter each keep M - 1 samples of their input
 * (on the device); AGC envelope and loop registers are carried anyway; AGC.apply still normalises by the maximum of each run's
 * band-passed samples (agc.py:67), so pieces are what the reference's primitives give when fed this way, not the single call.
 * A run that does not bring a stage up to its filter length produces nothing yet (*h_count = 0). */
#define PM_CHAIN_CARRY_HISTORY 2
typedef struct pm_chain_desc {
    int32_t modem;                                   /* PM_MODEM_* */
    int32_t flags;
    const double *input_fir;  int32_t n_input_fir;   /* input_bpf (afsk, bpsk, mpsk, afsk_pll) or input_lpf (fsk) */
    const double *mark_i, *mark_q, *space_i, *space_q; int32_t n_corr;      /* afsk */
    const double *hilbert;    int32_t n_hilbert, hilbert_delay;             /* mpsk: Hilbert taps; delay = (n_hilbert-1)/2 */
    const double *output_fir; int32_t n_output_fir;  /* output_lpf (afsk, afsk_pll) or the RRC matched filter (bpsk; mpsk both arms) */
    int32_t use_agc;          pm_agc_params agc;     /* bpsk, mpsk, afsk_pll */
    pm_loop loop;                                    /* carrier loop parameters and initial state */
    const double *wavetable;                         /* 256 entries (nco.py:22-24) */
    const int32_t *pd_table;                         /* 64 x 64 (phase_detector.py:36-44), mpsk */
    int32_t quadrature;                              /* 1: QuadratureSlicer (mpsk, qpsk), 0: BinarySlicer */
    pm_slicer_params slicer;
} pm_chain_desc;
typedef struct pm_chain pm_chain;
int pm_chain_create(pm_ctx *ctx, const pm_chain_desc *desc, pm_chain **out);
int pm_chain_run(pm_chain *chain, const int16_t *audio, int64_t n, int audio_on_device,
                 uint8_t *h_data, int64_t *h_addr, int64_t cap, int64_t *h_count);
int pm_chain_fetch(pm_chain *chain, uint8_t *h_data, int64_t *h_addr, int64_t cap, int64_t *h_count);   /* the last run's output, again */
int pm_chain_reset(pm_chain *chain);
int pm_chain_destroy(pm_chain *chain);

/* ---- batch engine for the carrier-loop modems: many recordings x chains in flight -----------------
 * A carrier loop (psk.py:173-189, psk.py:734-747, afsk_pll.py:153-165) is one dependent chain per sample: a GPU lane cannot make it
 * faster than a host core, it can only run hundreds at once.  pm_lbatch runs `recordings` recordings of equal length through
 * `chains` chains that share their front end (band-pass, AGC, Hilbert pair: the chains of configs/qpsk_2400.json differ in
 * carrier_freq only) in time chunks, all recordings x chains loops in ONE launch per chunk, every sequential state (AGC envelope,
 * loop registers, FIR histories) carried in device memory from chunk to chunk.  Output: the sign bitmap(s) of every chain's
 * demodulated stream (what modem.demod() followed by `>= 0` gives: psk.py:162-195, psk.py:705-773, afsk_pll.py:140-170,
 * psk.py:426-476), stream s = recording * chains + chain at d_bits_i + s * bits_stride words (and d_bits_q for the quadrature
 * modems), ready for pm_slice_batch.  Bit-identical to the per-recording entry points for every chunk length. */
typedef struct pm_lbatch_desc {
    int32_t modem;                                   /* PM_MODEM_BPSK | PM_MODEM_MPSK | PM_MODEM_AFSK_PLL | PM_MODEM_QPSK */
    int32_t recordings;                              /* most recordings a run will bring */
    int32_t chains;                                  /* carrier loops per recording */
    int32_t chunk;                                   /* final-filter outputs per chunk (rounded up to 2048s; 0 = 262144) */
    const double *input_fir;  int32_t n_input_fir;   /* input_bpf */
    const double *hilbert;    int32_t n_hilbert, hilbert_delay;             /* mpsk */
    const double *output_fir; int32_t n_output_fir;  /* RRC matched filter (bpsk, qpsk; mpsk both arms) or output_lpf (afsk_pll) */
    pm_agc_params agc;
    const pm_loop *loops;                            /* `chains` loops: parameters and initial state (the same for every recording) */
    const double *wavetable;                         /* 256 entries (nco.py:22-24) */
    const int32_t *pd_table;                         /* 64 x 64 (phase_detector.py:36-44), mpsk */
} pm_lbatch_desc;
typedef struct pm_lbatch pm_lbatch;
int pm_lbatch_create(pm_ctx *ctx, const pm_lbatch_desc *desc, pm_lbatch **out);      /* pointers inside desc are read here only */
/* samples per demodulated stream, outputs per chunk and chunks for recordings of n samples */
int pm_lbatch_geometry(pm_lbatch *batch, int64_t n, int64_t *h_nout, int64_t *h_chunk, int64_t *h_chunks);
/* h_d_audio: HOST array of `recordings` device pointers to int16 recordings of n samples each (they may be the same buffer).
 * Everything is enqueued (three streams: the context's and two of the engine's own); nothing waits for the GPU: the bitmaps are
 * complete when the context's stream has reached the end of the call.  Every run starts from fresh AGC and loop states.
 * bits_stride >= (nout + 63) / 64 + 1 words. */
int pm_lbatch_run(pm_lbatch *batch, const int16_t *const *h_d_audio, int recordings, int64_t n, uint64_t *d_bits_i, uint64_t *d_bits_q,
                  int64_t bits_stride, int64_t *h_nout);
/* The same run with the slicers inside it (slicer.py:59-107, :193-242): every stream (recording r, chain c: row r * chains + c) is
 * sliced chunk by chunk behind its matched filter by a lane of its own that carries the slicer's state through the run, so no sign
 * bitmap of a whole recording exists (16 384 streams of ten minutes: 59 GB each for I and Q) and nothing is left to slice when the run
 * ends.  h_params: one parameter set per chain; every stream starts from the just-tuned slicer state.  Output, per row: up to `cap`
 * (a multiple of 8) data bytes at d_data + row * cap, their address steps (address[i] - address[i-1], the first 0) at d_steps + row * cap,
 * and the row's record: count (bytes produced -- beyond cap they are counted, not stored, and flags has bit 1), the first and last
 * address, the slicer's end state (pm_slicer_state's fields) and flags (bit 0: a step did not fit 16 bits -- slice that run the
 * other way).  Results equal pm_lbatch_run + pm_slice_batch on every stream, byte for byte and address for address. */
typedef struct pm_rowslice_rec {
    int64_t count, first_addr, last_addr, seen;      /* seen: samples consumed (streamaddress) */
    double clk;                                      /* phase_clock */
    int32_t li_neg, lq_neg, wbyte, wbits, sreg, flags;
} pm_rowslice_rec;
int pm_lbatch_run_sliced(pm_lbatch *batch, const int16_t *const *h_d_audio, int recordings, int64_t n, const pm_slicer_params *h_params, int nparams,
                         uint8_t *d_data, uint16_t *d_steps, int64_t cap, pm_rowslice_rec *d_recs, int64_t *h_nout);
/* Rows [row0, row0 + nrows) of such a run as one dense block for the way to the host: row k at the sum of the sizes before it, its
 * min(count, cap) steps (padded to 8 bytes) then its data bytes (padded to 8).  On the ctx stream; nrows <= 4096. */
int pm_rows_gather(pm_ctx *ctx, const pm_rowslice_rec *d_recs, const uint8_t *d_data, const uint16_t *d_steps, int64_t cap, int64_t row0, int nrows,
                   void *d_block, size_t block_bytes);
pm_ctx *pm_lbatch_front_ctx(pm_lbatch *batch);       /* the engine's own contexts, for pm_prof_*: band-pass, AGC, Hilbert of chunk t + 1 ... */
pm_ctx *pm_lbatch_tail_ctx(pm_lbatch *batch);        /* ... and the matched filters of chunk t - 1, beside the loops of chunk t on the caller's */
pm_ctx *pm_lbatch_loop_ctx(pm_lbatch *batch);        /* ... or on the engine's loop context, when the loops have compute units of their own (else NULL) */
pm_ctx *pm_lbatch_slice_ctx(pm_lbatch *batch);       /* ... and the slicers of a sliced run, beside the matched filters of the next chunk */
int pm_lbatch_destroy(pm_lbatch *batch);

/* ---- host-integer stages (native C++, no GPU) --------------------------------------------------
 * These consume the slicer's byte stream; they are bit-serial state machines over KBs of data. */
/* LFSR.stream_unscramble_8bit (lfsr.py:22-52).  *h_shift_register is read and written.  h_out must not alias h_in. */
int pm_lfsr_unscramble(const uint8_t *h_in, int64_t n, uint64_t poly, int invert, uint64_t *h_shift_register, uint8_t *h_out);

/* Packet record shared by the codecs and the de-dup (PacketMeta, packet_meta.py:178-195).  IL2P packets are at most 1023 + 2 bytes
 * and AX.25 frames between flags likewise on any signal; the reference's AX.25 decoder can however close a frame of any length after
 * a long stretch without a flag (its byte counter wraps at 1023, the collected bytes stay: ax25.py:41-47).  Such a frame's row holds
 * its first PM_PKT_MAX bytes (len = PM_PKT_MAX); its CRC fields and valid_crc are those of the whole frame. */
#define PM_PKT_MAX 1280
typedef struct pm_packet {
    int64_t streamaddress;
    int32_t len;
    int32_t bytes_corrected;
    int32_t calculated_crc, carried_crc;
    int32_t valid_crc, valid_header;
    int32_t source_decoder;          /* chain index */
    int32_t correlated_count;        /* filled by pm_correlate */
    uint8_t data[PM_PKT_MAX];
} pm_packet;

/* The first 40 bytes of pm_packet: everything the de-dup reads.  Arrays of heads stand in for arrays of full rows wherever a
 * stride is passed (pm_correlate_strided), so rank 0 of a multi-GPU job never expands the payloads it gathers. */
typedef struct pm_packet_head {
    int64_t streamaddress;
    int32_t len;
    int32_t bytes_corrected;
    int32_t calculated_crc, carried_crc;
    int32_t valid_crc, valid_header;
    int32_t source_decoder;
    int32_t correlated_count;
} pm_packet_head;

typedef struct pm_codec pm_codec;
/* kind 0 = AX25Codec (ax25.py:11-93), 1 = IL2PCodec (il2p.py:110-519). */
int pm_codec_create(int kind, int crc, int disable_rs, int min_dist, int sync_tol, int source_decoder, pm_codec **out);
int pm_codec_destroy(pm_codec *c);
/* The index pm_codec_fetch writes into pm_packet.source_decoder from now on (the chain's place in the config: what the reference's
 * SourceDecoder name stands for, packet_meta.py:181). */
int pm_codec_set_source(pm_codec *c, int32_t source_decoder);
/* Feed n descrambled bytes with their stream addresses (state carries over between calls, like the reference's
 * codec objects); decoded packets queue inside the codec, *h_pending = packets waiting.  pm_codec_fetch moves up to
 * cap of them out, oldest first, with CRC and header validity filled (packet_meta.py:197-208). */
int pm_codec_decode(pm_codec *c, const uint8_t *h_data, const int64_t *h_addr, int64_t n, int64_t *h_pending);
int pm_codec_fetch(pm_codec *c, pm_packet *h_out, int64_t cap, int64_t *h_count);

/* The host half of a whole chain group in two calls (chain_execute.py:20-26 for every chain of the group at once): per job
 * pm_lfsr_unscramble into a library-owned buffer, then pm_codec_decode; per codec pm_codec_fetch into consecutive row blocks
 * of h_out (counts[j] rows each).  One chain per task on up to `threads` threads (the caller's included) that belong to the
 * library, so a language binding crosses the boundary twice per recording, not 3 x chains times.  No two jobs may share a
 * codec.  The first failing job's code is returned (its status field holds it too). */
typedef struct pm_host_job {
    pm_codec *codec;
    const uint8_t *h_data;        /* the slicer's bytes ...                       */
    const int64_t *h_addr;        /* ... and their stream addresses               */
    int64_t n;
    uint64_t lfsr_poly;
    uint64_t lfsr_state;          /* in/out: LFSR.shift_register                  */
    int64_t pending;              /* out: packets waiting in the codec            */
    int32_t lfsr_invert;
    int32_t status;               /* out */
    const uint16_t *h_addr_delta; /* with h_addr == NULL: the addresses in pm_slice_compact's form, address[i] = addr_first +  */
    int64_t addr_first;           /* delta[0] + ... + delta[i] (delta[0] = 0)                                                   */
    uint8_t *h_plain;             /* NULL, or n bytes: the LFSR's output is written here (and decoded from here) instead of a    */
                                  /* library-owned buffer -- for callers that want to see what the codec saw                    */
} pm_host_job;
int pm_host_decode_batch(pm_host_job *jobs, int njobs, int threads);
int pm_codec_fetch_batch(pm_codec *const *codecs, const int64_t *counts, int n, pm_packet *h_out, int threads);
int pm_crc16_ccitt(const uint8_t *h_data, int64_t n);                    /* crc_functions.py:44-55 */

/* Wire form of packet rows for the one exchange step of the multi-GPU path (the reference hands PacketMeta lists through a
 * multiprocessing.Queue, pymodem.py:140,157-163): per packet the 40-byte record header followed by its `len` payload bytes.
 * pm_packets_pack returns the bytes the n rows need and writes them when they fit in cap (returns the need either way);
 * pm_packets_unpack rebuilds full rows (payload tails zeroed) and returns how many it wrote, or < 0 on a malformed stream. */
int64_t pm_packets_pack(const pm_packet *h_rows, int64_t n, uint8_t *h_out, int64_t cap);
int64_t pm_packets_unpack(const uint8_t *h_in, int64_t bytes, pm_packet *h_rows, int64_t cap_rows);

/* Split a wire stream without expanding it: the record heads go to h_heads (dense), h_payload_at[k] is the offset of record
 * k's payload inside h_in.  Returns the number of records or < 0. */
int64_t pm_packets_index(const uint8_t *h_in, int64_t bytes, pm_packet_head *h_heads, int64_t *h_payload_at, int64_t cap_rows);

/* PacketMetaArray.Correlate (packet_meta.py:230-271): h_pkts hold all chains' packets in config order
 * (h_chain_counts[c] packets for chain c).  Writes indices of the unique packets, sorted by stream address,
 * to h_unique_idx and sets correlated_count on them.  Returns the number of unique packets or < 0. */
int64_t pm_correlate(pm_packet *h_pkts, const int64_t *h_chain_counts, int nchains, double address_distance,
                     int64_t *h_unique_idx, int32_t *h_corr_decoders, int64_t corr_cap);
/* Same on records `stride` bytes apart that each begin with a pm_packet_head (stride = sizeof(pm_packet) for full rows,
 * sizeof(pm_packet_head) for a dense array of heads). */
int64_t pm_correlate_strided(void *h_records, int64_t stride, const int64_t *h_chain_counts, int nchains, double address_distance,
                             int64_t *h_unique_idx, int32_t *h_corr_decoders, int64_t corr_cap);

/* ---- pipelined executor for an AFSK chain group: one call per recording -------------------------
 * What pymodem.py:140-163 does with a process per chain and a queue -- every chain of the config on the same recording, the
 * packets of all of them de-duplicated -- as a pipeline over RECORDINGS that lives entirely inside the library:
 *   pm_pipe_submit   launches the recording's demod stage on one of the demod streams -- for an AFSK group of up to two certified
 *                    sweeps ONE kernel: band-pass, every sweep and the exact chain for whatever it cannot certify, the int16 audio read
 *                    once and one bit per sample and chain written (csrc/pm_fir.hip: afsk_fused8_kernel; other groups: band-pass + a
 *                    launch per sweep) -- records an event, returns; blocks only while all bitmap slots are in use
 *   slicer threads   (own high-priority streams) wait for the event, take up to `slice_group` consecutive recordings, work off what a
 *                    sweep left on its list (normally nothing; an overflowed list: the exact kernels), run pm_slice_batch +
 *                    pm_slice_compact, whose output the kernel writes into a page-locked host block
 *   host threads     pm_host_decode_batch + pm_codec_fetch_batch (LFSR + AX.25/IL2P per chain, fresh stage objects per recording
 *                    as chain_builder.py makes them) and pm_correlate over the chains in config order
 *   pm_pipe_wait     the recording's packet rows, per-chain counts and the de-dup result, in the library's memory until
 *                    pm_pipe_release
 * Results equal process_chain on every chain + PacketMetaArray.Correlate, recording by recording.  One submitting thread.  The
 * device pointers inside the descs (taps) must stay valid for the pipeline's life; host arrays are copied by pm_pipe_create. */
typedef struct pm_pipe_fir {         /* a sign-FIR group: sign(FIR(int16 audio)) as one bitmap (pm_fir_signs_i16) -- FSKModem.demod, fsk.py:149-159 */
    const double *d_taps;
    int32_t m, flags;                /* PM_FIR_NEGATE: the modem's `invert` */
} pm_pipe_fir;
typedef struct pm_pipe_chain {
    int32_t sweep, slot;             /* which sweep of pm_pipe_desc.sweeps demodulates this chain, and its place in it (gain index);
                                        sweep = -(f + 1): the chain slices the bitmap of sign-FIR group f (slot ignored) */
    pm_slicer_params slicer;         /* binary slicer */
    uint64_t lfsr_poly;              /* lfsr.py:10-20 */
    int32_t lfsr_invert;
    int32_t codec_kind, crc, disable_rs, min_dist, sync_tol;   /* pm_codec_create */
    int32_t source_decoder;          /* the chain's place in the config */
} pm_pipe_chain;
typedef struct pm_pipe_desc {
    const double *d_bpf; int32_t mb; /* the group's shared input_bpf (device) */
    int32_t nsweeps;
    double x_bound;                  /* sum|input_bpf| * 32768 */
    const pm_afsk_sweep_desc *sweeps;/* h_bits ignored: the pipeline owns the bitmaps */
    const pm_pipe_chain *chains;     /* config order */
    int32_t nchains;
    int32_t slots;                   /* recordings between demod and slicer at most (0 = 16) */
    int32_t slice_workers;           /* 0 = 2 */
    int32_t slice_group;             /* most recordings per slicer batch (0 = 4) */
    int32_t slice_min_group;         /* a batch waits for this many recordings while later ones are queued (0 = slice_group) */
    int32_t demod_streams;           /* recordings take turns on this many demod streams: the context's own and further ones of the
                                        pipeline's, each with its own band-passed stream and sweep state (0 = 2) */
    int32_t host_threads;            /* recordings in the host stage at once (0 = min(12, 36 / chains), at least 2) */
    int32_t decode_threads;          /* threads inside one recording's host stage (0 = one per chain) */
    double address_distance;         /* PacketMetaArray.Correlate (packet_meta.py:230); < 0: no de-dup here (the chains are a part of
                                        the config: the rows go to the exchange, rank 0 de-duplicates), unique = 0 */
    int64_t max_samples;             /* longest recording */
    const pm_pipe_fir *firs;         /* sign-FIR groups (may be the only demodulators: nsweeps = 0, d_bpf / sweeps unused) */
    int32_t nfirs;
    int32_t keep_slices;             /* != 0: every finished recording keeps its slicers' bytes + addresses and the LFSR output of
                                        every chain for pm_pipe_slices (parity tests of the bitstream, slicer.py:59-107) */
} pm_pipe_desc;
typedef struct pm_pipe_result {
    int64_t ticket;
    int32_t status, reserved;        /* PM_OK or the stage error (pm_pipe_wait returns it too, message in pm_last_error) */
    int64_t rows;                    /* packets of all chains */
    const pm_packet *h_rows;         /* chain by chain, config order; correlated_count set on the unique ones */
    const int64_t *h_counts;         /* [nchains] */
    int64_t unique;
    const int64_t *h_unique_idx;     /* [unique] row indices by stream address */
    const int32_t *h_corr_decoders;  /* correlated decoders, consecutive runs of correlated_count per unique packet */
    double ms_to_demod_done, ms_to_sliced, ms_to_done;   /* from submit, host clock */
    double done_at_ms;               /* when the recording left the last stage, host clock since pm_pipe_create */
} pm_pipe_result;
typedef struct pm_pipe pm_pipe;
int pm_pipe_create(pm_ctx *ctx, const pm_pipe_desc *desc, pm_pipe **out);
/* d_audio stays untouched until the recording is sliced.  A recording refused for its shape (PM_ERR_ARG) takes no ticket; one whose
 * launches fail has a ticket (*h_ticket) that is finished with that error: pm_pipe_wait returns it, pm_pipe_release frees it. */
int pm_pipe_submit(pm_pipe *pipe, const int16_t *d_audio, int64_t n, int64_t *h_ticket);
/* `count` recordings in order from one call (blocks as pm_pipe_submit does, for the whole run): tickets *h_first_ticket ..
 * + count - 1, known before the call returns to anyone who read the next ticket -- pm_pipe_wait on a promised ticket waits for its
 * submission.  For hosts whose submitting thread would otherwise queue for an interpreter lock between recordings.  One submitter. */
int pm_pipe_submit_many(pm_pipe *pipe, const int16_t *const *d_audio, const int64_t *n, int count, int64_t *h_first_ticket);
/* Announces the next `count` tickets (a pm_pipe_submit_many that another thread is about to start): pm_pipe_wait on them waits. */
int pm_pipe_promise(pm_pipe *pipe, int count, int64_t *h_first_ticket);
int pm_pipe_wait(pm_pipe *pipe, int64_t ticket, pm_pipe_result *out);
/* A finished recording's slicer output for one chain (pipelines made with keep_slices): the bytes slicer.slice would return with
 * their stream addresses (slicer.py:59-107), and what stream_unscramble_8bit made of them (lfsr.py:22-52), *h_count entries each,
 * in the library's memory until pm_pipe_release.  Any of the three pointers may be NULL. */
int pm_pipe_slices(pm_pipe *pipe, int64_t ticket, int chain, const uint8_t **h_data, const int64_t **h_addr, const uint8_t **h_plain, int64_t *h_count);
/* A finished recording's SIGN BITMAP for one chain -- bit k = (modem.demod(audio)[k] >= 0), what its slicer read (slicer.py:85) -- copied
 * to h_words (`words` 64-bit words from the bitmap's start).  Pipelines made with keep_slices only, and only while the recording's
 * bitmap slot has not been taken by a later submission (fewer than pm_pipe_slots() submissions since): PM_ERR_ARG otherwise.  Test entry:
 * the demod stage's output compared bit by bit, whatever the slicer makes of it. */
int pm_pipe_bitmap(pm_pipe *pipe, int64_t ticket, int chain, uint64_t *h_words, int64_t words);
int pm_pipe_slots(pm_pipe *pipe);                        /* recordings between demod and slicer at most, as pm_pipe_create settled it */
int pm_pipe_release(pm_pipe *pipe, int64_t ticket);      /* the result's memory */
int pm_pipe_drain(pm_pipe *pipe);                        /* every recording submitted so far is through */
int pm_pipe_stats(pm_pipe *pipe, int64_t *h_batches, int64_t *h_batch_recordings, double *h_slice_busy_ms, double *h_host_busy_ms);
pm_ctx *pm_pipe_side_ctx(pm_pipe *pipe, int worker);     /* a slicer worker's context (NULL past the last): for pm_prof_* */
pm_ctx *pm_pipe_demod_ctx(pm_pipe *pipe, int k);         /* demod stream k (0 = the caller's context; NULL past the last) */
int pm_pipe_destroy(pm_pipe *pipe);                      /* drains first */

#ifdef __cplusplus
}
#endif
#endif /* PYMODEM_AMD_H */
