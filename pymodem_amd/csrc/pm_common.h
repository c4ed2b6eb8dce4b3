// Internal declarations shared by the translation units of libpymodem_amd.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../include/pymodem_amd.h"

constexpr size_t PM_PINNED_BYTES = 256 * 1024;
constexpr int kSweepRing = 64;
constexpr int kSweepCap = 65536;        // uncertain (sample, modem) pairs a certified sweep's list holds; more: the exact chains take over

// Counter and mailbox words of the certified sweeps of ONE recording, owned by whoever runs recordings side by side (pm_pipe.hip):
// sweep k counts the uncertain samples it could not decide itself in d_count[k], and the count arrives in the page-locked h_mail[k]
// when the recording's demod stage has finished.  With d_list (round 5) a matrix-pipe sweep decides its uncertain samples inside its
// workgroups and only what a workgroup could not take goes to d_list[k * kSweepCap ..], with no launch behind the sweep: the owner
// runs pm_afsk_sweep_exact_list over a list that is not empty.  The words belong to the recording from its submission until its
// owner is through with them -- no ring, no arithmetic on sequence numbers.
struct pm_sweep_cells {
    int *d_count = nullptr;                 // nsweeps live counters (zero when the recording's demod stage starts and again when it has mailed them)
                                            // + with d_list nsweeps more: the counts as mailed, for pm_afsk_sweep_exact_list
    int *h_mail = nullptr;                  // nsweeps words
    unsigned long long *d_list = nullptr;   // nullptr (every sweep ends with a launch that works its list off), or nsweeps * kSweepCap entries
};

// Diagnostic switches of the launchers: read from the environment ONCE, when a context is made (pm_tuning_from_env in
// pm_runtime.hip), or set per context with pm_ctx_tune -- no launch path calls getenv.  README.md lists what each one is for.
struct pm_tuning {
    int fir_no_short = 0;              // PM_FIR_NO_SHORT: short-tap sign FIR through the LDS kernel
    int fuse_run = 0;                  // PM_FUSE_RUN: 16 = round 1's run length in the fused AFSK kernel (0: the default)
    int afsk_unfused = 0;              // PM_AFSK_UNFUSED: certified AFSK path as separate kernels
    int afsk_lpf8 = 0;                 // PM_AFSK_LPF8: pm_afsk_sweep_signs_tones with a per-call matrix-pipe low-pass plan
    int loop_wide = -1;                // PM_LOOP_WIDE: carrier-loop kernel shape 0 / 1 / 2 (-1: by size)
    int64_t loop_lds_min = 0;          // PM_LOOP_LDS_MIN
    int lbatch_tail = -1;              // PM_LBATCH_TAIL: matched filters behind (0) or beside the next chunk's loops
    int agc_trace = 0;                 // PM_AGC_TRACE
    int slicer_max_chunk_words = 0, slicer_chunk_words = 0, slicer_quantum_words = 0;      // PM_SLICER_*_WORDS (0: the defaults)
    int slicer_compare_step = 0, slicer_mask_step = 0, slicer_compiled_step = 0;           // older forms of the slicer step
    int slicer_trace = 0, slicer_no_setprio = 0;
    int fir8 = 1;                      // PM_FIR8=0: the batch engine's matched filters in binary64 on the vector pipe
    int bpf8_max = 1;                  // PM_BPF8_MAX=0: the batch engine's pass for the AGC's `normal` as the reference's sums + a maximum
    int loop_agc = 1;                  // PM_LOOP_AGC=0: the batch engine's BPSK AGC as a pass of its own, not in the loop's lane
    int loop_vec = 1;                  // PM_LOOP_VEC=0: the direct loop shape moves its blocks with eight-byte accesses, a lane a row
    int lbatch_loop_cus = -1;          // PM_LBATCH_LOOP_CUS: compute units the batch engine's carrier loops have to themselves (0: none, -1: by size)
    int agc_rows_prio = 2;             // PM_AGC_ROWS_PRIO: wave priority of the rows AGC (the loops run at 3)
    int sweep_lds_templates = 0;       // PM_SWEEP_LDS_TEMPLATES: the fused kernel's sliding sums read their templates from LDS (as the split kernel does), not through the scalar cache
    int fused_lds_pad = 0;             // PM_FUSED_LDS_PAD: bytes of LDS the fused AFSK launch asks for beyond what it uses (12288: three workgroups per CU, 40960: two)
    int afsk_split = 0;                // PM_AFSK_SPLIT: the pipelined executor's AFSK stage as band-pass + one launch per sweep (round 4), not fused into one launch
    int sweep_no_tail = 0;             // PM_SWEEP_NO_TAIL: the matrix-pipe sweep sends every uncertain sample to the list (round 4), none to its own workgroup's exact chain
};
pm_tuning pm_tuning_from_env();

struct pm_ctx {
    pm_tuning tune = pm_tuning_from_env();
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // small pinned host mailbox + device scratch used by reductions / the slicer fixed point
    void *h_pinned = nullptr;       // PM_PINNED_BYTES
    void *d_scratch = nullptr;      // grows on demand
    size_t scratch_bytes = 0;
    // optional per-kernel-class timing (pm_prof_*)
    bool prof_on = false;
    struct ProfPair { hipEvent_t a, b; int cls; };
    std::vector<ProfPair> prof_pending;
    std::vector<hipEvent_t> prof_free;
    double prof_ms[PM_K_COUNT] = {0};
    int64_t prof_n[PM_K_COUNT] = {0};
    double prof_bytes[PM_K_COUNT] = {0}, prof_flops[PM_K_COUNT] = {0};   // algorithmic work of the tracked launches
    std::vector<float> prof_iv[PM_K_COUNT];   // (start, end) of every tracked launch in ms since the device's reference event (pm_prof_intervals)
    // slicer diagnostics
    int32_t sl_iterations = 0, sl_chunk_len = 0;
    int64_t sl_chunks = 0;
    int64_t sl_target_lanes = 16384;   // walkers (= chunks) a slicer batch is cut into (pm_slicer_tune)
    int64_t sl_max_chunk_words = 384;  // longest chunk in 64-sample words (pm_slicer_limits)
    int64_t sl_hint_shape = 0;         // chunk/launch geometry sl_launch_hint was learnt on
    int32_t sl_launch_hint = 0;        // lockstep launches to enqueue before the emit kernels without asking the device
    int *sweep_count = nullptr;        // device counter of the last pm_afsk_sweep_signs on this ctx (a slot of d_sweep)
    int *d_sweep = nullptr;            // ring of kSweepRing counters, one per certified sweep in flight (own allocation)
    int64_t sweep_seq = 0;
    int *h_sweep = nullptr;            // pinned mailbox ring: the deferred sweep's last launch writes its counter here too
    int64_t sweep_mail[64] = {0};      // ticket + 1 whose counter mailbox slot [ticket % ring] will hold (0: none)
    bool sweep_deferred = false;       // pm_afsk_sweep_mode: the overflow fallback is the caller's (pm_afsk_sweep_result), not three gated launches
};

int pm_set_error(int code, const char *fmt, ...);
int pm_scratch_reserve(pm_ctx *ctx, size_t bytes);     // ensures ctx->d_scratch >= bytes (contents undefined)

#define PM_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return pm_set_error(PM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                              \
    } while (0)

#define PM_ARG(cond)                                                                    \
    do {                                                                                \
        if (!(cond)) return pm_set_error(PM_ERR_ARG, "bad argument: %s (%s:%d)", #cond, \
                                         __FILE__, __LINE__);                           \
    } while (0)

// First line of every entry point that takes a ctx: validate it and make its device current on the CALLING thread (the HIP
// current device is per thread; a ctx may be driven from a worker thread, e.g. the slicer stage of the pipelined executor).
#define PM_CTX(c)                                   \
    do {                                            \
        PM_ARG((c) != nullptr);                     \
        PM_HIP(hipSetDevice((c)->device));          \
    } while (0)

// Bracket a launch: `PmProf p(ctx, PM_K_X); launch...; ` (destructor records the end event).
struct PmProf {
    pm_ctx *c; hipEvent_t a = nullptr, b = nullptr; int cls;
    PmProf(pm_ctx *ctx, int k);
    ~PmProf();
    // algorithmic (compulsory) HBM bytes and f64 flops of this launch: inputs read once, outputs written once, 2 flops per fma
    void work(double bytes, double flops) { if (c->prof_on) { c->prof_bytes[cls] += bytes; c->prof_flops[cls] += flops; } }
};
int pm_prof_fold(pm_ctx *ctx);      // sync + accumulate pending pairs

static inline int64_t pm_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- the group band-pass on the int8 matrix pipe (pm_bpf8.hip): a value within pm_bpf8_error() of the reference's sum, for the
// certified sweeps only.  A plan belongs to one tap set (host copy given once) and one device.
struct pm_bpf8_plan;
int pm_bpf8_plan_create(pm_ctx *ctx, const double *h_taps, int m, pm_bpf8_plan **out, int digits = 4);   // digits: 4, or pm_bpf8_max_digits() for pm_bpf8_rows_max
int pm_bpf8_max_digits(void);      // PM_ERR_ARG: more than 241 taps
void pm_bpf8_plan_destroy(pm_bpf8_plan *p);
double pm_bpf8_error(const pm_bpf8_plan *p);
int pm_bpf8_plan_view(const pm_bpf8_plan *p, int *kb, const void **d_btab, double *scales6);   // the plan's blocks (3 / 4), band table and 5 scales + constant
int pm_bpf8_digit_pairs(void);     // int8 digit products per tap and output of bpf8_kernel (and of fir8_kernel: pm_fir8_digit_pairs)
int pm_fir8_digit_pairs(void);
int pm_bpf8_taps(const pm_bpf8_plan *p);
// d_audio 16-byte aligned; d_clear: nclear (<= 64) ints the launch zeroes (a recording's sweep counters: the band-pass is the first
// launch of its demod stage, the sweeps behind it on the same stream start from zero without a memset of their own)
int pm_bpf8_run(pm_ctx *ctx, const pm_bpf8_plan *p, const int16_t *d_audio, int64_t n, double *d_y, int *d_clear = nullptr, int nclear = 0);
// d_out[r] = max(band-pass of row r), the reference's value exactly (matrix-pipe values pick the candidates, the canonical chain decides);
// d_keys: rows words of work space; d_redone: null or a counter of the outputs that took the exact chain.  Rows 16-byte aligned.
int pm_bpf8_rows_max(pm_ctx *ctx, const pm_bpf8_plan *p, const int16_t *const *d_rows, int rows, int64_t n, unsigned long long *d_keys, double *d_out,
                     unsigned long long *d_redone = nullptr);
// The certified sweeps' low-pass on the same pipe (afsk_slide_lpf8_kernel in pm_fir.hip): taps as three signed base-256 digits
// q = rint(h 2^S), |q| <= 2^22, the Toeplitz band as MFMA B operands [digit][block][lane] on the device.  ml <= 113.
struct pm_lpf8_plan {
    int ml = 0, S = 0, device = 0;
    double tapq_int = 0;             // sum |h 2^S - q|
    double qabs = 0;                 // sum |q|
    double dlow = 0;                 // bound on the digit product the kernel leaves out: 128 sum|q_0|
    double hmax = 0;
    void *d_btab = nullptr;
    // made by the fused launch the first time it meets the sweep's templates (pm_fir.hip: afsk_group_run_fused): the four correlator
    // templates reversed and interleaved, for scalar loads; tpl_src / tpl_m say which templates it was made from
    void *d_tpl = nullptr;
    const void *tpl_src[4] = {nullptr, nullptr, nullptr, nullptr};
    int tpl_m = 0;
};
int pm_lpf8_plan_create(pm_ctx *ctx, const double *h_taps, int ml, pm_lpf8_plan **out);
void pm_lpf8_plan_destroy(pm_lpf8_plan *p);
// pm_afsk_group_run with the band-pass from a plan (nullptr: the reference's sum); every sweep must then carry tones.
// lpf8: nullptr or one plan per sweep (entries may be nullptr): that sweep's low-passes on the matrix pipe
// cells: nullptr (the context's counter ring, results through pm_afsk_sweep_results) or the recording's own counters (nsweeps of
// each; h_tickets is not written then): the counts arrive in cells->h_mail[k] when the recording's demod stage has finished
int pm_afsk_group_run_plan(pm_ctx *ctx, const int16_t *d_audio, int64_t n, const double *d_bpf, int mb, double *d_bpf_out, double x_bound,
                           const pm_afsk_sweep_desc *h_sweeps, int nsweeps, int64_t *h_tickets, const pm_bpf8_plan *plan,
                           const pm_lpf8_plan *const *lpf8 = nullptr, const pm_sweep_cells *cells = nullptr);
// The exact chain (afsk.py:151-166, canonical order) for the (sample, modem) pairs of a sweep's list: min(*d_count, kSweepCap) entries of
// d_list, bits into h_bits[modem].  d_audio / d_bpf / mb: where the sweep's input came from (pm_afsk_group_run_plan with a plan).
int pm_afsk_sweep_exact_list(pm_ctx *ctx, const int16_t *d_audio, const double *d_bpf, int mb, const pm_afsk_sweep_desc *w, uint64_t *const *h_bits,
                             const unsigned long long *d_list, const int *d_count);

// pm_codec_fetch_batch into rows whose payload fields are already zero wherever a packet will not write (pm_pipe.hip keeps its row blocks
// that way): per packet the bytes it has, not the 1280 of the field
extern "C" int pm_codec_fetch_batch_clean(pm_codec *const *codecs, const int64_t *counts, int n, pm_packet *h_out, int threads);

// ---- the batch engine's slicers: one lane per stream, a chunk per launch, state carried in d_recs (pm_slicer.hip: rowslice_kernel)
struct pm_rowslice;
int pm_rowslice_create(pm_ctx *ctx, const pm_slicer_params *h_params, int chains, pm_rowslice **out);
void pm_rowslice_destroy(pm_rowslice *rs);
bool pm_rowslice_made_for(const pm_rowslice *rs, const pm_slicer_params *h_params, int chains);
// samples [first, first + count) of every row (first a multiple of 64): bit k of d_bi[row * stride + k / 64] is sample first + k
int pm_rowslice_chunk(pm_ctx *ctx, const pm_rowslice *rs, int rows, int quad, const uint64_t *d_bi, const uint64_t *d_bq, int64_t stride, int64_t first,
                      int64_t count, pm_rowslice_rec *d_recs, uint8_t *d_data, uint16_t *d_steps, int64_t cap);

// ---- long matched filters as certified signs on the int8 matrix pipe (pm_fir8.hip): bit k of row r = (canonical FIR sum >= 0), the
// bitmap pm_fir_rows(..., d_bits, ...) writes, for inputs of any magnitude.  A plan belongs to one tap set (m + 15 <= 1024) and device.
struct pm_fir8_plan;
int pm_fir8_plan_create(pm_ctx *ctx, const double *h_taps, int m, pm_fir8_plan **out);
void pm_fir8_plan_destroy(pm_fir8_plan *p);
int pm_fir8_taps(const pm_fir8_plan *p);
// rows of n doubles, pitch x_stride -> rows of n - m + 1 sign bits, pitch bits_stride words (bits past the last output of the last word
// zero).  d_count: nullptr, or a device counter that gains the number of outputs recomputed exactly (diagnostics).  x_room: 0, or how
// many doubles may be READ from each row's start (>= n: rows of a pitched block; whole windows are then loaded without bounds tests)
int pm_fir8_rows_signs(pm_ctx *ctx, const pm_fir8_plan *p, const double *d_x, int64_t x_stride, int rows, int64_t n, uint64_t *d_bits,
                       int64_t bits_stride, int *d_count, int64_t x_room = 0);

// ---- launchers shared between translation units (what the batch engine pm_loopbatch.hip strings together) ---------------------
// `rows` FIRs with the same taps over equal-length streams in one launch (pm_fir.hip).  Input row r: d_x + r * x_stride, or
// d_x_ptrs[r] + x_off (d_x_ptrs: DEVICE array of row pointers); x_aligned16: every input row starts on a 16-byte boundary.
// Exactly one of d_y (rows of n - m + 1 doubles, pitch y_stride) and d_bits (sign bitmaps, pitch bits_stride words) is given.
int pm_fir_rows(pm_ctx *ctx, bool i16, const void *d_x, int64_t x_stride, const void *const *d_x_ptrs, int64_t x_off, bool x_aligned16, int rows,
                int64_t n, const double *d_taps, int m, double *d_y, int64_t y_stride, uint64_t *d_bits, int64_t bits_stride, int flags);
// nloops carrier loops resident in device memory; loop l reads input row l / per_row (pm_loops.hip).  modem: PM_MODEM_*.
int pm_loops_rows(pm_ctx *ctx, int modem, pm_loop *d_loops, int nloops, int per_row, const double *d_table, const int32_t *d_pd,
                  const double *d_x0, const double *d_x1, int64_t x_stride, int64_t n, double *d_o0, double *d_o1, int64_t out_stride);
// BPSK with one chain per recording in the direct loop shape: the loop's lane steps the AGC too (pm_loops_rows_agc reads the BAND-PASSED
// rows; d_consts / d_state as pm_agc_rows) -- pm_loops_rows_take_agc says whether a launch of this size does
bool pm_loops_rows_take_agc(const pm_ctx *ctx, int modem, int nloops, int per_row);
int pm_loops_rows_agc(pm_ctx *ctx, pm_loop *d_loops, int nloops, const double *d_table, const double *d_x, int64_t x_stride, int64_t n, double *d_o,
                      int64_t out_stride, const pm_agc_params *hp, const double *d_consts, double *d_state);
// d_running[r] = max(d_running[r], max of row r) (first != 0: = max of row r).  d_partial: rows * pm_rows_max_parts() doubles of work space.
int pm_rows_max(pm_ctx *ctx, const double *d_x, int64_t x_stride, int rows, int64_t n, double *d_partial, double *d_running, int first);
int pm_rows_max_parts(void);
// d_consts[4r..] = {normal, attack step, decay step, -} from d_running[r] = max(buffer) (agc.py:15-16,67)
int pm_agc_rows_prepare(pm_ctx *ctx, const double *d_running, int rows, const pm_agc_params *hp, double *d_consts);
// the envelope follower of every row continued over n more samples: y = target * x / envelope; d_state[2r..] = {envelope, sustain}
int pm_agc_rows(pm_ctx *ctx, const double *d_x, int64_t x_stride, double *d_y, int64_t y_stride, int rows, int64_t n, const pm_agc_params *hp,
                const double *d_consts, double *d_state);
