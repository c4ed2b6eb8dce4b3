// Whole-chain entry points: modem + slicer of one demod_chain in one call (pm_chain_create / pm_chain_run), for hosts
// that do not want to sequence the stage kernels themselves.  Nothing new is computed here: pm_chain_run strings together the
// stage entry points of this library in the order of the reference's demod() methods (afsk.py:148-167, fsk.py:149-159,
// psk.py:162-195, psk.py:705-773, afsk_pll.py:140-170) followed by slicer.slice (slicer.py:59-107, 193-242), with every
// intermediate kept in device buffers owned by the chain object.  The last FIR of each modem writes only the sign bitmap.
#include "pm_common.h"
#include <cstring>
#include <vector>

struct pm_chain {
    pm_ctx *ctx = nullptr;
    pm_chain_desc d{};                       // pointers inside are NOT kept: the vectors below own the copies
    std::vector<double> h_taps;              // all host tap vectors, concatenated
    double *d_taps = nullptr;                // the same on the device
    size_t o_in = 0, o_mi = 0, o_mq = 0, o_si = 0, o_sq = 0, o_hil = 0, o_out = 0, o_wave = 0;
    int32_t *d_pd = nullptr;
    pm_loop loop0{}, loop{};
    double agc_state[2] = {0.0, 0.0};
    pm_slicer_state slicer_state{};
    // device work buffers (grown on demand)
    uint8_t *d_audio = nullptr; size_t audio_bytes = 0;
    double *d_a = nullptr, *d_b = nullptr; size_t a_n = 0, b_n = 0;
    uint64_t *d_bits_i = nullptr, *d_bits_q = nullptr; size_t bits_i_n = 0, bits_q_n = 0;
    uint8_t *d_data = nullptr; int64_t *d_addr = nullptr; size_t data_n = 0, addr_n = 0;
    int64_t last_count = -1;                 // bytes the last pm_chain_run produced (still in d_data / d_addr): pm_chain_fetch
    std::vector<int16_t> hist;               // PM_CHAIN_CARRY_HISTORY: the last sum(M - 1) input samples of the previous run
    // PM_CHAIN_CARRY_HISTORY for the carrier-loop modems: every later FIR of the cascade keeps the tail of ITS input stream on the device
    // ([0] AGC'd samples in front of the Hilbert pair, [1] / [2] loop outputs in front of the matched / output filter)
    double *d_tail[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};      // two each: a run writes the other one, a
    int tail_sel[3] = {0, 0, 0};                                                                 // successful run makes it current
    int64_t tail_n[3] = {0, 0, 0};
    double *d_win[3] = {nullptr, nullptr, nullptr}; size_t win_n[3] = {0, 0, 0};
};

namespace {

template <typename T>
int grow(pm_ctx *ctx, T *&p, size_t &have, size_t want)
{
    if (want <= have && p) return PM_OK;
    if (p) { if (int rc = pm_free(ctx, p)) return rc; p = nullptr; have = 0; }
    void *q = nullptr;
    if (int rc = pm_malloc(ctx, (want + want / 8 + 64) * sizeof(T), &q)) return rc;
    p = (T *)q;
    have = want + want / 8 + 64;
    return PM_OK;
}

size_t put(std::vector<double> &v, const double *src, int n)
{
    const size_t at = v.size();
    if (src && n > 0) v.insert(v.end(), src, src + n);
    while (v.size() % 2) v.push_back(0.0);                 // keep every vector 16-byte aligned on the device
    return at;
}

}  // namespace

extern "C" {

int pm_chain_create(pm_ctx *ctx, const pm_chain_desc *desc, pm_chain **out)
{
    PM_CTX(ctx);
    PM_ARG(desc != nullptr && out != nullptr);
    const pm_chain_desc &d = *desc;
    PM_ARG(d.modem >= PM_MODEM_AFSK && d.modem <= PM_MODEM_QPSK);
    PM_ARG(d.input_fir && d.n_input_fir >= 1);
    if (d.modem == PM_MODEM_AFSK) PM_ARG(d.mark_i && d.mark_q && d.space_i && d.space_q && d.n_corr >= 1);
    if (d.modem != PM_MODEM_FSK) PM_ARG(d.output_fir && d.n_output_fir >= 1);
    if (d.modem == PM_MODEM_MPSK) PM_ARG(d.hilbert && d.n_hilbert >= 1 && d.hilbert_delay >= 0 && d.hilbert_delay < d.n_hilbert && d.pd_table);
    if (d.modem == PM_MODEM_BPSK || d.modem == PM_MODEM_MPSK || d.modem == PM_MODEM_AFSK_PLL || d.modem == PM_MODEM_QPSK)
        PM_ARG(d.wavetable != nullptr);
    PM_ARG((d.quadrature != 0) == (d.modem == PM_MODEM_MPSK || d.modem == PM_MODEM_QPSK));
    // (carried FIR history makes pieces equal the whole for the modems whose demod() is FIRs and pointwise operations only; the
    // carrier-loop modems normalise by the maximum of each call's buffer, agc.py:67, and keep doing so: see pm_chain_run)
    pm_chain *c = new pm_chain();
    c->ctx = ctx;
    c->d = d;
    c->o_in = put(c->h_taps, d.input_fir, d.n_input_fir);
    if (d.modem == PM_MODEM_AFSK) {
        c->o_mi = put(c->h_taps, d.mark_i, d.n_corr);
        c->o_mq = put(c->h_taps, d.mark_q, d.n_corr);
        c->o_si = put(c->h_taps, d.space_i, d.n_corr);
        c->o_sq = put(c->h_taps, d.space_q, d.n_corr);
    }
    if (d.modem == PM_MODEM_MPSK) c->o_hil = put(c->h_taps, d.hilbert, d.n_hilbert);
    if (d.modem != PM_MODEM_FSK) c->o_out = put(c->h_taps, d.output_fir, d.n_output_fir);
    if (d.wavetable) c->o_wave = put(c->h_taps, d.wavetable, 256);
    void *p = nullptr;
    int rc = pm_malloc(ctx, c->h_taps.size() * sizeof(double), &p);
    if (!rc) { c->d_taps = (double *)p; rc = pm_h2d(ctx, c->d_taps, c->h_taps.data(), c->h_taps.size() * sizeof(double)); }
    if (!rc && d.modem == PM_MODEM_MPSK) {
        rc = pm_malloc(ctx, 4096 * sizeof(int32_t), &p);
        if (!rc) { c->d_pd = (int32_t *)p; rc = pm_h2d(ctx, c->d_pd, d.pd_table, 4096 * sizeof(int32_t)); }
    }
    if (rc) { pm_chain_destroy(c); return rc; }
    c->loop0 = c->loop = d.loop;
    // the desc's pointers belong to the caller: drop them
    c->d.input_fir = c->d.mark_i = c->d.mark_q = c->d.space_i = c->d.space_q = c->d.hilbert = c->d.output_fir = c->d.wavetable = nullptr;
    c->d.pd_table = nullptr;
    *out = c;
    return PM_OK;
}

int pm_chain_reset(pm_chain *c)
{
    PM_ARG(c != nullptr);
    c->loop = c->loop0;
    c->agc_state[0] = c->agc_state[1] = 0.0;
    c->slicer_state = pm_slicer_state{};
    c->hist.clear();
    c->tail_n[0] = c->tail_n[1] = c->tail_n[2] = 0;
    return PM_OK;
}

int pm_chain_destroy(pm_chain *c)
{
    if (!c) return PM_OK;
    pm_ctx *ctx = c->ctx;
    for (void *p : {(void *)c->d_taps, (void *)c->d_pd, (void *)c->d_audio, (void *)c->d_a, (void *)c->d_b, (void *)c->d_bits_i,
                    (void *)c->d_bits_q, (void *)c->d_data, (void *)c->d_addr, (void *)c->d_tail[0][0], (void *)c->d_tail[1][0], (void *)c->d_tail[2][0],
                    (void *)c->d_tail[0][1], (void *)c->d_tail[1][1], (void *)c->d_tail[2][1], (void *)c->d_win[0], (void *)c->d_win[1], (void *)c->d_win[2]})
        if (p) (void)pm_free(ctx, p);
    delete c;
    return PM_OK;
}

int pm_chain_run(pm_chain *c, const int16_t *audio, int64_t n, int audio_on_device, uint8_t *h_data, int64_t *h_addr, int64_t cap,
                 int64_t *h_count)
{
    PM_ARG(c != nullptr && h_count != nullptr && n >= 0 && cap >= 0 && (cap == 0 || (h_data && h_addr)));
    pm_ctx *ctx = c->ctx;
    PM_CTX(ctx);
    *h_count = 0;
    if (n == 0) return PM_OK;
    PM_ARG(audio != nullptr);
    c->last_count = -1;                                    // nothing to fetch until this run has produced its output
    const pm_chain_desc &d = c->d;
    const double *T = c->d_taps;
    const int16_t *x = audio;
    std::vector<int16_t> next_hist;                        // PM_CHAIN_CARRY_HISTORY: committed only when the run has succeeded, so that a
    bool have_next_hist = false;                           // failed run (allocation, slicer) can be repeated on the same samples
    if (d.flags & PM_CHAIN_CARRY_HISTORY) {
        // seamless pieces: [the previous run's last sum(M - 1) input samples | this run's] goes through the FIR
        // cascade, whose 'valid' output is then exactly the continuation of the previous run's (see _DeviceStage._with_history)
        const int64_t hlen = (int64_t)d.n_input_fir - 1 + (d.modem == PM_MODEM_AFSK ? (int64_t)d.n_corr - 1 + d.n_output_fir - 1 : 0);
        const int64_t nt = (int64_t)c->hist.size();
        if (int rc = grow(ctx, c->d_audio, c->audio_bytes, (size_t)(nt + n) * 2)) return rc;
        if (nt) { if (int rc = pm_h2d(ctx, c->d_audio, c->hist.data(), (size_t)nt * 2)) return rc; }
        if (audio_on_device) { if (int rc = pm_d2d(ctx, c->d_audio + nt * 2, audio, (size_t)n * 2)) return rc; }
        else { if (int rc = pm_h2d(ctx, c->d_audio + nt * 2, audio, (size_t)n * 2)) return rc; }
        // the new tail: the last hlen samples of [tail | audio]
        const int64_t take = std::min(hlen, n), keep = std::min(hlen - take, nt);
        std::vector<int16_t> next((size_t)(keep + take));
        if (keep) memcpy(next.data(), c->hist.data() + (nt - keep), (size_t)keep * 2);
        if (take) {
            if (audio_on_device) { if (int rc = pm_d2h(ctx, next.data() + keep, audio + (n - take), (size_t)take * 2)) return rc; }
            else memcpy(next.data() + keep, audio + (n - take), (size_t)take * 2);
        }
        if (int rc = pm_ctx_sync(ctx)) return rc;          // the tail's upload has left c->hist (it is replaced at the end of the run)
        next_hist.swap(next);
        have_next_hist = true;
        x = (const int16_t *)c->d_audio;
        n += nt;
        if (n <= hlen) {                                   // not one output yet: the samples wait in the tail
            c->hist.swap(next_hist);
            c->last_count = 0;
            return PM_OK;
        }
    } else if (!audio_on_device) {
        if (int rc = grow(ctx, c->d_audio, c->audio_bytes, (size_t)n * 2)) return rc;
        if (int rc = pm_h2d(ctx, c->d_audio, audio, (size_t)n * 2)) return rc;
        x = (const int16_t *)c->d_audio;
    }
    const int mi = d.n_input_fir;
    if (n < mi) return pm_set_error(PM_ERR_ARG, "pm_chain_run: %lld samples are fewer than the %d-tap input filter", (long long)n, mi);
    const bool carry = (d.flags & PM_CHAIN_CARRY_HISTORY) != 0;
    // One later FIR stage's input with carried history (carrier-loop modems): [tail k | the n_new samples at src] in the chain's window
    // buffer k, whose last h samples become the new tail -- all on the stream, nothing waits.  *win receives the window (src itself
    // when there is nothing to carry), *wn its length.
    int64_t new_tail_n[3] = {c->tail_n[0], c->tail_n[1], c->tail_n[2]};
    bool used_tail[3] = {false, false, false};
    auto carried = [&](int k, int h, const double *src, int64_t n_new, const double **win, int64_t *wn) -> int {
        *win = src;
        *wn = n_new;
        if (!carry || h <= 0) return PM_OK;
        const int64_t nt = c->tail_n[k], total = nt + n_new;
        for (int b = 0; b < 2; ++b) {
            if (c->d_tail[k][b]) continue;
            void *p = nullptr;
            if (int rc = pm_malloc(ctx, (size_t)h * sizeof(double), &p)) return rc;
            c->d_tail[k][b] = (double *)p;
        }
        const double *cur = c->d_tail[k][c->tail_sel[k]];
        double *next = c->d_tail[k][c->tail_sel[k] ^ 1];
        if (nt) {
            if (int rc = grow(ctx, c->d_win[k], c->win_n[k], (size_t)total)) return rc;
            if (int rc = pm_d2d(ctx, c->d_win[k], cur, (size_t)nt * sizeof(double))) return rc;
            if (n_new) { if (int rc = pm_d2d(ctx, c->d_win[k] + nt, src, (size_t)n_new * sizeof(double))) return rc; }
            *win = c->d_win[k];
            *wn = total;
        }
        // the new tail: the window's last h samples, into the OTHER tail buffer (the current one stays valid until the run succeeds)
        const int64_t keep = std::min<int64_t>(h, total);
        if (keep) { if (int rc = pm_d2d(ctx, next, *win + (total - keep), (size_t)keep * sizeof(double))) return rc; }
        new_tail_n[k] = keep;
        used_tail[k] = true;
        return PM_OK;
    };
    const int64_t n1 = n - mi + 1;                         // after the input filter
    int64_t ns = 0;                                        // samples the slicer sees
    auto bits_for = [&](int64_t count) -> int {
        const size_t words = (size_t)(count + 63) / 64 + 1;
        if (int rc = grow(ctx, c->d_bits_i, c->bits_i_n, words)) return rc;
        if (d.quadrature) { if (int rc = grow(ctx, c->d_bits_q, c->bits_q_n, words)) return rc; }
        return PM_OK;
    };
    if (d.modem == PM_MODEM_FSK) {                                              // fsk.py:149-159
        ns = n1;
        if (int rc = bits_for(ns)) return rc;
        if (int rc = pm_fir_signs_i16(ctx, x, n, T + c->o_in, mi, c->d_bits_i, (d.flags & PM_CHAIN_INVERT) ? PM_FIR_NEGATE : 0)) return rc;
    } else {
        if (int rc = grow(ctx, c->d_a, c->a_n, (size_t)n1)) return rc;
        if (int rc = pm_fir_valid_i16(ctx, x, n, T + c->o_in, mi, c->d_a, 0)) return rc;
        const int mo = d.n_output_fir;
        if (d.modem == PM_MODEM_AFSK) {                                         // afsk.py:148-167
            if (n1 < d.n_corr) return pm_set_error(PM_ERR_ARG, "pm_chain_run: input shorter than the correlators");
            const int64_t n2 = n1 - d.n_corr + 1;
            if (n2 < mo) return pm_set_error(PM_ERR_ARG, "pm_chain_run: input shorter than the output filter");
            if (int rc = grow(ctx, c->d_b, c->b_n, (size_t)n2)) return rc;
            if (int rc = pm_afsk_correlate(ctx, c->d_a, n1, T + c->o_mi, T + c->o_mq, T + c->o_si, T + c->o_sq, d.n_corr, c->d_b)) return rc;
            ns = n2 - mo + 1;
            if (int rc = bits_for(ns)) return rc;
            if (int rc = pm_fir_signs_f64(ctx, c->d_b, n2, T + c->o_out, mo, c->d_bits_i, 0)) return rc;
        } else {
            if (d.use_agc) { if (int rc = pm_agc_apply(ctx, c->d_a, n1, &d.agc, c->agc_state)) return rc; }
            // With PM_CHAIN_CARRY_HISTORY a stage whose [tail | new] is still shorter than its filter has no output yet (the samples
            // wait in its tail) and neither has anything behind it; without the flag that is an argument error, as before.
            const double *w0 = nullptr, *w1 = nullptr;
            int64_t wn0 = 0, wn1 = 0;
            if (d.modem == PM_MODEM_MPSK) {                                     // psk.py:705-773
                if (int rc = carried(0, d.n_hilbert - 1, c->d_a, n1, &w0, &wn0)) return rc;      // the AGC'd stream in front of the Hilbert pair
                if (wn0 < d.n_hilbert && !carry) return pm_set_error(PM_ERR_ARG, "pm_chain_run: input shorter than the Hilbert transformer");
                const int64_t n2 = wn0 >= d.n_hilbert ? wn0 - d.n_hilbert + 1 : 0;
                if (int rc = grow(ctx, c->d_b, c->b_n, (size_t)(n2 + 1) * 3)) return rc;      // imag | i_mix | q_mix
                double *imag = c->d_b, *i_mix = c->d_b + n2, *q_mix = c->d_b + 2 * n2;
                if (n2) {
                    if (int rc = pm_fir_valid_f64(ctx, w0, wn0, T + c->o_hil, d.n_hilbert, imag, 0)) return rc;
                    // the delay FIR [1,0,...,0] followed by [:-delay] is a pure shift (psk.py:714-716): real[k] = a[k + delay]
                    if (int rc = pm_mpsk_loop(ctx, &c->loop, 1, T + c->o_wave, c->d_pd, w0 + d.hilbert_delay, imag, 0, n2, i_mix, q_mix, n2)) return rc;
                }
                if (int rc = carried(1, mo - 1, i_mix, n2, &w0, &wn0)) return rc;
                if (int rc = carried(2, mo - 1, q_mix, n2, &w1, &wn1)) return rc;
                if (wn0 < mo && !carry) return pm_set_error(PM_ERR_ARG, "pm_chain_run: input shorter than the matched filter");
                ns = wn0 >= mo ? wn0 - mo + 1 : 0;
                if (int rc = bits_for(ns)) return rc;
                if (ns) {
                    if (int rc = pm_fir_signs_f64(ctx, w0, wn0, T + c->o_out, mo, c->d_bits_i, 0)) return rc;
                    if (int rc = pm_fir_signs_f64(ctx, w1, wn1, T + c->o_out, mo, c->d_bits_q, 0)) return rc;
                }
            } else if (d.modem == PM_MODEM_QPSK) {                              // psk.py:426-476
                if (int rc = grow(ctx, c->d_b, c->b_n, (size_t)n1 * 2)) return rc;      // i (sine branch) | q (cosine branch)
                double *i_arm = c->d_b, *q_arm = c->d_b + n1;
                if (int rc = pm_costas_qpsk(ctx, &c->loop, 1, T + c->o_wave, c->d_a, 0, n1, i_arm, q_arm, n1)) return rc;
                if (int rc = carried(1, mo - 1, i_arm, n1, &w0, &wn0)) return rc;
                if (int rc = carried(2, mo - 1, q_arm, n1, &w1, &wn1)) return rc;
                if (wn0 < mo && !carry) return pm_set_error(PM_ERR_ARG, "pm_chain_run: input shorter than the matched filter");
                ns = wn0 >= mo ? wn0 - mo + 1 : 0;
                if (int rc = bits_for(ns)) return rc;
                if (ns) {
                    if (int rc = pm_fir_signs_f64(ctx, w0, wn0, T + c->o_out, mo, c->d_bits_i, 0)) return rc;
                    if (int rc = pm_fir_signs_f64(ctx, w1, wn1, T + c->o_out, mo, c->d_bits_q, 0)) return rc;
                }
            } else {                                                            // psk.py:162-195, afsk_pll.py:140-170
                if (int rc = grow(ctx, c->d_b, c->b_n, (size_t)n1)) return rc;
                if (d.modem == PM_MODEM_BPSK) {
                    if (int rc = pm_costas_bpsk(ctx, &c->loop, 1, T + c->o_wave, c->d_a, 0, n1, c->d_b, n1)) return rc;
                } else {
                    if (int rc = pm_pll_afsk(ctx, &c->loop, 1, T + c->o_wave, c->d_a, 0, n1, c->d_b, n1)) return rc;
                }
                if (int rc = carried(1, mo - 1, c->d_b, n1, &w0, &wn0)) return rc;
                if (wn0 < mo && !carry) return pm_set_error(PM_ERR_ARG, "pm_chain_run: input shorter than the output filter");
                ns = wn0 >= mo ? wn0 - mo + 1 : 0;
                if (int rc = bits_for(ns)) return rc;
                if (ns) { if (int rc = pm_fir_signs_f64(ctx, w0, wn0, T + c->o_out, mo, c->d_bits_i, 0)) return rc; }
            }
        }
    }
    // slicer.slice: at most one symbol per sample
    const int64_t need = ns * d.slicer.bits_per_symbol / 8 + 5;
    if (int rc = grow(ctx, c->d_data, c->data_n, (size_t)need + 4)) return rc;
    if (int rc = grow(ctx, c->d_addr, c->addr_n, (size_t)need)) return rc;
    pm_slice_job job;
    memset(&job, 0, sizeof(job));
    job.d_bits_i = c->d_bits_i;
    job.d_bits_q = d.quadrature ? c->d_bits_q : nullptr;
    job.n = ns;
    job.params = d.slicer;
    job.d_data = c->d_data;
    job.d_addr = c->d_addr;
    job.cap = need;
    job.h_state = &c->slicer_state;                        // the slicer continues from run to run like the reference's object
    if (ns > 0) {
        if (int rc = pm_slice_batch(ctx, &job, 1)) return rc;
    } else {
        job.count = 0;                                     // nothing reached the slicer yet (carried history still filling)
    }
    if (have_next_hist) c->hist.swap(next_hist);           // the stream has moved on: only now
    for (int k = 0; k < 3; ++k) {
        if (!used_tail[k]) continue;
        c->tail_sel[k] ^= 1;
        c->tail_n[k] = new_tail_n[k];
    }
    c->last_count = job.count;
    return pm_chain_fetch(c, h_data, h_addr, cap, h_count);
}

int pm_chain_fetch(pm_chain *c, uint8_t *h_data, int64_t *h_addr, int64_t cap, int64_t *h_count)
{
    PM_ARG(c != nullptr && h_count != nullptr && cap >= 0);
    if (c->last_count < 0) return pm_set_error(PM_ERR_ARG, "pm_chain_fetch: no run to fetch from");
    pm_ctx *ctx = c->ctx;
    PM_CTX(ctx);
    const int64_t count = c->last_count;
    *h_count = count;
    // the chain's state has moved on with the run; its output stays in device memory until the next run, so a caller whose
    // buffers were too small comes back with larger ones instead of running (and advancing the stream) again
    if (count > cap) return pm_set_error(PM_ERR_CAPACITY, "pm_chain_run: %lld bytes decoded, caller's buffers hold %lld (pm_chain_fetch with larger ones)", (long long)count, (long long)cap);
    PM_ARG(count == 0 || (h_data && h_addr));
    if (count) {
        if (int rc = pm_d2h(ctx, h_data, c->d_data, (size_t)count)) return rc;
        if (int rc = pm_d2h(ctx, h_addr, c->d_addr, (size_t)count * sizeof(int64_t))) return rc;
    }
    return PM_OK;
}

}  // extern "C"
