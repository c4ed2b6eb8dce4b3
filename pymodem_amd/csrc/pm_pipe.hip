// Native pipelined executor for an AFSK chain group (chain_execute.py:30-52 for every chain of a group, recording after recording):
// the stages of pymodem_amd.chain_execute.RecordingPipeline -- demod on the caller's stream, slicer batches on high-priority side
// streams, LFSR + codec + cross-chain de-dup on host threads -- with NO interpreter between them.  Round 2 measured what the Python
// executor costs: one process delivers 235-241 Gsamples/s where two processes sharing the GPU deliver 308 together, and 9-10 ms of host
// CPU per recording, most of it threads handing the interpreter lock around between 20-microsecond native calls.  Here a recording is
// ONE call (pm_pipe_submit: launches only) and its packets come back through ONE call (pm_pipe_wait).
//
// Nothing new is computed: pm_afsk_group_run (shared band-pass + certified sweeps), pm_afsk_sweep_results (+ the exact kernels for a
// sweep whose list overflowed), pm_slice_batch + pm_slice_compact, pm_host_decode_batch + pm_codec_fetch_batch, pm_correlate --
// the entry points the Python executor sequences, in the same order, with the same batching rules (a slicer worker waits on the
// host for the oldest recording's demod event, takes along up to `slice_group` consecutive recordings, ONE worker collects at a
// time; bitmaps live in `slots` rotating sets; a slot is free again when its recording's slicer output is on the host).
#include "pm_common.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <pthread.h>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxChains = 64;
constexpr int kCompactHead = PM_COMPACT_HEAD;

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct HostBlock {                       // page-locked host memory for one slicer batch's compact output
    uint8_t *p = nullptr;
    size_t bytes = 0;
    ~HostBlock() { if (p) (void)hipHostFree(p); }
};

struct Rec {                             // one recording on its way through
    int64_t ticket = 0;
    int slot = 0;
    const int16_t *d_audio = nullptr;
    int64_t n = 0;
    std::vector<int64_t> nout;                          // per chain: the slicer's input length
    hipEvent_t demod_done = nullptr;
    pm_ctx *dctx = nullptr;                             // the context (stream) that demodulates this recording
    int cell = -1;                                      // its sweeps' counter / mailbox words (pm_pipe::cells), until the slicer worker has read them
    // pm_pipe_desc.keep_slices: what the slicer and the LFSR produced, chain by chain (pm_pipe_slices)
    std::vector<std::vector<uint8_t>> kept_data, kept_plain;
    std::vector<std::vector<int64_t>> kept_addr;
    // slicer output (compact form inside `block`)
    std::shared_ptr<HostBlock> block;
    std::vector<int64_t> off, count;
    std::vector<std::vector<int64_t>> full_addr;        // per chain: addresses in full when a step did not fit 16 bits (else empty)
    // result
    // the packet rows: a block out of the pipeline's pool, whose rows hold zeros wherever no packet has written (RowBlock)
    struct RowBlock *rowblock = nullptr;
    pm_packet *rows = nullptr;
    int64_t nrows = 0;
    pm_pipe *owner = nullptr;
    ~Rec();
    std::vector<int64_t> counts, unique_idx;
    std::vector<int32_t> corr;
    int64_t unique = 0;
    int status = PM_OK;
    std::string error;
    double t_submit = 0, t_ready = 0, t_sliced = 0, t_done = 0;
    bool done = false;
};

// Packet rows are 1320 bytes each (a 1280-byte payload field) and a recording of the headline config has 5800 of them: 7.7 MB that
// malloc handed out untouched and munmap took back, every recording, with every payload field's tail zeroed by hand in between --
// more host time than the decoding itself (tools/host_stage_probe.py: fetch 7.7 ms against decode 4.3 ms per recording on one core).
// A block stays with the pipeline instead, and stays CLEAN: zero wherever no packet has written.  Giving it back costs the bytes the
// packets had (their headers and payloads are zeroed again), taking it costs nothing.
struct RowBlock {
    pm_packet *rows = nullptr;
    int64_t cap = 0, used = 0;
    std::vector<int32_t> lens;           // what each row's packet wrote, noted by the library when it wrote it: the caller sees the rows
                                         // (read-only views, but memory it can reach) and a length read back from there could leave dirt in the pool
    ~RowBlock() { free(rows); }
};

}  // namespace

struct pm_pipe {
    pm_ctx *ctx = nullptr;               // the caller's context: the first demod stream
    std::vector<pm_ctx *> demod;         // demod streams, recordings take turns (demod[0] == ctx, the others are the pipeline's)
    std::vector<double *> d_bpf_outs;    // one band-passed stream per demod context
    pm_bpf8_plan *bpf8 = nullptr;        // the band-pass on the int8 matrix pipe (every sweep certified, <= 177 taps; PM_PIPE_BPF8=0: off)
    std::vector<pm_lpf8_plan *> lpf8;    // per sweep: its low-pass there too (tones, <= 113 taps; PM_PIPE_LPF8=0: off), else nullptr
    std::vector<hipEvent_t> handover;    // per slot: the point of the caller's stream a recording submitted to another demod stream starts behind
    std::vector<pm_ctx *> side;          // slicer streams, one per worker
    int nchains = 0, nsweeps = 0, slots = 16, group = 4, min_group = 4, host_threads = 3, decode_threads = 8;
    double address_distance = 0, x_bound = 0;
    int64_t max_samples = 0;
    int mb = 0;
    const double *d_bpf = nullptr;
    std::vector<pm_pipe_chain> chains;
    std::vector<pm_pipe_fir> firs;                       // sign-FIR groups (FSK modems: fsk.py:149-159): chains with sweep = -(f + 1)
    std::vector<int> bit_owner;                          // per chain: the chain whose bitmap it reads (itself, or the first chain of its FIR group)
    std::vector<int> job_owner;                          // per chain: the chain whose slicer job it shares (same bitmap, same slicer: fsk_9600's three)
    std::vector<pm_afsk_sweep_desc> sweeps;              // as given (device pointers stay the caller's; h_gains / h_tones copied below)
    std::vector<std::vector<double>> gains;
    std::vector<pm_afsk_tones> tones;
    std::vector<char> has_tones;
    // device memory owned by the pipeline
    std::vector<uint64_t *> d_bits;                      // [slot * nchains + chain]
    size_t bits_words = 0;
    std::vector<std::vector<std::vector<uint64_t *>>> sweep_bits_store;      // [slot][sweep] -> the device pointers of the sweep's bitmaps (what h_bits wants)
    std::vector<hipEvent_t> slot_event;
    // per worker: slicer output blocks on the device
    struct Work {
        uint8_t *d_out = nullptr, *d_dense = nullptr;
        size_t out_bytes = 0, dense_bytes = 0, tmp_n = 0, host_bytes = 0;
        double *d_tmp = nullptr;
        bool reserved = false;
    };
    // page-locked host blocks for the compact output: a block is in use until the host stages of all its recordings are through;
    // one made in front of a copy costs 8-20 ms there (touch + pin), so they are made once and go round
    std::mutex pool_mu;
    std::vector<HostBlock *> pool;
    std::vector<RowBlock *> row_pool;    // clean row blocks (pool_mu)
    std::vector<RowBlock *> dirty_rows;  // released blocks whose packets are still written in them (pool_mu): a host worker with nothing to decode, or the next
                                         // rows_get that finds no clean block, zeroes them -- not the thread that releases a recording (rows_retire)
    std::atomic<int> dirty_count{0};
    std::vector<Work> work;
    // queues
    std::mutex mu;
    std::condition_variable cv_slot, cv_slice, cv_host, cv_done;
    std::mutex collect_mu;
    std::deque<std::shared_ptr<Rec>> slice_q, host_q;
    std::map<int64_t, std::shared_ptr<Rec>> results;
    std::vector<char> slot_busy;
    // per-recording counter / mailbox words of the certified sweeps: block k = words [k * nsweeps, (k + 1) * nsweeps) of both
    // arrays; a block is the recording's from pm_pipe_submit until its slicer worker has read the mail, then back on the free list
    int *d_cells = nullptr, *h_cells = nullptr;
    unsigned long long *d_lists = nullptr;               // per block: nsweeps lists of kSweepCap entries (what a matrix-pipe sweep's workgroups did not decide themselves)
    std::vector<int> free_cells;
    bool keep_slices = false, trace = false;
    bool host_copy = false;              // PM_PIPE_HOST_COPY: the slicers' compact output through a device block and a copy (round 4), not written to the host block by the kernel
    bool skip_decode = false;            // PM_PIPE_SKIP_DECODE (diagnosis only): the host stage decodes nothing -- what the GPU stages alone sustain
    int64_t next_ticket = 0, submitted = 0, finished = 0;   // submitted: tickets handed out (each is in `results` from then on)
    int64_t promised = 0;                // pm_pipe_submit_many: tickets below this will exist; pm_pipe_wait waits for them to
    bool promise_failed = false;
    bool closing = false;
    std::vector<std::thread> threads;
    // statistics
    std::atomic<int64_t> batches{0}, batch_recordings{0};
    double busy_slice_ms = 0, busy_host_ms = 0, t_origin = now_ms();
};

namespace {

// clean again: what the packets wrote -- the header and len bytes of payload per row -- back to zero
void rows_clean(RowBlock *b)
{
    for (int64_t k = 0; k < b->used; ++k) {
        pm_packet &q = b->rows[k];
        const int32_t len = k < (int64_t)b->lens.size() ? b->lens[(size_t)k] : PM_PKT_MAX;      // (rows nobody noted: the whole field)
        memset(q.data, 0, (size_t)len);
        memset(&q, 0, offsetof(pm_packet, data));
    }
    b->used = 0;
    b->lens.clear();
}

// one released block zeroed and back among the clean ones -> true; nothing waiting -> false
bool rows_clean_one(pm_pipe *p)
{
    RowBlock *b = nullptr;
    {
        std::unique_lock<std::mutex> lk(p->pool_mu);
        if (p->dirty_rows.empty()) return false;
        b = p->dirty_rows.back();
        p->dirty_rows.pop_back();
        p->dirty_count.store((int)p->dirty_rows.size());
    }
    rows_clean(b);
    std::unique_lock<std::mutex> lk(p->pool_mu);
    if (p->row_pool.size() < 64) p->row_pool.push_back(b);
    else delete b;
    return true;
}

RowBlock *rows_get(pm_pipe *p, int64_t need)
{
    RowBlock *b = nullptr;
    bool dirty = false;
    {
        std::unique_lock<std::mutex> lk(p->pool_mu);
        for (size_t i = 0; i < p->row_pool.size(); ++i)
            if (p->row_pool[i]->cap >= need) {
                b = p->row_pool[i];
                p->row_pool.erase(p->row_pool.begin() + (ptrdiff_t)i);
                break;
            }
        for (size_t i = 0; !b && i < p->dirty_rows.size(); ++i)       // none clean: a released one that nobody has come round to yet
            if (p->dirty_rows[i]->cap >= need) {
                b = p->dirty_rows[i];
                p->dirty_rows.erase(p->dirty_rows.begin() + (ptrdiff_t)i);
                p->dirty_count.store((int)p->dirty_rows.size());
                dirty = true;
            }
    }
    if (dirty) rows_clean(b);
    if (!b) {
        b = new RowBlock();
        b->cap = std::max<int64_t>(need + need / 4, 64);
        b->rows = (pm_packet *)calloc((size_t)b->cap, sizeof(pm_packet));
        if (!b->rows) { delete b; return nullptr; }
    }
    b->used = need;
    return b;
}

// A recording's rows go back: as they are, onto the list of blocks to be zeroed.  The zeroing (the bytes the packets had: 0.16 ms for the
// headline's 5800 rows) used to run on the thread that released the recording, inside pm_pipe_release's lock on the pipeline -- the
// caller's thread, which at the end of a run releases the last batch's recordings one after the other while nothing else is left to do.
void rows_retire(pm_pipe *p, RowBlock *b)
{
    static const bool inline_clean = getenv("PM_PIPE_CLEAN_INLINE") != nullptr;      // the releasing thread zeroes the block itself (A/B runs)
    if (inline_clean) {
        rows_clean(b);
        std::unique_lock<std::mutex> lk(p->pool_mu);
        if (p->row_pool.size() < 64) p->row_pool.push_back(b);
        else delete b;
        return;
    }
    {
        std::unique_lock<std::mutex> lk(p->pool_mu);
        if (p->dirty_rows.size() + p->row_pool.size() >= 64) {
            lk.unlock();
            delete b;
            return;
        }
        p->dirty_rows.push_back(b);
        p->dirty_count.store((int)p->dirty_rows.size());
    }
    p->cv_host.notify_one();
}

Rec::~Rec()
{
    if (rowblock && owner) rows_retire(owner, rowblock);
    else delete rowblock;
}

int fail(Rec &r, int rc)
{
    char buf[512];
    pm_last_error(buf, sizeof(buf));
    r.status = rc;
    r.error = buf;
    return rc;
}

std::shared_ptr<HostBlock> block_get(pm_pipe *p, size_t need, size_t room)
{
    HostBlock *b = nullptr;
    {
        std::unique_lock<std::mutex> lk(p->pool_mu);
        for (size_t i = 0; i < p->pool.size(); ++i)
            if (p->pool[i]->bytes >= need) {
                b = p->pool[i];
                p->pool.erase(p->pool.begin() + i);
                break;
            }
    }
    if (!b) {
        b = new HostBlock();
        b->bytes = std::max<size_t>(std::max(need, room), 64);
        if (hipHostMalloc((void **)&b->p, b->bytes, hipHostMallocDefault) != hipSuccess) {
            b->p = nullptr;
            delete b;
            return nullptr;
        }
    }
    return std::shared_ptr<HostBlock>(b, [p](HostBlock *q) {
        std::unique_lock<std::mutex> lk(p->pool_mu);
        p->pool.push_back(q);
    });
}

// The exact chain of one modem on `side` (a sweep whose list of uncertain samples overflowed: digital silence, audio far below the
// stated bound): band-pass, this modem's own four correlators, low-pass + sign -- what resolve_sweeps does in the Python executor.
int exact_chain(pm_pipe *p, pm_ctx *side, pm_pipe::Work &w, const Rec &r, int chain)
{
    const pm_pipe_chain &c = p->chains[chain];
    const pm_afsk_sweep_desc &s = p->sweeps[c.sweep];
    const int64_t nb = r.n - p->mb + 1, nc = nb - s.m + 1;
    const size_t need = (size_t)nb + (size_t)nc + 16;
    if (w.tmp_n < need) {
        if (w.d_tmp) { if (int rc = pm_free(side, w.d_tmp)) return rc; w.d_tmp = nullptr; w.tmp_n = 0; }
        void *q = nullptr;
        if (int rc = pm_malloc(side, need * sizeof(double), &q)) return rc;
        w.d_tmp = (double *)q;
        w.tmp_n = need;
    }
    double *bpf = w.d_tmp, *corr = w.d_tmp + ((nb + 1) & ~(int64_t)1);
    if (int rc = pm_fir_valid_i16(side, r.d_audio, r.n, p->d_bpf, p->mb, bpf, 0)) return rc;
    const double *space = s.d_space + (size_t)c.slot * 2 * s.m;
    if (int rc = pm_afsk_correlate(side, bpf, nb, s.d_mark_i, s.d_mark_q, space, space + s.m, s.m, corr)) return rc;
    return pm_fir_signs_f64(side, corr, nc, s.d_lpf, s.ml, p->d_bits[(size_t)r.slot * p->nchains + chain], 0);
}

void slice_worker(pm_pipe *p, int wi)
{
    (void)pthread_setname_np(pthread_self(), "pm-slice");
    pm_ctx *side = p->side[wi];
    pm_pipe::Work &w = p->work[wi];
    (void)hipSetDevice(side->device);
    for (;;) {
        std::vector<std::shared_ptr<Rec>> batch;
        {
            // ONE worker at a time puts a batch together: consecutive recordings, started when its last demod is done
            std::unique_lock<std::mutex> collect(p->collect_mu);
            {
                std::unique_lock<std::mutex> lk(p->mu);
                p->cv_slice.wait(lk, [&] { return p->closing || !p->slice_q.empty(); });
                if (p->slice_q.empty()) return;
                batch.push_back(p->slice_q.front());
                p->slice_q.pop_front();
            }
            // (a demod stream that faulted: the recording's mailbox words hold whatever the block's last user left -- its bitmaps are not sliced)
            if (hipEventSynchronize(batch[0]->demod_done) != hipSuccess) fail(*batch[0], pm_set_error(PM_ERR_HIP, "the demod stage of recording %lld failed", (long long)batch[0]->ticket));
            batch[0]->t_ready = now_ms();
            while ((int)batch.size() < p->group) {
                std::shared_ptr<Rec> nxt;
                {
                    std::unique_lock<std::mutex> lk(p->mu);
                    if (p->slice_q.empty()) break;
                    nxt = p->slice_q.front();
                    const bool ready = hipEventQuery(nxt->demod_done) == hipSuccess;
                    if (!ready && (int)batch.size() >= p->min_group) break;
                    p->slice_q.pop_front();
                }
                // already queued on the GPU: a batch of four costs what one costs
                if (hipEventSynchronize(nxt->demod_done) != hipSuccess) fail(*nxt, pm_set_error(PM_ERR_HIP, "the demod stage of recording %lld failed", (long long)nxt->ticket));
                nxt->t_ready = now_ms();
                batch.push_back(nxt);
            }
        }
        const double t0 = now_ms();
        const int nch = p->nchains, nb = (int)batch.size();
        int rc = PM_OK;
        // What the certified sweeps left undecided is decided here, before the slicers read the bitmaps: a list that is not empty (a
        // workgroup with more uncertain samples than it takes on itself, digital silence) by the exact chain per entry, a list that
        // overflowed by the exact kernels over the whole recording.  Normally every word is zero and nothing is launched.
        std::vector<int> cells_back;
        for (auto &r : batch) {
            if (!p->nsweeps) break;
            // the recording's demod event has been waited for: its sweeps have left their counts in the mailbox words
            std::vector<int64_t> unc(p->nsweeps);
            for (int s = 0; s < p->nsweeps; ++s) unc[s] = ((volatile int *)p->h_cells)[(size_t)r->cell * p->nsweeps + s];
            for (int s = 0; s < p->nsweeps && !r->status; ++s) {
                if (unc[s] <= 0) continue;
                if (unc[s] <= kSweepCap) {
                    if ((rc = pm_afsk_sweep_exact_list(side, r->d_audio, p->d_bpf, p->mb, &p->sweeps[s], p->sweep_bits_store[r->slot][s].data(),
                                                       p->d_lists + ((size_t)r->cell * p->nsweeps + s) * kSweepCap, p->d_cells + (size_t)r->cell * 2 * p->nsweeps + p->nsweeps + s)))
                        fail(*r, rc);
                    continue;
                }
                for (int c = 0; c < nch && !r->status; ++c)
                    if (p->chains[c].sweep == s && (rc = exact_chain(p, side, w, *r, c))) fail(*r, rc);
            }
            cells_back.push_back(r->cell);                   // (its list may be read by a launch on this worker's stream: back when the batch is through)
            r->cell = -1;
        }
        // all slicers of the batch in one pm_slice_batch (groups of <= 64 jobs), compact form, one copy to the host
        // (chains that slice the same bitmap with the same slicer share a job: jidx maps (recording, chain) to it)
        std::vector<std::pair<int, int>> act;
        std::vector<int> jidx((size_t)nb * nch, -1);
        for (int b = 0; b < nb; ++b) {
            for (int c = 0; c < nch; ++c)
                if (p->job_owner[c] == c) { jidx[(size_t)b * nch + c] = (int)act.size(); act.emplace_back(b, c); }
            for (int c = 0; c < nch; ++c) jidx[(size_t)b * nch + c] = jidx[(size_t)b * nch + p->job_owner[c]];
        }
        std::vector<pm_slice_job> jobs(act.size());
        std::vector<pm_slicer_state> states(act.size());
        std::vector<int64_t> caps(jobs.size());
        auto lay_out = [&](bool tight) -> size_t {
            size_t at = 0;
            for (size_t j = 0; j < act.size(); ++j) {
                const int b = act[j].first, c = act[j].second;
                const pm_slicer_params &sp = p->chains[c].slicer;
                const int64_t no = batch[b]->nout[c];
                int64_t cap = no * sp.bits_per_symbol / 8 + 5;
                if (tight) cap = std::min<int64_t>(cap, (int64_t)(no * sp.bits_per_symbol / (8.0 * sp.samples_per_symbol) * 1.5) + 64);
                caps[j] = cap;
                at += (size_t)cap * 8;
            }
            for (size_t j = 0; j < caps.size(); ++j) at += ((size_t)caps[j] + 4 + 7) / 8 * 8;
            return at;
        };
        auto run = [&](bool tight) -> int {
            const size_t bytes = lay_out(tight);
            const size_t reserve = bytes * (size_t)p->group / (size_t)nb;       // sized for a full batch at first use: no free in mid-stream
            if (w.out_bytes < bytes) {
                if (w.d_out) { if (int rc2 = pm_free(side, w.d_out)) return rc2; w.d_out = nullptr; w.out_bytes = 0; }
                void *q = nullptr;
                if (int rc2 = pm_malloc(side, std::max(bytes, reserve), &q)) return rc2;
                w.d_out = (uint8_t *)q;
                w.out_bytes = std::max(bytes, reserve);
            }
            size_t a_at = 0, d_at = 0;
            for (size_t j = 0; j < caps.size(); ++j) d_at += (size_t)caps[j] * 8;
            for (size_t j = 0; j < act.size(); ++j) {
                const int b = act[j].first, c = act[j].second;
                pm_slice_job &q = jobs[j];
                memset(&q, 0, sizeof(q));
                memset(&states[j], 0, sizeof(pm_slicer_state));
                q.d_bits_i = p->d_bits[(size_t)batch[b]->slot * nch + p->bit_owner[c]];
                q.d_bits_q = nullptr;
                q.n = batch[b]->nout[c];
                q.params = p->chains[c].slicer;
                q.d_addr = (int64_t *)(w.d_out + a_at);
                q.d_data = w.d_out + d_at;
                q.cap = caps[j];
                q.h_state = &states[j];
                a_at += (size_t)caps[j] * 8;
                d_at += ((size_t)caps[j] + 4 + 7) / 8 * 8;
            }
            for (size_t j0 = 0; j0 < jobs.size(); j0 += 64) {
                const int nj = (int)std::min<size_t>(64, jobs.size() - j0);
                if (int rc2 = pm_slice_batch(side, jobs.data() + j0, nj)) return rc2;
            }
            return PM_OK;
        };
        const double t_a = now_ms();
        rc = run(true);
        if (rc == PM_ERR_CAPACITY) rc = run(false);
        const double t_b = now_ms();          // a stream with more than 1.5x its nominal symbol count: the full bound
        std::shared_ptr<HostBlock> hb;
        double t_c = t_b, t_d = t_b;
        std::vector<int64_t> offs(jobs.size());
        if (!rc) {
            size_t dense_cap = 0;
            for (size_t j = 0; j < caps.size(); ++j)
                dense_cap += kCompactHead + ((size_t)2 * jobs[j].count + 7) / 8 * 8 + ((size_t)jobs[j].count + 7) / 8 * 8;
            // The compact kernel writes the batch's output STRAIGHT into a page-locked host block (the device reaches it over the link:
            // hipHostMalloc memory is mapped) -- no device-side block, no copy operation behind the kernel.  As a copy of its own the
            // 8 MB took 0.2 ms in steady state but 4-12 ms whenever the demod streams had a dozen recordings queued (the first batches
            // of a run, every batch of a short one): the driver's 20-step figure was 0.81 or 1.1 ms per step by the luck of that
            // (gpurun_out r5m; PM_PIPE_HOST_COPY=1 brings the copy back for comparison).
            const bool direct = !p->host_copy;
            if (!direct && w.dense_bytes < dense_cap) {
                if (w.d_dense) { (void)pm_free(side, w.d_dense); w.d_dense = nullptr; w.dense_bytes = 0; }
                void *q = nullptr;
                const size_t want = dense_cap * 3 / 2 * (size_t)p->group / (size_t)nb + 4096;
                if (!(rc = pm_malloc(side, want, &q))) { w.d_dense = (uint8_t *)q; w.dense_bytes = want; }
            }
            // (host blocks in one size class: what a full batch needs, half as much again)
            const size_t padded = dense_cap + 256 * ((jobs.size() + 63) / 64);
            if (w.host_bytes < padded) w.host_bytes = padded * 3 / 2 * (size_t)p->group / (size_t)nb + 4096;
            if (!rc && !w.reserved) {
                // first batch of this worker: the stream's work block (checkpoints, symbol bitmaps, lists: ~150 MB per recording) sized
                // for a full batch now -- growing it later is a free + malloc in the middle of the pipeline -- and the host blocks made
                w.reserved = true;
                size_t have = 0;
                if (!(rc = pm_ctx_scratch(side, 0, &have))) rc = pm_ctx_scratch(side, have * (size_t)p->group / (size_t)nb, nullptr);
                std::vector<std::shared_ptr<HostBlock>> warm;
                for (int k = 0; k < 4 && !rc; ++k) {
                    warm.push_back(block_get(p, w.host_bytes, w.host_bytes));
                    if (!warm.back()) rc = pm_set_error(PM_ERR_HIP, "hipHostMalloc of %zu bytes failed", w.host_bytes);
                }
            }
            if (!rc) {
                hb = block_get(p, padded, w.host_bytes);
                if (!hb) rc = pm_set_error(PM_ERR_HIP, "hipHostMalloc of %zu bytes failed", w.host_bytes);
            }
            t_c = now_ms();
            uint8_t *const dst = direct ? (hb ? hb->p : nullptr) : w.d_dense;
            const size_t dst_bytes = direct ? (hb ? hb->bytes : 0) : w.dense_bytes;
            size_t used = 0, at = 0;
            for (size_t j0 = 0; j0 < jobs.size() && !rc; j0 += 64) {
                const int nj = (int)std::min<size_t>(64, jobs.size() - j0);
                size_t u = 0;
                rc = pm_slice_compact(side, jobs.data() + j0, nj, dst + at, dst_bytes - at, offs.data() + j0, &u);
                for (int k = 0; k < nj; ++k) offs[j0 + k] += (int64_t)at;
                at += (u + 255) & ~(size_t)255;
                used = at;
            }
            t_d = now_ms();
            if (!rc && !direct && hipMemcpyAsync(hb->p, w.d_dense, used, hipMemcpyDeviceToHost, side->stream) != hipSuccess) rc = pm_set_error(PM_ERR_HIP, "copy of the slicer output failed");
            if (!rc && hipStreamSynchronize(side->stream) != hipSuccess) rc = pm_set_error(PM_ERR_HIP, "slicer stream failed");
        }
        // a failed batch may have walkers or copies enqueued that still read the bitmaps: the slots go back when the stream is empty
        if (rc) (void)hipStreamSynchronize(side->stream);
        const double t1 = now_ms();
        if (p->trace)
            fprintf(stderr, "[pm_pipe] worker %d batch of %d (first %lld): demod done at %.2f, collected %.2f, sweeps checked +%.2f, sliced +%.2f, on the host +%.2f ms (host block +%.2f, compact enqueued +%.2f, until it is there +%.2f)\n", wi, nb,
                    (long long)batch[0]->ticket, batch[0]->t_ready - p->t_origin, t0 - p->t_origin, t_a - t0, t_b - t_a, t1 - t_b, t_c - t_b, t_d - t_c, t1 - t_d);
        for (int b = 0; b < nb; ++b) {
            Rec &r = *batch[b];
            if (rc && !r.status) fail(r, rc);
            if (!r.status) {
                r.block = hb;
                r.off.assign(nch, 0);
                r.count.assign(nch, 0);
                r.full_addr.assign(nch, {});
                for (int c = 0; c < nch && !r.status; ++c) {
                    const size_t j = (size_t)jidx[(size_t)b * nch + c];
                    r.off[c] = offs[j];
                    r.count[c] = jobs[j].count;
                    const uint8_t *flags = hb->p + offs[j] + 16;
                    bool wide = false;
                    for (int f = 0; f < 64 && jobs[j].count; ++f) wide = wide || flags[f] != 0;
                    if (wide) {                                  // a step beyond 16 bits: this stream's addresses in full
                        r.full_addr[c].resize((size_t)jobs[j].count);
                        if (hipMemcpy(r.full_addr[c].data(), jobs[j].d_addr, (size_t)jobs[j].count * 8, hipMemcpyDeviceToHost) != hipSuccess)
                            fail(r, pm_set_error(PM_ERR_HIP, "copy of the slicer addresses failed"));
                    }
                }
            }
            r.t_sliced = t1;
        }
        {
            std::unique_lock<std::mutex> lk(p->mu);
            for (int cell : cells_back) p->free_cells.push_back(cell);
            for (auto &r : batch) {
                p->slot_busy[r->slot] = 0;                       // the bitmaps are consumed and the output is on the host
                p->host_q.push_back(r);
            }
            p->busy_slice_ms += t1 - t0;
        }
        p->batches++;
        p->batch_recordings += nb;
        p->cv_slot.notify_all();
        p->cv_host.notify_all();
    }
}

void host_worker(pm_pipe *p)
{
    (void)pthread_setname_np(pthread_self(), "pm-host");
    for (;;) {
        std::shared_ptr<Rec> rp;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_host.wait(lk, [&] { return p->closing || !p->host_q.empty() || p->dirty_count.load() > 0; });
            if (p->host_q.empty()) {
                if (p->closing) return;
                lk.unlock();
                (void)rows_clean_one(p);                     // nothing to decode: a released block's packets back to zero
                continue;
            }
            rp = p->host_q.front();
            p->host_q.pop_front();
        }
        Rec &r = *rp;
        const double t0 = now_ms();
        const int nch = p->nchains;
        std::vector<pm_codec *> codecs(nch, nullptr);
        if (!r.status) {
            std::vector<pm_host_job> jobs(nch);
            int rc = PM_OK;
            for (int c = 0; c < nch && !rc; ++c) {
                const pm_pipe_chain &ch = p->chains[c];
                rc = pm_codec_create(ch.codec_kind, ch.crc, ch.disable_rs, ch.min_dist, ch.sync_tol, ch.source_decoder, &codecs[c]);
                pm_host_job &j = jobs[c];
                memset(&j, 0, sizeof(j));
                j.codec = codecs[c];
                const int64_t cnt = r.count[c];
                const uint8_t *base = r.block->p + r.off[c];
                j.n = p->skip_decode ? 0 : cnt;
                j.h_data = base + kCompactHead + ((size_t)2 * cnt + 7) / 8 * 8;
                if (!r.full_addr[c].empty()) {
                    j.h_addr = r.full_addr[c].data();
                } else {
                    j.h_addr = nullptr;
                    j.h_addr_delta = (const uint16_t *)(base + kCompactHead);
                    memcpy(&j.addr_first, base, 8);
                }
                j.lfsr_poly = ch.lfsr_poly;
                j.lfsr_state = 0;
                j.lfsr_invert = ch.lfsr_invert;
            }
            if (!rc && p->keep_slices) {
                // the slicer's output as it reached the host, addresses in full, and room for the LFSR's bytes (pm_host_job.h_plain)
                r.kept_data.resize(nch);
                r.kept_addr.resize(nch);
                r.kept_plain.resize(nch);
                for (int c = 0; c < nch; ++c) {
                    pm_host_job &j = jobs[c];
                    r.kept_data[c].assign(j.h_data, j.h_data + j.n);
                    r.kept_addr[c].resize((size_t)j.n);
                    if (j.h_addr) {
                        std::copy(j.h_addr, j.h_addr + j.n, r.kept_addr[c].begin());
                    } else {
                        int64_t a = j.addr_first;
                        for (int64_t i = 0; i < j.n; ++i) r.kept_addr[c][(size_t)i] = (a += j.h_addr_delta[i]);
                    }
                    r.kept_plain[c].resize((size_t)j.n);
                    j.h_plain = r.kept_plain[c].data();
                }
            }
            if (!rc) rc = pm_host_decode_batch(jobs.data(), nch, p->decode_threads);
            if (p->trace) {
                std::string line = "[pm_pipe] recording " + std::to_string(r.ticket) + " rc " + std::to_string(rc) + ": slicer bytes / packets per chain";
                for (int c = 0; c < nch; ++c) line += " " + std::to_string(r.count[c]) + "/" + std::to_string(jobs[c].pending);
                line += "; host stage began " + std::to_string(t0 - p->t_origin) + ", decode done " + std::to_string(now_ms() - p->t_origin);
                fprintf(stderr, "%s\n", line.c_str());
            }
            if (!rc) {
                r.counts.resize(nch);
                int64_t total = 0;
                for (int c = 0; c < nch; ++c) total += (r.counts[c] = jobs[c].pending);
                r.owner = p;
                r.rowblock = rows_get(p, std::max<int64_t>(total, 1));
                r.rows = r.rowblock ? r.rowblock->rows : nullptr;
                r.nrows = total;
                if (!r.rows) rc = pm_set_error(PM_ERR_ARG, "out of host memory for %lld packet rows", (long long)total);
                if (!rc) rc = pm_codec_fetch_batch_clean(codecs.data(), r.counts.data(), nch, r.rows, p->decode_threads);
                if (r.rowblock) {                             // the packets' lengths, before anyone else can touch the rows
                    r.rowblock->lens.resize((size_t)total);
                    for (int64_t k = 0; k < total; ++k) {
                        const int32_t len = r.rows[k].len;
                        r.rowblock->lens[(size_t)k] = len < 0 ? 0 : len > PM_PKT_MAX ? PM_PKT_MAX : len;
                    }
                }
                if (!rc) {
                    // PacketMetaArray.Correlate over the chains in config order (packet_meta.py:230-271)
                    r.unique_idx.resize((size_t)std::max<int64_t>(total, 1));
                    r.corr.resize((size_t)std::max<int64_t>(total, 1));
                    const int64_t k = total && p->address_distance >= 0 ? pm_correlate(r.rows, r.counts.data(), nch, p->address_distance, r.unique_idx.data(), r.corr.data(),
                                                           (int64_t)r.corr.size())
                                            : 0;
                    if (k < 0) rc = (int)k;
                    else r.unique = k;
                }
            }
            if (rc) fail(r, rc);
        }
        for (pm_codec *c : codecs)
            if (c) (void)pm_codec_destroy(c);
        r.block.reset();
        r.t_done = now_ms();
        if (p->trace) fprintf(stderr, "[pm_pipe] recording %lld done at %.2f\n", (long long)r.ticket, r.t_done - p->t_origin);
        {
            std::unique_lock<std::mutex> lk(p->mu);
            r.done = true;
            p->finished++;
            p->busy_host_ms += r.t_done - t0;
        }
        p->cv_done.notify_all();
    }
}

}  // namespace

extern "C" {

int pm_pipe_destroy(pm_pipe *p)
{
    if (!p) return PM_OK;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        // everything submitted goes through first
        p->cv_done.wait(lk, [&] { return p->finished == p->submitted; });
        p->closing = true;
    }
    p->cv_slice.notify_all();
    p->cv_host.notify_all();
    for (auto &t : p->threads) t.join();
    pm_ctx *ctx = p->ctx;
    (void)pm_ctx_sync(ctx);
    for (uint64_t *b : p->d_bits)
        if (b) (void)pm_free(ctx, b);
    for (size_t i = 1; i < p->demod.size(); ++i) (void)pm_ctx_sync(p->demod[i]);
    for (double *b : p->d_bpf_outs)
        if (b) (void)pm_free(ctx, b);
    pm_bpf8_plan_destroy(p->bpf8);
    for (pm_lpf8_plan *q : p->lpf8) pm_lpf8_plan_destroy(q);
    for (size_t i = 1; i < p->demod.size(); ++i) (void)pm_ctx_destroy(p->demod[i]);
    for (hipEvent_t e : p->handover)
        if (e) (void)hipEventDestroy(e);
    if (p->d_cells) (void)hipFree(p->d_cells);
    if (p->d_lists) (void)hipFree(p->d_lists);
    if (p->h_cells) (void)hipHostFree(p->h_cells);
    for (size_t i = 0; i < std::min(p->work.size(), p->side.size()); ++i) {      // (a create that failed half way: fewer contexts than blocks)
        pm_ctx *s = p->side[i];
        for (void *q : {(void *)p->work[i].d_out, (void *)p->work[i].d_dense, (void *)p->work[i].d_tmp})
            if (q) (void)pm_free(s, q);
    }
    for (hipEvent_t e : p->slot_event)
        if (e) (void)hipEventDestroy(e);
    for (pm_ctx *s : p->side) (void)pm_ctx_destroy(s);
    p->results.clear();
    for (HostBlock *b : p->pool) delete b;
    for (RowBlock *b : p->row_pool) delete b;
    for (RowBlock *b : p->dirty_rows) delete b;
    delete p;
    return PM_OK;
}

int pm_pipe_create(pm_ctx *ctx, const pm_pipe_desc *desc, pm_pipe **out)
{
    PM_CTX(ctx);
    PM_ARG(desc != nullptr && out != nullptr);
    const pm_pipe_desc &d = *desc;
    PM_ARG(d.nsweeps >= 0 && d.nsweeps <= 16 && d.nfirs >= 0 && d.nfirs <= 16 && d.nsweeps + d.nfirs >= 1 && d.chains && d.nchains >= 1 && d.nchains <= kMaxChains);
    PM_ARG(d.nsweeps == 0 || (d.d_bpf && d.mb >= 1 && d.x_bound > 0 && d.sweeps));
    PM_ARG(d.nfirs == 0 || d.firs != nullptr);
    for (int f = 0; f < d.nfirs; ++f) PM_ARG(d.firs[f].d_taps != nullptr && d.firs[f].m >= 1 && d.firs[f].m <= d.max_samples);
    PM_ARG(d.nsweeps == 0 || d.max_samples >= d.mb);
    pm_pipe *p = new pm_pipe();
    p->ctx = ctx;
    p->nchains = d.nchains;
    p->nsweeps = d.nsweeps;
    p->slots = d.slots > 0 ? std::min(d.slots, 32) : 16;
    // two: with the sweeps' sums on the matrix pipe three recordings' demod kernels at once take longer than three in a row
    // (demod alone 0.67 ms per recording with three streams, 0.49 with two, 0.84 with one)
    const int nd = d.demod_streams > 0 ? std::min(d.demod_streams, 4) : 2;
    p->slots = std::max(2, p->slots);
    p->keep_slices = d.keep_slices != 0;
    p->trace = getenv("PM_PIPE_TRACE") != nullptr;
    p->skip_decode = getenv("PM_PIPE_SKIP_DECODE") != nullptr;
    p->host_copy = getenv("PM_PIPE_HOST_COPY") != nullptr;
    p->group = d.slice_group > 0 ? std::min(d.slice_group, 16) : 4;
    p->min_group = d.slice_min_group > 0 ? std::min(d.slice_min_group, p->group) : p->group;
    // recordings in the host stage at a time: four for the headline's eight chains (each spreads its chains over decode_threads), twelve
    // for fsk_9600's three -- 9 / 10 / 12 / 14 there: 0.37-0.39 / 0.25-0.32 / 0.26-0.31 / 0.20-0.33 ms per step (profiles/r05_executor_knobs.txt)
    p->host_threads = d.host_threads > 0 ? d.host_threads : std::max(2, std::min(12, 36 / d.nchains));
    p->decode_threads = d.decode_threads > 0 ? d.decode_threads : d.nchains;
    p->address_distance = d.address_distance;
    p->x_bound = d.x_bound;
    p->max_samples = d.max_samples;
    p->mb = d.nsweeps ? d.mb : 1;
    p->d_bpf = d.d_bpf;
    if (d.nfirs) p->firs.assign(d.firs, d.firs + d.nfirs);
    p->bit_owner.assign(d.nchains, 0);
    p->chains.assign(d.chains, d.chains + d.nchains);
    if (d.nsweeps) p->sweeps.assign(d.sweeps, d.sweeps + d.nsweeps);
    p->gains.resize(d.nsweeps);
    p->tones.resize(d.nsweeps);
    p->has_tones.assign(d.nsweeps, 0);
    int rc = PM_OK;
    do {
        for (int s = 0; s < d.nsweeps; ++s) {
            pm_afsk_sweep_desc &w = p->sweeps[s];
            if (!(w.groups >= 1 && w.groups <= PM_AFSK_GROUP_MAX && w.h_gains && w.m >= 1 && w.ml >= 1)) { rc = pm_set_error(PM_ERR_ARG, "pm_pipe_create: sweep %d is malformed", s); break; }
            p->gains[s].assign(w.h_gains, w.h_gains + w.groups);
            w.h_gains = p->gains[s].data();
            if (w.h_tones) {
                p->tones[s] = *w.h_tones;
                p->has_tones[s] = 1;
            }
            w.h_tones = p->has_tones[s] ? &p->tones[s] : nullptr;
            w.h_bits = nullptr;
        }
        if (rc) break;
        // every chain names its sweep and its place in it exactly once
        std::vector<std::vector<int>> seen(d.nsweeps);
        for (int s = 0; s < d.nsweeps; ++s) seen[s].assign(p->sweeps[s].groups, -1);
        std::vector<int> fir_first(d.nfirs, -1);
        for (int c = 0; c < d.nchains && !rc; ++c) {
            const pm_pipe_chain &ch = p->chains[c];
            p->bit_owner[c] = c;
            if (ch.sweep < 0) {                                  // a sign-FIR group's chain: reads the bitmap of the group's first chain
                const int f = -ch.sweep - 1;
                if (f >= d.nfirs || ch.slicer.bits_per_symbol != 1) {
                    rc = pm_set_error(PM_ERR_ARG, "pm_pipe_create: chain %d does not name a sign-FIR group (or is not a binary-slicer chain)", c);
                } else {
                    if (fir_first[f] < 0) fir_first[f] = c;
                    p->bit_owner[c] = fir_first[f];
                }
                continue;
            }
            if (ch.sweep >= d.nsweeps || ch.slot < 0 || ch.slot >= p->sweeps[ch.sweep].groups || seen[ch.sweep][ch.slot] >= 0 ||
                ch.slicer.bits_per_symbol != 1)
                rc = pm_set_error(PM_ERR_ARG, "pm_pipe_create: chain %d does not name a free place of a sweep (or is not a binary-slicer chain)", c);
            else
                seen[ch.sweep][ch.slot] = c;
        }
        p->job_owner.assign(d.nchains, 0);
        for (int c = 0; c < d.nchains; ++c) {
            p->job_owner[c] = c;
            for (int e = 0; e < c; ++e)
                if (p->bit_owner[e] == p->bit_owner[c] && memcmp(&p->chains[e].slicer, &p->chains[c].slicer, sizeof(pm_slicer_params)) == 0) {
                    p->job_owner[c] = p->job_owner[e];
                    break;
                }
        }
        for (int s = 0; s < d.nsweeps && !rc; ++s)
            for (int g = 0; g < p->sweeps[s].groups; ++g)
                if (seen[s][g] < 0) rc = pm_set_error(PM_ERR_ARG, "pm_pipe_create: place %d of sweep %d has no chain", g, s);
        if (rc) break;
        for (int f = 0; f < d.nfirs && !rc; ++f)
            if (fir_first[f] < 0) rc = pm_set_error(PM_ERR_ARG, "pm_pipe_create: sign-FIR group %d has no chain", f);
        if (rc) break;
        const int64_t nb = d.nsweeps ? d.max_samples - d.mb + 1 : 1;
        void *q = nullptr;
        p->demod.push_back(ctx);
        for (int k = 1; k < nd && !rc; ++k) {
            pm_ctx *c = nullptr;
            if (!(rc = pm_ctx_create_prio(ctx->device, 0, &c))) p->demod.push_back(c);
        }
        for (int k = 0; k < nd && !rc; ++k)
            if (!(rc = pm_malloc(ctx, (size_t)nb * sizeof(double), &q))) p->d_bpf_outs.push_back((double *)q);
        if (rc) break;
        {
            // every consumer of the band-passed stream is a certified sweep: it may be a value with a bound (pm_bpf8.hip)
            bool all = true;
            for (int s = 0; s < d.nsweeps; ++s) all = all && p->has_tones[s];
            const char *e = getenv("PM_PIPE_BPF8");
            if (d.nsweeps && all && d.mb + 15 <= 192 && !(e && e[0] == '0')) {
                std::vector<double> h((size_t)d.mb);
                if (hipMemcpy(h.data(), d.d_bpf, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                    rc = pm_set_error(PM_ERR_HIP, "pm_pipe_create: reading the band-pass taps back failed");
                else
                    rc = pm_bpf8_plan_create(ctx, h.data(), d.mb, &p->bpf8);
            }
        }
        p->lpf8.assign(d.nsweeps, nullptr);
        {
            const char *e = getenv("PM_PIPE_LPF8");
            for (int s = 0; s < d.nsweeps && !rc && !(e && e[0] == '0'); ++s) {
                const pm_afsk_sweep_desc &w = p->sweeps[s];
                if (!p->has_tones[s] || w.ml + 15 > 128 || w.m < 2) continue;
                std::vector<double> h((size_t)w.ml);
                if (hipMemcpy(h.data(), w.d_lpf, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                    rc = pm_set_error(PM_ERR_HIP, "pm_pipe_create: reading the low-pass taps back failed");
                else
                    rc = pm_lpf8_plan_create(ctx, h.data(), w.ml, &p->lpf8[s]);
            }
        }
        if (rc) break;
        p->bits_words = (size_t)(d.max_samples + 63) / 64 + 2;
        p->d_bits.assign((size_t)p->slots * d.nchains, nullptr);
        for (size_t i = 0; i < p->d_bits.size() && !rc; ++i) {
            if (!(rc = pm_malloc(ctx, p->bits_words * 8, &q))) p->d_bits[i] = (uint64_t *)q;
        }
        if (rc) break;
        p->sweep_bits_store.resize(p->slots);
        for (int sl = 0; sl < p->slots; ++sl) {
            p->sweep_bits_store[sl].resize(d.nsweeps);
            for (int s = 0; s < d.nsweeps; ++s) {
                p->sweep_bits_store[sl][s].resize(p->sweeps[s].groups);
                for (int g = 0; g < p->sweeps[s].groups; ++g) p->sweep_bits_store[sl][s][g] = p->d_bits[(size_t)sl * d.nchains + seen[s][g]];
            }
        }
        if (d.nsweeps) {
            // one block of counter / mailbox words per recording between submission and its slicer batch: never more than `slots`
            // (device words per recording and sweep: the live counter and the count as mailed)
            const size_t words = (size_t)p->slots * d.nsweeps;
            if (hipMalloc((void **)&p->d_cells, 2 * words * sizeof(int)) != hipSuccess || hipMemset(p->d_cells, 0, 2 * words * sizeof(int)) != hipSuccess ||
                hipMalloc((void **)&p->d_lists, words * kSweepCap * sizeof(unsigned long long)) != hipSuccess ||
                hipHostMalloc((void **)&p->h_cells, words * sizeof(int), hipHostMallocDefault) != hipSuccess) {
                rc = pm_set_error(PM_ERR_HIP, "pm_pipe_create: no memory for the sweep counters");
                break;
            }
            memset(p->h_cells, 0, words * sizeof(int));
            for (int k = p->slots - 1; k >= 0; --k) p->free_cells.push_back(k);
        }
        p->slot_busy.assign(p->slots, 0);
        p->slot_event.assign(p->slots, nullptr);
        p->handover.assign(p->slots, nullptr);
        for (int sl = 0; sl < p->slots && !rc; ++sl)
            if (hipEventCreateWithFlags(&p->slot_event[sl], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&p->handover[sl], hipEventDisableTiming) != hipSuccess)
                rc = pm_set_error(PM_ERR_HIP, "hipEventCreate failed");
        if (rc) break;
        const int workers = d.slice_workers > 0 ? std::min(d.slice_workers, 4) : 2;
        p->work.resize(workers);
        for (int wk = 0; wk < workers && !rc; ++wk) {
            pm_ctx *s = nullptr;
            // (confining the slicer streams to 4 / 8 / 12 / 16 CUs per XCD was measured: 1.64 / 1.13 / 0.98 / 0.89 ms per step against
            // 0.81 -- the walkers need the width; at 8 per XCD they alone take 1.1 ms per recording, i.e. the slicers are worth a
            // third of the whole GPU's time per recording)
            if (!(rc = pm_ctx_create_prio(ctx->device, 1, &s))) {
                p->side.push_back(s);
                rc = pm_slicer_tune(s, 16384);
                // walkers of 32 k samples: the slicers are a third of the GPU's work now, and a walker re-walks its merge length
                // (10-20 k samples) whatever the chunk -- 1 + m/L lane-steps per sample; longer chunks cost depth, not work
                if (!rc) rc = pm_slicer_limits(s, 512);
            }
        }
        if (rc) break;
        for (int wk = 0; wk < workers; ++wk) p->threads.emplace_back(slice_worker, p, wk);
        for (int h = 0; h < p->host_threads; ++h) p->threads.emplace_back(host_worker, p);
    } while (0);
    if (rc) {
        char keep[512];
        pm_last_error(keep, sizeof(keep));
        pm_pipe_destroy(p);
        return pm_set_error(rc, "%s", keep);
    }
    *out = p;
    return PM_OK;
}

int pm_pipe_submit(pm_pipe *p, const int16_t *d_audio, int64_t n, int64_t *h_ticket)
{
    PM_ARG(p != nullptr && d_audio != nullptr && h_ticket != nullptr);
    PM_CTX(p->ctx);
    PM_ARG(n >= p->mb && n <= p->max_samples);
    // Everything that can refuse the recording by its shape is checked before a ticket exists.
    auto r = std::make_shared<Rec>();
    r->d_audio = d_audio;
    r->n = n;
    const int64_t nb = n - p->mb + 1;
    r->nout.resize(p->nchains);
    for (int c = 0; c < p->nchains; ++c) {
        if (p->chains[c].sweep < 0) {
            r->nout[c] = n - p->firs[-p->chains[c].sweep - 1].m + 1;
        } else {
            const pm_afsk_sweep_desc &w = p->sweeps[p->chains[c].sweep];
            r->nout[c] = nb - w.m - w.ml + 2;
        }
        if (r->nout[c] < 1) return pm_set_error(PM_ERR_ARG, "pm_pipe_submit: %lld samples are fewer than the filters of chain %d need", (long long)n, c);
    }
    {
        // From here on the ticket exists: it is in `results` (a wait finds it), it counts as submitted (drain and destroy wait for
        // it), and a launch that fails below finishes it with that error instead of leaving a hole in the ticket sequence.
        std::unique_lock<std::mutex> lk(p->mu);
        r->ticket = p->next_ticket++;
        p->results[r->ticket] = r;
        p->submitted++;
        r->slot = (int)(r->ticket % p->slots);
        const double tw = now_ms();
        p->cv_slot.wait(lk, [&] { return !p->slot_busy[r->slot] && (!p->nsweeps || !p->free_cells.empty()); });      // the recording that used this slot `slots` submissions ago is sliced
        p->slot_busy[r->slot] = 1;
        if (p->nsweeps) {
            r->cell = p->free_cells.back();
            p->free_cells.pop_back();
        }
        if (p->trace) fprintf(stderr, "[pm_pipe] submit %lld at %.2f (waited %.2f ms for its slot)\n", (long long)r->ticket, now_ms() - p->t_origin, now_ms() - tw);
    }
    *h_ticket = r->ticket;
    p->cv_done.notify_all();                                 // waits for a promised ticket: it exists now
    r->t_submit = now_ms();
    std::vector<pm_afsk_sweep_desc> sw(p->sweeps);
    for (int s = 0; s < p->nsweeps; ++s) sw[s].h_bits = p->sweep_bits_store[r->slot][s].data();
    const size_t di = (size_t)(r->ticket % (int64_t)p->demod.size());
    r->dctx = p->demod[di];
    int rc = PM_OK;
    if (r->dctx != p->ctx) {
        // the recording is in place at this point of the CALLER's stream (an upload enqueued there, an event it waits for): the other
        // demod stream starts behind that point
        if (hipEventRecord(p->handover[r->slot], p->ctx->stream) != hipSuccess || hipStreamWaitEvent(r->dctx->stream, p->handover[r->slot], 0) != hipSuccess)
            rc = pm_set_error(PM_ERR_HIP, "handing the recording to demod stream %zu failed", di);
    }
    if (!rc && p->nsweeps) {
        const pm_sweep_cells cells{p->d_cells + (size_t)r->cell * 2 * p->nsweeps, p->h_cells + (size_t)r->cell * p->nsweeps,
                                   p->d_lists + (size_t)r->cell * p->nsweeps * kSweepCap};
        rc = pm_afsk_group_run_plan(r->dctx, d_audio, n, p->d_bpf, p->mb, p->d_bpf_outs[di], p->x_bound, sw.data(), p->nsweeps, nullptr,
                                    ((uintptr_t)d_audio & 15) == 0 ? p->bpf8 : nullptr, p->lpf8.data(), &cells);
    }
    for (int c = 0; c < p->nchains && !rc; ++c)              // sign-FIR groups: one launch per group into its first chain's bitmap (fsk.py:149-159)
        if (p->chains[c].sweep < 0 && p->bit_owner[c] == c) {
            const pm_pipe_fir &f = p->firs[-p->chains[c].sweep - 1];
            rc = pm_fir_signs_i16(r->dctx, d_audio, n, f.d_taps, f.m, p->d_bits[(size_t)r->slot * p->nchains + c], f.flags);
        }
    if (!rc && hipEventRecord(p->slot_event[r->slot], r->dctx->stream) != hipSuccess) rc = pm_set_error(PM_ERR_HIP, "hipEventRecord failed");
    r->demod_done = p->slot_event[r->slot];
    if (rc) {
        // Some of the recording's launches may be on the stream: they write this slot's bitmaps and this block's counters, so both go
        // back only when the stream has passed them.  The ticket is finished, with the error.
        fail(*r, rc);
        (void)hipStreamSynchronize(r->dctx->stream);
        if (r->cell >= 0) (void)hipMemset(p->d_cells + (size_t)r->cell * 2 * p->nsweeps, 0, 2 * sizeof(int) * (size_t)p->nsweeps);      // (a stage that stopped half way never mailed and reset them)
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->slot_busy[r->slot] = 0;
            if (r->cell >= 0) p->free_cells.push_back(r->cell);
            r->cell = -1;
            r->done = true;
            r->t_done = r->t_sliced = r->t_ready = now_ms();
            p->finished++;
        }
        p->cv_slot.notify_all();
        p->cv_done.notify_all();
        return pm_set_error(rc, "%s", r->error.c_str());
    }
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->slice_q.push_back(r);
    }
    p->cv_slice.notify_one();
    return PM_OK;
}

int pm_pipe_promise(pm_pipe *p, int count, int64_t *h_first_ticket)
{
    // the next `count` tickets will be submitted (by a pm_pipe_submit_many about to start on another thread): waits for them may begin
    PM_ARG(p != nullptr && count >= 1 && h_first_ticket != nullptr);
    std::unique_lock<std::mutex> lk(p->mu);
    *h_first_ticket = p->next_ticket;
    p->promised = p->next_ticket + count;
    p->promise_failed = false;
    return PM_OK;
}

int pm_pipe_submit_many(pm_pipe *p, const int16_t *const *d_audio, const int64_t *n, int count, int64_t *h_first_ticket)
{
    // `count` recordings in order from ONE call: a host whose submitting thread shares an interpreter lock with the threads that take
    // the results (bench.py with the packet exchange behind the executor: 0.96 ms per pm_pipe_submit, most of it waiting for the lock
    // on the way back) stays out of the way for the whole run.  Tickets first .. first + count - 1; pm_pipe_wait on one of them
    // that is not submitted yet waits for it.  One submitting thread at a time.
    PM_ARG(p != nullptr && d_audio != nullptr && n != nullptr && count >= 1 && h_first_ticket != nullptr);
    {
        std::unique_lock<std::mutex> lk(p->mu);
        *h_first_ticket = p->next_ticket;
        p->promised = std::max(p->promised, p->next_ticket + count);
        p->promise_failed = false;
    }
    int rc = PM_OK;
    for (int i = 0; i < count && !rc; ++i) {
        int64_t t = -1;
        rc = pm_pipe_submit(p, d_audio[i], n[i], &t);
        if (rc && t >= 0) (void)pm_pipe_release(p, t);       // a ticket that finished with the launch error: nobody will ask for it
    }
    if (rc) {
        char keep[512];
        pm_last_error(keep, sizeof(keep));
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->promise_failed = true;
            p->promised = p->next_ticket;
        }
        p->cv_done.notify_all();
        return pm_set_error(rc, "%s", keep);
    }
    return PM_OK;
}

int pm_pipe_wait(pm_pipe *p, int64_t ticket, pm_pipe_result *out)
{
    PM_ARG(p != nullptr && out != nullptr);
    std::shared_ptr<Rec> r;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        // a ticket pm_pipe_submit_many has promised and not yet reached: wait until it exists (every ticket below next_ticket is in
        // `results` from the moment it was handed out until pm_pipe_release)
        p->cv_done.wait(lk, [&] { return ticket < p->next_ticket || ticket >= p->promised || p->promise_failed; });
        auto it = p->results.find(ticket);
        if (it == p->results.end()) return pm_set_error(PM_ERR_ARG, "pm_pipe_wait: ticket %lld is unknown (or released)", (long long)ticket);
        r = it->second;
        p->cv_done.wait(lk, [&] { return r->done; });
    }
    memset(out, 0, sizeof(*out));
    out->ticket = ticket;
    out->status = r->status;
    out->rows = r->nrows;
    out->h_rows = r->rows;
    out->h_counts = r->counts.data();
    out->unique = r->unique;
    out->h_unique_idx = r->unique_idx.data();
    out->h_corr_decoders = r->corr.data();
    out->ms_to_demod_done = r->t_ready - r->t_submit;
    out->ms_to_sliced = r->t_sliced - r->t_submit;
    out->ms_to_done = r->t_done - r->t_submit;
    out->done_at_ms = r->t_done - p->t_origin;
    if (r->status) return pm_set_error(r->status, "%s", r->error.c_str());
    return PM_OK;
}

int pm_pipe_release(pm_pipe *p, int64_t ticket)
{
    PM_ARG(p != nullptr);
    std::shared_ptr<Rec> gone;                              // (what the recording held goes back after the lock: ~Rec hands its rows to the pool)
    {
        std::unique_lock<std::mutex> lk(p->mu);
        auto it = p->results.find(ticket);
        if (it == p->results.end()) return PM_OK;
        if (!it->second->done) return pm_set_error(PM_ERR_ARG, "pm_pipe_release: recording %lld is still in flight", (long long)ticket);
        gone = std::move(it->second);
        p->results.erase(it);
    }
    return PM_OK;
}

int pm_pipe_drain(pm_pipe *p)
{
    PM_ARG(p != nullptr);
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->finished == p->submitted; });
    return PM_OK;
}

int pm_pipe_slots(pm_pipe *p) { return p ? p->slots : 0; }

int pm_pipe_slices(pm_pipe *p, int64_t ticket, int chain, const uint8_t **h_data, const int64_t **h_addr, const uint8_t **h_plain, int64_t *h_count)
{
    PM_ARG(p != nullptr && h_count != nullptr && chain >= 0 && chain < p->nchains);
    if (!p->keep_slices) return pm_set_error(PM_ERR_ARG, "pm_pipe_slices: the pipeline was made without keep_slices");
    std::unique_lock<std::mutex> lk(p->mu);
    auto it = p->results.find(ticket);
    if (it == p->results.end() || !it->second->done) return pm_set_error(PM_ERR_ARG, "pm_pipe_slices: recording %lld is unknown, released or still in flight", (long long)ticket);
    const Rec &r = *it->second;
    if (r.status) return pm_set_error(r.status, "%s", r.error.c_str());
    PM_ARG((size_t)chain < r.kept_data.size());
    *h_count = (int64_t)r.kept_data[chain].size();
    if (h_data) *h_data = r.kept_data[chain].data();
    if (h_addr) *h_addr = r.kept_addr[chain].data();
    if (h_plain) *h_plain = r.kept_plain[chain].data();
    return PM_OK;
}

int pm_pipe_bitmap(pm_pipe *p, int64_t ticket, int chain, uint64_t *h_words, int64_t words)
{
    PM_ARG(p != nullptr && h_words != nullptr && chain >= 0 && chain < p->nchains && words >= 1 && (size_t)words <= p->bits_words);
    if (!p->keep_slices) return pm_set_error(PM_ERR_ARG, "pm_pipe_bitmap: the pipeline was made without keep_slices");
    int slot = -1;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        auto it = p->results.find(ticket);
        if (it == p->results.end() || !it->second->done) return pm_set_error(PM_ERR_ARG, "pm_pipe_bitmap: recording %lld is unknown, released or still in flight", (long long)ticket);
        if (it->second->status) return pm_set_error(it->second->status, "%s", it->second->error.c_str());
        if (p->next_ticket - ticket > p->slots) return pm_set_error(PM_ERR_ARG, "pm_pipe_bitmap: recording %lld's bitmaps have been overwritten (%d slots)", (long long)ticket, p->slots);
        slot = it->second->slot;
    }
    PM_CTX(p->ctx);
    PM_HIP(hipMemcpy(h_words, p->d_bits[(size_t)slot * p->nchains + p->bit_owner[chain]], (size_t)words * 8, hipMemcpyDeviceToHost));
    return PM_OK;
}

int pm_pipe_stats(pm_pipe *p, int64_t *h_batches, int64_t *h_batch_recordings, double *h_slice_busy_ms, double *h_host_busy_ms)
{
    PM_ARG(p != nullptr);
    std::unique_lock<std::mutex> lk(p->mu);
    if (h_batches) *h_batches = p->batches.load();
    if (h_batch_recordings) *h_batch_recordings = p->batch_recordings.load();
    if (h_slice_busy_ms) *h_slice_busy_ms = p->busy_slice_ms;
    if (h_host_busy_ms) *h_host_busy_ms = p->busy_host_ms;
    return PM_OK;
}

pm_ctx *pm_pipe_side_ctx(pm_pipe *p, int worker)
{
    return (p && worker >= 0 && worker < (int)p->side.size()) ? p->side[worker] : nullptr;
}

pm_ctx *pm_pipe_demod_ctx(pm_pipe *p, int k)
{
    return (p && k >= 0 && k < (int)p->demod.size()) ? p->demod[k] : nullptr;
}

}  // extern "C"
