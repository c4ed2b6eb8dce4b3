// Context, memory, error and timer entry points of the C ABI (include/pymodem_amd.h).
#include "pm_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <mutex>

static thread_local char g_err[512] = "";

int pm_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {
struct TuneKey { const char *name, *env; int pm_tuning::*field; };
const TuneKey kTuneKeys[] = {
    {"fir_no_short", "PM_FIR_NO_SHORT", &pm_tuning::fir_no_short}, {"fuse_run", "PM_FUSE_RUN", &pm_tuning::fuse_run},
    {"afsk_unfused", "PM_AFSK_UNFUSED", &pm_tuning::afsk_unfused}, {"afsk_lpf8", "PM_AFSK_LPF8", &pm_tuning::afsk_lpf8},
    {"loop_wide", "PM_LOOP_WIDE", &pm_tuning::loop_wide}, {"lbatch_tail", "PM_LBATCH_TAIL", &pm_tuning::lbatch_tail},
    {"agc_trace", "PM_AGC_TRACE", &pm_tuning::agc_trace},
    {"slicer_max_chunk_words", "PM_SLICER_MAX_CHUNK_WORDS", &pm_tuning::slicer_max_chunk_words},
    {"slicer_chunk_words", "PM_SLICER_CHUNK_WORDS", &pm_tuning::slicer_chunk_words},
    {"slicer_quantum_words", "PM_SLICER_QUANTUM_WORDS", &pm_tuning::slicer_quantum_words},
    {"slicer_compare_step", "PM_SLICER_COMPARE_STEP", &pm_tuning::slicer_compare_step},
    {"slicer_mask_step", "PM_SLICER_MASK_STEP", &pm_tuning::slicer_mask_step},
    {"slicer_compiled_step", "PM_SLICER_COMPILED_STEP", &pm_tuning::slicer_compiled_step},
    {"slicer_trace", "PM_SLICER_TRACE", &pm_tuning::slicer_trace}, {"slicer_no_setprio", "PM_SLICER_NO_SETPRIO", &pm_tuning::slicer_no_setprio},
    {"fir8", "PM_FIR8", &pm_tuning::fir8}, {"bpf8_max", "PM_BPF8_MAX", &pm_tuning::bpf8_max}, {"loop_agc", "PM_LOOP_AGC", &pm_tuning::loop_agc}, {"loop_vec", "PM_LOOP_VEC", &pm_tuning::loop_vec},
    {"lbatch_loop_cus", "PM_LBATCH_LOOP_CUS", &pm_tuning::lbatch_loop_cus}, {"agc_rows_prio", "PM_AGC_ROWS_PRIO", &pm_tuning::agc_rows_prio},
    {"sweep_no_tail", "PM_SWEEP_NO_TAIL", &pm_tuning::sweep_no_tail}, {"afsk_split", "PM_AFSK_SPLIT", &pm_tuning::afsk_split}, {"fused_lds_pad", "PM_FUSED_LDS_PAD", &pm_tuning::fused_lds_pad}, {"sweep_lds_templates", "PM_SWEEP_LDS_TEMPLATES", &pm_tuning::sweep_lds_templates},
};
}  // namespace

// The environment is looked at here and nowhere on a launch path.  A variable that is set without a number ("PM_SLICER_TRACE=") counts as 1.
pm_tuning pm_tuning_from_env()
{
    pm_tuning t;
    for (const TuneKey &k : kTuneKeys)
        if (const char *e = getenv(k.env)) t.*(k.field) = (*e >= '0' && *e <= '9') || *e == '-' ? atoi(e) : 1;
    if (const char *e = getenv("PM_LOOP_LDS_MIN")) t.loop_lds_min = atol(e);
    return t;
}

extern "C" {

int pm_version(void) { return PM_VERSION; }

int pm_ctx_tune(pm_ctx *c, const char *name, int64_t value)
{
    PM_ARG(c != nullptr && name != nullptr);
    if (!strcmp(name, "loop_lds_min")) { c->tune.loop_lds_min = value; return PM_OK; }
    for (const TuneKey &k : kTuneKeys)
        if (!strcmp(name, k.name)) { c->tune.*(k.field) = (int)value; return PM_OK; }
    return pm_set_error(PM_ERR_ARG, "pm_ctx_tune: no switch named '%s'", name);
}

int pm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pm_last_error(char *buf, size_t cap)
{
    if (!buf || cap == 0) return PM_ERR_ARG;
    strncpy(buf, g_err, cap - 1);
    buf[cap - 1] = 0;
    return PM_OK;
}

int pm_ctx_create(int device, pm_ctx **out) { return pm_ctx_create_prio(device, 0, out); }

int pm_ctx_create_prio(int device, int high_priority, pm_ctx **out)
{
    PM_ARG(out != nullptr);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return pm_set_error(PM_ERR_NODEV, "no HIP device visible: the HIP path cannot run (there is no CPU fallback)");
    PM_ARG(device >= 0 && device < n);
    PM_HIP(hipSetDevice(device));
    pm_ctx *c = new pm_ctx();
    c->device = device;
    if (high_priority) {
        int lo = 0, hi = 0;                                   // numerically lower = higher priority
        PM_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        PM_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
    } else {
        PM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    PM_HIP(hipEventCreate(&c->ev0));
    PM_HIP(hipEventCreate(&c->ev1));
    PM_HIP(hipHostMalloc(&c->h_pinned, PM_PINNED_BYTES, hipHostMallocDefault));
    *out = c;
    return PM_OK;
}

int pm_ctx_create_cumask(int device, const uint32_t *cu_mask, int nwords, pm_ctx **out)
{
    PM_ARG(out != nullptr && cu_mask != nullptr && nwords >= 1);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return pm_set_error(PM_ERR_NODEV, "no HIP device visible: the HIP path cannot run (there is no CPU fallback)");
    PM_ARG(device >= 0 && device < n);
    PM_HIP(hipSetDevice(device));
    bool any = false;
    for (int k = 0; k < nwords; ++k) any = any || cu_mask[k] != 0;
    PM_ARG(any);
    pm_ctx *c = new pm_ctx();
    c->device = device;
    hipError_t e = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)nwords, cu_mask);
    if (e != hipSuccess) {
        delete c;
        return pm_set_error(PM_ERR_HIP, "hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
    }
    PM_HIP(hipEventCreate(&c->ev0));
    PM_HIP(hipEventCreate(&c->ev1));
    PM_HIP(hipHostMalloc(&c->h_pinned, PM_PINNED_BYTES, hipHostMallocDefault));
    *out = c;
    return PM_OK;
}

int pm_device_cus(int device)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 0;
    return p.multiProcessorCount;
}

int pm_ctx_destroy(pm_ctx *c)
{
    if (!c) return PM_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    pm_prof_fold(c);
    for (hipEvent_t e : c->prof_free) (void)hipEventDestroy(e);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_sweep) (void)hipFree(c->d_sweep);
    if (c->h_sweep) (void)hipHostFree(c->h_sweep);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return PM_OK;
}

int pm_event_record(pm_ctx *c, void **event)
{
    PM_CTX(c);
    PM_ARG(event != nullptr);
    if (*event == nullptr) {
        hipEvent_t e;
        PM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        *event = (void *)e;
    }
    PM_HIP(hipEventRecord((hipEvent_t)*event, c->stream));
    return PM_OK;
}

int pm_event_wait(pm_ctx *c, void *event)
{
    PM_CTX(c);
    PM_ARG(event != nullptr);
    PM_HIP(hipStreamWaitEvent(c->stream, (hipEvent_t)event, 0));
    return PM_OK;
}

int pm_event_query(void *event)
{
    if (!event) return pm_set_error(PM_ERR_ARG, "bad argument: event (%s:%d)", __FILE__, __LINE__);
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return pm_set_error(PM_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}

int pm_event_sync(void *event)
{
    PM_ARG(event != nullptr);
    PM_HIP(hipEventSynchronize((hipEvent_t)event));
    return PM_OK;
}

int pm_event_sync_relaxed(void *event, int poll_us)
{
    PM_ARG(event != nullptr && poll_us >= 1);
    // hipEventSynchronize spins: fine for the sub-millisecond waits of the pipelined executor, a core burnt for seconds behind a run
    // of the carrier-loop engine
    for (;;) {
        hipError_t e = hipEventQuery((hipEvent_t)event);
        if (e == hipSuccess) return PM_OK;
        if (e != hipErrorNotReady) return pm_set_error(PM_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
        usleep((useconds_t)poll_us);
    }
}

int pm_event_destroy(void *event)
{
    if (event) PM_HIP(hipEventDestroy((hipEvent_t)event));
    return PM_OK;
}

int pm_ctx_sync(pm_ctx *c)
{
    PM_CTX(c);
    PM_ARG(c != nullptr);
    PM_HIP(hipStreamSynchronize(c->stream));
    return PM_OK;
}

void *pm_ctx_stream(pm_ctx *c) { return c ? (void *)c->stream : nullptr; }

int pm_malloc(pm_ctx *c, size_t bytes, void **d_out)
{
    PM_CTX(c);
    PM_ARG(c != nullptr && d_out != nullptr);
    PM_HIP(hipSetDevice(c->device));
    PM_HIP(hipMalloc(d_out, bytes ? bytes : 8));
    return PM_OK;
}

int pm_free(pm_ctx *c, void *p)
{
    PM_CTX(c);
    PM_ARG(c != nullptr);
    if (!p) return PM_OK;
    // The context's own stream is waited for; a buffer that another context still reads (sign bitmaps by a slicer stream, the slicers'
    // output by the copy stream) is the CALLER's to keep alive until that reader is done -- the pipelined executor's slot bookkeeping
    // does.  (A device-wide wait here stalls a worker for as long as the other threads keep their streams busy: 15-25 ms measured.)
    PM_HIP(hipStreamSynchronize(c->stream));
    PM_HIP(hipFree(p));
    return PM_OK;
}

int pm_h2d(pm_ctx *c, void *d_dst, const void *h_src, size_t bytes)
{
    PM_CTX(c);
    PM_ARG(c != nullptr && (bytes == 0 || (d_dst && h_src)));
    if (bytes) PM_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    // pageable source: the runtime has staged the bytes when the call returns, the caller may reuse h_src
    return PM_OK;
}

int pm_d2h(pm_ctx *c, void *h_dst, const void *d_src, size_t bytes)
{
    PM_CTX(c);
    PM_ARG(c != nullptr && (bytes == 0 || (h_dst && d_src)));
    if (bytes) PM_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    PM_HIP(hipStreamSynchronize(c->stream));
    return PM_OK;
}

int pm_host_pin(pm_ctx *c, void *h_block, size_t bytes)
{
    PM_CTX(c);
    PM_ARG(c != nullptr && h_block != nullptr && bytes > 0);
    PM_HIP(hipHostRegister(h_block, bytes, hipHostRegisterDefault));
    return PM_OK;
}

int pm_host_unpin(void *h_block)
{
    if (!h_block) return PM_OK;
    PM_HIP(hipHostUnregister(h_block));
    return PM_OK;
}

int pm_d2d(pm_ctx *c, void *d_dst, const void *d_src, size_t bytes)
{
    PM_CTX(c);
    PM_ARG(bytes == 0 || (d_dst && d_src));
    if (bytes) PM_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return PM_OK;
}

int pm_memset(pm_ctx *c, void *d_dst, int value, size_t bytes)
{
    PM_CTX(c);
    PM_ARG(c != nullptr && (bytes == 0 || d_dst));
    if (bytes) PM_HIP(hipMemsetAsync(d_dst, value, bytes, c->stream));
    return PM_OK;
}

int pm_timer_start(pm_ctx *c)
{
    PM_CTX(c);
    PM_ARG(c != nullptr);
    PM_HIP(hipEventRecord(c->ev0, c->stream));
    return PM_OK;
}

int pm_timer_stop(pm_ctx *c, float *ms)
{
    PM_CTX(c);
    PM_ARG(c != nullptr && ms != nullptr);
    PM_HIP(hipEventRecord(c->ev1, c->stream));
    PM_HIP(hipEventSynchronize(c->ev1));
    PM_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return PM_OK;
}

}  // extern "C"

PmProf::PmProf(pm_ctx *ctx, int k) : c(ctx), cls(k)
{
    if (!c->prof_on) return;
    auto grab = [&]() {
        hipEvent_t e = nullptr;
        if (!c->prof_free.empty()) { e = c->prof_free.back(); c->prof_free.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    };
    a = grab();
    b = grab();
    if (a && b) (void)hipEventRecord(a, c->stream);
}

PmProf::~PmProf()
{
    if (!c->prof_on || !a || !b) return;
    (void)hipEventRecord(b, c->stream);
    c->prof_pending.push_back({a, b, cls});
    if (c->prof_pending.size() >= 2048) pm_prof_fold(c);
}

// One reference event per device, recorded (and finished) when profiling is first switched on there: launch intervals of every context
// of the device are measured from it, so that launches on different streams can be laid over each other (pm_prof_intervals).
static hipEvent_t g_prof_ref[64];
static std::mutex g_prof_ref_mu;

static hipEvent_t prof_ref(pm_ctx *c, bool make)
{
    std::lock_guard<std::mutex> lk(g_prof_ref_mu);
    if (c->device < 0 || c->device >= 64) return nullptr;
    if (!g_prof_ref[c->device] && make) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, c->stream) == hipSuccess && hipEventSynchronize(e) == hipSuccess) g_prof_ref[c->device] = e;
    }
    return g_prof_ref[c->device];
}

int pm_prof_fold(pm_ctx *c)
{
    if (c->prof_pending.empty()) return PM_OK;
    PM_HIP(hipStreamSynchronize(c->stream));
    hipEvent_t ref = prof_ref(c, false);
    for (auto &p : c->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->prof_ms[p.cls] += ms;
            c->prof_n[p.cls] += 1;
            float at = 0;
            if (ref && c->prof_iv[p.cls].size() < (size_t)1 << 22 && hipEventElapsedTime(&at, ref, p.a) == hipSuccess) {
                c->prof_iv[p.cls].push_back(at);
                c->prof_iv[p.cls].push_back(at + ms);
            }
        }
        c->prof_free.push_back(p.a);
        c->prof_free.push_back(p.b);
    }
    c->prof_pending.clear();
    return PM_OK;
}

extern "C" int pm_prof_enable(pm_ctx *c, int on)
{
    PM_ARG(c != nullptr);
    if (int rc = pm_prof_fold(c)) return rc;
    for (int k = 0; k < PM_K_COUNT; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; c->prof_bytes[k] = 0; c->prof_flops[k] = 0; c->prof_iv[k].clear(); }
    if (on) {
        PM_HIP(hipSetDevice(c->device));
        (void)prof_ref(c, true);
    }
    c->prof_on = on != 0;
    return PM_OK;
}

extern "C" int pm_prof_intervals(pm_ctx *c, int cls, double *h_start_ms, double *h_end_ms, int64_t cap, int64_t *h_n)
{
    PM_ARG(c != nullptr && cls >= 0 && cls < PM_K_COUNT && h_n != nullptr && cap >= 0 && (cap == 0 || (h_start_ms && h_end_ms)));
    if (int rc = pm_prof_fold(c)) return rc;
    const int64_t n = (int64_t)c->prof_iv[cls].size() / 2;
    for (int64_t k = 0; k < std::min(n, cap); ++k) {
        h_start_ms[k] = c->prof_iv[cls][2 * k];
        h_end_ms[k] = c->prof_iv[cls][2 * k + 1];
    }
    *h_n = n;
    return PM_OK;
}

extern "C" int pm_prof_read(pm_ctx *c, int cls, double *total_ms, int64_t *launches)
{
    PM_ARG(c != nullptr && cls >= 0 && cls < PM_K_COUNT);
    if (int rc = pm_prof_fold(c)) return rc;
    if (total_ms) *total_ms = c->prof_ms[cls];
    if (launches) *launches = c->prof_n[cls];
    return PM_OK;
}

extern "C" int pm_prof_work(pm_ctx *c, int cls, double *bytes, double *flops)
{
    PM_ARG(c != nullptr && cls >= 0 && cls < PM_K_COUNT);
    if (bytes) *bytes = c->prof_bytes[cls];
    if (flops) *flops = c->prof_flops[cls];
    return PM_OK;
}

extern "C" int pm_ctx_scratch(pm_ctx *c, size_t reserve_bytes, size_t *h_bytes)
{
    PM_CTX(c);
    if (reserve_bytes) { if (int rc = pm_scratch_reserve(c, reserve_bytes)) return rc; }
    if (h_bytes) *h_bytes = c->scratch_bytes;
    return PM_OK;
}

int pm_scratch_reserve(pm_ctx *c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return PM_OK;
    PM_HIP(hipStreamSynchronize(c->stream));
    if (c->d_scratch) PM_HIP(hipFree(c->d_scratch));
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    size_t want = bytes + bytes / 2 + 4096;      // grow in big steps: every growth is a stream wait and a hipFree
    PM_HIP(hipMalloc(&c->d_scratch, want));
    c->scratch_bytes = want;
    return PM_OK;
}
