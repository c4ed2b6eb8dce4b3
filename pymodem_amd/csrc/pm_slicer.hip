// Symbol-timing slicers (BinarySlicer.slice slicer.py:59-107, QuadratureSlicer.slice slicer.py:193-242)
// evaluated chunk-parallel on the sign bitmap(s) of the demodulated stream(s), many streams per launch.
//
// The reference recurrence, per sample k (1-based address k+1):
//     clk += 1.0;  if (clk >= sps/2 - 0.5) { clk -= sps;  take a symbol from sign(x[k]) }
//     if sign(x[k]) != sign(x[k-1]):  clk *= lock_rate
// is sequential in `clk` only, and two runs that see the same zero crossings contract towards each other by lock_rate per
// crossing until they are the same double (~140 crossings at 0.77).  Every stream is cut into chunks of L samples and every
// chunk gets a WALKER: a lane that starts at the chunk's first sample (walker 0 of a stream from the true state, the others from
// a cold 0.0), executes the reference's operations in the reference's order, and leaves behind, per 64-sample word, the clock it
// entered the word with (a checkpoint) and the bitmap of the samples at which it took a symbol.  A walker does not stop at the
// end of its chunk: it walks on into the next chunk, overwriting what that chunk's own walker left there, until it enters a
// word with exactly the clock the trail in front of it entered it with -- from there on its future IS that trail, word for word,
// and it retires.  All walkers advance in lockstep (one launch = the same number of words for every live walker), so a walker
// is always a whole chunk behind the one in front of it: it only ever reads words that were written by an earlier launch and
// nobody writes behind it but walkers that started further back.  Walker 0 is the sequential run; by induction the trail it
// merged into is the sequential run from that word on, and so on down the stream: when no walker is left, checkpoints and symbol
// bitmaps are those of the sequential run, bit for bit.  The cost is one pass plus the merge lengths (N (1 + m/L) lane-steps
// instead of one full pass per L/m of merge length), and live walkers are compacted into a dense list after every launch, so a
// launch has as many waves as there are walkers still on their way.  No crossings at all (digital silence) degrades to every
// walker reaching the end of its stream: sequential depth, never a wrong answer.
//
// A lone wave pays both the loop-carried chain (add -> compare -> select -> multiply, ~45 cycles with the latencies measured
// by tools/ubench) and the issue of the ~11 VALU instructions of a step (~5.5 cycles each): ~85 cycles per sample.
// After the last walker a count/scan/pack pipeline, chunked on its own (finely), turns symbol bitmap + sign bitmap(s) into
// bytes and the 1-based address of each byte's last symbol.  A slicer object's state (clock, last sign, open byte, address
// count, differential state) enters and leaves through pm_slicer_state.
#include "pm_common.h"
#include "pm_slicer_step.inc"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kMaxJobs = 64;
constexpr int kBlock = 256;     // four waves: one per SIMD when the grid is one workgroup per CU
constexpr size_t kPinnedBytes = PM_PINNED_BYTES;
constexpr int kShortLaunches = 10;    // launches of qwords words after the walkers' own chunks; 4 x qwords from then on

struct JobDev {
    const uint64_t *bi, *bq;       // sign bitmaps (bq null for binary)
    int64_t n, nwords;
    int64_t chunk0, nchunks;       // global chunk range of this stream
    int64_t word0;                 // offset of this stream in the global symbol bitmap
    double thr, sps, lock;
    double tp;                     // the smallest clock for which fl(clk + 1.0) >= thr (step32c)
    int bps, mask, quad, pad;
    int demap[16];
    uint32_t *data32;
    int64_t *addr;
    int64_t cap;
    // state carried in from the previous call on the same slicer object (all zero for a fresh one)
    double clk0;
    int li0, lq0;                  // 1 if the previous sample was >= 0
    int nb0, wb0;                  // bits already shifted into the working byte, and their value
    int sreg0;                     // previous symbol (quadrature)
    int64_t addr0;
    uint32_t *tail;                // the trailing partial byte of this call, left-aligned
};

__device__ __forceinline__ uint64_t dbits(double v) { return (uint64_t)__double_as_longlong(v); }
__device__ __forceinline__ double bitsd(uint64_t v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ double mkdouble(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }

// The last job whose first chunk is <= gc (chunk0 ascends along the table; streams without chunks share their successor's).
// Bisection: every probe is a dependent load at the head of a launch that is itself only tens of microseconds long, and the
// linear scan needed as many of them as the walker's stream has predecessors in the batch (up to 63).
__device__ __forceinline__ int find_job(const JobDev *jobs, int njobs, int64_t gc)
{
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (gc >= jobs[mid].chunk0) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// 32 samples, most significant bit first: zc = crossing flags (bit 31 = first sample).  Returns the symbol flags in
// the same orientation.  clk is updated in place.
//
// The step is arranged in four levels: a = clk + 1.0 | b = a - sps speculatively, beside the compare a >= thr | c = select(b, a) |
// clk = c * m, where m = {lock, 1.0} is picked from the crossing flag off the chain (c * 1.0 == c exactly).  Every
// arithmetic operation is the reference's own: clk + 1.0, clk >= thr, clk - sps, clk * lock_rate.
__device__ __forceinline__ uint32_t step32(double &clk, uint32_t zc, double thr, double neg_sps, double lock)
{
    uint32_t sym = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const double m = (int)zc < 0 ? lock : 1.0;          // crossing at this sample (slicer.py:99-104)
        zc <<= 1;
        const double a = clk + 1.0;                         // slicer.py:77
        const double b = a + neg_sps;                       // slicer.py:81, used only if the symbol is taken
        const bool s = a >= thr;                            // slicer.py:79
        sym = (sym << 1) | (s ? 1u : 0u);
        clk = (s ? b : a) * m;
    }
    return sym;
}

// The same 32 steps with every decision kept in vector registers as a 0 / -1 mask (no compare-to-scalar round trip, fewer
// instructions: a lone wave already takes most of its SIMD's issue slots, and these waves share SIMDs with FIR waves):
//     nm   = sign bits of (a - thr) smeared: -1 where the symbol is NOT taken.  a >= thr  <=>  a - thr >= +0: the difference of
//            two finite doubles is +0, never -0, when they are equal
//     c    = a + (-sps & ~nm): a + (+0) is a itself (a is never -0: it is a sum with 1.0), a + (-sps) is slicer.py:81
//     clk  = fma(c, lm1 & cm, c) with lm1 = lock_rate - 1 and cm = -1 on a crossing: c * (lock - 1) + c is c * lock in exact
//            arithmetic when lock - 1 is exact (the host checks; Sterbenz for 0.5 <= lock <= 2), so the single rounding of the fma
//            is the rounding of the reference's product (slicer.py:99-104); with a zero multiplier it returns c
// The symbol flags are gathered as acc = 2 acc + nm; the word is acc - 1 - ... see the caller (sum of (1 + nm_k) 2^(31-k)).
// LM0 / NS0: the low words of lock_rate - 1 / of sps are zero (0.75, 0.875, ...; every sps that is a small integer): one mask
// operation less each.
template <bool LM0, bool NS0>
__device__ __forceinline__ uint32_t step32m(double &clk, uint32_t zc, double thr, double neg_sps, double lm1)
{
    const int32_t ns_hi = __double2hiint(neg_sps), ns_lo = __double2loint(neg_sps);
    const int32_t lm_hi = __double2hiint(lm1), lm_lo = __double2loint(lm1);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int32_t cm = (int32_t)(zc << k) >> 31;                     // crossing at this sample: -1
        const double sel = __hiloint2double(lm_hi & cm, LM0 ? 0 : lm_lo & cm);
        const double a = clk + 1.0;                                      // slicer.py:77
        const int32_t nm = __double2hiint(a - thr) >> 31;                // slicer.py:79, negated
        acc = (acc << 1) + (uint32_t)nm;
        const double c = a + __hiloint2double(ns_hi & ~nm, NS0 ? 0 : ns_lo & ~nm);
        clk = __builtin_fma(c, sel, c);
    }
    return acc - 1u;        // sum_k (1 + nm_k) 2^(31-k) = (2^32 - 1) + acc  (mod 2^32)
}

// The same 32 steps with the symbol decision taken from the clock itself: fl(clk + 1.0) >= thr  <=>  clk >= tp, where tp is the
// smallest double whose successor-by-one reaches thr (x -> fl(x + 1.0) is monotone, so the clocks that take a symbol are exactly
// [tp, inf); the host finds tp by stepping through the neighbours of thr - 1).  That takes the compare off the chain behind the
// addition: compare | a = clk + 1.0 in parallel, then c = a + {0, -sps}, then the crossing's multiplication as one fma (see
// step32m for why fma(c, lock - 1, c) is the reference's rounded product).  Four dependent operations per sample instead of six,
// eight vector instructions instead of eleven: compare, select, two additions, fma, the symbol bit shifted in with the compare's
// carry, and two for the crossing mask.
template <bool LM0, bool NS0>
__device__ __forceinline__ uint32_t step32c(double &clk, uint32_t zc, double tp, double neg_sps, double lm1)
{
    const int32_t ns_hi = __double2hiint(neg_sps), ns_lo = __double2loint(neg_sps);
    const int32_t lm_hi = __double2hiint(lm1), lm_lo = __double2loint(lm1);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int32_t cm = (int32_t)(zc << k) >> 31;                     // crossing at this sample: -1
        const double sel = __hiloint2double(lm_hi & cm, LM0 ? 0 : lm_lo & cm);
        const bool s = clk >= tp;                                        // slicer.py:77-79: (clk + 1.0) >= thr
        const double a = clk + 1.0;                                      // slicer.py:77
        acc = acc + acc + (s ? 1u : 0u);
        const double c = a + __hiloint2double(s ? ns_hi : 0, NS0 ? 0 : (s ? ns_lo : 0));     // slicer.py:81; a + (+0) is a
        clk = __builtin_fma(c, sel, c);                                  // slicer.py:99-104
    }
    return acc;
}

// step32c as hand-scheduled assembly (tools/gen_slicer_step.py -> pm_slicer_step.inc): the compiler's version of the same
// source spends 13 instructions per sample (the crossing mask as and + compare + select, a move to rebuild the {0, -sps} pair,
// the symbol bit as select + shift + or); scheduled by hand it is 8 (10 when -sps and lock_rate - 1 have low words) with the
// next sample's crossing mask prepared in the shadow of the chain compare -> select -> add -> fma.
template <bool LM0, bool NS0>
__device__ __forceinline__ uint32_t step32a(double &clk, uint32_t zc, double tp, double neg_sps, double lm1)
{
    const int32_t ns_hi = __double2hiint(neg_sps), ns_lo = __double2loint(neg_sps);
    const int32_t lm_hi = __double2hiint(lm1), lm_lo = __double2loint(lm1);
    uint32_t acc = 0;
#define PM_STEP_ASM(BODY)                                                                                                    \
    asm volatile(BODY : "+v"(clk), "+v"(acc) : "v"(zc), "v"(tp), "v"(ns_hi), "v"(ns_lo), "v"(lm_hi), "v"(lm_lo)              \
                 : "vcc", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81")
    if (LM0 && NS0) PM_STEP_ASM(PM_SLICER_STEP32_LM1_NS1);
    else if (NS0) PM_STEP_ASM(PM_SLICER_STEP32_LM0_NS1);
    else if (LM0) PM_STEP_ASM(PM_SLICER_STEP32_LM1_NS0);
    else PM_STEP_ASM(PM_SLICER_STEP32_LM0_NS0);
#undef PM_STEP_ASM
    return acc;
}

// One lockstep launch of the walkers (see the header): every live walker advances by at most `qwords` words.  Work is a
// dense list of walker (= chunk) ids; the first launch runs every walker through its own chunk (list_in == nullptr: identity)
// and needs no comparison -- nobody has been there before.  Beyond its own chunk a walker compares, word by word, the clock it
// enters the word with against the checkpoint the trail in front of it left: equal means merged, it retires; otherwise it
// overwrites checkpoint and symbols and walks on.  Lockstep (qwords <= chunk length, the same for every walker of a launch)
// means the words a walker touches were last written by an EARLIER launch and are touched by nobody else in this one.
// A walker that reaches the end of its stream leaves the clock there as the stream's end state; walkers from further back
// arrive in later launches and overwrite it, or merge before the end, in which case what stands there is already theirs.
// STEP: 0 = step32 (compare and selects), 1 = step32m, 2 = step32m with zero low words in lock_rate - 1 and sps.  One kernel per
// form: with the 64 unrolled steps of several forms in one kernel the loop no longer fits the instruction cache comfortably.
template <int STEP>
__global__ __launch_bounds__(kBlock) void slice_walk_kernel(const JobDev *__restrict__ jobs, int njobs, int lc_words, int qwords,
                                                        uint64_t *__restrict__ wclk, int32_t *__restrict__ wpos,
                                                        const int32_t *__restrict__ list_in, int32_t *__restrict__ list_out,
                                                        int *__restrict__ counts, int iter, uint64_t *__restrict__ symmap,
                                                        uint64_t *__restrict__ ckmap, uint64_t *__restrict__ endstate, int prio)
{
    // These waves are bound by their own dependent chain; when FIR waves of another stream share the SIMD (pipelined executor)
    // every issue slot they lose lengthens the chain, while the FIR waves only need the slots in between: take issue priority.
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counts[iter]) return;                   // whole waves beyond the list leave at once
    const int64_t gc = list_in ? list_in[i] : i;
    const int j = find_job(jobs, njobs, gc);
    // Everything the loop needs from the job table in registers BEFORE the loop: the walker stores through ck / sm, so a field read
    // as jobs[j].x inside the loop is re-loaded in every word, and each such load (the pointer, then the word behind it, then n)
    // is a round trip on the walker's critical path -- a fifth of a word's time alone, and several times that while the FIR
    // kernels of the demod stream keep L1 and L2 busy.
    // (The bitmap pointers come out of a table, so the compiler would take them for generic pointers: `flat` loads, which may
    // return out of order and force a full `s_waitcnt vmcnt(0) lgkmcnt(0)` -- the previous word's stores included -- in front of
    // every use.  They are device memory: say so, and the wait in front of the next word counts past the two stores.)
    typedef const uint64_t __attribute__((address_space(1))) *gptr_c;
    typedef uint64_t __attribute__((address_space(1))) *gptr;
    const JobDev *Jp = jobs + j;
    const gptr_c bi = (gptr_c)Jp->bi;
    const gptr_c bq = (gptr_c)Jp->bq;
    const int64_t n = Jp->n, nwords = Jp->nwords, chunk0 = Jp->chunk0, word0 = Jp->word0;
    const bool quad = Jp->quad != 0;
    const double thr = Jp->thr, sps = Jp->sps, lock = Jp->lock, tp = Jp->tp;
    const int li0 = Jp->li0, lq0 = Jp->lq0;
    const int64_t c = gc - chunk0;
    const int64_t own_end = min((c + 1) * (int64_t)lc_words, nwords);
    int64_t w = wpos[gc];
    double clk = bitsd(wclk[gc]);
    const int64_t w_stop = min(w + (int64_t)qwords, nwords);
    // last_sample starts at 0.0, i.e. ">= 0" (slicer.py:55,164-165)
    uint64_t li = w == 0 ? (uint64_t)li0 : (bi[w - 1] >> 63);
    uint64_t lq = 1ull;
    if (quad) lq = w == 0 ? (uint64_t)lq0 : (bq[w - 1] >> 63);
    const double neg_sps = -sps;
    const double lm1 = lock - 1.0;
    const gptr sm = (gptr)(symmap + word0);
    const gptr ck = (gptr)(ckmap + word0);
    bool alive = true;
    // The next word's sign bits (and, beyond the own chunk, the checkpoint standing there) are loaded one word ahead: the 64 steps
    // of a word are ~2 us of dependent arithmetic, and a load issued at the top of the word it is needed in puts its whole latency
    // on top of every word.  Words [w, w_stop) are this walker's alone in this launch (lockstep), so what is read early is what
    // would have been read late.
    // Unconditional and branch-free (a binary slicer reads the in-phase word twice, the own chunk reads checkpoints it does not
    // compare, the last word of a launch reads itself again): with the loads under conditions the compiler cannot count them and
    // waits for everything outstanding, the previous word's stores included.
    const gptr_c bqe = quad ? bq : bi;
    uint64_t si_n = 0, sq_n = 0, ck_n = 0;
    if (w < w_stop) {
        si_n = bi[w];
        sq_n = bqe[w];
        ck_n = __builtin_nontemporal_load(&ck[w]);
    }
    // ... and the symbol word of a word is stored at the top of the NEXT one, together with that word's loads and checkpoint
    // store: every memory operation of a word then has the word's 64 steps to complete in, and the wait in front of the next
    // word's first use of a loaded value finds nothing young outstanding (stored at the end of its own word, the symbol store's
    // acknowledgement was waited for there, once per word).
    uint64_t sym_late = 0;
    bool late = false;
    for (; w < w_stop; ++w) {
        const uint64_t si = si_n, sq = sq_n, ck_here = ck_n;
        uint64_t cb = dbits(clk);
        if (w >= own_end && ck_here == cb) {         // merged into the trail in front: retire
            alive = false;
            break;
        }
        uint64_t zc = si ^ ((si << 1) | li);
        li = si >> 63;
        if (quad) {
            zc |= sq ^ ((sq << 1) | lq);
            lq = sq >> 63;
        }
        // every loaded value has been used above (one wait, for operations issued a whole word ago); now this word's traffic
        asm volatile("" : "+v"(zc), "+v"(cb) : : "memory");
        if (late) sm[w - 1] = sym_late;
        const int64_t wn = min(w + 1, w_stop - 1);
        si_n = bi[wn];
        sq_n = bqe[wn];
        ck_n = __builtin_nontemporal_load(&ck[wn]);
        ck[w] = cb;
        asm volatile("" : : : "memory");
        const int64_t left = n - (w << 6);
        uint64_t sym;
        if (left >= 64) {
            uint32_t lo, hi;
            if (STEP >= 5) {                       // 5 + LM0 + 2 NS0: the hand-scheduled form of step32c
                lo = step32a<((STEP - 5) & 1) != 0, ((STEP - 5) & 2) != 0>(clk, __brev((uint32_t)zc), tp, neg_sps, lm1);
                hi = step32a<((STEP - 5) & 1) != 0, ((STEP - 5) & 2) != 0>(clk, __brev((uint32_t)(zc >> 32)), tp, neg_sps, lm1);
            } else if (STEP == 4) {
                lo = step32c<true, true>(clk, __brev((uint32_t)zc), tp, neg_sps, lm1);
                hi = step32c<true, true>(clk, __brev((uint32_t)(zc >> 32)), tp, neg_sps, lm1);
            } else if (STEP == 3) {
                lo = step32c<false, false>(clk, __brev((uint32_t)zc), tp, neg_sps, lm1);
                hi = step32c<false, false>(clk, __brev((uint32_t)(zc >> 32)), tp, neg_sps, lm1);
            } else if (STEP == 2) {
                lo = step32m<true, true>(clk, __brev((uint32_t)zc), thr, neg_sps, lm1);
                hi = step32m<true, true>(clk, __brev((uint32_t)(zc >> 32)), thr, neg_sps, lm1);
            } else if (STEP == 1) {
                lo = step32m<false, false>(clk, __brev((uint32_t)zc), thr, neg_sps, lm1);
                hi = step32m<false, false>(clk, __brev((uint32_t)(zc >> 32)), thr, neg_sps, lm1);
            } else {
                lo = step32(clk, __brev((uint32_t)zc), thr, neg_sps, lock);
                hi = step32(clk, __brev((uint32_t)(zc >> 32)), thr, neg_sps, lock);
            }
            sym = ((uint64_t)__brev(hi) << 32) | (uint64_t)__brev(lo);
        } else {                                     // the stream's last, partial word
            sym = 0;
            for (int b = 0; b < (int)left; ++b) {
                clk += 1.0;
                if (clk >= thr) {
                    clk -= sps;
                    sym |= 1ull << b;
                }
                if ((zc >> b) & 1) clk = clk * lock;
            }
        }
        sym_late = sym;
        late = true;
    }
    if (late) sm[w - 1] = sym_late;                  // the last word walked (a walker that retired had not stored it yet either)
    if (alive && w >= nwords) {                      // the end of the stream: its end state (later arrivals are the truer ones)
        endstate[j] = dbits(clk);
        alive = false;
    }
    if (alive) {
        wpos[gc] = (int32_t)w;
        wclk[gc] = dbits(clk);
    }
    // survivors go on the next launch's list: one atomic per wave
    const uint64_t mask = __ballot(alive);
    if (mask) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)mask) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&counts[iter + 1], __popcll(mask));
        base = __shfl(base, leader);
        if (alive) list_out[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)gc;
    }
}

// Symbols per chunk and the (i<<1|q) bits of its last symbol (0xFF if it took none).
__global__ __launch_bounds__(kBlock) void slice_count_kernel(const JobDev *__restrict__ jobs, int njobs, int lc_words, int64_t total_chunks,
                                                         const uint64_t *__restrict__ symmap, uint32_t *__restrict__ count,
                                                         uint8_t *__restrict__ lastsym)
{
    __builtin_amdgcn_s_setprio(3);          // short kernels on the slicer stream's critical path (see slice_walk_kernel)
    const int64_t gc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the pack kernel ORs bit fields into the output words of every stream: clear them here, one launch earlier (instead of one
    // hipMemsetAsync per stream between scan and pack)
    for (int k = 0; k < njobs; ++k) {
        uint32_t *d32 = jobs[k].data32;
        const int64_t nd = (jobs[k].cap + 3) >> 2;
        for (int64_t d = gc; d < nd; d += (int64_t)gridDim.x * blockDim.x) d32[d] = 0u;
        // ... and the stream's trailing partial byte, which the pack kernel ORs together as well: an emission that ran before the
        // last walker had merged (the launch hint was too short) has left the bits of a not yet final trajectory in it
        if (gc == 0) *jobs[k].tail = 0u;
    }
    if (gc >= total_chunks) return;
    const int j = find_job(jobs, njobs, gc);
    const JobDev &J = jobs[j];
    const int64_t c = gc - J.chunk0;
    if (c >= J.nchunks) return;                   // the emit chunk ranges are padded to whole workgroups (slice_pack_kernel)
    const int64_t w0 = c * lc_words, w1 = min(w0 + (int64_t)lc_words, J.nwords);
    const uint64_t *sm = symmap + J.word0;
    uint32_t cnt = 0;
    uint32_t ls = 0xFF;
    for (int64_t w = w0; w < w1; ++w) {
        const uint64_t s = sm[w];
        cnt += (uint32_t)__popcll(s);
        if (J.quad && s) {
            const int b = 63 - __clzll((long long)s);
            ls = (uint32_t)((((J.bi[w] >> b) & 1) << 1) | ((J.bq[w] >> b) & 1));
        }
    }
    count[gc] = cnt;
    lastsym[gc] = (uint8_t)ls;
}

// Per stream: exclusive scan of symbol counts and the "last symbol before this chunk" carry.  One workgroup per stream.
__global__ __launch_bounds__(1024) void slice_scan_kernel(const JobDev *__restrict__ jobs, const uint32_t *__restrict__ count,
                                                          const uint8_t *__restrict__ lastsym, uint64_t *__restrict__ offset,
                                                          uint8_t *__restrict__ prevsym, uint64_t *__restrict__ totals,
                                                          const uint64_t *__restrict__ s_end, int njobs)
{
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint64_t sums[1024];
    __shared__ int lasts[1024];
    const JobDev &J = jobs[blockIdx.x];
    const uint32_t *cnt = count + J.chunk0;
    const uint8_t *lsym = lastsym + J.chunk0;
    uint64_t *off = offset + J.chunk0 + blockIdx.x;        // nchunks+1 entries per stream
    uint8_t *psym = prevsym + J.chunk0;
    const int t = threadIdx.x;
    const int64_t per = (J.nchunks + 1023) / 1024;
    const int64_t c0 = min((int64_t)t * per, J.nchunks), c1 = min(c0 + per, J.nchunks);
    uint64_t s = 0;
    int l = -1;
#pragma unroll 8
    for (int64_t c = c0; c < c1; ++c) {                    // unrolled: eight independent loads in flight instead of one
        s += cnt[c];
        if (lsym[c] != 0xFF) l = lsym[c];
    }
    sums[t] = s;
    lasts[t] = l;
    __syncthreads();
    // inclusive scan across the 1024 threads: sums add, "last symbol seen" takes the nearest defined one to the left
    for (int d = 1; d < 1024; d <<= 1) {
        uint64_t v = 0;
        int lv = -1;
        if (t >= d) {
            v = sums[t - d];
            lv = lasts[t - d];
        }
        __syncthreads();
        if (t >= d) {
            sums[t] += v;
            if (lasts[t] < 0) lasts[t] = lv;
        }
        __syncthreads();
    }
    const int carry0 = J.sreg0 & 3;                        // state_register starts at 0 (slicer.py:202) or where the last call left it
    const uint64_t before = t > 0 ? sums[t - 1] : 0;
    const int carry_before = t > 0 && lasts[t - 1] >= 0 ? lasts[t - 1] : carry0;
    if (t == 0) {
        const uint64_t all = sums[1023];
        const int carry = lasts[1023] >= 0 ? lasts[1023] : carry0;
        off[J.nchunks] = all;
        totals[blockIdx.x] = all;
        // end state for the next call on this slicer object: clock after the last chunk, signs of the last sample, last symbol
        totals[njobs + blockIdx.x] = s_end[blockIdx.x];   // left by the last walker to reach the end of the stream
        const int64_t last = J.n - 1;
        uint64_t signs = (J.bi[last >> 6] >> (last & 63)) & 1;
        if (J.quad) signs |= ((J.bq[last >> 6] >> (last & 63)) & 1) << 1;
        totals[2 * njobs + blockIdx.x] = signs;
        totals[3 * njobs + blockIdx.x] = (uint64_t)carry;
    }
    uint64_t run = before;
    int carry = carry_before;
#pragma unroll 8
    for (int64_t c = c0; c < c1; ++c) {
        off[c] = run;
        psym[c] = (uint8_t)carry;
        run += cnt[c];
        if (lsym[c] != 0xFF) carry = lsym[c];
    }
}

// Symbol bitmap + sign bitmap(s) -> packed bytes (MSB first) and the address of each byte's last symbol.
//
// A lane turns its chunk's few dozen symbols into two or three bytes, so lane by lane the output would be scattered 8-byte address
// stores and read-modify-writes of single bytes (measured: 830 MB of HBM traffic for 52 MB of output).  The chunks of a workgroup
// are consecutive chunks of ONE stream (the emit chunk ranges are padded to whole workgroups), so their output is one contiguous
// byte range: it is assembled in LDS and written out by the whole workgroup, addresses as coalesced 8-byte stores, data as whole
// dwords; only the first and the last dword of the range can hold bits of a neighbouring workgroup and go out as atomicOr.
constexpr int kPackBytes = 2048;
constexpr int kEmitWordsMax = 4;       // words per emit chunk at most (slice_pack_kernel stages 256 chunks of them in LDS).  Round 5: 8 -> 4 and
                                       // 32-bit staged addresses took the kernel from 51 KB of LDS to 26: the demod kernel beside it fills a CU's
                                       // 160 KB with four 40 KB workgroups, and a 51 KB workgroup -- at any priority -- waited until two of them
                                       // retired at once (1.5 ms per batch in the pipeline for a kernel that takes 0.2 alone)
__global__ __launch_bounds__(kBlock) void slice_pack_kernel(const JobDev *__restrict__ jobs, int njobs, int lc_words, int64_t total_chunks,
                                                        const uint64_t *__restrict__ symmap, const uint64_t *__restrict__ offset,
                                                        const uint8_t *__restrict__ prevsym)
{
    __shared__ uint32_t lb[kPackBytes / 4 + 1];
    __shared__ uint32_t la[kPackBytes + 4];                 // a byte's address as 1 + its distance in samples from the workgroup's first word (0: none)
    // the workgroup's words of the symbol bitmap and of the (in-phase) sign bitmap, loaded once, coalesced: lane by lane the reads
    // are 32-64 bytes apart and every cache line is fetched several times (PMC round 1: 6.5 x the algorithmic read)
    __shared__ uint64_t s_sym[kBlock * kEmitWordsMax], s_bi[kBlock * kEmitWordsMax];
    __builtin_amdgcn_s_setprio(3);          // short kernels on the slicer stream's critical path (see slice_walk_kernel)
    const int64_t gc0 = (int64_t)blockIdx.x * kBlock;
    if (gc0 >= total_chunks) return;
    const int j = find_job(jobs, njobs, gc0);
    const JobDev &J = jobs[j];
    const int64_t c0 = gc0 - J.chunk0;
    if (c0 >= J.nchunks) return;                           // padding only (uniform over the workgroup)
    const int t = threadIdx.x;
    const int64_t c = c0 + t;
    const int64_t gc = gc0 + t;
    const bool live = c < J.nchunks;
    const uint64_t *off = offset + J.chunk0 + j;
    const uint64_t total = off[J.nchunks];
    const int bps = J.bps;
    const uint64_t nb0 = (uint64_t)J.nb0;                  // bits the previous call left in the working byte come first
    const uint64_t nbytes = (nb0 + total * (uint64_t)bps) >> 3;   // a trailing partial byte is not emitted (slicer.py:94-96): it is the end state
    const uint64_t cap = (uint64_t)J.cap;
    // the workgroup's output bytes [byte_a, byte_b), staged from the dword boundary `base` at or below byte_a
    const uint64_t g_a = off[c0], g_b = off[min(c0 + (int64_t)kBlock, J.nchunks)];
    const uint64_t byte_a = c0 == 0 ? 0 : (nb0 + g_a * (uint64_t)bps) >> 3;
    const uint64_t byte_b = (nb0 + g_b * (uint64_t)bps + 7) >> 3;
    const uint64_t base = byte_a & ~3ull;
    const bool staged = byte_b - base <= (uint64_t)kPackBytes;
    const int span = staged ? (int)(byte_b - base) : 0;
    if (staged) {
        for (int i = t; i < (span + 3) / 4; i += kBlock) lb[i] = 0;
        for (int i = t; i < span; i += kBlock) la[i] = 0;
        __syncthreads();
    }
    // one byte's bits (and, from the lane that completes the byte, its address: never 0, addresses are 1-based)
    const int64_t wg0 = c0 * lc_words, wg1 = min(wg0 + (int64_t)kBlock * lc_words, J.nwords);
    const long long addr_base = (long long)(J.addr0 + (wg0 << 6));      // address = addr_base + (the 32-bit value staged)
    auto put = [&](uint64_t idx, uint32_t bits8, long long address) {
        if (staged) {
            const int rel = (int)(idx - base);
            atomicOr(&lb[rel >> 2], bits8 << ((rel & 3) * 8));
            if (address) la[rel] = (uint32_t)(address - addr_base);      // in 1 .. 64 * 256 * kEmitWordsMax
        } else if (idx < cap) {
            atomicOr(&J.data32[idx >> 2], bits8 << ((idx & 3) * 8));
            if (address) J.addr[idx] = address;
        }
    };
    {
        const uint64_t *smg = symmap + J.word0;
        for (int64_t i = t; i < wg1 - wg0; i += kBlock) {
            s_sym[i] = smg[wg0 + i];
            s_bi[i] = J.bi[wg0 + i];
        }
        __syncthreads();
    }
    if (live) {
        uint64_t g = off[c];
        if (c == 0 && nb0) {
            const uint32_t head = ((uint32_t)J.wb0 & ((1u << nb0) - 1u)) << (8 - nb0);
            if (nbytes > 0) put(0, head, 0);
            else atomicOr(J.tail, head);
        }
        uint32_t prev = prevsym[gc];
        const int64_t w0 = c * lc_words, w1 = min(w0 + (int64_t)lc_words, J.nwords);
        uint32_t acc = 0;
        bool pending = false;
        for (int64_t w = w0; w < w1; ++w) {
            uint64_t s = s_sym[w - wg0];
            if (!s) continue;
            const uint64_t si = s_bi[w - wg0];
            const uint64_t sq = J.quad ? J.bq[w] : 0;
            while (s) {
                const int b = __ffsll((long long)s) - 1;
                s &= s - 1;
                uint32_t v;
                if (J.quad) {
                    const uint32_t cur = (uint32_t)((((si >> b) & 1) << 1) | ((sq >> b) & 1));
                    v = (uint32_t)J.demap[((prev << 2) | cur) & (uint32_t)J.mask];        // slicer.py:210-217
                    prev = cur;
                } else {
                    v = (uint32_t)((si >> b) & 1);                                         // slicer.py:85-90
                }
                const uint64_t bitpos = nb0 + g * (uint64_t)bps;
                const int inbyte = (int)(bitpos & 7);
                acc |= v << (8 - bps - inbyte);
                pending = true;
                if (inbyte + bps == 8) {
                    // streamaddress, 1-based, continuing the previous call's count
                    put(bitpos >> 3, acc & 0xFF, (long long)(J.addr0 + (w << 6) + b + 1));
                    acc = 0;
                    pending = false;
                }
                ++g;
            }
        }
        if (pending) {                                     // head of a byte that a later chunk (or a later call) completes
            const uint64_t idx = (nb0 + (g - 1) * (uint64_t)bps) >> 3;
            if (idx < nbytes) put(idx, acc & 0xFF, 0);
            else atomicOr(J.tail, acc & 0xFF);
        }
    }
    if (!staged) return;
    __syncthreads();
    for (int i = t; i < span; i += kBlock) {
        const uint32_t a = la[i];
        if (a && base + i < cap) J.addr[base + i] = addr_base + (long long)a;
    }
    const int nd = (span + 3) / 4;
    for (int d = t; d < nd; d += kBlock) {
        const uint64_t b0 = base + 4ull * d;               // first byte of this dword
        if (b0 >= cap) continue;
        uint32_t v = lb[d];
        if (cap - b0 < 4) v &= (1u << ((cap - b0) * 8)) - 1u;
        if (d == 0 || d == nd - 1) {
            if (v) atomicOr(&J.data32[b0 >> 2], v);
        } else {
            J.data32[b0 >> 2] = v;
        }
    }
}

__global__ void slice_init_kernel(const JobDev *__restrict__ jobs, int njobs, int lc_words, int64_t total_chunks, uint64_t *wclk,
                                  int32_t *wpos, int *counts, int ncounts, uint32_t *tails)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t k = t; k < ncounts; k += stride) counts[k] = k == 0 ? (int)total_chunks : 0;       // the first launch runs every walker
    for (int64_t k = t; k < njobs; k += stride) tails[k] = 0u;
    for (int64_t gc = t; gc < total_chunks; gc += stride) {
        const int j = find_job(jobs, njobs, gc);
        const int64_t c = gc - jobs[j].chunk0;
        // cold start everywhere (phase_clock = 0.0); walker 0 of a stream starts from its true state, carried in or zero (slicer.py:50)
        wclk[gc] = c == 0 ? dbits(jobs[j].clk0) : 0ull;
        wpos[gc] = (int32_t)(c * lc_words);
    }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The smallest double x with fl(x + 1.0) >= thr (NaN if it cannot be pinned down: the caller then keeps the compare on the sum).
double symbol_clock_threshold(double thr)
{
    if (!(thr - thr == 0.0)) return NAN;
    volatile double x = thr - 1.0;
    for (int k = 0; k < 64; ++k) {                       // down while the predecessor still reaches thr
        volatile double p = nextafter((double)x, -INFINITY);
        volatile double sum = p + 1.0;
        if (!(sum >= thr)) break;
        x = p;
        if (k == 63) return NAN;
    }
    for (int k = 0; k < 64; ++k) {                       // up until it does
        volatile double sum = x + 1.0;
        if (sum >= thr) return x;
        x = nextafter((double)x, INFINITY);
    }
    return NAN;
}


// ---- one lane per stream, a chunk of the stream per launch (the carrier-loop batch engine, pm_loopbatch.hip) -------------------------
// The engine advances thousands of streams together, a time chunk at a time, and its loops are sequential in exactly the way the
// slicer's clock is: here the parallelism is across the streams, so every stream gets ONE lane that carries the true state from chunk
// to chunk -- no walkers, no merging, no symbol bitmap, and no sign bitmaps of whole recordings either (59 GB for 16 384 streams of ten
// minutes: the engine keeps one chunk of them).  Per 64-sample word the lane runs the same 64 steps as a walker (the step forms above:
// the reference's operations in the reference's order), then turns the word's symbols into bits as slice_pack_kernel does
// (slicer.py:85-96, :210-228), and stores a byte and its address step whenever eight bits are full.  A wave of 64 such lanes is bound
// by its dependent chain, ~2 us per word: a 65 536-sample chunk takes ~2.5 ms beside the 18-20 ms of the chunk's carrier loops.
struct RowParams {
    double thr, sps, lock, tp;
    int bps, mask;
    unsigned long long demap4;       // demap[16], four bits each
};

template <int STEP>
__global__ __launch_bounds__(64) void rowslice_kernel(const RowParams *__restrict__ params, int chains, int rows, int quad,
                                                      const uint64_t *__restrict__ bits_i, const uint64_t *__restrict__ bits_q, int64_t stride,
                                                      int64_t first, int64_t count, pm_rowslice_rec *__restrict__ recs,
                                                      uint8_t *__restrict__ data, uint16_t *__restrict__ steps, int64_t cap)
{
    const int row = (int)(blockIdx.x * 64 + threadIdx.x);
    if (row >= rows) return;
    typedef const uint64_t __attribute__((address_space(1))) *gptr_c;
    const RowParams P = params[row % chains];
    pm_rowslice_rec R = recs[row];
    const gptr_c bi = (gptr_c)(bits_i + (int64_t)row * stride);
    const gptr_c bq = quad ? (gptr_c)(bits_q + (int64_t)row * stride) : bi;
    const double thr = P.thr, sps = P.sps, lock = P.lock, tp = P.tp, neg_sps = -P.sps, lm1 = P.lock - 1.0;
    const int bps = P.bps;
    const uint32_t mask = (uint32_t)P.mask;
    const unsigned long long demap4 = P.demap4;
    double clk = R.clk;
    uint64_t li = R.li_neg ? 0ull : 1ull, lq = R.lq_neg ? 0ull : 1ull;
    uint32_t byte = (uint32_t)R.wbyte, prev = (uint32_t)R.sreg & 3u, flags = (uint32_t)R.flags;
    int nbits = R.wbits;
    int64_t cnt = R.count, last_addr = R.last_addr, first_addr = R.first_addr;
    uint8_t *const out_d = data + (int64_t)row * cap;
    uint16_t *const out_s = steps + (int64_t)row * cap;
    const int64_t nwords = (count + 63) >> 6;
    uint64_t si_n = 0, sq_n = 0;
    if (nwords > 0) {
        si_n = bi[0];
        sq_n = bq[0];
    }
    for (int64_t w = 0; w < nwords; ++w) {
        const uint64_t si = si_n, sq = sq_n;
        const int64_t wn = min(w + 1, nwords - 1);
        si_n = bi[wn];                                       // a word ahead: its latency under this word's 64 steps
        sq_n = bq[wn];
        uint64_t zc = si ^ ((si << 1) | li);
        if (quad) zc |= sq ^ ((sq << 1) | lq);
        const int64_t left = count - (w << 6);
        uint64_t sym;
        if (left >= 64) {
            li = si >> 63;
            lq = sq >> 63;
            uint32_t lo, hi;
            if (STEP >= 5) {
                lo = step32a<((STEP - 5) & 1) != 0, ((STEP - 5) & 2) != 0>(clk, __brev((uint32_t)zc), tp, neg_sps, lm1);
                hi = step32a<((STEP - 5) & 1) != 0, ((STEP - 5) & 2) != 0>(clk, __brev((uint32_t)(zc >> 32)), tp, neg_sps, lm1);
            } else if (STEP == 4) {
                lo = step32c<true, true>(clk, __brev((uint32_t)zc), tp, neg_sps, lm1);
                hi = step32c<true, true>(clk, __brev((uint32_t)(zc >> 32)), tp, neg_sps, lm1);
            } else if (STEP == 3) {
                lo = step32c<false, false>(clk, __brev((uint32_t)zc), tp, neg_sps, lm1);
                hi = step32c<false, false>(clk, __brev((uint32_t)(zc >> 32)), tp, neg_sps, lm1);
            } else if (STEP == 2) {
                lo = step32m<true, true>(clk, __brev((uint32_t)zc), thr, neg_sps, lm1);
                hi = step32m<true, true>(clk, __brev((uint32_t)(zc >> 32)), thr, neg_sps, lm1);
            } else if (STEP == 1) {
                lo = step32m<false, false>(clk, __brev((uint32_t)zc), thr, neg_sps, lm1);
                hi = step32m<false, false>(clk, __brev((uint32_t)(zc >> 32)), thr, neg_sps, lm1);
            } else {
                lo = step32(clk, __brev((uint32_t)zc), thr, neg_sps, lock);
                hi = step32(clk, __brev((uint32_t)(zc >> 32)), thr, neg_sps, lock);
            }
            sym = ((uint64_t)__brev(hi) << 32) | (uint64_t)__brev(lo);
        } else {                                             // the stream's last, partial word (bits past the end are not samples)
            li = (si >> (left - 1)) & 1;
            lq = (sq >> (left - 1)) & 1;
            sym = 0;
            for (int b = 0; b < (int)left; ++b) {
                clk += 1.0;
                if (clk >= thr) {
                    clk -= sps;
                    sym |= 1ull << b;
                }
                if ((zc >> b) & 1) clk = clk * lock;
            }
        }
        while (sym) {
            const int b = __ffsll((long long)sym) - 1;
            sym &= sym - 1;
            uint32_t v;
            if (quad) {
                const uint32_t cur = (uint32_t)((((si >> b) & 1) << 1) | ((sq >> b) & 1));
                v = (uint32_t)(demap4 >> (4 * (((prev << 2) | cur) & mask))) & 15u;          // slicer.py:210-217
                prev = cur;
            } else {
                v = (uint32_t)((si >> b) & 1);                                              // slicer.py:85-90
            }
            byte = ((byte << bps) | v) & 0xFFu;
            nbits += bps;
            if (nbits >= 8) {                                                               // slicer.py:91-96, :219-228
                nbits = 0;
                const int64_t a = first + (w << 6) + b + 1;                                 // 1-based address of the byte's last symbol
                if (cnt < cap) {
                    const int64_t step = cnt ? a - last_addr : 0;
                    if (step > 65535) flags |= 1u;
                    out_d[cnt] = (uint8_t)byte;
                    out_s[cnt] = (uint16_t)step;
                } else {
                    flags |= 2u;
                }
                if (cnt == 0) first_addr = a;
                last_addr = a;
                ++cnt;
            }
        }
    }
    R.clk = clk;
    R.li_neg = li ? 0 : 1;
    R.lq_neg = quad ? (lq ? 0 : 1) : 0;
    R.wbits = nbits;
    R.wbyte = nbits ? (int32_t)(byte & ((1u << nbits) - 1u)) : 0;
    R.sreg = (int32_t)prev;
    R.flags = (int32_t)flags;
    R.count = cnt;
    R.first_addr = first_addr;
    R.last_addr = last_addr;
    R.seen = first + count;
    recs[row] = R;
}

// rows of a sliced run -> one dense block: row k (of this call's rows) at the sum of the rows' sizes before it, count[k] address steps
// (uint16, padded to 8 bytes) then count[k] data bytes (padded to 8)
__global__ __launch_bounds__(256) void rows_gather_kernel(const pm_rowslice_rec *__restrict__ recs, const uint8_t *__restrict__ data,
                                                          const uint16_t *__restrict__ steps, int64_t cap, int64_t row0, uint8_t *__restrict__ block,
                                                          size_t block_bytes)
{
    __shared__ unsigned long long part[256];
    const int k = blockIdx.x, t = threadIdx.x;
    unsigned long long mine = 0;
    for (int j = t; j < k; j += 256) {
        const unsigned long long c = (unsigned long long)min(recs[row0 + j].count, cap);
        mine += ((2 * c + 7) & ~7ull) + ((c + 7) & ~7ull);
    }
    part[t] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) part[t] += part[t + s];
        __syncthreads();
    }
    const unsigned long long off = part[0];
    const unsigned long long c = (unsigned long long)min(recs[row0 + k].count, cap);
    const unsigned long long sw = (2 * c + 7) >> 3, dw = (c + 7) >> 3;
    if (off + 8 * (sw + dw) > block_bytes) return;           // (the host sized the block from the same counts)
    const uint64_t *src_s = reinterpret_cast<const uint64_t *>(steps + (row0 + k) * cap);      // cap is a multiple of 8: rows are aligned
    const uint64_t *src_d = reinterpret_cast<const uint64_t *>(data + (row0 + k) * cap);
    uint64_t *dst = reinterpret_cast<uint64_t *>(block + off);
    for (unsigned long long i = t; i < sw; i += 256) dst[i] = src_s[i];
    for (unsigned long long i = t; i < dw; i += 256) dst[sw + i] = src_d[i];
}

}  // namespace

extern "C" int pm_slice_batch(pm_ctx *ctx, pm_slice_job *jobs, int njobs)
{
    PM_CTX(ctx);
    PM_ARG(jobs && njobs >= 1 && njobs <= kMaxJobs);
    ctx->sl_iterations = 0;
    int64_t max_n = 0;
    for (int j = 0; j < njobs; ++j) {
        pm_slice_job &q = jobs[j];
        q.count = 0;
        PM_ARG(q.n >= 0 && q.cap >= 0);
        PM_ARG(q.params.bits_per_symbol == 1 || q.params.bits_per_symbol == 2);
        PM_ARG(q.params.samples_per_symbol > 0.0 && q.params.lock_rate == q.params.lock_rate);
        PM_ARG(q.n == 0 || (q.d_bits_i && (q.cap == 0 || (q.d_data && q.d_addr))));
        PM_ARG(((uintptr_t)q.d_data & 3) == 0);
        max_n = std::max(max_n, q.n);
    }
    if (max_n == 0) return PM_OK;

    // Chunk length L (a multiple of 64 samples) = distance between walkers.  Lane-steps are N (1 + m/L) with m the merge length
    // (10-20 k samples at lock 0.77), the depth is (L + longest merge) x t_step: long chunks are cheap, short ones are quick.
    // The batch is cut into about sl_target_lanes walkers, between 1024 and 65536 samples each.  PM_SLICER_CHUNK_WORDS overrides.
    int64_t all_words = 0, max_words = 0;
    for (int j = 0; j < njobs; ++j) {
        all_words += pm_cdiv(jobs[j].n, 64);
        max_words = std::max(max_words, pm_cdiv(jobs[j].n, 64));
    }
    // chunk length of a large batch: 384 words (24.6 k samples).  Longer chunks mean fewer walkers and fewer lane-steps per sample
    // (N (1 + m/L)), i.e. less of the vector ALU taken from the demod kernels beside them, shorter ones a shallower first launch;
    // measured in the pipelined executor (medians of interleaved runs, 20 / 400 steps): 256 words 1.52 / 1.17 ms per step, 384 words
    // 1.45 / 1.16, 512 words 1.48 / 1.25
    const pm_tuning &tn = ctx->tune;
    const int64_t lc_max = tn.slicer_max_chunk_words > 0 ? std::max(16, tn.slicer_max_chunk_words) : ctx->sl_max_chunk_words;
    int64_t lc_words = std::max<int64_t>(16, std::min<int64_t>(pm_cdiv(all_words, ctx->sl_target_lanes), lc_max));
    if (tn.slicer_chunk_words > 0) lc_words = tn.slicer_chunk_words;
    // Words per lockstep launch once the walkers are beyond their own chunks (never more than a chunk: see slice_walk_kernel).
    // A walker that retires mid-launch leaves its lane idle for the rest of it, so short launches waste less; each costs a dispatch.
    int64_t qwords = 32;
    if (tn.slicer_quantum_words > 0) qwords = tn.slicer_quantum_words;
    qwords = std::min(qwords, lc_words);
    const int64_t qtail = std::min<int64_t>(lc_words, 4 * qwords);
    std::vector<JobDev> jd;
    jd.reserve(njobs);
    std::vector<int> live;
    int64_t total_chunks = 0, total_words = 0;
    for (int j = 0; j < njobs; ++j) {
        const pm_slice_job &q = jobs[j];
        if (q.n == 0) continue;
        JobDev d;
        memset(&d, 0, sizeof(d));
        d.bi = q.d_bits_i;
        d.bq = q.d_bits_q;
        d.quad = q.d_bits_q != nullptr;
        d.n = q.n;
        d.nwords = pm_cdiv(q.n, 64);
        d.chunk0 = total_chunks;
        d.nchunks = pm_cdiv(d.nwords, lc_words);
        d.word0 = total_words;
        d.sps = q.params.samples_per_symbol;
        d.thr = (q.params.samples_per_symbol / 2.0) - 0.5;          // slicer.py:52
        d.tp = symbol_clock_threshold(d.thr);
        d.lock = q.params.lock_rate;
        d.bps = q.params.bits_per_symbol;
        d.mask = q.params.state_mask;
        for (int i = 0; i < 16; ++i) d.demap[i] = q.params.demap[i];
        d.data32 = (uint32_t *)q.d_data;
        d.addr = q.d_addr;
        d.cap = q.cap;
        d.li0 = d.lq0 = 1;
        if (const pm_slicer_state *st = q.h_state) {
            PM_ARG(st->working_bits >= 0 && st->working_bits < 8 && st->working_bits % q.params.bits_per_symbol == 0 && st->streamaddress >= 0);
            d.clk0 = st->phase_clock;
            d.li0 = st->last_i_negative ? 0 : 1;
            d.lq0 = st->last_q_negative ? 0 : 1;
            d.nb0 = st->working_bits;
            d.wb0 = st->working_byte;
            d.sreg0 = st->state_register;
            d.addr0 = st->streamaddress;
        }
        total_chunks += d.nchunks;
        total_words += d.nwords;
        jd.push_back(d);
        live.push_back(j);
    }
    const int nj = (int)jd.size();
    PM_ARG(total_chunks < (1ll << 31) && max_words < (1ll << 31));
    ctx->sl_chunk_len = (int32_t)(lc_words * 64);
    ctx->sl_chunks = total_chunks;
    // The count / scan / pack kernels only read the symbol bitmap the walkers left: they are cut independently of it, finely
    // (throughput kernels: ~4 waves per SIMD), however long the walkers' chunks are.
    const int64_t le_words = std::max<int64_t>(std::min<int64_t>(4, lc_words), std::min<int64_t>({lc_words, pm_cdiv(total_words, 524288), (int64_t)kEmitWordsMax}));
    std::vector<JobDev> je = jd;
    int64_t emit_chunks = 0;
    for (JobDev &d : je) {
        d.chunk0 = emit_chunks;
        d.nchunks = pm_cdiv(d.nwords, le_words);
        emit_chunks += pm_cdiv(d.nchunks, (int64_t)kBlock) * kBlock;      // a workgroup never straddles two streams (slice_pack_kernel)
    }

    // No walker walks further than its stream is long: that bounds the launches (reached only without any crossing at all).
    const int64_t max_launches = pm_cdiv(max_words, qwords) + 2;
    const int ncounts = (int)std::min<int64_t>(max_launches + 8, 1 << 22);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t ee = (size_t)emit_chunks + nj;
    // one host-to-device block (both job tables) and one device-to-host block (totals | tails | list sizes)
    const size_t res_tot = (size_t)nj * 8 * 4, res_tail = align_up((size_t)nj * 4, 8);
    const size_t o_jobs = carve(sizeof(JobDev) * nj * 2), o_wclk = carve((size_t)total_chunks * 8), o_wpos = carve((size_t)total_chunks * 4),
                 o_la = carve((size_t)total_chunks * 4), o_lb = carve((size_t)total_chunks * 4), o_end = carve((size_t)nj * 8),
                 o_res = carve(res_tot + res_tail + (size_t)ncounts * 4),
                 o_cnt = carve((size_t)emit_chunks * 4), o_ls = carve(emit_chunks), o_off = carve(ee * 8), o_ps = carve(emit_chunks),
                 o_sym = carve((size_t)total_words * 8), o_ck = carve((size_t)total_words * 8);
    if (int rc = pm_scratch_reserve(ctx, off)) return rc;
    char *base = (char *)ctx->d_scratch;
    JobDev *d_jobs = (JobDev *)(base + o_jobs), *d_ejobs = d_jobs + nj;
    uint64_t *wclk = (uint64_t *)(base + o_wclk), *endstate = (uint64_t *)(base + o_end);
    int32_t *wpos = (int32_t *)(base + o_wpos);
    int32_t *list_a = (int32_t *)(base + o_la), *list_b = (int32_t *)(base + o_lb);
    uint64_t *totals = (uint64_t *)(base + o_res);
    uint32_t *tails = (uint32_t *)(base + o_res + res_tot);
    int *counts = (int *)(base + o_res + res_tot + res_tail);
    uint32_t *cnt = (uint32_t *)(base + o_cnt);
    uint8_t *ls = (uint8_t *)(base + o_ls), *ps = (uint8_t *)(base + o_ps);
    uint64_t *offs = (uint64_t *)(base + o_off), *symmap = (uint64_t *)(base + o_sym), *ckmap = (uint64_t *)(base + o_ck);
    for (int k = 0; k < nj; ++k) jd[k].tail = je[k].tail = tails + k;

    std::vector<JobDev> both(jd);
    both.insert(both.end(), je.begin(), je.end());
    PM_HIP(hipMemcpyAsync(d_jobs, both.data(), sizeof(JobDev) * both.size(), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(slice_init_kernel, dim3((unsigned)std::min<int64_t>(pm_cdiv(std::max<int64_t>(total_chunks, ncounts), kBlock), 1024)),
                       dim3(kBlock), 0, ctx->stream, d_jobs, nj, (int)lc_words, total_chunks, wclk, wpos, counts, ncounts, tails);

    // step32m needs lock_rate - 1 to be exact for every stream of the batch (it is for 0.5 <= lock_rate <= 2) and finite clocks
    int masks = tn.slicer_compare_step ? 0 : 2;
    bool lm0 = true, ns0 = true;                     // low words of lock_rate - 1 / of sps zero in every stream
    for (const JobDev &d : jd) {
        const volatile double lm1 = d.lock - 1.0;
        if (!(lm1 + 1.0 == d.lock) || !(d.clk0 - d.clk0 == 0.0) || !(d.sps - d.sps == 0.0)) masks = 0;
        uint64_t lb, sb;
        const double l1 = lm1;
        memcpy(&lb, &l1, 8);
        memcpy(&sb, &d.sps, 8);
        if (masks == 2 && ((uint32_t)lb || (uint32_t)sb)) masks = 1;     // low words not zero: the general mask form
        lm0 = lm0 && (uint32_t)lb == 0;
        ns0 = ns0 && (uint32_t)sb == 0;
    }
    bool direct = masks != 0 && !tn.slicer_mask_step;          // step32c: the decision from the clock itself
    for (const JobDev &d : jd) direct = direct && d.tp == d.tp;
    const bool hand = direct && !tn.slicer_compiled_step;
    auto walk_kernel = hand ? (lm0 ? (ns0 ? slice_walk_kernel<8> : slice_walk_kernel<6>) : (ns0 ? slice_walk_kernel<7> : slice_walk_kernel<5>))
                       : direct ? (masks == 2 ? slice_walk_kernel<4> : slice_walk_kernel<3>)
                                : masks == 2 ? slice_walk_kernel<2> : masks == 1 ? slice_walk_kernel<1> : slice_walk_kernel<0>;
    const unsigned wgrid = (unsigned)pm_cdiv(total_chunks, kBlock);
    const bool trace = tn.slicer_trace != 0;
    const int prio = tn.slicer_no_setprio ? 0 : 1;
    // How many lockstep launches to enqueue before the emit kernels without asking the device: what the last batches of this
    // shape needed plus a margin (a launch that finds its list empty costs a dispatch and nothing else).  The list sizes come
    // back with the results; if walkers were still alive the rest is launched and the emit kernels run again.
    const int64_t shape = lc_words * 1000003 + qwords;
    if (ctx->sl_hint_shape != shape) { ctx->sl_hint_shape = shape; ctx->sl_launch_hint = (int32_t)std::min<int64_t>(kShortLaunches + 12, max_launches); }
    int launches = 0;                                  // walker launches so far; list `launches` is the one the next launch reads
    int burst = std::max(1, (int)std::min<int64_t>(ctx->sl_launch_hint, max_launches));
    char *h_res = (char *)ctx->h_pinned;
    const size_t h_cap_counts = (kPinnedBytes - res_tot - res_tail) / 4;
    uint64_t *h_tot = (uint64_t *)h_res;
    uint32_t *h_tail = (uint32_t *)(h_res + res_tot);
    int *h_counts = (int *)(h_res + res_tot + res_tail);
    for (;;) {
        {
            PmProf prof(ctx, PM_K_SLICE_ITER);
            for (int b = 0; b < burst; ++b) {
                // launch 0: every walker through its own chunk (identity list); then qwords at a time on the compacted lists
                const int32_t *lin = launches == 0 ? nullptr : ((launches & 1) ? list_a : list_b);
                int32_t *lout = (launches & 1) ? list_b : list_a;
                // the first launches after the chunks are short (most walkers retire there: a retired walker's lane idles until
                // the launch ends); the thin tail afterwards goes in longer launches: fewer dispatches, and few waves to waste
                const int64_t qw = launches == 0 ? lc_words : launches <= kShortLaunches ? qwords : qtail;
                hipLaunchKernelGGL(walk_kernel, dim3(wgrid), dim3(kBlock), 0, ctx->stream, d_jobs, nj, (int)lc_words, (int)qw, wclk, wpos,
                                   lin, lout, counts, launches, symmap, ckmap, endstate, prio);
                ++launches;
            }
        }
        {
            PmProf prof(ctx, PM_K_SLICE_EMIT);
            const unsigned egrid = (unsigned)pm_cdiv(emit_chunks, kBlock);
            hipLaunchKernelGGL(slice_count_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, d_ejobs, nj, (int)le_words, emit_chunks, symmap, cnt, ls);
            hipLaunchKernelGGL(slice_scan_kernel, dim3(nj), dim3(1024), 0, ctx->stream, d_ejobs, cnt, ls, offs, ps, totals, endstate, nj);
            hipLaunchKernelGGL(slice_pack_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, d_ejobs, nj, (int)le_words, emit_chunks, symmap, offs, ps);
        }
        const size_t ncopy = std::min<size_t>((size_t)launches + 1, h_cap_counts);
        PM_HIP(hipMemcpyAsync(h_res, totals, res_tot + res_tail + ncopy * 4, hipMemcpyDeviceToHost, ctx->stream));
        int *h_last = h_counts + launches;
        if ((size_t)launches + 1 > ncopy) {               // more launches than the mailbox holds list sizes for: fetch the last one on its own
            h_last = h_counts + ncopy - 1;
            PM_HIP(hipMemcpyAsync(h_last, counts + launches, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        }
        PM_HIP(hipStreamSynchronize(ctx->stream));
        PM_HIP(hipGetLastError());
        const int alive = *h_last;
        if (trace) {
            fprintf(stderr, "[slicer] chunk %lld words, %lld per launch, %lld walkers; alive after each of %d launches:", (long long)lc_words,
                    (long long)qwords, (long long)total_chunks, launches);
            for (size_t k = 1; k < ncopy; ++k) fprintf(stderr, " %d", h_counts[k]);
            fprintf(stderr, "\n");
        }
        if (alive == 0) {
            int used = launches;                        // the first launch that left nobody alive
            for (size_t k = 1; k < ncopy; ++k)
                if (h_counts[k] == 0) { used = (int)k; break; }
            ctx->sl_iterations = used;
            // next time: what this batch needed and a quarter more; decays slowly when batches get easier
            const int want = used + std::max(4, used / 4);
            ctx->sl_launch_hint = want >= ctx->sl_launch_hint ? want : ctx->sl_launch_hint - (ctx->sl_launch_hint - want + 7) / 8;
            break;
        }
        if (launches >= max_launches + 1)
            return pm_set_error(PM_ERR_NOCONVERGE, "slicer: %d walkers still alive after %d launches (%lld chunks)", alive, launches, (long long)total_chunks);
        burst = (int)std::min<int64_t>(std::max<int64_t>(8, launches / 2), max_launches + 1 - launches);
    }

    int rc = PM_OK;
    for (int k = 0; k < nj; ++k) {
        pm_slice_job &q = jobs[live[k]];
        const uint64_t bits = (uint64_t)jd[k].nb0 + h_tot[k] * (uint64_t)jd[k].bps;
        q.count = (int64_t)(bits >> 3);
        if (pm_slicer_state *st = q.h_state) {
            double clk;
            memcpy(&clk, &h_tot[(size_t)nj + k], 8);
            st->phase_clock = clk;
            st->last_i_negative = (h_tot[(size_t)2 * nj + k] & 1) ? 0 : 1;
            st->last_q_negative = jd[k].quad ? ((h_tot[(size_t)2 * nj + k] & 2) ? 0 : 1) : 0;
            st->working_bits = (int32_t)(bits & 7);
            st->working_byte = st->working_bits ? (int32_t)((h_tail[k] & 0xFF) >> (8 - st->working_bits)) : 0;
            st->state_register = (int32_t)h_tot[(size_t)3 * nj + k];
            st->streamaddress = jd[k].addr0 + jd[k].n;
        }
        if (q.count > q.cap)
            rc = pm_set_error(PM_ERR_CAPACITY, "slicer stream %d produced %lld bytes, capacity %lld", live[k], (long long)q.count, (long long)q.cap);
    }
    return rc;
}

// ---- compact form of a batch's output for the way to the host ------------------------------------------------------------------
// Per job: {first address, last address} (2 x int64), 64 flag bytes (one per workgroup of the packing launch: a step it packed did
// not fit 16 bits), count deltas as uint16 (address[i] - address[i-1], delta[0] = 0; padded to 8 bytes), count data bytes (padded
// to 8).  A byte's address is the sample at which its eighth bit was taken, so a delta is eight symbol periods (320 samples at
// 1200 Bd / 48 kHz) -- unless the input keeps the clock from ever reaching its threshold (a zero crossing at every sample does),
// so the check is a real one: a stream with a flag set has its addresses copied in full.  Every workgroup writes its own flag byte,
// set or not: nothing has to be cleared beforehand.
constexpr int kCompactHead = 16 + 64;
struct CompactJobs {
    const uint8_t *data[kMaxJobs];
    const int64_t *addr[kMaxJobs];
    int64_t count[kMaxJobs];
    int64_t off[kMaxJobs];           // byte offset of the job's header in the block
    int njobs;
};

__global__ __launch_bounds__(256) void slice_compact_kernel(CompactJobs J, uint8_t *block)
{
    const int j = blockIdx.y;
    const int64_t n = J.count[j];
    const int64_t *addr = J.addr[j];
    uint8_t *base = block + J.off[j];
    uint16_t *delta = reinterpret_cast<uint16_t *>(base + kCompactHead);
    uint8_t *bytes = base + kCompactHead + ((2 * n + 7) & ~int64_t(7));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<int64_t *>(base)[0] = n ? addr[0] : 0;
        reinterpret_cast<int64_t *>(base)[1] = n ? addr[n - 1] : 0;
    }
    int wide = 0;
    // four entries per thread: one 8-byte store of deltas, one 4-byte store of data
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; 4 * q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = 4 * q;
        int64_t prev = i ? addr[i - 1] : (n ? addr[0] : 0);
        uint64_t d4 = 0;
        uint32_t b4 = 0;
        for (int k = 0; k < 4; ++k) {
            if (i + k < n) {
                const int64_t a = addr[i + k];
                wide |= (uint64_t)(a - prev) > 65535u;
                d4 |= (uint64_t)(uint16_t)(a - prev) << (16 * k);
                b4 |= (uint32_t)J.data[j][i + k] << (8 * k);
                prev = a;
            }
        }
        *reinterpret_cast<uint64_t *>(delta + i) = d4;       // the padding makes the last, partly filled store legal
        *reinterpret_cast<uint32_t *>(bytes + i) = b4;
    }
    wide = __syncthreads_or(wide);
    if (threadIdx.x == 0) base[16 + blockIdx.x] = (uint8_t)(wide != 0);      // gridDim.x <= 64
    if (blockIdx.x == 0 && threadIdx.x >= gridDim.x && threadIdx.x < 64) base[16 + threadIdx.x] = 0;
}

extern "C" {

int pm_slice_binary(pm_ctx *ctx, const uint64_t *d_bits, int64_t n, const pm_slicer_params *h_params,
                    uint8_t *d_data, int64_t *d_addr, int64_t cap, int64_t *h_count)
{
    PM_ARG(h_params && h_count && h_params->bits_per_symbol == 1);
    pm_slice_job job;
    memset(&job, 0, sizeof(job));
    job.d_bits_i = d_bits;
    job.n = n;
    job.params = *h_params;
    job.d_data = d_data;
    job.d_addr = d_addr;
    job.cap = cap;
    const int rc = pm_slice_batch(ctx, &job, 1);
    *h_count = job.count;
    return rc;
}

int pm_slice_quadrature(pm_ctx *ctx, const uint64_t *d_bits_i, const uint64_t *d_bits_q, int64_t n,
                        const pm_slicer_params *h_params, uint8_t *d_data, int64_t *d_addr, int64_t cap, int64_t *h_count)
{
    PM_ARG(h_params && h_count && (n == 0 || d_bits_q));
    pm_slice_job job;
    memset(&job, 0, sizeof(job));
    job.d_bits_i = d_bits_i;
    job.d_bits_q = d_bits_q;
    job.n = n;
    job.params = *h_params;
    job.d_data = d_data;
    job.d_addr = d_addr;
    job.cap = cap;
    const int rc = pm_slice_batch(ctx, &job, 1);
    *h_count = job.count;
    return rc;
}

int pm_slice_compact(pm_ctx *ctx, const pm_slice_job *h_jobs, int njobs, void *d_block, size_t block_bytes, int64_t *h_offsets,
                     size_t *h_used)
{
    PM_CTX(ctx);
    PM_ARG(h_jobs && njobs >= 1 && njobs <= kMaxJobs && d_block && h_offsets && h_used);
    CompactJobs J;
    memset(&J, 0, sizeof(J));
    J.njobs = njobs;
    size_t at = 0;
    int64_t most = 0;
    for (int j = 0; j < njobs; ++j) {
        const pm_slice_job &q = h_jobs[j];
        PM_ARG(q.count >= 0 && q.count <= q.cap && (q.count == 0 || (q.d_data && q.d_addr)));
        J.data[j] = q.d_data;
        J.addr[j] = q.d_addr;
        J.count[j] = q.count;
        J.off[j] = h_offsets[j] = (int64_t)at;
        at += kCompactHead + (size_t)((2 * q.count + 7) & ~int64_t(7)) + (size_t)((q.count + 7) & ~int64_t(7));
        most = std::max(most, q.count);
    }
    *h_used = at;
    if (at > block_bytes) return pm_set_error(PM_ERR_CAPACITY, "pm_slice_compact: the block holds %zu bytes, the batch needs %zu", block_bytes, at);
    PmProf prof(ctx, PM_K_SLICE_EMIT);
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>(64, (most / 4 + 255) / 256));
    hipLaunchKernelGGL(slice_compact_kernel, dim3(bx, njobs), dim3(256), 0, ctx->stream, J, static_cast<uint8_t *>(d_block));
    PM_HIP(hipGetLastError());
    return PM_OK;
}

int pm_slicer_tune(pm_ctx *ctx, int64_t target_lanes)
{
    PM_ARG(ctx != nullptr && target_lanes >= 0);
    ctx->sl_target_lanes = target_lanes ? std::max<int64_t>(64, target_lanes) : 16384;
    return PM_OK;
}

int pm_slicer_limits(pm_ctx *ctx, int64_t max_chunk_words)
{
    PM_ARG(ctx != nullptr && max_chunk_words >= 0);
    ctx->sl_max_chunk_words = max_chunk_words ? std::max<int64_t>(16, max_chunk_words) : 384;
    return PM_OK;
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

// ---- the batch engine's row slicers (rowslice_kernel) ----------------------------------------------------------------------------
struct pm_rowslice {
    int chains = 0, step = 0, device = 0;
    RowParams *d_params = nullptr;
    std::vector<pm_slicer_params> made_for;
};

int pm_rowslice_create(pm_ctx *ctx, const pm_slicer_params *h_params, int chains, pm_rowslice **out)
{
    PM_CTX(ctx);
    PM_ARG(h_params != nullptr && out != nullptr && chains >= 1 && chains <= 4096);
    *out = nullptr;
    std::vector<RowParams> rp((size_t)chains);
    const pm_tuning &tn = ctx->tune;
    int masks = tn.slicer_compare_step ? 0 : 2;              // the choice of pm_slice_batch, over the chains' parameter sets
    bool lm0 = true, ns0 = true, direct = true;
    for (int c = 0; c < chains; ++c) {
        const pm_slicer_params &q = h_params[c];
        PM_ARG(q.bits_per_symbol == 1 || q.bits_per_symbol == 2);
        PM_ARG(q.samples_per_symbol > 0.0 && q.lock_rate == q.lock_rate);
        RowParams &d = rp[(size_t)c];
        d.sps = q.samples_per_symbol;
        d.thr = (q.samples_per_symbol / 2.0) - 0.5;          // slicer.py:52
        d.tp = symbol_clock_threshold(d.thr);
        d.lock = q.lock_rate;
        d.bps = q.bits_per_symbol;
        d.mask = q.state_mask;
        d.demap4 = 0;
        for (int i = 0; i < 16; ++i) {
            PM_ARG(q.demap[i] >= 0 && q.demap[i] < (1 << q.bits_per_symbol));
            d.demap4 |= (unsigned long long)q.demap[i] << (4 * i);
        }
        const volatile double lm1 = d.lock - 1.0;
        if (!(lm1 + 1.0 == d.lock) || !(d.sps - d.sps == 0.0)) masks = 0;
        uint64_t lb, sb;
        const double l1 = lm1;
        memcpy(&lb, &l1, 8);
        memcpy(&sb, &d.sps, 8);
        if (masks == 2 && ((uint32_t)lb || (uint32_t)sb)) masks = 1;
        lm0 = lm0 && (uint32_t)lb == 0;
        ns0 = ns0 && (uint32_t)sb == 0;
        direct = direct && d.tp == d.tp;
    }
    direct = direct && masks != 0 && !tn.slicer_mask_step;
    const bool hand = direct && !tn.slicer_compiled_step;
    pm_rowslice *rs = new pm_rowslice();
    rs->chains = chains;
    rs->device = ctx->device;
    rs->step = hand ? (lm0 ? (ns0 ? 8 : 6) : (ns0 ? 7 : 5)) : direct ? (masks == 2 ? 4 : 3) : masks;
    rs->made_for.assign(h_params, h_params + chains);
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void **)&rs->d_params, sizeof(RowParams) * (size_t)chains) != hipSuccess) {
        delete rs;
        return pm_set_error(PM_ERR_HIP, "row slicers: no device memory for the parameter table");
    }
    if (hipMemcpy(rs->d_params, rp.data(), sizeof(RowParams) * (size_t)chains, hipMemcpyHostToDevice) != hipSuccess) {
        pm_rowslice_destroy(rs);
        return pm_set_error(PM_ERR_HIP, "row slicers: copying the parameter table failed");
    }
    *out = rs;
    return PM_OK;
}

void pm_rowslice_destroy(pm_rowslice *rs)
{
    if (!rs) return;
    (void)hipSetDevice(rs->device);
    if (rs->d_params) (void)hipFree(rs->d_params);
    delete rs;
}

bool pm_rowslice_made_for(const pm_rowslice *rs, const pm_slicer_params *h_params, int chains)
{
    return rs && rs->chains == chains && memcmp(rs->made_for.data(), h_params, sizeof(pm_slicer_params) * (size_t)chains) == 0;
}

int pm_rowslice_chunk(pm_ctx *ctx, const pm_rowslice *rs, int rows, int quad, const uint64_t *d_bi, const uint64_t *d_bq, int64_t stride, int64_t first,
                      int64_t count, pm_rowslice_rec *d_recs, uint8_t *d_data, uint16_t *d_steps, int64_t cap)
{
    PM_CTX(ctx);
    PM_ARG(rs != nullptr && rows >= 1 && d_bi != nullptr && (!quad || d_bq != nullptr) && d_recs != nullptr && d_data != nullptr && d_steps != nullptr);
    PM_ARG(first >= 0 && first % 64 == 0 && count >= 1 && stride >= (count + 63) / 64 && cap >= 8 && cap % 8 == 0 && rs->device == ctx->device);
    PmProf prof(ctx, PM_K_SLICE_ITER);
    const dim3 grid((unsigned)pm_cdiv(rows, 64)), block(64);
#define PM_ROWSLICE(S) hipLaunchKernelGGL(rowslice_kernel<S>, grid, block, 0, ctx->stream, rs->d_params, rs->chains, rows, quad, d_bi, d_bq, stride, first, count, \
                                          d_recs, d_data, d_steps, cap)
    switch (rs->step) {
    case 8: PM_ROWSLICE(8); break;
    case 7: PM_ROWSLICE(7); break;
    case 6: PM_ROWSLICE(6); break;
    case 5: PM_ROWSLICE(5); break;
    case 4: PM_ROWSLICE(4); break;
    case 3: PM_ROWSLICE(3); break;
    case 2: PM_ROWSLICE(2); break;
    case 1: PM_ROWSLICE(1); break;
    default: PM_ROWSLICE(0); break;
    }
#undef PM_ROWSLICE
    PM_HIP(hipGetLastError());
    return PM_OK;
}

extern "C" int pm_rows_gather(pm_ctx *ctx, const pm_rowslice_rec *d_recs, const uint8_t *d_data, const uint16_t *d_steps, int64_t cap, int64_t row0, int nrows,
                              void *d_block, size_t block_bytes)
{
    PM_CTX(ctx);
    PM_ARG(d_recs != nullptr && d_data != nullptr && d_steps != nullptr && d_block != nullptr && cap >= 8 && cap % 8 == 0 && row0 >= 0 && nrows >= 1 &&
           nrows <= 4096 && ((uintptr_t)d_block & 7) == 0);
    hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)nrows), dim3(256), 0, ctx->stream, d_recs, d_data, d_steps, cap, row0, (uint8_t *)d_block, block_bytes);
    PM_HIP(hipGetLastError());
    return PM_OK;
}
