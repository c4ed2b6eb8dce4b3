// Symbol-timing slicers (BinarySlicer.slice slicer.py:59-107, QuadratureSlicer.slice slicer.py:193-242)
// evaluated chunk-parallel on the sign bitmap(s) of the demodulated stream(s), many streams per launch.
//
// The reference recurrence, per sample k (1-based address k+1):
//     clk += 1.0;  if (clk >= sps/2 - 0.5) { clk -= sps;  take a symbol from sign(x[k]) }
//     if sign(x[k]) != sign(x[k-1]):  clk *= lock_rate
// is sequential in `clk` only.  Every stream is cut into chunks of L samples, one lane per chunk.
// An iteration runs the chunks on its work list from the start states handed to them; a chunk whose end
// state changes hands it to the next chunk and puts that chunk on the next list; chunk 0 of a stream
// always starts from the true state.  When a list comes up empty, every chunk's last run started from
// the final end state of its predecessor, so by induction from chunk 0 the per-chunk runs ARE the
// sequential run, bit for bit: each lane executes the reference's operations in the reference's order,
// only the starting value is guessed.  Two trajectories that see the same zero crossings contract by lock_rate per
// crossing, so a few iterations suffice on real signals; the worst case (no crossings at all) degrades
// to nchunks iterations, i.e. sequential cost, never to a wrong answer.
//
// A lone wave pays both the loop-carried chain (add -> compare -> select -> multiply, ~45 cycles with the latencies measured
// by tools/ubench) and the issue of the ~12 VALU instructions of a step (~5.5 cycles each): ~85 cycles per sample.  Each run
// leaves a bitmap of the samples at which it took a symbol.  After the fixed point a count/scan/pack pipeline, chunked on its
// own (finely), turns symbol bitmap + sign bitmap(s) into bytes and the 1-based address of each byte's last symbol.  A slicer
// object's state (clock, last sign, open byte, address count, differential state) enters and leaves through pm_slicer_state.
#include "pm_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kMaxJobs = 64;
constexpr int kBlock = 256;     // four waves: one per SIMD when the grid is one workgroup per CU

struct JobDev {
    const uint64_t *bi, *bq;       // sign bitmaps (bq null for binary)
    int64_t n, nwords;
    int64_t chunk0, nchunks;       // global chunk range of this stream
    int64_t word0;                 // offset of this stream in the global symbol bitmap
    double thr, sps, lock;
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

__device__ __forceinline__ int find_job(const JobDev *jobs, int njobs, int64_t gc)
{
    int j = 0;
    while (j + 1 < njobs && gc >= jobs[j + 1].chunk0) ++j;
    return j;
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

// One fixed-point iteration.  Work is a list of chunk ids: iteration 0 holds every chunk; a chunk whose run changes its END state
// writes it into the state array and puts its successor on the next iteration's list.  Lists are dense, so the waves of an iteration
// are as many as there are chunks to re-run (the thin tail of the iteration costs a handful of waves, not the whole grid) and the
// iteration after the fixed point finds an empty list.  The state array is updated in place: a chunk may read its start state
// while its predecessor is rewriting it in the same iteration -- either value is a valid 8-byte state, and whenever the
// predecessor did change it the chunk is on the next list and runs again, so at the fixed point every chunk's last run started
// from the final end state of its predecessor (the induction of the header comment).
// STEP: 0 = step32 (compare and selects), 1 = step32m, 2 = step32m with zero low words in lock_rate - 1 and sps.  One kernel per
// form: with the 64 unrolled steps of several forms in one kernel the loop no longer fits the instruction cache comfortably.
template <int STEP>
__global__ __launch_bounds__(kBlock) void slice_iter_kernel(const JobDev *__restrict__ jobs, int njobs, int lc_words,
                                                        uint64_t *__restrict__ state, const int32_t *__restrict__ list_in,
                                                        int32_t *__restrict__ list_out, int *__restrict__ counts, int iter,
                                                        uint64_t *__restrict__ symmap)
{
    // These waves are bound by their own dependent chain; when FIR waves of another stream share the SIMD (pipelined executor)
    // every issue slot they lose lengthens the chain, while the FIR waves only need the slots in between: take issue priority.
    __builtin_amdgcn_s_setprio(3);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counts[iter]) return;                   // whole waves beyond the list leave at once
    const int64_t gc = list_in[i];
    const int j = find_job(jobs, njobs, gc);
    const JobDev &J = jobs[j];
    const int64_t c = gc - J.chunk0;
    const int64_t so = gc + j;                       // the state array holds nchunks+1 entries per stream
    double clk = bitsd(__builtin_nontemporal_load(&state[so]));
    const int64_t w0 = c * lc_words;
    const int64_t w1 = min(w0 + (int64_t)lc_words, J.nwords);
    // last_sample starts at 0.0, i.e. ">= 0" (slicer.py:55,164-165)
    uint64_t li = w0 == 0 ? (uint64_t)J.li0 : (J.bi[w0 - 1] >> 63);
    uint64_t lq = 1ull;
    if (J.quad) lq = w0 == 0 ? (uint64_t)J.lq0 : (J.bq[w0 - 1] >> 63);
    const double thr = J.thr, neg_sps = -J.sps;
    const double lock = J.lock;
    const double lm1 = lock - 1.0;
    uint64_t *sm = symmap + J.word0;
    for (int64_t w = w0; w < w1; ++w) {
        const uint64_t si = J.bi[w];
        uint64_t zc = si ^ ((si << 1) | li);
        li = si >> 63;
        if (J.quad) {
            const uint64_t sq = J.bq[w];
            zc |= sq ^ ((sq << 1) | lq);
            lq = sq >> 63;
        }
        const int64_t left = J.n - (w << 6);
        uint64_t sym;
        if (left >= 64) {
            uint32_t lo, hi;
            if (STEP == 2) {
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
                    clk -= J.sps;
                    sym |= 1ull << b;
                }
                if ((zc >> b) & 1) clk = clk * J.lock;
            }
        }
        sm[w] = sym;
    }
    const uint64_t e = dbits(clk);
    const bool ch = e != state[so + 1];
    if (ch) state[so + 1] = e;
    // successors of the chunks whose end state moved go on the next list: one atomic per wave
    const bool add = ch && c + 1 < J.nchunks;
    const uint64_t mask = __ballot(add);
    if (mask) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)mask) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&counts[iter + 1], __popcll(mask));
        base = __shfl(base, leader);
        if (add) list_out[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)(gc + 1);
    }
}

// Symbols per chunk and the (i<<1|q) bits of its last symbol (0xFF if it took none).
__global__ __launch_bounds__(kBlock) void slice_count_kernel(const JobDev *__restrict__ jobs, int njobs, int lc_words, int64_t total_chunks,
                                                         const uint64_t *__restrict__ symmap, uint32_t *__restrict__ count,
                                                         uint8_t *__restrict__ lastsym)
{
    __builtin_amdgcn_s_setprio(3);          // short kernels on the slicer stream's critical path (see slice_iter_kernel)
    const int64_t gc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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
                                                          const uint64_t *__restrict__ s_end, int njobs,
                                                          const JobDev *__restrict__ iter_jobs)
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
        const JobDev &I = iter_jobs[blockIdx.x];           // the state arrays are laid out by the ITERATION's chunks
        totals[njobs + blockIdx.x] = s_end[I.chunk0 + blockIdx.x + I.nchunks];
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
__global__ __launch_bounds__(kBlock) void slice_pack_kernel(const JobDev *__restrict__ jobs, int njobs, int lc_words, int64_t total_chunks,
                                                        const uint64_t *__restrict__ symmap, const uint64_t *__restrict__ offset,
                                                        const uint8_t *__restrict__ prevsym)
{
    __shared__ uint32_t lb[kPackBytes / 4 + 1];
    __shared__ long long la[kPackBytes + 4];
    __builtin_amdgcn_s_setprio(3);          // short kernels on the slicer stream's critical path (see slice_iter_kernel)
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
    auto put = [&](uint64_t idx, uint32_t bits8, long long address) {
        if (staged) {
            const int rel = (int)(idx - base);
            atomicOr(&lb[rel >> 2], bits8 << ((rel & 3) * 8));
            if (address) la[rel] = address;
        } else if (idx < cap) {
            atomicOr(&J.data32[idx >> 2], bits8 << ((idx & 3) * 8));
            if (address) J.addr[idx] = address;
        }
    };
    if (live) {
        uint64_t g = off[c];
        if (c == 0 && nb0) {
            const uint32_t head = ((uint32_t)J.wb0 & ((1u << nb0) - 1u)) << (8 - nb0);
            if (nbytes > 0) put(0, head, 0);
            else atomicOr(J.tail, head);
        }
        uint32_t prev = prevsym[gc];
        const int64_t w0 = c * lc_words, w1 = min(w0 + (int64_t)lc_words, J.nwords);
        const uint64_t *sm = symmap + J.word0;
        uint32_t acc = 0;
        bool pending = false;
        for (int64_t w = w0; w < w1; ++w) {
            uint64_t s = sm[w];
            if (!s) continue;
            const uint64_t si = J.bi[w];
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
        const long long a = la[i];
        if (a && base + i < cap) J.addr[base + i] = a;
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

__global__ void slice_init_kernel(const JobDev *__restrict__ jobs, int njobs, int64_t total_chunks, uint64_t *state, int32_t *list0,
                                  int *counts, int ncounts)
{
    const int64_t gc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gc < ncounts) counts[gc] = gc == 0 ? (int)total_chunks : 0;       // iteration 0 runs every chunk
    if (gc >= total_chunks) return;
    const int j = find_job(jobs, njobs, gc);
    const int64_t so = gc + j;
    // cold start everywhere (phase_clock = 0.0); chunk 0 of a stream starts from its true state, carried in or zero (slicer.py:50)
    state[so] = gc == jobs[j].chunk0 ? dbits(jobs[j].clk0) : 0ull;
    if (gc - jobs[j].chunk0 == jobs[j].nchunks - 1) state[so + 1] = ~0ull;   // "no end state yet": any first run differs from it
    list0[gc] = (int32_t)gc;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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

    // Chunk length (a multiple of 64 samples).  A lone wave is issue bound, so an iteration costs (waves per SIMD) x L x
    // t_step while the number of iterations falls as 1/L: the best L puts about one wave on each of the 1024 SIMDs
    // (65536 lanes), but never below 1024 samples.  PM_SLICER_CHUNK_WORDS overrides (tuning).
    int64_t all_words = 0;
    for (int j = 0; j < njobs; ++j) all_words += pm_cdiv(jobs[j].n, 64);
    int64_t lc_words = std::max<int64_t>(16, std::min<int64_t>(pm_cdiv(all_words, ctx->sl_target_lanes), 1024));
    if (const char *e = getenv("PM_SLICER_CHUNK_WORDS")) { if (atoi(e) > 0) lc_words = atoi(e); }
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
    ctx->sl_chunk_len = (int32_t)(lc_words * 64);
    ctx->sl_chunks = total_chunks;
    // The count / scan / pack kernels only read the symbol bitmap the iteration left: they are cut independently of it, finely
    // (throughput kernels: ~4 waves per SIMD), however long the iteration's chunks are.
    const int64_t le_words = std::max<int64_t>(4, std::min<int64_t>(lc_words, pm_cdiv(total_words, 524288)));
    std::vector<JobDev> je = jd;
    int64_t emit_chunks = 0;
    for (JobDev &d : je) {
        d.chunk0 = emit_chunks;
        d.nchunks = pm_cdiv(d.nwords, le_words);
        emit_chunks += pm_cdiv(d.nchunks, (int64_t)kBlock) * kBlock;      // a workgroup never straddles two streams (slice_pack_kernel)
    }

    const size_t e = (size_t)total_chunks + nj;             // nchunks+1 state entries per stream
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t ee = (size_t)emit_chunks + nj;
    int64_t most_chunks = 0;
    for (const JobDev &d : jd) most_chunks = std::max(most_chunks, d.nchunks);
    const int burst = 8;                                   // iterations after the fixed point find an empty list and cost microseconds
    const int64_t max_iters = most_chunks + 2;
    const int ncounts = (int)std::min<int64_t>(max_iters + 2 * burst + 8, 1 << 22);
    const size_t o_jobs = carve(sizeof(JobDev) * nj), o_ejobs = carve(sizeof(JobDev) * nj), o_state = carve(e * 8),
                 o_la = carve((size_t)total_chunks * 4), o_lb = carve((size_t)total_chunks * 4), o_counts = carve((size_t)ncounts * 4),
                 o_cnt = carve((size_t)emit_chunks * 4), o_ls = carve(emit_chunks), o_off = carve(ee * 8), o_ps = carve(emit_chunks),
                 o_sym = carve((size_t)total_words * 8), o_tot = carve((size_t)nj * 8 * 4), o_tail = carve((size_t)nj * 4);
    if (int rc = pm_scratch_reserve(ctx, off)) return rc;
    char *base = (char *)ctx->d_scratch;
    JobDev *d_jobs = (JobDev *)(base + o_jobs), *d_ejobs = (JobDev *)(base + o_ejobs);
    uint64_t *state = (uint64_t *)(base + o_state);
    int32_t *list_a = (int32_t *)(base + o_la), *list_b = (int32_t *)(base + o_lb);
    int *counts = (int *)(base + o_counts);
    uint32_t *cnt = (uint32_t *)(base + o_cnt);
    uint8_t *ls = (uint8_t *)(base + o_ls), *ps = (uint8_t *)(base + o_ps);
    uint64_t *offs = (uint64_t *)(base + o_off), *symmap = (uint64_t *)(base + o_sym), *totals = (uint64_t *)(base + o_tot);
    uint32_t *tails = (uint32_t *)(base + o_tail);
    for (int k = 0; k < nj; ++k) jd[k].tail = je[k].tail = tails + k;
    PM_HIP(hipMemsetAsync(tails, 0, (size_t)nj * 4, ctx->stream));

    PM_HIP(hipMemcpyAsync(d_jobs, jd.data(), sizeof(JobDev) * nj, hipMemcpyHostToDevice, ctx->stream));
    PM_HIP(hipMemcpyAsync(d_ejobs, je.data(), sizeof(JobDev) * nj, hipMemcpyHostToDevice, ctx->stream));
    const unsigned grid = (unsigned)pm_cdiv(std::max<int64_t>(total_chunks, ncounts), kBlock);
    hipLaunchKernelGGL(slice_init_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_jobs, nj, total_chunks, state, list_a, counts, ncounts);

    // step32m needs lock_rate - 1 to be exact for every stream of the batch (it is for 0.5 <= lock_rate <= 2) and finite clocks
    int masks = getenv("PM_SLICER_COMPARE_STEP") ? 0 : 2;
    for (const JobDev &d : jd) {
        const volatile double lm1 = d.lock - 1.0;
        if (!(lm1 + 1.0 == d.lock) || !(d.clk0 - d.clk0 == 0.0) || !(d.sps - d.sps == 0.0)) masks = 0;
        uint64_t lb, sb;
        const double l1 = lm1;
        memcpy(&lb, &l1, 8);
        memcpy(&sb, &d.sps, 8);
        if (masks == 2 && ((uint32_t)lb || (uint32_t)sb)) masks = 1;     // low words not zero: the general mask form
    }
    auto iter_kernel = masks == 2 ? slice_iter_kernel<2> : masks == 1 ? slice_iter_kernel<1> : slice_iter_kernel<0>;
    int *h_flag = (int *)ctx->h_pinned;
    int iters = 0;
    bool converged = false;
    while (!converged) {
        // a burst of iterations between host checks keeps the launch queue full
        for (int b = 0; b < burst; ++b) {
            PmProf prof(ctx, PM_K_SLICE_ITER);
            hipLaunchKernelGGL(iter_kernel, dim3((unsigned)pm_cdiv(total_chunks, kBlock)), dim3(kBlock), 0, ctx->stream, d_jobs, nj,
                               (int)lc_words, state, (iters & 1) ? list_b : list_a, (iters & 1) ? list_a : list_b, counts, iters, symmap);
            ++iters;
        }
        // counts[iters] = chunks the burst's last iteration put on the next list: none means the fixed point is reached
        PM_HIP(hipMemcpyAsync(h_flag, counts + iters, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(hipStreamSynchronize(ctx->stream));
        converged = (*h_flag == 0);
        if (getenv("PM_SLICER_TRACE")) fprintf(stderr, "[slicer] after %d iterations: %d chunks still to re-run\n", iters, *h_flag);
        if (!converged && (iters > max_iters + burst || iters + burst + 1 >= ncounts))
            return pm_set_error(PM_ERR_NOCONVERGE, "slicer fixed point not reached after %d iterations (%lld chunks)", iters, (long long)most_chunks);
    }
    ctx->sl_iterations = iters;

    {
        PmProf prof(ctx, PM_K_SLICE_EMIT);
        const unsigned egrid = (unsigned)pm_cdiv(emit_chunks, kBlock);
        hipLaunchKernelGGL(slice_count_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, d_ejobs, nj, (int)le_words, emit_chunks, symmap, cnt, ls);
        hipLaunchKernelGGL(slice_scan_kernel, dim3(nj), dim3(1024), 0, ctx->stream, d_ejobs, cnt, ls, offs, ps, totals, state, nj, d_jobs);
        for (const JobDev &d : jd)
            if (d.cap > 0) PM_HIP(hipMemsetAsync(d.data32, 0, align_up((size_t)d.cap, 4), ctx->stream));
        hipLaunchKernelGGL(slice_pack_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, d_ejobs, nj, (int)le_words, emit_chunks, symmap, offs, ps);
    }
    std::vector<uint64_t> h_tot((size_t)nj * 4);
    std::vector<uint32_t> h_tail(nj);
    PM_HIP(hipMemcpyAsync(h_tot.data(), totals, (size_t)nj * 8 * 4, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipMemcpyAsync(h_tail.data(), tails, (size_t)nj * 4, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    PM_HIP(hipGetLastError());
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

int pm_slicer_tune(pm_ctx *ctx, int64_t target_lanes)
{
    PM_ARG(ctx != nullptr && target_lanes >= 0);
    ctx->sl_target_lanes = target_lanes ? std::max<int64_t>(64, target_lanes) : 65536;
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
