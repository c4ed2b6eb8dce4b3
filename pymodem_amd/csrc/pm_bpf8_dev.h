// Device pieces of the int8 band-pass (pm_bpf8.hip has the picture and the error analysis) for kernels that run it as a STAGE of their
// own -- afsk_fused8_kernel in pm_fir.hip -- with the digit planes at run-time addresses.  The arithmetic is bpf8_kernel's, operation for
// operation: a value computed here is the value that kernel writes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pm_bpf8_dev {

typedef int i4 __attribute__((ext_vector_type(4)));

struct Scales { double s[5]; double c0; };      // = pm_bpf8.hip's (kMaxDigits + 1 weights)

// the two digit planes of samples [wg0, wg0 + span) of x (span a multiple of 8), eight samples per thread and step:
// x = 256 s1 + s0 + 128, s0 -> p0, s1 -> p1; samples past n count as zero
__device__ __forceinline__ void stage_planes_rt(const int16_t *__restrict__ x, int64_t n, int64_t wg0, int t, int threads, unsigned char *__restrict__ p0,
                                                unsigned char *__restrict__ p1, int span)
{
    for (int p = t * 8; p < span; p += threads * 8) {
        const int64_t gi = wg0 + p;
        uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        if (gi + 8 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(x + gi);
            d0 = v.x; d1 = v.y; d2 = v.z; d3 = v.w;
        } else {
            uint16_t s[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) s[q] = gi + q < n ? (uint16_t)x[gi + q] : (uint16_t)0;
            d0 = s[0] | (uint32_t)s[1] << 16; d1 = s[2] | (uint32_t)s[3] << 16; d2 = s[4] | (uint32_t)s[5] << 16; d3 = s[6] | (uint32_t)s[7] << 16;
        }
        const uint32_t lo0 = __builtin_amdgcn_perm(d1, d0, 0x06040200u) ^ 0x80808080u, lo1 = __builtin_amdgcn_perm(d3, d2, 0x06040200u) ^ 0x80808080u;
        const uint32_t hi0 = __builtin_amdgcn_perm(d1, d0, 0x07050301u), hi1 = __builtin_amdgcn_perm(d3, d2, 0x07050301u);
        *reinterpret_cast<uint2 *>(p0 + p) = make_uint2(lo0, lo1);
        *reinterpret_cast<uint2 *>(p1 + p) = make_uint2(hi0, hi1);
    }
}

// one tile of 256 outputs: lane (r, g) gets outputs tl + 16 (4 g + v) + r, v = 0..3
template <int KB, int D>
__device__ __forceinline__ void tile_values_rt(const unsigned char *__restrict__ p0, const unsigned char *__restrict__ p1, const i4 (&B)[D][KB], int tl, int lane,
                                               const Scales &sc, double (&val)[4])
{
    constexpr int kWeights = D + 1;
    const int r = lane & 15, g = lane >> 4;
    i4 acc[kWeights];
#pragma unroll
    for (int w = 0; w < kWeights; ++w) acc[w] = i4{0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int at = tl + 16 * r + 64 * kb + 16 * g;
        const i4 a0 = *reinterpret_cast<const i4 *>(p0 + at), a1 = *reinterpret_cast<const i4 *>(p1 + at);
#pragma unroll
        for (int b = 0; b < D; ++b) {
            acc[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, B[b][kb], acc[b], 0, 0, 0);
            acc[b + 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, B[b][kb], acc[b + 1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double x = sc.c0;
#pragma unroll
        for (int w = 0; w < kWeights; ++w) x = __builtin_fma((double)acc[w][v], sc.s[w], x);
        val[v] = x;
    }
}

// The same tile with the band operands fetched block by block from the plan's table (12 KB, in the vector cache after the first
// workgroup) instead of held in 4 D KB registers (48 for the sweeps' plan) for the kernel's whole life: a kernel that runs this as one stage among others keeps
// its register count -- and with it the room a co-resident wave of another kernel finds on the SIMD -- at what its other stages need.
template <int KB, int D>
__device__ __forceinline__ void tile_values_tab(const unsigned char *__restrict__ p0, const unsigned char *__restrict__ p1, const i4 *__restrict__ btab, int tl, int lane,
                                                const Scales &sc, double (&val)[4])
{
    constexpr int kWeights = D + 1;
    const int r = lane & 15, g = lane >> 4;
    i4 acc[kWeights];
#pragma unroll
    for (int w = 0; w < kWeights; ++w) acc[w] = i4{0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        i4 B[D];
#pragma unroll
        for (int b = 0; b < D; ++b) B[b] = btab[(b * KB + kb) * 64 + lane];
        const int at = tl + 16 * r + 64 * kb + 16 * g;
        const i4 a0 = *reinterpret_cast<const i4 *>(p0 + at), a1 = *reinterpret_cast<const i4 *>(p1 + at);
#pragma unroll
        for (int b = 0; b < D; ++b) {
            acc[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, B[b], acc[b], 0, 0, 0);
            acc[b + 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, B[b], acc[b + 1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double x = sc.c0;
#pragma unroll
        for (int w = 0; w < kWeights; ++w) x = __builtin_fma((double)acc[w][v], sc.s[w], x);
        val[v] = x;
    }
}

}  // namespace pm_bpf8_dev
