// Micro-benchmark: does the int8 matrix pipe (v_mfma_i32_16x16x64_i8, gfx950) run beside f64 vector fma?  Question behind it: the shared
// band-pass of the AFSK group (148 taps on int16 audio, f64) as exact int8-limb products on the matrix pipe (12 int8 MACs per tap).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
template <int MODE, int V>
__global__ __launch_bounds__(256) void k(double *out, double c1, double c2, int iters)
{
    i4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = i4{0, 0, 0, 0};
    i4 A = i4{(int)threadIdx.x, 1, 2, 3}, B = i4{7, (int)threadIdx.x, 5, 1};
    double a[16];
    for (int j = 0; j < 16; ++j) a[j] = threadIdx.x + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == 0 || MODE == 2) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[q], 0, 0, 0);
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int j = 0; j < V; ++j) a[(q * V + j) & 15] = __builtin_fma(a[(q * V + j) & 15], c2, c1);
            }
        }
    }
    double s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    for (int j = 0; j < 16; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int V>
double run(const char *name, int blocks, int iters)
{
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, V><<<blocks, 256>>>(out, 1.0, 0.999, iters);
    hipEventRecord(e0, 0);
    k<MODE, V><<<blocks, 256>>>(out, 1.0, 0.999, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4, mf = (MODE == 0 || MODE == 2) ? waves * iters * 4.0 * 16384 : 0, vf = (MODE == 1 || MODE == 2) ? waves * iters * 4.0 * V * 64 : 0;
    printf("%-44s %8.3f ms  matrix %8.2f T mac/s  vector %6.2f T fma/s\n", name, ms, mf / ms / 1e9, vf / ms / 1e9);
    hipFree(out);
    return ms;
}
int main()
{
    const int blocks = 4096, it = 4000;
    run<0, 4>("mfma i32 16x16x64 i8 only", blocks, it);
    run<1, 4>("vector fma f64 only (4 per slot)", blocks, it * 4);
    run<2, 2>("same wave: 1 mfma : 2 vector fma", blocks, it);
    run<2, 4>("same wave: 1 mfma : 4 vector fma", blocks, it);
    run<2, 8>("same wave: 1 mfma : 8 vector fma", blocks, it);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    double *o1, *o2; hipMalloc(&o1, 8 * blocks * 256); hipMalloc(&o2, 8 * blocks * 256);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    hipDeviceSynchronize();
    hipEventRecord(e0, s1);
    hipStreamWaitEvent(s2, e0, 0);
    k<0, 4><<<blocks / 2, 256, 0, s1>>>(o1, 1.0, 0.999, it);
    k<1, 4><<<blocks / 2, 256, 0, s2>>>(o2, 1.0, 0.999, it * 4);
    hipEventRecord(e1, s1); hipEventRecord(e2, s2);
    hipEventSynchronize(e1); hipEventSynchronize(e2);
    float m1, m2; hipEventElapsedTime(&m1, e0, e1); hipEventElapsedTime(&m2, e0, e2);
    const double waves = (double)blocks / 2 * 4;
    printf("two streams: matrix kernel %.3f ms (%.2f T mac/s), vector kernel %.3f ms (%.2f T fma/s)\n", m1, waves * it * 4.0 * 16384 / m1 / 1e9, m2,
           waves * it * 4 * 4.0 * 4 * 64 / m2 / 1e9);
    return 0;
}
