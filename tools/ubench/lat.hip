// Micro-benchmark: dependent-chain latency and single-wave issue rate of the f64 / f32 VALU ops the sequential
// kernels (slicer clock, carrier loops) are made of.  One wave per SIMD unless WAVES is raised.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N 4096
template <int MODE>
__global__ void k(double *out, double a0, double c1, double c2, long long *cyc)
{
    double a = a0 + threadIdx.x, b = a0 * 2 + threadIdx.x, c = a0 * 3, d = a0 * 4;
    long long t0 = __builtin_readcyclecounter();
    long long s0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 16; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c1)); }
            if (MODE == 1) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(c2)); }
            if (MODE == 2) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(c2), "v"(c1)); }
            if (MODE == 3) { asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(c1)); }
            if (MODE == 4) { float x = (float)a; asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"((float)c1)); a = x; }
            if (MODE == 5) {
                int lo = __double2loint(a), hi = __double2hiint(a);
                asm volatile("v_cmp_ge_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc"
                             : "+v"(lo), "+v"(hi) : "v"(a), "v"(c1), "v"(lo ^ 1), "v"(hi) : "vcc");
                a = __hiloint2double(hi, lo);
            }
            if (MODE == 6) { asm volatile("v_add_f64 %0, %0, %1\n v_mul_f64 %0, %0, %2" : "+v"(a) : "v"(c1), "v"(c2)); }
        }
    }
    long long s1 = clock64();
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = s1 - s0; }
}
template <int MODE>
void run(const char *name, int opsPerIter, int blocks, int threads)
{
    double *out; long long *cyc, h[2];
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 1.0, 1.0, 0.999, cyc);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, 1.0, 1.0, 0.999, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-34s blocks=%5d threads=%4d  %8.3f us  %7.2f ns/op-group  clock64 ticks/op %.2f  realtime ticks/op %.3f\n", name, blocks, threads, ms * 1e3,
           ms * 1e6 / (N * 1.0), (double)h[1] / N, (double)h[0] / N);
    hipFree(out); hipFree(cyc);
}
// Sustained vector-f64 FMA rate: 16 independent accumulators per lane, every SIMD full.
__global__ __launch_bounds__(256) void fma_peak(double *out, double c1, double c2, int iters)
{
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = threadIdx.x + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = __builtin_fma(a[j], c1, c2);
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Streaming copy: what HBM delivers to a plain 16-byte-per-lane read+write kernel (1 GiB in, 1 GiB out).
__global__ __launch_bounds__(256) void copy_kernel(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read_kernel(const double2 *__restrict__ a, double *__restrict__ out, size_t n)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}

static void copy_peak()
{
    const size_t n = (size_t)1 << 26;               // 2^26 x 16 B = 1 GiB
    double2 *a, *b; double *o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, 8);
    hipMemset(a, 1, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        copy_kernel<<<256 * 16, 256>>>(a, b, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipEventRecord(e0);
        read_kernel<<<256 * 16, 256>>>(a, o, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms2; hipEventElapsedTime(&ms2, e0, e1);
        printf("stream copy 1 GiB -> 1 GiB: %.3f ms = %.0f GB/s (read+write);  read-only 1 GiB: %.3f ms = %.0f GB/s\n", ms, 2.0 * n * 16 / ms / 1e6, ms2,
               1.0 * n * 16 / ms2 / 1e6);
    }
    hipFree(a); hipFree(b); hipFree(o);
}

typedef float f2 __attribute__((ext_vector_type(2)));
// Sustained packed-f32 FMA rate (v_pk_fma_f32: two fmas per lane per instruction).
__global__ __launch_bounds__(256) void pkfma_peak(float *out, float c1, float c2, int iters)
{
    f2 a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = f2{(float)threadIdx.x + j, (float)j};
    const f2 g = {c1, c1}, b = {c2, c2};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = __builtin_elementwise_fma(a[j], g, b);
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a[j].x + a[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static void pk_peak()
{
    float *out;
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    pkfma_peak<<<blocks, threads>>>(out, 0.999f, 0.5f, 100);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        pkfma_peak<<<blocks, threads>>>(out, 0.999f, 0.5f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fma = (double)blocks * threads * 32.0 * iters;
        printf("sustained v_pk_fma_f32: %.1f ms  %.2f TFMA/s = %.1f TFLOP/s\n", ms, fma / ms / 1e9, 2 * fma / ms / 1e9);
    }
    hipFree(out);
}

static void peak()
{
    double *out;
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fma_peak<<<blocks, threads>>>(out, 0.999, 0.5, 100);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        fma_peak<<<blocks, threads>>>(out, 0.999, 0.5, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fma = (double)blocks * threads * 16.0 * iters;
        printf("sustained v_fma_f64: %.1f ms  %.2f TFMA/s = %.1f TFLOP/s  (equivalent clock at 64 FMA/clk/CU x 256 CU: %.2f GHz)\n", ms, fma / ms / 1e9,
               2 * fma / ms / 1e9, fma / ms / 1e6 / (256.0 * 64.0) / 1e3 * 1e0);
    }
    hipFree(out);
}

int main()
{
    copy_peak();
    peak();
    pk_peak();
    for (int threads : {64, 128, 256, 512}) {
        const int blocks = 256;
        run<0>("dependent v_add_f64", 1, blocks, threads);
        run<1>("dependent v_mul_f64", 1, blocks, threads);
        run<2>("dependent v_fma_f64", 1, blocks, threads);
        run<3>("4 independent v_add_f64", 4, blocks, threads);
        run<5>("dependent cmp_f64+cndmask", 1, blocks, threads);
        run<6>("dependent add+mul f64", 2, blocks, threads);
    }
    return 0;
}
