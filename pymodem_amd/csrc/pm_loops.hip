// Sequential recurrences of the demod_chain path: AGC (agc.py:26-80) and the three carrier loops
// (BPSK Costas psk.py:173-189, MPSK psk.py:734-747, AFSK PLL afsk_pll.py:153-165).
//
// Each recurrence carries a binary64 state from one sample to the next through quantisers
// (int(), floor(), round()), so it has to be evaluated in the reference's order: one lane owns one
// loop and steps through the samples.  These kernels are dependent-latency-bound, not HBM- or
// ALU-bound; throughput comes from running many loops at once (one lane each, 8 per wave so that
// waves spread over CUs).  What the rest of the wave does: all 64 lanes stream the input tile into LDS
// (coalesced), the owning lanes iterate over it, then all lanes stream the output tile back; for the
// AGC the 64 lanes also do the per-sample division, which is not part of the recurrence.
//
// Built with -ffp-contract=off; no fma is used here because the reference has none.
#include "pm_common.h"
#include <cstring>
#include <vector>

namespace {

constexpr int kG = 8;            // loops per wave
constexpr int kTile = 256;       // samples per LDS tile
constexpr int kPad = kTile + 1;  // row pitch (doubles): rows of different lanes start on different banks
constexpr double kTwoPi = 2.0 * 3.141592653589793;

struct LoopRegs {
    double phase_scaling, index_scaling, set_frequency, b0, b1, a1, p_rate, i_rate, i_limit, gain;
    double phase, control, sine, cosine, x0, x1, y0, integral, proportional;
};

// The bodies below are written branch-free: a lone wave pays ~5 cycles per instruction and far more per taken branch, and
// every statement is on the loop-carried path.  Each select reproduces the reference's `if`/`while` exactly:
//   while (p >= 2pi) p -= 2pi  ==  one conditional subtract, then (never in practice) the loop for what is left.
__device__ __forceinline__ void nco_update(LoopRegs &L, const double *tab)
{
    double ph = L.phase + L.phase_scaling * (L.set_frequency + L.control);   // nco.py:35
    const double down = ph - kTwoPi;
    ph = ph >= kTwoPi ? down : ph;                                          // nco.py:36-37, first trip
    const double up = ph + kTwoPi;
    ph = ph < 0 ? up : ph;                                                  // nco.py:38-39, first trip
    if (__builtin_expect(!(ph >= 0 && ph < kTwoPi), 0)) {                   // |control| beyond one turn per sample
        while (ph >= kTwoPi) ph = ph - kTwoPi;
        while (ph < 0) ph = ph + kTwoPi;
    }
    L.phase = ph;
    const int idx = (int)(ph * L.index_scaling);                            // nco.py:40, int() truncates; 0..256
    const double s_new = tab[min(idx, 255) & 255];
    L.sine = idx < 256 ? s_new : L.sine;                                    // nco.py:41-45: index 256 keeps the old value
    L.cosine = tab[(idx + 64) & 255];                                       // nco.py:46-51
}

__device__ __forceinline__ double iir_update(LoopRegs &L, double sample)
{
    L.x1 = L.x0;                                                         // iir.py:40-42
    L.x0 = sample;
    double v = 0.0;
    v += L.x0 * L.b0;                                                    // iir.py:45-46
    v += L.x1 * L.b1;
    v += L.y0 * L.a1;                                                    // iir.py:48-52 (Y[1] is the previous output)
    L.y0 = v;
    return v;
}

__device__ __forceinline__ double pi_update(LoopRegs &L, double sample)
{
    L.proportional = L.gain * L.p_rate * sample;                         // pi_control.py:26, (gain*p_rate)*sample
    double in = L.integral + L.gain * (L.i_rate * sample);               // pi_control.py:27
    in = in > L.i_limit ? L.i_limit : in;                                // pi_control.py:28-31
    in = in < -L.i_limit ? -L.i_limit : in;
    L.integral = in;
    return L.proportional + in;                                          // pi_control.py:32
}

__device__ __forceinline__ int pd_lookup(const int32_t *tbl, double re, double im)
{
    // phase_detector.py:124-149, granularity 64: floor(x*64*0.5), clip to +-63, quadrant fold
    double fr = floor(re * 64 * 0.5), fi = floor(im * 64 * 0.5);
    fr = fmin(fmax(fr, -1e9), 1e9);
    fi = fmin(fmax(fi, -1e9), 1e9);
    int r = (int)fr, i = (int)fi;
    r = min(max(r, -63), 63);                       // >= 64 -> 63, <= -64 -> -63
    i = min(max(i, -63), 63);
    const int ar = abs(r), ai = abs(i);
    // Q1 T[r][i] | Q4 T[-i][r] | Q2 T[i][-r] | Q3 T[-r][-i]
    const bool swap = (r >= 0) != (i >= 0);
    const int row = swap ? ai : ar, col = swap ? ar : ai;
    return tbl[row * 64 + col];
}

enum { kCostas = 0, kPll = 1, kMpsk = 2 };

// LDS layout (doubles): tab[256] | in0[rows_in][kPad] | in1[...] (mpsk) | out0[kG][kPad] | out1[kG][kPad] (mpsk) | pd[4096 int32] (mpsk)
template <int MODE>
__global__ __launch_bounds__(64) void loop_kernel(pm_loop *__restrict__ loops, int nloops, const double *__restrict__ table,
                                                  const int32_t *__restrict__ pd, const double *__restrict__ x0,
                                                  const double *__restrict__ x1, int64_t x_stride, int64_t n,
                                                  double *__restrict__ o0, double *__restrict__ o1, int64_t out_stride)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * kG;
    const int ng = min(kG, nloops - g0);
    const bool shared_in = x_stride == 0;
    const int rows_in = shared_in ? 1 : kG;
    double *tab = lds;
    double *in0 = tab + 256;
    double *in1 = in0 + rows_in * kPad;
    double *out0 = MODE == kMpsk ? in1 + rows_in * kPad : in1;
    double *out1 = out0 + kG * kPad;
    int32_t *pdt = (int32_t *)(out1 + kG * kPad);

    for (int i = lane; i < 256; i += 64) tab[i] = table[i];
    if (MODE == kMpsk)
        for (int i = lane; i < 4096; i += 64) pdt[i] = pd[i];

    const bool active = lane < ng;
    LoopRegs L;
    if (active) {
        const pm_loop &s = loops[g0 + lane];
        L.phase_scaling = s.phase_scaling; L.index_scaling = s.index_scaling; L.set_frequency = s.set_frequency;
        L.b0 = s.b0; L.b1 = s.b1; L.a1 = s.a1;
        L.p_rate = s.p_rate; L.i_rate = s.i_rate; L.i_limit = s.i_limit; L.gain = s.gain;
        L.phase = s.phase; L.control = s.control; L.sine = s.sine; L.cosine = s.cosine;
        L.x0 = s.x0; L.x1 = s.x1; L.y0 = s.y0; L.integral = s.integral; L.proportional = s.proportional;
    }

    for (int64_t tile0 = 0; tile0 < n; tile0 += kTile) {
        const int len = (int)min((int64_t)kTile, n - tile0);
        // stream the tile in
        const int rows = shared_in ? 1 : ng;
        for (int r = 0; r < rows; ++r) {
            const int64_t off = (int64_t)(g0 + r) * x_stride + tile0;
            for (int k = lane; k < len; k += 64) {
                in0[r * kPad + k] = x0[off + k];
                if (MODE == kMpsk) in1[r * kPad + k] = x1[off + k];
            }
        }
        __syncthreads();
        if (active) {
            const double *p0 = in0 + (shared_in ? 0 : lane * kPad);
            const double *p1 = in1 + (shared_in ? 0 : lane * kPad);
            double *q0 = out0 + lane * kPad, *q1 = out1 + lane * kPad;
            for (int k = 0; k < len; ++k) {
                if (MODE == kCostas) {
                    const double s = p0[k];
                    nco_update(L, tab);
                    const double i_mixer = s * L.cosine;              // psk.py:177
                    const double q_mixer = s * (-L.sine);             // psk.py:182
                    const double lp = iir_update(L, i_mixer * q_mixer);
                    L.control = pi_update(L, lp);                     // psk.py:187
                    q0[k] = i_mixer;
                } else if (MODE == kPll) {
                    nco_update(L, tab);
                    const double mixer = p0[k] * L.sine;              // afsk_pll.py:156
                    const double lp = iir_update(L, mixer);
                    L.control = pi_update(L, lp);                     // afsk_pll.py:160
                    q0[k] = L.proportional;                           // afsk_pll.py:163
                } else {
                    const double sr = p0[k], si = p1[k];
                    nco_update(L, tab);
                    const double ar = L.cosine, ai = -L.sine;         // nco.py:52-53
                    const double re = (sr * ar) - (si * ai);          // complexmath.py:16
                    const double im = (ar * si) + (sr * ai);          // complexmath.py:17
                    const int e = pd_lookup(pdt, re, im);             // psk.py:739
                    const double lp = iir_update(L, (double)e);
                    L.control = rint(pi_update(L, lp));               // psk.py:740, round() is half-to-even
                    q0[k] = re;
                    q1[k] = im;
                }
            }
        }
        __syncthreads();
        // stream the tile out
        for (int r = 0; r < ng; ++r) {
            const int64_t off = (int64_t)(g0 + r) * out_stride + tile0;
            for (int k = lane; k < len; k += 64) {
                o0[off + k] = out0[r * kPad + k];
                if (MODE == kMpsk) o1[off + k] = out1[r * kPad + k];
            }
        }
    }
    if (active) {
        pm_loop &s = loops[g0 + lane];
        s.phase = L.phase; s.control = L.control; s.sine = L.sine; s.cosine = L.cosine;
        s.x0 = L.x0; s.x1 = L.x1; s.y0 = L.y0; s.integral = L.integral; s.proportional = L.proportional;
    }
}

size_t loop_lds_bytes(int mode, bool shared_in)
{
    const int rows_in = shared_in ? 1 : kG;
    size_t d = 256 + (size_t)rows_in * kPad * (mode == kMpsk ? 2 : 1) + (size_t)kG * kPad * 2;
    return d * 8 + (mode == kMpsk ? 4096 * 4 : 0);
}

template <int MODE>
int loop_launch(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table, const int32_t *d_pd,
                const double *d_x0, const double *d_x1, int64_t x_stride, int64_t n,
                double *d_o0, double *d_o1, int64_t out_stride)
{
    PM_ARG(ctx && h_loops && nloops >= 1 && d_table && n >= 0);
    if (n == 0) return PM_OK;
    PM_ARG(d_x0 && d_o0 && (MODE != kMpsk || (d_x1 && d_o1 && d_pd)));
    PM_ARG(nloops == 1 || out_stride >= n);
    const size_t bytes = sizeof(pm_loop) * (size_t)nloops;
    if (int rc = pm_scratch_reserve(ctx, bytes)) return rc;
    pm_loop *d_loops = (pm_loop *)ctx->d_scratch;
    PM_HIP(hipMemcpyAsync(d_loops, h_loops, bytes, hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = loop_lds_bytes(MODE, x_stride == 0);
    if (lds > 64 * 1024)
        PM_HIP(hipFuncSetAttribute((const void *)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        PmProf prof(ctx, PM_K_LOOP);
        hipLaunchKernelGGL((loop_kernel<MODE>), dim3((unsigned)pm_cdiv(nloops, kG)), dim3(64), lds, ctx->stream,
                           d_loops, nloops, d_table, d_pd, d_x0, d_x1, x_stride, n, d_o0, d_o1, out_stride);
    }
    PM_HIP(hipGetLastError());
    PM_HIP(hipMemcpyAsync(h_loops, d_loops, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

// ---- AGC ---------------------------------------------------------------------------------------
constexpr int kAgcTile = 1024;

__global__ __launch_bounds__(256) void max_partial_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[256];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double m = x[0];                                   // max() is order independent; seed with a real element
    for (; i < n; i += stride) {
        const double v = x[i];
        if (v > m) m = v;                              // Python max(): keeps the first maximum, NaN never wins unless first
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

struct AgcDev {
    double att, dec, sustain_time, sustain_inc, target;
};

// One wave.  Lane 0 runs the envelope follower over the LDS tile (agc.py:26-37); all lanes then scale the
// tile (agc.py:75-76) and stream it back.  state = {envelope, sustain_count}.
__global__ __launch_bounds__(64) void agc_kernel(double *__restrict__ buf, int64_t n, const double *__restrict__ partial, int npartial,
                                                 double scaled_attack, double scaled_decay, AgcDev P, double *__restrict__ state)
{
    __shared__ double xs[kAgcTile];
    __shared__ double es[kAgcTile];
    __shared__ double s_normal;
    const int lane = threadIdx.x;
    if (lane == 0) {
        double m = partial[0];
        for (int i = 1; i < npartial; ++i)
            if (partial[i] > m) m = partial[i];
        s_normal = m;                                   // agc.py:67 normal = max(buffer)
    }
    __syncthreads();
    const double normal = s_normal;
    const double att = scaled_attack * normal;          // agc.py:29 scaled_attack_rate * normal
    const double dec = scaled_decay * normal;           // agc.py:34
    double env = state[0], sustain = state[1];
    for (int64_t tile0 = 0; tile0 < n; tile0 += kAgcTile) {
        const int len = (int)min((int64_t)kAgcTile, n - tile0);
        for (int k = lane; k < len; k += 64) xs[k] = buf[tile0 + k];
        __syncthreads();
        if (lane == 0) {
            for (int k = 0; k < len; ++k) {
                const double cmp = fabs(xs[k]);
                const bool attack = cmp > env;                          // agc.py:28-32
                const double up = fmin(env + att, cmp);                 // env += att; if env > cmp: env = cmp
                env = attack ? up : env;
                sustain = attack ? 0.0 : sustain;
                const bool decay = sustain >= P.sustain_time;           // agc.py:33-36
                const double dn = env - dec;
                env = decay ? (dn < 0 ? 0.0 : dn) : env;
                sustain += P.sustain_inc;                               // agc.py:37
                es[k] = env;
            }
        }
        __syncthreads();
        for (int k = lane; k < len; k += 64) {
            const double e = es[k], s = xs[k];
            buf[tile0 + k] = e != 0 ? P.target * s / e : s;      // agc.py:75-76
        }
        __syncthreads();
    }
    if (lane == 0) {
        state[0] = env;
        state[1] = sustain;
    }
}

}  // namespace

extern "C" {

int pm_costas_bpsk(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table,
                   const double *d_x, int64_t x_stride, int64_t n, double *d_out, int64_t out_stride)
{
    return loop_launch<kCostas>(ctx, h_loops, nloops, d_table, nullptr, d_x, nullptr, x_stride, n, d_out, nullptr, out_stride);
}

int pm_pll_afsk(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table,
                const double *d_x, int64_t x_stride, int64_t n, double *d_out, int64_t out_stride)
{
    return loop_launch<kPll>(ctx, h_loops, nloops, d_table, nullptr, d_x, nullptr, x_stride, n, d_out, nullptr, out_stride);
}

int pm_mpsk_loop(pm_ctx *ctx, pm_loop *h_loops, int nloops, const double *d_table, const int32_t *d_pd_table,
                 const double *d_re, const double *d_im, int64_t x_stride, int64_t n,
                 double *d_i_out, double *d_q_out, int64_t out_stride)
{
    return loop_launch<kMpsk>(ctx, h_loops, nloops, d_table, d_pd_table, d_re, d_im, x_stride, n, d_i_out, d_q_out, out_stride);
}

int pm_agc_apply(pm_ctx *ctx, double *d_buf, int64_t n, const pm_agc_params *hp, double *h_state)
{
    PM_ARG(ctx && hp && h_state && n >= 0);
    if (n == 0) return PM_OK;
    PM_ARG(d_buf != nullptr && hp->sample_rate > 0);
    const int npartial = (int)std::min<int64_t>(1024, pm_cdiv(n, 256));
    if (int rc = pm_scratch_reserve(ctx, (size_t)(npartial + 2) * 8)) return rc;
    double *d_partial = (double *)ctx->d_scratch;
    double *d_state = d_partial + npartial;
    PM_HIP(hipMemcpyAsync(d_state, h_state, 16, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(max_partial_kernel, dim3(npartial), dim3(256), 0, ctx->stream, d_buf, n, d_partial);
    AgcDev P;
    P.sustain_time = hp->sustain_time;
    P.sustain_inc = 1 / hp->sample_rate;                            // agc.py:17
    P.target = hp->target_amplitude;
    P.att = P.dec = 0;
    {
        PmProf prof(ctx, PM_K_AGC);
        hipLaunchKernelGGL(agc_kernel, dim3(1), dim3(64), 0, ctx->stream, d_buf, n, d_partial, npartial,
                           hp->attack_rate / hp->sample_rate, hp->decay_rate / hp->sample_rate, P, d_state);   // agc.py:15-16
    }
    PM_HIP(hipGetLastError());
    PM_HIP(hipMemcpyAsync(h_state, d_state, 16, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

}  // extern "C"
