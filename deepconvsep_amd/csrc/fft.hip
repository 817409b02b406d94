// STFT / iSTFT kernels for gfx950.
//
// Replaces the per-frame Python loops of stft_norm / istft_norm
// (reference transform.py:277-396; script copy examples/dsd100/separate_dsd.py:49-111).
//
// One workgroup (256 threads, 4 wavefronts) transforms one frame.  The N-point real transform is
// done as an N/2-point complex Stockham FFT in LDS (radix-4 passes + one radix-2 pass when
// log2(N/2) is odd; ping-pong between two LDS buffers, natural-order output) followed by the
// usual even/odd split.  Twiddles exp(-2 pi i k / N), k = 0..N/2, are tabulated on the host in
// float64; the kernels read them through the vector L1 (the table is <= 32 KiB and shared by
// every workgroup).
//
// HBM traffic per frame (forward): hop new samples in (neighbouring frames hit L2), 2*(N/2+1)
// values out -- the stage is HBM-bound.
#include "dcs_internal.h"
#include "fft_wave.h"

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;

template <typename R> struct V2;
template <> struct V2<float> { using type = float2; };
template <> struct V2<double> { using type = double2; };

template <typename R2, typename R>
__device__ __forceinline__ R2 mk(R x, R y) {
    R2 r;
    r.x = x;
    r.y = y;
    return r;
}

// w(j) = exp(DIR * 2 pi i j / N) for 0 <= j < N from the half-circle table tw[0..N/2]
// (tw holds the forward sign, exp(-2 pi i j / N)).
template <typename R, typename R2, int DIR>
__device__ __forceinline__ R2 twiddle(const R2* __restrict__ tw, int j, int M) {
    R2 w;
    if (j <= M) {
        w = tw[j];
    } else {
        w = tw[j - M];
        w.x = -w.x;
        w.y = -w.y;
    }
    if (DIR > 0) w.y = -w.y;
    return w;
}

template <typename R2>
__device__ __forceinline__ R2 cmul(R2 a, R2 b) {
    R2 r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}

// In-LDS complex FFT of length M = 2^log2m, DIR = -1 forward / +1 inverse (unscaled).
// Input in `a`; returns the buffer that holds the natural-order result.  Every thread of the
// workgroup must call it; it ends with a barrier.
template <typename R, typename R2, int DIR>
__device__ R2* fft_lds(R2* a, R2* b, const R2* __restrict__ tw, int M, int log2m) {
    const int tid = threadIdx.x;
    int Ns = 1;
    int lg = 0;
    // radix-4 passes
    while (lg + 2 <= log2m) {
        const int quarter = M >> 2;
        const int step = (2 * M) / (4 * Ns);  // N / (4 Ns): table stride per unit of t*k
        for (int j = tid; j < quarter; j += kThreads) {
            const int k = j & (Ns - 1);
            R2 v0 = a[j];
            R2 v1 = a[j + quarter];
            R2 v2 = a[j + 2 * quarter];
            R2 v3 = a[j + 3 * quarter];
            if (Ns > 1) {
                v1 = cmul(v1, twiddle<R, R2, DIR>(tw, k * step, M));
                v2 = cmul(v2, twiddle<R, R2, DIR>(tw, 2 * k * step, M));
                v3 = cmul(v3, twiddle<R, R2, DIR>(tw, 3 * k * step, M));
            }
            R2 a02 = mk<R2, R>(v0.x + v2.x, v0.y + v2.y);
            R2 s02 = mk<R2, R>(v0.x - v2.x, v0.y - v2.y);
            R2 a13 = mk<R2, R>(v1.x + v3.x, v1.y + v3.y);
            R2 s13 = mk<R2, R>(v1.x - v3.x, v1.y - v3.y);
            // forward: y1 = s02 - i s13, y3 = s02 + i s13; inverse: swapped
            R2 ym = mk<R2, R>(s02.x + s13.y, s02.y - s13.x);  // s02 - i*s13
            R2 yp = mk<R2, R>(s02.x - s13.y, s02.y + s13.x);  // s02 + i*s13
            const int d = ((j - k) << 2) + k;                 // (j / Ns) * 4 Ns + k
            b[d] = mk<R2, R>(a02.x + a13.x, a02.y + a13.y);
            b[d + Ns] = (DIR < 0) ? ym : yp;
            b[d + 2 * Ns] = mk<R2, R>(a02.x - a13.x, a02.y - a13.y);
            b[d + 3 * Ns] = (DIR < 0) ? yp : ym;
        }
        __syncthreads();
        R2* t = a;
        a = b;
        b = t;
        Ns <<= 2;
        lg += 2;
    }
    if (lg < log2m) {  // one radix-2 pass
        const int half = M >> 1;
        const int step = (2 * M) / (2 * Ns);
        for (int j = tid; j < half; j += kThreads) {
            const int k = j & (Ns - 1);
            R2 v0 = a[j];
            R2 v1 = a[j + half];
            if (Ns > 1) v1 = cmul(v1, twiddle<R, R2, DIR>(tw, k * step, M));
            const int d = ((j - k) << 1) + k;
            b[d] = mk<R2, R>(v0.x + v1.x, v0.y + v1.y);
            b[d + Ns] = mk<R2, R>(v0.x - v1.x, v0.y - v1.y);
        }
        __syncthreads();
        R2* t = a;
        a = b;
        b = t;
    }
    return a;
}

__device__ __forceinline__ float dcs_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double dcs_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float dcs_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double dcs_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ void dcs_sincos(float a, float* s, float* c) { sincosf(a, s, c); }
__device__ __forceinline__ void dcs_sincos(double a, double* s, double* c) { sincos(a, s, c); }

// ------------------------------------------------------------------------------------------
// forward: frame t = window * padded[t*hop : t*hop+N], padded = [N/2 zeros | audio | zeros]
// (transform.py:309-333); mag = |X| / sqrt(N), phase = angle(X) (transform.py:244-247)
// ------------------------------------------------------------------------------------------
template <typename R, typename R2>
__global__ __launch_bounds__(kThreads) void stft_forward_kernel(
    const R* __restrict__ audio, int64_t L, const R* __restrict__ win, const R2* __restrict__ tw,
    R* __restrict__ mag, R* __restrict__ phase, R2* __restrict__ unit, int64_t ld, int N, int hop, int log2m,
    int64_t T, R sqrt_n, int tw_lds, int64_t rows_pc, int64_t audio_stride, int64_t il_clips) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = N >> 1;
    const int tid = threadIdx.x;
    // output row -> (clip, frame): clips of equal length are stacked with a pitch of rows_pc rows
    // (il_clips > 0: the clips are the channels of one signal and their rows are interleaved, [frame][channel])
    const int64_t row = blockIdx.x;
    const int64_t clip = row / rows_pc;
    const int64_t t = row - clip * rows_pc;
    audio += clip * audio_stride;
    const int64_t orow = il_clips > 0 ? t * il_clips + clip : row;
    R* mrow = mag + orow * ld;
    R* prow = phase ? phase + orow * ld : nullptr;
    R2* urow = unit ? unit + orow * ld : nullptr;
    if (t >= T) {  // rows past the last frame: zeros (zero-padding tiler, util.py:233)
        for (int k = tid; k < ld; k += kThreads) {
            mrow[k] = R(0);
            if (prow) prow[k] = R(0);
            if (urow) urow[k] = mk<R2, R>(R(1), R(0));
        }
        return;
    }
    R2* buf0 = reinterpret_cast<R2*>(smem);
    R2* buf1 = buf0 + M;
    // Twiddle table staged in LDS together with the frame: one exposed memory latency instead of one
    // dependent L1/L2 round trip per FFT pass (measured: ~10 us of workgroup time per 1024-point FFT
    // with the table in global memory).
    if (tw_lds) {
        R2* twl = buf1 + M;
        for (int k = tid; k <= M; k += kThreads) twl[k] = tw[k];
        tw = twl;
    }
    const int64_t base = t * (int64_t)hop - M;  // audio index of padded sample t*hop
    for (int m = tid; m < M; m += kThreads) {
        const int64_t p = base + 2 * m;
        R x0 = R(0), x1 = R(0);
        if (p >= 0 && p < L) x0 = audio[p] * win[2 * m];
        if (p + 1 >= 0 && p + 1 < L) x1 = audio[p + 1] * win[2 * m + 1];
        buf0[m] = mk<R2, R>(x0, x1);
    }
    __syncthreads();
    const R2* Z = fft_lds<R, R2, -1>(buf0, buf1, tw, M, log2m);
    for (int k = tid; k <= M; k += kThreads) {
        const R2 zk = Z[k & (M - 1)];
        const R2 zm = Z[(M - k) & (M - 1)];
        // E = (zk + conj(zm))/2 ; O = -i (zk - conj(zm))/2 ; X = E + w^k O
        const R er = R(0.5) * (zk.x + zm.x), ei = R(0.5) * (zk.y - zm.y);
        const R orr = R(0.5) * (zk.y + zm.y), oi = R(-0.5) * (zk.x - zm.x);
        const R2 w = tw[k];
        const R xr = er + (w.x * orr - w.y * oi);
        const R xi = ei + (w.x * oi + w.y * orr);
        const R ax = dcs_sqrt(xr * xr + xi * xi);
        mrow[k] = ax / sqrt_n;
        if (prow) prow[k] = dcs_atan2(xi, xr);
        // exp(j*angle(X)) without the angle: X/|X|, and 1 where X == 0 (np.angle(0) = 0)
        if (urow) urow[k] = (ax > R(0)) ? mk<R2, R>(xr / ax, xi / ax) : mk<R2, R>(R(1), R(0));
    }
    for (int k = M + 1 + tid; k < ld; k += kThreads) {  // row padding
        mrow[k] = R(0);
        if (prow) prow[k] = R(0);
        if (urow) urow[k] = mk<R2, R>(R(1), R(0));
    }
}

// ------------------------------------------------------------------------------------------
// inverse (compute_inverse, transform.py:271-273 + istft_norm, transform.py:337-396), one kernel:
//   X = (mag / pre_div) * sqrt(N) * exp(j phase)   (separate_dsd.py:304: pre_div = scale_factor)
//   frame = irfft(X)[:N]  (imaginary parts of DC / Nyquist ignored, as numpy does)
//   data[n*hop : n*hop+N] += window * frame ; norm += window * window ; drop N/2 ; data / norm
// A workgroup owns C consecutive hops of ONE source's output and walks the C + ceil(N/hop) - 1
// frames that overlap it in increasing n (the reference's accumulation order), adding each
// windowed frame into an LDS accumulator -- the [src][T][N] time-frame scratch of a two-pass
// scheme (2*N*4 bytes per frame per source through HBM) never exists.  exp(j phase) comes either
// from the angle (API path, sincos) or from the unit phasor the forward kernel stored (fused path).
// ------------------------------------------------------------------------------------------
// PF = bins per thread ( >= ceil((N/2+1)/256) ): the next frame's rows are prefetched into registers
// while the current frame is transformed.
template <typename R, typename R2, bool UNIT, int PF>
__global__ __launch_bounds__(kThreads) void istft_fused_kernel(
    const R* __restrict__ mag, int64_t src_stride, const R* __restrict__ phase, const R2* __restrict__ unit,
    int64_t ld, const R* __restrict__ win, const R* __restrict__ wsq, const R2* __restrict__ tw,
    R* __restrict__ audio, int64_t n_out, int N, int hop, int log2m, int64_t T, int C, R pre_div, R sqrt_n,
    int tw_lds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = N >> 1;
    const int tid = threadIdx.x;
    const int s = blockIdx.y;
    R2* buf0 = reinterpret_cast<R2*>(smem);
    R2* buf1 = buf0 + M;  // M+1 entries: holds X first, then serves as the FFT ping-pong buffer
    R2* X = buf1;
    R2* twl = buf1 + (M + 1);
    R* acc = reinterpret_cast<R*>(tw_lds ? twl + (M + 1) : twl);
    if (tw_lds) {
        for (int k = tid; k <= M; k += kThreads) twl[k] = tw[k];
        tw = twl;
    }
    const int span = C * hop;
    const int64_t p0 = (int64_t)blockIdx.x * span;  // first padded position of this chunk
    for (int q = tid; q < span; q += kThreads) acc[q] = R(0);

    int64_t n_hi = (p0 + span - 1) / hop;
    if (n_hi > T - 1) n_hi = T - 1;
    const int64_t n_lo = (p0 < N) ? 0 : (p0 - N) / hop + 1;
    const R inv_m = R(1) / R(M);
    const R2* w2 = reinterpret_cast<const R2*>(win);
    const R* msrc = mag + (int64_t)s * src_stride;

    R pm[PF];   // magnitudes of the frame about to be transformed
    R2 pu[PF];  // unit phasors (UNIT) or {phase, -} (!UNIT)
#define DCS_ROWS_LOAD(n_)                                                          \
    _Pragma("unroll") for (int u = 0; u < PF; ++u) {                               \
        const int k = tid + u * kThreads;                                          \
        if (k <= M) {                                                              \
            pm[u] = msrc[(n_) * ld + k];                                           \
            if (UNIT) pu[u] = unit[(n_) * ld + k];                                 \
            else pu[u].x = phase[(n_) * ld + k];                                   \
        }                                                                          \
    }
    if (n_lo <= n_hi) DCS_ROWS_LOAD(n_lo)
    for (int64_t n = n_lo; n <= n_hi; ++n) {
        __syncthreads();  // previous frame: accumulator updates done, FFT buffers free
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int k = tid + u * kThreads;
            if (k <= M) {
                const R a = (pm[u] / pre_div) * sqrt_n;
                R2 x;
                if (UNIT) {
                    x = mk<R2, R>(a * pu[u].x, a * pu[u].y);
                } else {
                    R sn, cs;
                    dcs_sincos(pu[u].x, &sn, &cs);
                    x = mk<R2, R>(a * cs, a * sn);
                }
                if (k == 0 || k == M) x.y = R(0);
                X[k] = x;
            }
        }
        if (n + 1 <= n_hi) DCS_ROWS_LOAD(n + 1)  // in flight during the transform of frame n
        __syncthreads();
        for (int k = tid; k < M; k += kThreads) {
            const R2 xk = X[k];
            const R2 xm = X[M - k];
            // E = (xk + conj(xm))/2 ; D = (xk - conj(xm))/2 ; O = D * conj(w^k) ; Z = E + i O
            const R er = R(0.5) * (xk.x + xm.x), ei = R(0.5) * (xk.y - xm.y);
            const R dr = R(0.5) * (xk.x - xm.x), di = R(0.5) * (xk.y + xm.y);
            const R2 w = tw[k];
            const R orr = dr * w.x + di * w.y;
            const R oi = di * w.x - dr * w.y;
            buf0[k] = mk<R2, R>(er - oi, ei + orr);
        }
        __syncthreads();
        const R2* z = fft_lds<R, R2, +1>(buf0, buf1, tw, M, log2m);
        const int64_t off = n * (int64_t)hop - p0;  // chunk-relative position of sample 0 of the frame
        for (int m = tid; m < M; m += kThreads) {
            const R2 v = z[m];
            const R2 w = w2[m];
            const int64_t q = off + 2 * m;
            if (q >= 0 && q < span) acc[q] += (v.x * inv_m) * w.x;
            if (q + 1 >= 0 && q + 1 < span) acc[q + 1] += (v.y * inv_m) * w.y;
        }
    }
    __syncthreads();
    const int half = N >> 1;
    for (int q = tid; q < span; q += kThreads) {
        const int64_t p = p0 + q;
        const int64_t m = p - half;
        if (m < 0 || m >= n_out) continue;
        int64_t f_hi = p / hop;
        if (f_hi > T - 1) f_hi = T - 1;
        const int64_t f_lo = (p < N) ? 0 : (p - N) / hop + 1;
        R norm = R(0);
        for (int64_t n = f_lo; n <= f_hi; ++n) norm += wsq[p - n * hop];
        if (norm == R(0)) norm = R(1);
        audio[(int64_t)s * n_out + m] = acc[q] / norm;
    }
#undef DCS_ROWS_LOAD
}

template <typename R, typename R2>
int launch_forward(dcs_stft* p, const R* win, const R2* tw, const R* audio, int64_t L, R* mag, R* phase, R2* unit,
                   int64_t ld, int64_t rows_out, int64_t T, int64_t n_clips = 1, int64_t audio_stride = 0,
                   bool interleave = false) {
    if (rows_out <= 0 || n_clips <= 0) return DCS_OK;
    const int M = p->frame / 2;
    size_t lds = (3 * (size_t)M + 1) * sizeof(R2);
    const int tw_lds = lds <= 64 * 1024;
    if (!tw_lds) lds = 2 * (size_t)M * sizeof(R2);
    auto kern = stft_forward_kernel<R, R2>;
    if (lds > 48 * 1024)
        DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    DcsTimer tm(p->ctx, DCS_TAG_STFT);
    hipLaunchKernelGGL(kern, dim3((unsigned)(rows_out * n_clips)), dim3(kThreads), lds, p->ctx->stream, audio, L, win, tw,
                       mag, phase, unit, ld, p->frame, p->hop, p->log2m, T, (R)sqrt((double)p->frame), tw_lds, rows_out,
                       audio_stride, interleave ? n_clips : (int64_t)0);
    tm.done();
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

template <typename R, typename R2, bool UNIT, int PF>
int launch_inverse_pf(dcs_stft* p, const dim3& grid, size_t lds, const R* win, const R2* tw, const R* wsq, const R* mag,
                      int64_t src_stride, const R* phase, const R2* unit, int64_t ld, int64_t T, R pre_div, R* audio,
                      int64_t n_out, int C, int tw_lds) {
    auto kern = istft_fused_kernel<R, R2, UNIT, PF>;
    if (lds > 48 * 1024)
        DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, p->ctx->stream, mag, src_stride, phase, unit, ld, win, wsq, tw,
                       audio, n_out, p->frame, p->hop, p->log2m, T, C, pre_div, (R)sqrt((double)p->frame), tw_lds);
    return DCS_OK;
}

template <typename R, typename R2, bool UNIT>
int launch_inverse_unit(dcs_stft* p, const dim3& grid, size_t lds, const R* win, const R2* tw, const R* wsq,
                        const R* mag, int64_t src_stride, const R* phase, const R2* unit, int64_t ld, int64_t T,
                        R pre_div, R* audio, int64_t n_out, int C, int tw_lds) {
    const int need = (p->frame / 2 + 1 + kThreads - 1) / kThreads;
#define DCS_PF(PF_)                                                                                                  \
    return launch_inverse_pf<R, R2, UNIT, PF_>(p, grid, lds, win, tw, wsq, mag, src_stride, phase, unit, ld, T, pre_div, \
                                               audio, n_out, C, tw_lds)
    if (need <= 3) DCS_PF(3);
    if (need <= 5) DCS_PF(5);
    if (need <= 9) DCS_PF(9);
    DCS_PF(17);
#undef DCS_PF
}

template <typename R, typename R2>
int launch_inverse(dcs_stft* p, const R* win, const R2* tw, const R* wsq, const R* mag, int64_t src_stride,
                   const R* phase, const R2* unit, int64_t ld, int64_t T, int n_src, R pre_div, R* audio,
                   int64_t n_out) {
    if (T <= 0 || n_src <= 0 || n_out <= 0) return DCS_OK;
    const int N = p->frame, hop = p->hop;
    const int M = N / 2;
    const int64_t hops = (n_out + N / 2 + hop - 1) / hop;  // padded positions [0, n_out + N/2)
    // hops per workgroup: enough workgroups to fill the chip first, then fewer redundant halo FFTs
    const int R_ = (N + hop - 1) / hop;
    int64_t C = hops * n_src / (int64_t)p->ctx->n_cu;
    if (C > 4 * R_) C = 4 * R_;
    if (C < 1) C = 1;
    const size_t fixed_tw = (3 * (size_t)M + 2) * sizeof(R2), fixed_notw = (2 * (size_t)M + 1) * sizeof(R2);
    const int tw_lds = fixed_tw + (size_t)hop * sizeof(R) <= 64 * 1024;
    const size_t fixed = tw_lds ? fixed_tw : fixed_notw;
    size_t lds = fixed + (size_t)C * hop * sizeof(R);
    // LDS per workgroup decides how many workgroups share a CU (their barriers overlap); 48 KiB = 3 per CU
    constexpr size_t lds_cap = 48 * 1024;
    while (lds > lds_cap && C > 1) {
        C = C / 2;
        lds = fixed + (size_t)C * hop * sizeof(R);
    }
    if (lds > 160 * 1024) DCS_FAIL(DCS_EUNSUPPORTED, "istft: frame %d needs %zu bytes of LDS", N, lds);
    const dim3 grid((unsigned)((hops + C - 1) / C), (unsigned)n_src);
    DcsTimer tm(p->ctx, DCS_TAG_ISTFT);
    int rc;
    if (unit)
        rc = launch_inverse_unit<R, R2, true>(p, grid, lds, win, tw, wsq, mag, src_stride, phase, unit, ld, T, pre_div,
                                              audio, n_out, (int)C, tw_lds);
    else
        rc = launch_inverse_unit<R, R2, false>(p, grid, lds, win, tw, wsq, mag, src_stride, phase, unit, ld, T, pre_div,
                                               audio, n_out, (int)C, tw_lds);
    tm.done();
    DCS_CHECK(rc);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

}  // namespace

int dcs_launch_stft_forward_f32_clips(dcs_stft* p, const float* audio, int64_t L, int64_t audio_stride, int64_t n_clips,
                                      float* mag, float* phase, float2* unit, int64_t ld, int64_t rows_out, int64_t T,
                                      bool interleave, const int64_t* clip_tab) {
    // few frames: the block-level kernel (4 waves share one frame's FFT) has the shorter critical path
    // (8.6 us vs 19 us for 186 frames); many frames: the wave-per-frame kernel has the higher throughput
    static const int64_t thr_env = getenv("DCS_STFT_WAVE_MIN") ? atoll(getenv("DCS_STFT_WAVE_MIN")) : -1;
    const int64_t thr = thr_env >= 0 ? thr_env : 4 * (int64_t)p->ctx->n_cu;
    if (clip_tab && !dcs_fft_wave_supported(p))
        DCS_FAIL(DCS_EUNSUPPORTED, "clips of different lengths need the wave STFT kernels (frameSize 1024 / 2048 / 4096)");
    if (dcs_fft_wave_supported(p) && (clip_tab || rows_out * n_clips >= thr)) {
        DcsTimer tm(p->ctx, DCS_TAG_STFT);
        const int rc = dcs_fft_wave_forward(p, audio, L, audio_stride, n_clips, mag, phase, unit, ld, rows_out, T,
                                            interleave, clip_tab);
        tm.done();
        DCS_CHECK(rc);
        DCS_HIP(hipGetLastError());
        return DCS_OK;
    }
    return launch_forward<float, float2>(p, p->win_f, p->tw_f, audio, L, mag, phase, unit, ld, rows_out, T, n_clips,
                                         audio_stride, interleave);
}
int dcs_launch_stft_forward_f32(dcs_stft* p, const float* audio, int64_t L, float* mag, float* phase, float2* unit,
                                int64_t ld, int64_t rows_out, int64_t T) {
    return dcs_launch_stft_forward_f32_clips(p, audio, L, 0, 1, mag, phase, unit, ld, rows_out, T, false);
}
int dcs_launch_stft_forward_f64(dcs_stft* p, const double* audio, int64_t L, double* mag, double* phase,
                                double2* unit, int64_t ld, int64_t rows_out, int64_t T) {
    return launch_forward<double, double2>(p, p->win_d, p->tw_d, audio, L, mag, phase, unit, ld, rows_out, T);
}
int dcs_launch_stft_forward_f64_clips(dcs_stft* p, const double* audio, int64_t L, int64_t audio_stride, int64_t n_clips,
                                      double* mag, double* phase, int64_t ld, int64_t rows_out, int64_t T) {
    return launch_forward<double, double2>(p, p->win_d, p->tw_d, audio, L, mag, phase, nullptr, ld, rows_out, T, n_clips,
                                           audio_stride, false);
}
int dcs_launch_stft_inverse_f32(dcs_stft* p, const float* mag, int64_t src_stride, const float* phase,
                                const float2* unit, int64_t ld, int64_t T, int n_src, float pre_div, float* audio,
                                int64_t n_out) {
    if (dcs_fft_wave_inverse_supported(p)) {
        if (T <= 0 || n_src <= 0 || n_out <= 0) return DCS_OK;
        DcsTimer tm(p->ctx, DCS_TAG_ISTFT);
        const int rc = dcs_fft_wave_inverse(p, mag, src_stride, phase, unit, 0, n_src, ld, T, n_src, pre_div, audio, n_out);
        tm.done();
        DCS_CHECK(rc);
        DCS_HIP(hipGetLastError());
        return DCS_OK;
    }
    return launch_inverse<float, float2>(p, p->win_f, p->tw_f, p->wsq_f, mag, src_stride, phase, unit, ld, T, n_src,
                                         pre_div, audio, n_out);
}
// n_clips clips of n_src sources each: sources are stacked (clip-major), the unit phasors are per clip
int dcs_launch_stft_inverse_f32_clips(dcs_stft* p, const float* mag, int64_t src_stride, const float2* unit,
                                      int64_t unit_clip_stride, int64_t ld, int64_t T, int n_src, int64_t n_clips,
                                      float pre_div, float* audio, int64_t n_out, const int64_t* clip_tab,
                                      int64_t out_stride) {
    if (T <= 0 || n_src <= 0 || n_out <= 0 || n_clips <= 0) return DCS_OK;
    if (clip_tab && (!dcs_fft_wave_inverse_supported(p) || !unit))
        DCS_FAIL(DCS_EUNSUPPORTED, "clips of different lengths need the wave iSTFT kernel (frameSize 1024 / 2048 / 4096, hop | frameSize)");
    if (!dcs_fft_wave_inverse_supported(p) || !unit) {
        for (int64_t c = 0; c < n_clips; ++c)
            DCS_CHECK(dcs_launch_stft_inverse_f32(p, mag + c * n_src * src_stride, src_stride, nullptr,
                                                  unit ? unit + c * unit_clip_stride : nullptr, ld, T, n_src, pre_div,
                                                  audio + c * n_src * n_out, n_out));
        return DCS_OK;
    }
    DcsTimer tm(p->ctx, DCS_TAG_ISTFT);
    const int rc = dcs_fft_wave_inverse(p, mag, src_stride, nullptr, unit, unit_clip_stride, n_src, ld, T,
                                        (int)(n_src * n_clips), pre_div, audio, n_out, clip_tab, out_stride);
    tm.done();
    DCS_CHECK(rc);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}
int dcs_launch_stft_inverse_f64(dcs_stft* p, const double* mag, int64_t src_stride, const double* phase,
                                const double2* unit, int64_t ld, int64_t T, int n_src, double pre_div, double* audio,
                                int64_t n_out) {
    return launch_inverse<double, double2>(p, p->win_d, p->tw_d, p->wsq_d, mag, src_stride, phase, unit, ld, T, n_src,
                                           pre_div, audio, n_out);
}
