// Wave-per-frame STFT / iSTFT for the float32 fast path (N = 1024, 2048, 4096), gfx950.
//
// The block-level radix-4 FFT of fft.hip spends ~60 % of a wave's life parked at workgroup barriers
// (PMC: SQ_WAIT_ANY, profiles/r01_b_pmc_summary.txt): 5 passes x 256 threads x 1 butterfly.  Here one
// 64-lane wavefront owns one frame: M = N/2 complex points, M/64 = 8/16/32 points per lane, three
// Stockham passes of radix 16/16/4 (M=1024), 16/16/8 (M=2048), 8/8/8 (M=512) done in registers and
// exchanged through a per-wave LDS buffer.  A wave's LDS operations are performed in program order, so
// no barrier is needed between passes; a 4-wave workgroup transforms 4 frames independently and only
// meets once to stage the shared twiddle table (and, in the inverse, to hand the finished frames to
// the overlap-add).  The exchange buffer is indexed through pad(i) = i + i/32 so that the stride-R
// stores of the first pass spread over the banks.
//
// Semantics are those of fft.hip (reference transform.py:277-396); the host side picks this file's
// kernels when the plan is float32 with N in {1024, 2048, 4096} and the block-level ones otherwise.
#include "dcs_internal.h"
#include "fft_wave.h"

#include <math.h>
#include <stdlib.h>

namespace {

__device__ __forceinline__ int pad(int i) { return i + (i >> 5); }

__device__ __forceinline__ float2 mk2(float x, float y) {
    float2 r;
    r.x = x;
    r.y = y;
    return r;
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return mk2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return mk2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return mk2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// exp(DIR * 2 pi i j / N), 0 <= j < N, from the half-circle table tw[0..M] = exp(-2 pi i j / N), N = 2M
template <int DIR>
__device__ __forceinline__ float2 tw_at(const float2* tw, int j, int M) {
    float2 w;
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

template <int DIR>
__device__ __forceinline__ void dft4(float2& a, float2& b, float2& c, float2& d) {
    const float2 a02 = cadd(a, c), s02 = csub(a, c), a13 = cadd(b, d), s13 = csub(b, d);
    const float2 ym = mk2(s02.x + s13.y, s02.y - s13.x);  // s02 - i*s13
    const float2 yp = mk2(s02.x - s13.y, s02.y + s13.x);  // s02 + i*s13
    a = cadd(a02, a13);
    c = csub(a02, a13);
    b = (DIR < 0) ? ym : yp;
    d = (DIR < 0) ? yp : ym;
}

// multiply by exp(DIR * 2 pi i m / 16)
template <int DIR, int m>
__device__ __forceinline__ float2 rot16(float2 v) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    constexpr int mm = m & 15;
    float cx, sx;  // exp(-2 pi i mm/16) = cx - i*sx ... expressed as (c, s) with s = sin(2 pi mm / 16)
    if (mm == 0) return v;
    if (mm == 1) { cx = C1; sx = S1; }
    else if (mm == 2) { cx = H; sx = H; }
    else if (mm == 3) { cx = S1; sx = C1; }
    else if (mm == 4) { cx = 0.f; sx = 1.f; }
    else if (mm == 6) { cx = -H; sx = H; }
    else if (mm == 9) { cx = -C1; sx = -S1; }
    else { cx = 1.f; sx = 0.f; }
    // forward: multiply by (cx, -sx); inverse: (cx, +sx)
    const float wy = (DIR < 0) ? -sx : sx;
    return mk2(v.x * cx - v.y * wy, v.x * wy + v.y * cx);
}

// In-register DFTs.  Input v[n], n = 0..R-1; output X[k] is left at v[perm(k)].
template <int DIR>
__device__ __forceinline__ void dft16(float2* v) {
    // n = 4a + b: DFT over a for each b
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // v[4c + b] = y[b][c]; twiddle w16^(b c)
    v[5] = rot16<DIR, 1>(v[5]);
    v[6] = rot16<DIR, 2>(v[6]);
    v[7] = rot16<DIR, 3>(v[7]);
    v[9] = rot16<DIR, 2>(v[9]);
    v[10] = rot16<DIR, 4>(v[10]);
    v[11] = rot16<DIR, 6>(v[11]);
    v[13] = rot16<DIR, 3>(v[13]);
    v[14] = rot16<DIR, 6>(v[14]);
    v[15] = rot16<DIR, 9>(v[15]);
    // DFT over b for each c: v[4c + d] = X[c + 4d]
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4<DIR>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}
__device__ __forceinline__ constexpr int perm16(int k) { return 4 * (k & 3) + (k >> 2); }

template <int DIR>
__device__ __forceinline__ void dft8(float2* v) {
    // n = 2a + b (a = 0..3, b = 0..1): DFT over a for each b
    dft4<DIR>(v[0], v[2], v[4], v[6]);
    dft4<DIR>(v[1], v[3], v[5], v[7]);
    // v[2c + b] = y[b][c]; twiddle w8^(b c) = w16^(2 b c)
    v[3] = rot16<DIR, 2>(v[3]);
    v[5] = rot16<DIR, 4>(v[5]);
    v[7] = rot16<DIR, 6>(v[7]);
    // 2-point DFT over b: v[2c + d] = X[c + 4d]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float2 t = v[2 * c];
        v[2 * c] = cadd(t, v[2 * c + 1]);
        v[2 * c + 1] = csub(t, v[2 * c + 1]);
    }
}
__device__ __forceinline__ constexpr int perm8(int k) { return 2 * (k & 3) + (k >> 2); }

template <int R, int DIR>
__device__ __forceinline__ void dftR(float2* v) {
    if (R == 16) dft16<DIR>(v);
    else if (R == 8) dft8<DIR>(v);
    else dft4<DIR>(v[0], v[1], v[2], v[3]);
}
template <int R>
__device__ __forceinline__ constexpr int permR(int k) { return R == 16 ? perm16(k) : (R == 8 ? perm8(k) : k); }

// One Stockham pass of radix R over the wave's M = 64*P points.  v holds the pass input
// v[b*R + t] = x[j_b + t*M/R], j_b = lane + 64 b; the output goes to buf (padded, wave-private):
// y[(j/Ns)*Ns*R + (j%Ns) + t*Ns].
template <int R, int P, int DIR>
__device__ __forceinline__ void fft_pass(float2 (&v)[P], int Ns, int lane, const float2* twl, float2* buf) {
    constexpr int NB = P / R;
    constexpr int M = 64 * P;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = lane + 64 * b;
        const int k = j & (Ns - 1);
        if (Ns > 1) {
            const int step = ((2 * M) / R / Ns) * k;
#pragma unroll
            for (int t = 1; t < R; ++t) v[b * R + t] = cmul(v[b * R + t], tw_at<DIR>(twl, t * step, M));
        }
        dftR<R, DIR>(&v[b * R]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = lane + 64 * b;
        const int k = j & (Ns - 1);
        const int d = (j - k) * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) buf[pad(d + t * Ns)] = v[b * R + permR<R>(t)];
    }
}

template <int R, int P>
__device__ __forceinline__ void load_pass(float2 (&v)[P], int lane, const float2* buf) {
    constexpr int NB = P / R;
    constexpr int stride = 64 * NB;  // M / R
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < R; ++t) v[b * R + t] = buf[pad(lane + 64 * b + t * stride)];
}

template <int LOG2M> struct Plan;
template <> struct Plan<9> { static constexpr int R1 = 8, R2 = 8, R3 = 8; };
template <> struct Plan<10> { static constexpr int R1 = 16, R2 = 16, R3 = 4; };
template <> struct Plan<11> { static constexpr int R1 = 16, R2 = 16, R3 = 8; };

// three passes; on entry v holds the first-pass input (index map of load_pass<R1>), on exit the
// natural-order result is in buf[pad(k)]
template <int LOG2M, int DIR>
__device__ __forceinline__ void fft_wave(float2 (&v)[(1 << LOG2M) / 64], int lane, const float2* twl, float2* buf) {
    constexpr int P = (1 << LOG2M) / 64;
    using PL = Plan<LOG2M>;
    fft_pass<PL::R1, P, DIR>(v, 1, lane, twl, buf);
    load_pass<PL::R2, P>(v, lane, buf);
    fft_pass<PL::R2, P, DIR>(v, PL::R1, lane, twl, buf);
    load_pass<PL::R3, P>(v, lane, buf);
    fft_pass<PL::R3, P, DIR>(v, PL::R1 * PL::R2, lane, twl, buf);
}

// ------------------------------------------------------------------------------------------------
// forward (compute_file): FPW frames per workgroup, one per wave
// ------------------------------------------------------------------------------------------------
template <int LOG2M>
__global__ void stft_forward_wave_kernel(const float* __restrict__ audio, int64_t L, const float* __restrict__ win,
                                         const float2* __restrict__ tw, float* __restrict__ mag,
                                         float* __restrict__ phase, float2* __restrict__ unit, int64_t ld, int hop,
                                         int64_t T, int64_t rows_out, float sqrt_n, int dbg) {
    constexpr int M = 1 << LOG2M, P = M / 64, MP = M + M / 32;
    constexpr int R1 = Plan<LOG2M>::R1, NB1 = P / R1, stride1 = 64 * NB1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* twl = reinterpret_cast<float2*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* buf = twl + (M + 1) + wave * MP;
    if (!(dbg & 4))
        for (int k = tid; k <= M; k += blockDim.x) twl[k] = tw[k];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (t >= rows_out) return;
    float* mrow = mag + t * ld;
    float* prow = phase ? phase + t * ld : nullptr;
    float2* urow = unit ? unit + t * ld : nullptr;
    if (t >= T) {
        for (int k = lane; k < ld; k += 64) {
            mrow[k] = 0.f;
            if (prow) prow[k] = 0.f;
            if (urow) urow[k] = mk2(1.f, 0.f);
        }
        return;
    }
    float2 v[P];
    const int64_t base = t * (int64_t)hop - M;
    // 32-bit window-relative bounds instead of two 64-bit compares per sample
    const int r_lo = base < 0 ? (int)(-base) : 0;
    const int64_t rem = L - base;
    const int r_hi = rem < 2 * M ? (rem < 0 ? 0 : (int)rem) : 2 * M;
    const float* ap = audio + base;
    const float inv_sqrt_n = 1.f / sqrt_n;
    const float2* w2 = reinterpret_cast<const float2*>(win);
#pragma unroll
    for (int b = 0; b < NB1; ++b)
#pragma unroll
        for (int tt = 0; tt < R1; ++tt) {
            const int i = lane + 64 * b + tt * stride1;
            const int r = 2 * i;
            const float2 w = (dbg & 8) ? mk2(0.5f, 0.25f) : w2[i];
            float x0 = 0.f, x1 = 0.f;
            if (dbg & 2) {
                x0 = (float)i * w.x;
                x1 = (float)(i + 1) * w.y;
            } else {
                if (r >= r_lo && r < r_hi) x0 = ap[r] * w.x;
                if (r + 1 >= r_lo && r + 1 < r_hi) x1 = ap[r + 1] * w.y;
            }
            v[b * R1 + tt] = mk2(x0, x1);
        }
    fft_wave<LOG2M, -1>(v, lane, twl, buf);
    // even/odd split: X[k] = E + w^k O with E = (Z[k] + conj Z[M-k])/2, O = -i (Z[k] - conj Z[M-k])/2
    for (int u = 0; u <= P; ++u) {
        const int k = lane + 64 * u;
        if (k > M) break;
        const float2 zk = buf[pad(k & (M - 1))];
        const float2 zm = buf[pad((M - k) & (M - 1))];
        const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);
        const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);
        const float2 w = twl[k];
        const float xr = er + (w.x * orr - w.y * oi);
        const float xi = ei + (w.x * oi + w.y * orr);
        const float ax = sqrtf(xr * xr + xi * xi);
        if ((dbg & 1) && !(ax == 12345.678f)) continue;  // keeps the arithmetic, drops the stores
        mrow[k] = ax * inv_sqrt_n;
        if (prow) prow[k] = atan2f(xi, xr);
        if (urow) {
            const float ra = 1.f / ax;  // one division for both components
            urow[k] = (ax > 0.f) ? mk2(xr * ra, xi * ra) : mk2(1.f, 0.f);
        }
    }
    for (int k = M + 1 + lane; k < ld; k += 64) {
        mrow[k] = 0.f;
        if (prow) prow[k] = 0.f;
        if (urow) urow[k] = mk2(1.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// inverse (compute_inverse): workgroup = C hops of one source; its 4 waves transform 4 consecutive
// frames at a time, then all threads add the 4 windowed frames into the chunk in frame order.
// ------------------------------------------------------------------------------------------------
template <int LOG2M, bool UNIT>
__global__ __launch_bounds__(256) void istft_wave_kernel(const float* __restrict__ mag, int64_t src_stride,
                                                         const float* __restrict__ phase,
                                                         const float2* __restrict__ unit, int64_t ld,
                                                         const float* __restrict__ win, const float* __restrict__ wsq,
                                                         const float2* __restrict__ tw, float* __restrict__ audio,
                                                         int64_t n_out, int hop, int64_t T, int C, float pre_div,
                                                         float sqrt_n) {
    constexpr int M = 1 << LOG2M, N = 2 * M, P = M / 64, MP = M + M / 32;
    constexpr int R1 = Plan<LOG2M>::R1, NB1 = P / R1, stride1 = 64 * NB1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* twl = reinterpret_cast<float2*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* fbuf = twl + (M + 1);
    float2* buf = fbuf + wave * MP;
    float* acc = reinterpret_cast<float*>(fbuf + 4 * MP);
    const int s = blockIdx.y;
    const int span = C * hop;
    const int64_t p0 = (int64_t)blockIdx.x * span;
    // hop | N and hop even (every reference config): a frame is N/hop whole hop-blocks of hop/2 sample pairs,
    // so which block of which frame lands where is wave-uniform scalar arithmetic
    const bool aligned = (N % hop) == 0 && (hop & 1) == 0;
    float2* winl = reinterpret_cast<float2*>(acc + span);  // window as (even, odd) pairs (aligned path)
    if (aligned) {
        const float2* w2g = reinterpret_cast<const float2*>(win);
        for (int k = tid; k < M; k += 256) winl[k] = w2g[k];
    }
    for (int k = tid; k <= M; k += 256) twl[k] = tw[k];
    for (int q = tid; q < span; q += 256) acc[q] = 0.f;
    int64_t n_hi = (p0 + span - 1) / hop;
    if (n_hi > T - 1) n_hi = T - 1;
    const int64_t n_lo = (p0 < N) ? 0 : (p0 - N) / hop + 1;
    const float inv_m = 1.f / (float)M;
    const float amp = sqrt_n / pre_div;  // (mag / scale_factor) * sqrt(N) with one multiply per bin
    const float* msrc = mag + (int64_t)s * src_stride;
    __syncthreads();

    for (int64_t nb = n_lo; nb <= n_hi; nb += 4) {
        const int64_t n = nb + wave;
        if (n <= n_hi) {
            const float* mrow = msrc + n * ld;
            float2 v[P];
#pragma unroll
            for (int b = 0; b < NB1; ++b)
#pragma unroll
                for (int tt = 0; tt < R1; ++tt) {
                    const int k = lane + 64 * b + tt * stride1;  // 0 <= k < M
                    const int km = M - k;                         // 1..M
                    float2 xk, xm;
                    {
                        const float a = mrow[k] * amp;
                        const float b2 = mrow[km] * amp;
                        if (UNIT) {
                            const float2 uk = unit[n * ld + k], um = unit[n * ld + km];
                            xk = mk2(a * uk.x, a * uk.y);
                            xm = mk2(b2 * um.x, b2 * um.y);
                        } else {
                            float sn, cs;
                            sincosf(phase[n * ld + k], &sn, &cs);
                            xk = mk2(a * cs, a * sn);
                            sincosf(phase[n * ld + km], &sn, &cs);
                            xm = mk2(b2 * cs, b2 * sn);
                        }
                    }
                    if (k == 0) xk.y = 0.f;      // imaginary parts of DC / Nyquist are ignored (numpy irfft)
                    if (km == M) xm.y = 0.f;
                    // E = (xk + conj xm)/2 ; D = (xk - conj xm)/2 ; O = D conj(w^k) ; Z = E + i O
                    const float er = 0.5f * (xk.x + xm.x), ei = 0.5f * (xk.y - xm.y);
                    const float dr = 0.5f * (xk.x - xm.x), di = 0.5f * (xk.y + xm.y);
                    const float2 w = twl[k];
                    const float orr = dr * w.x + di * w.y;
                    const float oi = di * w.x - dr * w.y;
                    v[b * R1 + tt] = mk2(er - oi, ei + orr);
                }
            fft_wave<LOG2M, +1>(v, lane, twl, buf);
        }
        __syncthreads();
        if (aligned) {
            // frames in increasing order (the reference's accumulation order); frame n2 covers the global
            // hop-blocks n2 .. n2 + N/hop - 1, chunk block j is global block p0/hop + j
            const int hp = hop >> 1, R = N / hop;
            const int64_t hb0 = p0 / hop;
            float2* acc2 = reinterpret_cast<float2*>(acc);
            for (int w = 0; w < 4; ++w) {
                const int64_t n2 = nb + w;
                if (n2 > n_hi) break;
                for (int j = 0; j < C; ++j) {
                    const int64_t d = hb0 + j - n2;
                    if (d < 0 || d >= R) continue;
                    const int fo = (int)d * hp;
                    for (int r = tid; r < hp; r += 256) {
                        const float2 z = fbuf[w * MP + pad(fo + r)];
                        const float2 ww = winl[fo + r];
                        float2 a2 = acc2[j * hp + r];
                        a2.x += (z.x * inv_m) * ww.x;
                        a2.y += (z.y * inv_m) * ww.y;
                        acc2[j * hp + r] = a2;
                    }
                }
            }
        } else {
            for (int q = tid; q < span; q += 256) {
                const int64_t p = p0 + q;
                float a = acc[q];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int64_t n2 = nb + w;
                    const int64_t off = p - n2 * hop;
                    if (n2 <= n_hi && off >= 0 && off < N) {
                        const float2 z = fbuf[w * MP + pad((int)(off >> 1))];
                        a += (((off & 1) ? z.y : z.x) * inv_m) * win[off];
                    }
                }
                acc[q] = a;
            }
        }
        __syncthreads();
    }
    const int half = N >> 1;
    for (int q = tid; q < span; q += 256) {
        const int64_t p = p0 + q;
        const int64_t m = p - half;
        if (m < 0 || m >= n_out) continue;
        int64_t f_hi = p / hop;
        if (f_hi > T - 1) f_hi = T - 1;
        const int64_t f_lo = (p < N) ? 0 : (p - N) / hop + 1;
        float norm = 0.f;
        for (int64_t n = f_lo; n <= f_hi; ++n) norm += wsq[p - n * hop];
        if (norm == 0.f) norm = 1.f;
        audio[(int64_t)s * n_out + m] = acc[q] / norm;
    }
}

template <int LOG2M>
int launch_fwd(dcs_stft* p, const float* audio, int64_t L, float* mag, float* phase, float2* unit, int64_t ld,
               int64_t rows_out, int64_t T) {
    constexpr int M = 1 << LOG2M, MP = M + M / 32;
    // frames per workgroup: 4 when there are plenty of frames, 1 to spread a short signal over the CUs
    static const int dbg = getenv("DCS_STFT_DBG") ? atoi(getenv("DCS_STFT_DBG")) : 0;
    static const int fpw_env = getenv("DCS_STFT_FPW") ? atoi(getenv("DCS_STFT_FPW")) : 0;
    int fpw = rows_out >= 8 * (int64_t)p->ctx->n_cu ? 4 : 1;
    if (fpw_env) fpw = fpw_env;
    const size_t lds = ((size_t)(M + 1) + (size_t)fpw * MP) * sizeof(float2);
    auto kern = stft_forward_wave_kernel<LOG2M>;
    if (lds > 48 * 1024)
        DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)dcs_cdiv(rows_out, fpw)), dim3(64 * fpw), lds, p->ctx->stream, audio, L,
                       p->win_f, p->tw_f, mag, phase, unit, ld, p->hop, T, rows_out, (float)sqrt((double)p->frame), dbg);
    return DCS_OK;
}

template <int LOG2M>
int launch_inv(dcs_stft* p, const float* mag, int64_t src_stride, const float* phase, const float2* unit, int64_t ld,
               int64_t T, int n_src, float pre_div, float* audio, int64_t n_out) {
    constexpr int M = 1 << LOG2M, MP = M + M / 32, N = 2 * M;
    const int hop = p->hop;
    const int64_t hops = (n_out + N / 2 + hop - 1) / hop;
    const int R_ = (N + hop - 1) / hop;
    // C hops per workgroup: C + R_ - 1 frames are transformed, 4 at a time
    static const int c_env = getenv("DCS_ISTFT_HOPS") ? atoi(getenv("DCS_ISTFT_HOPS")) : 0;
    int64_t C = hops * n_src / (int64_t)p->ctx->n_cu;
    if (C > 16 - (R_ - 1)) C = 16 - (R_ - 1);
    if (c_env > 0) C = c_env;
    if (C < 1) C = 1;
    const size_t fixed = ((size_t)(M + 1) + 4 * (size_t)MP + (size_t)M) * sizeof(float2);  // + window pairs
    size_t lds = fixed + (size_t)C * hop * sizeof(float);
    while (lds > 96 * 1024 && C > 1) {
        --C;
        lds = fixed + (size_t)C * hop * sizeof(float);
    }
    const dim3 grid((unsigned)((hops + C - 1) / C), (unsigned)n_src);
#define DCS_GO(UNIT_)                                                                                             \
    {                                                                                                             \
        auto kern = istft_wave_kernel<LOG2M, UNIT_>;                                                              \
        if (lds > 48 * 1024)                                                                                      \
            DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, p->ctx->stream, mag, src_stride, phase, unit, ld, p->win_f, \
                           p->wsq_f, p->tw_f, audio, n_out, hop, T, (int)C, pre_div, (float)sqrt((double)N));     \
    }
    if (unit) DCS_GO(true) else DCS_GO(false)
#undef DCS_GO
    return DCS_OK;
}

}  // namespace

bool dcs_fft_wave_supported(const dcs_stft* p) {
    static const bool off = getenv("DCS_FFT_BLOCK") != nullptr;  // debugging aid: force the block-level kernels
    return !off && p->log2m >= 9 && p->log2m <= 11;
}

int dcs_fft_wave_forward(dcs_stft* p, const float* audio, int64_t L, float* mag, float* phase, float2* unit, int64_t ld,
                         int64_t rows_out, int64_t T) {
    switch (p->log2m) {
        case 9: return launch_fwd<9>(p, audio, L, mag, phase, unit, ld, rows_out, T);
        case 10: return launch_fwd<10>(p, audio, L, mag, phase, unit, ld, rows_out, T);
        case 11: return launch_fwd<11>(p, audio, L, mag, phase, unit, ld, rows_out, T);
    }
    DCS_FAIL(DCS_EUNSUPPORTED, "wave FFT: frame size");
}

int dcs_fft_wave_inverse(dcs_stft* p, const float* mag, int64_t src_stride, const float* phase, const float2* unit,
                         int64_t ld, int64_t T, int n_src, float pre_div, float* audio, int64_t n_out) {
    switch (p->log2m) {
        case 9: return launch_inv<9>(p, mag, src_stride, phase, unit, ld, T, n_src, pre_div, audio, n_out);
        case 10: return launch_inv<10>(p, mag, src_stride, phase, unit, ld, T, n_src, pre_div, audio, n_out);
        case 11: return launch_inv<11>(p, mag, src_stride, phase, unit, ld, T, n_src, pre_div, audio, n_out);
    }
    DCS_FAIL(DCS_EUNSUPPORTED, "wave FFT: frame size");
}
