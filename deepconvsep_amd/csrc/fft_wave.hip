// Wave-per-frame STFT / iSTFT for the float32 fast path (N = 1024, 2048, 4096), gfx950.
//
// The block-level radix-4 FFT of fft.hip spends ~60 % of a wave's life parked at workgroup barriers
// (PMC: SQ_WAIT_ANY, profiles/r01_b_pmc_summary.txt): 5 passes x 256 threads x 1 butterfly.  Here one
// 64-lane wavefront owns one frame: M = N/2 complex points, M/64 = 8/16/32 points per lane, three
// Stockham passes of radix 16/16/4 (M=1024), 16/16/8 (M=2048), 8/8/8 (M=512) done in registers and
// exchanged through a per-wave LDS buffer.  A wave's LDS operations are performed in program order, so
// no barrier is needed between passes; a 4-wave workgroup transforms 4 frames independently.
//
// These kernels are VALU-issue bound (PMC: ~3-4 k wave instructions per frame, HBM far from busy), so
// the code is organised around the instruction count:
//   * complex values are (re, im) register pairs and every butterfly is packed fp32 math
//     (v_pk_add/mul/fma_f32); multiplications by +-i and complex products use the op_sel / neg
//     operand modifiers instead of moving halves around (c_add_i, c_mul below);
//   * the exchange buffer is indexed through pad(i) = i + i/32 (bank spread for the stride-R stores);
//     every access is written as pad(lane part) + pad(compile-time part) -- exact for the index sets used
//     here, see cpad() -- so the address of each ds_read/ds_write is one register plus an immediate;
//   * the pass twiddles depend only on the lane, not on the frame: they are fetched once per wave into
//     registers and reused for every frame the wave transforms.
//
// Semantics are those of fft.hip (reference transform.py:277-396); the host side picks this file's
// kernels when the plan is float32 with N in {1024, 2048, 4096} and the block-level ones otherwise.
#include "dcs_internal.h"
#include "fft_wave.h"

#include <math.h>
#include <stdlib.h>

#ifdef DCS_FFTW_TRACE
// experiment build: shader-clock stamps of wave 0 of the middle workgroup of the forward kernel (fftw_trace_dump)
__device__ unsigned long long fftw_trace_buf[64];
#define FW_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) fftw_trace_buf[slot] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#define FW_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#else
#define FW_STAMP(slot)
#define FW_DRAIN()
#endif

namespace {

typedef float cx __attribute__((ext_vector_type(2)));  // (re, im) in an aligned VGPR pair

__device__ __forceinline__ cx mk(float x, float y) { return cx{x, y}; }
__device__ __forceinline__ cx ldc(const float2* p) { return *reinterpret_cast<const cx*>(p); }
__device__ __forceinline__ void stc(float2* p, cx v) { *reinterpret_cast<cx*>(p) = v; }

// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cx c_add_i(cx a, cx b) {
    cx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cx c_sub_i(cx a, cx b) {
    cx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ cx c_add_conj(cx a, cx b) {
    cx r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cx c_sub_conj(cx a, cx b) {
    cx r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x): two packed instructions
__device__ __forceinline__ cx c_mul(cx a, cx w) {
    cx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
__device__ __forceinline__ cx c_mul_conj(cx a, cx w) {
    cx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// DIR < 0: forward transform, multiply by the table value w = exp(-i theta); DIR > 0: by its conjugate
template <int DIR>
__device__ __forceinline__ cx c_tw(cx a, cx w) { return DIR < 0 ? c_mul(a, w) : c_mul_conj(a, w); }

__device__ __forceinline__ int pad(int i) { return i + (i >> 5); }
// pad(l + c) == pad(l) + cpad(c) whenever (l & 31) + (c & 31) < 32; every (lane part, constant part) pair in
// this file satisfies it: the constant parts are multiples of 64, or the lane part is a multiple of the radix
// (a power of two <= 16... 32) while the constant's low bits stay below it (derivations at the call sites).
__device__ __forceinline__ constexpr int cpad(int c) { return c + (c >> 5); }

// exp(-2 pi i j / N), 0 <= j < N, from the half-circle table tw[0..M] (N = 2M)
__device__ __forceinline__ cx tw_fwd(const float2* tw, int j, int M) {
    const cx w = ldc(tw + (j <= M ? j : j - M));
    return j <= M ? w : -w;
}

// The same from the QUARTER-circle table th[0..M/2] (forward kernel, N <= 2048: half the LDS of the table, see
// stft_forward_wave_kernel): exp(-i pi k / M) for M/2 < k <= M is -conj(th[M - k]).
__device__ __forceinline__ cx tw_half(const float2* th, int k, int M) {   // 0 <= k <= M
    const bool hi = k > M / 2;
    const cx w = ldc(th + (hi ? M - k : k));
    return hi ? mk(-w[0], w[1]) : w;
}
__device__ __forceinline__ cx tw_fwd_half(const float2* th, int j, int M) {   // 0 <= j < 2M
    const cx w = tw_half(th, j <= M ? j : j - M, M);
    return j <= M ? w : -w;
}

template <int DIR>
__device__ __forceinline__ void dft4(cx& a, cx& b, cx& c, cx& d) {
    const cx a02 = a + c, s02 = a - c, a13 = b + d, s13 = b - d;
    const cx ym = c_sub_i(s02, s13), yp = c_add_i(s02, s13);
    a = a02 + a13;
    c = a02 - a13;
    b = (DIR < 0) ? ym : yp;
    d = (DIR < 0) ? yp : ym;
}

// multiply by exp(DIR * 2 pi i m / 16)
template <int DIR, int m>
__device__ __forceinline__ cx rot16(cx v) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    constexpr int mm = m & 15;
    if (mm == 0) return v;
    if (mm == 4) return DIR < 0 ? c_sub_i(mk(0.f, 0.f), v) : c_add_i(mk(0.f, 0.f), v);
    // table value exp(-2 pi i mm / 16) = (cos, -sin)
    const cx w = mm == 1 ? mk(C1, -S1) : mm == 2 ? mk(H, -H) : mm == 3 ? mk(S1, -C1) : mm == 6 ? mk(-H, -H) : mk(-C1, S1) /* 9 */;
    return c_tw<DIR>(v, w);
}

// In-register DFTs.  Input v[n], n = 0..R-1; output X[k] is left at v[perm(k)].
template <int DIR>
__device__ __forceinline__ void dft16(cx* v) {
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
__device__ __forceinline__ void dft8(cx* v) {
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
        const cx t = v[2 * c];
        v[2 * c] = t + v[2 * c + 1];
        v[2 * c + 1] = t - v[2 * c + 1];
    }
}
__device__ __forceinline__ constexpr int perm8(int k) { return 2 * (k & 3) + (k >> 2); }

// rot16 with the exponent known only after loop unrolling (m in {0,1,2,3,4,6,9})
template <int DIR>
__device__ __forceinline__ cx rot16_sel(cx v, int m) {
    switch (m) {
        case 0: return v;
        case 1: return rot16<DIR, 1>(v);
        case 2: return rot16<DIR, 2>(v);
        case 3: return rot16<DIR, 3>(v);
        case 4: return rot16<DIR, 4>(v);
        case 6: return rot16<DIR, 6>(v);
        default: return rot16<DIR, 9>(v);
    }
}

template <int R, int DIR>
__device__ __forceinline__ void dftR(cx* v) {
    if (R == 16) dft16<DIR>(v);
    else if (R == 8) dft8<DIR>(v);
    else dft4<DIR>(v[0], v[1], v[2], v[3]);
}
template <int R>
__device__ __forceinline__ constexpr int permR(int k) { return R == 16 ? perm16(k) : (R == 8 ? perm8(k) : k); }

template <int LOG2M> struct Plan;
template <> struct Plan<9> { static constexpr int R1 = 8, R2 = 8, R3 = 8; };
template <> struct Plan<10> { static constexpr int R1 = 16, R2 = 16, R3 = 4; };
template <> struct Plan<11> { static constexpr int R1 = 16, R2 = 16, R3 = 8; };

// Per-lane pass twiddles (table values exp(-i theta)), constant over the frames a wave transforms.
// Pass 2 (Ns = R1): k = lane mod R1 for every block b.  Pass 3 (Ns = R1 R2): k = lane + 64 b, twiddle
// w^(2 t (lane + 64 b)) = w^(2 t lane) * W16^(t b) for M = 1024: only the lane part is kept (R3 - 1 registers pairs),
// the block part is one of the constant rotations of rot16 (7 of the 12 are trivial or absent).  For M = 2048 the
// pass-3 twiddles stay in the LDS table (64 registers of data already).
template <int LOG2M>
struct WaveTw {
    using PL = Plan<LOG2M>;
    static constexpr int M = 1 << LOG2M, P = M / 64, NB3 = P / PL::R3;
    static constexpr bool REG3 = LOG2M <= 10;
    cx t2[PL::R2 - 1];
    cx t3[REG3 ? (PL::R3 - 1) : 1];
    static_assert(!REG3 || NB3 == 1 || M == 1024, "block part of the pass-3 twiddle is W16^(t b) only for M = 1024");
    __device__ __forceinline__ void init(const float2* tw, int lane) {
        const int step2 = ((2 * M) / PL::R2 / PL::R1) * (lane & (PL::R1 - 1));
#pragma unroll
        for (int t = 1; t < PL::R2; ++t) t2[t - 1] = tw_fwd(tw, t * step2, M);
        if (REG3) {
#pragma unroll
            for (int t = 1; t < PL::R3; ++t) t3[t - 1] = tw_fwd(tw, t * 2 * lane, M);
        }
    }
    __device__ __forceinline__ void init_half(const float2* th, int lane) {   // from the quarter-circle table (tw_half)
        const int step2 = ((2 * M) / PL::R2 / PL::R1) * (lane & (PL::R1 - 1));
#pragma unroll
        for (int t = 1; t < PL::R2; ++t) t2[t - 1] = tw_fwd_half(th, t * step2, M);
        if (REG3) {
#pragma unroll
            for (int t = 1; t < PL::R3; ++t) t3[t - 1] = tw_fwd_half(th, t * 2 * lane, M);
        }
    }
};

// Three Stockham passes.  On entry v holds the first-pass input v[b*R1 + t] = x[lane + 64 b + t*M/R1]; on exit
// the natural-order result is in buf[pad(k)] (wave-private).  Pass with sub-transform length Ns writes
// y[(j/Ns)*Ns*R + (j%Ns) + t*Ns], j = lane + 64 b.
template <int LOG2M, int DIR>
__device__ __forceinline__ void fft_wave(cx (&v)[(1 << LOG2M) / 64], int lane, const WaveTw<LOG2M>& w,
                                         const float2* twl, float2* buf) {
    using PL = Plan<LOG2M>;
    constexpr int M = 1 << LOG2M, P = M / 64;
    constexpr int R1 = PL::R1, R2 = PL::R2, R3 = PL::R3;
    constexpr int NB1 = P / R1, NB2 = P / R2, NB3 = P / R3;
    float2* bl = buf + pad(lane);  // loads of every pass: index lane + 64 (b + t NB), constant part % 64 == 0
    // ---- pass 1 (Ns = 1, no twiddles): index R1 (lane + 64 b) + t; lane part % R1 == 0, constant low bits t < R1
    {
        float2* bs = buf + pad(R1 * lane);
#pragma unroll
        for (int b = 0; b < NB1; ++b) dftR<R1, DIR>(&v[b * R1]);
#pragma unroll
        for (int b = 0; b < NB1; ++b)
#pragma unroll
            for (int t = 0; t < R1; ++t) stc(bs + cpad(R1 * 64 * b + t), v[b * R1 + permR<R1>(t)]);
    }
    if (DIR < 0) FW_STAMP(4);   // pass 1 done, stores issued
#pragma unroll
    for (int b = 0; b < NB2; ++b)
#pragma unroll
        for (int t = 0; t < R2; ++t) v[b * R2 + t] = ldc(bl + cpad(64 * b + t * 64 * NB2));
    if (DIR < 0) { FW_DRAIN(); FW_STAMP(5); }   // exchange 1 complete
    // ---- pass 2 (Ns = R1): k = lane % R1, index (lane - k) R2 + k + 64 b R2 + t R1; lane part & 31 = k < R1,
    // constant part & 31 is a multiple of R1 below 32
    {
        const int k = lane & (R1 - 1);
        float2* bs = buf + pad((lane - k) * R2 + k);
#pragma unroll
        for (int b = 0; b < NB2; ++b) {
#pragma unroll
            for (int t = 1; t < R2; ++t) v[b * R2 + t] = c_tw<DIR>(v[b * R2 + t], w.t2[t - 1]);
            dftR<R2, DIR>(&v[b * R2]);
        }
#pragma unroll
        for (int b = 0; b < NB2; ++b)
#pragma unroll
            for (int t = 0; t < R2; ++t) stc(bs + cpad(64 * b * R2 + t * R1), v[b * R2 + permR<R2>(t)]);
    }
    if (DIR < 0) FW_STAMP(6);   // pass 2 done
#pragma unroll
    for (int b = 0; b < NB3; ++b)
#pragma unroll
        for (int t = 0; t < R3; ++t) v[b * R3 + t] = ldc(bl + cpad(64 * b + t * 64 * NB3));
    if (DIR < 0) { FW_DRAIN(); FW_STAMP(7); }   // exchange 2 complete
    // ---- pass 3 (Ns = R1 R2 = M / R3 >= 64 NB3): k = j, index lane + 64 b + t Ns
    {
        constexpr int Ns = R1 * R2;
#pragma unroll
        for (int b = 0; b < NB3; ++b) {
#pragma unroll
            for (int t = 1; t < R3; ++t) {
                if (WaveTw<LOG2M>::REG3) {
                    v[b * R3 + t] = rot16_sel<DIR>(c_tw<DIR>(v[b * R3 + t], w.t3[t - 1]), t * b);
                } else {
                    v[b * R3 + t] = c_tw<DIR>(v[b * R3 + t], tw_fwd(twl, t * 2 * (lane + 64 * b), M));
                }
            }
            dftR<R3, DIR>(&v[b * R3]);
        }
#pragma unroll
        for (int b = 0; b < NB3; ++b)
#pragma unroll
            for (int t = 0; t < R3; ++t) stc(bl + cpad(64 * b + t * Ns), v[b * R3 + permR<R3>(t)]);
    }
}

// ------------------------------------------------------------------------------------------------
// forward (compute_file): FPW frames per workgroup, one per wave
// ------------------------------------------------------------------------------------------------
// (the pass twiddles gathered from global memory instead of an LDS table -- no fill, no barrier, 4 waves per SIMD -- measured
// slower at every size: 0.099 vs 0.089 ms at 4096 tiles, 25 - 27 vs 22 us at 640; scripts/README.md)
// per-wave staging of the forward kernel's outputs: 256 magnitudes (1 KB) + 128 phasors (1 KB), see the wide store path
constexpr int kStftStageF2 = 256;

// N <= 2048 (HALF): four workgroups of four waves per CU instead of three, so that the 3 720 - 3 800 frames of a 20 x 32-tile launch
// group are resident in ONE round (4 096 wave slots; with three per CU the launch was 1.24 rounds = two wave lives).  That
// needs <= 40 KB of LDS per workgroup and <= 128 registers:
//   * the twiddle table holds the quarter circle only (tw_half: exp(-i pi k / M), k <= M / 2; the rest by symmetry);
//   * no staging area of its own: bins k and M - k are computed TOGETHER from the one pair (Z[k], Z[M - k]) they share --
//     X[k] = E + w^k O, X[M - k] = conj(E - w^k O): half the LDS reads and twiddle products of the split -- 256 direct and 256
//     mirrored bins per phase into registers, and the wide stores are staged through the exchange-buffer slots that phase has
//     just read for the last time.
template <int LOG2M>
__global__ __launch_bounds__(256, LOG2M <= 10 ? 4 : 1) void stft_forward_wave_kernel(const float* __restrict__ audio, int64_t L, const float* __restrict__ win,
                                         const float2* __restrict__ tw, float* __restrict__ mag,
                                         float* __restrict__ phase, float2* __restrict__ unit, int64_t ld, int hop,
                                         int64_t T, int64_t rows_pc, int64_t n_clips, int64_t audio_stride,
                                         float sqrt_n, int interleave, const int64_t* __restrict__ clip_tab) {
    constexpr int M = 1 << LOG2M, P = M / 64, MP = M + M / 32;
    constexpr int R1 = Plan<LOG2M>::R1, NB1 = P / R1, stride1 = 64 * NB1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row pointers stay in SGPRs
    FW_STAMP(0);
    constexpr bool HALF = LOG2M <= 10;
    constexpr int kTab = HALF ? M / 2 + 2 : M + 1;       // table entries (HALF: an even count keeps the buffers 16-byte aligned)
    // !HALF: [waves][kStageF2] output staging (16-byte aligned: first in the segment), then the table, then the waves' exchange buffers
    float2* stage = reinterpret_cast<float2*>(smem) + wave * kStftStageF2;
    float2* lds0 = reinterpret_cast<float2*>(smem) + (HALF ? 0 : (blockDim.x >> 6) * kStftStageF2);
    const float2* twl = lds0;
    float2* buf = lds0 + kTab + wave * MP;
    for (int k = tid; k <= (HALF ? M / 2 : M); k += blockDim.x) lds0[k] = tw[k];
    __syncthreads();
    // output row -> (clip, frame): clips of equal length are stacked with a pitch of rows_pc rows
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (row >= rows_pc * n_clips) return;
    const int64_t clip = row / rows_pc;
    const int64_t t = row - clip * rows_pc;
    int64_t orow = interleave ? t * n_clips + clip : row;         // interleave: rows ordered [frame][clip]
    if (clip_tab) {   // clips of different lengths share the launch: kDcsClipTab entries per clip (dcs_internal.h)
        L = clip_tab[kDcsClipTab * clip];
        T = clip_tab[kDcsClipTab * clip + 1];
        const int64_t r0 = clip_tab[kDcsClipTab * clip + 3];
        if (r0 >= 0) {                                            // compact layout: the clip's own rows at its own offset
            if (t >= clip_tab[kDcsClipTab * clip + 5]) return;
            orow = r0 + t;
        }
    }
    audio += clip * audio_stride;
    float* mrow = mag + orow * ld;
    float* prow = phase ? phase + orow * ld : nullptr;
    float2* urow = unit ? unit + orow * ld : nullptr;
    if (t >= T) {
        for (int k = lane; k < ld; k += 64) {
            mrow[k] = 0.f;
            if (prow) prow[k] = 0.f;
            if (urow) stc(urow + k, mk(1.f, 0.f));
        }
        return;
    }
    FW_STAMP(1);   // table filled, barrier passed
    WaveTw<LOG2M> wt;
    if (HALF) wt.init_half(twl, lane); else wt.init(twl, lane);
    cx v[P];
    const int64_t base = t * (int64_t)hop - M;
    // 32-bit window-relative bounds instead of two 64-bit compares per sample
    const int r_lo = base < 0 ? (int)(-base) : 0;
    const int64_t rem = L - base;
    const int r_hi = rem < 2 * M ? (rem < 0 ? 0 : (int)rem) : 2 * M;
    const float* ap = audio + base;
    const float inv_sqrt_n = 1.f / sqrt_n;
    const float2* w2 = reinterpret_cast<const float2*>(win);
    // whole frame in range and its sample pairs 8-byte aligned: one 64-bit load per pair
    const bool inside = r_lo == 0 && r_hi == 2 * M && (reinterpret_cast<uintptr_t>(ap) & 7) == 0;
    // All sample and window loads are issued before the first value is used, on the edge frames too: their loads are
    // unconditional (index clamped into the clip) and the out-of-range samples are zeroed afterwards.  With `if (in range)
    // x[0] = ap[r]` the compiler waited for each of an edge frame's 32 loads in turn -- sixteen memory round trips in the
    // wave of the first / last frames of every clip, and a launch ends with its slowest wave.
    cx xs[P], ws[P];
    // (a load's immediate offset reaches 4 KB: the upper half of the window / frame goes through a second SCALAR base instead of
    // one address register per load -- eight registers that decide between 128 and 136 in the four-waves-per-SIMD build)
    auto uniform = [](uint64_t v) {                         // a wave-uniform address the compiler cannot prove uniform
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    const float2* w2_hi = reinterpret_cast<const float2*>(uniform(reinterpret_cast<uint64_t>(w2 + 512)));
    const float* ap_hi = reinterpret_cast<const float*>(uniform(reinterpret_cast<uint64_t>(ap + 1024)));
#pragma unroll
    for (int b = 0; b < NB1; ++b)
#pragma unroll
        for (int tt = 0; tt < R1; ++tt) {
            const int i = lane + 64 * b + tt * stride1;
            const int ic = 64 * b + tt * stride1;          // compile-time part
            ws[b * R1 + tt] = (HALF && ic >= 512) ? ldc(w2_hi + lane + (ic - 512)) : ldc(w2 + i);
        }
    if (inside) {
#pragma unroll
        for (int b = 0; b < NB1; ++b)
#pragma unroll
            for (int tt = 0; tt < R1; ++tt) {
                const int r = 2 * (lane + 64 * b + tt * stride1);
                const int rc = 2 * (64 * b + tt * stride1);
                xs[b * R1 + tt] = (HALF && rc >= 1024) ? *reinterpret_cast<const cx*>(ap_hi + 2 * lane + (rc - 1024))
                                                       : *reinterpret_cast<const cx*>(ap + r);
            }
    } else if (r_hi > r_lo) {
#pragma unroll
        for (int b = 0; b < NB1; ++b)
#pragma unroll
            for (int tt = 0; tt < R1; ++tt) {
                const int r = 2 * (lane + 64 * b + tt * stride1);
                const int r0 = r < r_lo ? r_lo : (r >= r_hi ? r_hi - 1 : r);
                const int r1 = r + 1 < r_lo ? r_lo : (r + 1 >= r_hi ? r_hi - 1 : r + 1);
                xs[b * R1 + tt] = mk(ap[r0], ap[r1]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < P; ++i) xs[i] = mk(0.f, 0.f);
    }
#pragma unroll
    for (int b = 0; b < NB1; ++b)
#pragma unroll
        for (int tt = 0; tt < R1; ++tt) {
            const int r = 2 * (lane + 64 * b + tt * stride1);
            cx x = xs[b * R1 + tt];
            if (!inside) {                                         // wave-uniform
                x[0] = (r >= r_lo && r < r_hi) ? x[0] : 0.f;
                x[1] = (r + 1 >= r_lo && r + 1 < r_hi) ? x[1] : 0.f;
            }
            v[b * R1 + tt] = x * ws[b * R1 + tt];
        }
    FW_STAMP(2);   // pass twiddles read, samples and window requested
    FW_DRAIN();
    FW_STAMP(3);   // arrived
    fft_wave<LOG2M, -1>(v, lane, wt, twl, buf);
    FW_STAMP(8);   // pass 3 done, stores issued
    // even/odd split: X[k] = E + w^k O with E = (Z[k] + conj Z[M-k])/2, O = -i (Z[k] - conj Z[M-k])/2.
    // k = lane + 64 u; M - k = (64 - lane) + (M - 64 (u + 1)): constant parts are multiples of 64.
    const float2* bk = buf + pad(lane);
    const float2* bm = buf + pad(64 - lane);
    // bin k = lane + 64 u (u == P: k = M, lane 0 only) -> magnitude before the 1 / sqrt(N) scale, and X[k]
    auto bin = [&](int u, float& ax, float& xr, float& xi) {
        const int k = lane + 64 * u;
        const cx zk = u < P ? ldc(bk + cpad(64 * u)) : ldc(buf);                      // Z[k mod M]
        const cx zm = (u == P || (u == 0 && lane == 0)) ? ldc(buf) : ldc(bm + cpad(M - 64 * (u + 1)));  // Z[(M-k) mod M]
        const cx e2 = c_add_conj(zk, zm);  // 2 E
        const cx d2 = c_sub_conj(zk, zm);  // 2 i O  ->  O = -i d2 / 2
        const cx o2 = c_sub_i(mk(0.f, 0.f), d2);
        const cx x2 = e2 + c_mul(o2, HALF ? tw_half(twl, k, M) : ldc(twl + k));
        xr = 0.5f * x2[0];
        xi = 0.5f * x2[1];
        // v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the correctly rounded sequences (~8 and ~10 instructions per bin):
        // far inside the float32 FFT's own rounding error
        ax = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
    };
    auto phasor = [](float ax, float xr, float xi) {
        const float ra = __builtin_amdgcn_rcpf(ax);  // one reciprocal for both components
        return (ax > 0.f) ? mk(xr * ra, xi * ra) : mk(1.f, 0.f);
    };
    // Wide stores (measured: the kernel is bound by ISSUING its 34 four- and eight-byte store instructions per frame, not by
    // their bytes -- without stores 0.062 instead of 0.088 ms at 4096 tiles): four 64-bin blocks of magnitudes / two of
    // phasors go through a wave-private 2 KB staging area and leave as ONE 16-byte store per lane (1 KB per instruction):
    // 4 + 8 + 3 store instructions per frame.  Needs rows of exactly M + 4 bins on 16-byte boundaries and no phase output.
    const bool wide = (P % 4 == 0) && ld == M + 4 && !prow && urow &&
                      ((reinterpret_cast<uintptr_t>(mag) | reinterpret_cast<uintptr_t>(unit)) & 15) == 0;
    if (wide && HALF) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        constexpr int NPH = P / 8;                            // phases of 4 x 64 pairs: 256 direct + 256 mirrored bins each
        static_assert(P % 8 == 0, "whole phases");
        // one pair: both bins' magnitude (before the 1 / sqrt(N) scale) and unit phasor
        auto pair = [&](cx zk, cx zm, cx w, float& axd, cx& phd, float& axm, cx& phm) {
            const cx e2 = c_add_conj(zk, zm);                 // 2 E
            const cx d2 = c_sub_conj(zk, zm);                 // 2 i O
            const cx t2 = c_mul(c_sub_i(mk(0.f, 0.f), d2), w);   // 2 w^k O
            const cx xd = (e2 + t2) * 0.5f;                   // X[k]
            const cx xc = (e2 - t2) * 0.5f;                   // conj X[M - k]
            axd = __builtin_amdgcn_sqrtf(xd[0] * xd[0] + xd[1] * xd[1]);
            axm = __builtin_amdgcn_sqrtf(xc[0] * xc[0] + xc[1] * xc[1]);
            phd = phasor(axd, xd[0], xd[1]);
            phm = phasor(axm, xc[0], -xc[1]);
        };
        // k = M / 2 is its own partner (lane 0): the first bin of the last phase's mirrored group -- the "carry" of a phase is
        // the bin one past its mirrored run: the mirror of its first pair
        float c_ax = 0.f;
        cx c_ph = mk(1.f, 0.f);
        {
            float axm;
            cx phm;
            const cx z = ldc(buf + pad(M / 2));
            pair(z, z, mk(0.f, -1.f), c_ax, c_ph, axm, phm);  // w^(M/2) = -i
        }
#pragma unroll
        for (int ph = NPH - 1; ph >= 0; --ph) {
            float axd[4], axm[4];
            cx phd[4], phm[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int u = 4 * ph + uu;                    // k = lane + 64 u < M / 2
                const cx zk = ldc(bk + cpad(64 * u));
                const cx zm = (u == 0 && lane == 0) ? ldc(buf) : ldc(bm + cpad(M - 64 * (u + 1)));   // Z[(M - k) mod M]
                pair(zk, zm, ldc(twl + lane + 64 * u), axd[uu], phd[uu], axm[uu], phm[uu]);
            }
            // every slot of Z this phase reads has now been read for the last time: k in [256 ph, 256 ph + 255] and its mirror run
            // [M - 256 ph - 255, M - 256 ph] -- the two staging areas (even slot index: 16-byte aligned, >= 256 slots each)
            float* sm = reinterpret_cast<float*>(buf + cpad(256 * ph));                          // 512 floats: direct, mirrored
            float2* su = buf + ((cpad(M - 256 * ph - 255) + 1) & ~1);                            // 256 phasors
            const int q = M / 256 - 1 - ph;                   // the mirrored group: bins [256 q, 256 q + 255]
            asm volatile("" ::: "memory");
            // mirrored bin of (uu, lane) = M - 256 ph - 64 uu - lane; its index in group q = 256 - 64 uu - lane (lane 0 of uu 0: the
            // NEXT group's first bin = this phase's carry out; index 0 of group q = the carry in)
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                sm[64 * uu + lane] = axd[uu] * inv_sqrt_n;
                if (uu > 0 || lane > 0) sm[256 + 256 - 64 * uu - lane] = axm[uu] * inv_sqrt_n;
            }
            if (lane == 0) sm[256] = c_ax * inv_sqrt_n;
            asm volatile("" ::: "memory");
            {
                const f4 a = *reinterpret_cast<const f4*>(sm + 4 * lane);
                const f4 b = *reinterpret_cast<const f4*>(sm + 256 + 4 * lane);
                asm volatile("" ::: "memory");
                *reinterpret_cast<f4*>(mrow + 256 * ph + 4 * lane) = a;
                *reinterpret_cast<f4*>(mrow + 256 * q + 4 * lane) = b;
            }
            // phasors: the direct group, then the mirrored one, 128 bins per 16-byte store instruction
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) stc(su + 64 * uu + lane, phd[uu]);
            asm volatile("" ::: "memory");
            {
                const f4 a = *reinterpret_cast<const f4*>(su + 2 * lane);
                const f4 b = *reinterpret_cast<const f4*>(su + 128 + 2 * lane);
                asm volatile("" ::: "memory");
                *reinterpret_cast<f4*>(urow + 256 * ph + 2 * lane) = a;
                *reinterpret_cast<f4*>(urow + 256 * ph + 128 + 2 * lane) = b;
            }
#pragma unroll
            for (int uu = 0; uu < 4; ++uu)
                if (uu > 0 || lane > 0) stc(su + 256 - 64 * uu - lane, phm[uu]);
            if (lane == 0) stc(su, c_ph);
            asm volatile("" ::: "memory");
            {
                const f4 a = *reinterpret_cast<const f4*>(su + 2 * lane);
                const f4 b = *reinterpret_cast<const f4*>(su + 128 + 2 * lane);
                asm volatile("" ::: "memory");
                *reinterpret_cast<f4*>(urow + 256 * q + 2 * lane) = a;
                *reinterpret_cast<f4*>(urow + 256 * q + 128 + 2 * lane) = b;
            }
            // carry out: the mirror of this phase's first pair (lane 0 of uu 0) = bin M - 256 ph
            c_ax = axm[0];
            c_ph = phm[0];
        }
        if (lane == 0) {   // bin M (the carry of phase 0) and the three padding bins of the row
            *reinterpret_cast<f4*>(mrow + M) = f4{c_ax * inv_sqrt_n, 0.f, 0.f, 0.f};
            *reinterpret_cast<f4*>(urow + M) = f4{c_ph[0], c_ph[1], 1.f, 0.f};
            *reinterpret_cast<f4*>(urow + M + 2) = f4{1.f, 0.f, 1.f, 0.f};
        }
    } else if (wide) {
        float* sm = reinterpret_cast<float*>(stage);          // [256] magnitudes of bins 256 g .. 256 g + 255
        float2* su = stage + 128;                             // [128] phasors of bins 128 h .. 128 h + 127
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int g = 0; g < P / 4; ++g) {
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                float ax, xr, xi;
                bin(4 * g + uu, ax, xr, xi);
                sm[64 * uu + lane] = ax * inv_sqrt_n;
                stc(su + 64 * (uu & 1) + lane, phasor(ax, xr, xi));
                if (uu & 1) {
                    // a wave's LDS instructions execute in order, so the other lanes' writes above are complete when this
                    // read executes; the empty asm keeps the COMPILER from moving accesses of different types across it
                    asm volatile("" ::: "memory");
                    const f4 q = *reinterpret_cast<const f4*>(su + 2 * lane);
                    asm volatile("" ::: "memory");
                    *reinterpret_cast<f4*>(urow + 128 * (2 * g + (uu >> 1)) + 2 * lane) = q;
                }
            }
            asm volatile("" ::: "memory");
            const f4 q = *reinterpret_cast<const f4*>(sm + 4 * lane);
            asm volatile("" ::: "memory");
            *reinterpret_cast<f4*>(mrow + 256 * g + 4 * lane) = q;
        }
        if (lane == 0) {   // bin M and the three padding bins of the row
            float ax, xr, xi;
            bin(P, ax, xr, xi);
            const cx ph = phasor(ax, xr, xi);
            *reinterpret_cast<f4*>(mrow + M) = f4{ax * inv_sqrt_n, 0.f, 0.f, 0.f};
            *reinterpret_cast<f4*>(urow + M) = f4{ph[0], ph[1], 1.f, 0.f};
            *reinterpret_cast<f4*>(urow + M + 2) = f4{1.f, 0.f, 1.f, 0.f};
        }
    } else {
#pragma unroll
        for (int u = 0; u <= P; ++u) {
            const int k = lane + 64 * u;
            if (u == P && lane > 0) break;  // k <= M
            float ax, xr, xi;
            bin(u, ax, xr, xi);
            mrow[k] = ax * inv_sqrt_n;
            if (prow) prow[k] = atan2f(xi, xr);
            if (urow) stc(urow + k, phasor(ax, xr, xi));
        }
        for (int k = M + 1 + lane; k < ld; k += 64) {
            mrow[k] = 0.f;
            if (prow) prow[k] = 0.f;
            if (urow) stc(urow + k, mk(1.f, 0.f));
        }
    }
    FW_STAMP(9);    // split + magnitude / phasor done, every store issued
    FW_DRAIN();
    FW_STAMP(10);   // stores acknowledged
}

// ------------------------------------------------------------------------------------------------
// inverse (compute_inverse): a workgroup walks C consecutive hop-blocks of one source.  Its 4 waves transform
// 4 consecutive frames per step; then every thread adds the windowed frames -- in frame order, the
// reference's accumulation order (transform.py:381-389) -- into a ring of hop-blocks in LDS, and the blocks
// that no later frame touches are normalised by sum(w^2) and written out.  A thread always works on the same
// sample pairs of a block, so the ring needs no synchronisation of its own, the accumulator is as small as
// N/hop + 3 blocks whatever C is, and only the N/hop - 1 frames before the first block are transformed
// twice (by this workgroup and by its left neighbour).  Requires hop | N and hop even.
// ------------------------------------------------------------------------------------------------
// exp(-i pi j / 16), j = 0..15: the part w^(64 j) of the pre-processing twiddle w^(lane + 64 j) for M = 1024 (every
// second entry for M = 512)
__device__ __forceinline__ cx w_pi16(int j) {
    constexpr float c[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                             0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f,
                             -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                             -0.92387953251128674f, -0.98078528040323043f};
    // -sin(pi j/16) = cos(pi (j+8)/16) for j < 8, = -cos(pi (j-8)/16) for j >= 8
    return mk(c[j & 15], c[(j + 8) & 15] * ((j & 15) >= 8 ? -1.f : 1.f));
}

// The general form: any hop | N (frameSize 4096, or more than four hop-blocks per frame); the configurations the reference ships
// for frameSize <= 2048 take the barrier-free kernels below.  (Round 1 - 4 kept variants of this kernel with the window pairs
// in registers for the shapes those kernels now cover; removed in round 5.)
template <int LOG2M, bool UNIT>
__global__ __launch_bounds__(256) void istft_wave_kernel(const float* __restrict__ mag, int64_t src_stride,
                                                         const float* __restrict__ phase,
                                                         const float2* __restrict__ unit, int64_t ld,
                                                         const float* __restrict__ win, const float* __restrict__ wsq,
                                                         const float2* __restrict__ tw, float* __restrict__ audio,
                                                         int64_t n_out, int hop, int64_t T, int C, int64_t n_blocks,
                                                         int n_chunks, int n_src, int ring_slots, float pre_div,
                                                         float sqrt_n, int64_t unit_clip_stride, int src_per_clip,
                                                         const int64_t* __restrict__ clip_tab, int64_t out_stride) {
    constexpr int M = 1 << LOG2M, N = 2 * M, P = M / 64, MP = M + M / 32;
    constexpr int R1 = Plan<LOG2M>::R1, NB1 = P / R1, stride1 = 64 * NB1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lp = reinterpret_cast<float2*>(smem);
    float2* twl = lp;                                    // [M + 1]
    lp += M + 1;
    float2* fbuf = lp;                                   // [4][MP] transformed frames
    lp += 4 * MP;
    float2* winl = lp;                                   // [M] window as (even, odd) pairs
    lp += M;
    float2* ring = lp;                                   // [ring_slots][hop/2]
    float* norm_s = reinterpret_cast<float*>(ring + (size_t)ring_slots * (hop >> 1));  // [hop] steady-state 1 / sum(w^2)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* buf = fbuf + wave * MP;
    // block -> (chunk, source): the n_src sources of a chunk read the same phasor rows; they get block ids 8
    // apart (same XCD, dispatched together), so one of them pulls a row from HBM and the others hit that L2
    int chunk, s;
    {
        const int b = blockIdx.x, full = (n_chunks >> 3) * 8 * n_src;
        if (b < full) {
            chunk = (b / (8 * n_src)) * 8 + (b & 7);
            s = (b >> 3) % n_src;
        } else {
            const int r = n_chunks & 7, b2 = b - full;
            chunk = (n_chunks >> 3) * 8 + b2 % r;
            s = b2 / r;
        }
    }
    const int hp = hop >> 1, R = N / hop, rmask = ring_slots - 1;
    if (clip_tab) {   // clips of different lengths: this source's own sample / frame / hop-block counts
        const int64_t c = s / src_per_clip;
        n_out = clip_tab[kDcsClipTab * c];
        T = clip_tab[kDcsClipTab * c + 1];
        n_blocks = (n_out + M + hop - 1) / hop;
        if ((int64_t)chunk * C >= n_blocks) return;   // workgroup-uniform, before any barrier
    }
    const int64_t hb0 = (int64_t)chunk * C;
    const int64_t hb1 = (hb0 + C < n_blocks) ? hb0 + C : n_blocks;

    {
        const float2* w2g = reinterpret_cast<const float2*>(win);
        for (int k = tid; k < M; k += 256) winl[k] = w2g[k];
        for (int k = tid; k <= M; k += 256) twl[k] = tw[k];
        for (int q = tid; q < ring_slots * hp; q += 256) stc(ring + q, mk(0.f, 0.f));
        for (int q = tid; q < hop; q += 256) {
            float nrm = 0.f;
            for (int d = R - 1; d >= 0; --d) nrm += wsq[q + d * hop];  // frames in increasing order
            norm_s[q] = nrm == 0.f ? 1.f : 1.f / nrm;  // reciprocal: the steady-state blocks multiply (one IEEE division
                                                       // is ~10 VALU instructions, 8 of them per lane and frame)
        }
    }
    int64_t n_hi = hb1 - 1;
    if (n_hi > T - 1) n_hi = T - 1;
    const int64_t n_lo = (hb0 < R) ? 0 : hb0 - (R - 1);
    const float inv_m = 1.f / (float)M;
    const float amp = 0.5f * (sqrt_n / pre_div);  // (mag / scale_factor) sqrt(N), and the 1/2 of the even/odd split
    const float* msrc = mag + (int64_t)s * src_stride;
    {   // stacked clips: each has its own phasor rows -- at a uniform pitch, or (compact ragged layout) at its row offset
        const int64_t c_ = s / src_per_clip;
        const int64_t r0_ = clip_tab ? clip_tab[kDcsClipTab * c_ + 3] : -1;
        unit += r0_ >= 0 ? r0_ * ld : c_ * unit_clip_stride;
    }
    float* dst = audio + (int64_t)s * out_stride;
    __syncthreads();
    WaveTw<LOG2M> wt;
    wt.init(twl, lane);

    int64_t g_done = hb0;  // next block to write out
    // flush blocks [g_done, g_end): every frame that touches them has been added (or does not exist)
    auto flush = [&](int64_t g_end, int64_t n_last) {
        for (int64_t g = g_done; g < g_end; ++g) {
            const int slot = (int)(g & rmask);
            const bool steady = g >= R - 1 && g <= T - 1;
            const bool live = g <= n_last + R - 1;  // beyond the last frame's reach: zeros
            for (int r = tid; r < hp; r += 256) {
                cx a = live ? ldc(ring + slot * hp + r) : mk(0.f, 0.f);
                if (live) stc(ring + slot * hp + r, mk(0.f, 0.f));
                const int64_t p = g * hop + 2 * r;
                const int64_t m = p - M;
                if (m + 1 < 0 || m >= n_out) continue;
                float nx, ny;
                if (steady) {
                    if (m >= 0) dst[m] = a[0] * norm_s[2 * r];
                    if (m + 1 < n_out) dst[m + 1] = a[1] * norm_s[2 * r + 1];
                    continue;
                } else {
                    const int64_t f_hi = g < T - 1 ? g : T - 1;
                    const int64_t f_lo = g < R ? 0 : g - (R - 1);
                    nx = 0.f;
                    ny = 0.f;
                    for (int64_t n = f_lo; n <= f_hi; ++n) {
                        nx += wsq[p - n * hop];
                        ny += wsq[p + 1 - n * hop];
                    }
                    if (nx == 0.f) nx = 1.f;
                    if (ny == 0.f) ny = 1.f;
                }
                if (m >= 0) dst[m] = a[0] / nx;
                if (m + 1 < n_out) dst[m + 1] = a[1] / ny;  // m + 1 >= 0 here
            }
        }
        if (g_end > g_done) g_done = g_end;
    };

    for (int64_t nb = n_lo; nb <= n_hi; nb += 4) {
        const int64_t n = nb + wave;
        if (n <= n_hi) {
            const float* mrow = msrc + n * ld;
            const float2* urow = unit + n * ld;
            const float* prow = phase + n * ld;
            cx v[P];
#pragma unroll
            for (int b = 0; b < NB1; ++b)
#pragma unroll
                for (int tt = 0; tt < R1; ++tt) {
                    const int k = lane + 64 * b + tt * stride1;  // 0 <= k < M
                    const int km = M - k;                         // 1..M
                    const float a = mrow[k] * amp;
                    const float b2 = mrow[km] * amp;
                    cx xk, xm;
                    if (UNIT) {
                        xk = ldc(urow + k) * a;
                        xm = ldc(urow + km) * b2;
                    } else {
                        float sn, cs;
                        sincosf(prow[k], &sn, &cs);
                        xk = mk(a * cs, a * sn);
                        sincosf(prow[km], &sn, &cs);
                        xm = mk(b2 * cs, b2 * sn);
                    }
                    if (b == 0 && tt == 0) {  // k == 0 only there: imaginary parts of DC / Nyquist are ignored (numpy irfft)
                        if (lane == 0) {
                            xk[1] = 0.f;
                            xm[1] = 0.f;
                        }
                    }
                    // E = (xk + conj xm)/2 ; D = (xk - conj xm)/2 ; O = D conj(w^k) ; Z = E + i O   (1/2 is in amp)
                    const cx e = c_add_conj(xk, xm), d = c_sub_conj(xk, xm);
                    const cx wk = ldc(twl + k);
                    const cx o = c_mul_conj(d, wk);
                    v[b * R1 + tt] = c_add_i(e, o);
                }
            fft_wave<LOG2M, +1>(v, lane, wt, twl, buf);
        }
        __syncthreads();
        // frame n2 covers the hop-blocks n2 .. n2 + R - 1
        for (int w = 0; w < 4; ++w) {
            const int64_t n2 = nb + w;
            if (n2 > n_hi) break;
            for (int d = 0; d < R; ++d) {
                const int64_t g = n2 + d;
                if (g < hb0 || g >= hb1) continue;
                const int slot = (int)(g & rmask);
                const int fo = d * hp;
                for (int r = tid; r < hp; r += 256) {
                    const cx z = ldc(fbuf + w * MP + pad(fo + r));
                    const cx ww = ldc(winl + fo + r);
                    cx a2 = ldc(ring + slot * hp + r);
                    a2 += (z * inv_m) * ww;
                    stc(ring + slot * hp + r, a2);
                }
            }
        }
        const bool last = nb + 4 > n_hi;
        flush(last ? hb1 : (nb + 4 < hb1 ? nb + 4 : hb1), n_hi);
        __syncthreads();
    }
    flush(hb1, n_hi < n_lo ? -R : n_hi);  // a chunk no frame reaches: zeros
}

// ------------------------------------------------------------------------------------------------
// inverse, barrier-free form (N <= 2048, hop | N with N / hop in {2, 4}, hop / 2 a multiple of 64): ONE WAVE walks C
// consecutive hop-blocks of one source on its own.  It transforms the frames that touch its blocks in increasing
// order (the R - 1 frames before the first block are transformed by this wave and by its left neighbour) and keeps the
// R hop-blocks a frame can touch in REGISTERS -- a lane owns hop / 128 sample pairs of every block -- so the windowed
// overlap-add (frame order = the reference's accumulation order, transform.py:381-389) is R packed multiply-adds per
// owned pair, and a block is normalised by sum(w^2) and stored (8-byte stores, 512 contiguous bytes per instruction)
// as soon as the frame of its own index has been added.  No ring in LDS, no workgroup barrier after the window table
// is staged, no flush loop: the four waves of a workgroup (the four sources of one chunk, which share the phasor
// rows) never wait for each other.  istft_wave_kernel above: 2 barriers per 4 frames, 16 LDS read-modify-writes per
// thread and step, ~300 scalar instructions per frame of 64-bit flush bookkeeping.  Measured at 4096 tiles (N = 2048,
// 4 sources): 0.302 -> 0.240 ms; a 16 x 32-tile launch group 56.7 -> 49.1 us.
// ------------------------------------------------------------------------------------------------
// STAGE (four sources per clip, phasor input, rows of M + 4 bins on 16-byte boundaries): the spectra come in through LDS.
// Measured on the plain form (profiles/r03_e_istft_ablation.txt): of 0.224 ms at 4096 tiles 0.090 are its 64 four- and
// eight-byte global loads per frame and source (0.134 with none of them), 0.057 the phasor loads alone.  Here a wave requests
// the NEXT frame's magnitude row (its own source) and its quarter of the phasor row (shared by the four sources = the four
// waves of the workgroup) as 16-byte-per-lane global_load_lds transfers -- 8 instead of 64 vector-memory instructions per
// frame, no registers, a whole FFT ahead of their use -- and reads them back lane by lane from LDS.  Costs one workgroup
// barrier per frame: the four waves walk the same frames, which they do anyway (same chunk, same clip).
template <int LOG2M, bool UNIT, int R, int STAGE = 0 /* 1: magnitudes and phasors through LDS */>
__global__ __launch_bounds__(256, 2) void istft_seq_kernel(const float* __restrict__ mag, int64_t src_stride,
                                                          const float* __restrict__ phase, const float2* __restrict__ unit,
                                                          int64_t ld, const float* __restrict__ win,
                                                          const float* __restrict__ wsq, const float2* __restrict__ tw,
                                                          float* __restrict__ audio, int64_t n_out, int64_t T, int C,
                                                          int64_t n_blocks, int n_chunks, int n_src, float pre_div,
                                                          float sqrt_n, int64_t unit_clip_stride, int src_per_clip,
                                                          const int64_t* __restrict__ clip_tab, int64_t out_stride) {
    constexpr int M = 1 << LOG2M, N = 2 * M, P = M / 64, MP = M + M / 32;
    constexpr int hop = N / R, hp = hop / 2, VPB = hp / 64;   // sample pairs a lane owns in one hop-block
    constexpr int R1 = Plan<LOG2M>::R1, NB1 = P / R1, stride1 = 64 * NB1;
    static_assert(WaveTw<LOG2M>::REG3 && VPB * R == P && VPB >= 1, "lean twiddles, whole pairs per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* winl = reinterpret_cast<float2*>(smem);          // [M] window pairs times 1/M
    float2* fbuf = winl + M;                                 // [4][MP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* buf = fbuf + wave * MP;
    // STAGE: [2][M + 4] phasor rows (frame parity), then [4][M + 4] magnitude rows (one per wave); all 16-byte aligned
    float2* sunit = fbuf + 4 * MP;
    float* smag = reinterpret_cast<float*>(sunit + 2 * (M + 4)) + wave * (M + 4);
    const float inv_m = 1.f / (float)M;                      // a power of two: (z / M) * w == z * (w / M) exactly
    {
        const float2* w2g = reinterpret_cast<const float2*>(win);
        for (int k = tid; k < M; k += 256) stc(winl + k, ldc(w2g + k) * inv_m);
    }
    __syncthreads();
    // wave -> (chunk, source): the sources of a chunk are neighbouring waves and read the same phasor rows
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int s = (int)(gw % n_src);
    const int64_t chunk = gw / n_src;
    if (chunk >= n_chunks) return;
    if (clip_tab) {   // clips of different lengths: this source's own sample / frame / hop-block counts
        const int64_t c = s / src_per_clip;
        n_out = clip_tab[kDcsClipTab * c];
        T = clip_tab[kDcsClipTab * c + 1];
        n_blocks = (n_out + M + hop - 1) / hop;
    }
    const int64_t hb0 = chunk * C;
    if (hb0 >= n_blocks) return;
    const int64_t hb1 = (hb0 + C < n_blocks) ? hb0 + C : n_blocks;
    const int64_t n_first = (hb0 < R) ? 0 : hb0 - (R - 1);
    const int64_t n_last = (hb1 - 1 < T - 1) ? hb1 - 1 : T - 1;
    const float amp = 0.5f * (sqrt_n / pre_div);  // (mag / scale_factor) sqrt(N), and the 1/2 of the even/odd split
    const float* msrc = mag + (int64_t)s * src_stride;
    {
        const int64_t c_ = s / src_per_clip;
        const int64_t r0_ = clip_tab ? clip_tab[kDcsClipTab * c_ + 3] : -1;
        unit += r0_ >= 0 ? r0_ * ld : c_ * unit_clip_stride;
    }
    float* dst = audio + (int64_t)s * out_stride;
    const bool dst_al = (reinterpret_cast<uintptr_t>(dst) & 7) == 0;   // sample pairs start at even offsets
    WaveTw<LOG2M> wt;
    wt.init(tw, lane);
    const cx wl = ldc(tw + lane);
    // steady-state normaliser of this lane's pairs: 1 / sum over the R frames that cover a block
    cx nrm[VPB];
#pragma unroll
    for (int v = 0; v < VPB; ++v) {
        const int q = 2 * (lane + 64 * v);
        float nx = 0.f, ny = 0.f;
#pragma unroll
        for (int d = R - 1; d >= 0; --d) {   // frames in increasing order
            nx += wsq[q + d * hop];
            ny += wsq[q + 1 + d * hop];
        }
        nrm[v] = mk(nx == 0.f ? 1.f : 1.f / nx, ny == 0.f ? 1.f : 1.f / ny);
    }
    cx acc[R][VPB];
#pragma unroll
    for (int d = 0; d < R; ++d)
#pragma unroll
        for (int v = 0; v < VPB; ++v) acc[d][v] = mk(0.f, 0.f);

    const float2* wlane = winl + lane;
    const float2* blane = buf + pad(lane);
    typedef float f4 __attribute__((ext_vector_type(4)));
    // frame n's rows -> LDS: this wave's magnitude row (M + 4 floats = 4 full transfers + one 16-byte piece) and pieces
    // 128 wave .. 128 wave + 127 of the phasor row (2 transfers; wave 3 also the last two pieces)
    auto request = [&](int64_t n) {
        const f4* m4 = reinterpret_cast<const f4*>(msrc + n * ld);
        f4* sm4 = reinterpret_cast<f4*>(smag);
#pragma unroll
        for (int c = 0; c < M / 256; ++c) __builtin_amdgcn_global_load_lds(m4 + 64 * c + lane, sm4 + 64 * c, 16, 0, 0);
        if (lane == 0) __builtin_amdgcn_global_load_lds(m4 + M / 4, sm4 + M / 4, 16, 0, 0);
        const f4* u4 = reinterpret_cast<const f4*>(unit + n * ld);
        f4* su4 = reinterpret_cast<f4*>(sunit + (int)(n & 1) * (M + 4));
        constexpr int UQ = M / 2 / 4;                         // 16-byte pieces per wave: M / 2 of the M / 2 * 4 + 2
#pragma unroll
        for (int c = 0; c < UQ / 64; ++c)
            __builtin_amdgcn_global_load_lds(u4 + wave * UQ + 64 * c + lane, su4 + wave * UQ + 64 * c, 16, 0, 0);
        if (wave == 3 && lane < 2) __builtin_amdgcn_global_load_lds(u4 + 4 * UQ + lane, su4 + 4 * UQ, 16, 0, 0);
    };
    if (STAGE && n_first <= n_last) request(n_first);
    for (int64_t nb = (n_first / R) * R; nb < hb1; nb += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int64_t n = nb + j;
            if (n >= n_first && n <= n_last) {       // wave-uniform
                const float* mrow = msrc + n * ld;
                const float2* urow = unit + n * ld;
                const float* prow = phase + n * ld;
                if (STAGE) {
                    // the rows requested one frame ago have landed, here and -- behind the barrier -- in the other three waves;
                    // the barrier also says that everybody is done reading frame n - 1's phasors.  (Waiting only down to the
                    // VPB block stores issued behind the request measured the same: 0.1974 vs 0.1980 ms at 4096 tiles.)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                const float2* su = sunit + (int)(n & 1) * (M + 4);
                cx v[P];
#pragma unroll
                for (int b = 0; b < NB1; ++b)
#pragma unroll
                    for (int tt = 0; tt < R1; ++tt) {
                        const int k = lane + 64 * b + tt * stride1;  // 0 <= k < M
                        const int km = M - k;                         // 1..M
                        const float a = (STAGE ? smag[k] : mrow[k]) * amp;
                        const float b2 = (STAGE ? smag[km] : mrow[km]) * amp;
                        cx xk, xm;
                        if (UNIT) {
                            xk = (STAGE == 1 ? ldc(su + k) : ldc(urow + k)) * a;
                            xm = (STAGE == 1 ? ldc(su + km) : ldc(urow + km)) * b2;
                        } else {
                            float sn, cs;
                            sincosf(prow[k], &sn, &cs);
                            xk = mk(a * cs, a * sn);
                            sincosf(prow[km], &sn, &cs);
                            xm = mk(b2 * cs, b2 * sn);
                        }
                        if (b == 0 && tt == 0) {  // k == 0 only there: imaginary parts of DC / Nyquist are ignored (numpy irfft)
                            if (lane == 0) {
                                xk[1] = 0.f;
                                xm[1] = 0.f;
                            }
                        }
                        const cx e = c_add_conj(xk, xm), d = c_sub_conj(xk, xm);
                        const int jj = (b + tt * NB1) * (1024 / M);   // k = lane + 64 j: w^k = w^lane * exp(-i pi 64 j / M)
                        const cx wk = jj == 0 ? wl : c_mul(wl, w_pi16(jj));
                        const cx o = c_mul_conj(d, wk);
                        v[b * R1 + tt] = c_add_i(e, o);
                    }
                if (STAGE) {
                    if (n + 1 <= n_last) {   // this wave's reads of its magnitude row are complete before it is overwritten
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        request(n + 1);
                    }
                }
                fft_wave<LOG2M, +1>(v, lane, wt, tw, buf);
                // frame n covers the hop-blocks n .. n + R - 1: block n + d lives in register slot (j + d) % R
#pragma unroll
                for (int d = 0; d < R; ++d)
#pragma unroll
                    for (int vv = 0; vv < VPB; ++vv) {
                        const int k = d * hp + 64 * vv;           // + lane; constant part a multiple of 64
                        const cx z = ldc(blane + cpad(k));
                        acc[(j + d) % R][vv] += z * ldc(wlane + k);
                    }
            }
            // block g = n has now received every frame that touches it
            const int64_t g = n;
            if (g >= hb0 && g < hb1) {
                const bool steady = g >= R - 1 && g <= T - 1;
                const int64_t m0 = g * hop - M;                   // output index of the block's first sample
                if (steady && dst_al && m0 >= 0 && m0 + hop <= n_out) {
                    float* op = dst + m0 + 2 * lane;
#pragma unroll
                    for (int vv = 0; vv < VPB; ++vv) *reinterpret_cast<cx*>(op + 128 * vv) = acc[j][vv] * nrm[vv];
                } else {
                    const int64_t f_hi = g < T - 1 ? g : T - 1;
                    const int64_t f_lo = g < R ? 0 : g - (R - 1);
#pragma unroll
                    for (int vv = 0; vv < VPB; ++vv) {
                        const int q = 2 * (lane + 64 * vv);
                        const int64_t m = m0 + q;
                        if (m + 1 < 0 || m >= n_out) continue;
                        float nx = 0.f, ny = 0.f;
                        for (int64_t f = f_lo; f <= f_hi; ++f) {
                            nx += wsq[(g - f) * hop + q];
                            ny += wsq[(g - f) * hop + q + 1];
                        }
                        if (nx == 0.f) nx = 1.f;
                        if (ny == 0.f) ny = 1.f;
                        if (m >= 0) dst[m] = steady ? acc[j][vv][0] * nrm[vv][0] : acc[j][vv][0] / nx;
                        if (m + 1 < n_out) dst[m + 1] = steady ? acc[j][vv][1] * nrm[vv][1] : acc[j][vv][1] / ny;
                    }
                }
            }
#pragma unroll
            for (int vv = 0; vv < VPB; ++vv) acc[j][vv] = mk(0.f, 0.f);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// inverse, chained form (round 5): the barrier-free kernel above gives every wave C hop-blocks AND the R - 1 frames in front
// of them to warm its accumulators up -- at the launch-group sizes of a batch-of-files server (C = 8, R = 4) three of a wave's
// eleven frames, and a launch lasts as long as its longest chain of frames.  Here the NW = 8 waves of a workgroup walk NW
// CONSECUTIVE runs of LC frames of ONE source: only the first wave of a workgroup warms up (its R - 1 first blocks belong to
// the workgroup on its left), every other wave starts cold and leaves the R - 1 blocks it cannot complete -- its "head" --
// as partial sums in LDS, and the wave on its left, whose last R - 1 accumulators -- its "tail" -- are the missing halves,
// adds the two after ONE workgroup barrier at the end: block = tail + head, i.e. (frames g-3.. of the left wave, added in
// order) + (frames .. g of the right wave, added in order).  The reference adds the frames of a block strictly in order
// (transform.py:381-389); here the R - 1 blocks at each of the NW - 1 seams of a workgroup add two in-order partial sums
// instead -- the same float32 error class (one rounding per addend either way), not the same bits; every other block is
// accumulated exactly as before.  A chain is LC frames instead of C + R - 1 for the same number of resident waves: 8 against 11
// at 20 x 32 tiles, 12 against 15 at 32 x 32.  Costs (R - 1) * hop floats of LDS per seam (118 KB per workgroup at N = 2048:
// one workgroup of 8 waves per CU -- the register budget of 2 waves per SIMD allows no more anyway).
// Phasor input only (the fused paths); frames of R in {2, 4} hop-blocks, N <= 2048, like the kernel above.
// ------------------------------------------------------------------------------------------------
constexpr int kChainWaves = 8;
template <int LOG2M, int R>
__global__ __launch_bounds__(64 * kChainWaves) void istft_chain_kernel(const float* __restrict__ mag, int64_t src_stride,
                                                                        const float2* __restrict__ unit, int64_t ld,
                                                                        const float* __restrict__ win, const float* __restrict__ wsq,
                                                                        const float2* __restrict__ tw, float* __restrict__ audio,
                                                                        int64_t n_out, int64_t T, int LC, int G /* workgroups per source */,
                                                                        int64_t n_blocks, int64_t n_units, float pre_div, float sqrt_n,
                                                                        int64_t unit_clip_stride, int src_per_clip,
                                                                        const int64_t* __restrict__ clip_tab, int64_t out_stride) {
    constexpr int M = 1 << LOG2M, N = 2 * M, P = M / 64, MP = M + M / 32, NW = kChainWaves;
    constexpr int hop = N / R, hp = hop / 2, VPB = hp / 64;   // sample pairs a lane owns in one hop-block
    constexpr int R1 = Plan<LOG2M>::R1, NB1 = P / R1, stride1 = 64 * NB1;
    static_assert(WaveTw<LOG2M>::REG3 && VPB * R == P && VPB >= 1, "lean twiddles, whole pairs per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* winl = reinterpret_cast<float2*>(smem);          // [M] window pairs times 1/M
    float2* fbuf = winl + M;                                 // [NW][MP]
    float2* heads = fbuf + NW * MP;                          // [NW - 1][R - 1][hp]: wave w's head blocks at (w - 1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* buf = fbuf + wave * MP;
    const float inv_m = 1.f / (float)M;
    {
        const float2* w2g = reinterpret_cast<const float2*>(win);
        for (int k = tid; k < M; k += 64 * NW) stc(winl + k, ldc(w2g + k) * inv_m);
    }
    // workgroup -> (unit = (clip, run of the clip), source of the clip): the sources of a unit read the same phasor rows; they get
    // block ids 8 apart (same XCD, dispatched together), so one of them pulls a row from HBM and the others hit that L2
    int64_t u;
    int s4;
    {
        const int64_t b = blockIdx.x, full = (n_units >> 3) * 8 * src_per_clip;
        if (b < full) {
            u = (b / (8 * src_per_clip)) * 8 + (b & 7);
            s4 = (int)((b >> 3) % src_per_clip);
        } else {
            const int64_t r = n_units & 7, b2 = b - full;
            u = (n_units >> 3) * 8 + b2 % r;
            s4 = (int)(b2 / r);
        }
    }
    const int64_t clip = u / G;
    const int jg = (int)(u - clip * G);
    const int64_t s = clip * src_per_clip + s4;
    if (clip_tab) {   // clips of different lengths: this source's own sample / frame / hop-block counts
        n_out = clip_tab[kDcsClipTab * clip];
        T = clip_tab[kDcsClipTab * clip + 1];
        n_blocks = (n_out + M + hop - 1) / hop;
    }
    const int64_t Sb = (int64_t)NW * LC - (R - 1);           // blocks a workgroup owns (the first one of a source R - 1 more)
    const int64_t wg0 = (int64_t)jg * Sb;                     // first frame slot of the workgroup
    const int64_t blk0 = jg == 0 ? 0 : wg0 + (R - 1);         // first block it owns
    if (blk0 >= n_blocks) return;                             // workgroup-uniform, before any barrier
    __syncthreads();
    const int64_t f0 = wg0 + (int64_t)wave * LC, f1 = f0 + LC;   // this wave's frame slots [f0, f1)
    const int64_t n_last = T - 1;
    const float amp = 0.5f * (sqrt_n / pre_div);
    const float* msrc = mag + s * src_stride;
    {
        const int64_t r0_ = clip_tab ? clip_tab[kDcsClipTab * clip + 3] : -1;
        unit += r0_ >= 0 ? r0_ * ld : clip * unit_clip_stride;
    }
    float* dst = audio + s * out_stride;
    const bool dst_al = (reinterpret_cast<uintptr_t>(dst) & 7) == 0;
    WaveTw<LOG2M> wt;
    wt.init(tw, lane);
    const cx wl = ldc(tw + lane);
    cx nrm[VPB];
#pragma unroll
    for (int v = 0; v < VPB; ++v) {
        const int q = 2 * (lane + 64 * v);
        float nx = 0.f, ny = 0.f;
#pragma unroll
        for (int d = R - 1; d >= 0; --d) {   // frames in increasing order
            nx += wsq[q + d * hop];
            ny += wsq[q + 1 + d * hop];
        }
        nrm[v] = mk(nx == 0.f ? 1.f : 1.f / nx, ny == 0.f ? 1.f : 1.f / ny);
    }
    cx acc[R][VPB];
#pragma unroll
    for (int d = 0; d < R; ++d)
#pragma unroll
        for (int v = 0; v < VPB; ++v) acc[d][v] = mk(0.f, 0.f);
    const float2* wlane = winl + lane;
    const float2* blane = buf + pad(lane);
    // block g is complete: normalise by the frames that reach it and store (the steady-state blocks multiply by the reciprocal
    // kept in registers, the first / last R - 1 blocks of a signal divide by their own sum -- as in the kernel above)
    auto store_block = [&](int64_t g, const cx (&blk)[VPB]) {
        const bool steady = g >= R - 1 && g <= T - 1;
        const int64_t m0 = g * hop - M;
        if (steady && dst_al && m0 >= 0 && m0 + hop <= n_out) {
            float* op = dst + m0 + 2 * lane;
#pragma unroll
            for (int vv = 0; vv < VPB; ++vv) *reinterpret_cast<cx*>(op + 128 * vv) = blk[vv] * nrm[vv];
            return;
        }
        const int64_t f_hi = g < T - 1 ? g : T - 1;
        const int64_t f_lo = g < R ? 0 : g - (R - 1);
#pragma unroll
        for (int vv = 0; vv < VPB; ++vv) {
            const int q = 2 * (lane + 64 * vv);
            const int64_t m = m0 + q;
            if (m + 1 < 0 || m >= n_out) continue;
            float nx = 0.f, ny = 0.f;
            for (int64_t f = f_lo; f <= f_hi; ++f) {
                nx += wsq[(g - f) * hop + q];
                ny += wsq[(g - f) * hop + q + 1];
            }
            if (nx == 0.f) nx = 1.f;
            if (ny == 0.f) ny = 1.f;
            if (m >= 0) dst[m] = steady ? blk[vv][0] * nrm[vv][0] : blk[vv][0] / nx;
            if (m + 1 < n_out) dst[m + 1] = steady ? blk[vv][1] * nrm[vv][1] : blk[vv][1] / ny;
        }
    };
    for (int64_t nb = (f0 / R) * R; nb < f1; nb += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int64_t n = nb + j;
            if (n < f0 || n >= f1) continue;        // wave-uniform; slots outside the wave's run leave the accumulators alone
            if (n <= n_last) {
                const float* mrow = msrc + n * ld;
                const float2* urow = unit + n * ld;
                cx v[P];
#pragma unroll
                for (int b = 0; b < NB1; ++b)
#pragma unroll
                    for (int tt = 0; tt < R1; ++tt) {
                        const int k = lane + 64 * b + tt * stride1;  // 0 <= k < M
                        const int km = M - k;                         // 1..M
                        const float a = mrow[k] * amp;
                        const float b2 = mrow[km] * amp;
                        cx xk = ldc(urow + k) * a;
                        cx xm = ldc(urow + km) * b2;
                        if (b == 0 && tt == 0) {  // k == 0 only there: imaginary parts of DC / Nyquist are ignored (numpy irfft)
                            if (lane == 0) {
                                xk[1] = 0.f;
                                xm[1] = 0.f;
                            }
                        }
                        const cx e = c_add_conj(xk, xm), d = c_sub_conj(xk, xm);
                        const int jj = (b + tt * NB1) * (1024 / M);
                        const cx wk = jj == 0 ? wl : c_mul(wl, w_pi16(jj));
                        const cx o = c_mul_conj(d, wk);
                        v[b * R1 + tt] = c_add_i(e, o);
                    }
                fft_wave<LOG2M, +1>(v, lane, wt, tw, buf);
                // frame n covers the hop-blocks n .. n + R - 1: block g lives in register slot g % R = (j + d) % R
#pragma unroll
                for (int d = 0; d < R; ++d)
#pragma unroll
                    for (int vv = 0; vv < VPB; ++vv) {
                        const int k = d * hp + 64 * vv;
                        const cx z = ldc(blane + cpad(k));
                        acc[(j + d) % R][vv] += z * ldc(wlane + k);
                    }
            }
            // block g = n has received every frame of THIS wave that touches it
            const int64_t g = n;
            const bool head = g < f0 + (R - 1) && f0 > 0;            // frames before f0 reach it too (f0 == 0: nothing before)
            if (!head) {
                if (g < n_blocks) store_block(g, acc[j]);
            } else if (wave > 0) {                                    // the left wave completes it after the barrier
                float2* hb = heads + ((wave - 1) * (R - 1) + (int)(g - f0)) * hp + lane;
#pragma unroll
                for (int vv = 0; vv < VPB; ++vv) stc(hb + 64 * vv, acc[j][vv]);
            }                                                         // wave 0 of a later workgroup: warm-up, the left workgroup owns it
#pragma unroll
            for (int vv = 0; vv < VPB; ++vv) acc[j][vv] = mk(0.f, 0.f);
        }
    }
    __syncthreads();
    if (wave < NW - 1) {   // the seam with the wave on the right: blocks f1 .. f1 + R - 2 = my tail + its head
#pragma unroll
        for (int d = 0; d < R - 1; ++d) {
            const int64_t g = f1 + d;
            if (g >= n_blocks) break;
            const float2* hb = heads + (wave * (R - 1) + d) * hp + lane;
            cx sum[VPB];
            const int slot = (int)(g % R);
#pragma unroll
            for (int q = 0; q < R; ++q)
                if (q == slot) {
#pragma unroll
                    for (int vv = 0; vv < VPB; ++vv) sum[vv] = acc[q][vv] + ldc(hb + 64 * vv);
                }
            store_block(g, sum);
        }
    }
}

template <int LOG2M>
int launch_fwd(dcs_stft* p, const float* audio, int64_t L, int64_t audio_stride, int64_t n_clips, float* mag,
               float* phase, float2* unit, int64_t ld, int64_t rows_out, int64_t T, bool interleave,
               const int64_t* clip_tab) {
    constexpr int M = 1 << LOG2M, MP = M + M / 32;
    // frames per workgroup: 4 when there are plenty of frames, 1 to spread a short signal over the CUs (1 / 2 / 3 per
    // workgroup at 640 tiles: 33.1 / 27.5 / 22.4 us against 22.0 with 4)
    const int64_t rows_all = rows_out * n_clips;
    const int fpw = rows_all >= 8 * (int64_t)p->ctx->n_cu ? 4 : 1;
    const size_t lds = LOG2M <= 10 ? ((size_t)(M / 2 + 2) + (size_t)fpw * MP) * sizeof(float2)       // quarter-circle table, no staging area
                                   : ((size_t)(M + 1) + (size_t)fpw * (MP + kStftStageF2)) * sizeof(float2);
    auto kern = stft_forward_wave_kernel<LOG2M>;
    if (lds > 48 * 1024)
        DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)dcs_cdiv(rows_all, fpw)), dim3(64 * fpw), lds, p->ctx->stream, audio, L,
                       p->win_f, p->tw_f, mag, phase, unit, ld, p->hop, T, rows_out, n_clips, audio_stride,
                       (float)sqrt((double)p->frame), interleave ? 1 : 0, clip_tab);
    return DCS_OK;
}

template <int LOG2M>
int launch_inv(dcs_stft* p, const float* mag, int64_t src_stride, const float* phase, const float2* unit,
               int64_t unit_clip_stride, int src_per_clip, int64_t ld, int64_t T, int n_src, float pre_div, float* audio,
               int64_t n_out, const int64_t* clip_tab, int64_t out_stride) {
    constexpr int M = 1 << LOG2M, MP = M + M / 32, N = 2 * M;
    const int hop = p->hop, R_ = N / hop;
    const int64_t n_blocks = (n_out + N / 2 + hop - 1) / hop;
    int ring_slots = 8;
    while (ring_slots < R_ + 3) ring_slots *= 2;
    const size_t lds = ((size_t)(M + 1) + 4 * (size_t)MP + (size_t)M + (size_t)ring_slots * (hop / 2)) * sizeof(float2) +
                       (size_t)hop * sizeof(float);
    // barrier-free wave-sequential kernel: N <= 2048, frames of 2 or 4 hop-blocks, hop / 2 a multiple of 64
    if constexpr (LOG2M <= 10) {
        if ((R_ == 2 || R_ == 4) && (hop / 2) % 64 == 0 && R_ * (hop / 2) == M) {
            static const int c_seq = getenv("DCS_ISTFT_SEQ_HOPS") ? atoi(getenv("DCS_ISTFT_SEQ_HOPS")) : 0;
            const int64_t total_s = n_blocks * n_src;
            // 2 waves per SIMD: the kernel needs ~240 registers (the R blocks in flight, the pass twiddles, a frame's points);
            // sized for 3 waves it spills 46 of them and runs 2.2x slower (0.52 vs 0.24 ms at 4096 tiles)
            const int64_t resident_w = (int64_t)p->ctx->n_cu * 4 * 2;
            int64_t Cs = (total_s + resident_w - 1) / resident_w;
            if (Cs < 1) Cs = 1;
            if (c_seq > 0) Cs = c_seq;
            const int n_chunks_s = (int)((n_blocks + Cs - 1) / Cs);
            const size_t lds_s = ((size_t)M + 4 * (size_t)MP) * sizeof(float2);
            const dim3 grid_s((unsigned)(((int64_t)n_chunks_s * n_src + 3) / 4));
            // spectra through LDS (istft_seq_kernel<..., true>): four sources per clip in one workgroup, phasor input, rows of
            // M + 4 bins whose starts are 16-byte aligned; DCS_ISTFT_STAGE=0 switches it off
            static const int stage_env = getenv("DCS_ISTFT_STAGE") ? atoi(getenv("DCS_ISTFT_STAGE")) : 1;
            const int spc = src_per_clip > 0 ? src_per_clip : n_src;
            // measured (profiles/r03_e_istft_stage_sweep.txt): 4096 tiles 0.229 -> 0.198 ms, 2048 tiles 0.129 -> 0.113, but 1024
            // tiles 0.063 -> 0.069 and 640 tiles 0.048 -> 0.052 -- with few hop-blocks per wave the chip's memory system is not
            // loaded and the plain loads cost less than the LDS round trip: staged from 16 hop-blocks per wave on
            static const int stage_min = getenv("DCS_ISTFT_STAGE_MIN") ? atoi(getenv("DCS_ISTFT_STAGE_MIN")) : 16;
            const bool stage = stage_env && Cs >= stage_min && unit && ld == M + 4 && n_src % 4 == 0 && spc % 4 == 0 && (src_stride & 3) == 0 &&
                               (unit_clip_stride & 1) == 0 &&
                               ((reinterpret_cast<uintptr_t>(mag) | reinterpret_cast<uintptr_t>(unit)) & 15) == 0;
            const size_t lds_stage = lds_s + (2 * (size_t)(M + 4)) * sizeof(float2) + 4 * (size_t)(M + 4) * sizeof(float);
            // chained form (istft_chain_kernel): when the staged form does not apply and its chain of frames is strictly shorter
            // than this kernel's Cs + R - 1.  LC = the fewest frames per wave for which every workgroup of the launch is resident
            // at once (one workgroup of 8 waves per CU at N = 2048).
            static const int chain_env = getenv("DCS_ISTFT_CHAIN") ? atoi(getenv("DCS_ISTFT_CHAIN")) : 1;   // 0: off; > 1: LC
            if (chain_env && unit && !stage && n_src % spc == 0) {
                const size_t lds_c = ((size_t)M + kChainWaves * (size_t)MP + (size_t)(kChainWaves - 1) * (R_ - 1) * (hop / 2)) * sizeof(float2);
                const int64_t n_clips_c = n_src / spc;
                // workgroups of 8 waves the chip holds at once: registers (230 .. 254 at N = 2048: one per CU) and LDS decide
                static int wg_cache[2][64];                          // [R == 4][device]; 0 = not asked yet (a race writes the same value twice)
                const int dev_i = p->ctx->device >= 0 && p->ctx->device < 64 ? p->ctx->device : 0;
                int wg_per_cu = wg_cache[R_ == 4][dev_i];
                if (wg_per_cu == 0) {
                    const hipError_t e = R_ == 4
                        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, istft_chain_kernel<LOG2M, 4>, 64 * kChainWaves, lds_c)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, istft_chain_kernel<LOG2M, 2>, 64 * kChainWaves, lds_c);
                    if (e != hipSuccess || wg_per_cu < 1) { (void)hipGetLastError(); wg_per_cu = 1; }
                    wg_cache[R_ == 4][dev_i] = wg_per_cu;
                }
                const int64_t resident_wg = (int64_t)p->ctx->n_cu * wg_per_cu;
                int64_t LC = R_ - 1 > 1 ? R_ - 1 : 1, Gc = 0;
                for (;; ++LC) {
                    const int64_t Sb = (int64_t)kChainWaves * LC - (R_ - 1);
                    Gc = n_blocks <= R_ - 1 ? 1 : (n_blocks - (R_ - 1) + Sb - 1) / Sb;
                    if (Gc * n_src <= resident_wg || LC >= Cs + R_ - 1) break;
                }
                if (chain_env > 1) {
                    LC = chain_env < R_ - 1 ? R_ - 1 : chain_env;
                    const int64_t Sb = (int64_t)kChainWaves * LC - (R_ - 1);
                    Gc = n_blocks <= R_ - 1 ? 1 : (n_blocks - (R_ - 1) + Sb - 1) / Sb;
                }
                if ((LC < Cs + R_ - 1 || chain_env > 1) && lds_c <= 160 * 1024 && Gc * n_src < 0x7fffffff) {
                    const int64_t n_units = n_clips_c * Gc;
#define DCS_CHAIN(R__)                                                                                               \
                    {                                                                                                \
                        auto kern = istft_chain_kernel<LOG2M, R__>;                                                  \
                        DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                             \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));        \
                        hipLaunchKernelGGL(kern, dim3((unsigned)(n_units * spc)), dim3(64 * kChainWaves), lds_c, p->ctx->stream, mag, \
                                           src_stride, unit, ld, p->win_f, p->wsq_f, p->tw_f, audio, n_out, T, (int)LC, (int)Gc, \
                                           n_blocks, n_units, pre_div, (float)sqrt((double)N), unit_clip_stride, spc, clip_tab, \
                                           out_stride > 0 ? out_stride : n_out);                                     \
                    }
                    if (R_ == 4) DCS_CHAIN(4) else DCS_CHAIN(2)
#undef DCS_CHAIN
                    return DCS_OK;
                }
            }
#define DCS_SEQ_STAGE(R__)                                                                                           \
            {                                                                                                        \
                auto kern = istft_seq_kernel<LOG2M, true, R__, 1>;                                                   \
                DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stage));            \
                hipLaunchKernelGGL(kern, grid_s, dim3(256), lds_stage, p->ctx->stream, mag, src_stride, phase, unit, ld, \
                                   p->win_f, p->wsq_f, p->tw_f, audio, n_out, T, (int)Cs, n_blocks, n_chunks_s, n_src, \
                                   pre_div, (float)sqrt((double)N), unit_clip_stride, spc, clip_tab,                 \
                                   out_stride > 0 ? out_stride : n_out);                                             \
            }
            if (stage) {
                if (R_ == 4) DCS_SEQ_STAGE(4) else DCS_SEQ_STAGE(2)
                return DCS_OK;
            }
#undef DCS_SEQ_STAGE
#define DCS_SEQ(UNIT_, R__)                                                                                          \
            {                                                                                                        \
                auto kern = istft_seq_kernel<LOG2M, UNIT_, R__>;                                                     \
                if (lds_s > 48 * 1024)                                                                               \
                    DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));            \
                hipLaunchKernelGGL(kern, grid_s, dim3(256), lds_s, p->ctx->stream, mag, src_stride, phase, unit, ld,  \
                                   p->win_f, p->wsq_f, p->tw_f, audio, n_out, T, (int)Cs, n_blocks, n_chunks_s, n_src, \
                                   pre_div, (float)sqrt((double)N), unit_clip_stride,                                \
                                   src_per_clip > 0 ? src_per_clip : n_src, clip_tab, out_stride > 0 ? out_stride : n_out); \
            }
            if (unit) { if (R_ == 4) DCS_SEQ(true, 4) else DCS_SEQ(true, 2) }
            else { if (R_ == 4) DCS_SEQ(false, 4) else DCS_SEQ(false, 2) }
#undef DCS_SEQ
            return DCS_OK;
        }
    }
    if (lds > 160 * 1024) DCS_FAIL(DCS_EUNSUPPORTED, "wave iSTFT: %zu bytes of LDS", lds);
    // C hop-blocks per workgroup (C + R_ - 1 frames are transformed, 4 per step): one round of resident
    // workgroups, all of the same length.  Measured on MI355X, N = 2048: 4096 tiles C = 161 0.384 ms, 81 0.403,
    // 41 0.425, 9 0.548; 32 tiles (752 blocks) C = 2 18.1 us, 1 22.3, 5 19.4.
    const int64_t total = n_blocks * n_src;
    const int64_t resident = (int64_t)p->ctx->n_cu * (lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3));
    int64_t C = (total + resident - 1) / resident;
    if (C >= 8) C = (C + R_ - 1 + 3) / 4 * 4 - (R_ - 1);  // C + R_ - 1 a multiple of 4: no half-empty last step
    if (C < 1) C = 1;
    const int n_chunks = (int)((n_blocks + C - 1) / C);
    const dim3 grid((unsigned)n_chunks * (unsigned)n_src);
#define DCS_GO(UNIT_)                                                                                             \
    {                                                                                                             \
        auto kern = istft_wave_kernel<LOG2M, UNIT_>;                                                              \
        if (lds > 48 * 1024)                                                                                      \
            DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, p->ctx->stream, mag, src_stride, phase, unit, ld, p->win_f, \
                           p->wsq_f, p->tw_f, audio, n_out, hop, T, (int)C, n_blocks, n_chunks, n_src, ring_slots,  \
                           pre_div, (float)sqrt((double)N), unit_clip_stride, src_per_clip > 0 ? src_per_clip : n_src, \
                           clip_tab, out_stride > 0 ? out_stride : n_out);                                         \
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

bool dcs_fft_wave_inverse_supported(const dcs_stft* p) {
    // hop | N and hop even (every reference configuration): a frame is N/hop whole hop-blocks of sample pairs
    return dcs_fft_wave_supported(p) && p->hop > 0 && (p->frame % p->hop) == 0 && (p->hop & 1) == 0;
}

int dcs_fft_wave_forward(dcs_stft* p, const float* audio, int64_t L, int64_t audio_stride, int64_t n_clips, float* mag,
                         float* phase, float2* unit, int64_t ld, int64_t rows_out, int64_t T, bool interleave,
                         const int64_t* clip_tab) {
    switch (p->log2m) {
        case 9: return launch_fwd<9>(p, audio, L, audio_stride, n_clips, mag, phase, unit, ld, rows_out, T, interleave, clip_tab);
        case 10: return launch_fwd<10>(p, audio, L, audio_stride, n_clips, mag, phase, unit, ld, rows_out, T, interleave, clip_tab);
        case 11: return launch_fwd<11>(p, audio, L, audio_stride, n_clips, mag, phase, unit, ld, rows_out, T, interleave, clip_tab);
    }
    DCS_FAIL(DCS_EUNSUPPORTED, "wave FFT: frame size");
}

int dcs_fft_wave_inverse(dcs_stft* p, const float* mag, int64_t src_stride, const float* phase, const float2* unit,
                         int64_t unit_clip_stride, int src_per_clip, int64_t ld, int64_t T, int n_src, float pre_div,
                         float* audio, int64_t n_out, const int64_t* clip_tab, int64_t out_stride) {
    switch (p->log2m) {
        case 9: return launch_inv<9>(p, mag, src_stride, phase, unit, unit_clip_stride, src_per_clip, ld, T, n_src, pre_div, audio, n_out, clip_tab, out_stride);
        case 10: return launch_inv<10>(p, mag, src_stride, phase, unit, unit_clip_stride, src_per_clip, ld, T, n_src, pre_div, audio, n_out, clip_tab, out_stride);
        case 11: return launch_inv<11>(p, mag, src_stride, phase, unit, unit_clip_stride, src_per_clip, ld, T, n_src, pre_div, audio, n_out, clip_tab, out_stride);
    }
    DCS_FAIL(DCS_EUNSUPPORTED, "wave FFT: frame size");
}

extern "C" DCS_API int fftw_trace_dump(unsigned long long* out, int n) {
#ifdef DCS_FFTW_TRACE
    if (!out || n < 64) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fftw_trace_buf), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
#else
    (void)out; (void)n;
    return -2;
#endif
}
