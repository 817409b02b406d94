// InverseLayer(conv2) + InverseLayer(conv1) of the Bach10 graph in ONE kernel with f32-class arithmetic
// (separate_bach10.py:219-227; the f16 switch OFF) -- round 4.
//
// colconv_deconv1_fused_kernel (colconv_wreg.hip) fuses the two transposed convolutions for the f16 switch: a wave keeps
// all 20 taps of the 20 x 1 filter in registers (160 VGPRs as f16) and nothing between the two layers reaches HBM.  As
// three bf16 planes the same weights are 480 registers, so the f32-class graph stayed on two kernels: the f32-MFMA column
// convolution (1.57 ms per 10 s clip, the last f32-MFMA convolution of the repository) and the transposed conv1 behind
// it (0.39 ms), with the 1.2 GB [n * 4, 30, 30, 505] intermediate written and read back between them.
//
// Here TWO waves share a column block of 16 x:
//   * wave `par` of the pair keeps the taps u = 2 k + par (k = 0 .. 9) as bf16 planes in registers (240 VGPRs), the pair
//     one copy of the block's input rows as pre-split bf16 planes in LDS ([row -1 .. 11][plane][K piece][x], rows -1 and 11
//     stay zero), filled once per block by both waves from the CHANNELS-LAST dense output (32 bytes per lane and (row, x):
//     generic.hip packs the dense weights with permuted columns when this kernel is planned);
//   * stage 1 (gather form, operands swapped as in the f16 kernel): for output row y the wave multiplies its tap k with
//     input row h = y - 19 + 2 k + par -- the LDS address is the only thing that depends on `par`, so both waves run the
//     same fully unrolled code; a (y, k) slot whose row is -1 or 11 for this wave meets the zero rows (120 slots per
//     wave and block for 110 products: 9 % of the stage);  every product is six bf16 MFMAs (three-way split operands:
//     f32-class), K = the 30 (32) input channels, two 16-channel halves of the output;
//   * the two partial rows are exchanged through LDS (each wave hands over the partial of the row its partner finishes,
//     one workgroup barrier per row pair; the workgroup IS the pair) and added: G[ci][x] of one row, complete, in the
//     layout of a B operand of the second MFMA;
//   * stage 2, the LDS shift-add, the carry between column blocks, the runs and the output stores are those of the f16
//     kernel (conv1^T on three-way split operands), one output row per wave instead of two.
// The workgroup is ONE pair (128 threads, 60 KB of LDS): two of them share a CU, each wave on a SIMD of its own (436
// registers).  A four-wave workgroup of two pairs that step through their runs in lockstep (s_barrier is workgroup-wide) was
// built and measured 10 % slower (1.27 against 1.15 ms per 10 s clip): the pairs wait for each other fifteen times a block.
// Per block and wave 1 440 + 180 MFMAs of 16 cycles; 668 images x 32 blocks over 512 pairs -> 0.45 ms of matrix-pipe
// time per 10 s clip, against 1.57 + 0.39 ms for the two kernels it replaces.  Index arithmetic emulated lane by lane in
// tests/test_generic_layouts_cpu.py::test_x3_decoder_*.
#include <string.h>

#include <type_traits>
#include <utility>

#include "dcs_internal.h"
#include "generic.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // rows are F floats apart, F odd
typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));    // channels-last positions are Cin (even) floats apart
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// timing ablations (scripts/build_exp.sh; results are wrong by construction): -DDCS_X3_ABL_NOTAIL drops everything behind a
// row pair's exchange barrier, -DDCS_X3_ABL_NOS1 the stage-1 MFMAs (fragment reads stay), -DDCS_X3_ABL_NOFILL the operand
// split and the LDS writes of the block fill, -DDCS_X3_ABL_NOXCHG the exchange write and its barrier
#if defined(DCS_X3_ABL_NOTAIL)
constexpr bool kAblTail = true;
#else
constexpr bool kAblTail = false;
#endif
#if defined(DCS_X3_ABL_NOS1)
constexpr bool kAblS1 = true;
#else
constexpr bool kAblS1 = false;
#endif
#if defined(DCS_X3_ABL_NOFILL)
constexpr bool kAblFill = true;
#else
constexpr bool kAblFill = false;
#endif
#if defined(DCS_X3_ABL_NOXCHG)
constexpr bool kAblX = true;
#else
constexpr bool kAblX = false;
#endif
// -DDCS_X3_TRACE: the two waves of workgroup 0 add up s_memtime intervals (shader clocks) per phase -- [wave][0] stage 1,
// [1] exchange write + barrier, [2] tail, [3] block fill, [4] whole run loop, [5] row pairs, [6 .. 11] the six pieces of
// the tail (not overlapped form) -- read with x3_trace_dump()
#if defined(DCS_X3_TRACE)
__device__ unsigned long long x3_trace_buf[32];
#define X3_NOW(v_)                                      \
    {                                                   \
        __builtin_amdgcn_sched_barrier(0);              \
        v_ = __builtin_amdgcn_s_memtime();              \
        __builtin_amdgcn_sched_barrier(0);              \
    }
#define X3_ADD(slot_, a_, b_) tr[slot_] += (b_) - (a_);
#else
#define X3_NOW(v_)
#define X3_ADD(slot_, a_, b_)
#endif
constexpr int kTh = 128;      // the workgroup is ONE pair of waves
constexpr int kPairTh = 128;

__device__ __forceinline__ f32x4 mma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned bf_trunc(float x) { return __float_as_uint(x) & 0xffff0000u; }

// Workgroup barrier that orders LDS traffic ONLY (the two waves of a pair talk through LDS alone; nobody reads `out`):
// no s_waitcnt vmcnt(0) for the output store each row pair leaves in flight.  (Measured: no difference on this compiler --
// its __syncthreads() did not wait for the stores either; kept because it states what the kernel needs.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// x = hi + mid + lo exactly (three bf16 by truncation); element j of a piece is k slot j
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = bf_trunc(x[j]);
        const float r1 = x[j] - __uint_as_float(h[j]);
        m[j] = bf_trunc(r1);
        l[j] = bf_trunc(r1 - __uint_as_float(m[j]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = (h[2 * q] >> 16) | h[2 * q + 1];
        mid[q] = (m[2 * q] >> 16) | m[2 * q + 1];
        lo[q] = (l[2 * q] >> 16) | l[2 * q + 1];
    }
}

// slot s = t * NK + k of a row pair (rows y, y + 1) is live when the input row of tap 2 k -- or of tap 2 k + 1, one further
// down -- exists: compile-time list walking for the fully unrolled kernel body
template <int H, int PH, int NK>
constexpr int x3_row0(int y, int s) { return (y + s / NK) - PH + 2 * (s % NK); }
template <int H, int PH, int NK>
constexpr bool x3_live(int y, int s) { return x3_row0<H, PH, NK>(y, s) >= -1 && x3_row0<H, PH, NK>(y, s) <= H - 1; }
template <int H, int PH, int NK>
constexpr int x3_next(int y, int s) {      // first live slot after s, or -1
    for (int s2 = s + 1; s2 < 2 * NK; ++s2)
        if (x3_live<H, PH, NK>(y, s2)) return s2;
    return -1;
}
template <class F, int... I>
__device__ __forceinline__ void x3_for_row_pairs(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, 2 * I>{}), ...);
}
template <class F, int... I>
__device__ __forceinline__ void x3_for_slots(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

struct DcsDecoderX3 {
    const u32x4* Wq;        // [2 parities][KH / 2 taps][3 planes][2 halves][64 lanes] pieces: conv2^T weights (A fragments)
    const u32x4* Wq1;       // [output channel][4 sections][2 tap halves][64 lanes] pieces of the padded conv1 filter
                            // (dcs_decoder_fused_pack; sections 0 .. 2 are the bf16 planes)
    float* out;             // [image][CO][HO][F]
    int F;
    int runs_per_image;
    int64_t n_runs;
};

// Measured dead end (round 4, profiles/r04_n_x3_timeline.txt): issuing the tail of a row pair (six pieces: partner's partial,
// operand split, stage 2, shift-add, carry, store) between the MFMA groups of the NEXT row pair's stage 1 -- two accumulator
// sets, sched_group_barrier interleaving -- made stage 1 exactly as much longer as the tail got shorter (issue is in order:
// every dependent instruction of the tail stalls the MFMA stream behind it): 1.20 against 1.14 ms.
//
// CO = the channels conv1^T produces (the INPUT channels of the graph): 1 for Bach10, 4 for the score-informed graph
// (bach10_scoreinformed/separate_bach10.py:388-447: one decoder branch, four output channels).  With CO > 1 stage 1 is
// unchanged (G[ci][x] of a row does not depend on the output channel) and the rest of a row's tail -- stage 2, shift-add,
// carry, store -- runs once per output channel on that channel's conv1 taps.  Those taps (six 16-byte fragments per lane and
// channel) do not fit the register file next to the conv2^T taps (436 registers) nor LDS next to a second workgroup
// (60 + 12 KB here, 25 KB of taps): they are read from global memory (24.6 KB, L2-resident, the same for every wave), each
// channel's set requested two channels ahead into the register set that channel's products have just released (two sets; the
// first two channels' sets by the last two channels of the row pair before), so that the loads land behind work that does
// not need them.
template <int KH, int H, int CO>
__global__ __launch_bounds__(kTh) void colconv_deconv1_fused_x3_kernel(const DcsColConv g, const DcsDecoderX3 d) {
    constexpr int HO = H + KH - 1, PH = KH - 1, NK = KH / 2;
    static_assert(KH % 2 == 0 && HO % 2 == 0, "taps and output rows are dealt to the two waves by parity");
    constexpr int kRowU = 3 * 4 * 16;                    // 16-byte units per input row: [plane][K piece kq][x]
    constexpr int kPlanes = (H + 2) * kRowU;             // rows -1 .. H (index h + 1); rows -1 and H are zero
    constexpr int kPb = 8 * 32;                          // per wave: [tap mm 8][32 slots] float4, slot 8 + x holds P[x][mm]
    constexpr int kCb = (HO / 2) * CO * 8;               // per wave: the carry of its 15 rows x CO channels, 8 float4 each
    constexpr int kTasks = H * 64;                       // (row, x, K piece) fetch / split tasks per block
    constexpr int NT = (kTasks + 127) / 128;             // per thread of a pair
    __shared__ u32x4 planes[kPlanes];
    __shared__ f32x4 PbS[2 * kPb];
    __shared__ f32x4 CbS[2 * kCb];
    __shared__ f32x4 Xs[2 * 2 * 128];                    // [buffer][writer parity][half][lane]: the partial row handed to the partner
    const int tid = threadIdx.x, lane = tid & 63;
    const int par = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    f32x4* Pb = PbS + par * kPb;
    f32x4* Cb = CbS + par * kCb;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 zeroq = u32x4{0u, 0u, 0u, 0u};
    for (int i = lane; i < kPb; i += 64) Pb[i] = zero4;
    for (int i = tid; i < kRowU; i += kPairTh) {
        planes[i] = zeroq;                               // row -1
        planes[(H + 1) * kRowU + i] = zeroq;             // row H
    }
    // this wave's taps, three planes, two output-channel halves: registers for the whole launch
    u32x4 w[NK][3][2];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) w[k][p][hf] = d.Wq[((((par * NK + k) * 3 + p) * 2) + hf) * 64 + lane];
    // conv1^T taps: CO == 1 keeps its one set for the whole launch; CO > 1 holds the set in use and the one in flight
    u32x4 w1s[CO > 1 ? 2 : 1][3][2];
    const u32x4* w1g = d.Wq1 + lane;                     // channel c, plane p, half mh: w1g[((c * 4 + p) * 2 + mh) * 64]
#define DCS_X3_W1_REQUEST(dst_, c_)                                                                   \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                                \
        dst_[p_][0] = w1g[(((c_) * 4 + p_) * 2) * 64];                                                \
        dst_[p_][1] = w1g[(((c_) * 4 + p_) * 2 + 1) * 64];                                            \
    }
    DCS_X3_W1_REQUEST(w1s[0], 0)
    if constexpr (CO > 1) DCS_X3_W1_REQUEST(w1s[1], 1)
    const int W = g.W, n_xb = g.n_xb, F = d.F, Cin = g.Cin;
    const int rpi = d.runs_per_image;
    // fetch / split task i = tid + 128 j (tid within the pair): row h = i / 64, x = (i % 64) / 4, K piece kqt = i % 4 (four consecutive threads
    // read the 120 consecutive bytes of one position; the last pair of kqt = 3 -- channels 30, 31 -- re-reads 28, 29 and
    // meets zero weights)
    int t_off[NT], t_dst[NT], t_x[NT];
    bool t_ok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int i = tid + kPairTh * j;
        const int h = i >> 6, rem = i & 63, x = rem >> 2, kqt = rem & 3;
        t_ok[j] = i < kTasks;
        t_x[j] = x;
        t_off[j] = (i < kTasks ? h : 0) * W * Cin + 8 * kqt;        // + (column) * Cin
        t_dst[j] = ((h + 1) * 3 * 4 + kqt) * 16 + x;                // + plane * 64
    }
    const bool hi_dup = 8 * (tid & 3) + 8 > Cin;                         // K piece 3 of a 30-channel input
    const int hi_off = hi_dup ? Cin - 4 - 8 * (tid & 3) : 4;              // relative to the task's first channel
    f32x4a8 rlo[NT], rhi[NT];
// Task j: the eight channels of one (row, x, K piece) as TWO 16-byte loads (8-byte aligned: a position is 120 bytes).  The
// second load of K piece 3 (channels 28 .. 31 of 30) starts two channels early and its upper half is used twice: the weights
// of channels 30 and 31 are zero, the values only have to be finite and inside the tensor.
// NOTHING touches the loaded registers before the next block's split, and no branch surrounds the loads (tasks past the
// last row re-read row 0): with element-wise copies or a select behind each load the compiler waited for every pair of loads
// in turn -- six memory latencies, 8 200 cycles per block (13 % of the kernel; profiles/r04_n_x3_timeline.txt).
#define DCS_X3_FETCH(img_, blk_)                                                                        \
    {                                                                                                   \
        const float* ib_ = g.in + (img_) * g.in_n_stride;                                               \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                \
            const int xl_ = (blk_) * 16 + t_x[j];                                                       \
            const float* ip_ = ib_ + (t_off[j] + (xl_ < W ? xl_ : W - 1) * Cin);                        \
            rlo[j] = *reinterpret_cast<const f32x4a8*>(ip_);                                            \
            rhi[j] = *reinterpret_cast<const f32x4a8*>(ip_ + hi_off);                                   \
        }                                                                                               \
    }
    // stage-2 write side: lane (x = fi, kq) owns taps mm = kq and kq + 4 of column x
    f32x4* pw = Pb + kq * 32 + 8 + fi;
    // read side: lane & 31 = q - 16 b (0 .. 22 are real); both half-waves compute the same sums, the first one stores
    const int rq = lane & 31;
    const f32x4* pr = Pb + (rq < 23 ? rq : 22) + 8;
    const u32x4* bl = planes + lane;                     // B fragment of (row, plane): unit (row * 3 + plane) * 64 + lane
    int xbuf = 0;
#if defined(DCS_X3_TRACE)
    unsigned long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tf0 = 0, tf1 = 0, tf2 = 0, tp0 = 0, tp1 = 0, ta = 0, tb = 0, tc = 0, td = 0, t_run0 = 0, t_run1 = 0;
    X3_NOW(t_run0)
#endif
    for (int64_t run = blockIdx.x; run < d.n_runs; run += gridDim.x) {
        const bool have = true;
        const int64_t img = run / rpi;
        const int rr = (int)(run - img * rpi);
        const int b_lo = (int)((int64_t)rr * n_xb / rpi), b_hi = (int)((int64_t)(rr + 1) * n_xb / rpi);
        const int b_first = b_lo > 0 ? b_lo - 1 : 0;         // the block to the left is recomputed for its carry
        const int n_it = b_hi - b_first;
        for (int i = lane; i < kCb; i += 64) Cb[i] = zero4;
        if (have) DCS_X3_FETCH(img, b_first)
        for (int it = 0; it < n_it; ++it) {
            const int b = b_first + it;
            constexpr bool active = true;
            X3_NOW(ta)
            lds_barrier();                                 // both waves are done with the previous block's planes
            X3_NOW(tf0)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (active && t_ok[j] && !(kAblFill && it > 0)) {
                    u32x4 p0, p1, p2;
                    float rawj[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) rawj[e] = rlo[j][e];
                    rawj[4] = hi_dup ? rhi[j][2] : rhi[j][0];
                    rawj[5] = hi_dup ? rhi[j][3] : rhi[j][1];
                    rawj[6] = rhi[j][2];
                    rawj[7] = rhi[j][3];
                    split8(rawj, p0, p1, p2);
                    const bool x_ok = b * 16 + t_x[j] < W;   // columns past W (last block): zero input, zero G, zero products
                    planes[t_dst[j]] = x_ok ? p0 : zeroq;
                    planes[t_dst[j] + 64] = x_ok ? p1 : zeroq;
                    planes[t_dst[j] + 128] = x_ok ? p2 : zeroq;
                }
            }
            X3_NOW(tf1)
            lds_barrier();
            X3_NOW(tf2)
            X3_ADD(12, ta, tf0)
            X3_ADD(13, tf0, tf1)
            X3_ADD(14, tf1, tf2)
            if (active) {
                const int nb = b + 1 < b_hi ? b + 1 : b;     // last block of the run: a harmless re-read
                DCS_X3_FETCH(img, nb)
            }
            X3_NOW(tb)
            X3_ADD(3, ta, tb)
            X3_ADD(15, tf2, tb)
            const bool keep = b >= b_lo;                     // false for the recomputed block
            const int f0 = 4 * (b * 16 + rq);
            float* orow = d.out + (img * CO * HO + par) * (int64_t)F + f0;      // channel c: + c * HO * F
            // state of the row pair being finished
            f32x4 acc[2][2];                                 // [row of the pair][channel half]
            f32x4 tm[2], tsum = zero4;
            f32x4 pq[CO > 1 ? 2 : 1][2][2];                  // stage-2 products of the channel being finished / the next one
            u32x4 tg[3];
            const f32x4* t_xr = Xs;
            constexpr int kPieces = 6;
            auto tail = [&](auto pc, auto ypc, auto cc) {
                constexpr int P = decltype(pc)::value, yp = decltype(ypc)::value, c = decltype(cc)::value;
                if constexpr (kAblTail) {
                    if constexpr (P == 0 && c == 0) tsum += acc[0][0] + acc[1][1];      // keeps stage 1 alive
                    if constexpr (P == 5 && yp == HO - 2 && c == 0)
                        if (keep && lane < 16 && f0 + 4 <= F) *reinterpret_cast<f32x4u*>(orow) = tsum;
                } else if constexpr (P == 0) {          // this wave finishes row yp + par: its own partial + the partner's
                    if constexpr (c == 0) {
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) tm[hf] = par ? acc[1][hf] : acc[0][hf];
                        tm[0] += t_xr[0];
                        tm[1] += t_xr[64];
                    }
                } else if constexpr (P == 1) {   // G[ci][x] of the row as three bf16 planes (a B operand of stage 2)
                    if constexpr (c == 0) {
                        float gv[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gv[e] = tm[0][e];
                            gv[4 + e] = tm[1][e];
                        }
                        split8(gv, tg[0], tg[1], tg[2]);
                    }
                } else if constexpr (P == 2) {
                    auto& w1 = w1s[c & 1];
                    // the two tap halves as FOUR chains of three products (smallest terms first in each): with one wave per
                    // SIMD a chain of six dependent MFMAs per half was ~430 cycles of pure latency per row pair
                    f32x4 (&pa)[2] = pq[c & 1][0];
                    f32x4 (&pb)[2] = pq[c & 1][1];
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        pa[mh] = mma_bf(w1[2][mh], tg[0], zero4);
                        pb[mh] = mma_bf(w1[0][mh], tg[2], zero4);
                    }
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        pa[mh] = mma_bf(w1[1][mh], tg[1], pa[mh]);
                        pb[mh] = mma_bf(w1[1][mh], tg[0], pb[mh]);
                    }
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        pa[mh] = mma_bf(w1[0][mh], tg[1], pa[mh]);
                        pb[mh] = mma_bf(w1[0][mh], tg[0], pb[mh]);
                    }
                    if constexpr (CO > 1) {              // this set is free: the taps of channel c + 2 (of the next row pair
                        __builtin_amdgcn_sched_barrier(0);      // for the last two channels) land behind the work in between
                        DCS_X3_W1_REQUEST(w1s[c & 1], (c + 2) % CO)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if constexpr (P == 3) {
                    pw[0] = pq[c & 1][0][0] + pq[c & 1][1][0];
                    pw[128] = pq[c & 1][0][1] + pq[c & 1][1][1];
                } else if constexpr (P == 4) {
                    asm volatile("" ::: "memory");           // the pieces of the row are written (LDS is in order per wave)
                    // all nine reads in flight before the first addition (left alone the compiler waits for each in turn)
                    f32x4 rd[8];
#pragma unroll
                    for (int mm = 0; mm < 8; ++mm) rd[mm] = pr[mm * 32 - mm];
                    const f32x4 cin = Cb[((yp >> 1) * CO + c) * 8 + (rq & 7)];
                    __builtin_amdgcn_sched_barrier(0);
                    tsum = ((rd[0] + rd[1]) + (rd[2] + rd[3])) + ((rd[4] + rd[5]) + (rd[6] + rd[7])) + (rq < 8 ? cin : zero4);
                } else {
                    asm volatile("" ::: "memory");           // every lane has read the carry before it is replaced
                    if (lane >= 16 && lane < 24) Cb[((yp >> 1) * CO + c) * 8 + lane - 16] = lane < 23 ? tsum : zero4;
#if defined(DCS_X3_ABL_NOSTORE)
                    if (keep && lane < 16 && tsum[0] == 12345.f) {
#else
                    if (keep && lane < 16) {
#endif
                        float* op = orow + ((int64_t)c * HO + yp) * F;
                        if (f0 + 4 <= F) {
                            *reinterpret_cast<f32x4u*>(op) = tsum;
                        } else {
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                if (f0 + e < F) op[e] = tsum[e];
                        }
                    }
                    asm volatile("" ::: "memory");       // (and the pieces of this channel are read before the next one's are written)
                }
            };
            // the 15 row pairs and their (row, tap) slots are unrolled through integer sequences: every weight register index
            // and every "is this slot live" is a compile-time constant (a `#pragma unroll` over 15 bodies of this size is
            // declined by the optimiser)
            x3_for_row_pairs([&](auto yc) {
                constexpr int y = decltype(yc)::value;
                X3_NOW(ta)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = zero4;
                // The B fragments of the NEXT live slot are requested before the twelve MFMAs of the current one are issued:
                // one wave per SIMD, so nobody else hides the LDS latency.
                u32x4 bc[3], bn[3];
                {
                    constexpr int s0 = x3_next<H, PH, NK>(y, -1);
                    const u32x4* bp = bl + (x3_row0<H, PH, NK>(y, s0) + 1 + par) * kRowU;
                    bc[0] = bp[0]; bc[1] = bp[64]; bc[2] = bp[128];
                }
                x3_for_slots([&](auto sc) {
                    constexpr int sl = decltype(sc)::value;
                    if constexpr (x3_live<H, PH, NK>(y, sl)) {
                        constexpr int t = sl / NK, k = sl % NK;
                        constexpr int s_next = x3_next<H, PH, NK>(y, sl);
                        if constexpr (s_next >= 0) {
                            const u32x4* bp = bl + (x3_row0<H, PH, NK>(y, s_next) + 1 + par) * kRowU;
                            bn[0] = bp[0]; bn[1] = bp[64]; bn[2] = bp[128];
                            // nothing moves across: the three requests stay IN FRONT of this slot's MFMAs (left alone the
                            // scheduler re-uses the fragment registers and issues them behind the last MFMA: ~100 idle
                            // cycles per slot with one wave per SIMD)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // smallest terms first; the two channel halves alternate so that no MFMA waits for the one before it
                        if constexpr (kAblS1) {
                            acc[t][0][0] += __uint_as_float(bc[0][0] ^ bc[1][1] ^ bc[2][2]) + __uint_as_float(w[k][0][0][0]);
                        } else {
                        acc[t][0] = mma_bf(w[k][2][0], bc[0], acc[t][0]);
                        acc[t][1] = mma_bf(w[k][2][1], bc[0], acc[t][1]);
                        acc[t][0] = mma_bf(w[k][0][0], bc[2], acc[t][0]);
                        acc[t][1] = mma_bf(w[k][0][1], bc[2], acc[t][1]);
                        acc[t][0] = mma_bf(w[k][1][0], bc[1], acc[t][0]);
                        acc[t][1] = mma_bf(w[k][1][1], bc[1], acc[t][1]);
                        acc[t][0] = mma_bf(w[k][1][0], bc[0], acc[t][0]);
                        acc[t][1] = mma_bf(w[k][1][1], bc[0], acc[t][1]);
                        acc[t][0] = mma_bf(w[k][0][0], bc[1], acc[t][0]);
                        acc[t][1] = mma_bf(w[k][0][1], bc[1], acc[t][1]);
                        acc[t][0] = mma_bf(w[k][0][0], bc[0], acc[t][0]);
                        acc[t][1] = mma_bf(w[k][0][1], bc[0], acc[t][1]);
                        }
                        if constexpr (s_next >= 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            bc[0] = bn[0]; bc[1] = bn[1]; bc[2] = bn[2];
                        }
                    }
                }, std::make_integer_sequence<int, 2 * NK>{});
                // this wave finishes row y + par: its partial of the OTHER row goes to the partner, the partner's partial of
                // this row comes back
                X3_NOW(tb)
                if constexpr (!kAblX) {
                    f32x4* xw = Xs + (xbuf * 2 + par) * 128 + lane;
                    xw[0] = par ? acc[0][0] : acc[1][0];
                    xw[64] = par ? acc[0][1] : acc[1][1];
                    lds_barrier();
                }
                t_xr = Xs + (xbuf * 2 + (1 - par)) * 128 + lane;
                xbuf ^= 1;
                X3_NOW(tc)
                // pieces of a row's tail: 0 partner's partial, 1 operand split (once per row); 2 stage-2 products, 3 their LDS
                // write, 4 shift-add, 5 carry + store (once per output channel).  The products of channel c + 1 are issued
                // BEFORE channel c is shifted, added and stored: their latency (three dependent MFMAs) passes behind that work.
                auto piece = [&](auto pc, auto cc) {
                    X3_NOW(tp0)
                    tail(pc, yc, cc);
                    X3_NOW(tp1)
                    X3_ADD(6 + decltype(pc)::value, tp0, tp1)
                };
                using I0 = std::integral_constant<int, 0>;
                piece(std::integral_constant<int, 0>{}, I0{});
                piece(std::integral_constant<int, 1>{}, I0{});
                piece(std::integral_constant<int, 2>{}, I0{});
                x3_for_slots([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
#if defined(DCS_X3_SEQ_TAILS)   // experiment build: channel c + 1's products only after channel c is stored
                    if constexpr (c > 0) piece(std::integral_constant<int, 2>{}, cc);
#else
                    if constexpr (c + 1 < CO) piece(std::integral_constant<int, 2>{}, std::integral_constant<int, c + 1>{});
#endif
                    piece(std::integral_constant<int, 3>{}, cc);
                    piece(std::integral_constant<int, 4>{}, cc);
                    piece(std::integral_constant<int, 5>{}, cc);
                }, std::make_integer_sequence<int, CO>{});
                X3_NOW(td)
                X3_ADD(0, ta, tb)
                X3_ADD(1, tb, tc)
                X3_ADD(2, tc, td)
                X3_ADD(5, ta, td)
            }, std::make_integer_sequence<int, HO / 2>{});
        }
        // the carry: the tail of the image (and the zeros up to F) for the last run, otherwise the next run recomputes it
        if (have && b_hi == n_xb) {
#pragma unroll
            for (int i = 0; i < (HO / 2 * CO + 7) / 8; ++i) {
                const int tc_ = (lane >> 3) + 8 * i, f = 4 * (16 * n_xb + (lane & 7));      // (row of this wave, channel)
                if (tc_ < HO / 2 * CO) {
                    const int t = tc_ / CO, c = tc_ - t * CO;
                    const f32x4 v = Cb[tc_ * 8 + (lane & 7)];
                    float* op = d.out + ((img * CO + c) * HO + 2 * t + par) * (int64_t)F + f;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (f + e < F) op[e] = v[e];
                }
            }
        }
        asm volatile("" ::: "memory");
    }
#if defined(DCS_X3_TRACE)
    X3_NOW(t_run1)
    tr[4] = t_run1 - t_run0;
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 16; ++i) x3_trace_buf[par * 16 + i] = tr[i];
#endif
#undef DCS_X3_FETCH
#undef DCS_X3_W1_REQUEST
}

}  // namespace

#if defined(DCS_X3_TRACE)
extern "C" __attribute__((visibility("default"))) int x3_trace_dump(unsigned long long* out32) {
    return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(x3_trace_buf), sizeof(unsigned long long) * 32);
}
#endif

// Wf: [kh][32 out][40] f32, in-channel fastest (the transposed conv2 filter as the column kernels take it) ->
// [2 parities][kh / 2][3 planes][2 halves][64 lanes] 16-byte pieces: lane (fi, kg) of tap u = 2 k + par, half hf holds
// W[u][out = fi + 16 hf][in = 8 kg .. 8 kg + 7] as bf16 plane p (x = p0 + p1 + p2 exactly)
void dcs_decoder_x3_pack(const float* Wf, int kh, std::vector<uint16_t>* out) {
    const int nk = kh / 2;
    out->assign((size_t)2 * nk * 3 * 2 * 64 * 8, 0);
    for (int par = 0; par < 2; ++par)
        for (int k = 0; k < nk; ++k)
            for (int hf = 0; hf < 2; ++hf)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int u = 2 * k + par;
                        float r = Wf[((size_t)u * 32 + (lane & 15) + 16 * hf) * 40 + (lane >> 4) * 8 + j];
                        for (int p = 0; p < 3; ++p) {
                            uint32_t bits;
                            memcpy(&bits, &r, 4);
                            bits &= 0xffff0000u;
                            float part;
                            memcpy(&part, &bits, 4);
                            r -= part;
                            (*out)[((((((size_t)par * nk + k) * 3 + p) * 2 + hf) * 64) + lane) * 8 + j] = (uint16_t)(bits >> 16);
                        }
                    }
}

// n_out: the channels conv1^T produces (the graph's input channels)
bool dcs_decoder_x3_ok(const DcsColConv& a, int F, int n_out) {
    static const bool on = !(getenv("DCS_DECODER_X3") && atoi(getenv("DCS_DECODER_X3")) == 0);
    // 28 .. 32 channels: the first half of K piece 3 (channels 24 .. 27) is read as it is
    return on && (n_out == 1 || n_out == 4) && dcs_decoder_fused_ok(a, F) && (a.Cin & 1) == 0 && a.Cin >= 28;
}

// the input MUST be channels-last ([image][H][W][Cin], 8-byte aligned): false = not launched
bool dcs_launch_decoder_x3(dcs_ctx* ctx, const DcsColConv& a, int64_t n_images, const void* Wq, const void* Wq1, float* out, int F,
                           int n_out) {
    if (!Wq || !Wq1 || !dcs_decoder_x3_ok(a, F, n_out)) return false;
    if ((a.in_n_stride & 1) || (reinterpret_cast<uintptr_t>(a.in) & 7)) return false;
    if (n_images <= 0) return true;
    // runs per image: fewest (rounds of wave pairs) x (blocks per run + the recomputed one); two pairs per CU
    const int64_t n_pairs = (int64_t)ctx->n_cu * 2;
    int best = 1;
    int64_t best_cost = -1;
    for (int rpi = 1; rpi <= a.n_xb; ++rpi) {
        const int64_t len = (a.n_xb + rpi - 1) / rpi + (rpi > 1 ? 1 : 0);
        const int64_t cost = dcs_cdiv(n_images * rpi, n_pairs) * len;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rpi; }
    }
    DcsDecoderX3 d{};
    d.Wq = reinterpret_cast<const u32x4*>(Wq);
    d.Wq1 = reinterpret_cast<const u32x4*>(Wq1);
    d.out = out;
    d.F = F;
    d.runs_per_image = best;
    d.n_runs = n_images * best;
    const unsigned grid = (unsigned)std::min<int64_t>(d.n_runs, n_pairs);
    if (n_out == 4)
        hipLaunchKernelGGL((colconv_deconv1_fused_x3_kernel<20, 11, 4>), dim3(grid), dim3(kTh), 0, ctx->stream, a, d);
    else
        hipLaunchKernelGGL((colconv_deconv1_fused_x3_kernel<20, 11, 1>), dim3(grid), dim3(kTh), 0, ctx->stream, a, d);
    return true;
}
