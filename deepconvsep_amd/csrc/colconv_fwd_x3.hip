// conv2 of the Bach10 / score-informed graphs (30 -> 30 channels, a 20 x 1 filter along time, no padding: 30 frames -> 11;
// separate_bach10.py:195-198, bach10_scoreinformed/separate_bach10.py:409-412) with f32-class arithmetic on the bf16 matrix
// pipe, weights in registers (round 5; the default for this shape since round 6, see dcs_colconv_fwd_x3_ok).
//
//   out[n][co][y][x] = bias[co] + sum_u sum_ci Wf[u][co][ci] * in[n][y + u][x][ci]        in: CHANNELS-LAST (conv1_mfma_kernel<C, true>)
//
// slabconv_ps_kernel, which served this layer before, streams the weight stages through LDS behind a barrier each: 0.64 of the
// f32 peak.  Here TWO waves share a column block of 16 x, one per HALF of the output channels: wave h keeps all 20 taps of the
// output channels 16 h .. 16 h + 15 as three bf16 planes in registers (240 VGPRs) and its eleven output rows in eleven
// accumulators (44) -- no partial sums to exchange, no epilogue but bias + store.  (A first version dealt the TAPS to the waves,
// as the fused decoder does: 88 accumulator registers per wave plus a hand-over of partial rows, and the compiler spilled
// 60 - 200 registers around every block; profiles/r05_q_*.)
//   * step s = 0 .. 29: both waves read the B fragments (three planes) of input row s from LDS and multiply them with every
//     tap u that maps the row onto an output row, y = s - u in 0 .. 10: up to eleven independent accumulators per step, six
//     MFMAs each (the products of the three-way split operands, smallest first);
//   * the input rows arrive two at a time: 128 (row, x, K piece) tasks = one per thread, requested three barrier intervals
//     ahead, split exactly into three bf16 terms and written to a ring of three chunks in LDS (18 KB), one barrier per chunk; the
//     pipeline runs across the blocks of a workgroup (the last two intervals of a block write the next block's first rows).
// 1 320 MFMAs per wave and block; 2 workgroups of one pair per CU, each wave on a SIMD of its own.
#include <string.h>

#include <type_traits>
#include <utility>

#include "dcs_internal.h"
#include "generic.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));    // channels-last positions are Cin (even) floats apart
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kTh = 128;      // the workgroup is ONE pair of waves

__device__ __forceinline__ f32x4 mma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned bf_trunc(float x) { return __float_as_uint(x) & 0xffff0000u; }
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

template <class F, int... I>
__device__ __forceinline__ void fwd_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

struct DcsColFwdX3 {
    const u32x4* Wq;        // [2 tap parities][KH / 2][3 planes][2 halves][64 lanes] pieces (dcs_decoder_x3_pack of the forward filter)
    int64_t n_items;        // images x column blocks
};

template <int KH, int H>
__global__ __launch_bounds__(kTh) void colconv_fwd_x3_kernel(const DcsColConv g, const DcsColFwdX3 d) {
    constexpr int HO = H - KH + 1, NK = KH / 2;
    constexpr int NCH = H / 2;                           // chunks of two input rows
    constexpr int kRowU = 3 * 4 * 16;                    // 16-byte units per input row: [plane][K piece kq][x]
    constexpr int kRing = 3;                             // chunks in LDS: the one being read, the next one, the one being filled
    static_assert(KH % 2 == 0 && H % 2 == 0 && NCH % kRing == 0 && NCH >= 2 * kRing, "rows in pairs, chunk slots line up across blocks");
    __shared__ u32x4 planes[kRing * 2 * kRowU];
    const int tid = threadIdx.x, lane = tid & 63;
    const int hf = __builtin_amdgcn_readfirstlane(tid >> 6);          // this wave's half of the output channels
    const int fi = lane & 15, kq = lane >> 4;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 zeroq = u32x4{0u, 0u, 0u, 0u};
    // all taps of this wave's output channels, three planes: registers for the whole launch (tap u = 2 k + parity in the packed array)
    u32x4 w[KH][3];
#pragma unroll
    for (int u = 0; u < KH; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p) w[u][p] = d.Wq[(((((u & 1) * NK + (u >> 1)) * 3 + p) * 2) + hf) * 64 + lane];
    const int W = g.W, n_xb = g.n_xb, Cin = g.Cin, Cout = g.Cout;
    // fetch / split task of a chunk: row rr = tid / 64 of the chunk, x = (tid % 64) / 4, K piece kqt = tid % 4 (four consecutive
    // threads read the 120 consecutive bytes of one position; the last pair of kqt = 3 -- channels 30, 31 -- re-reads 28, 29 and
    // meets zero weights)
    const int t_rr = tid >> 6, t_x = (tid & 63) >> 2, t_kq = tid & 3;
    const int t_dst = (t_rr * 3 * 4 + t_kq) * 16 + t_x;                  // + plane * 64 + ring slot * 2 * kRowU
    const bool hi_dup = 8 * t_kq + 8 > Cin;                              // K piece 3 of a 30-channel input
    const int hi_off = hi_dup ? Cin - 4 - 8 * t_kq : 4;                  // relative to the task's first channel
    f32x4a8 rlo[kRing], rhi[kRing];                                      // raw slot = chunk % kRing
// the two 16-byte loads of this thread's task of chunk ch_ (rows 2 ch_, 2 ch_ + 1); columns past W read column W - 1 (zeroed
// at the split).  Nothing touches the loaded registers before that chunk's split, and no branch surrounds the loads.
#define DCS_FWD_FETCH(ib_, xg_, ch_)                                                                    \
    {                                                                                                   \
        const float* ip_ = (ib_) + ((int64_t)(2 * (ch_) + t_rr) * W + (xg_)) * Cin + 8 * t_kq;         \
        rlo[(ch_) % kRing] = *reinterpret_cast<const f32x4a8*>(ip_);                                    \
        rhi[(ch_) % kRing] = *reinterpret_cast<const f32x4a8*>(ip_ + hi_off);                           \
    }
#define DCS_FWD_SPLIT(slot_, ok_)                                                                       \
    {                                                                                                   \
        u32x4 p0, p1, p2;                                                                               \
        float raw_[8];                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) raw_[e] = rlo[slot_][e];                         \
        raw_[4] = hi_dup ? rhi[slot_][2] : rhi[slot_][0];                                               \
        raw_[5] = hi_dup ? rhi[slot_][3] : rhi[slot_][1];                                               \
        raw_[6] = rhi[slot_][2];                                                                        \
        raw_[7] = rhi[slot_][3];                                                                        \
        split8(raw_, p0, p1, p2);                                                                       \
        u32x4* dst_ = planes + (slot_) * 2 * kRowU + t_dst;                                             \
        dst_[0] = (ok_) ? p0 : zeroq;                                                                   \
        dst_[64] = (ok_) ? p1 : zeroq;                                                                  \
        dst_[128] = (ok_) ? p2 : zeroq;                                                                 \
    }
    // item -> (image base, this thread's column, column inside the image?)
    auto locate = [&](int64_t item, const float*& ib, int& xg, bool& ok) {
        const int64_t img = item / n_xb;
        const int b = (int)(item - img * n_xb);
        ib = g.in + img * g.in_n_stride;
        const int xr = b * 16 + t_x;
        ok = xr < W;                                     // columns past W (last block): zero input, results never stored
        xg = ok ? xr : W - 1;
    };
    const u32x4* bl = planes + lane;                     // B fragment of (ring row, plane): unit (row * 3 + plane) * 64 + lane
    const float* ib = g.in;
    int xg = 0;
    bool x_ok = false;
    if ((int64_t)blockIdx.x < d.n_items) {               // first block: chunks 0 and 1 into the ring, 2 .. 4 in flight
        locate(blockIdx.x, ib, xg, x_ok);
        DCS_FWD_FETCH(ib, xg, 0)
        DCS_FWD_FETCH(ib, xg, 1)
        DCS_FWD_SPLIT(0, x_ok)
        DCS_FWD_SPLIT(1, x_ok)
        DCS_FWD_FETCH(ib, xg, 2)
        DCS_FWD_FETCH(ib, xg, 3)
        DCS_FWD_FETCH(ib, xg, 4)
    }
    for (int64_t item = blockIdx.x; item < d.n_items; item += gridDim.x) {
        const int64_t img = item / n_xb;
        const int b = (int)(item - img * n_xb);
        const float* ib_n = ib;                          // the next block of this workgroup (none: harmless re-reads of this one)
        int xg_n = xg;
        bool ok_n = x_ok;
        if (item + gridDim.x < d.n_items) locate(item + gridDim.x, ib_n, xg_n, ok_n);
        f32x4 acc[HO];
        u32x4 bc[3];
#pragma unroll
        for (int y = 0; y < HO; ++y) acc[y] = zero4;
        fwd_for([&](auto ic) {
            constexpr int i = decltype(ic)::value;       // chunk i = the input rows 2 i and 2 i + 1 = the steps s = 2 i, 2 i + 1
            lds_barrier();                               // chunk i (and i + 1) is written; nobody reads chunk i - 1 any more
            // chunk i + 2 (of the next block from i = NCH - 2 on) into the ring slot chunk i - 1 occupied, chunk i + 5 requested
            if constexpr (i + 2 < NCH) {
                DCS_FWD_SPLIT((i + 2) % kRing, x_ok)
            } else {
                DCS_FWD_SPLIT((i + 2) % kRing, ok_n)
            }
            if constexpr (i + 2 + kRing < NCH) {
                DCS_FWD_FETCH(ib, xg, i + 2 + kRing)
            } else {
                DCS_FWD_FETCH(ib_n, xg_n, i + 2 + kRing - NCH)
            }
            fwd_for([&](auto hc) {
                constexpr int s = 2 * i + decltype(hc)::value;       // input row s in ring row ((s / 2) % kRing) * 2 + s % 2
                if constexpr (s == 0) {                  // the first step of a block reads its own fragments ...
                    bc[0] = bl[0]; bc[1] = bl[64]; bc[2] = bl[128];
                }
                u32x4 bn[3];
                if constexpr (s + 1 < H) {               // ... every other step finds them requested in front of the step before
                    constexpr int q = (((s + 1) / 2) % kRing) * 2 + ((s + 1) % 2);
                    const u32x4* bp = bl + q * kRowU;
                    bn[0] = bp[0]; bn[1] = bp[64]; bn[2] = bp[128];
                    __builtin_amdgcn_sched_barrier(0);   // the requests stay IN FRONT of this step's MFMAs
                }
                // every tap that maps input row s onto an output row: y = s - u.  The slots of a step are independent
                // accumulators; their products are issued term by term across the slots (smallest terms first in every chain).
                constexpr int u_lo = s - (HO - 1) > 0 ? s - (HO - 1) : 0, u_hi = s < KH - 1 ? s : KH - 1;
                auto term = [&](auto pa_c, auto pb_c) {
                    constexpr int pa = decltype(pa_c)::value, pb = decltype(pb_c)::value;
                    fwd_for([&](auto uc) {
                        constexpr int u = u_lo + decltype(uc)::value, y = s - u;
                        static_assert(y >= 0 && y < HO && u < KH, "slot outside the output");
                        acc[y] = mma_bf(w[u][pa], bc[pb], acc[y]);
                    }, std::make_integer_sequence<int, u_hi - u_lo + 1>{});
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                term(I2{}, I0{});
                term(I0{}, I2{});
                term(I1{}, I1{});
                term(I1{}, I0{});
                term(I0{}, I1{});
                term(I0{}, I0{});
                if constexpr (s + 1 < H) {
                    __builtin_amdgcn_sched_barrier(0);
                    bc[0] = bn[0]; bc[1] = bn[1]; bc[2] = bn[2];
                }
            }, std::make_integer_sequence<int, 2>{});
        }, std::make_integer_sequence<int, NCH>{});
        // lane (x = fi, kq) of an accumulator: output channels 16 hf + 4 kq + e at column x
        {
            const int x0 = b * 16 + fi;
            if (x0 < W) {
                float* ob = g.out + img * g.out_n_stride + x0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = 16 * hf + 4 * kq + e;
                    if (co < Cout) {
                        const float bv = g.bias[co];
#pragma unroll
                        for (int y = 0; y < HO; ++y) ob[((int64_t)co * HO + y) * W] = acc[y][e] + bv;
                    }
                }
            }
        }
        ib = ib_n;
        xg = xg_n;
        x_ok = ok_n;
    }
#undef DCS_FWD_FETCH
#undef DCS_FWD_SPLIT
}

}  // namespace

// the input MUST be channels-last ([image][H][W][Cin], 8-byte aligned), the output is channel-first [image][Cout][HO][W]
// The default for this shape since round 6 (1.6 x faster than slabconv_ps_kernel; round 5 kept it opt-in because ONE
// ill-conditioned mask bin of one random draw landed at 1.8e-4 instead of 3.9e-5 with it -- a bin whose value depends on the last
// bit of the network output whatever kernel computes it; the parity criterion no longer counts such bins, INTEGRATION.md section 6;
// same-box A/B against slabconv_ps_kernel, which still serves the other shapes: profiles/r06_a_conv2_default_on_legs_ab.txt).
bool dcs_colconv_fwd_x3_ok(const DcsColConv& a) {
    return a.kh == 20 && a.ph == 0 && a.H == 30 && a.Ho == 11 && a.W >= 16 && a.Cout <= 32 && a.Cin <= 32 && a.Cin >= 28 &&
           (a.Cin & 1) == 0;
}

bool dcs_launch_colconv_fwd_x3(dcs_ctx* ctx, const DcsColConv& a0, int64_t n_images, const void* Wq) {
    if (!Wq || !dcs_colconv_fwd_x3_ok(a0)) return false;
    if ((a0.in_n_stride & 1) || (reinterpret_cast<uintptr_t>(a0.in) & 7)) return false;
    if (n_images <= 0) return true;
    DcsColConv a = a0;
    a.n_xb = (a.W + 15) / 16;
    DcsColFwdX3 d{};
    d.Wq = reinterpret_cast<const u32x4*>(Wq);
    d.n_items = n_images * a.n_xb;
    const int64_t n_pairs = (int64_t)ctx->n_cu * 2;                       // two pairs per CU
    // equal shares: the fewest rounds, then as many workgroups as give every one the same number of items (+- 1)
    const int64_t rounds = dcs_cdiv(d.n_items, n_pairs);
    const unsigned grid = (unsigned)dcs_cdiv(d.n_items, rounds);
    hipLaunchKernelGGL((colconv_fwd_x3_kernel<20, 30>), dim3(grid), dim3(kTh), 0, ctx->stream, a, d);
    return true;
}
