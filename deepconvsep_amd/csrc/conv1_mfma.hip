// conv1 of the Bach10 / score-informed graphs (30 filters of 1 x 30, frequency stride 4; separate_bach10.py:190-193,
// bach10_scoreinformed/separate_bach10.py:404-407) on the bf16 matrix pipe, f32-class.
//
//   out[n][o][t][j] = bias[o] + sum_c sum_u x[n][c][t][4 j + u] * Wc[o][c][u]
//
// conv1_reg_kernel (generic.hip) does this on the vector ALU: 900 multiply-adds per position and input channel, 0.33 ms for
// the 4-channel score-informed batch (28 % of the vector peak).  As a GEMM the K axis of a channel is its 30 taps padded
// to 32 = ONE MFMA K block, and with a stride of 4 a position's taps start 16 bytes after its neighbour's: a workgroup
// copies the 1 056 input floats that 256 positions of one (tile, frame) row can touch into LDS per channel, a lane
// (position fi, kg) reads its taps 8 kg .. 8 kg + 7 as two aligned ds_read_b128, splits them exactly into three bf16
// terms (truncation; six products kept, as everywhere in this library) and issues 12 MFMAs per (16 positions, channel)
// against weight fragments that stay in registers.  The accumulators (rows = positions) go through an LDS tile so that
// the stores are rows of up to 256 consecutive positions of one filter, not 64-byte pieces.
#include <string.h>

#include "dcs_internal.h"
#include "generic.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kThreads = 256;
constexpr int kPos = 256;                    // positions per workgroup
constexpr int kInW = 4 * kPos + 32;          // input floats per channel a workgroup can touch (taps padded to 32)
constexpr int kOutS = kPos + 4;              // row stride of the output tile in LDS (floats)

__device__ __forceinline__ unsigned bf_trunc(float x) { return __float_as_uint(x) & 0xffff0000u; }

__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4& hi, u32x4& mid, u32x4& lo) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? x0[j] : x1[j - 4];
        h[j] = bf_trunc(x);
        const float r1 = x - __uint_as_float(h[j]);
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

__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int C>
__global__ __launch_bounds__(kThreads) void conv1_mfma_kernel(const float* __restrict__ x, const u32x4* __restrict__ Wq,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int NF, int tc, int F, int w1) {
    __shared__ __attribute__((aligned(16))) float xin[C * kInW];
    __shared__ __attribute__((aligned(16))) float obuf[32 * kOutS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const int fi = lane & 15, kq = lane >> 4;
    const int64_t nt = blockIdx.x;            // n * tc + t
    const int64_t n = nt / tc;
    const int t = (int)(nt - n * tc);
    const int j_base = blockIdx.y * kPos;
    const int cnt = w1 - j_base < kPos ? w1 - j_base : kPos;      // positions of this workgroup
    // weights: B fragments [c][plane][filter half], lane (filter fi (+16), kg) = taps 8 kg .. 8 kg + 7
    u32x4 w[C][3][2];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            w[c][p][0] = Wq[((c * 3 + p) * 2) * 64 + lane];
            w[c][p][1] = Wq[((c * 3 + p) * 2 + 1) * 64 + lane];
        }
    const float b0 = fi < NF ? bias[fi] : 0.f, b1 = 16 + fi < NF ? bias[16 + fi] : 0.f;
    // input: the floats [4 j_base, 4 j_base + kInW) of the C rows, zero past the row
    const int f0 = 4 * j_base;
    for (int i = tid; i < C * kInW; i += kThreads) {
        const int c = i / kInW, k = i - c * kInW;
        xin[i] = f0 + k < F ? x[((n * C + c) * tc + t) * (int64_t)F + f0 + k] : 0.f;
    }
    __syncthreads();
    const int n_blk = (cnt + 15) >> 4;
    for (int b = wave; b < n_blk; b += 4) {
        f32x4 acc0 = f32x4{b0, b0, b0, b0}, acc1 = f32x4{b1, b1, b1, b1};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const f32x4* xp = reinterpret_cast<const f32x4*>(xin + c * kInW + 4 * (16 * b + fi) + 8 * kq);
            u32x4 a0, a1, a2;
            split8(xp[0], xp[1], a0, a1, a2);
            acc0 = mma(a2, w[c][0][0], acc0);   // smallest products first
            acc1 = mma(a2, w[c][0][1], acc1);
            acc0 = mma(a0, w[c][2][0], acc0);
            acc1 = mma(a0, w[c][2][1], acc1);
            acc0 = mma(a1, w[c][1][0], acc0);
            acc1 = mma(a1, w[c][1][1], acc1);
            acc0 = mma(a1, w[c][0][0], acc0);
            acc1 = mma(a1, w[c][0][1], acc1);
            acc0 = mma(a0, w[c][1][0], acc0);
            acc1 = mma(a0, w[c][1][1], acc1);
            acc0 = mma(a0, w[c][0][0], acc0);
            acc1 = mma(a0, w[c][0][1], acc1);
        }
        // lane (filter fi, kq): positions 16 b + 4 kq .. + 3
        *reinterpret_cast<f32x4*>(obuf + fi * kOutS + 16 * b + 4 * kq) = acc0;
        *reinterpret_cast<f32x4*>(obuf + (16 + fi) * kOutS + 16 * b + 4 * kq) = acc1;
    }
    __syncthreads();
    // rows of consecutive positions: a wave instruction writes 256 contiguous bytes of one filter's row.  (Groups of four
    // positions on 16-byte boundaries of `out` -- the alignment depends on the row -- with the head as scalars were slower:
    // 0.22 vs 0.19 ms on the score-informed batch.)
    for (int i = tid; i < NF * kPos; i += kThreads) {
        const int o = i / kPos, jj = i - o * kPos;
        if (jj < cnt) out[((n * NF + o) * tc + t) * (int64_t)w1 + j_base + jj] = obuf[o * kOutS + jj];
    }
}

}  // namespace

// Wc: [NF][C][kw] (the correlation filter, conv1_kernel's) -> [C][3 planes][2 filter halves][64 lanes][8] bf16
void dcs_conv1_mfma_pack(const float* Wc, int NF, int C, int kw, std::vector<uint16_t>* out) {
    out->assign((size_t)C * 3 * 2 * 64 * 8, 0);
    for (int c = 0; c < C; ++c)
        for (int h = 0; h < 2; ++h)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = (lane & 15) + 16 * h, u = 8 * (lane >> 4) + j;
                    float r = (o < NF && u < kw) ? Wc[((size_t)o * C + c) * kw + u] : 0.f;
                    for (int p = 0; p < 3; ++p) {
                        uint32_t bits;
                        memcpy(&bits, &r, 4);
                        bits &= 0xffff0000u;
                        float part;
                        memcpy(&part, &bits, 4);
                        r -= part;
                        (*out)[((((size_t)c * 3 + p) * 2 + h) * 64 + lane) * 8 + j] = (uint16_t)(bits >> 16);
                    }
                }
}

// false: shape not covered, nothing launched.  bias: [NF]
bool dcs_launch_conv1_mfma(dcs_ctx* ctx, const float* x, const void* Wq, const float* bias, float* out, int64_t n, int C,
                           int NF, int tc, int F, int kw, int sw, int w1) {
    static const bool on = !(getenv("DCS_CONV1_MFMA") && atoi(getenv("DCS_CONV1_MFMA")) == 0);
    if (!on || !Wq || sw != 4 || kw > 32 || NF > 32 || (C != 1 && C != 4) || n * tc > 0x7fffffff) return false;
    if (n <= 0) return true;
    const dim3 grid((unsigned)(n * tc), (unsigned)dcs_cdiv(w1, kPos));
    const u32x4* wq = reinterpret_cast<const u32x4*>(Wq);
    if (C == 1)
        hipLaunchKernelGGL((conv1_mfma_kernel<1>), grid, dim3(kThreads), 0, ctx->stream, x, wq, bias, out, NF, tc, F, w1);
    else
        hipLaunchKernelGGL((conv1_mfma_kernel<4>), grid, dim3(kThreads), 0, ctx->stream, x, wq, bias, out, NF, tc, F, w1);
    return true;
}
