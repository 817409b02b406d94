// conv1 of the Bach10 / score-informed graphs (30 filters of 1 x 30, frequency stride 4; separate_bach10.py:190-193,
// bach10_scoreinformed/separate_bach10.py:404-407) on the bf16 matrix pipe, f32-class.
//
//   out[n][o][t][j] = bias[o] + sum_c sum_u x[n][c][t][4 j + u] * Wc[o][c][u]
//
// conv1_reg_kernel (generic.hip) does this on the vector ALU: 900 multiply-adds per position and input channel, 0.33 ms for
// the 4-channel score-informed batch (28 % of the vector peak).  As a GEMM the K axis of a channel is its 30 taps padded
// to 32 = ONE MFMA K block, and with a stride of 4 a position's taps start 16 bytes after its neighbour's: a workgroup
// splits the 1 056 input floats that 256 positions of one (tile, frame) row can touch exactly into three bf16 terms
// (truncation; six products kept, as everywhere in this library) on their way into LDS, a lane (position fi, kg) reads its
// taps 8 kg .. 8 kg + 7 of every plane as two ds_read_b64 and issues 12 MFMAs per (16 positions, channel) against weight
// fragments that stay in registers.  The accumulators (rows = positions) go through an LDS tile so that
// the stores are rows of up to 256 consecutive positions of one filter, not 64-byte pieces.
#include <string.h>

#include "dcs_internal.h"
#include "generic.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // rows are F floats apart, F odd
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

// three bf16 planes of four consecutive floats (truncation, x = hi + mid + lo exactly), packed two per word in memory order
__device__ __forceinline__ void split4(const f32x4& x, u32x2& hi, u32x2& mid, u32x2& lo) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = bf_trunc(x[j]);
        const float r1 = x[j] - __uint_as_float(h[j]);
        m[j] = bf_trunc(r1);
        l[j] = bf_trunc(r1 - __uint_as_float(m[j]));
    }
    hi = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
    mid = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
    lo = u32x2{(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
}

// Round 5: the input floats are split ONCE, when they enter LDS (three bf16 planes, [plane][channel][float index]); a lane's A
// fragment of a plane is then 16 consecutive bytes of that plane -- two ds_read_b64 (the start is 8-byte aligned: position
// stride 4 floats = 8 bytes of bf16).  Before, every lane split the eight floats of its fragment itself: a float sits in the
// fragments of eight (position, K piece) pairs, so the split ran eight times per float -- 1 024 of the ~1 900 vector
// instructions of a wave, next to a per-float load loop with a division by 1 056; the kernel was bound by instruction issue at
// 0.46 of the f32 peak (score-informed batch: 0.197 ms).  Same terms, same products, same order: bit-identical results.
//
// CL: the output channels-last, out[n][t][j][o] (NF floats per position) -- the layout colconv_fwd_x3_kernel (conv2 of the
// f32-class Bach10 / score-informed graphs) reads a position's channels from; the tile in LDS is then [position][36].
// OUT = 2 (round 6): the channels-last output as f16 with 32 channels per position (30 + 2 zeros: the padded filters have zero
// weights and biases), out16[n][t][j][32] halves -- what the f16 forward conv2 (colconv_wreg_scatter_kernel<..., IN16>) reads as one
// 16-byte load per lane and row; half the bytes written here, and a quarter of the load instructions there.
template <int C, int OUT /* 0 channel-first f32 | 1 channels-last f32 | 2 channels-last f16 x 32 */>
__global__ __launch_bounds__(kThreads, C > 1 ? 3 : 4) void conv1_mfma_kernel(const float* __restrict__ x, const u32x4* __restrict__ Wq,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int NF, int tc, int F, int w1) {
    constexpr bool CL = OUT != 0;
    constexpr int kQ = kInW / 4;                                  // float4 chunks per channel
    constexpr int kPl = C * kInW / 4;                             // u32x2 units (4 bf16) per plane
    // the planes [plane][channel][chunk] and, once every wave is done with them, the output tile [32 filters][kOutS] in the
    // same bytes (a third barrier instead of 33 KB more LDS: three workgroups per CU instead of two)
    constexpr int kClS = 36;                                      // floats per position of the channels-last tile (32 + 4: the
                                                                  // four K-piece groups of a wave write 16 banks apart)
    constexpr int kTileBytes = CL ? kPos * kClS * 4 : 32 * kOutS * 4;
    constexpr int kLdsBytes = 3 * kPl * 8 > kTileBytes ? 3 * kPl * 8 : kTileBytes;
    __shared__ __attribute__((aligned(16))) char lds_raw[kLdsBytes];
    u32x2* xpl = reinterpret_cast<u32x2*>(lds_raw);
    float* obuf = reinterpret_cast<float*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const int fi = lane & 15, kq = lane >> 4;
    const int64_t nt = blockIdx.x;            // n * tc + t
    const int64_t n = nt / tc;
    const int t = (int)(nt - n * tc);
    const int j_base = blockIdx.y * kPos;
    const int cnt = w1 - j_base < kPos ? w1 - j_base : kPos;      // positions of this workgroup
    // input first (the longest wait): the floats [4 j_base, 4 j_base + kInW) of the C rows, zero past the row
    const int f0 = 4 * j_base;
    constexpr int kIt = (kQ + kThreads - 1) / kThreads;
    f32x4 raw[C][kIt];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float* row = x + ((n * C + c) * tc + t) * (int64_t)F + f0;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int q = tid + it * kThreads, k = 4 * q;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < kQ) {
                if (f0 + k + 3 < F) {
                    v = *reinterpret_cast<const f32x4u*>(row + k);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (f0 + k + e < F) v[e] = row[k + e];
                }
            }
            raw[c][it] = v;
        }
    }
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
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int q = tid + it * kThreads;
            if (q < kQ) {
                u32x2 hi, mid, lo;
                split4(raw[c][it], hi, mid, lo);
                xpl[c * kQ + q] = hi;
                xpl[kPl + c * kQ + q] = mid;
                xpl[2 * kPl + c * kQ + q] = lo;
            }
        }
    __syncthreads();
    const int n_blk = (cnt + 15) >> 4;
    constexpr int kBw = kPos / 16 / 4;                            // column blocks per wave
    f32x4 acc[kBw][2];
#pragma unroll
    for (int bi = 0; bi < kBw; ++bi) {
        const int b = wave + 4 * bi;
        f32x4 acc0 = f32x4{b0, b0, b0, b0}, acc1 = f32x4{b1, b1, b1, b1};
        if (b < n_blk) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            // floats 4 (16 b + fi) + 8 kq .. + 7 of channel c = chunks (16 b + fi) + 2 kq and the next one
            const u32x2* xp = xpl + c * kQ + (16 * b + fi) + 2 * kq;
            u32x4 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const u32x2 lo2 = xp[p * kPl], hi2 = xp[p * kPl + 1];
                a[p] = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
            }
            const u32x4 a0 = a[0], a1 = a[1], a2 = a[2];
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
        }
        acc[bi][0] = acc0;
        acc[bi][1] = acc1;
    }
    __syncthreads();                                              // every wave has read its planes: the tile may overwrite them
#pragma unroll
    for (int bi = 0; bi < kBw; ++bi) {
        const int b = wave + 4 * bi;
        if (b < n_blk) {   // lane (filter fi, kq): positions 16 b + 4 kq .. + 3
            if (CL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    obuf[(16 * b + 4 * kq + e) * kClS + fi] = acc[bi][0][e];
                    obuf[(16 * b + 4 * kq + e) * kClS + 16 + fi] = acc[bi][1][e];
                }
            } else {
                *reinterpret_cast<f32x4*>(obuf + fi * kOutS + 16 * b + 4 * kq) = acc[bi][0];
                *reinterpret_cast<f32x4*>(obuf + (16 + fi) * kOutS + 16 * b + 4 * kq) = acc[bi][1];
            }
        }
    }
    __syncthreads();
    // rows of consecutive positions: a wave instruction writes 256 contiguous bytes of one filter's row.  (Groups of four
    // positions on 16-byte boundaries of `out` -- the alignment depends on the row -- with the head as scalars were slower:
    // 0.22 vs 0.19 ms on the score-informed batch.)
    if (OUT == 2) {   // 32 halves per position: thread (position jj, octet oq) converts eight filters and stores 16 bytes
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        _Float16* op16 = reinterpret_cast<_Float16*>(out) + ((n * tc + t) * (int64_t)w1 + j_base) * 32;
        for (int i = tid; i < 4 * cnt; i += kThreads) {
            const int jj = i >> 2, oq = i & 3;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(obuf + jj * kClS + 8 * oq);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(obuf + jj * kClS + 8 * oq + 4);
            const f16x8 h = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                             (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
            *reinterpret_cast<f16x8*>(op16 + (int64_t)jj * 32 + 8 * oq) = h;
        }
        return;
    }
    if (CL) {   // NF * cnt consecutive floats of the output
        float* op = out + ((n * tc + t) * (int64_t)w1 + j_base) * NF;
        for (int i = tid; i < NF * cnt; i += kThreads) {
            const int jj = i / NF, o = i - jj * NF;
            op[i] = obuf[jj * kClS + o];
        }
        return;
    }
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
                           int NF, int tc, int F, int kw, int sw, int w1, bool channels_last, bool out_f16) {
    static const bool on = !(getenv("DCS_CONV1_MFMA") && atoi(getenv("DCS_CONV1_MFMA")) == 0);
    if (!on || !Wq || sw != 4 || kw > 32 || NF > 32 || (C != 1 && C != 4) || n * tc > 0x7fffffff) return false;
    if (n <= 0) return true;
    const dim3 grid((unsigned)(n * tc), (unsigned)dcs_cdiv(w1, kPos));
    const u32x4* wq = reinterpret_cast<const u32x4*>(Wq);
    if (out_f16) {                                        // f16, channels-last, 32 channels per position (single-channel graphs)
        if (C != 1 || !channels_last || (reinterpret_cast<uintptr_t>(out) & 15)) return false;
        hipLaunchKernelGGL((conv1_mfma_kernel<1, 2>), grid, dim3(kThreads), 0, ctx->stream, x, wq, bias, out, NF, tc, F, w1);
    } else if (C == 1 && channels_last)
        hipLaunchKernelGGL((conv1_mfma_kernel<1, 1>), grid, dim3(kThreads), 0, ctx->stream, x, wq, bias, out, NF, tc, F, w1);
    else if (C == 1)
        hipLaunchKernelGGL((conv1_mfma_kernel<1, 0>), grid, dim3(kThreads), 0, ctx->stream, x, wq, bias, out, NF, tc, F, w1);
    else if (channels_last)
        hipLaunchKernelGGL((conv1_mfma_kernel<4, 1>), grid, dim3(kThreads), 0, ctx->stream, x, wq, bias, out, NF, tc, F, w1);
    else
        hipLaunchKernelGGL((conv1_mfma_kernel<4, 0>), grid, dim3(kThreads), 0, ctx->stream, x, wq, bias, out, NF, tc, F, w1);
    return true;
}
