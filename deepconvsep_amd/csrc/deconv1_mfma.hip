// InverseLayer(conv1) of the Bach10 / score-informed graphs (stride 4, 30 taps padded to 32) on the bf16 matrix pipe,
// f32-class -- the second stage of colconv_deconv1_fused_kernel (colconv_wreg.hip) as a kernel of its own, for the graphs
// the fused decoder does not cover (f32 conv2^T; the 4-channel score-informed graph).
//
//   o[c][t][f] = sum_ci sum_x g[ci][t][x] * W[ci][c][f - 4 x],  0 <= f - 4 x < 32          (deconv1_reg_kernel, generic.hip)
//
// With f = 4 q + r:  o[c][t][4 q + r] = sum_mm P_c[(mm, r)][q - mm],  P_c[(mm, r)][x] = sum_ci W[ci][c][4 mm + r] g[ci][t][x]:
// a 32 x 32 x 16 product per (channel c, row t, 16 columns x) -- A = the filter (rows (mm, r), K = ci), B = g (lane (x, kg)
// loads its eight input channels 8 kg .. 8 kg + 7 of column x and splits them exactly into three bf16 terms; six products)
// -- after which lane (x, kq) holds the 16-byte piece of the output row at f = 4 (x + mm), mm = kq (+4).  The eight pieces
// that meet at one q are summed through a per-wave LDS array (plain 16-byte writes, shifted reads), a wave walks the column
// blocks of its two rows left to right, finished q leave as 256 contiguous bytes per row and 7 carry over.  See the fused
// kernel for the derivation; scripts/emu_fused_decoder.py is the lane-level emulation the index arithmetic was checked with.
#include <string.h>

#include "dcs_internal.h"
#include "generic.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kThreads = 256;

__device__ __forceinline__ f32x4 mma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned bf_trunc(float x) { return __float_as_uint(x) & 0xffff0000u; }

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

// g: [image][NF][tc][w1]; out: [image][C][tc][F]; a wave owns the rows (t, t + 1) of one image
template <int C>
__global__ __launch_bounds__(kThreads) void deconv1_mfma_kernel(const float* __restrict__ g, const u32x4* __restrict__ Wq,
                                                                float* __restrict__ out, int NF, int tc, int F, int w1,
                                                                int64_t n_units) {
    // per wave: Pb [2 rows][8 taps mm][32 slots] float4 (slot 8 + x holds P[x][mm]; slots 0..7, 24..31 stay zero) and the
    // carry Cb [C][2 rows][8] float4
    constexpr int kPb = 2 * 8 * 32, kCb = C * 2 * 8;       // float4 units
    __shared__ f32x4 lds[4 * (kPb + kCb)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    f32x4* Pb = lds + wave * (kPb + kCb);
    f32x4* Cb = Pb + kPb;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < kPb; i += 64) Pb[i] = zero4;
    u32x4 w[C][3][2];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            w[c][p][0] = Wq[((c * 3 + p) * 2) * 64 + lane];
            w[c][p][1] = Wq[((c * 3 + p) * 2 + 1) * 64 + lane];
        }
    const int n_xb = (w1 + 15) >> 4;
    const int pairs = (tc + 1) >> 1;
    const int64_t plane = (int64_t)tc * w1;
    f32x4* pw = Pb + kq * 32 + 8 + fi;
    const int rt = lane >> 5, rq = lane & 31;
    const f32x4* pr = Pb + rt * 256 + (rq < 23 ? rq : 22) + 8;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t unit = (int64_t)blockIdx.x * 4 + wave; unit < n_units; unit += stride) {
        const int64_t img = unit / pairs;
        const int y = (int)(unit - img * pairs) * 2;
        const bool two = y + 1 < tc;                         // wave-uniform: the last pair of an odd tc has one row
        for (int i = lane; i < kCb; i += 64) Cb[i] = zero4;
        const float* gb = g + img * NF * plane + (int64_t)y * w1;
        float* ob = out + (img * C * tc + y + rt) * (int64_t)F;
        float raw[2][8];
#define DCS_FETCH(blk_)                                                                                 \
        {                                                                                               \
            const int xl_ = (blk_) * 16 + fi;                                                           \
            const int xc_ = xl_ < w1 ? xl_ : w1 - 1;                                                    \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                             \
                const int ci_ = 8 * kq + j < NF ? 8 * kq + j : NF - 1;                                  \
                raw[0][j] = gb[ci_ * plane + xc_];                                                      \
                raw[1][j] = gb[ci_ * plane + (two ? w1 : 0) + xc_];                                     \
            }                                                                                           \
        }
        DCS_FETCH(0)
        for (int b = 0; b < n_xb; ++b) {
            u32x4 gp[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) split8(raw[t], gp[t][0], gp[t][1], gp[t][2]);
            {
                const int nb = b + 1 < n_xb ? b + 1 : b;     // last block: a harmless re-read
                DCS_FETCH(nb)
            }
            const bool x_ok = b * 16 + fi < w1;              // columns past w1 (last block only) must not reach the rows
            const bool edge = b * 16 + 16 > w1;
            const int f0 = 4 * (b * 16 + rq);
#pragma unroll
            for (int c = 0; c < C; ++c) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        f32x4 p = zero4;
                        p = mma_bf(w[c][2][mh], gp[t][0], p);  // smallest products first
                        p = mma_bf(w[c][0][mh], gp[t][2], p);
                        p = mma_bf(w[c][1][mh], gp[t][1], p);
                        p = mma_bf(w[c][1][mh], gp[t][0], p);
                        p = mma_bf(w[c][0][mh], gp[t][1], p);
                        p = mma_bf(w[c][0][mh], gp[t][0], p);
                        if (edge) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) p[e] = x_ok ? p[e] : 0.f;
                        }
                        pw[t * 256 + mh * 128] = p;
                    }
                }
                asm volatile("" ::: "memory");               // both rows' pieces are written (LDS is in order per wave)
                f32x4 sum = zero4;
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) sum += pr[mm * 32 - mm];
                const f32x4 cin = Cb[(c * 2 + rt) * 8 + (rq & 7)];
                if (rq < 8) sum += cin;
                if (rq >= 16 && rq < 24) Cb[(c * 2 + rt) * 8 + rq - 16] = rq < 23 ? sum : zero4;
                if (rq < 16 && (rt == 0 || two)) {
                    float* op = ob + (int64_t)c * tc * F + f0;
                    if (f0 + 4 <= F) {
                        *reinterpret_cast<f32x4u*>(op) = sum;
                    } else {
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                            if (f0 + e < F) op[e] = sum[e];
                    }
                }
                asm volatile("" ::: "memory");
            }
        }
        // the carry: the tail of the rows and the zeros up to F
        if (lane < C * 2 * 8) {
            const int c = lane >> 4, r = (lane >> 3) & 1, q = lane & 7;
            if (r == 0 || two) {
                const f32x4 v = Cb[lane];
                const int f = 4 * (16 * n_xb + q);
                float* op = out + ((img * C + c) * tc + y + r) * (int64_t)F + f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (f + e < F) op[e] = v[e];
            }
        }
        asm volatile("" ::: "memory");
#undef DCS_FETCH
    }
}

}  // namespace

// W1p: [nf1][C][32 taps] (tap axis zero-padded) -> [C][3 planes][2 tap halves][64 lanes][8] bf16: lane (fi, kg) of tap half mh
// holds, for row (mm = fi / 4 + 4 mh, r = fi % 4) = tap 4 mm + r, the input channels 8 kg .. 8 kg + 7 of output channel c
void dcs_deconv1_mfma_pack(const float* W1p, int nf1, int C, std::vector<uint16_t>* out) {
    out->assign((size_t)C * 3 * 2 * 64 * 8, 0);
    for (int c = 0; c < C; ++c)
        for (int mh = 0; mh < 2; ++mh)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int fi = lane & 15, kg = lane >> 4;
                    const int ci = 8 * kg + j;
                    const int tap = 4 * ((fi >> 2) + 4 * mh) + (fi & 3);
                    float r = ci < nf1 ? W1p[((size_t)ci * C + c) * 32 + tap] : 0.f;
                    for (int p = 0; p < 3; ++p) {
                        uint32_t bits;
                        memcpy(&bits, &r, 4);
                        bits &= 0xffff0000u;
                        float part;
                        memcpy(&part, &bits, 4);
                        r -= part;
                        (*out)[((((size_t)c * 3 + p) * 2 + mh) * 64 + lane) * 8 + j] = (uint16_t)(bits >> 16);
                    }
                }
}

// false: shape not covered, nothing launched
bool dcs_launch_deconv1_mfma(dcs_ctx* ctx, const float* g, const void* Wq, float* out, int64_t n_images, int NF, int C, int tc,
                             int F, int w1) {
    static const bool on = !(getenv("DCS_DECONV1_MFMA") && atoi(getenv("DCS_DECONV1_MFMA")) == 0);
    if (!on || !Wq || NF > 32 || (C != 1 && C != 4) || w1 < 16 || F < 4 * w1 + 26 || F > 4 * w1 + 32) return false;
    if (n_images <= 0) return true;
    const int64_t n_units = n_images * ((tc + 1) / 2);
    const unsigned grid = (unsigned)std::min<int64_t>(dcs_cdiv(n_units, 4), (int64_t)ctx->n_cu * 3);
    const u32x4* wq = reinterpret_cast<const u32x4*>(Wq);
    if (C == 1)
        hipLaunchKernelGGL((deconv1_mfma_kernel<1>), dim3(grid), dim3(kThreads), 0, ctx->stream, g, wq, out, NF, tc, F, w1, n_units);
    else
        hipLaunchKernelGGL((deconv1_mfma_kernel<4>), dim3(grid), dim3(kThreads), 0, ctx->stream, g, wq, out, NF, tc, F, w1, n_units);
    return true;
}
