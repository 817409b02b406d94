// Tiling and cross-fade overlap-add kernels (HBM-bound copies / blends).
//
// generate_overlapadd : examples/dsd100/separate_dsd.py:114-135 (script tiler), util.py:220-248
// overlapadd_multi    : util.py:297-327 (= separate_dsd.py:139-169), 2-source util.py:251-294
#include "dcs_internal.h"

namespace {

constexpr int kThreads = 256;

// tiles[i][c][j][:] = scale * mag[c][i*stride + j][:]  (zero rows past T: library tiler only;
// the script tiler never produces such rows because start + tc < T)
__global__ __launch_bounds__(kThreads) void tile_kernel(const float* __restrict__ mag, int64_t ch_stride,
                                                        int64_t ld, int C, int64_t T, int F, int tc, int stride,
                                                        float scale, float* __restrict__ tiles) {
    const int64_t i = blockIdx.x;
    const int cj = blockIdx.y;
    const int c = cj / tc, j = cj % tc;
    const int64_t t = i * stride + j;
    float* dst = tiles + ((i * C + c) * tc + j) * (int64_t)F;
    if (t < T) {
        const float* src = mag + c * ch_stride + t * ld;
        for (int f = threadIdx.x; f < F; f += kThreads) dst[f] = scale * src[f];
    } else {
        for (int f = threadIdx.x; f < F; f += kThreads) dst[f] = 0.f;
    }
}

// Frame-parallel form of the sequential stitch.  Output frame t is owned by the last tile k0 whose
// copy region [k0*st+ov, k0*st+tc) contains it (tile 0 also owns t < ov); every later tile k with
// k*st <= t then blends:  acc = fall[j]*acc + rise[j]*tile[k][j],  j = t - k*st < ov, in increasing
// k -- the same operands in the same order as util.py:321-325, so the float result is the
// sequential one.
__global__ __launch_bounds__(kThreads) void overlap_add_kernel(const float* __restrict__ out, int64_t n, int S,
                                                               int tc, int ov, int F,
                                                               const float* __restrict__ rise,
                                                               float* __restrict__ sep, int64_t sep_stride,
                                                               int64_t ld, int64_t out_src_stride) {
    const int64_t t = blockIdx.x;
    const int s = blockIdx.y;
    const int st = tc - ov;
    float* dst = sep + s * sep_stride + t * ld;
    int64_t k0 = (t < ov) ? 0 : (t - ov) / st;
    if (k0 > n - 1) k0 = n - 1;
    const int64_t j0 = (n > 0) ? t - k0 * st : tc;
    if (j0 >= tc) {  // past the last tile: the zeros of util.py:313
        for (int f = threadIdx.x; f < F; f += kThreads) dst[f] = 0.f;
        return;
    }
    const float* src = out + (int64_t)s * out_src_stride;
    for (int f = threadIdx.x; f < F; f += kThreads) {
        float acc = src[(k0 * tc + j0) * F + f];
        for (int64_t k = k0 + 1; k < n && k * st <= t; ++k) {
            const int j = (int)(t - k * st);
            // written out (a product rounded on its own, then one fused multiply-add) so that this kernel and mask_ola_kernel,
            // which blends the same operands, round identically whatever the compiler would contract
            acc = fmaf(rise[ov - 1 - j], acc, __fmul_rn(rise[j], src[(k * tc + j) * F + f]));
        }
        dst[f] = acc;
    }
}

// (audio_out * 32767).astype('int16') of the scripts (separate_dsd.py:307-309): truncation toward zero, no
// clipping -- an out-of-range product wraps the way NumPy's float -> int16 cast does on x86-64 (through int32).
__global__ __launch_bounds__(kThreads) void pcm_int16_kernel(const float* __restrict__ pcm, int64_t n,
                                                             int16_t* __restrict__ out) {
    const int64_t i0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4;
    if (i0 + 3 < n && (reinterpret_cast<uintptr_t>(pcm) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(pcm + i0);
        short4 o;
        o.x = (int16_t)(int32_t)(v.x * 32767.f);
        o.y = (int16_t)(int32_t)(v.y * 32767.f);
        o.z = (int16_t)(int32_t)(v.z * 32767.f);
        o.w = (int16_t)(int32_t)(v.w * 32767.f);
        *reinterpret_cast<short4*>(out + i0) = o;
    } else {
        for (int64_t i = i0; i < n && i < i0 + 4; ++i) out[i] = (int16_t)(int32_t)(pcm[i] * 32767.f);
    }
}

}  // namespace

extern "C" int dcs_pcm_to_int16(dcs_ctx* ctx, const float* pcm_d, int64_t n, int16_t* out_d) {
    if (!ctx || !pcm_d || !out_d || n < 0) DCS_FAIL(DCS_EINVAL, "dcs_pcm_to_int16: bad argument");
    if (n == 0) return DCS_OK;
    DCS_ON_DEVICE(ctx->device);
    hipLaunchKernelGGL(pcm_int16_kernel, dim3((unsigned)dcs_cdiv(n, (int64_t)kThreads * 4)), dim3(kThreads), 0, ctx->stream,
                       pcm_d, n, out_d);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}


int dcs_launch_tile(dcs_ctx* ctx, const float* mag, int64_t ch_stride, int64_t ld, int C, int64_t T, int F, int tc,
                    int ov, int tiler, float scale, float* tiles, int64_t n) {
    (void)tiler;
    DcsTimer tm(ctx, DCS_TAG_TILE);
    hipLaunchKernelGGL(tile_kernel, dim3((unsigned)n, (unsigned)(C * tc)), dim3(kThreads), 0, ctx->stream, mag,
                       ch_stride, ld, C, T, F, tc, tc - ov, scale, tiles);
    tm.done();
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

int dcs_launch_overlap_add(dcs_ctx* ctx, const float* out, int64_t n, int S, int tc, int ov, int F,
                           const float* rise_d, float* sep, int64_t sep_stride, int64_t ld, int64_t out_src_stride) {
    if (out_src_stride <= 0) out_src_stride = n * tc * (int64_t)F;   // dense [S][n][tc][F]
    const int64_t rows = n * (tc - ov) + tc;
    DcsTimer tm(ctx, DCS_TAG_OLA);
    hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)rows, (unsigned)S), dim3(kThreads), 0, ctx->stream, out,
                       n, S, tc, ov, F, rise_d, sep, sep_stride, ld, out_src_stride);
    tm.done();
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}
