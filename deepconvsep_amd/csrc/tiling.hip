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
// clipping -- an out-of-range product wraps the way NumPy's float -> int16 cast does on x86-64 (through int32).  The scripts'
// audio_out is float64: the product is formed in double (exact for a float32 sample: 24 + 15 significant bits), so the int16
// value is the one the host path gets from `(pcm.astype(float64) * 32767).astype('int16')` -- bit for bit, which a float32
// product is not (it rounds before the truncation: one LSB off on a few samples per million).
__device__ __forceinline__ int16_t pcm_i16(float v) { return (int16_t)(int32_t)((double)v * 32767.0); }

__global__ __launch_bounds__(kThreads) void pcm_int16_kernel(const float* __restrict__ pcm, int64_t n,
                                                             int16_t* __restrict__ out) {
    const int64_t i0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4;
    if (i0 + 3 < n && (reinterpret_cast<uintptr_t>(pcm) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(pcm + i0);
        short4 o;
        o.x = pcm_i16(v.x);
        o.y = pcm_i16(v.y);
        o.z = pcm_i16(v.z);
        o.w = pcm_i16(v.w);
        *reinterpret_cast<short4*>(out + i0) = o;
    } else {
        for (int64_t i = i0; i < n && i < i0 + 4; ++i) out[i] = pcm_i16(pcm[i]);
    }
}

// wav samples -> the mono float signal the scripts separate (separate_dsd.py:275-287, separate_ikala.py:229): int16 frames
// [n][channels] interleaved as scipy.io.wavfile.read delivers them; x = sample.astype('float') / 32767 in float64, then
//   mode 0 (DSD / hiphop / Bach10): (L + R) / 2 when there are two or more channels, the only channel otherwise
//   mode 1 (iKala):                 L + R (no halving)
// and the float32 value the device path has always worked on (the cast Context.to_device applies to the float64 array).
// Clips are stacked: clip c reads in + c * in_stride (int16 elements), writes out + c * out_stride.
__global__ __launch_bounds__(kThreads) void pcm16_to_float_kernel(const int16_t* __restrict__ in, int64_t n, int ch, int mode,
                                                                  int64_t in_stride, float* __restrict__ out, int64_t out_stride) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int16_t* p = in + (int64_t)blockIdx.y * in_stride + i * ch;
    const double l = (double)p[0] / 32767.0;
    double v = l;
    if (ch > 1) {
        const double r = (double)p[1] / 32767.0;
        v = mode == 1 ? l + r : (l + r) / 2;
    }
    out[(int64_t)blockIdx.y * out_stride + i] = (float)v;
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


extern "C" int dcs_pcm16_to_float(dcs_ctx* ctx, const int16_t* pcm16_d, int64_t n_frames, int channels, int mode, int64_t n_clips,
                                  int64_t in_stride, float* out_d, int64_t out_stride) {
    if (!ctx || !pcm16_d || !out_d || n_frames < 0 || n_clips < 0) DCS_FAIL(DCS_EINVAL, "dcs_pcm16_to_float: bad argument");
    if (channels < 1 || (mode != 0 && mode != 1)) DCS_FAIL(DCS_EINVAL, "dcs_pcm16_to_float: %d channels, mode %d", channels, mode);
    if (mode == 1 && channels < 2)
        DCS_FAIL(DCS_EINVAL, "too many indices for array");   // separate_ikala.py:229 indexes [:, 1] of a mono array: IndexError
    if (n_clips > 65535) DCS_FAIL(DCS_EINVAL, "dcs_pcm16_to_float: %lld clips in one call", (long long)n_clips);
    if (n_clips > 1 && (in_stride < n_frames * channels || out_stride < n_frames))
        DCS_FAIL(DCS_EINVAL, "dcs_pcm16_to_float: clip strides shorter than a clip");
    if (n_frames == 0 || n_clips == 0) return DCS_OK;
    DCS_ON_DEVICE(ctx->device);
    hipLaunchKernelGGL(pcm16_to_float_kernel, dim3((unsigned)dcs_cdiv(n_frames, kThreads), (unsigned)n_clips), dim3(kThreads), 0,
                       ctx->stream, pcm16_d, n_frames, channels, mode, in_stride, out_d, out_stride);
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
