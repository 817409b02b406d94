// Score-informed front-end on the device: filterSpec
// (examples/bach10_scoreinformed/separate_bach10.py:172-200) and the products mask_j * mag of :523-527.
//
// filterSpec paints, per instrument, the value 1 over the (time x harmonic-bin) rectangles of its notes on a floor
// of 1e-18 and divides by the instrument's maximum.  Two launches, no scratch: a streaming pass writes the floor
// value (x mag) everywhere, then one workgroup per rectangle overwrites its cells with the note value (x mag).
// Rectangles of one instrument may overlap; they all write the same value, so the order does not matter.
// HBM-bound: ninst * T * F * 4 B written (+ T * F * 4 read), 33.6 MB for a 10 s Bach10 file.
// The trainers' variant (LargeDatasetMask2.filterSpec, dataset.py:839-879: every bin divided by the sum over the instruments)
// takes a third pass, score_sumnorm_kernel below.
#include "dcs_internal.h"

#include <string.h>

#include <vector>

namespace {

struct ScoreRect {
    int inst, t0, t1, f0, f1;
};

__global__ __launch_bounds__(256) void score_floor_kernel(const float* __restrict__ mag, int64_t ld, int64_t T, int F,
                                                          int ninst, const float* __restrict__ lo /* [ninst] */,
                                                          float* __restrict__ out, float* __restrict__ mask,
                                                          float mag_scale) {
    const int64_t t = blockIdx.x;
    const int j = blockIdx.y;
    const float v = lo[j];
    const float* mrow = mag ? mag + t * ld : nullptr;    // mag == nullptr: the flag pass of the sum normalisation (v itself)
    for (int f = threadIdx.x; f < F; f += 256) {
        if (out) out[((int64_t)j * T + t) * F + f] = mrow ? v * (mag_scale * mrow[f]) : v;
        if (mask) mask[t * ((int64_t)ninst * F) + (int64_t)j * F + f] = v;
    }
}

__global__ __launch_bounds__(256) void score_rect_kernel(const float* __restrict__ mag, int64_t ld, int64_t T, int F,
                                                         int ninst, const ScoreRect* __restrict__ rects,
                                                         const float* __restrict__ hi /* [ninst] */,
                                                         float* __restrict__ out, float* __restrict__ mask,
                                                         float mag_scale) {
    const ScoreRect r = rects[blockIdx.x];
    const float v = hi[r.inst];
    const int w = r.f1 - r.f0;
    const int64_t cells = (int64_t)(r.t1 - r.t0) * w;
    for (int64_t c = threadIdx.x; c < cells; c += 256) {
        const int64_t t = r.t0 + c / w;
        const int f = r.f0 + (int)(c % w);
        if (out) out[((int64_t)r.inst * T + t) * F + f] = mag ? v * (mag_scale * mag[t * ld + f]) : v;
        if (mask) mask[t * ((int64_t)ninst * F) + (int64_t)r.inst * F + f] = v;
    }
}

// DCS_SCORE_NORM_SUM (LargeDatasetMask2.filterSpec, dataset.py:839-879): mask_j = filtered_j / sum_i filtered_i with filtered in
// {1e-18, 1} -- three passes over the outputs, no scratch: the floor / rectangle kernels above first leave FLAGS (0 = floor,
// 1 = inside a note rectangle) where the values will go, then this kernel turns the ninst flags of every (frame, bin) into the
// values: the float32 sum over the instruments in order (np.sum over axis 0 of a float32 array adds slice by slice), one IEEE
// division per instrument.  `flags` is out (layout [j][t][f]) when out is written, else mask ([t][j * F + f]).
__global__ __launch_bounds__(256) void score_sumnorm_kernel(const float* __restrict__ mag, int64_t ld, int64_t T, int F, int ninst,
                                                            float* __restrict__ out, float* __restrict__ mask, float mag_scale) {
    const int64_t t = blockIdx.x;
    const float* mrow = mag + t * ld;
    for (int f = threadIdx.x; f < F; f += 256) {
        unsigned on = 0;
        float total = 0.f;
        for (int j = 0; j < ninst; ++j) {
            const float flag = out ? out[((int64_t)j * T + t) * F + f] : mask[t * ((int64_t)ninst * F) + (int64_t)j * F + f];
            const float v = flag != 0.f ? 1.0f : 1e-18f;
            on |= (flag != 0.f ? 1u : 0u) << j;
            total = j == 0 ? v : total + v;
        }
        const float m = mag_scale * mrow[f];
        for (int j = 0; j < ninst; ++j) {
            const float v = (((on >> j) & 1u) ? 1.0f : 1e-18f) / total;
            if (out) out[((int64_t)j * T + t) * F + f] = v * m;
            if (mask) mask[t * ((int64_t)ninst * F) + (int64_t)j * F + f] = v;
        }
    }
}

}  // namespace

// mag_scale: the spectrogram is multiplied by it first (the scripts' scale_factor, separate_bach10.py:503; 1 = already scaled)
int dcs_score_masks_scaled(dcs_ctx* ctx, const float* mag_d, int64_t ld, int64_t n_frames, int F, const double* notes_h,
                           int ninst, int n_notes, int width, int64_t start, int64_t stop, float mag_scale, float* out_d,
                           float* mask_d, int normalise) {
    if (normalise != DCS_SCORE_NORM_MAX && normalise != DCS_SCORE_NORM_SUM)
        DCS_FAIL(DCS_EINVAL, "dcs_score_masks: normalise %d", normalise);
    if (normalise == DCS_SCORE_NORM_SUM && ninst > 32) DCS_FAIL(DCS_EUNSUPPORTED, "dcs_score_masks: sum normalisation of %d instruments", ninst);
    if (!ctx || !mag_d || !notes_h) DCS_FAIL(DCS_EINVAL, "dcs_score_masks: null argument");
    if (!out_d && !mask_d) DCS_FAIL(DCS_EINVAL, "dcs_score_masks: nothing to write");
    if (ninst < 1 || ninst > 65535 || n_notes < 0 || width < 5 || ((width - 3) & 1) || F < 1 || ld < F || n_frames < 0)
        DCS_FAIL(DCS_EINVAL, "dcs_score_masks: bad shape (ninst %d, notes %d, width %d, F %d)", ninst, n_notes, width, F);
    if (n_frames == 0) return DCS_OK;
    DCS_ON_DEVICE(ctx->device);
    // the reference's Python slicing: filtered[j, begin:end, range(start_f, stop_f)] with begin/end relative to
    // `start`; slices clip to the array, the fancy bin index would raise past F -- reported here as DCS_ESHAPE
    std::vector<ScoreRect> rects;
    std::vector<float> lo(ninst), hi(ninst);
    const int npairs = (width - 3) / 2;
    for (int j = 0; j < ninst; ++j) {
        bool any = false;
        for (int p = 0; p < n_notes; ++p) {
            const double* n = notes_h + ((size_t)j * n_notes + p) * width;
            const double n0 = n[0], n1 = n[1], midi = n[2];
            const double a = n0 > (double)start ? n0 : (double)start, b = n1 < (double)stop ? n1 : (double)stop;
            if (!(midi > 0) || !((b - a > 0 ? b - a : 0) > 0)) continue;
            int64_t t0 = (int64_t)a - start, t1 = (int64_t)b - start;
            if (t0 < 0) t0 = 0;
            if (t1 > n_frames) t1 = n_frames;
            for (int k = 0; k < npairs; ++k) {
                const double fs = n[3 + 2 * k], fe = n[4 + 2 * k];
                if (!(fe > 0)) continue;
                const int64_t f0 = (int64_t)fs, f1 = (int64_t)fe;
                if (f0 < 0 || f1 > F) DCS_FAIL(DCS_ESHAPE, "dcs_score_masks: bin range [%lld, %lld) outside 0..%d",
                                               (long long)f0, (long long)f1, F);
                if (t1 > t0 && f1 > f0) {
                    rects.push_back(ScoreRect{j, (int)t0, (int)t1, (int)f0, (int)f1});
                    any = true;
                }
            }
        }
        // filtered[j] / np.max(filtered[j]) in float32 (separate_bach10.py:194-195): an instrument without notes is
        // all floor, hence all ones after the division
        const float floor_v = 1e-18f, maxv = any ? 1.0f : floor_v;
        lo[j] = floor_v / maxv;
        hi[j] = 1.0f / maxv;
        if (normalise == DCS_SCORE_NORM_SUM) {   // flags first: the values depend on the other instruments' rectangles
            lo[j] = 0.f;
            hi[j] = 1.f;
        }
    }
    // rectangles + the two value rows travel as one block through the context's upload ring (no synchronisation here)
    const size_t b_rect = rects.size() * sizeof(ScoreRect), b_val = (size_t)ninst * sizeof(float);
    const size_t off_lo = (b_rect + 255) / 256 * 256, off_hi = off_lo + (b_val + 255) / 256 * 256;
    void *host = nullptr, *dev = nullptr;
    DCS_CHECK(ctx->score_ring.begin(off_hi + b_val, &host, &dev));
    if (b_rect) memcpy(host, rects.data(), b_rect);
    memcpy((char*)host + off_lo, lo.data(), b_val);
    memcpy((char*)host + off_hi, hi.data(), b_val);
    DCS_CHECK(ctx->score_ring.commit(off_hi + b_val, ctx->stream));
    char* base = (char*)dev;
    DcsTimer tm(ctx, DCS_TAG_SCORE);
    if (normalise == DCS_SCORE_NORM_SUM) {
        // flags go where the values will go (ONE of the two outputs, un-multiplied: mag = nullptr selects the flag form)
        float* f_out = out_d, *f_mask = out_d ? nullptr : mask_d;
        hipLaunchKernelGGL(score_floor_kernel, dim3((unsigned)n_frames, (unsigned)ninst), dim3(256), 0, ctx->stream,
                           (const float*)nullptr, ld, n_frames, F, ninst, (const float*)(base + off_lo), f_out, f_mask, 1.0f);
        if (!rects.empty())
            hipLaunchKernelGGL(score_rect_kernel, dim3((unsigned)rects.size()), dim3(256), 0, ctx->stream, (const float*)nullptr,
                               ld, n_frames, F, ninst, (const ScoreRect*)base, (const float*)(base + off_hi), f_out, f_mask, 1.0f);
        hipLaunchKernelGGL(score_sumnorm_kernel, dim3((unsigned)n_frames), dim3(256), 0, ctx->stream, mag_d, ld, n_frames, F,
                           ninst, out_d, mask_d, mag_scale);
    } else {
        hipLaunchKernelGGL(score_floor_kernel, dim3((unsigned)n_frames, (unsigned)ninst), dim3(256), 0, ctx->stream, mag_d, ld,
                           n_frames, F, ninst, (const float*)(base + off_lo), out_d, mask_d, mag_scale);
        if (!rects.empty())
            hipLaunchKernelGGL(score_rect_kernel, dim3((unsigned)rects.size()), dim3(256), 0, ctx->stream, mag_d, ld, n_frames,
                               F, ninst, (const ScoreRect*)base, (const float*)(base + off_hi), out_d, mask_d, mag_scale);
    }
    tm.done();
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

extern "C" int dcs_score_masks(dcs_ctx* ctx, const float* mag_d, int64_t ld, int64_t n_frames, int F,
                               const double* notes_h, int ninst, int n_notes, int width, int64_t start, int64_t stop,
                               float* out_d, float* mask_d) {
    return dcs_score_masks_scaled(ctx, mag_d, ld, n_frames, F, notes_h, ninst, n_notes, width, start, stop, 1.0f, out_d, mask_d,
                                  DCS_SCORE_NORM_MAX);
}

extern "C" int dcs_score_masks_norm(dcs_ctx* ctx, const float* mag_d, int64_t ld, int64_t n_frames, int F,
                                    const double* notes_h, int ninst, int n_notes, int width, int64_t start, int64_t stop,
                                    int normalise, float* out_d, float* mask_d) {
    return dcs_score_masks_scaled(ctx, mag_d, ld, n_frames, F, notes_h, ninst, n_notes, width, start, stop, 1.0f, out_d, mask_d,
                                  normalise);
}
