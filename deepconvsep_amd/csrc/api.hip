// C-ABI entry points that are not tied to the network: context, framing integers, STFT plan,
// tiling.  (Network + fused path: net.hip.)
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "dcs_internal.h"

static thread_local char g_err[512] = "";

void dcs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dcs_last_error(void) { return g_err; }
extern "C" int dcs_version(void) { return 100; }

// ------------------------------------------------------------------------------- buffers / timing
namespace {
struct GuardRegistry {
    std::mutex mu;
    std::vector<DcsBuffer*> live;
};
GuardRegistry& guard_registry() {
    static GuardRegistry r;
    return r;
}
size_t guard_bytes_env() {
    static const size_t v = getenv("DCS_WS_GUARD") ? (size_t)atoll(getenv("DCS_WS_GUARD")) / 256 * 256 : 0;
    return v;
}
int guard_poison_env() {
    static const int v = getenv("DCS_WS_POISON") ? (atoi(getenv("DCS_WS_POISON")) & 0xff) : 0xff;
    return v;
}
}  // namespace

int DcsBuffer::ensure(size_t need) {
    if (need <= bytes) return DCS_OK;
    if (ptr) {
        // other work on the stream may still read the old block
        DCS_HIP(hipDeviceSynchronize());
        release();
    }
    const size_t want = need + need / 8;
    const size_t g = guard_bytes_env();
    if (g == 0) {
        DCS_HIP(hipMalloc(&ptr, want));
        base = ptr;
        bytes = want;
        return DCS_OK;
    }
    const size_t payload = (want + 255) / 256 * 256;
    DCS_HIP(hipMalloc(&base, payload + 2 * g));
    DCS_HIP(hipMemset(base, guard_poison_env(), payload + 2 * g));
    DCS_HIP(hipDeviceSynchronize());
    ptr = (char*)base + g;
    bytes = payload;
    guard = g;
    GuardRegistry& r = guard_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    if (std::find(r.live.begin(), r.live.end(), this) == r.live.end()) r.live.push_back(this);
    return DCS_OK;
}

void DcsBuffer::release() {
    if (base) (void)hipFree(base);
    else if (ptr) (void)hipFree(ptr);
    ptr = base = nullptr;
    bytes = guard = 0;
}

DcsBuffer::~DcsBuffer() {
    if (guard_bytes_env() == 0) return;
    GuardRegistry& r = guard_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    r.live.erase(std::remove(r.live.begin(), r.live.end(), this), r.live.end());
}

long long dcs_buffers_check_guards(char* where, size_t where_len) {
    GuardRegistry& r = guard_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    long long bad = 0;
    const unsigned char poison = (unsigned char)guard_poison_env();
    std::vector<unsigned char> h;
    for (DcsBuffer* b : r.live) {
        if (!b->base || b->guard == 0) continue;
        h.resize(b->guard);
        for (int side = 0; side < 2; ++side) {
            const char* src = side == 0 ? (const char*)b->base : (const char*)b->ptr + b->bytes;
            if (hipMemcpy(h.data(), src, b->guard, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            for (size_t i = 0; i < b->guard; ++i)
                if (h[i] != poison) {
                    if (bad == 0 && where)
                        snprintf(where, where_len, "%s red zone of a %zu-byte scratch block, byte %zu (%s the block)",
                                 side == 0 ? "leading" : "trailing", b->bytes, i, side == 0 ? "counted from the red zone's start, before" : "past the end of");
                    ++bad;
                }
        }
    }
    return bad;
}

extern "C" int dcs_debug_check_guards(dcs_ctx* ctx, int64_t* n_blocks_out) {
    if (!ctx) DCS_FAIL(DCS_EINVAL, "dcs_debug_check_guards: null context");
    if (guard_bytes_env() == 0)
        DCS_FAIL(DCS_EUNSUPPORTED, "dcs_debug_check_guards: the process was started without DCS_WS_GUARD=<bytes>");
    DCS_ON_DEVICE(ctx->device);
    DCS_HIP(hipDeviceSynchronize());
    if (n_blocks_out) {
        GuardRegistry& r = guard_registry();
        std::lock_guard<std::mutex> lk(r.mu);
        int64_t n = 0;
        for (DcsBuffer* b : r.live) n += (b->base && b->guard) ? 1 : 0;
        *n_blocks_out = n;
    }
    char where[200] = "";
    const long long bad = dcs_buffers_check_guards(where, sizeof(where));
    if (bad < 0) DCS_FAIL(DCS_EHIP, "dcs_debug_check_guards: copying a red zone failed");
    if (bad > 0) DCS_FAIL(DCS_EHIP, "scratch red zones damaged: %lld bytes, first in the %s", bad, where);
    return DCS_OK;
}

int DcsUploadRing::begin(size_t bytes, void** host_out, void** dev_out) {
    if (bytes == 0) bytes = 16;
    const int slot = (int)(next++ % kSlots);
    if (ev[slot]) DCS_HIP(hipEventSynchronize(ev[slot]));   // the upload that last used this slot has run (long ago, normally)
    else DCS_HIP(hipEventCreateWithFlags(&ev[slot], hipEventDisableTiming));
    if (cap[slot] < bytes) {
        if (host[slot]) DCS_HIP(hipHostFree(host[slot]));
        host[slot] = nullptr;
        cap[slot] = 0;
        const size_t want = bytes + bytes / 4 + 256;
        DCS_HIP(hipHostMalloc(&host[slot], want, hipHostMallocDefault));
        cap[slot] = want;
    }
    DCS_CHECK(dev.ensure(bytes));   // synchronises the device when (and only when) the block has to grow
    cur = slot;
    *host_out = host[slot];
    *dev_out = dev.ptr;
    return DCS_OK;
}

int DcsUploadRing::commit(size_t bytes, hipStream_t stream) {
    if (cur < 0) DCS_FAIL(DCS_EINVAL, "upload ring: commit without begin");
    if (bytes) DCS_HIP(hipMemcpyAsync(dev.ptr, host[cur], bytes, hipMemcpyHostToDevice, stream));
    DCS_HIP(hipEventRecord(ev[cur], stream));
    cur = -1;
    return DCS_OK;
}

void DcsUploadRing::release() {
    for (int i = 0; i < kSlots; ++i) {
        if (ev[i]) { (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); }
        if (host[i]) (void)hipHostFree(host[i]);
        ev[i] = nullptr; host[i] = nullptr; cap[i] = 0;
    }
    dev.release();
    cur = -1;
}

DcsTimer::DcsTimer(dcs_ctx* c, int t) : ctx(c), tag(t), idx(0), on(t >= 0 && ((c->timing_mask >> t) & 1u)) {   // t < 0: no-op
    if (!on) return;
    if ((c->timing_seen[t]++ % (uint64_t)c->timing_stride) != 0) {
        on = false;
        return;
    }
    DcsTimingSlot& s = ctx->slots[tag];
    if (s.used == s.start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            on = false;
            return;
        }
        s.start.push_back(a);
        s.stop.push_back(b);
    }
    idx = s.used++;
    (void)hipEventRecord(s.start[idx], ctx->stream);
}

void DcsTimer::done() {
    if (on) (void)hipEventRecord(ctx->slots[tag].stop[idx], ctx->stream);
}

// nothing was launched after all: give the slot back (only valid while it is the tag's most recent one)
void DcsTimer::cancel() {
    if (!on) return;
    ctx->slots[tag].used = idx;
    on = false;
}

extern "C" int dcs_timing_enable(dcs_ctx* ctx, unsigned tag_mask) {
    if (!ctx) DCS_FAIL(DCS_EINVAL, "dcs_timing_enable: null ctx");
    ctx->timing_mask = tag_mask;
    return DCS_OK;
}

extern "C" int dcs_timing_stride(dcs_ctx* ctx, int stride) {
    if (!ctx || stride < 1) DCS_FAIL(DCS_EINVAL, "dcs_timing_stride: bad argument");
    ctx->timing_stride = stride;
    return DCS_OK;
}

extern "C" int dcs_timing_reset(dcs_ctx* ctx) {
    if (!ctx) DCS_FAIL(DCS_EINVAL, "dcs_timing_reset: null ctx");
    for (auto& s : ctx->slots) s.used = 0;
    for (auto& v : ctx->timing_seen) v = 0;
    return DCS_OK;
}

extern "C" int dcs_timing_query(dcs_ctx* ctx, int which, double* avg_ms, int64_t* launches) {
    if (!ctx || which < 0 || which >= DCS_TAG_COUNT) DCS_FAIL(DCS_EINVAL, "dcs_timing_query: bad argument");
    DCS_ON_DEVICE(ctx->device);
    DCS_HIP(hipStreamSynchronize(ctx->stream));
    DcsTimingSlot& s = ctx->slots[which];
    double total = 0.0;
    for (size_t i = 0; i < s.used; ++i) {
        float ms = 0.f;
        DCS_HIP(hipEventElapsedTime(&ms, s.start[i], s.stop[i]));
        total += ms;
    }
    if (avg_ms) *avg_ms = s.used ? total / (double)s.used : 0.0;
    if (launches) *launches = (int64_t)s.used;
    return DCS_OK;
}

// ------------------------------------------------------------------------------- context
extern "C" int dcs_create(int device, void* hip_stream, dcs_ctx** out) {
    if (!out) DCS_FAIL(DCS_EINVAL, "dcs_create: out is null");
    int count = 0;
    DCS_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) DCS_FAIL(DCS_EINVAL, "dcs_create: device %d of %d", device, count);
    DCS_ON_DEVICE(device);   // the caller's current device is restored on return
    hipDeviceProp_t prop;
    DCS_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        DCS_FAIL(DCS_EUNSUPPORTED, "dcs_create: device %d is %s; libdcs is built for gfx950 only", device,
                 prop.gcnArchName);
    dcs_ctx* c = new dcs_ctx();
    c->device = device;
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->n_cu = prop.multiProcessorCount;
    *out = c;
    return DCS_OK;
}

extern "C" int dcs_destroy(dcs_ctx* ctx) {
    if (!ctx) return DCS_OK;
    DCS_ON_DEVICE(ctx->device);
    for (auto& s : ctx->slots) {
        for (auto e : s.start) (void)hipEventDestroy(e);
        for (auto e : s.stop) (void)hipEventDestroy(e);
    }
    ctx->gemm_ws.release();
    ctx->score_ring.release();
    if (ctx->ola_rise_d) (void)hipFree(ctx->ola_rise_d);
    delete ctx;
    return DCS_OK;
}


extern "C" int dcs_synchronize(dcs_ctx* ctx) {
    if (!ctx) DCS_FAIL(DCS_EINVAL, "dcs_synchronize: null ctx");
    DCS_ON_DEVICE(ctx->device);
    DCS_HIP(hipStreamSynchronize(ctx->stream));
    return DCS_OK;
}

// ------------------------------------------------------------------------------- framing integers
extern "C" int64_t dcs_frame_count(int64_t n_samples, int hop) {
    if (hop <= 0 || n_samples < 0) return -1;
    // int(np.ceil(lengthData / np.double(hopsize)) + 2)        transform.py:309
    return (n_samples + hop - 1) / hop + 2;
}

extern "C" int64_t dcs_inverse_length(int64_t n_frames, int hop, int frame) {
    if (n_frames <= 0 || hop <= 0 || frame <= 0) return -1;
    // int(hopsize*(numberFrames-1) + lengthWindow) - int(lengthWindow/2.0)   transform.py:373,390
    return (int64_t)hop * (n_frames - 1) + frame - frame / 2;
}

extern "C" int64_t dcs_tile_count(int64_t n_frames, int tc, int ov, int tiler) {
    if (tc <= 0 || ov < 0 || ov >= tc) return -1;
    const int64_t guard = (tiler == DCS_TILER_SCRIPT) ? tc : ov;
    const int64_t stride = tc - ov;
    // tiles while start + guard < T, start = i * stride
    if (n_frames <= guard) return 0;
    return (n_frames - guard - 1) / stride + 1;
}

// ------------------------------------------------------------------------------- STFT plan
extern "C" int dcs_stft_plan(dcs_ctx* ctx, int frame, int hop, const double* window_h, dcs_stft** out) {
    if (!ctx || !window_h || !out) DCS_FAIL(DCS_EINVAL, "dcs_stft_plan: null argument");
    if (frame < 16 || frame > 8192 || (frame & (frame - 1)))
        DCS_FAIL(DCS_EUNSUPPORTED, "dcs_stft_plan: frameSize %d is not a power of two in [16, 8192]", frame);
    if (hop <= 0 || hop > frame) DCS_FAIL(DCS_EINVAL, "dcs_stft_plan: hopSize %d not in (0, %d]", hop, frame);
    DCS_ON_DEVICE(ctx->device);
    dcs_stft* p = new dcs_stft();
    p->ctx = ctx;
    p->frame = frame;
    p->hop = hop;
    int M = frame / 2, lg = 0;
    while ((1 << lg) < M) ++lg;
    p->log2m = lg;

    std::vector<float> wf(frame), wsqf(frame);
    std::vector<double> wsqd(frame);
    for (int i = 0; i < frame; ++i) {
        wf[i] = (float)window_h[i];
        wsqd[i] = window_h[i] * window_h[i];
        wsqf[i] = (float)wsqd[i];
    }
    std::vector<double2> twd(M + 1);
    std::vector<float2> twf(M + 1);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k <= M; ++k) {
        double a = -2.0 * pi * (double)k / (double)frame;
        twd[k].x = cos(a);
        twd[k].y = sin(a);
    }
    // exact values on the axes (keeps DC / Nyquist bins exactly real like numpy's rfft)
    twd[0].x = 1.0; twd[0].y = 0.0;
    twd[M].x = -1.0; twd[M].y = 0.0;
    if (M % 2 == 0) { twd[M / 2].x = 0.0; twd[M / 2].y = -1.0; }
    for (int k = 0; k <= M; ++k) {
        twf[k].x = (float)twd[k].x;
        twf[k].y = (float)twd[k].y;
    }
    auto up = [&](void** dst, const void* src, size_t bytes) -> int {
        DCS_HIP(hipMalloc(dst, bytes));
        DCS_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return DCS_OK;
    };
    int rc = DCS_OK;
    if ((rc = up((void**)&p->win_f, wf.data(), frame * sizeof(float))) ||
        (rc = up((void**)&p->win_d, window_h, frame * sizeof(double))) ||
        (rc = up((void**)&p->wsq_f, wsqf.data(), frame * sizeof(float))) ||
        (rc = up((void**)&p->wsq_d, wsqd.data(), frame * sizeof(double))) ||
        (rc = up((void**)&p->tw_f, twf.data(), (M + 1) * sizeof(float2))) ||
        (rc = up((void**)&p->tw_d, twd.data(), (M + 1) * sizeof(double2)))) {
        dcs_stft_plan_destroy(p);
        return rc;
    }
    *out = p;
    return DCS_OK;
}

extern "C" int dcs_stft_plan_destroy(dcs_stft* p) {
    if (!p) return DCS_OK;
    DCS_ON_DEVICE(p->ctx->device);
    (void)hipFree(p->win_f);
    (void)hipFree(p->win_d);
    (void)hipFree(p->wsq_f);
    (void)hipFree(p->wsq_d);
    (void)hipFree(p->tw_f);
    (void)hipFree(p->tw_d);
    delete p;
    return DCS_OK;
}

template <typename R, typename R2>
static int forward_checked(dcs_stft* p, const R* audio, int64_t L, R* mag, R* phase, int64_t ld, int64_t rows_out,
                           int (*launch)(dcs_stft*, const R*, int64_t, R*, R*, R2*, int64_t, int64_t, int64_t)) {
    if (!p || !mag || (!audio && L > 0)) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward: null argument");
    if (L < 0) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward: negative length");
    DCS_ON_DEVICE(p->ctx->device);
    const int64_t T = dcs_frame_count(L, p->hop);
    if (ld < p->frame / 2 + 1) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward: ld %lld < bins %d", (long long)ld, p->frame / 2 + 1);
    if (rows_out < T) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward: rows_out %lld < frames %lld", (long long)rows_out, (long long)T);
    return launch(p, audio, L, mag, phase, (R2*)nullptr, ld, rows_out, T);
}

extern "C" int dcs_stft_forward_f32(dcs_stft* p, const float* audio_d, int64_t n, float* mag_d, float* phase_d,
                                    int64_t ld, int64_t rows_out) {
    return forward_checked<float, float2>(p, audio_d, n, mag_d, phase_d, ld, rows_out, dcs_launch_stft_forward_f32);
}
extern "C" int dcs_stft_forward_f64(dcs_stft* p, const double* audio_d, int64_t n, double* mag_d, double* phase_d,
                                    int64_t ld, int64_t rows_out) {
    return forward_checked<double, double2>(p, audio_d, n, mag_d, phase_d, ld, rows_out, dcs_launch_stft_forward_f64);
}

// compute_transform (transform.py:80-131): every column of audio[t, i] -- n_clips signals of equal length -- in ONE launch
static int forward_clips_check(dcs_stft* p, const void* audio, int64_t L, int64_t n_clips, int64_t clip_stride, const void* mag,
                               int64_t ld, int64_t rows_out, int64_t* T) {
    if (!p || !mag || (!audio && L > 0)) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward_clips: null argument");
    if (L < 0 || n_clips < 0) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward_clips: negative size");
    if (n_clips > 1 && clip_stride < L) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward_clips: clip stride %lld < length %lld", (long long)clip_stride, (long long)L);
    *T = dcs_frame_count(L, p->hop);
    if (ld < p->frame / 2 + 1) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward: ld %lld < bins %d", (long long)ld, p->frame / 2 + 1);
    if (rows_out < *T) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward: rows_out %lld < frames %lld", (long long)rows_out, (long long)*T);
    if (rows_out * n_clips > 0x7fffffffLL) DCS_FAIL(DCS_EINVAL, "dcs_stft_forward_clips: %lld rows in one launch", (long long)(rows_out * n_clips));
    return DCS_OK;
}
extern "C" int dcs_stft_forward_f32_clips(dcs_stft* p, const float* audio_d, int64_t n, int64_t n_clips, int64_t clip_stride,
                                          float* mag_d, float* phase_d, int64_t ld, int64_t rows_out) {
    int64_t T = 0;
    DCS_CHECK(forward_clips_check(p, audio_d, n, n_clips, clip_stride, mag_d, ld, rows_out, &T));
    if (n_clips == 0) return DCS_OK;
    DCS_ON_DEVICE(p->ctx->device);
    return dcs_launch_stft_forward_f32_clips(p, audio_d, n, clip_stride, n_clips, mag_d, phase_d, nullptr, ld, rows_out, T, false);
}
extern "C" int dcs_stft_forward_f64_clips(dcs_stft* p, const double* audio_d, int64_t n, int64_t n_clips, int64_t clip_stride,
                                          double* mag_d, double* phase_d, int64_t ld, int64_t rows_out) {
    int64_t T = 0;
    DCS_CHECK(forward_clips_check(p, audio_d, n, n_clips, clip_stride, mag_d, ld, rows_out, &T));
    if (n_clips == 0) return DCS_OK;
    DCS_ON_DEVICE(p->ctx->device);
    return dcs_launch_stft_forward_f64_clips(p, audio_d, n, clip_stride, n_clips, mag_d, phase_d, ld, rows_out, T);
}

template <typename R, typename R2>
static int inverse_checked(dcs_stft* p, const R* mag, int64_t src_stride, const R* phase, int64_t ld, int64_t T,
                           int n_src, R pre_div, R* audio, int64_t n_out,
                           int (*launch)(dcs_stft*, const R*, int64_t, const R*, const R2*, int64_t, int64_t, int, R, R*,
                                         int64_t)) {
    if (!p || !mag || !phase || !audio) DCS_FAIL(DCS_EINVAL, "dcs_stft_inverse: null argument");
    if (T <= 0 || n_src <= 0) DCS_FAIL(DCS_EINVAL, "dcs_stft_inverse: empty input");
    DCS_ON_DEVICE(p->ctx->device);
    if (ld < p->frame / 2 + 1) DCS_FAIL(DCS_EINVAL, "dcs_stft_inverse: ld < bins");
    if (n_out < 0 || n_out > dcs_inverse_length(T, p->hop, p->frame))
        DCS_FAIL(DCS_EINVAL, "dcs_stft_inverse: n_out %lld exceeds %lld", (long long)n_out,
                 (long long)dcs_inverse_length(T, p->hop, p->frame));
    if (pre_div == R(0)) DCS_FAIL(DCS_EINVAL, "dcs_stft_inverse: pre_div is zero");
    return launch(p, mag, src_stride, phase, (const R2*)nullptr, ld, T, n_src, pre_div, audio, n_out);
}

extern "C" int dcs_stft_inverse_f32(dcs_stft* p, const float* mag_d, int64_t src_stride, const float* phase_d,
                                    int64_t ld, int64_t T, int n_src, float pre_div, float* audio_d, int64_t n_out) {
    return inverse_checked<float, float2>(p, mag_d, src_stride, phase_d, ld, T, n_src, pre_div, audio_d, n_out,
                                  dcs_launch_stft_inverse_f32);
}
extern "C" int dcs_stft_inverse_f64(dcs_stft* p, const double* mag_d, int64_t src_stride, const double* phase_d,
                                    int64_t ld, int64_t T, int n_src, double pre_div, double* audio_d,
                                    int64_t n_out) {
    return inverse_checked<double, double2>(p, mag_d, src_stride, phase_d, ld, T, n_src, pre_div, audio_d, n_out,
                                   dcs_launch_stft_inverse_f64);
}

// ------------------------------------------------------------------------------- tiling
extern "C" int dcs_tile(dcs_ctx* ctx, const float* mag_d, int64_t ch_stride, int64_t ld, int C, int64_t T, int F,
                        int tc, int ov, int tiler, float scale, float* tiles_d, int64_t n_tiles) {
    if (!ctx || !mag_d || (!tiles_d && n_tiles > 0)) DCS_FAIL(DCS_EINVAL, "dcs_tile: null argument");
    if (C < 1 || F < 1 || ld < F || tc < 1 || ov < 0 || ov >= tc) DCS_FAIL(DCS_EINVAL, "dcs_tile: bad shape");
    if (tiler != DCS_TILER_SCRIPT && tiler != DCS_TILER_LIBRARY) DCS_FAIL(DCS_EINVAL, "dcs_tile: bad tiler");
    if (tiler == DCS_TILER_SCRIPT && C != 1)
        DCS_FAIL(DCS_EINVAL, "dcs_tile: the script tiler takes a [T,F] spectrogram (separate_dsd.py:119)");
    if (n_tiles != dcs_tile_count(T, tc, ov, tiler))
        DCS_FAIL(DCS_EINVAL, "dcs_tile: n_tiles %lld != %lld", (long long)n_tiles,
                 (long long)dcs_tile_count(T, tc, ov, tiler));
    if (n_tiles == 0) return DCS_OK;
    DCS_ON_DEVICE(ctx->device);
    return dcs_launch_tile(ctx, mag_d, ch_stride, ld, C, T, F, tc, ov, tiler, scale, tiles_d, n_tiles);
}

extern "C" int dcs_overlap_add(dcs_ctx* ctx, const float* out_d, int64_t n, int S, int tc, int ov, int F,
                               const double* rise_h, float* sep_d, int64_t sep_stride, int64_t ld) {
    if (!ctx || !out_d || !sep_d || (!rise_h && ov > 0)) DCS_FAIL(DCS_EINVAL, "dcs_overlap_add: null argument");
    // overlap == 0 is legal in the reference (util.py:306-325 with an empty ramp): the tiles are laid end to end
    if (n < 0 || S < 1 || tc < 1 || ov < 0 || ov >= tc || F < 1 || ld < F)
        DCS_FAIL(DCS_EINVAL, "dcs_overlap_add: bad shape");
    if (ov > 256) DCS_FAIL(DCS_EUNSUPPORTED, "dcs_overlap_add: overlap > 256");
    DCS_ON_DEVICE(ctx->device);
    std::vector<float> rise(ov);
    for (int i = 0; i < ov; ++i) rise[i] = (float)rise_h[i];
    if (ov > 0 && (rise != ctx->ola_rise_h || !ctx->ola_rise_d)) {
        // a new ramp: earlier launches on the stream may still read the old table
        DCS_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->ola_rise_d) (void)hipFree(ctx->ola_rise_d);
        ctx->ola_rise_d = nullptr;
        ctx->ola_rise_h.clear();
        DCS_HIP(hipMalloc((void**)&ctx->ola_rise_d, ov * sizeof(float)));
        DCS_HIP(hipMemcpy(ctx->ola_rise_d, rise.data(), ov * sizeof(float), hipMemcpyHostToDevice));
        ctx->ola_rise_h = rise;
    }
    return dcs_launch_overlap_add(ctx, out_d, n, S, tc, ov, F, ctx->ola_rise_d, sep_d, sep_stride, ld);
}
