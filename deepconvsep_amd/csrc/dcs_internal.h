// Internal declarations shared by the libdcs translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/dcs.h"

void dcs_set_error(const char* fmt, ...);

#define DCS_FAIL(code, ...)          \
    do {                             \
        dcs_set_error(__VA_ARGS__);  \
        return (code);               \
    } while (0)

#define DCS_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            dcs_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,    \
                          __LINE__);                                                           \
            return e__ == hipErrorOutOfMemory ? DCS_ENOMEM : DCS_EHIP;                         \
        }                                                                                      \
    } while (0)

#define DCS_CHECK(expr)                 \
    do {                                \
        int rc__ = (expr);              \
        if (rc__ != DCS_OK) return rc__; \
    } while (0)

// Every public entry point runs on its context's device and leaves the calling thread's current device as it
// found it (a process may hold contexts on several GPUs; dcs_create itself must not switch the caller).
struct DcsDeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DcsDeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = (err == hipSuccess);
        }
    }
    ~DcsDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DcsDeviceGuard(const DcsDeviceGuard&) = delete;
    DcsDeviceGuard& operator=(const DcsDeviceGuard&) = delete;
};
#define DCS_ON_DEVICE(dev_)                                                                         \
    DcsDeviceGuard dcs_guard__(dev_);                                                               \
    if (dcs_guard__.err != hipSuccess)                                                              \
        DCS_FAIL(DCS_EHIP, "cannot select device %d: %s", (int)(dev_), hipGetErrorString(dcs_guard__.err))

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the CURRENT device's function object: a launcher that wants to
// pay for it once (the one-batch kernels: the host call is on the latency path) must remember it per device, thread-safely.
//   static DcsOncePerDevice once;
//   DCS_CHECK(once.run(ctx->device, [&]() -> int { DCS_HIP(hipFuncSetAttribute(...)); return DCS_OK; }));
// The device's bit is published only AFTER the attribute calls have succeeded, and they run under a mutex: a second thread
// either finds the bit set (the attributes are in place, it may launch) or waits for the first one and then finds it set; a
// failed attempt leaves the bit clear, so the next call tries again instead of launching a kernel that cannot get its LDS.
#include <mutex>
struct DcsOncePerDevice {
    unsigned long long mask = 0;   // devices 0..63; higher ordinals run `fn` every time
    std::mutex mu;
    template <typename Fn>
    int run(int device, Fn fn) {
        const bool tracked = device >= 0 && device < 64;
        const unsigned long long bit = tracked ? 1ull << device : 0;
        if (tracked && (__atomic_load_n(&mask, __ATOMIC_ACQUIRE) & bit)) return DCS_OK;
        std::lock_guard<std::mutex> lock(mu);
        if (tracked && (__atomic_load_n(&mask, __ATOMIC_ACQUIRE) & bit)) return DCS_OK;
        const int rc = fn();
        if (rc == DCS_OK && tracked) __atomic_fetch_or(&mask, bit, __ATOMIC_RELEASE);
        return rc;
    }
};

// A grow-only device scratch buffer.  Regions handed out keep their address until the
// buffer has to grow (then `generation` changes and zero-initialised regions are re-zeroed).
// Memory-safety aid (tests/test_gpu_guard.py): with DCS_WS_GUARD=<bytes> in the environment every block is allocated with
// that many bytes of red zone on either side and the WHOLE allocation (red zones and payload) is filled with the byte
// DCS_WS_POISON (default 0xFF: float / bf16 NaN, int64 -1) -- a kernel that reads scratch nobody wrote computes on
// NaNs, and dcs_debug_check_guards() finds writes outside the block.
struct DcsBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    void* base = nullptr;      // what hipMalloc returned (== ptr without red zones)
    size_t guard = 0;          // red-zone bytes on either side of [ptr, ptr + bytes)
    int ensure(size_t need);
    void release();
    ~DcsBuffer();              // leaves the guard registry (the memory itself is freed by release())
};
// number of damaged red-zone bytes over every live guarded DcsBuffer (0 = intact), < 0 on a HIP error
long long dcs_buffers_check_guards(char* where, size_t where_len);

// Small host tables (clip lengths, note rectangles) on their way to the device without a stream synchronisation in the
// call: begin() hands out a pinned host slot of a ring (waiting, if ever, for the upload that used the slot kSlots calls
// ago) and the device block, commit() enqueues ONE copy of the block and records the slot's event.  The device block is
// shared by consecutive calls -- stream order protects it, so every user must be on the same stream.
struct DcsUploadRing {
    static constexpr int kSlots = 4;
    void* host[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[kSlots] = {0, 0, 0, 0};
    hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    unsigned next = 0;
    int cur = -1;
    DcsBuffer dev;
    int begin(size_t bytes, void** host_out, void** dev_out);
    int commit(size_t bytes, hipStream_t stream);
    void release();
};

struct DcsTimingSlot {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
};

struct dcs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    unsigned timing_mask = 0;  // bit t set: kernels tagged t are bracketed by HIP events
    int timing_stride = 1;     // ... every timing_stride-th launch of the tag only
    uint64_t timing_seen[DCS_TAG_COUNT] = {0};
    DcsTimingSlot slots[DCS_TAG_COUNT];
    int n_cu = 256;
    DcsBuffer gemm_ws;         // partial sums of the K-split GEMM (grown on demand, never inside a graph capture)
    // cross-fade ramp of dcs_overlap_add, uploaded when its values change (one table per context; the fused paths keep
    // their own per model)
    float* ola_rise_d = nullptr;
    std::vector<float> ola_rise_h;
    DcsUploadRing score_ring;  // note rectangles and floor values of dcs_score_masks
};

// RAII-ish helper: records a start event on construction and a stop event in done().
struct DcsTimer {
    dcs_ctx* ctx;
    int tag;
    size_t idx;
    bool on;
    DcsTimer(dcs_ctx* c, int t);
    void done();
    void cancel();
};

static inline int64_t dcs_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int dcs_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------- FFT plan
struct dcs_stft {
    dcs_ctx* ctx = nullptr;
    int frame = 0;   // N
    int hop = 0;
    int log2m = 0;   // log2(N/2)
    // device tables
    float* win_f = nullptr;     // [N]
    double* win_d = nullptr;    // [N]
    float2* tw_f = nullptr;     // [N/2+1]  exp(-2 pi i k / N), k = 0..N/2
    double2* tw_d = nullptr;
    float* wsq_f = nullptr;     // [N] window*window in float32 (normaliser terms)
    double* wsq_d = nullptr;
};

// launchers implemented in fft.hip
// `unit` (nullable) = exp(j*angle(X)) as X/|X|, [rows, ld] complex: the fused path carries it instead
// of the angle so that neither atan2 nor sincos is evaluated.  The inverse takes `unit` when non-null,
// else `phase`.
int dcs_launch_stft_forward_f32(dcs_stft* p, const float* audio, int64_t L, float* mag, float* phase, float2* unit,
                                int64_t ld, int64_t rows_out, int64_t T);
int dcs_launch_stft_forward_f32_clips(dcs_stft* p, const float* audio, int64_t L, int64_t audio_stride, int64_t n_clips,
                                      float* mag, float* phase, float2* unit, int64_t ld, int64_t rows_out, int64_t T,
                                      bool interleave = false,   // interleave: rows ordered [frame][clip]
                                      const int64_t* clip_tab = nullptr);  // device {samples, frames, tiles} per clip
int dcs_launch_stft_inverse_f32_clips(dcs_stft* p, const float* mag, int64_t src_stride, const float2* unit,
                                      int64_t unit_clip_stride, int64_t ld, int64_t T, int n_src, int64_t n_clips,
                                      float pre_div, float* audio, int64_t n_out, const int64_t* clip_tab = nullptr,
                                      int64_t out_stride = 0);
int dcs_launch_stft_forward_f64(dcs_stft* p, const double* audio, int64_t L, double* mag, double* phase,
                                double2* unit, int64_t ld, int64_t rows_out, int64_t T);
int dcs_launch_stft_forward_f64_clips(dcs_stft* p, const double* audio, int64_t L, int64_t audio_stride, int64_t n_clips,
                                      double* mag, double* phase, int64_t ld, int64_t rows_out, int64_t T);
int dcs_launch_stft_inverse_f32(dcs_stft* p, const float* mag, int64_t src_stride, const float* phase,
                                const float2* unit, int64_t ld, int64_t T, int n_src, float pre_div, float* audio,
                                int64_t n_out);
int dcs_launch_stft_inverse_f64(dcs_stft* p, const double* mag, int64_t src_stride, const double* phase,
                                const double2* unit, int64_t ld, int64_t T, int n_src, double pre_div, double* audio,
                                int64_t n_out);

// ---------------------------------------------------------------------------------- clips of different lengths in one launch
// Device table of kDcsClipTab int64 per clip (separate_impl, net.hip): {samples, frames, tiles, row offset, tile offset, rows}.
// row offset < 0: the clips are stacked with a uniform pitch (the caller's strides apply; the generic graphs).  Otherwise the
// DSD path's compact layout: clip c owns rows [row offset, + rows) of the spectrogram-shaped buffers (mag, unit, H1, C2) and
// tiles [tile offset, + tiles) of the tile-shaped ones (Z, D, G) -- a group costs the SUM of its clips, not n x the longest.
constexpr int kDcsClipTab = 6;

// ---------------------------------------------------------------------------------- GEMM on rows
// C[row(r)][0..n_store) = act( a_scale * A[arow(r)][0..K) . B[K][ldb] + bias )
//   arow(r) = ((r / a_gdiv) * a_gmul + r % a_gdiv) * lda      (elements)
//   crow(r) = ((r / c_gdiv) * c_gmul + r % c_gdiv) * ldc
// B is padded with zero rows to a multiple of 128 (the largest K tile) and ldb is a multiple of 64.
struct DcsGemm {
    const float* A; int64_t lda; int a_gdiv; int64_t a_gmul; float a_scale;
    const int* a_rowmap;     // optional (device): arow(r) = a_rowmap[r] * lda instead of the grouped-row formula (ragged clip groups
                             // with per-clip row offsets: tile -> first row of its frames; f32 kernels of gemm.hip only)
    const float* B; int ldb;
    const float* bias;
    float* C; int64_t ldc; int c_gdiv; int64_t c_gmul;
    int64_t M; int n_cols;   // n_cols: padded N (multiple of 64) to compute
    int n_store;             // columns actually written (<= n_cols)
    int K;                   // valid K (A columns); multiple of 4 when a_vec
    int relu;
    int a_vec;               // 1: A rows are 16-byte aligned and lda % 4 == 0 -> float4 loads
    // set by the launcher only: K split over workgroups (few rows, very long K -- the 166 650-wide dense layer of
    // the Bach10 graph): slice z covers [z*kchunk, (z+1)*kchunk) and writes raw sums to partial[z][M][n_cols]
    float* partial; int kchunk;
    int xcd_slices;     // K-split launches: map (slice, tile) -> workgroup so that the tiles of a slice share an XCD (gemm_rows_kernel)
    // optional: B split into three bf16 planes by dcs_gemm_pack_bq (gemm_bf16x3.hip); launches that fill the chip then
    // run on the bf16 matrix pipe with f32-class results
    const void* Bq;
    // optional (all-rows kernel only): A already split into three bf16 planes by dcs_gemm_split_a, [k tile][plane][aq_rows][4 pieces]
    // -- the workgroups then copy their k tile's pieces into LDS instead of each splitting all the rows again
    const void* Aq; int aq_rows;
    // Bq holds the weights UNSPLIT, as 32-byte f32 pieces in the planes' piece order (dcs_gemm_pack_b32): the all-rows kernels split
    // them in registers -- 4 bytes per weight from HBM instead of 6.  Only those kernels take it (M 128 .. 176)
    int bq_f32;
    // optional (gemm_rows_splitk_kernel only): B once more in that kernel's FRAGMENT order, [n_cols / 16][(K + 15) / 16][64 lanes][4]:
    // lane (kq, fi) of column block nt and K chunk c holds B[16 c + 4 kq + e][16 nt + fi], e = 0 .. 3 (rows >= K zero) -- one
    // 16-byte load per lane and 16 K (1 KB per wave, contiguous) instead of four 4-byte loads on four 64-byte segments
    const float* Bfrag;
};

// row r of a grouped operand: (r / gdiv) * gmul + r % gdiv.  Most launches have ONE group (gdiv = 2^30 > M): a 64-bit division
// and remainder per row -- ~80 instructions, four times per thread in a GEMM epilogue, ~0.5 us of a 12 us launch with one wave
// per SIMD -- for a result that is r.  `flat` is wave-uniform (kernel arguments only).
#if defined(__HIPCC__)
__device__ __forceinline__ int64_t dcs_group_row(int64_t r, int gdiv, int64_t gmul, bool flat) {
    return flat ? r : (r / gdiv) * gmul + (r % gdiv);
}
#endif

int dcs_launch_gemm_rows(dcs_ctx* ctx, const DcsGemm& g, int tag);
size_t dcs_gemm_bq_bytes(int K, int n_cols);
// enqueued on the ctx stream; perm_c > 0: columns re-ordered from [channel perm_c][position perm_p] to [position][channel]
int dcs_gemm_pack_bq(dcs_ctx* ctx, const float* B_d, int K, int ldb, int n_cols, void* Bq_d, int perm_c = 0, int perm_p = 0);
size_t dcs_gemm_b32_bytes(int K, int n_cols);
int dcs_gemm_pack_b32(dcs_ctx* ctx, const float* B_d, int K, int ldb, int n_cols, void* B32_d, int perm_c = 0, int perm_p = 0);
bool dcs_launch_gemm_bf16x3(dcs_ctx* ctx, const DcsGemm& g);
struct DcsGemmBranches {         // (B planes, bias, C) of up to 4 GEMMs that share A and shape
    int n;
    const void* Bq[4];
    const float* bias[4];
    float* C[4];
};
bool dcs_launch_gemm_bf16x3_skinny(dcs_ctx* ctx, const DcsGemm& g, const DcsGemmBranches* br);   // false: not taken
bool dcs_launch_gemm_bf16x3_longk(dcs_ctx* ctx, const DcsGemm& g);                                   // false: not taken
void dcs_launch_gemm_longk_reduce(dcs_ctx* ctx, const DcsGemm& q, int ksplit);                       // slices of q.partial added in order, bias, rectifier
// A [M][K] f32 (lda) -> bf16 x 3 planes for DcsGemm::Aq: [(K + 31) / 32][3][rows_pad][4] 16-byte pieces, rows >= M and k >= K zero
size_t dcs_gemm_aq_bytes(int K, int rows_pad);
int dcs_gemm_split_a(dcs_ctx* ctx, const float* A_d, int64_t lda, int64_t M, int K, int rows_pad, void* Aq_d);

int dcs_score_masks_scaled(dcs_ctx* ctx, const float* mag_d, int64_t ld, int64_t n_frames, int F, const double* notes_h,
                           int ninst, int n_notes, int width, int64_t start, int64_t stop, float mag_scale, float* out_d,
                           float* mask_d, int normalise = DCS_SCORE_NORM_MAX);

// ---------------------------------------------------------------------------------- tiling kernels
int dcs_launch_tile(dcs_ctx* ctx, const float* mag, int64_t ch_stride, int64_t ld, int C, int64_t T, int F,
                    int tc, int ov, int tiler, float scale, float* tiles, int64_t n);
// out: source s of the n tiles at out + s * out_src_stride (0: dense [S][n][tc][F])
int dcs_launch_overlap_add(dcs_ctx* ctx, const float* out, int64_t n, int S, int tc, int ov, int F,
                           const float* rise_d, float* sep, int64_t sep_stride, int64_t ld, int64_t out_src_stride = 0);
