// conv1 / conv2 of the DSD graph at launch-group sizes (3 720 x 1 028 x 50 and 3 440 x 780 x 50 for 20 x 32 tiles;
// separate_dsd.py:198-203) with K split TWICE: over workgroups (grid z: every slice leaves raw sums in an array of its own, the
// next layer adds them while it loads its operand -- the hand-over of the one-batch kernels, dsd_lat.hip) and over the waves of a
// workgroup (one K block of 32 per wave, partial blocks added through LDS in wave order).  bf16 matrix pipe, f32-class results.
//
// Same contract as gemm_rows_kernel (gemm.hip): C[row(r)][0..n_store) = act(a_scale * A[arow(r)][0..K) . B[K][n_cols] + bias),
// with  A = sum of g.a_parts arrays (a_part_stride apart)  and, for nz = gridDim.z > 1, C replaced by nz arrays c_part_stride
// apart holding the raw sums of the slices (slice 0 carries the bias; no rectifier).
//
// Why this shape (round 6, in-kernel timelines profiles/r06_c_gemm_ks_first_form_ab_and_timeline.txt): a CU fills its L1 from L2 at
// ~25 bytes per clock whatever the kernel does (22 requests of 1 KB took a wave 3 300 - 4 800 clocks to ISSUE, the four waves of
// a SIMD one after the other), and a workgroup that multiplies 16 rows by ALL of B pulls all of B: 396 KB of bf16 planes (263 KB
// as f32) = 7 - 10 us per workgroup, 233 of them at once -- the 12.7 us of gemm_rows_kernel and of a first K-over-waves kernel
// alike.  What bounds the layer is therefore  (rows / rows per workgroup) x |B| + |A|  bytes through the L1s, spread over the CUs:
//   * 64 rows per workgroup (RB = 4 row blocks per wave: every B fragment meets four A fragments) and a QUARTER of K: a
//     workgroup pulls 99 KB of B and 66 KB of A instead of 396 + 66, and 58 x 4 = 232 workgroups still cover the chip;
//   * no operand goes through LDS: an A fragment (row fi, 8 consecutive k) is two 16-byte loads of the f32 row, split exactly
//     into three bf16 terms in registers; a B fragment is 1 KB of consecutive bytes of the planes dcs_gemm_pack_bq laid out
//     (Bq[k block][plane][column][4 pieces]); every request of a wave leaves before its first product (sched_barrier: left
//     alone the compiler sinks each load to its use and the wave's chain becomes one memory round trip per fragment).
// Arithmetic: x = hi + mid + lo exactly (truncation), the six products above 2^-24 accumulated in f32, smallest first; a slice's
// blocks are added in wave order, the slices in slice order by the consumer: deterministic.
#include <stdlib.h>

#include "dcs_internal.h"

#ifdef DCS_KS_TRACE
// in-kernel timeline (scripts/build_exp.sh kstrace gemm_ks.hip -DDCS_KS_TRACE=<K of the layer>; scripts/gpu_ks_trace.py): s_memtime
// stamps of lane 0 of every wave of the middle workgroup, [wave][8], read back through ks_trace_dump()
__device__ unsigned long long ks_trace_buf[16 * 8];
#define KS_STAMP(slot)                                                                                      \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (g.K == DCS_KS_TRACE && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && wave < 16) \
            ks_trace_buf[wave * 8 + (slot)] = __builtin_amdgcn_s_memtime();                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)
#else
#define KS_STAMP(slot)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf_trunc(float x) { return __float_as_uint(x) & 0xffff0000u; }

__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4& hi, u32x4& mid, u32x4& lo) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? x0[j] : x1[j - 4];
        h[j] = bf_trunc(x);
        const float r1 = x - __uint_as_float(h[j]);      // exact
        m[j] = bf_trunc(r1);
        l[j] = bf_trunc(r1 - __uint_as_float(m[j]));     // exact difference, at most 8 significant bits left
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

// KS waves = KS K blocks per slice; NCB column blocks and RB row blocks per wave; PARTS: arrays the operand is the sum of (1 | 4)
template <int KS, int NCB, int RB, int PARTS>
__global__ __launch_bounds__(64 * KS) void gemm_ks_kernel(const DcsGemm g) {
    constexpr int NBLK = RB * NCB;                       // 16 x 16 output blocks of the workgroup
    extern __shared__ f32x4 red[];                       // [KS][NBLK][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    KS_STAMP(0);    // wave started
    const int64_t m0 = (int64_t)blockIdx.x * (16 * RB);
    const int n0 = (int)blockIdx.y * (NCB * 16);
    const int z = blockIdx.z, nz = gridDim.z;
    const int gK = g.K, n_cols = g.n_cols;
    const int nkt = (gK + 31) >> 5;
    const int kt = z * KS + wave;                        // this wave's K block
    const bool live = kt < nkt;
    const int ktc = live ? kt : nkt - 1;
    // rows past M read row M - 1 (never stored); k past K reads the last four of the row and is multiplied by 0
    const int k0 = ktc * 32 + 8 * kq;
    const float s0 = (live && k0 < gK) ? g.a_scale : 0.f, s1 = (live && k0 + 4 < gK) ? g.a_scale : 0.f;
    const int ka = k0 < gK ? k0 : gK - 4, kb = k0 + 4 < gK ? k0 + 4 : gK - 4;
    f32x4 ra[PARTS][RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int64_t r = m0 + rb * 16 + fi < g.M ? m0 + rb * 16 + fi : g.M - 1;
        const unsigned ru = (unsigned)r, gd = (unsigned)g.a_gdiv;           // (rows fit 31 bits: checked by the launcher)
        const int64_t arow = g.a_gdiv >= g.M ? r : (int64_t)(ru / gd) * g.a_gmul + (ru % gd);
        const float* a_row = g.A + arow * g.lda;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) {
            ra[p][rb][0] = *reinterpret_cast<const f32x4*>(a_row + p * g.a_part_stride + ka);
            ra[p][rb][1] = *reinterpret_cast<const f32x4*>(a_row + p * g.a_part_stride + kb);
        }
    }
    const u32x4* bq = reinterpret_cast<const u32x4*>(g.Bq) + ((int64_t)ktc * 3 * n_cols + n0 + fi) * 4 + kq;
    const int64_t b_plane = (int64_t)n_cols * 4;
    u32x4 rb_[NCB][3];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int p = 0; p < 3; ++p) rb_[cb][p] = bq[p * b_plane + cb * 64];
    // every request above leaves before the first split
    __builtin_amdgcn_sched_barrier(0);
    KS_STAMP(1);    // A and B fragments requested

    f32x4 acc[RB][NCB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a[RB][3];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x4 x0 = ra[0][rb][0], x1 = ra[0][rb][1];
#pragma unroll
        for (int p = 1; p < PARTS; ++p) {                // the producer's slices, in slice order
            x0 += ra[p][rb][0];
            x1 += ra[p][rb][1];
        }
        split8(x0 * s0, x1 * s1, a[rb][0], a[rb][1], a[rb][2]);
    }
    KS_STAMP(2);    // operands landed and split
    // smallest terms first; the RB * NCB accumulator chains alternate so that no MFMA waits for the one before it
#define DCS_KS_TERM(pa_, pb_)                                                               \
    _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                       \
        _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mma(a[rb][pa_], rb_[cb][pb_], acc[rb][cb]);
    DCS_KS_TERM(2, 0) DCS_KS_TERM(0, 2) DCS_KS_TERM(1, 1) DCS_KS_TERM(1, 0) DCS_KS_TERM(0, 1) DCS_KS_TERM(0, 0)
#undef DCS_KS_TERM
    KS_STAMP(3);    // multiplied
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) red[(wave * NBLK + rb * NCB + cb) * 64 + lane] = acc[rb][cb];
    KS_STAMP(4);    // partial blocks written
    __syncthreads();
    KS_STAMP(5);    // barrier passed
    // block j of the workgroup (row block j / NCB, column block j % NCB) is summed in wave order by wave j % KS
    float* Cz = g.C + (nz > 1 ? (int64_t)z * g.c_part_stride : 0);
    const bool finish = nz == 1;                         // one slice: bias and rectifier here; else raw sums, slice 0 + bias
    for (int j = wave; j < NBLK; j += KS) {
        f32x4 sum = red[j * 64 + lane];
#pragma unroll
        for (int w = 1; w < KS; ++w) sum += red[(w * NBLK + j) * 64 + lane];
        const int rb = j / NCB, cb = j - rb * NCB;
        const int col = n0 + cb * 16 + fi;               // C/D layout of the 16x16 MFMA: column = lane & 15, row = (lane >> 4) * 4 + reg
        if (col < g.n_store) {
            const float bias = (g.bias && (finish || z == 0)) ? g.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + rb * 16 + kq * 4 + e;
                if (row < g.M) {
                    float v = sum[e] + bias;
                    if (finish && g.relu) v = fmaxf(v, 0.f);
                    Cz[dcs_group_row(row, g.c_gdiv, g.c_gmul, g.c_gdiv >= g.M) * g.ldc + col] = v;
                }
            }
        }
    }
    KS_STAMP(6);    // stores issued
}

template <int KS, int NCB, int RB, int PARTS>
int launch(dcs_ctx* ctx, const DcsGemm& g, int nz) {
    auto kern = gemm_ks_kernel<KS, NCB, RB, PARTS>;
    const size_t lds = (size_t)KS * RB * NCB * 64 * 16;
    static DcsOncePerDevice once;
    DCS_CHECK(once.run(ctx->device, [&]() -> int {
        DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        return DCS_OK;
    }));
    const dim3 grid((unsigned)dcs_cdiv(g.M, 16 * RB), (unsigned)dcs_cdiv(g.n_cols, NCB * 16), (unsigned)nz);
    hipLaunchKernelGGL(kern, grid, dim3(64 * KS), lds, ctx->stream, g);
    return DCS_OK;
}

}  // namespace

// Slices (grid z) the launch would use, 0 when dcs_launch_gemm_ks would not take it: the caller sizes the slice arrays
// (c_part_stride) and tells the consumer how many to add.  For the A-streaming layers of the encoder at launch-group sizes only:
// >= 512 rows, exactly 64 columns, B as planes, rows of A 16-byte aligned, K of 22 .. 36 blocks of 32 (conv1 at frameSize 2048:
// 33, conv2: 25), no rectifier.
int dcs_gemm_ks_slices(const DcsGemm& g) {
    static const int on = getenv("DCS_GEMM_KS") ? atoi(getenv("DCS_GEMM_KS")) : 1;    // 0: off (A/B against gemm.hip)
    if (!on || !g.Bq || !g.a_vec || g.partial || g.a_rowmap || (g.K & 3) || (g.lda & 3) || g.M < 512 || g.M > 0x7fffffff || g.n_cols != 64 ||
        g.a_gdiv < 1 || (g.a_parts > 1 && g.a_parts != 4) || g.relu)
        return 0;
    const int nkt = (g.K + 31) / 32;
    if (nkt < 22 || nkt > 36) return 0;                  // exactly four slices of 7 / 8 / 9 K blocks (the consumers add one or four arrays)
    return 4;
}

// true when the launch was taken (g.c_part_stride must be set when dcs_gemm_ks_slices(g) > 1); *rc: its status
bool dcs_launch_gemm_ks(dcs_ctx* ctx, const DcsGemm& g, int* rc) {
    const int nz = dcs_gemm_ks_slices(g);
    *rc = DCS_OK;
    if (nz < 1 || (nz > 1 && g.c_part_stride <= 0)) return false;
    const int nkt = (g.K + 31) / 32;
    const bool parts = g.a_parts == 4;
    static const int rb_env = getenv("DCS_GEMM_KS_RB") ? atoi(getenv("DCS_GEMM_KS_RB")) : 0;
    const int rb = (rb_env == 2 || rb_env == 4) ? rb_env : (parts ? 2 : 4);
#define DCS_KS_GO(KS_)                                                                              \
    *rc = parts ? (rb == 4 ? launch<KS_, 4, 4, 4>(ctx, g, nz) : launch<KS_, 4, 2, 4>(ctx, g, nz))   \
                : (rb == 4 ? launch<KS_, 4, 4, 1>(ctx, g, nz) : launch<KS_, 4, 2, 1>(ctx, g, nz))
    if (nkt <= 28) { DCS_KS_GO(7); }
    else if (nkt <= 32) { DCS_KS_GO(8); }
    else { DCS_KS_GO(9); }
#undef DCS_KS_GO
    return true;
}

#ifdef DCS_KS_TRACE
extern "C" __attribute__((visibility("default"))) int ks_trace_dump(unsigned long long* out, int n) {
    if (!out || n < 128) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ks_trace_buf), 128 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif
