// GEMM over gathered rows for launches whose duration is ONE WAVE'S DEPENDENT CHAIN (the encoder of the DSD graph at launch-group
// sizes: conv1 3 720 x 1 028 x 50, conv2 3 440 x 780 x 50, bottleneck 640 x 832 x 128, per-source dense 640 x 128 x 2 400 --
// separate_dsd.py:198-221): K is split over the WAVES of a workgroup, on the bf16 matrix pipe with f32-class results.
//
// Same contract as gemm_rows_kernel (gemm.hip): C[row(r)][0..n_store) = act(a_scale * A[arow(r)][0..K) . B[K][n_cols] + bias).
//
// Why (round 6): gemm_rows_kernel gives a 16-row block to four waves that walk all of K together -- nine K tiles of 128 for
// conv1, each a global load -> LDS -> barrier -> 32 dependent f32 MFMAs of 32 cycles: 12.7 / 9.9 us for 0.5 GFLOP with < 1 wave per
// SIMD, matrix pipe 14 - 16 % busy (profiles/r05_o_*, round-5 review).  Here
//   * a workgroup owns 16 rows x (CG * CBW) column blocks of 16; its KS * CG waves are (K slice ks, column group cg): wave
//     (ks, cg) multiplies the K blocks ks, ks + KS, ... (32 of K each) of its CBW column blocks -- at most MAXB of them, all
//     requested before the first product (A) / PF blocks ahead (B): the chain is one memory latency + MAXB x 6 CBW MFMAs of 16
//     cycles, not nine round trips;
//   * no operand goes through LDS: an A fragment (row fi, 8 consecutive k) is two 16-byte loads of the f32 row, split exactly
//     into three bf16 terms in registers; a B fragment is 1 KB of consecutive bytes of the planes dcs_gemm_pack_bq laid out
//     (Bq[k block][plane][column][4 pieces]), straight from L2;
//   * the KS partial blocks are added through LDS in slice order (deterministic), bias / rectifier / store by the first
//     CG * CBW waves.
// Arithmetic: x = hi + mid + lo exactly (truncation), the six products above 2^-24 accumulated in f32, smallest first -- the class
// of the f32-MFMA kernel (tests/test_host_cpu.py::test_bf16_three_way_split_is_exact_and_six_products_are_fp32_class).
#include <stdlib.h>

#include "dcs_internal.h"

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

// KS K slices x CG column groups of CBW column blocks; MAXB >= ceil(ceil(K / 32) / KS) K blocks per wave; B fragments PF blocks ahead
template <int KS, int CG, int CBW, int MAXB, int PF>
__global__ __launch_bounds__(64 * KS * CG) void gemm_ks_kernel(const DcsGemm g) {
    static_assert(PF >= 1 && PF <= MAXB, "prefetch depth");
    static_assert(KS == 1 || KS >= CBW, "the first CBW slices of a column group own its epilogue");
    constexpr int NCB = CG * CBW;
    __shared__ f32x4 red[KS > 1 ? KS * NCB * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % CG, ks = wave / CG;
    const int fi = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * 16;
    const int n0 = ((int)blockIdx.y * CG + cg) * (CBW * 16);
    const int gK = g.K, n_cols = g.n_cols;
    const int nkt = (gK + 31) >> 5;
    const float gscale = g.a_scale;
    // rows past M read row M - 1 (their results are never stored); k past K reads the last four of the row and is multiplied by 0
    const int64_t r = m0 + fi < g.M ? m0 + fi : g.M - 1;
    // (rows fit 31 bits, checked by the launcher: the grouped-row formula in 32-bit arithmetic, a fifth of the 64-bit division)
    const unsigned ru = (unsigned)r, gd = (unsigned)g.a_gdiv;
    const int64_t arow = g.a_rowmap ? (int64_t)g.a_rowmap[r] : (g.a_gdiv >= g.M ? r : (int64_t)(ru / gd) * g.a_gmul + (ru % gd));
    const float* a_row = g.A + arow * g.lda;
    const bool col_ok = n0 < n_cols;                     // (a launch whose last column group is partly past n_cols)
    const u32x4* bq = reinterpret_cast<const u32x4*>(g.Bq) + ((int64_t)((col_ok ? n0 : 0) + fi)) * 4 + kq;
    const int64_t b_plane = (int64_t)n_cols * 4, b_kt = 3 * b_plane;

    f32x4 ra[MAXB][2];
    float sa[MAXB][2];
    u32x4 rb[PF][CBW][3];
    // every A fragment of this wave, unconditionally (clamped addresses): nothing touches the registers before their split
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int kt = ks + i * KS;
        const int k0 = (kt < nkt ? kt : nkt - 1) * 32 + 8 * kq;
        const bool ok0 = kt < nkt && k0 < gK, ok1 = kt < nkt && k0 + 4 < gK;
        ra[i][0] = *reinterpret_cast<const f32x4*>(a_row + (k0 < gK ? k0 : gK - 4));
        ra[i][1] = *reinterpret_cast<const f32x4*>(a_row + (k0 + 4 < gK ? k0 + 4 : gK - 4));
        sa[i][0] = ok0 ? gscale : 0.f;
        sa[i][1] = ok1 ? gscale : 0.f;
    }
#define DCS_KS_LOAD_B(i_)                                                                               \
    {                                                                                                   \
        const int kt_ = ks + (i_) * KS;                                                                 \
        const u32x4* bp_ = bq + (int64_t)(kt_ < nkt ? kt_ : nkt - 1) * b_kt;                            \
        _Pragma("unroll") for (int cb = 0; cb < CBW; ++cb)                                              \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) rb[(i_) % PF][cb][p] = bp_[p * b_plane + cb * 64]; \
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) DCS_KS_LOAD_B(i)
    // every request above leaves before the first split: left to itself the compiler sinks each load to its use (fewest
    // registers), which turns the wave's chain back into MAXB memory round trips
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        // a K block past the end (kt >= nkt) has zero scales: its products are exact zeros (B is finite: a clamped, valid block)
        u32x4 a0, a1, a2;
        split8(ra[i][0] * sa[i][0], ra[i][1] * sa[i][1], a0, a1, a2);
        // smallest terms first; the column blocks alternate so that no MFMA waits for the one before it
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[cb] = mma(a2, rb[i % PF][cb][0], acc[cb]);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[cb] = mma(a0, rb[i % PF][cb][2], acc[cb]);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[cb] = mma(a1, rb[i % PF][cb][1], acc[cb]);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[cb] = mma(a1, rb[i % PF][cb][0], acc[cb]);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[cb] = mma(a0, rb[i % PF][cb][1], acc[cb]);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[cb] = mma(a0, rb[i % PF][cb][0], acc[cb]);
        if (i + PF < MAXB) {                             // the slot this block just freed takes block i + PF
            __builtin_amdgcn_sched_barrier(0);
            DCS_KS_LOAD_B(i + PF)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef DCS_KS_LOAD_B
    // the KS partial blocks of a (column group, column block) are added in slice order by wave (ks = cb, cg)
    int cb_mine = 0;
    f32x4 sum;
    if constexpr (KS > 1) {
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) red[(ks * NCB + cg * CBW + cb) * 64 + lane] = acc[cb];
        __syncthreads();
        if (ks >= CBW) return;
        cb_mine = ks;
        sum = red[(cg * CBW + cb_mine) * 64 + lane];
#pragma unroll
        for (int s = 1; s < KS; ++s) sum += red[(s * NCB + cg * CBW + cb_mine) * 64 + lane];
    }
    // epilogue: C/D layout of the 16x16 MFMA: column = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int cbi = 0; cbi < (KS > 1 ? 1 : CBW); ++cbi) {
        const int cb = KS > 1 ? cb_mine : cbi;
        const f32x4 v4 = KS > 1 ? sum : acc[cbi];
        const int col = n0 + cb * 16 + fi;
        if (col < g.n_store) {
            const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + kq * 4 + e;
                if (row < g.M) {
                    float v = v4[e] + bias;
                    if (g.relu) v = fmaxf(v, 0.f);
                    g.C[dcs_group_row(row, g.c_gdiv, g.c_gmul, g.c_gdiv >= g.M) * g.ldc + col] = v;
                }
            }
        }
    }
}

template <int KS, int CG, int CBW, int MAXB, int PF>
void launch(dcs_ctx* ctx, const DcsGemm& g) {
    const dim3 grid((unsigned)dcs_cdiv(g.M, 16), (unsigned)dcs_cdiv(g.n_cols, CG * CBW * 16));
    hipLaunchKernelGGL((gemm_ks_kernel<KS, CG, CBW, MAXB, PF>), grid, dim3(64 * KS * CG), 0, ctx->stream, g);
}

}  // namespace

// true when the launch was taken.  For launches of the encoder's size class only: at least 512 rows (below, the one-batch
// kernels of dsd_lat.hip or the f32 split-K tiles are cut for the chain), B as planes, rows of A 16-byte aligned.
bool dcs_launch_gemm_ks(dcs_ctx* ctx, const DcsGemm& g) {
    static const int cfg = getenv("DCS_GEMM_KS") ? atoi(getenv("DCS_GEMM_KS")) : 1;    // 0: off (A/B against gemm.hip)
    if (!cfg || !g.Bq || !g.a_vec || g.partial || (g.K & 3) || (g.lda & 3) || g.K < 32 || g.M < 512 || g.M > 0x7fffffff || (g.n_cols & 63) ||
        g.a_gdiv < 1)
        return false;
    const int nkt = (g.K + 31) / 32;
    if (g.n_cols <= 128 && nkt <= 40) {                  // A-streaming layers: K over 8 slices, 64 columns per workgroup
        if (cfg == 2) launch<16, 1, 4, 3, 1>(ctx, g);
        else if (cfg == 3) launch<4, 2, 2, 10, 3>(ctx, g);
        else if (nkt <= 32) launch<8, 2, 2, 4, 2>(ctx, g);
        else launch<8, 2, 2, 5, 2>(ctx, g);
        return true;
    }
    if (nkt <= 4) {                                      // short K, wide N (the per-source dense layers): no K split, no LDS
        if (cfg == 2) launch<4, 2, 2, 1, 1>(ctx, g);
        else launch<1, 4, 2, 4, 4>(ctx, g);
        return true;
    }
    return false;
}
