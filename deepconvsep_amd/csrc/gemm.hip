// fp32 MFMA GEMM over gathered rows (gfx950, v_mfma_f32_16x16x4_f32).
//
// Every dense / convolution layer of the reference's build_ca graph whose receptive field is a
// contiguous run of an activation matrix is this one kernel with a different row mapping:
//   conv1 (1 x F filter,      separate_dsd.py:198)  rows = frames,      A row = spectrogram row
//   conv2 (tc/2 x 1 filter,   separate_dsd.py:202)  rows = positions,   A row = kh consecutive H1 rows
//   DenseLayer bottleneck     (separate_dsd.py:206)  rows = tiles,       A row = H2 consecutive C2 rows
//   DenseLayer per source     (separate_dsd.py:209)  rows = tiles
// Arithmetic is exact f32 (the f32-input MFMA is an fmaf chain), accumulation order = k ascending.
//
// Tiling: workgroup = 4 wavefronts; block tile (16*RB) x 64, K step 32; wave w owns output columns
// [16w, 16w+16) of the block and RB row blocks (B fragment reused RB times).  A and B tiles are
// staged through LDS with 16-byte global loads, prefetched one K tile ahead in registers.
#include "dcs_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int BN = 64;
constexpr int BS = BN + 16;  // B tile row stride: rows kq, kq+1 land 16 banks apart

// BK = 32: many workgroups, several resident per CU hide each other's K-tile latency.
// BK = 128: few workgroups (the 32-tile batch): one memory latency per 128 of K instead of per 32.
template <int RB, int BK, bool VEC>
__global__ __launch_bounds__(kThreads) void gemm_rows_kernel(const DcsGemm g) {
    constexpr int AS = BK + 2;  // A tile row stride (floats): 2*i + kq distinct banks for the fragment read
    constexpr int BM = 16 * RB;
    constexpr int A_F4 = BM * (BK / 4);                        // float4 slots in an A tile
    constexpr int A_PER = (A_F4 + kThreads - 1) / kThreads;    // per thread
    constexpr int B_PER = (BK * BN / 4) / kThreads;            // = 2
    __shared__ __attribute__((aligned(16))) float lds[BM * AS + BK * BS];
    float* As = lds;
    float* Bs = lds + BM * AS;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    // K-split launches (gridDim.z slices): the row x column tiles of ONE K slice read the same A columns and B rows, so
    // they must run on one XCD, close in time, for a slice to come from HBM once and from that XCD's L2 afterwards.
    // Workgroups are dealt to the 8 XCDs round-robin by linear id: XCD x takes a contiguous run of (slice, tile) pairs.
    // (Counters before, Bach10 bottleneck 167 x 166 650 x 256 as 3 x 4 tiles x 64 slices: 929 MB moved for 282 MB of
    // operands -- A fetched by each of the 4 column tiles, B by each of the 3 row tiles, every one on another XCD.)
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.partial && g.xcd_slices) {
        const unsigned tiles = gridDim.x * gridDim.y, total = tiles * gridDim.z;
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned xcd = lin & 7, slot = lin >> 3, base = total >> 3, rem = total & 7;
        const unsigned virt = xcd * base + (xcd < rem ? xcd : rem) + slot;     // XCD x owns base + (x < rem) virtual ids
        bz = virt / tiles;
        const unsigned t = virt - bz * tiles;
        by = t / gridDim.x;
        bx = t - by * gridDim.x;
    }
    const int64_t m0 = (int64_t)bx * BM;
    const int n0 = by * BN;
    // kernel arguments as scalars (taking the struct's address would spill it to scratch)
    const int gK = g.K, gldb = g.ldb;
    const int64_t gM = g.M;
    const float gscale = g.a_scale;

    // per-thread A source rows
    const float* a_ptr[A_PER];
    int a_c4[A_PER];
    int a_row[A_PER];
    bool a_ok[A_PER];
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
        const int idx = tid + u * kThreads;
        const int row = idx / (BK / 4);
        a_row[u] = row;
        a_c4[u] = idx % (BK / 4);
        const int64_t r = m0 + row;
        a_ok[u] = (idx < A_F4) && (r < gM);
        const int64_t rr = a_ok[u] ? r : 0;
        a_ptr[u] = g.A + (g.a_rowmap ? (int64_t)g.a_rowmap[rr] : dcs_group_row(rr, g.a_gdiv, g.a_gmul, g.a_gdiv >= g.M)) * g.lda;
    }
    const float* b_ptr[B_PER];
    int b_row[B_PER], b_c4[B_PER];
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
        const int idx = tid + u * kThreads;
        b_row[u] = idx / (BN / 4);
        b_c4[u] = idx % (BN / 4);
        b_ptr[u] = g.B + (int64_t)b_row[u] * gldb + n0 + b_c4[u] * 4;
    }

    f32x4 ra[A_PER];  // native vectors: HIP's float4 struct did not survive SROA here (scratch)
    f32x4 rb[B_PER];
    // (macros, not lambdas: by-reference captures of ra/rb pushed the prefetch registers to scratch)
#define DCS_LOAD_TILES(kt_)                                                                   \
    {                                                                                         \
        const int k0_ = (kt_) * BK;                                                           \
        _Pragma("unroll") for (int u = 0; u < A_PER; ++u) {                                   \
            const int k = k0_ + a_c4[u] * 4;                                                  \
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                              \
            if (a_ok[u]) {                                                                    \
                if (VEC) {                                                                    \
                    if (k < gK) v = *reinterpret_cast<const f32x4*>(a_ptr[u] + k);            \
                } else {                                                                      \
                    if (k + 0 < gK) v[0] = a_ptr[u][k + 0];                                   \
                    if (k + 1 < gK) v[1] = a_ptr[u][k + 1];                                   \
                    if (k + 2 < gK) v[2] = a_ptr[u][k + 2];                                   \
                    if (k + 3 < gK) v[3] = a_ptr[u][k + 3];                                   \
                }                                                                             \
            }                                                                                 \
            ra[u] = v;                                                                        \
        }                                                                                     \
        _Pragma("unroll") for (int u = 0; u < B_PER; ++u)                                     \
            rb[u] = *reinterpret_cast<const f32x4*>(b_ptr[u] + (int64_t)k0_ * gldb);          \
    }
#define DCS_STORE_TILES()                                                                     \
    {                                                                                         \
        _Pragma("unroll") for (int u = 0; u < A_PER; ++u) {                                   \
            if (tid + u * kThreads < A_F4) {                                                  \
                float* d = As + a_row[u] * AS + a_c4[u] * 4;                                  \
                *reinterpret_cast<float2*>(d) = make_float2(gscale * ra[u][0], gscale * ra[u][1]);    \
                *reinterpret_cast<float2*>(d + 2) = make_float2(gscale * ra[u][2], gscale * ra[u][3]); \
            }                                                                                 \
        }                                                                                     \
        _Pragma("unroll") for (int u = 0; u < B_PER; ++u)                                     \
            *reinterpret_cast<f32x4*>(Bs + b_row[u] * BS + b_c4[u] * 4) = rb[u];              \
    }

    f32x4 acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K tiles of this workgroup: all of them, or slice blockIdx.z of a K-split launch (kchunk % BK == 0)
    const int kt0 = g.partial ? (int)bz * (g.kchunk / BK) : 0;
    int nkt = (gK + BK - 1) / BK;
    if (g.partial && kt0 + g.kchunk / BK < nkt) nkt = kt0 + g.kchunk / BK;
    if (kt0 < nkt) DCS_LOAD_TILES(kt0)
    for (int kt = kt0; kt < nkt; ++kt) {
        __syncthreads();
        DCS_STORE_TILES()
        __syncthreads();
        if (kt + 1 < nkt) DCS_LOAD_TILES(kt + 1)
        if constexpr (BK == 128) {
        // The few-workgroup shapes (BK = 128: one workgroup per CU, one wave per SIMD).  Operands of EIGHT k steps are read
        // from LDS at a time, the next eight while the MFMAs of these run.  Left to itself the compiler emitted read /
        // s_waitcnt lgkmcnt(0) / MFMA per step and every MFMA waited out an LDS latency -- the K tile of 128 cost ~3 100
        // cycles for 1 024 of MFMAs.  Same products in the same order.  (NOT for the many-workgroup shapes below: there
        // other waves hide the latency, and the 80 extra registers of RB = 4 cost occupancy -- Bach10's bottleneck layer
        // 0.167 -> 0.206 ms with this loop.)
        constexpr int KG = 8, NKG = (BK / 4) / KG;
        float bq[2][KG], aq[2][RB][KG];
#define DCS_READ_GROUP(g_)                                                                    \
        _Pragma("unroll") for (int i = 0; i < KG; ++i) {                                      \
            const int kk = (g_) * KG + i;                                                     \
            bq[(g_) & 1][i] = Bs[(kk * 4 + kq) * BS + wave * 16 + fi];                        \
            _Pragma("unroll") for (int r = 0; r < RB; ++r)                                    \
                aq[(g_) & 1][r][i] = As[(r * 16 + fi) * AS + kk * 4 + kq];                    \
        }
        DCS_READ_GROUP(0)
#pragma unroll
        for (int g_ = 0; g_ < NKG; ++g_) {
            if (g_ + 1 < NKG) DCS_READ_GROUP(g_ + 1)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KG; ++i)
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[g_ & 1][r][i], bq[g_ & 1][i], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DCS_READ_GROUP
        } else {
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const float b = Bs[(kk * 4 + kq) * BS + wave * 16 + fi];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float a = As[(r * 16 + fi) * AS + kk * 4 + kq];
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r], 0, 0, 0);
            }
        }
        }
    }

    // epilogue: C/D layout of the 16x16 MFMA: column = lane & 15, row = (lane >> 4) * 4 + reg
    const int col = n0 + wave * 16 + fi;
    if (g.partial) {   // raw sums of this K slice; gemm_ksplit_reduce_kernel adds the slices, bias, rectifier
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + r * 16 + kq * 4 + e;
                if (row < g.M) g.partial[((int64_t)bz * g.M + row) * g.n_cols + col] = acc[r][e];
            }
    } else if (col < g.n_store) {
        const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + r * 16 + kq * 4 + e;
                if (row < g.M) {
                    float v = acc[r][e] + bias;
                    if (g.relu) v = fmaxf(v, 0.f);
                    g.C[dcs_group_row(row, g.c_gdiv, g.c_gmul, g.c_gdiv >= g.M) * g.ldc + col] = v;
                }
            }
        }
    }
}

#undef DCS_LOAD_TILES
#undef DCS_STORE_TILES

// ------------------------------------------------------------------------------------------------
// Few-rows variant (the 32-tile batch: 12 row groups for conv1, 2 for the dense layers).  With so
// few workgroups a kernel's duration is the serial MFMA chain of one wave, so the chain is cut
// instead of the traffic: workgroup = one 16 x 16 output tile, the 4 waves take interleaved
// 64-wide slices of K, partial accumulators are reduced through LDS.  Operand fragments are
// loaded straight from global memory (each element is used by exactly one wave, so LDS staging
// would buy nothing); K is visited in the order k = kc + 16 j + 4 kq + e so that a lane's A values
// of four consecutive steps are one 16-byte load.  Two accumulators alternate to stay issue-bound
// (the 16x16x4 f32 MFMA has a 40-cycle dependent latency but a 32-cycle issue interval).
// ------------------------------------------------------------------------------------------------
// BF: B from DcsGemm::Bfrag (fragment order: one 16-byte load per lane and 16 K).
template <bool BF>
__global__ __launch_bounds__(kThreads) void gemm_rows_splitk_kernel(const DcsGemm g) {
    __shared__ float red[3 * 64 * 4];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * 16;
    const int n0 = blockIdx.y * 16;
    const int gK = g.K, gldb = g.ldb;
    const float gscale = g.a_scale;
    const int64_t r = m0 + fi;
    const bool row_ok = r < g.M;
    const int64_t rr = row_ok ? r : 0;
    const float* a_ptr = g.A + (g.a_rowmap ? (int64_t)g.a_rowmap[rr] : dcs_group_row(rr, g.a_gdiv, g.a_gmul, g.a_gdiv >= g.M)) * g.lda + 4 * kq;
    const float* b_ptr = g.B + (int64_t)(4 * kq) * gldb + n0 + fi;
    const f32x4* bf_ptr = BF ? reinterpret_cast<const f32x4*>(g.Bfrag) + (int64_t)blockIdx.y * ((gK + 15) >> 4) * 64 + lane : nullptr;

    f32x4 a_cur[4], a_nxt[4];
    float b_cur[16], b_nxt[16];
#define DCS_SK_LOAD(kc_, A_, B_)                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        const int k = (kc_) + 16 * j;                                                             \
        A_[j] = (row_ok && k + 4 * kq < gK) ? *reinterpret_cast<const f32x4*>(a_ptr + k)          \
                                            : f32x4{0.f, 0.f, 0.f, 0.f};                          \
        if (BF) {                                                                                 \
            const f32x4 bv_ = bf_ptr[(int64_t)(k >> 4) * 64];                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) B_[4 * j + e] = bv_[e];                 \
        } else {                                                                                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) B_[4 * j + e] = b_ptr[(int64_t)(k + e) * gldb]; \
        }                                                                                         \
    }
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    // K range of this workgroup: all of K, or slice blockIdx.z of the K-split launch (kchunk % 256 == 0)
    const int k_begin = g.partial ? (int)blockIdx.z * g.kchunk : 0;
    const int k_end = g.partial ? (k_begin + g.kchunk < gK ? k_begin + g.kchunk : gK) : gK;
    int kc = k_begin + wave * 64;
    if (kc < k_end) DCS_SK_LOAD(kc, a_cur, b_cur)
    for (; kc < k_end; kc += 256) {
        const int kn = kc + 256;
        if (kn < k_end) DCS_SK_LOAD(kn, a_nxt, b_nxt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gscale * a_cur[j][0], b_cur[4 * j + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gscale * a_cur[j][1], b_cur[4 * j + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gscale * a_cur[j][2], b_cur[4 * j + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gscale * a_cur[j][3], b_cur[4 * j + 3], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) a_cur[j] = a_nxt[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) b_cur[j] = b_nxt[j];
    }
#undef DCS_SK_LOAD
    const f32x4 acc = acc0 + acc1;
    if (wave > 0) *reinterpret_cast<f32x4*>(red + ((wave - 1) * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wave == 0) {
        f32x4 sum = acc;
#pragma unroll
        for (int w = 0; w < 3; ++w) sum += *reinterpret_cast<const f32x4*>(red + (w * 64 + lane) * 4);
        const int col = n0 + fi;
        if (g.partial) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + kq * 4 + e;
                if (row < g.M) g.partial[((int64_t)blockIdx.z * g.M + row) * g.n_cols + col] = sum[e];
            }
        } else if (col < g.n_store) {
            const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + kq * 4 + e;
                if (row < g.M) {
                    float v = sum[e] + bias;
                    if (g.relu) v = fmaxf(v, 0.f);
                    g.C[dcs_group_row(row, g.c_gdiv, g.c_gmul, g.c_gdiv >= g.M) * g.ldc + col] = v;
                }
            }
        }
    }
}

// second pass of the K-split launch: slices added in slice order (deterministic), then bias / rectify / store
__global__ __launch_bounds__(kThreads) void gemm_ksplit_reduce_kernel(const DcsGemm g, int ksplit) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= g.M * g.n_cols) return;
    const int64_t row = idx / g.n_cols;
    const int col = (int)(idx - row * g.n_cols);
    if (col >= g.n_store) return;
    // eight slices requested at a time (the additions stay in slice order): one thread per output with one load in flight
    // was a chain of ksplit memory latencies -- 14.6 us for 53 slices of 84 x 256
    float v = 0.f;
    const int64_t zs = g.M * (int64_t)g.n_cols;
    const float* p = g.partial + idx;
    int z = 0;
    for (; z + 8 <= ksplit; z += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[(z + u) * zs];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; z < ksplit; ++z) v += p[z * zs];
    v += g.bias ? g.bias[col] : 0.f;
    if (g.relu) v = fmaxf(v, 0.f);
    g.C[dcs_group_row(row, g.c_gdiv, g.c_gmul, g.c_gdiv >= g.M) * g.ldc + col] = v;
}

template <int RB, int BK>
void launch_rb(dcs_ctx* ctx, const DcsGemm& g) {
    dim3 grid((unsigned)dcs_cdiv(g.M, 16 * RB), (unsigned)(g.n_cols / BN));
    if (g.a_vec)
        hipLaunchKernelGGL((gemm_rows_kernel<RB, BK, true>), grid, dim3(kThreads), 0, ctx->stream, g);
    else
        hipLaunchKernelGGL((gemm_rows_kernel<RB, BK, false>), grid, dim3(kThreads), 0, ctx->stream, g);
}

}  // namespace

int dcs_launch_gemm_rows(dcs_ctx* ctx, const DcsGemm& g, int tag) {
    if (g.M <= 0) return DCS_OK;
    if (g.n_cols % BN) DCS_FAIL(DCS_EINVAL, "gemm_rows: n_cols %d not a multiple of %d", g.n_cols, BN);
    if (g.a_vec && ((g.K & 3) || (g.lda & 3))) DCS_FAIL(DCS_EINVAL, "gemm_rows: vector path needs K, lda %% 4 == 0");
    DcsTimer tm(ctx, tag);
    if (!g.a_rowmap && dcs_launch_gemm_bf16x3(ctx, g)) {   // B available as bf16 planes and the launch fills the chip
        tm.done();
        DCS_HIP(hipGetLastError());
        return DCS_OK;
    }
    // few rows: maximise the number of workgroups; many rows: reuse each B fragment 4 times
    const int64_t groups16 = (g.M + 15) / 16;
    const int64_t col_groups = g.n_cols / BN;
    // few rows and a very long K (Bach10's bottleneck layer: 167 x 166 650 x 256): the 16 x 16 tiles alone are ~176
    // workgroups streaming 170 MB of weights at 0.24 TB/s -> also split K over workgroups, two passes
    static const int ks_env = getenv("DCS_GEMM_KSPLIT") ? atoi(getenv("DCS_GEMM_KSPLIT")) : -1;   // 0 disables
    const int64_t tiles16 = groups16 * (g.n_cols / 16);
    if (g.a_vec && ks_env != 0 && g.K >= 16384 && tiles16 < 4 * (int64_t)ctx->n_cu) {
        if (ks_env < 0 && dcs_launch_gemm_bf16x3_longk(ctx, g)) {   // 128 .. 176 rows and B as bf16 planes: the matrix pipe's K-split
            tm.done();
            DCS_HIP(hipGetLastError());
            return DCS_OK;
        }
        constexpr int ks_tile = 1;
        const bool tiled = ks_tile && g.M >= 48;
        const int64_t units = tiled ? dcs_cdiv(g.M, 64) * (g.n_cols / BN) : tiles16;
        int ksplit = ks_env > 0 ? ks_env : (int)(((tiled ? 3 : 8) * (int64_t)ctx->n_cu + units - 1) / units);
        if (ksplit > 64) ksplit = 64;
        int kchunk = (int)dcs_round_up((g.K + ksplit - 1) / ksplit, 256);
        ksplit = (g.K + kchunk - 1) / kchunk;
        if (ksplit > 1) {
            DCS_CHECK(ctx->gemm_ws.ensure((size_t)ksplit * g.M * g.n_cols * sizeof(float)));
            DcsGemm q = g;
            q.partial = (float*)ctx->gemm_ws.ptr;
            q.kchunk = kchunk;
            constexpr bool xcd_env = true;
            q.xcd_slices = xcd_env ? 1 : 0;
            // 64 x 64 LDS tiles reuse every operand 4x more often than the 16 x 16 register tiles (which stream A and B
            // from L2 with 4 flop/B): worth it from a few row groups on
            if (tiled)
                hipLaunchKernelGGL((gemm_rows_kernel<4, 32, true>), dim3((unsigned)dcs_cdiv(g.M, 64), (unsigned)(g.n_cols / BN),
                                                                          (unsigned)ksplit),
                                   dim3(kThreads), 0, ctx->stream, q);
            else
                hipLaunchKernelGGL(gemm_rows_splitk_kernel<false>, dim3((unsigned)groups16, (unsigned)(g.n_cols / 16), (unsigned)ksplit),
                                   dim3(kThreads), 0, ctx->stream, q);
            hipLaunchKernelGGL(gemm_ksplit_reduce_kernel, dim3((unsigned)dcs_cdiv(g.M * g.n_cols, kThreads)), dim3(kThreads),
                               0, ctx->stream, q, ksplit);
            tm.done();
            DCS_HIP(hipGetLastError());
            return DCS_OK;
        }
    }
    constexpr int force = 0;   // (RB * 1000 + BK forced a tile shape in the round-1 .. 4 experiments; the heuristic below won every time)
    if (force) {
        switch (force) {
            case 1032: launch_rb<1, 32>(ctx, g); break;
            case 1128: launch_rb<1, 128>(ctx, g); break;
            case 2032: launch_rb<2, 32>(ctx, g); break;
            case 2064: launch_rb<2, 64>(ctx, g); break;
            case 2128: launch_rb<2, 128>(ctx, g); break;
            case 4032: launch_rb<4, 32>(ctx, g); break;
            case 4064: launch_rb<4, 64>(ctx, g); break;
            default: DCS_FAIL(DCS_EINVAL, "DCS_GEMM_FORCE=%d", force);
        }
        tm.done();
        DCS_HIP(hipGetLastError());
        return DCS_OK;
    }
    constexpr int64_t sk_env = -1;
    const int64_t sk_max = sk_env >= 0 ? sk_env : (int64_t)ctx->n_cu / 2;
    if (g.a_vec && groups16 * col_groups <= sk_max) {
        dim3 grid((unsigned)groups16, (unsigned)(g.n_cols / 16));
        if (g.Bfrag) hipLaunchKernelGGL(gemm_rows_splitk_kernel<true>, grid, dim3(kThreads), 0, ctx->stream, g);
        else hipLaunchKernelGGL(gemm_rows_splitk_kernel<false>, grid, dim3(kThreads), 0, ctx->stream, g);
    } else if (groups16 * col_groups <= 2 * (int64_t)ctx->n_cu)
        launch_rb<1, 128>(ctx, g);
    else if (groups16 * col_groups <= 4 * (int64_t)ctx->n_cu)
        launch_rb<1, 32>(ctx, g);
    else if (groups16 * col_groups <= 16 * (int64_t)ctx->n_cu)
        launch_rb<2, 32>(ctx, g);
    else
        launch_rb<4, 32>(ctx, g);
    tm.done();
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}
