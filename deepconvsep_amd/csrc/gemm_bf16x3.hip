// GEMM over gathered rows on the bf16 matrix pipe with f32-class results (gfx950, v_mfma_f32_16x16x32_bf16).
//
// Same contract as gemm_rows_kernel (gemm.hip): C[row(r)][0..n_store) = act(a_scale * A[arow(r)][0..K) . B[K][n_cols] + bias).
// Both operands are split exactly into three bf16 terms by truncation (x = hi + mid + lo, 8 + 8 + 8 significand bits)
// and the six term products above 2^-24 are accumulated in f32:
//     a.b ~= a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)        dropped: <= 3 * 2^-24 |a b|
// bf16 x bf16 products are exact in f32, so the only roundings are the accumulator's -- the class of the f32-MFMA kernel --
// at 6 MFMAs of 16 cycles per 16 x 16 x 32 block instead of 8 of 32, with the VALU (splitting A, address arithmetic)
// running beside the matrix pipe.
//   B is split once per model by gemm_pack_bq_kernel: Bq[k tile][plane][column][4 pieces], a 16-byte piece = the 8
//     consecutive k of one column a lane multiplies, so a wave's B fragment of a (k tile, plane) is 1 KB contiguous and
//     goes from L2 straight to registers (each element is used by one wave of the workgroup: no LDS for B).
//   A rows are fetched as f32 (8 consecutive k per thread), scaled, split in registers and stored to LDS as planes
//     [plane][row][4 pieces], double buffered, one barrier per k tile; the next tile's rows and B fragments are requested
//     before the current tile is multiplied.
// Workgroup = 4 waves = 4 column blocks of 16; RB row blocks of 16 per wave (B fragments reused RB times).
#include "dcs_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
constexpr int kRowU4 = 5;   // LDS row stride in 16-byte pieces: 4 pieces + 1 (20 words: 16 rows start in 16 different bank quads)

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

// B f32 [K rows][ldb] (rows past K are read as zero) -> Bq[kt][plane][n_cols][4] pieces.  perm_c > 0: output column
// j < perm_c * perm_p is source column (j % perm_c) * perm_p + j / perm_c -- a [channel][position] column axis re-ordered
// to [position][channel] (channels-last output of a dense layer whose consumer reads a position's channels together)
__global__ __launch_bounds__(kThreads) void gemm_pack_bq_kernel(const float* __restrict__ B, int K, int ldb, int n_cols,
                                                                u32x4* __restrict__ Bq, int64_t n_pieces /* kt * n_cols * 4 */,
                                                                int perm_c, int perm_p) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;   // (kt, n, kq), n fastest within kq? -> kq fastest
    if (idx >= n_pieces) return;
    const int kq = (int)(idx & 3);
    const int64_t t = idx >> 2;
    const int n = (int)(t % n_cols);
    const int64_t kt = t / n_cols;
    const int ns = (perm_c > 0 && n < perm_c * perm_p) ? (n % perm_c) * perm_p + n / perm_c : n;
    f32x4 x0, x1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t k = kt * 32 + kq * 8 + j;
        const float v = k < K ? B[k * ldb + ns] : 0.f;
        if (j < 4) x0[j] = v; else x1[j - 4] = v;
    }
    u32x4 p0, p1, p2;
    split8(x0, x1, p0, p1, p2);
    const int64_t base = (kt * 3 * n_cols + n) * 4 + kq;
    Bq[base] = p0;
    Bq[base + (int64_t)n_cols * 4] = p1;
    Bq[base + (int64_t)n_cols * 8] = p2;
}

// The same pieces UNSPLIT (round 6): B32[kt][column][4 pieces of 8 consecutive k] as f32, 32 bytes per piece -- 4 bytes per weight
// instead of the 6 of three bf16 planes.  For the all-rows kernels, whose time is the HBM stream of the weights (1 GB of planes
// for the Bach10 per-source layers): they split the piece in registers, beside the matrix pipe; same truncation, same planes,
// bit-identical products.
__global__ __launch_bounds__(kThreads) void gemm_pack_b32_kernel(const float* __restrict__ B, int K, int ldb, int n_cols,
                                                                 f32x4* __restrict__ B32, int64_t n_pieces, int perm_c, int perm_p) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;   // (kt, n, kq), kq fastest
    if (idx >= n_pieces) return;
    const int kq = (int)(idx & 3);
    const int64_t t = idx >> 2;
    const int n = (int)(t % n_cols);
    const int64_t kt = t / n_cols;
    const int ns = (perm_c > 0 && n < perm_c * perm_p) ? (n % perm_c) * perm_p + n / perm_c : n;
    f32x4 x0, x1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t k = kt * 32 + kq * 8 + j;
        const float v = k < K ? B[k * ldb + ns] : 0.f;
        if (j < 4) x0[j] = v; else x1[j - 4] = v;
    }
    B32[idx * 2] = x0;
    B32[idx * 2 + 1] = x1;
}

template <int RB>
__global__ __launch_bounds__(kThreads) void gemm_bf16x3_kernel(const DcsGemm g, int xcd_map) {
    constexpr int BM = 16 * RB;
    constexpr int kPlane = BM * kRowU4;                 // pieces per plane
    constexpr int kBuf = 3 * kPlane;
    __shared__ u32x4 As[2 * kBuf];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    // Workgroups are dealt to the 8 XCDs round-robin by their linear id.  The row tiles that share a 64-column B slab
    // (3 for Bach10: 167 rows, 1 GB of B planes) must run on ONE XCD, back to back, for the slab to be read from HBM
    // once and from that XCD's L2 afterwards: XCD x takes a contiguous run of (column group, row tile) pairs.
    const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7, slot = lin >> 3, base = total >> 3, rem = total & 7;
    const unsigned virt = xcd_map ? xcd * base + (xcd < rem ? xcd : rem) + slot : lin;   // XCD x owns base + (x < rem) ids
    const unsigned bx = virt % gridDim.x, by = virt / gridDim.x;
    const int64_t m0 = (int64_t)bx * BM;
    const int n0 = by * 64;
    const int gK = g.K, n_cols = g.n_cols;
    const int64_t gM = g.M;
    const float gscale = g.a_scale;
    // A staging: piece (row, kq) = 8 consecutive k of one row; BM * 4 pieces, one per thread for RB = 4
    constexpr int A_PER = (BM * 4 + kThreads - 1) / kThreads;
    const float* a_ptr[A_PER];
    bool a_ok[A_PER];
    int a_dst[A_PER], a_k0[A_PER];
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
        const int idx = tid + u * kThreads;
        const int row = idx >> 2, q = idx & 3;
        const int64_t r = m0 + row;
        a_ok[u] = idx < BM * 4 && r < gM;
        const int64_t rr = a_ok[u] ? r : 0;
        a_ptr[u] = g.A + (dcs_group_row(rr, g.a_gdiv, g.a_gmul, g.a_gdiv >= g.M)) * g.lda + q * 8;
        a_k0[u] = q * 8;
        a_dst[u] = row * kRowU4 + q;
    }
    const u32x4* Bq = reinterpret_cast<const u32x4*>(g.Bq) + ((int64_t)(n0 + wave * 16 + fi)) * 4 + kq;
    const int64_t b_plane = (int64_t)n_cols * 4, b_kt = 3 * b_plane;

    f32x4 ra[A_PER][2];
    u32x4 rb[3];
#define DCS_LOAD(kt_)                                                                                   \
    {                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < A_PER; ++u) {                                             \
            const int k = (kt_) * 32 + a_k0[u];                                                         \
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};                                                  \
            ra[u][0] = (a_ok[u] && k < gK) ? *reinterpret_cast<const f32x4*>(a_ptr[u] + (kt_) * 32) : z;      \
            ra[u][1] = (a_ok[u] && k + 4 < gK) ? *reinterpret_cast<const f32x4*>(a_ptr[u] + (kt_) * 32 + 4) : z; \
        }                                                                                               \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) rb[p] = Bq[(kt_) * b_kt + p * b_plane];           \
    }
#define DCS_STORE(buf_)                                                                                 \
    {                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < A_PER; ++u) {                                             \
            if (A_PER * kThreads == BM * 4 || tid + u * kThreads < BM * 4) {                            \
                u32x4 p0, p1, p2;                                                                       \
                split8(ra[u][0] * gscale, ra[u][1] * gscale, p0, p1, p2);                               \
                As[(buf_) * kBuf + a_dst[u]] = p0;                                                      \
                As[(buf_) * kBuf + kPlane + a_dst[u]] = p1;                                             \
                As[(buf_) * kBuf + 2 * kPlane + a_dst[u]] = p2;                                         \
            }                                                                                           \
        }                                                                                               \
    }

    f32x4 acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = (gK + 31) / 32;
    DCS_LOAD(0)
    for (int kt = 0; kt < nkt; ++kt) {
        // buffer kt & 1 was last read in iteration kt - 2; the barrier of iteration kt - 1 fences those reads
        DCS_STORE(kt & 1)
        u32x4 bcur[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bcur[p] = rb[p];
        __syncthreads();
        if (kt + 1 < nkt) DCS_LOAD(kt + 1)
        const u32x4* Ab = As + (kt & 1) * kBuf + fi * kRowU4 + kq;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const u32x4 a0 = Ab[r * 16 * kRowU4], a1 = Ab[kPlane + r * 16 * kRowU4], a2 = Ab[2 * kPlane + r * 16 * kRowU4];
            // smallest terms first
            acc[r] = mma(a2, bcur[0], acc[r]);
            acc[r] = mma(a0, bcur[2], acc[r]);
            acc[r] = mma(a1, bcur[1], acc[r]);
            acc[r] = mma(a1, bcur[0], acc[r]);
            acc[r] = mma(a0, bcur[1], acc[r]);
            acc[r] = mma(a0, bcur[0], acc[r]);
        }
    }
#undef DCS_LOAD
#undef DCS_STORE
    // epilogue: C/D layout of the 16x16 MFMA: column = lane & 15, row = (lane >> 4) * 4 + reg
    const int col = n0 + wave * 16 + fi;
    if (col < g.n_store) {
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

// ------------------------------------------------------------------------------------------------
// Few rows, very wide N (the per-source dense layers of the Bach10 / score-informed graphs: 167 x 256 x 666 600, a
// gigabyte of B planes): ALL rows in one workgroup, so that every byte of B is read exactly once and meets 11 row blocks.
// The product is taken transposed -- MFMA rows = 16 columns of C, MFMA columns = 16 rows of C -- and the four column
// blocks of a wave are interleaved (block cb, MFMA row i  <->  column 16 (i / 4) + 4 cb + i % 4), so that a lane ends up
// with 16 CONSECUTIVE columns of one row: four 16-byte stores, 256 contiguous bytes per row and wave (the 64 x 64 tiling
// above writes 64-byte pieces, which this chip's store path moves at a fraction of the rate, and reads B once per row tile).
// B fragments go global -> registers one k tile ahead; the A rows of a k tile are split once per workgroup into LDS planes
// (one buffer, two barriers per k tile: with two column blocks per wave two or three workgroups share a CU and fill each
// other's barrier, prologue and store-tail gaps; four blocks per wave = one wave per SIMD measured slower than the 64 x 64
// tiling: 0.56 vs 0.52 ms for the four Bach10 launches).
// ------------------------------------------------------------------------------------------------
// -DDCS_SKINNY_TRACE: wave 0 of workgroup (0, 0) adds up s_memtime intervals per phase (shader clocks): [0] split + LDS writes,
// [1] first barrier, [2] next tile's loads issued, [3] fragment reads + MFMAs, [4] second barrier, [5] epilogue, [6] whole kernel
#if defined(DCS_SKINNY_TRACE)
__device__ unsigned long long skinny_trace_buf[8];
#define SK_NOW(v_)                                      \
    {                                                   \
        __builtin_amdgcn_sched_barrier(0);              \
        v_ = __builtin_amdgcn_s_memtime();              \
        __builtin_amdgcn_sched_barrier(0);              \
    }
#define SK_ADD(i_, a_, b_) sk[i_] += (b_) - (a_);
#else
#define SK_NOW(v_)
#define SK_ADD(i_, a_, b_)
#endif
// AQ: the rows come pre-split (DcsGemm::Aq, round 5): the staging of a k tile is three 16-byte loads and three LDS writes per
// piece, no arithmetic -- with hundreds of workgroups each splitting all 128 .. 176 rows of every k tile the split was 22 % of
// the kernel (profiles/r04_o_skinny_timeline.txt).  Same planes, same products: bit-identical results.
// B32 (round 6): g.Bq holds the weights unsplit (gemm_pack_b32_kernel: 32-byte f32 pieces, 4 bytes per weight instead of 6); a wave
// splits its fragments in registers when it takes them over -- 112 vector instructions per k tile beside 132 MFMAs, for a third
// fewer bytes on a kernel whose time is the weight stream.
template <int RBT /* row blocks of 16 */, int CB /* 16-column blocks per wave */, bool AQ = false, bool B32 = false>
__global__ __launch_bounds__(kThreads, CB <= 2 ? 2 : 1) void gemm_bf16x3_skinny_kernel(const DcsGemm g0, const DcsGemmBranches br) {
    // blockIdx.y = branch: the same A against another (B planes, bias, C) triple -- one launch for all sources
    DcsGemm g = g0;
    if (br.n > 0) {
        g.Bq = br.Bq[blockIdx.y];
        g.bias = br.bias[blockIdx.y];
        g.C = br.C[blockIdx.y];
    }
    constexpr int ROWS = RBT * 16;
    constexpr int kPlane = ROWS * kRowU4;
    constexpr int CW = CB * 16;                            // columns per wave
    extern __shared__ u32x4 As[];                         // [3 planes][ROWS][kRowU4]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    const int n_cols = g.n_cols, gK = g.K;
    // K slices (g.partial: the long-K form, blockIdx.z = slice of g.kchunk columns of A): k tiles [kt_lo, kt_hi), raw sums out
    const int nkt_all = (gK + 31) / 32;
    const int kt_lo = g.partial ? (int)blockIdx.z * (g.kchunk / 32) : 0;
    const int nkt = (g.partial && kt_lo + g.kchunk / 32 < nkt_all) ? kt_lo + g.kchunk / 32 : nkt_all;
    const float gscale = g.a_scale;
    const int n0 = (blockIdx.x * 4 + wave) * CW;          // this wave's columns
    const bool live = n0 < n_cols;                        // n_cols is a multiple of 64
    // A staging: piece idx = (row, kg): 8 consecutive k of one row
    constexpr int A_PER = (ROWS * 4 + kThreads - 1) / kThreads;
    const float* a_ptr[A_PER];
    bool a_ok[A_PER];
    int a_dst[A_PER], a_k0[A_PER];
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
        const int idx = tid + u * kThreads;
        const int row = idx >> 2, q = idx & 3;
        a_ok[u] = idx < ROWS * 4 && row < g.M;
        const int64_t rr = a_ok[u] ? row : 0;
        a_ptr[u] = g.A + (dcs_group_row(rr, g.a_gdiv, g.a_gmul, g.a_gdiv >= g.M)) * g.lda + q * 8;
        a_k0[u] = q * 8;
        a_dst[u] = idx < ROWS * 4 ? row * kRowU4 + q : -1;
    }
    // B: piece (kt, plane, column, kg); MFMA row fi of block cb is column n0 + 4 CB (fi / 4) + 4 cb + fi % 4
    const int64_t b_plane = (int64_t)n_cols * 4, b_kt = 3 * b_plane;
    const u32x4* Bl = reinterpret_cast<const u32x4*>(g.Bq) + ((int64_t)((live ? n0 : 0) + (fi >> 2) * (4 * CB) + (fi & 3))) * 4 + kq;
    const f32x4* Bl32 = reinterpret_cast<const f32x4*>(g.Bq) + (((int64_t)((live ? n0 : 0) + (fi >> 2) * (4 * CB) + (fi & 3))) * 4 + kq) * 2;
    const int64_t b32_kt = (int64_t)n_cols * 8;          // f32x4 units per k tile
    f32x4 bn32[B32 ? CB : 1][2];
    f32x4 ra[AQ ? 1 : A_PER][2];
    u32x4 rq[AQ ? A_PER : 1][3];                           // AQ: the three planes of the piece in flight
    const u32x4* aq_ptr[A_PER];                            // AQ: plane 0 of this thread's piece of k tile 0
    const int64_t aq_plane = (int64_t)g.aq_rows * 4, aq_kt = 3 * aq_plane;
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
        const int idx = tid + u * kThreads;
        const int row = idx >> 2, q = idx & 3;
        aq_ptr[u] = reinterpret_cast<const u32x4*>(g.Aq) + (int64_t)(idx < ROWS * 4 ? row : 0) * 4 + q;
    }
    int ra_k[A_PER];                                       // first k of the piece in flight
    u32x4 bn[CB][3], bc[CB][3];
// The loads are UNCONDITIONAL (rows past M read row 0, k past K reads the last four of the row) and nothing touches the
// loaded registers before the next tile's split, where the out-of-range pieces are zeroed: with `ok ? load : 0` the compiler
// waited for each load to feed the select -- 1 640 cycles per k tile to "issue" twelve loads, 15 % of the kernel (in-kernel
// timeline, profiles/r04_o_skinny_timeline.txt).
#define DCS_LOAD_A(kt_)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < A_PER; ++u) {                                                 \
        if constexpr (AQ) {                                                                             \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) rq[u][p] = aq_ptr[u][(kt_) * aq_kt + p * aq_plane]; \
        } else {                                                                                        \
            const int k = (kt_) * 32 + a_k0[u];                                                         \
            ra[u][0] = *reinterpret_cast<const f32x4*>(a_ptr[u] + (k < gK ? k : gK - 4) - a_k0[u]);     \
            ra[u][1] = *reinterpret_cast<const f32x4*>(a_ptr[u] + (k + 4 < gK ? k + 4 : gK - 4) - a_k0[u]); \
            ra_k[u] = k;                                                                                \
        }                                                                                               \
    }
#define DCS_LOAD_B(kt_)                                                                                 \
    _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) {                                                 \
        if constexpr (B32) {                                                                            \
            bn32[cb][0] = Bl32[(kt_) * b32_kt + cb * 32];                                               \
            bn32[cb][1] = Bl32[(kt_) * b32_kt + cb * 32 + 1];                                           \
        } else {                                                                                        \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) bn[cb][p] = Bl[(kt_) * b_kt + p * b_plane + cb * 16]; \
        }                                                                                               \
    }
    f32x4 acc[RBT][CB];
#pragma unroll
    for (int r = 0; r < RBT; ++r)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[r][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(DCS_SKINNY_TRACE)
    unsigned long long sk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, sb = 0;
    SK_NOW(sb)
#endif
    DCS_LOAD_A(kt_lo)
    DCS_LOAD_B(kt_lo)
    for (int kt = kt_lo; kt < nkt; ++kt) {
        SK_NOW(s0)
#pragma unroll
        for (int u = 0; u < A_PER; ++u) {
            if (a_dst[u] >= 0) {
                u32x4* dst = As + a_dst[u];
                if constexpr (AQ) {                       // (rows past M and k past K are zero in the planes)
                    dst[0] = rq[u][0]; dst[kPlane] = rq[u][1]; dst[2 * kPlane] = rq[u][2];
                } else {
                    u32x4 p0, p1, p2;
                    const float s0 = (a_ok[u] && ra_k[u] < gK) ? gscale : 0.f, s1 = (a_ok[u] && ra_k[u] + 4 < gK) ? gscale : 0.f;
                    split8(ra[u][0] * s0, ra[u][1] * s1, p0, p1, p2);
                    dst[0] = p0; dst[kPlane] = p1; dst[2 * kPlane] = p2;
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if constexpr (B32) {
                split8(bn32[cb][0], bn32[cb][1], bc[cb][0], bc[cb][1], bc[cb][2]);
            } else {
#pragma unroll
                for (int p = 0; p < 3; ++p) bc[cb][p] = bn[cb][p];
            }
        }
        SK_NOW(s1)
        __syncthreads();
        SK_NOW(s2)
        const int ktn = kt + 1 < nkt ? kt + 1 : kt;       // last tile: a harmless re-read
        DCS_LOAD_A(ktn)
        DCS_LOAD_B(ktn)
        SK_NOW(s3)
        const u32x4* Ab = As + fi * kRowU4 + kq;
#pragma unroll
        for (int r = 0; r < RBT; ++r) {
            const u32x4 a0 = Ab[r * 16 * kRowU4], a1 = Ab[kPlane + r * 16 * kRowU4], a2 = Ab[2 * kPlane + r * 16 * kRowU4];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 c = acc[r][cb];
                c = mma(bc[cb][2], a0, c);                // smallest products first
                c = mma(bc[cb][0], a2, c);
                c = mma(bc[cb][1], a1, c);
                c = mma(bc[cb][1], a0, c);
                c = mma(bc[cb][0], a1, c);
                c = mma(bc[cb][0], a0, c);
                acc[r][cb] = c;
            }
        }
        SK_NOW(s4)
        __syncthreads();                                  // every wave has read this k tile's planes
        SK_NOW(s5)
        SK_ADD(0, s0, s1)
        SK_ADD(1, s1, s2)
        SK_ADD(2, s2, s3)
        SK_ADD(3, s3, s4)
        SK_ADD(4, s4, s5)
    }
#undef DCS_LOAD_A
#undef DCS_LOAD_B
    if (!live) return;
    // lane (row fi of block r, kq): columns n0 + 4 CB kq + 4 cb + e
    const int c0 = n0 + kq * (4 * CB);
    if (g.partial) {   // raw sums of this K slice, [slice][row][column]: the reduce pass adds the slices in order, bias, rectifier
#pragma unroll
        for (int r = 0; r < RBT; ++r) {
            const int64_t row = r * 16 + fi;
            if (row < g.M) {
                float* pp = g.partial + ((int64_t)blockIdx.z * g.M + row) * n_cols + c0;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<f32x4*>(pp + cb * 4) = acc[r][cb];
            }
        }
        return;
    }
    f32x4 bias[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[cb][e] = (g.bias && c0 + cb * 4 + e < g.n_store) ? g.bias[c0 + cb * 4 + e] : 0.f;
#pragma unroll
    for (int r = 0; r < RBT; ++r) {
        const int64_t row = r * 16 + fi;
        if (row < g.M) {
            float* cp = g.C + dcs_group_row(row, g.c_gdiv, g.c_gmul, g.c_gdiv >= g.M) * g.ldc + c0;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 v = acc[r][cb] + bias[cb];
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (c0 + cb * 4 + 4 <= g.n_store) {
                    *reinterpret_cast<f32x4*>(cp + cb * 4) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        if (c0 + cb * 4 + e < g.n_store) cp[cb * 4 + e] = v[e];
                }
            }
        }
    }
#if defined(DCS_SKINNY_TRACE)
    SK_NOW(s0)
    sk[5] = s0 - s5;
    sk[6] = s0 - sb;
    if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) skinny_trace_buf[i] = sk[i];
#endif
}

}  // namespace

#if defined(DCS_SKINNY_TRACE)
extern "C" __attribute__((visibility("default"))) int skinny_trace_dump(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(skinny_trace_buf), sizeof(unsigned long long) * 8);
}
#endif

namespace {

// Second pass of the long-K launch: 64 consecutive outputs per workgroup, the slices dealt to its four waves in four contiguous
// runs (each wave adds its run in slice order, sixteen loads in flight), the four run sums added in run order: deterministic,
// and a few hundred slices cost two or three memory round trips instead of one per eight slices.
__global__ __launch_bounds__(kThreads) void gemm_longk_reduce_kernel(const DcsGemm g, int ksplit) {
    __shared__ float part[4][64];
    const int tid = threadIdx.x, o = tid & 63, run = tid >> 6;
    const int64_t total = g.M * (int64_t)g.n_cols;
    const int64_t idx = (int64_t)blockIdx.x * 64 + o;
    const int per = (ksplit + 3) / 4;
    const int z0 = run * per, z1 = z0 + per < ksplit ? z0 + per : ksplit;
    float v = 0.f;
    if (idx < total) {
        const float* p = g.partial + idx;
        int z = z0;
        for (; z + 16 <= z1; z += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = p[(z + u) * total];
#pragma unroll
            for (int u = 0; u < 16; ++u) v += t[u];
        }
        for (; z < z1; ++z) v += p[z * total];
    }
    part[run][o] = v;
    __syncthreads();
    if (run == 0 && idx < total) {
        const int64_t row = idx / g.n_cols;
        const int col = (int)(idx - row * g.n_cols);
        if (col < g.n_store) {
            float s = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
            s += g.bias ? g.bias[col] : 0.f;
            if (g.relu) s = fmaxf(s, 0.f);
            g.C[dcs_group_row(row, g.c_gdiv, g.c_gmul, g.c_gdiv >= g.M) * g.ldc + col] = s;
        }
    }
}

}  // namespace

// Few rows, few columns, very long K (the bottleneck dense layer of the Bach10 / score-informed graphs: 128 .. 176 x 166 650 x
// 256) on the bf16 matrix pipe: the all-rows kernel above with K cut into slices (blockIdx.z; two workgroups per CU), raw
// sums per slice into ctx->gemm_ws, then gemm_longk_reduce_kernel.  false: not taken (shape, no planes, no scratch).
bool dcs_launch_gemm_bf16x3_longk(dcs_ctx* ctx, const DcsGemm& g) {
    static const bool on = !(getenv("DCS_GEMM_BF16") && atoi(getenv("DCS_GEMM_BF16")) == 0);   // (DCS_GEMM_KSPLIT=n forces the f32 K-split)
    if (!on || !g.Bq || !g.a_vec || g.partial || g.a_rowmap || (g.K & 3) || (g.lda & 3)) return false;
    if (g.M < 128 || g.M > 176 || (g.n_cols % 128) || g.n_cols > 1024 || g.K < 16384) return false;
    const int col_wgs = g.n_cols / 128;
    int ksplit = (int)((2 * (int64_t)ctx->n_cu + col_wgs - 1) / col_wgs);
    const int nkt = (g.K + 31) / 32;
    if (ksplit > nkt) ksplit = nkt;
    const int kts = (nkt + ksplit - 1) / ksplit;                   // k tiles per slice
    ksplit = (nkt + kts - 1) / kts;
    if (ksplit < 2) return false;
    if (ctx->gemm_ws.ensure((size_t)ksplit * g.M * g.n_cols * sizeof(float)) != DCS_OK) return false;
    DcsGemm q = g;
    q.partial = (float*)ctx->gemm_ws.ptr;
    q.kchunk = kts * 32;
    q.Aq = nullptr;
    const int rbt = g.M <= 128 ? 8 : 11;
    const size_t lds = (size_t)3 * rbt * 16 * kRowU4 * 16;
    auto kern = g.bq_f32 ? (rbt == 8 ? gemm_bf16x3_skinny_kernel<8, 2, false, true> : gemm_bf16x3_skinny_kernel<11, 2, false, true>)
                         : (rbt == 8 ? gemm_bf16x3_skinny_kernel<8, 2> : gemm_bf16x3_skinny_kernel<11, 2>);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    DcsGemmBranches none{};
    hipLaunchKernelGGL(kern, dim3((unsigned)col_wgs, 1, (unsigned)ksplit), dim3(kThreads), lds, ctx->stream, q, none);
    hipLaunchKernelGGL(gemm_longk_reduce_kernel, dim3((unsigned)dcs_cdiv(g.M * g.n_cols, 64)), dim3(kThreads), 0, ctx->stream, q, ksplit);
    return true;
}

// the second pass alone (the f16 long-K kernel of gemm_f16.hip leaves its slices in the same layout): q.partial, q.M, q.n_cols, q.n_store,
// q.bias, q.relu, q.C / ldc / c_gdiv / c_gmul
void dcs_launch_gemm_longk_reduce(dcs_ctx* ctx, const DcsGemm& q, int ksplit) {
    hipLaunchKernelGGL(gemm_longk_reduce_kernel, dim3((unsigned)dcs_cdiv(q.M * q.n_cols, 64)), dim3(kThreads), 0, ctx->stream, q, ksplit);
}

size_t dcs_gemm_b32_bytes(int K, int n_cols) { return (size_t)((K + 31) / 32) * (size_t)n_cols * 4 * 32; }

int dcs_gemm_pack_b32(dcs_ctx* ctx, const float* B_d, int K, int ldb, int n_cols, void* B32_d, int perm_c, int perm_p) {
    const int64_t n_pieces = (int64_t)((K + 31) / 32) * n_cols * 4;
    if (perm_c > 0 && (int64_t)perm_c * perm_p > n_cols) DCS_FAIL(DCS_EINVAL, "dcs_gemm_pack_b32: permutation wider than B");
    hipLaunchKernelGGL(gemm_pack_b32_kernel, dim3((unsigned)dcs_cdiv(n_pieces, kThreads)), dim3(kThreads), 0, ctx->stream, B_d, K,
                       ldb, n_cols, reinterpret_cast<f32x4*>(B32_d), n_pieces, perm_c, perm_p);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

size_t dcs_gemm_bq_bytes(int K, int n_cols) { return (size_t)((K + 31) / 32) * 3 * (size_t)n_cols * 4 * 16; }

int dcs_gemm_pack_bq(dcs_ctx* ctx, const float* B_d, int K, int ldb, int n_cols, void* Bq_d, int perm_c, int perm_p) {
    const int64_t n_pieces = (int64_t)((K + 31) / 32) * n_cols * 4;
    if (perm_c > 0 && (int64_t)perm_c * perm_p > n_cols) DCS_FAIL(DCS_EINVAL, "dcs_gemm_pack_bq: permutation wider than B");
    hipLaunchKernelGGL(gemm_pack_bq_kernel, dim3((unsigned)dcs_cdiv(n_pieces, kThreads)), dim3(kThreads), 0, ctx->stream, B_d, K,
                       ldb, n_cols, reinterpret_cast<u32x4*>(Bq_d), n_pieces, perm_c, perm_p);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

// true when the launch was taken (the caller's DcsTimer brackets it)
bool dcs_launch_gemm_bf16x3(dcs_ctx* ctx, const DcsGemm& g) {
    static const bool on = !(getenv("DCS_GEMM_BF16") && atoi(getenv("DCS_GEMM_BF16")) == 0);
    if (!on || !g.Bq || !g.a_vec || g.partial || (g.K & 3) || (g.lda & 3)) return false;
    if (g.bq_f32) return dcs_launch_gemm_bf16x3_skinny(ctx, g, nullptr);   // unsplit pieces: the all-rows kernel only
    const int64_t groups16 = (g.M + 15) / 16, col_groups = g.n_cols / 64;
    // launches that fill the chip only: with few workgroups a kernel's duration is one wave's dependent chain and the
    // f32 split-K kernels of gemm.hip cut that chain instead.  And wide B only (the per-source dense layers): the
    // A-streaming GEMMs of the encoder are bound by memory latency, not by the matrix pipe -- measured at 4096 tiles
    // conv1 0.041 -> 0.047 ms, fc 0.020 -> 0.021, while fc1x 0.042 -> 0.036; Bach10 fc1x (167 x 256 x 666 600) 0.72 -> 0.58
    if (groups16 * col_groups <= 4 * (int64_t)ctx->n_cu || g.n_cols < 1024 || g.M < 128) return false;
    if (dcs_launch_gemm_bf16x3_skinny(ctx, g, nullptr)) return true;
    constexpr int xcd_map = 1;
    if (groups16 * col_groups <= 16 * (int64_t)ctx->n_cu)
        hipLaunchKernelGGL((gemm_bf16x3_kernel<2>), dim3((unsigned)dcs_cdiv(g.M, 32), (unsigned)col_groups), dim3(kThreads), 0,
                           ctx->stream, g, xcd_map);
    else
        hipLaunchKernelGGL((gemm_bf16x3_kernel<4>), dim3((unsigned)dcs_cdiv(g.M, 64), (unsigned)col_groups), dim3(kThreads), 0,
                           ctx->stream, g, xcd_map);
    return true;
}

// All rows in one workgroup (gemm_bf16x3_skinny_kernel): 128 <= M <= 176 rows against very wide B planes.  `br` (optional):
// up to 4 (B planes, bias, C) triples that share g's A and shape -- one launch, blockIdx.y = branch.
bool dcs_launch_gemm_bf16x3_skinny(dcs_ctx* ctx, const DcsGemm& g, const DcsGemmBranches* br) {
    static const bool on = !(getenv("DCS_GEMM_BF16") && atoi(getenv("DCS_GEMM_BF16")) == 0);
    DcsGemmBranches b{};
    if (br) b = *br;
    const void* bq0 = b.n > 0 ? b.Bq[0] : g.Bq;
    if (!on || !bq0 || !g.a_vec || g.partial || (g.K & 3) || (g.lda & 3)) return false;
    if (g.M < 128 || g.M > 176 || g.n_cols < 8192 || (g.ldc & 3)) return false;
    // 16-byte stores need C rows and the column offset on 16-byte boundaries
    for (int i = 0; i < (b.n > 0 ? b.n : 1); ++i)
        if (((uintptr_t)(b.n > 0 ? b.C[i] : g.C) & 15) || (b.n > 0 && !b.Bq[i])) return false;
    const int rbt = g.M <= 128 ? 8 : 11;
    constexpr int cb_env = 2;   // four column blocks per wave measured 0.56 vs 0.52 ms (round 2)
    const int cb = cb_env == 4 ? 4 : 2;
    const size_t lds = (size_t)3 * rbt * 16 * kRowU4 * 16;
    // pre-split rows (dcs_gemm_split_a): only as made for this launch's row tiling, and only with a_scale == 1 (the planes hold A as it is)
#if defined(DCS_NO_AQ)   // experiment build: every workgroup splits the rows itself
    const bool aq = false;
#else
    const bool aq = g.Aq && g.aq_rows == rbt * 16 && g.a_scale == 1.f && g.a_gdiv >= g.M;
#endif
    auto kern = rbt == 8 ? (cb == 4 ? gemm_bf16x3_skinny_kernel<8, 4> : (aq ? gemm_bf16x3_skinny_kernel<8, 2, true> : gemm_bf16x3_skinny_kernel<8, 2>))
                         : (cb == 4 ? gemm_bf16x3_skinny_kernel<11, 4> : (aq ? gemm_bf16x3_skinny_kernel<11, 2, true> : gemm_bf16x3_skinny_kernel<11, 2>));
    if (g.bq_f32)
        kern = rbt == 8 ? (aq ? gemm_bf16x3_skinny_kernel<8, 2, true, true> : gemm_bf16x3_skinny_kernel<8, 2, false, true>)
                        : (aq ? gemm_bf16x3_skinny_kernel<11, 2, true, true> : gemm_bf16x3_skinny_kernel<11, 2, false, true>);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(kern, dim3((unsigned)dcs_cdiv(g.n_cols, 64 * cb), (unsigned)(b.n > 0 ? b.n : 1)), dim3(kThreads), lds,
                       ctx->stream, g, b);
    return true;
}

namespace {

// piece (kt, row, q): the 8 consecutive k of one row, split exactly into three bf16 terms
__global__ __launch_bounds__(kThreads) void gemm_split_a_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K, int rows_pad,
                                                                u32x4* __restrict__ Aq, int64_t n_pieces) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;   // q fastest, then row, then k tile
    if (idx >= n_pieces) return;
    const int q = (int)(idx & 3);
    const int64_t t = idx >> 2;
    const int row = (int)(t % rows_pad);
    const int64_t kt = t / rows_pad;
    f32x4 x0 = f32x4{0.f, 0.f, 0.f, 0.f}, x1 = x0;
    if (row < M) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t k = kt * 32 + q * 8 + j;
            const float v = k < K ? A[row * lda + k] : 0.f;
            if (j < 4) x0[j] = v; else x1[j - 4] = v;
        }
    }
    u32x4 p0, p1, p2;
    split8(x0, x1, p0, p1, p2);
    const int64_t base = (kt * 3 * rows_pad + row) * 4 + q;
    Aq[base] = p0;
    Aq[base + (int64_t)rows_pad * 4] = p1;
    Aq[base + (int64_t)rows_pad * 8] = p2;
}

}  // namespace

size_t dcs_gemm_aq_bytes(int K, int rows_pad) { return (size_t)((K + 31) / 32) * 3 * (size_t)rows_pad * 4 * 16; }

int dcs_gemm_split_a(dcs_ctx* ctx, const float* A_d, int64_t lda, int64_t M, int K, int rows_pad, void* Aq_d) {
    const int64_t n_pieces = (int64_t)((K + 31) / 32) * rows_pad * 4;
    hipLaunchKernelGGL(gemm_split_a_kernel, dim3((unsigned)dcs_cdiv(n_pieces, kThreads)), dim3(kThreads), 0, ctx->stream, A_d, lda, M, K,
                       rows_pad, reinterpret_cast<u32x4*>(Aq_d), n_pieces);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}
