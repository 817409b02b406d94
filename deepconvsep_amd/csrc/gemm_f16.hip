// The per-source dense layers of the Bach10 graph under the f16 switch (BASELINE configs[3], "fp16 MFMA conv path";
// separate_bach10.py:203-223: four DenseLayers 256 -> 30 x 11 x 505 = 166 650, rectified) with f16 WEIGHTS and an f16,
// channels-last OUTPUT -- what the fused f16 decoder behind them (colconv_deconv1_fused_kernel) multiplies anyway.
//
// Why (round 6): with the switch on, these layers were the largest kernel of the leg (0.37 of 1.15 ms) and ran the f32-class
// kernel of gemm_bf16x3.hip: 6 bytes of bf16 planes per weight (1.02 GB for the four branches) and a 445 MB f32 output that the
// decoder rounds to f16 on arrival -- 1.43 GB per launch at 3.7 TB/s.  SURVEY 8d prices this config as HBM-on-weights at 2 bytes
// per weight.  Here: one f16 plane (2 bytes per weight), the rows of Z split exactly-to-22-bits into two f16 terms (z = hi + lo;
// two MFMAs per block, so the only new rounding is the weights'), f32 accumulation, and D written ONCE as f16 in the layout the
// decoder reads: D16[tile][branch][position = y * W + x][32 channels] (30 real, 2 zero: a position is 64 bytes, a lane's eight
// channels one 16-byte load).  0.61 GB per launch.
//
// Kernel shape = gemm_bf16x3_skinny_kernel (all 128 .. 176 rows in one workgroup so that every byte of the weights is read exactly
// once; the product taken transposed so that a lane ends up with 8 consecutive columns = one 16-byte store).
#include "dcs_internal.h"
#include "generic.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
constexpr int kRowU4 = 5;   // LDS row stride in 16-byte pieces: 4 pieces + 1 (the 16 rows of a fragment read start in 16 different bank quads)

__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// B f32 [K rows][ldb], columns in the reference's flattened order (channel * npos + position)  ->  one f16 plane
// Bh[k tile][column][4 pieces of 8 k], output column j = position * chpad + channel (channels >= nch and columns past
// npos * chpad: zero weights)
__global__ __launch_bounds__(kThreads) void gemm_pack_bh_kernel(const float* __restrict__ B, int K, int ldb, int n_out, int nch, int npos,
                                                                int chpad, u32x4* __restrict__ Bh, int64_t n_pieces) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;   // (kt, column, kq), kq fastest
    if (idx >= n_pieces) return;
    const int kq = (int)(idx & 3);
    const int64_t t = idx >> 2;
    const int n = (int)(t % n_out);
    const int64_t kt = t / n_out;
    const int pos = n / chpad, ch = n - pos * chpad;
    const bool real = pos < npos && ch < nch;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t k = kt * 32 + kq * 8 + j;
        v[j] = (_Float16)((real && k < K) ? B[k * ldb + (int64_t)ch * npos + pos] : 0.f);
    }
    Bh[idx] = __builtin_bit_cast(u32x4, v);
}

// bias in the same column order, zero in the pad columns
__global__ __launch_bounds__(kThreads) void gemm_pack_bias_cl_kernel(const float* __restrict__ bias, int n_out, int nch, int npos, int chpad,
                                                                     float* __restrict__ out) {
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= n_out) return;
    const int pos = n / chpad, ch = n - pos * chpad;
    out[n] = (pos < npos && ch < nch) ? bias[(int64_t)ch * npos + pos] : 0.f;
}

// A f32 [M][K] -> two f16 planes, a = hi + lo to 22 bits: Ah[k tile][plane][rows_pad][4 pieces]; rows >= M and k >= K zero
__global__ __launch_bounds__(kThreads) void gemm_split_a_f16_kernel(const float* __restrict__ A, int64_t lda, int64_t M, int K, int rows_pad,
                                                                    u32x4* __restrict__ Ah, int64_t n_pieces) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;   // q fastest, then row, then k tile
    if (idx >= n_pieces) return;
    const int q = (int)(idx & 3);
    const int64_t t = idx >> 2;
    const int row = (int)(t % rows_pad);
    const int64_t kt = t / rows_pad;
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t k = kt * 32 + q * 8 + j;
        const float x = (row < M && k < K) ? A[row * lda + k] : 0.f;
        hi[j] = (_Float16)x;
        lo[j] = (_Float16)(x - (float)hi[j]);
    }
    const int64_t base = (kt * 2 * rows_pad + row) * 4 + q;
    Ah[base] = __builtin_bit_cast(u32x4, hi);
    Ah[base + (int64_t)rows_pad * 4] = __builtin_bit_cast(u32x4, lo);
}

struct DcsGemmF16 {
    const u32x4* Ah; int rows_pad;       // gemm_split_a_f16_kernel's planes of the rows (all of them: M <= rows_pad)
    int M, K, n_cols;                    // n_cols: a multiple of 128
    int n_br;                            // branches = blockIdx.y: (weights, bias, output) triples that share the rows
    const u32x4* Bh[4];
    const float* bias[4];
    _Float16* C[4];
    int64_t ldc;                         // halves between output rows
};

template <int RBT /* row blocks of 16 */>
__global__ __launch_bounds__(kThreads, 2) void gemm_f16_skinny_kernel(const DcsGemmF16 g) {
    constexpr int CB = 2, ROWS = RBT * 16, CW = CB * 16;
    constexpr int kPlane = ROWS * kRowU4;
    __shared__ u32x4 As[2 * kPlane];                      // [2 planes][ROWS][kRowU4]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    const int br = blockIdx.y;
    const int n_cols = g.n_cols;
    const int nkt = (g.K + 31) / 32;
    const int n0 = (blockIdx.x * 4 + wave) * CW;          // this wave's 32 columns (n_cols is a multiple of 128: always inside)
    // staging: piece (row, q) of both planes; rows past the planes' rows_pad do not exist (ROWS <= rows_pad is checked by the launcher)
    constexpr int A_PER = (ROWS * 4 + kThreads - 1) / kThreads;
    const int64_t a_plane = (int64_t)g.rows_pad * 4, a_kt = 2 * a_plane;
    const u32x4* a_src[A_PER];
    int a_dst[A_PER];
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
        const int idx = tid + u * kThreads;
        const int row = idx >> 2, q = idx & 3;
        const bool in = idx < ROWS * 4;
        a_src[u] = g.Ah + (int64_t)(in ? row : 0) * 4 + q;
        a_dst[u] = in ? row * kRowU4 + q : -1;
    }
    // B: MFMA row fi of block cb is column n0 + 8 (fi / 4) + 4 cb + fi % 4 (the four blocks... two blocks of a wave interleaved)
    const u32x4* Bl = g.Bh[br] + ((int64_t)(n0 + (fi >> 2) * (4 * CB) + (fi & 3))) * 4 + kq;
    const int64_t b_kt = (int64_t)n_cols * 4;
    u32x4 ra[A_PER][2], bn[CB], bc[CB];
    // unconditional loads; nothing touches the registers before the next tile's LDS write
#define DCS_LOAD(kt_)                                                                                   \
    {                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < A_PER; ++u) {                                             \
            ra[u][0] = a_src[u][(kt_) * a_kt];                                                          \
            ra[u][1] = a_src[u][(kt_) * a_kt + a_plane];                                                \
        }                                                                                               \
        _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) bn[cb] = Bl[(kt_) * b_kt + cb * 16];          \
    }
    f32x4 acc[RBT][CB];
#pragma unroll
    for (int r = 0; r < RBT; ++r)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[r][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    DCS_LOAD(0)
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int u = 0; u < A_PER; ++u)
            if (a_dst[u] >= 0) {
                As[a_dst[u]] = ra[u][0];
                As[kPlane + a_dst[u]] = ra[u][1];
            }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) bc[cb] = bn[cb];
        __syncthreads();
        {
            const int ktn = kt + 1 < nkt ? kt + 1 : kt;   // last tile: a harmless re-read
            DCS_LOAD(ktn)
        }
        const u32x4* Ab = As + fi * kRowU4 + kq;
#pragma unroll
        for (int r = 0; r < RBT; ++r) {
            const u32x4 a_hi = Ab[r * 16 * kRowU4], a_lo = Ab[kPlane + r * 16 * kRowU4];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 c = acc[r][cb];
                c = mma(bc[cb], a_lo, c);                 // the small term first
                c = mma(bc[cb], a_hi, c);
                acc[r][cb] = c;
            }
        }
        __syncthreads();                                  // every wave has read this k tile's planes
    }
#undef DCS_LOAD
    // lane (row fi of block r, kq): columns c0 .. c0 + 7, c0 = n0 + 8 kq -- bias, rectifier, eight halves, one 16-byte store
    const int c0 = n0 + kq * (4 * CB);
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = g.bias[br][c0 + e];
    _Float16* Cb = g.C[br];
#pragma unroll
    for (int r = 0; r < RBT; ++r) {
        const int row = r * 16 + fi;
        if (row < g.M) {
            f16x8 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (_Float16)fmaxf(acc[r][0][e] + bias[e], 0.f);
                h[4 + e] = (_Float16)fmaxf(acc[r][1][e] + bias[4 + e], 0.f);
            }
            *reinterpret_cast<u32x4*>(Cb + (int64_t)row * g.ldc + c0) = __builtin_bit_cast(u32x4, h);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The bottleneck layer under the f16 switch (128 .. 176 rows x K = 166 650 x 256): f16 weights (one plane, 85 MB instead of the
// 256 MB of bf16 planes), the f32 rows of A (conv2's output) split into two f16 terms ON THEIR WAY into LDS, K cut into slices
// (blockIdx.z; raw f32 sums per slice, then gemm_longk_reduce of gemm_bf16x3.hip adds the slices in order).  Same shape as the
// all-rows kernel above with four column blocks' worth of rows per workgroup.
// ------------------------------------------------------------------------------------------------
struct DcsGemmF16LongK {
    const float* A; int64_t lda; int M, K;        // K % 4 == 0, rows 16-byte aligned.  A16: A holds f16 rows of lda halves, K == lda % 32 == 0
    const u32x4* Bh; int n_cols;                  // f16 plane [k tile][n_cols][4 pieces]; n_cols a multiple of 128
    float* partial;                               // [slices][M][n_cols]
    int kts;                                      // k tiles per slice
};

// A16 (round 6): the rows arrive as f16 (the f16 conv2 writes them so, colconv_wreg_scatter_kernel<..., OUT16>): a piece is one
// 16-byte load, one plane, ONE MFMA per block.
template <int RBT, bool A16>
__global__ __launch_bounds__(kThreads, 2) void gemm_f16_longk_kernel(const DcsGemmF16LongK g) {
    constexpr int CB = 2, ROWS = RBT * 16, CW = CB * 16;
    constexpr int kPlane = ROWS * kRowU4;
    __shared__ u32x4 As[(A16 ? 1 : 2) * kPlane];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    const int n_cols = g.n_cols, gK = g.K;
    const int nkt_all = (gK + 31) / 32;
    const int kt_lo = (int)blockIdx.z * g.kts;
    const int nkt = kt_lo + g.kts < nkt_all ? kt_lo + g.kts : nkt_all;
    const int n0 = (blockIdx.x * 4 + wave) * CW;
    constexpr int A_PER = (ROWS * 4 + kThreads - 1) / kThreads;
    const float* a_ptr[A_PER];
    bool a_ok[A_PER];
    int a_dst[A_PER], a_k0[A_PER];
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
        const int idx = tid + u * kThreads;
        const int row = idx >> 2, q = idx & 3;
        a_ok[u] = idx < ROWS * 4 && row < g.M;
        a_ptr[u] = g.A + (int64_t)(a_ok[u] ? row : 0) * g.lda;
        a_k0[u] = q * 8;
        a_dst[u] = idx < ROWS * 4 ? row * kRowU4 + q : -1;
    }
    const u32x4* Bl = g.Bh + ((int64_t)(n0 + (fi >> 2) * (4 * CB) + (fi & 3))) * 4 + kq;
    const int64_t b_kt = (int64_t)n_cols * 4;
    f32x4 ra[A_PER][2];
    u32x4 rh[A_PER];
    int ra_k[A_PER];
    u32x4 bn[CB], bc[CB];
    const _Float16* ah_ptr[A_PER];
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
        const int idx = tid + u * kThreads;
        ah_ptr[u] = reinterpret_cast<const _Float16*>(g.A) + (int64_t)(a_ok[u] ? (idx >> 2) : 0) * g.lda + a_k0[u];
    }
    // unconditional loads (rows past M read row 0, k past K reads the last four of the row), masked at the split
#define DCS_LOAD(kt_)                                                                                   \
    {                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < A_PER; ++u) {                                             \
            const int k = (kt_) * 32 + a_k0[u];                                                         \
            if constexpr (A16) {                                                                        \
                rh[u] = *reinterpret_cast<const u32x4*>(ah_ptr[u] + (kt_) * 32);    /* K == lda: inside the row */ \
            } else {                                                                                    \
                ra[u][0] = *reinterpret_cast<const f32x4*>(a_ptr[u] + (k < gK ? k : gK - 4));           \
                ra[u][1] = *reinterpret_cast<const f32x4*>(a_ptr[u] + (k + 4 < gK ? k + 4 : gK - 4));   \
            }                                                                                           \
            ra_k[u] = k;                                                                                \
        }                                                                                               \
        _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) bn[cb] = Bl[(kt_) * b_kt + cb * 16];          \
    }
    f32x4 acc[RBT][CB];
#pragma unroll
    for (int r = 0; r < RBT; ++r)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[r][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    DCS_LOAD(kt_lo)
    for (int kt = kt_lo; kt < nkt; ++kt) {
#pragma unroll
        for (int u = 0; u < A_PER; ++u)
            if (a_dst[u] >= 0) {
                if constexpr (A16) {
                    As[a_dst[u]] = a_ok[u] ? rh[u] : u32x4{0u, 0u, 0u, 0u};
                    continue;
                }
                const float s0 = (a_ok[u] && ra_k[u] < gK) ? 1.f : 0.f, s1 = (a_ok[u] && ra_k[u] + 4 < gK) ? 1.f : 0.f;
                f16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = j < 4 ? ra[u][0][j] * s0 : ra[u][1][j - 4] * s1;
                    hi[j] = (_Float16)x;
                    lo[j] = (_Float16)(x - (float)hi[j]);
                }
                As[a_dst[u]] = __builtin_bit_cast(u32x4, hi);
                As[kPlane + a_dst[u]] = __builtin_bit_cast(u32x4, lo);
            }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) bc[cb] = bn[cb];
        __syncthreads();
        {
            const int ktn = kt + 1 < nkt ? kt + 1 : kt;
            DCS_LOAD(ktn)
        }
        const u32x4* Ab = As + fi * kRowU4 + kq;
#pragma unroll
        for (int r = 0; r < RBT; ++r) {
            const u32x4 a_hi = Ab[r * 16 * kRowU4];
            u32x4 a_lo = a_hi;
            if constexpr (!A16) a_lo = Ab[kPlane + r * 16 * kRowU4];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 c = acc[r][cb];
                if constexpr (!A16) c = mma(bc[cb], a_lo, c);
                c = mma(bc[cb], a_hi, c);
                acc[r][cb] = c;
            }
        }
        __syncthreads();
    }
#undef DCS_LOAD
    const int c0 = n0 + kq * (4 * CB);
#pragma unroll
    for (int r = 0; r < RBT; ++r) {
        const int row = r * 16 + fi;
        if (row < g.M) {
            float* pp = g.partial + ((int64_t)blockIdx.z * g.M + row) * n_cols + c0;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<f32x4*>(pp + cb * 4) = acc[r][cb];
        }
    }
}

}  // namespace

size_t dcs_gemm_bh_bytes(int K, int n_out) { return (size_t)((K + 31) / 32) * (size_t)n_out * 4 * 16; }
size_t dcs_gemm_ah_bytes(int K, int rows_pad) { return (size_t)((K + 31) / 32) * 2 * (size_t)rows_pad * 4 * 16; }

int dcs_gemm_pack_bh(dcs_ctx* ctx, const float* B_d, int K, int ldb, int n_out, int nch, int npos, int chpad, void* Bh_d) {
    const int64_t n_pieces = (int64_t)((K + 31) / 32) * n_out * 4;
    hipLaunchKernelGGL(gemm_pack_bh_kernel, dim3((unsigned)dcs_cdiv(n_pieces, kThreads)), dim3(kThreads), 0, ctx->stream, B_d, K, ldb, n_out,
                       nch, npos, chpad, reinterpret_cast<u32x4*>(Bh_d), n_pieces);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

int dcs_gemm_pack_bias_cl(dcs_ctx* ctx, const float* bias_d, int n_out, int nch, int npos, int chpad, float* out_d) {
    hipLaunchKernelGGL(gemm_pack_bias_cl_kernel, dim3((unsigned)dcs_cdiv(n_out, kThreads)), dim3(kThreads), 0, ctx->stream, bias_d, n_out, nch,
                       npos, chpad, out_d);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

// Z [M][K] f32 -> planes, then every branch in one launch: C[b][row][column] f16 (ldc halves apart), relu(Z . B_b + bias_b).
// false: the shape is not this kernel's (the caller keeps the f32-class path).
bool dcs_launch_gemm_f16_skinny(dcs_ctx* ctx, const float* Z, int64_t ldz, int M, int K, int n_cols, int n_br, const void* const* Bh,
                                const float* const* bias, void* const* C, int64_t ldc, void* Ah_scratch) {
    if (M < 128 || M > 176 || (n_cols % 128) || K < 32 || (K & 31) || n_br < 1 || n_br > 4 || (ldc & 7) || !Ah_scratch) return false;
    for (int b = 0; b < n_br; ++b)
        if (!Bh[b] || !bias[b] || !C[b] || (reinterpret_cast<uintptr_t>(C[b]) & 15)) return false;
    const int rows_pad = M <= 128 ? 128 : 176;
    const int64_t n_pieces = (int64_t)(K / 32) * rows_pad * 4;
    hipLaunchKernelGGL(gemm_split_a_f16_kernel, dim3((unsigned)dcs_cdiv(n_pieces, kThreads)), dim3(kThreads), 0, ctx->stream, Z, ldz,
                       (int64_t)M, K, rows_pad, reinterpret_cast<u32x4*>(Ah_scratch), n_pieces);
    DcsGemmF16 g{};
    g.Ah = reinterpret_cast<const u32x4*>(Ah_scratch); g.rows_pad = rows_pad;
    g.M = M; g.K = K; g.n_cols = n_cols; g.n_br = n_br; g.ldc = ldc;
    for (int b = 0; b < n_br; ++b) {
        g.Bh[b] = reinterpret_cast<const u32x4*>(Bh[b]);
        g.bias[b] = bias[b];
        g.C[b] = reinterpret_cast<_Float16*>(C[b]);
    }
    const dim3 grid((unsigned)(n_cols / 128), (unsigned)n_br);
    if (rows_pad == 128) hipLaunchKernelGGL((gemm_f16_skinny_kernel<8>), grid, dim3(kThreads), 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_f16_skinny_kernel<11>), grid, dim3(kThreads), 0, ctx->stream, g);
    return true;
}

// B f32 [K][ldb] in its own column order -> one f16 plane [k tile][n_cols][4 pieces] (gemm_pack_bh_kernel with one "channel")
int dcs_gemm_pack_bh_plain(dcs_ctx* ctx, const float* B_d, int K, int ldb, int n_cols, void* Bh_d) {
    return dcs_gemm_pack_bh(ctx, B_d, K, ldb, n_cols, 1, n_cols, 1, Bh_d);
}

// raw sums of the K slices into `partial` ([slices][M][n_cols] floats; the caller sized it with dcs_gemm_f16_longk_slices and runs
// the reduce pass).  Returns the slice count, 0: not taken.
int dcs_gemm_f16_longk_slices(const dcs_ctx* ctx, int M, int K, int n_cols) {
    if (M < 128 || M > 176 || (n_cols % 128) || n_cols > 1024 || K < 16384 || (K & 3)) return 0;
    const int col_wgs = n_cols / 128, nkt = (K + 31) / 32;
    int ksplit = (int)((2 * (int64_t)ctx->n_cu + col_wgs - 1) / col_wgs);
    if (ksplit > nkt) ksplit = nkt;
    const int kts = (nkt + ksplit - 1) / ksplit;
    ksplit = (nkt + kts - 1) / kts;
    return ksplit < 2 ? 0 : ksplit;
}

bool dcs_launch_gemm_f16_longk(dcs_ctx* ctx, const void* A, int64_t lda, int M, int K, int n_cols, const void* Bh, float* partial,
                               bool a_f16) {
    const int ksplit = dcs_gemm_f16_longk_slices(ctx, M, K, n_cols);
    if (ksplit < 2 || !Bh || !partial || (reinterpret_cast<uintptr_t>(A) & 15)) return false;
    if (a_f16 ? ((lda & 31) || K != lda) : ((lda & 3) != 0)) return false;
    const int nkt = (K + 31) / 32;
    DcsGemmF16LongK g{};
    g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.M = M; g.K = K; g.Bh = reinterpret_cast<const u32x4*>(Bh); g.n_cols = n_cols;
    g.partial = partial;
    g.kts = (nkt + ksplit - 1) / ksplit;
    const dim3 grid((unsigned)(n_cols / 128), 1, (unsigned)ksplit);
    if (a_f16) {
        if (M <= 128) hipLaunchKernelGGL((gemm_f16_longk_kernel<8, true>), grid, dim3(kThreads), 0, ctx->stream, g);
        else hipLaunchKernelGGL((gemm_f16_longk_kernel<11, true>), grid, dim3(kThreads), 0, ctx->stream, g);
    } else {
        if (M <= 128) hipLaunchKernelGGL((gemm_f16_longk_kernel<8, false>), grid, dim3(kThreads), 0, ctx->stream, g);
        else hipLaunchKernelGGL((gemm_f16_longk_kernel<11, false>), grid, dim3(kThreads), 0, ctx->stream, g);
    }
    return true;
}
