// Decoder kernels of the DSD100 / hiphop graph (examples/dsd100/separate_dsd.py:208-234) and
// its soft mask (:258-271), fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
//   deconv2_kernel : InverseLayer(l_reshape_i, l_conv2)  -- VJP of the (tc/2 x 1) convolution
//   final_kernel   : InverseLayer(., l_conv1) -> ConcatLayer -> BiasLayer -> rectify -> soft mask
//                    -> x mixture [-> cross-fade overlap-add of util.py:297-327 when FOLD]
#include "dcs_internal.h"
#include "dsd.h"

#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------
// Transposed conv2 as "GEMM + col2im".  For one (tile, branch) and one group of GS input channels:
//   P[t', c*kh + dt] = sum_co D[t', co] * W2c[co, ci0 + c, dt]     (MFMA, M = H2, K = CP, N = GS*kh)
//   G[t, ci0 + c]    = sum_dt P[t - dt, c*kh + dt],  0 <= t-dt < H2  (LDS reduction)
// which spends kh*CI*CP*H2 MACs -- the exact count of the transposed convolution -- instead of the
// (tc x kh*CP) x CI padded-GEMM form that multiplies mostly zeros.  Splitting the channels into NG
// groups gives NG x more workgroups (the 32-tile batch has only 96 (tile, branch) pairs).
// The A fragments (one row block of D) stay in registers; the NQ B values of a column block are
// all requested before the first MFMA that needs them, and the next block's are in flight meanwhile.
// ------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(kThreads) void deconv2_kernel(const float* __restrict__ D, const float* __restrict__ Bw,
                                                           float* __restrict__ G, int H2, int CP, int CI, int kh,
                                                           int tc, int GS, int gcols /* padded GS*kh, multiple of 16 */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nrb = (H2 + 15) >> 4;  // 1 or 2 row blocks
    const int as = CP + 2;
    float* As = smem;                 // [16*nrb][CP+2]
    float* P = smem + 16 * nrb * as;  // [16*nrb][gcols]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    const int64_t ks = blockIdx.x;
    const int grp = blockIdx.y;
    const float* Dp = D + ks * (int64_t)H2 * CP;

    for (int idx = tid; idx < 16 * nrb * CP; idx += kThreads) {
        const int row = idx / CP, c = idx - row * CP;
        As[row * as + c] = (row < H2) ? Dp[row * CP + c] : 0.f;
    }
    __syncthreads();

    float a0[NQ], a1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        a0[q] = As[fi * as + 4 * q + kq];
        a1[q] = (nrb > 1) ? As[(16 + fi) * as + 4 * q + kq] : 0.f;
    }

    const int ncb = gcols >> 4;
    const int ldb = gcols * (int)gridDim.y;
    const float* bbase = Bw + (int64_t)kq * ldb + grp * gcols + fi;
    float b[NQ], bn[NQ];
    int cb = wave;
    if (cb < ncb) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[q] = bbase[(int64_t)(4 * q) * ldb + cb * 16];
    }
    for (; cb < ncb; cb += 4) {
        const int nxt = cb + 4;
        if (nxt < ncb) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) bn[q] = bbase[(int64_t)(4 * q) * ldb + nxt * 16];
        }
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b[q], acc0, 0, 0, 0);
            if (nrb > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b[q], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            P[(kq * 4 + e) * gcols + cb * 16 + fi] = acc0[e];
            if (nrb > 1) P[(16 + kq * 4 + e) * gcols + cb * 16 + fi] = acc1[e];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[q] = bn[q];
    }
    __syncthreads();

    const int ci0 = grp * GS;
    const int ngg = (CI + kDsdGch - 1) / kDsdGch;
    float* Gp = G + ks * (int64_t)ngg * tc * kDsdGch;  // [channel group][t][8]
    for (int o = tid; o < tc * GS; o += kThreads) {
        const int t = o / GS, c = o - t * GS;
        const int ci = ci0 + c;
        if (ci >= CI) continue;
        int lo = t - (H2 - 1);
        if (lo < 0) lo = 0;
        const int hi = t < kh - 1 ? t : kh - 1;
        float sum = 0.f;
        for (int dt = lo; dt <= hi; ++dt) sum += P[(t - dt) * gcols + c * kh + dt];
        Gp[((ci / kDsdGch) * tc + t) * kDsdGch + (ci % kDsdGch)] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// Many-tiles variant: independent waves streaming over (tile, branch) items, no workgroup barrier after the
// weight slice is in LDS.  A workgroup owns one group of 8 input channels; its weight slice
// Bs[channel c][tap dt][52 output filters] (26.6 KB) stays in LDS.  (Keeping the B fragments in 104 registers
// instead was not faster: 0.194 vs 0.181 ms at 4096 tiles.)  For an item, a wave
//   * reads its A fragments straight from D (the K order final_chan() makes a lane's 13 filters of row t'
//     three float4 and one float), prefetching the next item's while it works;
//   * per pair of channels: 2 x 13 MFMAs give P[t'][dt] (one 16 x 16 block per channel, column = tap), which
//     it writes SKEWED, Ps[c][t' + dt][dt], into a private LDS buffer, so that the col2im sum
//     G[t, c] = sum_dt P[t - dt][dt] is a row sum: lane (t, c) reads 16 contiguous floats and adds them in
//     tap order -- the order of deconv2_kernel.  Entries no (t', dt) maps to are zeroed once and stay zero;
//   * collects the 30 x 8 outputs of the item in LDS and stores them as one contiguous 960-byte block of
//     G[item][channel group][t][8].
// fp32 MFMA and VALU instructions do not overlap on a SIMD (scripts/ubench/mfma_valu.hip), so what counts is
// the non-MFMA instruction count per MFMA: here ~45 VALU/LDS instructions per 26 MFMAs, against the one-shot
// kernel's staged A tile, P round trip with divisions in the reduction, and 3 barriers per item.
// Groups of one item run on the same XCD at about the same time (block -> (group, column) mapping), so the item's
// D rows are fetched from HBM once and served from that XCD's L2 to the other groups.
// ------------------------------------------------------------------------------------------------
constexpr int kD2PsStride = 20;  // floats per skewed row: 16 taps + 4 (16-byte aligned, spreads the banks)
// the second channel of a pair starts 32 banks further: the col2im reader lanes (t, 0) and (t, 1) of one 16-lane pass
// would otherwise hit the same banks (32 rows x 20 floats = 0 mod 64)
constexpr int kD2PsChan = 32 * kD2PsStride + 32;
constexpr int kD2PsSize = 2 * kD2PsChan;

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// x = hi + mid + lo exactly, each term a bf16 (truncation split: 8 + 8 + 8 significand bits); returned as the three
// 32-bit patterns whose upper halves are the bf16 values
__device__ __forceinline__ void bf16_split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);     // exact
    m = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);    // exact, at most 8 significant bits left
    l = __float_as_uint(r2) & 0xffff0000u;
}

// SPLIT: the outputs leave as three bf16 planes, Gs[item][channel group][t][plane][8 channels] (16 bytes per piece, the
// operand layout of final_bf16x3_kernel), instead of f32 G[item][channel group][t][8].
template <bool SPLIT>
__global__ __launch_bounds__(kThreads) void deconv2_stream_kernel(const float* __restrict__ D,
                                                                  const float* __restrict__ Bws,
                                                                  float* __restrict__ G, int64_t n_ks, int H2,
                                                                  int kh, int tc, int ngg, int n_full, int X,
                                                                  int tail_ch, int Xt) {
    constexpr int CP = 52, GS = kDsdGch;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                                   // [GS*16][CP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Ps = Bs + GS * 16 * CP + wave * (kD2PsSize + 32 * GS);  // [2][32][20] skewed products
    float* Os = Ps + kD2PsSize;                                     // [tc][8] outputs of the item
    const int fi = lane & 15, kq = lane >> 4;

    // block -> (channel group g, workgroup column x of nwx)
    const int b = blockIdx.x;
    int g, x, nwx, n_ch;
    if (b < n_full * X) {
        const int xl = b & 7, t = b >> 3;
        g = t % n_full;
        x = (t / n_full) * 8 + xl;
        nwx = X;
        n_ch = GS;
    } else {
        g = n_full;
        x = b - n_full * X;
        nwx = Xt;
        n_ch = tail_ch;
    }
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(Bws + (int64_t)g * GS * 16 * CP);
        f32x4* dst = reinterpret_cast<f32x4*>(Bs);
        for (int i = tid; i < GS * 16 * CP / 4; i += kThreads) dst[i] = src[i];
        for (int i = lane; i < kD2PsSize + 32 * GS; i += 64) Ps[i] = 0.f;
    }
    __syncthreads();

    const int n_pairs = (n_ch + 1) >> 1;
    // A fragments are fetched two items ahead (an item takes ~1.5 us).  Ablation at 4096 tiles, 0.183 ms in all:
    // without the col2im and stores 0.160, without the D loads 0.156, without the G stores 0.176; the MFMAs
    // alone would take 0.104 ms (scripts/ubench/mfma_chains.hip: 155 TFLOP/s sustained).
    f32x4 pa[3], pb[3];
    float pl = 0.f, pm = 0.f;
#define DCS_LOAD_D(ks_, pa_, pl_)                                                                    \
    {                                                                                                \
        const float* dp = D + ((ks_) * (int64_t)H2 + fi) * CP + 12 * kq;                             \
        const bool in = fi < H2;                                                  \
        _Pragma("unroll") for (int u = 0; u < 3; ++u)                                                \
            pa_[u] = in ? *reinterpret_cast<const f32x4*>(dp + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};   \
        pl_ = in ? dp[48 - 11 * kq] : 0.f;                                                           \
    }
    const int64_t kstep = (int64_t)nwx * 4;
    int64_t ks = (int64_t)x * 4 + wave;
    if (ks < n_ks) DCS_LOAD_D(ks, pa, pl)
    if (ks + kstep < n_ks) DCS_LOAD_D(ks + kstep, pb, pm)
    for (; ks < n_ks; ks += kstep) {
        float af[13];
#pragma unroll
        for (int q = 0; q < 12; ++q) af[q] = pa[q >> 2][q & 3];
        af[12] = pl;
#pragma unroll
        for (int u = 0; u < 3; ++u) pa[u] = pb[u];
        pl = pm;
        if (ks + 2 * kstep < n_ks) DCS_LOAD_D(ks + 2 * kstep, pb, pm)
        for (int cp = 0; cp < n_pairs; ++cp) {
            const float* b0 = Bs + ((2 * cp) * 16 + fi) * CP + 12 * kq;
            const float* b1 = b0 + 16 * CP;
            f32x4 bf0[3], bf1[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                bf0[u] = *reinterpret_cast<const f32x4*>(b0 + 4 * u);
                bf1[u] = *reinterpret_cast<const f32x4*>(b1 + 4 * u);
            }
            const float bl0 = b0[48 - 11 * kq], bl1 = b1[48 - 11 * kq];
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 13; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], q < 12 ? bf0[q >> 2][q & 3] : bl0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], q < 12 ? bf1[q >> 2][q & 3] : bl1, acc1, 0, 0, 0);
            }
            // acc[e] = P[t' = 4 kq + e][dt = fi]  ->  Ps[c][t' + dt][dt]
            if (fi < kh) {
                float* w0 = Ps + (4 * kq + fi) * kD2PsStride + fi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (4 * kq + e < H2) {
                        w0[e * kD2PsStride] = acc0[e];
                        w0[kD2PsChan + e * kD2PsStride] = acc1[e];
                    }
                }
            }
            if (lane < 2 * tc) {
                const int cc = lane & 1, t = lane >> 1;
                const f32x4* r = reinterpret_cast<const f32x4*>(Ps + cc * kD2PsChan + t * kD2PsStride);
                const f32x4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                float sum = 0.f;
#pragma unroll
                for (int d = 0; d < 4; ++d) sum += r0[d];
#pragma unroll
                for (int d = 0; d < 4; ++d) sum += r1[d];
#pragma unroll
                for (int d = 0; d < 4; ++d) sum += r2[d];
#pragma unroll
                for (int d = 0; d < 4; ++d) sum += r3[d];
                Os[t * GS + 2 * cp + cc] = sum;
            }
        }
        if constexpr (SPLIT) {
            // lane -> (t, half of the 8 channels): 4 values, all three planes; a (t, plane) piece is 16 bytes, the two
            // halves of a piece are written by neighbouring lanes
            if (lane < 2 * tc) {
                const int t = lane >> 1, half = lane & 1;
                const f32x4 v = reinterpret_cast<const f32x4*>(Os)[lane];
                unsigned h[4], md[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf16_split3(v[j], h[j], md[j], lo[j]);
                u32x2* dst = reinterpret_cast<u32x2*>(G) + (((ks * ngg + g) * (int64_t)tc + t) * 3) * 2 + half;
                dst[0] = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
                dst[2] = u32x2{(md[0] >> 16) | md[1], (md[2] >> 16) | md[3]};
                dst[4] = u32x2{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3]};
            }
        } else {
            if (lane < tc * GS / 4) {
                const f32x4 v = reinterpret_cast<const f32x4*>(Os)[lane];
                reinterpret_cast<f32x4*>(G + (ks * ngg + g) * (int64_t)tc * GS)[lane] = v;
            }
        }
    }
#undef DCS_LOAD_D
}

// ------------------------------------------------------------------------------------------------
// Transposed conv1 (K = CI input channels, N = F bins) fused with everything after it.
// Workgroup = 16 rows x 128 bins; wave w owns 32 bins = two MFMA column blocks (lane fi holds the
// adjacent bins 2fi, 2fi+1, so results leave as 8-byte stores in 128-byte row segments); the three
// branch accumulators of one (row, bin) live in the same lane, so bias + rectify + soft mask +
// x mixture happen in registers.  FOLD: a row is an output frame t and the loop over m walks the tiles
// that cover it in the order of the reference's sequential cross-fade (owner tile first, then the
// blends), so the masked tiles never exist in HBM.  !FOLD: a row is one frame of one tile
// (predict_function2).
//
// What bounds it (scripts/ubench/mfma_valu.hip, MI355X): fp32 MFMA and VALU instructions do NOT
// overlap on a SIMD -- not within a wave and not across waves (3 waves/SIMD: MFMA alone 13.6 ns,
// +4 v_fma 20.5 ns, +8 25.1 ns per MFMA) -- so the kernel's time is MFMA cycles PLUS VALU cycles and
// every non-MFMA instruction counts.  Hence: two column blocks per A fragment (staging, LDS reads and
// per-row cross-fade state amortised over twice the bins), cross-fade weights from an LDS table built
// once per workgroup (no per-m validity arithmetic), packed fp32 math (v_pk_add/mul/fma_f32 on the
// (e, e+1) register pairs of an accumulator) in the epilogue, 128-bit LDS fragment reads (a K order
// chosen so that a lane's 13 channels are 3 aligned float4 + 1 float).
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V>
struct IntTag {
    static constexpr int value = V;
};
typedef int i32x2 __attribute__((ext_vector_type(2)));

// max(x, lo), lo >= 0, on both halves, as a SIGNED INTEGER max of the bit patterns: non-negative floats order like
// their bits and every negative float (sign bit set) is a negative integer, so this is exact for lo >= 0.  fmaxf on a
// value that comes straight out of an MFMA makes the compiler canonicalise it first (one more v_max_f32 per element);
// the integer form needs no canonical input.
__device__ __forceinline__ f32x2 max2(f32x2 x, float lo) {
    const int l = __builtin_bit_cast(int, lo);
    const i32x2 xi = __builtin_bit_cast(i32x2, x);
    const i32x2 r = {xi[0] > l ? xi[0] : l, xi[1] > l ? xi[1] : l};
    return __builtin_bit_cast(f32x2, r);
}

// channel multiplied in MFMA step q by the lanes of K-quarter kq: 12 contiguous channels + 1 of the last 4
__device__ __forceinline__ constexpr int final_chan(int q, int kq) { return q < 12 ? 12 * kq + q : 48 + kq; }

template <bool FOLD, int MODE, int CBW /* column blocks per wave: 2, or 1 when there are few rows */,
          int NBR /* dense branches that reach the output: 3 in the DSD graph (separate_dsd.py:228, the 4th source
                     re-uses branch 1), 4 in the stereo trainer's graph (trainCNN_ILD_DSD100.py:98-104) */>
__global__ __launch_bounds__(kThreads, NBR == 3 ? 3 : 2) void final_kernel(const DsdFinalArgs a, int n_colg) {
    constexpr int NQ = 13;   // CI / 4 MFMA steps
    constexpr int CI = 52, AS = 56;  // LDS row stride: multiple of 4 floats (128-bit reads and writes)
    constexpr int kABuf = NBR * 16 * AS;
    constexpr int kMaxM = 16;
    __shared__ __attribute__((aligned(16))) float As[2 * kABuf];  // double-buffered A set
    __shared__ __attribute__((aligned(16))) float up_t[kMaxM * 16];    // cross-fade weight of tile m on row i
    __shared__ __attribute__((aligned(16))) float down_t[kMaxM * 16];  // weight kept of what is already there
    __shared__ int meta_k0[16];
    __shared__ int meta_j0[16];
    __shared__ int meta_mlim[16];   // last covering tile (index m) that has a weight on the row; -1: a dead row

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    // XCD-aware renumbering (block b runs on XCD b % 8; bijective for any grid size): the column groups of
    // one row group become consecutive on ONE XCD, so its L2 serves the G rows they all read.  PMC before:
    // FETCH_SIZE 746 MB per launch at 4096 tiles against ~160 MB of distinct input.
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const unsigned swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int64_t clip = blockIdx.y;  // stacked clips of equal length (dcs_separate_batch): same fold, shifted buffers
    const unsigned rg = swz / (unsigned)n_colg;
    const int row0 = (int)rg * 16;    // rows, tiles and workgroups all fit 31 bits (checked by the launcher)
    // Which wave takes which 16*CBW bins rotates with the row group: F = 64k + 1 leaves the last column group one
    // live wave (the Nyquist bin), and a fixed choice would put all of those on the same SIMD of every CU
    // (45 instead of 40 waves' worth of MFMAs at F = 1025).
    const int colw = (int)(swz - rg * (unsigned)n_colg) * (64 * CBW) + ((wave + (int)rg) & 3) * (16 * CBW);
    const int col = colw + CBW * fi;  // this lane's bins: col (cb = 0), col + 1 (cb = 1)
    const int tc = a.tc, st = a.st, ov = a.ov, mmax = a.mmax;
    int n = (int)a.n, rows = (int)a.rows;
    if (a.clip_tab) {   // stacked clips of different lengths: this clip's own tile and frame counts
        rows = (int)a.clip_tab[kDcsClipTab * clip + 1];
        n = (int)a.clip_tab[kDcsClipTab * clip + 2];
        if (row0 >= rows) return;   // workgroup-uniform, before any barrier
    }
    // where this clip's mixture rows and G live: uniform pitch, or the compact layout's per-clip offsets
    int64_t mix_off = clip * a.mix_clip_stride, g_off = clip * a.g_clip_stride;
    if (a.clip_tab && a.clip_tab[kDcsClipTab * clip + 3] >= 0) {
        mix_off = a.clip_tab[kDcsClipTab * clip + 3] * a.mix_ld;
        g_off = a.clip_tab[kDcsClipTab * clip + 4] * a.g_tile_stride;
    }

    // ---- per-row state, once per workgroup: owner tile k0 / position j0 of each row, then the table of
    // cross-fade weights (util.py:321-325): the owner tile (m = 0) overwrites (up 1, down 0), a later
    // tile blends with up = rise[j], down = rise[ov-1-j], a tile that does not cover the row leaves it
    // (up 0, down 1).  32-bit arithmetic: a 64-bit division is ~100 instructions on this path's critical chain.
    if (tid < 16 * mmax) {
        const int i = tid & 15, m = tid >> 4;
        const int r = row0 + i;
        int k0 = 0, j0 = -1;
        if (r < rows) {
            if (FOLD) {
                int kk = (r < ov) ? 0 : (int)((unsigned)(r - ov) / (unsigned)st);
                if (kk > n - 1) kk = n - 1;
                const int jj = r - kk * st;
                if (jj < tc) {
                    k0 = kk;
                    j0 = jj;
                }
            } else {
                k0 = (int)((unsigned)r / (unsigned)tc);
                j0 = r - k0 * tc;
            }
        }
        if (m == 0) {
            meta_k0[i] = k0;
            meta_j0[i] = j0;
            int lim = -1;
            if (j0 >= 0) {
                lim = (int)((unsigned)j0 / (unsigned)st);
                if (lim > n - 1 - k0) lim = n - 1 - k0;
            }
            meta_mlim[i] = lim;
        }
        const int j = j0 - m * st;
        const bool valid = j0 >= 0 && j >= 0 && k0 + m < n;
        float up = 0.f, down = 1.f;
        if (m == 0) {
            up = valid ? 1.f : 0.f;
            down = 0.f;
        } else if (valid) {
            up = a.rise[j];
            down = a.rise[ov - 1 - j];
        }
        up_t[m * 16 + i] = up;
        down_t[m * 16 + i] = down;
    }

    const bool live = colw < a.F;  // a wave whose 32 bins are all padding only helps staging
    const bool cb1_live = CBW == 2 && colw + 1 < a.F;  // wave-uniform; false for the wave that holds only the Nyquist bin
    // B fragments: Bw[c][bin] of this lane's two bins -- constant for the whole workgroup
    float breg[CBW][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float* bp = a.Bw + (final_chan(q, kq) * a.ldb + col);   // 32-bit offset from a uniform base
        if constexpr (CBW == 2) {
            const f32x2 b = live ? *reinterpret_cast<const f32x2*>(bp) : f32x2{0.f, 0.f};
            breg[0][q] = b[0];
            breg[1][q] = b[1];
        } else {
            breg[0][q] = live ? *bp : 0.f;
        }
    }
    // mixture value of this lane's 4 rows x 2 bins
    const bool vec = CBW == 2 && ((a.mix_ld | a.out_ld) & 1) == 0;  // rows 8-byte aligned (the fused path pads F to 4)
    f32x4 mixv[CBW];
    const float* mix0 = a.mix + mix_off + (int64_t)row0 * a.mix_ld;   // workgroup-uniform
    const int rows_here = rows - row0 < 16 ? rows - row0 : 16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ri = kq * 4 + e;
        float m0 = 0.f, m1 = 0.f;
        if (ri < rows_here) {
            // CBW == 1 is the few-workgroups variant, where the duration is one workgroup's dependent chain: with the
            // full 64-bit address the compiler issues these loads ahead of the staging plan (15.3 -> 12.6 us at 32 tiles);
            // with many workgroups the cheaper 32-bit offset wins
            const float* mp = CBW == 1 ? a.mix + mix_off + (int64_t)(row0 + ri) * a.mix_ld + col
                                       : mix0 + (ri * (int)a.mix_ld + col);
            if (vec && col + 1 < a.F) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(mp);
                m0 = v[0];
                m1 = v[1];
            } else {
                if (col < a.F) m0 = mp[0];
                if (CBW == 2 && col + 1 < a.F) m1 = mp[1];
            }
        }
        mixv[0][e] = a.mix_scale * m0;
        if constexpr (CBW == 2) mixv[1][e] = a.mix_scale * m1;
    }
    // output BiasLayer: one value per source, or per (source, input channel) when two channels sit side by side
    const int bsel = (a.bias_half > 0 && col >= a.bias_half) ? 1 : 0;
    const int bstr = a.bias_half > 0 ? 2 : 1;
    const float bias0 = a.bias[0 * bstr + bsel], bias1 = a.bias[1 * bstr + bsel], bias2 = a.bias[2 * bstr + bsel],
                bias3 = a.bias[3 * bstr + bsel];
    // eps * rand_num with the unseeded draw replaced by 0.5 (separate_dsd.py:245,256); stereo trainer: 1e-12 * one
    // standard deviation of its N(0, 0.1) draw (trainCNN_ILD_DSD100.py:155,166)
    const float eps_r = MODE == 3 ? 1e-13f : 5e-19f;

    f32x4 res[CBW][4];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int c = 0; c < 4; ++c) res[cb][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();

    // ---- staging plan of this thread: up to NSL float4 slots of the [NBR branches][16 rows][CI] A set.
    // Slot (s, i, c4) reads G[k0_i + m][s][c4 / 2][j0_i - m*st][4 (c4 % 2) ..]; going from m to m+1 moves the
    // address by a constant, and the slot is needed iff tile m has a non-zero weight on row i.
    constexpr int slots = NBR * 16 * NQ;
    constexpr int NGG = (CI + kDsdGch - 1) / kDsdGch;  // G is [tile][branch][channel group][t][8]
    const int m_delta = (NBR * NGG * tc - st) * kDsdGch;
    const int kbase = meta_k0[0];
    // workgroup-uniform base; the per-slot offsets stay 32-bit
    const float* gbase = a.G + g_off + (int64_t)kbase * NBR * NGG * tc * kDsdGch;
    constexpr int NSL = (slots + kThreads - 1) / kThreads;   // float4 slots per thread: 3 (NBR = 3) or 4
    // A slot is loaded for EVERY covering tile: where tile m has no weight on the slot's row (up = 0, down = 1: the
    // row keeps what it has) the address is clamped to the last tile that does -- finite values that the epilogue
    // multiplies by up = 0 -- so the loop body has no data-dependent branch and no LDS lookup in front of its loads.
    int goff[NSL], dst[NSL], mlim[NSL];
    bool in_slot[NSL];
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
        const int idx = tid + u * kThreads;
        const int s = idx / (16 * NQ);
        const int rem = idx - s * 16 * NQ;
        const int i = rem / NQ, c4 = rem - i * NQ;
        const bool in = (u + 1) * kThreads <= slots || idx < slots;   // compile-time true for all but the last slot
        in_slot[u] = in;
        const int lim = in ? meta_mlim[i] : -1;
        const int j0 = lim >= 0 ? meta_j0[i] : 0;            // a dead row reads row 0 of the first tile
        const int dk = lim >= 0 ? meta_k0[i] - kbase : 0;
        mlim[u] = lim >= 0 ? lim : 0;
        dst[u] = (s * 16 + i) * AS + c4 * 4;
        goff[u] = (((dk * NBR + s) * NGG + (c4 >> 1)) * tc + j0) * kDsdGch + (c4 & 1) * 4;
    }
    f32x4 pre[NSL];
#define DCS_LOAD_A(m_)                                                                          \
    _Pragma("unroll") for (int u = 0; u < NSL; ++u) {                                           \
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                    \
        if ((u + 1) * kThreads <= slots || in_slot[u]) {                                        \
            const int mm = (m_) < mlim[u] ? (m_) : mlim[u];                                     \
            v = *reinterpret_cast<const f32x4*>(gbase + (goff[u] + mm * m_delta));              \
        }                                                                                       \
        pre[u] = v;                                                                             \
    }
#define DCS_STORE_A(buf_)                                                                       \
    _Pragma("unroll") for (int u = 0; u < NSL; ++u) {                                           \
        if ((u + 1) * kThreads <= slots || in_slot[u])                                          \
            *reinterpret_cast<f32x4*>(As + (buf_) * kABuf + dst[u]) = pre[u];                   \
    }

    DCS_LOAD_A(0)
    for (int m = 0; m < mmax; ++m) {
        // buffer m&1 was last read in iteration m-2; the barrier of iteration m-1 fences those reads
        DCS_STORE_A(m & 1)
        __syncthreads();
        if (m + 1 < mmax) DCS_LOAD_A(m + 1)
        if (!live) continue;
        auto compute = [&](auto ncb_tag) __attribute__((always_inline)) {
        constexpr int NCB = decltype(ncb_tag)::value;   // column blocks with live bins: CBW, or 1 (second block all padding)
        const float* Ab = As + (m & 1) * kABuf + fi * AS + 12 * kq;
        // the output bias (separate_dsd.py:232) rides in as the MFMAs' initial accumulator value: no add afterwards
        f32x4 acc[4][CBW];   // acc[3] only with NBR == 4
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[0][cb] = f32x4{bias0, bias0, bias0, bias0};
            acc[1][cb] = f32x4{bias1, bias1, bias1, bias1};
            acc[2][cb] = f32x4{bias2, bias2, bias2, bias2};
            if (NBR == 4) acc[3][cb] = f32x4{bias3, bias3, bias3, bias3};
        }
        // A fragments of K-quarter kq: channels 12kq .. 12kq+11 (three float4) and channel 48+kq
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af[NBR];
#pragma unroll
            for (int s = 0; s < NBR; ++s) {
                if (g < 3)
                    af[s] = *reinterpret_cast<const f32x4*>(Ab + s * 16 * AS + 4 * g);
                else
                    af[s][0] = As[(m & 1) * kABuf + (s * 16 + fi) * AS + 48 + kq];
            }
#pragma unroll
            for (int t = 0; t < (g < 3 ? 4 : 1); ++t) {
                const int q = 4 * g + t;
#pragma unroll
                for (int s = 0; s < NBR; ++s)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][t], breg[cb][q], acc[s][cb], 0, 0, 0);
            }
        }
        // bias + rectify (separate_dsd.py:234), soft mask (:258-271), cross-fade (util.py:321-325) as one
        // fused multiply-add per source: res = down*res + s_c * (rcp(den) * mix * up), on (e, e+1) pairs.
        const f32x4 up4 = *reinterpret_cast<const f32x4*>(up_t + m * 16 + kq * 4);
        const f32x4 down4 = *reinterpret_cast<const f32x4*>(down_t + m * 16 + kq * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 up = {up4[2 * h], up4[2 * h + 1]};
            const f32x2 down = {down4[2 * h], down4[2 * h + 1]};
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const f32x2 x0 = {acc[0][cb][2 * h], acc[0][cb][2 * h + 1]};
                const f32x2 x1 = {acc[1][cb][2 * h], acc[1][cb][2 * h + 1]};
                const f32x2 x2 = {acc[2][cb][2 * h], acc[2][cb][2 * h + 1]};
                // 4th source: branch fc12 again in the DSD graph (:228; its accumulator carries bias1, so bias3 - bias1 is
                // added), its own branch in the stereo trainer's
                const f32x2 x3 = NBR == 4 ? f32x2{acc[NBR - 1][cb][2 * h], acc[NBR - 1][cb][2 * h + 1]} : x1 + (bias3 - bias1);
                // rectify; convention A adds eps*r = 5e-19 to every rectified value: max(x, 0) + eps == max(x, eps) in
                // float32 except for 0 < x < ~1e-11, where the two differ by at most eps itself
                const float lo = MODE == 0 ? eps_r : 0.f;
                const f32x2 p0 = max2(x0, lo), p1 = max2(x1, lo), p2 = max2(x2, lo), p3 = max2(x3, lo);
                const f32x2 mu = f32x2{mixv[cb][2 * h], mixv[cb][2 * h + 1]} * up;
                f32x2 s0 = p0, s1 = p1, s2 = p2, s3 = p3, w = up;
                if (MODE == 0) {  // convention A: m_i = s_i / sum(s), s_i = p_i + eps*r
                    const f32x2 den = ((s0 + s1) + s2) + s3;
                    w = f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])} * mu;
                } else if (MODE == 1 || MODE == 3) {  // convention B: m_i = p_i / (sum(p) + eps*r)
                    const f32x2 den = (((p0 + p1) + p2) + p3) + eps_r;
                    w = f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])} * mu;
                }
                const f32x2 o0 = __builtin_elementwise_fma(down, f32x2{res[cb][0][2 * h], res[cb][0][2 * h + 1]}, s0 * w);
                const f32x2 o1 = __builtin_elementwise_fma(down, f32x2{res[cb][1][2 * h], res[cb][1][2 * h + 1]}, s1 * w);
                const f32x2 o2 = __builtin_elementwise_fma(down, f32x2{res[cb][2][2 * h], res[cb][2][2 * h + 1]}, s2 * w);
                const f32x2 o3 = __builtin_elementwise_fma(down, f32x2{res[cb][3][2 * h], res[cb][3][2 * h + 1]}, s3 * w);
                res[cb][0][2 * h] = o0[0]; res[cb][0][2 * h + 1] = o0[1];
                res[cb][1][2 * h] = o1[0]; res[cb][1][2 * h + 1] = o1[1];
                res[cb][2][2 * h] = o2[0]; res[cb][2][2 * h + 1] = o2[1];
                res[cb][3][2 * h] = o3[0]; res[cb][3][2 * h + 1] = o3[1];
            }
        }
        };
        if (CBW == 2 && !cb1_live)
            compute(IntTag<1>{});
        else
            compute(IntTag<CBW>{});
    }
#undef DCS_LOAD_A
#undef DCS_STORE_A

    {
        float* out0 = a.out + clip * a.out_clip_stride + (int64_t)row0 * a.out_ld;   // workgroup-uniform; offsets below are 32-bit
        // stereo trainer: source = mask * input + eps*r (trainCNN_ILD_DSD100.py:180); the cross-fade weights of a frame
        // sum to one, so the constant is added once, after the fold
        if (MODE == 3) {
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                for (int c = 0; c < 4; ++c) res[cb][c] += eps_r;
        }
        if (CBW == 2 && vec && colw + 16 * CBW <= a.F) {
            // the whole wave inside the spectrum (wave-uniform): straight 8-byte stores
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ri = kq * 4 + e;
                if (ri < rows_here) {
                    float* op = out0 + (ri * (int)a.out_ld + col);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        *reinterpret_cast<f32x2*>(op + c * a.out_src_stride) = f32x2{res[0][c][e], res[CBW - 1][c][e]};
                }
            }
        } else if (col < a.F) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ri = kq * 4 + e;
                if (ri < rows_here) {
                    float* op = out0 + (ri * (int)a.out_ld + col);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        op[c * a.out_src_stride] = res[0][c][e];
                        if (CBW == 2 && col + 1 < a.F) op[c * a.out_src_stride + 1] = res[CBW - 1][c][e];
                    }
                }
            }
        }
    }
}

}  // namespace

int dcs_launch_dsd_deconv2(dcs_ctx* ctx, const float* D, const float* Bw, const float* Bws, float* G, int64_t n_ks,
                           int H2, int CP, int CI, int kh, int tc, int NG, int GS, int gcols, void* Gs, const void* Bq) {
    if (n_ks <= 0) return DCS_OK;
    const int nrb = (H2 + 15) / 16;
    if (nrb > 2) DCS_FAIL(DCS_EUNSUPPORTED, "deconv2: conv2 output height %d > 32", H2);
    if (CP != 52) DCS_FAIL(DCS_EUNSUPPORTED, "deconv2: built for 50 conv2 filters (CP=52), got CP=%d", CP);
    const size_t lds = (size_t)16 * nrb * ((CP + 2) + gcols) * sizeof(float);
    if (lds > 64 * 1024) DCS_FAIL(DCS_EUNSUPPORTED, "deconv2: LDS %zu too large", lds);
    DcsTimer tm(ctx, DCS_TAG_DECONV2);
    static const int force = getenv("DCS_DECONV2") ? atoi(getenv("DCS_DECONV2")) : 0;  // 1 one-shot, 2 streaming
    const int ngg = (CI + kDsdGch - 1) / kDsdGch;
    const bool can_stream = nrb == 1 && kh <= 16 && tc <= 32 && H2 + kh - 1 <= 32;
    const bool stream = can_stream && (force ? force == 2 : n_ks >= 4 * (int64_t)ctx->n_cu);
    if (stream && Gs && Bq) {
        // the consumer is the bf16x3 final kernel: the transposed conv2 runs on the bf16 matrix pipe too and writes the
        // three planes of G (dsd_bf16x3.hip)
        const int rc = dcs_launch_dsd_deconv2_bf16(ctx, D, Bq, Gs, n_ks, H2, CP, CI, kh, tc);
        tm.done();
        return rc;
    }
    if (stream) {
        // 3 workgroups per CU (50.6 KB of LDS each), one round: X columns for each full channel group, and a
        // proportionally smaller share for the last, partial group
        const int n_full = CI / kDsdGch, tail_ch = CI - n_full * kDsdGch;
        const int slots = 3 * ctx->n_cu;
        const double units = n_full + (tail_ch ? (double)((tail_ch + 1) / 2) / (kDsdGch / 2) : 0.0);
        int X = (int)(slots / units) / 8 * 8;
        if (X < 8) X = 8;
        int Xt = tail_ch ? slots - n_full * X : 0;
        if (tail_ch && Xt < 1) Xt = 1;
        const size_t lds2 = ((size_t)kDsdGch * 16 * CP + 4 * (kD2PsSize + 32 * kDsdGch)) * sizeof(float);
        auto kern = Gs ? deconv2_stream_kernel<true> : deconv2_stream_kernel<false>;
        if (lds2 > 48 * 1024)
            DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds2));
        // Gs: the consumer is the bf16x3 final kernel -- the three bf16 planes are written directly, f32 G is not
        hipLaunchKernelGGL(kern, dim3((unsigned)(n_full * X + Xt)), dim3(kThreads), lds2, ctx->stream, D, Bws,
                           Gs ? reinterpret_cast<float*>(Gs) : G, n_ks, H2, kh, tc, ngg, n_full, X, tail_ch, Xt);
        tm.done();
    } else {
        hipLaunchKernelGGL(deconv2_kernel<13>, dim3((unsigned)n_ks, (unsigned)NG), dim3(kThreads), lds, ctx->stream, D,
                           Bw, G, H2, CP, CI, kh, tc, GS, gcols);
        tm.done();
        if (Gs) DCS_CHECK(dcs_launch_dsd_gsplit(ctx, G, Gs, n_ks, tc, ngg));   // few tiles: one-shot kernel, then one pass that splits G
    }
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

int dsd_final_cbw(const dcs_ctx* ctx, int64_t rows, int F, int64_t n_clips) {
    static const int force = getenv("DCS_FINAL_CBW") ? atoi(getenv("DCS_FINAL_CBW")) : 0;
    if (force) return force;
    if (n_clips < 1) n_clips = 1;
    return dcs_cdiv(rows, 16) * ((F + 127) / 128) * n_clips >= 3 * (int64_t)ctx->n_cu ? 2 : 1;
}

bool dsd_final_bf16x3(const dcs_ctx* ctx, int64_t rows, int F, int64_t n_clips, int CI, int mask_mode) {
    return CI == 52 && mask_mode < 2 && dsd_final_cbw(ctx, rows, F, n_clips) == 2;
}

int dcs_launch_dsd_final(dcs_ctx* ctx, const DsdFinalArgs& a, bool fold) {
    if (a.rows <= 0) return DCS_OK;
    if (a.CI != 52) DCS_FAIL(DCS_EUNSUPPORTED, "final: built for 50 conv1 filters (CI=52), got CI=%d", a.CI);
    if (a.mmax < 1 || a.mmax > 16) DCS_FAIL(DCS_EUNSUPPORTED, "final: %d covering tiles per frame (max 16)", a.mmax);
    // workgroup = 16 rows x 128 bins (two column blocks per wave); with few rows (the 32-tile step has 12 row
    // groups) 16 x 64 gives twice the workgroups and half the work in each: 32 tiles 17.8 vs 19.1 us
    const int64_t n_rg = dcs_cdiv(a.rows, 16);
    const unsigned n_clips = a.n_clips > 0 ? (unsigned)a.n_clips : 1u;
    const int cbw = dsd_final_cbw(ctx, a.rows, a.F, n_clips);
    const int n_colg = (a.F + 64 * cbw - 1) / (64 * cbw);
    if (a.ldb < n_colg * 64 * cbw || (a.ldb & 1))
        DCS_FAIL(DCS_EINVAL, "final: weight pitch %d < %d", a.ldb, n_colg * 64 * cbw);
    const int64_t n_wg = n_rg * n_colg;
    if (n_wg > 0x7fffffff) DCS_FAIL(DCS_EUNSUPPORTED, "final: %lld workgroups", (long long)n_wg);
    DcsTimer tm(ctx, DCS_TAG_FINAL);
    if (a.Gs && a.Bpk && fold && a.nbr != 4 && cbw == 2 && a.mask_mode < 2 && a.bias_half == 0) {   // bf16x3 path (the caller asked dsd_final_bf16x3)
        const int rc = dcs_launch_dsd_final_bf16x3(ctx, a, n_colg, n_wg, n_clips);
        tm.done();
        return rc;
    }
#define DCS_FINAL(FOLD_, MODE_)                                                                                      \
    do {                                                                                                             \
        if (cbw == 2)                                                                                                \
            hipLaunchKernelGGL((final_kernel<FOLD_, MODE_, 2, 3>), dim3((unsigned)n_wg, n_clips), dim3(kThreads), 0,    \
                               ctx->stream, a, n_colg);                                                              \
        else                                                                                                         \
            hipLaunchKernelGGL((final_kernel<FOLD_, MODE_, 1, 3>), dim3((unsigned)n_wg, n_clips), dim3(kThreads), 0,    \
                               ctx->stream, a, n_colg);                                                              \
    } while (0)
    if (a.nbr == 4) {   // stereo trainer's graph: fused path only (fold, its own mask) or raw output
        if (a.mask_mode != 3 && a.mask_mode != 2) DCS_FAIL(DCS_EINVAL, "final: 4-branch graph with mask mode %d", a.mask_mode);
        if (a.mask_mode == 3) {
            if (fold) hipLaunchKernelGGL((final_kernel<true, 3, 1, 4>), dim3((unsigned)(n_rg * ((a.F + 63) / 64)), n_clips),
                                         dim3(kThreads), 0, ctx->stream, a, (a.F + 63) / 64);
            else hipLaunchKernelGGL((final_kernel<false, 3, 1, 4>), dim3((unsigned)(n_rg * ((a.F + 63) / 64)), n_clips),
                                    dim3(kThreads), 0, ctx->stream, a, (a.F + 63) / 64);
        } else {
            if (fold) hipLaunchKernelGGL((final_kernel<true, 2, 1, 4>), dim3((unsigned)(n_rg * ((a.F + 63) / 64)), n_clips),
                                         dim3(kThreads), 0, ctx->stream, a, (a.F + 63) / 64);
            else hipLaunchKernelGGL((final_kernel<false, 2, 1, 4>), dim3((unsigned)(n_rg * ((a.F + 63) / 64)), n_clips),
                                    dim3(kThreads), 0, ctx->stream, a, (a.F + 63) / 64);
        }
        tm.done();
        DCS_HIP(hipGetLastError());
        return DCS_OK;
    }
    if (fold) {
        if (a.mask_mode == 0) DCS_FINAL(true, 0);
        else if (a.mask_mode == 1) DCS_FINAL(true, 1);
        else DCS_FINAL(true, 2);
    } else {
        if (a.mask_mode == 0) DCS_FINAL(false, 0);
        else if (a.mask_mode == 1) DCS_FINAL(false, 1);
        else DCS_FINAL(false, 2);
    }
#undef DCS_FINAL
    tm.done();
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}
