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
    float* Gp = G + ks * (int64_t)tc * CI;
    for (int o = tid; o < tc * GS; o += kThreads) {
        const int t = o / GS, c = o - t * GS;
        if (ci0 + c >= CI) continue;
        int lo = t - (H2 - 1);
        if (lo < 0) lo = 0;
        const int hi = t < kh - 1 ? t : kh - 1;
        float sum = 0.f;
        for (int dt = lo; dt <= hi; ++dt) sum += P[(t - dt) * gcols + c * kh + dt];
        Gp[t * CI + ci0 + c] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// Many-tiles variant of deconv2_kernel: persistent workgroups.  With one workgroup per (tile, branch,
// group) every workgroup streams its 52 x 208 weight slice (43 KB) from L2 for 42 MFMAs per wave -- 2.1 GB
// of L2 traffic per launch at 4096 tiles.  Here a workgroup keeps the slice of its channel group in LDS
// and walks a strided list of (tile, branch) pairs; the next pair's 16 x 52 input rows are prefetched
// into registers while the current pair is multiplied and reduced.
// ------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(kThreads) void deconv2_persistent_kernel(const float* __restrict__ D,
                                                                      const float* __restrict__ Bw,
                                                                      float* __restrict__ G, int64_t n_ks, int H2,
                                                                      int CP, int CI, int kh, int tc, int GS, int gcols) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int as = CP + 2;
    float* Bs = smem;                      // [CP][gcols]   (gcols % 32 == 16: rows kq, kq+1 hit disjoint banks)
    float* As = Bs + CP * gcols;           // [16][CP+2]
    float* P = As + 16 * as;               // [16][gcols]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.y;
    const int ldb = gcols * (int)gridDim.y;
    for (int idx = tid; idx < CP * gcols; idx += kThreads) {
        const int r = idx / gcols, c = idx - r * gcols;
        Bs[idx] = Bw[(int64_t)r * ldb + grp * gcols + c];
    }
    const int a_slots = H2 * (CP >> 2);    // float4 slots of one (tile, branch) input block (H2 <= 16 rows)
    const int ncb = gcols >> 4;
    const int ci0 = grp * GS;
    f32x4 pre = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t ks = blockIdx.x;
    if (ks < n_ks && tid < a_slots) pre = *reinterpret_cast<const f32x4*>(D + ks * (int64_t)H2 * CP + tid * 4);
    for (; ks < n_ks; ks += gridDim.x) {
        __syncthreads();  // previous pair: P fully reduced, As fully read
        if (tid < 16 * (CP >> 2)) {
            const int r = tid / (CP >> 2), c4 = tid - r * (CP >> 2);
            const f32x4 v = (tid < a_slots) ? pre : f32x4{0.f, 0.f, 0.f, 0.f};
            float* d = As + r * as + c4 * 4;
            *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
            *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
        }
        __syncthreads();
        const int64_t nxt = ks + gridDim.x;
        if (nxt < n_ks && tid < a_slots) pre = *reinterpret_cast<const f32x4*>(D + nxt * (int64_t)H2 * CP + tid * 4);
        float a0[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) a0[q] = As[fi * as + 4 * q + kq];
        for (int cb = wave; cb < ncb; cb += 4) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], Bs[(4 * q + kq) * gcols + cb * 16 + fi], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) P[(kq * 4 + e) * gcols + cb * 16 + fi] = acc[e];
        }
        __syncthreads();
        float* Gp = G + ks * (int64_t)tc * CI;
        for (int o = tid; o < tc * GS; o += kThreads) {
            const int t = o / GS, c = o - t * GS;
            if (ci0 + c >= CI) continue;
            int lo = t - (H2 - 1);
            if (lo < 0) lo = 0;
            const int hi = t < kh - 1 ? t : kh - 1;
            float sum = 0.f;
            for (int dt = lo; dt <= hi; ++dt) sum += P[(t - dt) * gcols + c * kh + dt];
            Gp[t * CI + ci0 + c] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Transposed conv1 (K = CI input channels, N = F bins) fused with everything after it.
// Workgroup = 16 rows x 64 bins; wave w owns 16 bins; the three branch accumulators of one
// (row, bin) live in the same lane, so bias + rectify + soft mask + x mixture happen in
// registers.  FOLD: a row is an output frame t and the loop over m walks the tiles that cover it
// in the order of the reference's sequential cross-fade (owner tile first, then the blends), so
// the masked tiles never exist in HBM.  !FOLD: a row is one frame of one tile (predict_function2).
// ------------------------------------------------------------------------------------------------
template <bool FOLD, int NQ>
__global__ __launch_bounds__(kThreads) void final_kernel(const DsdFinalArgs a) {
    constexpr int NBR = 3;      // dense branches that reach the output (separate_dsd.py:228)
    constexpr int NQ_MAX = NQ;  // CI / 4, compile time: the K loop unrolls into one block of MFMAs
    constexpr int kABuf = NBR * 16 * (64 + 2);
    __shared__ __attribute__((aligned(16))) float As[2 * kABuf];  // double-buffered A set
    __shared__ int meta_k0[16];
    __shared__ int meta_j0[16];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    // XCD-aware renumbering (block b runs on XCD b % 8; bijective for any grid size): the column groups of
    // one row group become consecutive on ONE XCD, so its L2 serves the G rows they all read.  PMC before:
    // FETCH_SIZE 746 MB per launch at 4096 tiles against ~160 MB of distinct input.
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const unsigned swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int n_colg = a.ldb / 64;
    const int64_t row0 = (int64_t)(swz / n_colg) * 16;
    const int col = (int)(swz % n_colg) * 64 + wave * 16 + fi;
    constexpr int nq = NQ;
    const int CI = a.CI, as = a.CI + 2;
    const int tc = a.tc, st = a.st, ov = a.ov;
    const int64_t n = a.n;

    if (tid < 16) {
        const int64_t r = row0 + tid;
        int k0 = 0, j0 = -1;
        if (r < a.rows) {
            if (FOLD) {
                int64_t kk = (r < ov) ? 0 : (r - ov) / st;
                if (kk > n - 1) kk = n - 1;
                const int64_t jj = r - kk * st;
                if (jj < tc) {
                    k0 = (int)kk;
                    j0 = (int)jj;
                }
            } else {
                k0 = (int)(r / tc);
                j0 = (int)(r - (int64_t)k0 * tc);
            }
        }
        meta_k0[tid] = k0;
        meta_j0[tid] = j0;
    }

    // B fragments: Bw[c][bin], this lane's bin, rows 4q+kq -- constant for the whole workgroup
    float breg[NQ_MAX];
#pragma unroll
    for (int q = 0; q < NQ_MAX; ++q) breg[q] = a.Bw[(int64_t)(4 * q + kq) * a.ldb + col];

    // mixture value of this lane's 4 (row, bin) cells
    float mixv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t r = row0 + kq * 4 + e;
        mixv[e] = (r < a.rows && col < a.F) ? a.mix_scale * a.mix[r * a.mix_ld + col] : 0.f;
    }
    const float bias0 = a.bias[0], bias1 = a.bias[1], bias2 = a.bias[2], bias3 = a.bias[3];
    const float eps_r = 5e-19f;  // eps * rand_num with the unseeded draw replaced by 0.5 (separate_dsd.py:245,256)

    float res[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) res[c][e] = 0.f;

    __syncthreads();

    // ---- staging plan of this thread: up to 3 float4 slots of the [3 branches][16 rows][CI] A set.
    // Slot (s, i, c4) reads G[k0_i + m][s][j0_i - m*st][4 c4 ..]; going from m to m+1 moves the
    // address by the constant (NBR*tc - st)*CI, so only validity has to be re-evaluated per m.
    const int slots = NBR * 16 * nq;
    const int64_t m_delta = ((int64_t)NBR * tc - st) * CI;
    const float* src[3];
    int dst[3], sk0[3], sj0[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int idx = tid + u * kThreads;
        const int s = idx / (16 * nq);
        const int rem = idx - s * 16 * nq;
        const int i = rem / nq, c4 = rem - i * nq;
        const bool in = idx < slots;
        sk0[u] = in ? meta_k0[i] : 0;
        sj0[u] = in ? meta_j0[i] : -1;
        dst[u] = in ? (s * 16 + i) * as + c4 * 4 : 0;
        src[u] = a.G + (((int64_t)sk0[u] * NBR + s) * tc + (sj0[u] < 0 ? 0 : sj0[u])) * (int64_t)CI + c4 * 4;
    }
    f32x4 pre[3];
#define DCS_LOAD_A(m_)                                                                          \
    _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                             \
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                    \
        if (sj0[u] >= 0 && sj0[u] - (m_) * st >= 0 && (int64_t)sk0[u] + (m_) < n)               \
            v = *reinterpret_cast<const f32x4*>(src[u] + (m_) * m_delta);                       \
        pre[u] = v;                                                                             \
    }
#define DCS_STORE_A(buf_)                                                                       \
    _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                             \
        if (tid + u * kThreads < slots) {                                                       \
            float* d = As + (buf_) * kABuf + dst[u];                                            \
            *reinterpret_cast<float2*>(d) = make_float2(pre[u][0], pre[u][1]);                  \
            *reinterpret_cast<float2*>(d + 2) = make_float2(pre[u][2], pre[u][3]);              \
        }                                                                                       \
    }

    DCS_LOAD_A(0)
    DCS_STORE_A(0)
    __syncthreads();

    for (int m = 0; m < a.mmax; ++m) {
        const bool more = m + 1 < a.mmax;
        if (more) DCS_LOAD_A(m + 1)  // in flight while this m's MFMAs and epilogue run
        const float* Ab = As + (m & 1) * kABuf;

        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.prio) __builtin_amdgcn_s_setprio(1);  // matrix phase outranks the other waves' epilogues
#pragma unroll
        for (int q = 0; q < NQ_MAX; ++q) {
            const float b = breg[q];
            const int off = fi * as + 4 * q + kq;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ab[off], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ab[16 * as + off], b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ab[32 * as + off], b, acc2, 0, 0, 0);
        }
        if (a.prio) __builtin_amdgcn_s_setprio(0);

        // ---- bias + rectify (separate_dsd.py:234), soft mask (:258-271), cross-fade (util.py:321-325)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = kq * 4 + e;
            const int j0 = meta_j0[i];
            const int j = j0 - m * st;
            const bool valid = (j0 >= 0) && (j >= 0) && ((int64_t)meta_k0[i] + m < n);
            // output channels 0..3 use branches 0,1,2,1
            const float p0 = fmaxf(acc0[e] + bias0, 0.f);
            const float p1 = fmaxf(acc1[e] + bias1, 0.f);
            const float p2 = fmaxf(acc2[e] + bias2, 0.f);
            const float p3 = fmaxf(acc1[e] + bias3, 0.f);
            float v0, v1, v2, v3;
            if (a.mask_mode == 0) {  // convention A: m_i = s_i / sum(s), s_i = p_i + eps*r
                const float s0 = p0 + eps_r, s1 = p1 + eps_r, s2 = p2 + eps_r, s3 = p3 + eps_r;
                const float den = ((s0 + s1) + s2) + s3;             // >= 4*eps*r: never zero or denormal
                const float w = __builtin_amdgcn_rcpf(den) * mixv[e];  // one v_rcp_f32 (1 ulp) for 4 masks
                v0 = s0 * w; v1 = s1 * w; v2 = s2 * w; v3 = s3 * w;
            } else if (a.mask_mode == 1) {  // convention B: m_i = p_i / (sum(p) + eps*r)
                const float den = (((p0 + p1) + p2) + p3) + eps_r;
                const float w = __builtin_amdgcn_rcpf(den) * mixv[e];
                v0 = p0 * w; v1 = p1 * w; v2 = p2 * w; v3 = p3 * w;
            } else {  // raw network output (get_output)
                v0 = p0; v1 = p1; v2 = p2; v3 = p3;
            }
            if (m == 0) {
                if (valid) { res[0][e] = v0; res[1][e] = v1; res[2][e] = v2; res[3][e] = v3; }
            } else if (valid) {
                const float up = a.rise[j], down = a.rise[ov - 1 - j];
                res[0][e] = down * res[0][e] + up * v0;
                res[1][e] = down * res[1][e] + up * v1;
                res[2][e] = down * res[2][e] + up * v2;
                res[3][e] = down * res[3][e] + up * v3;
            }
        }
        if (more) {
            DCS_STORE_A((m + 1) & 1)  // the other buffer: last read in iteration m-1, fenced by its barrier
            __syncthreads();
        }
    }
#undef DCS_LOAD_A
#undef DCS_STORE_A

    if (col < a.F) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t r = row0 + kq * 4 + e;
            if (r < a.rows) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a.out[c * a.out_src_stride + r * a.out_ld + col] = res[c][e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Many-rows variant of final_kernel: one INDEPENDENT wave per 16 rows x (16*CBW) bins, no workgroup
// barrier.  (PMC on the cooperative kernel: 4 waves meeting every 39 MFMAs leave the matrix pipe idle
// ~55 % of the time.)  Each A fragment read from the wave-private LDS tile feeds CBW MFMAs; with CBW = 2
// the 65 (F=1025) / 33 (F=513) column blocks split into 33 / 17 groups with one block of padding.
// A wave's LDS operations are performed in program order, so staging the next covering tile's rows after
// the current MFMAs needs no fence.  Blocks are renumbered so that the column groups of one row group run
// on the same XCD (block b runs on XCD b % 8) and share its L2 for the G rows.
// ------------------------------------------------------------------------------------------------
template <bool FOLD, int CBW>
__global__ __launch_bounds__(64, 2) void final_wave_kernel(const DsdFinalArgs a, int n_colg) {
    constexpr int NBR = 3, NQ = 13, AS = 54, SLOTS = NBR * 16 * NQ, PER = (SLOTS + 63) / 64;
    __shared__ __attribute__((aligned(16))) float As[NBR * 16 * AS];
    __shared__ int meta_k0[16];
    __shared__ int meta_j0[16];
    const int lane = threadIdx.x;
    const int fi = lane & 15, kq = lane >> 4;
    // XCD-aware renumbering (bijective for any grid size)
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const unsigned swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int64_t row0 = (int64_t)(swz / n_colg) * 16;
    const int col0 = (int)(swz % n_colg) * (16 * CBW);
    const int CI = a.CI;
    const int tc = a.tc, st = a.st, ov = a.ov;
    const int64_t n = a.n;

    if (lane < 16) {
        const int64_t r = row0 + lane;
        int k0 = 0, j0 = -1;
        if (r < a.rows) {
            if (FOLD) {
                int64_t kk = (r < ov) ? 0 : (r - ov) / st;
                if (kk > n - 1) kk = n - 1;
                const int64_t jj = r - kk * st;
                if (jj < tc) {
                    k0 = (int)kk;
                    j0 = (int)jj;
                }
            } else {
                k0 = (int)(r / tc);
                j0 = (int)(r - (int64_t)k0 * tc);
            }
        }
        meta_k0[lane] = k0;
        meta_j0[lane] = j0;
    }

    float breg[CBW][NQ];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int q = 0; q < NQ; ++q) breg[cb][q] = a.Bw[(int64_t)(4 * q + kq) * a.ldb + col0 + cb * 16 + fi];
    float mixv[CBW][4];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t r = row0 + kq * 4 + e;
            const int col = col0 + cb * 16 + fi;
            mixv[cb][e] = (r < a.rows && col < a.F) ? a.mix_scale * a.mix[r * a.mix_ld + col] : 0.f;
        }
    const float bias0 = a.bias[0], bias1 = a.bias[1], bias2 = a.bias[2], bias3 = a.bias[3];
    const float eps_r = 5e-19f;
    float res[CBW][4][4];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) res[cb][c][e] = 0.f;

    // staging plan (see final_kernel): slot -> (branch s, row i, float4 c4).  Kept compact (32-bit offsets
    // relative to the first row's tile, packed row state) so that two waves fit on a SIMD.
    const int m_delta = (NBR * tc - st) * CI;
    const int kbase = meta_k0[0];
    const float* gbase = a.G + (int64_t)kbase * NBR * tc * CI;
    int dst[PER], off[PER], kj[PER];  // kj = (k0 - kbase) << 8 | (j0 + 1); 0 = row not covered
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = lane + u * 64;
        const int s = idx / (16 * NQ);
        const int rem = idx - s * 16 * NQ;
        const int i = rem / NQ, c4 = rem - i * NQ;
        const bool in = idx < SLOTS && meta_j0[i] >= 0;
        const int dk = in ? meta_k0[i] - kbase : 0;
        const int j0 = in ? meta_j0[i] : -1;
        kj[u] = in ? ((dk << 8) | (j0 + 1)) : 0;
        dst[u] = (idx < SLOTS) ? (s * 16 + i) * AS + c4 * 4 : -1;
        off[u] = ((dk * NBR + s) * tc + (j0 < 0 ? 0 : j0)) * CI + c4 * 4;
    }
    f32x4 pre[PER];
#define DCS_LOAD_A(m_)                                                                          \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                           \
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                    \
        const int j0p = kj[u] & 255;                                                            \
        if (j0p > 0 && j0p - 1 - (m_) * st >= 0 && (int64_t)kbase + (kj[u] >> 8) + (m_) < n)    \
            v = *reinterpret_cast<const f32x4*>(gbase + off[u] + (m_) * m_delta);               \
        pre[u] = v;                                                                             \
    }
#define DCS_STORE_A()                                                                           \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                           \
        if (dst[u] >= 0) {                                                                      \
            float* d = As + dst[u];                                                             \
            *reinterpret_cast<float2*>(d) = make_float2(pre[u][0], pre[u][1]);                  \
            *reinterpret_cast<float2*>(d + 2) = make_float2(pre[u][2], pre[u][3]);              \
        }                                                                                       \
    }
    DCS_LOAD_A(0)
    for (int m = 0; m < a.mmax; ++m) {
        DCS_STORE_A()  // after the previous iteration's fragment reads (program order within the wave)
        if (m + 1 < a.mmax) DCS_LOAD_A(m + 1)
        f32x4 acc[NBR][CBW];
#pragma unroll
        for (int s = 0; s < NBR; ++s)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) acc[s][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int off = fi * AS + 4 * q + kq;
            const float a0 = As[off], a1 = As[16 * AS + off], a2 = As[32 * AS + off];
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, breg[cb][q], acc[0][cb], 0, 0, 0);
                acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, breg[cb][q], acc[1][cb], 0, 0, 0);
                acc[2][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, breg[cb][q], acc[2][cb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = kq * 4 + e;
            const int j0 = meta_j0[i];
            const int j = j0 - m * st;
            const bool valid = (j0 >= 0) && (j >= 0) && ((int64_t)meta_k0[i] + m < n);
            float up = 0.f, down = 0.f;
            if (m > 0 && valid) {
                up = a.rise[j];
                down = a.rise[ov - 1 - j];
            }
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                const float p0 = fmaxf(acc[0][cb][e] + bias0, 0.f);
                const float p1 = fmaxf(acc[1][cb][e] + bias1, 0.f);
                const float p2 = fmaxf(acc[2][cb][e] + bias2, 0.f);
                const float p3 = fmaxf(acc[1][cb][e] + bias3, 0.f);
                float v0, v1, v2, v3;
                if (a.mask_mode == 0) {
                    const float s0 = p0 + eps_r, s1 = p1 + eps_r, s2 = p2 + eps_r, s3 = p3 + eps_r;
                    const float den = ((s0 + s1) + s2) + s3;
                    const float w = __builtin_amdgcn_rcpf(den) * mixv[cb][e];
                    v0 = s0 * w; v1 = s1 * w; v2 = s2 * w; v3 = s3 * w;
                } else if (a.mask_mode == 1) {
                    const float den = (((p0 + p1) + p2) + p3) + eps_r;
                    const float w = __builtin_amdgcn_rcpf(den) * mixv[cb][e];
                    v0 = p0 * w; v1 = p1 * w; v2 = p2 * w; v3 = p3 * w;
                } else {
                    v0 = p0; v1 = p1; v2 = p2; v3 = p3;
                }
                if (m == 0) {
                    if (valid) { res[cb][0][e] = v0; res[cb][1][e] = v1; res[cb][2][e] = v2; res[cb][3][e] = v3; }
                } else if (valid) {
                    res[cb][0][e] = down * res[cb][0][e] + up * v0;
                    res[cb][1][e] = down * res[cb][1][e] + up * v1;
                    res[cb][2][e] = down * res[cb][2][e] + up * v2;
                    res[cb][3][e] = down * res[cb][3][e] + up * v3;
                }
            }
        }
    }
#undef DCS_LOAD_A
#undef DCS_STORE_A
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) {
        const int col = col0 + cb * 16 + fi;
        if (col < a.F) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t r = row0 + kq * 4 + e;
                if (r < a.rows) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a.out[c * a.out_src_stride + r * a.out_ld + col] = res[cb][c][e];
                }
            }
        }
    }
}

}  // namespace

int dcs_launch_dsd_deconv2(dcs_ctx* ctx, const float* D, const float* Bw, float* G, int64_t n_ks, int H2, int CP,
                           int CI, int kh, int tc, int NG, int GS, int gcols) {
    if (n_ks <= 0) return DCS_OK;
    const int nrb = (H2 + 15) / 16;
    if (nrb > 2) DCS_FAIL(DCS_EUNSUPPORTED, "deconv2: conv2 output height %d > 32", H2);
    if (CP != 52) DCS_FAIL(DCS_EUNSUPPORTED, "deconv2: built for 50 conv2 filters (CP=52), got CP=%d", CP);
    const size_t lds = (size_t)16 * nrb * ((CP + 2) + gcols) * sizeof(float);
    if (lds > 64 * 1024) DCS_FAIL(DCS_EUNSUPPORTED, "deconv2: LDS %zu too large", lds);
    DcsTimer tm(ctx, DCS_TAG_DECONV2);
    static const int force = getenv("DCS_DECONV2") ? atoi(getenv("DCS_DECONV2")) : 0;  // 1 one-shot, 2 persistent
    const int per_group = 2 * ctx->n_cu / NG;                       // persistent workgroups per channel group
    const bool persistent = force ? force == 2 : (nrb == 1 && n_ks >= 8 * (int64_t)per_group);
    if (persistent) {
        const size_t lds2 = ((size_t)CP * gcols + 16 * (CP + 2) + (size_t)16 * gcols) * sizeof(float);
        auto kern = deconv2_persistent_kernel<13>;
        if (lds2 > 48 * 1024)
            DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds2));
        hipLaunchKernelGGL(kern, dim3((unsigned)per_group, (unsigned)NG), dim3(kThreads), lds2, ctx->stream, D, Bw, G,
                           n_ks, H2, CP, CI, kh, tc, GS, gcols);
    } else {
        hipLaunchKernelGGL(deconv2_kernel<13>, dim3((unsigned)n_ks, (unsigned)NG), dim3(kThreads), lds, ctx->stream, D,
                           Bw, G, H2, CP, CI, kh, tc, GS, gcols);
    }
    tm.done();
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

int dcs_launch_dsd_final(dcs_ctx* ctx, const DsdFinalArgs& a_in, bool fold) {
    if (a_in.rows <= 0) return DCS_OK;
    if (a_in.CI != 52) DCS_FAIL(DCS_EUNSUPPORTED, "final: built for 50 conv1 filters (CI=52), got CI=%d", a_in.CI);
    // DCS_FINAL_KERNEL=2 selects the barrier-free wave-tile variant (A/B experiments)
    static const int force = getenv("DCS_FINAL_KERNEL") ? atoi(getenv("DCS_FINAL_KERNEL")) : 0;  // 1 coop, 2 wave
    static const int prio = getenv("DCS_FINAL_PRIO") ? atoi(getenv("DCS_FINAL_PRIO")) : 0;
    DsdFinalArgs a = a_in;
    a.prio = prio;
    dim3 grid((unsigned)(dcs_cdiv(a.rows, 16) * (a.ldb / 64)));
    DcsTimer tm(ctx, DCS_TAG_FINAL);
    constexpr int kCBW = 2;  // column blocks per wave: 3 needs > 256 registers with the prefetch set
    const int n_colg = (a.F + 16 * kCBW - 1) / (16 * kCBW);
    const int64_t n_waves = (int64_t)dcs_cdiv(a.rows, 16) * n_colg;
    // measured at 4096 tiles (MI355X): cooperative 0.609 ms, wave-tile 0.649 ms -> cooperative is the default
    const bool wave_variant = force == 2 && n_colg * 16 * kCBW <= a.ldb;
    if (wave_variant) {
        if (fold)
            hipLaunchKernelGGL((final_wave_kernel<true, kCBW>), dim3((unsigned)n_waves), dim3(64), 0, ctx->stream, a, n_colg);
        else
            hipLaunchKernelGGL((final_wave_kernel<false, kCBW>), dim3((unsigned)n_waves), dim3(64), 0, ctx->stream, a, n_colg);
    } else if (fold)
        hipLaunchKernelGGL((final_kernel<true, 13>), grid, dim3(kThreads), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((final_kernel<false, 13>), grid, dim3(kThreads), 0, ctx->stream, a);
    tm.done();
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}
