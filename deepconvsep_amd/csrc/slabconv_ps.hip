// conv2 of the iKala graph (10 x 20 filters, 30 -> 30 channels, separate_ikala.py:181-183) and its InverseLayer on the
// 16-bit matrix pipe with the SLAB ALREADY SPLIT in LDS.
//
//   out[co][y][x] = bias[co] + sum_{u,v,ci} W[u][v][co][ci] * in[ci][y + u - ph][x + v - pw]      (zero outside the input)
//
// slabconv_mx_kernel (generic.hip) keeps the input slab in LDS as f32 and splits a lane's eight channels into three bf16
// terms every time a tap reads them: ~50 VALU instructions beside 12 MFMAs, 200 times per slab element, and the f32 slab
// (144 bytes per position) leaves conv2 one output row per workgroup -- 4 of 16 waves busy.  Here
//   * the slab holds 16 channels at a time as bf16 planes, record (row, x) = [plane][16 ci] = 96 bytes, written once per
//     element when the slab is filled.  No padding: with the lane groups a ds_read_b128 is really served in
//     ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md, LDS) a stride of 6 (or 2) 16-byte units puts the 16 lanes of every
//     group on 16 different bank quads, a padded stride of 7 does not (measured: 48 % of the LDS cycles were conflicts);
//     the kernel makes two passes (channels 0-15, 16-31) over the same accumulators;
//   * one MFMA K block (32) = TWO taps (v, v + 1) x 16 channels: lane (x = fi, kg) reads the 16 bytes of channels
//     8 (kg & 1) .. + 7 of column x + v + (kg >> 1) -- a plain ds_read_b128 per plane, no VALU on the operand;
//   * weights are packed on the host in the same k order and in FRAGMENT order, [half][u][tap pair][plane][co half][64
//     lanes] pieces, and streamed through a double-buffered LDS stage by the workgroup: a fragment read is 1 KB of
//     consecutive LDS, conflict-free without padding.
// A (tap pair, 16 x 16 block) step is 12 MFMAs (bf16 x 3: six products x two channel halves of the output) against
// three operand reads + six weight reads shared by the wave's blocks.  MODE 1 (f16 switch): one plane, 2 MFMAs.
#include <string.h>

#include "dcs_internal.h"
#include "generic.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ unsigned bf_trunc(float x) { return __float_as_uint(x) & 0xffff0000u; }

template <int MODE>
__device__ __forceinline__ void convert8(const float (&x)[8], u32x4 (&out)[MODE == 0 ? 3 : 1]) {
    if constexpr (MODE == 0) {
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = bf_trunc(x[j]);
            const float r1 = x[j] - __uint_as_float(h[j]);
            m[j] = bf_trunc(r1);
            l[j] = bf_trunc(r1 - __uint_as_float(m[j]));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            out[0][q] = (h[2 * q] >> 16) | h[2 * q + 1];
            out[1][q] = (m[2 * q] >> 16) | m[2 * q + 1];
            out[2][q] = (l[2 * q] >> 16) | l[2 * q + 1];
        }
    } else {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)x[j];
        out[0] = __builtin_bit_cast(u32x4, v);
    }
}

template <int MODE>
__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (MODE == 0)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// PU = false: a K block pairs the taps (u, v) and (u, v + 1) (filters wider than one column: iKala);
// PU = true:  kw == 1 (Bach10 / score-informed in f32): a K block pairs the taps u = 2 up and 2 up + 1 of the one column,
//             and a workgroup owns a tile of output columns (a 505-column row does not fit in LDS).
//
// FAST (wide filters only): the bounds arithmetic of the tap loop is done once per block -- a bit per tap pair for "some
// lane of the block reads inside the image" and one for "every lane does", the live filter rows as a range -- and once
// per weight stage (one address register per block: it advances by a constant per tap pair).
// Lanes of an edge block that fall outside the image read a record of zeros kept in front of the slab: one select on the
// address instead of twelve on the data.  The loop it replaces (FAST = false, DCS_SLABCONV_PS_FAST=0) spent 33 scalar
// and 18 vector instructions per twelve MFMAs on that arithmetic (PMC, profiles/r02_e_pmc_legs.txt).  Same products in
// the same order: the results are bit-identical.
template <int MODE, int NW /* waves per workgroup */, bool PU, bool FAST>
__global__ __launch_bounds__(64 * NW) void slabconv_ps_kernel(const DcsSlabConv g, const u32x4* __restrict__ Wq) {
    static_assert(!(PU && FAST), "the mask-driven tap loop covers filters wider than one column");
    constexpr int NP = MODE == 0 ? 3 : 1;
    constexpr int RP = NP * 2;                            // 16-byte pieces per slab record
    constexpr int NTH = 64 * NW, NBW = 32 / NW;           // threads; blocks per wave
    constexpr int kStage = NP * 128;                      // pieces per tap pair: [plane][co half][64 lanes]
    constexpr int kStageGlb = kStage;
    extern __shared__ u32x4 smem[];
    u32x4* Wl = smem;                                     // [2][pairs_per_stage][NP][2][64]
    const int wstage = g.pstage * kStage;
    u32x4* zrec = smem + 2 * wstage;                      // one record of zeros (FAST: what lanes outside the image read)
    u32x4* slab = zrec + RP;                              // [rows_max][columns of the tile][RP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    // workgroup = (image, band of output rows, tile of output columns)
    // (wide filters keep whole rows: n_xt == 1, and the tile arithmetic below folds to constants)
    const int xtile = blockIdx.x % g.n_xt;                // (n_xt == 1: whole rows)
    const int bb = blockIdx.x / g.n_xt;
    const int64_t img = bb / g.n_bands;
    const int y0 = (int)(bb - img * g.n_bands) * g.band;
    const int yb = y0 + g.band < g.Ho ? y0 + g.band : g.Ho;   // output rows [y0, yb)
    const int xt0 = xtile * g.xt;
    const int xt1 = xt0 + g.xt < g.Wo ? xt0 + g.xt : g.Wo;   // output columns [xt0, xt1)
    const float* in = g.in + img * g.in_n_stride;
    float* out = g.out + img * g.out_n_stride;
    int rbase = y0 - g.ph, rtop = yb - 1 - g.ph + g.kh - 1;
    if (rbase < 0) rbase = 0;
    if (rtop > g.H - 1) rtop = g.H - 1;
    const int rows = rtop - rbase + 1;
    const int HW = g.H * g.W;
    const int nxb = (xt1 - xt0 + 15) >> 4, nblk = (yb - y0) * nxb;
    const int nvp = PU ? 1 : (g.kw + 1) >> 1;             // tap pairs per filter row
    // input columns the tile can touch: [cx0, cx1) -- a wide filter's tap pairs reach 2 nvp - 1 columns past the block
    int cx0 = xt0 - g.pw;
    int cx1 = xt0 - g.pw + nxb * 16 + (PU ? 0 : 2 * nvp - 1);
    if (cx0 < 0) cx0 = 0;
    if (cx1 > g.W) cx1 = g.W;
    const int SW = cx1 > cx0 ? cx1 - cx0 : 0;
    int by[NBW], bx[NBW];
    f32x4 acc0[NBW], acc1[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int b = wave + NW * i;
        by[i] = b < nblk ? y0 + b / nxb : -1;
        bx[i] = b < nblk ? xt0 + (b % nxb) * 16 : 0;
        acc0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // FAST: per block -- tap pairs with a lane inside the image (lm) / with every lane inside (im), the live filter rows
    // [bu0, bu1], the input column of lane 0 at v = 0 (bxs) and the lane's slab index at (u, v) = (0, 0) (vb)
    unsigned lm[NBW], im[NBW];
    int bu0[NBW], bu1[NBW], bxs[NBW], vb[NBW];
    const int lx = fi + (kq >> 1);
    if constexpr (FAST) {
        if (tid < RP) zrec[tid] = u32x4{0u, 0u, 0u, 0u};   // ordered before its first read by the barriers of the fill
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            lm[i] = 0u; im[i] = 0u; bu0[i] = 1; bu1[i] = 0; bxs[i] = 0; vb[i] = 0;
            if (by[i] < 0) continue;
            const int xs0 = bx[i] - g.pw;
            for (int vp = 0; vp < nvp; ++vp) {
                const int xs = xs0 + 2 * vp;
                if (xs + 16 >= 0 && xs < g.W) lm[i] |= 1u << vp;
                if (xs >= 0 && xs + 17 <= g.W) im[i] |= 1u << vp;
            }
            bu0[i] = g.ph - by[i];
            bu1[i] = g.H - 1 + g.ph - by[i];
            bxs[i] = xs0;
            vb[i] = ((by[i] - g.ph - rbase) * SW + xs0 - cx0 + lx) * RP + (kq & 1);
        }
    }
    int u_lo = g.ph - (yb - 1), u_hi = g.ph - y0 + g.H - 1;
    if (u_lo < 0) u_lo = 0;
    if (u_hi > g.kh - 1) u_hi = g.kh - 1;
    // pairs of this band as (outer, inner): PU: one outer step, inner = up in [u_lo / 2, u_hi / 2];
    // otherwise outer = u in [u_lo, u_hi], inner = vp in [0, nvp)
    const int in_lo = PU ? u_lo >> 1 : 0;
    const int n_in = PU ? (u_hi >> 1) - in_lo + 1 : nvp;
    const int n_out = PU ? (u_hi >= u_lo ? 1 : 0) : u_hi - u_lo + 1;
    const int nvs = (n_in + g.pstage - 1) / g.pstage;     // stages per outer step
    const int n_stage = n_out * nvs;
    const int pairs_per_half = PU ? (g.kh + 1) >> 1 : g.kh * nvp;
    constexpr int kMaxPairs = 5;                          // tap pairs per weight stage (launcher: as many as fit)
    constexpr int WPRE = (kMaxPairs * kStageGlb + NTH - 1) / NTH;
    u32x4 wpre[WPRE];
    for (int hc = 0; hc < 2; ++hc) {
        __syncthreads();                                  // every wave is done with the previous half's slab and weights
        // slab fill: task = (row, x, channel octet); consecutive threads take consecutive x of one (octet, row)
        const int n_task = 2 * rows * SW;
        for (int i = tid; i < n_task; i += NTH) {
            const int o = i / (rows * SW), rem = i - o * (rows * SW);
            const int r = rem / SW, x = rem - r * SW;
            const int c0 = 16 * hc + 8 * o;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c0 + j < g.Cin ? in[(int64_t)(c0 + j) * HW + (rbase + r) * g.W + cx0 + x] : 0.f;
            u32x4 p[NP];
            convert8<MODE>(v, p);
#pragma unroll
            for (int q = 0; q < NP; ++q) slab[rem * RP + q * 2 + o] = p[q];
        }
        const u32x4* Wh = Wq + (int64_t)hc * pairs_per_half * kStageGlb;
#define DCS_PS_WFETCH(st_)                                                                              \
        {                                                                                               \
            const int o_ = (st_) / nvs, i0_ = ((st_) - o_ * nvs) * g.pstage;                            \
            const int np_ = i0_ + g.pstage <= n_in ? g.pstage : n_in - i0_;                             \
            const int pair_ = PU ? in_lo + i0_ : (u_lo + o_) * nvp + i0_;                               \
            _Pragma("unroll") for (int q = 0; q < WPRE; ++q) {                                          \
                const int e = tid + q * NTH;                                                            \
                wpre[q] = e < np_ * kStageGlb ? Wh[(int64_t)pair_ * kStageGlb + e] : u32x4{0u, 0u, 0u, 0u};    \
            }                                                                                           \
        }
        // Stages none of the workgroup's blocks can use are skipped by the whole workgroup -- no fetch, no barrier.  With whole
        // rows per workgroup every stage has a live block somewhere; with a COLUMN tile (round 6: the transposed convolution of
        // the iKala graph, kw - 1 = 19 columns of padding) the tile's blocks share their column range and so their live tap
        // pairs: the column tile at the right edge of an 83-wide row uses 2 of the 10 pairs of a filter row.
        auto stage_live = [&](int st_) -> bool {
            if (PU || !FAST) return true;
            const int i0_ = (st_ % nvs) * g.pstage;
            const int np_ = i0_ + g.pstage <= n_in ? g.pstage : n_in - i0_;
            const int min_xs = xt0 - g.pw + 2 * i0_, max_xs = xt0 + (nxb - 1) * 16 - g.pw + 2 * (i0_ + np_ - 1);
            return max_xs + 16 >= 0 && min_xs < g.W;
        };
        int st = 0, it = 0;
        while (st < n_stage && !stage_live(st)) ++st;
        if (st < n_stage) DCS_PS_WFETCH(st)
        for (; st < n_stage; ++it) {
            u32x4* Wb = Wl + (it & 1) * wstage;
#pragma unroll
            for (int q = 0; q < WPRE; ++q) {
                const int e = tid + q * NTH;              // the packed order is the LDS order
                if (e < g.pstage * kStageGlb) Wb[e] = wpre[q];
            }
            __syncthreads();     // also orders the slab fill before its first use; buffer it & 1 was last read two stages ago
            int st_next = st + 1;
            while (st_next < n_stage && !stage_live(st_next)) ++st_next;
            if (st_next < n_stage) DCS_PS_WFETCH(st_next)
            const int so = st / nvs, i0 = (st - so * nvs) * g.pstage;
            const int np = i0 + g.pstage <= n_in ? g.pstage : n_in - i0;
#define DCS_PS_MMA(i_, a0, a1, b)                                                                                \
            if constexpr (MODE == 0) { /* smallest terms first */                                       \
                acc0[i_] = mma<0>(a0[2], b[0], acc0[i_]);                                               \
                acc1[i_] = mma<0>(a1[2], b[0], acc1[i_]);                                               \
                acc0[i_] = mma<0>(a0[0], b[2], acc0[i_]);                                               \
                acc1[i_] = mma<0>(a1[0], b[2], acc1[i_]);                                               \
                acc0[i_] = mma<0>(a0[1], b[1], acc0[i_]);                                               \
                acc1[i_] = mma<0>(a1[1], b[1], acc1[i_]);                                               \
                acc0[i_] = mma<0>(a0[1], b[0], acc0[i_]);                                               \
                acc1[i_] = mma<0>(a1[1], b[0], acc1[i_]);                                               \
                acc0[i_] = mma<0>(a0[0], b[1], acc0[i_]);                                               \
                acc1[i_] = mma<0>(a1[0], b[1], acc1[i_]);                                               \
                acc0[i_] = mma<0>(a0[0], b[0], acc0[i_]);                                               \
                acc1[i_] = mma<0>(a1[0], b[0], acc1[i_]);                                               \
            } else {                                                                                    \
                acc0[i_] = mma<1>(a0[0], b[0], acc0[i_]);                                               \
                acc1[i_] = mma<1>(a1[0], b[0], acc1[i_]);                                               \
            }
            if constexpr (FAST) {
                const int u = u_lo + so;
                unsigned m[NBW], ne[NBW], many = 0u;
                int va[NBW];
#pragma unroll
                for (int i = 0; i < NBW; ++i) {
                    const bool rl = u >= bu0[i] && u <= bu1[i];          // (a slot without a block has an empty range)
                    m[i] = rl ? (lm[i] >> i0) & ((1u << np) - 1u) : 0u;
                    ne[i] = ~im[i] >> i0;
                    va[i] = vb[i] + (u * SW + 2 * i0) * RP;
                    many |= m[i];
                }
                const u32x4* wp = Wb + lane;
                // (not unrolled: with the five pairs of a stage unrolled the accumulators left every conditional block
                // in other registers -- fourteen copies per step)
#pragma nounroll
                for (int tp = 0; tp < np; ++tp, wp += kStage) {
                    if ((many >> tp) & 1u) {
                        u32x4 a0[NP], a1[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            a0[p] = wp[(p * 2) * 64];
                            a1[p] = wp[(p * 2 + 1) * 64];
                        }
#pragma unroll
                        for (int i = 0; i < NBW; ++i) {
                            if (!((m[i] >> tp) & 1u)) continue;
                            int idx = va[i];
                            if ((ne[i] >> tp) & 1u) {                    // an edge block: lanes outside read the zeros
                                const int xc = bxs[i] + 2 * (i0 + tp) + lx;
                                idx = (unsigned)xc < (unsigned)g.W ? idx : (kq & 1) - RP;
                            }
                            u32x4 b[NP];
#pragma unroll
                            for (int p = 0; p < NP; ++p) b[p] = slab[idx + p * 2];
                            DCS_PS_MMA(i, a0, a1, b)
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NBW; ++i) va[i] += 2 * RP;
                }
            } else
            for (int tp = 0; tp < np; ++tp) {
                const int u = PU ? 2 * (in_lo + i0 + tp) : u_lo + so;      // first tap of the pair
                const int v = PU ? 0 : 2 * (i0 + tp);
                const u32x4* wp = Wb + tp * kStage + lane;
                u32x4 a0[NP], a1[NP];
                bool have = false;
#pragma unroll
                for (int i = 0; i < NBW; ++i) {
                    if (by[i] < 0) continue;
                    const int r = by[i] + u - g.ph;                  // input row of the first tap (uniform per block)
                    const int xs = bx[i] + v - g.pw;                 // input column of lane 0, first tap
                    bool inner;                                      // every row / column of both taps inside: no select
                    if (PU) {
                        if (r + 1 < 0 || r >= g.H || xs + 15 < 0 || xs >= g.W) continue;
                        inner = r >= 0 && r + 1 < g.H && u + 1 < g.kh && xs >= 0 && xs + 16 <= g.W;
                    } else {
                        if (r < 0 || r >= g.H || xs + 16 < 0 || xs >= g.W) continue;
                        inner = xs >= 0 && xs + 17 <= g.W;
                    }
                    if (!have) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            a0[p] = wp[(p * 2) * 64];
                            a1[p] = wp[(p * 2 + 1) * 64];
                        }
                        have = true;
                    }
                    const int rr = PU ? r + (kq >> 1) : r;
                    const int xc = PU ? xs + fi : xs + fi + (kq >> 1);
                    // (a second tap past the filter has zero weights, but its row may lie outside the slab: no read)
                    const bool ok = xc >= 0 && xc < g.W && rr >= 0 && rr < g.H && (!PU || u + (kq >> 1) < g.kh);
                    const u32x4* sp = slab + (ok ? (rr - rbase) * SW + xc - cx0 : 0) * RP + (kq & 1);
                    u32x4 b[NP];
                    if (inner) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) b[p] = sp[p * 2];
                    } else {
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const u32x4 t = sp[p * 2];
                            b[p] = ok ? t : u32x4{0u, 0u, 0u, 0u};
                        }
                    }
                    DCS_PS_MMA(i, a0, a1, b)
                }
            }
#undef DCS_PS_MMA
            st = st_next;
        }
#undef DCS_PS_WFETCH
    }
    const int HoWo = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        if (by[i] < 0 || bx[i] + fi >= g.Wo) continue;
        float* op = out + (int64_t)by[i] * g.Wo + bx[i] + fi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = 4 * kq + e;
            if (co < g.Cout) op[(int64_t)co * HoWo] = acc0[i][e] + g.bias[co];
            if (co + 16 < g.Cout) op[(int64_t)(co + 16) * HoWo] = acc1[i][e] + g.bias[co + 16];
        }
    }
}

}  // namespace

// Wf: [kh * kw taps][1024] f32 with element (ci, co) of a tap at wslot(ci, co) -> the kernel's order
// [half hc][pair][plane][co half][lane = 16 kg + (co & 15)][8], channel of k slot (kg, j) = 16 hc + 8 (kg & 1) + j and
//   kw > 1:  pair = u * ceil(kw / 2) + vp, tap (u, 2 vp + (kg >> 1));     kw == 1:  pair = up, tap (2 up + (kg >> 1), 0).
// mode 0: three bf16 planes (truncation split, exact); mode 1: one f16 plane (round to nearest even).
void dcs_slabconv_ps_pack(const float* Wf, int kh, int kw, int (*wslot)(int, int), int mode, std::vector<uint16_t>* out) {
    const bool pu = kw == 1;
    const int np = mode == 0 ? 3 : 1, nvp = (kw + 1) / 2;
    const int pairs = pu ? (kh + 1) / 2 : kh * nvp;
    out->assign((size_t)2 * pairs * np * 128 * 8, 0);
    for (int hc = 0; hc < 2; ++hc)
        for (int pr = 0; pr < pairs; ++pr)
            for (int co = 0; co < 32; ++co)
                for (int kg = 0; kg < 4; ++kg)
                    for (int j = 0; j < 8; ++j) {
                        const int u = pu ? 2 * pr + (kg >> 1) : pr / nvp;
                        const int v = pu ? 0 : 2 * (pr % nvp) + (kg >> 1);
                        const int ci = 16 * hc + 8 * (kg & 1) + j;
                        float r = (u < kh && v < kw) ? Wf[(size_t)(u * kw + v) * 1024 + wslot(ci, co)] : 0.f;
                        const size_t base = (((size_t)hc * pairs + pr) * np) * 128 * 8;
                        const size_t idx = (((size_t)(co >> 4) * 64) + kg * 16 + (co & 15)) * 8 + j;
                        if (mode == 0) {
                            for (int p = 0; p < 3; ++p) {
                                uint32_t bits;
                                memcpy(&bits, &r, 4);
                                bits &= 0xffff0000u;
                                float part;
                                memcpy(&part, &bits, 4);
                                r -= part;
                                (*out)[base + (size_t)p * 128 * 8 + idx] = (uint16_t)(bits >> 16);
                            }
                        } else {
                            const _Float16 hv = (_Float16)r;
                            uint16_t hb;
                            memcpy(&hb, &hv, 2);
                            (*out)[base + idx] = hb;
                        }
                    }
}

// false: the shape does not fit; nothing launched
bool dcs_launch_slabconv_ps(dcs_ctx* ctx, DcsSlabConv a, int64_t n_images, const void* Wq, int mode) {
    static const bool on = !(getenv("DCS_SLABCONV_PS") && atoi(getenv("DCS_SLABCONV_PS")) == 0);
    if (!on || !Wq || a.Cin > 32 || a.Cout > 32) return false;
    const bool pu = a.kw == 1;
    const int np = mode == 0 ? 3 : 1;
    const int nvp = (a.kw + 1) / 2;
    const int nxb_all = (a.Wo + 15) / 16;
    const size_t rec = (size_t)np * 2 * 16;
    // (rows, columns) of outputs per workgroup: a workgroup has 16 waves x 2 block slots and streams all the weights
    // whatever its share, so the best split is the one that wastes the fewest (wave, slot) pairs -- counting short last
    // bands / tiles and CUs left without a workgroup -- among those whose slab fits, discounted by the number of weight
    // stages (barriers) the leftover LDS forces; ties go to the larger weight stage, then to the larger share.  Wide filters keep whole rows (the halo of a column tile would be kw - 1 columns).
    // Measured dead ends of round 4 (profiles/r04_m_ikala_*): (i) 8 waves x 4 blocks, or 16 x 4 on ten-row bands, halve the
    // LDS reads of the weight fragments per MFMA and are 10 - 25 % SLOWER; (ii) requesting both blocks' input fragments and
    // the next pair's weight fragments ahead of the MFMAs (software pipeline inside the wave) is 12 % slower.  Counters of
    // the kernel as it is: LDS 31 % busy, 3 % of the wave-cycles wait for LDS, 47 % wait at s_barrier / s_waitcnt -- the
    // sixteen waves of a stage have between 2 and 10 live (pair, block) steps (column blocks 0 and 5 of an 83-wide row see
    // 8 and 2 of the 10 pairs) and the stage ends when the slowest is done.
    constexpr bool fast_on = true;
    const bool fast = fast_on && !pu && nvp <= 32;         // a bit per tap pair of a filter row
    const int nw = 16, nbw = 2;
    const int slots = nw * nbw;
    int band = 0, xt = 0, ps_best = 1;
    size_t lds = 0;
    double best = 0.0;
    for (int cxt = pu ? 1 : nxb_all; cxt <= nxb_all && cxt <= slots; ++cxt) {     // 16-column blocks per tile
        const int n_xt = (nxb_all + cxt - 1) / cxt;
        int sw = cxt * 16 + (pu ? 0 : 2 * nvp - 1);
        if (sw > a.W) sw = a.W;
        for (int cand = 1; cand <= a.Ho && cand * cxt <= slots; ++cand) {
            int rows = cand + a.kh - 1;
            if (rows > a.H) rows = a.H;
            // weight stage: as many tap pairs as fit beside the slab (a barrier per stage: the fewer the better), at most 5
            int ps = 5;
            size_t need = 0;
            for (;; --ps) {
                if (ps < 1) break;
                need = (size_t)2 * ps * np * 128 * 16 + (size_t)(rows * sw + 1) * rec;   // + the record of zeros
                if (need <= 160 * 1024) break;
            }
            if (ps < 1) break;
            const int64_t n_wg = n_images * ((a.Ho + cand - 1) / cand) * n_xt;
            const int rounds = (cand * cxt + nw - 1) / nw;            // block slots used per wave
            double eff = (double)n_images * a.Ho * nxb_all / ((double)n_wg * nw * rounds);
            if (n_wg < ctx->n_cu) eff *= (double)n_wg / ctx->n_cu;
            eff /= 1.0 + 0.6 / ps;                                    // a barrier per stage: measured, iKala 0.83 -> 0.67 ms
            const bool tie = eff > best - 1e-9;
            if (eff > best + 1e-9 || (tie && (ps > ps_best || (ps == ps_best && cand * cxt >= band * (xt / 16))))) {
                best = eff; band = cand; xt = cxt * 16; lds = need; ps_best = ps;
            }
        }
    }
    // Wide filters WITH padding (the transposed convolution of the iKala graph: 10 x 20, 9 / 19 of padding): column strips
    // instead of row bands.  With whole rows per workgroup the sixteen waves of a weight stage have between 2 and 10 live (tap
    // pair, block) steps -- which tap pairs reach inside the image depends on the block's COLUMN -- and the stage ends with the
    // slowest: 47 % of the wave-cycles waited at the stage barrier (PMC, profiles/r04_m_ikala_pmc.txt).  A workgroup that owns
    // ONE column block of every output row has the same live tap pairs in all its blocks: the stages balance, and the stages
    // no block of the strip can use are skipped by the whole workgroup (slabconv_ps_kernel: stage_live).
    // (same-box A/B: conv2^T 0.673 -> 0.627 ms per 10 s clip, profiles/r06_g_ikala_column_strips_ab.txt)
    if (fast && a.pw > 0 && a.Ho <= slots && nxb_all >= 2) {
        int rows = a.Ho + a.kh - 1;
        if (rows > a.H) rows = a.H;
        int sw = 16 + 2 * nvp - 1;
        if (sw > a.W) sw = a.W;
        for (int ps = 5; ps >= 1; --ps) {
            const size_t need = (size_t)2 * ps * np * 128 * 16 + (size_t)(rows * sw + 1) * rec;
            if (need <= 160 * 1024) {
                band = a.Ho; xt = 16; lds = need; ps_best = ps;
                break;
            }
        }
    }
    if (band < 1 || band * (xt / 16) < 8) return false;    // fewer than 8 blocks: the f32-slab kernels are no worse
    a.pstage = ps_best;
    a.band = band;
    a.n_bands = (a.Ho + band - 1) / band;
    a.rows_max = band + a.kh - 1;
    a.xt = xt;
    a.n_xt = (nxb_all * 16 + xt - 1) / xt;
    typedef void (*kern_t)(const DcsSlabConv, const u32x4*);
    kern_t kern;
    if (pu)
        kern = mode == 0 ? (kern_t)slabconv_ps_kernel<0, 16, true, false> : (kern_t)slabconv_ps_kernel<1, 16, true, false>;
    else if (!fast)
        kern = mode == 0 ? (kern_t)slabconv_ps_kernel<0, 16, false, false> : (kern_t)slabconv_ps_kernel<1, 16, false, false>;
    else
        kern = mode == 0 ? (kern_t)slabconv_ps_kernel<0, 16, false, true> : (kern_t)slabconv_ps_kernel<1, 16, false, true>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_images * a.n_bands * a.n_xt)), dim3(64 * nw), lds, ctx->stream, a,
                       reinterpret_cast<const u32x4*>(Wq));
    return true;
}
