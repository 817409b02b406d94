// Column convolution (kh x 1 filters: conv2 of the Bach10 / score-informed graphs, separate_bach10.py:196-200, and its
// InverseLayer) with f16 operands, f32 accumulation and the WEIGHTS IN REGISTERS.
//
//   out[co][y][x] = bias[co] + sum_u sum_ci W[u][co][ci] * in[ci][y + u - ph][x]        (rows outside 0 <= . < H are zero)
//
// Every column x is an independent 1-D problem along y.  colconv_f16_kernel (generic.hip) keeps slab and weights in
// LDS and reads three 16-byte operands per pair of MFMAs: with four SIMDs sharing 128 B/clk the LDS pipe is asked for
// three times what it has, and the 512-thread workgroup pays two barriers per 16 columns.  Here a WAVE owns 16
// columns of one image and nothing is shared, so there is no LDS and no barrier:
//   * all kh x 2 weight fragments (B operand, 32 in-channels x 16 out-channels each) stay in 8 kh VGPRs for the whole
//     launch (160 for kh = 20); the kernel runs one wave per SIMD with the 512-VGPR budget;
//   * the slab goes global -> registers directly as the A operand (rows = x): lane (x = lane & 15, kg = lane >> 4) loads
//     in[ci = 8 kg + j][row][x0 + x], 16 consecutive x per channel (64-byte segments), rounds to f16 in registers;
//   * the accumulator then holds 4 consecutive x of one output channel per lane: one 16-byte store per (row, 16 channels).
// Two forms, both fully unrolled so that every register index is static:
//   gather  (transpose: H = 11 input rows, 30 output rows): all input rows resident as fragments, output rows two at
//           a time (four independent MFMA chains); the next unit's rows are requested before the first MFMA;
//   scatter (forward: 30 input rows, 11 output rows): all output rows accumulate in registers while the input rows
//           stream through a ring of kAhead rows in flight.
// MFMA work: 220 (row, tap) pairs x 2 per unit either way = 440 MFMAs of 16 cycles; 668 images x 32 units over 1024
// SIMDs is 21 units per wave, 0.06 ms at the matrix pipe's rate.  Measured on the Bach10 leg: the LDS kernel 0.93
// (transpose) / 0.27 ms (forward), these two 0.73 / 0.15 ms -- the transpose cannot ISSUE its 1.2 GB of 4-byte-aligned
// 16-byte stores any faster (profiles/r02_e_pmc_legs.txt), which is why the default for the single-channel graph is the
// fused kernel at the end of this file; the gather form remains for the 4-channel graph with the f16 switch on.
#include <string.h>

#include "dcs_internal.h"
#include "generic.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // 4-byte aligned: rows are W floats apart, W odd
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;

__device__ __forceinline__ f32x4 mma(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ h8 as_h8(u32x4 v) {
    union { u32x4 u; h8 h; } x;
    x.u = v;
    return x.h;
}

__device__ __forceinline__ h8 round8(const float (&r)[8]) {
    h8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)r[j];
    return v;
}

// Addresses are (uniform pointer)[32-bit lane index] throughout: one SGPR base per row + one VGPR offset per channel.
// Nothing carries a column predicate: the last column block of an image starts at W - 16 and overlaps its neighbour
// (the shared columns are computed twice, to the same bits, and stored twice).  A lane of a padded channel reads the
// last real channel instead and meets zero weights.
struct LaneIn {
    int idx[8];
};
__device__ __forceinline__ int block_x(int block, int W) { return block * 16 + 16 <= W ? block * 16 : W - 16; }

__device__ __forceinline__ LaneIn lane_in(const DcsColConv& g, int kg, int xc, int HW) {
    LaneIn li;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = kg * 8 + j < g.Cin ? kg * 8 + j : g.Cin - 1;
        li.idx[j] = c * HW + xc;
    }
    return li;
}

// ---- gather form: out row y takes input rows r = y + u - PH, PH = KH - 1 (InverseLayer of a 'valid' convolution)
template <int KH, int H>
__global__ __launch_bounds__(kThreads) void colconv_wreg_gather_kernel(const DcsColConv g, const u32x4* __restrict__ Wq,
                                                                       int64_t n_units) {
    constexpr int HO = H + KH - 1, PH = KH - 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = lane & 15, kg = lane >> 4;
    h8 w[KH][2];
#pragma unroll
    for (int u = 0; u < KH; ++u) {
        w[u][0] = as_h8(Wq[(u * 2) * 64 + lane]);
        w[u][1] = as_h8(Wq[(u * 2 + 1) * 64 + lane]);
    }
    const float bias0 = g.bias[fi], bias1 = g.bias[16 + fi];
    const bool c1_ok = fi + 16 < g.Cout;
    const int W = g.W, n_xb = g.n_xb;
    const int HW = H * W;
    const int out_lane = fi * HO * W + kg * 4;          // + 16 HO W for the second channel half
    // the four waves of a workgroup take four adjacent column blocks: their 64-byte row segments share cache lines
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    float raw[H][8];
#define DCS_FETCH(unit_)                                                                                \
    {                                                                                                   \
        const int64_t img_ = (unit_) / n_xb;                                                            \
        const int xb_ = block_x((int)((unit_) - img_ * n_xb), W);                                       \
        const LaneIn li_ = lane_in(g, kg, xb_ + fi, HW);                                                \
        const float* ib_ = g.in + img_ * g.in_n_stride;                                                 \
        _Pragma("unroll") for (int h = 0; h < H; ++h) {                                                 \
            const float* ir_ = ib_ + h * W;                                                             \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) raw[h][j] = ir_[li_.idx[j]];                  \
        }                                                                                               \
    }
    if (unit < n_units) DCS_FETCH(unit)
    for (; unit < n_units; unit += stride) {
        h8 a[H];
#pragma unroll
        for (int h = 0; h < H; ++h) a[h] = round8(raw[h]);
        {
            const int64_t next = unit + stride < n_units ? unit + stride : unit;     // last round: a harmless re-read
            DCS_FETCH(next)
        }
        const int64_t img = unit / n_xb;
        const int xb = block_x((int)(unit - img * n_xb), W);
        float* ob = g.out + img * g.out_n_stride + xb;
#pragma unroll
        for (int y = 0; y < HO; y += 2) {
            f32x4 acc[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < H; ++h) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int u = h - (y + t) + PH;
                    if (y + t < HO && u >= 0 && u < KH) {
                        acc[t][0] = mma(a[h], w[u < 0 || u >= KH ? 0 : u][0], acc[t][0]);
                        acc[t][1] = mma(a[h], w[u < 0 || u >= KH ? 0 : u][1], acc[t][1]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[t][0] += bias0;
                acc[t][1] += bias1;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (y + t < HO) *reinterpret_cast<f32x4u*>(ob + (y + t) * W + out_lane) = acc[t][0];
            if (c1_ok) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (y + t < HO) *reinterpret_cast<f32x4u*>(ob + (16 * HO + y + t) * W + out_lane) = acc[t][1];
            }
        }
    }
#undef DCS_FETCH
}

// ---- scatter form: 'valid' convolution, input row r feeds output rows y = r - u, HO = H - KH + 1
// OUT16 (round 6): the output map as f16 (same [channel][row][x] order, g.out_n_stride in halves) -- the bottleneck layer of the
// f16 switch multiplies f16 rows (gemm_f16_longk_kernel); half the bytes written here and read there.
// IN16 (round 6): the input as f16, channels-last with 32 channels per position (conv1_mfma_kernel<1, 2>): a lane's eight channels of
// a row are ONE 16-byte load instead of eight 4-byte loads 30 x 505 floats apart, and need no conversion.
template <int KH, int H, int AHEAD /* input rows in flight */, bool OUT16 = false, bool IN16 = false>
__global__ __launch_bounds__(kThreads) void colconv_wreg_scatter_kernel(const DcsColConv g, const u32x4* __restrict__ Wq,
                                                                        int64_t n_units) {
    constexpr int HO = H - KH + 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = lane & 15, kg = lane >> 4;
    h8 w[KH][2];
#pragma unroll
    for (int u = 0; u < KH; ++u) {
        w[u][0] = as_h8(Wq[(u * 2) * 64 + lane]);
        w[u][1] = as_h8(Wq[(u * 2 + 1) * 64 + lane]);
    }
    const float bias0 = g.bias[fi], bias1 = g.bias[16 + fi];
    const bool c1_ok = fi + 16 < g.Cout;
    const int W = g.W, n_xb = g.n_xb;
    const int HW = H * W;
    const int out_lane = fi * HO * W + kg * 4;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= n_units) return;
    // the first AHEAD rows of a unit are requested while the previous unit's last rows are multiplied
    float pre[IN16 ? 1 : AHEAD][8];
    u32x4 preh[IN16 ? AHEAD : 1];
    int64_t img = unit / n_xb;
    int xb = block_x((int)(unit - img * n_xb), W);
    LaneIn li = lane_in(g, kg, xb + fi, HW);
    const float* ib = g.in + img * g.in_n_stride;
    // IN16: this lane's eight channels of (row r, column xb + fi) are the 16 bytes at ih + r * W * 32
    const _Float16* ih = reinterpret_cast<const _Float16*>(g.in) + img * g.in_n_stride + ((xb + fi) * 32 + 8 * kg);
#pragma unroll
    for (int r = 0; r < AHEAD; ++r) {
        if constexpr (IN16) preh[r] = *reinterpret_cast<const u32x4*>(ih + (int64_t)r * W * 32);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) pre[r][j] = (ib + r * W)[li.idx[j]];
        }
    }
    for (; unit < n_units; unit += stride) {
        const int64_t next = unit + stride < n_units ? unit + stride : unit;         // last round: a harmless re-read
        const int64_t img_n = next / n_xb;
        const int xb_n = block_x((int)(next - img_n * n_xb), W);
        const LaneIn li_n = lane_in(g, kg, xb_n + fi, HW);
        const float* ib_n = g.in + img_n * g.in_n_stride;
        const _Float16* ih_n = reinterpret_cast<const _Float16*>(g.in) + img_n * g.in_n_stride + ((xb_n + fi) * 32 + 8 * kg);
        float raw[IN16 ? 1 : H][8];           // fully unrolled: only AHEAD + 1 rows are live at any point
        u32x4 rawh[IN16 ? H : 1];
#pragma unroll
        for (int r = 0; r < AHEAD; ++r) {
            if constexpr (IN16) rawh[r] = preh[r];
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[r][j] = pre[r][j];
            }
        }
        f32x4 acc[HO][2];
#pragma unroll
        for (int y = 0; y < HO; ++y) acc[y][0] = acc[y][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < H; ++r) {
            if constexpr (IN16) {
                if (r + AHEAD < H) rawh[r + AHEAD] = *reinterpret_cast<const u32x4*>(ih + (int64_t)(r + AHEAD) * W * 32);
                else preh[r + AHEAD - H] = *reinterpret_cast<const u32x4*>(ih_n + (int64_t)(r + AHEAD - H) * W * 32);
            } else if (r + AHEAD < H) {
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[r + AHEAD][j] = (ib + (r + AHEAD) * W)[li.idx[j]];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) pre[r + AHEAD - H][j] = (ib_n + (r + AHEAD - H) * W)[li_n.idx[j]];
            }
            h8 a;
            if constexpr (IN16) a = as_h8(rawh[r]);
            else a = round8(raw[r]);
#pragma unroll
            for (int y = 0; y < HO; ++y) {
                const int u = r - y;
                if (u >= 0 && u < KH) {
                    acc[y][0] = mma(a, w[u < 0 || u >= KH ? 0 : u][0], acc[y][0]);
                    acc[y][1] = mma(a, w[u < 0 || u >= KH ? 0 : u][1], acc[y][1]);
                }
            }
        }
#pragma unroll
        for (int y = 0; y < HO; ++y) {
            acc[y][0] += bias0;
            acc[y][1] += bias1;
        }
        if constexpr (OUT16) {
            typedef _Float16 h4u __attribute__((ext_vector_type(4), aligned(2)));     // rows of W (odd) halves: 2-byte alignment
            _Float16* ob = reinterpret_cast<_Float16*>(g.out) + img * g.out_n_stride + xb;
#pragma unroll
            for (int y = 0; y < HO; ++y) {
                const h4u v = {(_Float16)acc[y][0][0], (_Float16)acc[y][0][1], (_Float16)acc[y][0][2], (_Float16)acc[y][0][3]};
                *reinterpret_cast<h4u*>(ob + y * W + out_lane) = v;
            }
            if (c1_ok) {
#pragma unroll
                for (int y = 0; y < HO; ++y) {
                    const h4u v = {(_Float16)acc[y][1][0], (_Float16)acc[y][1][1], (_Float16)acc[y][1][2], (_Float16)acc[y][1][3]};
                    *reinterpret_cast<h4u*>(ob + (16 * HO + y) * W + out_lane) = v;
                }
            }
        } else {
        float* ob = g.out + img * g.out_n_stride + xb;
#pragma unroll
        for (int y = 0; y < HO; ++y) *reinterpret_cast<f32x4u*>(ob + y * W + out_lane) = acc[y][0];
        if (c1_ok) {
#pragma unroll
            for (int y = 0; y < HO; ++y) *reinterpret_cast<f32x4u*>(ob + (16 * HO + y) * W + out_lane) = acc[y][1];
        }
        }
        img = img_n; xb = xb_n; li = li_n; ib = ib_n; ih = ih_n;
    }
}

// ------------------------------------------------------------------------------------------------
// InverseLayer(conv2) + InverseLayer(conv1) of the Bach10 graph in one kernel (separate_bach10.py:219-227; f16 switch on).
//
// Unfused, the 30 x 30 x 505 activations between the two layers are 1.2 GB per 10 s clip, written and read back in
// 64-byte pieces that start on 4-byte boundaries (rows are 505 floats): both kernels ran at the rate the memory system
// absorbs such pieces (0.73 + 0.54 ms), an order of magnitude above their arithmetic.  Here they never leave the wave:
//   stage 1  the gather form above with the operands swapped (A = weights, B = slab), so that a lane ends up with
//            G[ci = 4 kq + e (+16)][t][x = fi] -- eight input channels of ONE column, which is exactly a B-operand
//            fragment of a second MFMA whose K axis is the channel (k slot j <-> ci = 4 kq + j | 16 + 4 kq + j - 4);
//   stage 2  conv1^T with stride 4 and 30 (padded 32) taps is  y[4 q + r] = sum_mm sum_ci G[ci][q - mm] W1[ci][4 mm + r]:
//            P^T[(mm, r)][x] = sum_ci W1[ci][4 mm + r] G[ci][x] is a 32 x 32 x 16 product per row (bf16 pipe, both
//            operands split three ways, six products: f32-class), after which lane (x, kq) holds the float4
//            P[x][mm = kq (+4)][r = 0..3] = a 16-byte piece of the output row at f = 4 (x + mm);
//   shift-add  the eight pieces that meet at one q come from different lanes.  Each lane writes its two pieces to a
//            per-wave LDS array P[row][mm][slot 8 + x] (zero guard slots either side); then lane q of the first (second)
//            half-wave adds the eight slots P[row 0 (1)][mm][8 + q - mm] of its q = 16 b .. 16 b + 22 -- plain
//            ds_write_b128 / ds_read_b128, no conflicts, no atomics (ds_add_f32 costs ~150 cycles per wave instruction
//            and made the first version 1.9 ms).  A wave walks consecutive column blocks of one image: q < 16 b + 16 is
//            complete after block b (its inputs are x = q - 7 .. q) and goes out as 256 contiguous bytes per row, with the
//            carry of block b - 1 added; q = 16 b + 16 .. + 22 are this block's carry (LDS, [row][8]).
// An image's blocks are cut into runs (one wave each); a run that does not start at x = 0 first recomputes the block to
// its left and keeps only the carry.  HBM traffic: the dense-layer output once (445 MB) + the 164 MB result.
// (scripts/emu_fused_decoder.py: the same index arithmetic lane by lane in NumPy against the direct formula.)
// ------------------------------------------------------------------------------------------------
struct DcsDecoderFused {
    const u32x4* Wq1;       // the padded conv1 filter as packed by dcs_decoder_fused_pack; this kernel reads section 3 (the f16 fragments)
    float* out;             // [image][HO][F]
    int F;
    int runs_per_image;
    int64_t n_runs;
};

// CL (round 4): the dense layer's output arrives CHANNELS-LAST, in[image][row][x][Cin] (its weight columns are permuted when
// the bf16 planes are packed, dcs_gemm_pack_bq), so a lane's eight channels of a (row, x) are 32 consecutive bytes -- four
// 8-byte loads, 44 per column block instead of 88 four-byte ones, and a wave's loads of a row cover ONE contiguous run of
// 16 x Cin floats.  With the channel-first layout (rows of W = 505 floats: every 64-byte segment of 16 x starts on a 4-byte
// boundary and straddles two lines) the counters showed 2.68 x the algorithmic bytes moving (profiles/r03_traffic.json).
// Stage 2 -- the transposed conv1 -- takes f16 inputs like stage 1 (round 4; the three-way split form it replaced was a switch
// until round 6): G is rounded to f16 once (it is the output of an f16-input
// convolution already) and meets the f16 conv1 filter in ONE MFMA per tap half instead of six bf16 ones behind a 52-
// instruction operand split per row -- 500 instead of 800 MFMAs and ~1 500 fewer vector instructions per column block.
// BASELINE configs[3] names an "fp16 MFMA conv path"; its stated tolerance (network output 2e-3) is unchanged, the
// f32-class result of this graph is the path with the switch off.
// IN16 (round 6): the input arrives as f16, channels-last with the channel axis padded to 32 (gemm_f16.hip writes it so under the
// f16 switch): in[image][row][x][32] halves, a lane's eight channels ONE 16-byte load and no conversion -- half the bytes of
// the f32 input it replaces, which this kernel rounded to f16 on arrival anyway.  g.in / g.in_n_stride are then in halves.
template <int KH, int H, bool CL, bool IN16 = false>
__global__ __launch_bounds__(kThreads) void colconv_deconv1_fused_kernel(const DcsColConv g, const u32x4* __restrict__ Wq,
                                                                         const DcsDecoderFused d) {
    static_assert(!IN16 || CL, "the f16 input is channels-last");
    constexpr int HO = H + KH - 1, PH = KH - 1;
    static_assert(HO % 2 == 0, "output rows are processed in pairs");
    // per wave: Pb [2 rows][8 taps mm][32 slots] float4 -- slot 8 + x holds P[x][mm], slots 0..7 and 24..31 stay zero --
    // and the carry Cb [HO][8] float4 (q = 16 b + 16 .. + 23 of the block just finished; the last one is always zero)
    constexpr int kPb = 2 * 8 * 32 * 4, kCb = HO * 8 * 4;
    __shared__ __attribute__((aligned(16))) float lds[4 * (kPb + kCb)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    f32x4* Pb = reinterpret_cast<f32x4*>(lds + wave * (kPb + kCb));
    f32x4* Cb = Pb + kPb / 4;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < kPb / 4; i += 64) Pb[i] = zero4;
    h8 w[KH][2];
#pragma unroll
    for (int u = 0; u < KH; ++u) {
        w[u][0] = as_h8(Wq[(u * 2) * 64 + lane]);
        w[u][1] = as_h8(Wq[(u * 2 + 1) * 64 + lane]);
    }
    u32x4 w1[2];
    w1[0] = d.Wq1[(3 * 2) * 64 + lane];                  // section 3 of the packed array: the f16 fragments
    w1[1] = d.Wq1[(3 * 2 + 1) * 64 + lane];
    // g.bias is not read: an InverseLayer has no bias (the generic path's vector for this layer is all zeros)
    const int W = g.W, n_xb = g.n_xb, F = d.F;
    const int HW = H * W;
    const int rpi = d.runs_per_image;
    const int64_t stride = (int64_t)gridDim.x * 4;
    // write side: lane (x = fi, kq) owns taps mm = kq and kq + 4 of column x
    f32x4* pw = Pb + kq * 32 + 8 + fi;
    // read side: lanes 0..31 sum row 0 of a pair, lanes 32..63 row 1; lane & 31 = q - 16 b (0..22 are real)
    const int rt = lane >> 5, rq = lane & 31;
    const f32x4* pr = Pb + rt * 256 + (rq < 23 ? rq : 22) + 8;
    float raw[IN16 ? 1 : H][8];
    u32x4 rawh[IN16 ? H : 1];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // CL: the lane's channels 8 kq .. 8 kq + 7 as four pairs; pairs past Cin (30: the last pair of kq = 3) re-read the pair
    // before them -- finite numbers that meet zero weights -- instead of running into the next position / past the buffer
    const int Cin = g.Cin;
    int pair_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 8 * kq + 2 * q;
        pair_off[q] = c + 2 <= Cin ? c : Cin - 2;
    }
#define DCS_FETCH(img_, blk_)                                                                           \
    {                                                                                                   \
        const int xl_ = (blk_) * 16 + fi;                                                               \
        const float* ib_ = g.in + (img_) * g.in_n_stride;                                               \
        if constexpr (IN16) {                                                                           \
            const _Float16* ih_ = reinterpret_cast<const _Float16*>(g.in) + (img_) * g.in_n_stride +    \
                                  ((xl_ < W ? xl_ : W - 1) * 32 + 8 * kq);                              \
            _Pragma("unroll") for (int h = 0; h < H; ++h)                                               \
                rawh[h] = *reinterpret_cast<const u32x4*>(ih_ + (int64_t)h * W * 32);                   \
        } else if constexpr (CL) {                                                                      \
            const float* ip_ = ib_ + (xl_ < W ? xl_ : W - 1) * Cin;                                     \
            _Pragma("unroll") for (int h = 0; h < H; ++h) {                                             \
                const float* ir_ = ip_ + h * W * Cin;                                                   \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                         \
                    const f32x2 v_ = *reinterpret_cast<const f32x2*>(ir_ + pair_off[q]);                \
                    raw[h][2 * q] = v_[0];                                                              \
                    raw[h][2 * q + 1] = v_[1];                                                          \
                }                                                                                       \
            }                                                                                           \
        } else {                                                                                        \
            const LaneIn li_ = lane_in(g, kq, xl_ < W ? xl_ : W - 1, HW);                               \
            _Pragma("unroll") for (int h = 0; h < H; ++h) {                                             \
                const float* ir_ = ib_ + h * W;                                                         \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) raw[h][j] = ir_[li_.idx[j]];              \
            }                                                                                           \
        }                                                                                               \
    }
    for (int64_t run = (int64_t)blockIdx.x * 4 + wave; run < d.n_runs; run += stride) {
        const int64_t img = run / rpi;
        const int rr = (int)(run - img * rpi);
        const int b_lo = (int)((int64_t)rr * n_xb / rpi), b_hi = (int)((int64_t)(rr + 1) * n_xb / rpi);
        const int b_first = b_lo > 0 ? b_lo - 1 : 0;         // the block to the left is recomputed for its carry
        for (int i = lane; i < kCb / 4; i += 64) Cb[i] = zero4;
        DCS_FETCH(img, b_first)
        for (int b = b_first; b < b_hi; ++b) {
            h8 a[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                if constexpr (IN16) a[h] = as_h8(rawh[h]);
                else a[h] = round8(raw[h]);
            }
            {
                const int nb = b + 1 < b_hi ? b + 1 : b;     // last block of the run: a harmless re-read
                DCS_FETCH(img, nb)
            }
            const bool x_ok = b * 16 + fi < W;               // columns past W (last block only) must not reach the rows
            const bool edge = b * 16 + 16 > W;
            if (edge) {                                      // wave-uniform.  No bias in either InverseLayer: a zero input column
#pragma unroll                                               // gives a zero G column and zero products -- masked once, at the source
                for (int h = 0; h < H; ++h)
                    if (!x_ok) a[h] = h8{0, 0, 0, 0, 0, 0, 0, 0};
            }
            const bool keep = b >= b_lo;                     // wave-uniform: false for the recomputed block
            const int f0 = 4 * (b * 16 + rq);
            float* orow = d.out + (img * HO + rt) * (int64_t)F + f0;
#pragma unroll
            for (int y = 0; y < HO; y += 2) {
                f32x4 acc[2][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = zero4;
#pragma unroll
                for (int h = 0; h < H; ++h) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int u = h - (y + t) + PH;
                        if (u >= 0 && u < KH) {
                            acc[t][0] = mma(w[u < 0 || u >= KH ? 0 : u][0], a[h], acc[t][0]);
                            acc[t][1] = mma(w[u < 0 || u >= KH ? 0 : u][1], a[h], acc[t][1]);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float gv[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        gv[e] = acc[t][0][e];
                        gv[4 + e] = acc[t][1][e];
                    }
                    const h8 gh = round8(gv);
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) pw[t * 256 + mh * 128] = mma(as_h8(w1[mh]), gh, zero4);
                }
                asm volatile("" ::: "memory");               // the pieces of both rows are written (LDS is in order per wave)
                // all nine reads in flight before the first addition (left alone the compiler waits for each in turn: one
                // wave per SIMD, nobody hides the latency), a select instead of a branch for the carry
                f32x4 rd[8];
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) rd[mm] = pr[mm * 32 - mm];
                const f32x4 cin = Cb[(y + rt) * 8 + (rq & 7)];
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 sum = ((rd[0] + rd[1]) + (rd[2] + rd[3])) + ((rd[4] + rd[5]) + (rd[6] + rd[7])) + (rq < 8 ? cin : zero4);
                if (rq >= 16 && rq < 24) Cb[(y + rt) * 8 + rq - 16] = rq < 23 ? sum : zero4;
                if (keep && rq < 16) {
                    float* op = orow + (int64_t)y * F;
                    if (f0 + 4 <= F) {
                        *reinterpret_cast<f32x4u*>(op) = sum;
                    } else {
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                            if (f0 + e < F) op[e] = sum[e];
                    }
                }
                asm volatile("" ::: "memory");
            }
        }
        // the carry: the tail of the image (and the zeros up to F) for the last run, otherwise the next run recomputes it
        if (b_hi == n_xb) {
#pragma unroll
            for (int i = 0; i < (HO + 7) / 8; ++i) {
                const int t = (lane >> 3) + 8 * i, f = 4 * (16 * n_xb + (lane & 7));
                if (t < HO) {
                    const f32x4 v = Cb[t * 8 + (lane & 7)];
                    float* op = d.out + (img * HO + t) * (int64_t)F + f;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (f + e < F) op[e] = v[e];
                }
            }
        }
        asm volatile("" ::: "memory");
    }
#undef DCS_FETCH
}

}  // namespace

// Wh: [kh][32 out][40] halves, in-channel fastest (the LDS kernel's weights) -> [kh][2 halves][64 lanes] 16-byte pieces:
// lane (fi, kg) of (u, half) = W[u][out = fi + 16 half][in = 8 kg .. 8 kg + 7]
void dcs_colconv_wreg_pack(const _Float16* Wh, int kh, std::vector<_Float16>* out) {
    out->assign((size_t)kh * 2 * 64 * 8, (_Float16)0.f);
    for (int u = 0; u < kh; ++u)
        for (int half = 0; half < 2; ++half)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    (*out)[(((size_t)u * 2 + half) * 64 + lane) * 8 + j] =
                        Wh[((size_t)u * 32 + (lane & 15) + 16 * half) * 40 + (lane >> 4) * 8 + j];
}

static bool wreg_on() {
    static const bool on = !(getenv("DCS_COLCONV_WREG") && atoi(getenv("DCS_COLCONV_WREG")) == 0);
    return on;
}

bool dcs_colconv_wreg_scatter_ok(const DcsColConv& a) {
    return wreg_on() && a.Cin <= 32 && a.Cout <= 32 && a.kh == 20 && a.W >= 16 && a.ph == 0 && a.H == 30 && a.Ho == 11;
}

bool dcs_launch_colconv_wreg(dcs_ctx* ctx, const DcsColConv& a, int64_t n_images, const void* Wq) {
    const bool on = wreg_on();
    if (!on || !Wq || a.Cin > 32 || a.Cout > 32 || a.kh != 20 || a.W < 16) return false;
    if ((a.out_f16 || a.in_f16) && !dcs_colconv_wreg_scatter_ok(a)) return false;
    const int64_t n_units = n_images * a.n_xb;
    if (n_units <= 0) return true;
    const unsigned grid = (unsigned)std::min<int64_t>(dcs_cdiv(n_units, 4), ctx->n_cu);
    const u32x4* wq = reinterpret_cast<const u32x4*>(Wq);
    if (a.ph == a.kh - 1 && a.H == 11 && a.Ho == 30) {
        hipLaunchKernelGGL((colconv_wreg_gather_kernel<20, 11>), dim3(grid), dim3(kThreads), 0, ctx->stream, a, wq, n_units);
        return true;
    }
    if (a.ph == 0 && a.H == 30 && a.Ho == 11) {
        if (a.in_f16 && ((a.in_n_stride & 7) || (reinterpret_cast<uintptr_t>(a.in) & 15))) return false;
#define DCS_SC(O16_, I16_) hipLaunchKernelGGL((colconv_wreg_scatter_kernel<20, 30, 8, O16_, I16_>), dim3(grid), dim3(kThreads), 0, ctx->stream, a, wq, n_units)
        if (a.out_f16) { if (a.in_f16) DCS_SC(true, true); else DCS_SC(true, false); }
        else { if (a.in_f16) DCS_SC(false, true); else DCS_SC(false, false); }
#undef DCS_SC
        return true;
    }
    return false;
}

// W1p: [nf1][C][32 taps] (tap axis zero-padded) -> per output channel c of conv1^T (= input channel of the graph)
// [3 planes][2 tap halves][64 lanes][8] bf16: lane (fi, kg) of tap half mh holds, for row (mm = fi / 4 + 4 mh, r = fi % 4), the k
// slots j <-> ci = 4 kg + j (j < 4) | 16 + 4 kg + j - 4 ... followed by the same fragments rounded to f16 ([2 tap halves][64
// lanes][8]: section 3, the S2H kernel's operand).  Channel c's four sections start at c * 4 * 2 * 64 * 8.
void dcs_decoder_fused_pack(const float* W1p, int nf1, int C, std::vector<uint16_t>* out) {
    out->assign((size_t)C * 4 * 2 * 64 * 8, 0);
    for (int c = 0; c < C; ++c)
        for (int mh = 0; mh < 2; ++mh)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int fi = lane & 15, kg = lane >> 4;
                    const int ci = j < 4 ? 4 * kg + j : 16 + 4 * kg + (j - 4);
                    const int tap = 4 * ((fi >> 2) + 4 * mh) + (fi & 3);
                    float r = ci < nf1 ? W1p[((size_t)ci * C + c) * 32 + tap] : 0.f;
                    uint16_t* o = out->data() + (size_t)c * 4 * 2 * 64 * 8;
                    {
                        const _Float16 hv = (_Float16)r;
                        uint16_t hb;
                        memcpy(&hb, &hv, 2);
                        o[(((size_t)3 * 2 + mh) * 64 + lane) * 8 + j] = hb;
                    }
                    for (int p = 0; p < 3; ++p) {
                        uint32_t bits;
                        memcpy(&bits, &r, 4);
                        bits &= 0xffff0000u;
                        float part;
                        memcpy(&part, &bits, 4);
                        r -= part;
                        o[(((size_t)p * 2 + mh) * 64 + lane) * 8 + j] = (uint16_t)(bits >> 16);
                    }
                }
}

bool dcs_decoder_fused_ok(const DcsColConv& a, int F) {
    static const bool on = !(getenv("DCS_DECODER_FUSED") && atoi(getenv("DCS_DECODER_FUSED")) == 0);
    return on && a.Cin <= 32 && a.Cout <= 32 && a.kh == 20 && a.ph == 19 && a.H == 11 && a.Ho == 30 && a.W >= 16 &&
           F >= 4 * a.W + 26 && F <= 4 * a.W + 32;
}

bool dcs_launch_decoder_fused(dcs_ctx* ctx, const DcsColConv& a, int64_t n_images, const void* Wq, const void* Wq1, float* out,
                              int F, bool in_channels_last, bool in_f16) {
    if (!Wq || !Wq1 || !dcs_decoder_fused_ok(a, F)) return false;
    // f16 input (gemm_f16.hip): [image][row][x][32 halves], 16-byte loads
    if (in_f16 && (!in_channels_last || (a.in_n_stride & 7) || (reinterpret_cast<uintptr_t>(a.in) & 15))) return false;
    // channels-last input: pairs of channels are fetched as 8-byte loads
    if (in_channels_last && !in_f16 && ((a.Cin & 1) || (a.in_n_stride & 1) || (reinterpret_cast<uintptr_t>(a.in) & 7))) return false;
    if (n_images <= 0) return true;
    // runs per image: fewest (rounds of waves) x (blocks per run + the recomputed one)
    const int64_t n_waves = (int64_t)ctx->n_cu * 4;
    int best = 1;
    int64_t best_cost = -1;
    for (int rpi = 1; rpi <= a.n_xb; ++rpi) {
        const int64_t len = (a.n_xb + rpi - 1) / rpi + (rpi > 1 ? 1 : 0);
        const int64_t cost = dcs_cdiv(n_images * rpi, n_waves) * len;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rpi; }
    }
    DcsDecoderFused d{};
    d.Wq1 = reinterpret_cast<const u32x4*>(Wq1);
    d.out = out;
    d.F = F;
    d.runs_per_image = best;
    d.n_runs = n_images * best;
    const unsigned grid = (unsigned)std::min<int64_t>(dcs_cdiv(d.n_runs, 4), ctx->n_cu);
    const u32x4* wq = reinterpret_cast<const u32x4*>(Wq);
#define DCS_GO(CL_, IN16_) \
    hipLaunchKernelGGL((colconv_deconv1_fused_kernel<20, 11, CL_, IN16_>), dim3(grid), dim3(kThreads), 0, ctx->stream, a, wq, d)
    if (in_f16) DCS_GO(true, true);
    else if (in_channels_last) DCS_GO(true, false);
    else DCS_GO(false, false);
#undef DCS_GO
    return true;
}
