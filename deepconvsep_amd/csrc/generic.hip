// Generic build_ca path: iKala (separate_ikala.py:172-192), Bach10 (separate_bach10.py:172-229) and the
// score-informed graph (bach10_scoreinformed/separate_bach10.py:388-447) on gfx950.
//
//   conv1 (1 x 30, stride 3|4)            conv1_kernel          direct, VALU (2-9 % of the graph's FLOPs)
//   MaxPool2DLayer((1,4))                 pool_kernel
//   conv2 (10 x 20 | 20 x 1)              conv_igemm_kernel     implicit GEMM, LDS-staged im2col, f32 MFMA
//   DenseLayer x (1 + branches)           gemm_rows_kernel      (gemm.hip; weight streaming, HBM-bound)
//   InverseLayer(., conv2)                conv_igemm_kernel     'full' correlation with W2 (channels swapped)
//   InverseLayer(., pool)                 unpool_kernel         tie routing ALL (Theano CPU) | FIRST (cuDNN)
//   InverseLayer(., conv1)                deconv1_kernel        direct, VALU
//   concat + bias + rectify + soft mask   mask_kernel
//
// Layouts are the reference's NCHW ([tile, channel, time, frequency]); branches are folded into the tile
// axis ([n*S, ...]) from the per-source dense layers on.
#include "generic.h"

#include <string.h>

#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------------------------
// conv1: out[n,o,t,j] = bias[o] + sum_{c,u} x[n,c,t,j*sw+u] * Wc[o,c,u]     (Wc = flipped W1)
// block = 256 consecutive j of one (n,t) row; the input row segments live in LDS, the weights too.
// ------------------------------------------------------------------------------------------------
template <int NF>
__global__ __launch_bounds__(kThreads) void conv1_kernel(const float* __restrict__ x, const float* __restrict__ Wc,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int C, int tc, int F, int kw, int sw, int w1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                   // [NF][C][kw]
    float* xs = smem + NF * C * kw;     // [C][256*sw + kw]
    const int tid = threadIdx.x;
    const int64_t nt = blockIdx.y;      // n*tc + t
    const int64_t n = nt / tc;
    const int t = (int)(nt - n * tc);
    const int j0 = blockIdx.x * kThreads;
    const int seg = kThreads * sw + kw;
    for (int i = tid; i < NF * C * kw; i += kThreads) Ws[i] = Wc[i];
    for (int c = 0; c < C; ++c) {
        const float* xr = x + ((n * C + c) * tc + t) * (int64_t)F;
        for (int i = tid; i < seg; i += kThreads) {
            const int f = j0 * sw + i;
            xs[c * seg + i] = (f < F) ? xr[f] : 0.f;
        }
    }
    __syncthreads();
    const int j = j0 + tid;
    if (j >= w1) return;
    float acc[NF];
#pragma unroll
    for (int o = 0; o < NF; ++o) acc[o] = bias[o];
    for (int c = 0; c < C; ++c) {
        for (int u = 0; u < kw; ++u) {
            const float xv = xs[c * seg + tid * sw + u];
#pragma unroll
            for (int o = 0; o < NF; ++o) acc[o] = fmaf(xv, Ws[(o * C + c) * kw + u], acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < NF; ++o) out[((n * NF + o) * tc + t) * (int64_t)w1 + j] = acc[o];
}

// conv1, register-blocked: a thread owns one output position j of a row and all NF filters; the kw inputs of the current
// channel sit in registers (for a stride of 4 the window starts 16-byte aligned: eight 16-byte loads, shared with the
// neighbouring threads through L1), the filter taps -- the same for every thread -- come through scalar loads from
// Wt[c][u][32] (filters fastest), so the inner loop is NF multiply-adds per tap with one vector and one scalar operand.
// The LDS kernel above does one (broadcast) LDS read per multiply-add and is bound by that.
// POOL: MaxPool2DLayer((1, 4)) fused in (iKala): the four lanes of a pooling window take the window maximum with two
// quad permutes, the workgroup writes the POOLED rows [n][NF][tc][wp] and -- instead of the full-resolution activations,
// which only the un-pooling VJP would read again -- the routing of that VJP as 4 bits per window (the positions that equal
// the maximum: all of them, Theano 0.9's CPU MaxPoolGrad, or only the first with tie_first), 8 windows per 32-bit word,
// mw words per row: a wave's ballot IS its two words.  deconv1_reg_kernel<., ., true> consumes the pair.
template <int NF, int SW, bool POOL = false>
__global__ __launch_bounds__(kThreads) void conv1_reg_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int C, int tc, int F, int kw, int w1,
                                                             unsigned* __restrict__ maskw = nullptr, int wp = 0, int mw = 0,
                                                             int tie_first = 0) {
    constexpr int KW = 32;                  // taps held in registers (kw <= 32)
    const int64_t nt = blockIdx.y;          // n*tc + t
    const int64_t n = nt / tc;
    const int t = (int)(nt - n * tc);
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= w1) return;
    float acc[NF];
#pragma unroll
    for (int o = 0; o < NF; ++o) acc[o] = bias[o];
    for (int c = 0; c < C; ++c) {
        const float* xr = x + ((n * C + c) * tc + t) * (int64_t)F + (int64_t)j * SW;
        float xv[KW];
        const bool vec = SW == 4 && (F & 3) == 0 && j * SW + KW <= F && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
        if (vec) {
#pragma unroll
            for (int q = 0; q < KW / 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[4 * q + e] = v[e];
            }
        } else {
#pragma unroll
            for (int u = 0; u < KW; ++u) xv[u] = (u < kw && j * SW + u < F) ? xr[u] : 0.f;
        }
        const float* wc = Wt + (int64_t)c * kw * 32;     // uniform: scalar loads
#pragma unroll
        for (int u = 0; u < KW; ++u) {
            if (u < kw) {
                // one v_fmac with a scalar-register operand per filter (written out: left to itself the compiler packs
                // pairs into v_pk_fma_f32 and spends three moves per multiply-add bringing the scalar taps into vector pairs)
#pragma unroll
                for (int o = 0; o < NF; ++o) {
                    const float w = wc[u * 32 + o];
                    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[o]) : "s"(w), "v"(xv[u]));
                }
            }
        }
    }
    if (POOL) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int o = 0; o < NF; ++o) {
            const int ai = __builtin_bit_cast(int, acc[o]);
            const float m1 = fmaxf(acc[o], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ai, ai, 0xB1, 0xf, 0xf, false)));
            const int mi = __builtin_bit_cast(int, m1);
            const float m = fmaxf(m1, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(mi, mi, 0x4E, 0xf, 0xf, false)));
            unsigned long long b = __builtin_amdgcn_ballot_w64(acc[o] == m);
            if (tie_first)   // keep the lowest set bit of every nibble
                b &= ~(((b << 1) & 0xEEEEEEEEEEEEEEEEull) | ((b << 2) & 0xCCCCCCCCCCCCCCCCull) | ((b << 3) & 0x8888888888888888ull));
            const int64_t row = (n * NF + o) * tc + t;
            if ((lane & 3) == 0 && (j >> 2) < wp) out[row * wp + (j >> 2)] = m;
            if ((lane & 31) == 0) maskw[row * mw + (j >> 5)] = lane ? (unsigned)(b >> 32) : (unsigned)b;
        }
        return;
    }
#pragma unroll
    for (int o = 0; o < NF; ++o) out[((n * NF + o) * tc + t) * (int64_t)w1 + j] = acc[o];
}

// ------------------------------------------------------------------------------------------------
// the pad columns [width, pitch) of `rows` rows of a row-major buffer: the dense layers read rows of `pitch` = width rounded
// up to 4 floats, the producers write `width` (one thread per pad element: 1 - 3 per row.  A whole-buffer hipMemsetAsync in
// front of the producers zero-filled 265 MB per Bach10 clip for 2 KB of pad: 27 us per launch, profiles/r05_u_*)
__global__ __launch_bounds__(kThreads) void zero_pad_cols_kernel(float* __restrict__ buf, int64_t rows, int width, int pitch) {
    const int npad = pitch - width;
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= rows * npad) return;
    const int64_t r = idx / npad;
    buf[r * pitch + width + (int)(idx - r * npad)] = 0.f;
}

// conv2 -> BiasLayer -> bottleneck DenseLayer folded into one affine map of the tile's conv2 INPUT (model creation, once):
//   Z[h] = relu(c[h] + sum_{ci,r,x} p1[ci][r][x] W2fc[(ci, r, x)][h]),
//   W2fc[(ci, r, x)][h] = sum_{u,v : 0 <= r-u < h2, 0 <= x-v < w2} sum_co Wf[u][v][co][ci] Bfc[(co, r-u, x-v)][h]
// (no nonlinearity between the two layers: separate_ikala.py:181-186).  float64 sums, rounded once to float32.
// One workgroup per input position k = (ci, r, x), one thread per hidden unit; Wf is read through scalar loads.
__global__ __launch_bounds__(kThreads) void fold_conv2_fc_kernel(const float* __restrict__ Wf /* [kh*kw][nf2][nf1] */,
                                                                 const float* __restrict__ Bfc /* [flat_p][hid64] */,
                                                                 float* __restrict__ out /* [nf1*tc*wp][hid64] */, int nf1, int nf2,
                                                                 int kh, int kw, int tc, int wp, int h2, int w2, int hid64) {
    const int k = blockIdx.x, ci = k / (tc * wp), r = (k / wp) % tc, x = k % wp;
    for (int h = threadIdx.x; h < hid64; h += kThreads) {
        double acc = 0.0;
        for (int u = r - h2 + 1 > 0 ? r - h2 + 1 : 0; u < kh && u <= r; ++u)
            for (int v = x - w2 + 1 > 0 ? x - w2 + 1 : 0; v < kw && v <= x; ++v) {
                const float* wrow = Wf + ((size_t)(u * kw + v) * nf2) * nf1 + ci;
                const float* brow = Bfc + ((size_t)(r - u) * w2 + (x - v)) * hid64 + h;
                for (int co = 0; co < nf2; ++co)
                    acc += (double)wrow[(size_t)co * nf1] * (double)brow[(size_t)co * h2 * w2 * hid64];
            }
        out[(size_t)k * hid64 + h] = (float)acc;
    }
}

// max-pool (1,pw), stride pw, ignore_border: rows of w1 -> rows of wp
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void pool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int64_t rows, int w1, int wp, int pw) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= rows * wp) return;
    const int64_t r = idx / wp;
    const int j = (int)(idx - r * wp);
    const float* p = in + r * w1 + j * pw;
    float m = p[0];
    for (int q = 1; q < pw; ++q) m = fmaxf(m, p[q]);
    out[idx] = m;
}

// VJP of the pool at the forward input `a` (rows of w1): position f receives g[f/pw] when a[f] equals
// its window maximum -- every such position (tie_all, Theano 0.9 CPU MaxPoolGrad) or only the first.
// g rows are indexed [n*S + s], a rows [n] (g has S branch copies per tile).  One thread per pooling WINDOW: it reads
// the window of `a` once, decides the routing and writes the pw outputs (one 16-byte store for pw = 4 and aligned
// rows); the border columns past wp*pw are zeroed by the first threads of a row.  32-bit indexing (the launcher checks).
template <int PW>
__global__ __launch_bounds__(kThreads) void unpool_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                          float* __restrict__ out, unsigned rows_g, unsigned rows_per_tile,
                                                          unsigned S, int w1, int wp, int tie_first) {
    const unsigned rg = blockIdx.x * 4u + (threadIdx.x >> 6);   // a wave per row of g / out
    if (rg >= rows_g) return;
    const unsigned tile_s = rg / rows_per_tile;             // n*S + s
    const unsigned ra = (tile_s / S) * rows_per_tile + (rg - tile_s * rows_per_tile);
    float* orow = out + (int64_t)rg * w1;
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < w1 - wp * PW; j += 64) orow[wp * PW + j] = 0.f;   // columns the pool ignored
    for (int j = lane; j < wp; j += 64) {
    const float* p = a + (int64_t)ra * w1 + j * PW;
    float win[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) win[q] = p[q];
    float m = win[0];
#pragma unroll
    for (int q = 1; q < PW; ++q) m = fmaxf(m, win[q]);
    const float gv = g[(int64_t)rg * wp + j];
    float v[PW];
    bool seen = false;
#pragma unroll
    for (int q = 0; q < PW; ++q) {
        const bool hit = win[q] == m && !(tie_first && seen);
        seen = seen || win[q] == m;
        v[q] = hit ? gv : 0.f;
    }
    if (PW == 4 && (w1 & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        *reinterpret_cast<f32x4*>(orow + j * PW) = f32x4{v[0], v[1], v[2], v[3 % PW]};
    } else {
#pragma unroll
        for (int q = 0; q < PW; ++q) orow[j * PW + q] = v[q];
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM correlation, stride 1, virtual zero padding (ph, pw):
//   out[n, co, y, x] = bias[co] + sum_{ci,u,v} in[n, ci, y+u-ph, x+v-pw] * Wm[(ci,u,v)][co]
// M = n*Ho*Wo output positions, N = 32 (Cout <= 32, zero padded), K = Cin*kh*kw.
// Workgroup = 128 positions x 32 channels; K tile 32.  A tile: lane <-> position (consecutive x ->
// coalesced gathers), im2col offsets from a per-k table; staged in LDS with the B tile; 4 waves x
// (2 row blocks x 2 column blocks) of v_mfma_f32_16x16x4_f32.  The output tile goes back through
// LDS so that the NCHW store is coalesced along x.
// ------------------------------------------------------------------------------------------------
struct IgemmArgs {
    const float* in; int64_t in_n_stride; int Cin, H, W;
    const float* Wm;            // [Kpad][32]
    const int* koff;            // [Kpad] ci*H*W + u*W + v  (0 for padding rows, their weights are 0)
    const int* kuv;             // [Kpad] (u << 16) | v
    const float* bias;          // [32]
    float* out; int64_t out_n_stride; int Cout, Ho, Wo;
    int ph, pw, K;
    int64_t M;                  // n*Ho*Wo
    int kh, k_per_u;            // k_per_u > 0: K is ordered tap-row major, k = u*k_per_u + ..., u = 0..kh-1 (see below)
};

// Zero-tap skipping for the transposed ('full') correlation.  With ph = kh-1 rows of implicit zero padding only
// the taps u with 0 <= y + u - ph < H touch real input: for the Bach10 graph (H = 11, kh = 20, 30 output rows)
// that is 7.3 of 20 on average -- 63 % of a dense K loop multiplies zeros.  With the K axis ordered tap-row major
// the taps that can be non-zero for the rows of a workgroup form one interval of k; the K loop runs over the
// tiles of that interval only (the per-element bounds test still zeroes what is outside at its ends).
__device__ __forceinline__ void igemm_k_range(const IgemmArgs& g, int64_t m0, int BM, int BK, int& kt0, int& kt1) {
    kt0 = 0;
    kt1 = (g.K + BK - 1) / BK;
    if (g.k_per_u <= 0 || g.ph == 0) return;
    const int HoWo = g.Ho * g.Wo;
    int64_t ma = m0, mb = m0 + BM - 1;
    if (mb > g.M - 1) mb = g.M - 1;
    const int64_t na = ma / HoWo, nb = mb / HoWo;
    const int ya = (int)((ma - na * HoWo) / g.Wo), yb = (int)((mb - nb * HoWo) / g.Wo);
    if (nb != na) return;                            // the workgroup straddles two images: keep the full loop
    int u_lo = g.ph - yb, u_hi = g.ph - ya + g.H - 1;
    if (u_lo < 0) u_lo = 0;
    if (u_hi > g.kh - 1) u_hi = g.kh - 1;
    if (u_hi < u_lo) { kt1 = 0; return; }
    kt0 = (u_lo * g.k_per_u) / BK;
    const int ke = (u_hi + 1) * g.k_per_u;
    kt1 = (ke + BK - 1) / BK;
    const int kmax = (g.K + BK - 1) / BK;
    if (kt1 > kmax) kt1 = kmax;
}

__global__ __launch_bounds__(kThreads) void conv_igemm_kernel(const IgemmArgs g) {
    constexpr int BM = 128, BK = 32, BN = 32, AS = BK + 2, BS = BN + 16;
    __shared__ __attribute__((aligned(16))) float lds[BM * AS + BK * BS];  // reused as Cs[BN][BM+1] at the end
    float* As = lds;
    float* Bs = lds + BM * AS;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    const bool padded = (g.ph | g.pw) != 0;

    // A gather plan: this thread serves position row = tid % 128 and k = kt*32 + (tid/128) + 2*i
    const int row = tid & (BM - 1);
    const int ksub = tid >> 7;
    const int64_t m = m0 + row;
    const bool row_ok = m < g.M;
    int64_t n = 0;
    int y = 0, x = 0;
    if (row_ok) {
        n = m / HoWo;
        const int r = (int)(m - n * HoWo);
        y = r / g.Wo;
        x = r - y * g.Wo;
    }
    const float* in_base = g.in + n * g.in_n_stride + (int64_t)(y - g.ph) * g.W + (x - g.pw);
    // B tile plan: 32 x 32 floats = 256 float4
    const int b_row = tid >> 3, b_c4 = tid & 7;

    float ra[16];
    f32x4 rb;
#define DCS_IG_LOAD(kt_)                                                                       \
    {                                                                                          \
        const int k0_ = (kt_) * BK;                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                       \
            const int k = k0_ + ksub + 2 * i;                                                  \
            float v = 0.f;                                                                     \
            if (row_ok && k < g.K) {                                                           \
                bool ok = true;                                                                \
                if (padded) {                                                                  \
                    const int uv = g.kuv[k];                                                   \
                    const int yy = y + (uv >> 16) - g.ph, xx = x + (uv & 0xffff) - g.pw;       \
                    ok = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;                           \
                }                                                                              \
                if (ok) v = in_base[g.koff[k]];                                                \
            }                                                                                  \
            ra[i] = v;                                                                         \
        }                                                                                      \
        rb = *reinterpret_cast<const f32x4*>(g.Wm + (int64_t)(k0_ + b_row) * BN + b_c4 * 4);   \
    }
#define DCS_IG_STORE()                                                                         \
    {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) As[row * AS + ksub + 2 * i] = ra[i];    \
        *reinterpret_cast<f32x4*>(Bs + b_row * BS + b_c4 * 4) = rb;                            \
    }

    f32x4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    int kt_first, nkt;
    igemm_k_range(g, m0, BM, BK, kt_first, nkt);
    if (kt_first < nkt) DCS_IG_LOAD(kt_first)
    for (int kt = kt_first; kt < nkt; ++kt) {
        __syncthreads();
        DCS_IG_STORE()
        __syncthreads();
        if (kt + 1 < nkt) DCS_IG_LOAD(kt + 1)
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const float b0 = Bs[(kk * 4 + kq) * BS + fi];
            const float b1 = Bs[(kk * 4 + kq) * BS + 16 + fi];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float a = As[(wave * 32 + r * 16 + fi) * AS + kk * 4 + kq];
                acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[r][1], 0, 0, 0);
            }
        }
    }
#undef DCS_IG_LOAD
#undef DCS_IG_STORE
    __syncthreads();
    // transpose through LDS: Cs[co][position]
    float* Cs = lds;
    constexpr int CS = BM + 1;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                Cs[(c * 16 + fi) * CS + wave * 32 + r * 16 + kq * 4 + e] = acc[r][c][e];
    __syncthreads();
    for (int idx = tid; idx < g.Cout * BM; idx += kThreads) {
        const int co = idx / BM, p = idx - co * BM;
        const int64_t mm = m0 + p;
        if (mm < g.M) {
            const int64_t nn = mm / HoWo;
            const int r = (int)(mm - nn * HoWo);
            g.out[nn * g.out_n_stride + (int64_t)co * HoWo + r] = Cs[co * CS + p] + g.bias[co];
        }
    }
    (void)HW;
}

// ------------------------------------------------------------------------------------------------
// Column convolution (filter kh x 1: conv2 of the Bach10 / score-informed graphs and its transpose).
//   out[co][y][x] = bias[co] + sum_u sum_ci Wk[u][ci][co] * in[ci][y + u - ph][x]     (zero outside 0 <= row < H)
// forward: ph = 0, Ho = H - kh + 1;  transposed ('full' correlation): ph = kh - 1, Ho = H + kh - 1.
// Every x is an independent 1-D problem along y, and an input element feeds up to kh outputs.  The implicit GEMM
// above re-gathers each of them from L2 (16 scalar loads + bounds logic per thread per K tile -- as many issue
// cycles as the MFMAs) and, for the transpose, multiplied the zero padding.  Here a workgroup owns 16 columns x of
// one image: the input slab [ci][H][16 x] and ALL the weights [u][ci][co] sit in LDS, each of the 8 waves takes
// output rows y = w, w+8, ... and walks only the taps with a real input row (7.3 of 20 on average for the
// Bach10 transpose).  MFMA operands: A = weights (rows = co), B = slab (columns = x), so a lane ends up with 4
// channels of one x and a store instruction writes 16 consecutive x of one (co, y).  A workgroup keeps its
// weights for several column blocks of the same image.  LDS strides / the co swizzle keep the two K-quarters of a
// 32-lane half on disjoint banks.  Measured (Bach10, 10 s): transpose 5.2 -> see DESIGN.md, forward 0.95 -> idem.
// ------------------------------------------------------------------------------------------------
typedef DcsColConv ColConvArgs;

__host__ __device__ __forceinline__ int colconv_wslot(int u, int ci, int co) {
    return (u * 32 + ci) * 32 + ((co + 16 * (ci & 1)) & 31);
}

constexpr int kColThreads = 512;

// LDS layouts (round 3): a lane's eight K values of a tap (channels kq, kq + 4, ..., kq + 28) are TWO 16-byte reads for
// each operand instead of eight 4-byte ones -- weights Wl[u][kq][half][co 32][4], slab[kq][row][half][x 16][4]: within a
// (u / row, kq, half) the 16 lanes of a fragment read 256 contiguous bytes, and the kq planes are whole multiples of 256
// bytes apart, so the reads are conflict-free for the lane groups a ds_read_b128 is served in.  6 LDS reads per tap and
// wave instead of 24 (the f32 MFMAs wait for their operands: Bach10 f32 conv2^T 1.73 ms, 0.48 of the f32 MFMA peak before).
__global__ __launch_bounds__(kColThreads) void colconv_kernel(const ColConvArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                         // [kh][4 kq][2][32 co][4]
    float* slab = smem + g.kh * 1024;         // [4 kq][H][2][16 x][4]   (512 H floats; the launcher reserves 32 (16 H + 16))
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    const int groups = (g.n_xb + g.xb_per_wg - 1) / g.xb_per_wg;
    const int64_t img = blockIdx.x / groups;
    const int xb0 = (int)(blockIdx.x - img * groups) * g.xb_per_wg;
    const float* in = g.in + img * g.in_n_stride;
    float* out = g.out + img * g.out_n_stride;
    const int HoW = g.Ho * g.W;
    const int H = g.H;

    // weights: global [u][ci][co swizzled] (colconv_wslot) -> fragment order
    for (int i = tid; i < g.kh * 1024; i += kColThreads) {
        const int u = i >> 10, ci = (i >> 5) & 31, slot = i & 31;
        const int co = (slot - 16 * (ci & 1)) & 31;
        const int kk = ci >> 2;
        Wl[((((u * 4 + (ci & 3)) * 2 + (kk >> 2)) * 32 + co) << 2) + (kk & 3)] = g.Wk[i];
    }
    for (int i = tid; i < 512 * H; i += kColThreads) slab[i] = 0.f;   // channels >= Cin stay zero
    const float bias_lo[4] = {g.bias[4 * kq], g.bias[4 * kq + 1], g.bias[4 * kq + 2], g.bias[4 * kq + 3]};
    const float bias_hi[4] = {g.bias[16 + 4 * kq], g.bias[17 + 4 * kq], g.bias[18 + 4 * kq], g.bias[19 + 4 * kq]};

    // slab staging plan: element e = tid + 512 i  ->  (x = e % 16, cr = e / 16 = ci*H + r); the next column block's
    // elements are fetched into registers while the current one is multiplied (one workgroup per CU: nothing else
    // would hide the loads)
    constexpr int kPre = 32;
    const int n_el = g.Cin * H * 16;
    int soff[kPre];
    float pre[kPre];
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
        const int e = tid + i * kColThreads;
        const int cr = e >> 4, ci = cr / H, r = cr - ci * H, kk = ci >> 2;
        soff[i] = e < n_el ? (((((ci & 3) * H + r) * 2 + (kk >> 2)) * 16 + (e & 15)) << 2) + (kk & 3) : -1;
    }
#define DCS_COL_FETCH(x0_)                                                                               \
    _Pragma("unroll") for (int i = 0; i < kPre; ++i) {                                                   \
        const int e = tid + i * kColThreads;                                                             \
        pre[i] = (soff[i] >= 0 && (x0_) + (e & 15) < g.W) ? in[(int64_t)(e >> 4) * g.W + (x0_) + (e & 15)] : 0.f; \
    }
    const int xb_end = xb0 + g.xb_per_wg < g.n_xb ? xb0 + g.xb_per_wg : g.n_xb;
    if (xb0 < xb_end) DCS_COL_FETCH(xb0 * 16)
    const f32x4* sl4 = reinterpret_cast<const f32x4*>(slab) + kq * H * 32 + fi;      // + (row * 2 + half) * 16
    const f32x4* wl4 = reinterpret_cast<const f32x4*>(Wl) + kq * 64 + fi;            // + u * 256 + half * 32 (+ 16: co + 16)
    for (int xb = xb0; xb < xb_end; ++xb) {
        const int x0 = xb * 16;
        __syncthreads();                      // previous slab fully read (and, first time, the weights staged)
#pragma unroll
        for (int i = 0; i < kPre; ++i)
            if (soff[i] >= 0) slab[soff[i]] = pre[i];
        __syncthreads();
        if (xb + 1 < xb_end) DCS_COL_FETCH((xb + 1) * 16)
        for (int y = wave; y < g.Ho; y += kColThreads / 64) {
            int u_lo = g.ph - y, u_hi = g.ph - y + H - 1;            // 0 <= y + u - ph < H
            if (u_lo < 0) u_lo = 0;
            if (u_hi > g.kh - 1) u_hi = g.kh - 1;
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int u = u_lo; u <= u_hi; ++u) {
                const f32x4* sp = sl4 + (y + u - g.ph) * 32;
                const f32x4* wp = wl4 + u * 256;
                const f32x4 b0 = sp[0], b1 = sp[16];
                const f32x4 a00 = wp[0], a01 = wp[32], a10 = wp[16], a11 = wp[48];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a00[kk], b0[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a10[kk], b0[kk], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a01[kk], b1[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a11[kk], b1[kk], acc1, 0, 0, 0);
                }
            }
            if (x0 + fi < g.W) {
                float* op = out + (int64_t)y * g.W + x0 + fi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = 4 * kq + e;
                    if (co < g.Cout) op[(int64_t)co * HoW] = acc0[e] + bias_lo[e];
                    if (co + 16 < g.Cout) op[(int64_t)(co + 16) * HoW] = acc1[e] + bias_hi[e];
                }
            }
        }
    }
#undef DCS_COL_FETCH
}

// ------------------------------------------------------------------------------------------------
// The same implicit GEMM with f16 inputs and f32 accumulation (v_mfma_f32_16x16x32_f16, 16x the f32 MFMA
// rate) -- the "fp16 MFMA conv path" of BASELINE config 3.  Activations are rounded to f16 when the
// im2col tile is stored to LDS, weights are pre-rounded on the host and packed [k tile][32 channels][32 k]
// so that a lane's 8 consecutive k values are one 16-byte LDS read for both operands.  Opt-in
// (dcs_model_set_conv_precision): results differ from the f32 path at the 1e-3 level (f16 has an 11-bit
// significand); the parity test states the tolerance it meets.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(kThreads) void conv_igemm_f16_kernel(const IgemmArgs g, const _Float16* __restrict__ Wh) {
    constexpr int BM = 128, BK = 32, BN = 32, AS = BK + 8, BS = BK + 8;  // strides in halves (80 bytes)
    __shared__ __attribute__((aligned(16))) float lds[BN * (BM + 1)];   // >= A tile + B tile; reused for the output
    _Float16* As = reinterpret_cast<_Float16*>(lds);          // [BM][AS]
    _Float16* Bs = As + BM * AS;                              // [BN][BS]  (channel-major, k contiguous)
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kg = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int HoWo = g.Ho * g.Wo;
    const bool padded = (g.ph | g.pw) != 0;
    const int row = tid & (BM - 1);
    const int ksub = tid >> 7;
    const int64_t m = m0 + row;
    const bool row_ok = m < g.M;
    int64_t n = 0;
    int y = 0, x = 0;
    if (row_ok) {
        n = m / HoWo;
        const int r = (int)(m - n * HoWo);
        y = r / g.Wo;
        x = r - y * g.Wo;
    }
    const float* in_base = g.in + n * g.in_n_stride + (int64_t)(y - g.ph) * g.W + (x - g.pw);
    float ra[16];
    f32x4 rb = f32x4{0.f, 0.f, 0.f, 0.f};  // 8 halves of the weight tile (threads 0..127)
#define DCS_IGH_LOAD(kt_)                                                                      \
    {                                                                                          \
        const int k0_ = (kt_) * BK;                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                       \
            const int k = k0_ + ksub + 2 * i;                                                  \
            float v = 0.f;                                                                     \
            if (row_ok && k < g.K) {                                                           \
                bool ok = true;                                                                \
                if (padded) {                                                                  \
                    const int uv = g.kuv[k];                                                   \
                    const int yy = y + (uv >> 16) - g.ph, xx = x + (uv & 0xffff) - g.pw;       \
                    ok = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;                           \
                }                                                                              \
                if (ok) v = in_base[g.koff[k]];                                                \
            }                                                                                  \
            ra[i] = v;                                                                         \
        }                                                                                      \
        if (tid < 128) rb = *reinterpret_cast<const f32x4*>(Wh + (int64_t)(kt_) * (BN * BK) + tid * 8); \
    }
#define DCS_IGH_STORE()                                                                        \
    {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) As[row * AS + ksub + 2 * i] = (_Float16)ra[i]; \
        if (tid < 128) *reinterpret_cast<f32x4*>(Bs + (tid >> 2) * BS + (tid & 3) * 8) = rb;   \
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kt_first, nkt;
    igemm_k_range(g, m0, BM, BK, kt_first, nkt);
    if (kt_first < nkt) DCS_IGH_LOAD(kt_first)
    for (int kt = kt_first; kt < nkt; ++kt) {
        __syncthreads();
        DCS_IGH_STORE()
        __syncthreads();
        if (kt + 1 < nkt) DCS_IGH_LOAD(kt + 1)
        const h8 b0 = *reinterpret_cast<const h8*>(Bs + fi * BS + kg * 8);
        const h8 b1 = *reinterpret_cast<const h8*>(Bs + (16 + fi) * BS + kg * 8);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const h8 a = *reinterpret_cast<const h8*>(As + (wave * 32 + r * 16 + fi) * AS + kg * 8);
            acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc[r][0], 0, 0, 0);
            acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc[r][1], 0, 0, 0);
        }
    }
#undef DCS_IGH_LOAD
#undef DCS_IGH_STORE
    __syncthreads();
    float* Cs = lds;
    constexpr int CS = BM + 1;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                Cs[(c * 16 + fi) * CS + wave * 32 + r * 16 + kg * 4 + e] = acc[r][c][e];
    __syncthreads();
    for (int idx = tid; idx < g.Cout * BM; idx += kThreads) {
        const int co = idx / BM, p = idx - co * BM;
        const int64_t mm = m0 + p;
        if (mm < g.M) {
            const int64_t nn = mm / HoWo;
            const int r = (int)(mm - nn * HoWo);
            g.out[nn * g.out_n_stride + (int64_t)co * HoWo + r] = Cs[co * CS + p] + g.bias[co];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Slab convolution for general kh x kw filters (conv2 of the iKala graph, 10 x 20, and its transpose).
//   out[co][y][x] = bias[co] + sum_{u,v} sum_ci Wk[u][v][ci][co] * in[ci][y + u - ph][x + v - pw]   (zero outside)
// A workgroup owns a band of output rows of one image.  The input rows the band can touch -- at most band + kh - 1
// -- sit in LDS for the whole workgroup, the weights stream through LDS a few taps at a time (double buffered), and
// each wave keeps the accumulators of its (row, 16-column) blocks in registers across all taps.  Per tap a wave reads
// the 16 weight fragments once and, for each of its blocks, 8 slab values (lane = column; a lane whose shifted column
// falls outside the input gets 0) for 16 MFMAs; taps whose input row lies outside the image, and (tap, block) pairs
// whose shifted columns all do, are skipped -- the implicit GEMM multiplies all of them (63 % of its K loop for the
// iKala transpose) and re-gathers every operand from L2.
// ------------------------------------------------------------------------------------------------
typedef DcsSlabConv SlabConvArgs;

__global__ __launch_bounds__(kColThreads) void slabconv_kernel(const SlabConvArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wstage = g.tstage * 1024;            // floats per weight stage
    float* Wl = smem;                              // [2][tstage][32][32]
    float* slab = smem + 2 * wstage;               // [32][rows_max][W]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    const int64_t img = blockIdx.x / g.n_bands;
    const int y0 = (int)(blockIdx.x - img * g.n_bands) * g.band;
    const int yb = y0 + g.band < g.Ho ? y0 + g.band : g.Ho;   // rows [y0, yb)
    const float* in = g.in + img * g.in_n_stride;
    float* out = g.out + img * g.out_n_stride;
    // input rows the band can touch
    int rbase = y0 - g.ph, rtop = yb - 1 - g.ph + g.kh - 1;
    if (rbase < 0) rbase = 0;
    if (rtop > g.H - 1) rtop = g.H - 1;
    const int rows = rtop - rbase + 1;
    const int PS = g.rows_max * g.W;               // plane stride of the slab
    for (int i = tid; i < 32 * PS; i += kColThreads) {
        const int ci = i / PS, rem = i - ci * PS;
        const int r = rem / g.W;
        slab[i] = (ci < g.Cin && r < rows) ? in[((int64_t)ci * g.H + rbase + r) * g.W + (rem - r * g.W)] : 0.f;
    }
    // this wave's blocks: b = wave + 8 i -> (row y0 + b / nxb, column block b % nxb)
    const int nxb = (g.Wo + 15) >> 4, nblk = (yb - y0) * nxb;
    int by[4], bx[4];
    f32x4 acc0[4], acc1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = wave + 8 * i;
        by[i] = b < nblk ? y0 + b / nxb : -1;
        bx[i] = b < nblk ? (b % nxb) * 16 : 0;
        acc0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int c0 = (fi + 16 * (kq & 1)) & 31, c1 = (fi + 16 + 16 * (kq & 1)) & 31;
    const int nvs = (g.kw + g.tstage - 1) / g.tstage;   // stages per tap row
    // taps rows u that touch the band at all: 0 <= y + u - ph < H for some y in [y0, yb)
    int u_lo = g.ph - (yb - 1), u_hi = g.ph - y0 + g.H - 1;
    if (u_lo < 0) u_lo = 0;
    if (u_hi > g.kh - 1) u_hi = g.kh - 1;
    const int n_stage = (u_hi - u_lo + 1) * nvs;
    f32x4 wpre[2];                                  // 2 float4 per thread cover a stage of <= 4 taps (4096 floats)
#define DCS_SLAB_WFETCH(st_)                                                                             \
    {                                                                                                    \
        const int u_ = u_lo + (st_) / nvs, v_ = ((st_) % nvs) * g.tstage;                                \
        const int nt_ = v_ + g.tstage <= g.kw ? g.tstage : g.kw - v_;                                     \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                  \
            const int e = (tid + q * kColThreads) * 4;                                                   \
            wpre[q] = e < nt_ * 1024 ? *reinterpret_cast<const f32x4*>(g.Wk + ((int64_t)(u_ * g.kw + v_)) * 1024 + e) \
                                     : f32x4{0.f, 0.f, 0.f, 0.f};                                        \
        }                                                                                                \
    }
    if (n_stage > 0) DCS_SLAB_WFETCH(0)
    for (int st = 0; st < n_stage; ++st) {
        float* Wb = Wl + (st & 1) * wstage;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = (tid + q * kColThreads) * 4;
            if (e < wstage) *reinterpret_cast<f32x4*>(Wb + e) = wpre[q];
        }
        __syncthreads();       // also orders the slab fill before the first use; buffer st&1 was last read at st-2
        if (st + 1 < n_stage) DCS_SLAB_WFETCH(st + 1)
        const int u = u_lo + st / nvs, v0 = (st % nvs) * g.tstage;
        const int nt = v0 + g.tstage <= g.kw ? g.tstage : g.kw - v0;
        for (int tv = 0; tv < nt; ++tv) {
            const int v = v0 + tv;
            const float* wp = Wb + tv * 1024 + kq * 32;
            float a0[8], a1[8];
            bool have = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (by[i] < 0) continue;
                const int r = by[i] + u - g.ph;                      // input row (uniform per block)
                const int xs = bx[i] + v - g.pw;                     // shifted column of lane 0
                if (r < 0 || r >= g.H || xs + 15 < 0 || xs >= g.W) continue;
                if (!have) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        a0[kk] = wp[4 * kk * 32 + c0];
                        a1[kk] = wp[4 * kk * 32 + c1];
                    }
                    have = true;
                }
                const int xl = xs + fi;
                const bool ok = xl >= 0 && xl < g.W;
                const float* sp = slab + kq * PS + (r - rbase) * g.W + (ok ? xl : 0);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float bv = sp[4 * kk * PS];
                    const float b = ok ? bv : 0.f;
                    acc0[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kk], b, acc0[i], 0, 0, 0);
                    acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kk], b, acc1[i], 0, 0, 0);
                }
            }
        }
    }
#undef DCS_SLAB_WFETCH
    const int HoWo = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
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

// ------------------------------------------------------------------------------------------------
// The slab convolution on the 16-bit matrix pipe.  One MFMA (16 x 16 x 32) covers a whole tap -- 32 input channels --
// for 16 output channels x 16 columns:
//   MODE 0 (default, f32-class results): both operands split exactly into three bf16 terms (x = hi + mid + lo by
//           truncation), the six term products above 2^-24 accumulated in f32 -- 12 MFMAs of 16 cycles per (tap, block)
//           instead of 16 of 32, with the VALU (splitting the slab values) running beside the matrix pipe;
//   MODE 1 (dcs_model_set_conv_precision(f16)): operands rounded to f16, one product -- 2 MFMAs per (tap, block).
// The slab stays f32 in LDS, channel-fastest [row][x][36] (32 channels + 4 words: a lane's 8 channels are two 16-byte
// reads, 16 consecutive x start in 16 different bank quads); a lane converts its 8 values when it uses them.  The
// weights are packed per model [tap][plane][32 co][4 pieces] (a 16-byte piece = the 8 input channels a lane multiplies)
// and stream through LDS two taps at a time with a row stride of 5 pieces.  Bands, tap / block skipping and the
// accumulator layout are those of slabconv_kernel.
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kSlabCi = 36;                 // floats per (row, x) of the slab
constexpr int kSlabWRow = 5;                // 16-byte pieces per (plane, co) weight row in LDS: 4 + 1 pad

__device__ __forceinline__ unsigned bf_trunc_u(float x) { return __float_as_uint(x) & 0xffff0000u; }

template <int MODE>
__device__ __forceinline__ void slab_convert(const f32x4& x0, const f32x4& x1, u32x4 (&out)[MODE == 0 ? 3 : 1]) {
    if constexpr (MODE == 0) {
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = j < 4 ? x0[j] : x1[j - 4];
            h[j] = bf_trunc_u(x);
            const float r1 = x - __uint_as_float(h[j]);
            m[j] = bf_trunc_u(r1);
            l[j] = bf_trunc_u(r1 - __uint_as_float(m[j]));
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
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)(j < 4 ? x0[j] : x1[j - 4]);
        out[0] = __builtin_bit_cast(u32x4, v);
    }
}

template <int MODE>
__device__ __forceinline__ f32x4 slab_mma(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (MODE == 0)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int MODE, int NW /* waves per workgroup: 8 or 16 */>
__global__ __launch_bounds__(64 * NW) void slabconv_mx_kernel(const SlabConvArgs g, const u32x4* __restrict__ Wq) {
    constexpr int NP = MODE == 0 ? 3 : 1;
    constexpr int NTH = 64 * NW, NBW = 32 / NW;          // threads; (row, 16-column) blocks per wave
    constexpr int kTapLds = NP * 32 * kSlabWRow;          // pieces per tap in LDS
    constexpr int kTapGlb = NP * 32 * 4;                  // pieces per tap in the packed array
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem);           // [2][tstage][NP][32][5]
    const int wstage = g.tstage * kTapLds;                // pieces per weight stage
    float* slab = smem + 2 * wstage * 4;                  // [rows_max][W][36]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    const int64_t img = blockIdx.x / g.n_bands;
    const int y0 = (int)(blockIdx.x - img * g.n_bands) * g.band;
    const int yb = y0 + g.band < g.Ho ? y0 + g.band : g.Ho;   // rows [y0, yb)
    const float* in = g.in + img * g.in_n_stride;
    float* out = g.out + img * g.out_n_stride;
    int rbase = y0 - g.ph, rtop = yb - 1 - g.ph + g.kh - 1;
    if (rbase < 0) rbase = 0;
    if (rtop > g.H - 1) rtop = g.H - 1;
    const int rows = rtop - rbase + 1;
    const int RW = g.rows_max * g.W;
    // slab fill: element (ci, r, x) with x fastest (coalesced reads of the channel-major input); channels >= Cin and
    // rows past the band's reach are zero
    for (int i = tid; i < 32 * RW; i += NTH) {
        const int ci = i / RW, rem = i - ci * RW;
        const int r = rem / g.W;
        slab[rem * kSlabCi + ci] = (ci < g.Cin && r < rows) ? in[((int64_t)ci * g.H + rbase + r) * g.W + (rem - r * g.W)] : 0.f;
    }
    const int nxb = (g.Wo + 15) >> 4, nblk = (yb - y0) * nxb;
    int by[NBW], bx[NBW];
    f32x4 acc0[NBW], acc1[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int b = wave + NW * i;
        by[i] = b < nblk ? y0 + b / nxb : -1;
        bx[i] = b < nblk ? (b % nxb) * 16 : 0;
        acc0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nvs = (g.kw + g.tstage - 1) / g.tstage;
    int u_lo = g.ph - (yb - 1), u_hi = g.ph - y0 + g.H - 1;
    if (u_lo < 0) u_lo = 0;
    if (u_hi > g.kh - 1) u_hi = g.kh - 1;
    const int n_stage = (u_hi - u_lo + 1) * nvs;
    // weight staging: a stage is tstage * NP * 128 <= 768 pieces
    constexpr int WPRE = NW == 8 ? 2 : 1;
    u32x4 wpre[WPRE];
#define DCS_SLABQ_WFETCH(st_)                                                                           \
    {                                                                                                   \
        const int u_ = u_lo + (st_) / nvs, v_ = ((st_) % nvs) * g.tstage;                               \
        const int nt_ = v_ + g.tstage <= g.kw ? g.tstage : g.kw - v_;                                    \
        _Pragma("unroll") for (int q = 0; q < WPRE; ++q) {                                              \
            const int e = tid + q * NTH;                                                        \
            wpre[q] = e < nt_ * kTapGlb ? Wq[(int64_t)(u_ * g.kw + v_) * kTapGlb + e] : u32x4{0u, 0u, 0u, 0u}; \
        }                                                                                               \
    }
    if (n_stage > 0) DCS_SLABQ_WFETCH(0)
    for (int st = 0; st < n_stage; ++st) {
        u32x4* Wb = Wl + (st & 1) * wstage;
#pragma unroll
        for (int q = 0; q < WPRE; ++q) {
            const int e = tid + q * NTH;           // (tap, plane * 32 + co, kq) in the packed order
            if (e < g.tstage * kTapGlb) Wb[(e >> 2) * kSlabWRow + (e & 3)] = wpre[q];
        }
        __syncthreads();       // also orders the slab fill before the first use; buffer st&1 was last read at st-2
        if (st + 1 < n_stage) DCS_SLABQ_WFETCH(st + 1)
        const int u = u_lo + st / nvs, v0 = (st % nvs) * g.tstage;
        const int nt = v0 + g.tstage <= g.kw ? g.tstage : g.kw - v0;
        for (int tv = 0; tv < nt; ++tv) {
            const int v = v0 + tv;
            const u32x4* wp = Wb + tv * kTapLds + fi * kSlabWRow + kq;
            u32x4 a0[NP], a1[NP];
            bool have = false;
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                if (by[i] < 0) continue;
                const int r = by[i] + u - g.ph;                      // input row (uniform per block)
                const int xs = bx[i] + v - g.pw;                     // shifted column of lane 0
                if (r < 0 || r >= g.H || xs + 15 < 0 || xs >= g.W) continue;
                if (!have) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        a0[p] = wp[(p * 32) * kSlabWRow];
                        a1[p] = wp[(p * 32 + 16) * kSlabWRow];
                    }
                    have = true;
                }
                const int xl = xs + fi;
                const bool ok = xl >= 0 && xl < g.W;
                const float* sp = slab + ((r - rbase) * g.W + (ok ? xl : 0)) * kSlabCi + 8 * kq;
                f32x4 x0 = *reinterpret_cast<const f32x4*>(sp), x1 = *reinterpret_cast<const f32x4*>(sp + 4);
                if (!ok) {
                    x0 = f32x4{0.f, 0.f, 0.f, 0.f};
                    x1 = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                u32x4 b[NP];
                slab_convert<MODE>(x0, x1, b);
                if constexpr (MODE == 0) {   // smallest terms first
                    acc0[i] = slab_mma<0>(a0[2], b[0], acc0[i]);
                    acc1[i] = slab_mma<0>(a1[2], b[0], acc1[i]);
                    acc0[i] = slab_mma<0>(a0[0], b[2], acc0[i]);
                    acc1[i] = slab_mma<0>(a1[0], b[2], acc1[i]);
                    acc0[i] = slab_mma<0>(a0[1], b[1], acc0[i]);
                    acc1[i] = slab_mma<0>(a1[1], b[1], acc1[i]);
                    acc0[i] = slab_mma<0>(a0[1], b[0], acc0[i]);
                    acc1[i] = slab_mma<0>(a1[1], b[0], acc1[i]);
                    acc0[i] = slab_mma<0>(a0[0], b[1], acc0[i]);
                    acc1[i] = slab_mma<0>(a1[0], b[1], acc1[i]);
                    acc0[i] = slab_mma<0>(a0[0], b[0], acc0[i]);
                    acc1[i] = slab_mma<0>(a1[0], b[0], acc1[i]);
                } else {
                    acc0[i] = slab_mma<1>(a0[0], b[0], acc0[i]);
                    acc1[i] = slab_mma<1>(a1[0], b[0], acc1[i]);
                }
            }
        }
    }
#undef DCS_SLABQ_WFETCH
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

// ------------------------------------------------------------------------------------------------
// The column convolution with f16 inputs and f32 accumulation (v_mfma_f32_16x16x32_f16): one MFMA covers a whole
// tap (32 input channels) for a 16 co x 16 x block, 16x the f32 rate, so this variant is LDS- and store-bound.
// Slab and weights are stored channel-fastest -- slab [row][x][ci], weights [u][co][ci], rows of 32 halves padded to
// 40 (80 bytes, the bank-friendly stride of conv_igemm_f16_kernel) -- so both operands are one 16-byte LDS read per
// lane.  Activations are rounded to f16 when the slab is written, weights are pre-rounded on the host.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kColThreads) void colconv_f16_kernel(const ColConvArgs g, const _Float16* __restrict__ Wh) {
    constexpr int RS = 40;                                   // halves per (row, x) / (u, co) line
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Wl = reinterpret_cast<_Float16*>(smem);         // [kh][32 co][RS]
    _Float16* slab = Wl + g.kh * 32 * RS;                      // [H][16 x][RS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kg = lane >> 4;
    const int groups = (g.n_xb + g.xb_per_wg - 1) / g.xb_per_wg;
    const int64_t img = blockIdx.x / groups;
    const int xb0 = (int)(blockIdx.x - img * groups) * g.xb_per_wg;
    const float* in = g.in + img * g.in_n_stride;
    float* out = g.out + img * g.out_n_stride;
    const int HoW = g.Ho * g.W;

    for (int i = tid; i < g.kh * 32 * RS / 8; i += kColThreads)
        reinterpret_cast<f32x4*>(Wl)[i] = reinterpret_cast<const f32x4*>(Wh)[i];
    for (int i = tid; i < g.H * 16 * RS / 2; i += kColThreads) reinterpret_cast<float*>(slab)[i] = 0.f;  // pad channels
    const float bias_lo[4] = {g.bias[4 * kg], g.bias[4 * kg + 1], g.bias[4 * kg + 2], g.bias[4 * kg + 3]};
    const float bias_hi[4] = {g.bias[16 + 4 * kg], g.bias[17 + 4 * kg], g.bias[18 + 4 * kg], g.bias[19 + 4 * kg]};

    constexpr int kPre = 32;
    const int n_el = g.Cin * g.H * 16;
    int soff[kPre];
    float pre[kPre];
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
        const int e = tid + i * kColThreads;
        const int cr = e >> 4, ci = cr / g.H;
        soff[i] = e < n_el ? ((cr - ci * g.H) * 16 + (e & 15)) * RS + ci : -1;
    }
#define DCS_COL_FETCH(x0_)                                                                               \
    _Pragma("unroll") for (int i = 0; i < kPre; ++i) {                                                   \
        const int e = tid + i * kColThreads;                                                             \
        pre[i] = (soff[i] >= 0 && (x0_) + (e & 15) < g.W) ? in[(int64_t)(e >> 4) * g.W + (x0_) + (e & 15)] : 0.f; \
    }
    const int xb_end = xb0 + g.xb_per_wg < g.n_xb ? xb0 + g.xb_per_wg : g.n_xb;
    if (xb0 < xb_end) DCS_COL_FETCH(xb0 * 16)
    for (int xb = xb0; xb < xb_end; ++xb) {
        const int x0 = xb * 16;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kPre; ++i)
            if (soff[i] >= 0) slab[soff[i]] = (_Float16)pre[i];
        __syncthreads();
        if (xb + 1 < xb_end) DCS_COL_FETCH((xb + 1) * 16)
        for (int y = wave; y < g.Ho; y += kColThreads / 64) {
            int u_lo = g.ph - y, u_hi = g.ph - y + g.H - 1;
            if (u_lo < 0) u_lo = 0;
            if (u_hi > g.kh - 1) u_hi = g.kh - 1;
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int u = u_lo; u <= u_hi; ++u) {
                const h8 b = *reinterpret_cast<const h8*>(slab + ((y + u - g.ph) * 16 + fi) * RS + kg * 8);
                const h8 a0 = *reinterpret_cast<const h8*>(Wl + (u * 32 + fi) * RS + kg * 8);
                const h8 a1 = *reinterpret_cast<const h8*>(Wl + (u * 32 + 16 + fi) * RS + kg * 8);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, acc1, 0, 0, 0);
            }
            if (x0 + fi < g.W) {
                float* op = out + (int64_t)y * g.W + x0 + fi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = 4 * kg + e;
                    if (co < g.Cout) op[(int64_t)co * HoW] = acc0[e] + bias_lo[e];
                    if (co + 16 < g.Cout) op[(int64_t)(co + 16) * HoW] = acc1[e] + bias_hi[e];
                }
            }
        }
    }
#undef DCS_COL_FETCH
}

// ------------------------------------------------------------------------------------------------
// VJP of conv1 for a frequency stride that divides 16 (Bach10 / score-informed: kw = 30, stride 4), register-blocked.
// With f = sw q + r the sum is  o[f] = sum_o' sum_mm g[o'][q - mm] * W[o'][sw mm + r],  mm < ceil(kw / sw): a thread
// owns 16 consecutive f (16/sw values of q, all r) of one row, keeps the 16 sums and the 16/sw + ceil(kw/sw) - 1 inputs
// of the current channel in registers, and reads the filter row of that channel -- the same for every thread -- with
// scalar loads (the filter is padded to sw*ceil(kw/sw) taps so the unrolled loop needs no tap test).  128 multiply-adds
// per 11 vector loads for the Bach10 shape; the direct kernel below does one LDS read per operand (2 per
// multiply-add) and is LDS-bandwidth bound: 1.3 ms -> see DESIGN.md for 167 tiles.
// ------------------------------------------------------------------------------------------------
// POOLED (SW = 3, NT = 10: iKala): the un-pooling VJP fused in.  g is the POOLED gradient [m][NF][tc][wp] and maskw the routing
// bits conv1_reg_kernel<., ., true> left (rows [m / nb][NF][tc][mw]): a thread's 13 inputs j0 .. j0 + 12 start at position 3
// of window k = qb - 3 and cover windows k + 1 .. k + 3 whole, so it loads 4 pooled values (16 bytes) and 16 routing bits (one
// 8-byte load) per channel instead of 13 un-pooled values -- and the un-pooled tensor (4x the bytes, written and read once) and
// its kernel do not exist.
template <int SW, int NT /* taps per residue = ceil(kw / SW) */, bool POOLED = false>
__global__ __launch_bounds__(kThreads) void deconv1_reg_kernel(const float* __restrict__ g, const float* __restrict__ Wp,
                                                               float* __restrict__ out, int NF, int C, int tc, int F,
                                                               int w1, int nqb, const unsigned* __restrict__ maskw = nullptr,
                                                               int wp = 0, int mw = 0, int nb = 1) {
    // a thread owns QB values of q and all SW residues: FB = SW * QB consecutive bins (16 for a stride of 4, 12 for the
    // iKala stride of 3, whose 30 taps are exactly 10 per residue)
    constexpr int QB = 4, FB = SW * QB, XI = QB + NT - 1;
    const int64_t m = blockIdx.y;
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= tc * nqb) return;
    const int t = idx / nqb, qb = idx - t * nqb;
    const int q0 = qb * QB, j0 = q0 - (NT - 1);            // inputs j0 .. j0 + XI - 1
    static_assert(!POOLED || (QB == 4 && (NT - 1) % 4 == 1), "pooled input: j0 = 4 (qb - (NT + 2) / 4) + 3");
    const float* grow = g + (m * NF * tc + t) * (int64_t)(POOLED ? wp : w1);
    const unsigned* mrow = POOLED ? maskw + ((m / nb) * NF * tc + t) * (int64_t)mw : nullptr;
    const int k = qb - (NT + 2) / 4;                       // POOLED: first window touched (position 3 of it only)
    for (int c = 0; c < C; ++c) {
        float acc[FB];
#pragma unroll
        for (int i = 0; i < FB; ++i) acc[i] = 0.f;
        for (int o = 0; o < NF; ++o) {
            const float* gp = grow + (int64_t)o * tc * (POOLED ? wp : w1);
            float gw[XI];
            constexpr int NV = (XI + 3) / 4;
            if (POOLED) {
                const unsigned* mp = mrow + (int64_t)o * tc * mw;
                float gv[4];
                unsigned bits;
                if (k >= 0 && k + 3 < wp) {               // interior: one 16-byte and one 8-byte load
                    const f32x4 v = *reinterpret_cast<const f32x4*>(gp + k);
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[i] = v[i];
                    const unsigned long long w64 = *reinterpret_cast<const unsigned long long*>(mp + (k >> 3));
                    bits = (unsigned)(w64 >> (4 * (k & 7))) & 0xffffu;
                } else {
                    bits = 0u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int w = k + i;
                        const bool in = w >= 0 && w < wp;
                        gv[i] = in ? gp[w] : 0.f;
                        if (in) bits |= ((mp[w >> 3] >> (4 * (w & 7))) & 15u) << (4 * i);
                    }
                }
                // input x of the thread = position (x + 3) & 3 of window (x + 3) >> 2 (relative to k)
#pragma unroll
                for (int x = 0; x < XI; ++x) gw[x] = ((bits >> (x + 3)) & 1u) ? gv[(x + 3) >> 2] : 0.f;
            } else if (j0 >= 0 && j0 + 4 * NV <= w1) {           // interior: three or four (unaligned) 16-byte loads
                f32x4 v[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q] = *reinterpret_cast<const f32x4*>(gp + j0 + 4 * q);
#pragma unroll
                for (int x = 0; x < XI; ++x) gw[x] = v[x >> 2][x & 3];
            } else {
#pragma unroll
                for (int x = 0; x < XI; ++x) {
                    const int j = j0 + x;
                    gw[x] = (j >= 0 && j < w1) ? gp[j] : 0.f;
                }
            }
            const float* wr = Wp + ((int64_t)o * C + c) * (SW * NT);   // uniform: scalar loads
#pragma unroll
            for (int mm = 0; mm < NT; ++mm)
#pragma unroll
                for (int r = 0; r < SW; ++r) {
                    const float w = wr[SW * mm + r];
#pragma unroll
                    for (int i = 0; i < QB; ++i) acc[i * SW + r] = fmaf(gw[i + (NT - 1) - mm], w, acc[i * SW + r]);
                }
        }
        float* op = out + ((m * C + c) * tc + t) * (int64_t)F + (int64_t)q0 * SW;
#pragma unroll
        for (int i = 0; i < FB; ++i)
            if (q0 * SW + i < F) op[i] = acc[i];
    }
}

// ------------------------------------------------------------------------------------------------
// VJP of conv1: o[m, c, t, f] = sum_{o', j : 0 <= f - j*sw < kw} g[m, o', t, j] * Wc[o', c, f - j*sw]
// block = 256 consecutive f of one (m, t) row, all C output channels.
// ------------------------------------------------------------------------------------------------
template <int NF>
__global__ __launch_bounds__(kThreads) void deconv1_kernel(const float* __restrict__ g, const float* __restrict__ Wc,
                                                           float* __restrict__ out, int C, int tc, int F, int kw, int sw,
                                                           int w1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                         // [NF][C][kw]
    const int span = kThreads / sw + kw / sw + 3;
    float* gs = smem + NF * C * kw;           // [NF][span]
    const int tid = threadIdx.x;
    const int64_t mt = blockIdx.y;            // m*tc + t
    const int64_t mrow = mt / tc;
    const int t = (int)(mt - mrow * tc);
    const int f0 = blockIdx.x * kThreads;
    // j range feeding f in [f0, f0+256): j >= ceil((f0-kw+1)/sw), j <= (f0+255)/sw
    int jlo = f0 - kw + 1;
    jlo = jlo <= 0 ? 0 : (jlo + sw - 1) / sw;
    for (int i = tid; i < NF * C * kw; i += kThreads) Ws[i] = Wc[i];
    for (int i = tid; i < NF * span; i += kThreads) {
        const int o = i / span, jj = i - o * span;
        const int j = jlo + jj;
        gs[i] = (j < w1) ? g[((mrow * NF + o) * tc + t) * (int64_t)w1 + j] : 0.f;
    }
    __syncthreads();
    const int f = f0 + tid;
    if (f >= F) return;
    int ja = f - kw + 1;
    ja = ja <= 0 ? 0 : (ja + sw - 1) / sw;
    int jb = f / sw;
    if (jb > w1 - 1) jb = w1 - 1;
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        for (int j = ja; j <= jb; ++j) {
            const int u = f - j * sw;
#pragma unroll 6
            for (int o = 0; o < NF; ++o) acc = fmaf(gs[o * span + (j - jlo)], Ws[(o * C + c) * kw + u], acc);
        }
        out[((mrow * C + c) * tc + t) * (int64_t)F + f] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// concat + BiasLayer + rectify + soft mask (separate_ikala.py:190-217, separate_bach10.py:229-264).
// o: [n, CH, tc, F] with CH = branches*C; masks use channels 0..S-1, the mixture is input channel 0 (every separate script) or
// the sum of the first mix_n = C input channels (score-informed trainers, trainCNNrwc.py:258-263).
// mode 0/1: out[s][n][t][f]; mode 2: p[ch][n][t][f] for all CH channels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void mask_kernel(const float* __restrict__ o, const float* __restrict__ bias,
                                                        const float* __restrict__ x, float* __restrict__ out,
                                                        int64_t n, int CH, int S, int C, int64_t plane /* tc*F */,
                                                        int mode, int mix_n /* input channels the mixture adds up: 1 or C */) {
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= n * plane) return;
    const int64_t k = idx / plane, r = idx - k * plane;
    const float eps_r = 5e-19f;
    if (mode == 2) {
        for (int ch = 0; ch < CH; ++ch)
            out[((int64_t)ch * n + k) * plane + r] = fmaxf(o[(k * CH + ch) * plane + r] + bias[ch], 0.f);
        return;
    }
    float p[4];
    float den = 0.f;
    for (int s = 0; s < S; ++s) {
        p[s] = fmaxf(o[(k * CH + s) * plane + r] + bias[s], 0.f);
        if (mode == 0) p[s] += eps_r;
        den = (s == 0) ? p[s] : den + p[s];
    }
    if (mode == 1) den += eps_r;
    float mix = x[(k * C) * plane + r];
    for (int c = 1; c < mix_n; ++c) mix += x[(k * C + c) * plane + r];   // the trainers' mixture: channels added left to right
    for (int s = 0; s < S; ++s) out[((int64_t)s * n + k) * plane + r] = (p[s] / den) * mix;
}

// ------------------------------------------------------------------------------------------------
// mask_kernel + overlap_add_kernel (tiling.hip) in one pass for the whole-path entry point: the masked tiles
// [S][n][tc][F] (164 MB per 10 s of Bach10, written once and read once) never exist.  Output frame t is owned by the last
// tile k0 whose copy region contains it and blended with the later tiles that reach it, in increasing k -- the operands
// and the order of overlap_add_kernel -- and every tile value is computed where it is consumed with the expressions of
// mask_kernel: the result is bit-identical to the two kernels.  Each (tile, frame, bin) of o is read exactly once.
// o: [n_all][CH][tc][F] raw decoder output, x: the tiles [n_all][C][tc][F]; this clip's tiles start at k_off.
// ------------------------------------------------------------------------------------------------
// MM > 0 (the frame's covering tiles are at most MM, S == 4): the tiles that reach a frame are the same for every thread of the
// workgroup, so all MM x 5 values are REQUESTED before the first is used -- the loop form below asks for one tile's five
// values, waits, blends, and only then asks for the next tile's (2.6 TB/s on the Bach10 clip).
template <int MM>
__global__ __launch_bounds__(kThreads) void mask_ola_kernel(const float* __restrict__ o, const float* __restrict__ bias,
                                                            const float* __restrict__ x, int64_t n, int64_t k_off, int CH,
                                                            int S, int C, int tc, int ov, int F,
                                                            const float* __restrict__ rise, float* __restrict__ sep,
                                                            int64_t sep_stride, int64_t ld, int mode, int mix_n) {
    const int64_t t = blockIdx.x;
    const int st = tc - ov;
    const int64_t plane = (int64_t)tc * F;
    int64_t k0 = (t < ov) ? 0 : (t - ov) / st;
    if (k0 > n - 1) k0 = n - 1;
    const int64_t j0 = (n > 0) ? t - k0 * st : tc;
    const float eps_r = 5e-19f;
    const int f = blockIdx.y * kThreads + threadIdx.x;   // one (frame, bin) per thread: enough waves to cover the latency
    if (MM > 0) {
        if (f >= F) return;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (j0 < tc) {
            constexpr int MA = MM > 0 ? MM : 1;
            float v[MA][8];                                            // [0..3] raw outputs, [4..7] input channels of the mixture
            bool valid[MA];
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const int64_t k = k0 + m;
                valid[m] = k < n && k * st <= t;                       // workgroup-uniform
                const int64_t kc = valid[m] ? k : k0;                  // clamped: every load is issued, none sits behind a branch
                const int64_t r = (t - kc * st) * F + f;
#pragma unroll
                for (int s = 0; s < 4; ++s) v[m][s] = o[((k_off + kc) * CH + s) * plane + r];
#pragma unroll
                for (int c = 0; c < 4; ++c)                            // mix_n is 1 or C <= 4, uniform over the launch
                    v[m][4 + c] = c < mix_n ? x[((k_off + kc) * C + c) * plane + r] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                if (!valid[m]) continue;
                const int j = (int)(t - (k0 + m) * st);
                float p[4];
                float den = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    p[s] = fmaxf(v[m][s] + bias[s], 0.f);
                    if (mode == 0) p[s] += eps_r;
                    den = (s == 0) ? p[s] : den + p[s];
                }
                if (mode == 1) den += eps_r;
                float mix = v[m][4];
#pragma unroll
                for (int c = 1; c < 4; ++c)
                    if (c < mix_n) mix += v[m][4 + c];
                if (m == 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[s] = (p[s] / den) * mix;
                } else {
                    const float down = rise[ov - 1 - j], up = rise[j];
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[s] = fmaf(down, acc[s], __fmul_rn(up, (p[s] / den) * mix));   // one explicit form: see overlap_add_kernel
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) sep[s * sep_stride + t * ld + f] = acc[s];
        return;
    }
    if (f < F) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (j0 < tc) {                                   // else: past the last tile, the zeros of util.py:313
            for (int64_t k = k0; k < n && k * st <= t; ++k) {
                const int j = (int)(t - k * st);
                const int64_t r = (int64_t)j * F + f;
                float p[4];
                float den = 0.f;
                for (int s = 0; s < S; ++s) {
                    p[s] = fmaxf(o[((k_off + k) * CH + s) * plane + r] + bias[s], 0.f);
                    if (mode == 0) p[s] += eps_r;
                    den = (s == 0) ? p[s] : den + p[s];
                }
                if (mode == 1) den += eps_r;
                float mix = x[((k_off + k) * C) * plane + r];
                for (int c = 1; c < mix_n; ++c) mix += x[((k_off + k) * C + c) * plane + r];
                if (k == k0) {
                    for (int s = 0; s < S; ++s) acc[s] = (p[s] / den) * mix;
                } else {
                    const float down = rise[ov - 1 - j], up = rise[j];
                    for (int s = 0; s < S; ++s) acc[s] = fmaf(down, acc[s], __fmul_rn(up, (p[s] / den) * mix));   // one explicit form: see overlap_add_kernel
                }
            }
        }
        for (int s = 0; s < S; ++s) sep[s * sep_stride + t * ld + f] = acc[s];
    }
}

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
    DCS_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
    DCS_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return DCS_OK;
}

}  // namespace

struct DcsGenericNet {
    dcs_ctx* ctx = nullptr;
    DcsGenericDims d;
    int C = 1, tc = 30, F = 0;
    int flat_p = 0, flat64 = 0, hid64 = 0;
    // conv1 / deconv1
    float *W1c = nullptr, *bias1 = nullptr;
    float* W1t = nullptr;      // conv1 filters [c][u][32] (filters fastest) for conv1_reg_kernel
    // conv2 as implicit GEMM
    float *W2m = nullptr, *bias2 = nullptr;
    int *k2off = nullptr, *k2uv = nullptr;
    int K2 = 0;
    // transposed conv2
    float *W2t = nullptr, *bias0 = nullptr;
    int *kt_off = nullptr, *kt_uv = nullptr;
    // f16 copies of the two conv2 weight matrices, packed [k tile][32 channels][32 k]
    _Float16 *W2m_h = nullptr, *W2t_h = nullptr;
    int conv_f16 = 0;
    // column convolution (kw2 == 1): weights [kh][32 ci][32 co swizzled] of conv2 and of its transpose
    float *Wslab = nullptr, *Wslab_t = nullptr;   // [kh][kw][32][32] conv2 / its transpose for slabconv_kernel
    // the same for slabconv_mx_kernel: [tap][plane][32 co][4 pieces of 8 ci]; q3 = three bf16 planes, h = one f16 plane
    uint16_t *Wslab_q3 = nullptr, *Wslab_t_q3 = nullptr, *Wslab_h = nullptr, *Wslab_t_h = nullptr;
    uint16_t *Wps_q3 = nullptr, *Wps_t_q3 = nullptr, *Wps_h = nullptr, *Wps_t_h = nullptr;   // slabconv_ps.hip orders
    uint16_t *Wpc_q3 = nullptr, *Wpc_t_q3 = nullptr;     // the kh x 1 filters in the same orders (bf16 x 3)
    int use_slabconv = 0;
    float* W1p = nullptr;      // conv1 filters padded to sw1*ceil(kw1/sw1) taps (register-blocked transpose of conv1)
    float *Wcol = nullptr, *Wcol_t = nullptr;
    _Float16 *Wcol_h = nullptr, *Wcol_t_h = nullptr;     // [kh][32 co][40] halves, channel-fastest
    _Float16 *Wcol_r = nullptr, *Wcol_t_r = nullptr;     // the same weights as MFMA fragments (colconv_wreg.hip)
    uint16_t* W1q = nullptr;                             // padded conv1 filter as bf16 x 3 fragments (fused decoder)
    uint16_t* Wx3 = nullptr;                             // transposed conv2 filter as bf16 x 3 fragments, taps dealt by parity (colconv_x3.hip)
    uint16_t* Wfx3 = nullptr;                            // the forward conv2 filter in the same fragment format (colconv_fwd_x3.hip)
    uint16_t* W1m = nullptr;                             // conv1 filter as bf16 x 3 fragments (conv1_mfma.hip)
    uint16_t* W1dq = nullptr;                            // the same for its InverseLayer (deconv1_mfma.hip)
    int use_colconv = 0;
    // dense
    float *Bfc = nullptr, *biasfc = nullptr;
    // whole-path calls on ONE clip: the clip's scaled frames [C][frames_rows][F], tile k = rows k * frames_st .. + tc (set around
    // dcs_generic_forward by the separate path, null otherwise): conv1 is a per-row operation, so it runs once per FRAME
    const float* frames_src = nullptr;
    int frames_rows = 0, frames_st = 0;
    float *B2fc = nullptr, *bias2fc = nullptr;          // conv2 + bias + bottleneck layer as one affine map of conv2's input (fold_conv2_fc_kernel)
    int K2fc = 0;                                        // its K = nf1 * tc * wp (0: not folded)
    void* Bfcq = nullptr;                                // the bottleneck weights as bf16 x 3 planes (long-K launches of 128 .. 176 rows), on first need
    bool bfcq_failed = false;
    float* Bd[4] = {nullptr, nullptr, nullptr, nullptr};
    void* Bdq[4] = {nullptr, nullptr, nullptr, nullptr};  // per-source dense weights as bf16 planes (gemm_bf16x3.hip)
    bool bdq_failed = false;                              // the planes did not fit in memory: the dense layers stay on the f32 GEMM
    float* biasd[4] = {nullptr, nullptr, nullptr, nullptr};
    float* biasd_cl[4] = {nullptr, nullptr, nullptr, nullptr};   // the same biases in [position][channel] order (channels-last D)
    bool bdq_cl = false;                                         // column order the bf16 planes Bdq are packed in
    // the same weights UNSPLIT in the planes' piece order (dcs_gemm_pack_b32: 4 bytes per weight instead of 6) for passes of
    // 128 .. 176 tiles, which run the all-rows kernel: it splits the pieces in registers (round 6); Bdq serves the other row counts
    void* Bd32[4] = {nullptr, nullptr, nullptr, nullptr};
    bool bd32_cl = false;
    // f16 switch with the fused decoder (gemm_f16.hip): the same weights as ONE f16 plane, columns [position][32 channels]
    // (30 real), their biases in that order; n_out16 columns per branch (a multiple of 128); made on first need
    void* Bfch = nullptr;                                        // the bottleneck weights as one f16 plane (long-K launches under the switch)
    bool bfch_failed = false;
    void* Bdh[4] = {nullptr, nullptr, nullptr, nullptr};
    float* biasd_h[4] = {nullptr, nullptr, nullptr, nullptr};
    int n_out16 = 0;
    bool bdh_failed = false;
    float* bout = nullptr;
    DcsBuffer ws;
    float* rise_d = nullptr;
    int rise_ov = -1;
    const float* raw_o = nullptr;        // decoder output of the last deferred-mask pass (scratch of ws), its channel count
    int raw_ch = 0;
    // score-informed graphs: the reference's two semantics (dcs_model_set_score_semantics)
    int score_norm = DCS_SCORE_NORM_MAX;  // harmonic masks / own maximum (script) or / sum over instruments (trainers)
    int mix_sum = 0;                      // soft masks x input channel 0 (script) or x the channel sum (trainers)
};

int dcs_generic_create(dcs_ctx* ctx, const DcsGenericDims& d, int C, int tc, int F,
                       const std::vector<std::vector<float>>& P, DcsGenericNet** out) {
    if (d.nf1 != 30 || d.nf2 != 30) DCS_FAIL(DCS_EUNSUPPORTED, "generic path is built for 30-filter layers");
    if (d.n_fc > 4 || d.S > 4) DCS_FAIL(DCS_EUNSUPPORTED, "generic path: at most 4 branches");
    DcsGenericNet* g = new DcsGenericNet();
    g->ctx = ctx;
    g->d = d;
    g->C = C;
    g->tc = tc;
    g->F = F;
    g->flat_p = (int)dcs_round_up(d.flat, 4);
    g->flat64 = (int)dcs_round_up(d.flat, 64);
    g->hid64 = (int)dcs_round_up(d.hidden, 64);
    const float *W1 = P[0].data(), *b1 = P[1].data(), *b1b = P[2].data();
    const float *W2 = P[3].data(), *b2 = P[4].data(), *b2b = P[5].data();
    const int kw1 = d.kw1, kh = d.kh2, kw = d.kw2, nf1 = d.nf1, nf2 = d.nf2;
    // conv1: Wc[o][c][u] = W1[o][c][0][kw1-1-u]
    std::vector<float> W1c((size_t)nf1 * C * kw1), bias1(nf1);
    for (int o = 0; o < nf1; ++o) {
        for (int c = 0; c < C; ++c)
            for (int u = 0; u < kw1; ++u) W1c[((size_t)o * C + c) * kw1 + u] = W1[((size_t)o * C + c) * kw1 + (kw1 - 1 - u)];
        bias1[o] = b1[o] + b1b[o];
    }
    std::vector<float> W1t((size_t)C * kw1 * 32, 0.f);
    for (int o = 0; o < nf1; ++o)
        for (int c = 0; c < C; ++c)
            for (int u = 0; u < kw1; ++u) W1t[((size_t)c * kw1 + u) * 32 + o] = W1c[((size_t)o * C + c) * kw1 + u];
    // the same filters with the tap axis zero-padded to a multiple of the stride (deconv1_reg_kernel)
    std::vector<float> W1p;
    if (16 % d.sw1 == 0 || d.sw1 == 3) {
        const int ntap = d.sw1 * ((kw1 + d.sw1 - 1) / d.sw1);
        W1p.assign((size_t)nf1 * C * ntap, 0.f);
        for (int o = 0; o < nf1; ++o)
            for (int c = 0; c < C; ++c)
                for (int u = 0; u < kw1; ++u) W1p[((size_t)o * C + c) * ntap + u] = W1c[((size_t)o * C + c) * kw1 + u];
    }
    // conv2: k = (ci, u, v); Wm[k][co] = W2[co][ci][kh-1-u][kw-1-v]; input plane = [tc, wp]
    const int K2 = nf1 * kh * kw;
    const int K2p = (int)dcs_round_up(K2, 32);
    g->K2 = K2;
    std::vector<float> W2m((size_t)K2p * 32, 0.f), bias2(32, 0.f), W2t((size_t)K2p * 32, 0.f), bias0(32, 0.f);
    std::vector<int> k2off(K2p, 0), k2uv(K2p, 0), kt_off(K2p, 0), kt_uv(K2p, 0);
    for (int ci = 0; ci < nf1; ++ci)
        for (int u = 0; u < kh; ++u)
            for (int v = 0; v < kw; ++v) {
                const int k = (ci * kh + u) * kw + v;
                k2off[k] = ci * tc * d.wp + u * d.wp + v;
                k2uv[k] = (u << 16) | v;
                for (int co = 0; co < nf2; ++co)
                    W2m[(size_t)k * 32 + co] = W2[(((size_t)co * nf1 + ci) * kh + (kh - 1 - u)) * kw + (kw - 1 - v)];
            }
    for (int co = 0; co < nf2; ++co) bias2[co] = b2[co] + b2b[co];
    // transposed conv2 = 'full' correlation of d[n, co, h2, w2] with Wt[(co,u,v)][ci] = W2[co][ci][u][v]
    // K ordered tap-row major, k = (u, co, v): see igemm_k_range
    for (int co = 0; co < nf2; ++co)
        for (int u = 0; u < kh; ++u)
            for (int v = 0; v < kw; ++v) {
                const int k = (u * nf2 + co) * kw + v;
                kt_off[k] = co * d.h2 * d.w2 + u * d.w2 + v;
                kt_uv[k] = (u << 16) | v;
                for (int ci = 0; ci < nf1; ++ci)
                    W2t[(size_t)k * 32 + ci] = W2[(((size_t)co * nf1 + ci) * kh + u) * kw + v];
            }
    // f16 packing of both matrices (round-to-nearest-even from the f32 weights)
    auto pack_h = [&](const std::vector<float>& Wm) {
        std::vector<_Float16> h((size_t)K2p * 32);
        for (int kt = 0; kt < K2p / 32; ++kt)
            for (int c = 0; c < 32; ++c)
                for (int kk = 0; kk < 32; ++kk)
                    h[((size_t)kt * 32 + c) * 32 + kk] = (_Float16)Wm[(size_t)(kt * 32 + kk) * 32 + c];
        return h;
    };
    const std::vector<_Float16> W2m_h = pack_h(W2m), W2t_h = pack_h(W2t);
    // slab convolution (general kh x kw): Wslab[u][v][ci][co] = W2[co][ci][kh-1-u][kw-1-v]; transpose
    // Wslab_t[u][v][co][ci] = W2[co][ci][u][v]
    std::vector<float> Wslab, Wslab_t;
    static const int slab_env = getenv("DCS_SLABCONV") ? atoi(getenv("DCS_SLABCONV")) : 1;
    g->use_slabconv = (kw > 1 && nf1 <= 32 && nf2 <= 32 && slab_env) ? 1 : 0;
    if (g->use_slabconv) {
        Wslab.assign((size_t)kh * kw * 1024, 0.f);
        Wslab_t.assign((size_t)kh * kw * 1024, 0.f);
        for (int co = 0; co < nf2; ++co)
            for (int ci = 0; ci < nf1; ++ci)
                for (int u = 0; u < kh; ++u)
                    for (int v = 0; v < kw; ++v) {
                        Wslab[(size_t)(u * kw + v) * 1024 + colconv_wslot(0, ci, co)] =
                            W2[(((size_t)co * nf1 + ci) * kh + (kh - 1 - u)) * kw + (kw - 1 - v)];
                        Wslab_t[(size_t)(u * kw + v) * 1024 + colconv_wslot(0, co, ci)] =
                            W2[(((size_t)co * nf1 + ci) * kh + u) * kw + v];
                    }
    }
    // 16-bit packings of both slab weight sets: piece (tap, plane, co, kq) = input channels 8 kq .. 8 kq + 7
    std::vector<uint16_t> Wslab_q3, Wslab_t_q3, Wslab_h, Wslab_t_h;
    if (g->use_slabconv) {
        auto pack = [&](const std::vector<float>& Wf, std::vector<uint16_t>& q3, std::vector<uint16_t>& h1) {
            const int taps = kh * kw;
            q3.assign((size_t)taps * 3 * 32 * 32, 0);
            h1.assign((size_t)taps * 32 * 32, 0);
            for (int t = 0; t < taps; ++t)
                for (int co = 0; co < 32; ++co)
                    for (int ci = 0; ci < 32; ++ci) {
                        float r = Wf[(size_t)t * 1024 + colconv_wslot(0, ci, co)];
                        const _Float16 hv = (_Float16)r;
                        uint16_t hb;
                        memcpy(&hb, &hv, 2);
                        h1[((size_t)t * 32 + co) * 32 + ci] = hb;
                        for (int pl = 0; pl < 3; ++pl) {
                            uint32_t bits;
                            memcpy(&bits, &r, 4);
                            bits &= 0xffff0000u;
                            float part;
                            memcpy(&part, &bits, 4);
                            r -= part;
                            q3[(((size_t)t * 3 + pl) * 32 + co) * 32 + ci] = (uint16_t)(bits >> 16);
                        }
                    }
        };
        pack(Wslab, Wslab_q3, Wslab_h);
        pack(Wslab_t, Wslab_t_q3, Wslab_t_h);
    }
    std::vector<uint16_t> Wps_q3, Wps_t_q3, Wps_h, Wps_t_h;
    if (g->use_slabconv) {
        auto slot = +[](int ci, int co) { return colconv_wslot(0, ci, co); };
        dcs_slabconv_ps_pack(Wslab.data(), kh, kw, slot, 0, &Wps_q3);
        dcs_slabconv_ps_pack(Wslab.data(), kh, kw, slot, 1, &Wps_h);
        dcs_slabconv_ps_pack(Wslab_t.data(), kh, kw, slot, 0, &Wps_t_q3);
        dcs_slabconv_ps_pack(Wslab_t.data(), kh, kw, slot, 1, &Wps_t_h);
    }
    // column convolution (kw == 1): Wcol[u][ci][co] = W2[co][ci][kh-1-u] (true convolution), transpose
    // Wcol_t[u][co][ci] = W2[co][ci][u]
    std::vector<float> Wcol, Wcol_t;
    std::vector<_Float16> Wcol_h, Wcol_t_h;
    std::vector<float> Wcol_t_f;                         // the transposed filter once more in f32, [kh][32 out][40] (colconv_x3.hip)
    std::vector<float> Wcol_f;                           // the forward filter in that format (colconv_fwd_x3.hip)
    static const int col_env = getenv("DCS_COLCONV") ? atoi(getenv("DCS_COLCONV")) : 1;
    g->use_colconv = (kw == 1 && nf1 <= 32 && nf2 <= 32 && col_env) ? 1 : 0;
    if (g->use_colconv) {
        Wcol.assign((size_t)kh * 1024, 0.f);
        Wcol_t.assign((size_t)kh * 1024, 0.f);
        Wcol_h.assign((size_t)kh * 32 * 40, (_Float16)0.f);
        Wcol_t_h.assign((size_t)kh * 32 * 40, (_Float16)0.f);
        Wcol_t_f.assign((size_t)kh * 32 * 40, 0.f);
        Wcol_f.assign((size_t)kh * 32 * 40, 0.f);
        for (int co = 0; co < nf2; ++co)
            for (int ci = 0; ci < nf1; ++ci)
                for (int u = 0; u < kh; ++u) {
                    const float wf = W2[((size_t)co * nf1 + ci) * kh + (kh - 1 - u)], wt = W2[((size_t)co * nf1 + ci) * kh + u];
                    Wcol[colconv_wslot(u, ci, co)] = wf;
                    Wcol_t[colconv_wslot(u, co, ci)] = wt;
                    Wcol_h[((size_t)u * 32 + co) * 40 + ci] = (_Float16)wf;      // forward: out channel co, in channel ci
                    Wcol_t_h[((size_t)u * 32 + ci) * 40 + co] = (_Float16)wt;    // transpose: out channel ci, in channel co
                    Wcol_t_f[((size_t)u * 32 + ci) * 40 + co] = wt;
                    Wcol_f[((size_t)u * 32 + co) * 40 + ci] = wf;
                }
    }
    // dense layers: the flattened [nf2, h2, w2] order is the storage order of a2b, so no permutation
    const int Kfc = g->flat_p;
    std::vector<float> Bfc((size_t)dcs_round_up(Kfc, 128) * g->hid64, 0.f), biasfc(g->hid64, 0.f);
    for (int i = 0; i < d.flat; ++i)
        memcpy(&Bfc[(size_t)i * g->hid64], &P[6][(size_t)i * d.hidden], d.hidden * sizeof(float));
    for (int h = 0; h < d.hidden; ++h) biasfc[h] = P[7][h];
    int rc = DCS_OK;
#define UP(dst, src)                                \
    if (rc == DCS_OK) rc = upload(&(dst), (src));
    UP(g->W1c, W1c) UP(g->W1t, W1t) UP(g->bias1, bias1) UP(g->W2m, W2m) UP(g->bias2, bias2) UP(g->k2off, k2off) UP(g->k2uv, k2uv)
    UP(g->W2t, W2t) UP(g->bias0, bias0) UP(g->kt_off, kt_off) UP(g->kt_uv, kt_uv) UP(g->Bfc, Bfc) UP(g->biasfc, biasfc)
    UP(g->W2m_h, W2m_h) UP(g->W2t_h, W2t_h)
    if (!W1p.empty()) { UP(g->W1p, W1p) }
    if (d.sw1 == 4 && kw1 <= 32 && (C == 1 || C == 4)) {
        std::vector<uint16_t> W1m;
        dcs_conv1_mfma_pack(W1c.data(), nf1, C, kw1, &W1m);
        UP(g->W1m, W1m)
        if (kw1 > 28 && !W1p.empty()) {                  // the padded tap axis is 32 long
            std::vector<uint16_t> W1dq;
            dcs_deconv1_mfma_pack(W1p.data(), nf1, C, &W1dq);
            UP(g->W1dq, W1dq)
        }
    }
    // Fold conv2 + BiasLayer + bottleneck layer where it pays: the folded layer multiplies tc * wp / (h2 * w2) times the
    // bottleneck's weights, conv2 goes away.  iKala (10 x 20 filters): 159 MFLOP of conv2 per tile against 12 more in the
    // dense layer -- folded; the column-filter graphs (Bach10: 200 against 147, on a layer that is bound by its weight
    // stream) stay layer by layer.  DCS_FOLD_CONV2=0: off.
    {
        static const bool fold_env = !(getenv("DCS_FOLD_CONV2") && atoi(getenv("DCS_FOLD_CONV2")) == 0);
        const int64_t Kf = (int64_t)nf1 * tc * d.wp;
        const double conv2_flop = 2.0 * nf2 * nf1 * kh * kw * d.h2 * d.w2, more_fc_flop = 2.0 * (double)(Kf - d.flat) * d.hidden;
        if (rc == DCS_OK && fold_env && kw > 1 && (Kf & 3) == 0 && Kf < (1 << 30) && conv2_flop >= 4.0 * more_fc_flop) {
            std::vector<float> Wf2((size_t)kh * kw * nf2 * nf1);
            for (int co = 0; co < nf2; ++co)
                for (int ci = 0; ci < nf1; ++ci)
                    for (int u = 0; u < kh; ++u)
                        for (int v = 0; v < kw; ++v)
                            Wf2[((size_t)(u * kw + v) * nf2 + co) * nf1 + ci] = W2[(((size_t)co * nf1 + ci) * kh + (kh - 1 - u)) * kw + (kw - 1 - v)];
            std::vector<double> cb(g->hid64, 0.0);
            for (int h = 0; h < d.hidden; ++h) cb[h] = P[7][h];
            for (int co = 0; co < nf2; ++co) {
                const double b = (double)b2[co] + (double)b2b[co];
                for (int i = 0; i < d.h2 * d.w2; ++i) {
                    const float* fr = &P[6][((size_t)co * d.h2 * d.w2 + i) * d.hidden];
                    for (int h = 0; h < d.hidden; ++h) cb[h] += b * (double)fr[h];
                }
            }
            std::vector<float> bias2fc(g->hid64, 0.f);
            for (int h = 0; h < d.hidden; ++h) bias2fc[h] = (float)cb[h];
            float* Wf2_d = nullptr;
            UP(Wf2_d, Wf2) UP(g->bias2fc, bias2fc)
            const size_t rows = (size_t)dcs_round_up(Kf, 128);
            if (rc == DCS_OK && hipMalloc((void**)&g->B2fc, rows * g->hid64 * sizeof(float)) != hipSuccess) {
                (void)hipGetLastError();
                g->B2fc = nullptr;                       // does not fit: layer by layer
            }
            if (rc == DCS_OK && g->B2fc) {
                if (hipMemsetAsync(g->B2fc, 0, rows * g->hid64 * sizeof(float), ctx->stream) != hipSuccess) rc = DCS_EHIP;
                hipLaunchKernelGGL(fold_conv2_fc_kernel, dim3((unsigned)Kf), dim3(kThreads), 0, ctx->stream, Wf2_d, g->Bfc, g->B2fc, nf1,
                                   nf2, kh, kw, tc, d.wp, d.h2, d.w2, g->hid64);
                if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = DCS_EHIP;
                g->K2fc = (int)Kf;
            }
            if (Wf2_d) (void)hipFree(Wf2_d);
        }
    }
    if (g->use_slabconv) { UP(g->Wslab, Wslab) UP(g->Wslab_t, Wslab_t) UP(g->Wslab_q3, Wslab_q3) UP(g->Wslab_t_q3, Wslab_t_q3) UP(g->Wslab_h, Wslab_h) UP(g->Wslab_t_h, Wslab_t_h) UP(g->Wps_q3, Wps_q3) UP(g->Wps_t_q3, Wps_t_q3) UP(g->Wps_h, Wps_h) UP(g->Wps_t_h, Wps_t_h) }
    if (g->use_colconv) {
        std::vector<_Float16> Wcol_r, Wcol_t_r;
        dcs_colconv_wreg_pack(Wcol_h.data(), kh, &Wcol_r);
        dcs_colconv_wreg_pack(Wcol_t_h.data(), kh, &Wcol_t_r);
        // Wcol is [u][1024] with (ci, co) at colconv_wslot(0, ci, co): the slab kernel's source format with kw = 1
        std::vector<uint16_t> Wpc_q3, Wpc_t_q3;
        auto slot = +[](int ci, int co) { return colconv_wslot(0, ci, co); };
        dcs_slabconv_ps_pack(Wcol.data(), kh, 1, slot, 0, &Wpc_q3);
        dcs_slabconv_ps_pack(Wcol_t.data(), kh, 1, slot, 0, &Wpc_t_q3);
        UP(g->Wpc_q3, Wpc_q3) UP(g->Wpc_t_q3, Wpc_t_q3)
        UP(g->Wcol, Wcol) UP(g->Wcol_t, Wcol_t) UP(g->Wcol_h, Wcol_h) UP(g->Wcol_t_h, Wcol_t_h) UP(g->Wcol_r, Wcol_r) UP(g->Wcol_t_r, Wcol_t_r)
        if (kh % 2 == 0 && !d.pool_w) {                   // f32-class conv2 with the weights in registers (colconv_fwd_x3.hip)
            std::vector<uint16_t> Wfx3;
            dcs_decoder_x3_pack(Wcol_f.data(), kh, &Wfx3);
            UP(g->Wfx3, Wfx3)
        }
        // both InverseLayers in one kernel: single-channel graphs (Bach10), and the f32-class kernel also with the four
        // output channels of the score-informed graph
        if ((C == 1 || C == 4) && d.sw1 == 4 && kw1 > 28 && kw1 <= 32 && !d.pool_w && !W1p.empty()) {
            std::vector<uint16_t> W1q;
            dcs_decoder_fused_pack(W1p.data(), nf1, C, &W1q);
            UP(g->W1q, W1q)
            if (kh % 2 == 0) {                           // the f32-class fused decoder deals the taps to two waves by parity
                std::vector<uint16_t> Wx3;
                dcs_decoder_x3_pack(Wcol_t_f.data(), kh, &Wx3);
                UP(g->Wx3, Wx3)
            }
        }
    }
    for (int s = 0; s < d.n_fc && rc == DCS_OK; ++s) {
        std::vector<float> Bd((size_t)dcs_round_up(g->hid64, 128) * g->flat64, 0.f), bd(g->flat64, 0.f);
        for (int h = 0; h < d.hidden; ++h)
            memcpy(&Bd[(size_t)h * g->flat64], &P[8 + 2 * s][(size_t)h * d.flat], d.flat * sizeof(float));
        memcpy(bd.data(), P[9 + 2 * s].data(), d.flat * sizeof(float));
        UP(g->Bd[s], Bd) UP(g->biasd[s], bd)
        {
            // channels-last order of the same bias (see forward_chunk: the fused decoder reads D as [row][x][channel])
            std::vector<float> bd_cl(bd);
            const int Cc = d.nf2, Pp = d.h2 * d.w2;
            for (int j = 0; j < Cc * Pp; ++j) bd_cl[j] = bd[(size_t)(j % Cc) * Pp + j / Cc];
            UP(g->biasd_cl[s], bd_cl)
        }
    }
    std::vector<float> bout(P[8 + 2 * d.n_fc]);
    UP(g->bout, bout)
#undef UP
    // (the per-source dense weights as three bf16 planes -- 1.5x their f32 size, a gigabyte for Bach10 -- are made by the
    // first forward pass large enough to run on the bf16 matrix pipe: ensure_bdq)
    if (rc != DCS_OK) {
        dcs_generic_destroy(g);
        return rc;
    }
    *out = g;
    return DCS_OK;
}

void dcs_generic_destroy(DcsGenericNet* g) {
    if (!g) return;
    void* ptrs[] = {g->Wpc_q3, g->Wpc_t_q3, g->Wps_q3, g->Wps_t_q3, g->Wps_h, g->Wps_t_h, g->W1t, g->Wslab, g->Wslab_t, g->Wslab_q3, g->Wslab_t_q3, g->Wslab_h, g->Wslab_t_h, g->W1p, g->Wcol, g->Wcol_t, g->Wcol_h, g->Wcol_t_h, g->Wcol_r, g->Wcol_t_r, g->W1q, g->W1m, g->W1dq, g->W2m_h, g->W2t_h, g->W1c, g->bias1, g->W2m, g->bias2, g->k2off, g->k2uv, g->W2t, g->bias0, g->kt_off, g->kt_uv,
                    g->Bfc, g->biasfc, g->Bd[0], g->Bd[1], g->Bd[2], g->Bd[3], g->biasd[0], g->biasd[1], g->biasd[2],
                    g->biasd[3], g->bout, g->rise_d, g->Bdq[0], g->Bdq[1], g->Bdq[2], g->Bdq[3], g->biasd_cl[0], g->biasd_cl[1],
                    g->biasd_cl[2], g->biasd_cl[3], g->Wx3, g->Bfcq, g->Wfx3, g->Bdh[0], g->Bdh[1], g->Bdh[2], g->Bdh[3], g->biasd_h[0],
                    g->biasd_h[1], g->biasd_h[2], g->biasd_h[3], g->Bfch, g->Bd32[0], g->Bd32[1], g->Bd32[2], g->Bd32[3], g->B2fc, g->bias2fc};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    g->ws.release();
    delete g;
}

namespace {

// false: the shape does not fit (LDS); the caller falls back to the implicit GEMM.
// Wq: weights packed for the 16-bit matrix pipe (mode 0: three bf16 planes, f32-class results -- the default; mode 1: one
// f16 plane); null: the f32-MFMA kernel (DCS_SLABCONV_MX=0).
bool launch_slabconv(dcs_ctx* ctx, SlabConvArgs a, int64_t n_images, const uint16_t* Wq = nullptr, int mode = 0,
                     const uint16_t* Wps = nullptr) {
    if (Wq && Wps && dcs_launch_slabconv_ps(ctx, a, n_images, Wps, mode)) return true;
    const int nxb = (a.Wo + 15) / 16;
    const int np = mode == 0 ? 3 : 1;
    a.tstage = Wq ? (a.kw < 2 ? a.kw : 2) : (a.kw < 4 ? a.kw : 4);
    const size_t w_bytes = Wq ? (size_t)2 * a.tstage * np * 32 * kSlabWRow * 16 : (size_t)2 * a.tstage * 1024 * sizeof(float);
    const size_t row_bytes = Wq ? (size_t)a.W * kSlabCi * sizeof(float) : (size_t)32 * a.W * sizeof(float);
    int band = 32 / nxb;                                   // 4 blocks per wave at most
    if (band < 1) return false;
    if (band > a.Ho) band = a.Ho;
    // more workgroups than two per CU if the image count alone does not give them
    // a workgroup streams all kh*kw weight tiles whatever its band: shrink the band only until every CU has a workgroup
    while (band > 1 && n_images * ((a.Ho + band - 1) / band) < (int64_t)ctx->n_cu) --band;
    band = (a.Ho + (a.Ho + band - 1) / band - 1) / ((a.Ho + band - 1) / band);   // equal bands
    size_t lds;
    for (;; --band) {
        if (band < 1) return false;
        lds = w_bytes + row_bytes * (size_t)(band + a.kh - 1);
        if (lds <= 160 * 1024) break;
    }
    a.band = band;
    a.n_bands = (a.Ho + band - 1) / band;
    a.rows_max = band + a.kh - 1;
    const dim3 grid((unsigned)(n_images * a.n_bands));
    if (Wq) {
        // 16 waves per workgroup: LDS allows one workgroup per CU, so the waves that hide each other's LDS latency have
        // to come from inside it
        constexpr int nw_env = 16;
        auto kern = nw_env == 8 ? (mode == 0 ? slabconv_mx_kernel<0, 8> : slabconv_mx_kernel<1, 8>)
                                : (mode == 0 ? slabconv_mx_kernel<0, 16> : slabconv_mx_kernel<1, 16>);
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        hipLaunchKernelGGL(kern, grid, dim3(nw_env == 8 ? 512 : 1024), lds, ctx->stream, a, reinterpret_cast<const u32x4*>(Wq));
        return true;
    }
    auto kern = slabconv_kernel;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(kern, grid, dim3(kColThreads), lds, ctx->stream, a);
    return true;
}

int launch_colconv(dcs_ctx* ctx, ColConvArgs a, int64_t n_images, const _Float16* Wh = nullptr, const _Float16* Wr = nullptr,
                   const uint16_t* Wps = nullptr) {
    a.n_xb = (a.W + 15) / 16;
    if (Wh && Wr && dcs_launch_colconv_wreg(ctx, a, n_images, Wr)) return DCS_OK;
    if (a.out_f16 || a.in_f16) DCS_FAIL(DCS_EHIP, "column convolution: f16 input / output asked of a kernel that cannot take it");
    constexpr bool ps_col = true;
    // f32-class forward conv2: bf16 x 3 with the slab pre-split in LDS (0.41 -> 0.25 ms on the score-informed batch).  The
    // transpose stays with the f32 column kernel: 7 of 20 taps are valid on average there and that kernel walks only
    // those (0.34 against 0.38 ms)
    if (!Wh && Wps && ps_col && a.ph == 0) {
        SlabConvArgs c{};
        c.in = a.in; c.in_n_stride = a.in_n_stride; c.Cin = a.Cin; c.H = a.H; c.W = a.W;
        c.Wk = a.Wk; c.bias = a.bias; c.out = a.out; c.out_n_stride = a.out_n_stride;
        c.Cout = a.Cout; c.Ho = a.Ho; c.Wo = a.W; c.kh = a.kh; c.kw = 1; c.ph = a.ph; c.pw = 0;
        if (dcs_launch_slabconv_ps(ctx, c, n_images, Wps, 0)) return DCS_OK;
    }
    if (Wh) {
        int per = 8;
        while (per > 1 && n_images * ((a.n_xb + per - 1) / per) < 4 * (int64_t)ctx->n_cu) per >>= 1;
        a.xb_per_wg = per;
        const size_t lds = ((size_t)a.kh * 32 * 40 + (size_t)a.H * 16 * 40) * sizeof(_Float16);
        if (lds > 160 * 1024 || a.Cin * a.H * 16 > 32 * kColThreads || a.Cin > 32)
            DCS_FAIL(DCS_EUNSUPPORTED, "column convolution (f16): %zu bytes of LDS", lds);
        auto kern = colconv_f16_kernel;
        if (lds > 48 * 1024)
            DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)(n_images * ((a.n_xb + per - 1) / per))), dim3(kColThreads), lds, ctx->stream,
                           a, Wh);
        return DCS_OK;
    }
    // column blocks per workgroup: amortise the 80 KB weight staging, but keep >= 4 workgroups per CU
    int per = 8;
    while (per > 1 && n_images * ((a.n_xb + per - 1) / per) < 4 * (int64_t)ctx->n_cu) per >>= 1;
    a.xb_per_wg = per;
    const size_t lds = ((size_t)a.kh * 1024 + (size_t)32 * (a.H * 16 + 16)) * sizeof(float);
    if (lds > 160 * 1024 || a.Cin * a.H * 16 > 32 * kColThreads)
        DCS_FAIL(DCS_EUNSUPPORTED, "column convolution: %zu bytes of LDS, %d x %d input rows", lds, a.Cin, a.H);
    auto kern = colconv_kernel;
    if (lds > 48 * 1024)
        DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t groups = (a.n_xb + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_images * groups)), dim3(kColThreads), lds, ctx->stream, a);
    return DCS_OK;
}

constexpr bool kF16Igemm = false;
constexpr bool kSlabMx = true;

// Will both InverseLayers run as ONE kernel, and does that kernel want the dense output channels-last?  One place decides it:
// forward_chunk (layout of the dense layers' output) and dcs_generic_forward (chunk size: the layout can only be produced by
// the all-rows dense kernel, 128 .. 176 tiles per launch).
bool plans_channels_last(const DcsGenericNet* g, bool* fuse_planned_out, bool* fuse_x3_out) {
    const DcsGenericDims& d = g->d;
    bool fuse_planned = false, fuse_x3 = false;
    if (g->use_colconv && g->W1q && (g->conv_f16 ? g->Wcol_t_r != nullptr : g->Wx3 != nullptr)) {
        ColConvArgs c0{};
        c0.Cin = d.nf2; c0.H = d.h2; c0.W = d.w2; c0.Cout = d.nf1; c0.Ho = g->tc; c0.ph = d.kh2 - 1; c0.kh = d.kh2;
        fuse_planned = g->conv_f16 ? (g->C == 1 && dcs_decoder_fused_ok(c0, g->F)) : dcs_decoder_x3_ok(c0, g->F, g->C);
        fuse_x3 = fuse_planned && !g->conv_f16;
    }
    static const bool cl_env = !(getenv("DCS_DECODER_CL") && atoi(getenv("DCS_DECODER_CL")) == 0);
    if (fuse_planned_out) *fuse_planned_out = fuse_planned;
    if (fuse_x3_out) *fuse_x3_out = fuse_x3;
    return fuse_planned && (cl_env || fuse_x3) && (d.nf2 & 1) == 0 && (g->flat_p & 1) == 0;
}

// one chunk of tiles through the graph; scratch carved from `w`
int forward_chunk(DcsGenericNet* g, const float* tiles, int64_t n, int64_t n_total, int64_t k_first, int mask_mode,
                  int tie_mode, float* out, char* w) {
    const DcsGenericDims& d = g->d;
    dcs_ctx* ctx = g->ctx;
    const int C = g->C, tc = g->tc, F = g->F;
    // Branches that reach the requested output.  The masks use the first S output channels (prediction2[:, 0:S]); with C
    // input channels per branch those belong to the first ceil(S / C) branches -- for the score-informed graph (C = 4,
    // S = 4) branch 0 alone: the other three branches of its decoder are dead code for predict_function2
    // (bach10_scoreinformed/separate_bach10.py:475-488; Theano prunes them from the compiled function as well).
    // dcs_model_forward (mask_mode 2: the whole network output) evaluates all of them.
    const int NB = mask_mode == 2 ? d.n_branch : (d.S + C - 1) / C < d.n_branch ? (d.S + C - 1) / C : d.n_branch;
    const int64_t plane1 = (int64_t)tc * d.w1, planep = (int64_t)tc * d.wp;
    bool want_a1_cl = false, a1_cl = false;             // conv1's output channels-last: asked for / written that way
    bool want_a1_16 = false, a1_16 = false;             // ... as f16 with 32 channels per position (f16 switch)
    int a1_rows = tc;                                   // rows from one tile's first row to the next's in the channels-last map (per-frame conv1: the stride)
    float* a1b = (float*)w; w += align256((size_t)n * d.nf1 * plane1 * 4);
    float* p1 = a1b;
    if (d.pool_w) { p1 = (float*)w; w += align256((size_t)n * d.nf1 * planep * 4); }
    float* a2b = (float*)w; w += align256((size_t)n * g->flat_p * 4);
    float* Z = (float*)w; w += align256((size_t)n * g->hid64 * 4);
    void* Zq = (void*)w; w += align256(dcs_gemm_aq_bytes(g->hid64, 176));   // Z as bf16 x 3 planes for the all-rows dense kernel
    float* D = (float*)w; w += align256((size_t)n * NB * g->flat_p * 4);
    float* g2 = (float*)w; w += align256((size_t)n * NB * d.nf1 * planep * 4);
    float* g1 = g2;
    if (d.pool_w) { g1 = (float*)w; w += align256((size_t)n * NB * d.nf1 * plane1 * 4); }
    float* o = (float*)w; w += align256((size_t)n * NB * C * tc * F * 4);
    // iKala: max-pool fused into conv1 and its VJP into conv1^T (register kernels, 30 filters of 30 taps at a stride of 3, pool
    // width 4).  The full-resolution activations a1b then never exist; their region holds the routing words instead
    // (pool_mw + 1 <= w1 words per row).  DCS_POOL_FUSED=0: the four separate kernels.
    static const int pool_fused_env = getenv("DCS_POOL_FUSED") ? atoi(getenv("DCS_POOL_FUSED")) : 1;
    static const int reg1_env = getenv("DCS_CONV1_REG") ? atoi(getenv("DCS_CONV1_REG")) : 1;
    static const int dreg_env = getenv("DCS_DECONV1_REG") ? atoi(getenv("DCS_DECONV1_REG")) : 1;
    const int pool_mw = (int)dcs_cdiv(d.w1, 64) * 2;
    const bool pool_fused = pool_fused_env && reg1_env && dreg_env && d.pool_w == 4 && d.nf1 == 30 && d.sw1 == 3 && d.kw1 <= 32 &&
                            (d.kw1 + 2) / 3 == 10 && g->W1t && g->W1p && !g->W1m && !g->W1dq &&
                            pool_mw + 1 <= d.w1 && d.wp * 4 <= d.w1;
    unsigned* pool_bits = reinterpret_cast<unsigned*>(a1b);

    // conv1 + both biases
    {
        const size_t lds = ((size_t)d.nf1 * C * d.kw1 + (size_t)C * (kThreads * d.sw1 + d.kw1)) * 4;
        auto kern = conv1_kernel<30>;
        if (lds > 48 * 1024)
            DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
        DcsTimer tm(ctx, DCS_TAG_CONV1);
        // f32-class column conv2 with the weights in registers (colconv_fwd_x3.hip) reads a position's channels together: conv1
        // then writes its output channels-last (only the bf16-pipe conv1 kernel can; nothing else reads a1b in these graphs)
        {
            ColConvArgs c2{};
            c2.in = a1b; c2.in_n_stride = (int64_t)d.nf1 * plane1; c2.Cin = d.nf1; c2.H = tc; c2.W = d.w1;
            c2.Cout = d.nf2; c2.Ho = d.h2; c2.ph = 0; c2.kh = d.kh2;
            want_a1_cl = !g->conv_f16 && g->use_colconv && g->Wfx3 && g->W1m && !d.pool_w && d.wp == d.w1 && dcs_colconv_fwd_x3_ok(c2) &&
                         (c2.in_n_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(a1b) & 7) == 0;
            // f16 switch, one input channel (round 6): the weights-in-registers f16 conv2 takes its input as f16, channels-last, 32
            // channels per position -- conv1 writes a1b16[n][t][x][32] halves (half the bytes; a lane of conv2 loads 16 bytes per row)
            static const bool cl_on = !(getenv("DCS_DECODER_CL") && atoi(getenv("DCS_DECODER_CL")) == 0);
            want_a1_16 = cl_on && g->conv_f16 && C == 1 && g->use_colconv && g->Wcol_h && g->Wcol_r && g->W1m && !d.pool_w && d.wp == d.w1 &&
                         d.nf1 <= 32 && dcs_colconv_wreg_scatter_ok(c2) && (reinterpret_cast<uintptr_t>(a1b) & 15) == 0;
        }
        // One clip, all its tiles in this chunk, a channels-last hand-over to conv2: conv1 (a per-row operation) runs once per
        // FRAME of the clip instead of once per tile row -- tc / (tc - overlap) = 6 x fewer rows at the reference's settings --
        // and conv2 takes tile k as the window of tc rows from row k * st of that map (its tile stride is a parameter).
        // Same arithmetic per row: bit-identical to the per-tile form.
        const bool per_frame = g->frames_src && n == n_total && k_first == 0 && g->frames_st > 0 &&
                               g->frames_rows == (n - 1) * g->frames_st + tc && g->frames_st < tc;
        static const int reg1 = getenv("DCS_CONV1_REG") ? atoi(getenv("DCS_CONV1_REG")) : 1;
        const dim3 grid1((unsigned)dcs_cdiv(d.w1, kThreads), (unsigned)(n * tc));
        if (pool_fused) {   // conv1 + max-pool: pooled rows to p1, the un-pooling routing bits where the activations would go
            hipLaunchKernelGGL((conv1_reg_kernel<30, 3, true>), grid1, dim3(kThreads), 0, ctx->stream, tiles, g->W1t, g->bias1, p1, C,
                               tc, F, d.kw1, d.w1, pool_bits, d.wp, pool_mw, tie_mode == DCS_TIE_FIRST ? 1 : 0);
        } else if (per_frame && want_a1_16 &&
                   dcs_launch_conv1_mfma(ctx, g->frames_src, g->W1m, g->bias1, a1b, 1, C, d.nf1, g->frames_rows, F, d.kw1, d.sw1, d.w1, true, true)) {
            a1_16 = true; a1_rows = g->frames_st;
        } else if (per_frame && want_a1_cl &&
                   dcs_launch_conv1_mfma(ctx, g->frames_src, g->W1m, g->bias1, a1b, 1, C, d.nf1, g->frames_rows, F, d.kw1, d.sw1, d.w1, true)) {
            a1_cl = true; a1_rows = g->frames_st;
        } else if (want_a1_16 && dcs_launch_conv1_mfma(ctx, tiles, g->W1m, g->bias1, a1b, n, C, d.nf1, tc, F, d.kw1, d.sw1, d.w1, true, true)) {
            a1_16 = true;
        } else if (g->W1m && dcs_launch_conv1_mfma(ctx, tiles, g->W1m, g->bias1, a1b, n, C, d.nf1, tc, F, d.kw1, d.sw1, d.w1, want_a1_cl)) {
            a1_cl = want_a1_cl;
        } else if (reg1 && d.nf1 == 30 && d.kw1 <= 32 && d.sw1 == 4)   // the register kernel is built for 30 filters
            hipLaunchKernelGGL((conv1_reg_kernel<30, 4>), grid1, dim3(kThreads), 0, ctx->stream, tiles, g->W1t, g->bias1, a1b, C,
                               tc, F, d.kw1, d.w1);
        else if (reg1 && d.nf1 == 30 && d.kw1 <= 32 && d.sw1 == 3)
            hipLaunchKernelGGL((conv1_reg_kernel<30, 3>), grid1, dim3(kThreads), 0, ctx->stream, tiles, g->W1t, g->bias1, a1b, C,
                               tc, F, d.kw1, d.w1);
        else
            hipLaunchKernelGGL(kern, grid1, dim3(kThreads), lds, ctx->stream, tiles, g->W1c, g->bias1, a1b, C, tc, F, d.kw1,
                               d.sw1, d.w1);
        tm.done();
    }
    if (d.pool_w && !pool_fused) {
        DcsTimer tm(ctx, DCS_TAG_POOL);
        hipLaunchKernelGGL(pool_kernel, dim3((unsigned)dcs_cdiv(n * d.nf1 * tc * d.wp, kThreads)), dim3(kThreads), 0,
                           ctx->stream, a1b, p1, n * d.nf1 * tc, d.w1, d.wp, d.pool_w);
        tm.done();
    }
    // f16 switch (round 6): when the bottleneck layer will run on its f16 weight plane (gemm_f16_longk_kernel: 128 .. 176 tiles, K >=
    // 16 384) AND conv2 is the weights-in-registers f16 kernel, conv2 writes its map as f16 -- a2b16[n][pitch16] halves, pitch16 =
    // flat rounded up to 32 so that every K tile of the layer lies inside a row -- and the layer multiplies those rows as they
    // are: half the bytes on both sides of the hand-over, one MFMA per block instead of two.  The plane is made on first need.
    static const bool fc16_on = !(getenv("DCS_GEMM_BF16") && atoi(getenv("DCS_GEMM_BF16")) == 0);
    const int pitch16 = (int)dcs_round_up(d.flat, 32);
    int ks16 = 0;
    bool a2b16 = false;
    if (fc16_on && g->conv_f16 && !g->bfch_failed) {
        ks16 = dcs_gemm_f16_longk_slices(ctx, (int)n, pitch16, g->hid64);
        if (ks16 >= 2 && !g->Bfch) {
            void* plane = nullptr;
            if (hipMalloc(&plane, dcs_gemm_bh_bytes(g->flat_p, g->hid64)) != hipSuccess) {
                (void)hipGetLastError();
                g->bfch_failed = true;
            } else {
                const int rc = dcs_gemm_pack_bh_plain(ctx, g->Bfc, g->flat_p, g->hid64, g->hid64, plane);
                if (rc != DCS_OK) {
                    (void)hipFree(plane);
                    return rc;
                }
                g->Bfch = plane;
            }
        }
        if (ks16 >= 2 && g->Bfch && g->use_colconv && g->Wcol_h && g->Wcol_r && !d.pool_w && (d.flat & 1) == 0) {
            ColConvArgs c2{};
            c2.Cin = d.nf1; c2.H = tc; c2.W = d.wp; c2.Cout = d.nf2; c2.Ho = d.h2; c2.ph = 0; c2.kh = d.kh2;
            a2b16 = dcs_colconv_wreg_scatter_ok(c2);
        }
    }
    // conv2 + BiasLayer + bottleneck layer as one folded affine map of conv2's input (created with the model where it pays:
    // the iKala graph; f32-class only -- the f16 switch asks for conv2 in f16)
    const bool fold2 = g->B2fc && g->K2fc > 0 && !g->conv_f16 && !a1_cl && !a1_16;
    // conv2 + both biases -> a2b[n][flat_p] (pad columns zeroed; f16: the pitch16 - flat pad halves, as flat / 2 .. pitch16 / 2 floats)
    if (fold2) {
    } else if (a2b16) {
        if (pitch16 != d.flat)
            hipLaunchKernelGGL(zero_pad_cols_kernel, dim3((unsigned)dcs_cdiv(n * ((pitch16 - d.flat) / 2), kThreads)), dim3(kThreads), 0,
                               ctx->stream, a2b, n, d.flat / 2, pitch16 / 2);
    } else if (g->flat_p != d.flat)
        hipLaunchKernelGGL(zero_pad_cols_kernel, dim3((unsigned)dcs_cdiv(n * (g->flat_p - d.flat), kThreads)), dim3(kThreads), 0,
                           ctx->stream, a2b, n, d.flat, g->flat_p);
    if (!fold2) {
        IgemmArgs a{};
        a.in = p1; a.in_n_stride = (int64_t)d.nf1 * planep; a.Cin = d.nf1; a.H = tc; a.W = d.wp;
        a.Wm = g->W2m; a.koff = g->k2off; a.kuv = g->k2uv; a.bias = g->bias2;
        a.out = a2b; a.out_n_stride = g->flat_p; a.Cout = d.nf2; a.Ho = d.h2; a.Wo = d.w2;
        a.ph = 0; a.pw = 0; a.K = g->K2; a.M = n * d.h2 * d.w2;
        DcsTimer tm(ctx, DCS_TAG_CONV2);
        if (g->use_colconv) {
            ColConvArgs c{};
            c.in = a.in; c.in_n_stride = a.in_n_stride; c.Cin = a.Cin; c.H = a.H; c.W = a.W;
            c.Wk = g->Wcol; c.bias = a.bias; c.out = a.out; c.out_n_stride = a.out_n_stride; c.Cout = a.Cout; c.Ho = a.Ho;
            c.ph = 0; c.kh = d.kh2;
            if (a2b16) { c.out_f16 = 1; c.out_n_stride = pitch16; }
            if (a1_16) { c.in_f16 = 1; c.in_n_stride = (int64_t)a1_rows * d.w1 * 32; }
            else if (a1_cl) c.in_n_stride = (int64_t)a1_rows * d.w1 * d.nf1;
            if (a1_cl) {
                if (!dcs_launch_colconv_fwd_x3(ctx, c, n, g->Wfx3))
                    DCS_FAIL(DCS_EHIP, "generic graph: conv2 refused the channels-last input conv1 was asked to write");
            } else
            DCS_CHECK(launch_colconv(ctx, c, n, g->conv_f16 ? g->Wcol_h : nullptr, g->Wcol_r, g->Wpc_q3));
            // (launch_colconv fails a launch that asks for f16 output and is not taken by the weights-in-registers kernel)
        } else if (g->conv_f16 && !(g->use_slabconv && !kF16Igemm))
            // general filters (iKala, 10 x 20) go through the slab kernel in either precision unless DCS_F16_IGEMM=1 asks
            // for the f16 implicit GEMM
            hipLaunchKernelGGL(conv_igemm_f16_kernel, dim3((unsigned)dcs_cdiv(a.M, 128)), dim3(kThreads), 0, ctx->stream, a,
                               g->W2m_h);
        else {
            bool done = false;
            if (g->use_slabconv) {
                SlabConvArgs c{};
                c.in = a.in; c.in_n_stride = a.in_n_stride; c.Cin = a.Cin; c.H = a.H; c.W = a.W;
                c.Wk = g->Wslab; c.bias = a.bias; c.out = a.out; c.out_n_stride = a.out_n_stride;
                c.Cout = a.Cout; c.Ho = a.Ho; c.Wo = a.Wo; c.kh = d.kh2; c.kw = d.kw2; c.ph = 0; c.pw = 0;
                done = launch_slabconv(ctx, c, n, kSlabMx ? (g->conv_f16 ? g->Wslab_h : g->Wslab_q3) : nullptr, g->conv_f16 ? 1 : 0,
                                       g->conv_f16 ? g->Wps_h : g->Wps_q3);
            }
            if (!done)
                hipLaunchKernelGGL(conv_igemm_kernel, dim3((unsigned)dcs_cdiv(a.M, 128)), dim3(kThreads), 0, ctx->stream, a);
        }
        tm.done();
    }
    // bottleneck dense (rectify)
    {
        DcsGemm q{};
        q.A = a2b; q.lda = g->flat_p; q.a_gdiv = 1 << 30; q.a_gmul = 0; q.a_scale = 1.f;
        q.B = g->Bfc; q.ldb = g->hid64; q.bias = g->biasfc;
        q.C = Z; q.ldc = g->hid64; q.c_gdiv = 1 << 30; q.c_gmul = 0;
        q.M = n; q.n_cols = g->hid64; q.n_store = g->hid64; q.K = g->flat_p; q.relu = 1; q.a_vec = 1;
        if (fold2) {   // the tile's conv2 input p1[ci][row][x], contiguous per tile
            q.A = p1; q.lda = (int64_t)d.nf1 * planep; q.K = g->K2fc; q.B = g->B2fc; q.bias = g->bias2fc;
        }
        // 128 .. 176 tiles against a very long K (166 650 for the Bach10 graphs): the all-rows bf16 x 3 kernel with K cut into
        // slices (dcs_launch_gemm_bf16x3_longk) -- the planes (1.5 x the f32 weights) are made on first need; without them,
        // or for any other row count, the f32 K-split of gemm.hip
        static const bool fcq_on = !(getenv("DCS_GEMM_BF16") && atoi(getenv("DCS_GEMM_BF16")) == 0);
        if (!fold2 && fcq_on && n >= 128 && n <= 176 && g->flat_p >= 16384 && (g->hid64 % 128) == 0 && !g->Bfcq && !g->bfcq_failed &&
            (!g->conv_f16 || g->bfch_failed)) {           // (under the f16 switch the layer takes the f16 plane below instead)
            void* planes = nullptr;                       // (unsplit 32-byte pieces since round 6: the long-K launch is the all-rows kernel)
            if (hipMalloc(&planes, dcs_gemm_b32_bytes(g->flat_p, g->hid64)) != hipSuccess) {
                (void)hipGetLastError();
                g->bfcq_failed = true;
            } else {
                const int rc = dcs_gemm_pack_b32(ctx, g->Bfc, g->flat_p, g->hid64, g->hid64, planes);
                if (rc != DCS_OK) {
                    (void)hipFree(planes);
                    return rc;
                }
                g->Bfcq = planes;
            }
        }
        q.Bq = fold2 ? nullptr : g->Bfcq;
        q.bq_f32 = 1;
        // f16 switch: the same long-K launch on f16 weights (gemm_f16.hip: one plane, 2 bytes per weight instead of 6); the rows
        // of A are conv2's f16 map (a2b16, above) or, when conv2 could not write it, f32 rows split into two f16 terms on their
        // way into LDS; the slices are added by the same second pass
        bool fc16 = false;
        const int ks_fc = a2b16 ? ks16 : ((fc16_on && g->conv_f16 && g->Bfch) ? dcs_gemm_f16_longk_slices(ctx, (int)n, g->flat_p, g->hid64) : 0);
        if (ks_fc >= 2 && g->Bfch && ctx->gemm_ws.ensure((size_t)ks_fc * n * g->hid64 * sizeof(float)) == DCS_OK) {
            DcsTimer tm(ctx, DCS_TAG_FC);
            fc16 = a2b16 ? dcs_launch_gemm_f16_longk(ctx, a2b, pitch16, (int)n, pitch16, g->hid64, g->Bfch, (float*)ctx->gemm_ws.ptr, true)
                         : dcs_launch_gemm_f16_longk(ctx, a2b, g->flat_p, (int)n, g->flat_p, g->hid64, g->Bfch, (float*)ctx->gemm_ws.ptr, false);
            if (fc16) {
                DcsGemm r = q;
                r.partial = (float*)ctx->gemm_ws.ptr;
                dcs_launch_gemm_longk_reduce(ctx, r, ks_fc);
                tm.done();
            } else
                tm.cancel();
        }
        if (a2b16 && !fc16) DCS_FAIL(DCS_EHIP, "generic graph: conv2 wrote an f16 map and the f16 bottleneck layer refused it");
        if (!fc16) DCS_CHECK(dcs_launch_gemm_rows(ctx, q, DCS_TAG_FC));
    }
    // per-source dense (rectify): D[n][branch][flat_p]; aliased branches (none in these graphs) would reuse a layer
    if (g->flat_p != d.flat)
        hipLaunchKernelGGL(zero_pad_cols_kernel, dim3((unsigned)dcs_cdiv(n * NB * (g->flat_p - d.flat), kThreads)), dim3(kThreads), 0,
                           ctx->stream, D, n * NB, d.flat, g->flat_p);
    // Will both InverseLayers run as ONE kernel (Bach10 graph with the f16 switch)?  Known before the dense layers run, and it
    // decides their output layout: the fused decoder reads a position's channels together, so D is written CHANNELS-LAST
    // ([branch][row][x][channel]) by packing the bf16 planes of the dense weights with permuted columns (DCS_DECODER_CL=0:
    // channel-first as every other consumer takes it).
    // With the switch off the same fusion runs on three-way split operands (colconv_x3.hip, round 4: two waves per column
    // block); that kernel takes the channels-last layout only, so it is planned when the layout can be had.
    bool fuse_planned = false, fuse_x3 = false;
    const bool want_cl = plans_channels_last(g, &fuse_planned, &fuse_x3);
    // f16 switch, fused decoder, 128 .. 176 tiles, one input channel: the per-source dense layers on f16 weights with an f16,
    // channels-last (32 channels per position) output -- gemm_f16.hip: 2 bytes per weight instead of 6, and D written and read
    // at half the size.  The planes (0.36 GB for Bach10) are made on first need; if they do not fit, the f32-class path stays.
    // (same-box A/B against the f32-class dense layers in front of the same decoder: profiles/r06_f_bach10_f16_dense_legs_ab.txt)
    bool dense16 = false, d_cl = false;
    if (g->conv_f16 && fuse_planned && !fuse_x3 && want_cl && C == 1 && n >= 128 && n <= 176 && !g->bdh_failed &&
        (g->hid64 & 31) == 0) {
        const int npos = d.h2 * d.w2;
        const int n_out = (int)dcs_round_up((int64_t)npos * 32, 128);
        g->n_out16 = n_out;
        for (int b = 0; b < NB && !g->bdh_failed; ++b) {
            const int s2 = d.branch_fc[b];
            if (g->Bdh[s2]) continue;
            void* planes = nullptr;
            float* bias_h = nullptr;
            if (hipMalloc(&planes, dcs_gemm_bh_bytes(g->hid64, n_out)) != hipSuccess || hipMalloc((void**)&bias_h, (size_t)n_out * 4) != hipSuccess) {
                (void)hipGetLastError();
                if (planes) (void)hipFree(planes);
                g->bdh_failed = true;
                break;
            }
            int rc = dcs_gemm_pack_bh(ctx, g->Bd[s2], g->hid64, g->flat64, n_out, d.nf2, npos, 32, planes);
            if (rc == DCS_OK) rc = dcs_gemm_pack_bias_cl(ctx, g->biasd[s2], n_out, d.nf2, npos, 32, bias_h);
            if (rc != DCS_OK) {
                (void)hipFree(planes);
                (void)hipFree(bias_h);
                return rc;
            }
            g->Bdh[s2] = planes;                         // published only when packed
            g->biasd_h[s2] = bias_h;
        }
        if (!g->bdh_failed) {
            const void* bh[4] = {nullptr, nullptr, nullptr, nullptr};
            const float* bs[4] = {nullptr, nullptr, nullptr, nullptr};
            void* cs[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int b = 0; b < NB; ++b) {
                const int s2 = d.branch_fc[b];
                bh[b] = g->Bdh[s2]; bs[b] = g->biasd_h[s2];
                cs[b] = reinterpret_cast<_Float16*>(D) + (int64_t)b * n_out;      // D16[tile][branch][position][32]
            }
            DcsTimer tm(ctx, DCS_TAG_FC1X);
            dense16 = dcs_launch_gemm_f16_skinny(ctx, Z, g->hid64, (int)n, g->hid64, n_out, NB, bh, bs, cs, (int64_t)NB * n_out, Zq);
            if (dense16) tm.done(); else tm.cancel();
            d_cl = dense16;
        }
    }
    if (!dense16) {
    // the bf16 planes of the dense weights, on first need: a launch of >= 128 rows against >= 1024 columns (smaller ones stay
    // on the f32 kernels whatever is packed, dcs_launch_gemm_bf16x3); same stream, so no synchronisation
    static const bool bf16_on = !(getenv("DCS_GEMM_BF16") && atoi(getenv("DCS_GEMM_BF16")) == 0);
    // (same-box A/B against the planes: Bach10 fc1x 0.396 -> 0.352 ms, profiles/r06_m_unsplit_weight_pieces_legs_ab.txt)
    const bool win = n <= 176;                           // the all-rows kernel's window: unsplit pieces (Bd32), else planes (Bdq)
    void** store = win ? g->Bd32 : g->Bdq;
    bool& store_cl = win ? g->bd32_cl : g->bdq_cl;
    if (bf16_on && n >= 128 && g->flat64 >= 1024) {
        const int rows = (int)dcs_round_up(g->hid64, 128);
        const int pc = want_cl ? d.nf2 : 0, pp = want_cl ? d.h2 * d.w2 : 0;
        auto pack = [&](const float* B, void* dst) {
            return win ? dcs_gemm_pack_b32(ctx, B, rows, g->flat64, g->flat64, dst, pc, pp)
                       : dcs_gemm_pack_bq(ctx, B, rows, g->flat64, g->flat64, dst, pc, pp);
        };
        if (store_cl != want_cl) {
            // the precision switch was flipped since the weights were packed: the existing blocks are re-packed in place, in
            // the other column order (stream-ordered behind every earlier use)
            for (int s2 = 0; s2 < d.n_fc; ++s2)
                if (store[s2]) DCS_CHECK(pack(g->Bd[s2], store[s2]));
            store_cl = want_cl;
        }
        for (int s2 = 0; s2 < d.n_fc && !g->bdq_failed; ++s2) {
            if (store[s2]) continue;
            // published only when packed: a failed pack must not leave a non-null, unpacked block behind (later calls
            // would multiply by uninitialised memory).  Out of memory here (~0.7 GB for Bach10, outside the chunk budget) is
            // not an error of the forward pass: the layer stays on the f32 GEMM (q.Bq == nullptr) for the model's lifetime.
            void* planes = nullptr;
            if (hipMalloc(&planes, win ? dcs_gemm_b32_bytes(rows, g->flat64) : dcs_gemm_bq_bytes(rows, g->flat64)) != hipSuccess) {
                (void)hipGetLastError();
                g->bdq_failed = true;
                break;
            }
            const int rc = pack(g->Bd[s2], planes);
            if (rc != DCS_OK) {
                (void)hipFree(planes);
                return rc;
            }
            store[s2] = planes;
        }
    }
    // planes in channels-last order serve the all-branches launch below and nothing else: a launch that falls back to the
    // per-branch GEMMs runs them on the f32 weights (channel-first), and the decoder is told which layout it got
    const bool planes_cl = store_cl;
    bool branches_done = false;
    if (NB > 1 || want_cl) {                             // every live branch in one launch when the shape allows it
        DcsGemm q{};
        q.A = Z; q.lda = g->hid64; q.a_gdiv = 1 << 30; q.a_gmul = 0; q.a_scale = 1.f;
        q.ldb = g->flat64; q.ldc = (int64_t)NB * g->flat_p; q.c_gdiv = 1 << 30; q.c_gmul = 0;
        q.M = n; q.n_cols = g->flat64; q.n_store = d.flat; q.K = g->hid64; q.relu = 1; q.a_vec = 1;
        DcsGemmBranches br{};
        br.n = NB;
        for (int b = 0; b < NB; ++b) {
            const int s = d.branch_fc[b];
            br.Bq[b] = store[s]; br.bias[b] = planes_cl ? g->biasd_cl[s] : g->biasd[s]; br.C[b] = D + (int64_t)b * g->flat_p;
        }
        q.B = g->Bd[d.branch_fc[0]]; q.bias = br.bias[0]; q.Bq = br.Bq[0]; q.C = br.C[0]; q.bq_f32 = win ? 1 : 0;
        DcsTimer tm(ctx, DCS_TAG_FC1X);
        // the launch below takes the all-rows kernel: split Z once for all its workgroups (several branches: 0.413 -> 0.392 ms for
        // Bach10; with one branch the extra launch costs what it saves)
        if (NB > 1 && n >= 128 && n <= 176 && br.Bq[0]) {
            const int rows_pad = n <= 128 ? 128 : 176;
            DCS_CHECK(dcs_gemm_split_a(ctx, Z, g->hid64, n, g->hid64, rows_pad, Zq));
            q.Aq = Zq; q.aq_rows = rows_pad;
        }
        branches_done = dcs_launch_gemm_bf16x3_skinny(ctx, q, &br);
        if (branches_done) tm.done(); else tm.cancel();
    }
    d_cl = branches_done && planes_cl;                   // layout of D as the decoder will find it
    for (int b = 0; b < NB && !branches_done; ++b) {
        const int s = d.branch_fc[b];
        DcsGemm q{};
        q.A = Z; q.lda = g->hid64; q.a_gdiv = 1 << 30; q.a_gmul = 0; q.a_scale = 1.f;
        q.B = g->Bd[s]; q.ldb = g->flat64; q.bias = g->biasd[s]; q.Bq = planes_cl ? nullptr : store[s]; q.bq_f32 = win ? 1 : 0;
        q.C = D + (int64_t)b * g->flat_p; q.ldc = (int64_t)NB * g->flat_p; q.c_gdiv = 1 << 30; q.c_gmul = 0;
        q.M = n; q.n_cols = g->flat64; q.n_store = d.flat; q.K = g->hid64; q.relu = 1; q.a_vec = 1;
        DCS_CHECK(dcs_launch_gemm_rows(ctx, q, DCS_TAG_FC1X));
    }
    }
    // InverseLayer(., conv2): [n*NB, nf2, h2, w2] -> [n*NB, nf1, tc, wp]
    bool decoder_fused = false;
    {
        IgemmArgs a{};
        a.in = D; a.in_n_stride = g->flat_p; a.Cin = d.nf2; a.H = d.h2; a.W = d.w2;
        a.Wm = g->W2t; a.koff = g->kt_off; a.kuv = g->kt_uv; a.bias = g->bias0;
        a.out = g2; a.out_n_stride = (int64_t)d.nf1 * planep; a.Cout = d.nf1; a.Ho = tc; a.Wo = d.wp;
        a.ph = d.kh2 - 1; a.pw = d.kw2 - 1; a.K = g->K2; a.M = n * NB * planep;
        a.kh = d.kh2; a.k_per_u = d.nf2 * d.kw2;
        ColConvArgs c{};
        if (g->use_colconv) {
            c.in = a.in; c.in_n_stride = a.in_n_stride; c.Cin = a.Cin; c.H = a.H; c.W = a.W;
            c.Wk = g->Wcol_t; c.bias = a.bias; c.out = a.out; c.out_n_stride = a.out_n_stride; c.Cout = a.Cout; c.Ho = a.Ho;
            c.ph = d.kh2 - 1; c.kh = d.kh2; c.n_xb = (c.W + 15) / 16;
            decoder_fused = g->conv_f16 ? (C == 1 && g->W1q && g->Wcol_t_r && dcs_decoder_fused_ok(c, F))
                                        : (fuse_x3 && d_cl);       // f32-class: only on the channels-last layout
            if (dense16) {
                if (!decoder_fused) DCS_FAIL(DCS_EHIP, "generic graph: f16 dense output planned without the fused decoder");
                c.in_n_stride = g->n_out16;                    // halves: D16[image = tile * NB + branch][position][32]
            }
        }
        if (d_cl && !decoder_fused) {
            // cannot happen while the plan above and the launch conditions agree (the layout is only asked for when the fused
            // decoder is planned); if they ever disagree, redo the dense layers channel-first on the f32 weights instead of
            // failing the call: every other consumer takes that layout
            for (int b = 0; b < NB; ++b) {
                const int s = d.branch_fc[b];
                DcsGemm q{};
                q.A = Z; q.lda = g->hid64; q.a_gdiv = 1 << 30; q.a_gmul = 0; q.a_scale = 1.f;
                q.B = g->Bd[s]; q.ldb = g->flat64; q.bias = g->biasd[s]; q.Bq = nullptr;
                q.C = D + (int64_t)b * g->flat_p; q.ldc = (int64_t)NB * g->flat_p; q.c_gdiv = 1 << 30; q.c_gmul = 0;
                q.M = n; q.n_cols = g->flat64; q.n_store = d.flat; q.K = g->hid64; q.relu = 1; q.a_vec = 1;
                DCS_CHECK(dcs_launch_gemm_rows(ctx, q, DCS_TAG_FC1X));
            }
        }
        if (decoder_fused) {                                 // both InverseLayers in one kernel: o directly
            DcsTimer tmf(ctx, DCS_TAG_DECODER);
            const bool ok = g->conv_f16 ? dcs_launch_decoder_fused(ctx, c, n * NB, g->Wcol_t_r, g->W1q, o, F, d_cl, dense16)
                                        : dcs_launch_decoder_x3(ctx, c, n * NB, g->Wx3, g->W1q, o, F, C);
            tmf.done();
            if (!ok) DCS_FAIL(DCS_EHIP, "generic graph: the fused decoder refused a launch it had accepted (channels-last %d)", (int)d_cl);
        }
        DcsTimer tm(ctx, decoder_fused ? -1 : DCS_TAG_DECONV2);
        if (decoder_fused) {
        } else if (g->use_colconv) {
            DCS_CHECK(launch_colconv(ctx, c, n * NB, g->conv_f16 ? g->Wcol_t_h : nullptr, g->Wcol_t_r, g->Wpc_t_q3));
        } else if (g->conv_f16 && !(g->use_slabconv && !kF16Igemm))
            hipLaunchKernelGGL(conv_igemm_f16_kernel, dim3((unsigned)dcs_cdiv(a.M, 128)), dim3(kThreads), 0, ctx->stream, a,
                               g->W2t_h);
        else {
            bool done = false;
            if (g->use_slabconv) {
                SlabConvArgs c{};
                c.in = a.in; c.in_n_stride = a.in_n_stride; c.Cin = a.Cin; c.H = a.H; c.W = a.W;
                c.Wk = g->Wslab_t; c.bias = a.bias; c.out = a.out; c.out_n_stride = a.out_n_stride;
                c.Cout = a.Cout; c.Ho = a.Ho; c.Wo = a.Wo; c.kh = d.kh2; c.kw = d.kw2; c.ph = d.kh2 - 1; c.pw = d.kw2 - 1;
                done = launch_slabconv(ctx, c, n * NB, kSlabMx ? (g->conv_f16 ? g->Wslab_t_h : g->Wslab_t_q3) : nullptr,
                                       g->conv_f16 ? 1 : 0, g->conv_f16 ? g->Wps_t_h : g->Wps_t_q3);
            }
            if (!done)
                hipLaunchKernelGGL(conv_igemm_kernel, dim3((unsigned)dcs_cdiv(a.M, 128)), dim3(kThreads), 0, ctx->stream, a);
        }
        tm.done();
    }
    // InverseLayer(., pool)
    if (d.pool_w && !pool_fused) {
        DcsTimer tm(ctx, DCS_TAG_UNPOOL);
        const int64_t rows_g = n * NB * d.nf1 * tc;
        if (d.pool_w != 4 || rows_g > 0x7fffffff) DCS_FAIL(DCS_EUNSUPPORTED, "un-pool: pool width %d, %lld rows", d.pool_w, (long long)rows_g);
        hipLaunchKernelGGL((unpool_kernel<4>), dim3((unsigned)dcs_cdiv(rows_g, 4)), dim3(kThreads), 0,
                           ctx->stream, g2, a1b, g1, (unsigned)rows_g, (unsigned)(d.nf1 * tc), (unsigned)NB, d.w1, d.wp,
                           tie_mode == DCS_TIE_FIRST ? 1 : 0);
        tm.done();
    }
    // InverseLayer(., conv1): [n*NB, nf1, tc, w1] -> [n*NB, C, tc, F] = [n, NB*C, tc, F]
    if (!decoder_fused) {
        const int span = kThreads / d.sw1 + d.kw1 / d.sw1 + 3;
        const size_t lds = ((size_t)d.nf1 * C * d.kw1 + (size_t)d.nf1 * span) * 4;
        auto kern = deconv1_kernel<30>;
        if (lds > 48 * 1024)
            DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
        DcsTimer tm(ctx, DCS_TAG_FINAL);
        static const int reg_env = getenv("DCS_DECONV1_REG") ? atoi(getenv("DCS_DECONV1_REG")) : 1;
        if (pool_fused) {   // un-pool + conv1^T: reads the pooled gradient and the routing bits
            const int nqb = (F + 11) / 12;
            hipLaunchKernelGGL((deconv1_reg_kernel<3, 10, true>), dim3((unsigned)dcs_cdiv((int64_t)tc * nqb, kThreads), (unsigned)(n * NB)),
                               dim3(kThreads), 0, ctx->stream, g2, g->W1p, o, d.nf1, C, tc, F, d.w1, nqb, pool_bits, d.wp, pool_mw, NB);
        } else if (g->W1dq && dcs_launch_deconv1_mfma(ctx, g1, g->W1dq, o, n * NB, d.nf1, C, tc, F, d.w1)) {
        } else if (g->W1p && reg_env && d.sw1 == 3 && (d.kw1 + 2) / 3 == 10) {
            const int nqb = (F + 11) / 12;
            hipLaunchKernelGGL((deconv1_reg_kernel<3, 10>), dim3((unsigned)dcs_cdiv((int64_t)tc * nqb, kThreads), (unsigned)(n * NB)),
                               dim3(kThreads), 0, ctx->stream, g1, g->W1p, o, d.nf1, C, tc, F, d.w1, nqb);
        } else if (g->W1p && reg_env && d.sw1 == 4 && (d.kw1 + 3) / 4 == 8) {
            const int nqb = (F + 15) / 16;
            hipLaunchKernelGGL((deconv1_reg_kernel<4, 8>), dim3((unsigned)dcs_cdiv((int64_t)tc * nqb, kThreads), (unsigned)(n * NB)),
                               dim3(kThreads), 0, ctx->stream, g1, g->W1p, o, d.nf1, C, tc, F, d.w1, nqb);
        } else {
            hipLaunchKernelGGL(kern, dim3((unsigned)dcs_cdiv(F, kThreads), (unsigned)(n * NB * tc)), dim3(kThreads), lds,
                               ctx->stream, g1, g->W1c, o, C, tc, F, d.kw1, d.sw1, d.w1);
        }
        tm.done();
    }
    // concat + bias + rectify + mask.  out is [S or CH][n_total][tc][F]; this chunk starts at tile k_first.
    if (mask_mode < 0) {                                 // deferred to mask_ola_kernel (single chunk): o stays in scratch
        g->raw_o = o;
        g->raw_ch = NB * C;
    } else {
        const int64_t plane = (int64_t)tc * F;
        const int CH = NB * C;
        DcsTimer tm(ctx, DCS_TAG_MASK);
        // the kernel indexes out as [ch][n][plane] with n = chunk size; point it at the chunk and pass the
        // total tile count as the channel stride through a strided launch: do it per channel group instead
        // -> simplest exact form: launch with n_total as `n` stride when the chunk is the whole batch.
        if (n == n_total) {
            hipLaunchKernelGGL(mask_kernel, dim3((unsigned)dcs_cdiv(n * plane, kThreads)), dim3(kThreads), 0, ctx->stream,
                               o, g->bout, tiles, out, n, CH, d.S, C, plane, mask_mode, g->mix_sum ? C : 1);
        } else {
            // chunked batch: write into a compact [ch][n][plane] staging area, then scatter rows
            float* stage = (float*)w;
            const int nch = mask_mode == 2 ? CH : d.S;
            hipLaunchKernelGGL(mask_kernel, dim3((unsigned)dcs_cdiv(n * plane, kThreads)), dim3(kThreads), 0, ctx->stream,
                               o, g->bout, tiles, stage, n, CH, d.S, C, plane, mask_mode, g->mix_sum ? C : 1);
            for (int ch = 0; ch < nch; ++ch)
                DCS_HIP(hipMemcpyAsync(out + ((int64_t)ch * n_total + k_first) * plane, stage + (int64_t)ch * n * plane,
                                       (size_t)n * plane * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        tm.done();
    }
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

size_t chunk_bytes(const DcsGenericNet* g, int64_t n) {
    const DcsGenericDims& d = g->d;
    const int NB = d.n_branch;
    const int64_t plane1 = (int64_t)g->tc * d.w1, planep = (int64_t)g->tc * d.wp;
    size_t b = align256((size_t)n * d.nf1 * plane1 * 4) + align256((size_t)n * g->flat_p * 4) +
               align256((size_t)n * g->hid64 * 4) + align256(dcs_gemm_aq_bytes(g->hid64, 176)) + align256((size_t)n * NB * g->flat_p * 4) +
               align256((size_t)n * NB * d.nf1 * planep * 4) + align256((size_t)n * NB * g->C * g->tc * g->F * 4);
    if (d.pool_w) b += align256((size_t)n * d.nf1 * planep * 4) + align256((size_t)n * NB * d.nf1 * plane1 * 4);
    b += align256((size_t)n * NB * g->C * g->tc * g->F * 4);  // mask staging for chunked batches
    return b;
}

}  // namespace

int dcs_generic_set_score_semantics(DcsGenericNet* g, int normalise, int mixture) {
    if (!g) DCS_FAIL(DCS_EINVAL, "null network");
    if ((normalise != DCS_SCORE_NORM_MAX && normalise != DCS_SCORE_NORM_SUM) || (mixture != DCS_MIX_CH0 && mixture != DCS_MIX_SUM))
        DCS_FAIL(DCS_EINVAL, "dcs_model_set_score_semantics: normalise %d, mixture %d", normalise, mixture);
    if (g->C < 2) DCS_FAIL(DCS_EUNSUPPORTED, "score semantics belong to the multi-channel (score-informed) graphs");
    if (g->C > 4) DCS_FAIL(DCS_EUNSUPPORTED, "channel-sum mixture: at most 4 input channels");
    g->score_norm = normalise;
    g->mix_sum = mixture == DCS_MIX_SUM ? 1 : 0;
    return DCS_OK;
}

int dcs_generic_set_conv_f16(DcsGenericNet* g, int on) {
    if (!g) DCS_FAIL(DCS_EINVAL, "null network");
    g->conv_f16 = on ? 1 : 0;
    return DCS_OK;
}

int dcs_generic_forward(DcsGenericNet* g, const float* tiles, int64_t n, int mask_mode, int tie_mode, float* out) {
    if (!g) DCS_FAIL(DCS_EINVAL, "generic forward: null network");
    // Tiles go through the graph in chunks that bound the scratch (the Bach10 graph needs ~13 MB per tile) at 4 GiB of
    // the 288 GB: every chunk re-reads the dense weights (170 MB + 4 x 170 MB for Bach10), so few large chunks.
    static const int64_t chunk_env = getenv("DCS_GENERIC_CHUNK") ? atoll(getenv("DCS_GENERIC_CHUNK")) : 0;
    int64_t chunk = chunk_env > 0 ? chunk_env : (int64_t)(((size_t)4 << 30) / (chunk_bytes(g, 64) / 64 + 1));
    if (chunk < 64 && chunk_env <= 0) chunk = 64;
    // Graphs whose decoder is the fused kernel on channels-last input (Bach10, f32-class score-informed): that layout only comes out of the all-rows dense
    // kernel, which takes 128 .. 176 tiles per launch.  A longer pass would drop to the f32 GEMM for its 256 x 166 650 dense
    // layers AND to the two-kernel decoder -- a 20 s clip several times slower per tile than a 10 s one.  Cut such passes into
    // equal pieces inside the window instead (the dense weights are re-read once per piece: 0.85 GB, ~0.2 ms; the deferred-mask
    // form below is single-chunk, so such a pass takes the separate mask and cross-fade kernels).
    constexpr bool cap_env = true;
    if (cap_env && chunk_env <= 0 && n > 176 && g->flat64 >= 8192 && plans_channels_last(g, nullptr, nullptr)) {
        const int64_t pieces = (n + 175) / 176, per = (n + pieces - 1) / pieces;
        const int64_t capped = per >= 128 ? per : 176;
        if (capped < chunk) chunk = capped;
    }
    if (mask_mode < 0 && n > chunk) return DCS_EUNSUPPORTED;   // deferred mask: one chunk only (quiet: the caller falls back)
    const int64_t per = n < chunk ? n : chunk;
    DCS_CHECK(g->ws.ensure(chunk_bytes(g, per)));
    const int64_t tile_elems = (int64_t)g->C * g->tc * g->F;
    for (int64_t k = 0; k < n; k += chunk) {
        const int64_t m = n - k < chunk ? n - k : chunk;
        DCS_CHECK(forward_chunk(g, tiles + k * tile_elems, m, n, k, mask_mode, tie_mode, out, (char*)g->ws.ptr));
    }
    return DCS_OK;
}

int dcs_generic_separate(DcsGenericNet* g, dcs_stft* plan, const float* audio, int64_t L, int ov, int tiler, float scale,
                         int eps_mode, int tie_mode, float* pcm, float* sep_out, float* mag_out, float* phase_out,
                         int64_t ld_out, DcsBuffer* ws, int64_t n_clips, int64_t audio_stride, const DcsScoreNotes* notes,
                         const int64_t* lens_h, const int64_t* clip_tab_d, int64_t pcm_stride) {
    // The un-fused composition of the public operators: STFT -> tiles -> network -> cross-fade -> iSTFT.  Clips share the
    // launches: one STFT / iSTFT launch over all clips and ONE pass of all their tiles through the network (the dense
    // layers' weights -- 853 MB for Bach10 -- are then read once for the whole group).  lens_h: clips of different lengths
    // (dcs_separate_ragged): L is then the longest, every clip is framed, tiled and cross-faded with its own counts (its
    // tiles sit behind those of the clips before it), the STFT / iSTFT kernels read them from the device table.
    dcs_ctx* ctx = g->ctx;
    const int tc = g->tc, F = g->F, st = tc - ov, S = g->d.S;
    const int64_t T = dcs_frame_count(L, plan->hop);
    const int64_t n = dcs_tile_count(T, tc, ov, tiler);
    const int64_t ld = dcs_round_up(F, 4);
    const int64_t rows = n * st + tc;  // rows of the stitched spectrogram (>= T), of the longest clip
    std::vector<int64_t> Tc((size_t)n_clips, T), nc((size_t)n_clips, n), off((size_t)n_clips + 1, 0);
    for (int64_t c = 0; c < n_clips; ++c) {
        if (lens_h) {
            Tc[c] = dcs_frame_count(lens_h[c], plan->hop);
            nc[c] = dcs_tile_count(Tc[c], tc, ov, tiler);
        }
        off[c + 1] = off[c] + nc[c];
    }
    const int64_t n_all = off[n_clips];
    const size_t b_mag = align256((size_t)n_clips * T * ld * 4), b_unit = 2 * b_mag, b_ph = phase_out ? b_mag : 0;
    const size_t b_tiles = align256((size_t)n_all * g->C * tc * F * 4), b_out = align256((size_t)S * n_all * tc * F * 4);
    const size_t b_sep = align256((size_t)n_clips * S * rows * ld * 4);
    const size_t b_inp = notes ? align256((size_t)g->C * T * F * 4) : 0;   // score-informed network input [C][T][F]
    // one clip through a graph whose conv1 can hand its map to conv2 channels-last: the clip's scaled frames once more as
    // [C][Tn][F] rows (a sixth of the tiles), so that conv1 runs per frame instead of per tile row (forward_chunk: per_frame)
    const int64_t Tn = (n - 1) * st + tc;
#ifdef DCS_EXP_NO_PER_FRAME      // experiment build (scripts/build_exp.sh): the per-tile conv1 of round 5 for a same-box A/B
    const bool want_frames = false;
#else
    const bool want_frames = n_clips == 1 && !lens_h && n >= 2 && st > 0 && st < tc && g->W1m && g->use_colconv && g->C * Tn <= 65535;
#endif
    const size_t b_frames = want_frames ? align256((size_t)g->C * Tn * F * 4) : 0;
    if (notes && (n_clips != 1 || notes->ninst != g->C))
        DCS_FAIL(DCS_EINVAL, "score-informed path: one clip, %d score channels (got %d)", g->C, notes->ninst);
    if (lens_h && (notes || g->C != 1 || !clip_tab_d || phase_out || sep_out || mag_out))
        DCS_FAIL(DCS_EUNSUPPORTED, "clips of different lengths: single-channel graphs, PCM output");
    DCS_CHECK(ws->ensure(b_mag + b_unit + b_ph + b_tiles + b_out + b_sep + b_inp + b_frames));
    char* p = (char*)ws->ptr;
    float* mag = (float*)p; p += b_mag;
    float2* unit = (float2*)p; p += b_unit;
    float* phase = phase_out ? (float*)p : nullptr; p += b_ph;
    float* tiles = (float*)p; p += b_tiles;
    float* outm = (float*)p; p += b_out;
    float* sep = (float*)p; p += b_sep;
    float* inp = (float*)p; p += b_inp;
    float* frames = (float*)p; p += b_frames;
    DCS_CHECK(dcs_launch_stft_forward_f32_clips(plan, audio, L, audio_stride, n_clips, mag, phase, unit, ld, T, T, false, clip_tab_d));
    if (notes) {
        // separate_bach10.py (score-informed) :503-527: scaled magnitudes x filterSpec masks, one channel per instrument,
        // then the C-channel tiles (the masks multiply channel 0 of a tile, as in the script)
        DCS_CHECK(dcs_score_masks_scaled(ctx, mag, ld, T, F, notes->notes_h, notes->ninst, notes->n_notes, notes->width, 0, T,
                                         scale, inp, nullptr, g->score_norm));
        DCS_CHECK(dcs_launch_tile(ctx, inp, T * (int64_t)F, F, g->C, T, F, tc, ov, tiler, 1.0f, tiles, n));
    } else {
        for (int64_t c = 0; c < n_clips; ++c)
            DCS_CHECK(dcs_launch_tile(ctx, mag + c * T * ld, 0, ld, 1, Tc[c], F, tc, ov, tiler, scale, tiles + off[c] * tc * F, nc[c]));
    }
    struct FramesScope {                                 // the frames belong to this call only
        DcsGenericNet* g;
        ~FramesScope() { g->frames_src = nullptr; g->frames_rows = 0; g->frames_st = 0; }
    } frames_scope{g};
    if (want_frames) {
        if (notes) DCS_CHECK(dcs_launch_tile(ctx, inp, T * (int64_t)F, F, g->C, T, F, (int)Tn, 0, tiler, 1.0f, frames, 1));
        else DCS_CHECK(dcs_launch_tile(ctx, mag, 0, ld, 1, T, F, (int)Tn, 0, tiler, scale, frames, 1));
        g->frames_src = frames; g->frames_rows = (int)Tn; g->frames_st = st;
    }
    // mask + cross-fade in one kernel when all tiles go through the graph in one chunk (the masked tiles then never exist)
    static const bool fuse_env = !(getenv("DCS_MASK_OLA") && atoi(getenv("DCS_MASK_OLA")) == 0);
    bool mask_fused = false;
    if (fuse_env && S <= 4) {
        const int rc = dcs_generic_forward(g, tiles, n_all, -1, tie_mode, outm);
        if (rc == DCS_OK) mask_fused = true;
        else if (rc != DCS_EUNSUPPORTED) return rc;
    }
    if (!mask_fused) DCS_CHECK(dcs_generic_forward(g, tiles, n_all, eps_mode, tie_mode, outm));
    if (g->rise_ov != ov) {
        std::vector<float> r(ov > 0 ? ov : 1, 0.f);
        if (ov > 1) {
            const double step = 1.0 / (double)(ov - 1);
            for (int i = 0; i < ov; ++i) r[i] = (float)((double)i * step);
            r[ov - 1] = 1.0f;
        }
        if (g->rise_d) {
            DCS_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipFree(g->rise_d);
            g->rise_d = nullptr;
        }
        DCS_HIP(hipMalloc((void**)&g->rise_d, r.size() * sizeof(float)));
        DCS_HIP(hipMemcpy(g->rise_d, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice));
        g->rise_ov = ov;
    }
    // outm is [S][n_all][tc][F]: clip c's tiles of source s start at (s * n_all + off[c]) tiles.  A clip's stitched rows
    // past its own n_c * st + tc stay unwritten (the iSTFT only reads its own T_c frames)
    if (mask_fused) {
        DcsTimer tm(ctx, DCS_TAG_MASK);
        constexpr int mm_env = 1;   // (0 selected the loop form: 2.6 TB/s on the Bach10 clip against the all-requests-first form)
        const int mmax = (ov + st - 1) / st + 1;             // tiles that can reach one frame
        for (int64_t c = 0; c < n_clips; ++c) {
            const dim3 grid((unsigned)(nc[c] * st + tc), (unsigned)dcs_cdiv(F, kThreads));
#define DCS_MASK_OLA(MM_)                                                                                                   \
            hipLaunchKernelGGL((mask_ola_kernel<MM_>), grid, dim3(kThreads), 0, ctx->stream, g->raw_o, g->bout, tiles, nc[c], off[c], \
                               g->raw_ch, S, g->C, tc, ov, F, g->rise_d, sep + c * S * rows * ld, rows * ld, ld, eps_mode,         \
                               g->mix_sum ? g->C : 1)
            if (mm_env && S == 4 && mmax <= 6) DCS_MASK_OLA(6);
            else DCS_MASK_OLA(0);
#undef DCS_MASK_OLA
        }
        tm.done();
    } else {
        for (int64_t c = 0; c < n_clips; ++c)
            DCS_CHECK(dcs_launch_overlap_add(ctx, outm + off[c] * tc * F, nc[c], S, tc, ov, F, g->rise_d, sep + c * S * rows * ld,
                                             rows * ld, ld, n_all * tc * (int64_t)F));
    }
    // pad columns of sep (F..ld) are never written by the stitch; the iSTFT only reads bins < F
    if (pcm)
        DCS_CHECK(dcs_launch_stft_inverse_f32_clips(plan, sep, rows * ld, unit, T * ld, ld, T, S, n_clips, scale, pcm, L, clip_tab_d,
                                                    lens_h ? pcm_stride : 0));
    for (int s = 0; s < S && sep_out; ++s)
        DCS_HIP(hipMemcpy2DAsync(sep_out + (int64_t)s * T * ld_out, ld_out * 4, sep + (int64_t)s * rows * ld, ld * 4,
                                 (size_t)F * 4, (size_t)T, hipMemcpyDeviceToDevice, ctx->stream));
    if (mag_out)
        DCS_HIP(hipMemcpy2DAsync(mag_out, ld_out * 4, mag, ld * 4, (size_t)F * 4, (size_t)T, hipMemcpyDeviceToDevice,
                                 ctx->stream));
    if (phase_out)
        DCS_HIP(hipMemcpy2DAsync(phase_out, ld_out * 4, phase, ld * 4, (size_t)F * 4, (size_t)T,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    return DCS_OK;
}
