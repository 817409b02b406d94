// The fused final kernel of dsd.hip on the 16-bit-input matrix pipe with fp32-class results (the default for launches
// large enough for 128-bin workgroups; DCS_FINAL_BF16X3=0 selects the f32 kernel).
//
// Why: scripts/ubench/mfma16_valu.hip (profiles/r01_d_ubench_mfma16_valu.txt) -- v_mfma_f32_16x16x32_bf16 takes 6.9 ns
// per SIMD for 16 384 flop and runs BESIDE the VALU, while the f32 MFMA (13.9 ns for 2 048 flop) executes on the
// vector FMA lanes and serialises with every VALU instruction of the epilogue.
//
// How: every f32 operand is split exactly into three bf16 terms by truncation, x = hi + mid + lo (8 + 8 + 8
// significand bits), and the product sum keeps the six term pairs above 2^-24:
//     a.b ~= a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)          dropped: a1 b2 + a2 b1 + a2 b2 <= 3 * 2^-24 |a b|
// Each pair is a K = 64 (50 channels, zero padded) bf16 MFMA chain with f32 accumulation: 12 MFMAs per (branch,
// column block, covering tile) instead of 13 f32 ones.  bf16 products are exact in f32, so the only roundings are the
// accumulator's -- the class of the f32 kernel.  G arrives already split: the streaming deconv2 kernel writes the three
// planes itself (Gs[item][channel group][t][plane][8 channels], 16-byte pieces; 6 instead of 4 bytes per value and no
// f32 copy), the one-shot deconv2 kernel of small launches is followed by g_split_kernel; the transposed-conv1 weights
// are split when the model is packed.
//
// Everything else -- row/tile bookkeeping, cross-fade tables, soft mask, fold, stores -- is final_kernel<true, MODE, 2, 3>.
#include "dcs_internal.h"
#include "dsd.h"

#include <stdlib.h>

#ifdef DCS_FINAL_TRACE
// in-kernel timeline (scripts/build_exp.sh finaltrace dsd_bf16x3.hip -DDCS_FINAL_TRACE; scripts/gpu_final_trace.py): s_memtime
// stamps of wave 0 of the middle workgroup of the middle clip, read back through final_trace_dump()
__device__ unsigned long long final_trace_buf[64];
#define FT_STAMP(slot)                                                                                      \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2)                 \
            final_trace_buf[slot] = __builtin_amdgcn_s_memtime();                                           \
    } while (0)
#else
#define FT_STAMP(slot)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
constexpr int kNgg = 7;                 // channel groups of 8 that carry data (50 filters -> 56); group 7 of the K = 64 axis is zero

__device__ __forceinline__ unsigned bf_trunc(float x) { return __float_as_uint(x) & 0xffff0000u; }

// G f32 [item][channel group][t][8]  ->  Gs bf16 [item][channel group][t][plane][8], x = plane0 + plane1 + plane2 exactly
__global__ __launch_bounds__(kThreads) void g_split_kernel(const float* __restrict__ G, u32x4* __restrict__ Gs,
                                                           int64_t n_rows /* items * channel groups * tc */) {
    const int64_t it = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (it >= n_rows) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(G + it * 8);
    const f32x4 x0 = p[0], x1 = p[1];
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = x0[j];
        v[4 + j] = x1[j];
    }
    unsigned pl[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned h = bf_trunc(v[j]);
        const float r1 = v[j] - __uint_as_float(h);      // exact
        const unsigned m = bf_trunc(r1);
        const float r2 = r1 - __uint_as_float(m);        // exact, at most 8 significant bits left
        pl[0][j] = h;
        pl[1][j] = m;
        pl[2][j] = bf_trunc(r2);
    }
#pragma unroll
    for (int q3 = 0; q3 < 3; ++q3) {
        u32x4 w;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = (pl[q3][2 * q] >> 16) | (pl[q3][2 * q + 1] & 0xffff0000u);
        Gs[it * 3 + q3] = w;
    }
}

__device__ __forceinline__ f32x2 max2(f32x2 x, float lo) {
    const int l = __builtin_bit_cast(int, lo);
    const i32x2 xi = __builtin_bit_cast(i32x2, x);
    const i32x2 r = {xi[0] > l ? xi[0] : l, xi[1] > l ? xi[1] : l};
    return __builtin_bit_cast(f32x2, r);
}

// 16 bytes per lane from global memory straight into LDS (lane l lands at lds + 16 l; lds is wave-uniform).  A __device__
// function of its own: called from a __global__ template the builtin silently suppresses the HOST stub of the kernel.
__device__ __forceinline__ void lds_dma16(const u32x4* src, u32x4* lds) { __builtin_amdgcn_global_load_lds(src, lds, 16, 0, 0); }

__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The A set of a covering tile goes from HBM straight into LDS (global_load_lds_dwordx4: lane l's 16 bytes land at M0 + 16 l, scripts/ubench/lds_async.hip)
// in PIECE-major order -- unit (sp, i) = piece column sp = branch * 21 + plane * 7 + group of row i at sp * 16 + i:
//   * no prefetch registers and no LDS destination registers: 168 registers = THREE workgroups per CU (round 3's form, which
//     staged the pieces through registers, needed 188, spilled 21 when sized for three and measured 65 vs 58.5 us at 640 tiles;
//     removed in round 5);
//   * no ds_write of the staged pieces, no second pass over them in registers;
//   * an MFMA A fragment (row fi, K piece kq) of (branch, plane, K block 0) is unit (..)*16 + 16 kq + fi = a constant + the
//     LANE index: every ds_read_b128 of the kernel reads 1 KB of consecutive LDS -- conflict-free by construction;
//   * K channels 56..63 (K block 1, kq = 3) do not exist in G: their B rows are zero (Bpk, net.hip), so those lanes read
//     K piece 3 of the SAME row again -- finite numbers times zero -- and no LDS is zero-filled.
template <int MODE>
__global__ __launch_bounds__(kThreads, 3) void final_bf16x3_kernel(const DsdFinalArgs a, int n_colg) {
    constexpr int CBW = 2, NBR = 3;
    constexpr int kABuf = 1024;
    constexpr int kMaxM = 16;
    // the two buffers are two ARRAYS and the loop over the covering tiles is unrolled by two, so that the compiler
    // sees transfers into one array and fragment reads from the other -- with one array and a buffer index it must assume
    // that an LDS read may alias a transfer in flight and waits for vmcnt(0) in front of the first ds_read of every tile
    __shared__ u32x4 As[kABuf];
    __shared__ u32x4 As1[kABuf];
    __shared__ __attribute__((aligned(16))) float up_t[kMaxM * 16];
    __shared__ __attribute__((aligned(16))) float down_t[kMaxM * 16];
    __shared__ int meta_k0[16];
    __shared__ int meta_j0[16];
    __shared__ int meta_mlim[16];

    FT_STAMP(0);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, kq = lane >> 4;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const unsigned swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int64_t clip = blockIdx.y;
    const unsigned rg = swz / (unsigned)n_colg;
    const int row0 = (int)rg * 16;        // rows, tiles and workgroups all fit 31 bits (checked by the launcher)
    const int colw = (int)(swz - rg * (unsigned)n_colg) * (64 * CBW) + ((wave + (int)rg) & 3) * (16 * CBW);
    const int col = colw + CBW * fi;
    const int tc = a.tc, st = a.st, ov = a.ov, mmax = a.mmax;
    int n = (int)a.n, rows = (int)a.rows;
    if (a.clip_tab) {
        rows = (int)a.clip_tab[kDcsClipTab * clip + 1];
        n = (int)a.clip_tab[kDcsClipTab * clip + 2];
        if (row0 >= rows) return;
    }
    // where this clip's mixture rows and G planes live: uniform pitch, or the compact layout's per-clip offsets
    int64_t mix_off = clip * a.mix_clip_stride, gs_off = clip * a.gs_clip_stride;
    if (a.clip_tab && a.clip_tab[kDcsClipTab * clip + 3] >= 0) {
        mix_off = a.clip_tab[kDcsClipTab * clip + 3] * a.mix_ld;
        gs_off = a.clip_tab[kDcsClipTab * clip + 4] * a.gs_tile_stride;
    }

    if (tid < 16 * mmax) {
        const int i = tid & 15, m = tid >> 4;
        const int r = row0 + i;
        int k0 = 0, j0 = -1;
        if (r < rows) {
            int kk = (r < ov) ? 0 : (int)((unsigned)(r - ov) / (unsigned)st);
            if (kk > n - 1) kk = n - 1;
            const int jj = r - kk * st;
            if (jj < tc) {
                k0 = kk;
                j0 = jj;
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

    const bool live = colw < a.F;
    // B fragments: the three planes of Bw[c][bin] for this lane's two bins and K-quarter, constant for the workgroup
    // (one lane-dependent address; the 12 pieces of a lane are compile-time offsets from it)
    const u32x4* Bpk = reinterpret_cast<const u32x4*>(a.Bpk) + (col * 24 + kq);
    u32x4 breg[CBW][3][2];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                breg[cb][p][kb] = live ? Bpk[cb * 24 + (p * 2 + kb) * 4] : u32x4{0u, 0u, 0u, 0u};

    const bool vec = ((a.mix_ld | a.out_ld) & 1) == 0;
    f32x4 mixv[CBW];
    const float* mix0 = a.mix + mix_off + (int64_t)row0 * a.mix_ld;
    const int rows_here = rows - row0 < 16 ? rows - row0 : 16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ri = kq * 4 + e;
        float m0 = 0.f, m1 = 0.f;
        if (ri < rows_here) {
            const float* mp = mix0 + (ri * (int)a.mix_ld + col);
            if (vec && col + 1 < a.F) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(mp);
                m0 = v[0];
                m1 = v[1];
            } else {
                if (col < a.F) m0 = mp[0];
                if (col + 1 < a.F) m1 = mp[1];
            }
        }
        mixv[0][e] = a.mix_scale * m0;
        mixv[1][e] = a.mix_scale * m1;
    }
    const float bias0 = a.bias[0], bias1 = a.bias[1], bias2 = a.bias[2], bias3 = a.bias[3];
    const float eps_r = 5e-19f;

    f32x4 res[CBW][4];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int c = 0; c < 4; ++c) res[cb][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    FT_STAMP(1);    // tables written, B fragments and mixture rows requested
    __syncthreads();
    FT_STAMP(2);    // table barrier passed

    // staging plan: the A set of a covering tile is [3 branches][16 rows][3 planes][7 channel groups] 16-byte pieces
    // (1008; the 8th group of the K = 64 axis is zeroed once below).  Slot idx -> (branch s, piece pg = plane * 7 + g, row i)
    // with the row fastest: 16 consecutive lanes fetch the same piece of 16 consecutive frames, 48 bytes apart.
    constexpr int slots = NBR * 16 * 3 * kNgg;               // 1008
    constexpr int NSL = (slots + kThreads - 1) / kThreads;   // 4
    const int m_delta = (NBR * kNgg * tc - st) * 3;          // 16-byte units from covering tile m to m + 1
    const int kbase = meta_k0[0];
    const u32x4* gbase = reinterpret_cast<const u32x4*>(a.Gs) + gs_off + (int64_t)kbase * NBR * kNgg * tc * 3;
    // every slot is loaded for every covering tile, with the tile index clamped to the last one that has a weight on
    // the slot's row (see final_kernel in dsd.hip): no data-dependent branch in front of the loads
    int goff[NSL], mlim[NSL];
    bool in_slot[NSL];
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
        const int idx = tid + u * kThreads;
        const int i = idx & 15, sp = idx >> 4;               // sp = s * 21 + plane * 7 + g
        const int s = sp / (3 * kNgg), pg = sp - s * (3 * kNgg);
        const int plane = pg / kNgg, g = pg - plane * kNgg;
        const bool in = (u + 1) * kThreads <= slots || idx < slots;   // compile-time true for all but the last slot
        in_slot[u] = in;
        const int lim = in ? meta_mlim[i] : -1;
        const int j0 = lim >= 0 ? meta_j0[i] : 0;
        const int dk = lim >= 0 ? meta_k0[i] - kbase : 0;
        mlim[u] = lim >= 0 ? lim : 0;
        goff[u] = (((dk * NBR + s) * kNgg + g) * tc + j0) * 3 + plane;
    }
    // slot idx = tid + 256 u IS LDS unit idx (row fastest, then the piece column): wave w's transfer u covers units
    // 256 u + 64 w .. + 63, a wave-uniform LDS base
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define DCS_LOAD_A_DIRECT(m_, buf_)                                                             \
    _Pragma("unroll") for (int u = 0; u < NSL; ++u) {                                           \
        if ((u + 1) * kThreads <= slots || in_slot[u]) {                                        \
            const int mm = (m_) < mlim[u] ? (m_) : mlim[u];                                     \
            lds_dma16(gbase + (goff[u] + mm * m_delta), (buf_) + (u * kThreads + wave_u * 64)); \
        }                                                                                       \
    }

    // fragment bases: K block 0 -> unit lane (= 16 kq + fi), K block 1 -> group 4 + kq, or group 3 again for kq = 3
    const int lane1 = kq < 3 ? lane + 64 : lane;
    // one covering tile: fragments from a0 (K block 0) / a1 (K block 1), 72 MFMAs, mask + fold in registers
    auto compute = [&](const int m, const u32x4* a0, const u32x4* a1) {
        f32x4 acc[NBR][CBW];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
            acc[0][cb] = f32x4{bias0, bias0, bias0, bias0};
            acc[1][cb] = f32x4{bias1, bias1, bias1, bias1};
            acc[2][cb] = f32x4{bias2, bias2, bias2, bias2};
        }
#pragma unroll
        for (int s = 0; s < NBR; ++s) {
            u32x4 af[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    af[p][kb] = (kb ? a1 : a0)[(s * 3 * kNgg + p * kNgg) * 16];
                }
            // smallest terms first; the two column blocks alternate so that no MFMA waits for the one before it
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) acc[s][cb] = mma(af[2][kb], breg[cb][0][kb], acc[s][cb]);
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) acc[s][cb] = mma(af[0][kb], breg[cb][2][kb], acc[s][cb]);
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) acc[s][cb] = mma(af[1][kb], breg[cb][1][kb], acc[s][cb]);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) acc[s][cb] = mma(af[1][kb], breg[cb][0][kb], acc[s][cb]);
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) acc[s][cb] = mma(af[0][kb], breg[cb][1][kb], acc[s][cb]);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) acc[s][cb] = mma(af[0][kb], breg[cb][0][kb], acc[s][cb]);
        }
        const f32x4 up4 = *reinterpret_cast<const f32x4*>(up_t + m * 16 + kq * 4);
        const f32x4 down4 = *reinterpret_cast<const f32x4*>(down_t + m * 16 + kq * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 up = {up4[2 * h], up4[2 * h + 1]};
            const f32x2 down = {down4[2 * h], down4[2 * h + 1]};
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                const f32x2 x0 = {acc[0][cb][2 * h], acc[0][cb][2 * h + 1]};
                const f32x2 x1 = {acc[1][cb][2 * h], acc[1][cb][2 * h + 1]};
                const f32x2 x2 = {acc[2][cb][2 * h], acc[2][cb][2 * h + 1]};
                const f32x2 x3 = x1 + (bias3 - bias1);
                const float lo = MODE == 0 ? eps_r : 0.f;
                const f32x2 p0 = max2(x0, lo), p1 = max2(x1, lo), p2 = max2(x2, lo), p3 = max2(x3, lo);
                const f32x2 mu = f32x2{mixv[cb][2 * h], mixv[cb][2 * h + 1]} * up;
                f32x2 w;
                if (MODE == 0) {
                    const f32x2 den = ((p0 + p1) + p2) + p3;
                    w = f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])} * mu;
                } else {
                    const f32x2 den = (((p0 + p1) + p2) + p3) + eps_r;
                    w = f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])} * mu;
                }
                const f32x2 o0 = __builtin_elementwise_fma(down, f32x2{res[cb][0][2 * h], res[cb][0][2 * h + 1]}, p0 * w);
                const f32x2 o1 = __builtin_elementwise_fma(down, f32x2{res[cb][1][2 * h], res[cb][1][2 * h + 1]}, p1 * w);
                const f32x2 o2 = __builtin_elementwise_fma(down, f32x2{res[cb][2][2 * h], res[cb][2][2 * h + 1]}, p2 * w);
                const f32x2 o3 = __builtin_elementwise_fma(down, f32x2{res[cb][3][2 * h], res[cb][3][2 * h + 1]}, p3 * w);
                res[cb][0][2 * h] = o0[0]; res[cb][0][2 * h + 1] = o0[1];
                res[cb][1][2 * h] = o1[0]; res[cb][1][2 * h + 1] = o1[1];
                res[cb][2][2 * h] = o2[0]; res[cb][2][2 * h + 1] = o2[1];
                res[cb][3][2 * h] = o3[0]; res[cb][3][2 * h + 1] = o3[1];
            }
        }
    };
    // the A set of covering tile m + 1 is requested while tile m is multiplied (requesting it two tiles ahead costs 20
    // more registers and measured slower: 0.360 vs 0.346 ms at 4096 tiles)
    {
        DCS_LOAD_A_DIRECT(0, As)
        FT_STAMP(3);    // staging plan made, first A set requested
        for (int m = 0; m < mmax; m += 2) {
            // tile m's transfers (issued one compute phase ago) have landed -- here, and behind the barrier in the other
            // waves; the barrier also says that everybody is done with tile m - 1, whose buffer tile m + 1 now overwrites
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FT_STAMP(4 + 4 * m);     // own transfers of tile m landed
            __syncthreads();
            FT_STAMP(5 + 4 * m);     // barrier passed
            if (m + 1 < mmax) DCS_LOAD_A_DIRECT(m + 1, As1)
            FT_STAMP(6 + 4 * m);     // next tile requested
            if (live) compute(m, As + lane, As + lane1);
            FT_STAMP(7 + 4 * m);     // fragments read, 72 MFMAs, mask + fold done
            if (m + 1 >= mmax) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FT_STAMP(8 + 4 * m);
            __syncthreads();
            FT_STAMP(9 + 4 * m);
            if (m + 2 < mmax) DCS_LOAD_A_DIRECT(m + 2, As)
            FT_STAMP(10 + 4 * m);
            if (live) compute(m + 1, As1 + lane, As1 + lane1);
            FT_STAMP(11 + 4 * m);
        }
    }
#undef DCS_LOAD_A_DIRECT

    FT_STAMP(28);   // every covering tile folded
    // per source a workgroup-uniform base (scalar registers) plus one 32-bit lane offset per row
    float* out0 = a.out + clip * a.out_clip_stride + (int64_t)row0 * a.out_ld;
    float* outc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) outc[c] = out0 + c * a.out_src_stride;
    if (vec && colw + 16 * CBW <= a.F) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ri = kq * 4 + e;
            if (ri < rows_here) {
                const int off = ri * (int)a.out_ld + col;
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x2*>(outc[c] + off) = f32x2{res[0][c][e], res[1][c][e]};
            }
        }
    } else if (col < a.F) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ri = kq * 4 + e;
            if (ri < rows_here) {
                const int off = ri * (int)a.out_ld + col;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    outc[c][off] = res[0][c][e];
                    if (col + 1 < a.F) outc[c][off + 1] = res[1][c][e];
                }
            }
        }
    }
#ifdef DCS_FINAL_TRACE
    FT_STAMP(29);   // stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FT_STAMP(30);   // stores acknowledged
#endif
}

// ------------------------------------------------------------------------------------------------
// Streaming transposed conv2 (deconv2_stream_kernel of dsd.hip) on the bf16 matrix pipe.  Same decomposition -- per
// (tile, branch) item and input channel c the product P[t'][dt] = sum_co D[t'][co] W2[co, c, dt] (M = 16 positions,
// N = 16 tap slots, K = 50 filters padded to 64), written skewed so that the col2im sum is a row sum, 8 channels per
// workgroup, independent waves streaming over items -- but every operand is split exactly into three bf16 terms and
// the six products above 2^-24 are accumulated in f32 (12 MFMAs of 16 cycles per channel instead of 13 of 32), and
// the VALU work (splitting the item's D rows, skew / row sums, packing the outputs) runs beside the matrix pipe.
//   A: lane (t' = fi, kq) holds filters 8 kq .. 8 kq + 7 (+ 32 for the second K block) of row t' of D, split in
//      registers when the item is consumed; rows are fetched two items ahead.
//   B: the group's weights as bf16 planes in LDS, Bq[channel][tap][plane][K piece] 16-byte pieces, tap stride 25 pieces
//      (100 words: the 16 taps of a 16-lane pass start in 16 different bank quads), copied from the packed global array.
//   Output: the three bf16 planes of G, Gs[item][channel group][t][plane][8] (what final_bf16x3_kernel stages).
// ------------------------------------------------------------------------------------------------
constexpr int kD2TapU4 = kDsdD2qTapU4;              // 16-byte pieces per (channel, tap): 3 planes x 8 K pieces + padding (dsd.h)
constexpr int kD2ChanU4 = 16 * kD2TapU4;           // per channel
constexpr int kD2GroupU4 = kDsdGch * kD2ChanU4;    // per channel group: 3200 pieces = 51 200 bytes
constexpr int kD2PsStride = 20;                    // floats per skewed row: 16 taps + 4
constexpr int kD2PsChan = 32 * kD2PsStride + 32;   // second channel of a pair starts 32 banks further
constexpr int kD2PsSize = 2 * kD2PsChan;

__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4& hi, u32x4& mid, u32x4& lo) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? x0[j] : x1[j - 4];
        h[j] = bf_trunc(x);
        const float r1 = x - __uint_as_float(h[j]);
        m[j] = bf_trunc(r1);
        l[j] = bf_trunc(r1 - __uint_as_float(m[j]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = (h[2 * q] >> 16) | h[2 * q + 1];
        mid[q] = (m[2 * q] >> 16) | m[2 * q + 1];
        lo[q] = (l[2 * q] >> 16) | l[2 * q + 1];
    }
}

// NW waves per workgroup share one copy of the group's weights (53 KB); a wave's own skew / output buffers are 6.4 KB.  NW = 4:
// two workgroups per CU = 2 waves per SIMD; NW = 16: one workgroup per CU (152 KB) = 4 waves per SIMD -- the kernel has no
// barrier after the weight copy, so the only price of the big workgroup is that copy's barrier.
template <int NW>
__global__ __launch_bounds__(64 * NW, NW <= 4 ? 2 : 1) void deconv2_stream_bf16_kernel(const float* __restrict__ D,
                                                                          const u32x4* __restrict__ Bq,
                                                                          u32x4* __restrict__ Gs, int64_t n_ks, int H2,
                                                                          int kh, int tc, int ngg, int n_full, int X,
                                                                          int tail_ch, int Xt) {
    constexpr int CP = 52, GS = kDsdGch;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u32x4* Bs = reinterpret_cast<u32x4*>(smem_raw);                 // [GS][16][25]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Ps = reinterpret_cast<float*>(Bs + kD2GroupU4) + wave * (kD2PsSize + 32 * GS);  // [2][32][20] skewed products
    float* Os = Ps + kD2PsSize;                                                             // [tc][8] outputs of the item
    const int fi = lane & 15, kq = lane >> 4;

    // block -> (channel group g, workgroup column x of nwx): as deconv2_stream_kernel
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
        const u32x4* src = Bq + (int64_t)g * kD2GroupU4;
        for (int i = tid; i < kD2GroupU4; i += 64 * NW) Bs[i] = src[i];
        for (int i = lane; i < kD2PsSize + 32 * GS; i += 64) Ps[i] = 0.f;
    }
    __syncthreads();

    const int n_pairs = (n_ch + 1) >> 1;
    // raw D rows of an item: filters 8 kq .. +7 (K block 0) and 32 + 8 kq .. +7 (K block 1; filters >= 52 do not exist)
    f32x4 ra[4], rb[4];
#define DCS_LOAD_D(ks_, r_)                                                                             \
    {                                                                                                   \
        const float* dp = D + ((ks_) * (int64_t)H2 + fi) * CP + 8 * kq;                                 \
        const bool in = fi < H2;                                                                        \
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
        r_[0] = in ? *reinterpret_cast<const f32x4*>(dp) : z;                                           \
        r_[1] = in ? *reinterpret_cast<const f32x4*>(dp + 4) : z;                                       \
        r_[2] = (in && kq < 3) ? *reinterpret_cast<const f32x4*>(dp + 32) : z;                          \
        r_[3] = (in && kq < 2) ? *reinterpret_cast<const f32x4*>(dp + 36) : z;                          \
    }
    const int64_t kstep = (int64_t)nwx * NW;
    int64_t ks = (int64_t)x * NW + wave;
    if (ks < n_ks) DCS_LOAD_D(ks, ra)
    if (ks + kstep < n_ks) DCS_LOAD_D(ks + kstep, rb)
    const u32x4* bl = Bs + fi * kD2TapU4 + kq;      // this lane's tap and K piece
    for (; ks < n_ks; ks += kstep) {
        u32x4 ap[3][2];                              // A planes x K blocks
        split8(ra[0], ra[1], ap[0][0], ap[1][0], ap[2][0]);
        split8(ra[2], ra[3], ap[0][1], ap[1][1], ap[2][1]);
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[u] = rb[u];
        if (ks + 2 * kstep < n_ks) DCS_LOAD_D(ks + 2 * kstep, rb)
        for (int cp = 0; cp < n_pairs; ++cp) {
            const u32x4* b0 = bl + (2 * cp) * kD2ChanU4;
            const u32x4* b1 = b0 + kD2ChanU4;
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            // smallest terms first; the two channels alternate so that no MFMA waits for the one before it
#define DCS_D2_TERM(pa_, pb_)                                                                           \
            _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                          \
                acc0 = mma(ap[pa_][kb], b0[(pb_) * 8 + kb * 4], acc0);                                  \
                acc1 = mma(ap[pa_][kb], b1[(pb_) * 8 + kb * 4], acc1);                                  \
            }
            DCS_D2_TERM(2, 0) DCS_D2_TERM(0, 2) DCS_D2_TERM(1, 1) DCS_D2_TERM(1, 0) DCS_D2_TERM(0, 1) DCS_D2_TERM(0, 0)
#undef DCS_D2_TERM
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
        // lane -> (t, half of the 8 channels): 4 values, all three planes; a (t, plane) piece is 16 bytes
        if (lane < 2 * tc) {
            const int t = lane >> 1, half = lane & 1;
            const f32x4 v = reinterpret_cast<const f32x4*>(Os)[lane];
            unsigned h[4], md[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = bf_trunc(v[j]);
                const float r1 = v[j] - __uint_as_float(h[j]);
                md[j] = bf_trunc(r1);
                lo[j] = bf_trunc(r1 - __uint_as_float(md[j]));
            }
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2* dst = reinterpret_cast<u32x2*>(Gs) + (((ks * ngg + g) * (int64_t)tc + t) * 3) * 2 + half;
            dst[0] = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
            dst[2] = u32x2{(md[0] >> 16) | md[1], (md[2] >> 16) | md[3]};
            dst[4] = u32x2{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3]};
        }
    }
#undef DCS_LOAD_D
}

}  // namespace

int dcs_launch_dsd_deconv2_bf16(dcs_ctx* ctx, const float* D, const void* Bq, void* Gs, int64_t n_ks, int H2, int CP,
                                int CI, int kh, int tc) {
    if (n_ks <= 0) return DCS_OK;
    if (CP != 52 || H2 > 16 || kh > 16 || tc > 32 || H2 + kh - 1 > 32)
        DCS_FAIL(DCS_EUNSUPPORTED, "bf16 deconv2: built for 50 conv2 filters and tc <= 32");
    const int ngg = (CI + kDsdGch - 1) / kDsdGch;
    const int n_full = CI / kDsdGch, tail_ch = CI - n_full * kDsdGch;
    // waves per workgroup: 16 (one workgroup per CU, 4 waves per SIMD) from 1024 (tile, branch) items on, else 4 (two
    // workgroups per CU).  Round 3 measured the big workgroup 10-13 % faster for THIS kernel but the register-staged final
    // kernel behind it 4-8 % slower, a wash; with the direct-to-LDS final kernel of round 4 the step gains: 20 x 32 tiles
    // deconv2 33.2 -> 29.6 us with the final kernel unchanged (63.1 / 63.3), 4096 tiles 0.8526 -> 0.8449 ms per clip
    // (profiles/r04_b_*).  DCS_DECONV2_WAVES=4 | 16 forces one.
    constexpr int nw_env = 0;
    const int nw = nw_env == 16 ? 16 : (nw_env == 4 ? 4 : (n_ks >= 1024 ? 16 : 4));
    const int slots = (nw == 16 ? 1 : 2) * ctx->n_cu;   // workgroups resident at once (77 / 152 KB of LDS each), one round
    const double units = n_full + (tail_ch ? (double)((tail_ch + 1) / 2) / (kDsdGch / 2) : 0.0);
    int X = (int)(slots / units) / 8 * 8;
    if (X < 8) X = 8;
    int Xt = tail_ch ? slots - n_full * X : 0;
    if (tail_ch && Xt < 1) Xt = 1;
    const size_t lds = (size_t)kD2GroupU4 * 16 + (size_t)nw * (kD2PsSize + 32 * kDsdGch) * sizeof(float);
    auto kern = nw == 16 ? deconv2_stream_bf16_kernel<16> : deconv2_stream_bf16_kernel<4>;
    DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_full * X + Xt)), dim3(64 * nw), lds, ctx->stream, D,
                       reinterpret_cast<const u32x4*>(Bq), reinterpret_cast<u32x4*>(Gs), n_ks, H2, kh, tc, ngg, n_full, X,
                       tail_ch, Xt);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

int dcs_launch_dsd_gsplit(dcs_ctx* ctx, const float* G, void* Gs, int64_t n_items, int tc, int ngg) {
    if (n_items <= 0) return DCS_OK;
    if (ngg != kNgg) DCS_FAIL(DCS_EUNSUPPORTED, "bf16x3 final kernel: built for 50 conv1 filters (7 channel groups), got %d", ngg);
    const int64_t n_rows = n_items * ngg * tc;
    hipLaunchKernelGGL(g_split_kernel, dim3((unsigned)dcs_cdiv(n_rows, kThreads)), dim3(kThreads), 0, ctx->stream, G,
                       reinterpret_cast<u32x4*>(Gs), n_rows);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

int dcs_launch_dsd_final_bf16x3(dcs_ctx* ctx, const DsdFinalArgs& a, int n_colg, int64_t n_wg, unsigned n_clips) {
    const dim3 grid((unsigned)n_wg, n_clips), block(kThreads);
    if (a.mask_mode == 0) hipLaunchKernelGGL((final_bf16x3_kernel<0>), grid, block, 0, ctx->stream, a, n_colg);
    else hipLaunchKernelGGL((final_bf16x3_kernel<1>), grid, block, 0, ctx->stream, a, n_colg);
    DCS_HIP(hipGetLastError());
    return DCS_OK;
}

extern "C" DCS_API int final_trace_dump(unsigned long long* out, int n) {
#ifdef DCS_FINAL_TRACE
    if (!out || n < 64) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(final_trace_buf), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
#else
    (void)out; (void)n;
    return -2;
#endif
}
