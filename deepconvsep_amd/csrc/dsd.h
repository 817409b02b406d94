// Launch interface of the DSD decoder kernels (dsd.hip).
#pragma once
#include "dcs_internal.h"

// Transposed-conv2 outputs G are stored as [tile][branch][channel group][t][8]: a (tile, branch, group) block is
// contiguous (the streaming deconv2 kernel writes it in one piece) and a row's 8 channels are two float4
// slots of the final kernel's A operand.
constexpr int kDsdGch = 8;
inline int dsd_g_pitch(int CI, int tc) { return ((CI + kDsdGch - 1) / kDsdGch) * tc * kDsdGch; }  // floats per (tile, branch)

struct DsdFinalArgs {
    const float* G;       // [n][3][CI/8][tc][8] transposed-conv2 outputs
    const float* Bw;      // [CI][ldb]  Bw[c][f] = W1[c,0,0,F-1-f]; ldb = F rounded up to 128, zero padded
    int ldb;
    const float* bias;    // [4] output BiasLayer
    const float* mix;     // mixture magnitudes: FOLD mag[T][mix_ld]; else tiles[n*tc][F]
    int64_t mix_ld;
    float mix_scale;      // scale_factor (FOLD) or 1
    float* out;           // FOLD sep[c][t][out_ld]; else out[c][n*tc][F]
    int64_t out_src_stride;
    int64_t out_ld;
    const float* rise;    // [ov] = np.linspace(0, 1, ov)
    int64_t n;            // tiles
    int64_t rows;         // FOLD: frames to produce; else n*tc
    int tc, ov, st;
    int F, CI;
    int mmax;             // FOLD: ceil(ov/st)+1 covering tiles per frame; else 1
    int mask_mode;        // 0 = convention A, 1 = convention B, 2 = raw network output, 3 = stereo trainer (per input channel)
    int nbr;              // 3: the DSD graph (4th output = branch 1 again); 4: one branch per source (stereo trainer)
    int bias_half;        // 0: bias[4] per source; > 0: two input channels side by side, bins >= bias_half use bias[2 s + 1]
    int n_clips;          // stacked clips of equal length (0 or 1: a single clip); clip c uses G + c*g_clip_stride,
    int64_t g_clip_stride, mix_clip_stride, out_clip_stride;  // mix + c*mix_clip_stride, out + c*out_clip_stride
    // bf16x3 path (dsd_bf16x3.hip): G split into three bf16 planes [item][channel group][t][3][8] and the weights split
    // and laid out per (bin, plane, K block, lane group); null = the f32 kernel
    const void* Gs;
    const void* Bpk;
    int64_t gs_clip_stride;   // 16-byte units
    const int64_t* clip_tab;  // device table (kDcsClipTab entries per clip, dcs_internal.h) when the stacked clips differ in
                              // length (n and rows above are then the maxima that size the grid and the output strides),
                              // else null.  With row offsets >= 0 in it (compact layout) the mixture rows of clip c start at
                              // row offset[c] and its G at tile offset[c] * g_tile_stride (gs_tile_stride: 16-byte units)
    int64_t g_tile_stride, gs_tile_stride;
};

// Bw:  [CP][NG * gcols], column (g, c, dt) -> g*gcols + c*kh + dt holds W2c[co, g*GS + c, dt]   (few tiles)
// Bws: [ci][16 taps][CP] holds W2c[co, ci, dt], ci padded to whole groups of 8, tap 15 zero   (many tiles)
// Gs != null: the outputs are (also, for few tiles) produced as three bf16 planes for the bf16x3 final kernel,
// Gs[item][channel group][t][plane][8 channels] -- 16-byte pieces, dsd_gs_pitch() of them per (tile, branch)
int dcs_launch_dsd_deconv2(dcs_ctx* ctx, const float* D, const float* Bw, const float* Bws, float* G, int64_t n_ks,
                           int H2, int CP, int CI, int kh, int tc, int NG, int GS, int gcols, void* Gs = nullptr,
                           const void* Bq = nullptr);
int dcs_launch_dsd_final(dcs_ctx* ctx, const DsdFinalArgs& a, bool fold);
// column blocks per wave the final kernel will use for this launch (2: 128-bin workgroups, 1: 64-bin ones for few rows)
int dsd_final_cbw(const dcs_ctx* ctx, int64_t rows, int F, int64_t n_clips);
// whether the fused (fold) launch of these dimensions runs the bf16x3 kernel -- the caller then has deconv2 produce the
// split planes (DsdFinalArgs::Gs) instead of f32 G
bool dsd_final_bf16x3(const dcs_ctx* ctx, int64_t rows, int F, int64_t n_clips, int CI, int mask_mode);

// ---- bf16x3 variant of the fused final kernel (dsd_bf16x3.hip; DCS_FINAL_BF16X3=0 switches it off)
inline int dsd_gs_pitch(int CI, int tc) { return ((CI + kDsdGch - 1) / kDsdGch) * tc * 3; }  // 16-byte pieces per (tile, branch)
int dcs_launch_dsd_gsplit(dcs_ctx* ctx, const float* G, void* Gs, int64_t n_items, int tc, int ngg);
// streaming transposed conv2 on the bf16 matrix pipe (many items): D f32 in, the bf16 planes of G out.  Bq: the conv2
// weights as bf16 planes, [channel group][8 channels][16 taps][25 pieces of 16 bytes] (dsd_d2q_bytes() per model)
// 16-byte pieces per (channel, tap) of the packed conv2 weights: 3 planes x 8 K pieces = 24 used, the rest padding.  26: the
// 16 taps of a ds_read_b128 lane group then start in 16 different bank quads (scripts/lds_bank_model.py; 25 is 2-way for the real
// lane groups -- profiles/r03_b_pmc_summary.txt: 1.48 conflict cycles per LDS-active cycle in deconv2_stream_bf16_kernel)
#ifndef DCS_D2Q_TAP_U4
#define DCS_D2Q_TAP_U4 26
#endif
constexpr int kDsdD2qTapU4 = DCS_D2Q_TAP_U4;
constexpr int kDsdD2qGroupBytes = kDsdGch * 16 * kDsdD2qTapU4 * 16;
inline size_t dsd_d2q_bytes(int CI) { return (size_t)((CI + kDsdGch - 1) / kDsdGch) * kDsdD2qGroupBytes; }
int dcs_launch_dsd_deconv2_bf16(dcs_ctx* ctx, const float* D, const void* Bq, void* Gs, int64_t n_ks, int H2, int CP, int CI,
                                int kh, int tc);
int dcs_launch_dsd_final_bf16x3(dcs_ctx* ctx, const DsdFinalArgs& a, int n_colg, int64_t n_wg, unsigned n_clips);
