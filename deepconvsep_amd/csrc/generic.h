// Generic build_ca path (strided conv1, optional max-pool, 2-D conv2, large dense layers):
// iKala, Bach10 and score-informed graphs.  Implemented in generic.hip.
#pragma once
#include <stdint.h>

#include <vector>

#include "dcs_internal.h"

struct DcsGenericDims {
    int nf1, kw1, sw1, w1, pool_w, wp, nf2, kh2, kw2, h2, w2, flat, hidden, n_fc, n_branch, S;
    int branch_fc[4];
};

struct DcsGenericNet;

// expandMidi's note table on the host (see dcs_score_masks): the score-informed front-end of the whole-path entry
struct DcsScoreNotes {
    const double* notes_h;
    int ninst, n_notes, width;
};

// column convolution (kh x 1): see colconv_kernel in generic.hip for the operation
struct DcsColConv {
    const float* in; int64_t in_n_stride; int Cin, H, W;
    const float* Wk;            // [kh][32][32] (ci, co swizzled: see colconv_wslot)
    const float* bias;          // [32]
    float* out; int64_t out_n_stride; int Cout, Ho;
    int ph, kh;
    int xb_per_wg;              // column blocks (16 x each) a workgroup walks
    int n_xb;                   // column blocks per image
    int in_f16;                 // 1: `in` holds f16, channels-last with 32 channels per position ([image][H][W][32] halves, in_n_stride in
                                // halves): conv1_mfma_kernel<1, 2> writes it so for colconv_wreg_scatter_kernel (f16 switch)
    int out_f16;                // 1: `out` holds f16 ([image][Cout][Ho][W] halves, out_n_stride in halves) -- the f16 forward conv2 feeding
                                // the f16 bottleneck layer (colconv_wreg_scatter_kernel only: dcs_colconv_wreg_scatter_ok)
};
// slab convolution (general kh x kw): see slabconv_kernel in generic.hip for the operation
struct DcsSlabConv {
    const float* in; int64_t in_n_stride; int Cin, H, W;
    const float* Wk;            // [kh][kw][32][32] (ci, co swizzled: colconv_wslot(0, ci, co) within a tap)
    const float* bias;          // [32]
    float* out; int64_t out_n_stride; int Cout, Ho, Wo;
    int kh, kw, ph, pw;
    int band, n_bands;          // output rows per workgroup, workgroups per image
    int rows_max;               // band + kh - 1 (slab rows allocated)
    int tstage;                 // taps (along v) staged per step
    int pstage;                 // tap pairs staged per step (slabconv_ps.hip)
    int xt, n_xt;               // output columns per workgroup (multiple of 16), column tiles per band (slabconv_ps.hip)
};
// pre-split slab variant (slabconv_ps.hip): false = shape not covered, nothing launched
void dcs_slabconv_ps_pack(const float* Wf, int kh, int kw, int (*wslot)(int, int), int mode, std::vector<uint16_t>* out);
bool dcs_launch_slabconv_ps(dcs_ctx* ctx, DcsSlabConv a, int64_t n_images, const void* Wq, int mode);
// conv1 with a frequency stride of 4 on the bf16 matrix pipe (conv1_mfma.hip): false = shape not covered, nothing launched
void dcs_conv1_mfma_pack(const float* Wc, int NF, int C, int kw, std::vector<uint16_t>* out);
bool dcs_launch_conv1_mfma(dcs_ctx* ctx, const float* x, const void* Wq, const float* bias, float* out, int64_t n, int C,
                           int NF, int tc, int F, int kw, int sw, int w1, bool channels_last = false, bool out_f16 = false);
// conv2 of the Bach10 / score-informed graphs, f32-class, weights in registers (colconv_fwd_x3.hip): channels-last input
bool dcs_colconv_fwd_x3_ok(const DcsColConv& a);
bool dcs_launch_colconv_fwd_x3(dcs_ctx* ctx, const DcsColConv& a, int64_t n_images, const void* Wq);
// InverseLayer(conv1) with a frequency stride of 4 on the bf16 matrix pipe (deconv1_mfma.hip)
void dcs_deconv1_mfma_pack(const float* W1p, int nf1, int C, std::vector<uint16_t>* out);
bool dcs_launch_deconv1_mfma(dcs_ctx* ctx, const float* g, const void* Wq, float* out, int64_t n_images, int NF, int C, int tc,
                             int F, int w1);
// weights-in-registers f16 variant (colconv_wreg.hip): false = shape not covered, nothing launched
void dcs_colconv_wreg_pack(const _Float16* Wh, int kh, std::vector<_Float16>* out);
bool dcs_launch_colconv_wreg(dcs_ctx* ctx, const DcsColConv& a, int64_t n_images, const void* Wq);
bool dcs_colconv_wreg_scatter_ok(const DcsColConv& a);   // the forward ('valid') shape of the weights-in-registers kernel: may write f16
// InverseLayer(conv2) + InverseLayer(conv1) in one kernel (Bach10 graph, f16 switch on): out [image][Ho][F]
void dcs_decoder_fused_pack(const float* W1p, int nf1, int C, std::vector<uint16_t>* out);
bool dcs_decoder_fused_ok(const DcsColConv& a, int F);
// gemm_f16.hip: the per-source dense layers on f16 weights, f16 channels-last output (f16 switch + fused decoder)
size_t dcs_gemm_bh_bytes(int K, int n_out);
int dcs_gemm_pack_bh(dcs_ctx* ctx, const float* B_d, int K, int ldb, int n_out, int nch, int npos, int chpad, void* Bh_d);
int dcs_gemm_pack_bias_cl(dcs_ctx* ctx, const float* bias_d, int n_out, int nch, int npos, int chpad, float* out_d);
int dcs_gemm_pack_bh_plain(dcs_ctx* ctx, const float* B_d, int K, int ldb, int n_cols, void* Bh_d);   // B in its own column order
int dcs_gemm_f16_longk_slices(const dcs_ctx* ctx, int M, int K, int n_cols);                          // 0: the shape is not the long-K kernel's
// A: f32 rows (split into two f16 terms on the way into LDS) or, a_f16, f16 rows of lda halves (lda % 32 == 0, K == lda)
bool dcs_launch_gemm_f16_longk(dcs_ctx* ctx, const void* A, int64_t lda, int M, int K, int n_cols, const void* Bh, float* partial,
                               bool a_f16 = false);
bool dcs_launch_gemm_f16_skinny(dcs_ctx* ctx, const float* Z, int64_t ldz, int M, int K, int n_cols, int n_br, const void* const* Bh,
                                const float* const* bias, void* const* C, int64_t ldc, void* Ah_scratch);
// in_channels_last: a.in is [image][H][W][Cin] (the dense layer wrote a position's channels together) instead of [image][Cin][H][W]
bool dcs_launch_decoder_fused(dcs_ctx* ctx, const DcsColConv& a, int64_t n_images, const void* Wq, const void* Wq1, float* out,
                              int F, bool in_channels_last = false, bool in_f16 = false);   // in_f16: [image][H][W][32] halves (gemm_f16.hip)

// the same fusion with f32-class arithmetic (colconv_x3.hip): two waves per column block, ten taps each as bf16 planes in
// registers; the input MUST be channels-last.  false = shape not covered / not launched
void dcs_decoder_x3_pack(const float* Wf /* [kh][32 out][40], in-channel fastest */, int kh, std::vector<uint16_t>* out);
bool dcs_decoder_x3_ok(const DcsColConv& a, int F, int n_out = 1);
bool dcs_launch_decoder_x3(dcs_ctx* ctx, const DcsColConv& a, int64_t n_images, const void* Wq, const void* Wq1, float* out, int F,
                           int n_out = 1);

int dcs_generic_create(dcs_ctx* ctx, const DcsGenericDims& d, int C, int tc, int F,
                       const std::vector<std::vector<float>>& params, DcsGenericNet** out);
void dcs_generic_destroy(DcsGenericNet* g);
int dcs_generic_set_conv_f16(DcsGenericNet* g, int on);
int dcs_generic_set_score_semantics(DcsGenericNet* g, int normalise, int mixture);
// tiles [n, C, tc, F] -> mask_mode 0/1: out [S, n, tc, F] masked; mask_mode 2: p [n, n_branch*C, tc, F]
int dcs_generic_forward(DcsGenericNet* g, const float* tiles, int64_t n, int mask_mode, int tie_mode, float* out);
// n_clips equal-length clips (clip c at audio + c * audio_stride) go through one set of launches: their tiles are
// stacked into one batch for the network; pcm [n_clips][S][L].  The spectra outputs are single-clip only.
int dcs_generic_separate(DcsGenericNet* g, dcs_stft* plan, const float* audio, int64_t L, int ov, int tiler, float scale,
                         int eps_mode, int tie_mode, float* pcm, float* sep_out, float* mag_out, float* phase_out,
                         int64_t ld_out, DcsBuffer* ws, int64_t n_clips = 1, int64_t audio_stride = 0,
                         const DcsScoreNotes* notes = nullptr,
                         const int64_t* lens_h = nullptr /* host: samples per clip (clips of different lengths) */,
                         const int64_t* clip_tab_d = nullptr /* device {samples, frames, tiles} per clip */,
                         int64_t pcm_stride = 0);
