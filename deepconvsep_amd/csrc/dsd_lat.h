// One-batch ("latency") kernels of the DSD path (dsd_lat.hip): the reference calls predict_function2 once per batch of
// 32 tiles (examples/dsd100/separate_dsd.py:296-298), i.e. ~190 frames and 0.57 GFLOP per call -- 3.6 us of the chip's
// f32 peak.  At that size a kernel's duration is its longest dependent chain (launch -> global loads -> compute ->
// stores), not its throughput, so these kernels are cut for a SHORT chain: every global load of a workgroup is issued
// in one round up front, K is split over the 16 waves of a workgroup instead of looped over, nothing is staged twice.
// They read and write the same HBM buffers (mag, unit, H1, C2, Z, D, G / Gs, sep) as the throughput kernels, so every
// stage can be switched between the two families (dcs_set_latency_stages) and checked against the other.
#pragma once
#include "dcs_internal.h"
#include "dsd.h"

// stage bits of the latency path
enum {
    DCS_LAT_STFT = 1, DCS_LAT_CONV1 = 2, DCS_LAT_CONV2 = 4, DCS_LAT_FC = 8, DCS_LAT_FC1X = 16, DCS_LAT_DECONV2 = 32,
    DCS_LAT_FINAL = 64, DCS_LAT_ISTFT = 128,
    DCS_LAT_ALL = 255
};
// (Round 3 also built conv2 .. conv2^T as ONE launch of 8-workgroup clusters exchanging through tagged granules, and STFT +
// conv1 as one launch: 51.9 / 54.1 us per step against 49.7 -- removed in round 5, numbers in scripts/README.md.)

// C[r][0..n_store) = act(a_scale * A_r[0..K) . B + bias),  A_r = A + r * a_row_stride (K contiguous floats, 16-byte
// aligned).  K is cut into n_slices slices of slice_len (multiple of 4); wave s of a workgroup multiplies slice s of one
// 16 x 16 output block, the slices are added through LDS in slice order.  Bp is B in fragment order
// (dcs_lat_pack_b): [slice][column block][j][lane][4].
// Split over workgroups: nz > 1 cuts the slices into nz groups, one workgroup each (4 waves -- one per matrix pipe --
// instead of 16 on one CU); group z writes its partial sum to C + z * c_part_stride (bias in part 0, no rectifier), and
// the NEXT layer adds the parts while it loads its operand: a_parts = 4, a_part_stride, relu_in.
struct DcsLatGemm {
    const float* A; int64_t a_row_stride; float a_scale;
    const float* Bp; const float* bias;
    float* C; int64_t ldc;
    int M, n_store, K, slice_len, n_slices, n_cb, relu;
    int nz; int64_t c_part_stride;                 // output split (0 / 1: none)
    int a_parts; int64_t a_part_stride; int relu_in;   // operand = sum of a_parts arrays (1 or 4), then rectified if relu_in
    int a_gdiv; int64_t a_gmul;                    // stacked clips: operand row of output row r = (r / a_gdiv) * a_gmul + r % a_gdiv
};
inline int dcs_lat_j(int slice_len) { return (slice_len + 15) / 16; }
// host: B[k][ldb] (k-major, n_cb*16 <= ldb columns) -> fragment order
void dcs_lat_pack_b(const float* B, int ldb, int K, int n_cb, int slice_len, int n_slices, std::vector<float>* out);
int dcs_launch_lat_gemm(dcs_ctx* ctx, const DcsLatGemm& g, int tag);

// transposed conv2 of the DSD graph (H2 = 16, kh = 15, tc = 30, CP = 52): D [items][16][52] -> G f32
// [items][7][30][8] (nullable) and / or its three bf16 planes Gs [items][7][30][3] x 16 bytes (nullable).
// Wp: [56 channels][4][64 lanes][4] fragment order (dcs_lat_pack_deconv2 from Bw2s [ci][16 taps][52]).
void dcs_lat_pack_deconv2(const float* Bw2s, int n_ci8, std::vector<float>* out);
int dcs_launch_lat_deconv2(dcs_ctx* ctx, const float* D, const float* Wp, float* G, void* Gs, int64_t n_items);

// fused transposed conv1 + bias + rectify + soft mask + cross-fade (final_bf16x3_kernel's arithmetic) with 16 rows x 64
// bins per workgroup and every covering tile's A set staged at once; needs a.Gs / a.Bpk, one clip, mask_mode 0 / 1
int dcs_lat_final_max_covers();   // ceil(overlap / stride) + 1 must not exceed this (6)
bool dcs_lat_final_supported(const DsdFinalArgs& a);
int dcs_launch_lat_final(dcs_ctx* ctx, const DsdFinalArgs& a);

// STFT / iSTFT with one workgroup per frame / per (source, hop block); frameSize 1024 / 2048, frameSize / hop in {2, 4}
bool dcs_lat_stft_supported(const dcs_stft* p);
int dcs_launch_lat_stft(dcs_stft* p, const float* audio, int64_t L, float* mag, float* phase /* nullable */, float2* unit,
                        int64_t ld, int64_t rows_out, int64_t T);
// the inverse is two launches (every frame transformed once, then the overlap-add); frames: scratch of
// dcs_lat_istft_scratch_bytes() bytes, [source][frame][frameSize] float32
size_t dcs_lat_istft_scratch_bytes(const dcs_stft* p, int64_t T, int n_src);
int dcs_launch_lat_istft(dcs_stft* p, const float* sep, int64_t src_stride, const float2* unit, int64_t ld, int64_t T,
                         int n_src, float pre_div, float* audio, int64_t n_out, float* frames);
