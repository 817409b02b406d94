// Wave-per-frame float32 STFT kernels (fft_wave.hip) -- the fast path for N = 1024 / 2048 / 4096.
#pragma once
#include "dcs_internal.h"

bool dcs_fft_wave_supported(const dcs_stft* p);
bool dcs_fft_wave_inverse_supported(const dcs_stft* p);  // additionally: hop | N, hop even
// clips of equal length: rows_out rows per clip (stacked), audio of clip c at audio + c * audio_stride
int dcs_fft_wave_forward(dcs_stft* p, const float* audio, int64_t L, int64_t audio_stride, int64_t n_clips, float* mag,
                         float* phase, float2* unit, int64_t ld, int64_t rows_out, int64_t T, bool interleave = false,
                         const int64_t* clip_tab = nullptr /* device: {samples, frames, tiles} per clip */);
// n_src stacked sources; source s uses the phasor rows of clip s / src_per_clip at unit + clip * unit_clip_stride
int dcs_fft_wave_inverse(dcs_stft* p, const float* mag, int64_t src_stride, const float* phase, const float2* unit,
                         int64_t unit_clip_stride, int src_per_clip, int64_t ld, int64_t T, int n_src, float pre_div,
                         float* audio, int64_t n_out, const int64_t* clip_tab = nullptr, int64_t out_stride = 0);
