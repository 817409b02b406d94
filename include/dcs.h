/*
 * libdcs -- MI355X (gfx950) separation hot path of MTG/DeepConvSep behind a C ABI.
 *
 * The reference (pure Python 2 + Theano/Lasagne) has no FFI of its own; its
 * integration surfaces are two Python call surfaces:
 *
 *   transformFFT.compute_file / compute_inverse          transform.py:224-274
 *   the separate_*.py train_auto() body                  examples/dsd100/separate_dsd.py:239-313
 *     compute_file -> generate_overlapadd -> predict_function2 ->
 *     overlapadd_multi -> compute_inverse
 *
 * Every entry point below names the reference function whose arithmetic it
 * replaces.  The Python package deepconvsep_amd/ binds these with ctypes and
 * re-creates the reference's function signatures on top.
 *
 * Conventions
 *   - plain C, no exceptions; every function returns 0 (DCS_OK) or a negative
 *     dcs_status; dcs_last_error() gives a thread-local message.
 *   - pointers suffixed _d are DEVICE pointers (HBM), _h are host pointers.
 *     The caller owns every buffer it passes in.
 *   - a dcs_ctx binds one device + one HIP stream; all work of the handles
 *     created from it is enqueued on that stream, asynchronously.  A ctx (and
 *     its plans / models) is not thread-safe; distinct ctx objects are.
 *   - spectrogram matrices are row-major [frames, ld] with ld >= bins; the
 *     "dense" layout of the reference is ld == bins.
 */
#ifndef DCS_H
#define DCS_H

#include <stdint.h>

/* libdcs.so is built with -fvisibility=hidden: the entry points below are the ONLY dynamic symbols it exports. */
#define DCS_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dcs_ctx dcs_ctx;
typedef struct dcs_stft dcs_stft;
typedef struct dcs_model dcs_model;

typedef enum {
    DCS_OK = 0,
    DCS_EINVAL = -1,       /* bad argument (shape, size, null pointer)            */
    DCS_EUNSUPPORTED = -2, /* legal in the reference, not built here (see DESIGN) */
    DCS_EHIP = -3,         /* a HIP runtime call failed                            */
    DCS_ENOMEM = -4,       /* device allocation failed                             */
    DCS_ESHAPE = -5        /* parameter count/shape mismatch (set_all_param_values) */
} dcs_status;

/* build_ca variants (examples/<x>/separate_<x>.py) */
enum { DCS_ARCH_DSD = 0, DCS_ARCH_IKALA = 1, DCS_ARCH_BACH10 = 2, DCS_ARCH_BACH10_SI = 3,
       DCS_ARCH_DSD_ILD = 4 /* stereo DSD100 graph of examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py:66-115 */,
       DCS_ARCH_IKALA_NOPOOL = 5 /* the iKala TRAINER's graph (examples/ikala/trainCNN.py:87-118): as DCS_ARCH_IKALA without
                                    the (1, 4) max-pool, fc.W has 30 * 21 * 143 = 90 090 rows at 513 bins */,
       DCS_ARCH_BACH10_SI1 = 6 /* the single-branch score-informed graph (examples/bach10_scoreinformed/trainCNNrwc_samp.py:
                                  195-235): 4 input channels, ONE per-source dense layer and pair of InverseLayers, 4 output
                                  channels, 11 arrays -- also what predict_function2 of the 17-array DCS_ARCH_BACH10_SI graph
                                  evaluates (its other three branches never reach the masks) */ };
/* soft-mask epsilon convention: A = separate_dsd.py:258-266, B = separate_bach10.py:251-259 */
enum { DCS_EPS_A = 0, DCS_EPS_B = 1 };
/* max-pool gradient tie routing: ALL = Theano 0.9 CPU MaxPoolGrad, FIRST = cuDNN */
enum { DCS_TIE_ALL = 0, DCS_TIE_FIRST = 1 };
/* score-informed path, the two places where the separate script and the trainers differ:
 * harmonic masks divided by each instrument's own maximum (script filterSpec, separate_bach10.py:195; dataset.py:781) or, bin by
 * bin, by the sum over the instruments (LargeDatasetMask2.filterSpec, dataset.py:862 -- the class trainCNNrwc.py:657 trains on);
 * soft masks applied to input channel 0 (script, separate_bach10.py:485) or to the sum of the input channels (trainers,
 * trainCNNrwc.py:258-263, trainCNNrwc_samp.py:300-305). */
enum { DCS_SCORE_NORM_MAX = 0, DCS_SCORE_NORM_SUM = 1 };
enum { DCS_MIX_CH0 = 0, DCS_MIX_SUM = 1 };
/* tiler: SCRIPT = separate_dsd.py:114-135 (drops the tail), LIBRARY = util.py:220-248 (zero pads) */
enum { DCS_TILER_SCRIPT = 0, DCS_TILER_LIBRARY = 1 };

/* ------------------------------------------------------------------ library / context */
DCS_API int dcs_version(void);
DCS_API const char* dcs_last_error(void);

/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the
 * device's default stream. */
DCS_API int dcs_create(int device, void* hip_stream, dcs_ctx** out);
DCS_API int dcs_destroy(dcs_ctx* ctx);
DCS_API int dcs_synchronize(dcs_ctx* ctx);

/* ------------------------------------------------------------------ framing integers (host) */
/* numberFrames of stft_norm: int(ceil(L/hop) + 2)                      transform.py:309 */
DCS_API int64_t dcs_frame_count(int64_t n_samples, int hop);
/* len(istft_norm(...)) = hop*(T-1) + N - N/2                            transform.py:373,390 */
DCS_API int64_t dcs_inverse_length(int64_t n_frames, int hop, int frame);
/* number of tiles the reference tilers cut from T frames                separate_dsd.py:121-125, util.py:228-232 */
DCS_API int64_t dcs_tile_count(int64_t n_frames, int time_context, int overlap, int tiler);

/* ------------------------------------------------------------------ STFT  (transform.py:224-396) */
/* frame must be a power of two in [64, 8192]; window_h = window(frame) as the reference
 * materialises it in Transforms.__init__ (transform.py:78). */
DCS_API int dcs_stft_plan(dcs_ctx* ctx, int frame, int hop, const double* window_h, dcs_stft** out);
DCS_API int dcs_stft_plan_destroy(dcs_stft* plan);

/* compute_file: mag = |rfft(w * frame)| / sqrt(N), phase = angle(.)     transform.py:243-247
 * audio_d [n_samples]; mag_d / phase_d [rows_out, ld] with rows >= dcs_frame_count() written
 * as zero rows (used by the zero-padding tiler); phase_d may be NULL (phase=False). */
DCS_API int dcs_stft_forward_f32(dcs_stft* plan, const float* audio_d, int64_t n_samples, float* mag_d,
                         float* phase_d, int64_t ld, int64_t rows_out);
DCS_API int dcs_stft_forward_f64(dcs_stft* plan, const double* audio_d, int64_t n_samples, double* mag_d,
                         double* phase_d, int64_t ld, int64_t rows_out);

/* compute_transform (transform.py:80-131; caller examples/dsd100/compute_features.py:83-112): the same transform for every
 * column of audio[t, i] -- n_clips signals of n_samples each, signal c at audio_d + c * clip_stride -- in ONE launch:
 * mag_d / phase_d [n_clips][rows_out, ld], i.e. the reference's mags[i] = compute_file(audio[:, i]) stacked, ready to be
 * written as the .data file.  phase_d may be NULL. */
DCS_API int dcs_stft_forward_f32_clips(dcs_stft* plan, const float* audio_d, int64_t n_samples, int64_t n_clips, int64_t clip_stride,
                               float* mag_d, float* phase_d, int64_t ld, int64_t rows_out);
DCS_API int dcs_stft_forward_f64_clips(dcs_stft* plan, const double* audio_d, int64_t n_samples, int64_t n_clips, int64_t clip_stride,
                               double* mag_d, double* phase_d, int64_t ld, int64_t rows_out);

/* compute_inverse for n_src magnitude matrices sharing one phase:       transform.py:271-273, 337-396
 *   X = (mag / pre_div) * sqrt(N) * exp(j*phase) -> irfft -> window -> overlap-add -> / sum(w*w)
 * mag_d [n_src][n_frames, ld] (source stride src_stride elements), phase_d [n_frames, ld],
 * audio_d [n_src][n_out] with n_out <= dcs_inverse_length() (the caller's truncation,
 * separate_dsd.py:305-306).  pre_div is the scale_factor division of separate_dsd.py:304 (1.0 for
 * the plain transform API). */
DCS_API int dcs_stft_inverse_f32(dcs_stft* plan, const float* mag_d, int64_t src_stride, const float* phase_d,
                         int64_t ld, int64_t n_frames, int n_src, float pre_div, float* audio_d,
                         int64_t n_out);
DCS_API int dcs_stft_inverse_f64(dcs_stft* plan, const double* mag_d, int64_t src_stride, const double* phase_d,
                         int64_t ld, int64_t n_frames, int n_src, double pre_div, double* audio_d,
                         int64_t n_out);

/* ------------------------------------------------------------------ tiling (separate_dsd.py:114-169, util.py:220-327) */
/* generate_overlapadd: tiles_d [n, C, tc, F] = scale * mag_d[C][T, ld] windows; n = dcs_tile_count().
 * (the reference multiplies by scale_factor before tiling, separate_dsd.py:290) */
DCS_API int dcs_tile(dcs_ctx* ctx, const float* mag_d, int64_t ch_stride, int64_t ld, int C, int64_t n_frames, int F,
             int time_context, int overlap, int tiler, float scale, float* tiles_d, int64_t n_tiles);

/* overlapadd_multi / overlapadd: cross-fade stitch of out_d [S, n, tc, F] into
 * sep_d [S][n*(tc-ov)+tc, ld] (source stride sep_stride).  rise_h = np.linspace(0,1,overlap) as
 * float64 (util.py:306); the fall ramp is its reverse (util.py:307). */
DCS_API int dcs_overlap_add(dcs_ctx* ctx, const float* out_d, int64_t n_tiles, int S, int time_context, int overlap,
                    int F, const double* rise_h, float* sep_d, int64_t sep_stride, int64_t ld);

/* ------------------------------------------------------------------ network (build_ca + mask) */
/* params_d: device float32 arrays in lasagne.layers.get_all_params order (= the .pkl order),
 * shapes: nparams x 4 int64 (unused trailing dims = 1).  Fails with DCS_ESHAPE exactly where
 * lasagne.layers.set_all_param_values would raise (separate_dsd.py:250). */
DCS_API int dcs_model_create(dcs_ctx* ctx, int arch, int in_channels, int time_context, int F,
                     const float* const* params_d, const int64_t* shapes, int nparams, dcs_model** out);
DCS_API int dcs_model_destroy(dcs_model* m);
DCS_API int dcs_model_num_sources(const dcs_model* m);
/* f16 = 1: conv2 and its transpose of the ikala / bach10 / score-informed graphs run with f16 inputs and
 * f32 accumulation on the matrix cores (BASELINE config 3, "fp16 MFMA conv path"); 0 (default): f32-class arithmetic
 * everywhere (f32 MFMA, or the bf16 pipe with operands split exactly into three bf16 terms).  With the switch on, the
 * single-channel bach10 graph runs both InverseLayers in one kernel (colconv_wreg.hip): conv2^T in f16 and -- the default since
 * round 4 -- conv1^T in f16 as well: the activations between the two InverseLayers are rounded to f16 and meet an f16 conv1
 * filter in one MFMA per tile (f32 accumulation).  Results stay within the f16 path's stated tolerance (2e-3 of the network
 * output, tests/test_gpu_configs.py), not within 1e-4.  Since round 6 the DENSE layers of that graph follow the switch as well
 * when a pass has 128 .. 176 tiles (the window of the all-rows kernels): the bottleneck layer and the per-source layers multiply
 * f16 weights (one plane, 2 bytes per weight -- SURVEY 8d prices this config as HBM-on-weights at fp16), conv1 hands its map to
 * conv2 as f16 (the same values conv2 rounded to before, now rounded once by the producer), conv2 hands its map to
 * the bottleneck layer as f16, and the per-source layers write their output once as f16 in the layout the fused decoder reads
 * (gemm_f16.hip); the measured error of the network output is unchanged (3e-5: the f16 convolutions dominate).  The ikala
 * graph (10 x 20 filters) takes the same slab kernel in either precision (one f16 plane instead of three bf16 planes). */
DCS_API int dcs_model_set_conv_precision(dcs_model* m, int f16);
/* Score-informed graphs (DCS_ARCH_BACH10_SI / _SI1): which of the reference's two semantics dcs_separate_scoreinformed and
 * dcs_model_forward_masked follow (the enums above).  Default = the separate script's: DCS_SCORE_NORM_MAX, DCS_MIX_CH0.
 * A model trained by trainCNNrwc.py saw sum-normalised inputs and a channel-sum mixture: (DCS_SCORE_NORM_SUM, DCS_MIX_SUM).
 * DCS_MIX_SUM adds the C input channels left to right in float32.  DCS_EUNSUPPORTED for single-channel graphs. */
DCS_API int dcs_model_set_score_semantics(dcs_model* m, int normalise, int mixture);
/* Which stages of dcs_separate run on the one-batch ("latency") kernels of csrc/dsd_lat.hip -- the shape of the
 * reference's own call, predict_function2 on ONE batch of 32 tiles (separate_dsd.py:296-298), where a kernel's duration
 * is its chain of dependent memory latencies.  stages = -1 (default): automatic, all of them for one clip of at most
 * 1024 frames; 0: the throughput kernels; else a bit set: 1 STFT, 2 conv1, 4 conv2, 8 bottleneck,
 * 16 per-source dense, 32 transposed conv2, 64 final (transposed conv1 + mask + cross-fade), 128 iSTFT.  Both families
 * read and write the same buffers, so any mix is valid (tests compare each stage against the other family).  DSD graph
 * only (DCS_EUNSUPPORTED otherwise). */
DCS_API int dcs_model_set_latency_stages(dcs_model* m, int stages);
/* Testing aids (host only, no GPU): the weight re-layouts of the one-batch kernels.  dcs_lat_pack_b_host: B[K][ldb]
 * (k-major) -> [slice][column block][j][lane][4], the order in which lane (fi = lane & 15, kq = lane >> 4) of wave
 * `slice` feeds v_mfma_f32_16x16x4_f32: element e of piece j is B[slice * slice_len + 16 j + 4 kq + e][16 cb + fi].
 * dcs_lat_pack_deconv2_host: Bw2s[ci][16 taps][52] -> [ci][j][lane][4] with column fi = tap.  Both return the number of
 * floats of the packed array (and fill `out` when out_len is large enough) or a negative status. */
DCS_API int64_t dcs_lat_pack_b_host(const float* B, int ldb, int K, int n_cb, int slice_len, int n_slices, float* out, int64_t out_len);
DCS_API int64_t dcs_lat_pack_deconv2_host(const float* Bw2s, int n_ci8, float* out, int64_t out_len);

/* predict_function2 (separate_dsd.py:273,298): tiles_d [n, C, tc, F] -> out_d [S, n, tc, F]
 * = soft-masked magnitudes of the S sources. */
DCS_API int dcs_model_forward_masked(dcs_model* m, const float* tiles_d, int64_t n_tiles, int eps_mode, int tie_mode,
                             float* out_d);
/* lasagne.layers.get_output(network2): p_d [n, channels_out, tc, F] before masking (testing aid) */
DCS_API int dcs_model_forward(dcs_model* m, const float* tiles_d, int64_t n_tiles, int tie_mode, float* p_d);
DCS_API int dcs_model_out_channels(const dcs_model* m);
/* Which kernel the fused path (dcs_separate*) runs for the decoder's last stage (transposed conv1 + bias + rectify + mask
 * + cross-fade) on n_clips clips of n_frames frames each: 0 = f32 MFMA, 64-bin workgroups (small launches); 1 = f32
 * MFMA, 128-bin workgroups; 2 = bf16 MFMA on operands split exactly into three bf16 terms (f32-class results, the
 * default for launches that fill the chip; DSD / hiphop graph); 3 = the one-batch kernel of csrc/dsd_lat.hip (one clip of
 * at most 1024 frames, see dcs_model_set_latency_stages; the same bf16x3 arithmetic, 16 x 64 workgroups);
 * negative: not a fused-kernel graph.  bench.py prices its roofline block with this. */
DCS_API int dcs_model_final_kernel(const dcs_model* m, int64_t n_frames, int64_t n_clips, int eps_mode);

/* ------------------------------------------------------------------ fused file-level path */
/* The separation block of train_auto (separate_dsd.py:289-306) for one mono signal already in
 * HBM:  STFT -> x scale -> tiles -> network -> mask -> cross-fade overlap-add -> / scale ->
 * iSTFT -> truncate to n_samples.  pcm_d [S, n_samples] float32.  n_tiles_out / n_frames_out
 * (host, optional) receive the tile and frame counts.  Returns DCS_EINVAL when the tiler yields
 * zero tiles (the reference raises in overlapadd_multi in that case). */
DCS_API int dcs_separate(dcs_model* m, dcs_stft* plan, const float* audio_d, int64_t n_samples, int overlap, int tiler,
                 float scale, int eps_mode, int tie_mode, float* pcm_d, int64_t* n_tiles_out,
                 int64_t* n_frames_out);

/* The separation block of the score-informed script (examples/bach10_scoreinformed/separate_bach10.py:497-571) in one
 * call: STFT -> x scale -> filterSpec masks of the note table (notes_h exactly as for dcs_score_masks, frame window
 * (0, n_frames)) -> ninst-channel tiles of the library tiler -> network -> masks (applied to input channel 0, or to the channel
 * sum after dcs_model_set_score_semantics; the same call selects the sum-normalised harmonic masks) -> cross-fade
 * -> / scale -> iSTFT.  The model must have ninst input channels.  pcm_d [S, n_samples] float32.  All tiles go through the
 * network in one pass (the script's batch loop gives the same values tile by tile).  Asynchronous (the note rectangles are
 * staged through the context's pinned upload ring; notes_h may be reused when the call returns). */
DCS_API int dcs_separate_scoreinformed(dcs_model* m, dcs_stft* plan, const float* audio_d, int64_t n_samples, const double* notes_h,
                               int ninst, int n_notes, int width, int overlap, float scale, int eps_mode, int tie_mode,
                               float* pcm_d, int64_t* n_tiles_out, int64_t* n_frames_out);

/* The same pipeline for n_clips mono signals of EQUAL length in one set of launches (the batch-of-files
 * driver of SURVEY 8f.1 / separate_multiple.ipynb; equal-length segments of one long file): clip c starts at
 * audio_d + c * clip_stride (clip_stride >= n_samples), pcm_d [n_clips][S][n_samples].  Every clip is processed
 * exactly as dcs_separate would process it alone (same tiles, same cross-fade) -- the clips only share kernel
 * launches; outputs agree with the single-clip call to fp32 rounding (the FFT / GEMM kernel variants are
 * chosen by the total amount of work).  The ikala / bach10 graphs stack the tiles of all clips into one pass of the
 * network (their dense-layer weights are then read once per group).  n_tiles_out / n_frames_out are per clip. */
DCS_API int dcs_separate_batch(dcs_model* m, dcs_stft* plan, const float* audio_d, int64_t n_samples, int64_t n_clips,
                       int64_t clip_stride, int overlap, int tiler, float scale, int eps_mode, int tie_mode,
                       float* pcm_d, int64_t* n_tiles_out, int64_t* n_frames_out);

/* The same for clips of DIFFERENT lengths (a directory of songs, separate_multiple.ipynb): clip c has
 * n_samples_h[c] samples (host array) starting at audio_d + c * clip_stride, and its S signals are written to
 * pcm_d + (c * S + s) * pcm_stride (pcm_stride >= the longest clip; samples past a clip's own length are not
 * written).  Strides and grids are sized by the longest clip; the STFT, the cross-fade fold and the iSTFT read
 * every clip's own sample / frame / tile counts from a small device table, so each clip gets exactly the frames,
 * the tiles and the cross-fade dcs_separate gives it alone (shorter clips cost the launch the work of the longest).
 * Every single-channel graph: DSD / hiphop through the fused path; ikala / bach10 with one STFT / iSTFT launch over all
 * clips and one pass of all their tiles through the network (a clip's tiles sit behind those of the clips before it).
 * Needs the wave STFT kernels (frameSize 1024 / 2048 / 4096 with hop | frameSize, DCS_EUNSUPPORTED otherwise).
 * n_tiles_out / n_frames_out: [n_clips] or NULL.  Equal lengths with pcm_stride == length take the
 * dcs_separate_batch path.  Asynchronous like every other call: the clip table goes out through a pinned staging ring owned
 * by the model (n_samples_h may be reused as soon as the call returns). */
DCS_API int dcs_separate_ragged(dcs_model* m, dcs_stft* plan, const float* audio_d, const int64_t* n_samples_h,
                        int64_t n_clips, int64_t clip_stride, int overlap, int tiler, float scale, int eps_mode,
                        int tie_mode, float* pcm_d, int64_t pcm_stride, int64_t* n_tiles_out, int64_t* n_frames_out);

/* Stereo separation, the "Separating" block of examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py:291-325, for a
 * DCS_ARCH_DSD_ILD model: audio_d holds the two channels (channel c at audio_d + c * channel_stride), one STFT per
 * channel, 2-channel tiles (pass DCS_TILER_LIBRARY: the trainer calls util.generate_overlapadd), one network pass,
 * per input channel the mask of :176-180 (p / (sum over sources + 1e-12 r), source = mask * input + 1e-12 r, r = 0.1
 * standing in for the trainer's N(0, 0.1) draw), cross-fade, and the iSTFT with that channel's phase.
 * pcm_d [2][S][n_samples]; sep_d (optional) [2][S][n_frames, ld_out] scaled magnitudes.  Either may be NULL. */
DCS_API int dcs_separate_stereo(dcs_model* m, dcs_stft* plan, const float* audio_d, int64_t n_samples, int64_t channel_stride,
                        int overlap, int tiler, float scale, float* pcm_d, float* sep_d, int64_t ld_out,
                        int64_t* n_tiles_out, int64_t* n_frames_out);

/* Same pipeline stopped before the iSTFT: sep_d [S][n_frames, ld_out] (scaled magnitudes, what the
 * reference calls mm[i,:len(ph)]) and phase_d [n_frames, ld_out]; either may be NULL. */
DCS_API int dcs_separate_spectra(dcs_model* m, dcs_stft* plan, const float* audio_d, int64_t n_samples, int overlap,
                         int tiler, float scale, int eps_mode, int tie_mode, float* sep_d, float* mag_d,
                         float* phase_d, int64_t ld_out);

/* The wav sample format of every script: out_d[i] = (int16)(pcm_d[i] * 32767), truncation toward zero, no
 * clipping (separate_dsd.py:307-309); the product is formed in double like the scripts' float64 `audio_out * maxn`, so the
 * values are those of `(pcm.astype(float64) * 32767).astype('int16')` bit for bit.  Halves the bytes of the multi-GPU PCM
 * gather and of the batch driver's device-to-host copy. */
DCS_API int dcs_pcm_to_int16(dcs_ctx* ctx, const float* pcm_d, int64_t n, int16_t* out_d);

/* The other end of a wav file: int16 frames as scipy.io.wavfile.read returns them ([n_frames][channels], interleaved) -> the
 * mono float32 signal the scripts separate: sample.astype('float') / 32767 (separate_dsd.py:275-282, float64), then
 * mode 0: (L + R) / 2 for two or more channels, the channel itself for mono (separate_dsd.py:285-287, separate_bach10.py);
 * mode 1: L + R (separate_ikala.py:229; a mono file is DCS_EINVAL with NumPy's IndexError text), computed in double and
 * rounded to float32 once -- the value the host path `to_device(to_mono(read_wav(f)))` uploads, bit for bit, for half (stereo)
 * or a quarter (mono) of its PCIe bytes and none of its host arithmetic.  n_clips stacked clips: clip c at
 * pcm16_d + c * in_stride (int16 elements, >= n_frames * channels), out_d + c * out_stride. */
DCS_API int dcs_pcm16_to_float(dcs_ctx* ctx, const int16_t* pcm16_d, int64_t n_frames, int channels, int mode, int64_t n_clips,
                       int64_t in_stride, float* out_d, int64_t out_stride);

/* ---- wav files of the batch-of-files driver (host side, no device work) ----------------------------------------------
 * What scipy.io.wavfile.read / .write do in every script (separate_dsd.py:275-282, :307-309) for 16-bit PCM files, by a pool of
 * I/O threads and without the float detour: int16 frames go from the file into the caller's (pinned) staging memory and from
 * it into a file behind the 44-byte header scipy.io.wavfile.write produces -- the division by 32767, the mix-down and the int16
 * conversion are dcs_pcm16_to_float / dcs_pcm_to_int16 on the device.  A batch of reads or writes is one call that returns at
 * once; the caller collects it with dcs_wav_batch_wait.  Thread-safe; the threads never call back into the caller. */
typedef struct dcs_wav_pool dcs_wav_pool;
typedef struct dcs_wav_batch dcs_wav_batch;
DCS_API int dcs_wav_pool_create(int n_threads, dcs_wav_pool** out);
/* Completes every enqueued batch first.  Batches that were not waited for stay valid for dcs_wav_batch_wait. */
DCS_API void dcs_wav_pool_destroy(dcs_wav_pool* pool);
/* Enqueue n reads.  File i: its frames go to dst_h[i] (capacity cap[i] bytes); after the wait status[i] is 0 (rate[i],
 * n_frames[i], channels[i] are set, the frames lie at dst_h[i] as [n_frames][channels] int16), 1 (not plain 16-bit PCM, e.g.
 * float / 24-bit samples, a damaged header, or larger than cap[i]: take the scripts' float path for this file) or -errno (the
 * file could not be opened or read).  The paths are copied; every other array must stay valid until the wait. */
DCS_API int dcs_wav_read_pcm16_async(dcs_wav_pool* pool, int n, const char* const* paths, void* const* dst_h, const int64_t* cap,
                                     int32_t* rate, int64_t* n_frames, int32_t* channels, int32_t* status, dcs_wav_batch** out);
/* Enqueue n writes.  File i = the 44 header bytes of scipy.io.wavfile.write(path, rate[i], int16 [n_frames[i]] or
 * [n_frames[i], channels[i]]) followed by the frames at data_h[i], one writev; missing parent directories are created.
 * status[i] after the wait: 0 or -errno.  DCS_EINVAL for a file that would not fit a wav header (>= 4 GiB). */
DCS_API int dcs_wav_write_pcm16_async(dcs_wav_pool* pool, int n, const char* const* paths, const int16_t* const* data_h,
                                      const int64_t* n_frames, const int32_t* channels, const int32_t* rate, int32_t* status,
                                      dcs_wav_batch** out);
/* 1 when every file of the batch has been handled (the batch stays valid), 0 otherwise. */
DCS_API int dcs_wav_batch_done(dcs_wav_batch* batch);
/* Blocks until the batch is complete and releases it. */
DCS_API int dcs_wav_batch_wait(dcs_wav_batch* batch);

/* The one exchange of the multi-GPU path (tiles / clips are sharded over one process per GPU and nothing else is shared;
 * SURVEY 8b `dcs_gather(h, ncclComm_t, shard, count, full, root)`, counted in BYTES here so that the scripts' int16 PCM of
 * dcs_pcm_to_int16 travels as it is).  nccl_comm is the caller's ncclComm_t (RCCL: ncclCommInitRank, one rank per GPU);
 * librccl is dlopen'ed on first use -- libdcs has no link-time dependency on it (DCS_EUNSUPPORTED if it cannot be loaded).
 * root < 0: all-gather, every rank's full_d [n_ranks][bytes] in rank order.  root >= 0: only that rank receives (grouped
 * ncclSend / ncclRecv: what a single writer process needs, 1 / n_ranks of the all-gather's traffic per link); full_d may be
 * NULL elsewhere.  Enqueued on the ctx stream like every other call; shard_d may be the rank's own slot of full_d. */
DCS_API int dcs_gather(dcs_ctx* ctx, void* nccl_comm, const void* shard_d, int64_t bytes, void* full_d, int root);

/* ------------------------------------------------------------------ score-informed front-end */
/* filterSpec (examples/bach10_scoreinformed/separate_bach10.py:172-200) and the network input of :520-527.
 * notes_h: HOST table [ninst][n_notes][width] of doubles exactly as expandMidi returns it (util.py:424-512),
 * width = 2*nharmonics+3: begin frame, end frame, midi number, then (first bin, one-past-last bin) pairs, zeros
 * unused.  start/stop: filterSpec's frame window (0, nframes in the script).  Per instrument j the mask is 1 on
 * the note rectangles and 1e-18 elsewhere, divided by its maximum (so an instrument without notes gets an all-ones
 * mask, as in the reference).  out_d [ninst][n_frames][F] = mask_j * mag_d (mag_d [n_frames, ld], already scaled by
 * the caller like :503); mask_d [n_frames][ninst*F] = filterSpec's return value.  Either output may be NULL.
 * A bin range outside [0, F) is DCS_ESHAPE (NumPy raises IndexError there).  Asynchronous: the note rectangles go out through
 * a pinned staging ring owned by the context. */
DCS_API int dcs_score_masks(dcs_ctx* ctx, const float* mag_d, int64_t ld, int64_t n_frames, int F, const double* notes_h,
                    int ninst, int n_notes, int width, int64_t start, int64_t stop, float* out_d, float* mask_d);

/* The same with the normalisation of the masks chosen by the caller: DCS_SCORE_NORM_MAX = dcs_score_masks; DCS_SCORE_NORM_SUM =
 * LargeDatasetMask2.filterSpec (dataset.py:839-879): mask_j = filtered_j / sum_i filtered_i, float32, instruments added in order
 * -- a bin nobody plays is 1e-18 / (4 x 1e-18 added in turn) = 0.25 for four instruments, a bin k instruments play is 1 / k for
 * them and 1e-18 / k for the others.  At most 32 instruments. */
DCS_API int dcs_score_masks_norm(dcs_ctx* ctx, const float* mag_d, int64_t ld, int64_t n_frames, int F, const double* notes_h,
                         int ninst, int n_notes, int width, int64_t start, int64_t stop, int normalise, float* out_d,
                         float* mask_d);

/* ------------------------------------------------------------------ timing aid for bench.py */
/* Average duration (ms) of the kernels tagged `which` since the last reset, measured with HIP events
 * on the ctx stream.  tag_mask bit t enables the tag t (two event records per tagged launch);
 * 0 disables all. */
enum { DCS_TAG_STFT = 0, DCS_TAG_CONV1 = 1, DCS_TAG_CONV2 = 2, DCS_TAG_FC = 3, DCS_TAG_FC1X = 4,
       DCS_TAG_DECONV2 = 5, DCS_TAG_FINAL = 6, DCS_TAG_ISTFT = 7, DCS_TAG_OLA = 8, DCS_TAG_TILE = 9, DCS_TAG_POOL = 10,
       DCS_TAG_UNPOOL = 11, DCS_TAG_MASK = 12, DCS_TAG_SCORE = 13,
       DCS_TAG_DECODER = 14 /* transposed conv2 + transposed conv1 in one kernel (Bach10 graph, f16 switch) */,
       DCS_TAG_COUNT = 15 };
DCS_API int dcs_timing_enable(dcs_ctx* ctx, unsigned tag_mask);
/* bracket only every stride-th launch of an enabled tag (an event pair costs ~6 us of stream time each side) */
DCS_API int dcs_timing_stride(dcs_ctx* ctx, int stride);
DCS_API int dcs_timing_reset(dcs_ctx* ctx);
DCS_API int dcs_timing_query(dcs_ctx* ctx, int which, double* avg_ms, int64_t* launches);

/* ------------------------------------------------------------------ memory-safety aid (tests/test_gpu_guard.py) */
/* With DCS_WS_GUARD=<bytes> in the environment (read once per process) every scratch block libdcs allocates -- the
 * per-model workspace, the K-split partial sums, the clip / note tables -- sits between two red zones of that size and is
 * born filled with the byte DCS_WS_POISON (default 0xFF: float NaN).  This call synchronises the device and verifies every
 * live red zone: DCS_OK, DCS_EHIP with the first damaged byte in dcs_last_error(), DCS_EUNSUPPORTED without the switch.
 * No reference counterpart (the reference is NumPy / Theano); n_blocks_out (nullable) = guarded blocks alive. */
DCS_API int dcs_debug_check_guards(dcs_ctx* ctx, int64_t* n_blocks_out);

#ifdef __cplusplus
}
#endif
#endif /* DCS_H */
