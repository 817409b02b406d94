"""CPU restatement of the separation block of ``train_auto`` (test infrastructure).

Follows ``examples/dsd100/separate_dsd.py:289-306`` (ikala: ``separate_ikala.py:229-246``,
bach10: ``separate_bach10.py:282-300``):

    mag, ph = compute_file(audio, phase=True)            float64
    mag     = scale_factor * mag.astype(np.float32)      float32 product
    batches, nchunks = generate_overlapadd(mag, ...)     float64 container
    output  = [predict_function2(batch) for batch in batches]
    mm      = overlapadd_multi(np.array(output), batches, nchunks, overlap)
    audio_i = compute_inverse(mm[i, :len(ph)] / scale_factor, ph)[:len(audio)]

with ``predict_function2`` = ``oracle.net_ref.predict`` (float64, as Theano's CPU default
floatX).  Garbage slots of the last batch (``np.empty`` in the reference) are zero tiles
here; batch items never interact, so they cannot influence real tiles.
"""
import numpy as np

from . import net_ref, stft_np, tiling_np


def separate(arch, params, audio, scale_factor=0.3, time_context=30, overlap=25, batch_size=32,
             frameSize=1024, hopSize=512, window=np.hanning, tiler=tiling_np.SCRIPT, tie_mode='all',
             eps_mode=None, return_spectra=False):
    audio = np.asarray(audio, dtype=np.float64)
    mag, ph = stft_np.compute_file(audio, phase=True, frameSize=frameSize, hopSize=hopSize, window=window)
    mag = scale_factor * mag.astype(np.float32)
    batches, nchunks = tiling_np.generate_overlapadd(mag, mag.shape[-1], time_context, overlap, batch_size,
                                                     tiler=tiler, fill=0.0)
    if nchunks == 0:
        raise IndexError("tuple index out of range")
    output = []
    for batch in batches:
        output.append(net_ref.predict(arch, params, batch, tie_mode=tie_mode, eps_mode=eps_mode))
    output = np.array(output)
    mm = tiling_np.overlapadd_multi(output, nchunks, overlap=overlap)
    pcm = []
    for i in range(mm.shape[0]):
        audio_out = stft_np.compute_inverse(mm[i, :len(ph)] / scale_factor, ph, frameSize=frameSize,
                                            hopSize=hopSize, window=window)
        if len(audio_out) > len(audio):
            audio_out = audio_out[:len(audio)]
        pcm.append(audio_out)
    pcm = np.stack(pcm)
    if return_spectra:
        return pcm, mm[:, :len(ph)], mag, ph
    return pcm


def separate_scoreinformed(params, audio, melody, scale_factor=0.3, time_context=30, overlap=25, batch_size=32,
                           frameSize=4096, hopSize=512, window=None, tie_mode='all', return_input=False, harmonics=None,
                           normalise='max', mixture='ch0', arch='bach10_si'):
    """``examples/bach10_scoreinformed/separate_bach10.py:497-541``: the network input is one channel per instrument,
    ``filterSpec`` mask x scaled magnitudes (:520-527); tiles come from the LIBRARY tiler (:531, ``util.
    generate_overlapadd``, ``toverlap`` there is an undefined name -- the script's ``overlap`` is meant); the masks
    are built from the first four output channels and applied to input channel 0 (:473-486); ``overlapadd_multi``
    and the iSTFT as in the other scripts.

    ``normalise='sum'`` / ``mixture='sum'``: the semantics of the TRAINERS' own separation block
    (bach10_scoreinformed/trainCNNrwc.py:360-416): masks from ``LargeDatasetMask2.filterSpec`` (dataset.py:839-866, every
    bin divided by the sum over the instruments) and the soft masks applied to the sum of the four input channels
    (trainCNNrwc.py:258-263).  ``arch='bach10_si1'``: the single-branch graph of trainCNNrwc_samp.py:195-235 (11 arrays)."""
    from . import score_np
    if window is None:
        from scipy.signal.windows import blackmanharris as window
    audio = np.asarray(audio, dtype=np.float64)
    nframes = int(np.ceil(len(audio) / np.double(hopSize))) + 2
    mag, ph = stft_np.compute_file(audio, phase=True, frameSize=frameSize, hopSize=hopSize, window=window)
    mag = scale_factor * mag.astype(np.float32)
    masks = score_np.network_input(mag, np.asarray(melody), nframes, harmonics, normalise=normalise)   # [C, T, F] float64 (harmonics: timbre model)
    batches, nchunks = tiling_np.generate_overlapadd(masks, masks.shape[-1], time_context, overlap, batch_size,
                                                     tiler=tiling_np.LIBRARY, fill=0.0)
    if nchunks == 0:
        raise IndexError("tuple index out of range")
    output = []
    for batch in batches:
        output.append(net_ref.predict(arch, params, batch, tie_mode=tie_mode, mixture=mixture))
    output = np.array(output)
    mm = tiling_np.overlapadd_multi(output, nchunks, overlap=overlap)
    pcm = []
    for i in range(mm.shape[0]):
        audio_out = stft_np.compute_inverse(mm[i, :len(ph)] / scale_factor, ph, frameSize=frameSize,
                                            hopSize=hopSize, window=window)
        if len(audio_out) > len(audio):
            audio_out = audio_out[:len(audio)]
        pcm.append(audio_out)
    pcm = np.stack(pcm)
    if return_input:
        return pcm, masks
    return pcm


def separate_stereo(params, audio, scale_factor=0.3, time_context=30, overlap=25, batch_size=32, frameSize=1024,
                    hopSize=512, window=np.hanning, return_spectra=False):
    """The "Separating" block of the stereo trainer (examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py:291-325):
    ``audio [L, 2]``; ``compute_transform`` gives ``mag, ph [2, T, F]`` (one STFT per channel, transform.py:80-131);
    the LIBRARY tiler cuts 2-channel tiles (:303); ``predict_function`` returns per input channel the S masked
    sources (:176-189); per channel j the sources are cross-faded (``np.swapaxes(output[:, j:j+1], 1, 3)``, :313) and
    inverted with that channel's phase (:315).  Returns ``[L, S, 2]`` like ``sep_audio`` (:299)."""
    audio = np.asarray(audio, dtype=np.float64)
    L, C = audio.shape
    mags, phs = [], []
    for c in range(C):
        m, p = stft_np.compute_file(audio[:, c], phase=True, frameSize=frameSize, hopSize=hopSize, window=window)
        mags.append(m)
        phs.append(p)
    mag = scale_factor * np.stack(mags).astype(np.float32)
    batches, nchunks = tiling_np.generate_overlapadd(mag, mag.shape[-1], time_context, overlap, batch_size,
                                                     tiler=tiling_np.LIBRARY, fill=0.0)
    if nchunks == 0:
        raise IndexError("tuple index out of range")
    output = np.array([net_ref.predict_ild(params, batch, inverse='explicit') for batch in batches])   # [nb, C, B, S, tc, F]
    T = phs[0].shape[0]
    sep_audio = np.zeros((L, output.shape[3], C))
    spectra = []
    for j in range(C):
        mm = tiling_np.overlapadd_multi(np.swapaxes(output[:, j:j + 1], 1, 3), nchunks, overlap=overlap)
        spectra.append(mm[:, :T])
        for i in range(mm.shape[0]):
            audio_out = stft_np.compute_inverse(mm[i, :T] / scale_factor, phs[j], frameSize=frameSize, hopSize=hopSize,
                                                window=window)
            sep_audio[:, i, j] = audio_out[:L]
    if return_spectra:
        return sep_audio, np.stack(spectra, axis=1), mag, np.stack(phs)    # spectra [S, C, T, F]
    return sep_audio
