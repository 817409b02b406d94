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
                           frameSize=4096, hopSize=512, window=None, tie_mode='all', return_input=False):
    """``examples/bach10_scoreinformed/separate_bach10.py:497-541``: the network input is one channel per instrument,
    ``filterSpec`` mask x scaled magnitudes (:520-527); tiles come from the LIBRARY tiler (:531, ``util.
    generate_overlapadd``, ``toverlap`` there is an undefined name -- the script's ``overlap`` is meant); the masks
    are built from the first four output channels and applied to input channel 0 (:473-486); ``overlapadd_multi``
    and the iSTFT as in the other scripts."""
    from . import score_np
    if window is None:
        from scipy.signal.windows import blackmanharris as window
    audio = np.asarray(audio, dtype=np.float64)
    nframes = int(np.ceil(len(audio) / np.double(hopSize))) + 2
    mag, ph = stft_np.compute_file(audio, phase=True, frameSize=frameSize, hopSize=hopSize, window=window)
    mag = scale_factor * mag.astype(np.float32)
    masks = score_np.network_input(mag, np.asarray(melody), nframes)          # [C, T, F] float64
    batches, nchunks = tiling_np.generate_overlapadd(masks, masks.shape[-1], time_context, overlap, batch_size,
                                                     tiler=tiling_np.LIBRARY, fill=0.0)
    if nchunks == 0:
        raise IndexError("tuple index out of range")
    output = []
    for batch in batches:
        output.append(net_ref.predict('bach10_si', params, batch, tie_mode=tie_mode))
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
