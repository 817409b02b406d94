"""NumPy restatement of the reference STFT / iSTFT (test infrastructure).

Follows ``transform.py:277-335`` (``stft_norm``), ``transform.py:337-396``
(``istft_norm``), ``transform.py:224-274`` (``transformFFT.compute_file`` /
``compute_inverse``); the script copies are
``examples/dsd100/separate_dsd.py:24-111``.

Everything is float64 / complex128 exactly like the reference; the framing
integers (frame count, padding, slice bounds, output length) are the part that
must be bit-exact on the GPU.
"""
import numpy as np


def frame_count(n_samples, hop):
    """``numberFrames`` of ``stft_norm`` (transform.py:309)."""
    return int(np.ceil(n_samples / np.double(hop)) + 2)


def n_bins(frame):
    """``numberFrequencies`` (transform.py:322)."""
    return int(frame / 2 + 1)


def inverse_length(n_frames, hop, frame):
    """Length of ``istft_norm``'s result: hop*(T-1)+N minus the dropped
    half window (transform.py:373,390)."""
    return int(hop * (n_frames - 1) + frame) - int(frame / 2.0)


def stft_norm(data, window, hop, nfft):
    """Complex STFT ``[T, nfft/2+1]`` (transform.py:277-335).

    ``lengthWindow/2`` zeros are put in front so that frame 0 is centred on
    sample 0 (:316); zeros are appended so that T whole frames exist (:319);
    frame n is ``window * padded[n*hop : n*hop+lengthWindow]`` (:329-331)
    transformed with ``rfft(., nfft)`` (:332).
    """
    data = np.asarray(data)
    win_len = window.size
    T = frame_count(data.size, hop)
    padded_len = int((T - 1) * hop + win_len)
    head = int(win_len / 2.0)
    padded = np.zeros(padded_len)
    padded[head:head + data.size] = data
    out = np.zeros((T, n_bins(nfft)), dtype=complex)
    for n in range(T):
        a = int(n * hop)
        out[n] = np.fft.rfft(window * padded[a:a + win_len], int(nfft))
    return out


def istft_norm(X, window, analysis_window, hop, nfft):
    """Weighted overlap-add inverse (transform.py:337-396).

    Frame n -> ``irfft(X[n], nfft)[:lengthWindow]`` (:382-383), accumulated as
    ``window * frame`` (:387-388) while ``window * analysisWindow`` accumulates
    the normaliser (:384-386); the first half window is dropped (:390-391),
    zeros in the normaliser become 1 (:392), then the division (:394).
    """
    X = np.asarray(X)
    T = X.shape[0]
    win_len = int(window.size)
    total = int(hop * (T - 1) + win_len)
    acc = np.zeros(total)
    norm = np.zeros(total)
    ww = window * analysis_window
    for n in range(T):
        a = int(n * hop)
        frame = np.fft.irfft(X[n], int(nfft))[:win_len]
        norm[a:a + win_len] = norm[a:a + win_len] + ww
        acc[a:a + win_len] = acc[a:a + win_len] + window * frame
    head = int(win_len / 2.0)
    acc = acc[head:]
    norm = norm[head:]
    norm[norm == 0] = 1.0
    return acc / norm


def compute_file(audio, phase=False, frameSize=1024, hopSize=512, window=np.hanning):
    """``mag = |X| / sqrt(N)``, ``ph = angle(X)`` (transform.py:243-247).
    ``window`` is a callable (``np.hanning``, ``blackmanharris``) or an array."""
    win = window(frameSize) if callable(window) else np.asarray(window, dtype=float)
    X = stft_norm(audio, win, float(hopSize), float(frameSize))
    mag = np.abs(X) / np.sqrt(frameSize)
    if phase:
        return mag, np.angle(X)
    return mag


def compute_inverse(mag, phase, frameSize=1024, hopSize=512, window=np.hanning):
    """``X = mag*sqrt(N)*exp(j*phase)`` then ``istft_norm`` with the same
    window for analysis and synthesis (transform.py:271-273)."""
    win = window(frameSize) if callable(window) else np.asarray(window, dtype=float)
    X = (np.asarray(mag) * np.sqrt(frameSize)) * np.exp(1j * np.asarray(phase))
    return istft_norm(X, win, win, float(hopSize), float(frameSize))
