"""Seeded synthetic inputs: no dataset, no published weights are reachable
offline (reference README.md:21,30,39,48 only links them), so parity tests and
``bench.py`` run on these."""
import numpy as np

from .arch import ARCHS


def synth_audio(n_samples, seed=0, channels=1, silence=True):
    """5 sinusoids + 0.1 uniform noise (+ a digital-silence gap) quantised to
    int16 and scaled the way ``scipy.io.wavfile.read`` + ``/iinfo.max`` deliver
    it (separate_dsd.py:275-282).  Returns float64 ``[n]`` or ``[n, channels]``."""
    rs = np.random.RandomState(seed)
    t = np.arange(n_samples) / 44100.0
    cols = []
    for _ in range(channels):
        x = np.zeros(n_samples)
        for f0, a in zip(rs.uniform(80.0, 6000.0, 5), rs.uniform(0.05, 0.2, 5)):
            x += a * np.sin(2 * np.pi * f0 * t + rs.uniform(0, 2 * np.pi))
        x += 0.1 * rs.uniform(-1, 1, n_samples)
        if silence:
            g0 = n_samples // 3
            x[g0:g0 + max(1, n_samples // 16)] = 0.0
        cols.append(x)
    x = np.stack(cols, axis=1)
    q = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
    out = q.astype('float') / 32767.0
    return out[:, 0] if channels == 1 else out


def synth_params(arch_name, tc=30, F=513, seed=1, bias_scale=0.05, gain=1.0):
    """Glorot-uniform float32 weights with the exact ``.pkl`` shape list of the
    architecture; small non-zero biases so every bias term is exercised.  ``gain`` scales every weight
    matrix / filter bank (trained networks have larger activations than a fresh Glorot draw)."""
    arch = ARCHS[arch_name]
    rs = np.random.RandomState(seed)
    params = []
    for shp in arch.param_shapes(tc, F):
        if len(shp) == 1:
            p = rs.uniform(-bias_scale, bias_scale, shp)
        elif len(shp) == 2:
            lim = gain * np.sqrt(6.0 / (shp[0] + shp[1]))
            p = rs.uniform(-lim, lim, shp)
        else:
            rf = shp[2] * shp[3]
            lim = gain * np.sqrt(6.0 / ((shp[0] + shp[1]) * rf))
            p = rs.uniform(-lim, lim, shp)
        params.append(p.astype(np.float32))
    return params


_NOTE_NAMES = ("C", "C#", "D", "Eb", "E", "F", "F#", "G", "Ab", "A", "Bb", "B")


def synth_score_text(seed, seconds, lo=45, hi=76, mean_note=0.35):
    """A seeded monophonic score in the text format the score-informed script reads
    (``onset,offset,name`` per line, examples/bach10_scoreinformed/separate_bach10.py:455-518): notes of random
    length and pitch in MIDI ``[lo, hi)`` back to back with occasional rests, until ``seconds``."""
    rs = np.random.RandomState(seed)
    t, lines = float(rs.uniform(0.0, 0.3)), []
    while True:
        dur = float(rs.uniform(0.4, 1.6)) * mean_note
        if t + dur > seconds:
            break
        midi = int(rs.randint(lo, hi))
        lines.append("%.3f,%.3f,%s%d" % (t, t + dur, _NOTE_NAMES[midi % 12], midi // 12 - 1))
        t += dur + (float(rs.uniform(0.05, 0.4)) if rs.uniform() < 0.25 else 0.0)
    return "\n".join(lines) + "\n"
