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


def synth_params(arch_name, tc=30, F=513, seed=1, bias_scale=0.05):
    """Glorot-uniform float32 weights with the exact ``.pkl`` shape list of the
    architecture; small non-zero biases so every bias term is exercised."""
    arch = ARCHS[arch_name]
    rs = np.random.RandomState(seed)
    params = []
    for shp in arch.param_shapes(tc, F):
        if len(shp) == 1:
            p = rs.uniform(-bias_scale, bias_scale, shp)
        elif len(shp) == 2:
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            p = rs.uniform(-lim, lim, shp)
        else:
            rf = shp[2] * shp[3]
            lim = np.sqrt(6.0 / ((shp[0] + shp[1]) * rf))
            p = rs.uniform(-lim, lim, shp)
        params.append(p.astype(np.float32))
    return params
