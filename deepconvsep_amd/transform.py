"""Drop-in for the reference's ``transform.py`` STFT classes, computed on the MI355X.

Mirrors ``Transforms`` / ``transformFFT`` (reference transform.py:52-274): same
constructor signature, same attributes (``frameSize hopSize sampleRate window
suffix`` ...), same ``compute_file`` / ``compute_inverse`` /
``compute_transform`` semantics and return types (fresh float64 ndarrays).
The per-frame NumPy loops of ``stft_norm`` / ``istft_norm``
(transform.py:277-396) are replaced by the HIP kernels of ``csrc/fft.hip``.

``precision='float64'`` (default) runs the float64 kernels so that results agree
with the reference to ~1e-13; ``precision='float32'`` is what the fused
separation path uses internally.
"""
import re

import numpy as np

from . import _lib
from .runtime import StftPlan, default_context


def sinebell(lengthWindow):
    """``sin(pi t / L)`` window (transform.py:35-49)."""
    return np.sin((np.pi * (np.arange(lengthWindow))) / (1.0 * lengthWindow))


class Transforms(object):
    """Base class (transform.py:52-198): parameters, tensor save/load helpers."""

    def __init__(self, ttype='fft', bins=48, frameSize=1024, hopSize=256, tffmin=25, tffmax=18000, iscale='lin',
                 suffix='', sampleRate=44100, window=np.hanning, precision='float64', device=None):
        self.bins = bins
        self.frameSize = frameSize
        self.hopSize = hopSize
        self.fmin = tffmin
        self.fmax = tffmax
        self.iscale = iscale
        self.suffix = suffix
        self.sampleRate = sampleRate
        self.ttype = ttype
        self.window = window(self.frameSize)  # materialised once, transform.py:78
        if precision not in ('float64', 'float32'):
            raise ValueError("precision must be 'float64' or 'float32'")
        self.precision = precision
        self._device = device
        self._plan = None

    # -- transform.py:80-131 -------------------------------------------------------------------
    def compute_transform(self, audio, out_path=None, phase=False, save=True):
        """STFT of every column of ``audio[t, i]``; saved as ``.data``/``.shape`` or returned as
        ``[i, T, F]`` float64."""
        self.out_path = out_path
        mags = phs = None
        for i in range(audio.shape[1]):
            if phase:
                mag, ph = self.compute_file(audio[:, i], phase=True, sampleRate=self.sampleRate)
            else:
                mag = self.compute_file(audio[:, i], phase=False, sampleRate=self.sampleRate)
            if i == 0:
                mags = np.zeros((audio.shape[1], mag.shape[0], mag.shape[1]))
                if phase:
                    phs = np.zeros((audio.shape[1], ph.shape[0], ph.shape[1]))
            mags[i] = mag
            if phase:
                phs[i] = ph
        if save and self.out_path is not None:
            self.saveTensor(mags, '_' + self.suffix + '_m_')
            if phase:
                self.saveTensor(phs, '_' + self.suffix + '_p_')
            return None
        if phase:
            return mags, phs
        return mags

    def compute_file(self, audio, phase=False):
        return None

    def compute_inverse(self, mag, phase):
        return None

    # -- transform.py:159-197 ---------------------------------------------------------------------
    def saveTensor(self, t, name='_cqt_m_'):
        t.tofile(self.out_path.replace('.data', name + '.data'))
        self.shape = t.shape
        self.save_shape(self.out_path.replace('.data', name + '.shape'), t.shape)

    def loadTensor(self, name='_cqt_m_'):
        f_in = np.fromfile(self.out_path.replace('.data', name + '.data'))
        shape = self.get_shape(self.out_path.replace('.data', '.shape'))
        if self.shape == shape:
            return f_in.reshape(shape)
        print('Shape of loaded array does not match with the original shape of the transform')

    def save_shape(self, shape_file, shape):
        with open(shape_file, 'w') as fout:
            fout.write(u'#' + '\t'.join(str(e) for e in shape) + '\n')

    def get_shape(self, shape_file):
        with open(shape_file, 'rb') as f:
            line = f.readline().decode('ascii')
            if line.startswith('#'):
                return tuple(map(int, re.findall(r'(\d+)', line)))
            raise IOError('Failed to find shape in file')


class transformFFT(Transforms):
    """STFT on the GPU with the reference's API (transform.py:201-274).

    >>> tt1 = transformFFT(frameSize=2048, hopSize=512, sampleRate=44100)
    >>> mag, ph = tt1.compute_file(audio, phase=True)
    >>> audio = tt1.compute_inverse(mag, ph)
    """

    def __init__(self, ttype='fft', bins=48, frameSize=1024, hopSize=256, tffmin=25, tffmax=18000, iscale='lin',
                 suffix='', sampleRate=44100, window=np.hanning, precision='float64', device=None):
        super(transformFFT, self).__init__(ttype='fft', bins=bins, frameSize=frameSize, hopSize=hopSize,
                                           tffmin=tffmin, tffmax=tffmax, iscale=iscale, suffix=suffix,
                                           sampleRate=sampleRate, window=window, precision=precision, device=device)

    def _get_plan(self):
        if self._plan is None:
            self._plan = StftPlan(default_context(self._device), self.frameSize, self.hopSize, self.window)
        return self._plan

    def _np_dtype(self):
        return np.float64 if self.precision == 'float64' else np.float32

    def compute_file(self, audio, phase=False, sampleRate=44100):
        """``mag = |STFT| / sqrt(frameSize)`` (``[T, frameSize/2+1]`` float64) and, with
        ``phase=True``, ``angle(STFT)`` (transform.py:243-247)."""
        plan = self._get_plan()
        audio = np.asarray(audio)
        if audio.ndim != 1:
            audio = audio.reshape(-1)  # stft_norm uses data.size / 1-D concatenation (transform.py:303,316)
        a = plan.ctx.to_device(audio, self._np_dtype())
        mag, ph = plan.forward(a, phase=phase)
        mag = mag.cpu().numpy().astype(np.float64, copy=False)
        if phase:
            return mag, ph.cpu().numpy().astype(np.float64, copy=False)
        return mag

    def compute_inverse(self, mag, phase, sampleRate=44100):
        """Inverse STFT of ``mag * sqrt(frameSize) * exp(1j*phase)`` (transform.py:271-273); length
        ``hopSize*(T-1) + frameSize/2``."""
        plan = self._get_plan()
        mag = np.asarray(mag)
        phase = np.asarray(phase)
        if mag.shape != phase.shape or mag.ndim != 2:
            raise ValueError("operands could not be broadcast together with shapes %r %r" % (mag.shape, phase.shape))
        if mag.shape[1] != plan.bins:
            raise ValueError("expected %d frequency bins, got %d" % (plan.bins, mag.shape[1]))
        m = plan.ctx.to_device(mag, self._np_dtype())
        p = plan.ctx.to_device(phase, self._np_dtype())
        out = plan.inverse(m, p)
        return out.cpu().numpy().astype(np.float64, copy=False)


# BASELINE.json spells the class with a capital T
TransformFFT = transformFFT


# Functional forms with the signatures of the copies inside the separate_*.py scripts
# (examples/dsd100/separate_dsd.py:24-41).
def compute_file(audio, phase=False, frameSize=1024, hopSize=512, sampleRate=44100, window=np.hanning,
                 precision='float64'):
    tt = transformFFT(frameSize=frameSize, hopSize=hopSize, sampleRate=sampleRate, window=window, precision=precision)
    return tt.compute_file(audio, phase=phase, sampleRate=sampleRate)


def compute_inverse(mag, phase, frameSize=1024, hopSize=512, sampleRate=44100, window=np.hanning,
                    precision='float64'):
    tt = transformFFT(frameSize=frameSize, hopSize=hopSize, sampleRate=sampleRate, window=window, precision=precision)
    return tt.compute_inverse(mag, phase, sampleRate=sampleRate)
