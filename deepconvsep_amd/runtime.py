"""Thin object layer over the C ABI: Context, StftPlan, Network.

PyTorch-ROCm is used for exactly three things here: allocating device memory
(``torch.empty(..., device='cuda')``), host<->device copies, and naming the HIP
stream the kernels run on.  No torch operator computes anything on the path.
"""
import ctypes
from ctypes import POINTER, byref, c_double, c_int64, c_void_p

import numpy as np

from . import _lib
from .arch import (ARCHS, EPS_A, EPS_B, TIE_ALL, TIE_FIRST, TILER_LIBRARY, TILER_SCRIPT, check_params, live_params,
                   resolve as resolve_arch)


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("deepconvsep_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                           "and there is no CPU fallback")
    return torch


def _on_ctx_stream(method):
    """Run a method of an object with a ``.ctx`` inside ``torch.cuda.stream(ctx.torch_stream)``: the tensors it
    allocates belong to that stream in torch's caching allocator and the copies it issues are ordered with the
    kernels libdcs enqueues there, whatever stream is current in the caller."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with self.ctx.stream_scope():
            return method(self, *args, **kwargs)
    return wrapper


class Context(object):
    """One device + one HIP stream (``dcs_ctx``).

    The stream is the one current on ``device`` when the context is made (or ``stream=``); it is kept as a
    ``torch.cuda.Stream`` so that every host<->device copy, allocation and ``.cpu()`` the package issues runs on the
    stream the kernels run on (:meth:`stream_scope`).  A caller that produces inputs or consumes outputs on another
    stream synchronises with :attr:`torch_stream` as with any torch stream."""

    def __init__(self, device=None, stream=None):
        torch = require_gpu()
        lib = _lib.load()
        if device is None:
            device = stream.device.index if stream is not None else torch.cuda.current_device()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.torch_stream = stream if stream is not None else torch.cuda.current_stream(self.device_index)
        if self.torch_stream.device.index != self.device_index:
            raise ValueError("stream belongs to device %d, context to device %d"
                             % (self.torch_stream.device.index, self.device_index))
        h = c_void_p()
        _lib.check(lib.dcs_create(self.device_index, c_void_p(self.torch_stream.cuda_stream), byref(h)))
        self._h = h
        self._lib = lib

    def stream_scope(self):
        """``with ctx.stream_scope():`` makes the context's stream torch's current stream."""
        return _torch().cuda.stream(self.torch_stream)

    # -- memory plumbing ------------------------------------------------------------------
    def to_device(self, array, dtype):
        torch = _torch()
        a = np.ascontiguousarray(array, dtype=dtype)
        with self.stream_scope():
            return torch.from_numpy(a).to(self.device)

    def to_host(self, tensor):
        """Device tensor -> ndarray, ordered after everything enqueued on the context's stream."""
        with self.stream_scope():
            return tensor.cpu().numpy()

    def empty(self, shape, dtype):
        torch = _torch()
        tdt = {np.float32: torch.float32, np.float64: torch.float64}[dtype]
        with self.stream_scope():
            return torch.empty(tuple(int(s) for s in shape), dtype=tdt, device=self.device)

    def synchronize(self):
        _lib.check(self._lib.dcs_synchronize(self._h))

    def check_guards(self):
        """``dcs_debug_check_guards``: with ``DCS_WS_GUARD=<bytes>`` in the environment, verify the red zones around every
        scratch block libdcs holds (raises on damage); returns the number of guarded blocks."""
        n = c_int64()
        _lib.check(self._lib.dcs_debug_check_guards(self._h, byref(n)))
        return n.value

    # -- kernel timing (bench.py) ---------------------------------------------------------------
    def timing_stride(self, stride):
        _lib.check(self._lib.dcs_timing_stride(self._h, int(stride)))

    def timing(self, tags):
        """Bracket the kernels of the named tags (``_lib.TAGS`` keys; ``'all'``; ``None`` = off) with HIP events."""
        if not tags:
            mask = 0
        elif tags == 'all':
            mask = (1 << len(_lib.TAGS)) - 1
        else:
            mask = 0
            for t in tags:
                mask |= 1 << _lib.TAGS[t]
        _lib.check(self._lib.dcs_timing_enable(self._h, mask))

    def timing_reset(self):
        _lib.check(self._lib.dcs_timing_reset(self._h))

    def timing_query(self, tag):
        ms, cnt = c_double(), c_int64()
        _lib.check(self._lib.dcs_timing_query(self._h, _lib.TAGS[tag], byref(ms), byref(cnt)))
        return ms.value, cnt.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dcs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    torch = require_gpu()
    idx = torch.cuda.current_device() if device is None else int(device)
    if idx not in _default_ctx:
        _default_ctx[idx] = Context(idx)
    return _default_ctx[idx]


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)


class StftPlan(object):
    """``dcs_stft``: window + twiddle tables for one (frameSize, hopSize, window)."""

    def __init__(self, ctx, frame, hop, window):
        self.ctx = ctx
        self.frame, self.hop = int(frame), int(hop)
        self.bins = self.frame // 2 + 1
        w = np.ascontiguousarray(window, dtype=np.float64)
        if w.shape != (self.frame,):
            raise ValueError("window must have frameSize=%d samples, got %r" % (self.frame, w.shape))
        self.window = w
        h = c_void_p()
        _lib.check(ctx._lib.dcs_stft_plan(ctx._h, self.frame, self.hop, w.ctypes.data_as(POINTER(c_double)), byref(h)))
        self._h = h

    @_on_ctx_stream
    def forward(self, audio_t, phase=True, rows_out=None, ld=None):
        """audio_t: 1-D float32/float64 device tensor.  Returns (mag, phase|None) device tensors
        ``[rows_out, ld]`` (defaults: the reference's dense ``[T, bins]``)."""
        torch = _torch()
        L = int(audio_t.numel())
        T = _lib.frame_count(L, self.hop)
        rows = T if rows_out is None else int(rows_out)
        ld = self.bins if ld is None else int(ld)
        f64 = audio_t.dtype == torch.float64
        mag = torch.empty((rows, ld), dtype=audio_t.dtype, device=audio_t.device)
        ph = torch.empty((rows, ld), dtype=audio_t.dtype, device=audio_t.device) if phase else None
        fn = self.ctx._lib.dcs_stft_forward_f64 if f64 else self.ctx._lib.dcs_stft_forward_f32
        _lib.check(fn(self._h, _ptr(audio_t), L, _ptr(mag), _ptr(ph), ld, rows))
        return mag, ph

    @_on_ctx_stream
    def forward_clips(self, audio_t, phase=True, mag_out=None, phase_out=None):
        """``compute_transform`` (transform.py:80-131): audio_t ``[clips, samples]`` float32 / float64 device tensor (rows
        contiguous) -> (mag, phase|None) ``[clips, T, bins]`` from ONE launch (``dcs_stft_forward_f32/f64_clips``)."""
        torch = _torch()
        if audio_t.dim() != 2 or audio_t.stride(1) != 1:
            raise ValueError("forward_clips expects a [clips, samples] tensor with contiguous rows")
        B, L = int(audio_t.shape[0]), int(audio_t.shape[1])
        T = _lib.frame_count(L, self.hop)
        f64 = audio_t.dtype == torch.float64
        mag = mag_out if mag_out is not None else torch.empty((B, T, self.bins), dtype=audio_t.dtype, device=audio_t.device)
        ph = None
        if phase:
            ph = phase_out if phase_out is not None else torch.empty((B, T, self.bins), dtype=audio_t.dtype, device=audio_t.device)
        fn = self.ctx._lib.dcs_stft_forward_f64_clips if f64 else self.ctx._lib.dcs_stft_forward_f32_clips
        _lib.check(fn(self._h, _ptr(audio_t), L, B, int(audio_t.stride(0)), _ptr(mag), _ptr(ph), self.bins, T))
        return mag, ph

    @_on_ctx_stream
    def inverse(self, mag_t, phase_t, n_out=None, pre_div=1.0):
        """mag_t ``[S, T, ld]`` or ``[T, ld]``, phase_t ``[T, ld]`` (same ld).  Returns ``[S, n_out]``
        (or ``[n_out]``)."""
        torch = _torch()
        squeeze = mag_t.dim() == 2
        if squeeze:
            mag_t = mag_t.unsqueeze(0)
        S, T, ld = (int(x) for x in mag_t.shape)
        if tuple(phase_t.shape) != (T, ld):
            raise ValueError("phase shape %r does not match magnitude %r" % (tuple(phase_t.shape), (T, ld)))
        mag_t = mag_t.contiguous()
        phase_t = phase_t.contiguous()
        full = _lib.inverse_length(T, self.hop, self.frame)
        n_out = full if n_out is None else min(int(n_out), full)
        out = torch.empty((S, n_out), dtype=mag_t.dtype, device=mag_t.device)
        if mag_t.dtype == torch.float64:
            _lib.check(self.ctx._lib.dcs_stft_inverse_f64(self._h, _ptr(mag_t), T * ld, _ptr(phase_t), ld, T, S,
                                                          float(pre_div), _ptr(out), n_out))
        else:
            _lib.check(self.ctx._lib.dcs_stft_inverse_f32(self._h, _ptr(mag_t), T * ld, _ptr(phase_t), ld, T, S,
                                                          float(pre_div), _ptr(out), n_out))
        return out[0] if squeeze else out

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.dcs_stft_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Network(object):
    """``dcs_model``: one ``build_ca`` graph with its parameters resident in HBM.

    ``params`` is the list ``lasagne.layers.get_all_param_values`` pickled by the reference's
    trainers (examples/dsd100/trainCNN.py:59-64); a count/shape mismatch raises ``ValueError``
    like ``set_all_param_values`` does (examples/dsd100/separate_dsd.py:250).
    """

    def __init__(self, ctx, arch, params, time_context=30, feat_size=513, live_only=True):
        self.ctx = ctx
        self.tc, self.F = int(time_context), int(feat_size)
        params = [np.asarray(p) for p in params]
        # 'ikala': pooled or trainer (no-pool) graph by fc.W; 'bach10_si': 17-array or single-branch 11-array graph by count
        self.arch = resolve_arch(arch, params, self.tc, self.F)
        check_params(self.arch, params, self.tc, self.F)
        self.graph_arch = self.arch                                  # the graph the .pkl belongs to
        if live_only:
            # the 17-array score-informed graph: three of its four decoder branches never reach predict_function2 -- they are
            # neither uploaded nor allocated (3 x 171 MB of dense weights at 2049 bins); forward_raw then returns the four
            # live channels.  live_only=False keeps the whole graph (what lasagne.layers.get_output would evaluate).
            self.arch, params = live_params(self.arch, params)
        # weight tensors hosted by torch (float32, as the pickles store them)
        self._params = [ctx.to_device(p, np.float32) for p in params]
        n = len(self._params)
        ptrs = (c_void_p * n)(*[p.data_ptr() for p in self._params])
        shapes = (c_int64 * (4 * n))()
        for i, p in enumerate(params):
            shp = list(p.shape) + [1] * (4 - p.ndim)
            for k in range(4):
                shapes[4 * i + k] = shp[k]
        h = c_void_p()
        _lib.check(ctx._lib.dcs_model_create(ctx._h, self.arch.code, self.arch.C, self.tc, self.F, ptrs, shapes, n,
                                             byref(h)))
        self._h = h
        self.S = int(ctx._lib.dcs_model_num_sources(h))
        self.out_channels = int(ctx._lib.dcs_model_out_channels(h))

    def final_kernel(self, n_frames, n_clips=1, eps_mode=None):
        """Which kernel the fused path runs for the last decoder stage on a launch of this size: ``'f32x64'``,
        ``'f32x128'``, ``'bf16x3'`` or ``'one_batch'`` (the kernel of csrc/dsd_lat.hip a single short clip takes) -- ``dcs_model_final_kernel``; None for graphs
        without a fused decoder."""
        eps = self.arch.eps_mode if eps_mode is None else eps_mode
        code = int(self.ctx._lib.dcs_model_final_kernel(self._h, int(n_frames), int(n_clips), int(eps)))
        return {0: 'f32x64', 1: 'f32x128', 2: 'bf16x3', 3: 'one_batch'}.get(code)

    def set_latency_stages(self, stages):
        """Stages of the fused path that run on the one-batch kernels (``dcs_model_set_latency_stages``): -1 automatic,
        0 none, else bits 1 STFT, 2 conv1, 4 conv2, 8 bottleneck, 16 per-source dense, 32 transposed conv2, 64 final,
        128 iSTFT."""
        _lib.check(self.ctx._lib.dcs_model_set_latency_stages(self._h, int(stages)))

    def set_score_semantics(self, normalise='max', mixture='ch0'):
        """Score-informed graphs: ``normalise`` ``'max'`` (the separate script: every instrument's harmonic mask divided by
        its own maximum, separate_bach10.py:195) or ``'sum'`` (the trainers' dataset class: by the sum over the instruments,
        dataset.py:862); ``mixture`` ``'ch0'`` (script: soft masks x input channel 0, :485) or ``'sum'`` (trainers: x the sum
        of the input channels, trainCNNrwc.py:258-263).  ``dcs_model_set_score_semantics``."""
        try:
            n, m = {'max': 0, 'sum': 1}[normalise], {'ch0': 0, 'sum': 1}[mixture]
        except KeyError:
            raise ValueError("normalise must be 'max' or 'sum', mixture 'ch0' or 'sum'")
        _lib.check(self.ctx._lib.dcs_model_set_score_semantics(self._h, n, m))
        self.score_normalise, self.score_mixture = normalise, mixture

    def set_conv_precision(self, dtype):
        """``'f16'``: conv2 and its transpose use f16-input / f32-accumulate MFMA (BASELINE config 3);
        ``'f32'`` (default): exact f32."""
        if dtype not in ('f16', 'f32'):
            raise ValueError("conv precision must be 'f16' or 'f32'")
        _lib.check(self.ctx._lib.dcs_model_set_conv_precision(self._h, 1 if dtype == 'f16' else 0))

    @_on_ctx_stream
    def forward_masked(self, tiles_t, eps_mode=None, tie_mode=TIE_ALL):
        """tiles_t ``[n, C, tc, F]`` float32 device tensor -> ``[S, n, tc, F]``."""
        torch = _torch()
        n = int(tiles_t.shape[0])
        if tuple(tiles_t.shape[1:]) != (self.arch.C, self.tc, self.F):
            raise ValueError("input tiles %r do not match network input (n, %d, %d, %d)"
                             % (tuple(tiles_t.shape), self.arch.C, self.tc, self.F))
        tiles_t = tiles_t.contiguous()
        out = torch.empty((self.S, n, self.tc, self.F), dtype=torch.float32, device=tiles_t.device)
        eps = self.arch.eps_mode if eps_mode is None else eps_mode
        _lib.check(self.ctx._lib.dcs_model_forward_masked(self._h, _ptr(tiles_t), n, int(eps), int(tie_mode), _ptr(out)))
        return out

    @_on_ctx_stream
    def forward_raw(self, tiles_t, tie_mode=TIE_ALL):
        """Network output before masking, ``[n, out_channels, tc, F]`` (testing aid).  For the DSD
        graph the kernel emits channel-major ``[out_channels, n, tc, F]``; it is permuted here."""
        torch = _torch()
        n = int(tiles_t.shape[0])
        tiles_t = tiles_t.contiguous()
        out = torch.empty((self.out_channels, n, self.tc, self.F), dtype=torch.float32, device=tiles_t.device)
        _lib.check(self.ctx._lib.dcs_model_forward(self._h, _ptr(tiles_t), n, int(tie_mode), _ptr(out)))
        return out.permute(1, 0, 2, 3).contiguous()

    @_on_ctx_stream
    def separate(self, plan, audio_t, overlap, tiler=TILER_SCRIPT, scale=0.3, eps_mode=None, tie_mode=TIE_ALL,
                 out=None):
        """Fused file-level path: 1-D float32 device tensor -> ``[S, L]`` float32 PCM."""
        torch = _torch()
        L = int(audio_t.numel())
        if out is None:
            out = torch.empty((self.S, L), dtype=torch.float32, device=audio_t.device)
        eps = self.arch.eps_mode if eps_mode is None else eps_mode
        nt, nf = c_int64(), c_int64()
        _lib.check(self.ctx._lib.dcs_separate(self._h, plan._h, _ptr(audio_t), L, int(overlap), int(tiler),
                                              float(scale), int(eps), int(tie_mode), _ptr(out), byref(nt), byref(nf)))
        self.last_tiles, self.last_frames = nt.value, nf.value
        return out

    @_on_ctx_stream
    def separate_scoreinformed(self, plan, audio_t, notes, overlap, scale=0.3, eps_mode=None, tie_mode=TIE_ALL, out=None):
        """``dcs_separate_scoreinformed``: 1-D float32 device tensor + the note table ``[C, P, W]`` of
        :func:`deepconvsep_amd.score` (``expandMidi``'s layout) -> ``[S, L]`` float32 PCM, one call."""
        torch = _torch()
        notes = np.ascontiguousarray(notes, dtype=np.float64)
        if notes.ndim != 3:
            raise ValueError("notes must be [instruments, notes, 2*nharmonics+3]")
        L = int(audio_t.numel())
        if out is None:
            out = torch.empty((self.S, L), dtype=torch.float32, device=audio_t.device)
        eps = self.arch.eps_mode if eps_mode is None else eps_mode
        nt, nf = c_int64(), c_int64()
        _lib.check(self.ctx._lib.dcs_separate_scoreinformed(
            self._h, plan._h, _ptr(audio_t), L, notes.ctypes.data_as(POINTER(c_double)), int(notes.shape[0]),
            int(notes.shape[1]), int(notes.shape[2]), int(overlap), float(scale), int(eps), int(tie_mode), _ptr(out),
            byref(nt), byref(nf)))
        self.last_tiles, self.last_frames = nt.value, nf.value
        return out

    @_on_ctx_stream
    def separate_batch(self, plan, audio_t, overlap, tiler=TILER_SCRIPT, scale=0.3, eps_mode=None, tie_mode=TIE_ALL,
                       out=None):
        """Fused path for equal-length clips sharing one set of launches: ``[B, L]`` float32 device tensor
        (rows contiguous) -> ``[B, S, L]`` float32 PCM; every clip gets the tiles and cross-fade :meth:`separate`
        would give it alone (DSD / hiphop graph)."""
        torch = _torch()
        if audio_t.dim() != 2 or audio_t.stride(1) != 1:
            raise ValueError("separate_batch expects a [clips, samples] tensor with contiguous rows")
        B, L = int(audio_t.shape[0]), int(audio_t.shape[1])
        if out is None:
            out = torch.empty((B, self.S, L), dtype=torch.float32, device=audio_t.device)
        eps = self.arch.eps_mode if eps_mode is None else eps_mode
        nt, nf = c_int64(), c_int64()
        _lib.check(self.ctx._lib.dcs_separate_batch(self._h, plan._h, _ptr(audio_t), L, B, int(audio_t.stride(0)),
                                                    int(overlap), int(tiler), float(scale), int(eps), int(tie_mode),
                                                    _ptr(out), byref(nt), byref(nf)))
        self.last_tiles, self.last_frames = nt.value, nf.value
        return out

    @_on_ctx_stream
    def separate_ragged(self, plan, audio_t, lengths, overlap, tiler=TILER_SCRIPT, scale=0.3, eps_mode=None,
                        tie_mode=TIE_ALL, out=None):
        """Clips of different lengths sharing one set of launches: ``audio_t [B, Lmax]`` float32 device tensor (row c
        holds ``lengths[c]`` samples), returns ``[B, S, Lmax]`` float32 PCM of which ``[c, :, :lengths[c]]`` is valid.
        Every clip gets the frames, tiles and cross-fade :meth:`separate` would give it alone (every single-channel
        graph; frameSize 1024 / 2048 / 4096)."""
        torch = _torch()
        if audio_t.dim() != 2 or audio_t.stride(1) != 1:
            raise ValueError("separate_ragged expects a [clips, samples] tensor with contiguous rows")
        B, Lmax = int(audio_t.shape[0]), int(audio_t.shape[1])
        lens = (c_int64 * B)(*[int(x) for x in lengths])
        if len(lengths) != B or max(lens) > Lmax:
            raise ValueError("separate_ragged: %d lengths for %d clips of at most %d samples" % (len(lengths), B, Lmax))
        if out is None:
            out = torch.zeros((B, self.S, Lmax), dtype=torch.float32, device=audio_t.device)
        eps = self.arch.eps_mode if eps_mode is None else eps_mode
        nt, nf = (c_int64 * B)(), (c_int64 * B)()
        _lib.check(self.ctx._lib.dcs_separate_ragged(self._h, plan._h, _ptr(audio_t), lens, B, int(audio_t.stride(0)),
                                                     int(overlap), int(tiler), float(scale), int(eps), int(tie_mode),
                                                     _ptr(out), int(out.stride(1)), nt, nf))
        self.last_tiles, self.last_frames = list(nt), list(nf)
        return out

    @_on_ctx_stream
    def separate_stereo(self, plan, audio_t, overlap, tiler=TILER_LIBRARY, scale=0.3, want_spectra=False):
        """Stereo (ILD) graph: ``[2, L]`` float32 device tensor (rows contiguous) -> PCM ``[2, S, L]`` (and, with
        ``want_spectra``, the cross-faded scaled magnitudes ``[2, S, T, F]``)."""
        torch = _torch()
        if audio_t.dim() != 2 or audio_t.shape[0] != 2 or audio_t.stride(1) != 1:
            raise ValueError("separate_stereo expects a [2, samples] tensor with contiguous rows")
        L = int(audio_t.shape[1])
        T = _lib.frame_count(L, plan.hop)
        out = torch.empty((2, self.S, L), dtype=torch.float32, device=audio_t.device)
        sep = torch.empty((2, self.S, T, self.F), dtype=torch.float32, device=audio_t.device) if want_spectra else None
        nt, nf = c_int64(), c_int64()
        _lib.check(self.ctx._lib.dcs_separate_stereo(self._h, plan._h, _ptr(audio_t), L, int(audio_t.stride(0)),
                                                     int(overlap), int(tiler), float(scale), _ptr(out),
                                                     _ptr(sep) if sep is not None else None, self.F, byref(nt), byref(nf)))
        self.last_tiles, self.last_frames = nt.value, nf.value
        return (out, sep) if want_spectra else out

    @_on_ctx_stream
    def separate_spectra(self, plan, audio_t, overlap, tiler=TILER_SCRIPT, scale=0.3, eps_mode=None,
                         tie_mode=TIE_ALL):
        """Fused path stopped before the iSTFT: (sep ``[S,T,F]``, mag ``[T,F]``, phase ``[T,F]``)."""
        torch = _torch()
        L = int(audio_t.numel())
        T = _lib.frame_count(L, plan.hop)
        F = self.F
        sep = torch.empty((self.S, T, F), dtype=torch.float32, device=audio_t.device)
        mag = torch.empty((T, F), dtype=torch.float32, device=audio_t.device)
        ph = torch.empty((T, F), dtype=torch.float32, device=audio_t.device)
        eps = self.arch.eps_mode if eps_mode is None else eps_mode
        _lib.check(self.ctx._lib.dcs_separate_spectra(self._h, plan._h, _ptr(audio_t), L, int(overlap), int(tiler),
                                                      float(scale), int(eps), int(tie_mode), _ptr(sep), _ptr(mag),
                                                      _ptr(ph), F))
        return sep, mag, ph

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.dcs_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tile(ctx, mag_t, time_context, overlap, tiler, scale=1.0):
    """``generate_overlapadd`` on the device: mag_t ``[T,F]`` or ``[C,T,F]`` float32 -> (tiles
    ``[n,C,tc,F]``, n)."""
    torch = _torch()
    if mag_t.dim() == 2:
        mag_t = mag_t.unsqueeze(0)
    with ctx.stream_scope():
        mag_t = mag_t.contiguous()
    C, T, F = (int(x) for x in mag_t.shape)
    n = _lib.tile_count(T, time_context, overlap, tiler)
    with ctx.stream_scope():
        tiles = torch.empty((n, C, time_context, F), dtype=torch.float32, device=mag_t.device)
        if n:
            _lib.check(ctx._lib.dcs_tile(ctx._h, _ptr(mag_t), T * F, F, C, T, F, int(time_context), int(overlap),
                                         int(tiler), float(scale), _ptr(tiles), n))
    return tiles, n


def pcm_to_int16(ctx, pcm_t, out=None):
    """``(audio_out * 32767).astype('int16')`` on the device (separate_dsd.py:307-309): float32 tensor of any shape
    (contiguous) -> int16 tensor of the same shape."""
    torch = _torch()
    with ctx.stream_scope():
        pcm_t = pcm_t.contiguous()
        if out is None:
            out = torch.empty(pcm_t.shape, dtype=torch.int16, device=pcm_t.device)
        _lib.check(ctx._lib.dcs_pcm_to_int16(ctx._h, _ptr(pcm_t), int(pcm_t.numel()), _ptr(out)))
    return out


def pcm16_to_float(ctx, pcm16_t, channels, mode=0, out=None):
    """``dcs_pcm16_to_float``: int16 wav frames on the device -> the mono float32 signal the scripts separate.  ``pcm16_t``
    ``[B, stride]`` int16 (row c = clip c's interleaved frames, zero padded), ``channels`` per frame; returns ``[B, stride //
    channels]`` float32.  ``mode`` 0: (L + R) / 2 for two or more channels, the channel itself for mono (separate_dsd.py:
    285-287); 1: L + R (separate_ikala.py:229)."""
    torch = _torch()
    if pcm16_t.dim() != 2 or pcm16_t.stride(1) != 1 or pcm16_t.dtype != torch.int16:
        raise ValueError("pcm16_to_float expects a [clips, interleaved samples] int16 tensor with contiguous rows")
    B, n = int(pcm16_t.shape[0]), int(pcm16_t.shape[1]) // int(channels)
    with ctx.stream_scope():
        if out is None:
            out = torch.empty((B, n), dtype=torch.float32, device=pcm16_t.device)
        _lib.check(ctx._lib.dcs_pcm16_to_float(ctx._h, _ptr(pcm16_t), n, int(channels), int(mode), B, int(pcm16_t.stride(0)),
                                               _ptr(out), int(out.stride(0))))
    return out


def overlap_add(ctx, out_t, overlap):
    """``overlapadd_multi`` on the device: out_t ``[S, n, tc, F]`` float32 -> ``[S, n*(tc-ov)+tc, F]``."""
    torch = _torch()
    S, n, tc, F = (int(x) for x in out_t.shape)
    rows = n * (tc - overlap) + tc
    rise = np.ascontiguousarray(np.linspace(0., 1.0, num=overlap), dtype=np.float64)
    with ctx.stream_scope():
        out_t = out_t.contiguous()
        sep = torch.empty((S, rows, F), dtype=torch.float32, device=out_t.device)
        _lib.check(ctx._lib.dcs_overlap_add(ctx._h, _ptr(out_t), n, S, tc, int(overlap), F,
                                            rise.ctypes.data_as(POINTER(c_double)), _ptr(sep), rows * F, F))
    return sep
