"""Host-side mirror of the reference's separation scripts, running on the MI355X.

Function names, argument meaning, defaults and error behaviour follow
``examples/dsd100/separate_dsd.py`` (and its ikala / bach10 / hiphop siblings):

  load_model            separate_dsd.py:17-21
  generate_overlapadd   separate_dsd.py:114-135 (script tiler) / util.py:220-248 (library tiler)
  overlapadd_multi      separate_dsd.py:139-169 = util.py:297-327
  overlapadd            util.py:251-294 (2 sources, separate_ikala.py:138-169)
  build_ca + theano.function(predict_function2)  ->  ``PredictFunction``   separate_dsd.py:172-273
  train_auto            separate_dsd.py:239-313

The arithmetic runs in libdcs (HIP); NumPy only moves data in and out.
"""
import os
import pickle

import numpy as np
import scipy.io.wavfile

from . import _lib
from .arch import ARCHS, TIE_ALL, TILER_LIBRARY, TILER_SCRIPT
from .runtime import Network, StftPlan, _on_ctx_stream, default_context, overlap_add, pcm16_to_float, pcm_to_int16, tile


def blackmanharris(n):
    """``scipy.signal.blackmanharris`` of the reference's era (separate_bach10.py:4) lives in
    ``scipy.signal.windows`` today."""
    from scipy.signal.windows import blackmanharris as _bh
    return _bh(n)


def load_model(filename):
    """Python-2 cPickle list of float32 ndarrays (examples/dsd100/trainCNN.py:59-64)."""
    with open(filename, 'rb') as f:
        try:
            params = pickle.load(f, encoding='latin1')
        except TypeError:
            params = pickle.load(f)
    return params


def save_model(filename, params):
    with open(filename, 'wb') as f:
        pickle.dump([np.asarray(p, dtype=np.float32) for p in params], f, protocol=2)


_TILERS = {'script': TILER_SCRIPT, 'library': TILER_LIBRARY, TILER_SCRIPT: TILER_SCRIPT, TILER_LIBRARY: TILER_LIBRARY}


def generate_overlapadd(allmix, input_size=513, time_context=30, overlap=10, batch_size=32, sampleRate=44100,
                        tiler='script', device=None):
    """Tiles of ``time_context`` frames every ``time_context-overlap`` frames, packed as
    ``fbatch[nb, batch_size, C, time_context, input_size]`` (float64), plus the tile count.

    ``tiler='script'`` drops the tail like separate_dsd.py:123 (``start+tc < T``); slots past the
    last tile -- uninitialised ``np.empty`` memory in the reference -- are zero here.
    ``tiler='library'`` zero-pads like util.py:230-243 and accepts ``[C, T, F]`` input.
    """
    allmix = np.asarray(allmix)
    code = _TILERS[tiler]
    if code == TILER_SCRIPT:
        if allmix.ndim != 2 or input_size != allmix.shape[-1]:
            # separate_dsd.py:119: the body is skipped and `fbatch` is unbound
            raise UnboundLocalError("local variable 'fbatch' referenced before assignment")
    else:
        assert input_size == allmix.shape[-1], "Feature size must be the same as the last dimension of the spectrogram"
    ctx = default_context(device)
    tiles, n = tile(ctx, ctx.to_device(allmix, np.float32), time_context, overlap, code, 1.0)
    C = tiles.shape[1]
    nb = int(np.ceil(float(n) / batch_size))
    fbatch = np.zeros([nb, batch_size, C, time_context, input_size])
    if n:
        fbatch.reshape(nb * batch_size, C, time_context, input_size)[:n] = ctx.to_host(tiles)
    return fbatch, n


def _tiles_from_batches(fbatch, nchunks):
    """``[nb, S, B, 1, tc, F]`` (np.array of predict_function2 outputs) -> ``[S, nchunks, tc, F]``."""
    fbatch = np.asarray(fbatch)
    nb, S, B = fbatch.shape[0], fbatch.shape[1], fbatch.shape[2]
    tc, F = fbatch.shape[-2], fbatch.shape[-1]
    t = np.transpose(fbatch[:, :, :, 0], (1, 0, 2, 3, 4)).reshape(S, nb * B, tc, F)
    return np.ascontiguousarray(t[:, :nchunks])


def overlapadd_multi(fbatch, obatch, nchunks, overlap=10, device=None):
    """Cross-fade stitch of the per-source network outputs (util.py:297-327) ->
    ``sep[S, nchunks*(tc-overlap)+tc, F]`` float64."""
    ctx = default_context(device)
    tiles = _tiles_from_batches(fbatch, nchunks)
    sep = overlap_add(ctx, ctx.to_device(tiles, np.float32), overlap)
    return ctx.to_host(sep).astype(np.float64)


def overlapadd(fbatch, obatch, nchunks, overlap=10, device=None):
    """2-source form (util.py:251-294): returns ``(sep1, sep2)``."""
    sep = overlapadd_multi(np.asarray(fbatch)[:, :2], obatch, nchunks, overlap=overlap, device=device)
    return sep[0], sep[1]


class PredictFunction(object):
    """What ``theano.function([input_var2], [source_1..source_S])`` is in the reference
    (separate_dsd.py:273): ``batch[B, C, tc, F] -> list of S arrays [B, 1, tc, F]``.

    ``batch_size`` is not baked into the graph here (the reference fixes it through the
    InputLayer / ReshapeLayer shapes, separate_dsd.py:192,210): any B works.
    """

    def __init__(self, arch, params, time_context=30, input_size=513, eps_mode=None, tie_mode=TIE_ALL, device=None):
        self.ctx = default_context(device)
        self.net = Network(self.ctx, arch, params, time_context, input_size)
        self.eps_mode, self.tie_mode = eps_mode, tie_mode

    def __call__(self, batch):
        batch = np.asarray(batch)
        out = self.net.forward_masked(self.ctx.to_device(batch, np.float32), self.eps_mode, self.tie_mode)
        out = self.ctx.to_host(out).astype(np.float64)
        return [out[s][:, None] for s in range(out.shape[0])]


def read_wav(filein):
    """wav -> float64 in [-1, 1] the way every script does it (separate_dsd.py:275-282)."""
    sampleRate, audioObj = scipy.io.wavfile.read(filein)
    try:
        maxv = np.finfo(audioObj.dtype).max
    except Exception:
        maxv = np.iinfo(audioObj.dtype).max
    return sampleRate, audioObj.astype('float') / maxv


def write_wav(path, audio_out, sampleRate):
    """``(audio * 32767).astype('int16')`` -- truncation, no clipping (separate_dsd.py:307-309)."""
    maxn = np.iinfo(np.int16).max
    scipy.io.wavfile.write(filename=path, rate=sampleRate, data=(audio_out * maxn).astype('int16'))


def to_mono(audioObj, arch_name):
    """DSD / Bach10: mean of L and R when stereo, mono passes through (separate_dsd.py:285-287).
    iKala: L + R, no halving, and a mono file is an IndexError (separate_ikala.py:229)."""
    if arch_name == 'ikala':
        return audioObj[:, 0] + audioObj[:, 1]
    if len(audioObj.shape) > 1 and audioObj.shape[1] > 1:
        return (audioObj[:, 0] + audioObj[:, 1]) / 2
    if len(audioObj.shape) > 1:
        return audioObj[:, 0]
    return audioObj


def length_groups(sizes, max_group=16, max_ratio=1.5):
    """Indices of ``sizes`` sorted by size and cut into runs of at most ``max_group`` entries whose largest is at most
    ``max_ratio`` times the smallest (a launch group costs what its longest clip costs, times its size).  Empty
    clips stay alone (the single-clip path raises for them what the reference raises)."""
    order = sorted(range(len(sizes)), key=lambda i: (sizes[i], i))
    groups, cur = [], []
    for i in order:
        if cur and (len(cur) >= max_group or sizes[cur[0]] == 0 or sizes[i] > max_ratio * sizes[cur[0]]):
            groups.append(cur)
            cur = []
        cur.append(i)
    if cur:
        groups.append(cur)
    return groups


class Pcm16Pending(object):
    """What ``Separator.separate_many_pcm16(..., wait=False)`` returns: the enqueued work of one call."""

    def __init__(self, event, out, views, keep):
        self._event, self._out, self._views, self._keep = event, out, views, keep

    def done(self):
        return self._event is None or self._event.query()

    def result(self):
        if self._event is not None:
            self._event.synchronize()
            self._event = None
            for i, v in self._views:
                self._out[i] = v.numpy()
            self._views, self._keep = [], None
        return self._out


class Separator(object):
    """Model + STFT plan resident on one GPU; ``separate(audio)`` is the body of ``train_auto``
    between reading and writing the wav files (separate_dsd.py:289-306)."""

    def __init__(self, arch, params, scale_factor=0.3, time_context=30, overlap=25, batch_size=32, input_size=513,
                 frameSize=1024, hopSize=512, window=np.hanning, tiler='script', tie_mode=TIE_ALL, device=None,
                 ctx=None, score_normalise='max', score_mixture='ch0'):
        self.arch_name = arch
        self.arch = ARCHS[arch]
        self.scale_factor, self.tc, self.overlap, self.batch_size = scale_factor, time_context, overlap, batch_size
        self.input_size, self.frameSize, self.hopSize = input_size, frameSize, hopSize
        self.tiler, self.tie_mode = _TILERS[tiler], tie_mode
        if frameSize // 2 + 1 != input_size:
            raise ValueError("frameSize %d gives %d bins but the network takes %d" % (frameSize, frameSize // 2 + 1,
                                                                                      input_size))
        self.ctx = ctx if ctx is not None else default_context(device)   # ctx: one per HIP stream
        self.window = window(frameSize)
        self.plan = StftPlan(self.ctx, frameSize, hopSize, self.window)
        self.net = Network(self.ctx, arch, params, time_context, input_size)
        # score-informed graphs: the separate script's semantics by default, the trainers' on request:
        # score_normalise 'max' | 'sum' (separate_bach10.py:195 | dataset.py:862), score_mixture 'ch0' | 'sum'
        # (separate_bach10.py:485 | trainCNNrwc.py:258-263)
        self.score_normalise, self.score_mixture = score_normalise, score_mixture
        if (score_normalise, score_mixture) != ('max', 'ch0'):
            self.net.set_score_semantics(score_normalise, score_mixture)

    @_on_ctx_stream
    def separate(self, audio):
        """Fused device path: float audio ``[L]`` -> float64 ``[S, L]``."""
        a = self.ctx.to_device(np.asarray(audio), np.float32)
        pcm = self.net.separate(self.plan, a, self.overlap, self.tiler, self.scale_factor, None, self.tie_mode)
        return self.ctx.to_host(pcm).astype(np.float64)

    def separate_many(self, audios, max_group=16, max_ratio=None, on_error='raise'):
        """A list of clips -> a list of float64 ``[S, L_i]``.  Clips share sets of kernel launches (DSD / hiphop graph):
        sorted by length, they are cut into groups of at most ``max_group`` clips whose longest is at most
        ``max_ratio`` times the shortest (default 3 for the DSD graph, whose ragged launches cost the SUM of their clips since
        round 4 -- per-clip row and tile offsets; 1.5 for the other graphs, where a group still costs what its longest clip
        costs, times its size); a group goes
        through ``dcs_separate_ragged`` (``dcs_separate_batch`` when its lengths are equal).  Every clip gets exactly
        the frames, the tiles and the cross-fade :meth:`separate` gives it alone.  Frame sizes the wave STFT kernels do
        not cover share launches between clips of equal length only (``dcs_separate_batch``); groups of one go through
        :meth:`separate`.  Clips that cannot be separated
        (empty, or too short for one tile) never join a shared launch; ``on_error='return'`` puts the exception the
        single-clip path raises for them into their slot of the result instead of raising it."""
        audios = [np.asarray(a) for a in audios]
        if max_ratio is None:
            max_ratio = 3.0 if self.arch_name in ("dsd", "hiphop") else 1.5
        for i, a in enumerate(audios):
            if a.ndim != 1:
                # [L, 1] / [1, L] mono arrays are flattened; reshape(-1) of a real [L, 2] array would interleave its
                # channels: mix down first (to_mono)
                if a.size == max(a.shape + (0,)):
                    audios[i] = a.reshape(-1)
                else:
                    raise ValueError("separate_many takes mono clips ([L] arrays), got shape %r" % (a.shape,))
        out = [None] * len(audios)
        # clips of different lengths share launches on every single-channel graph (dcs_separate_ragged: DSD fused path;
        # ikala / bach10: every clip framed, tiled and cross-faded with its own counts, one pass of all tiles through the
        # network) when the wave STFT kernels cover the frame size
        ragged_ok = (self.arch.C == 1 and self.arch_name != "dsd_ild" and self.frameSize in (1024, 2048, 4096)
                     and self.frameSize % self.hopSize == 0 and self.hopSize % 2 == 0)
        # ikala / bach10: clips of EQUAL length share the launches (their tiles go through the network as one batch)
        equal_ok = self.arch.C == 1 and self.arch_name != "dsd_ild"
        shared = ragged_ok or equal_ok
        if not ragged_ok:
            max_ratio = 1.0
        # a clip too short to yield a tile makes the shared launch fail as a whole: such clips go through the single-clip
        # path, where the failure stays confined to that clip (the reference runs one process per file)
        def n_tiles(a):
            return _lib.tile_count(_lib.frame_count(a.size, self.hopSize), self.tc, self.overlap, self.tiler)
        alone = set(i for i, a in enumerate(audios) if a.size == 0 or n_tiles(a) < 1)
        sizes = [0 if i in alone else a.size for i, a in enumerate(audios)]
        groups = length_groups(sizes, max_group if shared else 1, max_ratio)
        for idx in groups:
            if len(idx) == 1:
                i = idx[0]
                if on_error == 'raise' or i not in alone:
                    out[i] = self.separate(audios[i])
                else:
                    try:
                        out[i] = self.separate(audios[i])
                    except (ValueError, IndexError) as exc:
                        out[i] = exc
                continue
            lens = [int(audios[i].size) for i in idx]
            stack = np.zeros((len(idx), max(lens)), dtype=np.float32)
            for b, i in enumerate(idx):
                stack[b, :lens[b]] = audios[i]
            try:
                dev = self.ctx.to_device(stack, np.float32)                                          # [B, Lmax]
                if min(lens) == max(lens):
                    pcm = self.net.separate_batch(self.plan, dev, self.overlap, self.tiler, self.scale_factor, None,
                                                  self.tie_mode)
                else:
                    pcm = self.net.separate_ragged(self.plan, dev, lens, self.overlap, self.tiler, self.scale_factor,
                                                   None, self.tie_mode)
                pcm = self.ctx.to_host(pcm)
            except Exception:
                if on_error == 'raise':
                    raise
                # a failure of the shared launch (e.g. out of memory on a large group) must not discard the clips
                # already separated: its members go through the single-clip path, each with its own outcome
                for i in idx:
                    try:
                        out[i] = self.separate(audios[i])
                    except Exception as exc:
                        out[i] = exc
                continue
            for b, i in enumerate(idx):
                out[i] = pcm[b, :, :lens[b]].astype(np.float64)
        return out

    @_on_ctx_stream
    def separate_many_pcm16(self, clips, max_group=16, max_ratio=None, on_error='raise', ring=None, wait=True):
        """The batch-of-files path without host arithmetic: ``clips`` are the int16 frames of 16-bit PCM wav files exactly as
        ``scipy.io.wavfile.read`` returns them (``[L]`` mono or ``[L, channels]``; NumPy arrays or pinned torch CPU tensors, e.g.
        from :func:`deepconvsep_amd.wavio.read_pcm16`), the result a list of int16 arrays ``[S, L_i]`` -- the samples the
        scripts write (``(audio_out * 32767).astype('int16')``, separate_dsd.py:307-309).  The division by 32767, the mix-down
        (separate_dsd.py:285-287; iKala: L + R, separate_ikala.py:229) and the int16 conversion run on the device
        (``dcs_pcm16_to_float`` / ``dcs_pcm_to_int16``, float64 arithmetic like the scripts'): the values are those of
        ``separate_many(to_mono(read_wav(f)))`` + ``write_wav`` bit for bit, for a quarter of the PCIe bytes.  Grouping as in
        :meth:`separate_many`.  ``ring=0`` returns fresh arrays (the default with ``wait=True``: what :meth:`separate_many`
        callers expect); ``ring=k`` > 0 returns views of a ring of ``k`` pinned host buffers that stay valid until ``k - 1``
        further calls have been made (the default, 3, with ``wait=False`` -- the batch driver's writer threads stream the
        samples to the files straight from pinned memory).  A different ``ring`` on a later call re-sizes the ring.

        ``wait=False`` returns as soon as the copies and kernels are enqueued: the result is a :class:`Pcm16Pending` whose
        ``result()`` waits for the device and hands out the list.  The caller's input buffers must stay untouched until then
        (they are the source of an asynchronous host-to-device copy); a driver that reads group i + 1 and writes group i - 1
        while the device separates group i needs nothing else (``examples/separate_batch.py``)."""
        import torch
        from .runtime import _torch as _device_torch
        dtorch = _device_torch()            # device allocations (the memory-safety harness routes them through guarded arenas)
        if max_ratio is None:
            max_ratio = 3.0 if self.arch_name in ("dsd", "hiphop") else 1.5
        mode = 1 if self.arch_name in ("ikala", "ikala_nopool") else 0
        S = self.net.S
        tens = []
        for c in clips:
            t = c if isinstance(c, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(c))
            if t.dtype != torch.int16 or t.dim() not in (1, 2):
                raise ValueError("separate_many_pcm16 takes int16 frames [L] or [L, channels]")
            tens.append(t.contiguous())
        chans = [1 if t.dim() == 1 else int(t.shape[1]) for t in tens]
        frames = [int(t.shape[0]) for t in tens]
        out = [None] * len(tens)
        ragged_ok = (self.arch.C == 1 and self.frameSize in (1024, 2048, 4096) and self.frameSize % self.hopSize == 0
                     and self.hopSize % 2 == 0)
        if not ragged_ok:
            max_ratio = 1.0

        def n_tiles(L):
            return _lib.tile_count(_lib.frame_count(L, self.hopSize), self.tc, self.overlap, self.tiler)
        bad = {}
        for i in range(len(tens)):
            if mode == 1 and chans[i] < 2:
                bad[i] = IndexError("too many indices for array")            # separate_ikala.py:229 on a mono file
            elif frames[i] == 0 or n_tiles(frames[i]) < 1:
                bad[i] = ValueError("dcs_separate: %d frames give no tile (the reference fails in overlapadd_multi)"
                                    % _lib.frame_count(frames[i], self.hopSize))
        if bad and on_error == 'raise':
            raise bad[min(bad)]
        for i, exc in bad.items():
            out[i] = exc
        # groups by length alone, exactly as separate_many forms them (the channel count only matters for the mix-down, which
        # is done clip by clip): the same clips share the same launches on either path, so the results agree bit for bit
        idx_ok = [i for i in range(len(tens)) if i not in bad]
        groups = [[idx_ok[k] for k in g]
                  for g in length_groups([frames[i] for i in idx_ok], max_group if self.arch.C == 1 else 1, max_ratio)]
        if not groups:
            return out if wait else Pcm16Pending(None, out, [], None)
        # one pinned output block per call, cut out of a small ring: the writer threads of the caller stream the int16
        # samples to the files straight from it
        total = sum(S * frames[i] for g in groups for i in g)
        if ring is None:
            ring = 0 if wait else 3
        if ring:
            if len(getattr(self, "_pcm16_ring", ())) != int(ring):       # first call, or the caller changed the ring length
                self._pcm16_ring, self._pcm16_next = [None] * int(ring), 0
            slot = self._pcm16_next % len(self._pcm16_ring)
            self._pcm16_next += 1
            if self._pcm16_ring[slot] is None or self._pcm16_ring[slot].numel() < total:
                self._pcm16_ring[slot] = torch.empty((max(total, 1 << 20) * 9 // 8,), dtype=torch.int16, pin_memory=True)
            host = self._pcm16_ring[slot]
        else:
            host = torch.empty((total,), dtype=torch.int16)
        pos = 0
        views = []
        dev = self.ctx.device
        # clips cut out of ONE pinned staging block (the batch driver's read arena) travel as one host-to-device copy
        arena_dev, arena_lo = None, 0
        live = [i for g in groups for i in g]
        whole = None
        try:                                            # storage introspection only: anything unusual means "no arena"
            bases = set(tens[i].untyped_storage().data_ptr() for i in live)
            if len(bases) == 1 and len(live) > 1 and all(tens[i].is_pinned() for i in live):
                lo = min(tens[i].data_ptr() for i in live)
                hi = max(tens[i].data_ptr() + tens[i].numel() * 2 for i in live)
                if hi - lo <= 2 * sum(tens[i].numel() * 2 for i in live) + 4096:       # the block is (nearly) all payload
                    st0 = tens[live[0]].untyped_storage()
                    whole = torch.empty(0, dtype=torch.uint8).set_(st0, lo - st0.data_ptr(), (hi - lo,))
                    arena_lo = lo
        except (RuntimeError, AttributeError, TypeError):
            whole = None
        if whole is not None:                           # allocation / copy errors (out of memory ...) are the caller's to see
            arena_dev = dtorch.empty((whole.numel(),), dtype=torch.uint8, device=dev)
            arena_dev.copy_(whole, non_blocking=True)
        for idx in groups:
            lens = [frames[i] for i in idx]
            B, Lmax = len(idx), max(lens)
            mono = dtorch.zeros((B, Lmax), dtype=torch.float32, device=dev) if min(lens) != Lmax else \
                dtorch.empty((B, Lmax), dtype=torch.float32, device=dev)
            for b, i in enumerate(idx):
                nel = lens[b] * chans[i]
                if arena_dev is not None:
                    off = tens[i].data_ptr() - arena_lo
                    raw = arena_dev[off:off + 2 * nel].view(torch.int16).view(1, nel)
                else:
                    raw = dtorch.empty((1, nel), dtype=torch.int16, device=dev)
                    raw[0].copy_(tens[i].reshape(-1), non_blocking=True)
                pcm16_to_float(self.ctx, raw, chans[i], mode, out=mono[b:b + 1, :lens[b]])
            if B == 1:
                pcm = self.net.separate(self.plan, mono[0], self.overlap, self.tiler, self.scale_factor, None, self.tie_mode)[None]
            elif min(lens) == Lmax:
                pcm = self.net.separate_batch(self.plan, mono, self.overlap, self.tiler, self.scale_factor, None, self.tie_mode)
            else:
                pcm = self.net.separate_ragged(self.plan, mono, lens, self.overlap, self.tiler, self.scale_factor, None,
                                               self.tie_mode)
            p16 = pcm_to_int16(self.ctx, pcm)                                                       # [B, S, Lmax] int16
            if min(lens) == Lmax:                      # equal lengths: the group's samples leave as one copy
                blk = host[pos:pos + B * S * Lmax].view(B, S, Lmax)
                blk.copy_(p16, non_blocking=True)
                for b, i in enumerate(idx):
                    views.append((i, blk[b]))
                pos += B * S * Lmax
            else:
                for b, i in enumerate(idx):
                    n = S * lens[b]
                    dst = host[pos:pos + n].view(S, lens[b])
                    dst.copy_(p16[b, :, :lens[b]], non_blocking=True)
                    views.append((i, dst))
                    pos += n
        if not wait:
            done = torch.cuda.Event()
            done.record(self.ctx.torch_stream)
            return Pcm16Pending(done, out, views, (tens, host))
        self.ctx.torch_stream.synchronize()
        for i, v in views:
            out[i] = v.numpy()
        return out

    @_on_ctx_stream
    def separate_stepwise(self, audio):
        """The reference's control flow, stage by stage through the public operators
        (compute_file -> x scale -> generate_overlapadd -> predict_function2 per batch ->
        overlapadd_multi -> compute_inverse); float32 on the device like ``separate``."""
        import torch
        a = self.ctx.to_device(np.asarray(audio), np.float32)
        mag, ph = self.plan.forward(a, phase=True)
        tiles, n = tile(self.ctx, mag, self.tc, self.overlap, self.tiler, self.scale_factor)
        if n == 0:
            raise IndexError("tuple index out of range")  # what overlapadd_multi hits on an empty output array
        outs = []
        for b0 in range(0, n, self.batch_size):
            outs.append(self.net.forward_masked(tiles[b0:b0 + self.batch_size], None, self.tie_mode))
        out = torch.cat(outs, dim=1)
        mm = overlap_add(self.ctx, out, self.overlap)
        T = mag.shape[0]
        sep = mm[:, :T].contiguous()
        pcm = self.plan.inverse(sep, ph, n_out=int(a.numel()), pre_div=self.scale_factor)
        return self.ctx.to_host(pcm).astype(np.float64)


    @_on_ctx_stream
    def separate_stereo(self, audio):
        """The "Separating" block of the stereo trainer (examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py:291-325):
        ``audio [L, 2]`` -> ``sep_audio [L, S, 2]`` (one stereo signal per source, :299,316)."""
        a = self.ctx.to_device(np.ascontiguousarray(np.asarray(audio).T), np.float32)          # [2, L]
        pcm = self.net.separate_stereo(self.plan, a, self.overlap, TILER_LIBRARY, self.scale_factor)
        return np.ascontiguousarray(self.ctx.to_host(pcm).astype(np.float64).transpose(2, 1, 0))   # [L, S, 2]

    @_on_ctx_stream
    def separate_scoreinformed(self, audio, melody, timbre_model_path=None):
        """Score-informed separation (examples/bach10_scoreinformed/separate_bach10.py:497-541), stage by stage on the
        device: STFT -> x scale -> harmonic masks of the score x spectrogram (``dcs_score_masks``) -> library tiler
        with one input channel per instrument (``util.generate_overlapadd``, :531) -> network (masks from the first
        ``S`` output channels, mixture = input channel 0, :473-486) -> overlapadd_multi -> iSTFT.
        ``melody``: note tables ``[instruments, notes, 2*nharmonics+3]`` (``score.melody_table``).
        ``timbre_model_path``: the pickled harmonic templates of ``filterSpec(..., timbre_model_path)`` -- the masks are then
        the template-weighted ones (host, ``score.timbre_masks``) and the path runs stage by stage."""
        a = self.ctx.to_device(np.asarray(audio), np.float32)
        if timbre_model_path is not None:
            from .score import load_timbre_model
            pcm = self._separate_scoreinformed_staged(a, melody, load_timbre_model(timbre_model_path))
            return self.ctx.to_host(pcm).astype(np.float64)
        return self.ctx.to_host(self.separate_scoreinformed_device(a, melody)).astype(np.float64)

    def separate_scoreinformed_device(self, a, melody, staged=False):
        """:meth:`separate_scoreinformed` on a float32 device tensor ``[L]``; returns the float32 PCM ``[S, L]`` on the
        device.  One ``dcs_separate_scoreinformed`` call (all tiles in one pass of the network; the script's
        ``batch_size`` loop gives the same values tile by tile).  ``staged=True``: the same path composed from the
        stage-level operators, ``batch_size`` tiles at a time (what the reference's control flow looks like when each
        helper is swapped for its ``dcs_*`` counterpart)."""
        if self.arch.C != np.asarray(melody).shape[0]:
            raise ValueError("the network takes %d score channels, the note table has %d"
                             % (self.arch.C, np.asarray(melody).shape[0]))
        if staged:
            return self._separate_scoreinformed_staged(a, melody)
        return self.net.separate_scoreinformed(self.plan, a, melody, self.overlap, self.scale_factor,
                                               tie_mode=self.tie_mode)

    @_on_ctx_stream
    def _separate_scoreinformed_staged(self, a, melody, harmonics=None):
        import torch
        from .score import score_masks, timbre_masks
        mag, ph = self.plan.forward(a, phase=True)
        T = int(mag.shape[0])
        mag = mag * np.float32(self.scale_factor)                       # :503
        if harmonics is None:
            inp, _ = score_masks(self.ctx, mag, melody, 0, T, normalise=self.score_normalise)   # :520-527, [C, T, F]
        else:
            # timbre-model masks (host) x spectrogram, float32 products as the script forms them (:523-527)
            mag_h = self.ctx.to_host(mag)
            C = int(np.asarray(melody).shape[0])
            F = int(mag_h.shape[1])
            m = timbre_masks(mag_h.shape, melody, C, 0, T, harmonics, normalise=self.score_normalise)
            inp = self.ctx.to_device(np.stack([m[:, j * F:(j + 1) * F] * mag_h for j in range(C)]), np.float32)
        tiles, n = tile(self.ctx, inp, self.tc, self.overlap, TILER_LIBRARY, 1.0)
        outs = []
        for b0 in range(0, n, self.batch_size):
            outs.append(self.net.forward_masked(tiles[b0:b0 + self.batch_size], None, self.tie_mode))
        out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
        mm = overlap_add(self.ctx, out, self.overlap)
        sep = mm[:, :T].contiguous()
        return self.plan.inverse(sep, ph, n_out=int(a.numel()), pre_div=self.scale_factor)


_SCRIPT_DEFAULTS = {
    # arch: (frameSize, hopSize, window, overlap in main(), input_size)     separate_<x>.py main()
    'dsd': (1024, 512, np.hanning, 25, 513),       # separate_dsd.py:24,332
    'hiphop': (1024, 512, np.hanning, 25, 513),    # separate_hhds.py
    'ikala': (1024, 512, np.hanning, 20, 513),     # separate_ikala.py:24,275
    'bach10': (4096, 512, blackmanharris, 25, 2049),  # separate_bach10.py:282,325
    'bach10_si': (4096, 512, blackmanharris, 25, 2049),  # bach10_scoreinformed/separate_bach10.py:450,572
}

# score files next to the wav, one per instrument (bach10_scoreinformed/separate_bach10.py:455)
SI_SCORE_FILES = ['bassoon_b.txt', 'clarinet_b.txt', 'saxophone_b.txt', 'violin_b.txt']
SI_SCORE_PARAMS = dict(interval=50, tuning_freq=440, nharmonics=20)     # :457-460


def output_paths(arch_name, filein, outdir):
    """Output file names of each script (separate_dsd.py:309, separate_ikala.py:253-254,
    separate_bach10.py:302)."""
    path, filename = os.path.split(filein)
    if arch_name in ('dsd', 'hiphop'):
        return [os.path.join(outdir, s + ".wav") for s in ARCHS['dsd'].source_names]
    if arch_name == 'ikala':
        return [os.path.join(outdir, filename.replace(".wav", "-voice.wav")),
                os.path.join(outdir, filename.replace(".wav", "-music.wav"))]
    return [os.path.join(outdir, filename.replace(".wav", "_" + s + ".wav")) for s in ARCHS[arch_name].source_names]


def train_auto(arch_name, filein, outdir, model, scale_factor=0.3, time_context=30, overlap=20, batch_size=32,
               input_size=513, frameSize=None, hopSize=None, window=None, fused=True, device=None, score_normalise='max',
               score_mixture='ch0'):
    """``train_auto`` of the separate scripts: wav in, one wav per source out.  ``score_normalise`` / ``score_mixture``
    (score-informed only): ``('sum', 'sum')`` = what the trainer's own separation block computes
    (bach10_scoreinformed/trainCNNrwc.py:360-416), for models that trainer wrote; the default is the separate script's."""
    d_frame, d_hop, d_win, _, _ = _SCRIPT_DEFAULTS[arch_name]
    frameSize = d_frame if frameSize is None else frameSize
    hopSize = d_hop if hopSize is None else hopSize
    window = d_win if window is None else window
    params = load_model(model) if isinstance(model, str) else model
    sep = Separator(arch_name, params, scale_factor, time_context, overlap, batch_size, input_size, frameSize,
                    hopSize, window, tiler='library' if arch_name == 'bach10_si' else 'script', device=device,
                    score_normalise=score_normalise, score_mixture=score_mixture)
    sampleRate, audioObj = read_wav(filein)
    if sampleRate == 44100:
        audio = to_mono(audioObj, arch_name)
        if arch_name == 'bach10_si':
            from .score import melody_table
            nframes = int(np.ceil(len(audio) / np.double(hopSize))) + 2     # :500
            melody = melody_table(SI_SCORE_FILES, os.path.dirname(filein), nframes, sampleRate, hopSize, frameSize,
                                  **SI_SCORE_PARAMS)
            pcm = sep.separate_scoreinformed(audio, melody)
        else:
            pcm = sep.separate(audio) if fused else sep.separate_stepwise(audio)
        for path, audio_out in zip(output_paths(arch_name, filein, outdir), pcm):
            write_wav(path, audio_out, sampleRate)
    else:
        print("Sample rate is not 44100")
