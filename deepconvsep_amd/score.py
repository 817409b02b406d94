"""Score-informed front-end: score text -> note table -> harmonic masks -> network input.

Mirrors the reference's surface for this step:

  expandMidi(instrument, FilePath, beginTime, finishTime, interval, tuning_freq, nharmonics, samplerate, hop, window,
             timeSpan_on, timeSpan_off, nframes, fermata=0.)     util.py:424-512
  getMidiNum(instrument, FilePath, beginTime, finishTime)         util.py:526-579
  filterSpec(mag, notes, ninst, start, stop)                      examples/bach10_scoreinformed/separate_bach10.py:172-200
  str2midi, midi2freq                                             util.py:581-606, 126-127

The note table is a few hundred integers computed on the host (vectorised NumPy; the Python-2 integer arithmetic of
the reference -- ``samplerate / hop`` floors, ``round`` rounds halves away from zero -- is reproduced explicitly).
Painting the masks and multiplying them into the spectrogram runs on the GPU (``dcs_score_masks``, csrc/score.hip);
there is no CPU path for it.
"""
import os
from ctypes import POINTER, c_double

import numpy as np

from . import _lib

_NAME = {"c": -9, "d": -7, "e": -5, "f": -4, "g": -2, "a": 0, "b": 2}
_ACC = {"b": -1, "#": 1, "x": 2}


def str2midi(note_string):
    """"Bb4" -> 70 (util.py:581-606).  Accidentals directly after the letter come before the octave digits;
    otherwise the second character is a one-digit octave and accidentals may follow it."""
    if note_string == "?":
        return float("nan")
    s = note_string.strip().lower()
    tail = s[1:] if s[1] in _ACC else s[2:]
    k = 0
    while k < len(tail) and tail[k] in _ACC:
        k += 1
    octave = int(s[1 + k:]) if s[1] in _ACC else int(s[1])
    return 69 + _NAME[s[0]] + sum(_ACC[c] for c in tail[:k]) + 12 * (octave - 4)


def midi2freq(midi_number, tuning_freq=440., MIDI_A4=69.):
    return float(tuning_freq) * 2.0 ** ((float(midi_number) - float(MIDI_A4)) * (1. / 12.))


def _half_away(x):
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def harmonic_bins(midi, size, interval=30, tuning_freq=440, nharmonics=20, sampleRate=44100):
    """FFT bin ranges [first, last+1) of the harmonics 1..nharmonics-1 of a note, +-interval cents wide, overlapping
    ranges merged the way util.remove_overlap does (a merged run ends at the stop of its LAST member), ranges reaching
    past the Nyquist bin dropped (util.py:171-191, 140-157)."""
    if not midi > 0:
        return np.zeros((0, 2), dtype=np.int64)
    k = np.arange(1, nharmonics, dtype=np.float64)
    f0 = midi2freq(midi, tuning_freq=tuning_freq)
    factor = 2.0 ** (interval / 1200.0)
    binfactor = float(size) / float(sampleRate)
    lo = 1 + np.floor(k * f0 / float(factor) * binfactor).astype(np.int64)
    hi = 1 + np.ceil(k * f0 * float(factor) * binfactor).astype(np.int64)
    order = np.lexsort((hi, lo))
    lo, hi = lo[order], hi[order]
    merged = []
    reach = -1                                   # running maximum of the stops seen in the current run
    for a, b in zip(lo.tolist(), hi.tolist()):
        if a > reach:
            merged.append([a, b])
            reach = b
        else:
            merged[-1][1] = b
            reach = max(reach, b)
    out = np.array([m for m in merged if m[1] <= size // 2 + 1], dtype=np.int64).reshape(-1, 2)
    return out


def read_score(path):
    """Lines ``onset,offset,name``; ``!`` starts a comment.  Times are float32 values like the reference's
    ``np.genfromtxt(dtype=["f","f","S3"])``; names keep their first three characters."""
    on, off, names = [], [], []
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.split("!")[0].strip(" \r\n")
            if line:
                a, b, c = line.split(",")[:3]
                on.append(np.float32(a))
                off.append(np.float32(b))
                names.append(c[:3])
    return np.asarray(on, dtype=np.float32).astype(np.float64), np.asarray(off, dtype=np.float32).astype(np.float64), names


def _score_path(instrument, FilePath):
    p = os.path.join(FilePath, instrument)
    return p if os.path.isfile(p) or instrument.endswith(".txt") else p + ".txt"


def _select(on, off, beginTime, finishTime):
    """Index window of the notes that touch [beginTime, finishTime] (util.py:433-441), then times relative to
    beginTime clamped to [0, finishTime - beginTime] (:444-458).  None when fewer than two notes remain."""
    first = int(np.searchsorted(off, beginTime, side="right"))
    last = int(np.searchsorted(on, finishTime, side="left"))
    if off[first] < float(beginTime):           # IndexError past the end, like the reference
        first += 1
    if last >= len(on):
        last = len(on) - 1
    elif on[last] > float(finishTime):
        last -= 1
    if not first < last:
        return None
    span = finishTime - beginTime
    b = np.clip(on[first:last + 1] - beginTime, 0.0, span)
    e = np.clip(off[first:last + 1] - beginTime, 0.0, span)
    return first, last, b, e


def getMidiNum(instrument, FilePath, beginTime, finishTime):
    on, off, _ = read_score(_score_path(instrument, FilePath))
    sel = _select(on, off, beginTime, finishTime)
    if sel is None:
        return 1
    _, _, b, e = sel
    return int(np.count_nonzero(~((e <= 0) | (e <= b) | ((e - b) < 0.01))))


def expandMidi(instrument, FilePath, beginTime, finishTime, interval, tuning_freq, nharmonics, samplerate, hop, window,
               timeSpan_on, timeSpan_off, nframes, fermata=0.):
    """Note table ``[notes, 2*nharmonics+3]`` (float64, integer valued): first frame, one-past-last frame, MIDI number,
    then the harmonic bin ranges.  A note starts ``timeSpan_on`` s early; it ends ``timeSpan_off`` s late when another
    note begins inside its extended span, otherwise it rings for ``max(timeSpan_off, fermata)`` s but not into the
    attack region of the next note (util.py:476-495)."""
    fermata = max(timeSpan_off, fermata)
    on, off, names = read_score(_score_path(instrument, FilePath))
    sel = _select(on, off, beginTime, finishTime)
    if sel is None:
        return None
    first, last, b, e = sel
    names = names[first:last + 1]
    tframes = float(nframes) * float(hop) / float(samplerate)
    keep = ~((e <= 0) | (e <= b) | (b >= tframes) | ((e - b) < 0.01))
    b, e = b[keep], e[keep]
    names = [n for n, k in zip(names, keep.tolist()) if k]
    per_s = samplerate // hop if isinstance(samplerate, int) and isinstance(hop, int) else samplerate / hop
    fps = float(_half_away(float(per_s)))
    on_frames = int(_half_away(timeSpan_on * float(per_s)))
    off_frames = int(_half_away(timeSpan_off * float(per_s)))
    end_melody = int((finishTime - beginTime) * fps)
    n = len(b)
    table = np.zeros((n, 2 * nharmonics + 3))
    if n == 0:
        return table
    table[:, 0] = np.maximum(0, (b * fps).astype(np.int64) - on_frames)
    # pairwise: does note q begin after note i and inside i's extended span?
    bq, eq = b[None, :], e[None, :]
    bi, ei = b[:, None], e[:, None]
    crowded = ((bq > bi) & ((eq + timeSpan_off) >= (bi - timeSpan_on)) & ((bq - timeSpan_on) <= (ei + timeSpan_off))).any(axis=1)
    later = np.where((bq - timeSpan_on) > (ei + timeSpan_off), bq, np.inf).min(axis=1)   # earliest clearly-later onset
    ring = np.where(np.isfinite(later), np.minimum(e + fermata, np.maximum(0, later - timeSpan_on)), e + fermata)
    end_free = (ring * fps).astype(np.int64)
    end_crowded = (e * fps).astype(np.int64) + off_frames
    table[:, 1] = np.minimum(nframes, np.minimum(end_melody, np.where(crowded, end_crowded, end_free)))
    for m, name in enumerate(names):
        midi = str2midi(name)
        table[m, 2] = midi
        bins = harmonic_bins(midi, size=window, interval=interval, tuning_freq=tuning_freq, nharmonics=nharmonics,
                             sampleRate=samplerate)
        table[m, 3:3 + 2 * len(bins):2] = bins[:, 0]
        table[m, 4:4 + 2 * len(bins):2] = bins[:, 1]
    return table


def melody_table(instruments, FilePath, nframes, samplerate=44100, hop=512, window=4096, interval=50, tuning_freq=440,
                 nharmonics=20, beginTime=0, finishTime=40.0, timeSpan_on=0.2, timeSpan_off=0.2, fermata=0.5):
    """The script's note tables stacked to ``[ninst, max notes, 2*nharmonics+3]``
    (examples/bach10_scoreinformed/separate_bach10.py:508-518)."""
    nelem = 1
    for ins in instruments:
        nelem = max(getMidiNum(ins, FilePath, beginTime, finishTime), nelem)
    melody = np.zeros((len(instruments), int(nelem), 2 * nharmonics + 3))
    for i, ins in enumerate(instruments):
        tmp = expandMidi(ins, FilePath, beginTime, finishTime, interval, tuning_freq, nharmonics, samplerate, hop, window,
                         timeSpan_on, timeSpan_off, nframes, fermata)
        if tmp is None:
            raise AttributeError("'NoneType' object has no attribute 'shape'")   # what the script hits (:517)
        melody[i, :tmp.shape[0], :] = tmp
    return melody


_NORMALISE = {'max': 0, 'sum': 1}


def score_masks(ctx, mag_t, notes, start, stop, want_input=True, want_mask=False, normalise='max'):
    """Device ``filterSpec`` and the products of separate_bach10.py:523-527.  mag_t ``[T, F]`` float32 device tensor
    (already scaled); notes ``[ninst, P, W]``.  Returns (input ``[ninst, T, F]`` or None, mask ``[T, ninst*F]`` or
    None), float32 device tensors.  ``normalise``: ``'max'`` -- every instrument's field divided by its own maximum (the
    script, separate_bach10.py:195); ``'sum'`` -- bin by bin by the sum over the instruments (the trainers' dataset class,
    ``LargeDatasetMask2.filterSpec``, dataset.py:839-879)."""
    if normalise not in _NORMALISE:
        raise ValueError("normalise must be 'max' or 'sum'")
    from .runtime import _ptr, _torch
    torch = _torch()
    if mag_t.dim() != 2 or mag_t.stride(1) != 1:
        raise ValueError("score_masks expects a [frames, bins] tensor with contiguous rows")
    notes = np.ascontiguousarray(notes, dtype=np.float64)
    if notes.ndim != 3:
        raise ValueError("notes must be [instruments, notes, 2*nharmonics+3]")
    ninst, P, W = notes.shape
    T, F = int(mag_t.shape[0]), int(mag_t.shape[1])
    with ctx.stream_scope():
        inp = torch.empty((ninst, T, F), dtype=torch.float32, device=mag_t.device) if want_input else None
        mask = torch.empty((T, ninst * F), dtype=torch.float32, device=mag_t.device) if want_mask else None
        _lib.check(ctx._lib.dcs_score_masks_norm(ctx._h, _ptr(mag_t), int(mag_t.stride(0)), T, F,
                                                 notes.ctypes.data_as(POINTER(c_double)), ninst, P, W, int(start), int(stop),
                                                 _NORMALISE[normalise],
                                                 _ptr(inp) if inp is not None else None,
                                                 _ptr(mask) if mask is not None else None))
    return inp, mask


def load_timbre_model(path):
    """The pickled harmonic templates ``filterSpec`` takes (``harmonics[instrument, midi pitch, harmonic]``,
    examples/bach10_scoreinformed/separate_bach10.py:173-175); Python-2 pickles load with ``latin1``."""
    import pickle
    with open(path, 'rb') as fh:
        try:
            h = pickle.load(fh)
        except UnicodeDecodeError:
            fh.seek(0)
            h = pickle.load(fh, encoding='latin1')
    h = np.asarray(h)
    if h.ndim != 3:
        raise ValueError("timbre model must be [instruments, pitches, harmonics], got shape %r" % (h.shape,))
    return h


def timbre_masks(shape, notes, ninst, start, stop, harmonics, normalise='max'):
    """The timbre-model branch of ``filterSpec`` (separate_bach10.py:189-191,195-196) on the host: a float32 field per
    instrument that starts at 1e-18, to which every harmonic band of every note sounding in ``[start, stop)`` ADDS the
    template weight ``harmonics[j, pitch, k]`` (float32 adds, notes in table order, harmonics in slot order: the sums are
    order-dependent in the last bit, so the order is the reference's), divided by its own maximum.  Bands of one note are
    not merged here -- overlapping bands add twice -- and the zero-width slots past the Nyquist bin add to nothing.
    A few thousand small slice updates: this branch is not on ``train_auto``'s path (it never passes a model), so it stays
    NumPy; the binary branch is the device kernel ``dcs_score_masks``.  ``normalise='sum'``: the fields are divided by their
    sum over the instruments instead (dataset.py:862; float32, instruments added in order)."""
    if normalise not in _NORMALISE:
        raise ValueError("normalise must be 'max' or 'sum'")
    T, F = int(shape[0]), int(shape[1])
    notes = np.asarray(notes, dtype=np.float64)
    weights = np.asarray(harmonics)
    mask = np.empty((T, ninst * F), dtype=np.float32)
    fields = []
    for j in range(ninst):
        field = np.full((T, F), np.float32(1e-18), dtype=np.float32)
        tab = notes[j]
        sounding = (tab[:, 2] > 0) & (np.minimum(tab[:, 1], stop) - np.maximum(tab[:, 0], start) > 0)
        for row in tab[sounding]:
            t0 = int(max(row[0], start)) - start
            t1 = int(min(row[1], stop)) - start
            lo, hi = row[3::2].astype(np.int64), row[4::2].astype(np.int64)
            w = weights[j, int(row[2])].astype(np.float32)
            for k in range(len(lo)):
                band = field[t0:t1, lo[k]:hi[k]]
                np.add(band, w[k], out=band)
        if normalise == 'sum':
            fields.append(field)
        else:
            np.divide(field, field.max(), out=mask[:, j * F:(j + 1) * F])
    if normalise == 'sum':
        total = fields[0].copy()
        for field in fields[1:]:
            np.add(total, field, out=total)
        for j, field in enumerate(fields):
            np.divide(field, total, out=mask[:, j * F:(j + 1) * F])
    return mask


def filterSpec(mag, notes, ninst, start, stop, timbre_model_path=None, ctx=None, normalise='max'):
    """Drop-in ``filterSpec``: NumPy ``mag [T, F]`` in, float32 mask ``[T, ninst*F]`` out -- the binary harmonic masks on
    the GPU (``dcs_score_masks``), or with ``timbre_model_path`` the template-weighted masks (:func:`timbre_masks`).
    ``normalise='sum'`` gives ``LargeDatasetMask2.filterSpec`` (dataset.py:839-879), the masks the trainers feed."""
    if timbre_model_path is not None:
        return timbre_masks(np.shape(mag), np.asarray(notes)[:ninst], ninst, int(start), int(stop),
                            load_timbre_model(timbre_model_path), normalise=normalise)
    from .runtime import default_context
    ctx = ctx if ctx is not None else default_context()
    mag_t = ctx.to_device(np.asarray(mag), np.float32)
    _, mask = score_masks(ctx, mag_t, np.asarray(notes)[:ninst], start, stop, want_input=False, want_mask=True,
                          normalise=normalise)
    return ctx.to_host(mask)
