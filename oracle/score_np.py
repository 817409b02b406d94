"""Score-informed front-end, CPU restatement (TEST INFRASTRUCTURE -- only tests/, smoke() and bench.py's
cpu_baseline may import this).

Follows the reference line by line:

  str2midi          util.py:581-606   note name -> MIDI number
  midi2freq         util.py:126-127
  getfreqs          util.py:183-191   harmonic band edges, +-interval cents  (harmonics 1 .. nharmonics-1)
  remove_overlap    util.py:140-157   merges overlapping bin ranges (keeps the LAST stop of a merged run)
  slicefft_slices   util.py:171-180   band edges -> FFT bin ranges
  expandMidi        util.py:424-512   score text -> [notes, 2*nharmonics+3] = (begin frame, end frame, midi, bin ranges)
  getMidiNum        util.py:526-579   number of notes kept (sizes the note table)
  filterSpec        examples/bach10_scoreinformed/separate_bach10.py:172-200   binary harmonic masks (or timbre-model weights), floor 1e-18,
                    normalised by the per-instrument max
  front end         examples/bach10_scoreinformed/separate_bach10.py:500-527   net input [4, T, F] = mask_j * mag

Python-2 arithmetic of the reference is kept explicit: ``samplerate / hop`` on ints is floor division and
``round`` rounds halves away from zero.

Parity status: PINNED -- tests/test_oracle_golden.py runs the reference's own code (oracle/ref_exec.py ``score()``)
on seeded scores and compares every table and mask bit for bit; tests/golden/score_*.npz carry those answers to the
GPU box.
"""
import math
import os

import numpy as np

MIDI_A4 = 69  # util.py:123


def _round_py2(x):
    """Python 2 ``round``: halves away from zero, returns a float."""
    return float(math.floor(abs(x) + 0.5)) * (1.0 if x >= 0 else -1.0)


def str2midi(note):  # util.py:581-606
    if note == "?":
        return float("nan")
    data = note.strip().lower()
    name2delta = {"c": -9, "d": -7, "e": -5, "f": -4, "g": -2, "a": 0, "b": 2}
    accident2delta = {"b": -1, "#": 1, "x": 2}

    def run(chars):
        out = []
        for ch in chars:
            if ch not in accident2delta:
                break
            out.append(ch)
        return out
    if data[1] in accident2delta:
        accidents = run(data[1:])
        octave_delta = int(data[len(accidents) + 1:]) - 4
    else:
        accidents = run(data[2:])
        octave_delta = int(data[1]) - 4
    return MIDI_A4 + name2delta[data[0]] + sum(accident2delta[a] for a in accidents) + 12 * octave_delta


def midi2freq(midi_number, tuning_freq=440.0):  # util.py:126-127
    return float(tuning_freq) * 2.0 ** ((float(midi_number) - 69.0) * (1.0 / 12.0))


def getfreqs(midinote, interval=30, tuning_freq=440, nharmonics=20):  # util.py:183-191
    factor = 2.0 ** (interval / 1200.0)
    f0 = float(midi2freq(midinote, tuning_freq=tuning_freq))
    fdowns = [f * f0 / float(factor) for f in range(1, nharmonics)]
    fups = [f * f0 * float(factor) for f in range(1, nharmonics)]
    return fups, fdowns


def remove_overlap(ranges):  # util.py:140-157
    result = []
    current_start = -1
    current_stop = -1
    for start, stop in sorted(ranges):
        if start > current_stop:
            result.append((start, stop))
            current_start, current_stop = start, stop
        else:
            result[-1] = (current_start, stop)          # the merged run takes THIS stop, even a smaller one
            current_stop = max(current_stop, stop)
    return result


def slicefft_ranges(pitch, size, interval=30, tuning_freq=440, nharmonics=20, sampleRate=44100):  # util.py:171-180
    if not pitch > 0:
        return []
    binfactor = float(size) / float(sampleRate)
    fups, fdowns = getfreqs(pitch, interval=interval, tuning_freq=tuning_freq, nharmonics=nharmonics)
    ranges = tuple((1 + int(np.floor(fdowns[f] * binfactor)), 1 + int(np.ceil(fups[f] * binfactor)))
                   for f in range(len(fdowns)))
    ranges = remove_overlap(ranges)
    return [r for r in ranges if r[1] <= (size // 2 + 1)]


def read_score(path):
    """onset,offset,name per line; '!' starts a comment (np.genfromtxt(comments='!', delimiter=','), util.py:427-428).
    Times go through float32 like dtype "f"."""
    begins, ends, names = [], [], []
    with open(path, "r") as fh:
        for line in fh:
            line = line.split("!")[0].strip()
            if not line:
                continue
            a, b, c = line.split(",")[:3]
            begins.append(float(np.float32(float(a))))
            ends.append(float(np.float32(float(b))))
            names.append(c[:3])
    return begins, ends, names


def _window(beginsO, endsO, beginTime, finishTime):  # util.py:433-441 / 534-542
    import bisect
    startTime = bisect.bisect_right(endsO, beginTime)
    endTime = bisect.bisect_left(beginsO, finishTime)
    if endsO[startTime] < float(beginTime):
        startTime = startTime + 1
    if endTime >= len(beginsO):
        endTime = len(beginsO) - 1
    elif beginsO[endTime] > float(finishTime):
        endTime = endTime - 1
    return startTime, endTime


def _clip_times(beginsO, endsO, startTime, endTime, beginTime, finishTime):  # util.py:444-458
    begins = [x - beginTime for x in beginsO[startTime:endTime + 1]]
    ends = [x - beginTime for x in endsO[startTime:endTime + 1]]
    span = finishTime - beginTime
    for i in range(len(begins)):
        if begins[i] < 0:
            begins[i] = 0.0
        if ends[i] < 0:
            ends[i] = 0.0
        if ends[i] > span:
            ends[i] = span
        if begins[i] > span:
            begins[i] = span
    return begins, ends


def getMidiNum(path, beginTime, finishTime):  # util.py:526-579
    beginsO, endsO, names = read_score(path)
    startTime, endTime = _window(beginsO, endsO, beginTime, finishTime)
    if not startTime < endTime:
        return 1
    begins, ends = _clip_times(beginsO, endsO, startTime, endTime, beginTime, finishTime)
    keep = [i for i in range(len(begins))
            if not (ends[i] <= 0 or ends[i] <= begins[i] or (ends[i] - begins[i]) < 0.01)]
    return len(keep)


def expandMidi(path, beginTime, finishTime, interval, tuning_freq, nharmonics, samplerate, hop, window, timeSpan_on,
               timeSpan_off, nframes, fermata=0.0):  # util.py:424-512
    fermata = max(timeSpan_off, fermata)
    beginsO, endsO, namesO = read_score(path)
    startTime, endTime = _window(beginsO, endsO, beginTime, finishTime)
    if not startTime < endTime:
        return None                                      # the reference falls off the end of the function
    begins, ends = _clip_times(beginsO, endsO, startTime, endTime, beginTime, finishTime)
    names = namesO[startTime:endTime + 1]
    tframes = float(nframes) * float(hop) / float(samplerate)
    keep = [i for i in range(len(begins))
            if not (ends[i] <= 0 or ends[i] <= begins[i] or begins[i] >= tframes or (ends[i] - begins[i]) < 0.01)]
    begins = [begins[i] for i in keep]
    ends = [ends[i] for i in keep]
    names = [names[i] for i in keep]
    fps = _round_py2(float(samplerate // hop))           # round(float(samplerate / hop)), Python-2 ints
    maxAllowed_on = int(_round_py2(timeSpan_on * float(samplerate // hop)))
    maxAllowed_off = int(_round_py2(timeSpan_off * float(samplerate // hop)))
    endMelody = int((finishTime - beginTime) * fps)
    melodyBegin, melodyEnd = [], []
    for i in range(len(ends)):
        melodyBegin.append(max(0, int(begins[i] * fps) - maxAllowed_on))
        intersect = [mb for mb, me in zip(begins, ends)
                     if (mb > begins[i]) and (me + timeSpan_off) >= (begins[i] - timeSpan_on)
                     and (mb - timeSpan_on) <= (ends[i] + timeSpan_off)]
        if len(intersect) == 0:
            notesafter = [x for x in begins if (x - timeSpan_on) > (ends[i] + timeSpan_off)]
            if len(notesafter) > 0:
                newoffset = min(ends[i] + fermata, max(0, min(notesafter) - timeSpan_on))
            else:
                newoffset = ends[i] + fermata
            melodyEnd.append(min(nframes, min(endMelody, int(newoffset * fps))))
        else:
            melodyEnd.append(min(nframes, min(endMelody, int(ends[i] * fps) + maxAllowed_off)))
    notes = [str2midi(n) for n in names]
    intervals = np.zeros((len(notes), 2 * nharmonics + 3))
    for m in range(len(notes)):
        intervals[m, 0] = melodyBegin[m]
        intervals[m, 1] = melodyEnd[m]
        intervals[m, 2] = notes[m]
        rs = slicefft_ranges(notes[m], size=window, interval=interval, tuning_freq=tuning_freq,
                             nharmonics=nharmonics, sampleRate=samplerate)
        intervals[m, 3:2 * len(rs) + 3:2] = [r[0] for r in rs]
        intervals[m, 4:2 * len(rs) + 4:2] = [r[1] for r in rs]
    return intervals


def filterSpec(mag, notes, ninst, start, stop, harmonics=None, normalise='max'):  # separate_bach10.py:172-200
    """``normalise``: ``'max'`` divides every instrument's field by its own maximum (the script, separate_bach10.py:195, and
    dataset.py:781 ``LargeDatasetMask``); ``'sum'`` divides by the sum over the instruments, bin by bin (the class the
    score-informed trainers use: ``LargeDatasetMask2.filterSpec``, dataset.py:839-866, ``filtered[j] / np.sum(filtered,
    axis=0)`` -- float32, instruments added in order).  ``harmonics`` None: the binary masks of ``timbre_model_path=None``.  Otherwise the timbre branch (:173-175,189-191)
    with ``harmonics [ninst, midi pitch, nharmonics]`` = the unpickled model: every harmonic band of every active note ADDS
    the instrument's template weight of that pitch and harmonic (in note, then harmonic order; float32 adds -- under the
    reference's NumPy a float64 scalar met a float32 array as float32, which is what a float32 model gives under any NumPy),
    bands of a note are NOT merged (overlapping bands add twice) and zero-width slots past the Nyquist bin add to nothing."""
    T, F = mag.shape
    filtered = np.ones((ninst, T, F), dtype=np.float32) * np.float32(1e-18)
    for j in range(ninst):
        for p in range(len(notes[j])):
            n0, n1, midi = notes[j, p, 0], notes[j, p, 1], notes[j, p, 2]
            if midi > 0 and max(0, min(n1, stop) - max(n0, start)) > 0:
                begin = int(max(n0, start)) - start
                end = int(min(n1, stop)) - start
                starts, stops = notes[j, p, 3::2], notes[j, p, 4::2]
                if harmonics is None:
                    for f in range(min(len(starts), len(stops))):
                        if stops[f] > 0:
                            filtered[j, begin:end, int(starts[f]):int(stops[f])] = 1.0
                else:
                    for k in range(len(starts)):
                        filtered[j, begin:end, int(starts[k]):int(stops[k])] += np.float32(harmonics[j, int(midi), k])
    mask = np.zeros((T, ninst * F), dtype=np.float32)
    if normalise == 'sum':
        total = np.sum(filtered, axis=0)                      # float32, ((f0 + f1) + f2) + ...
        for j in range(ninst):
            mask[:, j * F:(j + 1) * F] = filtered[j] / total
        return mask
    if normalise != 'max':
        raise ValueError(normalise)
    for j in range(ninst):
        mask[:, j * F:(j + 1) * F] = filtered[j] / np.max(filtered[j])
    return mask


def melody_table(score_paths, nframes, samplerate=44100, hop=512, window=4096, interval=50, tuning_freq=440,
                 nharmonics=20):
    """separate_bach10.py:508-518: per-instrument note tables stacked into [ninst, max notes, 2*nharmonics+3]."""
    nelem = 1
    for pth in score_paths:
        nelem = max(getMidiNum(pth, 0, 40.0), nelem)
    melody = np.zeros((len(score_paths), int(nelem), 2 * nharmonics + 3))
    for i, pth in enumerate(score_paths):
        tmp = expandMidi(pth, 0, 40.0, interval, tuning_freq, nharmonics, samplerate, hop, window, 0.2, 0.2, nframes, 0.5)
        melody[i, :tmp.shape[0], :] = tmp
    return melody


def network_input(mag, melody, nframes, harmonics=None, normalise='max'):
    """separate_bach10.py:520-527 (trainCNNrwc.py:388-396 with ``normalise='sum'``): [ninst, T, F] float64 = mask_j * mag
    (float32 product stored in float64)."""
    ninst = melody.shape[0]
    jump = mag.shape[-1]
    masks_temp = filterSpec(mag, melody, ninst, 0, nframes, harmonics, normalise=normalise)
    masks = np.ones((ninst, mag.shape[0], mag.shape[1]))
    for j in range(ninst):
        masks[j] = masks_temp[:, j * jump:(j + 1) * jump] * mag
    return masks


def synth_score(path, seed, n_notes=24, total=9.5, lo=40, hi=76, with_comment=True):
    """Seeded score text: monophonic line with rests, a very short note, an unknown pitch ('?' is not exercised --
    the reference's str2midi returns nan for it and the note is then skipped by ``notes[j,p,2] > 0``)."""
    rs = np.random.RandomState(seed)
    names = ["C", "C#", "D", "Eb", "E", "F", "F#", "G", "Ab", "A", "Bb", "B"]
    t = float(rs.uniform(0.0, 0.6))
    lines = []
    if with_comment:
        lines.append("! onset,offset,note")
    for k in range(n_notes):
        dur = float(rs.choice([0.005, 0.12, 0.3, 0.45, 0.8], p=[0.06, 0.2, 0.3, 0.3, 0.14]))
        gap = float(rs.choice([0.0, 0.02, 0.15, 0.6], p=[0.35, 0.25, 0.25, 0.15]))
        midi = int(rs.randint(lo, hi))
        name = "%s%d" % (names[midi % 12], midi // 12 - 1)
        if t + dur > total:
            break
        lines.append("%.3f,%.3f,%s" % (t, t + dur, name))
        t += dur + gap
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    return path
