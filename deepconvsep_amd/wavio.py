"""16-bit PCM wav files without the float detour (the batch-of-files driver).

Every script of the reference reads its input with ``scipy.io.wavfile.read`` and divides by the integer maximum
(separate_dsd.py:275-282), and writes ``(audio_out * 32767).astype('int16')`` with ``scipy.io.wavfile.write``
(:307-309).  For the 16-bit PCM files of the datasets those two calls are a header and a block of int16 frames: this module
reads the block as it lies in the file (one ``np.fromfile``), hands it to the device as int16 (the mix-down and the division
happen there, ``dcs_pcm16_to_float``) and writes the int16 block the device returns (``dcs_pcm_to_int16``) behind the 44-byte
header ``scipy.io.wavfile.write`` produces -- the files are byte-identical to the reference's, the host does no arithmetic.
Anything else (float or 24-bit samples, exotic chunk layouts) returns ``None`` and the caller falls back to scipy.
"""
import struct

import numpy as np

_PCM, _EXTENSIBLE = 1, 0xFFFE
_PCM_SUBFORMAT = b"\x01\x00\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"


def _find_data(fh):
    """(rate, channels, data offset, data bytes) of an open 16-bit PCM wav, or None."""
    head = fh.read(12)
    if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        return None
    rate = channels = None
    while True:
        ck = fh.read(8)
        if len(ck) < 8:
            return None
        tag, size = ck[:4], struct.unpack("<I", ck[4:])[0]
        if tag == b"fmt ":
            fmt = fh.read(size)
            if size < 16:
                return None
            code, channels, rate, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
            if code == _EXTENSIBLE and size >= 40 and fmt[24:40] == _PCM_SUBFORMAT:
                code = _PCM
            if code != _PCM or bits != 16 or channels < 1 or align != 2 * channels:
                return None
            if size & 1:
                fh.seek(1, 1)
        elif tag == b"data":
            if rate is None:
                return None
            offset = fh.tell()
            fh.seek(0, 2)
            avail = fh.tell() - offset
            return rate, channels, offset, min(size, avail)
        else:
            fh.seek(size + (size & 1), 1)


def read_pcm16_into(path, buf):
    """Read the int16 frames of a 16-bit PCM wav straight into ``buf`` (a writable byte buffer, e.g. a slice of a pinned
    staging area at least as large as the file): ``(rate, n_frames, channels)``, or ``None`` when the file is not plain 16-bit
    PCM or does not fit (the caller then falls back to the float path)."""
    with open(path, "rb", buffering=0) as fh:
        info = _find_data(fh)
        if info is None:
            return None
        rate, channels, offset, nbytes = info
        frames = nbytes // (2 * channels)
        nbytes = frames * 2 * channels
        mv = memoryview(buf).cast("B")
        if nbytes > len(mv):
            return None
        fh.seek(offset)
        got = 0
        while got < nbytes:
            k = fh.readinto(mv[got:nbytes])
            if not k:
                return None
            got += k
        return rate, frames, channels


def read_pcm16(path):
    """``(rate, frames)`` with ``frames`` an int16 array ``[n]`` (mono) or ``[n, channels]`` -- exactly what
    ``scipy.io.wavfile.read`` returns for a 16-bit PCM file -- or ``None`` when the file is not plain 16-bit PCM."""
    with open(path, "rb") as fh:
        info = _find_data(fh)
    if info is None:
        return None
    rate, channels, offset, nbytes = info
    frames = nbytes // (2 * channels)
    data = np.fromfile(path, dtype="<i2", count=frames * channels, offset=offset)
    return rate, (data if channels == 1 else data.reshape(frames, channels))


def header_pcm16(rate, n_frames, channels=1):
    """The 44 bytes ``scipy.io.wavfile.write`` puts in front of int16 data."""
    nbytes = int(n_frames) * 2 * channels
    return (b"RIFF" + struct.pack("<I", 36 + nbytes) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, _PCM, channels, int(rate), int(rate) * 2 * channels, 2 * channels, 16) +
            b"data" + struct.pack("<I", nbytes))


def write_pcm16(path, rate, samples):
    """``scipy.io.wavfile.write(path, rate, samples)`` for an int16 array ``[n]`` or ``[n, channels]``: header + the bytes, one
    ``writev`` on a raw descriptor (no buffered file object between the pinned samples and the page cache)."""
    import os
    samples = np.ascontiguousarray(samples, dtype="<i2")
    channels = 1 if samples.ndim == 1 else samples.shape[1]
    head = header_pcm16(rate, samples.shape[0], channels)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        if samples.size:
            body = memoryview(samples.reshape(-1)).cast("B")
            done = os.writev(fd, [head, body])
            done -= len(head)
            while done < len(body):                      # a short write (signal, quota): finish with plain writes
                done += os.write(fd, body[done:])
        else:
            os.write(fd, head)
    finally:
        os.close(fd)
