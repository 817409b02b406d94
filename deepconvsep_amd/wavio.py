"""16-bit PCM wav files without the float detour (the batch-of-files driver).

Every script of the reference reads its input with ``scipy.io.wavfile.read`` and divides by the integer maximum
(separate_dsd.py:275-282), and writes ``(audio_out * 32767).astype('int16')`` with ``scipy.io.wavfile.write``
(:307-309).  For the 16-bit PCM files of the datasets those two calls are a header and a block of int16 frames: this module
reads the block as it lies in the file (one ``np.fromfile``), hands it to the device as int16 (the mix-down and the division
happen there, ``dcs_pcm16_to_float``) and writes the int16 block the device returns (``dcs_pcm_to_int16``) behind the 44-byte
header ``scipy.io.wavfile.write`` produces -- the files are byte-identical to the reference's, the host does no arithmetic.
Anything else (float or 24-bit samples, exotic chunk layouts) returns ``None`` and the caller falls back to scipy.

:class:`WavPool` is the same pair of operations for a batch of files at a time on the I/O threads of ``libdcs.so``
(``dcs_wav_read_pcm16_async`` / ``dcs_wav_write_pcm16_async``, ``csrc/wavio.hip``): one call per batch that returns at once,
no interpreter lock on the workers' side.
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
            if size < 16 or len(fmt) < 16:               # a file that ends inside the fmt chunk: not plain PCM, as documented
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
        body = memoryview(samples.reshape(-1)).cast("B") if samples.size else memoryview(b"")
        total = len(head) + len(body)
        done = os.writev(fd, [head, body] if len(body) else [head])
        while done < total:                              # a short write (signal, quota) may end inside the header as well:
            if done < len(head):                         # one remaining-bytes counter over header + body, as the C writer does
                done += os.write(fd, head[done:])
            else:
                done += os.write(fd, body[done - len(head):])
    finally:
        os.close(fd)


class WavBatch(object):
    """An enqueued batch of :class:`WavPool`; ``result()`` waits for it."""

    def __init__(self, lib, handle, keep, finish):
        self._lib, self._h, self._keep, self._finish, self._res = lib, handle, keep, finish, None

    def done(self):
        return self._h is None or bool(self._lib.dcs_wav_batch_done(self._h))

    def result(self):
        if self._h is not None:
            from . import _lib
            h, self._h = self._h, None
            _lib.check(self._lib.dcs_wav_batch_wait(h))
            self._res = self._finish()
            self._keep = self._finish = None
        return self._res

    def __del__(self):
        # a batch dropped without result() (fire-and-forget write): the native I/O threads still read the sample arrays and
        # write status / rate / frames through the pointers in self._keep -- wait for them before those objects are freed
        # (dcs_wav_batch_wait also releases the native batch object)
        h = getattr(self, "_h", None)
        if h is not None:
            self._h = None
            try:
                self._lib.dcs_wav_batch_wait(h)
            except Exception:
                pass


class WavPool(object):
    """I/O threads of ``libdcs.so`` for 16-bit PCM wav files (``dcs_wav_pool_create``).

    ``read_into(paths, buffers)``: the frames of file i go into ``buffers[i]`` (a writable, C-contiguous byte buffer such as a
    slice of a pinned staging area); the batch's ``result()`` is a list with, per file, ``(rate, n_frames, channels)``, ``None``
    (not plain 16-bit PCM or larger than the buffer -- :func:`read_pcm16_into`'s contract: the caller falls back to scipy) or an
    ``OSError``.  ``write(paths, rates, arrays)``: file i = ``scipy.io.wavfile.write(paths[i], rates[i], arrays[i])`` for int16
    arrays ``[n]`` / ``[n, channels]`` (C-contiguous; they are NOT copied and must stay untouched until ``result()``, which is a
    list of ``None`` or ``OSError``); missing parent directories are created."""

    def __init__(self, n_threads=16):
        import ctypes
        from . import _lib
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self._lib.dcs_wav_pool_create(int(n_threads), ctypes.byref(h)))
        self._h = h
        self.n_threads = int(n_threads)

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.dcs_wav_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @staticmethod
    def _paths(paths):
        import ctypes
        import os
        enc = [os.fsencode(p) for p in paths]
        return enc, (ctypes.c_char_p * len(enc))(*enc)

    def read_into(self, paths, buffers):
        import ctypes
        from . import _lib
        n = len(paths)
        if len(buffers) != n:
            raise ValueError("one buffer per path")
        enc, cpaths = self._paths(paths)
        arrs = [np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b.reshape(-1).view(np.uint8) for b in buffers]
        for a in arrs:
            if not a.flags.writeable or not a.flags.c_contiguous:
                raise ValueError("read_into needs writable contiguous buffers")
        dst = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
        cap = (ctypes.c_int64 * n)(*[a.nbytes for a in arrs])
        rate, frames = (ctypes.c_int32 * n)(), (ctypes.c_int64 * n)()
        chans, status = (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)()
        h = ctypes.c_void_p()
        _lib.check(self._lib.dcs_wav_read_pcm16_async(self._h, n, cpaths, dst, cap, rate, frames, chans, status, ctypes.byref(h)))

        def finish():
            import os
            out = []
            for i in range(n):
                if status[i] == 0:
                    out.append((int(rate[i]), int(frames[i]), int(chans[i])))
                elif status[i] > 0:
                    out.append(None)
                else:
                    out.append(OSError(-status[i], os.strerror(-status[i]), paths[i]))
            return out
        return WavBatch(self._lib, h, (enc, cpaths, arrs, dst, cap, rate, frames, chans, status), finish)

    def write(self, paths, rates, arrays):
        import ctypes
        from . import _lib
        n = len(paths)
        if len(arrays) != n or len(rates) != n:
            raise ValueError("one rate and one array per path")
        enc, cpaths = self._paths(paths)
        arrs = []
        for a in arrays:
            a = np.asarray(a)
            if a.dtype != np.int16 or a.ndim not in (1, 2) or not a.flags.c_contiguous:
                a = np.ascontiguousarray(a, dtype="<i2")
                if a.ndim not in (1, 2):
                    raise ValueError("write takes int16 arrays [n] or [n, channels]")
            arrs.append(a)
        data = (ctypes.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
        nfr = (ctypes.c_int64 * n)(*[a.shape[0] for a in arrs])
        chans = (ctypes.c_int32 * n)(*[1 if a.ndim == 1 else a.shape[1] for a in arrs])
        crates = (ctypes.c_int32 * n)(*[int(r) for r in rates])
        status = (ctypes.c_int32 * n)()
        h = ctypes.c_void_p()
        _lib.check(self._lib.dcs_wav_write_pcm16_async(self._h, n, cpaths, data, nfr, chans, crates, status, ctypes.byref(h)))

        def finish():
            import os
            return [None if status[i] == 0 else OSError(-status[i], os.strerror(-status[i]), paths[i]) for i in range(n)]
        return WavBatch(self._lib, h, (enc, cpaths, arrs, data, nfr, chans, crates, status), finish)
