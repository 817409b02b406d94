"""NumPy restatement of the tilers and the cross-fade overlap-add (test
infrastructure).

Follows ``examples/dsd100/separate_dsd.py:114-135`` (script tiler),
``util.py:220-248`` (library tiler), ``util.py:251-294`` (2-source
``overlapadd``) and ``util.py:297-327`` = ``separate_dsd.py:139-169``
(``overlapadd_multi``).
"""
import numpy as np

SCRIPT = "script"
LIBRARY = "library"


def tile_starts(n_frames, time_context, overlap, tiler):
    """Start frame of every tile.

    script tiler : tiles while ``start + time_context < T`` (strict;
                   separate_dsd.py:123,131) -- the tail is dropped.
    library tiler: tiles while ``start + overlap < T`` (util.py:230,237) --
                   the last tile is zero padded.
    Both advance by ``time_context - overlap`` (:125 / util.py:246).
    """
    stride = time_context - overlap
    guard = time_context if tiler == SCRIPT else overlap
    starts = []
    s = 0
    while s + guard < n_frames:
        starts.append(s)
        s += stride
    return starts


def generate_overlapadd(allmix, input_size, time_context=30, overlap=10, batch_size=32,
                        tiler=SCRIPT, fill=0.0):
    """Cut ``allmix`` (``[T,F]`` or ``[C,T,F]``) into ``[nb,B,C,tc,F]`` float64.

    The script tiler allocates with ``np.empty`` (separate_dsd.py:126), so slots
    past the last tile hold whatever was in memory; this restatement fills them
    with ``fill`` and returns the count so callers can ignore them.  The
    library tiler zero-initialises (util.py:233) and zero-pads the last tile
    (util.py:238-243).
    """
    allmix = np.asarray(allmix)
    if allmix.ndim > 2:
        if tiler == SCRIPT:
            raise ValueError("the script tiler only takes [T,F] input")
        nch = allmix.shape[0]
    else:
        nch = 1
    assert input_size == allmix.shape[-1], \
        "Feature size must be the same as the last dimension of the spectrogram"
    T = allmix.shape[-2]
    starts = tile_starts(T, time_context, overlap, tiler)
    n = len(starts)
    nb = int(np.ceil(float(n) / batch_size))
    fbatch = np.full((nb, batch_size, nch, time_context, input_size), fill, dtype=np.float64)
    if tiler == LIBRARY:
        fbatch[...] = 0.0
    for i, s in enumerate(starts):
        e = min(s + time_context, T)
        b, k = int(i / batch_size), int(i % batch_size)
        if tiler == LIBRARY:
            fbatch[b, k, :, :, :] = 0.0
        if allmix.ndim > 2:
            fbatch[b, k, :, :e - s, :] = allmix[:, s:e, :]
        else:
            fbatch[b, k, :, :e - s, :] = allmix[s:e, :]
    return fbatch, n


def _crossfade(overlap):
    rise = np.linspace(0.0, 1.0, num=overlap)
    return rise, rise[::-1]


def overlapadd_multi(fbatch, nchunks, overlap=10):
    """Sequential cross-fade stitch (util.py:297-327).

    ``fbatch`` is ``[nb, S, B, 1, tc, F]`` (``np.array`` of the per-batch lists of
    per-source outputs).  Tile 0 is copied (:321-322); tile k at
    ``start = k*(tc-overlap)`` overwrites ``[start+overlap, start+tc)`` (:324) and
    blends ``[start, start+overlap)`` as ``fall*old + rise*new`` (:325).
    Result ``[S, nchunks*(tc-overlap)+tc, F]`` (:313).
    """
    fbatch = np.asarray(fbatch)
    F = fbatch.shape[-1]
    tc = fbatch.shape[-2]
    B = fbatch.shape[2]
    S = fbatch.shape[1]
    stride = tc - overlap
    rise, fall = _crossfade(overlap)
    rise = rise[:, None]
    fall = fall[:, None]
    sep = np.zeros((S, nchunks * stride + tc, F))
    for s in range(S):
        start = 0
        for i in range(nchunks):
            tile = fbatch[int(i / B), s, int(i % B), 0]
            if start == 0:
                sep[s, 0:tc] = tile
            else:
                sep[s, start + overlap:start + tc] = tile[overlap:tc]
                sep[s, start:start + overlap] = fall * sep[s, start:start + overlap] + rise * tile[:overlap]
            start += stride
    return sep


def overlapadd(fbatch, nchunks, overlap=10):
    """2-source variant (util.py:251-294): same stitch, returns ``(sep1, sep2)``."""
    sep = overlapadd_multi(np.asarray(fbatch)[:, :2], nchunks, overlap)
    return sep[0], sep[1]


# --------------------------------------------------------------------------
# Closed form of the sequential stitch, used to check the GPU formulation.
# --------------------------------------------------------------------------
def overlapadd_frame_parallel(tiles, overlap):
    """Per-frame fold equivalent to ``overlapadd_multi`` for ONE source.

    ``tiles`` is ``[n, tc, F]``.  Output frame t is first written by the last
    tile k0 with ``k0*stride + overlap <= t`` (copy region; tile 0 also owns
    frames < overlap), then every later tile k that still covers t
    (``k*stride <= t``, local index ``j = t - k*stride < overlap``) applies
    ``acc = fall[j]*acc + rise[j]*tile[k, j]`` in increasing k.  The arithmetic
    (operands and order) is the reference's, so the result is bit-identical.
    """
    tiles = np.asarray(tiles)
    n, tc, F = tiles.shape
    stride = tc - overlap
    rise, fall = _crossfade(overlap)
    total = n * stride + tc
    out = np.zeros((total, F))
    if n == 0:
        return out
    for t in range(total):
        # owner: largest k with k*stride + overlap <= t  (k=0 owns t < overlap too)
        k0 = 0 if t < overlap else min((t - overlap) // stride, n - 1)
        j0 = t - k0 * stride
        if j0 >= tc:
            continue  # beyond the last tile: stays zero (only when n == 0 rows)
        acc = tiles[k0, j0].copy()
        k = k0 + 1
        while k < n and k * stride <= t:
            j = t - k * stride
            acc = fall[j] * acc + rise[j] * tiles[k, j]
            k += 1
        out[t] = acc
    return out
