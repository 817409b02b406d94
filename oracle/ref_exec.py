"""Run the reference's OWN pure-NumPy function bodies (test infrastructure).

The reference tree is Python 2 (``print "..."``, ``file()``, ``cPickle``) and
imports Theano/Lasagne at module level, so its modules cannot be imported.
Its STFT / tiling / overlap-add helpers however are plain NumPy and execute
unchanged under Python 3 / NumPy 2 when sliced out by line range.  This module
slices them out of ``/root/reference`` AT RUN TIME (nothing is copied into the
repo) and exposes them as a namespace.

``/root/reference`` exists only in the build container; on the GPU box
``available()`` is False and everything that depends on this module is skipped
-- the committed ``tests/golden/*.npz`` carry the reference's answers there.
"""
import os

import numpy as np

REFERENCE_ROOT = os.environ.get("DCS_REFERENCE_ROOT", "/root/reference")

# (file, first line, last line) -- 1-based, inclusive.
_SLICES = {
    # compute_file, compute_inverse, sinebell, stft_norm, istft_norm,
    # generate_overlapadd (script tiler), overlapadd_multi
    "script_dsd": ("examples/dsd100/separate_dsd.py", 24, 169),
    # same helpers + the 2-source overlapadd
    "script_ikala": ("examples/ikala/separate_ikala.py", 24, 169),
    # library: sinebell
    "lib_sinebell": ("transform.py", 35, 49),
    # library: stft_norm, istft_norm
    "lib_stft": ("transform.py", 277, 396),
    # library: generate_overlapadd (zero-padding tiler), overlapadd, overlapadd_multi
    "lib_tiling": ("util.py", 220, 327),
}


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "transform.py"))


def _slice(relpath, first, last):
    with open(os.path.join(REFERENCE_ROOT, relpath), "r") as fh:
        lines = fh.readlines()
    return "".join(lines[first - 1:last])


class _NS(dict):
    __getattr__ = dict.__getitem__


def load(name):
    """Execute one slice and return its namespace (functions as attributes)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    relpath, first, last = _SLICES[name]
    ns = _NS(np=np, __name__="ref_exec." + name)
    if name == "lib_stft":
        # stft_norm's default argument calls sinebell() at def time
        exec(compile(_slice(*_SLICES["lib_sinebell"]), "transform.py[sinebell]", "exec"), ns)
    exec(compile(_slice(relpath, first, last), relpath, "exec"), ns)
    return ns


def script_dsd():
    return load("script_dsd")


def script_ikala():
    return load("script_ikala")


def lib_stft():
    return load("lib_stft")


def lib_tiling():
    return load("lib_tiling")
