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
    # score-informed front-end.  library: MIDI constants + midi2freq, remove_overlap, slicefft_slices + getfreqs,
    # expandMidi, getMidiNum + str2midi; script: filterSpec
    "lib_midi_const": ("util.py", 123, 127),
    "lib_remove_overlap": ("util.py", 140, 157),
    "lib_slices": ("util.py", 171, 191),
    "lib_expandmidi": ("util.py", 424, 512),
    "lib_midinum": ("util.py", 526, 606),
    "script_si_filterspec": ("examples/bach10_scoreinformed/separate_bach10.py", 172, 200),
    # the dataset class the score-informed trainers instantiate (trainCNNrwc.py:657): LargeDatasetMask2.filterSpec, a method
    # (self.ninst, self.tensortype, self.timbre_model_path, self.harmonics) -- every bin divided by the sum over instruments
    "lib_dataset_filterspec_sum": ("dataset.py", 839, 879),
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


class _NpText(object):
    """numpy with ONE Python-2 text semantic restored: ``genfromtxt(dtype=[..., "S3"])`` yields ``str`` note names
    (under Python 3 they would be ``bytes`` and str2midi's character tests would compare ints)."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def genfromtxt(fname, **kw):
        dt = kw.get("dtype")
        if isinstance(dt, (list, tuple)):
            kw["dtype"] = [d.replace("S", "U") if isinstance(d, str) else d for d in dt]
        return np.genfromtxt(fname, **kw)


def score():
    """expandMidi / getMidiNum / filterSpec and their helpers, executed from the reference tree.  Python-2 names the
    bodies expect are supplied: ``filter`` returning a list, ``bisect``/``itertools`` imports of util.py:28-29,
    ``nan``.  Integer divisions: ``samplerate / hop`` only appears inside ``round(float(...))`` or multiplied and
    rounded, and ``size/2+1`` in a comparison -- for sr=44100 and the hop sizes used here (512, 256, 441) Python 2
    and 3 give the same integers (checked in tests/test_oracle_golden.py)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import bisect
    import itertools
    ns = _NS(np=_NpText(), os=os, bisect_left=bisect.bisect_left, bisect_right=bisect.bisect_right, it=itertools,
             nan=float("nan"), filter=lambda f, xs: [x for x in xs if f(x)], pickle=__import__("pickle"),
             __name__="ref_exec.score")
    for name in ("lib_midi_const", "lib_remove_overlap", "lib_slices", "lib_expandmidi", "lib_midinum",
                 "script_si_filterspec"):
        relpath, first, last = _SLICES[name]
        exec(compile(_slice(relpath, first, last), relpath, "exec"), ns)
    return ns


def dataset_filterspec_sum(tensortype=np.float32, harmonics=None):
    """``LargeDatasetMask2.filterSpec`` (dataset.py:839-879), the body of the method executed as it stands with a stand-in
    ``self`` that carries the four attributes it reads: ``f(mag, notes, start, stop) -> mask [T, ninst*F]``.  The trainers
    build the class with ``tensortype=theano.config.floatX`` (trainCNNrwc.py:657-658): float32."""
    import textwrap
    import types
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    relpath, first, last = _SLICES["lib_dataset_filterspec_sum"]
    ns = _NS(np=np, __name__="ref_exec.dataset")
    exec(compile("\n" * (first - 1) + textwrap.dedent(_slice(relpath, first, last)), relpath, "exec"), ns)
    fn = ns["filterSpec"]

    def run(mag, notes, start, stop):
        me = types.SimpleNamespace(tensortype=tensortype, timbre_model_path=None if harmonics is None else "model",
                                   harmonics=harmonics)
        return fn(me, mag, notes, start, stop)
    return run


def script_dsd():
    return load("script_dsd")


def script_ikala():
    return load("script_ikala")


def lib_stft():
    return load("lib_stft")


def lib_tiling():
    return load("lib_tiling")


# ------------------------------------------------------------------------------------------------ networks
# The reference's own build_ca source and mask expressions, executed with oracle.lasagne_np standing in for
# ``lasagne`` (eager NumPy float64).  Graph construction, filter sizes, the fc12 alias, the parameter order of
# set_all_param_values and the mask arithmetic then come from the reference's code; the semantics of the nine Lasagne
# layer classes are restated third-party behaviour (lasagne_np's docstring).
_NET_FILES = {
    "dsd": "examples/dsd100/separate_dsd.py",
    "hiphop": "examples/hiphopss/separate_hhds.py",
    "ikala": "examples/ikala/separate_ikala.py",
    "ikala_nopool": "examples/ikala/trainCNN.py",                      # the trainer's graph: no MaxPool2DLayer
    "bach10": "examples/bach10/separate_bach10.py",
    "bach10_si": "examples/bach10_scoreinformed/separate_bach10.py",
    "bach10_si1": "examples/bach10_scoreinformed/trainCNNrwc_samp.py",   # a trainer: the single-branch build_ca (:195-235)
    "bach10_si_trainer": "examples/bach10_scoreinformed/trainCNNrwc.py", # a trainer: the 17-array graph, masks x channel SUM
    "dsd_ild": "examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py",     # a trainer: build_ca only, returns a dict of layers
}


def _file_lines(relpath):
    with open(os.path.join(REFERENCE_ROOT, relpath), "r") as fh:
        return fh.readlines()


def _def_block(lines, name):
    """Source of the top-level function ``name``: from its ``def`` to the line before the next top-level statement."""
    start = next(i for i, l in enumerate(lines) if l.startswith("def %s(" % name))
    end = start + 1
    while end < len(lines) and (lines[end].strip() == "" or lines[end][0] in " \t#"):
        end += 1
    return "".join(lines[start:end]), start + 1


def build_network(arch, x, time_context=None, **extra):
    """Run the reference's ``build_ca`` of ``arch`` on the input batch ``x [B, C, tc, F]`` (the ndarray is what the
    script passes as ``input_var``).  Returns (output layer, lasagne stand-in module)."""
    import textwrap  # noqa: F401
    from . import lasagne_np
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    x = np.asarray(x, dtype=np.float64)
    B, C, tc, F = x.shape
    src, first = _def_block(_file_lines(_NET_FILES[arch]), "build_ca")
    ns = dict(np=np, lasagne=lasagne_np, __name__="ref_exec.net." + arch)
    exec(compile("\n" * (first - 1) + src, _NET_FILES[arch], "exec"), ns)
    kw = dict(input_var=x, batch_size=B, time_context=tc, feat_size=F)
    if arch in ("bach10_si", "bach10_si1", "bach10_si_trainer", "dsd_ild"):
        kw["nchannels"] = C
    kw.update(extra)
    net = ns["build_ca"](**kw)
    return (net["l_out"] if isinstance(net, dict) else net), lasagne_np


def network_param_shapes(arch, B, C, tc, F):
    """Shapes in the order ``lasagne.layers.set_all_param_values`` expects for the reference's graph."""
    out, L = build_network(arch, np.zeros((B, C, tc, F)))
    return [p.shape for p in L.get_all_params(out)]


def network_output(arch, params, x, tie_mode="all"):
    """``lasagne.layers.get_output(network2, deterministic=True)`` of the reference's graph with ``params`` set by
    ``set_all_param_values`` (separate_dsd.py:246-252): float64 ``[B, branches*C, tc, F]``."""
    out, L = build_network(arch, x)
    L.set_all_param_values(out, [np.asarray(p, dtype=np.float64) for p in params])
    L.TIE_MODE[0] = tie_mode
    try:
        return L.get_output(out, deterministic=True)
    finally:
        L.TIE_MODE[0] = "all"


def mask_sources(arch, prediction2, x, rand=0.5):
    """The reference's mask expressions (the lines of ``train_auto`` between the last ``rand_num = np.random.uniform``
    draw and ``predict_function2=theano.function(...)``), executed on ndarrays; the unseeded draw is replaced by the
    constant ``rand``.  Returns the list of outputs named in the ``theano.function`` call."""
    import re
    import textwrap
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    lines = _file_lines(_NET_FILES[arch])
    t0 = next(i for i, l in enumerate(lines) if l.startswith("def train_auto("))
    fn = next(i for i in range(t0, len(lines)) if "predict_function2" in lines[i] and "theano.function" in lines[i])
    draw = max(i for i in range(t0, fn) if "rand_num = np.random.uniform" in lines[i])
    eps_line = next(l for l in lines[t0:fn] if re.match(r"\s*eps\s*=", l))
    # a trainer's train_auto goes on with the losses and the update rules (lasagne.objectives / theano.function) before it
    # reaches predict_function2: the mask expressions end where the first of those begins
    stop = next((i for i in range(draw + 1, fn) if "lasagne." in lines[i] or "theano." in lines[i]), fn)
    body = textwrap.dedent("".join(lines[draw + 1:stop]))
    names = re.search(r"\[input_var2\]\s*,\s*\[([^\]]*)\]", lines[fn]).group(1).split(",")
    x = np.asarray(x, dtype=np.float64)
    ns = dict(np=np, prediction2=np.asarray(prediction2, dtype=np.float64), input_var2=x,
              rand_num=np.full((x.shape[0], 1) + x.shape[2:], float(rand)))
    exec(textwrap.dedent(eps_line), ns)
    exec(compile("\n" * (draw + 1) + body, _NET_FILES[arch], "exec"), ns)
    return [np.asarray(ns[n.strip()]) for n in names]
