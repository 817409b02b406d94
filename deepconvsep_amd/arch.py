"""Static descriptions of the reference's ``build_ca`` graphs.

Each entry states what the reference builds with Lasagne for one separation
script; the HIP model (``csrc/``) is instantiated from it.

  dsd / hiphop  examples/dsd100/separate_dsd.py:172-236  (hiphopss/separate_hhds.py:171-235)
  ikala         examples/ikala/separate_ikala.py:172-192   (max-pool variant)
  ikala_nopool  examples/ikala/trainCNN.py:87-118         (the trainer's graph: no pooling layer; a .pkl written by that
                trainer has 90 090 rows in fc.W instead of 13 230 -- `resolve` picks the graph from the shapes)
  bach10        examples/bach10/separate_bach10.py:172-229
  bach10_si     examples/bach10_scoreinformed/separate_bach10.py:388-447
  bach10_si1    examples/bach10_scoreinformed/trainCNNrwc_samp.py:195-235   (the single-branch form, 11 arrays: the branch
                ``predict_function2`` reads from the 17-array graph, written out on its own)
  dsd_ild       examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py:66-115   (stereo input, 4 branches x 2 channels)
"""
import numpy as np

# enum values shared with include/dcs.h
ARCH_DSD, ARCH_IKALA, ARCH_BACH10, ARCH_BACH10_SI, ARCH_DSD_ILD, ARCH_IKALA_NOPOOL, ARCH_BACH10_SI1 = 0, 1, 2, 3, 4, 5, 6
SCORE_NORM_MAX, SCORE_NORM_SUM = 0, 1     # harmonic masks / own maximum (script) | / sum over instruments (trainers)
MIX_CH0, MIX_SUM = 0, 1                   # soft masks x input channel 0 (script) | x channel sum (trainers)
EPS_A, EPS_B = 0, 1
EPS_ILD = 3   # per input channel, p / (sum + 1e-12 r), + 1e-12 r (trainCNN_ILD_DSD100.py:176-180); dcs_separate_stereo only
TIE_ALL, TIE_FIRST = 0, 1
TILER_SCRIPT, TILER_LIBRARY = 0, 1


class Arch(object):
    def __init__(self, name, code, in_channels, conv1, pool_w, conv2, hidden, branch_fc,
                 n_sources, eps_mode, source_names):
        self.name = name
        self.code = code
        self.C = in_channels
        self.conv1 = conv1            # (filters, kernel width or 'F', stride along frequency)
        self.pool_w = pool_w          # 0 = no pooling layer
        self.conv2 = conv2            # (filters, kernel height fn(tc), kernel width)
        self.hidden = hidden
        self.branch_fc = list(branch_fc)
        self.n_fc = max(branch_fc) + 1
        self.S = n_sources
        self.eps_mode = eps_mode
        self.source_names = source_names

    def dims(self, tc, F):
        nf1, kw1, sw1 = self.conv1
        kw1 = F if kw1 == 'F' else kw1
        w1 = (F - kw1) // sw1 + 1
        wp = w1 // self.pool_w if self.pool_w else w1
        nf2, kh2, kw2 = self.conv2
        kh2 = kh2(tc)
        h2, w2 = tc - kh2 + 1, wp - kw2 + 1
        return dict(nf1=nf1, kw1=kw1, sw1=sw1, w1=w1, wp=wp, nf2=nf2, kh2=kh2, kw2=kw2,
                    h2=h2, w2=w2, flat=nf2 * h2 * w2)

    def param_shapes(self, tc, F):
        """Shapes in ``lasagne.layers.get_all_params`` order = the ``.pkl`` order."""
        d = self.dims(tc, F)
        shapes = [(d['nf1'], self.C, 1, d['kw1']), (d['nf1'],), (d['nf1'],),
                  (d['nf2'], d['nf1'], d['kh2'], d['kw2']), (d['nf2'],), (d['nf2'],),
                  (d['flat'], self.hidden), (self.hidden,)]
        for _ in range(self.n_fc):
            shapes += [(self.hidden, d['flat']), (d['flat'],)]
        shapes.append((len(self.branch_fc) * self.C,))
        return shapes

    def live_branches(self):
        """Decoder branches that reach the masks: the masks use the first S output channels
        (``prediction2[:, 0:S]``) and a branch contributes C of them -- all branches for the single-channel graphs,
        branch 0 alone for the score-informed graph (C = 4, S = 4; the other three are dead code for
        ``predict_function2``, bach10_scoreinformed/separate_bach10.py:475-488)."""
        return min(len(self.branch_fc), -(-self.S // self.C))

    def flops_per_tile(self, tc, F, live_only=False):
        """Algorithmic multiply-add FLOPs of one tile through the reference graph
        (every output branch computed, as Theano does for the aliased DSD branch only once);
        ``live_only``: only the branches ``predict_function2`` needs."""
        d = self.dims(tc, F)
        conv1 = 2 * d['nf1'] * self.C * d['kw1'] * tc * d['w1']
        conv2 = 2 * d['nf2'] * d['nf1'] * d['kh2'] * d['kw2'] * d['h2'] * d['w2']
        fc = 2 * d['flat'] * self.hidden
        nb = min(self.n_fc, self.live_branches()) if live_only else self.n_fc
        return conv1 + conv2 + fc + nb * (fc + conv2 + conv1)


ARCHS = {
    'dsd': Arch('dsd', ARCH_DSD, 1, (50, 'F', 1), 0, (50, lambda tc: int(tc / 2), 1), 128,
                [0, 1, 2, 1], 4, EPS_A, ['vocals', 'bass', 'drums', 'other']),
    'ikala': Arch('ikala', ARCH_IKALA, 1, (30, 30, 3), 4, (30, lambda tc: 10, 20), 256,
                  [0, 1], 2, EPS_A, ['voice', 'music']),
    'bach10': Arch('bach10', ARCH_BACH10, 1, (30, 30, 4), 0, (30, lambda tc: int(2 * tc / 3), 1), 256,
                   [0, 1, 2, 3], 4, EPS_B, ['bassoon', 'clarinet', 'saxphone', 'violin']),
    'bach10_si': Arch('bach10_si', ARCH_BACH10_SI, 4, (30, 30, 4), 0,
                      (30, lambda tc: int(2 * tc / 3), 1), 256, [0, 1, 2, 3], 4, EPS_B,
                      ['bassoon', 'clarinet', 'saxphone', 'violin']),
    'dsd_ild': Arch('dsd_ild', ARCH_DSD_ILD, 2, (50, 'F', 1), 0, (50, lambda tc: int(tc / 2), 1), 256,
                    [0, 1, 2, 3], 4, EPS_ILD, ['vocals', 'bass', 'drums', 'other']),
}
ARCHS['bach10_si1'] = Arch('bach10_si1', ARCH_BACH10_SI1, 4, (30, 30, 4), 0, (30, lambda tc: int(2 * tc / 3), 1), 256, [0], 4,
                           EPS_B, ['bassoon', 'clarinet', 'saxphone', 'violin'])
ARCHS['hiphop'] = ARCHS['dsd']
ARCHS['ikala_nopool'] = Arch('ikala_nopool', ARCH_IKALA_NOPOOL, 1, (30, 30, 3), 0, (30, lambda tc: 10, 20), 256,
                             [0, 1], 2, EPS_A, ['voice', 'music'])


def resolve(arch, params, tc, F):
    """The graph a parameter list belongs to.  'ikala' is two graphs in the reference: separate_ikala.py builds it with a
    (1, 4) max-pool (separate_ikala.py:176), ikala/trainCNN.py:92-100 without; the trainer's .pkl files therefore only load
    into the no-pool graph.  The bottleneck's input size tells them apart (fc.W rows 13 230 vs 90 090 at 513 bins)."""
    a = ARCHS[arch] if isinstance(arch, str) else arch
    # score-informed: the 17-array graph of the separate script / trainCNNrwc.py, or the 11-array single-branch graph of
    # trainCNNrwc_samp.py:195-235 -- the same script loads either .pkl
    if a.name == 'bach10_si' and len(params) == len(ARCHS['bach10_si1'].param_shapes(tc, F)):
        return ARCHS['bach10_si1']
    if a.name == 'ikala' and len(params) > 6 and np.ndim(params[6]) == 2:
        rows = int(np.shape(params[6])[0])
        if rows != a.dims(tc, F)['flat'] and rows == ARCHS['ikala_nopool'].dims(tc, F)['flat']:
            return ARCHS['ikala_nopool']
    return a


def live_params(arch, params):
    """The parameters ``predict_function2`` can reach, as a graph of their own.  Only the 17-array score-informed graph has
    dead ones: its masks read ``prediction2[:, 0:4]`` = the four channels of decoder branch 0
    (bach10_scoreinformed/separate_bach10.py:475-488), so three per-source dense layers (3 x 171 MB at 2049 bins) and twelve
    output biases are never used -- what is left is exactly the 11-array graph ``bach10_si1``.  Returns (arch, params)."""
    if arch.name == 'bach10_si' and len(params) == 17:
        live = list(params[:10]) + [np.asarray(params[16])[:arch.C]]
        return ARCHS['bach10_si1'], live
    return arch, params


def check_params(arch, params, tc, F):
    """Same failure behaviour as ``lasagne.layers.set_all_param_values``
    (separate_dsd.py:250): ValueError on a count or shape mismatch."""
    shapes = arch.param_shapes(tc, F)
    if len(params) != len(shapes):
        raise ValueError("mismatch: got %d values to set %d parameters" % (len(params), len(shapes)))
    for p, s in zip(params, shapes):
        if tuple(np.shape(p)) != tuple(s):
            raise ValueError("mismatch: parameter has shape %r but value to set has shape %r"
                             % (tuple(s), tuple(np.shape(p))))
