"""torch-CPU float64 restatement of the reference networks + soft mask (test
infrastructure; PARITY UNPINNED -- see ``oracle/__init__.py``).

Graphs restated (all ``build_ca``):
  dsd / hiphop   ``examples/dsd100/separate_dsd.py:172-236``
  ikala (pool)   ``examples/ikala/separate_ikala.py:172-192``
  bach10         ``examples/bach10/separate_bach10.py:172-229``
  bach10_si      ``examples/bach10_scoreinformed/separate_bach10.py:388-447``
  bach10_si1     ``examples/bach10_scoreinformed/trainCNNrwc_samp.py:195-235`` (the single-branch form: 11 arrays)
  dsd_ild        ``examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py:66-115`` (stereo input, one branch
                 per source, every branch returns both channels: output channel ``s*C + c``)
Mask expressions: ``separate_dsd.py:258-271`` (convention A),
``separate_bach10.py:251-264`` (convention B), ``trainCNN_ILD_DSD100.py:176-189`` (per input
channel, ``predict_ild``).

Third-party semantics encoded here (Lasagne master / Theano 0.9, the versions
``requirements.txt:1-2`` names; neither is vendored in the reference):

* ``Conv2DLayer(pad='valid', nonlinearity=None)``: ``W[out,in,kh,kw]``, bias
  ``b[out]``, ``flip_filters=True`` -> a TRUE convolution, i.e. a
  cross-correlation with ``W[:, :, ::-1, ::-1]``.
* ``BiasLayer``: one more per-channel bias ``b[C]`` (shared over axes 0,2,3).
* ``DenseLayer``: flatten C-order to ``[B, C*H*W]``, ``x @ W[in,out] + b``,
  default nonlinearity = rectify.
* ``MaxPool2DLayer((1,p))``: stride = pool size, ``ignore_border=True``.
* ``InverseLayer(incoming, layer)`` = ``theano.grad(None, wrt=layer.input,
  known_grads={layer.output: incoming})``: the vector-Jacobian product of
  ``layer`` evaluated at the forward activations.  Biases contribute nothing.
  For max-pooling Theano 0.9's CPU ``MaxPoolGrad`` sends the incoming value
  to EVERY position that equals the window maximum (``tie_mode='all'``);
  cuDNN sends it to the first one only (``tie_mode='first'``).
* ``ConcatLayer(axis=1)`` -> ``BiasLayer`` -> ``rectify``.
* Parameter order = ``get_all_params``: conv1.W, conv1.b, conv1b.b, conv2.W,
  conv2.b, conv2b.b, fc.W, fc.b, then (W, b) of every REACHABLE per-source
  dense layer, then the output bias.  In the DSD graph the 4th branch is built
  on ``l_fc12`` (``separate_dsd.py:228``) so ``l_fc14`` is unreachable: 15
  arrays, and output channel 3 repeats branch 2.
"""
import numpy as np
import torch
import torch.nn.functional as Fnn

EPS = 1e-18
RAND = 0.5  # deterministic stand-in for the unseeded uniform draw (separate_dsd.py:256)


class NetSpec(object):
    """Static description of one ``build_ca`` variant."""

    def __init__(self, name, in_channels, conv1, pool_w, conv2, hidden, branch_fc, n_sources,
                 eps_mode):
        self.name = name
        self.C = in_channels
        self.conv1 = conv1          # (filters, kw or 'F', stride_w)
        self.pool_w = pool_w        # None or pool width
        self.conv2 = conv2          # (filters, kh or callable(tc), kw)
        self.hidden = hidden
        self.branch_fc = branch_fc  # per output branch: index of the dense layer it uses
        self.n_fc = max(branch_fc) + 1
        self.S = n_sources          # number of masks / separated sources
        self.eps_mode = eps_mode    # 'A' or 'B'

    def dims(self, tc, F):
        nf1, kw1, sw1 = self.conv1
        kw1 = F if kw1 == 'F' else kw1
        w1 = (F - kw1) // sw1 + 1
        wp = w1 // self.pool_w if self.pool_w else w1
        nf2, kh2, kw2 = self.conv2
        kh2 = kh2(tc) if callable(kh2) else kh2
        h2 = tc - kh2 + 1
        w2 = wp - kw2 + 1
        return dict(nf1=nf1, kw1=kw1, sw1=sw1, w1=w1, wp=wp, nf2=nf2, kh2=kh2, kw2=kw2,
                    h2=h2, w2=w2, flat=nf2 * h2 * w2)

    def param_shapes(self, tc, F):
        d = self.dims(tc, F)
        shapes = [(d['nf1'], self.C, 1, d['kw1']), (d['nf1'],), (d['nf1'],),
                  (d['nf2'], d['nf1'], d['kh2'], d['kw2']), (d['nf2'],), (d['nf2'],),
                  (d['flat'], self.hidden), (self.hidden,)]
        for _ in range(self.n_fc):
            shapes += [(self.hidden, d['flat']), (d['flat'],)]
        shapes.append((len(self.branch_fc) * self.C,))
        return shapes


SPECS = {
    'dsd': NetSpec('dsd', 1, (50, 'F', 1), None, (50, lambda tc: int(tc / 2), 1), 128,
                   [0, 1, 2, 1], 4, 'A'),
    'ikala': NetSpec('ikala', 1, (30, 30, 3), 4, (30, 10, 20), 256, [0, 1], 2, 'A'),
    'ikala_nopool': NetSpec('ikala_nopool', 1, (30, 30, 3), None, (30, 10, 20), 256, [0, 1], 2, 'A'),   # ikala/trainCNN.py:87-118
    'bach10': NetSpec('bach10', 1, (30, 30, 4), None, (30, lambda tc: int(2 * tc / 3), 1), 256,
                      [0, 1, 2, 3], 4, 'B'),
    'bach10_si': NetSpec('bach10_si', 4, (30, 30, 4), None, (30, lambda tc: int(2 * tc / 3), 1),
                         256, [0, 1, 2, 3], 4, 'B'),
    # the single-branch score-informed graph of bach10_scoreinformed/trainCNNrwc_samp.py:195-235: one dense layer back, one
    # pair of InverseLayers, 4 output channels = the only channels predict_function2 reads from the 17-array graph
    'bach10_si1': NetSpec('bach10_si1', 4, (30, 30, 4), None, (30, lambda tc: int(2 * tc / 3), 1),
                          256, [0], 4, 'B'),
    'dsd_ild': NetSpec('dsd_ild', 2, (50, 'F', 1), None, (50, lambda tc: int(tc / 2), 1), 256,
                       [0, 1, 2, 3], 4, 'ILD'),
}

EPS_ILD = 1e-12   # trainCNN_ILD_DSD100.py:155
RAND_ILD = 0.1    # deterministic stand-in for theano_rng.normal(avg=0, std=0.1) (:166): one standard deviation


def _t(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float64))


def _flip(W):
    return torch.flip(W, dims=(2, 3))


def _conv_vjp(g, x_shape, Wc, stride, how):
    """VJP of ``conv2d(x, Wc, stride)`` w.r.t. x, applied to g."""
    if how == 'autograd':
        x = torch.zeros(x_shape, dtype=g.dtype, requires_grad=True)
        y = Fnn.conv2d(x, Wc, stride=stride)
        (gx,) = torch.autograd.grad(y, x, grad_outputs=g)
        return gx
    # explicit transposed convolution, zero padded up to the forward input size
    y = Fnn.conv_transpose2d(g, Wc, stride=stride)
    pad_h = x_shape[2] - y.shape[2]
    pad_w = x_shape[3] - y.shape[3]
    return Fnn.pad(y, (0, pad_w, 0, pad_h))


def _pool_fwd(x, pw):
    W = x.shape[-1]
    wp = W // pw
    xr = x[..., :wp * pw].reshape(x.shape[:-1] + (wp, pw))
    return xr.max(dim=-1).values


def _pool_vjp(g, x, pw, tie_mode):
    """VJP of the (1,pw) max-pool evaluated at x (forward input of the pool)."""
    W = x.shape[-1]
    wp = W // pw
    xr = x[..., :wp * pw].reshape(x.shape[:-1] + (wp, pw))
    m = xr.max(dim=-1, keepdim=True).values
    eq = (xr == m)
    if tie_mode == 'first':
        first = eq.to(torch.int64).argmax(dim=-1, keepdim=True)
        eq = torch.zeros_like(eq).scatter_(-1, first, True)
    elif tie_mode != 'all':
        raise ValueError(tie_mode)
    gx = (eq.to(g.dtype) * g.unsqueeze(-1)).reshape(x.shape[:-1] + (wp * pw,))
    return Fnn.pad(gx, (0, W - wp * pw))


def forward(arch, params, x, tie_mode='all', inverse='autograd'):
    """Network output ``p = rectify(concat(branches) + bias)``: ``[B, nb*C, tc, F]``."""
    spec = SPECS[arch]
    x = _t(x)
    B, C, tc, F = x.shape
    assert C == spec.C
    P = [_t(p) for p in params]
    shapes = spec.param_shapes(tc, F)
    if len(P) != len(shapes):
        raise ValueError("mismatch: got %d values to set %d parameters" % (len(P), len(shapes)))
    for p, s in zip(P, shapes):
        if tuple(p.shape) != tuple(s):
            raise ValueError("mismatch: parameter has shape %r but value to set has shape %r"
                             % (tuple(s), tuple(p.shape)))
    d = spec.dims(tc, F)
    W1, b1, b1b, W2, b2, b2b, Wfc, bfc = P[:8]
    W1c, W2c = _flip(W1), _flip(W2)
    s1 = (1, d['sw1'])

    a1 = Fnn.conv2d(x, W1c, stride=s1) + b1.view(1, -1, 1, 1)
    a1b = a1 + b1b.view(1, -1, 1, 1)
    h = _pool_fwd(a1b, spec.pool_w) if spec.pool_w else a1b
    a2 = Fnn.conv2d(h, W2c) + b2.view(1, -1, 1, 1)
    a2b = a2 + b2b.view(1, -1, 1, 1)
    z = torch.relu(a2b.reshape(B, -1) @ Wfc + bfc)

    dense_out = []
    for i in range(spec.n_fc):
        Wi, bi = P[8 + 2 * i], P[9 + 2 * i]
        dense_out.append(torch.relu(z @ Wi + bi).reshape(a2.shape))
    outs = []
    for fc_idx in spec.branch_fc:
        g = _conv_vjp(dense_out[fc_idx], h.shape, W2c, (1, 1), inverse)
        if spec.pool_w:
            g = _pool_vjp(g, a1b, spec.pool_w, tie_mode)
        outs.append(_conv_vjp(g, x.shape, W1c, s1, inverse))
    y = torch.cat(outs, dim=1) + P[-1].view(1, -1, 1, 1)
    return torch.relu(y)


def soft_mask(arch, p, x, eps_mode=None, mixture='ch0'):
    """Masked magnitudes, list of S tensors ``[B,1,tc,F]``.

    ``mixture``: what the masks multiply -- ``'ch0'`` input channel 0 (every script; separate_bach10.py:485 for the
    score-informed one), ``'sum'`` the sum of the input channels, left to right (the score-informed TRAINERS:
    ``input_var = input_var2[:,0:1] + [:,1:2] + [:,2:3] + [:,3:4]``, trainCNNrwc.py:258-263, trainCNNrwc_samp.py:300-305).

    Convention A (separate_dsd.py:258-271): ``s_i = p_i + eps*r``,
    ``m_i = s_i / sum_j s_j``.  Convention B (separate_bach10.py:251-264):
    ``m_i = p_i / (sum_j p_j + eps*r)``.  The mixture is input channel 0
    (``input_var2[:,0:1]``; ``separate_bach10.py:261``).  Only the first S
    channels of p take part (score-informed: ``prediction2[:,0:4]``).
    """
    spec = SPECS[arch]
    mode = eps_mode or spec.eps_mode
    x = _t(x)
    mix = x[:, 0:1]
    if mixture == 'sum':
        for c in range(1, x.shape[1]):
            mix = mix + x[:, c:c + 1]
    elif mixture != 'ch0':
        raise ValueError(mixture)
    r = EPS * RAND
    ch = [p[:, i:i + 1] for i in range(spec.S)]
    if mode == 'A':
        s = [c + r for c in ch]
        den = s[0]
        for c in s[1:]:
            den = den + c
        return [(c / den) * mix for c in s]
    if mode == 'B':
        den = ch[0]
        for c in ch[1:]:
            den = den + c
        den = den + r
        return [(c / den) * mix for c in ch]
    raise ValueError(mode)


def predict(arch, params, x, tie_mode='all', inverse='autograd', eps_mode=None, mixture='ch0'):
    """``predict_function2`` of the reference: list of S float64 ndarrays
    ``[B,1,tc,F]`` (separate_dsd.py:273)."""
    with torch.no_grad() if inverse != 'autograd' else torch.enable_grad():
        p = forward(arch, params, x, tie_mode=tie_mode, inverse=inverse)
        outs = soft_mask(arch, p.detach(), x, eps_mode=eps_mode, mixture=mixture)
    return [o.detach().numpy() for o in outs]


def predict_ild(params, x, inverse='autograd'):
    """``predict_function`` of the stereo trainer (trainCNN_ILD_DSD100.py:176-189, 231): a list over the input
    channels j of ``[B, S, tc, F]`` arrays,

        mask   = p[:, j::C] / (sum_s p[:, j::C] + eps*r)
        source = mask * x[:, j:j+1] + eps*r

    with eps = 1e-12 and r a draw from N(0, 0.1) in the reference (one standard deviation here)."""
    spec = SPECS['dsd_ild']
    with torch.no_grad() if inverse != 'autograd' else torch.enable_grad():
        p = forward('dsd_ild', params, x, inverse=inverse).detach()
    xt = _t(x)
    e = EPS_ILD * RAND_ILD
    out = []
    for j in range(spec.C):
        pj = p[:, j::spec.C]
        den = pj[:, 0:1]
        for s in range(1, pj.shape[1]):
            den = den + pj[:, s:s + 1]
        out.append(((pj / (den + e)) * xt[:, j:j + 1] + e).numpy())
    return out
