"""The folded encoder maps of round 6 (csrc/net.hip: pack_dsd; csrc/generic.hip: fold_conv2_fc_kernel), restated in NumPy with the
index conventions the packers use, against the layer-by-layer evaluation of oracle/net_ref.py: conv2 + BiasLayer + bottleneck
DenseLayer are ONE affine map of conv2's input because the reference puts no nonlinearity between them
(examples/dsd100/separate_dsd.py:198-206, examples/ikala/separate_ikala.py:181-186).  The GPU tests check the kernels; this pins the
algebra (which taps meet which dense rows, the flip of the filters, the bias term) on shapes small enough for float64 loops."""
import numpy as np
import torch

from oracle import net_ref


def _fold_rows(W2, Wfc, b2, bfc, tc, wp):
    """W2 [nf2, nf1, kh, kw] (Lasagne: true convolution), Wfc [(nf2 * h2 * w2), hidden] in the flattened [nf2, h2, w2] order,
    b2 = conv2 bias + BiasLayer bias.  Returns W2fc [(nf1, tc, wp) flattened, hidden] and the folded bias: the formula of
    fold_conv2_fc_kernel (and of pack_dsd with kw = 1, wp = 1 and rows before channels)."""
    nf2, nf1, kh, kw = W2.shape
    h2, w2 = tc - kh + 1, wp - kw + 1
    hidden = Wfc.shape[1]
    Wf = W2[:, :, ::-1, ::-1]                                # correlation form: out[co][y][x] = sum Wf[co][ci][u][v] in[ci][y + u][x + v]
    F = Wfc.reshape(nf2, h2, w2, hidden)
    out = np.zeros((nf1, tc, wp, hidden))
    for u in range(kh):
        for v in range(kw):
            # input position (r, c) = (y + u, x + v) meets dense row (co, y, x)
            out[:, u:u + h2, v:v + w2, :] += np.einsum('oi,oyxh->iyxh', Wf[:, :, u, v], F)
    c = bfc + np.einsum('o,oyxh->h', b2, F)
    return out.reshape(-1, hidden), c


def _layer_by_layer(x, W2, b2, Wfc, bfc):
    a2 = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(W2[:, :, ::-1, ::-1].copy())).numpy()
    a2 = a2 + b2[None, :, None, None]
    return np.maximum(a2.reshape(x.shape[0], -1) @ Wfc + bfc, 0.0)


def test_folded_conv2_bottleneck_map_equals_the_two_layers():
    rng = np.random.default_rng(5)
    for nf1, nf2, kh, kw, tc, wp, hidden in ((5, 4, 3, 1, 8, 1, 7),      # DSD-like: column filter on a one-column map
                                             (3, 4, 3, 4, 7, 9, 6),      # iKala-like: kh x kw filter
                                             (2, 2, 1, 1, 3, 2, 3)):
        h2, w2 = tc - kh + 1, wp - kw + 1
        W2 = rng.standard_normal((nf2, nf1, kh, kw))
        b2 = rng.standard_normal(nf2)
        Wfc = rng.standard_normal((nf2 * h2 * w2, hidden))
        bfc = rng.standard_normal(hidden)
        x = rng.standard_normal((4, nf1, tc, wp))
        want = _layer_by_layer(x, W2, b2, Wfc, bfc)
        Wf, c = _fold_rows(W2, Wfc, b2, bfc, tc, wp)
        got = np.maximum(x.reshape(4, -1) @ Wf + c, 0.0)
        assert np.max(np.abs(got - want)) < 1e-10 * max(1.0, float(np.max(np.abs(want))))


def test_the_reference_graphs_have_no_nonlinearity_between_conv2_and_the_bottleneck_layer():
    """oracle/net_ref.py (pinned against the reference's own build_ca source, tests/test_oracle_net.py) evaluates
    z = relu((conv2(h) + b2 + b2b).reshape @ Wfc + bfc) for every graph: the fold is valid for all of them."""
    import inspect
    src = inspect.getsource(net_ref.forward)
    assert "a2 = Fnn.conv2d(h, W2c) + b2.view(1, -1, 1, 1)" in src
    assert "z = torch.relu(a2b.reshape(B, -1) @ Wfc + bfc)" in src
