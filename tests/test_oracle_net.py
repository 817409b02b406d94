"""Self-consistency of the network oracle (PARITY UNPINNED: Theano/Lasagne are not available,
see oracle/__init__.py).  Two independent formulations of every InverseLayer must agree, and
the documented edge semantics (mask conventions, tie routing, uncovered columns) must hold."""
import numpy as np
import pytest
import torch

from deepconvsep_amd.arch import ARCHS, check_params
from deepconvsep_amd.synth import synth_params
from oracle import net_ref

CASES = [("dsd", 30, 513), ("dsd", 30, 65), ("ikala", 30, 513), ("bach10", 30, 257), ("bach10_si", 30, 129)]


def _input(arch, B, tc, F, seed=3):
    rs = np.random.RandomState(seed)
    C = ARCHS[arch].C
    return (0.3 * rs.uniform(0, 2, (B, C, tc, F)).astype(np.float32)).astype(np.float64)


@pytest.mark.parametrize("arch,tc,F", CASES)
def test_autograd_vjp_equals_explicit_transpose(arch, tc, F):
    params = synth_params(arch, tc, F, seed=5)
    check_params(ARCHS[arch], params, tc, F)
    assert [tuple(p.shape) for p in params] == [tuple(s) for s in net_ref.SPECS[arch].param_shapes(tc, F)]
    x = _input(arch, 2, tc, F)
    a = net_ref.forward(arch, params, x, inverse='autograd').detach().numpy()
    b = net_ref.forward(arch, params, x, inverse='explicit').detach().numpy()
    assert a.shape == (2, len(ARCHS[arch].branch_fc) * ARCHS[arch].C, tc, F)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    assert (a >= 0).all()


def test_param_counts_match_survey():
    assert len(ARCHS["dsd"].param_shapes(30, 513)) == 15
    assert len(ARCHS["ikala"].param_shapes(30, 513)) == 13
    assert len(ARCHS["bach10"].param_shapes(30, 2049)) == 17
    assert ARCHS["bach10_si"].param_shapes(30, 2049)[-1] == (16,)
    assert ARCHS["dsd"].param_shapes(30, 513)[6] == (800, 128)
    assert ARCHS["ikala"].param_shapes(30, 513)[6] == (13230, 256)
    assert ARCHS["ikala_nopool"].param_shapes(30, 513)[6] == (90090, 256)      # the trainer's graph
    from deepconvsep_amd.arch import resolve
    shapes = ARCHS["ikala_nopool"].param_shapes(30, 150)
    fake = [np.zeros(s_, dtype=np.float32) for s_ in shapes]
    assert resolve("ikala", fake, 30, 150).name == "ikala_nopool"           # picked from fc.W's rows
    assert resolve("ikala", [np.zeros(s_, dtype=np.float32) for s_ in ARCHS["ikala"].param_shapes(30, 270)], 30, 270).name == "ikala"
    assert ARCHS["bach10"].param_shapes(30, 2049)[6] == (166650, 256)
    # SURVEY 8a-4': 11.78 MFLOP per DSD tile (aliased branch computed once)
    assert abs(ARCHS["dsd"].flops_per_tile(30, 513) / 1e6 - 11.78) < 0.02
    assert abs(ARCHS["dsd"].flops_per_tile(30, 1025) / 1e6 - 17.92) < 0.02


def test_dsd_fourth_channel_reuses_second_branch():
    params = synth_params("dsd", 30, 65, seed=7)
    x = _input("dsd", 1, 30, 65)
    p = net_ref.forward("dsd", params, x, inverse='explicit')
    pre1 = p[:, 1] - 0  # relu(o1 + b[1]);  channel 3 = relu(o1 + b[3])
    b = params[-1]
    o1 = None
    # recover o1 where channel 1 is active, then check channel 3 there
    act = (p[:, 1] > 0) & (p[:, 3] > 0)
    o1 = p[:, 1][act] - float(b[1])
    np.testing.assert_allclose((o1 + float(b[3])).numpy(), p[:, 3][act].numpy(), atol=1e-12)
    assert pre1.shape == p[:, 3].shape


def test_mask_conventions_on_silence():
    # all-zero network output: convention A -> masks 1/S, convention B -> 0  
    x = np.full((1, 1, 2, 3), 0.7)
    p = torch.zeros((1, 4, 2, 3), dtype=torch.float64)
    a = net_ref.soft_mask("dsd", p, x, eps_mode='A')
    b = net_ref.soft_mask("dsd", p, x, eps_mode='B')
    for m in a:
        np.testing.assert_allclose(m.numpy(), 0.7 / 4)
    for m in b:
        np.testing.assert_allclose(m.numpy(), 0.0)
    # masks of a non-degenerate output sum to the mixture
    rs = np.random.RandomState(0)
    p = torch.as_tensor(rs.uniform(0, 1, (1, 4, 2, 3)))
    np.testing.assert_allclose(sum(m.numpy() for m in net_ref.soft_mask("dsd", p, x, 'A')), x, atol=1e-15)


def test_pool_tie_modes_differ_only_on_ties():
    params = synth_params("ikala", 30, 513, seed=9)
    x = _input("ikala", 1, 30, 513)
    x[0, 0, 5:9] = 0.0  # digital silence rows: conv1b output is constant -> every pooling window ties
    all_ = net_ref.forward("ikala", params, x, tie_mode='all', inverse='explicit').numpy()
    first = net_ref.forward("ikala", params, x, tie_mode='first', inverse='explicit').numpy()
    rows = np.zeros(30, bool)
    rows[5:9] = True
    np.testing.assert_allclose(all_[:, :, ~rows], first[:, :, ~rows], atol=1e-12)
    assert np.abs(all_[:, :, rows] - first[:, :, rows]).max() > 1e-6


def test_uncovered_columns_only_see_the_bias():
    # bach10 conv1: (F-30) % 4 trailing columns are not covered by any filter position (SURVEY 8a-4')
    F = 257  # (257-30) % 4 = 3
    params = synth_params("bach10", 30, F, seed=11)
    x = _input("bach10", 1, 30, F)
    p = net_ref.forward("bach10", params, x, inverse='explicit').numpy()
    for c in range(4):
        np.testing.assert_allclose(p[0, c, :, F - 3:], max(float(params[-1][c]), 0.0), atol=1e-15)


def test_set_all_param_values_failure_modes():
    params = synth_params("dsd", 30, 513)
    with pytest.raises(ValueError):
        check_params(ARCHS["dsd"], params[:-1], 30, 513)
    bad = list(params)
    bad[6] = bad[6][:, :100]
    with pytest.raises(ValueError):
        check_params(ARCHS["dsd"], bad, 30, 513)
    with pytest.raises(ValueError):
        net_ref.forward("dsd", bad, _input("dsd", 1, 30, 513))


def test_stereo_ild_graph_autograd_equals_explicit_and_masks_per_channel():
    """The stereo trainer's graph (examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py:66-115): VJP by autograd vs explicit
    transposed convolutions, 17 parameter arrays, output channel s*2 + c, and the per-input-channel masks of
    :176-180 summing to (almost) the input where any source is active."""
    import numpy as np
    from oracle import net_ref
    spec = net_ref.SPECS['dsd_ild']
    tc, F = 30, 65
    rs = np.random.RandomState(0)
    shapes = spec.param_shapes(tc, F)
    assert len(shapes) == 17 and shapes[0] == (50, 2, 1, F) and shapes[6] == (800, 256) and shapes[-1] == (8,)
    params = [rs.uniform(-0.1, 0.1, s) for s in shapes]
    x = np.abs(rs.randn(2, 2, tc, F))
    a = net_ref.forward('dsd_ild', params, x, inverse='autograd').detach().numpy()
    b = net_ref.forward('dsd_ild', params, x, inverse='explicit').numpy()
    assert a.shape == (2, 8, tc, F) and np.max(np.abs(a - b)) < 1e-12
    out = net_ref.predict_ild(params, x, inverse='explicit')
    assert len(out) == 2 and out[0].shape == (2, 4, tc, F)
    for j in range(2):
        den = b[:, j::2].sum(axis=1)
        tot = out[j].sum(axis=1)
        ok = den > 1e-6
        assert np.max(np.abs(tot[ok] - x[:, j][ok])) < 1e-6


# ------------------------------------------------------------------------------------------------ reference graphs
# tests/golden/net_*.npz: the reference's own build_ca source and mask expressions executed on the NumPy Lasagne
# stand-in (oracle/lasagne_np.py, oracle/ref_exec.py:build_network) -- see make_golden.py:networks().
from oracle import cases, ref_exec  # noqa: E402

NET_NAMES = [c[0] for c in cases.NET_CASES]


def _case(golden, name):
    g = golden(name)
    arch, F, seed, kind = str(g["arch"]), int(g["F"]), int(g["seed"]), str(g["kind"])
    params = cases.case_params(arch, 30, F, seed, kind, g["out_bias"] if kind != "glorot" else None)
    assert len(params) == int(g["n_params"])
    return g, arch, params


@pytest.mark.parametrize("name", NET_NAMES)
def test_net_ref_matches_the_reference_graph_fixtures(golden, name):
    """oracle.net_ref (torch float64, both InverseLayer formulations) against the outputs of the reference's own
    graph-construction code on the Lasagne stand-in: network output, tie routing, masks -- 1e-12."""
    g, arch, params = _case(golden, name)
    x = g["x"].astype(np.float64)
    for how in ("explicit", "autograd"):
        p = net_ref.forward(arch, params, x, inverse=how).detach().numpy()
        assert p.shape == g["p"].shape
        assert np.max(np.abs(p - g["p"])) < 1e-12
    assert abs(float((g["p"] == 0).mean()) - float(g["zero_fraction"])) < 1e-12
    if str(g["kind"]) == "sparse":
        live = g["p"][np.abs(x).sum(axis=(1, 2, 3)) > 0]
        assert (live == 0).mean() >= 0.9            # trained-like statistics: >= 90 % of the outputs are exactly zero
    if "p_tie_first" in g.files:
        p1 = net_ref.forward(arch, params, x, tie_mode='first', inverse='explicit').numpy()
        assert np.max(np.abs(p1 - g["p_tie_first"])) < 1e-12
        assert np.max(np.abs(g["p_tie_first"] - g["p"])) > 1e-6      # the silent rows do tie
    if "masked" in g.files:
        m = net_ref.predict(arch, params, x, inverse='explicit')
        assert len(m) == g["masked"].shape[0]
        for a, b in zip(m, g["masked"]):
            assert np.max(np.abs(a - b)) < 1e-12
    if "masked_sum" in g.files:      # the score-informed TRAINERS' mask expressions: x the sum of the input channels
        m = net_ref.predict(arch, params, x, inverse='explicit', mixture='sum')
        for a, b in zip(m, g["masked_sum"]):
            assert np.max(np.abs(a - b)) < 1e-12
        assert np.max(np.abs(g["masked_sum"] - g["masked"])) > 1e-3      # the two semantics are different functions


@pytest.mark.skipif(not ref_exec.available(), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("arch,C,F", [("dsd", 1, 513), ("dsd", 1, 1025), ("hiphop", 1, 513), ("ikala", 1, 513),
                                      ("ikala", 1, 1025), ("ikala_nopool", 1, 513), ("bach10", 1, 2049), ("bach10_si", 4, 2049),
                                      ("bach10_si1", 4, 2049), ("dsd_ild", 2, 513)])
def test_parameter_order_comes_from_the_reference_graph(arch, C, F):
    """``get_all_params`` of the reference's own build_ca at the real sizes == the shape list the HIP model checks
    (deepconvsep_amd.arch) and the one the torch oracle uses -- count, order and shapes (13 / 15 / 17 arrays)."""
    shapes = [tuple(s) for s in ref_exec.network_param_shapes(arch, 32, C, 30, F)]
    a = "dsd" if arch == "hiphop" else arch
    assert shapes == [tuple(s) for s in ARCHS[a].param_shapes(30, F)]
    assert shapes == [tuple(s) for s in net_ref.SPECS[a].param_shapes(30, F)]


@pytest.mark.skipif(not ref_exec.available(), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("name", ["net_dsd_f33_tiny", "net_ikala_f270_sparse", "net_bach10si_f129_sparse"])
def test_fixtures_regenerate_bit_for_bit(golden, name):
    g, arch, params = _case(golden, name)
    p = ref_exec.network_output(arch, params, g["x"])
    assert np.array_equal(p, g["p"])
    assert np.array_equal(np.stack(ref_exec.mask_sources(arch, p, g["x"])), g["masked"])
    x = g["x"]
    fresh = cases.calibrate(arch, 30, int(g["F"]), int(g["seed"]), str(g["kind"]), x)
    assert np.array_equal(fresh[-1], g["out_bias"])


def test_single_branch_score_informed_graph_is_the_live_part_of_the_17_array_graph():
    """trainCNNrwc_samp.py:195-235 (11 arrays) against separate_bach10.py:388-447 (17 arrays): with the first ten arrays and
    the first four output biases shared, the single-branch graph's four channels ARE channels 0..3 of the big graph -- the
    only ones predict_function2 reads (:475-488).  `arch.live_params` makes that cut; `arch.resolve` tells the two .pkl
    layouts apart by their length."""
    from deepconvsep_amd.arch import live_params, resolve
    tc, F = 30, 129
    params = synth_params("bach10_si", tc, F, seed=8)
    a1, live = live_params(ARCHS["bach10_si"], params)
    assert a1.name == "bach10_si1" and len(live) == 11 and live[-1].shape == (4,)
    check_params(a1, live, tc, F)
    assert [tuple(p.shape) for p in live] == [tuple(s_) for s_ in net_ref.SPECS["bach10_si1"].param_shapes(tc, F)]
    assert resolve("bach10_si", live, tc, F).name == "bach10_si1" and resolve("bach10_si", params, tc, F).name == "bach10_si"
    assert live_params(ARCHS["bach10"], synth_params("bach10", tc, F, seed=8))[0].name == "bach10"     # nothing to cut elsewhere
    x = _input("bach10_si", 2, tc, F)
    big = net_ref.forward("bach10_si", params, x, inverse='explicit').numpy()
    one = net_ref.forward("bach10_si1", live, x, inverse='explicit').numpy()
    assert one.shape == (2, 4, tc, F) and np.array_equal(one, big[:, :4])
    for mixture in ("ch0", "sum"):
        for a, b in zip(net_ref.predict("bach10_si", params, x, inverse='explicit', mixture=mixture),
                        net_ref.predict("bach10_si1", live, x, inverse='explicit', mixture=mixture)):
            assert np.array_equal(a, b)
    assert ARCHS["bach10_si1"].flops_per_tile(tc, F) == ARCHS["bach10_si"].flops_per_tile(tc, F, live_only=True)


@pytest.mark.skipif(not ref_exec.available(), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("seed", range(4))
def test_single_branch_graph_and_trainer_masks_equal_the_executed_reference(seed):
    """Seeded draws of the single-branch graph: the reference's own build_ca of trainCNNrwc_samp.py and the TRAINERS' mask
    lines (trainCNNrwc_samp.py:288-305, trainCNNrwc.py:246-263: masks x channel sum) executed, against net_ref."""
    rs = np.random.RandomState(500 + seed)
    tc, F, B = 30, 129, int(rs.randint(1, 3))
    params = synth_params("bach10_si1", tc, F, seed=int(rs.randint(1 << 30)), gain=float(rs.uniform(0.6, 2.0)))
    x = (0.3 * rs.uniform(0, 3, (B, 4, tc, F)) * (rs.uniform(size=(B, 4, tc, F)) < 0.6)).astype(np.float32)
    y = cases.pre_bias_output("bach10_si1", params, x)
    params[-1] = np.array([-np.quantile(y[:, c], 0.7) for c in range(4)], dtype=np.float32)
    p_ref = ref_exec.network_output("bach10_si1", params, x)
    p = net_ref.forward("bach10_si1", params, x.astype(np.float64), inverse='explicit').numpy()
    assert p.shape == p_ref.shape == (B, 4, tc, F) and np.max(np.abs(p - p_ref)) < 1e-11
    for which, mixture in (("bach10_si1", "sum"), ("bach10_si_trainer", "sum"), ("bach10_si", "ch0")):
        m_ref = ref_exec.mask_sources(which, p_ref, x)
        m = net_ref.predict("bach10_si1", params, x.astype(np.float64), inverse='explicit', mixture=mixture)
        for a, b in zip(m, m_ref):
            assert np.max(np.abs(a - b)) < 1e-11


def test_lasagne_standin_layer_order_and_alias():
    """The stand-in's get_all_layers / get_all_params on a hand-built graph with a shared layer and an InverseLayer:
    incoming layers before the layer, each once, W before b."""
    from oracle import lasagne_np as L
    x = np.zeros((2, 1, 6, 8))
    l_in = L.InputLayer((2, 1, 6, 8), input_var=x)
    c1 = L.Conv2DLayer(l_in, 3, (1, 8), nonlinearity=None)
    b1 = L.BiasLayer(c1)
    d = L.DenseLayer(b1, 5)
    d2 = L.DenseLayer(d, 3 * 6)
    r = L.ReshapeLayer(d2, (2, 3, 6, 1))
    inv = L.InverseLayer(r, c1)
    out = L.NonlinearityLayer(L.BiasLayer(L.ConcatLayer([inv, inv], axis=1)))
    names = [type(l).__name__ for l in L.get_all_layers(out)]
    assert names == ["InputLayer", "Conv2DLayer", "BiasLayer", "DenseLayer", "DenseLayer", "ReshapeLayer",
                     "InverseLayer", "ConcatLayer", "BiasLayer", "NonlinearityLayer"]
    assert [p.shape for p in L.get_all_params(out)] == [(3, 1, 1, 8), (3,), (3,), (18, 5), (5,), (5, 18), (18,), (2,)]
    with pytest.raises(ValueError):
        L.set_all_param_values(out, [np.zeros(1)])
    assert L.get_output(out).shape == (2, 2, 6, 8)


@pytest.mark.skipif(not ref_exec.available(), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("seed", range(24))
def test_random_draws_net_ref_equals_the_executed_reference_graph(seed):
    """24 seeded draws (graph, weights of either bias sign and gain 0.5 .. 2.2, dense / few-level / sparse inputs,
    10 .. 97 % of the outputs cut to exact zeros): the torch oracle against the reference's own build_ca source executed on
    the Lasagne stand-in -- network output, both tie routings of the iKala graph, masked sources."""
    arch, F, params, x = cases.random_draw(seed)
    p_ref = ref_exec.network_output(arch, params, x)
    p = net_ref.forward(arch, params, x.astype(np.float64), inverse='explicit').numpy()
    assert p.shape == p_ref.shape
    assert np.max(np.abs(p - p_ref)) < 1e-11
    if arch == "ikala":
        p1 = net_ref.forward(arch, params, x.astype(np.float64), tie_mode='first', inverse='explicit').numpy()
        assert np.max(np.abs(p1 - ref_exec.network_output(arch, params, x, tie_mode="first"))) < 1e-11
    m_ref = ref_exec.mask_sources("ikala" if arch == "ikala_nopool" else arch, p_ref, x)   # the separate script's masks
    m = net_ref.predict(arch, params, x.astype(np.float64), inverse='explicit')
    assert len(m) == len(m_ref)
    for a, b in zip(m, m_ref):
        assert np.max(np.abs(a - b)) < 1e-11
