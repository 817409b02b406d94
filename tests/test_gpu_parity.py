"""Parity of the HIP path (through the C ABI) against the committed golden vectors of the
reference's own code and against the CPU oracle on seeded inputs.  Run on the MI355X box:

    python -m pytest tests -m gpu -x -q

Tolerances: framing / indexing bit-exact; float64 kernels 1e-11; float32 kernels 1e-4 absolute
on (scaled) magnitudes per masked bin (BASELINE.json north_star), PCM 1e-4.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import deepconvsep_amd as dcs  # noqa: E402
from deepconvsep_amd import _lib  # noqa: E402
from deepconvsep_amd.arch import ARCHS, EPS_A, EPS_B, TILER_LIBRARY, TILER_SCRIPT  # noqa: E402
from deepconvsep_amd.runtime import Network, StftPlan, default_context, overlap_add, tile  # noqa: E402
from deepconvsep_amd.synth import synth_audio, synth_params  # noqa: E402
from oracle import net_ref, pipeline, stft_np, tiling_np  # noqa: E402

STFT_CASES = ["stft_n1024_hann", "stft_n2048_hann", "stft_n4096_bh", "stft_n1024_hop256_hann",
              "stft_n512_hop200_hann"]
TILE_CASES = ["tile_t83_tc30_ov25", "tile_t61_tc30_ov20", "tile_t30_tc30_ov25", "tile_t131_tc30_ov25"]
OLA_CASES = ["ola_n11_tc30_ov25_s4", "ola_n7_tc30_ov20_s2", "ola_n1_tc30_ov25_s4", "ola_n37_tc30_ov25_s4"]


@pytest.fixture(scope="module")
def ctx():
    return default_context()


def _tt(g, precision):
    win = g["window"]
    return dcs.transformFFT(frameSize=int(g["frame"]), hopSize=int(g["hop"]), window=lambda n: win,
                            precision=precision)


# ------------------------------------------------------------------------------------------ STFT
@pytest.mark.parametrize("name", STFT_CASES)
@pytest.mark.parametrize("precision,tol", [("float64", 1e-11), ("float32", 2e-5)])
def test_compute_file_matches_reference(golden, name, precision, tol):
    g = golden(name)
    mag, ph = _tt(g, precision).compute_file(g["audio"], phase=True)
    assert mag.shape == g["mag"].shape and mag.dtype == np.float64          # framing: bit-exact
    assert np.max(np.abs(mag - g["mag"])) < tol
    # raw phase is ill-conditioned where mag ~ 0 and at +-pi: compare mag * exp(j phase)
    z = mag * np.exp(1j * ph)
    zr = g["mag"] * np.exp(1j * g["phase"])
    assert np.max(np.abs(z - zr)) < 2 * tol
    only_mag = _tt(g, precision).compute_file(g["audio"], phase=False)
    assert np.array_equal(only_mag, mag)


@pytest.mark.parametrize("name", STFT_CASES)
@pytest.mark.parametrize("precision,tol", [("float64", 1e-11), ("float32", 1e-5)])
def test_compute_inverse_matches_reference(golden, name, precision, tol):
    g = golden(name)
    back = _tt(g, precision).compute_inverse(g["mag"], g["phase"])
    assert back.shape == g["inverse"].shape
    L = g["audio"].size
    assert np.max(np.abs(back[:L] - g["inverse"][:L])) < tol
    # past the end of the signal (samples every caller truncates, separate_dsd.py:305) the reference
    # divides w*frame by a vanishing sum of w*w: rounding noise is amplified by 1/w there
    assert np.max(np.abs(back[L:] - g["inverse"][L:])) < (tol if precision == "float64" else 5e-3)


@pytest.mark.parametrize("N,hop", [(1024, 512), (2048, 512), (4096, 512), (256, 64), (8192, 2048)])
def test_round_trip_property_full_size(N, hop):
    """README.md:60-70 contract at BASELINE size (10 s): compute_inverse(compute_file(x)) == x."""
    audio = synth_audio(441000, seed=4)
    tt = dcs.transformFFT(frameSize=N, hopSize=hop, precision="float64")
    mag, ph = tt.compute_file(audio, phase=True)
    assert mag.shape == (stft_np.frame_count(audio.size, hop), N // 2 + 1)
    back = tt.compute_inverse(mag, ph)
    assert back.size == stft_np.inverse_length(mag.shape[0], hop, N)
    assert np.max(np.abs(back[:audio.size] - audio)) < 1e-11
    tt32 = dcs.transformFFT(frameSize=N, hopSize=hop, precision="float32")
    m32, p32 = tt32.compute_file(audio, phase=True)
    assert np.max(np.abs(m32 - mag)) < 3e-5
    assert np.max(np.abs(tt32.compute_inverse(m32, p32)[:audio.size] - audio)) < 2e-5


def test_stft_linearity_and_silence():
    tt = dcs.transformFFT(frameSize=2048, hopSize=512, precision="float64")

    def spec(x):
        m, p = tt.compute_file(x, True)
        return m * np.exp(1j * p)

    a, b = synth_audio(30000, seed=1, silence=False), synth_audio(30000, seed=2, silence=False)
    assert np.max(np.abs(spec(2.0 * a - 0.5 * b) - (2.0 * spec(a) - 0.5 * spec(b)))) < 1e-11
    m0, p0 = tt.compute_file(np.zeros(5000), True)
    assert not m0.any() and not (m0 * np.exp(1j * p0)).any()
    assert not tt.compute_inverse(m0, p0).any()


def test_compute_transform_writes_reference_format(tmp_path):
    audio = np.stack([synth_audio(9000, seed=5), synth_audio(9000, seed=6)], axis=1)
    tt = dcs.transformFFT(frameSize=1024, hopSize=512, suffix="x")
    mags = tt.compute_transform(audio, phase=False, save=False)
    assert mags.shape == (2, stft_np.frame_count(9000, 512), 513)
    np.testing.assert_allclose(mags[1], stft_np.compute_file(audio[:, 1], frameSize=1024, hopSize=512), atol=1e-11)
    out = str(tmp_path / "song.data")
    assert tt.compute_transform(audio, out_path=out, phase=True, save=True) is None
    back = np.fromfile(out.replace(".data", "_x_m_.data")).reshape(tt.get_shape(out.replace(".data", "_x_m_.shape")))
    assert np.array_equal(back, mags)


@pytest.mark.parametrize("N,hop,precision,cols,L", [(1024, 512, 'float64', 5, 30011), (4096, 512, 'float64', 3, 20000),
                                                     (2048, 512, 'float32', 5, 30011), (512, 200, 'float64', 2, 7001)])
def test_compute_transform_all_columns_in_one_launch(N, hop, precision, cols, L, tmp_path):
    """transformFFT.compute_transform (transform.py:80-131): every column of audio[t, i] through ONE STFT launch
    (dcs_stft_forward_f64_clips / _f32_clips) equals compute_file column by column -- the same kernel per frame: bit for bit --
    and the oracle (float64 1e-11, float32 2e-5); magnitudes and phases land in the .data / .shape files in the
    reference's layout [i, T, F]; the reference's failure modes for a 1-D array and for zero columns."""
    audio = np.stack([synth_audio(L, seed=80 + c) for c in range(cols)], axis=1)
    audio[:, 1] = 0.0                                                # a silent column
    tt = dcs.transformFFT(frameSize=N, hopSize=hop, suffix="s", precision=precision)
    mags, phs = tt.compute_transform(audio, phase=True, save=False)
    T, F = stft_np.frame_count(L, hop), N // 2 + 1
    assert mags.shape == phs.shape == (cols, T, F) and mags.dtype == phs.dtype == np.float64
    tol = 1e-11 if precision == 'float64' else 2e-5
    for c in range(cols):
        m1, p1 = tt.compute_file(audio[:, c], phase=True)
        assert np.array_equal(mags[c], m1) and np.array_equal(phs[c], p1)
        m0, p0 = stft_np.compute_file(audio[:, c], phase=True, frameSize=N, hopSize=hop)
        assert np.max(np.abs(mags[c] - m0)) < tol
        z = mags[c] * np.exp(1j * phs[c]) - m0 * np.exp(1j * p0)      # phases compare through the spectrum (angle wraps)
        assert np.max(np.abs(z)) < 2 * tol
    assert not mags[1].any()
    out = str(tmp_path / "song.data")
    assert tt.compute_transform(audio, out_path=out, phase=True, save=True) is None
    for tag, want in (("_s_m_", mags), ("_s_p_", phs)):
        shape = tt.get_shape(out.replace(".data", tag + ".shape"))
        assert shape == (cols, T, F)
        assert np.array_equal(np.fromfile(out.replace(".data", tag + ".data")).reshape(shape), want)
    only = tt.compute_transform(audio, phase=False, save=False)
    assert np.array_equal(only, mags)
    with pytest.raises(IndexError):
        tt.compute_transform(audio[:, 0])
    with pytest.raises(UnboundLocalError):
        tt.compute_transform(audio[:, :0])


# ------------------------------------------------------------------------------------------ tiling
@pytest.mark.parametrize("name", TILE_CASES)
def test_generate_overlapadd_matches_reference(golden, name):
    g = golden(name)
    tc, ov, B = int(g["tc"]), int(g["ov"]), int(g["B"])
    mag = g["mag"]
    fb, n = dcs.generate_overlapadd(mag, input_size=mag.shape[-1], time_context=tc, overlap=ov, batch_size=B)
    assert n == int(g["script_n"]) and fb.dtype == np.float64
    assert fb.shape[0] == int(np.ceil(float(n) / B))
    assert np.array_equal(fb.reshape((-1,) + fb.shape[2:])[:n], g["script_tiles"])           # copies: bit-exact
    fb, n = dcs.generate_overlapadd(mag, mag.shape[-1], tc, ov, B, tiler='library')
    assert n == int(g["library_n"]) and np.array_equal(fb, g["library_batches"])
    fb, n = dcs.generate_overlapadd(g["mag3"], mag.shape[-1], tc, ov, B, tiler='library')
    assert n == int(g["library3_n"]) and np.array_equal(fb, g["library3_batches"])


@pytest.mark.parametrize("name", OLA_CASES)
def test_overlapadd_matches_reference(golden, name):
    g = golden(name)
    n, ov = int(g["n"]), int(g["ov"])
    sep = dcs.overlapadd_multi(g["out"], None, n, overlap=ov)
    assert sep.shape == g["sep"].shape
    assert np.max(np.abs(sep - g["sep"])) < 5e-7    # float32 blend vs the reference's float64
    if "sep1" in g.files:
        s1, s2 = dcs.overlapadd(g["out"], None, n, overlap=ov)
        assert np.max(np.abs(s1 - g["sep1"])) < 5e-7 and np.max(np.abs(s2 - g["sep2"])) < 5e-7


def test_overlapadd_of_identical_tiles_is_identity():
    """Property: if every tile holds the same frames of one spectrogram, the cross-fade returns it."""
    rs = np.random.RandomState(5)
    T, F, tc, ov = 240, 33, 30, 25
    spec = rs.uniform(0, 1, (T, F)).astype(np.float32)
    fb, n = dcs.generate_overlapadd(spec, F, tc, ov, 32)
    out = np.stack([fb, 2 * fb], axis=1)[:, :, :, :, :, :]          # [nb, S=2, B, 1, tc, F]
    sep = dcs.overlapadd_multi(out, None, n, overlap=ov)
    covered = (n - 1) * (tc - ov) + tc
    assert np.max(np.abs(sep[0, :covered] - spec[:covered])) < 1e-6
    assert np.max(np.abs(sep[1, :covered] - 2 * spec[:covered])) < 1e-6
    assert not sep[:, covered:].any()


# ------------------------------------------------------------------------------------------ network
def _tiles(arch, n, tc, F, seed, silent_rows=True):
    rs = np.random.RandomState(seed)
    C = ARCHS[arch].C
    x = (0.3 * rs.uniform(0, 3, (n, C, tc, F)).astype(np.float32))
    if silent_rows and n > 1:
        x[1, :, 4:9] = 0.0      # digital silence inside a tile
        x[n - 1] = 0.0          # a fully silent tile: every p_i can be 0 -> mask edge case
    return x.astype(np.float32)


@pytest.mark.parametrize("F,n", [(513, 5), (1025, 3), (513, 37), (65, 2)])
def test_dsd_predict_function_matches_oracle(F, n):
    tc = 30
    params = synth_params("dsd", tc, F, seed=2)
    x = _tiles("dsd", n, tc, F, seed=8)
    pf = dcs.PredictFunction("dsd", params, tc, F)
    got = pf(x)
    want = net_ref.predict("dsd", params, x.astype(np.float64), inverse='explicit')
    assert len(got) == 4
    for g_, w_ in zip(got, want):
        assert g_.shape == w_.shape == (n, 1, tc, F)
        assert np.max(np.abs(g_ - w_)) < 1e-4
    # masks partition the mixture (convention A): sum of sources == input
    assert np.max(np.abs(sum(got) - x.astype(np.float64))) < 1e-5


def test_dsd_raw_output_and_mask_conventions():
    tc, F, n = 30, 513, 3
    params = synth_params("dsd", tc, F, seed=3)
    x = _tiles("dsd", n, tc, F, seed=9)
    ctx = default_context()
    net = Network(ctx, "dsd", params, tc, F)
    p = net.forward_raw(ctx.to_device(x, np.float32)).cpu().numpy()
    want = net_ref.forward("dsd", params, x.astype(np.float64), inverse='explicit').numpy()
    assert p.shape == want.shape == (n, 4, tc, F)
    assert np.max(np.abs(p - want)) < 1e-4
    for mode, name in ((EPS_A, 'A'), (EPS_B, 'B')):
        got = net.forward_masked(ctx.to_device(x, np.float32), eps_mode=mode).cpu().numpy()
        ref = net_ref.predict("dsd", params, x.astype(np.float64), inverse='explicit', eps_mode=name)
        for s in range(4):
            assert np.max(np.abs(got[s] - ref[s][:, 0])) < 1e-4


def test_all_zero_network_output_gives_uniform_masks():
    """p == 0 everywhere: convention A -> 1/S of the mixture, convention B -> 0."""
    tc, F, n = 30, 513, 2
    params = [np.zeros_like(p) for p in synth_params("dsd", tc, F, seed=4)]
    x = _tiles("dsd", n, tc, F, seed=10, silent_rows=False)
    ctx = default_context()
    net = Network(ctx, "dsd", params, tc, F)
    a = net.forward_masked(ctx.to_device(x, np.float32), eps_mode=EPS_A).cpu().numpy()
    b = net.forward_masked(ctx.to_device(x, np.float32), eps_mode=EPS_B).cpu().numpy()
    for s in range(4):
        np.testing.assert_allclose(a[s], 0.25 * x[:, 0], rtol=1e-6)
        assert not b[s].any()


def test_param_mismatch_raises_like_set_all_param_values():
    params = synth_params("dsd", 30, 513)
    with pytest.raises(ValueError):
        dcs.PredictFunction("dsd", params[:-1], 30, 513)
    with pytest.raises(ValueError):
        dcs.PredictFunction("dsd", params, 30, 1025)


# ------------------------------------------------------------------------------------------ whole path
@pytest.mark.parametrize("N,tiler", [(1024, 'script'), (2048, 'script'), (1024, 'library')])
def test_dsd_separation_matches_oracle(N, tiler):
    F = N // 2 + 1
    params = synth_params("dsd", 30, F, seed=2)
    audio = synth_audio(44100, seed=0)          # 1 s: 89 frames, 12 tiles (script) / 13 (library)
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, 512, np.hanning, tiler=tiler)
    want, mm, mag, ph = pipeline.separate("dsd", params, audio, 0.3, 30, 25, 32, N, 512, np.hanning,
                                          tiler=tiling_np.SCRIPT if tiler == 'script' else tiling_np.LIBRARY,
                                          return_spectra=True)
    ctx = default_context()
    a = ctx.to_device(audio, np.float32)
    s_d, m_d, p_d = sep.net.separate_spectra(sep.plan, a, 25, sep.tiler, 0.3)
    assert tuple(s_d.shape) == mm.shape                     # framing / tile bookkeeping: exact
    assert np.max(np.abs(m_d.cpu().numpy() * 0.3 - mag)) < 1e-5
    assert np.max(np.abs(s_d.cpu().numpy() - mm)) < 1e-4    # per masked bin
    got = sep.separate(audio)
    assert got.shape == want.shape == (4, audio.size)
    assert np.max(np.abs(got - want)) < 1e-4
    step = sep.separate_stepwise(audio)                     # stage-by-stage operators agree with the fused path
    assert np.max(np.abs(step - got)) < 2e-5
    # int16 files: at most one LSB apart from the float64 reference path (truncation, separate_dsd.py:309)
    d = np.abs((got * 32767).astype('int16').astype(int) - (want * 32767).astype('int16').astype(int))
    assert d.max() <= 2


def test_long_clip_matches_oracle_on_interior_segments():
    """BASELINE-size check of the many-rows kernel variants (4096 tiles, 3 min 58 s, N=2048): tiles only
    see local audio, so the separated spectrogram / PCM of a segment cut on the tile grid equals the
    whole-clip result away from the segment's edges.  The oracle runs on three 1.5 s segments."""
    N, hop, F, tc, ov = 2048, 512, 1025, 30, 25
    st = tc - ov
    params = synth_params("dsd", tc, F, seed=2)
    n_tiles = 4096
    L = (tc + 1 + (n_tiles - 1) * st - 2) * hop
    audio = synth_audio(L, seed=7)
    sep = dcs.Separator("dsd", params, 0.3, tc, ov, 32, F, N, hop, np.hanning)
    ctx = default_context()
    s_d, _, _ = sep.net.separate_spectra(sep.plan, ctx.to_device(audio, np.float32), ov, sep.tiler, 0.3)
    assert tuple(s_d.shape) == (4, _lib.frame_count(L, hop), F)
    pcm = sep.separate(audio)
    seg_frames = 130
    for q in (0, 1531, 4040):            # segment starts, in tiles
        a0 = q * st * hop
        seg = audio[a0:a0 + (seg_frames - 2) * hop]
        want_pcm, mm, _, _ = pipeline.separate("dsd", params, seg, 0.3, tc, ov, 32, N, hop, np.hanning,
                                               return_spectra=True)
        lo = 0 if q == 0 else tc + 4                     # frames whose covering tiles and samples are all inside
        hi = mm.shape[1] - tc - 8
        got = s_d[:, q * st + lo:q * st + hi].cpu().numpy()
        assert np.max(np.abs(got - mm[:, lo:hi])) < 1e-4
        s_lo, s_hi = lo * hop + N, (hi - 4) * hop
        assert np.max(np.abs(pcm[:, a0 + s_lo:a0 + s_hi] - want_pcm[:, s_lo:s_hi])) < 1e-4


def test_separation_edge_cases():
    F = 513
    params = synth_params("dsd", 30, F, seed=2)
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, 1024, 512)
    # too short for a single tile: the reference dies in overlapadd_multi; we raise ValueError
    with pytest.raises(ValueError):
        sep.separate(synth_audio(512 * 20, seed=1))
    # pure digital silence -> silence (masks 1/4 of zero)
    out = sep.separate(np.zeros(44100))
    assert out.shape == (4, 44100) and not out.any()
    # the four sources add up to the part of the mixture the tiles cover (masks sum to 1)
    audio = synth_audio(66150, seed=7)
    got = sep.separate(audio)
    n = _lib.tile_count(_lib.frame_count(audio.size, 512), 30, 25, TILER_SCRIPT)
    covered = ((n - 1) * 5 + 30 - 2) * 512 - 1024     # samples whose every contributing frame is tiled
    assert np.max(np.abs(got.sum(0)[:covered] - audio[:covered])) < 1e-4


def test_train_auto_writes_the_reference_files(tmp_path):
    import scipy.io.wavfile
    F = 513
    params = synth_params("dsd", 30, F, seed=2)
    model = str(tmp_path / "model.pkl")
    dcs.save_model(model, params)
    audio = synth_audio(44100, seed=3, channels=2)
    wav = str(tmp_path / "mix.wav")
    scipy.io.wavfile.write(wav, 44100, (audio * 32767).astype('int16'))
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "separate_dsd", os.path.join(os.path.dirname(__file__), "..", "examples", "dsd100", "separate_dsd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(["-i", wav, "-o", str(tmp_path), "-m", model])
    sr, mono = dcs.separation.read_wav(wav)
    want = pipeline.separate("dsd", params, (mono[:, 0] + mono[:, 1]) / 2, 0.3, 30, 25, 32, 1024, 512, np.hanning)
    for i, name in enumerate(["vocals", "bass", "drums", "other"]):
        sr, data = scipy.io.wavfile.read(str(tmp_path / (name + ".wav")))
        assert sr == 44100 and data.dtype == np.int16 and data.shape == (44100,)
        assert np.abs(data.astype(int) - (want[i] * 32767).astype('int16').astype(int)).max() <= 2


# ------------------------------------------------------------------------------------------ other graphs
def _assert_masked(got, ref, p_ref, mix, S, tol=1e-4, well=1e-3, p_got=None, conv='B', label=None):
    """Masked sources against the float64 oracle on EVERY bin (tests/maskcheck.py): the error of each bin is held to
    the bound that follows from the mask's own conditioning, |err| <= mix * min(1, (S+1) d / (D - S d)) with d the
    largest error of the network output before masking and D the reference's mask denominator, and to the plain 1e-4
    bar wherever that bound is below it; the number of bins outside 1e-4 is counted and written to
    gpurun_out/mask_bins.txt (they all sit where every source is (nearly) zero and the mask is discontinuous)."""
    from maskcheck import check_masked
    if p_got is None:
        p_got = p_ref
    return check_masked(got, np.stack([r[:, 0] for r in ref]) if isinstance(ref, list) else ref, p_ref, p_got, mix, S,
                        conv, tol=tol, label=label)


@pytest.mark.parametrize("arch,F,n", [("ikala", 513, 3), ("bach10", 257, 3), ("bach10_si", 257, 2), ("ikala", 1025, 2)])
def test_generic_graphs_match_oracle(arch, F, n):
    """iKala (max-pool / un-pool), Bach10 (strided conv1 with uncovered tail columns) and the 4-channel
    score-informed graph: network output before masking and the masked sources."""
    tc = 30
    params = synth_params(arch, tc, F, seed=3)
    x = _tiles(arch, n, tc, F, seed=12)
    ctx = default_context()
    net = Network(ctx, arch, params, tc, F, live_only=False)      # the whole graph (all 16 channels of the score-informed one)
    xd = ctx.to_device(x, np.float32)
    p = net.forward_raw(xd).cpu().numpy()
    want = net_ref.forward(arch, params, x.astype(np.float64), inverse='explicit').numpy()
    assert p.shape == want.shape
    assert np.max(np.abs(p - want)) < 1e-4
    got = net.forward_masked(xd).cpu().numpy()
    ref = net_ref.predict(arch, params, x.astype(np.float64), inverse='explicit')
    assert got.shape == (ARCHS[arch].S, n, tc, F)
    _assert_masked(got, ref, want, x[:, 0].astype(np.float64), ARCHS[arch].S, p_got=p,
                   conv='A' if arch == 'ikala' else 'B', label="%s F=%d glorot" % (arch, F))


def test_ikala_trainer_graph_realistic_weights_every_mask_bin_within_1e4():
    """north_star "within 1e-4 fp32 per mask bin" on the graph the round-4 review singled out (the iKala trainer's no-pool
    graph) at its real size, with trained-like parameters: non-zero biases in every layer, an output bias that does
    not sit on a plateau of the pre-bias output, tiles cut from the scaled magnitude spectrogram of a signal (with a
    digital-silence gap).  Every mask bin is held to the four-part criterion of oracle/maskcheck.py under the graph's own
    convention (A: an all-zero bin is 1/S).  The only bins that can ever leave 1e-4 are those where every source is within float32
    rounding of zero WITHOUT being exactly zero in both implementations (INTEGRATION.md "mask bins")."""
    from oracle import cases, stft_np, tiling_np
    arch, tc, F, N = "ikala_nopool", 30, 513, 1024
    rs = np.random.RandomState(2024)
    params = synth_params(arch, tc, F, seed=77, gain=1.4)
    for i in (1, 2, 4, 5):
        params[i] = rs.uniform(-0.1, 0.1, params[i].shape).astype(np.float32)
    for i in (7, 9, 11):
        params[i] = rs.uniform(-0.3, 0.1, params[i].shape).astype(np.float32)
    audio = synth_audio(int(1.4 * 44100), seed=17)
    mag = 0.3 * stft_np.compute_file(audio, phase=False, frameSize=N, hopSize=512, window=np.hanning).astype(np.float32)
    fb, n = tiling_np.generate_overlapadd(mag, F, tc, 20, 32, tiler=tiling_np.SCRIPT, fill=0.0)
    x = fb.reshape((-1,) + fb.shape[2:])[:n].astype(np.float32)
    assert n >= 3
    y = cases.pre_bias_output(arch, params, x)
    params[-1] = np.array([-np.quantile(y[:, c], 0.6) + 0.013 * (1 + c) for c in range(2)], dtype=np.float32)
    # no plateau at the rectifier's threshold: the pre-bias outputs within 1e-6 of -bias are a vanishing share of the bins
    near = np.abs(y + params[-1].astype(np.float64)[None, :, None, None]) < 1e-6
    assert near.mean() < 1e-4
    ctx = default_context()
    net = Network(ctx, arch, params, tc, F)
    xd = ctx.to_device(x, np.float32)
    want = net_ref.forward(arch, params, x.astype(np.float64), inverse='explicit').numpy()
    p = net.forward_raw(xd).cpu().numpy()
    assert np.max(np.abs(p - want)) < 1e-4
    assert 0.2 < float((want == 0).mean()) < 0.9                    # the rectifier cuts a large part to exact zeros
    got = net.forward_masked(xd).cpu().numpy()
    ref = net_ref.predict(arch, params, x.astype(np.float64), inverse='explicit')
    rec = _assert_masked(got, ref, want, x[:, 0].astype(np.float64), 2, p_got=p, conv='A',
                         label="ikala_nopool F=513 realistic weights, %d tiles" % n)
    # the criterion of oracle/maskcheck.py is asserted inside; on top of it this draw has been inside 1e-4 on EVERY bin with
    # every kernel so far -- kept as a tripwire only where no bin is ill-conditioned (otherwise the count is a report)
    assert rec["mask_consistent"], rec
    if rec["unconditioned_bins"] == 0:
        assert rec["bins_outside_1e4"] == 0 and rec["max_err"] < 1e-4, rec


@pytest.mark.parametrize("n", [150, 128])
def test_dense_layers_with_all_rows_in_one_workgroup(n):
    """128 ... 176 tiles in one launch: the per-source dense layers (256 x 18 810 here, 256 x 166 650 at F = 2049) run on
    the bf16 pipe with three-way split operands and every row in one workgroup (gemm_bf16x3_skinny_kernel, 11 or 8 row
    blocks).  f32 path against the oracle at 1e-4; with the f16 switch on (conv2 / conv2^T in f16, both InverseLayers in
    the fused decoder kernel, one run per image here) against the f16 path's stated tolerance."""
    arch, F, tc = "bach10", 257, 30
    S = ARCHS[arch].S
    params = synth_params(arch, tc, F, seed=3)
    x = _tiles(arch, n, tc, F, seed=21)
    ctx = default_context()
    net = Network(ctx, arch, params, tc, F)
    xd = ctx.to_device(x, np.float32)
    want = net_ref.forward(arch, params, x.astype(np.float64), inverse='explicit').numpy()
    p = net.forward_raw(xd).cpu().numpy()
    assert np.max(np.abs(p - want)) < 1e-4
    got = net.forward_masked(xd).cpu().numpy()
    ref = net_ref.predict(arch, params, x.astype(np.float64), inverse='explicit')
    _assert_masked(got, ref, want, x[:, 0].astype(np.float64), S, p_got=p, conv='B', label="bach10 F=257, %d tiles" % n)
    net.set_conv_precision('f16')
    p16 = net.forward_raw(xd).cpu().numpy()
    net.set_conv_precision('f32')
    assert np.max(np.abs(p16 - want)) < 2e-3


def test_ikala_pool_tie_modes():
    """Digital-silence rows make every pooling window tie: Theano's CPU gradient feeds all tied positions
    (default), cuDNN only the first."""
    from deepconvsep_amd.arch import TIE_ALL, TIE_FIRST
    tc, F, n = 30, 513, 2
    params = synth_params("ikala", tc, F, seed=9)
    x = _tiles("ikala", n, tc, F, seed=13)
    x[0, 0, 10:14] = 0.0
    ctx = default_context()
    net = Network(ctx, "ikala", params, tc, F)
    xd = ctx.to_device(x, np.float32)
    for mode, name in ((TIE_ALL, 'all'), (TIE_FIRST, 'first')):
        p = net.forward_raw(xd, tie_mode=mode).cpu().numpy()
        want = net_ref.forward("ikala", params, x.astype(np.float64), tie_mode=name, inverse='explicit').numpy()
        assert np.max(np.abs(p - want)) < 1e-4
    a = net.forward_raw(xd, tie_mode=TIE_ALL).cpu().numpy()
    b = net.forward_raw(xd, tie_mode=TIE_FIRST).cpu().numpy()
    assert np.abs(a - b).max() > 1e-5


def _run_children(child_src, args, envs, timeout=300, parallel=4, per_child_args=None):
    """One fresh process per environment (the switches of libdcs are read once per process), `parallel` of them at a time on the
    one GPU -- a child is ~3 s of interpreter + torch start-up around a fraction of a second of kernels, so running the variants
    of a test side by side is what keeps the suite short.  Every child must exit 0; the failing environments are reported."""
    import subprocess
    pending, running, failed = list(enumerate(envs)), [], []
    while pending or running:
        while pending and len(running) < parallel:
            i, env = pending.pop(0)
            child_env = dict(os.environ)
            child_env.update(env)
            extra = list(per_child_args[i]) if per_child_args else []        # e.g. the child's own output file
            running.append((env, subprocess.Popen([sys.executable, "-c", child_src] + [str(a) for a in list(args) + extra], env=child_env,
                                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        env, proc = running.pop(0)
        try:
            out, err = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            proc.kill()
            out, err = proc.communicate()
            failed.append((env, "timeout", out[-400:], err[-800:]))
            continue
        if proc.returncode != 0:
            failed.append((env, proc.returncode, out[-400:], err[-800:]))
    assert not failed, failed


_IKALA_POOL_CHILD = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from deepconvsep_amd.arch import TIE_ALL, TIE_FIRST
from deepconvsep_amd.runtime import Network, default_context
from deepconvsep_amd.synth import synth_params
z = np.load(sys.argv[2])
ctx = default_context()
worst = 0.0
for F in (513, 1025, 271):                 # every size in one process: the switches are read once either way
    net = Network(ctx, 'ikala', synth_params('ikala', 30, F, seed=9), 30, F)
    xd = ctx.to_device(z['x%d' % F], np.float32)
    for mode, key in ((TIE_ALL, 'want_all%d' % F), (TIE_FIRST, 'want_first%d' % F)):
        p = net.forward_raw(xd, tie_mode=mode).cpu().numpy()
        err = float(np.max(np.abs(p - z[key])))
        print('F %d %s max err %.3e' % (F, key, err))
        worst = max(worst, err)
sys.exit(0 if worst < 1e-4 else 3)
"""
_IKALA_POOL_CASE = {}


def test_ikala_pool_fused_and_separate_kernels_agree_with_the_oracle(tmp_path):
    """The iKala graph's max-pool runs inside conv1 and its VJP inside conv1^T by default (routing bits instead of the
    full-resolution activations); DCS_POOL_FUSED=0 (or either register kernel switched off) takes the four separate
    kernels.  Both against the oracle for both tie routings, on tiles with digital-silence rows and a silent band (every
    window over them ties exactly: the bias alone), and F = 271 / 513 (w1 = 81 / 162: rows that do not end on a window)."""
    import subprocess
    # max-pool routing is discontinuous: two conv1 outputs of a window closer than float32 rounding route the gradient
    # differently in float64 -- seed 30 is the draw (of seeds 30..79, three tiles) with the widest smallest margin at all three
    # sizes: 2.6e-6 of the largest activation, ~10x the rounding error of a 30-tap float32 dot product
    n = 3
    if not _IKALA_POOL_CASE:                             # inputs and oracle outputs, once per session
        for F in (513, 1025, 271):
            x = _tiles("ikala", n, 30, F, seed=30)
            x[0, 0, 10:14] = 0.0
            x[2, 0, :, 40:90] = 0.0
            params = synth_params("ikala", 30, F, seed=9)
            _IKALA_POOL_CASE["x%d" % F] = x
            _IKALA_POOL_CASE["want_all%d" % F] = net_ref.forward("ikala", params, x.astype(np.float64), tie_mode='all',
                                                                 inverse='explicit').numpy()
            _IKALA_POOL_CASE["want_first%d" % F] = net_ref.forward("ikala", params, x.astype(np.float64), tie_mode='first',
                                                                   inverse='explicit').numpy()
    f = tmp_path / "case.npz"
    np.savez(f, **_IKALA_POOL_CASE)
    _run_children(_IKALA_POOL_CHILD, [ROOT, f], [{}, {"DCS_POOL_FUSED": "0"}, {"DCS_POOL_FUSED": "0", "DCS_CONV1_REG": "0"},
                                                 {"DCS_DECONV1_REG": "0"}], timeout=200)


def test_generic_chunked_batch_equals_small_batches():
    """More tiles than one scratch chunk: the chunked batch equals evaluation in small batches (the chunk is sized
    from a 4 GiB scratch budget; DCS_GENERIC_CHUNK forces small chunks in the variant test below)."""
    tc, F, n = 30, 257, 70
    params = synth_params("bach10", tc, F, seed=4)
    x = _tiles("bach10", n, tc, F, seed=14)
    ctx = default_context()
    net = Network(ctx, "bach10", params, tc, F)
    xd = ctx.to_device(x, np.float32)
    # the GEMM variant (and with it the summation order) depends on the number of rows: equal to rounding
    whole = net.forward_raw(xd).cpu().numpy()
    parts = np.concatenate([net.forward_raw(xd[i:i + 7]).cpu().numpy() for i in range(0, n, 7)], axis=0)
    assert np.max(np.abs(whole - parts)) < 1e-6
    whole_m = net.forward_masked(xd).cpu().numpy()
    ref = net_ref.predict("bach10", params, x.astype(np.float64), inverse='explicit')
    want = net_ref.forward("bach10", params, x.astype(np.float64), inverse='explicit').numpy()
    _assert_masked(whole_m, ref, want, x[:, 0].astype(np.float64), 4, p_got=whole, label="bach10 F=257 chunked")


def test_ikala_separation_matches_oracle():
    """BASELINE configs[0]: iKala 2-source, frameSize 1024 hop 512, overlap 20, stereo wav summed L+R."""
    F, N = 513, 1024
    params = synth_params("ikala", 30, F, seed=1)
    stereo = synth_audio(3 * 44100, seed=0, channels=2)
    audio = stereo[:, 0] + stereo[:, 1]                       # separate_ikala.py:229
    sep = dcs.Separator("ikala", params, 0.3, 30, 20, 32, F, N, 512, np.hanning)
    got = sep.separate(audio)
    want = pipeline.separate("ikala", params, audio, 0.3, 30, 20, 32, N, 512, np.hanning)
    assert got.shape == want.shape == (2, audio.size)
    assert np.max(np.abs(got - want)) < 1e-4
    assert np.max(np.abs(sep.separate_stepwise(audio) - got)) < 2e-5


def test_bach10_full_size_separation_matches_oracle():
    """BASELINE configs[3] shapes: frameSize 4096 blackmanharris, 2049 bins, 17-array model (853 MB of
    float32 weights, K = 166 650 dense layers), convention-B masks."""
    F, N = 2049, 4096
    params = synth_params("bach10", 30, F, seed=3)
    audio = synth_audio(22050 + 9000, seed=5)
    sep = dcs.Separator("bach10", params, 0.3, 30, 25, 32, F, N, 512, dcs.blackmanharris)
    got = sep.separate(audio)
    want = pipeline.separate("bach10", params, audio, 0.3, 30, 25, 32, N, 512, dcs.blackmanharris)
    assert got.shape == want.shape == (4, audio.size)
    assert np.max(np.abs(got - want)) < 1e-4


def test_long_file_plan_on_one_gpu():
    """deepconvsep_amd.dist.plan_long_file: separating the two halo-extended halves one after the other on
    this GPU and keeping each one's interior reproduces the whole-file result."""
    from deepconvsep_amd.dist import plan_long_file
    F, N = 513, 1024
    params = synth_params("dsd", 30, F, seed=2)
    audio = synth_audio(4 * 44100, seed=8)
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, 512, np.hanning)
    whole = sep.separate(audio)
    pieces = [sep.separate(audio[p['a0']:p['a1']])[:, p['s0'] - p['a0']:p['s1'] - p['a0']]
              for p in plan_long_file(audio.size, 2, N, 512, 30, 25)]
    assert np.max(np.abs(np.concatenate(pieces, axis=1) - whole)) < 2e-6


def test_batch_driver_matches_single_file_runs(tmp_path):
    import importlib.util
    import os
    import scipy.io.wavfile
    F = 513
    params = synth_params("dsd", 30, F, seed=2)
    model = str(tmp_path / "model.pkl")
    dcs.save_model(model, params)
    wavs = []
    for i in range(3):
        a = synth_audio(30000 + 5000 * i, seed=20 + i)
        w = str(tmp_path / ("song%d.wav" % i))
        scipy.io.wavfile.write(w, 44100, (a * 32767).astype('int16'))
        wavs.append(w)
    spec = importlib.util.spec_from_file_location(
        "separate_batch", os.path.join(os.path.dirname(__file__), "..", "examples", "separate_batch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "out"
    out.mkdir()
    mod.main(["-a", "dsd", "-m", model, "-o", str(out), "--group", "1"] + wavs)      # one file per set of launches
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, 1024, 512, np.hanning)
    for i, w in enumerate(wavs):
        sr, audio = dcs.separation.read_wav(w)
        want = sep.separate(audio)
        for s, name in enumerate(["vocals", "bass", "drums", "other"]):
            sr2, data = scipy.io.wavfile.read(str(out / ("song%d" % i) / (name + ".wav")))
            assert np.array_equal(data, (want[s] * 32767).astype('int16'))


def test_pcm16_frames_to_mono_float_is_the_scripts_arithmetic():
    """dcs_pcm16_to_float: int16 wav frames -> sample / 32767 (float64), (L + R) / 2 or L + R or the mono channel, one rounding
    to float32 -- bit for bit the array `to_device(to_mono(read_wav(f)))` uploads (separate_dsd.py:275-287,
    separate_ikala.py:229), for stacked clips with padded strides; a mono file under the iKala rule is the reference's
    IndexError."""
    import torch
    from deepconvsep_amd.runtime import pcm16_to_float
    from deepconvsep_amd.separation import to_mono
    ctx = default_context()
    rs = np.random.RandomState(5)
    for ch in (1, 2, 3):
        for n in (1, 255, 4097):
            x = rs.randint(-32768, 32768, (2, n, ch)).astype(np.int16)
            x[0, 0] = 32767
            x[1, -1] = -32768
            dev = torch.zeros((2, (n + 5) * ch), dtype=torch.int16, device=ctx.device)
            dev[:, :n * ch] = torch.from_numpy(x.reshape(2, -1)).to(ctx.device)
            for mode, arch in ((0, "dsd"), (1, "ikala")):
                if mode == 1 and ch < 2:
                    with pytest.raises(ValueError):
                        pcm16_to_float(ctx, dev, ch, mode)
                    continue
                got = pcm16_to_float(ctx, dev, ch, mode).cpu().numpy()
                assert got.shape == (2, n + 5) and got.dtype == np.float32
                for c in range(2):
                    f = x[c].astype('float') / 32767
                    want = to_mono(f if ch > 1 else f[:, 0], arch).astype(np.float32)
                    assert np.array_equal(got[c, :n], want) and not got[c, n:].any()


def test_batch_driver_int16_path_writes_the_same_bytes_as_the_float_path(tmp_path):
    """examples/separate_batch.py: 16-bit PCM files go to the GPU as int16 frames (pinned staging, device mix-down and int16
    conversion, RIFF header written directly) -- the wav files are byte-identical to those of --float-path (read_wav / to_mono
    / float32 upload / float64 download / scipy writes), for stereo and mono inputs of equal and different lengths in one
    group; a float32 wav in the same run takes the float path; an unreadable file fails alone."""
    import importlib.util
    import os
    import scipy.io.wavfile
    F = 513
    params = synth_params("dsd", 30, F, seed=2)
    model = str(tmp_path / "model.pkl")
    dcs.save_model(model, params)
    wavs = []
    for i, (n, ch) in enumerate([(40000, 2), (40000, 2), (31000, 1), (52000, 2), (40000, 1), (36000, 3)]):
        a = synth_audio(n, seed=60 + i, channels=ch)
        w = str(tmp_path / ("clip%d.wav" % i))
        scipy.io.wavfile.write(w, 44100, (a * 32767).astype('int16'))
        wavs.append(w)
    fl = str(tmp_path / "floaty.wav")
    scipy.io.wavfile.write(fl, 44100, synth_audio(33000, seed=70).astype(np.float32))
    broken = str(tmp_path / "broken.wav")
    with open(broken, "wb") as fh:
        fh.write(b"RIFF\x00\x00\x00\x00WAVEjunk")
    spec = importlib.util.spec_from_file_location(
        "separate_batch", os.path.join(os.path.dirname(__file__), "..", "examples", "separate_batch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    outs = {}
    for name, extra in (("int16", []), ("float", ["--float-path"])):
        out = tmp_path / name
        out.mkdir()
        rc = mod.main(["-a", "dsd", "-m", model, "-o", str(out), "--group", "4"] + extra + wavs + [fl, broken])
        assert rc == 1                                               # the broken file is reported, the others are done
        outs[name] = out
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, 1024, 512, np.hanning)
    for w in wavs + [fl]:
        stem = os.path.splitext(os.path.basename(w))[0]
        sr, audio = dcs.separation.read_wav(w)
        alone = sep.separate(dcs.separation.to_mono(audio, "dsd"))
        for s_, src in enumerate(["vocals", "bass", "drums", "other"]):
            a = open(str(outs["int16"] / stem / (src + ".wav")), "rb").read()
            b = open(str(outs["float"] / stem / (src + ".wav")), "rb").read()
            assert a == b, (stem, src)
            sr2, data = scipy.io.wavfile.read(str(outs["int16"] / stem / (src + ".wav")))
            assert sr2 == 44100 and data.dtype == np.int16 and data.shape[0] == audio.shape[0]
            assert np.max(np.abs(data.astype(np.int32) - (alone[s_] * 32767).astype('int16').astype(np.int32))) <= 1
    assert not (outs["int16"] / "broken").exists() or not os.listdir(str(outs["int16"] / "broken"))


def test_batch_driver_groups_equal_lengths(tmp_path):
    """--group: files share sets of launches (Separator.separate_many -> dcs_separate_ragged / dcs_separate_batch); every
    file still gets what the single-file path gives it (kernel variants depend on the launch size, so float results
    agree to ~1e-6 and the int16 samples to one step)."""
    import importlib.util
    import os
    import scipy.io.wavfile
    F = 513
    params = synth_params("dsd", 30, F, seed=2)
    model = str(tmp_path / "model.pkl")
    dcs.save_model(model, params)
    lengths = [40000, 31000, 40000, 40000, 52000]
    wavs = []
    for i, n in enumerate(lengths):
        w = str(tmp_path / ("clip%d.wav" % i))
        scipy.io.wavfile.write(w, 44100, (synth_audio(n, seed=40 + i) * 32767).astype('int16'))
        wavs.append(w)
    spec = importlib.util.spec_from_file_location(
        "separate_batch", os.path.join(os.path.dirname(__file__), "..", "examples", "separate_batch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "out"
    out.mkdir()
    mod.main(["-a", "dsd", "-m", model, "-o", str(out), "--group", "5"] + wavs)
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, 1024, 512, np.hanning)
    many = sep.separate_many([dcs.separation.to_mono(dcs.separation.read_wav(w)[1], "dsd") for w in wavs])
    for i, w in enumerate(wavs):
        sr, audio = dcs.separation.read_wav(w)
        want = sep.separate(dcs.separation.to_mono(audio, "dsd"))
        assert many[i].shape == want.shape and np.max(np.abs(many[i] - want)) < 5e-6
        ref = pipeline.separate("dsd", params, dcs.separation.to_mono(audio, "dsd"), 0.3, 30, 25, 32, 1024, 512, np.hanning)
        assert np.max(np.abs(many[i] - ref)) < 1e-4
        for s_, name in enumerate(["vocals", "bass", "drums", "other"]):
            sr2, data = scipy.io.wavfile.read(str(out / ("clip%d" % i) / (name + ".wav")))
            d = data.astype(np.int32) - (want[s_] * 32767).astype('int16').astype(np.int32)
            assert data.shape[0] == lengths[i] and np.max(np.abs(d)) <= 1


@pytest.mark.parametrize("N,hop,tiler,lengths", [
    (1024, 512, "script", [40000, 31000, 40000, 52000, 19000, 47011]),
    (2048, 512, "script", [94208, 60000, 94208, 70001]),
    (1024, 256, "library", [30000, 20011, 25000]),
    # frameSize/2 > 2*hop: a clip's last samples depend on frames past its own end (per-clip normalisation and row
    # zeroing matter); lengths on, just below and just above a hop multiple
    (4096, 512, "script", [51200, 51199, 51201, 40000, 61440]),
    (2048, 256, "script", [25600, 25601, 25599, 33000]),
])
def test_separate_ragged_equals_clip_by_clip(N, hop, tiler, lengths):
    """dcs_separate_ragged: clips of different lengths in one set of launches -- each clip's frames, tiles and
    cross-fade are its own (device table of {samples, frames, tiles}); results agree with the single-clip call to
    fp32 rounding (kernel variants depend on the launch size) and with the oracle within 1e-4."""
    import torch
    F = N // 2 + 1
    params = synth_params("dsd", 30, F, seed=2)
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, hop, np.hanning, tiler=tiler)
    clips = [synth_audio(n, seed=60 + i) for i, n in enumerate(lengths)]
    Lmax = max(lengths)
    stack = np.zeros((len(clips), Lmax + 17), dtype=np.float32)          # a row pitch larger than the longest clip
    for b, a in enumerate(clips):
        stack[b, :a.size] = a
    dev = sep.ctx.to_device(stack, np.float32)
    got = sep.net.separate_ragged(sep.plan, dev[:, :Lmax], lengths, 25, sep.tiler, 0.3).cpu().numpy()
    assert list(sep.net.last_frames) == [stft_np.frame_count(n, hop) for n in lengths]
    for b, a in enumerate(clips):
        alone = sep.separate(a)
        assert np.max(np.abs(got[b, :, :a.size] - alone)) < 5e-6
        assert not got[b, :, a.size:].any()                               # nothing written past a clip's own length
        want = pipeline.separate("dsd", params, a, 0.3, 30, 25, 32, N, hop, np.hanning,
                                 tiler=tiling_np.SCRIPT if tiler == "script" else tiling_np.LIBRARY)
        assert np.max(np.abs(got[b, :, :a.size] - want)) < 1e-4
    many = sep.separate_many(clips)
    for b, a in enumerate(clips):
        assert many[b].shape == (4, a.size) and np.max(np.abs(many[b] - sep.separate(a))) < 5e-6


@pytest.mark.parametrize("arch,N,ov,lengths", [("ikala", 1024, 20, [40000, 31000, 52000, 40000, 26011]),
                                                ("bach10", 4096, 25, [60000, 45001, 60000, 38000])])
def test_separate_ragged_of_the_generic_graphs_equals_clip_by_clip(arch, N, ov, lengths):
    """dcs_separate_ragged on the ikala / bach10 graphs: one STFT / iSTFT launch over all clips and ONE pass of all their
    tiles through the network, every clip framed, tiled and cross-faded with its own counts (its tiles sit behind those of
    the clips before it): equal to the single-clip call to fp32 rounding, to the oracle within 1e-4, nothing written past a
    clip's own length; separate_many takes this path for mixed lengths."""
    F = N // 2 + 1
    window = dcs.blackmanharris if arch == "bach10" else np.hanning
    params = synth_params(arch, 30, F, seed=4)
    sep = dcs.Separator(arch, params, 0.3, 30, ov, 32, F, N, 512, window)
    clips = [synth_audio(n, seed=80 + i) for i, n in enumerate(lengths)]
    Lmax = max(lengths)
    stack = np.zeros((len(clips), Lmax + 5), dtype=np.float32)
    for b, a in enumerate(clips):
        stack[b, :a.size] = a
    dev = sep.ctx.to_device(stack, np.float32)
    got = sep.net.separate_ragged(sep.plan, dev[:, :Lmax], lengths, ov, sep.tiler, 0.3).cpu().numpy()
    assert list(sep.net.last_frames) == [stft_np.frame_count(n, 512) for n in lengths]
    S = ARCHS[arch].S
    for b, a in enumerate(clips):
        alone = sep.separate(a)
        assert got[b].shape[0] == S and np.max(np.abs(got[b, :, :a.size] - alone)) < 5e-6
        assert not got[b, :, a.size:].any()
    want = pipeline.separate(arch, params, clips[1], 0.3, 30, ov, 32, N, 512, window)
    assert np.max(np.abs(got[1, :, :clips[1].size] - want)) < 1e-4
    many = sep.separate_many(clips)
    for b, a in enumerate(clips):
        assert many[b].shape == (S, a.size) and np.max(np.abs(many[b] - got[b, :, :a.size])) < 5e-6


def test_separate_ragged_rejects_what_it_cannot_do():
    import torch
    params = synth_params("dsd", 30, 257, seed=2)
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, 257, 512, 200, np.hanning)      # block-level FFT only
    dev = sep.ctx.to_device(np.zeros((2, 30000), dtype=np.float32), np.float32)
    with pytest.raises(NotImplementedError):
        sep.net.separate_ragged(sep.plan, dev, [30000, 20000], 25, sep.tiler, 0.3)
    many = sep.separate_many([synth_audio(30000, seed=1), synth_audio(20000, seed=2)])   # falls back to one by one
    assert many[0].shape == (4, 30000) and many[1].shape == (4, 20000)
    params2 = synth_params("dsd", 30, 513, seed=2)
    sep2 = dcs.Separator("dsd", params2, 0.3, 30, 25, 32, 513, 1024, 512, np.hanning)
    dev2 = sep2.ctx.to_device(np.zeros((2, 30000), dtype=np.float32), np.float32)
    with pytest.raises(ValueError):
        sep2.net.separate_ragged(sep2.plan, dev2, [30000, 5], 25, sep2.tiler, 0.3)     # 5 samples: no tile


def test_graph_replay_recomputes_on_a_side_stream(monkeypatch):
    """With DCS_GRAPH=1 (opt-in since round 6: eager launches are at least as fast at every measured shape) the fused step is
    captured into a hipGraph on the second identical call (non-default stream) and replayed afterwards: replays must track new
    input written into the same buffers."""
    import torch
    monkeypatch.setenv("DCS_GRAPH", "1")                    # read per call by libdcs
    from deepconvsep_amd.runtime import Context
    F, N = 513, 1024
    params = synth_params("dsd", 30, F, seed=2)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx2 = Context()
        sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, 512, np.hanning, ctx=ctx2)
        a1, a2 = synth_audio(44100, seed=31), synth_audio(44100, seed=32)
        buf = ctx2.to_device(a1, np.float32)
        out = torch.empty((4, a1.size), dtype=torch.float32, device=buf.device)
        res = []
        for i in range(4):                                   # eager, capture+launch, replay, replay
            sep.net.separate(sep.plan, buf, 25, sep.tiler, 0.3, out=out)
            stream.synchronize()
            res.append(out.cpu().numpy().copy())
        buf.copy_(torch.from_numpy(a2.astype(np.float32)).to(buf.device))
        sep.net.separate(sep.plan, buf, 25, sep.tiler, 0.3, out=out)
        stream.synchronize()
        other = out.cpu().numpy().copy()
    for r in res[1:]:
        assert np.array_equal(r, res[0])                    # same kernels, same order: bit-identical
    want = pipeline.separate("dsd", params, a2, 0.3, 30, 25, 32, N, 512, np.hanning)
    assert np.max(np.abs(other - want)) < 1e-4
    assert np.max(np.abs(other - res[0])) > 1e-3


@pytest.mark.parametrize("N,seconds,clips,tiler,hop", [(1024, 1.0, 3, "script", 512), (2048, 2.14, 8, "script", 512),
                                                         (1024, 0.9, 5, "library", 512),
                                                         (512, 0.8, 3, "script", 200),     # block-level FFT kernels
                                                         (4096, 1.3, 3, "script", 512),    # frameSize/2 > 2*hop
                                                         (2048, 0.75, 4, "script", 256)])
def test_separate_batch_equals_clip_by_clip(N, seconds, clips, tiler, hop):
    """dcs_separate_batch: equal-length clips sharing one set of launches are each separated exactly as
    dcs_separate separates them alone -- same tiles, same cross-fade; only fp32 rounding may differ, because the
    FFT / GEMM kernel variants are picked by the total amount of work -- and match the oracle."""
    import torch
    F = N // 2 + 1
    params = synth_params("dsd", 30, F, seed=2)
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, hop, np.hanning, tiler=tiler)
    L = int(44100 * seconds)
    audio = np.stack([synth_audio(L, seed=40 + c) for c in range(clips)]).astype(np.float32)
    audio[1, L // 3: L // 2] = 0.0                               # a digital-silence gap in one clip
    buf = sep.ctx.to_device(audio, np.float32)
    got = sep.net.separate_batch(sep.plan, buf, 25, sep.tiler, 0.3).cpu().numpy()
    assert got.shape == (clips, 4, L)
    for c in range(clips):
        alone = sep.net.separate(sep.plan, buf[c], 25, sep.tiler, 0.3).cpu().numpy()
        assert np.max(np.abs(got[c] - alone)) < 2e-6, "clip %d differs from the single-clip path" % c
    want = pipeline.separate("dsd", params, audio[1].astype(np.float64), 0.3, 30, 25, 32, N, hop, np.hanning,
                             tiler=tiling_np.SCRIPT if tiler == 'script' else tiling_np.LIBRARY)
    assert np.max(np.abs(got[1] - want)) < 1e-4
    # a strided view (clip stride > length) goes through the same entry point
    wide = sep.ctx.to_device(np.concatenate([audio, np.zeros((clips, 37), np.float32)], axis=1), np.float32)
    got2 = sep.net.separate_batch(sep.plan, wide[:, :L], 25, sep.tiler, 0.3).cpu().numpy()
    assert np.array_equal(got2, got)


@pytest.mark.parametrize("clips", [20, 32])
def test_bench_launch_shapes_match_oracle(clips):
    """The launch groups bench.py times -- 20 (the driver's --steps 20) and 32 (the default) independent 32-tile batches
    of N = 2048 in ONE set of launches (dcs_separate_batch), the bf16x3 final kernel selected by size, not by an
    environment switch -- against the oracle, clip by clip: PCM within 1e-4 on every sample of every clip."""
    N, F, hop, tc, ov = 2048, 1025, 512, 30, 25
    params = synth_params("dsd", tc, F, seed=2)
    L = (tc + 1 + 31 * (tc - ov) - 2) * hop
    audio = np.stack([synth_audio(L, seed=300 + c) for c in range(clips)])
    audio[3, 20000:30000] = 0.0
    sep = dcs.Separator("dsd", params, 0.3, tc, ov, 32, F, N, hop, np.hanning)
    T = _lib.frame_count(L, hop)
    assert sep.net.final_kernel(T, clips) == "bf16x3"
    ctx = default_context()
    pcm = sep.net.separate_batch(sep.plan, ctx.to_device(audio, np.float32), ov, sep.tiler, 0.3).cpu().numpy()
    assert pcm.shape == (clips, 4, L) and sep.net.last_tiles == 32
    worst = 0.0
    for c in range(clips):
        want = pipeline.separate("dsd", params, audio[c], 0.3, tc, ov, 32, N, hop, np.hanning)
        worst = max(worst, float(np.max(np.abs(pcm[c] - want))))
    assert worst < 1e-4


_WHOLE_PATH_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import deepconvsep_amd as dcs
from deepconvsep_amd.synth import synth_audio, synth_params
arch, N, ov = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
F = N // 2 + 1
sep = dcs.Separator(arch, synth_params(arch, 30, F, seed=7), 0.3, 30, ov, 32, F, N, 512, np.hanning)
np.save(sys.argv[2], sep.separate(synth_audio(int(3.2 * 44100), seed=33)))
"""


@pytest.mark.parametrize("arch,ov", [("bach10", 25), ("ikala", 20)])
def test_whole_path_fallbacks_of_the_generic_graphs_agree(arch, ov, tmp_path):
    """dcs_separate on the generic graphs: the one-chunk path with mask + cross-fade in one kernel (default), the same
    with the two kernels (DCS_MASK_OLA=0) and the several-chunk path, which cannot defer the mask (DCS_GENERIC_CHUNK=8:
    46 tiles in 6 chunks), must give the same PCM: the fused kernel is bit-compatible by construction, chunking only
    changes which GEMM variants the sizes select."""
    import subprocess
    jobs = (("default", {}), ("two_kernels", {"DCS_MASK_OLA": "0"}), ("chunks", {"DCS_GENERIC_CHUNK": "8"}))
    files = {name: str(tmp_path / (name + ".npy")) for name, _ in jobs}
    # (the child takes its output file first: the shared arguments follow)
    _run_children(_WHOLE_PATH_CHILD, [ROOT], [env for _, env in jobs], per_child_args=[[files[name], arch, "1024", str(ov)] for name, _ in jobs])
    res = {name: np.load(files[name]) for name, _ in jobs}
    assert np.max(np.abs(res["default"])) > 1e-3
    assert np.array_equal(res["default"], res["two_kernels"])
    assert np.max(np.abs(res["default"] - res["chunks"])) < 5e-6


@pytest.mark.parametrize("arch,N,ov,clips,seconds", [("ikala", 1024, 20, 3, 1.1), ("bach10", 1024, 25, 2, 0.9)])
def test_separate_batch_of_the_generic_graphs_equals_clip_by_clip(arch, N, ov, clips, seconds):
    """dcs_separate_batch for the ikala / bach10 graphs: equal-length clips share the launches -- one STFT / iSTFT
    launch and one pass of all their tiles through the network; every clip equals its single-clip result and the
    oracle.  The stereo (ILD) graph and the spectra outputs stay single-clip."""
    F = N // 2 + 1
    params = synth_params(arch, 30, F, seed=1)
    sep = dcs.Separator(arch, params, 0.3, 30, ov, 32, F, N, 512, np.hanning)
    L = int(44100 * seconds)
    audio = np.stack([synth_audio(L, seed=140 + c) for c in range(clips)]).astype(np.float32)
    buf = sep.ctx.to_device(audio, np.float32)
    got = sep.ctx.to_host(sep.net.separate_batch(sep.plan, buf, ov, sep.tiler, 0.3))
    assert got.shape == (clips, ARCHS[arch].S, L)
    for c in range(clips):
        alone = sep.ctx.to_host(sep.net.separate(sep.plan, buf[c], ov, sep.tiler, 0.3))
        assert np.max(np.abs(got[c] - alone)) < 5e-6, "clip %d differs from the single-clip path" % c
    want = pipeline.separate(arch, params, audio[1].astype(np.float64), 0.3, 30, ov, 32, N, 512, np.hanning)
    assert np.max(np.abs(got[1] - want)) < 1e-4
    many = sep.separate_many([a.astype(np.float64) for a in audio] + [synth_audio(L + 999, seed=5)])
    assert np.max(np.abs(many[0] - got[0])) < 5e-6 and many[-1].shape == (ARCHS[arch].S, L + 999)
    ild = dcs.Separator("dsd_ild", synth_params("dsd_ild", 30, 513, seed=1), 0.3, 30, 25, 32, 513, 1024, 512, np.hanning)
    with pytest.raises(NotImplementedError):
        ild.net.separate_batch(ild.plan, buf, 25, TILER_LIBRARY, 0.3)


@pytest.mark.parametrize("arch,F,n", [("bach10", 257, 4), ("ikala", 513, 3)])
def test_f16_mfma_conv_path_stated_tolerance(arch, F, n):
    """BASELINE config 3: conv2 and its transpose with f16 inputs / f32 accumulation on the matrix cores.
    f16 keeps 11 significant bits, so this path does NOT meet the 1e-4 bar of the f32 path.  Tolerance it is
    held to: network output before masking max |err| < 2e-3; masked magnitudes 99.9th percentile < 1e-3 and
    max < 5e-3 over the bins where the mask is well conditioned (reference sum of branch outputs > 1e-2 --
    where all branches are cut to ~0 by the rectifier the mask itself is discontinuous)."""
    import os
    tc = 30
    S = ARCHS[arch].S
    params = synth_params(arch, tc, F, seed=3)
    x = _tiles(arch, n, tc, F, seed=12)
    ctx = default_context()
    net = Network(ctx, arch, params, tc, F)
    xd = ctx.to_device(x, np.float32)
    p_ref = net_ref.forward(arch, params, x.astype(np.float64), inverse='explicit').numpy()
    ref = np.stack([r[:, 0] for r in net_ref.predict(arch, params, x.astype(np.float64), inverse='explicit')])
    f32 = net.forward_masked(xd).cpu().numpy()
    net.set_conv_precision('f16')
    f16 = net.forward_masked(xd).cpu().numpy()
    p16 = net.forward_raw(xd).cpu().numpy()
    net.set_conv_precision('f32')
    assert np.array_equal(net.forward_masked(xd).cpu().numpy(), f32)     # the switch is reversible
    perr = np.abs(p16 - p_ref)
    err = np.abs(f16 - ref)
    well = (p_ref[:, :S].sum(axis=1) > 1e-2)[None].repeat(S, axis=0)       # [S, n, tc, F]
    stats = ("%s F=%d f16 conv path: raw output max|err| %.3e; masked p99.9 %.3e, mean %.3e, max over "
             "well-conditioned bins %.3e (%.1f%% of bins), max overall %.3e; f32 path max %.3e" % (
                 arch, F, perr.max(), np.percentile(err, 99.9), err.mean(), err[well].max(), 100.0 * well.mean(),
                 err.max(), np.abs(f32 - ref).max()))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/f16_stats.txt", "a") as fh:
        fh.write(stats + "\n")
    assert perr.max() < 2e-3, stats
    assert np.percentile(err, 99.9) < 1e-3 and err[well].max() < 5e-3, stats
    well32 = (p_ref[:, :S].sum(axis=1) > 1e-3)[None].repeat(S, axis=0)    # see _assert_masked
    assert np.abs(f32 - ref)[well32].max() < 1e-4
    with pytest.raises(NotImplementedError):
        Network(ctx, "dsd", synth_params("dsd", tc, 513), tc, 513).set_conv_precision('f16')


_VARIANT_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import deepconvsep_amd as dcs
from deepconvsep_amd.synth import synth_params
z = np.load(sys.argv[2])
worst = 0.0
for N in (1024, 2048):                    # both frame sizes in one process: the switches are read once either way
    F = N // 2 + 1
    sep = dcs.Separator('dsd', synth_params('dsd', 30, F, seed=2), 0.3, 30, 25, 32, F, N, 512, np.hanning)
    sep.net.set_latency_stages(0)         # the throughput kernels these switches select among (a 3 s clip would take the one-batch family)
    for rep in range(3):                  # eager, graph capture, graph replay
        got = sep.separate(z['audio'])
    err = float(np.max(np.abs(got - z['want%d' % N])))
    print('N %d max err %.3e' % (N, err))
    worst = max(worst, err)
sys.exit(0 if worst < 1e-4 else 3)
"""


_IKALA_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from deepconvsep_amd.runtime import Network, default_context
from deepconvsep_amd.synth import synth_params
z = np.load(sys.argv[2])
ctx = default_context()
F = int(z['F'])
net = Network(ctx, 'ikala', synth_params('ikala', 30, F, seed=4), 30, F)
p = net.forward_raw(ctx.to_device(z['x'], np.float32)).cpu().numpy()
err = float(np.max(np.abs(p - z['want'])))
print('max err %.3e' % err)
sys.exit(0 if err < 1e-4 else 3)
"""


@pytest.mark.parametrize("F,n", [(513, 9), (1025, 3)])
def test_ikala_conv2_kernels_agree_with_the_oracle(F, n, tmp_path):
    """conv2 of the iKala graph (10 x 20) and its transpose: slab kernel on the bf16 matrix pipe with the slab pre-split
    into bf16 planes (slabconv_ps.hip: tap loop driven by per-block bit masks), the one that splits per tap
    (DCS_SLABCONV_PS=0) and the implicit-GEMM fallback, on batch sizes that give several row bands per image; and the
    folded conv2 + bottleneck map that replaces the forward conv2 by default (generic.hip: fold_conv2_fc_kernel)."""
    import subprocess
    x = _tiles("ikala", n, 30, F, seed=16)
    want = net_ref.forward("ikala", synth_params("ikala", 30, F, seed=4), x.astype(np.float64), inverse='explicit').numpy()
    f = tmp_path / "case.npz"
    np.savez(f, x=x, want=want, F=F)
    # (since round 6 the default folds conv2 + bottleneck layer into one affine map for this graph: the forward conv2 kernels
    # run with DCS_FOLD_CONV2=0; the first child is the folded default, the second the default conv2 kernel)
    unfolded = {"DCS_FOLD_CONV2": "0"}
    _run_children(_IKALA_CHILD, [ROOT, f], [{}, unfolded, dict(unfolded, DCS_SLABCONV="0"), dict(unfolded, DCS_SLABCONV="1"),
                                            dict(unfolded, DCS_SLABCONV_PS="0")], timeout=200)


_GENERIC_CASE = []
_GENERIC_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from deepconvsep_amd.runtime import Network, default_context
from deepconvsep_amd.synth import synth_params
z = np.load(sys.argv[2])
ctx = default_context()
F = int(z['F'])
net = Network(ctx, 'bach10', synth_params('bach10', 30, F, seed=4), 30, F)
import os
f16 = os.environ.get('DCS_TEST_F16') == '1'
if f16:
    net.set_conv_precision('f16')
p = net.forward_raw(ctx.to_device(z['x'], np.float32)).cpu().numpy()
err = float(np.max(np.abs(p - z['want'])))
print('max err %.3e' % err)
sys.exit(0 if err < (2e-3 if f16 else 1e-4) else 3)
"""


_GENERIC_ENVS = [{"DCS_GENERIC_CHUNK": "8"}, {"DCS_GEMM_KSPLIT": "0"}, {"DCS_GEMM_KSPLIT": "5"},
                                 {"DCS_GEMM_KSPLIT": "64"},
                                 {"DCS_COLCONV": "0"}, {"DCS_GEMM_BF16": "0"},
                                 {"DCS_CONV1_MFMA": "0"}, {"DCS_CONV1_MFMA": "0", "DCS_CONV1_REG": "0"}, {"DCS_DECONV1_MFMA": "0"},
                                 {"DCS_DECONV1_MFMA": "0", "DCS_DECONV1_REG": "0"},
                                 {"DCS_TEST_F16": "1"}, {"DCS_TEST_F16": "1", "DCS_DECODER_FUSED": "0"},
                                 {"DCS_TEST_F16": "1", "DCS_COLCONV_WREG": "0"},
                                 {"DCS_TEST_F16": "1", "DCS_GENERIC_CHUNK": "8"}]


def test_generic_graph_variants_agree_with_the_oracle(tmp_path):
    """Scratch chunking, the K-split of the long dense layer (register- and LDS-tiled), the column convolution and the
    register-blocked transposed conv1 all have a fallback or a size rule; force each on a 52-tile Bach10 batch (fresh
    process) and compare the network output with the oracle.  DCS_TEST_F16=1 (read by the child, not by libdcs) turns the
    f16 switch on and holds the output to that path's 2e-3: the fused decoder (default), conv2^T and conv1^T as two
    kernels, the LDS column kernel instead of the weights-in-registers one, and the fused decoder on 8-tile chunks."""
    import subprocess
    F, n = 257, 52
    if not _GENERIC_CASE:                                # once per session
        x = _tiles("bach10", n, 30, F, seed=15)
        _GENERIC_CASE.append((x, net_ref.forward("bach10", synth_params("bach10", 30, F, seed=4), x.astype(np.float64),
                                                 inverse='explicit').numpy()))
    x, want = _GENERIC_CASE[0]
    f = tmp_path / "case.npz"
    np.savez(f, x=x, want=want, F=F)
    _run_children(_GENERIC_CHILD, [ROOT, f], _GENERIC_ENVS, timeout=200)


_CL_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from deepconvsep_amd.runtime import Network, default_context
from deepconvsep_amd.synth import synth_params
z = np.load(sys.argv[2])
ctx = default_context()
F = int(z['F'])
net = Network(ctx, 'bach10', synth_params('bach10', 30, F, seed=4), 30, F)
net.set_conv_precision('f16')
x = ctx.to_device(z['x'], np.float32)
p = net.forward_raw(x).cpu().numpy()
net.set_conv_precision('f32')                      # flips the layout of the packed planes back ...
q = net.forward_raw(x).cpu().numpy()
net.set_conv_precision('f16')                      # ... and forth: the planes are re-packed in place
p2 = net.forward_raw(x).cpu().numpy()
np.savez(sys.argv[3], p=p, q=q, same=np.array_equal(p, p2))
"""


def test_channels_last_dense_output_feeds_the_fused_decoder(tmp_path):
    """From 128 tiles on, the Bach10 graph hands D to its fused decoder channels-last.  Under the f16 switch (round 6) the
    per-source dense layers run on f16 weights and write D as f16 with 32 channels per position (gemm_f16.hip), which the
    decoder reads with 16-byte loads; DCS_DECODER_CL=0 keeps the round-4 form (f32-class dense layers, channel-first f32 D,
    the same decoder).  140 tiles at F = 257: both meet the f16 path's stated 2e-3 against the oracle; flipping the precision
    switch forth and back reproduces the f16 result to the bit and the f32-class result within 1e-4; with the switch off the
    f32-class fused decoder (bf16 planes packed with permuted columns) agrees with the two kernels it replaces, and the
    bottleneck layer's bf16 x 3 K-slices with the f32 K-split."""
    import subprocess
    F, n = 257, 140
    x = _tiles("bach10", n, 30, F, seed=21)
    want = net_ref.forward("bach10", synth_params("bach10", 30, F, seed=4), x.astype(np.float64), inverse='explicit').numpy()
    f = tmp_path / "case.npz"
    np.savez(f, x=x, F=F)
    res = {}
    jobs = (("h16", {}), ("cf", {"DCS_DECODER_CL": "0"}), ("nox3", {"DCS_DECODER_X3": "0"}), ("f32fc", {"DCS_GEMM_KSPLIT": "40"}))
    outs = {name: str(tmp_path / (name + ".npz")) for name, _ in jobs}
    _run_children(_CL_CHILD, [ROOT, f], [env for _, env in jobs], per_child_args=[[outs[name]] for name, _ in jobs], parallel=2)
    res = {name: np.load(outs[name]) for name, _ in jobs}
    # round 6: under the f16 switch the dense layers themselves run on f16 weights and write D as f16, channels-last, 32 channels
    # per position (gemm_f16.hip).  DCS_DECODER_CL=0 keeps the round-4 form: f32-class dense layers, channel-first D, the same
    # decoder -- the same f16 convolutions behind different dense arithmetic: both inside the stated 2e-3, 2.4e-5 apart
    assert np.max(np.abs(res["h16"]["p"] - want)) < 2e-3
    assert np.max(np.abs(res["h16"]["q"] - want)) < 1e-4          # f32-class after the switch back
    assert bool(res["h16"]["same"])                                 # and the f16 result again, to the bit
    assert np.max(np.abs(res["cf"]["p"] - want)) < 2e-3
    assert np.array_equal(res["h16"]["q"], res["cf"]["q"])          # the f32-class path knows neither layout switch
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "f16_stats.txt"), "a") as fh:
        fh.write("bach10 F=257, 140 tiles, f16 switch: raw output max|err| vs oracle: f16 dense + f16 D %.3e, f32-class dense %.3e; "
                 "between the two %.3e\n" % (np.max(np.abs(res["h16"]["p"] - want)), np.max(np.abs(res["cf"]["p"] - want)),
                                             np.max(np.abs(res["h16"]["p"] - res["cf"]["p"]))))
    res["cl"] = res["h16"]
    # switch off: the f32-class fused decoder (colconv_x3.hip: two waves per column block, taps dealt by parity, three-way
    # split operands) against the two kernels it replaces (DCS_DECODER_X3=0: f32-MFMA column convolution + transposed conv1)
    assert np.max(np.abs(res["nox3"]["q"] - want)) < 1e-4
    assert np.max(np.abs(res["cl"]["q"] - res["nox3"]["q"])) < 2e-6 * max(1.0, float(np.max(np.abs(want))))
    assert np.array_equal(res["cl"]["p"], res["nox3"]["p"])          # (the f16 result does not involve the x3 decoder)
    # the bottleneck layer (140 x 18 810 x 256): bf16 x 3 all-rows kernel with K cut into slices (default from 128 rows and
    # 16 384 columns of A on) against the f32 K-split it replaces (DCS_GEMM_KSPLIT=n forces it)
    assert np.max(np.abs(res["f32fc"]["q"] - want)) < 1e-4
    assert np.max(np.abs(res["cl"]["q"] - res["f32fc"]["q"])) < 2e-6 * max(1.0, float(np.max(np.abs(want))))


_SI_X3_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from deepconvsep_amd.runtime import Network, default_context
from deepconvsep_amd.synth import synth_params
z = np.load(sys.argv[2])
ctx = default_context()
F = int(z['F'])
net = Network(ctx, 'bach10_si1', synth_params('bach10_si1', 30, F, seed=4), 30, F)
out = {}
for n in (140, 128, 300, 64):                       # one launch | one launch | pieces of 150 | below the window: two kernels
    out['n%d' % n] = net.forward_raw(ctx.to_device(z['x'][:n], np.float32)).cpu().numpy()
ctx.timing(['decoder', 'deconv2', 'final'])
net.forward_raw(ctx.to_device(z['x'][:140], np.float32)); ctx.synchronize()
out['decoder_ms'] = np.float64(ctx.timing_query('decoder')[0])
out['deconv2_ms'] = np.float64(ctx.timing_query('deconv2')[0])
np.savez(sys.argv[3], **out)
"""


def test_scoreinformed_decoder_runs_on_the_fused_three_way_split_kernel(tmp_path):
    """The score-informed graph's two InverseLayers (one decoder branch, FOUR output channels) through the fused f32-class
    decoder (colconv_x3.hip, CO = 4: stage 1 once per row, stage 2 / shift-add / carry / store once per output channel with
    that channel's conv1 taps requested one channel ahead): against the float64 oracle 1e-4, against the two kernels it
    replaces (DCS_DECODER_X3=0) 2e-6 relative, for a launch inside the all-rows dense kernel's window (128, 140 tiles), a pass
    cut into pieces (300) and one below the window (64: the two-kernel path either way, identical bits); the timer tags tell
    which path ran."""
    import subprocess
    F = 257
    x = _tiles("bach10_si1", 300, 30, F, seed=23)
    want = net_ref.forward("bach10_si1", synth_params("bach10_si1", 30, F, seed=4), x.astype(np.float64), inverse='explicit').numpy()
    f = tmp_path / "case.npz"
    np.savez(f, x=x, F=F)
    res = {}
    jobs = (("x3", {}), ("two", {"DCS_DECODER_X3": "0"}))
    outs = {name: str(tmp_path / (name + ".npz")) for name, _ in jobs}
    _run_children(_SI_X3_CHILD, [ROOT, f], [env for _, env in jobs], timeout=600, per_child_args=[[outs[name]] for name, _ in jobs], parallel=2)
    res = {name: np.load(outs[name]) for name, _ in jobs}
    scale = max(1.0, float(np.max(np.abs(want))))
    for n in (140, 128, 300, 64):
        k = "n%d" % n
        assert res["x3"][k].shape == want[:n].shape
        assert np.max(np.abs(res["x3"][k] - want[:n])) < 1e-4, n
        assert np.max(np.abs(res["two"][k] - want[:n])) < 1e-4, n
        assert np.max(np.abs(res["x3"][k] - res["two"][k])) < 2e-6 * scale, n
    assert np.array_equal(res["x3"]["n64"], res["two"]["n64"])
    assert float(res["x3"]["decoder_ms"]) > 0 and float(res["x3"]["deconv2_ms"]) == 0      # fused
    assert float(res["two"]["decoder_ms"]) == 0 and float(res["two"]["deconv2_ms"]) > 0    # two kernels


_VARIANT_ENVS = [
    {"DCS_FINAL_CBW": "2"}, {"DCS_FINAL_CBW": "1"},               # 128-bin workgroups (bf16x3 kernel, G split by a pass) / 64-bin (f32)
    {"DCS_FINAL_CBW": "2", "DCS_DECONV2": "2"},                   # bf16x3 kernel fed by the streaming deconv2 (writes the planes)
    {"DCS_DECONV2": "2"}, {"DCS_DECONV2": "1"},                   # streaming / one-shot transposed conv2
    {"DCS_STFT_WAVE_MIN": "1"}, {"DCS_FFT_BLOCK": "1"},           # wave-per-frame / block-level FFT everywhere
    {"DCS_ISTFT_CHAIN": "0", "DCS_ISTFT_SEQ_HOPS": "1"}, {"DCS_ISTFT_CHAIN": "0", "DCS_ISTFT_SEQ_HOPS": "37"},   # blocks per wave of the barrier-free iSTFT
    {"DCS_GRAPH": "1"},                                            # hipGraph replay of the step (opt-in)
    {"DCS_ISTFT_STAGE_MIN": "1"},                                  # spectra through LDS (long clips' iSTFT) on a short clip
    {"DCS_ISTFT_STAGE": "0"},
    {"DCS_ISTFT_CHAIN": "0"}, {"DCS_ISTFT_CHAIN": "3"}, {"DCS_ISTFT_CHAIN": "7"},   # chained iSTFT off / forced frames per wave
    {"DCS_FOLD_CONV2": "0"},                                       # conv2 and the bottleneck layer as two launches (the unfolded weights)
]


def test_kernel_variants_selected_by_size_agree_with_the_oracle(tmp_path):
    """The launchers pick kernel variants by problem size; the debugging switches force each variant on the
    same 3 s clip at both frame sizes (fresh process: the switches are read once) and every one must meet the parity bar."""
    import subprocess
    audio, want = _variant_case()
    f = tmp_path / "case.npz"
    np.savez(f, audio=audio, want1024=want[1024], want2048=want[2048])
    _run_children(_VARIANT_CHILD, [ROOT, f], _VARIANT_ENVS, timeout=200)


_VARIANT_CASE = []


def _variant_case():
    """The 3 s clip of the kernel-variant tests and its oracle output per frame size (computed once per session)."""
    if not _VARIANT_CASE:
        audio = synth_audio(3 * 44100, seed=77)
        audio[40000:52000] = 0.0
        _VARIANT_CASE.append((audio, {N: pipeline.separate("dsd", synth_params("dsd", 30, N // 2 + 1, seed=2), audio, 0.3, 30, 25,
                                                          32, N, 512, np.hanning) for N in (1024, 2048)}))
    return _VARIANT_CASE[0]


_BF16X3_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import deepconvsep_amd as dcs
from deepconvsep_amd.runtime import default_context
z = np.load(sys.argv[2], allow_pickle=True)
N = int(z['N']); F = N // 2 + 1
params = [z['p%d' % i] for i in range(int(z['n_params']))]
sep = dcs.Separator('dsd', params, 0.3, 30, 25, 32, F, N, 512, np.hanning)
ctx = default_context()
a = ctx.to_device(z['audio'], np.float32)
for rep in range(2):
    s_d, m_d, _ = sep.net.separate_spectra(sep.plan, a, 25, sep.tiler, 0.3)
np.save(sys.argv[3], ctx.to_host(s_d))
pcm = sep.separate(z['audio'])
np.save(sys.argv[4], pcm)
"""


_STAGED_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import deepconvsep_amd as dcs
from deepconvsep_amd.synth import synth_audio, synth_params
z = np.load(sys.argv[2])
ok = True
for N in (1024, 2048):
    F = N // 2 + 1
    sep = dcs.Separator('dsd', synth_params('dsd', 30, F, seed=2), 0.3, 30, 25, 32, F, N, 512, np.hanning)
    sep.net.set_latency_stages(0)
    clips = [z['a0'], z['a1'], z['a2'], z['a0'][:len(z['a0']) - 4111]]
    many = sep.separate_many(clips)                  # sorted by length, one ragged group: per-clip frame counts in one launch
    alone = [sep.separate(c) for c in clips]
    same = sep.ctx.to_host(sep.net.separate_batch(sep.plan, sep.ctx.to_device(np.stack([z['a1'], z['a1'][::-1].copy()]), np.float32), 25, sep.tiler, 0.3))
    e_many = max(float(np.max(np.abs(m - a))) for m, a in zip(many, alone))
    e_same = float(np.max(np.abs(same[0] - alone[1])))
    e_ref = float(np.max(np.abs(alone[2] - z['want2_%d' % N])))
    print('N %d: ragged vs alone %.2e, batch vs alone %.2e, vs oracle %.2e' % (N, e_many, e_same, e_ref))
    ok = ok and e_many < 5e-6 and e_same < 5e-6 and e_ref < 1e-4
sys.exit(0 if ok else 3)
"""


_STAGED_CASE = []


_STAGED_ENVS = [{"DCS_ISTFT_STAGE_MIN": "1"}, {"DCS_ISTFT_STAGE": "0"},
                                 {"DCS_ISTFT_CHAIN": "0"}, {"DCS_ISTFT_CHAIN": "3"}, {"DCS_ISTFT_CHAIN": "5"},   # chained iSTFT: off, forced run lengths (seams at other places)
                                 ]


def test_staged_istft_on_ragged_groups_and_batches(tmp_path):
    """The LDS-staged inverse STFT (normally long clips only) forced onto short ones: a ragged group (per-clip frame counts
    from the device table: the four waves of a workgroup must still walk the same frames), an equal-length batch and single
    clips agree with each other, and a single clip with the oracle.  DCS_ISTFT_STAGE=0: the same with the plain loads."""
    import subprocess
    if not _STAGED_CASE:                                 # the clips and the oracle's output for one of them, once per session
        a = [synth_audio(int(44100 * sec) + odd, seed=60 + i) for i, (sec, odd) in enumerate(((1.9, 0), (1.5, 0), (1.7, 313)))]
        _STAGED_CASE.append((a, {N: pipeline.separate("dsd", synth_params("dsd", 30, N // 2 + 1, seed=2), a[2], 0.3, 30, 25, 32, N,
                                                      512, np.hanning) for N in (1024, 2048)}))
    a, want2 = _STAGED_CASE[0]
    f = tmp_path / "case.npz"
    np.savez(f, a0=a[0], a1=a[1], a2=a[2], want2_1024=want2[1024], want2_2048=want2[2048])
    _run_children(_STAGED_CHILD, [ROOT, f], _STAGED_ENVS, timeout=300)


@pytest.mark.parametrize("kind", ["glorot", "sparse", "tiny"])
def test_bf16x3_final_kernel_meets_the_parity_bar(kind, tmp_path):
    """The final kernel of large launches (dsd_bf16x3.hip: bf16 matrix pipe, both operands split exactly into three
    bf16 terms, the six products above 2^-24 kept, f32 accumulation) against the oracle and against the f32 kernel, on
    a fresh Glorot draw and on trained-like / tiny-output weight sets (oracle/cases.py): separated spectrograms within
    the all-bin mask check of tests/maskcheck.py, PCM within 1e-4, and within 1e-6 of the f32 kernel's PCM."""
    import subprocess
    from maskcheck import check_masked
    from oracle import cases
    N, F = 2048, 1025
    audio = synth_audio(3 * 44100, seed=77)
    audio[40000:52000] = 0.0
    mag0, _ = stft_np.compute_file(audio, phase=True, frameSize=N, hopSize=512, window=np.hanning)
    tiles, n = tiling_np.generate_overlapadd(0.3 * mag0.astype(np.float32), F, 30, 25, 32, tiler=tiling_np.SCRIPT, fill=0.0)
    x = tiles.reshape((-1,) + tiles.shape[2:])[:min(n, 8)].astype(np.float32)
    params = cases.calibrate("dsd", 30, F, 52, kind, x)
    want, mm, mag, ph = pipeline.separate("dsd", params, audio, 0.3, 30, 25, 32, N, 512, np.hanning, return_spectra=True)
    f = tmp_path / "case.npz"
    np.savez(f, audio=audio, N=N, n_params=len(params), **{"p%d" % i: p for i, p in enumerate(params)})
    res = {}
    jobs = (("bf16x3", {"DCS_FINAL_CBW": "2", "DCS_DECONV2": "2"}),
            ("f32", {"DCS_FINAL_CBW": "1", "DCS_DECONV2": "2"}))                     # the f32-MFMA kernel (64-bin workgroups)
    outs = {name: (str(tmp_path / (name + "_sep.npy")), str(tmp_path / (name + "_pcm.npy"))) for name, _ in jobs}
    _run_children(_BF16X3_CHILD, [ROOT, f], [env for _, env in jobs], per_child_args=[list(outs[name]) for name, _ in jobs], parallel=2)
    res = {name: (np.load(outs[name][0]), np.load(outs[name][1])) for name, _ in jobs}
    sep16, pcm16 = res["bf16x3"]
    sep32, pcm32 = res["f32"]
    assert np.max(np.abs(pcm16 - want)) < 1e-4
    assert np.max(np.abs(pcm16 - pcm32)) < 1e-6
    if kind == "glorot":
        assert np.max(np.abs(sep16 - mm)) < 1e-4            # per masked bin, every bin
    # the fold is linear with weights in [0, 1] that sum to one, so a bin of the separated spectrogram inherits the
    # conditioning of the masks of the tiles that cover it; checked on the masks of every tile directly:
    T = mm.shape[1]
    assert sep16.shape == mm.shape == (4, T, F)
    assert np.max(np.abs(sep16 - sep32)) < 1e-5 * max(1.0, float(np.max(mag)))
    err = np.abs(sep16 - mm)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "mask_bins.txt"), "a") as fh:
        fh.write("dsd fused path (bf16x3 final kernel), %s weights: %d bins of the separated spectrogram, outside 1e-4: %d, "
                 "max err %.2e; vs the f32 kernel max %.2e\n" % (kind, err[0].size, int((err.max(axis=0) > 1e-4).sum()),
                                                                  err.max(), np.abs(sep16 - sep32).max()))
    # what is measured: 0 bins outside 1e-4 (glorot, sparse), 1 of 267 525 (tiny).  Allowed: 0 / at most 2, and only where the
    # reference itself has (nearly) nothing in every source -- there the mask is discontinuous
    bad = err.max(axis=0) > 1e-4
    assert int(bad.sum()) <= (0 if kind == "glorot" else 2)
    if bad.any():
        assert float(np.max(np.abs(mm[:, bad]))) < 1e-5


# ------------------------------------------------------------------ score-informed path
SI_INSTS = ["bassoon_b", "clarinet_b", "saxophone_b", "violin_b"]


def _golden_scores(g, d):
    for ins, text in zip(SI_INSTS, g["texts"]):
        (d / (ins + ".txt")).write_text(str(text))



def test_scoreinformed_path_with_a_timbre_model_matches_oracle(golden, tmp_path):
    """Separator.separate_scoreinformed(..., timbre_model_path): the template-weighted masks of filterSpec's timbre branch
    (host; bit-exact against the reference's masks in tests/test_oracle_golden.py) through the device path -- STFT, tiles,
    network, masks, cross-fade, iSTFT -- against oracle.pipeline.separate_scoreinformed with the same templates."""
    import pickle
    from deepconvsep_amd import score
    g = golden("score_timbre_n1024_hop512")
    N, hop, L = int(g["frame"]), int(g["hop"]), int(g["n_samples"])
    F = N // 2 + 1
    _golden_scores(g, tmp_path)
    nframes = int(np.ceil(L / float(hop))) + 2
    melody = score.melody_table([i + ".txt" for i in SI_INSTS], str(tmp_path), nframes, 44100, hop, N)
    assert np.array_equal(melody, g["melody"])
    model = tmp_path / "timbre.pkl"
    with open(model, "wb") as fh:
        pickle.dump(g["harmonics"], fh, protocol=2)
    params = synth_params("bach10_si", 30, F, seed=5)
    audio = synth_audio(L, seed=9)
    sep = dcs.Separator("bach10_si", params, 0.3, 30, 25, 32, F, N, hop, np.hanning, tiler='library')
    got = sep.separate_scoreinformed(audio, melody, timbre_model_path=str(model))
    want = pipeline.separate_scoreinformed(params, audio, melody, 0.3, 30, 25, 32, N, hop, np.hanning, harmonics=g["harmonics"])
    assert got.shape == want.shape and np.max(np.abs(got - want)) < 1e-4
    plain = sep.separate_scoreinformed(audio, melody)
    assert np.max(np.abs(plain - got)) > 1e-3                      # the templates do change the separation

@pytest.mark.parametrize("name", ["score_n4096_hop512", "score_n1024_hop512", "score_n2048_hop256"])
def test_score_masks_match_the_reference_filterSpec(golden, name, tmp_path):
    """dcs_score_masks against the masks the reference's own filterSpec produced (golden) and against the oracle's
    mask x spectrogram products: binary float32 masks and single float32 products -- bit exact."""
    from deepconvsep_amd import score
    from oracle import score_np
    g = golden(name)
    N, nframes = int(g["frame"]), int(g["nframes"])
    rs = np.random.RandomState(int(g["mag_seed"]))
    mag = (0.3 * np.abs(rs.randn(nframes, N // 2 + 1)).astype(np.float32)).astype(np.float32)
    ctx = default_context()
    mag_t = ctx.to_device(mag, np.float32)
    inp, mask = score.score_masks(ctx, mag_t, g["melody"], 0, nframes, want_input=True, want_mask=True)
    mask = mask.cpu().numpy()
    assert int((mask == 1).sum()) == int(g["mask_ones"])
    assert np.array_equal(mask.astype(np.float64).sum(axis=1), g["mask_rowsum"])
    assert np.array_equal(mask.astype(np.float64).sum(axis=0), g["mask_colsum"])
    want = score_np.network_input(mag, g["melody"], nframes)
    assert np.array_equal(inp.cpu().numpy().astype(np.float64), want)
    # frame window not starting at 0, one instrument without notes (all-ones mask), drop-in filterSpec signature
    melody2 = g["melody"].copy()
    melody2[2] = 0
    got = score.filterSpec(mag[40:140], melody2, 4, 40, 140)
    assert got.dtype == np.float32 and np.array_equal(got, g["mask_win"])
    # a strided spectrogram (rows padded) and fewer instruments than table rows
    wide = ctx.to_device(np.concatenate([mag, np.zeros((nframes, 3), np.float32)], axis=1), np.float32)
    inp2, _ = score.score_masks(ctx, wide[:, :mag.shape[1]], g["melody"][:2], 0, nframes)
    assert np.array_equal(inp2.cpu().numpy(), inp.cpu().numpy()[:2])
    bad = g["melody"].copy()
    bad[0, 0, 4] = N // 2 + 5                                        # bin range past the spectrum: IndexError in NumPy
    with pytest.raises(ValueError):
        score.score_masks(ctx, mag_t, bad, 0, nframes)


@pytest.mark.parametrize("name", ["score_n1024_hop512", "score_n4096_hop512"])
def test_sum_normalised_score_masks_match_the_trainers_dataset_class(golden, name):
    """dcs_score_masks_norm(DCS_SCORE_NORM_SUM) against the masks the reference's LargeDatasetMask2.filterSpec body produced
    (dataset.py:839-879; tests/golden/score_sum_*.npz): every bin divided by the float32 sum over the instruments -- bit exact,
    with either output alone (the flags of the first two passes live in whichever buffer is written), a frame window that
    does not start at 0 with a silent instrument, and the products mask x spectrogram against the oracle."""
    from deepconvsep_amd import score
    from oracle import score_np
    g, gs = golden(name), golden(name.replace("score_", "score_sum_"))
    N, nframes = int(g["frame"]), int(g["nframes"])
    F = N // 2 + 1
    mag = (0.3 * np.abs(np.random.RandomState(int(g["mag_seed"])).randn(nframes, F)).astype(np.float32)).astype(np.float32)
    ctx = default_context()
    mag_t = ctx.to_device(mag, np.float32)
    inp, mask = score.score_masks(ctx, mag_t, g["melody"], 0, nframes, want_input=True, want_mask=True, normalise='sum')
    assert np.array_equal(mask.cpu().numpy(), gs["mask"])
    want = score_np.network_input(mag, g["melody"], nframes, normalise='sum')
    assert np.array_equal(inp.cpu().numpy().astype(np.float64), want)
    _, only_mask = score.score_masks(ctx, mag_t, g["melody"], 0, nframes, want_input=False, want_mask=True, normalise='sum')
    assert np.array_equal(only_mask.cpu().numpy(), gs["mask"])
    only_inp, _ = score.score_masks(ctx, mag_t, g["melody"], 0, nframes, want_input=True, want_mask=False, normalise='sum')
    assert np.array_equal(only_inp.cpu().numpy(), inp.cpu().numpy())
    melody2 = g["melody"].copy()
    melody2[2] = 0
    got = score.filterSpec(mag[40:140], melody2, 4, 40, 140, normalise='sum')
    assert got.dtype == np.float32 and np.array_equal(got, gs["mask_win"])
    # the default is still the script's normalisation, and an unknown one is refused
    _, mx = score.score_masks(ctx, mag_t, g["melody"], 0, nframes, want_input=False, want_mask=True)
    assert int((mx.cpu().numpy() == 1).sum()) == int(g["mask_ones"])
    with pytest.raises(ValueError):
        score.score_masks(ctx, mag_t, g["melody"], 0, nframes, normalise='mean')


@pytest.mark.parametrize("normalise,mixture", [("sum", "sum"), ("sum", "ch0"), ("max", "sum")])
def test_scoreinformed_separation_with_the_trainers_semantics(normalise, mixture, tmp_path):
    """the whole score-informed path with the harmonic masks divided by their sum over the instruments
    (dataset.py:862) and / or the soft masks applied to the sum of the input channels (trainCNNrwc.py:258-263), one fused call
    and the stage-level composition, against oracle.pipeline.separate_scoreinformed with the same switches -- and the four
    combinations really are four different functions."""
    from deepconvsep_amd import score
    from oracle import score_np
    N, seconds = 1024, 3.0
    F, L = N // 2 + 1, int(seconds * 44100)
    audio = synth_audio(L, seed=91)
    for i, ins in enumerate(SI_INSTS):
        score_np.synth_score(str(tmp_path / (ins + ".txt")), 700 + i, n_notes=16, total=seconds + 0.5, lo=40 + 5 * i,
                             hi=64 + 6 * i)
    nframes = int(np.ceil(L / 512.0)) + 2
    melody = score.melody_table([i + ".txt" for i in SI_INSTS], str(tmp_path), nframes, 44100, 512, N)
    params = synth_params("bach10_si", 30, F, seed=5)
    sep = dcs.Separator("bach10_si", params, 0.3, 30, 25, 32, F, N, 512, np.hanning, tiler='library',
                        score_normalise=normalise, score_mixture=mixture)
    got = sep.separate_scoreinformed(audio, melody)
    want = pipeline.separate_scoreinformed(params, audio, melody, 0.3, 30, 25, 32, N, 512, np.hanning,
                                           normalise=normalise, mixture=mixture)
    assert got.shape == want.shape == (4, L) and np.max(np.abs(got - want)) < 1e-4
    a = sep.ctx.to_device(audio, np.float32)
    staged = sep.ctx.to_host(sep.separate_scoreinformed_device(a, melody, staged=True)).astype(np.float64)
    assert np.max(np.abs(staged - want)) < 1e-4 and np.max(np.abs(staged - got)) < 2e-5
    script = pipeline.separate_scoreinformed(params, audio, melody, 0.3, 30, 25, 32, N, 512, np.hanning)
    assert np.max(np.abs(want - script)) > 1e-3
    # the switch is model state: back to the script's semantics on the same handle
    sep.net.set_score_semantics('max', 'ch0')
    sep.score_normalise = 'max'
    assert np.max(np.abs(sep.separate_scoreinformed(audio, melody) - script)) < 1e-4
    with pytest.raises(ValueError):
        sep.net.set_score_semantics('sum', 'all')
    with pytest.raises(NotImplementedError):
        dcs.Separator("bach10", synth_params("bach10", 30, F, seed=5), 0.3, 30, 25, 32, F, N, 512, np.hanning,
                      score_mixture='sum')


def test_single_branch_score_informed_model_and_pruned_17_array_model():
    """The 11-array .pkl of trainCNNrwc_samp.py:195-235 loads through the same 'bach10_si' entry (arch.resolve), and a
    17-array .pkl is instantiated as its live part only (arch.live_params: branch 0; three dense layers never uploaded):
    network output and masked sources, script and trainer mixtures, against the float64 oracle; live_only=False keeps all
    16 channels and its first four agree with the pruned model to fp32 rounding."""
    tc, F, n = 30, 257, 3
    params17 = synth_params("bach10_si", tc, F, seed=3)
    params11 = synth_params("bach10_si1", tc, F, seed=4)
    x = _tiles("bach10_si", n, tc, F, seed=12)
    ctx = default_context()
    xd = ctx.to_device(x, np.float32)
    one = Network(ctx, "bach10_si", params11, tc, F)
    assert one.arch.name == "bach10_si1" and one.out_channels == 4 and one.S == 4
    p = one.forward_raw(xd).cpu().numpy()
    want = net_ref.forward("bach10_si1", params11, x.astype(np.float64), inverse='explicit').numpy()
    assert p.shape == want.shape == (n, 4, tc, F) and np.max(np.abs(p - want)) < 1e-4
    for mixture in ("ch0", "sum"):
        one.set_score_semantics('max', mixture)
        got = one.forward_masked(xd).cpu().numpy()
        ref = net_ref.predict("bach10_si1", params11, x.astype(np.float64), inverse='explicit', mixture=mixture)
        mix = x[:, 0].astype(np.float64) if mixture == "ch0" else x.astype(np.float64).sum(axis=1)
        _assert_masked(got, ref, want, mix, 4, p_got=p, conv='B', label="bach10_si1 %s" % mixture)
    pruned = Network(ctx, "bach10_si", params17, tc, F)
    assert pruned.graph_arch.name == "bach10_si" and pruned.arch.name == "bach10_si1" and pruned.out_channels == 4
    full = Network(ctx, "bach10_si", params17, tc, F, live_only=False)
    assert full.arch.name == "bach10_si" and full.out_channels == 16
    pf = full.forward_raw(xd).cpu().numpy()
    pp = pruned.forward_raw(xd).cpu().numpy()
    w17 = net_ref.forward("bach10_si", params17, x.astype(np.float64), inverse='explicit').numpy()
    assert np.max(np.abs(pf - w17)) < 1e-4 and np.max(np.abs(pp - w17[:, :4])) < 1e-4
    assert np.max(np.abs(pp - pf[:, :4])) < 2e-6 * max(1.0, float(np.max(np.abs(w17))))
    assert np.max(np.abs(pruned.forward_masked(xd).cpu().numpy() - full.forward_masked(xd).cpu().numpy())) < 2e-6
    with pytest.raises(ValueError):
        Network(ctx, "bach10_si", params17[:12], tc, F)               # neither layout: set_all_param_values would raise


@pytest.mark.parametrize("N,seconds", [(1024, 3.0), (4096, 2.0)])
def test_scoreinformed_separation_matches_oracle(N, seconds, tmp_path):
    """The whole score-informed path (STFT, harmonic masks x spectrogram, 4-channel library tiling, network, mask on
    input channel 0, cross-fade, iSTFT) against oracle.pipeline.separate_scoreinformed."""
    from scipy.signal.windows import blackmanharris
    from deepconvsep_amd import score
    from oracle import score_np
    F = N // 2 + 1
    L = int(seconds * 44100)
    audio = synth_audio(L, seed=91)
    audio[L // 2: L // 2 + 6000] = 0.0
    for i, ins in enumerate(SI_INSTS):
        score_np.synth_score(str(tmp_path / (ins + ".txt")), 700 + i, n_notes=16, total=seconds + 0.5, lo=40 + 5 * i,
                             hi=64 + 6 * i)
    nframes = int(np.ceil(L / 512.0)) + 2
    melody = score.melody_table([i + ".txt" for i in SI_INSTS], str(tmp_path), nframes, 44100, 512, N)
    params = synth_params("bach10_si", 30, F, seed=5)
    sep = dcs.Separator("bach10_si", params, 0.3, 30, 25, 32, F, N, 512, blackmanharris, tiler='library')
    got = sep.separate_scoreinformed(audio, melody)                      # one dcs_separate_scoreinformed call
    want = pipeline.separate_scoreinformed(params, audio, melody, 0.3, 30, 25, 32, N, 512, blackmanharris)
    assert got.shape == want.shape == (4, L)
    assert np.max(np.abs(got - want)) < 1e-4
    assert np.max(np.abs(want)) > 1e-3
    # the same path composed from the stage-level operators, batch_size tiles at a time (phase angles instead of
    # unit phasors between STFT and iSTFT, separate mask and cross-fade kernels)
    a = sep.ctx.to_device(audio, np.float32)
    staged = sep.ctx.to_host(sep.separate_scoreinformed_device(a, melody, staged=True)).astype(np.float64)
    assert np.max(np.abs(staged - want)) < 1e-4
    assert np.max(np.abs(staged - got)) < 2e-5
    with pytest.raises(ValueError):
        sep.separate_scoreinformed(audio, melody[:3])


def test_scoreinformed_command_line(tmp_path):
    """examples/bach10_scoreinformed/separate_bach10.py end to end on a 1.5 s wav with its four score files."""
    import importlib.util
    import scipy.io.wavfile
    from scipy.signal.windows import blackmanharris
    from oracle import score_np
    L = int(1.5 * 44100)
    audio = synth_audio(L, seed=92)
    wav = tmp_path / "mix.wav"
    scipy.io.wavfile.write(str(wav), 44100, (audio * 32767).astype(np.int16))
    for i, ins in enumerate(SI_INSTS):
        score_np.synth_score(str(tmp_path / (ins + ".txt")), 800 + i, n_notes=10, total=2.0, lo=40 + 5 * i, hi=64 + 6 * i)
    params = synth_params("bach10_si", 30, 2049, seed=6)
    model = tmp_path / "model.pkl"
    dcs.save_model(str(model), params)
    out = tmp_path / "out"
    out.mkdir()
    spec = importlib.util.spec_from_file_location("si_cli", os.path.join(ROOT, "examples", "bach10_scoreinformed",
                                                                         "separate_bach10.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    cli.main(["-i", str(wav), "-o", str(out), "-m", str(model)])
    sr, a16 = scipy.io.wavfile.read(str(wav))
    a = a16.astype('float') / 32767
    nframes = int(np.ceil(len(a) / 512.0)) + 2
    melody = score_np.melody_table([str(tmp_path / (i + ".txt")) for i in SI_INSTS], nframes, 44100, 512, 4096)
    want = pipeline.separate_scoreinformed(params, a, melody, 0.3, 30, 25, 32, 4096, 512, blackmanharris)
    for i, s in enumerate(["bassoon", "clarinet", "saxphone", "violin"]):
        sr2, got = scipy.io.wavfile.read(str(out / ("mix_%s.wav" % s)))
        assert sr2 == 44100 and got.dtype == np.int16 and len(got) == L
        assert np.max(np.abs(got.astype(np.int64) - (want[i] * 32767).astype('int16').astype(np.int64))) <= 2


def test_pcm_to_int16_is_the_scripts_wav_format():
    """dcs_pcm_to_int16 = (audio_out * 32767).astype('int16') of separate_dsd.py:307-309: truncation toward zero,
    no clipping (out-of-range values wrap like NumPy's cast), any length / alignment."""
    from deepconvsep_amd.runtime import pcm_to_int16
    ctx = default_context()
    rs = np.random.RandomState(3)
    for n in (1, 3, 4, 5, 1023, 94208, 94211):
        x = rs.uniform(-1.0, 1.0, n).astype(np.float32)
        x[:min(n, 4)] = np.array([0.99999, -0.99999, 3.05e-5, -3.06e-5], np.float32)[:min(n, 4)]
        if n > 100:
            x[50:54] = np.array([1.0, -1.0, 1.2, -1.3], np.float32)          # clipping input: wraps, as in the reference
        want = (x.astype(np.float64) * 32767.0).astype(np.int32).astype(np.int16)      # the scripts' float64 product, exact
        got = pcm_to_int16(ctx, ctx.to_device(x, np.float32)).cpu().numpy()
        assert got.dtype == np.int16 and np.array_equal(got, want)
        if n > 8:                                                             # unaligned views
            t = ctx.to_device(x, np.float32)
            assert np.array_equal(pcm_to_int16(ctx, t[1:]).cpu().numpy(), want[1:])
    # = what the float path writes: the float32 PCM widened to float64, times 32767, truncated -- identical, not "one LSB apart"
    x = rs.uniform(-1.0, 1.0, 50000).astype(np.float32)
    ref = (x.astype(np.float64) * 32767).astype('int16')
    got = pcm_to_int16(ctx, ctx.to_device(x, np.float32)).cpu().numpy()
    assert np.array_equal(got, ref)


# ------------------------------------------------------------------ stereo (ILD) graph, SURVEY 8f-4
@pytest.mark.parametrize("N,seconds,hop", [(1024, 2.0, 512), (2048, 1.2, 512), (512, 1.0, 200)])
def test_stereo_ild_separation_matches_oracle(N, seconds, hop):
    """dcs_separate_stereo (examples/dsd100_2ch_ILD/trainCNN_ILD_DSD100.py:291-325 on the DSD kernel family:
    per-channel STFT, 2-channel library tiles, 4 branches x 2 channels, per-channel masks, per-channel phase)
    against oracle.pipeline.separate_stereo: cross-faded magnitudes per masked bin and PCM."""
    F = N // 2 + 1
    L = int(seconds * 44100)
    stereo = synth_audio(L, seed=61, channels=2)
    stereo[L // 3: L // 3 + 5000, 1] = 0.0                      # one channel silent for a while
    params = synth_params("dsd_ild", 30, F, seed=7)
    sep = dcs.Separator("dsd_ild", params, 0.3, 30, 25, 32, F, N, hop, np.hanning, tiler='library')
    want, spectra, mag, ph = pipeline.separate_stereo(params, stereo, 0.3, 30, 25, 32, N, hop, np.hanning,
                                                      return_spectra=True)
    a = sep.ctx.to_device(np.ascontiguousarray(stereo.T), np.float32)
    pcm, spec = sep.net.separate_stereo(sep.plan, a, 25, sep.tiler, 0.3, want_spectra=True)
    spec = spec.cpu().numpy()                                   # [2, S, T, F]
    assert spec.shape == (2, 4, mag.shape[1], F)
    assert np.max(np.abs(spec - spectra.transpose(1, 0, 2, 3))) < 1e-4
    got = sep.separate_stereo(stereo)
    assert got.shape == want.shape == (L, 4, 2)
    assert np.max(np.abs(got - want)) < 1e-4
    assert np.max(np.abs(want)) > 1e-3
    # the two channels are separated with their own masks: swapping the input channels swaps the outputs
    # only if the network is symmetric in them -- it is not; check instead that channel 1 differs from channel 0
    assert np.max(np.abs(got[:, :, 0] - got[:, :, 1])) > 1e-3


def test_stereo_ild_model_rejects_other_entry_points():
    params = synth_params("dsd_ild", 30, 513, seed=7)
    sep = dcs.Separator("dsd_ild", params, 0.3, 30, 25, 32, 513, 1024, 512, np.hanning, tiler='library')
    tiles = sep.ctx.to_device(np.zeros((2, 2, 30, 513), np.float32), np.float32)
    with pytest.raises((NotImplementedError, ValueError)):       # its mask is not one of the two script conventions
        sep.net.forward_masked(tiles)
    with pytest.raises(NotImplementedError):                      # tile-level operator: fused stereo path only
        sep.net.forward_raw(tiles)
    mono = dcs.Separator("dsd", synth_params("dsd", 30, 513, seed=2), 0.3, 30, 25, 32, 513, 1024, 512, np.hanning)
    with pytest.raises(ValueError):
        mono.net.separate_stereo(mono.plan, sep.ctx.to_device(np.zeros((2, 30000), np.float32), np.float32), 25)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_randomised_lengths_and_batches_match_oracle(seed):
    """Size-dependent launch decisions (hop-blocks per iSTFT workgroup and its block -> (chunk, source) map, column
    groups of the final kernel, row counts that are not multiples of anything): random clip lengths, both tilers,
    both frame sizes, single clips and stacked clips against the oracle."""
    rs = np.random.RandomState(1000 + seed)
    N = int(rs.choice([1024, 2048]))
    F = N // 2 + 1
    tiler = str(rs.choice(["script", "library"]))
    L = int(rs.randint(16500, 120000))
    clips = int(rs.choice([1, 2, 5, 7]))
    params = synth_params("dsd", 30, F, seed=2)
    sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, 512, np.hanning, tiler=tiler)
    audio = np.stack([synth_audio(L, seed=2000 + 10 * seed + c) for c in range(clips)]).astype(np.float32)
    buf = sep.ctx.to_device(audio, np.float32)
    got = (sep.net.separate_batch(sep.plan, buf, 25, sep.tiler, 0.3) if clips > 1
           else sep.net.separate(sep.plan, buf[0], 25, sep.tiler, 0.3)[None]).cpu().numpy()
    for c in sorted(set([0, clips - 1])):
        want = pipeline.separate("dsd", params, audio[c].astype(np.float64), 0.3, 30, 25, 32, N, 512, np.hanning,
                                 tiler=tiling_np.SCRIPT if tiler == 'script' else tiling_np.LIBRARY)
        assert got[c].shape == want.shape
        assert np.max(np.abs(got[c] - want)) < 1e-4, (N, tiler, L, clips, c)
