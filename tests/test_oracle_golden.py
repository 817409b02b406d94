"""Pin the CPU oracle against outputs of the reference's own code
(tests/golden/*.npz, produced by tests/golden/make_golden.py) and, when the
reference tree is present (build container), against the live reference code."""
import numpy as np
import pytest

from oracle import ref_exec, stft_np, tiling_np

STFT_CASES = ["stft_n1024_hann", "stft_n2048_hann", "stft_n4096_bh",
              "stft_n1024_hop256_hann", "stft_n512_hop200_hann"]
TILE_CASES = ["tile_t83_tc30_ov25", "tile_t61_tc30_ov20", "tile_t30_tc30_ov25", "tile_t131_tc30_ov25"]
OLA_CASES = ["ola_n11_tc30_ov25_s4", "ola_n7_tc30_ov20_s2", "ola_n1_tc30_ov25_s4", "ola_n37_tc30_ov25_s4"]


@pytest.mark.parametrize("name", STFT_CASES)
def test_stft_matches_reference_bitwise(golden, name):
    g = golden(name)
    N, hop = int(g["frame"]), int(g["hop"])
    mag, ph = stft_np.compute_file(g["audio"], phase=True, frameSize=N, hopSize=hop, window=g["window"])
    assert mag.shape == g["mag"].shape == (stft_np.frame_count(g["audio"].size, hop), stft_np.n_bins(N))
    assert np.array_equal(mag, g["mag"])
    assert np.array_equal(ph, g["phase"])


@pytest.mark.parametrize("name", STFT_CASES)
def test_istft_matches_reference_bitwise(golden, name):
    g = golden(name)
    N, hop = int(g["frame"]), int(g["hop"])
    back = stft_np.compute_inverse(g["mag"], g["phase"], frameSize=N, hopSize=hop, window=g["window"])
    assert back.size == stft_np.inverse_length(g["mag"].shape[0], hop, N)
    assert np.array_equal(back, g["inverse"])
    # README.md:60-70 round trip: the inverse reproduces the input on its support
    L = g["audio"].size
    assert np.max(np.abs(back[:L] - g["audio"])) < 1e-12


@pytest.mark.parametrize("name", TILE_CASES)
def test_tilers_match_reference(golden, name):
    g = golden(name)
    tc, ov, B = int(g["tc"]), int(g["ov"]), int(g["B"])
    mag = g["mag"]
    fb, n = tiling_np.generate_overlapadd(mag, mag.shape[-1], tc, ov, B, tiler=tiling_np.SCRIPT)
    assert n == int(g["script_n"])
    flat = fb.reshape((-1,) + fb.shape[2:])[:n]
    assert np.array_equal(flat, g["script_tiles"])
    fb, n = tiling_np.generate_overlapadd(mag, mag.shape[-1], tc, ov, B, tiler=tiling_np.LIBRARY)
    assert n == int(g["library_n"])
    assert np.array_equal(fb, g["library_batches"])
    fb, n = tiling_np.generate_overlapadd(g["mag3"], mag.shape[-1], tc, ov, B, tiler=tiling_np.LIBRARY)
    assert n == int(g["library3_n"])
    assert np.array_equal(fb, g["library3_batches"])


@pytest.mark.parametrize("name", OLA_CASES)
def test_overlapadd_matches_reference_bitwise(golden, name):
    g = golden(name)
    n, ov, B = int(g["n"]), int(g["ov"]), int(g["B"])
    sep = tiling_np.overlapadd_multi(g["out"], n, ov)
    assert np.array_equal(sep, g["sep"])
    if "sep1" in g.files:
        s1, s2 = tiling_np.overlapadd(g["out"], n, ov)
        assert np.array_equal(s1, g["sep1"]) and np.array_equal(s2, g["sep2"])
    # the frame-parallel closed form (what the GPU kernel implements) is bit-identical too
    out = g["out"]
    S = out.shape[1]
    tiles = np.stack([out[i // B, :, i % B, 0] for i in range(n)], axis=1)   # [S, n, tc, F]
    for s in range(S):
        assert np.array_equal(tiling_np.overlapadd_frame_parallel(tiles[s], ov), g["sep"][s])


@pytest.mark.parametrize("name", ["score_n1024_hop512", "score_n4096_hop512"])
def test_sum_normalised_masks_match_the_trainers_dataset_class(golden, name, tmp_path):
    """the score-informed TRAINERS feed masks divided by the sum over the instruments
    (LargeDatasetMask2.filterSpec, dataset.py:839-879 -- fixtures made by that method body, make_golden.py score_sum): the
    oracle's restatement and the product's host implementation (timbre branch; the binary branch is the device kernel,
    tests/test_gpu_parity.py) reproduce them bit for bit, and they differ from the script's max-normalised masks."""
    from oracle import score_np
    from deepconvsep_amd import score
    g, gs = golden(name), golden(name.replace("score_", "score_sum_"))
    nframes, F = int(g["nframes"]), int(g["frame"]) // 2 + 1
    mag = (0.3 * np.abs(np.random.RandomState(int(g["mag_seed"])).randn(nframes, F)).astype(np.float32)).astype(np.float32)
    mask = score_np.filterSpec(mag, g["melody"], 4, 0, nframes, normalise='sum')
    assert mask.dtype == np.float32 and np.array_equal(mask, gs["mask"])
    # the values a four-instrument score can produce: 1 / k and 1e-18 / k for k playing instruments, 1e-18 / (4 x 1e-18) = 0.25
    assert set(np.unique(mask).tolist()) <= set(np.float32(v) for v in gs["values"].tolist())
    cols = mask.reshape(nframes, 4, F).astype(np.float64).sum(axis=1)
    assert np.max(np.abs(cols - 1.0)) < 1e-6                          # the masks of a bin sum to one: what "sum" buys
    melody2 = g["melody"].copy()
    melody2[2] = 0
    assert np.array_equal(score_np.filterSpec(mag[40:140], melody2, 4, 40, 140, normalise='sum'), gs["mask_win"])
    assert not np.array_equal(mask, score_np.filterSpec(mag, g["melody"], 4, 0, nframes))
    if "mask_timbre" in gs.files:
        t = golden("score_timbre_n1024_hop512")
        assert np.array_equal(score_np.filterSpec(mag, t["melody"], 4, 0, nframes, t["harmonics"], normalise='sum'), gs["mask_timbre"])
        import pickle
        model = tmp_path / "timbre.pkl"
        with open(str(model), "wb") as fh:
            pickle.dump(t["harmonics"], fh, protocol=2)
        got = score.filterSpec(mag, t["melody"], 4, 0, nframes, timbre_model_path=str(model), normalise='sum')
        assert got.dtype == np.float32 and np.array_equal(got, gs["mask_timbre"])


@pytest.mark.skipif(not ref_exec.available(), reason="reference tree only exists in the build container")
def test_live_trainer_filterspec_agrees_on_fresh_scores():
    """The reference's LargeDatasetMask2.filterSpec executed now against the restatement: seeded rectangles that overlap
    within and across instruments, windows that do not start at frame 0, an instrument without notes."""
    from oracle import score_np
    f = ref_exec.dataset_filterspec_sum()
    for seed in range(6):
        rs = np.random.RandomState(700 + seed)
        T, F, ninst = int(rs.randint(40, 90)), int(rs.randint(60, 300)), int(rs.randint(2, 6))
        mag = rs.rand(T, F).astype(np.float32)
        notes = np.zeros((ninst, 6, 43))
        for j in range(ninst):
            for p in range(0 if j == ninst - 1 and seed % 2 else int(rs.randint(1, 6))):
                b = int(rs.randint(0, T + 20))
                notes[j, p, :3] = (b, b + int(rs.randint(1, 30)), 40 + p)
                for k in range(int(rs.randint(1, 12))):
                    f0 = int(rs.randint(0, F - 16))
                    notes[j, p, 3 + 2 * k], notes[j, p, 4 + 2 * k] = f0, f0 + int(rs.randint(1, 16))
        start = int(rs.randint(0, 10))
        want = f(mag[start:], notes, start, T)
        got = score_np.filterSpec(mag[start:], notes, ninst, start, T, normalise='sum')
        assert want.dtype == got.dtype == np.float32 and np.array_equal(want, got)


@pytest.mark.skipif(not ref_exec.available(), reason="reference tree only exists in the build container")
def test_live_reference_agrees_on_fresh_input():
    rs = np.random.RandomState(99)
    audio = rs.uniform(-1, 1, 7001)
    ref = ref_exec.script_dsd()
    for N, hop in [(1024, 512), (2048, 512)]:
        m0, p0 = ref.compute_file(audio, phase=True, frameSize=N, hopSize=hop)
        m1, p1 = stft_np.compute_file(audio, phase=True, frameSize=N, hopSize=hop)
        assert np.array_equal(m0, m1) and np.array_equal(p0, p1)
        assert np.array_equal(ref.compute_inverse(m0, p0, frameSize=N, hopSize=hop),
                              stft_np.compute_inverse(m1, p1, frameSize=N, hopSize=hop))


# ------------------------------------------------------------------ score-informed front-end
SCORE_CASES = ["score_n4096_hop512", "score_n1024_hop512", "score_n2048_hop256"]
SCORE_INSTS = ["bassoon_b", "clarinet_b", "saxophone_b", "violin_b"]


def write_scores(g, d):
    paths = []
    for ins, text in zip(SCORE_INSTS, g["texts"]):
        p = d / (ins + ".txt")
        p.write_text(str(text))
        paths.append(str(p))
    return paths


def golden_mag(g):
    N = int(g["frame"])
    rs = np.random.RandomState(int(g["mag_seed"]))
    return (0.3 * np.abs(rs.randn(int(g["nframes"]), N // 2 + 1)).astype(np.float32)).astype(np.float32)


@pytest.mark.parametrize("name", SCORE_CASES)
def test_score_oracle_matches_reference_tables_and_masks(golden, name, tmp_path):
    from oracle import score_np
    g = golden(name)
    paths = write_scores(g, tmp_path)
    N, hop, nframes = int(g["frame"]), int(g["hop"]), int(g["nframes"])
    for i, p in enumerate(paths):
        t = score_np.expandMidi(p, 0, 40.0, 50, 440, 20, 44100, hop, N, 0.2, 0.2, nframes, 0.5)
        assert score_np.getMidiNum(p, 0, 40.0) == int(g["nums"][i])
        assert np.array_equal(t, g["melody"][i, :t.shape[0]])
        assert not g["melody"][i, t.shape[0]:].any()
    assert np.array_equal(score_np.melody_table(paths, nframes, 44100, hop, N), g["melody"])
    mag = golden_mag(g)
    mask = score_np.filterSpec(mag, g["melody"], 4, 0, nframes)
    assert mask.dtype == np.float32 and int((mask == 1).sum()) == int(g["mask_ones"])
    assert np.array_equal(mask.astype(np.float64).sum(axis=1), g["mask_rowsum"])
    assert np.array_equal(mask.astype(np.float64).sum(axis=0), g["mask_colsum"])
    melody2 = g["melody"].copy()
    melody2[2] = 0
    assert np.array_equal(score_np.filterSpec(mag[40:140], melody2, 4, 40, 140), g["mask_win"])


def test_timbre_model_masks_match_the_reference_filterSpec(golden, tmp_path):
    """The timbre-model branch of filterSpec (separate_bach10.py:173-175,189-191): the fixture holds the masks the
    reference's own code produced from a seeded float32 template array; the oracle's restatement and the SHIPPED host
    implementation (deepconvsep_amd.score.filterSpec with a pickled model; pure NumPy, no GPU) both match bit for bit, on the
    whole clip and on a frame window with a silent instrument."""
    import pickle
    from oracle import score_np
    from deepconvsep_amd import score
    g = golden("score_timbre_n1024_hop512")
    nframes = int(g["nframes"])
    mag = golden_mag(g)
    H = g["harmonics"]
    assert np.array_equal(score_np.filterSpec(mag, g["melody"], 4, 0, nframes, H), g["mask"])
    melody2 = g["melody"].copy()
    melody2[2] = 0
    assert np.array_equal(score_np.filterSpec(mag[40:140], melody2, 4, 40, 140, H), g["mask_win"])
    model = tmp_path / "timbre.pkl"
    with open(model, "wb") as fh:
        pickle.dump(H, fh, protocol=2)
    got = score.filterSpec(mag, g["melody"], 4, 0, nframes, timbre_model_path=str(model))
    assert got.dtype == np.float32 and np.array_equal(got, g["mask"])
    assert np.array_equal(score.filterSpec(mag[40:140], melody2, 4, 40, 140, timbre_model_path=str(model)), g["mask_win"])
    assert len(np.unique(g["mask"])) > 100 and float(g["mask"].max()) == 1.0      # weighted, not binary; normalised per instrument


@pytest.mark.skipif(not ref_exec.available(), reason="reference tree only exists in the build container")
def test_live_reference_timbre_branch_agrees_on_fresh_models(tmp_path):
    import pickle
    from oracle import score_np
    from deepconvsep_amd import score
    sc = ref_exec.score()
    for seed in range(6):
        rs = np.random.RandomState(900 + seed)
        p = score_np.synth_score(str(tmp_path / "t_b.txt"), 50 + seed, n_notes=25, total=6.0)
        tab = sc.expandMidi("t_b", str(tmp_path), 0, 40.0, 50, 440, 20, 44100, 512, 2048, 0.2, 0.2, 520, 0.5)
        melody = np.zeros((2, tab.shape[0], 43))
        melody[0] = tab
        melody[1, : tab.shape[0] // 2] = tab[: tab.shape[0] // 2]
        H = rs.uniform(0.0, 2.0, (2, 128, 20)).astype(np.float32)
        model = str(tmp_path / ("m%d.pkl" % seed))
        with open(model, "wb") as fh:
            pickle.dump(H, fh, protocol=2)
        mag = np.abs(rs.randn(520, 1025)).astype(np.float32)
        want = sc.filterSpec(mag[30:400], melody, 2, 30, 400, timbre_model_path=model)
        assert np.array_equal(score_np.filterSpec(mag[30:400], melody, 2, 30, 400, H), want)
        assert np.array_equal(score.filterSpec(mag[30:400], melody, 2, 30, 400, timbre_model_path=model), want)


@pytest.mark.skipif(not ref_exec.available(), reason="reference tree only exists in the build container")
def test_live_reference_score_code_agrees_on_fresh_scores(tmp_path):
    """The reference's own expandMidi / getMidiNum / filterSpec (Python-2 bodies executed with the shims of
    oracle.ref_exec.score) against the restatement, many seeded scores, three (frameSize, hop) pairs and a window
    that does not start at zero.  For sr = 44100 and these hops Python 2's integer ``samplerate / hop`` and Python 3's
    true division give the same frame numbers, which is what makes the executed reference a valid pin."""
    from oracle import score_np
    sc = ref_exec.score()
    for hop in (512, 256, 441):
        assert round(float(44100 / hop)) == round(float(44100 // hop))
        assert int(round(0.2 * float(44100 / hop))) == int(round(0.2 * float(44100 // hop)))
    for seed in range(25):
        p = score_np.synth_score(str(tmp_path / "s_b.txt"), seed, n_notes=30, total=12.0 if seed % 2 else 9.5)
        for hop, N, nfr in ((512, 4096, 864), (256, 1024, 1725), (441, 2048, 1002)):
            a = sc.expandMidi("s_b", str(tmp_path), 0, 40.0, 50, 440, 20, 44100, hop, N, 0.2, 0.2, nfr, 0.5)
            b = score_np.expandMidi(p, 0, 40.0, 50, 440, 20, 44100, hop, N, 0.2, 0.2, nfr, 0.5)
            assert np.array_equal(a, b)
            assert sc.getMidiNum("s_b", str(tmp_path), 0, 40.0) == score_np.getMidiNum(p, 0, 40.0)
        a = sc.expandMidi("s_b", str(tmp_path), 2.0, 8.0, 30, 442, 20, 44100, 512, 4096, 0.1, 0.3, 500, 0.0)
        b = score_np.expandMidi(p, 2.0, 8.0, 30, 442, 20, 44100, 512, 4096, 0.1, 0.3, 500, 0.0)
        assert np.array_equal(a, b)
    for note in ("C4", "A4", "Bb3", "F#5", "Cx2", "Db1", "B0", "E10"[:3]):
        assert sc.str2midi(note) == score_np.str2midi(note)
