"""CPU-only checks of the host layer: the C-ABI library loads and exports every symbol
include/dcs.h declares, the framing integers agree with the oracle, the product fails loudly
without a GPU, and the script surface behaves like the reference's (no compute calls here)."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

import deepconvsep_amd as dcs
from deepconvsep_amd import _lib
from deepconvsep_amd.arch import TILER_LIBRARY, TILER_SCRIPT
from oracle import stft_np, tiling_np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.header_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.dcs_version() >= 100


def test_library_exports_nothing_but_the_c_abi():
    """libdcs.so is built with -fvisibility=hidden, DCS_API on the declarations of include/dcs.h and a linker version script
    (csrc/libdcs.map): its dynamic symbol table is the C ABI and the three trace dumps of the experiment builds -- no C++
    internals, no std:: template instantiations."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == sorted(_lib.header_symbols() + ["final_trace_dump", "fftw_trace_dump", "lat_trace_dump"]), exported


def test_framing_integers_match_the_oracle():
    for hop in (200, 256, 512):
        for L in list(range(0, 2100, 37)) + [441000, 441001, 10 ** 7 + 3]:
            assert _lib.frame_count(L, hop) == stft_np.frame_count(L, hop)
    for N, hop in ((1024, 512), (2048, 512), (4096, 512), (1024, 256), (512, 200)):
        for T in (1, 2, 3, 17, 864):
            assert _lib.inverse_length(T, hop, N) == stft_np.inverse_length(T, hop, N)


def test_tile_counts_match_both_reference_tilers():
    for tc, ov in ((30, 25), (30, 20), (30, 10), (20, 19), (8, 1)):
        for T in range(0, 260):
            assert _lib.tile_count(T, tc, ov, TILER_SCRIPT) == len(tiling_np.tile_starts(T, tc, ov, tiling_np.SCRIPT))
            assert _lib.tile_count(T, tc, ov, TILER_LIBRARY) == len(tiling_np.tile_starts(T, tc, ov, tiling_np.LIBRARY))
    # 10 s at 44.1 kHz -> 864 frames -> 167 tiles (ov=25) / 84 tiles (ov=20)
    assert _lib.frame_count(441000, 512) == 864
    assert _lib.tile_count(864, 30, 25, TILER_SCRIPT) == 167
    assert _lib.tile_count(864, 30, 20, TILER_SCRIPT) == 84


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dcs.transformFFT(frameSize=1024, hopSize=512).compute_file(np.zeros(2048))
    with pytest.raises(RuntimeError):
        dcs.generate_overlapadd(np.zeros((40, 5), np.float32), input_size=5)


def test_transform_object_keeps_reference_attributes():
    tt = dcs.transformFFT(frameSize=2048, hopSize=512, sampleRate=44100)
    assert (tt.frameSize, tt.hopSize, tt.sampleRate, tt.ttype, tt.suffix) == (2048, 512, 44100, 'fft', '')
    assert tt.window.shape == (2048,) and np.array_equal(tt.window, np.hanning(2048))
    assert dcs.TransformFFT is dcs.transformFFT
    np.testing.assert_allclose(dcs.sinebell(8), np.sin(np.pi * np.arange(8) / 8.0))


def test_compute_transform_dispatches_through_a_subclass_compute_file(tmp_path):
    """transform.py:107-121: compute_transform calls self.compute_file once per column, so a subclass that customises
    compute_file is honoured (the one-launch device path is only taken for transformFFT's own compute_file)."""
    calls = []

    class Mine(dcs.transformFFT):
        def compute_file(self, audio, phase=False, sampleRate=44100):
            calls.append(np.array(audio))
            mag = np.full((3, 5), float(len(calls)))
            return (mag, -mag) if phase else mag

    tt = Mine(frameSize=8, hopSize=4, suffix='x')
    audio = np.arange(12.0).reshape(6, 2)
    mags, phs = tt.compute_transform(audio, phase=True, save=False)
    assert len(calls) == 2 and np.array_equal(calls[1], audio[:, 1])
    assert mags.shape == (2, 3, 5) and np.all(mags[1] == 2.0) and np.all(phs[0] == -1.0)
    out = str(tmp_path / "song.data")
    assert tt.compute_transform(audio, out_path=out, phase=False, save=True) is None
    assert tt.shape == (2, 3, 5) and os.path.isfile(str(tmp_path / "song_x_m_.data"))


def test_model_pickle_round_trip_py2_protocol(tmp_path):
    from deepconvsep_amd.synth import synth_params
    params = synth_params("dsd", 30, 33)
    path = str(tmp_path / "model.pkl")
    dcs.save_model(path, params)
    back = dcs.load_model(path)
    assert len(back) == 15 and all(np.array_equal(a, b) for a, b in zip(params, back))
    with open(path, "rb") as fh:
        assert pickle.load(fh, encoding="latin1")[0].dtype == np.float32


@pytest.mark.parametrize("script", ["examples/dsd100/separate_dsd.py", "examples/ikala/separate_ikala.py",
                                    "examples/bach10/separate_bach10.py", "examples/hiphopss/separate_hhds.py"])
def test_cli_usage_and_bad_option_exit_code(script):
    exe = os.path.join(ROOT, script)
    r = subprocess.run([sys.executable, exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "-i <inputfile> -o <outputdir> -m <path_to_model.pkl>" in r.stdout
    r = subprocess.run([sys.executable, exe, "--nonsense"], capture_output=True, text=True)
    assert r.returncode == 2  # separate_dsd.py:319-321


def test_output_names_follow_each_script():
    from deepconvsep_amd.separation import output_paths
    assert [os.path.basename(p) for p in output_paths("dsd", "/a/mix.wav", "/o")] == \
        ["vocals.wav", "bass.wav", "drums.wav", "other.wav"]
    assert [os.path.basename(p) for p in output_paths("ikala", "/a/x.wav", "/o")] == ["x-voice.wav", "x-music.wav"]
    assert os.path.basename(output_paths("bach10", "/a/x.wav", "/o")[2]) == "x_saxphone.wav"  # sic, separate_bach10.py:236


def test_mono_mixdown_rules():
    from deepconvsep_amd.separation import to_mono
    st = np.array([[0.2, 0.4], [1.0, -1.0]])
    np.testing.assert_allclose(to_mono(st, "dsd"), [0.3, 0.0])
    np.testing.assert_allclose(to_mono(st, "ikala"), [0.6, 0.0])  # sum, not mean (separate_ikala.py:229)
    np.testing.assert_allclose(to_mono(st[:, 0], "dsd"), st[:, 0])
    with pytest.raises(IndexError):
        to_mono(st[:, 0], "ikala")


# ------------------------------------------------------------------ score-informed front-end, host logic
@pytest.mark.parametrize("name", ["score_n4096_hop512", "score_n1024_hop512", "score_n2048_hop256"])
def test_product_note_tables_match_the_reference(golden, name, tmp_path):
    """deepconvsep_amd.score (the shipped host code: vectorised, own structure) against the tables the reference's
    own expandMidi / getMidiNum produced (tests/golden/score_*.npz) -- integers, bit exact."""
    from deepconvsep_amd import score
    g = golden(name)
    insts = ["bassoon_b", "clarinet_b", "saxophone_b", "violin_b"]
    for ins, text in zip(insts, g["texts"]):
        (tmp_path / (ins + ".txt")).write_text(str(text))
    N, hop, nframes = int(g["frame"]), int(g["hop"]), int(g["nframes"])
    for i, ins in enumerate(insts):
        assert score.getMidiNum(ins, str(tmp_path), 0, 40.0) == int(g["nums"][i])
        t = score.expandMidi(ins, str(tmp_path), 0, 40.0, 50, 440, 20, 44100, hop, N, 0.2, 0.2, nframes, 0.5)
        assert t.dtype == np.float64 and np.array_equal(t, g["melody"][i, :t.shape[0]])
        # the script passes file names with their extension (separate_bach10.py:455), util.py appends it (:426)
        t2 = score.expandMidi(ins + ".txt", str(tmp_path), 0, 40.0, 50, 440, 20, 44100, hop, N, 0.2, 0.2, nframes, 0.5)
        assert np.array_equal(t, t2)
    mel = score.melody_table([i + ".txt" for i in insts], str(tmp_path), nframes, 44100, hop, N)
    assert np.array_equal(mel, g["melody"])


def test_product_score_helpers_against_the_oracle(tmp_path):
    from deepconvsep_amd import score
    from oracle import score_np
    for note in ("C4", "A4", "Bb3", "F#5", "Cx2", "Db1", "B0", "G9", "a#4", " E3"):
        assert score.str2midi(note) == score_np.str2midi(note)
    assert np.isnan(score.str2midi("?"))
    for midi in (24, 36, 45.0, 60, 72, 84, 96, 108, 120):
        for N, itv, nh in ((4096, 50, 20), (1024, 30, 20), (2048, 100, 12)):
            want = score_np.slicefft_ranges(midi, N, interval=itv, nharmonics=nh)
            got = score.harmonic_bins(midi, N, interval=itv, nharmonics=nh)
            assert got.tolist() == [list(r) for r in want]
    assert score.harmonic_bins(float("nan"), 4096).shape == (0, 2) and score.harmonic_bins(0, 4096).shape == (0, 2)
    # windows, tunings and time spans other than the script's, and a single-note score (reference returns None)
    for seed in range(12):
        p = score_np.synth_score(str(tmp_path / "v_b.txt"), 500 + seed, n_notes=26, total=11.0)
        for args in ((0, 40.0, 50, 440, 20, 44100, 512, 4096, 0.2, 0.2, 864, 0.5),
                     (2.0, 8.0, 30, 442, 20, 44100, 512, 4096, 0.1, 0.3, 500, 0.0),
                     (1.0, 6.5, 80, 415, 10, 44100, 256, 1024, 0.05, 0.0, 900, 1.0)):
            want = score_np.expandMidi(p, *args)
            got = score.expandMidi("v_b", str(tmp_path), *args)
            assert np.array_equal(got, want)
            assert score.getMidiNum("v_b", str(tmp_path), args[0], args[1]) == score_np.getMidiNum(p, args[0], args[1])
    (tmp_path / "one_b.txt").write_text("0.5,1.5,C4\n")
    assert score.expandMidi("one_b", str(tmp_path), 0, 40.0, 50, 440, 20, 44100, 512, 4096, 0.2, 0.2, 100, 0.5) is None
    assert score.getMidiNum("one_b", str(tmp_path), 0, 40.0) == 1


def test_length_groups_policy():
    """Separator.separate_many: clips sorted by length, groups of bounded size and bounded longest / shortest ratio."""
    from deepconvsep_amd.separation import length_groups
    sizes = [40000, 31000, 40000, 40000, 52000, 0, 100, 149, 151]
    groups = length_groups(sizes, max_group=3, max_ratio=1.5)
    assert sorted(i for g in groups for i in g) == list(range(len(sizes)))          # a partition
    assert groups[0] == [5]                                                            # the empty clip stays alone
    for g in groups:
        assert len(g) <= 3
        lo, hi = min(sizes[i] for i in g), max(sizes[i] for i in g)
        assert hi <= 1.5 * lo or lo == 0
    assert [6, 7] in groups and [8] in groups                                          # 151 > 1.5 * 100
    assert length_groups([5, 5, 5], max_group=1) == [[0], [1], [2]]
    assert length_groups([]) == []



def test_bf16_three_way_split_is_exact_and_six_products_are_fp32_class():
    """The arithmetic behind the opt-in bf16x3 kernel (csrc/dsd_bf16x3.hip), restated in NumPy: truncating a float32 to
    its top 16 bits three times gives x = hi + mid + lo EXACTLY, and the six kept term pairs reproduce a 50-long dot
    product to within a few 2^-24 of sum |a b| -- the error class of a float32 accumulation."""
    rs = np.random.RandomState(5)

    def split(x):
        parts, r = [], x.astype(np.float32)
        for _ in range(3):
            p = (r.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
            parts.append(p)
            r = (r - p).astype(np.float32)                      # exact: no rounding happens in these subtractions
        return parts, r

    x = np.concatenate([rs.randn(4000), 1e-3 * rs.randn(2000), 300.0 * rs.randn(2000), [0.0, 1.0, -2.5, 3e-20]]).astype(np.float32)
    (h, m, l), rest = split(x)
    assert not rest.any()                                          # nothing left after three 8-bit pieces
    assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), x)
    for p in (h, m, l):
        assert not (p.view(np.uint32) & np.uint32(0xffff)).any()   # every piece is a bf16 value

    a = rs.randn(64, 50).astype(np.float32) * 10.0
    b = (rs.randn(50, 33) * 0.05).astype(np.float32)
    (a0, a1, a2), _ = split(a)
    (b0, b1, b2), _ = split(b)
    f = np.float64
    kept = (a2.astype(f) @ b0 + a0.astype(f) @ b2 + a1.astype(f) @ b1) + (a1.astype(f) @ b0 + a0.astype(f) @ b1) + a0.astype(f) @ b0
    exact = a.astype(f) @ b.astype(f)
    scale = np.abs(a).astype(f) @ np.abs(b).astype(f)
    assert np.max(np.abs(kept - exact) / scale) < 3.0 * 2.0 ** -24   # the three dropped pairs
    f32 = (a @ b).astype(f)                                        # a float32 accumulation for comparison
    assert np.max(np.abs(kept - exact) / scale) < 4.0 * np.max(np.abs(f32 - exact) / scale) + 2.0 ** -24


def test_wavio_reads_and_writes_what_scipy_does(tmp_path):
    """deepconvsep_amd.wavio (the batch driver's int16 path): `read_pcm16` / `read_pcm16_into` return exactly the frames
    `scipy.io.wavfile.read` returns for 16-bit PCM files (mono, stereo, three channels, empty, odd chunk layouts) and decline
    everything else; `write_pcm16` produces the bytes `scipy.io.wavfile.write` produces."""
    import struct
    import scipy.io.wavfile
    from deepconvsep_amd import wavio
    rs = np.random.RandomState(0)
    a, b = str(tmp_path / "a.wav"), str(tmp_path / "b.wav")
    for ch in (1, 2, 3):
        for n in (0, 1, 5, 4411):
            x = rs.randint(-32768, 32768, (n, ch) if ch > 1 else (n,)).astype(np.int16)
            scipy.io.wavfile.write(a, 44100, x)
            wavio.write_pcm16(b, 44100, x)
            assert open(a, "rb").read() == open(b, "rb").read(), (ch, n)
            rate, y = wavio.read_pcm16(a)
            sr, want = scipy.io.wavfile.read(a)
            assert rate == sr and y.dtype == want.dtype and y.shape == want.shape and np.array_equal(y, want)
            buf = np.zeros(os.path.getsize(a), np.uint8)
            assert wavio.read_pcm16_into(a, buf) == (44100, n, ch)
            assert np.array_equal(buf[:2 * n * ch].view(np.int16), x.reshape(-1))
            if n:
                assert wavio.read_pcm16_into(a, buf[:2 * n * ch - 1]) is None          # does not fit: the caller falls back
    # a LIST chunk in front of the data and an odd-sized chunk with its pad byte (scipy reads these too)
    x = rs.randint(-32768, 32768, (100, 2)).astype(np.int16)
    fmt = struct.pack("<HHIIHH", 1, 2, 44100, 44100 * 4, 4, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"LIST" + struct.pack("<I", 5) + b"abcde\x00" + \
        b"data" + struct.pack("<I", x.nbytes) + x.tobytes()
    with open(a, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    sr, want = scipy.io.wavfile.read(a)
    rate, y = wavio.read_pcm16(a)
    assert rate == sr == 44100 and np.array_equal(y, want) and np.array_equal(y, x)
    # not 16-bit PCM: float32 samples, 8-bit samples, no RIFF header at all
    scipy.io.wavfile.write(a, 44100, rs.rand(100).astype(np.float32))
    assert wavio.read_pcm16(a) is None and wavio.read_pcm16_into(a, np.zeros(4096, np.uint8)) is None
    scipy.io.wavfile.write(a, 22050, rs.randint(0, 255, 100).astype(np.uint8))
    assert wavio.read_pcm16(a) is None
    with open(a, "wb") as fh:
        fh.write(b"RIFF\x00\x00\x00\x00WAVEjunk")
    assert wavio.read_pcm16(a) is None
    # a file that ends inside its fmt chunk: None (the caller falls back to scipy), not struct.error
    with open(a, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 100) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt[:9])
    assert wavio.read_pcm16(a) is None and wavio.read_pcm16_into(a, np.zeros(64, np.uint8)) is None
    # a short writev that ends INSIDE the 44-byte header: the rest of the header and the body still arrive
    real_writev = os.writev
    try:
        os.writev = lambda fd, bufs: os.write(fd, bytes(bufs[0])[:17])
        wavio.write_pcm16(b, 44100, x)
    finally:
        os.writev = real_writev
    scipy.io.wavfile.write(a, 44100, x)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_wav_batch_dropped_without_result_still_finishes(tmp_path):
    """A WavBatch that is dropped without result() (fire-and-forget write) waits for the native I/O threads in its finalizer:
    they read the sample arrays and write the status array the batch object keeps alive."""
    import gc
    import scipy.io.wavfile
    from deepconvsep_amd import wavio
    rs = np.random.RandomState(2)
    paths = [str(tmp_path / ("ff%02d.wav" % i)) for i in range(24)]
    want = []
    with wavio.WavPool(3) as pool:
        for rnd in range(4):
            arrays = [rs.randint(-32768, 32768, (50000, 2)).astype(np.int16) for _ in paths]
            want = [a.copy() for a in arrays]
            pool.write(paths, [44100] * len(paths), arrays)      # the batch object and `arrays` die here
            del arrays
            gc.collect()
    for p, x in zip(paths, want):
        sr, y = scipy.io.wavfile.read(p)
        assert sr == 44100 and np.array_equal(y, x), p


def test_wav_pool_of_libdcs_reads_and_writes_what_scipy_does(tmp_path):
    """wavio.WavPool = the I/O threads of libdcs.so (dcs_wav_read_pcm16_async / dcs_wav_write_pcm16_async, csrc/wavio.hip; host
    code, no device): a batch of writes produces the bytes `scipy.io.wavfile.write` produces (parent directories made on the way),
    a batch of reads delivers the frames `scipy.io.wavfile.read` returns into the caller's buffers -- same verdict per file as
    the Python reader `read_pcm16_into` (frames | None for anything that is not plain 16-bit PCM or does not fit | OSError)."""
    import struct
    import scipy.io.wavfile
    from deepconvsep_amd import wavio
    rs = np.random.RandomState(1)
    ref = str(tmp_path / "ref.wav")
    cases = [(n, ch) for ch in (1, 2, 3) for n in (0, 1, 5, 4411, 70001)]
    arrays = [rs.randint(-32768, 32768, (n, ch) if ch > 1 else (n,)).astype(np.int16) for n, ch in cases]
    rates = [44100 if i % 2 == 0 else 22050 for i in range(len(cases))]
    paths = [str(tmp_path / "deep" / ("d%d" % (i % 3)) / ("f%02d.wav" % i)) for i in range(len(cases))]
    with wavio.WavPool(5) as pool:
        batch = pool.write(paths, rates, arrays)
        assert batch.result() == [None] * len(cases) and batch.done() and batch.result() == [None] * len(cases)
        for p, x, sr in zip(paths, arrays, rates):
            scipy.io.wavfile.write(ref, sr, x)
            assert open(ref, "rb").read() == open(p, "rb").read(), p
        # odd chunk layouts, an extensible header, not-PCM files, a missing file, a buffer that is too small -- one batch
        x = rs.randint(-32768, 32768, (100, 2)).astype(np.int16)
        fmt = struct.pack("<HHIIHH", 1, 2, 44100, 44100 * 4, 4, 16)
        odd = str(tmp_path / "odd.wav")
        body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"LIST" + struct.pack("<I", 5) + b"abcde\x00" + \
            b"data" + struct.pack("<I", x.nbytes) + x.tobytes()
        with open(odd, "wb") as fh:
            fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
        ext = str(tmp_path / "ext.wav")
        fmt_ext = struct.pack("<HHIIHH", 0xFFFE, 2, 44100, 44100 * 4, 4, 16) + struct.pack("<HHI", 22, 16, 3) + \
            b"\x01\x00\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        body = b"WAVE" + b"fmt " + struct.pack("<I", 40) + fmt_ext + b"data" + struct.pack("<I", x.nbytes) + x.tobytes()
        with open(ext, "wb") as fh:
            fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
        trunc = str(tmp_path / "trunc.wav")              # the data chunk promises more than the file holds: whole frames of the rest
        with open(trunc, "wb") as fh:
            fh.write(open(odd, "rb").read()[:-7])
        flt, u8, junk = str(tmp_path / "flt.wav"), str(tmp_path / "u8.wav"), str(tmp_path / "junk.wav")
        scipy.io.wavfile.write(flt, 44100, rs.rand(100).astype(np.float32))
        scipy.io.wavfile.write(u8, 22050, rs.randint(0, 255, 100).astype(np.uint8))
        with open(junk, "wb") as fh:
            fh.write(b"RIFF\x00\x00\x00\x00WAVEjunk")
        files = paths + [odd, ext, trunc, flt, u8, junk, str(tmp_path / "missing.wav"), paths[4]]
        bufs = [np.zeros(os.path.getsize(f) if os.path.exists(f) else 16, np.uint8) for f in files]
        bufs[-1] = np.zeros(2 * 70001 - 1, np.uint8)                                   # one byte short
        got = pool.read_into(files, bufs).result()
        for i, (f, b) in enumerate(zip(files, bufs)):
            if i == len(files) - 2:
                assert isinstance(got[i], OSError) and got[i].errno == 2
                continue
            assert got[i] == wavio.read_pcm16_into(f, np.zeros_like(b)), f             # the Python reader's verdict
            if got[i] is not None:
                sr, want = scipy.io.wavfile.read(f) if f != trunc else (44100, x[:98])
                rate, n, ch = got[i]
                assert rate == sr and n == want.shape[0] and np.array_equal(b[:2 * n * ch].view(np.int16), want.reshape(-1)), f
        assert got[len(paths)] == (44100, 100, 2) and got[len(paths) + 1] == (44100, 100, 2) and got[len(paths) + 2] == (44100, 98, 2)
        assert got[len(paths) + 3] is None and got[len(paths) + 4] is None and got[len(paths) + 5] is None and got[-1] is None
        # a directory that cannot be made: the file fails alone
        blocker = str(tmp_path / "blocker")
        open(blocker, "w").close()
        res = pool.write([os.path.join(blocker, "x.wav"), str(tmp_path / "fine.wav")], [44100, 44100], [arrays[1], arrays[1]]).result()
        assert isinstance(res[0], OSError) and res[1] is None
        # many small batches in flight at once
        batches = [pool.write([str(tmp_path / ("m%d_%d.wav" % (k, j))) for j in range(7)], [8000] * 7, [arrays[2]] * 7) for k in range(20)]
        assert all(b.result() == [None] * 7 for b in batches)
    with pytest.raises(ValueError):
        wavio.WavPool(0)
