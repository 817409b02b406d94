#!/usr/bin/env python3
"""Generate tests/golden/*.npz FROM THE REFERENCE'S OWN CODE.

Run in the build container (``/root/reference`` must exist):

    python tests/golden/make_golden.py

The reference's pure-NumPy helpers are executed by line range through
``oracle.ref_exec`` (nothing is copied) on seeded inputs; inputs and outputs
are written as small compressed fixtures.  The GPU box has no reference tree,
so the parity tests there read these files.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_exec  # noqa: E402


def seeded_audio(n, seed):
    """5 sinusoids + 0.1 uniform noise + a digital-silence gap, in [-1, 1]."""
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 44100.0
    x = np.zeros(n)
    for f0, a in zip(rs.uniform(80.0, 6000.0, 5), rs.uniform(0.05, 0.2, 5)):
        x += a * np.sin(2 * np.pi * f0 * t + rs.uniform(0, 2 * np.pi))
    x += 0.1 * rs.uniform(-1, 1, n)
    g0 = n // 3
    x[g0:g0 + n // 8] = 0.0
    # int16 quantise the way a wav read would deliver it (separate_dsd.py:275-282)
    q = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
    return q.astype('float') / 32767.0


def blackmanharris(n):
    from scipy.signal.windows import blackmanharris as bh
    return bh(n)


def main():
    if not ref_exec.available():
        raise SystemExit("reference tree not found; fixtures can only be generated in the build container")
    dsd = ref_exec.script_dsd()
    ika = ref_exec.script_ikala()
    lib = ref_exec.lib_stft()
    libt = ref_exec.lib_tiling()

    # ---- STFT / iSTFT through the script's compute_file / compute_inverse -------------
    cases = [
        ("stft_n1024_hann", 1024, 512, np.hanning, 5003, 11),
        ("stft_n2048_hann", 2048, 512, np.hanning, 6000, 12),
        ("stft_n4096_bh", 4096, 512, blackmanharris, 5121, 13),
        ("stft_n1024_hop256_hann", 1024, 256, np.hanning, 3000, 14),   # transformFFT default hop
        ("stft_n512_hop200_hann", 512, 200, np.hanning, 2111, 15),     # hop does not divide N
    ]
    for name, N, hop, win, L, seed in cases:
        audio = seeded_audio(L, seed)
        mag, ph = dsd.compute_file(audio, phase=True, frameSize=N, hopSize=hop, window=win)
        back = dsd.compute_inverse(mag, ph, frameSize=N, hopSize=hop, window=win)
        # library functions on the same input must agree with the script copies
        X = lib.stft_norm(audio, window=win(N), hopsize=float(hop), nfft=float(N), fs=44100.0)
        assert np.array_equal(np.abs(X) / np.sqrt(N), mag)
        back_lib = lib.istft_norm((mag * np.sqrt(N)) * np.exp(1j * ph), window=win(N),
                                  analysisWindow=win(N), hopsize=float(hop), nfft=float(N))
        assert np.array_equal(back_lib, back)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), audio=audio, frame=N, hop=hop,
                            window=win(N), mag=mag, phase=ph, inverse=back)
        print(name, mag.shape, back.shape)

    # ---- tilers ---------------------------------------------------------------------------
    rs = np.random.RandomState(21)
    F = 9
    for name, T, tc, ov, B in [("tile_t83_tc30_ov25", 83, 30, 25, 4),
                               ("tile_t61_tc30_ov20", 61, 30, 20, 4),
                               ("tile_t30_tc30_ov25", 30, 30, 25, 4),     # script tiler: zero tiles
                               ("tile_t131_tc30_ov25", 131, 30, 25, 32)]:
        mag = rs.uniform(0, 1, (T, F)).astype(np.float32)
        # script tiler leaves unused slots uninitialised (np.empty): record only the count
        # and the filled tiles
        fb_s, n_s = dsd.generate_overlapadd(mag, input_size=F, time_context=tc, overlap=ov, batch_size=B)
        fb_s = fb_s.reshape((-1,) + fb_s.shape[2:])[:n_s] if n_s else np.zeros((0, 1, tc, F))
        fb_l, n_l = libt.generate_overlapadd(mag, input_size=F, time_context=tc, overlap=ov, batch_size=B)
        mag3 = rs.uniform(0, 1, (3, T, F)).astype(np.float32)
        fb_l3, n_l3 = libt.generate_overlapadd(mag3, input_size=F, time_context=tc, overlap=ov, batch_size=B)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), mag=mag, mag3=mag3, tc=tc, ov=ov, B=B,
                            script_tiles=fb_s, script_n=n_s, library_batches=fb_l, library_n=n_l,
                            library3_batches=fb_l3, library3_n=n_l3)
        print(name, "script n =", n_s, "library n =", n_l)

    # ---- cross-fade overlap-add -------------------------------------------------------------
    for name, n, tc, ov, B, S in [("ola_n11_tc30_ov25_s4", 11, 30, 25, 4, 4),
                                  ("ola_n7_tc30_ov20_s2", 7, 30, 20, 4, 2),
                                  ("ola_n1_tc30_ov25_s4", 1, 30, 25, 4, 4),
                                  ("ola_n37_tc30_ov25_s4", 37, 30, 25, 32, 4)]:
        nb = int(np.ceil(float(n) / B))
        out = rs.uniform(0, 1, (nb, S, B, 1, tc, F))
        sep = dsd.overlapadd_multi(out, None, n, overlap=ov)
        sep_lib = libt.overlapadd_multi(out, None, n, overlap=ov)
        assert np.array_equal(sep, sep_lib)
        d = dict(out=out, n=n, tc=tc, ov=ov, B=B, sep=sep)
        if S == 2:
            s1, s2 = ika.overlapadd(out, None, n, overlap=ov)
            assert np.array_equal(s1, sep[0]) and np.array_equal(s2, sep[1])
            d["sep1"], d["sep2"] = s1, s2
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, sep.shape)

    # ---- score-informed front-end: util.expandMidi / getMidiNum / script filterSpec ---------------
    import tempfile
    from oracle import score_np
    sc = ref_exec.score()
    for name, N, hop, seconds, seed in [("score_n4096_hop512", 4096, 512, 10.0, 301),
                                        ("score_n1024_hop512", 1024, 512, 4.0, 302),
                                        ("score_n2048_hop256", 2048, 256, 5.0, 303)]:
        L = int(seconds * 44100)
        nframes = int(np.ceil(L / np.double(hop))) + 2
        F = N // 2 + 1
        with tempfile.TemporaryDirectory() as d:
            texts, tables, nums = [], [], []
            insts = ["bassoon_b", "clarinet_b", "saxophone_b", "violin_b"]
            for i, ins in enumerate(insts):
                pth = score_np.synth_score(os.path.join(d, ins + ".txt"), seed * 10 + i, n_notes=40,
                                           total=seconds + 1.5, lo=36 + 6 * i, hi=60 + 8 * i)
                texts.append(open(pth).read())
                tables.append(sc.expandMidi(ins, d, 0, 40.0, 50, 440, 20, 44100, hop, N, 0.2, 0.2, nframes, 0.5))
                nums.append(sc.getMidiNum(ins, d, 0, 40.0))
            P = max(max(nums), 1)
            melody = np.zeros((4, P, 43))
            for i, t in enumerate(tables):
                melody[i, :t.shape[0]] = t
            mag = (0.3 * np.abs(np.random.RandomState(seed).randn(nframes, F)).astype(np.float32)).astype(np.float32)
            mask = sc.filterSpec(mag, melody, 4, 0, nframes)
            # a window that does not start at 0 and an instrument without notes (all-ones mask)
            melody2 = melody.copy()
            melody2[2] = 0
            mask_win = sc.filterSpec(mag[40:140], melody2, 4, 40, 140)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), frame=N, hop=hop, n_samples=L, nframes=nframes,
                            texts=np.array(texts), melody=melody, nums=np.array(nums), mag_seed=seed,
                            mask_rowsum=mask.astype(np.float64).sum(axis=1), mask_colsum=mask.astype(np.float64).sum(axis=0),
                            mask_ones=np.int64((mask == 1).sum()), mask_win=mask_win)
        print(name, melody.shape, nums, int((mask == 1).sum()))


def score_timbre():
    """The timbre-model branch of the reference's filterSpec (separate_bach10.py:173-175,189-191), executed: the model is a
    seeded float32 array [instruments, 128 pitches, 20 harmonics] pickled to a temporary file (the reference's NumPy added a
    float64 template weight to the float32 mask as float32; a float32 model makes that the arithmetic under any NumPy)."""
    import pickle
    import tempfile
    from oracle import score_np
    sc = ref_exec.score()
    N, hop, seconds, seed = 1024, 512, 4.0, 302
    L = int(seconds * 44100)
    nframes = int(np.ceil(L / np.double(hop))) + 2
    F = N // 2 + 1
    harmonics = np.random.RandomState(77).uniform(0.05, 1.0, (4, 128, 20)).astype(np.float32)
    harmonics[1, :, 3] = 0.0                                          # a harmonic the template does not have
    with tempfile.TemporaryDirectory() as d:
        texts, tables, nums = [], [], []
        for i, ins in enumerate(["bassoon_b", "clarinet_b", "saxophone_b", "violin_b"]):
            pth = score_np.synth_score(os.path.join(d, ins + ".txt"), seed * 10 + i, n_notes=40, total=seconds + 1.5,
                                       lo=36 + 6 * i, hi=60 + 8 * i)
            texts.append(open(pth).read())
            tables.append(sc.expandMidi(ins, d, 0, 40.0, 50, 440, 20, 44100, hop, N, 0.2, 0.2, nframes, 0.5))
            nums.append(sc.getMidiNum(ins, d, 0, 40.0))
        melody = np.zeros((4, max(max(nums), 1), 43))
        for i, t in enumerate(tables):
            melody[i, :t.shape[0]] = t
        mag = (0.3 * np.abs(np.random.RandomState(seed).randn(nframes, F)).astype(np.float32)).astype(np.float32)
        model = os.path.join(d, "timbre.pkl")
        with open(model, "wb") as fh:
            pickle.dump(harmonics, fh, protocol=2)
        mask = sc.filterSpec(mag, melody, 4, 0, nframes, timbre_model_path=model)
        melody2 = melody.copy()
        melody2[2] = 0
        mask_win = sc.filterSpec(mag[40:140], melody2, 4, 40, 140, timbre_model_path=model)
    assert mask.dtype == np.float32
    np.savez_compressed(os.path.join(HERE, "score_timbre_n1024_hop512.npz"), frame=N, hop=hop, n_samples=L, nframes=nframes,
                        texts=np.array(texts), melody=melody, nums=np.array(nums), mag_seed=seed, harmonics=harmonics,
                        mask=mask, mask_win=mask_win)
    print("score_timbre_n1024_hop512", mask.shape, "distinct values", len(np.unique(mask)))


def score_sum():
    """The TRAINERS' harmonic masks: ``LargeDatasetMask2.filterSpec`` (dataset.py:839-879, every bin divided by the sum over
    the instruments), the class bach10_scoreinformed/trainCNNrwc.py:657 instantiates with ``tensortype=floatX`` -- its method
    body executed as it stands on the note tables and magnitudes of the committed score fixtures (binary branch), and on the
    timbre fixture's templates (timbre branch)."""
    for base in ("score_n1024_hop512", "score_n4096_hop512"):
        g = np.load(os.path.join(HERE, base + ".npz"))
        nframes, N, seed = int(g["nframes"]), int(g["frame"]), int(g["mag_seed"])
        F = N // 2 + 1
        mag = (0.3 * np.abs(np.random.RandomState(seed).randn(nframes, F)).astype(np.float32)).astype(np.float32)
        f = ref_exec.dataset_filterspec_sum()
        mask = f(mag, g["melody"], 0, nframes)
        melody2 = g["melody"].copy()
        melody2[2] = 0                                               # an instrument without notes
        mask_win = ref_exec.dataset_filterspec_sum()(mag[40:140], melody2, 40, 140)
        assert mask.dtype == np.float32 and mask.shape == (nframes, 4 * F)
        d = dict(base=base, mask=mask, mask_win=mask_win, values=np.unique(mask))
        if base == "score_n1024_hop512":
            t = np.load(os.path.join(HERE, "score_timbre_n1024_hop512.npz"))
            d["mask_timbre"] = ref_exec.dataset_filterspec_sum(harmonics=t["harmonics"])(mag, t["melody"], 0, nframes)
        np.savez_compressed(os.path.join(HERE, base.replace("score_", "score_sum_") + ".npz"), **d)
        print(base, "sum-normalised", mask.shape, "values", np.unique(mask))


def networks(only=None):
    """Network fixtures: the reference's OWN ``build_ca`` source and mask expressions executed on the NumPy Lasagne
    stand-in (oracle/lasagne_np.py) -- graph wiring, filter sizes, parameter order and mask arithmetic come from the
    reference's code; the layer semantics are restated Lasagne / Theano behaviour (PARITY of those stays unpinned)."""
    from oracle import cases
    for name, arch, F, B, seed, kind in cases.NET_CASES:
        if only and name not in only:
            continue
        x = cases.make_input(arch, B, 30, F, seed + 500)
        params = cases.calibrate(arch, 30, F, seed, kind, x)
        shapes = ref_exec.network_param_shapes(arch, B, x.shape[1], 30, F)
        assert [tuple(s_) for s_ in shapes] == [tuple(p_.shape) for p_ in params], name
        p = ref_exec.network_output(arch, params, x)
        d = dict(arch=arch, F=F, B=B, seed=seed, kind=kind, x=x, out_bias=params[-1], p=p,
                 n_params=len(params), zero_fraction=float((p == 0).mean()))
        if arch == "ikala":
            d["p_tie_first"] = ref_exec.network_output(arch, params, x, tie_mode="first")
        if arch == "ikala_nopool":
            # the masks of the separate script (separate_ikala.py:211-216, eps 1e-18): the drop-in surface is that script fed
            # a .pkl the trainer wrote; the trainer's own block uses its training epsilon 1e-8 (ikala/trainCNN.py:155)
            d["masked"] = np.stack(ref_exec.mask_sources("ikala", p, x))
        elif arch == "bach10_si1":
            # a trainer's graph (trainCNNrwc_samp.py): its own mask expressions multiply the SUM of the input channels
            # (:300-305); the separate script fed this .pkl multiplies channel 0 (separate_bach10.py:485)
            d["masked_sum"] = np.stack(ref_exec.mask_sources("bach10_si1", p, x))
            d["masked"] = np.stack(ref_exec.mask_sources("bach10_si", p, x))
        elif arch != "dsd_ild":
            d["masked"] = np.stack(ref_exec.mask_sources(arch, p, x))
        if arch == "bach10_si":
            d["masked_sum"] = np.stack(ref_exec.mask_sources("bach10_si_trainer", p, x))   # trainCNNrwc.py:253-263
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, p.shape, "zeros %.3f" % d["zero_fraction"], "max %.3f" % p.max())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "networks":
        networks(only=sys.argv[2:] or None)
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "score_sum":
        score_sum()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "score_timbre":
        score_timbre()
        raise SystemExit(0)

    main()
