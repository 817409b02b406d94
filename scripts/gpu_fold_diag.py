"""Diagnostic (round 6, visit W): the folded conv2 + bottleneck map on the adversarial weight sets of oracle/cases.py.
One process per setting of DCS_FOLD_CONV2 (read once per process): PCM and separated spectra of the one-batch path (never folded)
and of the throughput path against the oracle, with the four-part mask criterion of oracle/maskcheck.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_latency as T
import deepconvsep_amd as dcs
from deepconvsep_amd.runtime import default_context
from deepconvsep_amd.synth import synth_audio
from oracle import cases, pipeline, stft_np, tiling_np, maskcheck

def main():
    for kind in ("sparse", "tiny", "glorot"):
        N = 2048; F = N // 2 + 1; TC = 30
        audio = synth_audio(T.samples_for_tiles(32) + 11, seed=77)
        audio[40000:52000] = 0.0
        mag0, _ = stft_np.compute_file(audio, phase=True, frameSize=N, hopSize=512, window=np.hanning)
        tiles, n = tiling_np.generate_overlapadd(0.3 * mag0.astype(np.float32), F, TC, 25, 32, tiler=tiling_np.SCRIPT, fill=0.0)
        x = tiles.reshape((-1,) + tiles.shape[2:])[:min(n, 8)].astype(np.float32)
        params = cases.calibrate("dsd", TC, F, 52, kind, x)
        want, mm, mag, ph = pipeline.separate("dsd", params, audio, 0.3, TC, 25, 32, N, 512, np.hanning, return_spectra=True)
        sep = dcs.Separator("dsd", params, 0.3, TC, 25, 32, F, N, 512, np.hanning)
        ctx = default_context(); a = ctx.to_device(audio, np.float32)
        out = {}
        for name, stages in (("lat", 255), ("thr", 0)):
            sep.net.set_latency_stages(stages)
            s_d, _, _ = sep.net.separate_spectra(sep.plan, a, 25, sep.tiler, 0.3)
            out[name] = (s_d.cpu().numpy(), sep.separate(audio))
        for name in ("lat", "thr"):
            s, pcm = out[name]
            err = np.abs(s - mm)
            bad = err.max(axis=0) > 1e-4
            i = np.unravel_index(np.argmax(np.abs(pcm - want)), pcm.shape)
            print("fold=%s %-6s %-3s: pcm max|err| vs oracle %.3e (source %d sample %d), spectra max %.3e, bins outside 1e-4: %d of %d, max |oracle| at those bins %.3e"
                  % (os.environ.get("DCS_FOLD_CONV2", "1"), kind, name, np.abs(pcm - want).max(), i[0], i[1], err.max(), int(bad.sum()), err[0].size,
                     float(np.max(np.abs(mm[:, bad]))) if bad.any() else 0.0))
        print("fold=%s %-6s lat vs thr: pcm %.3e spectra %.3e" % (os.environ.get("DCS_FOLD_CONV2", "1"), kind, np.abs(out["lat"][1] - out["thr"][1]).max(), np.abs(out["lat"][0] - out["thr"][0]).max()))

main()
