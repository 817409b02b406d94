"""Per-kernel times of the fused DSD path for one library build (DCS_LIB / env switches decide which):
4096-tile clip (saturating) and a 16 x 32-tile launch group (the headline's launch shape), plus parity of a 3 s clip."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import deepconvsep_amd as dcs
from deepconvsep_amd import _lib
from deepconvsep_amd.synth import synth_audio, synth_params

label = sys.argv[1] if len(sys.argv) > 1 else "default"
check = os.environ.get("DCS_EXP_CHECK", "1") == "1"
N = 2048; F = N // 2 + 1
params = synth_params("dsd", 30, F, seed=2)
sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, 512, np.hanning)
err = float("nan")
if check:
    from oracle import pipeline
    audio = synth_audio(3 * 44100, seed=77); audio[40000:52000] = 0.0
    got = sep.separate(audio)
    want = pipeline.separate("dsd", params, audio, 0.3, 30, 25, 32, N, 512, np.hanning)
    err = float(np.max(np.abs(got - want))) if np.isfinite(got).all() else float("inf")

def times(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    sep.ctx.timing("all"); sep.ctx.timing_reset()
    for _ in range(reps):
        fn()
    ks = {}
    for tag in _lib.TAGS:
        ms, cnt = sep.ctx.timing_query(tag)
        if cnt: ks[tag] = round(ms * cnt / reps, 4)
    sep.ctx.timing(None); sep.ctx.timing_reset()
    return dt * 1e3, ks

L = (31 + 4095 * 5 - 2) * 512
a = sep.ctx.to_device(synth_audio(L, seed=7), np.float32)
out = torch.empty((4, L), dtype=torch.float32, device=a.device)
t_sat, k_sat = times(lambda: sep.net.separate(sep.plan, a, 25, sep.tiler, 0.3, out=out), 5)
del a, out
Lb = (31 + 31 * 5 - 2) * 512
ab = sep.ctx.to_device(np.stack([synth_audio(Lb, seed=20 + i) for i in range(16)]), np.float32)
t_grp, k_grp = times(lambda: sep.net.separate_batch(sep.plan, ab, 25, sep.tiler, 0.3), 20)
print("EXP %-22s err %.2e | 4096 tiles %.4f ms final %.4f deconv2 %.4f istft %.4f stft %.4f | 16x32 tiles %.4f ms final %.4f deconv2 %.4f"
      % (label, err, t_sat, k_sat.get("final", 0), k_sat.get("deconv2", 0), k_sat.get("istft", 0), k_sat.get("stft", 0),
         t_grp, k_grp.get("final", 0), k_grp.get("deconv2", 0)))
print("   sat", k_sat); print("   grp", k_grp)
