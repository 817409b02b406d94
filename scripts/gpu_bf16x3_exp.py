"""Experiment: the opt-in bf16x3 final kernel (DCS_FINAL_BF16X3=1, DCS_FINAL_CBW=2) against the oracle and the f32 kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import deepconvsep_amd as dcs
from deepconvsep_amd.synth import synth_audio, synth_params
from oracle import pipeline

N = 2048; F = N // 2 + 1
params = synth_params("dsd", 30, F, seed=2)
sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, 512, np.hanning)
audio = synth_audio(3 * 44100, seed=77); audio[40000:52000] = 0.0
got = sep.separate(audio)
if len(sys.argv) > 1:
    np.save(sys.argv[1], got)
want = pipeline.separate("dsd", params, audio, 0.3, 30, 25, 32, N, 512, np.hanning)
print("env BF16X3=%s: max |pcm - oracle| = %.3e (finite %s)" % (os.environ.get("DCS_FINAL_BF16X3"), np.max(np.abs(got - want)), np.isfinite(got).all()))
# saturating timing
L = 4096 * 5 * 512
a = sep.ctx.to_device(synth_audio(L, seed=7), np.float32)
out = torch.empty((4, L), dtype=torch.float32, device=a.device)
for _ in range(2):
    sep.net.separate(sep.plan, a, 25, sep.tiler, 0.3, out=out)
sep.ctx.timing("all"); sep.ctx.timing_reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    sep.net.separate(sep.plan, a, 25, sep.tiler, 0.3, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
from deepconvsep_amd import _lib
ks = {}
for tag in _lib.TAGS:
    ms, cnt = sep.ctx.timing_query(tag)
    if cnt: ks[tag] = round(ms, 4)
print("4096 tiles: %.3f ms per clip" % (dt * 1e3), ks)
