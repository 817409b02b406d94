"""Micro-experiment: forward STFT kernel alone, 20.5k frames (N=2048), timed with torch events."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepconvsep_amd as dcs
from deepconvsep_amd.runtime import StftPlan, default_context
from deepconvsep_amd.synth import synth_audio
ctx = default_context()
N, hop = 2048, 512
plan = StftPlan(ctx, N, hop, np.hanning(N))
L = 20504 * hop
a = ctx.to_device(synth_audio(L, seed=1), np.float32)
for ph in (True,):
    for _ in range(3): plan.forward(a, phase=ph)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): m, p = plan.forward(a, phase=ph)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    T = m.shape[0]
    print("dbg=%s fpw=%s phase=%s: %.4f ms  (%.2f TB/s of %d B/frame)" % (os.environ.get("DCS_STFT_DBG", "0"), os.environ.get("DCS_STFT_FPW", "-"), ph, ms, T * (2048 + 2 * 1025 * 4) / ms / 1e9, 2048 + 8200))
