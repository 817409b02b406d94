"""Host-issue time vs total time per 32-tile step, eager vs hipGraph, 1/4/8 streams."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepconvsep_amd as dcs
from deepconvsep_amd.runtime import Context
from deepconvsep_amd.synth import synth_audio, synth_params
from deepconvsep_amd.arch import TILER_SCRIPT
N, F, L = 2048, 1025, 184 * 512
params = synth_params("dsd", 30, F, seed=2)
class Lane:
    def __init__(s, i):
        s.stream = torch.cuda.Stream()
        with torch.cuda.stream(s.stream):
            s.ctx = Context(); s.sep = dcs.Separator("dsd", params, 0.3, 30, 25, 32, F, N, 512, np.hanning, ctx=s.ctx)
            s.a = s.ctx.to_device(synth_audio(L, seed=i), np.float32); s.o = torch.empty((4, L), dtype=torch.float32, device=s.a.device)
    def step(s):
        with torch.cuda.stream(s.stream):
            s.sep.net.separate(s.sep.plan, s.a, 25, TILER_SCRIPT, 0.3, out=s.o)
for ns in (1, 4, 8):
    lanes = [Lane(i) for i in range(ns)]
    for i in range(40): lanes[i % ns].step()
    torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter()
    for i in range(K): lanes[i % ns].step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("graph=%s streams=%d: issue %.1f us/step, total %.1f us/step" % (os.environ.get("DCS_GRAPH", "1"), ns, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
