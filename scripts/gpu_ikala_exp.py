"""iKala graph (frameSize 2048, 10 s): per-kernel times of the whole path for the current library / env switches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import deepconvsep_amd as dcs
from deepconvsep_amd import _lib
from deepconvsep_amd.synth import synth_audio, synth_params
label = sys.argv[1] if len(sys.argv) > 1 else "default"
N = int(os.environ.get("IKALA_N", "2048")); F = N // 2 + 1
params = synth_params("ikala", 30, F, seed=1)
sep = dcs.Separator("ikala", params, 0.3, 30, 20, 32, F, N, 512, np.hanning)
if os.environ.get("IKALA_F16"):
    sep.net.set_conv_precision('f16')
st = synth_audio(441000, seed=0, channels=2)
a = sep.ctx.to_device(st[:, 0] + st[:, 1], np.float32)
out = torch.empty((2, a.numel()), dtype=torch.float32, device=a.device)
run = lambda: sep.net.separate(sep.plan, a, 20, sep.tiler, 0.3, out=out)
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
sep.ctx.timing("all"); sep.ctx.timing_reset()
for _ in range(3): run()
ks = {}
for tag in _lib.TAGS:
    ms, cnt = sep.ctx.timing_query(tag)
    if cnt: ks[tag] = round(ms * cnt / 3, 4)
print("IKALA %-20s N=%d %.3f ms  conv2 %.4f deconv2 %.4f  | %s" % (label, N, dt * 1e3, ks.get("conv2", 0), ks.get("deconv2", 0), ks))
