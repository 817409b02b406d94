"""Does issuing from several host threads raise the 32-tile throughput?  (ctypes drops the GIL in the call)"""
import os, sys, time, threading, ctypes
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
        net, plan = s.sep.net, s.sep.plan
        s.call = (s.ctx._lib.dcs_separate, (net._h, plan._h, ctypes.c_void_p(s.a.data_ptr()), L, 25, TILER_SCRIPT, ctypes.c_float(0.3), 0, 0, ctypes.c_void_p(s.o.data_ptr()), None, None))
    def step(s):
        fn, a = s.call; fn(*a)
for nthreads, ns in ((1, 8), (2, 8), (4, 8), (2, 16), (4, 16)):
    lanes = [Lane(i) for i in range(ns)]
    for i in range(5 * ns): lanes[i % ns].step()
    torch.cuda.synchronize()
    K = 1600
    def work(tix):
        mine = lanes[tix::nthreads]
        for i in range(K // nthreads): mine[i % len(mine)].step()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    [t.start() for t in th]; [t.join() for t in th]
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("threads=%d streams=%d: issue %.1f us/step, total %.1f us/step" % (nthreads, ns, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
