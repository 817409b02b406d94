"""Timing of the other BASELINE configs: iKala 2-source and Bach10 4-instrument on 10 s of audio
(configs[0], configs[3] shapes; fp32), plus the CPU oracle on the same input for configs[0]."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepconvsep_amd as dcs
from deepconvsep_amd.runtime import default_context
from deepconvsep_amd.synth import synth_audio, synth_params
from deepconvsep_amd import _lib
ctx = default_context()
for arch, N, F, ov, win in (("ikala", 1024, 513, 20, np.hanning), ("bach10", 4096, 2049, 25, dcs.blackmanharris), ("dsd", 1024, 513, 25, np.hanning)):
    params = synth_params(arch, 30, F, seed=3)
    audio = synth_audio(441000, seed=0)
    sep = dcs.Separator(arch, params, 0.3, 30, ov, 32, F, N, 512, win)
    if os.environ.get("DCS_ARCH_F16") and arch != "dsd":
        sep.net.set_conv_precision('f16')
    a = ctx.to_device(audio, np.float32)
    out = sep.net.separate(sep.plan, a, ov, sep.tiler, 0.3)
    torch.cuda.synchronize()
    ctx.timing("all"); ctx.timing_reset()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K): sep.net.separate(sep.plan, a, ov, sep.tiler, 0.3, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    ks = {t: round(ctx.timing_query(t)[0] * ctx.timing_query(t)[1] / K, 4) for t in _lib.TAGS if ctx.timing_query(t)[1]}
    ctx.timing(None)
    T = sep.net.last_frames; n = sep.net.last_tiles
    fl = dcs.ARCHS[arch].flops_per_tile(30, F) * n
    print("%s: 10 s, %d frames, %d tiles: %.3f ms -> %.0f frames/s (%.0fx real time), %.1f TFLOP/s algorithmic; per-stage ms/step %s" % (arch, T, n, dt * 1e3, T / dt, 10.0 / dt, fl / dt / 1e12, ks))
    if arch == "ikala" and os.environ.get("DCS_CPU", "1") == "1":
        from oracle import pipeline
        torch.set_num_threads(16)
        t0 = time.perf_counter(); pipeline.separate(arch, params, audio, 0.3, 30, ov, 32, N, 512, win); c = time.perf_counter() - t0
        print("ikala CPU oracle (16 threads): %.2f s -> %.0f frames/s (%.1fx real time)" % (c, T / c, 10.0 / c))
