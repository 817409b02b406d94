#!/usr/bin/env python3
"""Is one batch per call (32 tiles, one stream) bound by the host's launch rate or by the GPU?  K steps are enqueued
without a synchronisation; the time until the last enqueue returns is the host's, the wait behind it the GPU's backlog."""
import ctypes
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import deepconvsep_amd as dcs  # noqa: E402
from deepconvsep_amd import _lib  # noqa: E402
from deepconvsep_amd.arch import TILER_SCRIPT  # noqa: E402
from deepconvsep_amd.runtime import Context  # noqa: E402
from deepconvsep_amd.synth import synth_audio, synth_params  # noqa: E402

N, TILES = 2048, 32
F, HOP, TC, OV = N // 2 + 1, 512, 30, 25
L = (TC + 1 + (TILES - 1) * (TC - OV) - 2) * HOP


def main():
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    ctx = Context(stream=stream)
    with torch.cuda.stream(stream):
        sep = dcs.Separator("dsd", synth_params("dsd", TC, F, seed=2), 0.3, TC, OV, 32, F, N, HOP, np.hanning, ctx=ctx)
        audio = ctx.to_device(synth_audio(L, seed=100), np.float32)
        pcm = torch.empty((4, L), dtype=torch.float32, device=audio.device)
    stream.synchronize()
    net, plan = sep.net, sep.plan
    fn = ctx._lib.dcs_separate_batch
    args = (net._h, plan._h, ctypes.c_void_p(audio.data_ptr()), L, 1, L, OV, TILER_SCRIPT, ctypes.c_float(0.3),
            net.arch.eps_mode, 0, ctypes.c_void_p(pcm.data_ptr()), None, None)
    for k in (50, 400, 2000):
        for _ in range(8):
            fn(*args)
        torch.cuda.synchronize()
        host, total = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(k):
                rc = fn(*args)
                if rc:
                    _lib.check(rc)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) / k * 1e6)
            total.append((t2 - t0) / k * 1e6)
        print("K = %4d steps per round: host enqueue %.2f us per step, until the GPU is idle %.2f us per step (backlog at the last enqueue %.1f us)"
              % (k, statistics.median(host), statistics.median(total), (statistics.median(total) - statistics.median(host)) * k))


if __name__ == "__main__":
    main()
