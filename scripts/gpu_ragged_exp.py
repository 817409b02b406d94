#!/usr/bin/env python3
"""Round 4: what a ragged launch group costs (dcs_separate_ragged, DSD graph, frameSize 2048).  Clips whose lengths are 1 : 3
(8 of 32 tiles + 8 of 96 tiles) in ONE call, against equal-length batches holding the same frames: 16 x 64 tiles in one
dcs_separate_batch call (the sum-of-frames cost) and 16 x 96 tiles (the n x longest cost the uniform-pitch layout pays).
Run once per layout: DCS_RAGGED_COMPACT=1 (default) / 0.     python scripts/gpu_ragged_exp.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import deepconvsep_amd as dcs  # noqa: E402
from deepconvsep_amd.synth import synth_audio, synth_params  # noqa: E402

N, F, HOP = 2048, 1025, 512


def samples(n_tiles):
    return (30 + 1 + (n_tiles - 1) * 5 - 2) * HOP


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    sep = dcs.Separator("dsd", synth_params("dsd", 30, F, seed=2), 0.3, 30, 25, 32, F, N, HOP, np.hanning)
    sep.net.set_latency_stages(0)
    ctx = sep.ctx
    lens = [samples(32)] * 8 + [samples(96)] * 8
    Lmax = max(lens)
    stack = np.zeros((16, Lmax), dtype=np.float32)
    for i, L in enumerate(lens):
        stack[i, :L] = synth_audio(L, seed=70 + i)
    dev = ctx.to_device(stack, np.float32)
    out = torch.zeros((16, 4, Lmax), dtype=torch.float32, device=dev.device)
    t_ragged = timed(lambda: sep.net.separate_ragged(sep.plan, dev, lens, 25, sep.tiler, 0.3, out=out))
    same = {}
    for nt in (64, 96, 32):
        a = ctx.to_device(np.stack([synth_audio(samples(nt), seed=90 + i) for i in range(16 if nt != 32 else 8)]), np.float32)
        o = torch.empty((a.shape[0], 4, a.shape[1]), dtype=torch.float32, device=a.device)
        same[nt] = timed(lambda: sep.net.separate_batch(sep.plan, a, 25, sep.tiler, 0.3, out=o))
    a96 = ctx.to_device(np.stack([synth_audio(samples(96), seed=120 + i) for i in range(8)]), np.float32)
    o96 = torch.empty((8, 4, a96.shape[1]), dtype=torch.float32, device=a96.device)
    t_8x96 = timed(lambda: sep.net.separate_batch(sep.plan, a96, 25, sep.tiler, 0.3, out=o96))
    # parity of the ragged call against clip-by-clip separation
    got = ctx.to_host(sep.net.separate_ragged(sep.plan, dev, lens, 25, sep.tiler, 0.3))
    worst = 0.0
    for i in (0, 7, 8, 15):
        alone = sep.separate(stack[i, :lens[i]])
        worst = max(worst, float(np.max(np.abs(got[i, :, :lens[i]] - alone))))
    print("layout %s: ragged 8 x 32 + 8 x 96 tiles (1024 tiles) %.4f ms | equal-length batches: 16 x 64 (same sum of frames) %.4f ms, "
          "16 x 96 (n x longest) %.4f ms, 8 x 32 + 8 x 96 as two calls %.4f ms | ragged / sum-of-frames %.3f | max |ragged - alone| %.2e"
          % ("compact" if os.environ.get("DCS_RAGGED_COMPACT", "1") != "0" else "uniform pitch", t_ragged, same[64], same[96],
             same[32] + t_8x96, t_ragged / same[64], worst))


if __name__ == "__main__":
    main()
