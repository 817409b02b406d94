#!/usr/bin/env python3
"""One batch of 32 tiles per call on ONE stream (BASELINE configs[1] as the reference runs it): microseconds per step for
the throughput kernels (stages 0), the one-batch kernels (255) and every mix that differs from 255 / 0 by one stage, then
the per-kernel HIP-event times of both families.  DCS_LAT_EXP_N=1024 for the as-shipped frame size; DCS_LAT_EXP_TILES."""
import ctypes
import json
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

N = int(os.environ.get("DCS_LAT_EXP_N", "2048"))
TILES = int(os.environ.get("DCS_LAT_EXP_TILES", "32"))
F, HOP, TC, OV = N // 2 + 1, 512, 30, 25
L = (TC + 1 + (TILES - 1) * (TC - OV) - 2) * HOP
NAMES = ["stft", "conv1", "conv2", "fc", "fc1x", "deconv2", "final", "istft"]


def main():
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    ctx = Context(stream=stream)
    params = synth_params("dsd", TC, F, seed=2)
    with torch.cuda.stream(stream):
        sep = dcs.Separator("dsd", params, 0.3, TC, OV, 32, F, N, HOP, np.hanning, ctx=ctx)
        audio = ctx.to_device(synth_audio(L, seed=100), np.float32)
        pcm = torch.empty((4, L), dtype=torch.float32, device=audio.device)
    stream.synchronize()
    net, plan = sep.net, sep.plan
    fn = ctx._lib.dcs_separate_batch
    args = (net._h, plan._h, ctypes.c_void_p(audio.data_ptr()), L, 1, L, OV, TILER_SCRIPT, ctypes.c_float(0.3),
            net.arch.eps_mode, 0, ctypes.c_void_p(pcm.data_ptr()), None, None)

    def step():
        rc = fn(*args)
        if rc:
            _lib.check(rc)

    def per_step(stages, k=400, rounds=7):
        net.set_latency_stages(stages)
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        out = []
        for _ in range(rounds):
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / k * 1e6)
        return statistics.median(out), min(out)

    res = {"N": N, "tiles": TILES, "us_per_step": {}}
    ref = None
    sweep = [0, 511, 255] + [511 ^ (1 << b) for b in (0, 1, 6, 7)] + [255 ^ (1 << b) for b in range(2, 6)]
    if os.environ.get("DCS_LAT_EXP_STAGES"):
        sweep = [int(x, 0) for x in os.environ["DCS_LAT_EXP_STAGES"].split(",")]
    for stages in sweep:
        med, lo = per_step(stages)
        net.set_latency_stages(stages)
        step()
        torch.cuda.synchronize()
        out = pcm.cpu().numpy().copy()
        if ref is None:
            ref = out
        res["us_per_step"]["0x%02x" % stages] = {"median": round(med, 2), "min": round(lo, 2),
                                                 "max_abs_diff_vs_throughput": float(np.max(np.abs(out - ref)))}
        print("stages 0x%02x: %.2f us/step (min %.2f), max |pcm - throughput| %.3g"
              % (stages, med, lo, float(np.max(np.abs(out - ref)))), flush=True)

    def breakdown(stages, reps=50):
        net.set_latency_stages(stages)
        ctx.timing("all")
        ctx.timing_reset()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        out = {}
        for tag in _lib.TAGS:
            ms, cnt = ctx.timing_query(tag)
            if cnt:
                out[tag] = round(ms * cnt / reps * 1e3, 2)
        ctx.timing(None)
        ctx.timing_reset()
        return out
    res["events_us"] = {"throughput": breakdown(0), "one_batch": breakdown(511)}
    print("HIP-event us per kernel, throughput kernels:", res["events_us"]["throughput"])
    print("HIP-event us per kernel, one-batch kernels :", res["events_us"]["one_batch"])
    # no graph: eager launches (what a first call costs)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "lat_exp_N%d_T%d.json" % (N, TILES)), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
