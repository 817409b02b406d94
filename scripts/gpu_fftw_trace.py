#!/usr/bin/env python3
"""In-kernel timeline of stft_forward_wave_kernel under full load (experiment build: scripts/build_exp.sh fftwtrace
fft_wave.hip -DDCS_FFTW_TRACE; DCS_LIB=deepconvsep_amd/_exp_fftwtrace.so): shader-clock stamps of wave 0 of the middle
workgroup of a 4096-tile launch, as differences between consecutive marks."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import deepconvsep_amd as dcs  # noqa: E402
from deepconvsep_amd.runtime import default_context  # noqa: E402
from deepconvsep_amd.synth import synth_audio, synth_params  # noqa: E402

N, HOP, TC, OV = 2048, 512, 30, 25
TILES = int(os.environ.get("DCS_TRACE_TILES", "4096"))
CLIPS = int(os.environ.get("DCS_TRACE_CLIPS", "1"))       # > 1: one dcs_separate_batch launch group (the driver's shape: 20 x 32)
L = (TC + 1 + (TILES - 1) * (TC - OV) - 2) * HOP
MARKS = ["table fill + barrier", "twiddles read, samples + window requested", "samples arrive", "pass 1 + stores issued",
         "exchange 1 (LDS)", "pass 2 + stores issued", "exchange 2 (LDS)", "pass 3 + stores issued", "split + sqrt / rcp + global stores issued",
         "stores acknowledged"]


def main():
    ctx = default_context()
    F = N // 2 + 1
    sep = dcs.Separator("dsd", synth_params("dsd", TC, F, seed=2), 0.3, TC, OV, 32, F, N, HOP, np.hanning)
    audio = ctx.to_device(synth_audio(L, seed=3) if CLIPS == 1 else np.stack([synth_audio(L, seed=3 + c) for c in range(CLIPS)]), np.float32)
    out = torch.empty((CLIPS, 4, L), dtype=torch.float32, device=audio.device) if CLIPS > 1 else None
    if CLIPS > 1:
        sep.net.set_latency_stages(0)
    dump = ctx._lib.fftw_trace_dump
    dump.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 64)()
    rows = []
    for rep in range(12):
        if CLIPS > 1:
            sep.net.separate_batch(sep.plan, audio, OV, sep.tiler, 0.3, out=out)
        else:
            sep.net.separate(sep.plan, audio, OV, sep.tiler, 0.3)
        torch.cuda.synchronize()
        if dump(buf, 64) != 0:
            raise SystemExit("library was not built with -DDCS_FFTW_TRACE")
        st = np.array(buf[:11], dtype=np.int64)
        if rep >= 2:
            rows.append(np.diff(st))
    d = np.median(np.array(rows), axis=0)
    print("stft_forward_wave_kernel, %d clip(s) x %d tiles, wave 0 of the middle workgroup, median of %d launches (shader clocks):" % (CLIPS, TILES, len(rows)))
    for name, c in zip(MARKS, d):
        print("  %-48s %7.0f clk" % (name, c))
    print("  %-48s %7.0f clk" % ("wave life", d.sum()))


if __name__ == "__main__":
    main()
