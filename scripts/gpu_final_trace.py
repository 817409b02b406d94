#!/usr/bin/env python3
"""In-kernel timeline of final_bf16x3_kernel at the driver's launch shape (20 clips x 32 tiles in one dcs_separate_batch
call; experiment build: scripts/build_exp.sh finaltrace dsd_bf16x3.hip -DDCS_FINAL_TRACE; DCS_LIB=deepconvsep_amd/
_exp_finaltrace.so): s_memtime stamps of wave 0 of the middle workgroup of the middle clip, as differences between
consecutive marks, in shader clocks.   DCS_TRACE_CLIPS / DCS_TRACE_TILES change the shape."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import deepconvsep_amd as dcs  # noqa: E402
from deepconvsep_amd.synth import synth_audio, synth_params  # noqa: E402

N, HOP, TC, OV = 2048, 512, 30, 25
CLIPS = int(os.environ.get("DCS_TRACE_CLIPS", "20"))
TILES = int(os.environ.get("DCS_TRACE_TILES", "32"))
L = (TC + 1 + (TILES - 1) * (TC - OV) - 2) * HOP


def main():
    F = N // 2 + 1
    sep = dcs.Separator("dsd", synth_params("dsd", TC, F, seed=2), 0.3, TC, OV, 32, F, N, HOP, np.hanning)
    sep.net.set_latency_stages(0)
    ctx = sep.ctx
    audio = ctx.to_device(np.stack([synth_audio(L, seed=3 + c) for c in range(CLIPS)]), np.float32)
    out = torch.empty((CLIPS, 4, L), dtype=torch.float32, device=audio.device)
    dump = ctx._lib.final_trace_dump
    dump.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 64)()
    rows = []
    for rep in range(14):
        if CLIPS > 1:
            sep.net.separate_batch(sep.plan, audio, OV, sep.tiler, 0.3, out=out)
        else:
            sep.net.separate(sep.plan, audio[0], OV, sep.tiler, 0.3, out=out[0])
        torch.cuda.synchronize()
        if dump(buf, 64) != 0:
            raise SystemExit("library was not built with -DDCS_FINAL_TRACE")
        if rep >= 2:
            rows.append(np.array(buf[:31], dtype=np.int64))
    st = np.median(np.array(rows), axis=0)
    tick = 1.0                                                  # s_memtime counts shader clocks on gfx950 (a 58 us kernel: ~53 k per wave life)
    marks = [("tables written; B fragments + mixture rows requested", 0, 1), ("table barrier", 1, 2),
             ("staging plan, first A set requested", 2, 3)]
    for j in range(6):
        b = 4 + 4 * j
        marks += [("tile %d: own transfers landed (vmcnt 0)" % j, b - 1 if j else 3, b), ("tile %d: barrier" % j, b, b + 1),
                  ("tile %d: next tile requested" % j, b + 1, b + 2), ("tile %d: fragments + 72 MFMAs + mask + fold" % j, b + 2, b + 3)]
    marks += [("(loop exit)", 27, 28), ("16 row stores issued", 28, 29), ("stores acknowledged", 29, 30)]
    print("final_bf16x3_kernel, %d clips x %d tiles, wave 0 of the middle workgroup of the middle clip, median of %d launches:" % (CLIPS, TILES, len(rows)))
    for name, a, b in marks:
        print("  %-58s %8.0f clk" % (name, (st[b] - st[a]) * tick))
    print("  %-58s %8.0f clk" % ("wave life", (st[30] - st[0]) * tick))


if __name__ == "__main__":
    main()
