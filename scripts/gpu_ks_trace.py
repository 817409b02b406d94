#!/usr/bin/env python3
"""In-kernel timeline of gemm_ks_kernel for ONE layer of the DSD encoder at the driver's launch shape (20 clips x 32 tiles):
experiment build `scripts/build_exp.sh kstrace gemm_ks.hip -DDCS_KS_TRACE=<K>` (K = 1028 conv1, 780 conv2, 832 bottleneck, 128
per-source dense), DCS_LIB=deepconvsep_amd/_exp_kstrace.so.  s_memtime stamps of lane 0 of every wave of the middle workgroup,
as differences in shader clocks (median of the launches)."""
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
    dump = ctx._lib.ks_trace_dump
    dump.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 128)()
    rows = []
    for rep in range(14):
        sep.net.separate_batch(sep.plan, audio, OV, sep.tiler, 0.3, out=out)
        torch.cuda.synchronize()
        if dump(buf, 128) != 0:
            raise SystemExit("library was not built with -DDCS_KS_TRACE")
        if rep >= 2:
            rows.append(np.array(buf[:128], dtype=np.int64).reshape(16, 8))
    st = np.median(np.array(rows), axis=0)
    t0 = st[:, 0].min()
    names = ["start (after the earliest wave)", "A + B requested", "operands landed and split", "multiplied",
             "partials written", "barrier passed", "stores issued"]
    print("gemm_ks_kernel, layer %s, %d clips x %d tiles; clocks since the workgroup's earliest wave:" % (os.environ.get("DCS_KS_LAYER", "?"), CLIPS, TILES))
    print("  wave " + " ".join("%9s" % n[:9] for n in ["start", "requested", "landed", "multiplied", "partials", "barrier", "stored"]))
    for w in range(16):
        if st[w, 0] == 0:
            continue
        print("  %4d " % w + " ".join("%9.0f" % (st[w, j] - t0 if st[w, j] else float('nan')) for j in range(7)))
    print("  (" + "; ".join(names) + ")")


if __name__ == "__main__":
    main()
