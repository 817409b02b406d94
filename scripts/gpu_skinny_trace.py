#!/usr/bin/env python3
"""Phase times inside gemm_bf16x3_skinny_kernel (per-source dense layers of the Bach10 graph) from s_memtime sums of wave 0 of
the middle workgroup (experiment build: scripts/build_exp.sh sk_trace gemm_bf16x3.hip -DDCS_SKINNY_TRACE; DCS_LIB=<that .so>),
in shader clocks, on a 10 s Bach10 clip's worth of tiles (167)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deepconvsep_amd.runtime import Network, default_context  # noqa: E402
from deepconvsep_amd.synth import synth_params  # noqa: E402


def main():
    F, n = 2049, int(os.environ.get("DCS_TRACE_TILES", "167"))
    ctx = default_context()
    net = Network(ctx, "bach10", synth_params("bach10", 30, F, seed=4), 30, F)
    x = ctx.to_device(np.abs(np.random.RandomState(1).randn(n, 1, 30, F)).astype(np.float32), np.float32)
    dump = ctx._lib.skinny_trace_dump
    dump.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 8)()
    rows = []
    for rep in range(8):
        net.forward_raw(x)
        torch.cuda.synchronize()
        if dump(buf) != 0:
            raise SystemExit("skinny_trace_dump failed")
        if rep >= 2:
            rows.append(np.array(buf[:8], dtype=np.int64))
    st = np.median(np.array(rows), axis=0)
    names = ["split + LDS writes", "first barrier", "next tile's loads issued", "fragment reads + MFMAs", "second barrier", "epilogue", "whole kernel"]
    print("gemm_bf16x3_skinny_kernel (Bach10 per-source dense layers, %d tiles), wave 0 of the middle workgroup, median of %d launches (shader clocks):" % (n, len(rows)))
    print("  " + "; ".join("%s %.0f (%.0f %%)" % (names[i], st[i], 100.0 * st[i] / max(st[6], 1)) for i in range(7)))


if __name__ == "__main__":
    main()
