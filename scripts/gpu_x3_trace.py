#!/usr/bin/env python3
"""Phase times inside colconv_deconv1_fused_x3_kernel (Bach10 f32-class decoder) from s_memtime sums of the two waves of
workgroup 0 (experiment build: scripts/build_exp.sh x3_trace colconv_x3.hip -DDCS_X3_TRACE; DCS_LIB=<that .so>), in shader
clocks, on a 10 s Bach10 clip's worth of tiles (167).  """
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
    dump = ctx._lib.x3_trace_dump
    dump.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 32)()
    rows = []
    for rep in range(8):
        net.forward_raw(x)
        torch.cuda.synchronize()
        if dump(buf) != 0:
            raise SystemExit("x3_trace_dump failed")
        if rep >= 2:
            rows.append(np.array(buf[:32], dtype=np.int64))
    st = np.median(np.array(rows), axis=0)
    names = ["stage 1 (slots of a row pair)", "exchange write + barrier", "tail (behind the barrier)", "block fill (2 barriers, split, LDS writes)",
             "whole run loop", "row pairs, start to end", "tail 0: partner's partial", "tail 1: operand split", "tail 2: stage 2 + piece writes",
             "tail 3: -", "tail 4: shift-add reads", "tail 5: carry + store", "fill: first barrier", "fill: split + LDS writes", "fill: second barrier",
             "fill: next block's loads issued"]
    print("colconv_deconv1_fused_x3_kernel, %d tiles, workgroup 0, median of %d launches (shader clocks):"
          % (n, len(rows)))
    for w in range(2):
        tot = st[16 * w + 4]
        print("  wave %d: " % w + "; ".join("%s %.0f (%.0f %%)" % (names[i], st[16 * w + i], 100.0 * st[16 * w + i] / max(tot, 1)) for i in range(16)))


if __name__ == "__main__":
    main()
