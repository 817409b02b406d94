#!/usr/bin/env python3
"""In-kernel timeline of the one-batch kernels (experiment build with -DDCS_LAT_TRACE, scripts/build_exp.sh lattrace
dsd_lat.hip -DDCS_LAT_TRACE; run with DCS_LIB=deepconvsep_amd/_exp_lattrace.so): shader-clock stamps of thread 0 of the
first and the last workgroup of every launch, as differences between consecutive marks."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import deepconvsep_amd as dcs  # noqa: E402
from deepconvsep_amd.arch import TILER_SCRIPT  # noqa: E402
from deepconvsep_amd.runtime import Context  # noqa: E402
from deepconvsep_amd.synth import synth_audio, synth_params  # noqa: E402

N = int(os.environ.get("DCS_LAT_EXP_N", "2048"))
F, HOP, TC, OV = N // 2 + 1, 512, 30, 25
L = (TC + 1 + 31 * (TC - OV) - 2) * HOP
MARKS = {
    0: ("conv1", ["args+request", "operands arrive", "mfma+lds write", "barrier", "reduce+store issue", "store ack"]),
    1: ("conv2", ["args+request", "operands arrive", "mfma+lds write", "barrier", "reduce+store issue", "store ack"]),
    2: ("fc", ["args+request", "operands arrive", "mfma+lds write", "barrier", "reduce+store issue", "store ack"]),
    3: ("fc1x", ["args+request", "operands arrive", "mfma+lds write", "barrier", "reduce+store issue", "store ack"]),
    4: ("mid", ["weights requested", "conv2+publish", "sweep C2", "fc+publish", "sweep Z", "dense+publish", "sweep D",
                "deconv2+store issue", "store ack"]),
    5: ("final", ["tables+request B/mix", "barrier", "plan+request A", "A arrives", "LDS fill+barrier", "fold", "swap+store issue",
                  "store ack"]),
    6: ("stft", ["request+LDS fill", "barrier", "fft", "post+store issue", "store ack"]),
    7: ("ifft", ["request+LDS fill", "barrier", "pre+fft", "window+store issue", "store ack"]),
}


def main():
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    ctx = Context(stream=stream)
    params = synth_params("dsd", TC, F, seed=2)
    with torch.cuda.stream(stream):
        sep = dcs.Separator("dsd", params, 0.3, TC, OV, 32, F, N, HOP, np.hanning, ctx=ctx)
        audio = ctx.to_device(synth_audio(L, seed=100), np.float32)
        pcm = torch.empty((4, L), dtype=torch.float32, device=audio.device)
    sep.net.set_latency_stages(int(os.environ.get('DCS_LAT_TRACE_STAGES', '511')))
    fn = ctx._lib.dcs_separate_batch
    args = (sep.net._h, sep.plan._h, ctypes.c_void_p(audio.data_ptr()), L, 1, L, OV, TILER_SCRIPT, ctypes.c_float(0.3),
            sep.net.arch.eps_mode, 0, ctypes.c_void_p(pcm.data_ptr()), None, None)
    dump = ctx._lib.lat_trace_dump
    dump.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * (16 * 64))()
    acc = {}
    reps = 40
    for rep in range(reps + 10):
        for _ in range(8):                    # steady state: graph replays back to back
            fn(*args)
        torch.cuda.synchronize()
        if dump(buf, 16 * 64) != 0:
            raise SystemExit("library was not built with -DDCS_LAT_TRACE")
        if rep < 10:
            continue
        a = np.array(buf[:], dtype=np.uint64).reshape(16, 64).astype(np.int64)
        for kid, (name, marks) in MARKS.items():
            n = len(marks) + 1
            for which, off in (("first wg", 0), ("last wg", 32)):
                st = a[kid, off:off + n]
                d = np.diff(st)
                acc.setdefault((kid, which), []).append(d)
            rt = a[kid, 31] - a[kid, 30]      # 100 MHz ticks between the first and the last stamp of the first workgroup
            acc.setdefault((kid, "rt"), []).append([rt, a[kid, n - 1] - a[kid, 0]])
    for kid, (name, marks) in MARKS.items():
        rt = np.median(np.array(acc[(kid, "rt")]), axis=0)
        clk_per_us = rt[1] / (rt[0] / 100.0) if rt[0] > 0 else float("nan")
        print("%s: first stamp -> last stamp %.2f us (shader clock %.0f MHz)" % (name, rt[0] / 100.0, clk_per_us))
        for which in ("first wg", "last wg"):
            d = np.median(np.array(acc[(kid, which)]), axis=0)
            print("   %-8s " % which + " | ".join("%s %d clk (%.2f us)" % (m, x, x / clk_per_us) for m, x in zip(marks, d)))


if __name__ == "__main__":
    main()
