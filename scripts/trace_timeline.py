#!/usr/bin/env python3
"""Timeline of one step from a rocprofv3 --kernel-trace CSV: the launches are cut into steps at every occurrence of the
first kernel (default: a name containing 'stft' that is not 'istft'), and for every position in the step the average
kernel duration and the average gap to the next kernel's start are printed -- what is inside kernels and what is between.

    python scripts/trace_timeline.py gpurun_out/prof_lat [first-kernel-substring]
"""
import collections
import csv
import glob
import os
import sys

from trace_by_grid import short


def main():
    root = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else None
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()

    def is_first(name):
        if first:
            return first in name
        return "stft" in name and "istft" not in name
    steps, cur = [], []
    for r in rows:
        if is_first(r[2]) and cur:
            steps.append(cur)
            cur = []
        cur.append(r)
    if cur:
        steps.append(cur)
    by_shape = collections.defaultdict(list)
    for st in steps:
        by_shape[tuple(k[2] for k in st)].append(st)
    for shape, sts in sorted(by_shape.items(), key=lambda kv: -len(kv[1]))[:6]:
        if len(sts) < 20:
            continue
        n = len(shape)
        dur = [0.0] * n
        gap = [0.0] * n
        span = 0.0
        period = []
        for st in sts:
            for i, (s, e, _) in enumerate(st):
                dur[i] += (e - s) / 1e3
                if i + 1 < n:
                    gap[i] += (st[i + 1][0] - e) / 1e3
            span += (st[-1][1] - st[0][0]) / 1e3
        starts = [st[0][0] for st in sts]
        for a, b in zip(starts, starts[1:]):
            if 0 < b - a < 1e6:
                period.append((b - a) / 1e3)
        period.sort()
        k = len(sts)
        print("step shape (%d steps): first start -> last end %.2f us; median start-to-start of consecutive steps %.2f us"
              % (k, span / k, period[len(period) // 2] if period else float("nan")))
        for i, name in enumerate(shape):
            print("   %-40s  dur %6.2f us   gap to next %6.2f us" % (name, dur[i] / k, gap[i] / k if i + 1 < n else float("nan")))
        print("   sum of durations %.2f us, sum of gaps %.2f us" % (sum(dur) / k, sum(gap) / k))
    return 0


if __name__ == "__main__":
    sys.exit(main())
