#!/usr/bin/env python3
"""Split a rocprofv3 --kernel-trace CSV by (kernel, grid size): a bench run mixes launch sizes (one 32-tile batch,
a launch group, one 4096-tile clip, the other graphs' clips) that the --stats summary averages together.

    python scripts/trace_by_grid.py gpurun_out/prof > profiles/rNN_kernel_durations_by_grid.txt
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void\s+)?([A-Za-z0-9_]+)(<[^(]*>)?", name)
    if not m:
        return name[:40]
    base, targs = m.group(1), m.group(2) or ""
    return base + (targs if len(targs) <= 24 else "")


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no *kernel_trace.csv under", root)
        return 1
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            grid = int(r.get("Grid_Size") or int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)))
            agg[(short(r["Kernel_Name"]), grid)].append(dur)
    print("kernel, grid threads, launches, avg us, min us, max us, total ms   (rocprofv3 --kernel-trace)")
    for (k, g), v in sorted(agg.items()):
        print("%-44s %10d %6d %9.1f %9.1f %9.1f %9.3f" % (k, g, len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3))
    return 0


if __name__ == "__main__":
    sys.exit(main())
