#!/usr/bin/env python3
"""HBM traffic per kernel launch from the rocprofv3 --pmc passes of scripts/gpu_traffic_r04.sh -> gpurun_out/traffic.json
(committed as profiles/r04_traffic.json, read by bench.py):

  "all":  { "<kernel>@grid_threads=<n>": {FETCH_SIZE_KiB, WRITE_SIZE_KiB, launches} }   headline / one-batch / saturating shapes
  "legs": { "<leg>": { "<kernel>@grid_threads=<n>": {...} } }                               one pass pair per leg (--only-legs)

FETCH_SIZE and WRITE_SIZE come from separate passes (they do not fit the TCC counter slots together); values are averaged
over the launches of the same kernel and grid.  bench.py reports 2*FETCH + WRITE (gfx950: FETCH_SIZE counts half the bytes of
wide coalesced reads, MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void\s+)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:48]


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def merge(out, tag):
    fetch, write = load(os.path.join(out, "pmc_fetch_" + tag)), load(os.path.join(out, "pmc_write_" + tag))
    recs = {}
    for k in sorted(set(fetch) | set(write)):
        rec = {}
        if "FETCH_SIZE" in fetch.get(k, {}):
            v = fetch[k]["FETCH_SIZE"]
            rec["FETCH_SIZE_KiB"] = round(sum(v) / len(v), 2)
            rec["launches"] = len(v)
        if "WRITE_SIZE" in write.get(k, {}):
            v = write[k]["WRITE_SIZE"]
            rec["WRITE_SIZE_KiB"] = round(sum(v) / len(v), 2)
        recs["%s@grid_threads=%d" % k] = rec
    return recs


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    tags = sorted(set(os.path.basename(p)[len("pmc_fetch_"):] for p in glob.glob(os.path.join(out, "pmc_fetch_*")) if os.path.isdir(p)))
    doc = {"all": {}, "legs": {}, "passes": tags,
           "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes per workload (scripts/gpu_traffic_r04.sh); "
                   "per launch, KiB, averaged over the launches of the same kernel and grid"}
    for t in tags:
        recs = merge(out, t)
        if t.startswith("leg_"):
            doc["legs"][t[4:]] = recs
        else:
            for k, v in recs.items():
                if k not in doc["all"] or v.get("launches", 0) > doc["all"][k].get("launches", 0):
                    doc["all"][k] = v
    json.dump(doc, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    for k in sorted(doc["all"]):
        r = doc["all"][k]
        if "FETCH_SIZE_KiB" in r and "WRITE_SIZE_KiB" in r:
            print("%-60s fetch %10.1f KiB  write %10.1f KiB  2F+W %8.2f MB  (%d launches)" % (
                k, r["FETCH_SIZE_KiB"], r["WRITE_SIZE_KiB"], (2 * r["FETCH_SIZE_KiB"] + r["WRITE_SIZE_KiB"]) / 1024.0, r.get("launches", 0)))
    for leg, recs in doc["legs"].items():
        print("leg", leg)
        for k in sorted(recs):
            r = recs[k]
            if "FETCH_SIZE_KiB" in r and "WRITE_SIZE_KiB" in r:
                print("   %-58s 2F+W %8.2f MB" % (k, (2 * r["FETCH_SIZE_KiB"] + r["WRITE_SIZE_KiB"]) / 1024.0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
