#!/usr/bin/env python3
"""Merge the records of a partial traffic visit (gpurun_out/traffic.json from scripts/gpu_traffic_r03.sh with
DCS_TRAFFIC_LEGS set) into the committed file:  python scripts/traffic_merge.py gpurun_out/traffic.json profiles/r03_traffic.json"""
import json
import sys

new, dst = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
for k, v in new.get("all", {}).items():
    dst.setdefault("all", {})[k] = v
for leg, recs in new.get("legs", {}).items():
    dst.setdefault("legs", {})[leg] = recs
dst["passes"] = sorted(set(dst.get("passes", [])) | set(new.get("passes", [])))
json.dump(dst, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print("merged legs %s, %d headline records" % (sorted(new.get("legs", {})), len(new.get("all", {}))))
