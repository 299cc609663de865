#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/collect_pmc.sh: per-kernel average duration and per-launch PMC values."""
import csv, glob, os, sys, collections, json
out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        res.setdefault(r["Name"].split("(")[0], {})["stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "max_ns": float(r["MaxNs"]), "total_ns": float(r["TotalDurationNs"])}
for d in glob.glob(os.path.join(out, "pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, vals in cs.items():
                res.setdefault(k, {}).setdefault("pmc", {})[c] = {"launches": len(vals), "max": max(vals), "mean": sum(vals) / len(vals), "sum": sum(vals)}
print(json.dumps(res, indent=1, sort_keys=True))
