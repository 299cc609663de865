#!/usr/bin/env python3
"""Digest of the rocprofv3 passes of tools/gpu_configs.sh: per-kernel HBM bytes (FETCH_SIZE doubled per MI355X_MICROARCH.md,
WRITE_SIZE; both in KiB) and durations for BASELINE configs 4 (fri) and 5 (sponge).  Writes <dir>/<workload>_pmc.json, stamped with
bench.py's code fingerprint (tools/bench_configs.py quotes counter bytes only from a file collected on the code it runs)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import code_fingerprint  # noqa: E402
for wl in ("fri", "sponge"):
    res = {}
    f = os.path.join(out, "%s_kernel_stats.csv" % wl)
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            res.setdefault(r["Name"].split("(")[0], {})["stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"])}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        for g in glob.glob(os.path.join(out, "pmc_%s_%s" % (wl, cname), "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(g)):
                if r["Counter_Name"] == cname:
                    agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
            for k, vals in agg.items():
                scale = 2 * 1024 if cname == "FETCH_SIZE" else 1024
                res.setdefault(k, {})[cname.lower() + "_bytes_total"] = scale * sum(vals)
                res[k][cname.lower() + "_launches"] = len(vals)
    if not res:
        continue
    tot_f = sum(v.get("fetch_size_bytes_total", 0) for v in res.values())
    tot_w = sum(v.get("write_size_bytes_total", 0) for v in res.values())
    tot_ns = sum(v.get("stats", {}).get("total_ns", 0) for v in res.values())
    digest = {"workload": wl, "code_fingerprint": code_fingerprint(), "calls_profiled": 4,
              "note": "totals over the whole profiled command (tools/bench_configs.py %s: warm-up + 3 timed repetitions + set-up); "
                                      "FETCH_SIZE doubled (gfx950 correction), separate --pmc passes" % wl,
              "total_fetch_bytes": tot_f, "total_write_bytes": tot_w, "total_kernel_ms": tot_ns / 1e6,
              "aggregate_GBps_over_kernel_time": (tot_f + tot_w) / max(tot_ns, 1), "kernels": res}
    json.dump(digest, open(os.path.join(out, "%s_pmc.json" % wl), "w"), indent=1, sort_keys=True)
    print(wl, "fetch %.2f GB write %.2f GB kernel time %.1f ms -> %.0f GB/s" % (tot_f / 1e9, tot_w / 1e9, tot_ns / 1e6, (tot_f + tot_w) / max(tot_ns, 1)))
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("stats", {}).get("total_ns", 0))[:12]:
        st = v.get("stats", {})
        b = v.get("fetch_size_bytes_total", 0) + v.get("write_size_bytes_total", 0)
        print("  %-56s calls %5d total %8.2f ms  %8.2f GB  %6.0f GB/s" % (k[:56], st.get("calls", 0), st.get("total_ns", 0) / 1e6, b / 1e9, b / max(st.get("total_ns", 1), 1)))
