#!/usr/bin/env python3
"""Copy what a final pass (tools/gpu_final.sh TAG, merged back under gpurun_out/) produced into the tracked profiles/ directory.
   python tools/collect_profiles.py TAG"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [
    (f"{tag}/bench.json", f"{tag}_bench.json"), (f"{tag}/pytest.log", f"{tag}_pytest_gpu.log"), (f"{tag}/seg16.json", f"{tag}_segment12_2_16.json"),
    (f"prof_{tag}/kernel_stats.csv", f"{tag}_kernel_stats.csv"), (f"prof_{tag}/pmc_summary.json", f"{tag}_pmc_summary.json"),
    (f"prof_{tag}/pmc_latest.json", "pmc_latest.json"),
    (f"{tag}_cfg/configs_with_counters.json", f"{tag}_configs_4_5.json"), (f"{tag}_cfg/fri_pmc.json", "fri_2_22_pmc.json"),
    (f"{tag}_cfg/sponge_pmc.json", "keccak_sponge_2_20_pmc.json"), (f"{tag}_cfg/fri_kernel_stats.csv", f"{tag}_fri_2_22_kernel_stats.csv"),
    (f"{tag}_cfg/sponge_kernel_stats.csv", f"{tag}_keccak_sponge_2_20_kernel_stats.csv"),
    (f"{tag}_lockstep_kernel_stats.csv", f"{tag}_lockstep_kernel_stats.csv"), (f"{tag}_lockstep_valu.txt", f"{tag}_lockstep_valu.txt"),
    ("lockstep_valu_latest.json", "lockstep_valu_latest.json"),
]
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.isfile(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("copied", src, "->", "profiles/" + dst)
    else:
        print("MISSING", src)
# the full-size CPU oracle time written by the GPU suite's 2^20 parity test: merged into the tracked file, history kept
new = os.path.join(G, "cpu_oracle_full_size_64_threads.json")
if os.path.isfile(new):
    cur_path = os.path.join(P, "cpu_oracle_full_size.json")
    cur = json.load(open(cur_path)) if os.path.isfile(cur_path) else {}
    d = json.load(open(new))
    hist = cur.get("history", {})
    if cur.get("full_size_s") and abs(cur["full_size_s"] - d["full_size_s"]) > 1e-9:
        hist["before_" + tag] = {k: cur.get(k) for k in ("full_size_s", "threads", "cpu_quota")}
    d["history"] = hist
    json.dump(d, open(cur_path, "w"), indent=1)
    print("updated profiles/cpu_oracle_full_size.json: %.1f s on %s threads" % (d["full_size_s"], d.get("threads")))
