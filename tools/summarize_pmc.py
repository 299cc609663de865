#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/collect_pmc.sh: per-kernel average duration and per-launch PMC values, and write the
digest bench.py quotes (pmc_latest.json: HBM traffic of the dominant kernel and of the NTT kernels, VALU instructions per wave),
stamped with the fingerprint of the kernel sources so that a stale file is never quoted.

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 it reports half the bytes
of wide coalesced streaming reads)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 3      # proofs made by the profiled command (steps + warmup)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True)

# ---- the digest bench.py reads
try:
    from bench import code_fingerprint
    fp = code_fingerprint()
except Exception:
    fp = None
latest = {"source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md; "
                    "the 262-column trace launch = max over the launches)" % os.path.basename(out.rstrip("/")),
          "code_fingerprint": fp, "proofs_profiled": proofs}
# the one-lane leaf kernel: k_merkle_leaves_mfma when the tuning key leaf_mfma is on (the default), k_merkle_leaves otherwise; the digest
# keeps the key "k_merkle_leaves" and names the kernel it was taken from
leaf_kernel = "k_merkle_leaves_mfma" if "k_merkle_leaves_mfma" in res else "k_merkle_leaves"
ml = res.get(leaf_kernel, {}).get("pmc", {})
if "FETCH_SIZE" in ml and "WRITE_SIZE" in ml:
    fb, wb = 2 * 1024 * ml["FETCH_SIZE"]["max"], 1024 * ml["WRITE_SIZE"]["max"]
    latest["k_merkle_leaves"] = {"kernel": leaf_kernel, "fetch_bytes": fb, "write_bytes": wb, "traffic_bytes": fb + wb}
    if "SQ_INSTS_VALU" in ml and "SQ_WAVES" in ml:
        latest["k_merkle_leaves"]["valu_insts_per_wave"] = ml["SQ_INSTS_VALU"]["max"] / ml["SQ_WAVES"]["max"]
    if "GRBM_GUI_ACTIVE" in ml:
        latest["k_merkle_leaves"]["grbm_gui_active_max"] = ml["GRBM_GUI_ACTIVE"]["max"]
        st = res.get(leaf_kernel, {}).get("stats", {})
        if st.get("max_ns"):
            # GPU cycles of the launch (the counter is summed over the 8 XCDs) over its duration in the kernel-trace pass: the clock the
            # kernel really ran at, which is what its issue fractions should be taken against (nominal: 2.4 GHz)
            latest["k_merkle_leaves"]["clock_ghz_measured"] = ml["GRBM_GUI_ACTIVE"]["max"] / 8 / st["max_ns"]
# the whole proof against the VALU-issue roof: wave-level VALU instructions of every kernel, per proof
valu_all = sum(v["pmc"]["SQ_INSTS_VALU"]["sum"] for k, v in res.items() if "SQ_INSTS_VALU" in v.get("pmc", {}))
if valu_all:
    latest["whole_proof"] = {"valu_insts_per_proof": valu_all / proofs,
                             "valu_insts_per_proof_by_kernel": {k: v["pmc"]["SQ_INSTS_VALU"]["sum"] / proofs for k, v in sorted(
                                 res.items(), key=lambda kv: -kv[1].get("pmc", {}).get("SQ_INSTS_VALU", {}).get("sum", 0))[:12]
                                 if "SQ_INSTS_VALU" in v.get("pmc", {})}}
is_ntt = lambda k: "k_ntt_" in k or "k_lde_upper" in k   # every NTT / LDE kernel of ntt.hip
fetch = sum(v["pmc"]["FETCH_SIZE"]["sum"] for k, v in res.items() if is_ntt(k) and "FETCH_SIZE" in v.get("pmc", {}))
write = sum(v["pmc"]["WRITE_SIZE"]["sum"] for k, v in res.items() if is_ntt(k) and "WRITE_SIZE" in v.get("pmc", {}))
if fetch and write:
    latest["ntt"] = {"fetch_bytes_per_proof": 2 * 1024 * fetch / proofs, "write_bytes_per_proof": 1024 * write / proofs,
                     "traffic_bytes_per_proof": (2 * 1024 * fetch + 1024 * write) / proofs,
                     "kernel_ms_per_proof": sum(v["stats"]["total_ns"] for k, v in res.items() if is_ntt(k) and "stats" in v) / proofs / 1e6}
json.dump(latest, open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
print(json.dumps(latest, indent=1))
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("stats", {}).get("total_ns", 0))[:14]:
    if "stats" in v:
        print("%-60s calls %4d  avg %9.1f us  total %8.2f ms" % (k[:60], v["stats"]["calls"], v["stats"]["avg_ns"] / 1e3, v["stats"]["total_ns"] / 1e6))
