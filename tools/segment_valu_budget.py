"""How much VALU issue a 2^16-cycle segment costs: sums SQ_INSTS_VALU (wave-level VALU instructions) and SQ_WAVES over every kernel of a
rocprofv3 --pmc pass of tools/auxpipe_ab.py (one context, R segments after 2 warm-ups + 1 reference) or tools/lockstep_run.py (K
segments in lock-step) and prices them at the leaf kernel's measured issue rate.
usage: segment_valu_budget.py <counter_collection.csv> <segments in the run> [out.json: the digest bench.py quotes, with the code fingerprint]"""
import csv
import collections
import sys


def main():
    path, nseg = sys.argv[1], float(sys.argv[2])
    per = collections.defaultdict(lambda: collections.Counter())
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            per[k]["rows"] += 1
    tot = collections.Counter()
    rows = []
    for k, c in per.items():
        rows.append((c["SQ_INSTS_VALU"], k, c["SQ_WAVES"]))
        tot["valu"] += c["SQ_INSTS_VALU"]
        tot["waves"] += c["SQ_WAVES"]
    rows.sort(reverse=True)
    simds, ghz, cyc = 1024, 2.2, 3.6      # the leaf kernel (k_merkle_leaves 5 waves / k_merkle_leaves_mfma 4 waves per SIMD): 3.5-3.7 cycles per VALU instruction
    ms = lambda v: v / nseg / simds * cyc / (ghz * 1e6)
    print("VALU wave-instructions per segment: %.3e in %.3e waves -> %.2f ms of a full GPU at %.1f cycles per instruction, %.1f GHz" %
          (tot["valu"] / nseg, tot["waves"] / nseg, ms(tot["valu"]), cyc, ghz))
    for v, k, w in rows[:28]:
        print("  %-48s %10.3e inst  %9.0f waves  %6.2f ms" % (k, v / nseg, w / nseg, ms(v)))
    if len(sys.argv) > 3:
        import json
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import code_fingerprint
        json.dump({"what": "wave-level VALU instructions of ONE 2^16-cycle twelve-table segment proven in a lock-step group (rocprofv3 --pmc SQ_INSTS_VALU "
                           "SQ_WAVES over every kernel of tools/lockstep_run.py, divided by the segments of the run)",
                   "valu_insts_per_segment": tot["valu"] / nseg, "waves_per_segment": tot["waves"] / nseg, "simds": simds, "clock_ghz": ghz,
                   "cycles_per_valu_inst": cyc, "budget_ms_per_segment": ms(tot["valu"]),
                   "pricing": "the leaf kernel's measured rate: 3.6 SIMD cycles per wave-level VALU instruction at five waves per SIMD, 2.2 GHz measured clock "
                              "(profiles/pmc_latest.json); 1024 SIMDs",
                   "by_kernel": {k: v / nseg for v, k, w in rows[:20]}, "segments_in_run": nseg, "code_fingerprint": code_fingerprint()},
                  open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
