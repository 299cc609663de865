#!/usr/bin/env python3
"""tools/bench_host_line.py -- one line of the host-resident figures of a bench.py JSON line on stdin (A/B runs)."""
import json
import sys
j = json.loads(sys.stdin.read())
h = j.get("host_resident") or {}
ls = [(x.get("contexts"), x.get("segments_per_call"), round(x.get("segments_per_s", 0), 1)) for x in (j.get("segment_2_16") or {}).get("lockstep", [])] if isinstance((j.get("segment_2_16") or {}).get("lockstep"), list) else None
print(sys.argv[1] if len(sys.argv) > 1 else "", "value %.2f" % j["value"], "intra %s" % (h.get("intra_proof") or {}).get("proofs_per_s"), "cross %s" % h.get("proofs_per_s"),
      "h2d %s" % h.get("pcie_h2d_GBps"), "err %s" % j.get("host_resident_error"), "fri %s" % (j.get("fri_2_22") or {}).get("ms"), "lockstep %s" % ls)
