#!/usr/bin/env python3
"""Where the wall time of ONE 2^16-cycle segment goes, from a rocprofv3 kernel trace of tools/bench_segment.py 16 single:
the last segment of the trace cut into its phases by marker kernels (k_gather_queries ends a table proof; the first
k_quotient* launch starts the serial phase) -- per phase: wall, GPU-busy time on the merged timeline, launches.

  python tools/segment_phases.py <kernel_trace.csv>
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]) for r in rows)
gq = [i for i, e in enumerate(ev) if e[2].startswith("k_gather_queries")]
assert len(gq) >= 24, "need at least two segments"
seg = ev[gq[-13] + 1:gq[-1] + 1]
t0 = seg[0][0]


def busy(evs):
    b, cs, ce = 0, None, None
    for s, e, _ in sorted(evs):
        if cs is None:
            cs, ce = s, e
        elif s > ce:
            b += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return b + (ce - cs if cs is not None else 0)


first_q = next(i for i, e in enumerate(seg) if e[2].startswith("k_quotient"))
par = seg[:first_q]
print("segment: %.2f ms wall, %d launches, busy %.2f ms" % ((max(e[1] for e in seg) - t0) / 1e6, len(seg), busy(seg) / 1e6))
print("parallel phases (trace commitments, CTL data, auxiliary commitments): %.2f ms wall, %d launches, busy %.2f ms" % (
    (seg[first_q][0] - t0) / 1e6, len(par), busy(par) / 1e6))
# table proofs: from the first quotient launch (or the end of the previous gather) to the table's gather
bounds = [i for i, e in enumerate(seg) if e[2].startswith("k_gather_queries")]
start = first_q
print("table proofs (serial, one transcript):")
for t, b in enumerate(bounds):
    part = seg[start:b + 1]
    w = (part[-1][1] - (seg[start - 1][1] if start > first_q else part[0][0])) / 1e6
    names = {}
    for s, e, n in part:
        names[n] = names.get(n, 0) + (e - s)
    top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
    print("  table %2d: %6.2f ms wall, %3d launches, busy %5.2f ms | %s" % (t, w, len(part), busy(part) / 1e6,
                                                                           ", ".join("%s %.0f us" % (n[:28], v / 1e3) for n, v in top)))
    start = b + 1

# ---- the parallel phases in detail: every launch of >= 80 us with its queue, start offset and duration (who is the critical lane?)
qcol = "Queue_Id" if "Queue_Id" in rows[0] else None
if qcol and "--lanes" in sys.argv:
    evq = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r[qcol]) for r in rows)
    lo, hi = par[0][0], seg[first_q][0]
    print("parallel phases, launches >= 80 us (offset ms, duration us, queue, kernel):")
    for s, e, n, q in evq:
        if lo <= s < hi and e - s >= 80000:
            print("  %7.3f %8.1f  q%-3s %s" % ((s - lo) / 1e6, (e - s) / 1e3, q, n))
    per_q = {}
    for s, e, n, q in evq:
        if lo <= s < hi:
            per_q.setdefault(q, []).append((s, e))
    print("per queue in the parallel phases: launches, busy ms, first start, last end (ms)")
    for q, iv in sorted(per_q.items()):
        print("  q%-3s %4d %7.2f %7.2f %7.2f" % (q, len(iv), sum(e - s for s, e in iv) / 1e6, (min(s for s, _ in iv) - lo) / 1e6, (max(e for _, e in iv) - lo) / 1e6))
