#!/usr/bin/env python3
"""Idle-time analysis of a rocprofv3 kernel trace of tools/bench_segment.py 16 single (development aid).

  python tools/segment_timeline.py <kernel_trace.csv> [nsegments=4]
Takes the LAST segment of the run (the last 1/nsegments of the time between the first and last prover kernel is a good enough window
when the segments are equal), merges the busy intervals of all queues and prints: busy / idle time, the idle gaps by the kernel that
ends before the gap -> the kernel that starts after it, and the longest gaps in order.
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]) for r in rows)
# the segments: find the last launch of the first kernel of a segment proof by looking for the largest idle gaps (between segments the host
# downloads the proof and starts over) -- simpler: window = last quarter by count of "k_gather_queries" launches (12 per segment)
gq = [i for i, e in enumerate(ev) if e[2].startswith("k_gather_queries")]
per = 12
assert len(gq) >= 2 * per, "need at least two segments in the trace"
lo = gq[-per - 1] + 1          # first kernel after the last gather of the previous segment
hi = gq[-1]
seg = ev[lo:hi + 1]
t0, t1 = seg[0][0], max(e[1] for e in seg)
busy, cur_s, cur_e, last_name = 0, seg[0][0], seg[0][1], seg[0][2]
gaps = []
end_name = seg[0][2]
for s, e, n in seg[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, end_name, n, cur_e - t0))
        busy += cur_e - cur_s
        cur_s, cur_e, end_name = s, e, n
    elif e > cur_e:
        cur_e, end_name = e, n
busy += cur_e - cur_s
tot = t1 - t0
print("segment window %.2f ms, %d launches: busy %.2f ms, idle %.2f ms in %d gaps" % (tot / 1e6, len(seg), busy / 1e6, (tot - busy) / 1e6, len(gaps)))
by = defaultdict(lambda: [0, 0])
for g, a, b, _ in gaps:
    by[(a, b)][0] += g
    by[(a, b)][1] += 1
print("idle by (kernel before -> kernel after):")
for (a, b), (g, k) in sorted(by.items(), key=lambda kv: -kv[1][0])[:28]:
    print("  %8.1f us in %3d gaps (%.1f us each)  %s -> %s" % (g / 1e3, k, g / k / 1e3, a, b))
hist = defaultdict(int)
for g, *_ in gaps:
    hist[min(int(g / 5e3) * 5, 100)] += 1
print("gap histogram (us bucket: count):", dict(sorted(hist.items())))
print("longest gaps (us, at ms offset): before -> after")
for g, a, b, off in sorted(gaps, key=lambda x: -x[0])[:14]:
    print("  %8.1f us at %6.2f ms  %s -> %s" % (g / 1e3, off / 1e6, a, b))
tot_by = defaultdict(lambda: [0, 0])
for s_, e_, n_ in seg:
    tot_by[n_][0] += e_ - s_
    tot_by[n_][1] += 1
print("kernel time in the window (ms, launches, mean us):")
for n_, (t_, k_) in sorted(tot_by.items(), key=lambda kv: -kv[1][0])[:30]:
    print("  %7.3f ms %4d %8.1f us  %s" % (t_ / 1e6, k_, t_ / k_ / 1e3, n_))
