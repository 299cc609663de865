#!/usr/bin/env python3
"""GPU occupancy over time of a multi-context lock-step run from a rocprofv3 kernel trace: in the steady-state window (by default the 1.2 s that end
0.3 s before the last kernel: tools/sweep_lockstep.py times its last three calls per context, ~1.8 s at 8 x 8) -- the fraction of time ANY kernel is running, the mean number of kernels in
flight, and the same for the "large" kernels only (>= 1 ms: leaf hashing, big transforms).   python tools/lockstep_timeline.py <kernel_trace.csv> [window_s=1.2] [tail_s=0.3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows if not r["Kernel_Name"].startswith("__amd"))
t_first, t_last = ev[0][0], max(e[1] for e in ev)
wlen = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
tail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
hi = t_last - tail * 1e9
lo = max(t_first, hi - wlen * 1e9)
win = [(max(s, lo), min(e, hi), n) for s, e, n in ev if e > lo and s < hi]


def union(iv):
    busy, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if cs is None:
            cs, ce = s, e
        elif s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + (ce - cs if cs is not None else 0)


tot = hi - lo
allb = union([(s, e) for s, e, _ in win])
big = [(s, e) for s, e, n in win if e - s >= 1e6]
print("window %.1f ms, %d kernel launches in it" % (tot / 1e6, len(win)))
print("some kernel running: %.1f %% of the time; kernels in flight on average: %.2f" % (100.0 * allb / tot, sum(e - s for s, e, _ in win) / tot))
print("a kernel of >= 1 ms running: %.1f %% of the time (%.2f in flight on average)" % (100.0 * union(big) / tot, sum(e - s for s, e in big) / tot))
by = {}
for s, e, n in win:
    by.setdefault(n, [0, 0])
    by[n][0] += e - s
    by[n][1] += 1
print("kernel-time share in the window (sum of durations / window):")
for n, (t, k) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  %6.2f  %5d launches  mean %8.1f us  %s" % (t / tot, k, t / k / 1e3, n))
