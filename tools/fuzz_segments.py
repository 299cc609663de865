#!/usr/bin/env python3
"""Differential fuzz of zkm_prove_segment against the CPU oracle's prove_with_traces: the committed twelve-table test segment with
every table tiled to a random height (its own height + 0 .. 4 doublings; tiled rows are not a valid witness across the seams, neither
prover looks at validity), random public values; all twelve proof blobs and the CTL challenges word for word.  Test infrastructure.

  python tools/fuzz_segments.py [cases=12] [seed=1] [stack=0]
stack > 0: every case is ONE zkm_prove_segments call of 2 .. stack segments in lock-step (each with its own random heights, row rotation
and public values; max_stack random per case); every blob is compared with the oracle's prove_with_traces of that segment."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import zkm_amd  # noqa: E402
from oracle import oracle_py  # noqa: E402
from zkm_amd import tables as T  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
stack = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
o = oracle_py.Oracle()
o.set_threads(min(64, os.cpu_count() or 1))
ctx = zkm_amd.Context(0)
for kv in filter(None, os.environ.get("ZKM_SEG_TUNING", "").split(",")):   # "key=value,...": zkm_ctx_set_tuning
    ctx.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
seg = np.load(os.path.join(ROOT, "tests", "golden", "segment12.npz"))
base = [int(x) for x in seg["log_n"]]
ctl_tables, ctls = T.all_cross_table_lookups()
bad = 0
t0 = time.time()
nsegs = 0
for k in range(cases if stack else 0):
    K = int(rng.integers(2, stack + 1))
    segs = []
    for v in range(K):
        traces, log_n = [], []
        for i in range(12):
            w = T.WIDTH[T.TABLE_ENUM_ORDER[i]]
            up = int(rng.integers(0, 4)) if w < 1000 else int(rng.integers(0, 2))
            t = np.roll(seg["t%d" % i].reshape(w, -1), int(rng.integers(0, 8)) * (i + 1), axis=1)
            traces.append(np.ascontiguousarray(np.tile(t, (1, 1 << up))).reshape(-1))
            log_n.append(base[i] + up)
        segs.append((traces, log_n, [int(x) for x in rng.integers(0, 1 << 32, int(rng.integers(0, 9)))]))
    ctx.set_tuning("max_stack", int(rng.choice([2, 3, 5, 32])))
    got = ctx.prove_segments(segs)
    for v, (traces, log_n, pub) in enumerate(segs):
        tables = [(T.TABLE_ENUM_ORDER[i], traces[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
        ref, rchal, roffs = o.prove_with_traces(tables, ctls, public_values=pub)
        nsegs += 1
        if not (list(got[v][2]) == list(roffs) and bool((got[v][1] == rchal).all()) and got[v][0].size == ref.size and bool((got[v][0] == ref).all())):
            bad += 1
            print("MISMATCH case %d segment %d heights %r" % (k, v, log_n))
if stack:
    print("fuzz_segments (lock-step): %d calls, %d segments, %d mismatches, %.1f s (seed %d)" % (cases, nsegs, bad, time.time() - t0, seed))
    sys.exit(1 if bad else 0)
for k in range(cases):
    traces, log_n = [], []
    for i in range(12):
        w = T.WIDTH[T.TABLE_ENUM_ORDER[i]]
        up = int(rng.integers(0, 5)) if w < 1000 else int(rng.integers(0, 3))
        traces.append(np.ascontiguousarray(np.tile(seg["t%d" % i].reshape(w, -1), (1, 1 << up))).reshape(-1))
        log_n.append(base[i] + up)
    pub = [int(x) for x in rng.integers(0, 1 << 32, int(rng.integers(0, 9)))]
    got, chal, offs = ctx.prove_segment(traces, log_n, public_values=pub)
    tables = [(T.TABLE_ENUM_ORDER[i], traces[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
    ref, rchal, roffs = o.prove_with_traces(tables, ctls, public_values=pub)
    ok = list(offs) == list(roffs) and bool((chal == rchal).all()) and got.size == ref.size and bool((got == ref).all())
    if not ok:
        bad += 1
        print("MISMATCH case %d heights %r" % (k, log_n))
print("fuzz_segments: %d cases, %d mismatches, %.1f s (seed %d)" % (cases, bad, time.time() - t0, seed))
sys.exit(1 if bad else 0)
