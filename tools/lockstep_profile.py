#!/usr/bin/env python3
"""Per-kernel HIP-event profile of ONE context proving K 2^16-cycle segments in lock-step on one stream (commit_lanes 1): kernel
milliseconds per SEGMENT next to the wall time -- where a lock-step group spends its time.   python tools/lockstep_profile.py [K] [key=value ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import zkm_amd  # noqa: E402
from tools.bench_segment import tiled_segment  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
c = zkm_amd.Context(0)
c.set_tuning("throughput_profile", 1)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    c.set_tuning(k, int(v))
bufs, logs = tiled_segment(c, 16)
segs = [(bufs, logs, [1, 2, 3, 0, j]) for j in range(K)]
c.prove_segments(segs)
c.synchronize()
reps = 2
t0 = time.perf_counter()
for _ in range(reps):
    c.prove_segments(segs)
c.synchronize()
wall = (time.perf_counter() - t0) / reps
c.profile(True)
c.profile_reset()
for _ in range(reps):
    c.prove_segments(segs)
c.synchronize()
rec = c.profile_records()
c.profile(False)
krec = {k: v for k, v in rec.items() if not k.startswith("stage/")}
ksum = sum(v[1] for v in krec.values()) / reps
out = {"segments_per_call": K, "ms_per_call": wall * 1e3, "ms_per_segment": wall * 1e3 / K, "kernel_ms_per_segment": ksum / K,
       "launches_per_call": sum(v[0] for v in krec.values()) / reps,
       "kernel_ms_per_segment_by_name": {k: round(v[1] / reps / K, 4) for k, v in sorted(krec.items(), key=lambda kv: -kv[1][1])[:30]},
       "launches_per_call_by_name": {k: round(v[0] / reps, 1) for k, v in sorted(krec.items(), key=lambda kv: -kv[1][0])[:16]},
       "stage_ms_per_call": {k[6:]: round(v[1] / reps, 3) for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1]) if k.startswith("stage/")}}
print(json.dumps(out, indent=1))
