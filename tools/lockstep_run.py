#!/usr/bin/env python3
"""Plain runner for profilers: one context proves K 2^16-cycle segments in lock-step, WARM warm-up calls then REPS timed calls.
   python tools/lockstep_run.py K REPS [WARM] [key=value ...]      (ZKM_SEG_TUNING-style keys; throughput_profile=1 is the default)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import zkm_amd  # noqa: E402
from tools.bench_segment import tiled_segment  # noqa: E402

K, reps = int(sys.argv[1]), int(sys.argv[2])
warm = int(sys.argv[3]) if len(sys.argv) > 3 and "=" not in sys.argv[3] else 1
c = zkm_amd.Context(0)
c.set_tuning("throughput_profile", 1)
for kv in sys.argv[3:]:
    if "=" in kv:
        k, v = kv.split("=")
        c.set_tuning(k, int(v))
bufs, logs = tiled_segment(c, 16)
segs = [(bufs, logs, [1, 2, 3, 0, j]) for j in range(K)]
for _ in range(warm):
    c.prove_segments(segs)
c.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    c.prove_segments(segs)
c.synchronize()
print("lockstep_run: K %d, %d calls, %.2f ms per segment" % (K, reps, (time.perf_counter() - t0) * 1e3 / reps / K))
