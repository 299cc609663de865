#!/usr/bin/env python3
"""2^20-cycle twelve-table segments (table heights of tools/bench_segment.py HEIGHTS[20]) through zkm_prove_segments, K = 1, 2, 4 per call, one context."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import zkm_amd  # noqa: E402
from tools.bench_segment import tiled_segment  # noqa: E402
c = zkm_amd.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    c.set_tuning(k, int(v))
bufs, logs = tiled_segment(c, 20)
for K in (1, 2, 4):
    segs = [(bufs, logs, [1, 2, 3, j]) for j in range(K)]
    c.prove_segments(segs)
    c.synchronize()
    t0 = time.perf_counter()
    c.prove_segments(segs)
    c.synchronize()
    dt = time.perf_counter() - t0
    print("2^20-cycle segments, %d per call: %.1f ms per segment; allocator live / cached GB %s" % (K, dt * 1e3 / K, [round(x / 2**30, 1) for x in c.memory()]))
