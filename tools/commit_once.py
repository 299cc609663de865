#!/usr/bin/env python3
"""One from_values commitment of a 262 x 2^20 PoseidonStark trace (warm-up + REPS timed), for profilers: python tools/commit_once.py [reps=2] [log_n=20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkm_amd  # noqa: E402
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
c = zkm_amd.Context(0)
t = c.poseidon_trace(100, 1 << log_n, log_n)
b = zkm_amd.PolynomialBatch.from_values(c, t, 262, log_n)
b.free()
c.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    b = zkm_amd.PolynomialBatch.from_values(c, t, 262, log_n)
    c.synchronize()
    b.free()
print("commit_once: %.2f ms per commitment" % ((time.perf_counter() - t0) * 1e3 / reps))
