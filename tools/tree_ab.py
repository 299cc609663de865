#!/usr/bin/env python3
"""tools/tree_ab.py -- A/B of a tuning key on the commitment of 262 x 2^20 and on config 4 (prove_openings at 2^22): per-kernel means from
the library's own event records, both settings alternating in one process.  usage: tree_ab.py key [rounds]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkm_amd
key = sys.argv[1] if len(sys.argv) > 1 else "leaf_mfma"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
log_n = 20
ctx = zkm_amd.Context(0)
trace = ctx.poseidon_trace(seed=2, num_perms=(1 << log_n) - 3, log_n=log_n)
caps = {}
for r in range(rounds):
    for val in (0, 1):
        ctx.set_tuning(key, val)
        b = zkm_amd.PolynomialBatch.from_values(ctx, trace, 262, log_n, 2, 4)
        caps[val] = b.cap().copy()
        b.free()
        ctx.profile(True)
        ctx.profile_reset()
        t0 = time.time()
        for _ in range(3):
            b = zkm_amd.PolynomialBatch.from_values(ctx, trace, 262, log_n, 2, 4)
            b.free()
        ctx.synchronize()
        wall = (time.time() - t0) / 3
        recs = ctx.profile_records()
        ctx.profile(False)
        print("%s=%d commit %.2f ms; " % (key, val, wall * 1e3) + "; ".join("%s %.3f" % (k, v[1] / 3) for k, v in sorted(recs.items(), key=lambda kv: -kv[1][1])[:5]), flush=True)
assert (caps[0] == caps[1]).all(), "caps differ between the two settings"
trace.free()
ctx.trim()
# config 4
rng = np.random.default_rng(5)
P = 0xFFFFFFFF00000001
log_n = 22
n = 1 << log_n
W, A, Q, Z = 13, 4, 4, 2
tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (W, A, Q))
tb, ab = zkm_amd.PolynomialBatch.from_values(ctx, tv, W, log_n), zkm_amd.PolynomialBatch.from_values(ctx, av, A, log_n)
qb = zkm_amd.PolynomialBatch.from_coeffs(ctx, qc, Q, log_n)
blobs = {}
for r in range(rounds):
    for val in (0, 1):
        ctx.set_tuning(key, val)
        blobs[val] = ctx.prove_openings(tb, ab, qb, Z)
        ctx.profile(True)
        ctx.profile_reset()
        t0 = time.time()
        for _ in range(5):
            ctx.prove_openings(tb, ab, qb, Z)
        ctx.synchronize()
        wall = (time.time() - t0) / 5
        recs = ctx.profile_records()
        ctx.profile(False)
        print("%s=%d prove_openings 2^22 %.3f ms; " % (key, val, wall * 1e3) + "; ".join("%s %.3f" % (k, v[1] / 5) for k, v in sorted(recs.items(), key=lambda kv: -kv[1][1])[:6] if not k.startswith("stage/")), flush=True)
assert (blobs[0] == blobs[1]).all(), "proofs differ between the two settings"
print("bit-exact both ways")
