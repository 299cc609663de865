#!/usr/bin/env python3
"""tools/leaf_only.py -- one trace commitment (262 x 2^LOG_N: NTT + LDE + k_merkle_leaves + tree) for counter passes:
   rocprofv3 --pmc <counters> -- python tools/leaf_only.py [log_n] [repeats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkm_amd

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = zkm_amd.Context(0)
trace = ctx.poseidon_trace(seed=2, num_perms=(1 << log_n) - 3, log_n=log_n)
for _ in range(reps):
    b = zkm_amd.PolynomialBatch.from_values(ctx, trace, 262, log_n, 2, 4)
    b.free()
ctx.synchronize()
print("done")
