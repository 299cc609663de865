#!/usr/bin/env python3
"""Ad-hoc timing of a full prove_single_table with the per-kernel HIP-event breakdown (development aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkm_amd as z
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = z.Context(0)
n = 1 << log_n
trace = ctx.poseidon_trace(2, n, log_n)
aux = ctx.alloc(4 * n).upload(np.zeros(4 * n, dtype=np.uint64))
ctx.prove_single_table(trace, log_n, aux, [1, 1])  # warm-up
ts = []
for _ in range(3):
    t = time.time(); ctx.prove_single_table(trace, log_n, aux, [1, 1]); ts.append(time.time() - t)
ctx.profile(True); ctx.profile_reset()
t = time.time(); ctx.prove_single_table(trace, log_n, aux, [1, 1]); tp = time.time() - t
rec = ctx.profile_records()
print(json.dumps({"log_n": log_n, "wall_s": ts, "profiled_wall_s": tp, "kernel_ms_total": sum(v[1] for v in rec.values()),
                  "kernels": {k: {"n": v[0], "ms": round(v[1], 3)} for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1])}}, indent=1))
