#!/usr/bin/env python3
"""tools/leaf_time.py -- time of trace commitments (262 x 2^20: NTT + LDE + k_merkle_leaves + tree) with the library's own per-kernel
records: prints the leaf kernel's mean duration.  ZKM_HIP_LIB selects an A/B build."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkm_amd
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = zkm_amd.Context(0)
if os.environ.get('LEAF_MFMA'): ctx.set_tuning('leaf_mfma', int(os.environ['LEAF_MFMA']))
trace = ctx.poseidon_trace(seed=2, num_perms=(1 << log_n) - 3, log_n=log_n)
b = zkm_amd.PolynomialBatch.from_values(ctx, trace, 262, log_n, 2, 4); cap = b.cap().copy(); b.free()
ctx.profile(True); ctx.profile_reset()
t0 = time.time()
for _ in range(3):
    b = zkm_amd.PolynomialBatch.from_values(ctx, trace, 262, log_n, 2, 4); b.free()
ctx.synchronize()
wall = (time.time() - t0) / 3
recs = ctx.profile_records()   # name -> (launches, total ms)
top = sorted(recs.items(), key=lambda kv: -kv[1][1])[:4]
print(os.environ.get("ZKM_HIP_LIB", "default"), "mfma=" + os.environ.get("LEAF_MFMA", "-"), "commit %.2f ms;" % (wall * 1e3), "; ".join("%s %.3f ms x%d" % (k, v[1] / max(v[0], 1), v[0]) for k, v in top), "cap0 %016x" % int(cap.reshape(-1)[0]))
