#!/usr/bin/env python3
"""Ad-hoc GPU timing of the primitives with the library's own HIP-event profiler (development aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkm_amd as z

ctx = z.Context(0)
out = {}
rng = np.random.default_rng(0)
k = 1 << 22
st = rng.integers(0, z.P, 12 * k, dtype=np.uint64)
buf = ctx.alloc(st.size).upload(st)
ctx.poseidon_permute_batch(buf)
ctx.profile(True); ctx.profile_reset()
for _ in range(3): ctx.poseidon_permute_batch(buf)
r = ctx.profile_records(); n, ms = r["poseidon_permute"]
out["poseidon_permute_Gperm_s"] = k * n / ms / 1e6
buf.free()
kk = 1 << 22
st = rng.integers(0, 2**64, 25 * kk, dtype=np.uint64)
buf = ctx.alloc(st.size).upload(st)
ctx.keccakf_batch(buf); ctx.profile_reset()
for _ in range(3): ctx.keccakf_batch(buf)
n, ms = ctx.profile_records()["keccakf"]
out["keccakf_Gperm_s"] = kk * n / ms / 1e6
buf.free()
for log_n in (16, 18, 20):
    tr = ctx.poseidon_trace(1, 1 << log_n, log_n)
    ctx.profile_reset()
    t = time.time()
    b = z.PolynomialBatch.from_values(ctx, tr, 262, log_n)
    dt = time.time() - t
    out["commit_2^%d" % log_n] = {"wall_s": dt, "kernels": {k_: {"n": v[0], "ms": v[1]} for k_, v in ctx.profile_records().items()}}
    b.free(); tr.free()
print(json.dumps(out, indent=1))
