#!/usr/bin/env python3
"""Development aid: wall time and per-kernel HIP-event profile of zkm_prove_with_traces on the twelve-table test segment
(tests/cpu_fixtures.build_full_segment) and of a 2^LOG-row CPU-table proof.  Uses the oracle only to build the fixture."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zkm_amd as z
import zkm_amd.tables
from oracle.oracle_py import Oracle
from tests import cpu_fixtures as CF
from tests.test_gpu_tables import fake_ctl_aux

ctx = z.Context(0)
o = Oracle()
out = {}
tables, ctls = CF.build_full_segment(o)
ctx.prove_with_traces(tables, ctls)
ctx.profile(True); ctx.profile_reset()
t = time.time()
ctx.prove_with_traces(tables, ctls)
out["segment12_wall_s"] = time.time() - t
out["segment12_kernels"] = {k: {"n": v[0], "ms": round(v[1], 3)} for k, v in sorted(ctx.profile_records().items(), key=lambda kv: -kv[1][1])[:14]}
# the same twelve tables tiled to the sizes of a 2^20-cycle segment (no longer a valid witness -- timing only)
SIZES = {z.tables.TABLE_CPU: 20, z.tables.TABLE_MEMORY: 21, z.tables.TABLE_ARITHMETIC: 19, z.tables.TABLE_LOGIC: 17, z.tables.TABLE_KECCAK: 15,
         z.tables.TABLE_KECCAK_SPONGE: 12, z.tables.TABLE_POSEIDON: 14, z.tables.TABLE_POSEIDON_SPONGE: 14, z.tables.TABLE_SHA_EXTEND: 16,
         z.tables.TABLE_SHA_EXTEND_SPONGE: 16, z.tables.TABLE_SHA_COMPRESS: 16, z.tables.TABLE_SHA_COMPRESS_SPONGE: 10}
big = []
for tid, tr, w, ln, ct in tables:
    L = max(SIZES[tid], ln)
    t2 = np.ascontiguousarray(np.tile(np.asarray(tr).reshape(w, -1), (1, 1 << (L - ln)))).reshape(-1)
    big.append((tid, ctx.alloc(t2.size).upload(t2), w, L, ct))
ctx.prove_with_traces(big, ctls)
ctx.profile_reset()
t = time.time()
ctx.prove_with_traces(big, ctls)
out["segment12_2^20_wall_s"] = time.time() - t
out["segment12_2^20_kernels"] = {k: {"n": v[0], "ms": round(v[1], 3)} for k, v in sorted(ctx.profile_records().items(), key=lambda kv: -kv[1][1])[:24]}
out["segment12_2^20_memory"] = ctx.memory()
for b in big:
    b[1].free()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
m = CF.Machine()
while len(m.rows) + 260 < (1 << 14):
    CF.sample_program(m)
small = m.trace(14).reshape(259, -1)
trace = np.ascontiguousarray(np.tile(small, (1, 1 << (log_n - 14)))).reshape(-1)   # not a valid witness across the seams; timing only
aux = fake_ctl_aux(log_n)
d = ctx.alloc(trace.size).upload(trace)
ctx.prove_single_table(d, log_n, aux, [2], ncols=259, table_id=11)
ctx.profile_reset()
t = time.time()
ctx.prove_single_table(d, log_n, aux, [2], ncols=259, table_id=11)
out["cpu_table_2^%d_wall_s" % log_n] = time.time() - t
out["cpu_table_kernels"] = {k: {"n": v[0], "ms": round(v[1], 3)} for k, v in sorted(ctx.profile_records().items(), key=lambda kv: -kv[1][1])[:10]}
print(json.dumps(out, indent=1))
