#!/usr/bin/env python3
"""Differential fuzz of the commitment + openings + FRI path against the CPU oracle on random shapes (test infrastructure: the oracle
is the checker).  Random column counts (1 .. 700), heights 2^5 .. 2^13, auxiliary widths and numbers of CTL Zs: every size threshold
of the library sits inside this box (16-lane / four-lane / one-lane hashing, sliced FRI combination, division scan levels, FRI layer
counts).  Every case: three commitments (caps), then zkm_prove_openings == oracle.prove_openings, word for word.

  python tools/fuzz_openings.py [cases=150] [seed=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import zkm_amd  # noqa: E402
from oracle import oracle_py  # noqa: E402

P = 0xFFFFFFFF00000001
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
o = oracle_py.Oracle()
o.set_threads(min(32, os.cpu_count() or 1))
ctx = zkm_amd.Context(0)
t0 = time.time()
bad = 0
for k in range(cases):
    log_n = int(rng.integers(5, 14))
    W = int(rng.choice([1, 2, 3, 5, 8, 9, 31, 32, 33, 64, 127, 128, 129, 262, 470, int(rng.integers(1, 700))]))
    if (W << log_n) > (1 << 21):
        W = max(1, (1 << 21) >> log_n)
    A = int(rng.integers(1, 40))
    Z = int(rng.integers(1, min(A, 4) + 1))
    n = 1 << log_n
    tv = rng.integers(0, P, W * n, dtype=np.uint64)
    av = rng.integers(0, P, A * n, dtype=np.uint64)
    qc = rng.integers(0, P, 4 * n, dtype=np.uint64)
    tb, ab = zkm_amd.PolynomialBatch.from_values(ctx, tv, W, log_n), zkm_amd.PolynomialBatch.from_values(ctx, av, A, log_n)
    qb = zkm_amd.PolynomialBatch.from_coeffs(ctx, qc, 4, log_n)
    otb, oab, oqb = o.batch_from_values(tv, W, log_n), o.batch_from_values(av, A, log_n), o.batch_from_coeffs(qc, 4, log_n)
    ok = (tb.cap() == otb.cap()).all() and (ab.cap() == oab.cap()).all() and (qb.cap() == oqb.cap()).all()
    if ok:
        got = ctx.prove_openings(tb, ab, qb, Z)
        want = o.prove_openings(otb, oab, oqb, Z)
        ok = got.size == want.size and bool((got == want).all())
    if not ok:
        bad += 1
        print("MISMATCH case %d: log_n %d W %d A %d Z %d" % (k, log_n, W, A, Z))
    for b in (tb, ab, qb):
        b.free()
print("fuzz_openings: %d cases, %d mismatches, %.1f s (seed %d)" % (cases, bad, time.time() - t0, seed))
sys.exit(1 if bad else 0)
