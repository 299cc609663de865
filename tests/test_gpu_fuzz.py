"""Differential fuzz inside the GPU suite (the driver, not a text file, vouches for these seeds): commitments + zkm_prove_openings
against the oracle on random shapes around the library's size thresholds (sixteen seeds of tools/fuzz_openings.py's generator), and
lock-step groups of twelve-table segments at random heights against the single-segment path and the oracle.  Longer runs of the same
generators: profiles/r03_fuzz.txt, profiles/r04_fuzz.txt."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.mark.parametrize("seed", list(range(101, 117)))
def test_random_shapes_commit_and_open_like_the_oracle(ctx, zkm, oracle, seed):
    rng = np.random.default_rng(seed)
    for _ in range(7):
        log_n = int(rng.integers(5, 13))
        W = int(rng.choice([1, 3, 8, 9, 33, 127, 129, 262, int(rng.integers(1, 600))]))
        W = min(W, max(1, (1 << 20) >> log_n))
        A = int(rng.integers(1, 24))
        Z = int(rng.integers(1, min(A, 3) + 1))
        n = 1 << log_n
        tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (W, A, 4))
        tb, ab = zkm.PolynomialBatch.from_values(ctx, tv, W, log_n), zkm.PolynomialBatch.from_values(ctx, av, A, log_n)
        qb = zkm.PolynomialBatch.from_coeffs(ctx, qc, 4, log_n)
        otb, oab, oqb = oracle.batch_from_values(tv, W, log_n), oracle.batch_from_values(av, A, log_n), oracle.batch_from_coeffs(qc, 4, log_n)
        assert (tb.cap() == otb.cap()).all() and (ab.cap() == oab.cap()).all() and (qb.cap() == oqb.cap()).all(), (log_n, W, A)
        got, want = ctx.prove_openings(tb, ab, qb, Z), oracle.prove_openings(otb, oab, oqb, Z)
        assert got.size == want.size and (got == want).all(), (log_n, W, A, Z)
        for b in (tb, ab, qb):
            b.free()


@pytest.mark.parametrize("seed", [201, 202, 203])
def test_random_lockstep_groups_prove_like_single_segments(ctx, zkm, oracle, seed):
    """zkm_prove_segments on 2 .. 5 twelve-table segments, every table of every segment tiled to a random height (its own + 0 .. 2
    doublings), random public values, random max_stack: every blob == zkm_prove_segment's for that segment alone, and one segment of
    every case == the oracle's prove_with_traces."""
    import os
    from zkm_amd import tables as T
    rng = np.random.default_rng(seed)
    seg = np.load(os.path.join(os.path.dirname(__file__), "golden", "segment12.npz"))
    base = [int(x) for x in seg["log_n"]]
    ctl_tables, ctls = T.all_cross_table_lookups()
    K = int(rng.integers(2, 6))
    segs = []
    for v in range(K):
        traces, log_n = [], []
        for i in range(12):
            w = T.WIDTH[T.TABLE_ENUM_ORDER[i]]
            up = int(rng.integers(0, 3)) if w < 1000 else int(rng.integers(0, 2))
            t = np.roll(seg["t%d" % i].reshape(w, -1), v * (i + 1), axis=1)
            traces.append(np.ascontiguousarray(np.tile(t, (1, 1 << up))).reshape(-1))
            log_n.append(base[i] + up)
        segs.append((traces, log_n, [int(x) for x in rng.integers(0, 1 << 32, int(rng.integers(0, 6)))]))
    c2 = zkm.Context(0)
    try:
        c2.set_tuning("max_stack", int(rng.choice([2, 3, 32])))
        got = c2.prove_segments(segs)
    finally:
        c2.close()
    for v, (traces, log_n, pub) in enumerate(segs):
        want, wchal, woffs = ctx.prove_segment(traces, log_n, public_values=pub)
        assert list(got[v][2]) == list(woffs) and (got[v][1] == wchal).all() and (got[v][0] == want).all(), (seed, v, log_n)
    v = int(rng.integers(0, K))
    traces, log_n, pub = segs[v]
    tables = [(T.TABLE_ENUM_ORDER[i], traces[i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
    ref, rchal, _ = oracle.prove_with_traces(tables, ctls, public_values=pub)
    assert (got[v][0] == ref).all() and (got[v][1] == rchal).all()
