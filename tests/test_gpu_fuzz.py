"""A slice of tools/fuzz_openings.py inside the GPU suite: commitments + zkm_prove_openings against the oracle on random shapes around
the library's size thresholds (the long runs are recorded in profiles/r03_fuzz.txt)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.mark.parametrize("seed", [101, 102])
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
