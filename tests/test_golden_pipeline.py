"""Pipeline golden vectors (tests/golden/pipeline_golden.json): the oracle must keep reproducing them (CPU), and
the HIP path must reproduce them WITHOUT the oracle in the loop (GPU)."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.json")))
P = 0xFFFFFFFF00000001


def h(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def seeded(seed, n):
    return np.random.default_rng(seed).integers(0, P, n, dtype=np.uint64)


def fake_aux(log_n, seed=3):
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    hh = [[int(x) for x in rng.integers(0, 1 << 62, n)] for _ in range(2)]
    z, acc = [0] * n, 0
    for i in range(n - 1, -1, -1):
        acc = (acc + hh[0][i] + hh[1][i]) % P
        z[i] = acc
    return np.array(hh[0] + hh[1] + z, dtype=np.uint64)


def table_cases(gen):
    rng = np.random.default_rng(9)
    ops = np.stack([rng.integers(0, 4, 50), rng.integers(0, 1 << 32, 50), rng.integers(0, 1 << 32, 50)], axis=1)
    yield 1, gen.logic_trace(ops, 6)
    yield 3, gen.keccak_trace(rng.integers(0, 1 << 64, (2, 25), dtype=np.uint64), [5, 9], 6)
    yield 6, gen.sha_extend_trace(rng.integers(0, 256, (20, 16), dtype=np.uint8), np.arange(20) + 1, 5)
    yield 8, gen.sha_compress_trace(rng.integers(0, 1 << 32, (1, 8), dtype=np.uint64), rng.integers(0, 1 << 32, (1, 64), dtype=np.uint64),
                                    np.array([[0, 0, 64, 9, 512, 0, 0, 0]], dtype=np.uint64), 7)


def cpu_trace(log_n):
    from . import cpu_fixtures as CF
    return CF.sample_program(CF.Machine()).trace(log_n)


def test_oracle_reproduces_goldens(oracle):
    for g in GOLD["ntt"]:
        r = oracle.ntt(seeded(g["seed"], g["ncols"] << g["log_n"]), g["log_n"], inverse=g["inverse"], coset_shift=g["coset_shift"])
        assert h(r) == g["sha256"] and [int(x) for x in r[:4]] == g["first"]
    for g in GOLD["commit"]:
        b = oracle.batch_from_values(seeded(g["seed"], g["ncols"] << g["log_n"]), g["ncols"], g["log_n"])
        assert h(b.coeffs()) == g["coeffs_sha256"] and [int(x) for x in b.cap().reshape(-1)] == g["cap"]
        assert h(b.lde_row(3)) == g["lde_row_3_sha256"] and h(b.merkle_path(5)) == g["merkle_path_5_sha256"]
    g = GOLD["proof"]
    proof = oracle.prove(oracle.poseidon_trace(g["seed"], g["num_perms"], g["log_n"]), g["log_n"], np.zeros(4 << g["log_n"], dtype=np.uint64), [1, 1])
    assert proof.size == g["words"] and h(proof) == g["sha256"]
    for (tid, tr), g in zip(table_cases(oracle), GOLD["tables"]):
        assert tid == g["table_id"] and h(tr) == g["trace_sha256"]
        assert h(oracle.prove(tr, g["log_n"], fake_aux(g["log_n"]), [2], ncols=g["ncols"], table_id=tid)) == g["proof_sha256"]
    g = GOLD["cpu"]
    tr = cpu_trace(g["log_n"])
    assert h(tr) == g["trace_sha256"]
    assert h(oracle.prove(tr, g["log_n"], fake_aux(g["log_n"]), [2], ncols=259, table_id=11)) == g["proof_sha256"]


@pytest.mark.gpu
def test_hip_path_reproduces_goldens_without_the_oracle(ctx, zkm):
    for g in GOLD["ntt"]:
        r = ctx.ntt(seeded(g["seed"], g["ncols"] << g["log_n"]), g["ncols"], g["log_n"], inverse=g["inverse"], coset_shift=g["coset_shift"])
        assert h(r) == g["sha256"] and [int(x) for x in r[:4]] == g["first"]
    for g in GOLD["commit"]:
        b = zkm.PolynomialBatch.from_values(ctx, seeded(g["seed"], g["ncols"] << g["log_n"]), g["ncols"], g["log_n"])
        assert h(b.coeffs()) == g["coeffs_sha256"] and [int(x) for x in b.cap().reshape(-1)] == g["cap"]
        assert h(b.lde_row(3)) == g["lde_row_3_sha256"] and h(b.merkle_path(5)) == g["merkle_path_5_sha256"]
    g = GOLD["proof"]
    trace = ctx.poseidon_trace(g["seed"], g["num_perms"], g["log_n"])
    proof = ctx.prove_single_table(trace, g["log_n"], np.zeros(4 << g["log_n"], dtype=np.uint64), [1, 1])
    assert proof.size == g["words"] and h(proof) == g["sha256"]
    C = 64
    assert [int(x) for x in proof[28:28 + C]] == g["trace_cap"] and [int(x) for x in proof[28 + 2 * C:28 + 3 * C]] == g["quotient_cap"]

    class Gen:  # the GPU witness generators with the oracle's call shape
        def __getattr__(self, name):
            return lambda *a: getattr(ctx, name)(*a).download()
    for (tid, tr), g in zip(table_cases(Gen()), GOLD["tables"]):
        assert h(tr) == g["trace_sha256"]
        assert h(ctx.prove_single_table(tr, g["log_n"], fake_aux(g["log_n"]), [2], ncols=g["ncols"], table_id=tid)) == g["proof_sha256"]
    g = GOLD["cpu"]
    tr = cpu_trace(g["log_n"])
    assert h(tr) == g["trace_sha256"]
    assert h(ctx.prove_single_table(tr, g["log_n"], fake_aux(g["log_n"]), [2], ncols=259, table_id=11)) == g["proof_sha256"]
