"""CpuStark (cpu/cpu_stark.rs:260-285, 741 constraints): the oracle's restatement vanishes on rows produced the way the reference's
witness generators fill them (tests/cpu_fixtures.py: every instruction class the constraints mention), every constraint is live
(some single-cell corruption makes it nonzero), and the proof of the table verifies; the GPU kernel is bit-exact against it."""
import numpy as np
import pytest

from zkm_amd import tables as T
from . import cpu_fixtures as CF
from .test_gpu_tables import fake_ctl_aux

P = 0xFFFFFFFF00000001
LOG_N = 8


@pytest.fixture(scope="module")
def machine():
    return CF.sample_program(CF.Machine())


def test_cpu_constraints_vanish_on_generated_rows(oracle, machine):
    trace = machine.trace(LOG_N)
    count, bad = oracle.debug_constraints(T.TABLE_CPU, trace, 259, LOG_N)
    assert count == 741 and bad is None
    flags = trace.reshape(259, -1)[7:40].sum(axis=1)
    exercised = {n for n, c in zip(CF.OPS, flags) if c}
    assert exercised >= {"binary_op", "binary_imm_op", "logic_op", "movz_op", "movn_op", "clz_op", "clo_op", "shift", "shift_imm", "jumps",
                         "jumpi", "jumpdirect", "branch", "m_op_load", "m_op_store", "nop", "ext", "ins", "maddu", "rdhwr", "signext8",
                         "signext16", "swaphalf", "teq", "ror", "syscall"}


def test_every_cpu_constraint_is_live(oracle, machine):
    """Single-cell corruptions (+1 and +2^32) of the generated rows: every constraint except the channel-3..8 bootstrap address
    checks (the fixture boots through channels 0..2) and the last-row check becomes nonzero for at least one of them."""
    n = 1 << LOG_N
    rows = machine.trace(LOG_N).reshape(259, n).T.copy()
    live = np.zeros(741, dtype=bool)
    for r in range(len(machine.rows)):
        for c in range(259):
            old = rows[r, c]
            for delta in (1, 1 << 32):
                rows[r, c] = (int(old) + delta) % P
                live |= oracle.row_constraints(T.TABLE_CPU, rows[r], rows[(r + 1) % n], r == 0, r == n - 1) != 0
                if r:
                    live |= oracle.row_constraints(T.TABLE_CPU, rows[r - 1], rows[r], r == 1, False) != 0
            rows[r, c] = old
    dead = set(np.nonzero(~live)[0].tolist())
    assert dead <= {1} | set(range(9, 21)), sorted(dead)


def test_cpu_table_proof_verifies_and_rejects_corruption(oracle, machine):
    trace = machine.trace(LOG_N)
    aux = fake_ctl_aux(LOG_N)
    proof = oracle.prove(trace, LOG_N, aux, [2], ncols=259, table_id=T.TABLE_CPU)
    assert oracle.verify(proof, 3, [2], ncols=259, table_id=T.TABLE_CPU) == 0
    n = 1 << LOG_N
    t = trace.reshape(259, n)
    cases = []
    for name in ("branch", "m_op_load", "m_op_store", "syscall", "ext", "ins", "ror", "maddu", "clz_op", "jumps", "signext8"):
        r = int(np.nonzero(t[CF.OP[name]])[0][0])
        cases.append((name, CF.ch({"syscall": 4, "ins": 2, "jumps": 0}.get(name, 1), 5), r))          # a channel value the instruction constrains
    cases.append(("boot", CF.IS_BOOT, 0))
    for name, col, r in cases:
        bad = trace.copy()
        bad[col * n + r] = (int(bad[col * n + r]) + 1) % P
        p2 = oracle.prove(bad, LOG_N, aux, [2], ncols=259, table_id=T.TABLE_CPU)
        assert oracle.verify(p2, 3, [2], ncols=259, table_id=T.TABLE_CPU) != 0, name


@pytest.mark.gpu
def test_cpu_quotient_and_proof_are_bit_exact(ctx, zkm, oracle, machine):
    trace = machine.trace(LOG_N)
    aux = fake_ctl_aux(LOG_N)
    want = oracle.prove(trace, LOG_N, aux, [2], ncols=259, table_id=T.TABLE_CPU)
    got = ctx.prove_single_table(trace, LOG_N, aux, [2], ncols=259, table_id=T.TABLE_CPU)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify(got, 3, [2], ncols=259, table_id=T.TABLE_CPU) == 0
    # arbitrary (invalid) rows: every constraint is nonzero, so the quotient values check all 741 expressions at once
    rng = np.random.default_rng(9)
    log_n = 6
    rnd = rng.integers(0, P, 259 << log_n, dtype=np.uint64)
    aux = fake_ctl_aux(log_n)
    alphas = [int(x) for x in rng.integers(1, P, 2, dtype=np.uint64)]
    tb_o, ab_o = oracle.batch_from_values(rnd, 259, log_n), oracle.batch_from_values(aux, 3, log_n)
    want = oracle.quotient(tb_o, ab_o, [2], alphas, table_id=T.TABLE_CPU)
    tb, ab = zkm.PolynomialBatch.from_values(ctx, rnd, 259, log_n), zkm.PolynomialBatch.from_values(ctx, aux, 3, log_n)
    got = ctx.quotient(tb, ab, [2], alphas, table_id=T.TABLE_CPU)
    assert (np.asarray(got) == np.asarray(want)).all()
    for nal in (1,):
        want = oracle.quotient(tb_o, ab_o, [2], alphas[:nal], table_id=T.TABLE_CPU)
        got = ctx.quotient(tb, ab, [2], alphas[:nal], table_id=T.TABLE_CPU)
        assert (np.asarray(got) == np.asarray(want)).all()


@pytest.mark.gpu
def test_cpu_table_large_proof_verifies(ctx, oracle):
    """2^14 rows (the sample program repeated): GPU proof accepted by the oracle's verifier."""
    m = CF.Machine()
    for _ in range(60):
        CF.sample_program(m)
    log_n = 14
    trace = m.trace(log_n)
    assert oracle.debug_constraints(T.TABLE_CPU, trace, 259, log_n)[1] is None
    aux = fake_ctl_aux(log_n)
    got = ctx.prove_single_table(trace, log_n, aux, [2], ncols=259, table_id=T.TABLE_CPU)
    assert oracle.verify(got, 3, [2], ncols=259, table_id=T.TABLE_CPU) == 0


def test_cpu_segment_cross_table_lookups(oracle):
    """CPU + Memory + Logic + Arithmetic: ctl_arithmetic, the CPU looker of ctl_logic and the nine CPU memory channels hold on
    the generated segment, the oracle's proofs verify, and a changed register value breaks only the cross-table check."""
    tables, ctls, m = CF.build_cpu_segment(oracle)
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0
    n = 1 << tables[0][3]
    cpu = tables[0][1].copy()
    r = int(np.nonzero(cpu.reshape(259, n)[CF.OP["binary_op"]])[0][0])
    cpu[CF.ch(2, 5) * n + r] += 1                     # ADDU result: unconstrained inside the CPU table, pinned by the lookups
    assert oracle.debug_constraints(T.TABLE_CPU, cpu, 259, tables[0][3])[1] is None
    bad = [(T.TABLE_CPU, cpu) + tables[0][2:]] + tables[1:]
    assert oracle.check_ctls(bad, ctls) != 0


@pytest.mark.gpu
def test_cpu_segment_proofs_are_bit_exact(ctx, oracle):
    tables, ctls, m = CF.build_cpu_segment(oracle)
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
    got, chal, offs = ctx.prove_with_traces(tables, ctls)
    assert offs == woffs and (chal == wchal).all()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal) == 0


def test_full_segment_all_fifteen_lookups(oracle):
    """Twelve tables, the fifteen lookups of all_stark::all_cross_table_lookups() with every looking table the reference lists:
    consistent (check_ctls), proved and verified by the oracle; a wrong digest word in a CPU precompile row breaks the lookup."""
    tables, ctls = CF.build_full_segment(oracle)
    assert [t[0] for t in tables] == [T.TABLE_ARITHMETIC, T.TABLE_CPU, T.TABLE_POSEIDON, T.TABLE_POSEIDON_SPONGE, T.TABLE_KECCAK,
                                      T.TABLE_KECCAK_SPONGE, T.TABLE_SHA_EXTEND, T.TABLE_SHA_EXTEND_SPONGE, T.TABLE_SHA_COMPRESS,
                                      T.TABLE_SHA_COMPRESS_SPONGE, T.TABLE_LOGIC, T.TABLE_MEMORY]
    assert len(ctls) == 15 and sum(len(looking) for looking, _ in ctls) == 2 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + (1 + 34 + 4 + 12) + \
        (9 + 32 + 136 + 16 + 32 + 4)
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    assert oracle.verify_all(tables, ctls, proofs, chal, public_values=[1, 2, 3]) == 0
    # auxiliary-column counts of the real AllStark (Arithmetic 22, Cpu 26, Poseidon 4, PoseidonSponge 40, Keccak 4,
    # KeccakSponge 180, Logic 2, Memory 6; the SHA tables follow from the same rules)
    assert [int(proofs[offs[i] + 3]) for i in range(12)] == [22, 26, 4, 40, 4, 180, 10, 24, 24, 40, 2, 6]
    n = 1 << tables[1][3]
    cpu = tables[1][1].copy()
    for flag in (T.CPU_IS_KECCAK_SPONGE, T.CPU_IS_POSEIDON_SPONGE, T.CPU_IS_SHA_EXTEND_SPONGE, T.CPU_IS_SHA_COMPRESS_SPONGE):
        r = int(np.nonzero(cpu.reshape(259, n)[flag])[0][0])
        bad = cpu.copy()
        bad[T.CPU_GENERAL * n + r] += 1
        assert oracle.check_ctls([tables[0], (T.TABLE_CPU, bad) + tables[1][2:]] + tables[2:], ctls) != 0


@pytest.mark.gpu
def test_full_segment_proofs_are_bit_exact(ctx, zkm, oracle):
    """prove_with_traces on all twelve tables with all fifteen lookups: GPU == oracle word for word, verify_proof accepts; the same
    through a ZKMTRACE image."""
    tables, ctls = CF.build_full_segment(oracle)
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    got, chal, offs = ctx.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    assert offs == woffs and (chal == wchal).all()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal, public_values=[1, 2, 3]) == 0
    img = zkm.segment_image(tables, ctls, public_values=[1, 2, 3])
    got2, chal2, offs2 = ctx.prove_segment_image(img)
    assert offs2 == offs and (chal2 == chal).all() and (got2 == got).all()
