"""GPU parity: cross-table lookup data (K6), CTL constraint checks in the quotient kernel and the multi-table
driver prove_with_traces, all bit-exact against the CPU oracle."""
import numpy as np
import pytest

from tests.ctl_fixtures import build, colsets
from zkm_amd.ctl import CtlTable, make_zs

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def rich_table():
    t = CtlTable()
    a, m = colsets(t, "a"), colsets(t, "m")
    # a column set with next-row terms (Column::linear_combination_and_next_row_with_constant).  Its filter must be 0
    # on the last row: eval_table treats the missing next row as 0 while the constraints wrap around
    # (cross_table_lookup.rs:274-282 vs prover.rs:704) -- the FILTER column is 0 on the padding rows at the end.
    first = t.column(local=[(1, 3)], next=[(2, 5), (13, 1)], constant=9)
    t.column(next=[(1, 1)])
    f = t.single(0)
    nx = t.colset(range(first, first + 2), filter_constants=[f])
    return t, a, m, nx


@pytest.mark.parametrize("log_n,k", [(5, 20), (9, 400), (12, 4000)])
def test_ctl_data_matches_oracle(ctx, oracle, log_n, k):
    trace = oracle.poseidon_trace(6, k, log_n)
    t, a, m, nx = rich_table()
    zs, ids = make_zs([([a, m], 0x1234567890ABCDEF % P, 0x0FEDCBA987654321 % P), ([m], 7, 11), ([nx, a, m], 13, 17), ([nx], 19, 23)])
    got = ctx.ctl_data(t, zs, ids, trace, 262, log_n)
    want = oracle.ctl_data(t, zs, ids, trace, 262, log_n)
    assert (got == want).all()
    # device-resident trace and output
    tb = ctx.alloc(trace.size).upload(trace)
    out = ctx.alloc(got.size)
    ctx.ctl_data(t, zs, ids, tb, 262, log_n, out=out)
    assert (out.download() == want).all()
    tb.free()
    out.free()


def test_non_binary_filter_is_an_error(ctx, zkm, oracle):
    log_n = 5
    trace = oracle.poseidon_trace(6, 20, log_n)
    t = CtlTable()
    bad = t.singles_set([1, 2], filter_col=3)  # an input column is not 0/1
    zs, ids = make_zs([([bad], 3, 5)])
    with pytest.raises(zkm.ZkmError, match="Non-binary filter"):
        ctx.ctl_data(t, zs, ids, trace, 262, log_n)


def test_malformed_description_is_rejected(ctx, zkm, oracle):
    trace = oracle.poseidon_trace(6, 20, 5)
    t = CtlTable()
    a = colsets(t, "a")
    zs, ids = make_zs([([a, a], 3, 5, 0)])  # two column sets need one helper column
    with pytest.raises(zkm.ZkmError):
        ctx.ctl_data(t, zs, ids, trace, 262, 5)
    zs, ids = make_zs([([a + 5], 3, 5)])  # column-set index out of range
    with pytest.raises(zkm.ZkmError):
        ctx.ctl_data(t, zs, ids, trace, 262, 5)


@pytest.mark.parametrize("log_n", [5, 8])
def test_single_table_proof_with_real_ctl_data_is_bit_exact(ctx, oracle, log_n):
    n = 1 << log_n
    trace = oracle.poseidon_trace(4, n - 7, log_n)
    t, a, m, nx = rich_table()
    zs, ids = make_zs([([a, m, m], 3, 5), ([a], 7, 11), ([m, nx], 13, 17), ([nx], 29, 31)])
    aux = ctx.ctl_data(t, zs, ids, trace, 262, log_n)
    want = oracle.prove_ctl(trace, log_n, aux, t, zs, ids)
    got = ctx.prove_single_table_ctl(trace, log_n, aux, t, zs, ids)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_ctl(got, aux.size >> log_n, t, zs, ids) == 0


def test_prove_with_traces_is_bit_exact_and_verifies(ctx, oracle):
    tables, ctls = build(oracle)
    pub = [9, 8, 7]
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls, public_values=pub)
    got, chal, offs = ctx.prove_with_traces(tables, ctls, public_values=pub)
    assert offs == woffs and (chal == wchal).all()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal, public_values=pub) == 0


def test_prove_with_traces_larger_tables_verify(ctx, oracle):
    # 2^12 / 2^13-row tables: GPU proofs accepted by the oracle's verify_proof (incl. the cross-table sums)
    tables, ctls = build(oracle, log_small=12, k0=3000, k1=1000)
    proofs, chal, offs = ctx.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0
    bad = proofs.copy()
    bad[offs[3] + 500] ^= 1
    assert oracle.verify_all(tables, ctls, bad, chal) != 0


def test_inconsistent_tables_fail_only_the_cross_table_check(ctx, oracle):
    tables, ctls = build(oracle)
    tid, tr, w, log_n, ct = tables[2]
    n = 1 << log_n
    cols = tr.reshape(262, n).copy()
    cols[0][0] = 0
    tables2 = list(tables)
    tables2[2] = (tid, np.ascontiguousarray(cols).reshape(-1), w, log_n, ct)
    proofs, chal, offs = ctx.prove_with_traces(tables2, ctls)
    rc = oracle.verify_all(tables2, ctls, proofs, chal)
    assert 50 <= rc < 60, rc


@pytest.mark.parametrize("nlook", [1, 2, 5])
def test_lookup_helper_columns_match_oracle(ctx, oracle, nlook):
    # logUp helper columns (lookup.rs:46-124): parity on arbitrary looking / table / frequency columns
    log_n = 10
    trace = oracle.poseidon_trace(8, 700, log_n)
    t = CtlTable()
    sets = []
    for i in range(nlook):
        c = t.column(local=[(1 + i, 1), (14 + i, 3)], constant=i)
        f = t.single(0) if i % 2 == 0 else None
        sets.append(t.colset([c], filter_constants=[f] if f is not None else None))
    table_col = t.column(local=[(30, 1)], next=[(31, 2)])
    freq_col = t.single(40)
    ch = 0xABCDEF0123456789 % P
    got = ctx.lookup_helper_columns(t, sets, table_col, freq_col, ch, trace, 262, log_n)
    want = oracle.lookup_helper_columns(t, sets, table_col, freq_col, ch, trace, 262, log_n)
    assert got.size == ((nlook + 1) // 2 + 1) << log_n
    assert (got == want).all()


def test_error_in_a_table_proof_while_lanes_build_later_commitments(ctx, zkm, oracle):
    """A segment of short tables proves table t while the commit lanes are still building the auxiliary commitments of the tables after
    it (zkm_ctx_set_tuning "aux_pipeline").  A table that takes part in no lookup fails in its turn with the reference's message
    (prover.rs:509); the lanes are stopped and joined before anything they refer to goes away, nothing leaks, and the same context
    proves the complete instance afterwards, word for word."""
    tables, ctls = build(oracle)
    ctx.prove_with_traces(tables, ctls)              # (tables and caches of the session's context are in place)
    live0, _ = ctx.memory()
    for bad_ctls in ([ctls[0]], [ctls[1]]):          # table 3 / tables 1 and 2 without a lookup
        with pytest.raises(zkm.ZkmError, match="No CTL"):
            ctx.prove_with_traces(tables, bad_ctls)
    live, cached = ctx.memory()
    assert live == live0
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
    for pipelined in (1, 0):
        ctx.set_tuning("aux_pipeline", pipelined)
        got, chal, offs = ctx.prove_with_traces(tables, ctls)
        assert offs == woffs and (chal == wchal).all() and (got == want).all()
    ctx.set_tuning("aux_pipeline", 1)


def test_check_constraints_entry_point(ctx, zkm, oracle):
    """zkm_check_constraints == check_constraints (prover.rs:793-910): the whole vanishing polynomial (table constraints, CTL checks) on
    every row of the trace domain.  A valid table with its real CTL data passes; a corrupted trace cell fails at the row the oracle's
    constraint debugger names; a corrupted Z column fails the CTL checks although the table constraints hold."""
    log_n = 8
    n = 1 << log_n
    trace = oracle.poseidon_trace(4, n - 7, log_n)
    t, a, m, nx = rich_table()
    zs, ids = make_zs([([a, m, m], 3, 5), ([a], 7, 11), ([m, nx], 13, 17), ([nx], 29, 31)])
    aux = ctx.ctl_data(t, zs, ids, trace, 262, log_n)
    alphas = [0x1234567 % P, 0xFEDCBA9876543210 % P]
    assert ctx.check_constraints(trace, log_n, aux, t, zs, ids, alphas) is None
    assert ctx.check_constraints(trace, log_n, aux, t, zs, ids, alphas[:1]) is None
    # a corrupted trace cell: the first failing row is the oracle debugger's
    bad = trace.copy()
    r = 37
    bad.reshape(262, n)[20][r] ^= 1
    count, where = oracle.debug_constraints(0, bad, 262, log_n)
    assert where is not None
    got = ctx.check_constraints(bad, log_n, aux, t, zs, ids, alphas)
    assert got is not None and got <= where[0] and got >= r - 1       # (the CTL checks read the same cell and may fire one row earlier)
    # a corrupted running sum: only the cross-table checks see it
    aux2 = aux.copy()
    naux = aux.size >> log_n
    aux2.reshape(naux, n)[naux - 1][100] = (int(aux2.reshape(naux, n)[naux - 1][100]) + 1) % P
    got = ctx.check_constraints(trace, log_n, aux2, t, zs, ids, alphas)
    assert got in (99, 100)
    # the benchmark's fake CTL shape (helper columns, no column sets; poseidon_stark.rs:786-799) on the GPU witness
    tr = ctx.poseidon_trace(seed=3, num_perms=n - 2, log_n=log_n)
    assert ctx.check_constraints(tr, log_n, np.zeros(4 * n, dtype=np.uint64), None, [1, 1], None, alphas) is None
    tr.free()


def test_check_constraints_on_a_table_with_lookups(ctx, zkm, oracle):
    """MemoryStark: its range-check lookup columns (memory_stark.rs:476-483: RANGE_CHECK looked up in COUNTER with FREQUENCIES) come
    first among the auxiliary columns, per challenge helper column then Z (prover.rs:475-508)."""
    from zkm_amd import tables as T
    from zkm_amd.ctl import make_zs
    from .test_oracle_tables import random_memory_ops
    log_n = 8
    n = 1 << log_n
    trace, natural = oracle.memory_trace(random_memory_ops(log_n, 200), log_n)
    assert natural == n
    t = CtlTable()
    cs = T.memory_ctl_data(t)
    zs, ids = make_zs([([cs], 3, 5), ([cs], 7, 11)])
    ctl_aux = ctx.ctl_data(t, zs, ids, trace, 13, log_n)
    betas = [3, 7]                                     # the lookup challenges are the betas of the CTL challenges (prover.rs:468-474)
    lt = CtlTable()
    looking = lt.colset([lt.single(10)])
    table_col, freq_col = lt.single(11), lt.single(12)
    lk = [ctx.lookup_helper_columns(lt, [looking], table_col, freq_col, b, trace, 13, log_n) for b in betas]
    aux = np.concatenate(lk + [ctl_aux])
    assert aux.size == (4 + 2) * n
    alphas = [5, 7]
    assert ctx.check_constraints(trace, log_n, aux, t, zs, ids, alphas, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=betas) is None
    aux2 = aux.copy()
    aux2[3] = (int(aux2[3]) + 1) % P                   # a helper value of the first lookup challenge
    got = ctx.check_constraints(trace, log_n, aux2, t, zs, ids, alphas, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=betas)
    assert got in (2, 3)
    with pytest.raises(zkm.ZkmError, match="lookup challenges"):
        ctx.check_constraints(trace, log_n, aux, t, zs, ids, alphas, ncols=13, table_id=T.TABLE_MEMORY)
