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
