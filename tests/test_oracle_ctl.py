"""CPU suite: cross-table lookup data, check_ctls and prove_with_traces -> verify_proof of the oracle."""
import numpy as np

from tests.ctl_fixtures import build, colsets
from zkm_amd.ctl import CtlTable, make_zs

P = 0xFFFFFFFF00000001


def test_check_ctls_accepts_and_rejects(oracle):
    tables, ctls = build(oracle)
    assert oracle.check_ctls(tables, ctls) == 0
    # break one looked row: multisets differ (cross_table_lookup.rs:1486-1581 would panic)
    tid, tr, w, log_n, ct = tables[2]
    bad = tr.copy()
    bad[1 * (1 << log_n) + 0] = (int(bad[1 * (1 << log_n) + 0]) + 1) % P
    tables2 = list(tables)
    tables2[2] = (tid, bad, w, log_n, ct)
    assert oracle.check_ctls(tables2, ctls) != 0


def test_ctl_data_matches_definition(oracle):
    # helper column h = sum f_i / (combine_i), Z = upside-down running sum (cross_table_lookup.rs:709-872)
    log_n, k = 5, 20
    n = 1 << log_n
    trace = oracle.poseidon_trace(3, k, log_n)
    t = CtlTable()
    a = colsets(t, "a")
    m = colsets(t, "m")
    beta, gamma = 0x1234567890ABCDEF % P, 0x0FEDCBA987654321 % P
    zs, ids = make_zs([([a, m], beta, gamma), ([m], gamma, beta)])
    assert list(zs["num_helpers"]) == [1, 0]
    aux = oracle.ctl_data(t, zs, ids, trace, 262, log_n).reshape(3, n)
    cols = trace.reshape(262, n)

    def comb_a(d):
        vals = [int(cols[c][d]) for c in list(range(1, 13)) + [25]]
        return (sum(v * pow(beta, i, P) for i, v in enumerate(vals)) + gamma) % P

    def comb_m(d, b, g):
        vals = [(int(cols[1][d]) + 2 * int(cols[2][d]) + 5) % P, 7 * int(cols[3][d]) % P,
                (int(cols[4][d]) + 2 * int(cols[5][d]) + 4 * int(cols[6][d])) % P, (int(cols[13][d]) + 3 * int(cols[25][d]) - 1) % P]
        return (sum(v * pow(b, i, P) for i, v in enumerate(vals)) + g) % P

    h = [((pow(comb_a(d), P - 2, P) + pow(comb_m(d, beta, gamma), P - 2, P)) % P) if cols[0][d] == 1 else 0 for d in range(n)]
    assert [int(x) for x in aux[0]] == h
    z0 = [sum(h[d:]) % P for d in range(n)]
    assert [int(x) for x in aux[1]] == z0
    h1 = [pow(comb_m(d, gamma, beta), P - 2, P) if cols[0][d] == 1 else 0 for d in range(n)]
    assert [int(x) for x in aux[2]] == [sum(h1[d:]) % P for d in range(n)]


def test_single_table_with_real_ctl_data_proves_and_verifies(oracle):
    log_n, k = 5, 17
    n = 1 << log_n
    trace = oracle.poseidon_trace(4, k, log_n)
    t = CtlTable()
    a, m = colsets(t, "a"), colsets(t, "m")
    m2 = colsets(t, "m")
    zs, ids = make_zs([([a, m, m2], 3, 5), ([a], 7, 11), ([m, a], 13, 17)])
    assert list(zs["num_helpers"]) == [2, 0, 1]
    aux = oracle.ctl_data(t, zs, ids, trace, 262, log_n)
    proof = oracle.prove_ctl(trace, log_n, aux, t, zs, ids)
    assert oracle.verify_ctl(proof, aux.size >> log_n, t, zs, ids) == 0
    # wrong challenge in the description -> the CTL constraints no longer vanish
    zs_bad = zs.copy()
    zs_bad[1]["beta"] = 8
    assert oracle.verify_ctl(proof, aux.size >> log_n, t, zs_bad, ids) != 0
    # corrupted Z column -> rejected
    aux_bad = aux.copy()
    aux_bad[-3] = (int(aux_bad[-3]) + 1) % P
    bad = oracle.prove_ctl(trace, log_n, aux_bad, t, zs, ids)
    assert oracle.verify_ctl(bad, aux.size >> log_n, t, zs, ids) != 0


def test_prove_with_traces_then_verify(oracle):
    tables, ctls = build(oracle)
    pub = [1, 2, 3, 4, 5]
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls, public_values=pub)
    assert len(offs) == len(tables) + 1 and offs[-1] == proofs.size
    assert oracle.verify_all(tables, ctls, proofs, chal, public_values=pub) == 0
    # different public values -> different transcript -> rejected
    assert oracle.verify_all(tables, ctls, proofs, chal, public_values=[1, 2, 3, 4, 6]) != 0
    # tamper with one table's proof
    bad = proofs.copy()
    bad[offs[2] + 200] ^= 1
    assert oracle.verify_all(tables, ctls, bad, chal, public_values=pub) != 0


def test_cross_table_sum_catches_inconsistent_tables(oracle):
    # each table's own proof is valid, but the looked table misses one row: only verify_cross_table_lookups notices
    tables, ctls = build(oracle)
    tid, tr, w, log_n, ct = tables[2]
    n = 1 << log_n
    cols = tr.reshape(262, n).copy()
    cols[0][0] = 0  # clear FILTER on one real row of the looked table (its Poseidon constraints still hold)
    tables2 = list(tables)
    tables2[2] = (tid, np.ascontiguousarray(cols).reshape(-1), w, log_n, ct)
    assert oracle.check_ctls(tables2, ctls) != 0
    proofs, chal, offs = oracle.prove_with_traces(tables2, ctls)
    rc = oracle.verify_all(tables2, ctls, proofs, chal)
    assert 50 <= rc < 60, rc


def test_lookup_helper_columns_definition(oracle):
    # Z(1) = 0 and Z(gx) = Z(x) + sum h_i(x) - m(x) / (x + t(x))  (lookup.rs:40-44, 111-121)
    log_n = 4
    n = 1 << log_n
    trace = oracle.poseidon_trace(9, 10, log_n)
    cols = trace.reshape(262, n)
    t = CtlTable()
    s0 = t.colset([t.single(1)], filter_constants=[t.single(0)])
    s1 = t.colset([t.single(2)])
    s2 = t.colset([t.column(local=[(3, 2)], constant=1)])
    tc, fc = t.single(5), t.single(0)
    ch = 12345
    out = oracle.lookup_helper_columns(t, [s0, s1, s2], tc, fc, ch, trace, 262, log_n).reshape(3, n)
    inv = lambda x: pow(x % P, P - 2, P)
    h0 = [((inv(ch + int(cols[1][d])) if cols[0][d] == 1 else 0) + inv(ch + int(cols[2][d]))) % P for d in range(n)]
    h1 = [inv(ch + 2 * int(cols[3][d]) + 1) for d in range(n)]
    assert [int(x) for x in out[0]] == h0 and [int(x) for x in out[1]] == h1
    z = [0]
    for i in range(n - 1):
        z.append((z[-1] + h0[i] + h1[i] - int(cols[0][i]) * inv(ch + int(cols[5][i]))) % P)
    assert [int(x) for x in out[2]] == z
