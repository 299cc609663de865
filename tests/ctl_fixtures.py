"""Synthetic multi-table instance used by the CTL / prove_with_traces tests (CPU oracle and GPU parity).

Four PoseidonStark tables linked by two cross-table lookups (the reference links 12 tables with 15 CTLs,
all_stark.rs:136-154; only the Poseidon table has a HIP constraint kernel so far):
  CTL 0: looking T0.inputs, T1.inputs            -> looked T2.inputs          (T2 holds the rows of T0 and T1)
  CTL 1: looking T0.inputs, T0.inputs' (two column sets of one table -> a helper column) -> looked T3.mixed
         (T3 holds every row of T0 twice)
Column sets use linear combinations with constants and a product filter so every descriptor field is exercised.
"""
import numpy as np

from zkm_amd.ctl import CtlTable

P = 0xFFFFFFFF00000001
W = 262


def rows_to_trace(rows, log_n, default_row):
    n = 1 << log_n
    full = np.tile(default_row, (n, 1))
    full[:len(rows)] = rows
    return np.ascontiguousarray(full.T).reshape(-1)


def trace_rows(trace, log_n, k):
    n = 1 << log_n
    return trace.reshape(W, n).T[:k].copy()


def colsets(t, variant):
    """variant 'a': Column::singles(in0..in11, timestamp) + Filter::new_simple(FILTER)
       variant 'm': mixed linear forms + product filter FILTER*FILTER"""
    if variant == "a":
        return t.singles_set(list(range(1, 13)) + [25], filter_col=0)
    first = t.column(local=[(1, 1), (2, 2)], constant=5)            # in0 + 2 in1 + 5
    t.column(local=[(3, 7)])                                          # 7 in2
    t.le_bits([4, 5, 6])                                              # in3 + 2 in4 + 4 in5
    t.column(local=[(13, 1), (25, 3)], constant=P - 1)               # out0 + 3 ts - 1
    f = t.single(0)
    return t.colset(range(first, first + 4), filter_products=[(f, f)])


def build(oracle, log_small=5, k0=20, k1=9):
    default_row = oracle.poseidon_witness_row([0] * 12, 0, 0)
    t0 = oracle.poseidon_trace(11, k0, log_small)
    t1 = oracle.poseidon_trace(12, k1, log_small)
    r0, r1 = trace_rows(t0, log_small, k0), trace_rows(t1, log_small, k1)
    log_big = log_small + 1
    rng = np.random.default_rng(5)
    r2 = np.concatenate([r0, r1])
    rng.shuffle(r2, axis=0)
    r3 = np.concatenate([r0, r0])
    rng.shuffle(r3, axis=0)
    t2 = rows_to_trace(r2, log_big, default_row)
    t3 = rows_to_trace(r3, log_big, default_row)
    ctl0, ctl1, ctl2, ctl3 = CtlTable(), CtlTable(), CtlTable(), CtlTable()
    a0 = colsets(ctl0, "a")
    m0 = colsets(ctl0, "m")
    m0b = colsets(ctl0, "m")
    a1 = colsets(ctl1, "a")
    a2 = colsets(ctl2, "a")
    m3 = colsets(ctl3, "m")
    tables = [(0, t0, W, log_small, ctl0), (0, t1, W, log_small, ctl1), (0, t2, W, log_big, ctl2), (0, t3, W, log_big, ctl3)]
    ctls = [([(0, a0), (1, a1)], (2, a2)),
            ([(0, m0), (0, m0b)], (3, m3))]
    return tables, ctls
