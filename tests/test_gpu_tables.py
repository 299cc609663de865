"""GPU parity for the Logic and KeccakSponge tables (N2): witness generation, quotient kernels, single-table proofs and
the KeccakSponge -> Logic cross-table lookup (all_stark.rs:340-355), bit-exact against the CPU oracle."""
import numpy as np
import pytest

from zkm_amd import tables as T

from . import logic_fixtures
from .sponge_fixtures import ops_for_rows

pytestmark = pytest.mark.gpu
EMPTY = np.zeros(0, dtype=np.uint64)


def random_logic_ops(seed, k):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, 4, k), rng.integers(0, 1 << 32, k), rng.integers(0, 1 << 32, k)], axis=1).astype(np.uint32)


def fake_ctl_aux(log_n, seed=3):
    """h0, h1 random; Z(w^i) = sum_{k >= i} (h0 + h1)(w^k): passes the last-row / transition checks on Z."""
    P = 0xFFFFFFFF00000001
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    h = [[int(x) for x in rng.integers(0, 1 << 62, n)] for _ in range(2)]
    z, acc = [0] * n, 0
    for i in range(n - 1, -1, -1):
        acc = (acc + h[0][i] + h[1][i]) % P
        z[i] = acc
    return np.array(h[0] + h[1] + z, dtype=np.uint64)


def table_trace(oracle, table_id, log_n, seed=2):
    if table_id == T.TABLE_KECCAK:
        k = (1 << log_n) // 24
        rng = np.random.default_rng(seed)
        return oracle.keccak_trace(rng.integers(0, 1 << 64, (k, 25), dtype=np.uint64), rng.integers(0, 1 << 30, k), log_n)
    if table_id == T.TABLE_LOGIC:
        return oracle.logic_trace(random_logic_ops(seed, (1 << log_n) - 5), log_n)
    data, off, meta, rows, nops = ops_for_rows(seed, (1 << log_n) - 3)
    return oracle.keccak_sponge_trace(data, off, meta, log_n)[0]


@pytest.mark.parametrize("log_n,k", [(3, 8), (6, 40), (12, 4000), (16, 1 << 16)])
def test_logic_trace_matches_oracle(ctx, oracle, log_n, k):
    ops = random_logic_ops(log_n, k)
    ops[: min(k, 4), 1:] = [[0, 0], [0xFFFFFFFF, 0xFFFFFFFF], [0, 0xFFFFFFFF], [0x80000000, 1]][: min(k, 4)]
    got = ctx.logic_trace(ops, log_n)
    assert (got.download() == oracle.logic_trace(ops, log_n)).all()
    got.free()
    # device-resident operations
    d_ops = ctx.alloc((ops.size * 4 + 7) // 8)
    d_ops.upload(np.frombuffer(ops.tobytes() + b"\0" * (-ops.size * 4 % 8), dtype=np.uint64))
    import ctypes as C
    out = ctx.alloc(69 << log_n)
    err = C.c_char_p()
    assert ctx.L.zkm_logic_trace(ctx.h, d_ops.ptr, k, log_n, out.ptr, C.byref(err)) == 0
    assert (out.download() == oracle.logic_trace(ops, log_n)).all()


def test_logic_trace_errors(ctx, zkm):
    with pytest.raises(zkm.ZkmError, match="more operations"):
        ctx.logic_trace(random_logic_ops(1, 9), 3)
    bad = random_logic_ops(1, 8)
    bad[5, 0] = 4
    with pytest.raises(zkm.ZkmError, match="op code"):
        ctx.logic_trace(bad, 3)
    # empty: all-zero padding rows
    assert not ctx.logic_trace(np.zeros((0, 3), dtype=np.uint32), 3).download().any()


@pytest.mark.parametrize("log_n,k", [(5, 1), (7, 5), (10, 42), (13, 300)])
def test_keccak_trace_matches_oracle(ctx, oracle, log_n, k):
    rng = np.random.default_rng(log_n)
    inputs = rng.integers(0, 1 << 64, (k, 25), dtype=np.uint64)
    inputs[0] = 0
    ts = rng.integers(0, 1 << 40, k).astype(np.uint64)
    want = oracle.keccak_trace(inputs, ts, log_n)
    got = ctx.keccak_trace(inputs, ts, log_n)
    assert (got.download() == want).all()
    # device-resident inputs, caller-provided output
    d_in = ctx.alloc(inputs.size).upload(inputs.reshape(-1))
    d_ts = ctx.alloc(ts.size).upload(ts)
    ctx.keccak_trace(d_in, d_ts, log_n, out=got)
    assert (got.download() == want).all()


def test_keccak_trace_errors(ctx, zkm):
    with pytest.raises(zkm.ZkmError, match="more rows"):
        ctx.keccak_trace(np.zeros((2, 25), dtype=np.uint64), [1, 2], 5)
    assert not ctx.keccak_trace(np.zeros((0, 25), dtype=np.uint64), np.zeros(0, dtype=np.uint64), 5).download().any()


@pytest.mark.parametrize("table_id,log_n", [(T.TABLE_LOGIC, 5), (T.TABLE_LOGIC, 11), (T.TABLE_KECCAK_SPONGE, 5),
                                            (T.TABLE_KECCAK_SPONGE, 10), (T.TABLE_KECCAK, 5), (T.TABLE_KECCAK, 8)])
@pytest.mark.parametrize("nalphas", [1, 2])
def test_quotient_matches_oracle(ctx, zkm, oracle, table_id, log_n, nalphas):
    W = T.WIDTH[table_id]
    trace = table_trace(oracle, table_id, log_n)
    rng = np.random.default_rng(5)
    aux = rng.integers(0, 1 << 63, 3 << log_n, dtype=np.uint64)     # 2 helpers + 1 Z (fake CTL data, as the benchmark)
    alphas = [int(x) for x in rng.integers(1, 1 << 62, nalphas)]
    tb = zkm.PolynomialBatch.from_values(ctx, trace, W, log_n)
    ab = zkm.PolynomialBatch.from_values(ctx, aux, 3, log_n)
    got = ctx.quotient(tb, ab, [2], alphas, table_id=table_id)
    otb = oracle.batch_from_values(trace, W, log_n)
    oab = oracle.batch_from_values(aux, 3, log_n)
    want = oracle.quotient(otb, oab, [2], alphas, table_id=table_id)
    assert (got == want).all()


@pytest.mark.parametrize("table_id,log_n", [(T.TABLE_LOGIC, 4), (T.TABLE_LOGIC, 10), (T.TABLE_KECCAK_SPONGE, 4),
                                            (T.TABLE_KECCAK_SPONGE, 9), (T.TABLE_KECCAK, 5), (T.TABLE_KECCAK, 8)])
def test_single_table_proof_is_bit_exact(ctx, oracle, table_id, log_n):
    W = T.WIDTH[table_id]
    trace = table_trace(oracle, table_id, log_n, seed=7)
    # benchmark-style CTL stand-in (prover.rs:420-438): a Z column consistent with its two helper columns
    aux = fake_ctl_aux(log_n)
    want = oracle.prove(trace, log_n, aux, [2], ncols=W, table_id=table_id)
    got = ctx.prove_single_table(trace, log_n, aux, [2], ncols=W, table_id=table_id)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify(got, 3, [2], ncols=W, table_id=table_id) == 0


def test_wrong_width_or_missing_ctl_is_rejected(ctx, zkm, oracle):
    trace = table_trace(oracle, T.TABLE_LOGIC, 4)
    aux = fake_ctl_aux(4)
    with pytest.raises(zkm.ZkmError):
        ctx.prove_single_table(trace, 4, aux, [2], ncols=69, table_id=T.TABLE_KECCAK_SPONGE)
    with pytest.raises(zkm.ZkmError):
        ctx.prove_single_table(trace, 4, aux, [2], ncols=69, table_id=7)
    with pytest.raises(zkm.ZkmError, match="No CTL"):   # the reference asserts the same (prover.rs:509)
        ctx.prove_single_table(trace, 4, EMPTY, [], ncols=69, table_id=T.TABLE_LOGIC)


@pytest.mark.parametrize("log_sponge", [4, 7])
def test_sponge_logic_lookup_is_bit_exact_and_verifies(ctx, oracle, log_sponge):
    tables, ctls, ops = logic_fixtures.build(oracle, log_sponge=log_sponge)
    # witnesses from the GPU generators match the oracle's
    tid, sponge, w, ls, cs = tables[0]
    tid1, logic, w1, ll, cl = tables[1]
    assert (ctx.logic_trace(ops, ll).download() == logic).all()
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
    got, chal, offs = ctx.prove_with_traces(tables, ctls)
    assert offs == woffs and (chal == wchal).all()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal) == 0


def test_sponge_logic_lookup_device_resident_large(ctx, oracle):
    """2^11 sponge rows -> 2^17 Logic rows, all witnesses generated and kept on the device; the oracle verifies."""
    log_sponge = 11
    data, off, meta, rows, nops = ops_for_rows(33, (1 << log_sponge) - 2)
    d_sponge, used = ctx.keccak_sponge_trace(data, off, meta, log_sponge)
    assert used == rows
    sponge = d_sponge.download()
    ops = logic_fixtures.logic_ops_from_sponge(sponge, log_sponge, rows)
    log_logic = int(np.ceil(np.log2(len(ops))))
    d_logic = ctx.logic_trace(ops, log_logic)
    from zkm_amd.ctl import CtlTable
    cs, cl = CtlTable(), CtlTable()
    looking, looked = T.ctl_logic_keccak_sponge(0, 1, cs, cl)
    dev_tables = [(T.TABLE_KECCAK_SPONGE, d_sponge, 470, log_sponge, cs), (T.TABLE_LOGIC, d_logic, 69, log_logic, cl)]
    proofs, chal, offs = ctx.prove_with_traces(dev_tables, [(looking, looked)])
    host_tables = [(T.TABLE_KECCAK_SPONGE, sponge, 470, log_sponge, cs), (T.TABLE_LOGIC, d_logic.download(), 69, log_logic, cl)]
    assert oracle.verify_all(host_tables, [(looking, looked)], proofs, chal) == 0


def test_sponge_keccak_logic_is_bit_exact_and_verifies(ctx, oracle):
    """The three tables of the Keccak precompile path with their three lookups (all_stark.rs:214-240, 340-355)."""
    tables, ctls, (ops, inputs, ts) = logic_fixtures.build3(oracle, log_sponge=4)
    assert (ctx.keccak_trace(inputs, ts, tables[1][3]).download() == tables[1][1]).all()
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
    got, chal, offs = ctx.prove_with_traces(tables, ctls)
    assert offs == woffs and (chal == wchal).all()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal) == 0


def test_sponge_keccak_logic_device_resident_large(ctx, oracle):
    """2^9 sponge rows -> 2^14 Keccak rows (2431 columns) and 2^15 Logic rows, witnesses generated and kept on the device."""
    log_sponge = 9
    data, off, meta, rows, nops = ops_for_rows(35, (1 << log_sponge) - 1)
    d_sponge, used = ctx.keccak_sponge_trace(data, off, meta, log_sponge)
    sponge = d_sponge.download()
    ops = logic_fixtures.logic_ops_from_sponge(sponge, log_sponge, rows)
    inputs, ts = logic_fixtures.keccak_inputs_from_sponge(sponge, log_sponge, rows)
    log_logic, log_keccak = int(np.ceil(np.log2(len(ops)))), int(np.ceil(np.log2(24 * rows)))
    d_logic, d_keccak = ctx.logic_trace(ops, log_logic), ctx.keccak_trace(inputs, ts, log_keccak)
    from zkm_amd.ctl import CtlTable
    cs, cl, ck = CtlTable(), CtlTable(), CtlTable()
    ctls = [T.ctl_keccak_inputs(0, 1, cs, ck), T.ctl_keccak_outputs(0, 1, cs, ck), T.ctl_logic_keccak_sponge(0, 2, cs, cl)]
    dev = [(T.TABLE_KECCAK_SPONGE, d_sponge, 470, log_sponge, cs), (T.TABLE_KECCAK, d_keccak, 2431, log_keccak, ck),
           (T.TABLE_LOGIC, d_logic, 69, log_logic, cl)]
    proofs, chal, offs = ctx.prove_with_traces(dev, ctls)
    host = [(t, b.download(), w, l, c) for (t, b, w, l, c) in dev]
    assert oracle.verify_all(host, ctls, proofs, chal) == 0


@pytest.mark.parametrize("log_n,k", [(8, 200), (12, 3000)])
def test_memory_table_proof_with_range_check_lookup_is_bit_exact(ctx, zkm, oracle, log_n, k):
    """MemoryStark: 13 trace columns + the logUp range check (lookup.rs:46-198) whose helper columns the prover builds."""
    from zkm_amd.ctl import CtlTable, make_zs
    from .test_oracle_tables import random_memory_ops
    trace, natural = oracle.memory_trace(random_memory_ops(log_n, k), log_n)
    assert natural == 1 << log_n
    t = CtlTable()
    cs = T.memory_ctl_data(t)
    zs, ids = make_zs([([cs], 3, 5), ([cs], 7, 11)])
    aux = ctx.ctl_data(t, zs, ids, trace, 13, log_n)
    assert (aux == oracle.ctl_data(t, zs, ids, trace, 13, log_n)).all()
    lk = [0x1234567, 0x89ABCDEF01]
    want = oracle.prove_ctl(trace, log_n, aux, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=lk)
    got = ctx.prove_single_table_ctl(trace, log_n, aux, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=lk)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert got[3] == 4 + 2    # 2 x (1 helper + 1 Z) lookup columns in front of the 2 CTL Zs
    assert oracle.verify_ctl(got, 2, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=lk) == 0
    # the lookup needs its challenges and the trace values
    with pytest.raises(zkm.ZkmError, match="lookup challenges"):
        ctx.prove_single_table_ctl(trace, log_n, aux, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY)
    tb = zkm.PolynomialBatch.from_values(ctx, trace, 13, log_n)
    with pytest.raises(zkm.ZkmError, match="trace values"):
        ctx.prove_single_table_ctl(None, log_n, aux, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, trace_batch=tb, lookup_challenges=lk)


def test_precompile_path_four_tables_is_bit_exact_and_verifies(ctx, oracle):
    """Memory -> KeccakSponge -> (Keccak, Logic): every cross-table lookup the reference defines among these four tables
    (all_stark.rs:214-240, 340-355, 479-542), including the 136 per-byte memory reads (68 helper columns per challenge)."""
    tables, ctls, _ = logic_fixtures.build4(oracle, log_sponge=3)
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
    got, chal, offs = ctx.prove_with_traces(tables, ctls)
    assert offs == woffs and (chal == wchal).all()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal) == 0


def test_precompile_path_four_tables_larger(ctx, oracle):
    tables, ctls, _ = logic_fixtures.build4(oracle, log_sponge=7, seed=31)
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = ctx.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0


@pytest.mark.parametrize("log_sponge", [4, 9])
def test_poseidon_sponge_path_is_bit_exact_and_verifies(ctx, oracle, log_sponge):
    """Memory -> PoseidonSponge -> Poseidon: witnesses from the GPU generators, proofs bit-exact, oracle verify_proof accepts
    (all_stark.rs:169-195, 487-493)."""
    tables, ctls, (data, off, meta, inputs, ts, mem_ops) = logic_fixtures.build_poseidon_path(oracle, log_sponge=log_sponge)
    d_sponge, used = ctx.poseidon_sponge_trace(data, off, meta, log_sponge)
    assert (d_sponge.download() == tables[0][1]).all()
    assert (ctx.poseidon_trace_inputs(inputs, ts, tables[1][3]).download() == tables[1][1]).all()
    got, chal, offs = ctx.prove_with_traces(tables, ctls)
    if log_sponge <= 6:
        want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
        assert offs == woffs and (chal == wchal).all()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal) == 0


def test_poseidon_sponge_single_table_quotient(ctx, zkm, oracle):
    tables, ctls, _ = logic_fixtures.build_poseidon_path(oracle, log_sponge=6)
    tid, trace, W, log_n, cs = tables[0]
    aux = fake_ctl_aux(log_n)
    want = oracle.prove(trace, log_n, aux, [2], ncols=W, table_id=tid)
    got = ctx.prove_single_table(trace, log_n, aux, [2], ncols=W, table_id=tid)
    assert (got == want).all()
    assert oracle.verify(got, 3, [2], ncols=W, table_id=tid) == 0


def test_both_precompile_paths_share_the_memory_table(ctx, oracle):
    """Six tables in one proof: Memory is looked up by both sponges (168 looking column sets -> 84 helper columns per
    challenge), as in all_stark::ctl_memory."""
    from zkm_amd.ctl import CtlTable
    kt, kctls, (ops, kin, kts, kmem) = logic_fixtures.build4(oracle, log_sponge=3)
    pt, pctls, (data, off, meta, pin, pts, pmem) = logic_fixtures.build_poseidon_path(oracle, log_sponge=4)
    mem_ops = np.concatenate([kmem, pmem])
    log_mem = int(np.ceil(np.log2(len(mem_ops)))) + 1
    memory, natural = oracle.memory_trace(mem_ops, log_mem)
    if natural < (1 << log_mem):
        log_mem -= 1
        memory, natural = oracle.memory_trace(mem_ops, log_mem)
    cks, ck, cl, cps, cp, cm = (CtlTable() for _ in range(6))
    tables = [(kt[0][0], kt[0][1], 470, kt[0][3], cks), (kt[1][0], kt[1][1], 2431, kt[1][3], ck), (kt[2][0], kt[2][1], 69, kt[2][3], cl),
              (pt[0][0], pt[0][1], 110, pt[0][3], cps), (pt[1][0], pt[1][1], 262, pt[1][3], cp), (T.TABLE_MEMORY, memory, 13, log_mem, cm)]
    ctls = [T.ctl_poseidon_inputs(3, 4, cps, cp), T.ctl_poseidon_outputs(3, 4, cps, cp),
            T.ctl_keccak_inputs(0, 1, cks, ck), T.ctl_keccak_outputs(0, 1, cks, ck), T.ctl_logic_keccak_sponge(0, 2, cks, cl),
            (T.memory_lookers_keccak_sponge(0, cks) + T.memory_lookers_poseidon_sponge(3, cps), (5, T.memory_ctl_data(cm)))]
    assert oracle.check_ctls(tables, ctls) == 0
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
    got, chal, offs = ctx.prove_with_traces(tables, ctls)
    assert offs == woffs and (chal == wchal).all() and (got == want).all()
    assert oracle.verify_all(tables, ctls, got, chal) == 0


@pytest.mark.parametrize("nblocks", [1, 20])
def test_sha_extend_path_is_bit_exact_and_verifies(ctx, oracle, nblocks):
    """Memory -> ShaExtendSponge -> ShaExtend -> Logic: SHA-256 message schedules (all_stark.rs:256-282, 356-385, 503-509)."""
    tables, ctls, (w16, meta, inputs, ts, ops, mem_ops) = logic_fixtures.build_sha_extend_path(oracle, nblocks=nblocks)
    log_s = tables[0][3]
    assert (ctx.sha_extend_sponge_trace(w16, meta, log_s).download() == tables[0][1]).all()
    assert (ctx.sha_extend_trace(inputs, ts, log_s).download() == tables[1][1]).all()
    got, chal, offs = ctx.prove_with_traces(tables, ctls)
    if nblocks == 1:
        want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
        assert offs == woffs and (chal == wchal).all()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal) == 0


@pytest.mark.parametrize("t_idx", [0, 1])
def test_sha_extend_single_table_proofs(ctx, oracle, t_idx):
    tables, ctls, _ = logic_fixtures.build_sha_extend_path(oracle, nblocks=2)
    tid, trace, W, log_n, cs = tables[t_idx]
    aux = fake_ctl_aux(log_n)
    want = oracle.prove(trace, log_n, aux, [2], ncols=W, table_id=tid)
    got = ctx.prove_single_table(trace, log_n, aux, [2], ncols=W, table_id=tid)
    assert (got == want).all()
    assert oracle.verify(got, 3, [2], ncols=W, table_id=tid) == 0


def test_sha_extend_trace_errors(ctx, zkm):
    with pytest.raises(zkm.ZkmError, match="48 each"):
        ctx.sha_extend_sponge_trace(np.zeros((2, 16), dtype=np.uint32), np.zeros((2, 4), dtype=np.uint64), 6)
    with pytest.raises(zkm.ZkmError, match="more rows"):
        ctx.sha_extend_trace(np.zeros((9, 16), dtype=np.uint8), np.zeros(9, dtype=np.uint64), 3)


@pytest.mark.parametrize("ncomp", [1, 9])
def test_sha_compress_path_is_bit_exact_and_verifies(ctx, oracle, ncomp):
    """Memory -> ShaCompressSponge -> ShaCompress -> Logic: SHA-256 compressions (all_stark.rs:298-324, 387-470, 511-525)."""
    tables, ctls, (hx, w, meta, ops, mem_ops) = logic_fixtures.build_sha_compress_path(oracle, ncomp=ncomp)
    assert (ctx.sha_compress_sponge_trace(hx, w, meta, tables[0][3]).download() == tables[0][1]).all()
    assert (ctx.sha_compress_trace(hx, w, meta, tables[1][3]).download() == tables[1][1]).all()
    got, chal, offs = ctx.prove_with_traces(tables, ctls)
    if ncomp == 1:
        want, wchal, woffs = oracle.prove_with_traces(tables, ctls)
        assert offs == woffs and (chal == wchal).all()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal) == 0


@pytest.mark.parametrize("t_idx", [0, 1])
def test_sha_compress_single_table_proofs(ctx, oracle, t_idx):
    tables, ctls, _ = logic_fixtures.build_sha_compress_path(oracle, ncomp=3)
    tid, trace, W, log_n, cs = tables[t_idx]
    aux = fake_ctl_aux(log_n)
    want = oracle.prove(trace, log_n, aux, [2], ncols=W, table_id=tid)
    got = ctx.prove_single_table(trace, log_n, aux, [2], ncols=W, table_id=tid)
    assert (got == want).all()
    assert oracle.verify(got, 3, [2], ncols=W, table_id=tid) == 0


def test_arithmetic_table_proof_is_bit_exact(ctx, zkm, oracle):
    """ArithmeticStark: all 26 operations + the 18-column range-check lookup (20 lookup columns built on the GPU), bit-exact
    against the oracle; corrupted rows of every constraint family are rejected by the oracle's verifier on GPU proofs too."""
    from zkm_amd.ctl import CtlTable, make_zs
    from . import arith_fixtures as A
    log_n, n = 16, 1 << 16
    ops = A.random_ops(1, 200)
    trace = A.generate_trace(ops)
    t = CtlTable()
    cs = T.arithmetic_ctl_rows(t)          # the column set the CPU table looks up (arithmetic_stark.rs:66-126)
    zs, ids = make_zs([([cs], 3, 5), ([cs], 7, 11)])
    aux = ctx.ctl_data(t, zs, ids, trace, 54, log_n)
    assert (aux == oracle.ctl_data(t, zs, ids, trace, 54, log_n)).all()
    lk = [0x1234567, 0x89ABCDEF01]
    want = oracle.prove_ctl(trace, log_n, aux, t, zs, ids, ncols=54, table_id=T.TABLE_ARITHMETIC, lookup_challenges=lk)
    got = ctx.prove_single_table_ctl(trace, log_n, aux, t, zs, ids, ncols=54, table_id=T.TABLE_ARITHMETIC, lookup_challenges=lk)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert got[3] == 20 + 2
    assert oracle.verify_ctl(got, 2, t, zs, ids, ncols=54, table_id=T.TABLE_ARITHMETIC, lookup_challenges=lk) == 0
    first_row, r = {}, 0
    for op, a, b in ops:
        first_row.setdefault(op, r)
        r += 2 if op in (A.IS_DIV, A.IS_DIVU, A.IS_SRL, A.IS_SRLV, A.IS_SRA, A.IS_SRAV) else 1
    cases = [(A.IS_ADD, A.OUT, 0), (A.IS_SUB, A.AUX0, 0), (A.IS_MUL, A.OUT + 1, 0), (A.IS_MULT, A.OUT_HI, 0), (A.IS_MULTU, A.OUT_LO, 0),
             (A.IS_SLT, A.OUT, 0), (A.IS_SLTU, A.AUX0, 0), (A.IS_LUI, A.OUT + 1, 0), (A.IS_DIVU, A.OUT_HI, 0), (A.IS_DIV, A.QUOT_ABS, 0),
             (A.IS_DIV, A.NV_DENOM_IS_ZERO + 5, 1), (A.IS_SLL, A.OUT, 0), (A.IS_SRL, A.OUT, 0), (A.IS_SRA, A.OUT + 1, 0), (A.IS_SRAV, A.AUX_EXTRA + 3, 1),
             (A.IS_MFLO, A.OUT, 0), (A.IS_ADDI, A.OUT, 0), (A.IS_SLTI, A.AUX1, 0), (A.IS_SLLV, A.AUX0, 0), (A.IS_SRLV, A.AUX0, 0)]
    for op, col, second in cases:
        badt = trace.copy()
        idx = col * n + first_row[op] + second
        badt[idx] = (int(badt[idx]) + 1) & 0xFFFF if col < A.RANGE_COUNTER else int(badt[idx]) + 1
        tr = badt.reshape(54, n)
        tr[A.RC_FREQ] = 0
        tr[A.RC_FREQ] += np.bincount(tr[26:44].reshape(-1).astype(np.int64), minlength=n)[:n].astype(np.uint64)
        aux2 = ctx.ctl_data(t, zs, ids, badt, 54, log_n)
        p2 = ctx.prove_single_table_ctl(badt, log_n, aux2, t, zs, ids, ncols=54, table_id=T.TABLE_ARITHMETIC, lookup_challenges=lk)
        assert oracle.verify_ctl(p2, 2, t, zs, ids, ncols=54, table_id=T.TABLE_ARITHMETIC, lookup_challenges=lk) != 0, (op, col)


def test_eleven_tables_in_one_proof(ctx, oracle):
    """Every table with a constraint kernel in ONE prove_with_traces call on one transcript: the Keccak, Poseidon, SHA-extend and
    SHA-compress paths share the Logic and Memory tables exactly as in all_stark::ctl_logic / ctl_memory (minus the CPU lookers), and
    the Arithmetic table rides along with its own column set.  GPU proof == oracle proof, and verify_proof accepts."""
    from zkm_amd.ctl import CtlTable
    from . import arith_fixtures as A
    kt, _, (kops, kin, kts, kmem) = logic_fixtures.build4(oracle, log_sponge=3)
    pt, _, (pdata, poff, pmeta, pin, pts, pmem) = logic_fixtures.build_poseidon_path(oracle, log_sponge=4)
    et, _, (ew16, emeta, ein, ets, eops, emem) = logic_fixtures.build_sha_extend_path(oracle, nblocks=1)
    ct, _, (chx, cw, cmeta, cops, cmem) = logic_fixtures.build_sha_compress_path(oracle, ncomp=1)

    def memory_table(ops):
        log_m = int(np.ceil(np.log2(len(ops)))) + 1
        tr, natural = oracle.memory_trace(ops, log_m)
        if natural < (1 << log_m):
            log_m -= 1
            tr, natural = oracle.memory_trace(ops, log_m)
        return tr, log_m
    memory, log_mem = memory_table(np.concatenate([kmem, pmem, emem, cmem]))
    lops = np.concatenate([kops, eops, cops])
    np.random.default_rng(77).shuffle(lops, axis=0)
    log_logic = int(np.ceil(np.log2(len(lops))))
    logic = oracle.logic_trace(lops, log_logic)
    arith = A.generate_trace(A.random_ops(5, 60))
    c = [CtlTable() for _ in range(11)]
    KS, KK, PS, PO, SES, SE, SCS, SC, LO, ME, AR = range(11)
    tables = [(kt[0][0], kt[0][1], 470, kt[0][3], c[KS]), (kt[1][0], kt[1][1], 2431, kt[1][3], c[KK]),
              (pt[0][0], pt[0][1], 110, pt[0][3], c[PS]), (pt[1][0], pt[1][1], 262, pt[1][3], c[PO]),
              (et[0][0], et[0][1], 76, et[0][3], c[SES]), (et[1][0], et[1][1], 78, et[1][3], c[SE]),
              (ct[0][0], ct[0][1], 127, ct[0][3], c[SCS]), (ct[1][0], ct[1][1], 224, ct[1][3], c[SC]),
              (T.TABLE_LOGIC, logic, 69, log_logic, c[LO]), (T.TABLE_MEMORY, memory, 13, log_mem, c[ME]), (T.TABLE_ARITHMETIC, arith, 54, 16, c[AR])]
    arith_set = T.arithmetic_ctl_rows(c[AR])
    logic_lookers = [(KS, T.keccak_sponge_looking_logic(c[KS], i)) for i in range(T.NUM_LOGIC_CTLS)] + T.logic_lookers_sha_extend(SE, c[SE]) + \
        T.logic_lookers_sha_compress(SC, c[SC])
    memory_lookers = T.memory_lookers_keccak_sponge(KS, c[KS]) + T.memory_lookers_poseidon_sponge(PS, c[PS]) + \
        T.memory_lookers_sha_extend_sponge(SES, c[SES]) + T.memory_lookers_sha_compress_sponge(SCS, c[SCS]) + T.memory_lookers_sha_compress(SC, c[SC])
    ctls = [T.ctl_poseidon_inputs(PS, PO, c[PS], c[PO]), T.ctl_poseidon_outputs(PS, PO, c[PS], c[PO]),
            T.ctl_keccak_inputs(KS, KK, c[KS], c[KK]), T.ctl_keccak_outputs(KS, KK, c[KS], c[KK]),
            T.ctl_sha_extend_inputs(SES, SE, c[SES], c[SE]), T.ctl_sha_extend_outputs(SES, SE, c[SES], c[SE]),
            T.ctl_sha_compress_inputs(SCS, SC, c[SCS], c[SC]), T.ctl_sha_compress_outputs(SCS, SC, c[SCS], c[SC]),
            (logic_lookers, (LO, T.logic_ctl_data(c[LO]))), (memory_lookers, (ME, T.memory_ctl_data(c[ME]))),
            ([(AR, arith_set)], (AR, T.arithmetic_ctl_rows(c[AR])))]     # stand-in for ctl_arithmetic: the table looks itself up
    assert oracle.check_ctls(tables, ctls) == 0
    want, wchal, woffs = oracle.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    got, chal, offs = ctx.prove_with_traces(tables, ctls, public_values=[1, 2, 3])
    assert offs == woffs and (chal == wchal).all()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing word %d" % bad[0]
    assert oracle.verify_all(tables, ctls, got, chal, public_values=[1, 2, 3]) == 0


@pytest.mark.parametrize("table_id,log_n", [(T.TABLE_KECCAK, 15), (T.TABLE_SHA_COMPRESS, 17), (T.TABLE_SHA_EXTEND, 18), (T.TABLE_POSEIDON_SPONGE, 18)])
def test_large_tables_gpu_proof_verifies(ctx, oracle, table_id, log_n):
    """Sizes the oracle would take minutes to prove: witness and proof on the GPU, the oracle only verifies (openings, constraint
    identity at zeta, FRI)."""
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    if table_id == T.TABLE_KECCAK:
        k = n // 24
        trace = ctx.keccak_trace(rng.integers(0, 1 << 64, (k, 25), dtype=np.uint64), rng.integers(0, 1 << 30, k), log_n)
    elif table_id == T.TABLE_SHA_COMPRESS:
        k = n // 65
        meta = np.zeros((k, 8), dtype=np.uint64)
        meta[:, 2], meta[:, 3], meta[:, 4] = np.arange(k) * 4096, np.arange(k) + 5, np.arange(k) * 4096 + 1024
        trace = ctx.sha_compress_trace(rng.integers(0, 1 << 32, (k, 8), dtype=np.uint64), rng.integers(0, 1 << 32, (k, 64), dtype=np.uint64), meta, log_n)
    elif table_id == T.TABLE_SHA_EXTEND:
        trace = ctx.sha_extend_trace(rng.integers(0, 256, (n - 9, 16), dtype=np.uint8), np.arange(n - 9) + 1, log_n)
    else:
        data, off, meta, rows, nops = logic_fixtures.poseidon_sponge_ops(3, n - 5)
        trace, used = ctx.poseidon_sponge_trace(data, off, meta, log_n)
        assert used == rows
    W = T.WIDTH[table_id]
    aux = fake_ctl_aux(log_n) if log_n <= 16 else None
    if aux is None:   # the Python stand-in is too slow beyond 2^16 rows: zeros are a valid helper / Z pair too
        aux = np.zeros(3 << log_n, dtype=np.uint64)
    proof = ctx.prove_single_table(trace, log_n, aux, [2], ncols=W, table_id=table_id)
    assert oracle.verify(proof, 3, [2], ncols=W, table_id=table_id) == 0
    proof[len(proof) // 2] ^= 1
    assert oracle.verify(proof, 3, [2], ncols=W, table_id=table_id) != 0
