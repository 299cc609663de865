"""Oracle: LogicStark and KeccakSpongeStark constraints (logic.rs:199-248, keccak_sponge_stark.rs:456-567) accept honest
witnesses, reject corrupted ones, and the KeccakSponge -> Logic cross-table lookup (all_stark.rs:340-355) balances."""
import numpy as np
import pytest

from zkm_amd import tables as T

from . import logic_fixtures
from .sponge_fixtures import ops_for_rows

EMPTY = np.zeros(0, dtype=np.uint64)
P = 0xFFFFFFFF00000001


def test_logic_trace_ops(oracle):
    rng = np.random.default_rng(1)
    ops = np.stack([rng.integers(0, 4, 50), rng.integers(0, 1 << 32, 50), rng.integers(0, 1 << 32, 50)], axis=1).astype(np.uint32)
    tr = oracle.logic_trace(ops, 6).reshape(69, 64)
    for r, (op, a, b) in enumerate(ops.tolist()):
        want = [a & b, a | b, a ^ b, ~(a | b) & 0xFFFFFFFF][op]
        assert int(tr[68, r]) == want
        assert tr[:4, r].tolist() == [int(op == k) for k in range(4)]
        assert sum(int(tr[4 + i, r]) << i for i in range(32)) == a
    assert not tr[:, 50:].any()


def test_keccak_trace_rows(oracle):
    """keccak_stark.rs:655-686: the output limbs of row 23 are keccakf(input); plus round flags / timestamps / padding."""
    rng = np.random.default_rng(4)
    inputs = rng.integers(0, 1 << 64, (5, 25), dtype=np.uint64)
    ts = np.arange(5, dtype=np.uint64) * 3 + 2
    n = 128
    tr = oracle.keccak_trace(inputs, ts, 7).reshape(2431, n)
    for p in range(5):
        want = oracle.keccakf(inputs[p])
        row = 24 * p + 23
        got = [int(tr[T.kk_reg_output_limb(2 * i), row]) | (int(tr[T.kk_reg_output_limb(2 * i + 1), row]) << 32) for i in range(25)]
        assert got == [int(x) for x in want]
        first = [int(tr[T.kk_reg_input_limb(2 * i), 24 * p]) | (int(tr[T.kk_reg_input_limb(2 * i + 1), 24 * p]) << 32) for i in range(25)]
        assert first == [int(x) for x in inputs[p]]
        assert (tr[:24, 24 * p:24 * p + 24] == np.eye(24, dtype=np.uint64)).all()
        assert (tr[24, 24 * p:24 * p + 24] == ts[p]).all()
    assert not tr[:, 120:].any()
    assert tr[75:2315].max() == 1    # C, C', A' are bits


@pytest.mark.parametrize("table_id,log_n", [(T.TABLE_LOGIC, 6), (T.TABLE_KECCAK_SPONGE, 5), (T.TABLE_KECCAK, 6)])
def test_single_table_proof(oracle, table_id, log_n):
    if table_id == T.TABLE_KECCAK:
        rng = np.random.default_rng(3)
        trace = oracle.keccak_trace(rng.integers(0, 1 << 64, (2, 25), dtype=np.uint64), [5, 9], log_n)
    elif table_id == T.TABLE_LOGIC:
        rng = np.random.default_rng(2)
        ops = np.stack([rng.integers(0, 4, 40), rng.integers(0, 1 << 32, 40), rng.integers(0, 1 << 32, 40)], axis=1)
        trace = oracle.logic_trace(ops, log_n)
    else:
        data, off, meta, rows, nops = ops_for_rows(9, (1 << log_n) - 3)
        trace, _ = oracle.keccak_sponge_trace(data, off, meta, log_n)
    W = T.WIDTH[table_id]
    proof = oracle.prove(trace, log_n, EMPTY, [], ncols=W, table_id=table_id)
    assert oracle.verify(proof, 0, [], ncols=W, table_id=table_id) == 0
    # a corrupted witness cell makes the quotient a non-polynomial: verification must fail
    bad = trace.copy()
    col = {T.TABLE_LOGIC: 68, T.TABLE_KECCAK_SPONGE: 39, T.TABLE_KECCAK: 900}[table_id]   # RESULT / already_absorbed_bytes / an A' bit
    bad[col * (1 << log_n) + 1] ^= 1
    proof = oracle.prove(bad, log_n, EMPTY, [], ncols=W, table_id=table_id)
    assert oracle.verify(proof, 0, [], ncols=W, table_id=table_id) != 0


def test_sponge_logic_ctl(oracle):
    tables, ctls, ops = logic_fixtures.build(oracle)
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0
    # drop one XOR from the Logic table: the lookup no longer balances
    bad_ops = ops.copy()
    bad_ops[0, 1] ^= 4
    tid, tr, w, log_n, ct = tables[1]
    bad_tables = [tables[0], (tid, oracle.logic_trace(bad_ops, log_n), w, log_n, ct)]
    assert oracle.check_ctls(bad_tables, ctls) != 0


def test_sponge_keccak_logic_ctls(oracle):
    tables, ctls, (ops, inputs, ts) = logic_fixtures.build3(oracle)
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0
    # a Keccak table that permutes a different input: the inputs lookup no longer balances
    bad_inputs = inputs.copy()
    bad_inputs[0, 3] ^= 1
    tid, tr, w, log_n, ct = tables[1]
    bad_tables = [tables[0], (tid, oracle.keccak_trace(bad_inputs, ts, log_n), w, log_n, ct), tables[2]]
    assert oracle.check_ctls(bad_tables, ctls) != 0


def random_memory_ops(seed, k):
    rng = np.random.default_rng(seed)
    addr = rng.integers(0, 40, k)
    ops = np.zeros((k, 6), dtype=np.uint64)
    ops[:, 0] = rng.integers(0, 2, k)          # context
    ops[:, 1] = rng.integers(0, 5, k)          # segment
    ops[:, 2] = addr * 4
    ops[:, 3] = rng.permutation(k) * 3 + 1     # distinct timestamps
    # reads must return the last written value: replay in timestamp order per address
    order = np.argsort(ops[:, 3])
    mem = {}
    for i in order:
        key = (int(ops[i, 0]), int(ops[i, 1]), int(ops[i, 2]))
        if key not in mem or rng.integers(0, 3) == 0:
            ops[i, 4], ops[i, 5] = 0, rng.integers(0, 1 << 32)
            mem[key] = int(ops[i, 5])
            if key == (0, 4, 0):
                mem[key] = 0                    # writes to R0 are stored as 0 (memory_stark.rs:68-76)
        else:
            ops[i, 4], ops[i, 5] = 1, mem[key]
    return ops


def test_memory_trace_and_proof(oracle):
    ops = random_memory_ops(1, 200)
    log_n = 8
    trace, natural = oracle.memory_trace(ops, log_n)
    assert natural == 256
    tr = trace.reshape(13, 256)
    key = [tuple(int(tr[c, i]) for c in (3, 4, 5, 1)) for i in range(256)]
    assert key == sorted(key)                                   # ordered by (context, segment, virt, timestamp)
    assert int(tr[0].sum()) == 200                              # every real operation present, padding has filter 0
    assert (tr[11] == np.arange(256)).all() and int(tr[12].sum()) == 256
    assert int(tr[10].max()) < 256
    # single-table proof with the range-check lookup (lookup.rs:138-198): lookup challenges are the CTL betas
    from zkm_amd.ctl import CtlTable, make_zs
    t = CtlTable()
    cs = T.memory_ctl_data(t)
    zs, ids = make_zs([([cs], 3, 5), ([cs], 7, 11)])
    aux = oracle.ctl_data(t, zs, ids, trace, 13, log_n)
    proof = oracle.prove_ctl(trace, log_n, aux, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=[3, 7])
    assert oracle.num_lookup_columns(T.TABLE_MEMORY) == 4 and proof[3] == 4 + 2
    assert oracle.verify_ctl(proof, 2, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=[3, 7]) == 0
    assert oracle.verify_ctl(proof, 2, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=[3, 8]) != 0
    # a value that breaks read consistency / a range-check value outside the counter range
    for col, row in ((6, 17), (10, 30)):
        bad = trace.copy()
        bad[col * 256 + row] += 1
        if col == 6:
            while bad[2 * 256 + row] != 1 or tr[5, row] != tr[5, row - 1]:   # need a read that follows the same address
                bad[col * 256 + row] -= 1
                row += 1
                bad[col * 256 + row] += 1
        aux = oracle.ctl_data(t, zs, ids, bad, 13, log_n)
        proof = oracle.prove_ctl(bad, log_n, aux, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=[3, 7])
        assert oracle.verify_ctl(proof, 2, t, zs, ids, ncols=13, table_id=T.TABLE_MEMORY, lookup_challenges=[3, 7]) != 0


def test_memory_fill_gaps(oracle):
    # two accesses of one address 10^4 timestamps apart, and two addresses 10^5 words apart: dummy reads bridge both gaps
    ops = np.array([[0, 1, 8, 5, 0, 77], [0, 1, 8, 10005, 1, 77], [0, 1, 100008, 3, 0, 9], [0, 1, 100008, 4, 1, 9]], dtype=np.uint64)
    trace, natural = oracle.memory_trace(ops, 16)
    tr = trace.reshape(13, -1)
    assert int(tr[0].sum()) == 4 and natural > 4 and int(tr[10].max()) < natural


def test_precompile_path_four_tables(oracle):
    tables, ctls, _ = logic_fixtures.build4(oracle)
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0


def test_poseidon_sponge_trace_and_path(oracle):
    tables, ctls, (data, off, meta, inputs, ts, mem_ops) = logic_fixtures.build_poseidon_path(oracle)
    tid, sponge, w, log_n, cs = tables[0]
    tr = sponge.reshape(110, -1)
    # the digest of every operation's last row is the Poseidon sponge hash of its input (poseidon_sponge_stark.rs:143-166)
    row = 0
    for op in range(len(off) - 1):
        msg = bytes(data[int(off[op]):int(off[op + 1])])
        row += len(msg) // 32
        padded = bytearray(msg) + bytearray((len(msg) // 32 + 1) * 32 - len(msg))
        if len(msg) % 32 == 31:
            padded[len(msg)] = 0x81
        else:
            padded[len(msg)] = 1
            padded[-1] = 0x80
        st = [0] * 12
        for b in range(0, len(padded), 32):
            st[:8] = [int.from_bytes(padded[b + 4 * i:b + 4 * i + 4], "little") for i in range(8)]
            st = [int(x) for x in oracle.poseidon_permute(st)]
        assert [int(x) for x in tr[106:110, row]] == st[:4]
        row += 1
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0
    # a sponge row whose output does not continue into the next row breaks a transition constraint
    bad = sponge.copy()
    n = 1 << log_n
    full_rows = np.nonzero(tr[0])[0]
    bad[106 * n + int(full_rows[0])] ^= 1
    aux_tables = [(tid, bad, w, log_n, cs)] + tables[1:]
    proofs, chal, offs = oracle.prove_with_traces(aux_tables, ctls)
    assert oracle.verify_all(aux_tables, ctls, proofs, chal) != 0


def test_sha_extend_path(oracle):
    import hashlib, struct
    tables, ctls, (w16, meta, inputs, ts, ops, mem_ops) = logic_fixtures.build_sha_extend_path(oracle)
    # the 48 output words are the SHA-256 message schedule w[16..63] of the block
    tr = tables[0][1].reshape(76, -1)
    w = [int(x) for x in w16[0]]
    rotr = lambda x, r: ((x >> r) | (x << (32 - r))) & 0xFFFFFFFF
    for i in range(16, 64):
        s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
        s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
        w.append((w[i - 16] + s0 + w[i - 7] + s1) & 0xFFFFFFFF)
        assert sum(int(tr[64 + j, i - 16]) << (8 * j) for j in range(4)) == w[i]
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0
    # a wrong carry flag in the ShaExtend table / a skipped round in the sponge table
    for t_idx, col in ((1, 4), (0, 75)):
        tid, trace, width, log_n, ct = tables[t_idx]
        bad = trace.copy()
        bad[col * (1 << log_n) + 5] += 1
        bt = list(tables)
        bt[t_idx] = (tid, bad, width, log_n, ct)
        proofs, chal, offs = oracle.prove_with_traces(bt, ctls)
        assert oracle.verify_all(bt, ctls, proofs, chal) != 0


def test_sha_compress_path(oracle):
    import hashlib, struct
    tables, ctls, (hx, w, meta, ops, mem_ops) = logic_fixtures.build_sha_compress_path(oracle)
    # with the SHA-256 IV and the schedule of a one-block message, output_hx is the SHA-256 digest
    iv = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
    msg = b"abc"
    block = msg + b"\x80" + b"\0" * (55 - len(msg)) + struct.pack(">Q", 8 * len(msg))
    ws = list(struct.unpack(">16I", block))
    rotr = lambda x, r: ((x >> r) | (x << (32 - r))) & 0xFFFFFFFF
    for i in range(16, 64):
        s0 = rotr(ws[i - 15], 7) ^ rotr(ws[i - 15], 18) ^ (ws[i - 15] >> 3)
        s1 = rotr(ws[i - 2], 17) ^ rotr(ws[i - 2], 19) ^ (ws[i - 2] >> 10)
        ws.append((ws[i - 16] + s0 + ws[i - 7] + s1) & 0xFFFFFFFF)
    tr = oracle.sha_compress_sponge_trace([iv], [ws], np.zeros((1, 8), dtype=np.uint64), 3).reshape(127, 8)
    digest = b"".join(struct.pack(">I", sum(int(tr[64 + 6 * q + j, 0]) << (8 * j) for j in range(4))) for q in range(8))
    assert digest == hashlib.sha256(msg).digest()
    assert oracle.check_ctls(tables, ctls) == 0
    proofs, chal, offs = oracle.prove_with_traces(tables, ctls)
    assert oracle.verify_all(tables, ctls, proofs, chal) == 0
    # a corrupted chained state byte in the compress table / a wrong carry flag in the sponge table
    for t_idx, col, row in ((1, 5, 7), (0, 68, 0)):
        tid, trace, width, log_n, ct = tables[t_idx]
        bad = trace.copy()
        bad[col * (1 << log_n) + row] ^= 1
        bt = list(tables)
        bt[t_idx] = (tid, bad, width, log_n, ct)
        proofs, chal, offs = oracle.prove_with_traces(bt, ctls)
        assert oracle.verify_all(bt, ctls, proofs, chal) != 0


def _arith_proof_ok(oracle, trace, log_n=16):
    from zkm_amd.ctl import CtlTable, make_zs
    t = CtlTable()
    cs = T.arithmetic_ctl_rows(t)          # the column set the CPU table looks up (arithmetic_stark.rs:66-126)
    zs, ids = make_zs([([cs], 3, 5), ([cs], 7, 11)])
    aux = oracle.ctl_data(t, zs, ids, trace, 54, log_n)
    proof = oracle.prove_ctl(trace, log_n, aux, t, zs, ids, ncols=54, table_id=T.TABLE_ARITHMETIC, lookup_challenges=[3, 7])
    return oracle.verify_ctl(proof, 2, t, zs, ids, ncols=54, table_id=T.TABLE_ARITHMETIC, lookup_challenges=[3, 7]) == 0


def test_arithmetic_table_all_operations(oracle):
    """ArithmeticStark (arithmetic_stark.rs:214-276): all 26 operations, the 2^16-row range-check lookup over the 18 shared columns
    (20 lookup columns in front of the CTL columns), and one corrupted cell per constraint family is rejected."""
    from . import arith_fixtures as A
    assert oracle.num_lookup_columns(T.TABLE_ARITHMETIC) == 20
    ops = A.random_ops(1, 200)
    trace = A.generate_trace(ops)
    assert _arith_proof_ok(oracle, trace)
    n = 1 << 16
    # row of the first operation of each kind, then flip a witness cell that only that operation's constraints see
    first_row, r = {}, 0
    for op, a, b in ops:
        first_row.setdefault(op, r)
        r += 2 if op in (A.IS_DIV, A.IS_DIVU, A.IS_SRL, A.IS_SRLV, A.IS_SRA, A.IS_SRAV) else 1
    cases = [(A.IS_ADD, A.OUT, 0), (A.IS_SUB, A.AUX0, 0), (A.IS_MUL, A.OUT + 1, 0), (A.IS_MULT, A.OUT_HI, 0), (A.IS_MULTU, A.OUT_LO, 0),
             (A.IS_SLT, A.OUT, 0), (A.IS_SLTU, A.AUX0, 0), (A.IS_LUI, A.OUT + 1, 0), (A.IS_DIVU, A.OUT_HI, 0), (A.IS_DIV, A.QUOT_ABS, 0),
             (A.IS_DIV, A.NV_DENOM_IS_ZERO + 5, 1), (A.IS_SLL, A.OUT, 0), (A.IS_SRL, A.OUT, 0), (A.IS_SRA, A.OUT + 1, 0), (A.IS_SRAV, A.AUX_EXTRA + 3, 1),
             (A.IS_MFLO, A.OUT, 0)]
    for op, col, second in cases[::3]:       # every third case here (each is a full 2^16-row proof); the rest run in the GPU parity test
        bad = trace.copy()
        idx = col * n + first_row[op] + second
        bad[idx] = (int(bad[idx]) + 1) & 0xFFFF if col < A.RANGE_COUNTER else int(bad[idx]) + 1
        tr = bad.reshape(54, n)
        tr[A.RC_FREQ] = 0
        tr[A.RC_FREQ] += np.bincount(tr[26:44].reshape(-1).astype(np.int64), minlength=n)[:n].astype(np.uint64)   # keep the lookup consistent
        assert not _arith_proof_ok(oracle, bad), "corruption of op %d col %d accepted" % (op, col)
    # a value outside the 16-bit range fails the lookup even when no operation flag is set on its row
    bad = trace.copy()
    bad[30 * n + 40000] = 1 << 16
    assert not _arith_proof_ok(oracle, bad)


# ---- known-answer vectors held by the reference's own SHA table tests
REF_SHA_W = [
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 34013193, 67559435, 1711661200, 3020350282, 1447362251, 3118632270,
    4004188394, 690615167, 6070360, 1105370215, 2385558114, 2348232513, 507799627, 2098764358, 5845374, 823657968, 2969863067,
    3903496557, 4274682881, 2059629362, 1849247231, 2656047431, 835162919, 2096647516, 2259195856, 1779072524, 3152121987,
    4210324067, 1557957044, 376930560, 982142628, 3926566666, 4164334963, 789545383, 1028256580, 2867933222, 3843938318,
    1135234440, 390334875, 2025924737, 3318322046, 3436065867, 652746999, 4261492214, 2543173532, 3334668051, 3166416553,
    634956631]
REF_SHA_H = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
REF_SHA_OUTPUT_HX = [3592665057, 2164530888, 1223339564, 3041196771, 2006723467, 2963045520, 3851824201, 3453903005]


def _le4(tr, col, row):
    return sum(int(tr[col + j][row]) << (8 * j) for j in range(4))


def test_sha_compress_reference_vector(oracle):
    """sha_compress/sha_compress_stark.rs:933-965 test_generation: round 0 on the SHA-256 IV with w_0 = 0, K_0:
    temp1_add_temp2.value == 4228417613 and d_add_temp1.value == 2563236514 (columns 140.. and 134.. of columns.rs:41-43)."""
    tr = oracle.sha_compress_trace([REF_SHA_H], [REF_SHA_W], np.zeros((1, 8), dtype=np.uint64), 7).reshape(224, 128)
    assert _le4(tr, 140, 0) == 4228417613
    assert _le4(tr, 134, 0) == 2563236514
    assert _le4(tr, 36, 0) == 0 and _le4(tr, 40, 0) == 0x428a2f98  # w_0, K_0 (constants.rs SHA_COMPRESS_K[0])


def test_sha_compress_sponge_reference_vector(oracle):
    """sha_compress_sponge/sha_compress_sponge_stark.rs:380-448 test_generation: the eight output_hx words of one
    compression of W on the IV (output_hx[q].value = columns 64 + 6 q .. + 3, columns.rs:6-12)."""
    tr = oracle.sha_compress_sponge_trace([REF_SHA_H], [REF_SHA_W], np.zeros((1, 8), dtype=np.uint64), 3).reshape(127, 8)
    assert [_le4(tr, 64 + 6 * q, 0) for q in range(8)] == REF_SHA_OUTPUT_HX
    # the reference's W is itself a message schedule: w[16..] extend w[0..15] (sha_extend) -- ties the two tables together
    sp, used = oracle.sha_extend_sponge_trace([REF_SHA_W[:16]], np.zeros((1, 4), dtype=np.uint64), 6)
    sp = sp.reshape(76, 64)
    assert used == 48 and [_le4(sp, 64, r) for r in range(48)] == REF_SHA_W[16:]


def test_sha_extend_reference_vector(oracle):
    """sha_extend/sha_extend_stark.rs:443-476 test_correction: inputs w[i-15], w[i-2], w[i-16], w[i-7] = 0, 1, 2, 3."""
    inp = np.array([0, 1, 2, 3], dtype="<u4").view(np.uint8)
    tr = oracle.sha_extend_trace(inp, [0], 2).reshape(78, 4)
    rotr = lambda x, r: ((x >> r) | (x << (32 - r))) & 0xFFFFFFFF
    s0 = rotr(0, 7) ^ rotr(0, 18) ^ (0 >> 3)
    s1 = rotr(1, 17) ^ rotr(1, 19) ^ (1 >> 10)
    assert _le4(tr, 0, 0) == (s1 + 2 + s0 + 3) & 0xFFFFFFFF == 40965
    # and against the reference's W: w[16] = extend(w[1], w[14], w[0], w[9])
    inp = np.array([REF_SHA_W[1], REF_SHA_W[14], REF_SHA_W[0], REF_SHA_W[9]], dtype="<u4").view(np.uint8)
    assert _le4(oracle.sha_extend_trace(inp, [0], 2).reshape(78, 4), 0, 0) == REF_SHA_W[16] == 34013193


# ---- the reference's per-table low-degree test (every *_stark.rs `test_stark_degree` -> stark_testing.rs:21-70)
@pytest.mark.parametrize("table_id", list(range(12)))
def test_stark_degree(oracle, table_id):
    """Random witness polynomials of degree < 32, extended by log2_ceil(constraint_degree + 1) = 2 bits on the plain subgroup; the
    alpha-combined constraint polynomial must have degree <= 32 * 3 - 1 (constraint_degree() is 3 for every table: stark.rs /
    each *_stark.rs).  A constraint transcribed with one factor too many (in the oracle or, through the parity tests, in
    constraints_dev.h) shows up here as a non-zero top coefficient; the witness satisfies nothing, so nothing cancels by accident."""
    from zkm_amd import tables as T
    W, log_w, rate_bits, degree = T.WIDTH[table_id], 5, 2, 3
    size = (1 << log_w) << rate_bits
    rng = np.random.default_rng(1000 + table_id)
    coeffs = np.zeros((W, size), dtype=np.uint64)
    coeffs[:, :1 << log_w] = rng.integers(0, P, size=(W, 1 << log_w), dtype=np.uint64)   # random_low_degree_matrix (:141-149)
    rows = oracle.ntt(coeffs, log_w + rate_bits).reshape(W, size)
    alpha = int(rng.integers(1, P, dtype=np.uint64))
    evals = oracle.constraint_evals(table_id, rows, log_w, rate_bits, alpha)
    assert evals.any(), "a random witness must violate the constraints"
    poly = oracle.ntt(evals, log_w + rate_bits, inverse=True)
    nz = np.nonzero(poly)[0]
    assert nz.max() <= (1 << log_w) * degree - 1, "constraint polynomial of table %d has degree %d" % (table_id, nz.max())
    # ... and the evaluation is not vacuous: products of two columns (degree 2 * 31) occur in every table; the sponge tables stop at
    # 2 * 31 + 1 (quadratic transition constraints times z_last), the others have cubic terms
    assert nz.max() >= 2 * ((1 << log_w) - 1)
