"""KeccakSponge + Logic instance: the KeccakSponge -> Logic part of all_stark::ctl_logic (all_stark.rs:340-355).
Every non-padding sponge row looks up 34 XORs (original_rate_u32 ^ block_u32 = xored_rate_u32) in the Logic table."""
import numpy as np

from zkm_amd import tables as T
from zkm_amd.ctl import CtlTable

from .sponge_fixtures import ops_for_rows


def logic_ops_from_sponge(sponge_trace, log_n, rows, extra_seed=3, extra=5):
    """The XOR operations the sponge rows request, plus a few unrelated AND/OR/NOR rows with filter... no: every Logic
    row with a flag set is looked, so the Logic table holds exactly the requested XORs (shuffled)."""
    n = 1 << log_n
    tr = sponge_trace.reshape(T.WIDTH[T.TABLE_KECCAK_SPONGE], n)
    ops = []
    for r in range(rows):
        for i in range(T.NUM_LOGIC_CTLS):
            a = int(tr[T.KS_ORIG_RATE + i, r])
            b = sum(int(tr[T.KS_BLOCK + 4 * i + j, r]) << (8 * j) for j in range(4))
            ops.append((T.OP_XOR, a, b))
    ops = np.array(ops, dtype=np.uint32)
    np.random.default_rng(extra_seed).shuffle(ops, axis=0)
    return ops


def build(oracle, log_sponge=4, seed=21):
    data, off, meta, rows, nops = ops_for_rows(seed, (1 << log_sponge) - 2)
    sponge, _ = oracle.keccak_sponge_trace(data, off, meta, log_sponge)
    ops = logic_ops_from_sponge(sponge, log_sponge, rows)
    log_logic = max(3, int(np.ceil(np.log2(len(ops)))))
    logic = oracle.logic_trace(ops, log_logic)
    cs, cl = CtlTable(), CtlTable()
    looking, looked = T.ctl_logic_keccak_sponge(0, 1, cs, cl)
    tables = [(T.TABLE_KECCAK_SPONGE, sponge, 470, log_sponge, cs), (T.TABLE_LOGIC, logic, 69, log_logic, cl)]
    return tables, [(looking, looked)], ops


def keccak_inputs_from_sponge(sponge_trace, log_n, rows):
    """The permutation inputs (xored rate ++ original capacity, as u64 words) and timestamps the sponge rows request."""
    n = 1 << log_n
    tr = sponge_trace.reshape(T.WIDTH[T.TABLE_KECCAK_SPONGE], n)
    u32 = np.concatenate([tr[T.KS_XORED:T.KS_XORED + 34, :rows], tr[T.KS_ORIG_CAP:T.KS_ORIG_CAP + 16, :rows]]).T   # rows x 50
    inputs = (u32[:, 0::2] | (u32[:, 1::2] << np.uint64(32))).astype(np.uint64)
    return np.ascontiguousarray(inputs), tr[T.KS_TIMESTAMP, :rows].copy()


def build3(oracle, log_sponge=3, seed=22):
    """KeccakSponge + Keccak + Logic with the three cross-table lookups that link them in the reference
    (all_stark.rs:214-240, 340-355)."""
    data, off, meta, rows, nops = ops_for_rows(seed, (1 << log_sponge) - 1)
    sponge, _ = oracle.keccak_sponge_trace(data, off, meta, log_sponge)
    ops = logic_ops_from_sponge(sponge, log_sponge, rows)
    log_logic = max(3, int(np.ceil(np.log2(len(ops)))))
    logic = oracle.logic_trace(ops, log_logic)
    inputs, ts = keccak_inputs_from_sponge(sponge, log_sponge, rows)
    log_keccak = int(np.ceil(np.log2(24 * rows)))
    keccak = oracle.keccak_trace(inputs, ts, log_keccak)
    cs, cl, ck = CtlTable(), CtlTable(), CtlTable()
    tables = [(T.TABLE_KECCAK_SPONGE, sponge, 470, log_sponge, cs), (T.TABLE_KECCAK, keccak, 2431, log_keccak, ck),
              (T.TABLE_LOGIC, logic, 69, log_logic, cl)]
    ctls = [T.ctl_keccak_inputs(0, 1, cs, ck), T.ctl_keccak_outputs(0, 1, cs, ck), T.ctl_logic_keccak_sponge(0, 2, cs, cl)]
    return tables, ctls, (ops, inputs, ts)


def memory_ops_from_poseidon_sponge(trace, log_n, rows):
    """Memory reads requested by PoseidonSponge rows (poseidon_sponge_stark.rs:63-104, :120-134)."""
    n = 1 << log_n
    tr = trace.reshape(T.WIDTH[T.TABLE_POSEIDON_SPONGE], n)
    ops = []
    for r in range(rows):
        full = int(tr[T.PS_FULL, r])
        rem = 32 if full else int(np.argmax(tr[T.PS_FINAL_LEN:T.PS_FINAL_LEN + 32, r]))
        for i in range(rem):
            s = (i // 4) * 4
            b = [int(tr[T.PS_BLOCK + s + j, r]) for j in range(4)]
            ops.append((int(tr[T.PS_CONTEXT, r]), int(tr[T.PS_SEGMENT, r]), int(tr[T.PS_VIRT + i // 4, r]), int(tr[T.PS_TIMESTAMP, r]), 1,
                        (b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]))
    return np.array(ops, dtype=np.uint64).reshape(-1, 6)


def poseidon_inputs_from_sponge(trace, log_n, rows):
    n = 1 << log_n
    tr = trace.reshape(T.WIDTH[T.TABLE_POSEIDON_SPONGE], n)
    inputs = np.concatenate([tr[T.PS_NEW_RATE:T.PS_NEW_RATE + 8, :rows], tr[T.PS_ORIG_CAP:T.PS_ORIG_CAP + 4, :rows]]).T
    return np.ascontiguousarray(inputs), tr[T.PS_TIMESTAMP, :rows].copy()


def poseidon_sponge_ops(seed, target_rows, max_len=200):
    rng = np.random.default_rng(seed)
    nops = max(1, target_rows // 5)
    while True:
        lens = rng.integers(1, max_len, nops)
        rows = int(np.sum(lens // 32 + 1))
        if rows <= target_rows:
            break
        nops = max(1, int(nops * 0.9))
    off = np.zeros(nops + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    data = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
    meta = np.zeros((nops, 4), dtype=np.uint64)
    meta[:, 1] = 2
    meta[:, 2] = (1 << 20) + np.arange(nops) * 64      # disjoint word ranges, apart from the Keccak fixtures'
    meta[:, 3] = np.arange(nops) * 5 + 3
    return data, off, meta.reshape(-1), rows, nops


def build_poseidon_path(oracle, log_sponge=4, seed=41, ts=None):
    """Memory + PoseidonSponge + Poseidon with the lookups the reference defines among them
    (all_stark.rs:169-195, 487-493)."""
    data, off, meta, rows, nops = poseidon_sponge_ops(seed, (1 << log_sponge) - 1)
    if ts is not None:
        meta = meta.reshape(-1, 4).copy()
        meta[:, 3] = ts(nops)
        meta = meta.reshape(-1)
    sponge, used = oracle.poseidon_sponge_trace(data, off, meta, log_sponge)
    assert used == rows
    inputs, ts = poseidon_inputs_from_sponge(sponge, log_sponge, rows)
    log_pos = max(3, int(np.ceil(np.log2(rows))))
    poseidon = oracle.poseidon_trace_inputs(inputs, ts, log_pos)
    mem_ops = memory_ops_from_poseidon_sponge(sponge, log_sponge, rows)
    log_mem = int(np.ceil(np.log2(len(mem_ops)))) + 1
    memory, natural = oracle.memory_trace(mem_ops, log_mem)
    if natural < (1 << log_mem):
        log_mem -= 1
        memory, natural = oracle.memory_trace(mem_ops, log_mem)
    cs, cp, cm = CtlTable(), CtlTable(), CtlTable()
    tables = [(T.TABLE_POSEIDON_SPONGE, sponge, 110, log_sponge, cs), (T.TABLE_POSEIDON, poseidon, 262, log_pos, cp),
              (T.TABLE_MEMORY, memory, 13, log_mem, cm)]
    ctls = [T.ctl_poseidon_inputs(0, 1, cs, cp), T.ctl_poseidon_outputs(0, 1, cs, cp),
            (T.memory_lookers_poseidon_sponge(0, cs), (2, T.memory_ctl_data(cm)))]
    return tables, ctls, (data, off, meta, inputs, ts, mem_ops)


def memory_ops_from_sponge(sponge_trace, log_n, rows):
    """The memory reads the sponge rows request (keccak_sponge_stark.rs:91-124, :174-186): one per input byte, each
    carrying the big-endian word that holds the byte.  Returns nops x 6 (context, segment, virt, timestamp, is_read, value)."""
    n = 1 << log_n
    tr = sponge_trace.reshape(T.WIDTH[T.TABLE_KECCAK_SPONGE], n)
    ops = []
    for r in range(rows):
        full = int(tr[T.KS_FULL, r])
        rem = 136 if full else int(np.argmax(tr[T.KS_FINAL_LEN:T.KS_FINAL_LEN + 136, r]))
        for i in range(rem):
            s = (i // 4) * 4
            b = [int(tr[T.KS_BLOCK + s + j, r]) for j in range(4)]
            ops.append((int(tr[T.KS_CONTEXT, r]), int(tr[T.KS_SEGMENT, r]), int(tr[T.KS_VIRT + i // 4, r]), int(tr[T.KS_TIMESTAMP, r]), 1,
                        (b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]))
    return np.array(ops, dtype=np.uint64).reshape(-1, 6)


def build4(oracle, log_sponge=3, seed=23, ts=None):
    """Memory + KeccakSponge + Keccak + Logic: the Keccak precompile's data path with every cross-table lookup the
    reference defines among these tables (all_stark.rs:214-240, 340-355, 479-542)."""
    data, off, meta, rows, nops = ops_for_rows(seed, (1 << log_sponge) - 1)
    meta = meta.reshape(-1, 4).copy()
    meta[:, 2] = np.arange(nops) * 512          # disjoint word ranges: reads of one address always see one value
    if ts is not None:
        meta[:, 3] = ts(nops)
        meta[:, 0], meta[:, 1] = 0, 1               # keep clear of the CPU's register file / shift table / code segments
    meta = meta.reshape(-1)
    sponge, _ = oracle.keccak_sponge_trace(data, off, meta, log_sponge)
    ops = logic_ops_from_sponge(sponge, log_sponge, rows)
    log_logic = max(3, int(np.ceil(np.log2(len(ops)))))
    logic = oracle.logic_trace(ops, log_logic)
    inputs, ts = keccak_inputs_from_sponge(sponge, log_sponge, rows)
    log_keccak = int(np.ceil(np.log2(24 * rows)))
    keccak = oracle.keccak_trace(inputs, ts, log_keccak)
    mem_ops = memory_ops_from_sponge(sponge, log_sponge, rows)
    log_mem = int(np.ceil(np.log2(len(mem_ops)))) + 1
    memory, natural = oracle.memory_trace(mem_ops, log_mem)
    if natural < (1 << log_mem):
        log_mem -= 1
        memory, natural = oracle.memory_trace(mem_ops, log_mem)
    cs, cl, ck, cm = CtlTable(), CtlTable(), CtlTable(), CtlTable()
    tables = [(T.TABLE_KECCAK_SPONGE, sponge, 470, log_sponge, cs), (T.TABLE_KECCAK, keccak, 2431, log_keccak, ck),
              (T.TABLE_LOGIC, logic, 69, log_logic, cl), (T.TABLE_MEMORY, memory, 13, log_mem, cm)]
    ctls = [T.ctl_keccak_inputs(0, 1, cs, ck), T.ctl_keccak_outputs(0, 1, cs, ck), T.ctl_logic_keccak_sponge(0, 2, cs, cl),
            T.ctl_memory_keccak_sponge(0, 3, cs, cm)]
    return tables, ctls, (ops, inputs, ts, mem_ops)


def build_sha_extend_path(oracle, nblocks=1, seed=51, ts=None):
    """Memory + ShaExtendSponge + ShaExtend + Logic: SHA-256 message schedules with the lookups the reference defines among
    these tables (all_stark.rs:256-282, 356-385, 503-509)."""
    rng = np.random.default_rng(seed)
    w16 = rng.integers(0, 1 << 32, (nblocks, 16), dtype=np.uint64).astype(np.uint32)
    meta = np.zeros((nblocks, 4), dtype=np.uint64)
    meta[:, 1] = 1
    meta[:, 2] = (1 << 22) + np.arange(nblocks) * 1024
    meta[:, 3] = 7 + np.arange(nblocks) * 2000 if ts is None else ts(nblocks)
    log_s = int(np.ceil(np.log2(48 * nblocks + 1)))
    sponge, used = oracle.sha_extend_sponge_trace(w16, meta, log_s)
    n = 1 << log_s
    tr = sponge.reshape(76, n)
    rows = 48 * nblocks
    inputs = tr[T.SES_W15:T.SES_W15 + 16, :rows].T.astype(np.uint8)
    ts = tr[T.SES_TIMESTAMP, :rows].copy()
    extend = oracle.sha_extend_trace(inputs, ts, log_s)
    ex = extend.reshape(78, n)
    le = lambda c, r: sum(int(ex[c + j, r]) << (8 * j) for j in range(4))
    ops, mem = [], []
    for r in range(rows):
        for a, b in ((T.SE_RR7, T.SE_RR18), (T.SE_S0_INTER, T.SE_RS3), (T.SE_RR17, T.SE_RR19), (T.SE_S1_INTER, T.SE_RS10)):
            ops.append((T.OP_XOR, le(a, r), le(b, r)))
        for q in range(4):
            val = sum(int(tr[T.SES_W15 + 4 * q + j, r]) << (8 * j) for j in range(4))
            mem += [(int(tr[T.SES_CONTEXT, r]), int(tr[T.SES_SEGMENT, r]), int(tr[T.SES_IN_VIRT + q, r]), int(tr[T.SES_TIMESTAMP, r]), 1, val)] * 4
    ops = np.array(ops, dtype=np.uint32)
    np.random.default_rng(seed + 1).shuffle(ops, axis=0)
    log_logic = int(np.ceil(np.log2(len(ops))))
    logic = oracle.logic_trace(ops, log_logic)
    mem_ops = np.array(mem, dtype=np.uint64).reshape(-1, 6)
    log_mem = int(np.ceil(np.log2(len(mem_ops)))) + 1
    memory, natural = oracle.memory_trace(mem_ops, log_mem)
    if natural < (1 << log_mem):
        log_mem -= 1
        memory, natural = oracle.memory_trace(mem_ops, log_mem)
    cs, ce, cl, cm = CtlTable(), CtlTable(), CtlTable(), CtlTable()
    tables = [(T.TABLE_SHA_EXTEND_SPONGE, sponge, 76, log_s, cs), (T.TABLE_SHA_EXTEND, extend, 78, log_s, ce),
              (T.TABLE_LOGIC, logic, 69, log_logic, cl), (T.TABLE_MEMORY, memory, 13, log_mem, cm)]
    ctls = [T.ctl_sha_extend_inputs(0, 1, cs, ce), T.ctl_sha_extend_outputs(0, 1, cs, ce),
            (T.logic_lookers_sha_extend(1, ce), (2, T.logic_ctl_data(cl))),
            (T.memory_lookers_sha_extend_sponge(0, cs), (3, T.memory_ctl_data(cm)))]
    return tables, ctls, (w16, meta, inputs, ts, ops, mem_ops)


def build_sha_compress_path(oracle, ncomp=1, seed=61, ts=None):
    """Memory + ShaCompressSponge + ShaCompress + Logic: SHA-256 compressions with the lookups the reference defines among
    these tables (all_stark.rs:298-324, 387-470, 511-525)."""
    rng = np.random.default_rng(seed)
    hx = rng.integers(0, 1 << 32, (ncomp, 8), dtype=np.uint64).astype(np.uint32)
    w = rng.integers(0, 1 << 32, (ncomp, 64), dtype=np.uint64).astype(np.uint32)
    meta = np.zeros((ncomp, 8), dtype=np.uint64)
    meta[:, 1] = 0                                        # hx: context 0, segment Code (witness/operation.rs:1318)
    meta[:, 2] = (1 << 23) + np.arange(ncomp) * 2048      # hx address
    meta[:, 3] = 30 + np.arange(ncomp) * 10 if ts is None else ts(ncomp)   # timestamp
    meta[:, 4] = (1 << 23) + np.arange(ncomp) * 2048 + 512  # w address
    log_c = int(np.ceil(np.log2(65 * ncomp + 1)))
    log_s = max(3, int(np.ceil(np.log2(ncomp + 1))))
    compress = oracle.sha_compress_trace(hx, w, meta, log_c)
    sponge = oracle.sha_compress_sponge_trace(hx, w, meta, log_s)
    n = 1 << log_c
    tr = compress.reshape(224, n)
    le = lambda c, r: sum(int(tr[c + j, r]) << (8 * j) for j in range(4))
    ops, mem = [], []
    for e in range(ncomp):
        for q in range(8):
            mem += [(int(meta[e, 0]), int(meta[e, 1]), int(meta[e, 2]) + 4 * q, int(meta[e, 3]), 1, int(hx[e, q]))] * 4
        for rd in range(64):
            r = 65 * e + rd
            for opcode, in0, in1, res in T.SHA_COMPRESS_LOGIC:
                ops.append((T.OP_XOR if opcode == (0b100110 << 6) else T.OP_AND, le(in0, r), le(in1, r)))
            mem += [(int(meta[e, 6]), int(meta[e, 5]), int(meta[e, 4]) + 4 * rd, int(meta[e, 3]), 1, int(w[e, rd]))] * 4
    ops = np.array(ops, dtype=np.uint32)
    np.random.default_rng(seed + 1).shuffle(ops, axis=0)
    log_logic = int(np.ceil(np.log2(len(ops))))
    logic = oracle.logic_trace(ops, log_logic)
    mem_ops = np.array(mem, dtype=np.uint64).reshape(-1, 6)
    log_mem = int(np.ceil(np.log2(len(mem_ops)))) + 1
    memory, natural = oracle.memory_trace(mem_ops, log_mem)
    if natural < (1 << log_mem):
        log_mem -= 1
        memory, natural = oracle.memory_trace(mem_ops, log_mem)
    cs, cc, cl, cm = CtlTable(), CtlTable(), CtlTable(), CtlTable()
    tables = [(T.TABLE_SHA_COMPRESS_SPONGE, sponge, 127, log_s, cs), (T.TABLE_SHA_COMPRESS, compress, 224, log_c, cc),
              (T.TABLE_LOGIC, logic, 69, log_logic, cl), (T.TABLE_MEMORY, memory, 13, log_mem, cm)]
    ctls = [T.ctl_sha_compress_inputs(0, 1, cs, cc), T.ctl_sha_compress_outputs(0, 1, cs, cc),
            (T.logic_lookers_sha_compress(1, cc), (2, T.logic_ctl_data(cl))),
            (T.memory_lookers_sha_compress_sponge(0, cs) + T.memory_lookers_sha_compress(1, cc), (3, T.memory_ctl_data(cm)))]
    return tables, ctls, (hx, w, meta, ops, mem_ops)
