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


def build4(oracle, log_sponge=3, seed=23):
    """Memory + KeccakSponge + Keccak + Logic: the Keccak precompile's data path with every cross-table lookup the
    reference defines among these tables (all_stark.rs:214-240, 340-355, 479-542)."""
    data, off, meta, rows, nops = ops_for_rows(seed, (1 << log_sponge) - 1)
    meta = meta.reshape(-1, 4).copy()
    meta[:, 2] = np.arange(nops) * 512          # disjoint word ranges: reads of one address always see one value
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
