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
