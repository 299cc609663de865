"""Table descriptions for the STARK tables that have HIP constraint kernels: table ids, widths and the cross-table
lookup column sets the reference defines for them.  Pure host-side marshalling (descriptor arrays for
include/zkm_hip.h); mirrors
  - Table ids / widths: zkm_hip.h ZKM_TABLE_*; prover/src/poseidon/columns.rs, logic.rs:25-50, keccak_sponge/columns.rs:19-70
  - logic::ctl_data / ctl_filter                      logic.rs:52-76
  - keccak_sponge::ctl_looking_logic(i) / _filter     keccak_sponge_stark.rs:126-194
  - all_stark::ctl_logic (KeccakSponge lookers)       all_stark.rs:340-355
"""
from .ctl import CtlTable

TABLE_POSEIDON, TABLE_LOGIC, TABLE_KECCAK_SPONGE = 0, 1, 2
WIDTH = {TABLE_POSEIDON: 262, TABLE_LOGIC: 69, TABLE_KECCAK_SPONGE: 470}

# LogicStark columns (logic.rs:25-50)
LOGIC_IS_AND, LOGIC_IS_OR, LOGIC_IS_XOR, LOGIC_IS_NOR = 0, 1, 2, 3
LOGIC_INPUT0, LOGIC_INPUT1, LOGIC_RESULT = 4, 36, 68
OP_AND, OP_OR, OP_XOR, OP_NOR = 0, 1, 2, 3

# KeccakSpongeStark columns (keccak_sponge/columns.rs:19-70)
KS_FULL, KS_FINAL_LEN, KS_ORIG_RATE, KS_BLOCK, KS_XORED = 0, 40, 176, 226, 362
KECCAK_RATE_BYTES, KECCAK_RATE_U32S = 136, 34
NUM_LOGIC_CTLS = KECCAK_RATE_BYTES // 4   # num_logic_ctls(): U8S_PER_CTL = 4, U32S_PER_CTL = 1


def logic_ctl_data(t: CtlTable):
    """logic::ctl_data() with logic::ctl_filter() as one column set of `t`."""
    first = t.column(local=[(LOGIC_IS_AND, 0b100100 << 6), (LOGIC_IS_OR, 0b100101 << 6), (LOGIC_IS_XOR, 0b100110 << 6),
                            (LOGIC_IS_NOR, 0b100111 << 6)])
    t.le_bits(range(LOGIC_INPUT0, LOGIC_INPUT0 + 32))
    t.le_bits(range(LOGIC_INPUT1, LOGIC_INPUT1 + 32))
    t.single(LOGIC_RESULT)
    f = t.sum([LOGIC_IS_AND, LOGIC_IS_OR, LOGIC_IS_XOR, LOGIC_IS_NOR])
    return t.colset(range(first, first + 4), filter_constants=[f])


def keccak_sponge_looking_logic(t: CtlTable, i):
    """keccak_sponge_stark::ctl_looking_logic(i) with ctl_looking_logic_filter()."""
    assert 0 <= i < NUM_LOGIC_CTLS
    first = t.constant(0b100110 << 6)
    t.single(KS_ORIG_RATE + i)
    t.le_bytes(range(KS_BLOCK + 4 * i, KS_BLOCK + 4 * i + 4))
    t.single(KS_XORED + i)
    f = t.sum([KS_FULL] + list(range(KS_FINAL_LEN, KS_FINAL_LEN + KECCAK_RATE_BYTES)))
    return t.colset(range(first, first + 4), filter_constants=[f])


def ctl_logic_keccak_sponge(sponge_index, logic_index, sponge_ctl: CtlTable, logic_ctl: CtlTable):
    """The KeccakSponge -> Logic part of all_stark::ctl_logic(): 34 looking column sets, one looked."""
    looking = [(sponge_index, keccak_sponge_looking_logic(sponge_ctl, i)) for i in range(NUM_LOGIC_CTLS)]
    return looking, (logic_index, logic_ctl_data(logic_ctl))
