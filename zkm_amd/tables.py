"""Table descriptions for the STARK tables that have HIP constraint kernels: table ids, widths and the cross-table
lookup column sets the reference defines for them.  Pure host-side marshalling (descriptor arrays for
include/zkm_hip.h); mirrors
  - Table ids / widths: zkm_hip.h ZKM_TABLE_*; prover/src/poseidon/columns.rs, logic.rs:25-50, keccak_sponge/columns.rs:19-70
  - logic::ctl_data / ctl_filter                      logic.rs:52-76
  - keccak_sponge::ctl_looking_logic(i) / _filter     keccak_sponge_stark.rs:126-194
  - all_stark::ctl_logic (KeccakSponge lookers)       all_stark.rs:340-355
"""
from .ctl import CtlTable

TABLE_POSEIDON, TABLE_LOGIC, TABLE_KECCAK_SPONGE, TABLE_KECCAK, TABLE_MEMORY, TABLE_POSEIDON_SPONGE = 0, 1, 2, 3, 4, 5
TABLE_SHA_EXTEND, TABLE_SHA_EXTEND_SPONGE, TABLE_SHA_COMPRESS, TABLE_SHA_COMPRESS_SPONGE, TABLE_ARITHMETIC, TABLE_CPU = 6, 7, 8, 9, 10, 11
WIDTH = {TABLE_POSEIDON: 262, TABLE_LOGIC: 69, TABLE_KECCAK_SPONGE: 470, TABLE_KECCAK: 2431, TABLE_MEMORY: 13, TABLE_POSEIDON_SPONGE: 110,
         TABLE_SHA_EXTEND: 78, TABLE_SHA_EXTEND_SPONGE: 76, TABLE_SHA_COMPRESS: 224, TABLE_SHA_COMPRESS_SPONGE: 127, TABLE_ARITHMETIC: 54, TABLE_CPU: 259}

# LogicStark columns (logic.rs:25-50)
LOGIC_IS_AND, LOGIC_IS_OR, LOGIC_IS_XOR, LOGIC_IS_NOR = 0, 1, 2, 3
LOGIC_INPUT0, LOGIC_INPUT1, LOGIC_RESULT = 4, 36, 68
OP_AND, OP_OR, OP_XOR, OP_NOR = 0, 1, 2, 3

# KeccakSpongeStark columns (keccak_sponge/columns.rs:19-70)
KS_FULL, KS_FINAL_LEN, KS_ORIG_RATE, KS_BLOCK, KS_XORED = 0, 40, 176, 226, 362
KS_CONTEXT, KS_SEGMENT, KS_VIRT = 1, 2, 3
KS_TIMESTAMP, KS_ORIG_CAP, KS_PARTIAL, KS_DIGEST = 37, 210, 396, 438

# PoseidonStark columns (poseidon/columns.rs:3-54) and PoseidonSpongeStark columns (poseidon_sponge/columns.rs:17-66)
POS_FILTER, POS_IN, POS_OUT, POS_TIMESTAMP = 0, 1, 13, 25
PS_FULL, PS_CONTEXT, PS_SEGMENT, PS_VIRT, PS_TIMESTAMP, PS_LEN, PS_ABSORBED, PS_FINAL_LEN = 0, 1, 2, 3, 11, 12, 13, 14
PS_ORIG_RATE, PS_ORIG_CAP, PS_BLOCK, PS_NEW_RATE, PS_PARTIAL, PS_DIGEST = 46, 54, 58, 90, 98, 106
POSEIDON_RATE_BYTES = 32

# ShaExtendStark (sha_extend/columns.rs:8-36) and ShaExtendSpongeStark (sha_extend_sponge/columns.rs:7-33) columns
SE_W_I, SE_W15, SE_W2, SE_W16, SE_W7, SE_S0_INTER, SE_S0, SE_S1_INTER, SE_S1 = 0, 8, 12, 16, 20, 24, 28, 32, 36
SE_RR7, SE_RR18, SE_RR17, SE_RR19, SE_RS10, SE_RS3, SE_TIMESTAMP, SE_IS_REAL = 40, 46, 52, 58, 64, 70, 76, 77
SES_ROUND, SES_W15, SES_W2, SES_W16, SES_W7, SES_W_I, SES_IN_VIRT, SES_OUT_VIRT, SES_CONTEXT, SES_SEGMENT, SES_TIMESTAMP = (
    0, 48, 52, 56, 60, 64, 68, 72, 73, 74, 75)

# ShaCompressStark (sha_compress/columns.rs:9-55) and ShaCompressSpongeStark (sha_compress_sponge/columns.rs:7-27) columns
SC_STATE, SC_E_NOT, SC_W_I, SC_K_I, SC_S1_INTER, SC_S1, SC_E_AND_F, SC_ENOT_AND_G, SC_CH = 0, 32, 36, 40, 44, 48, 52, 56, 60
SC_S0_INTER, SC_S0, SC_A_AND_B, SC_A_AND_C, SC_B_AND_C, SC_MAJ_INTER, SC_MAJ = 64, 68, 72, 76, 80, 84, 88
SC_E_RR6, SC_E_RR11, SC_E_RR25, SC_A_RR2, SC_A_RR13, SC_A_RR22 = 92, 98, 104, 110, 116, 122
SC_TIMESTAMP, SC_SEGMENT, SC_CONTEXT, SC_W_I_VIRT, SC_ROUND = 146, 147, 148, 149, 159
SCS_HX, SCS_OUT_STATE, SCS_OUT_HX, SCS_HX_VIRT, SCS_W_VIRT, SCS_TIMESTAMP, SCS_CONTEXT, SCS_SEGMENT = 0, 32, 64, 112, 120, 121, 122, 123
SCS_W_SEGMENT, SCS_W_CONTEXT, SCS_IS_REAL = 124, 125, 126

# MemoryStark columns (memory/columns.rs)
MEM_FILTER, MEM_TIMESTAMP, MEM_IS_READ, MEM_CONTEXT, MEM_SEGMENT, MEM_VIRTUAL, MEM_VALUE = 0, 1, 2, 3, 4, 5, 6
KECCAK_RATE_BYTES, KECCAK_RATE_U32S = 136, 34

# KeccakStark registers (keccak/columns.rs:7-134)
KK_ROUNDS, KK_TIMESTAMP = 24, 24


def kk_reg_a(x, y):
    return 25 + (x * 5 + y) * 2


def kk_reg_a_prime_prime_prime(x, y):
    return 2429 if (x, y) == (0, 0) else 2315 + x * 10 + y * 2


def kk_reg_input_limb(i):
    """reg_input_limb(i): limb i of the y-major 5x5 input (columns.rs:15-27)."""
    y, x = divmod(i // 2, 5)
    return kk_reg_a(x, y) + i % 2


def kk_reg_output_limb(i):
    y, x = divmod(i // 2, 5)
    return kk_reg_a_prime_prime_prime(x, y) + i % 2
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


def keccak_ctl_data_inputs(t: CtlTable):
    """keccak_stark::ctl_data_inputs() with ctl_filter_inputs() (keccak_stark.rs:34-50)."""
    return t.singles_set([kk_reg_input_limb(i) for i in range(50)] + [KK_TIMESTAMP], filter_col=0)


def keccak_ctl_data_outputs(t: CtlTable):
    """keccak_stark::ctl_data_outputs() with ctl_filter_outputs() (keccak_stark.rs:40-54)."""
    return t.singles_set([kk_reg_output_limb(i) for i in range(50)] + [KK_TIMESTAMP], filter_col=KK_ROUNDS - 1)


def _sponge_keccak_filter(t: CtlTable):
    return t.sum([KS_FULL] + list(range(KS_FINAL_LEN, KS_FINAL_LEN + KECCAK_RATE_BYTES)))


def keccak_sponge_looking_keccak_inputs(t: CtlTable):
    """keccak_sponge_stark::ctl_looking_keccak_inputs() with ctl_looking_keccak_filter() (:53-67, :196-201)."""
    first = len(t._cols)
    for c in list(range(KS_XORED, KS_XORED + 34)) + list(range(KS_ORIG_CAP, KS_ORIG_CAP + 16)) + [KS_TIMESTAMP]:
        t.single(c)
    return t.colset(range(first, first + 51), filter_constants=[_sponge_keccak_filter(t)])


def keccak_sponge_looking_keccak_outputs(t: CtlTable):
    """keccak_sponge_stark::ctl_looking_keccak_outputs() with ctl_looking_keccak_filter() (:69-89, :196-201)."""
    first = len(t._cols)
    for l in range(8):
        t.le_bytes(range(KS_DIGEST + 4 * l, KS_DIGEST + 4 * l + 4))
    for c in list(range(KS_PARTIAL, KS_PARTIAL + 42)) + [KS_TIMESTAMP]:
        t.single(c)
    return t.colset(range(first, first + 51), filter_constants=[_sponge_keccak_filter(t)])


def ctl_keccak_inputs(sponge_index, keccak_index, sponge_ctl, keccak_ctl):
    """all_stark::ctl_keccak_inputs() (all_stark.rs:214-226)."""
    return [(sponge_index, keccak_sponge_looking_keccak_inputs(sponge_ctl))], (keccak_index, keccak_ctl_data_inputs(keccak_ctl))


def ctl_keccak_outputs(sponge_index, keccak_index, sponge_ctl, keccak_ctl):
    """all_stark::ctl_keccak_outputs() (all_stark.rs:228-240)."""
    return [(sponge_index, keccak_sponge_looking_keccak_outputs(sponge_ctl))], (keccak_index, keccak_ctl_data_outputs(keccak_ctl))


def memory_ctl_data(t: CtlTable):
    """memory_stark::ctl_data() with ctl_filter() (memory_stark.rs:32-43; VALUE_LIMBS = 1)."""
    return t.singles_set([MEM_IS_READ, MEM_CONTEXT, MEM_SEGMENT, MEM_VIRTUAL, MEM_VALUE, MEM_TIMESTAMP], filter_col=MEM_FILTER)


def keccak_sponge_looking_memory(t: CtlTable, i):
    """keccak_sponge_stark::ctl_looking_memory(i) with ctl_looking_memory_filter(i) (:91-124, :174-186): byte i of the block
    is read as part of the big-endian word at virt[i / 4]."""
    start = (i // 4) * 4
    first = t.constant(1)                                                      # is_read
    t.single(KS_CONTEXT)
    t.single(KS_SEGMENT)
    t.single(KS_VIRT + i // 4)
    t.le_bytes([KS_BLOCK + start + 3, KS_BLOCK + start + 2, KS_BLOCK + start + 1, KS_BLOCK + start])
    t.single(KS_TIMESTAMP)
    if i == KECCAK_RATE_BYTES - 1:
        f = t.single(KS_FULL)
    else:
        f = t.sum([KS_FULL] + list(range(KS_FINAL_LEN + i + 1, KS_FINAL_LEN + KECCAK_RATE_BYTES)))
    return t.colset(range(first, first + 6), filter_constants=[f])


def ctl_memory_keccak_sponge(sponge_index, memory_index, sponge_ctl, memory_ctl):
    """The KeccakSponge part of all_stark::ctl_memory() (all_stark.rs:479-542): 136 looking column sets."""
    looking = [(sponge_index, keccak_sponge_looking_memory(sponge_ctl, i)) for i in range(KECCAK_RATE_BYTES)]
    return looking, (memory_index, memory_ctl_data(memory_ctl))


def poseidon_ctl_data_inputs(t: CtlTable):
    """poseidon_stark::ctl_data_inputs() with ctl_filter_inputs() (poseidon_stark.rs:29-45)."""
    return t.singles_set(list(range(POS_IN, POS_IN + 12)) + [POS_TIMESTAMP], filter_col=POS_FILTER)


def poseidon_ctl_data_outputs(t: CtlTable):
    """poseidon_stark::ctl_data_outputs() with ctl_filter_outputs() (poseidon_stark.rs:36-49)."""
    return t.singles_set(list(range(POS_OUT, POS_OUT + 12)) + [POS_TIMESTAMP], filter_col=POS_FILTER)


def _poseidon_sponge_filter(t: CtlTable):
    return t.sum([PS_FULL] + list(range(PS_FINAL_LEN, PS_FINAL_LEN + POSEIDON_RATE_BYTES)))


def poseidon_sponge_looking_poseidon_inputs(t: CtlTable):
    """poseidon_sponge_stark::ctl_looking_poseidon_inputs() with ctl_looking_poseidon_filter() (:44-51, :136-141)."""
    first = len(t._cols)
    for c in list(range(PS_NEW_RATE, PS_NEW_RATE + 8)) + list(range(PS_ORIG_CAP, PS_ORIG_CAP + 4)) + [PS_TIMESTAMP]:
        t.single(c)
    return t.colset(range(first, first + 13), filter_constants=[_poseidon_sponge_filter(t)])


def poseidon_sponge_looking_poseidon_outputs(t: CtlTable):
    """poseidon_sponge_stark::ctl_looking_poseidon_outputs() with ctl_looking_poseidon_filter() (:53-61, :136-141)."""
    first = len(t._cols)
    for c in list(range(PS_DIGEST, PS_DIGEST + 4)) + list(range(PS_PARTIAL, PS_PARTIAL + 8)) + [PS_TIMESTAMP]:
        t.single(c)
    return t.colset(range(first, first + 13), filter_constants=[_poseidon_sponge_filter(t)])


def poseidon_sponge_looking_memory(t: CtlTable, i):
    """poseidon_sponge_stark::ctl_looking_memory(i) with ctl_looking_memory_filter(i) (:63-104, :120-134)."""
    start = (i // 4) * 4
    first = t.constant(1)
    t.single(PS_CONTEXT)
    t.single(PS_SEGMENT)
    t.single(PS_VIRT + i // 4)
    t.le_bytes([PS_BLOCK + start + 3, PS_BLOCK + start + 2, PS_BLOCK + start + 1, PS_BLOCK + start])
    t.single(PS_TIMESTAMP)
    if i == POSEIDON_RATE_BYTES - 1:
        f = t.single(PS_FULL)
    else:
        f = t.sum([PS_FULL] + list(range(PS_FINAL_LEN + i + 1, PS_FINAL_LEN + POSEIDON_RATE_BYTES)))
    return t.colset(range(first, first + 6), filter_constants=[f])


def ctl_poseidon_inputs(sponge_index, poseidon_index, sponge_ctl, poseidon_ctl):
    """all_stark::ctl_poseidon_inputs() (all_stark.rs:169-181)."""
    return [(sponge_index, poseidon_sponge_looking_poseidon_inputs(sponge_ctl))], (poseidon_index, poseidon_ctl_data_inputs(poseidon_ctl))


def ctl_poseidon_outputs(sponge_index, poseidon_index, sponge_ctl, poseidon_ctl):
    """all_stark::ctl_poseidon_outputs() (all_stark.rs:183-195)."""
    return [(sponge_index, poseidon_sponge_looking_poseidon_outputs(sponge_ctl))], (poseidon_index, poseidon_ctl_data_outputs(poseidon_ctl))


def memory_lookers_poseidon_sponge(sponge_index, sponge_ctl):
    """The PoseidonSponge part of all_stark::ctl_memory() (all_stark.rs:487-493): 32 looking column sets."""
    return [(sponge_index, poseidon_sponge_looking_memory(sponge_ctl, i)) for i in range(POSEIDON_RATE_BYTES)]


def memory_lookers_keccak_sponge(sponge_index, sponge_ctl):
    return [(sponge_index, keccak_sponge_looking_memory(sponge_ctl, i)) for i in range(KECCAK_RATE_BYTES)]


def sha_extend_ctl_data_inputs(t: CtlTable):
    """sha_extend_stark::ctl_data_inputs() with ctl_filter() (sha_extend_stark.rs:31-45, :110-114)."""
    cols = list(range(SE_W15, SE_W15 + 4)) + list(range(SE_W2, SE_W2 + 4)) + list(range(SE_W16, SE_W16 + 4)) + list(range(SE_W7, SE_W7 + 4))
    return t.singles_set(cols + [SE_TIMESTAMP], filter_col=SE_IS_REAL)


def sha_extend_ctl_data_outputs(t: CtlTable):
    """sha_extend_stark::ctl_data_outputs() with ctl_filter() (sha_extend_stark.rs:47-52)."""
    return t.singles_set(list(range(SE_W_I, SE_W_I + 4)) + [SE_TIMESTAMP], filter_col=SE_IS_REAL)


def sha_extend_looking_logic(t: CtlTable, which):
    """ctl_s_0_inter / s_0 / s_1_inter / s_1 _looking_logic() with ctl_filter() (sha_extend_stark.rs:54-108): an XOR of two
    little-endian byte quadruples."""
    in0, in1, out = {"s_0_inter": (SE_RR7, SE_RR18, SE_S0_INTER), "s_0": (SE_S0_INTER, SE_RS3, SE_S0),
                     "s_1_inter": (SE_RR17, SE_RR19, SE_S1_INTER), "s_1": (SE_S1_INTER, SE_RS10, SE_S1)}[which]
    first = t.constant(0b100110 << 6)
    for c in (in0, in1, out):
        t.le_bytes(range(c, c + 4))
    f = t.single(SE_IS_REAL)
    return t.colset(range(first, first + 4), filter_constants=[f])


def _ses_filter(t: CtlTable):
    return t.sum(range(SES_ROUND, SES_ROUND + 48))


def sha_extend_sponge_looking_inputs(t: CtlTable):
    """sha_extend_sponge_stark::ctl_looking_sha_extend_inputs() with ctl_looking_sha_extend_filter() (:31-45, :106-110)."""
    first = len(t._cols)
    for c in list(range(SES_W15, SES_W15 + 16)) + [SES_TIMESTAMP]:
        t.single(c)
    return t.colset(range(first, first + 17), filter_constants=[_ses_filter(t)])


def sha_extend_sponge_looking_outputs(t: CtlTable):
    first = len(t._cols)
    for c in list(range(SES_W_I, SES_W_I + 4)) + [SES_TIMESTAMP]:
        t.single(c)
    return t.colset(range(first, first + 5), filter_constants=[_ses_filter(t)])


def sha_extend_sponge_looking_memory(t: CtlTable, i):
    """sha_extend_sponge_stark::ctl_looking_memory(i) (:67-104): byte i belongs to input word i / 4, read as a little-endian u32."""
    q = i // 4
    first = t.constant(1)
    t.single(SES_CONTEXT)
    t.single(SES_SEGMENT)
    t.single(SES_IN_VIRT + q)
    t.le_bytes(range(SES_W15 + 4 * q, SES_W15 + 4 * q + 4))
    t.single(SES_TIMESTAMP)
    return t.colset(range(first, first + 6), filter_constants=[_ses_filter(t)])


def ctl_sha_extend_inputs(sponge_index, extend_index, sponge_ctl, extend_ctl):
    """all_stark::ctl_sha_extend_inputs() (all_stark.rs:256-268)."""
    return [(sponge_index, sha_extend_sponge_looking_inputs(sponge_ctl))], (extend_index, sha_extend_ctl_data_inputs(extend_ctl))


def ctl_sha_extend_outputs(sponge_index, extend_index, sponge_ctl, extend_ctl):
    """all_stark::ctl_sha_extend_outputs() (all_stark.rs:270-282)."""
    return [(sponge_index, sha_extend_sponge_looking_outputs(sponge_ctl))], (extend_index, sha_extend_ctl_data_outputs(extend_ctl))


def logic_lookers_sha_extend(extend_index, extend_ctl):
    """The ShaExtend part of all_stark::ctl_logic() (all_stark.rs:356-385), in the reference's order."""
    return [(extend_index, sha_extend_looking_logic(extend_ctl, w)) for w in ("s_0_inter", "s_0", "s_1_inter", "s_1")]


def memory_lookers_sha_extend_sponge(sponge_index, sponge_ctl):
    """The ShaExtendSponge part of all_stark::ctl_memory() (all_stark.rs:503-509): 16 looking column sets."""
    return [(sponge_index, sha_extend_sponge_looking_memory(sponge_ctl, i)) for i in range(16)]


def sha_compress_ctl_data_inputs(t: CtlTable):
    """sha_compress_stark::ctl_data_inputs() with ctl_filter_inputs() (sha_compress_stark.rs:39-50, :213-217)."""
    return t.singles_set(list(range(SC_STATE, SC_STATE + 32)) + [SC_TIMESTAMP, SC_SEGMENT, SC_CONTEXT, SC_W_I_VIRT], filter_col=SC_ROUND)


def sha_compress_ctl_data_outputs(t: CtlTable):
    """sha_compress_stark::ctl_data_outputs() with ctl_filter_outputs() (sha_compress_stark.rs:52-57, :219-223)."""
    return t.singles_set(list(range(SC_STATE, SC_STATE + 32)) + [SC_TIMESTAMP], filter_col=SC_ROUND + 64)


# (opcode, input 0, input 1, output) of the twelve logic lookups of a compression round, in the order of all_stark.rs:387-470
_XOR, _AND = 0b100110 << 6, 0b100100 << 6
SHA_COMPRESS_LOGIC = [
    (_XOR, SC_E_RR6, SC_E_RR11, SC_S1_INTER), (_XOR, SC_S1_INTER, SC_E_RR25, SC_S1), (_AND, SC_STATE + 16, SC_STATE + 20, SC_E_AND_F),
    (_AND, SC_E_NOT, SC_STATE + 24, SC_ENOT_AND_G), (_XOR, SC_E_AND_F, SC_ENOT_AND_G, SC_CH), (_XOR, SC_A_RR2, SC_A_RR13, SC_S0_INTER),
    (_XOR, SC_S0_INTER, SC_A_RR22, SC_S0), (_AND, SC_STATE, SC_STATE + 4, SC_A_AND_B), (_AND, SC_STATE, SC_STATE + 8, SC_A_AND_C),
    (_AND, SC_STATE + 4, SC_STATE + 8, SC_B_AND_C), (_XOR, SC_A_AND_B, SC_A_AND_C, SC_MAJ_INTER), (_XOR, SC_MAJ_INTER, SC_B_AND_C, SC_MAJ)]


def _sc_round_filter(t: CtlTable):
    """ctl_logic_filter(): rounds 0..63 (sha_compress_stark.rs:225-229)."""
    return t.sum(range(SC_ROUND, SC_ROUND + 64))


def logic_lookers_sha_compress(index, t: CtlTable):
    """The ShaCompress part of all_stark::ctl_logic() (all_stark.rs:387-470; columns sha_compress_stark.rs:59-191)."""
    out = []
    for opcode, in0, in1, res in SHA_COMPRESS_LOGIC:
        first = t.constant(opcode)
        for c in (in0, in1, res):
            t.le_bytes(range(c, c + 4))
        out.append((index, t.colset(range(first, first + 4), filter_constants=[_sc_round_filter(t)])))
    return out


def memory_lookers_sha_compress(index, t: CtlTable):
    """The ShaCompress part of all_stark::ctl_memory() (all_stark.rs:519-525): w_i is read four times per round."""
    out = []
    for _ in range(4):
        first = t.constant(1)
        t.single(SC_CONTEXT)
        t.single(SC_SEGMENT)
        t.single(SC_W_I_VIRT)
        t.le_bytes(range(SC_W_I, SC_W_I + 4))
        t.single(SC_TIMESTAMP)
        out.append((index, t.colset(range(first, first + 6), filter_constants=[_sc_round_filter(t)])))
    return out


def sha_compress_sponge_looking_inputs(t: CtlTable):
    """sha_compress_sponge_stark::ctl_looking_sha_compress_inputs() with ctl_looking_sha_compress_filter() (:29-39, :81-86)."""
    first = len(t._cols)
    for c in list(range(SCS_HX, SCS_HX + 32)) + [SCS_TIMESTAMP, SCS_W_SEGMENT, SCS_W_CONTEXT, SCS_W_VIRT]:
        t.single(c)
    return t.colset(range(first, first + 36), filter_constants=[t.single(SCS_IS_REAL)])


def sha_compress_sponge_looking_outputs(t: CtlTable):
    first = len(t._cols)
    for c in list(range(SCS_OUT_STATE, SCS_OUT_STATE + 32)) + [SCS_TIMESTAMP]:
        t.single(c)
    return t.colset(range(first, first + 33), filter_constants=[t.single(SCS_IS_REAL)])


def memory_lookers_sha_compress_sponge(index, t: CtlTable):
    """The ShaCompressSponge part of all_stark::ctl_memory() (all_stark.rs:511-517; columns sha_compress_sponge_stark.rs:63-79)."""
    out = []
    for i in range(32):
        q = i // 4
        first = t.constant(1)
        t.single(SCS_CONTEXT)
        t.single(SCS_SEGMENT)
        t.single(SCS_HX_VIRT + q)
        t.le_bytes(range(SCS_HX + 4 * q, SCS_HX + 4 * q + 4))
        t.single(SCS_TIMESTAMP)
        out.append((index, t.colset(range(first, first + 6), filter_constants=[t.single(SCS_IS_REAL)])))
    return out


def ctl_sha_compress_inputs(sponge_index, compress_index, sponge_ctl, compress_ctl):
    """all_stark::ctl_sha_compress_inputs() (all_stark.rs:298-310)."""
    return [(sponge_index, sha_compress_sponge_looking_inputs(sponge_ctl))], (compress_index, sha_compress_ctl_data_inputs(compress_ctl))


def ctl_sha_compress_outputs(sponge_index, compress_index, sponge_ctl, compress_ctl):
    """all_stark::ctl_sha_compress_outputs() (all_stark.rs:312-324)."""
    return [(sponge_index, sha_compress_sponge_looking_outputs(sponge_ctl))], (compress_index, sha_compress_ctl_data_outputs(compress_ctl))


# ArithmeticStark (arithmetic/columns.rs): operation flags 0..25, 16-bit limb registers
ARITH_IN0, ARITH_IN1, ARITH_OUT = 26, 28, 32
ARITH_COMBINED_OPS = [(0, 0b100000 << 6), (1, 0b100001 << 6), (2, 0b001000), (3, 0b001001), (4, 0b100010 << 6), (5, 0b100011 << 6),
                      (6, 0b011000 << 6), (7, 0b011001 << 6), (8, 0b011100 + (0b000010 << 6)), (9, 0b011010 << 6), (10, 0b011011 << 6),
                      (11, 0b000100 << 6), (12, 0b000110 << 6), (13, 0b000111 << 6), (14, 0), (15, 0b000010 << 6), (16, 0b000011 << 6),
                      (17, 0b101010 << 6), (18, 0b101011 << 6), (19, 0b001010), (20, 0b001011), (21, 0b001111), (22, 0b010000 << 6),
                      (23, 0b010001 << 6), (24, 0b010010 << 6), (25, 0b010011 << 6)]


def arithmetic_ctl_rows(t: CtlTable):
    """arithmetic_stark::ctl_arithmetic_rows() (arithmetic_stark.rs:26-126): the opcode combination, then INPUT_REGISTER_0,
    INPUT_REGISTER_1 and OUTPUT_REGISTER as 32-bit values (limb0 + 2^16 limb1); filter = sum of the 26 flags.  The CPU table is the
    only looker (all_stark.rs:156-164), so this set is exercised on its own until the CPU table has a kernel."""
    first = t.column(local=[(c, code) for c, code in ARITH_COMBINED_OPS])
    for reg in (ARITH_IN0, ARITH_IN1, ARITH_OUT):
        t.column(local=[(reg, 1), (reg + 1, 1 << 16)])
    f = t.sum([c for c, _ in ARITH_COMBINED_OPS])
    return t.colset(range(first, first + 4), filter_constants=[f])


# ---------------------------------------------------------------------------------------------------------------- CpuStark
# cpu/columns/mod.rs:62-96; the CTL column builders are cpu/cpu_stark.rs:25-250.
CPU_OP_BINARY, CPU_OP_BINARY_IMM, CPU_OP_LOGIC, CPU_OP_SHIFT, CPU_OP_SHIFT_IMM = 7, 8, 10, 16, 17
CPU_OPCODE_BITS, CPU_FUNC_BITS = 50, 76
CPU_IS_POSEIDON_SPONGE, CPU_IS_KECCAK_SPONGE, CPU_IS_SHA_EXTEND_SPONGE, CPU_IS_SHA_COMPRESS_SPONGE = 82, 83, 84, 85
CPU_GENERAL, CPU_CLOCK, CPU_CHANNELS, CPU_NUM_GP_CHANNELS, CPU_NUM_CHANNELS = 86, 204, 205, 9, 10
KS_LEN = 38
SCS_OUT_HX = 64


def cpu_channel(i, field):
    """field: 0 used, 1 is_read, 2 addr_context, 3 addr_segment, 4 addr_virtual, 5 value."""
    return CPU_CHANNELS + 6 * i + field


def _cpu_binops(t: CtlTable):
    for i in range(3):
        t.single(cpu_channel(i, 5))


def cpu_looking_logic(t: CtlTable):
    """cpu_stark::ctl_data_logic() / ctl_filter_logic() (:131-145)."""
    first = t.le_bits(list(range(CPU_OPCODE_BITS, CPU_OPCODE_BITS + 6)) + list(range(CPU_FUNC_BITS, CPU_FUNC_BITS + 6)))
    _cpu_binops(t)
    return t.colset(range(first, first + 4), filter_constants=[t.single(CPU_OP_LOGIC)])


def cpu_looking_arithmetic(t: CtlTable):
    """cpu_stark::ctl_arithmetic_base_rows() (:150-171): filter = binary_op + shift + shift_imm."""
    first = t.le_bits(list(range(CPU_OPCODE_BITS, CPU_OPCODE_BITS + 6)) + list(range(CPU_FUNC_BITS, CPU_FUNC_BITS + 6)))
    _cpu_binops(t)
    return t.colset(range(first, first + 4), filter_constants=[t.sum([CPU_OP_BINARY, CPU_OP_SHIFT, CPU_OP_SHIFT_IMM])])


def cpu_looking_arithmetic_imm(t: CtlTable):
    """cpu_stark::ctl_arithmetic_imm_base_rows() (:173-186)."""
    first = t.le_bits(range(CPU_OPCODE_BITS, CPU_OPCODE_BITS + 6))
    _cpu_binops(t)
    return t.colset(range(first, first + 4), filter_constants=[t.single(CPU_OP_BINARY_IMM)])


def cpu_looking_memory(t: CtlTable, channel):
    """cpu_stark::ctl_data_gp_memory(channel) / ctl_filter_gp_memory(channel) (:220-246): timestamp = clock * NUM_CHANNELS."""
    first = t.single(cpu_channel(channel, 1))
    for f in (2, 3, 4, 5):
        t.single(cpu_channel(channel, f))
    t.column(local=[(CPU_CLOCK, CPU_NUM_CHANNELS)])
    return t.colset(range(first, first + 6), filter_constants=[t.single(cpu_channel(channel, 0))])


def _cpu_looking_sponge(t: CtlTable, nchan, nvalues, flag):
    first = t.single(cpu_channel(0, 5))
    for i in range(1, nchan):
        t.single(cpu_channel(i, 5))
    t.column(local=[(CPU_CLOCK, CPU_NUM_CHANNELS)])
    for i in range(nvalues):
        t.single(CPU_GENERAL + i)
    return t.colset(range(first, first + nchan + 1 + nvalues), filter_constants=[t.single(flag)])


def cpu_looking_keccak_sponge(t: CtlTable):
    """cpu_stark::ctl_data_keccak_sponge() (:25-43): context, segment, virt, len, timestamp, khash().value[8]."""
    return _cpu_looking_sponge(t, 4, 8, CPU_IS_KECCAK_SPONGE)


def cpu_looking_poseidon_sponge(t: CtlTable):
    """cpu_stark::ctl_data_poseidon_sponge() (:94-112): ..., hash().value[4]."""
    return _cpu_looking_sponge(t, 4, 4, CPU_IS_POSEIDON_SPONGE)


def cpu_looking_sha_extend_sponge(t: CtlTable):
    """cpu_stark::ctl_data_sha_extend_sponge() (:45-61): context, segment, virt, timestamp, element().value."""
    return _cpu_looking_sponge(t, 3, 1, CPU_IS_SHA_EXTEND_SPONGE)


def cpu_looking_sha_compress_sponge(t: CtlTable):
    """cpu_stark::ctl_data_sha_compress_sponge() (:63-80): context, segment, virt, timestamp, shash().value[8]."""
    return _cpu_looking_sponge(t, 3, 8, CPU_IS_SHA_COMPRESS_SPONGE)


def keccak_sponge_looked_data(t: CtlTable):
    """keccak_sponge_stark::ctl_looked_data() / ctl_looked_filter() (:28-50, :167-171): the digest as eight big-endian words in
    reverse order."""
    first = t.single(KS_CONTEXT)
    for c in (KS_SEGMENT, KS_VIRT, KS_LEN, KS_TIMESTAMP):
        t.single(c)
    for i in reversed(range(8)):
        t.column(local=[(KS_DIGEST + 4 * i + j, 1 << (24 - 8 * j)) for j in range(4)])
    f = t.sum(range(KS_FINAL_LEN, KS_FINAL_LEN + KECCAK_RATE_BYTES))
    return t.colset(range(first, first + 13), filter_constants=[f])


def poseidon_sponge_looked_data(t: CtlTable):
    """poseidon_sponge_stark::ctl_looked_data() / ctl_looked_filter() (:28-43, :102-106)."""
    first = t.single(PS_CONTEXT)
    for c in [PS_SEGMENT, PS_VIRT, PS_LEN, PS_TIMESTAMP] + list(range(PS_DIGEST, PS_DIGEST + 4)):
        t.single(c)
    f = t.sum(range(PS_FINAL_LEN, PS_FINAL_LEN + POSEIDON_RATE_BYTES))
    return t.colset(range(first, first + 9), filter_constants=[f])


def sha_extend_sponge_looked_data(t: CtlTable):
    """sha_extend_sponge_stark::ctl_looked_data() / ctl_looking_sha_extend_filter() (:55-62, :97-101)."""
    first = t.single(SES_CONTEXT)
    for c in (SES_SEGMENT, SES_OUT_VIRT, SES_TIMESTAMP):
        t.single(c)
    t.le_bytes(range(SES_W_I, SES_W_I + 4))
    return t.colset(range(first, first + 5), filter_constants=[t.sum(range(SES_ROUND, SES_ROUND + 48))])


def sha_compress_sponge_looked_data(t: CtlTable):
    """sha_compress_sponge_stark::ctl_looked_data() / ctl_looked_filter() (:49-61, :89-95); output_hx[i] = WrappingAdd2Op
    {value[4], carry[2]}."""
    first = t.single(SCS_CONTEXT)
    for c in (SCS_SEGMENT, SCS_HX_VIRT, SCS_TIMESTAMP):
        t.single(c)
    for i in range(8):
        t.le_bytes(range(SCS_OUT_HX + 6 * i, SCS_OUT_HX + 6 * i + 4))
    return t.colset(range(first, first + 12), filter_constants=[t.single(SCS_IS_REAL)])


def ctl_arithmetic(cpu_index, arith_index, cpu_ctl, arith_ctl):
    """all_stark::ctl_arithmetic() (all_stark.rs:156-164)."""
    return [(cpu_index, cpu_looking_arithmetic(cpu_ctl)), (cpu_index, cpu_looking_arithmetic_imm(cpu_ctl))], \
        (arith_index, arithmetic_ctl_rows(arith_ctl))


def logic_lookers_cpu(cpu_index, cpu_ctl):
    """The CPU looker of all_stark::ctl_logic() (all_stark.rs:326-338)."""
    return [(cpu_index, cpu_looking_logic(cpu_ctl))]


def memory_lookers_cpu(cpu_index, cpu_ctl):
    """The nine general-purpose channels of the CPU in all_stark::ctl_memory() (all_stark.rs:480-486)."""
    return [(cpu_index, cpu_looking_memory(cpu_ctl, ch)) for ch in range(CPU_NUM_GP_CHANNELS)]


def ctl_keccak_sponge(cpu_index, sponge_index, cpu_ctl, sponge_ctl):
    """all_stark::ctl_keccak_sponge() (all_stark.rs:242-254)."""
    return [(cpu_index, cpu_looking_keccak_sponge(cpu_ctl))], (sponge_index, keccak_sponge_looked_data(sponge_ctl))


def ctl_poseidon_sponge(cpu_index, sponge_index, cpu_ctl, sponge_ctl):
    """all_stark::ctl_poseidon_sponge() (all_stark.rs:197-209)."""
    return [(cpu_index, cpu_looking_poseidon_sponge(cpu_ctl))], (sponge_index, poseidon_sponge_looked_data(sponge_ctl))


def ctl_sha_extend_sponge(cpu_index, sponge_index, cpu_ctl, sponge_ctl):
    """all_stark::ctl_sha_extend_sponge() (all_stark.rs:284-296)."""
    return [(cpu_index, cpu_looking_sha_extend_sponge(cpu_ctl))], (sponge_index, sha_extend_sponge_looked_data(sponge_ctl))


def ctl_sha_compress_sponge(cpu_index, sponge_index, cpu_ctl, sponge_ctl):
    """all_stark::ctl_sha_compress_sponge() (all_stark.rs:326-338)."""
    return [(cpu_index, cpu_looking_sha_compress_sponge(cpu_ctl))], (sponge_index, sha_compress_sponge_looked_data(sponge_ctl))


# ---- the whole AllStark (all_stark.rs:96-155): table order of Table::all() and all_cross_table_lookups()
TABLE_ENUM_ORDER = [TABLE_ARITHMETIC, TABLE_CPU, TABLE_POSEIDON, TABLE_POSEIDON_SPONGE, TABLE_KECCAK, TABLE_KECCAK_SPONGE, TABLE_SHA_EXTEND,
                    TABLE_SHA_EXTEND_SPONGE, TABLE_SHA_COMPRESS, TABLE_SHA_COMPRESS_SPONGE, TABLE_LOGIC, TABLE_MEMORY]


def all_cross_table_lookups():
    """all_cross_table_lookups() (all_stark.rs:136-155) on the twelve tables in Table::all() order.
    Returns (ctl_tables, ctls): ctl_tables[i] = the CtlTable (column sets) of table TABLE_ENUM_ORDER[i]; ctls = the fifteen
    lookups as (looking sides, looked side), sides = (table index in enum order, column-set index).  The construction order of
    the column sets is part of the contract: it fixes the column-set indices the C tables of csrc/all_stark_ctl.inc are
    generated from (tools/gen_all_stark_ctl.py)."""
    c = [CtlTable() for _ in range(12)]
    AR, CPU, PO, PS, KK, KS, SE, SES, SC, SCS, LO, ME = range(12)
    logic_lookers = logic_lookers_cpu(CPU, c[CPU]) + [(KS, keccak_sponge_looking_logic(c[KS], i)) for i in range(NUM_LOGIC_CTLS)] + \
        logic_lookers_sha_extend(SE, c[SE]) + logic_lookers_sha_compress(SC, c[SC])
    # ctl_memory() chains cpu, keccak_sponge, poseidon_sponge, sha_extend_sponge, sha_compress_sponge, sha_compress (all_stark.rs:527-534):
    # a consumer that indexes the looking sides positionally (verifier, recursion) sees the reference's order
    memory_lookers = memory_lookers_cpu(CPU, c[CPU]) + memory_lookers_keccak_sponge(KS, c[KS]) + \
        memory_lookers_poseidon_sponge(PS, c[PS]) + memory_lookers_sha_extend_sponge(SES, c[SES]) + \
        memory_lookers_sha_compress_sponge(SCS, c[SCS]) + memory_lookers_sha_compress(SC, c[SC])
    ctls = [ctl_arithmetic(CPU, AR, c[CPU], c[AR]),
            ctl_poseidon_sponge(CPU, PS, c[CPU], c[PS]), ctl_poseidon_inputs(PS, PO, c[PS], c[PO]), ctl_poseidon_outputs(PS, PO, c[PS], c[PO]),
            ctl_keccak_sponge(CPU, KS, c[CPU], c[KS]), ctl_keccak_inputs(KS, KK, c[KS], c[KK]), ctl_keccak_outputs(KS, KK, c[KS], c[KK]),
            ctl_sha_extend_sponge(CPU, SES, c[CPU], c[SES]), ctl_sha_extend_inputs(SES, SE, c[SES], c[SE]),
            ctl_sha_extend_outputs(SES, SE, c[SES], c[SE]),
            ctl_sha_compress_sponge(CPU, SCS, c[CPU], c[SCS]), ctl_sha_compress_inputs(SCS, SC, c[SCS], c[SC]),
            ctl_sha_compress_outputs(SCS, SC, c[SCS], c[SC]),
            (logic_lookers, (LO, logic_ctl_data(c[LO]))), (memory_lookers, (ME, memory_ctl_data(c[ME])))]
    return c, ctls
