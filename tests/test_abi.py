"""CPU suite: the C-ABI library loads and exports every symbol include/zkm_hip.h declares; host-side
transcript code (no GPU needed) agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "zkm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(zkm):
    lib = zkm.load()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libzkmhip.so does not export %s" % n
    assert set(names) == set(zkm.EXPORTS), set(names) ^ set(zkm.EXPORTS)
    assert b"gfx950" in lib.zkm_version()


def test_no_gpu_fails_loudly(zkm):
    import torch
    if torch.cuda.is_available():
        return
    try:
        zkm.Context(0)
    except zkm.ZkmError as e:
        assert "hip" in str(e).lower()
    else:
        raise AssertionError("Context creation must fail without a GPU (no CPU fallback)")


def test_host_challenger_matches_oracle(zkm, oracle):
    rng = np.random.default_rng(11)
    a, b = zkm.challenger_new(), oracle.challenger()
    for step in range(40):
        k = int(rng.integers(0, 20))
        elems = rng.integers(0, zkm.P, k, dtype=np.uint64)
        zkm.challenger_observe(a, elems)
        oracle.observe(b, elems)
        for _ in range(int(rng.integers(0, 11))):
            assert zkm.challenger_get(a) == oracle.challenge(b)
    assert list(a.state) == list(b.state)


def test_host_permutation_edge_states_match_oracle(zkm, oracle):
    """The host-side permutation (csrc/host_poseidon.hip: sparse partial rounds, 128-bit accumulation, vectorised MDS) on the states
    that exercise its carries and borrows: all-zero, all p - 1, single words at the limb boundaries, long runs of random blocks --
    through the Challenger (the only way host code reaches it), against the oracle's independent permutation."""
    P = zkm.P
    edge = [0, 1, 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - 1, P - 2, P - (1 << 32), (1 << 63), (1 << 63) - 1, P >> 1]
    rng = np.random.default_rng(12)
    blocks = [[e] * 8 for e in edge]
    blocks += [[edge[(i + k) % len(edge)] for k in range(8)] for i in range(len(edge))]
    blocks += [[int(x) for x in rng.integers(0, P, 8, dtype=np.uint64)] for _ in range(3000)]
    a, b = zkm.challenger_new(), oracle.challenger()
    for i, blk in enumerate(blocks):
        zkm.challenger_observe(a, blk)
        oracle.observe(b, blk)
        if i % 64 == 0 or i < 30:
            assert list(a.state) == list(b.state), i
    assert list(a.state) == list(b.state)
    for _ in range(8):
        assert zkm.challenger_get(a) == oracle.challenge(b)


def test_proof_layout_sizes_agree(zkm, oracle):
    cfg_o = oracle.standard_config()
    cfg = zkm.StarkConfig()
    zkm.load().zkm_standard_config(C.byref(cfg))
    for f, _ in cfg._fields_:
        assert getattr(cfg, f) == getattr(cfg_o, f)
    lib = zkm.load()
    for log_n in (5, 7, 12, 16, 20, 22):
        assert lib.zkm_proof_words(C.byref(cfg), log_n, 262, 4, 2) == oracle.proof_words(cfg_o, log_n, 262, 4, 2)


def test_all_stark_ctl_inc_is_current():
    """csrc/all_stark_ctl.inc (the AllStark lookups compiled into the library) is generated from zkm_amd/tables.py."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_all_stark_ctl", os.path.join(ROOT, "tools", "gen_all_stark_ctl.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.render() == open(os.path.join(ROOT, "zkm_amd", "csrc", "all_stark_ctl.inc")).read(), "run tools/gen_all_stark_ctl.py"


def test_all_stark_looking_side_order_is_the_references():
    """The looking sides of every lookup come in the order the reference chains them (all_stark.rs): a consumer that indexes them
    positionally (verify_cross_table_lookups, the recursion circuits) must see the same sequence.  ctl_memory (:479-542): the CPU's
    NUM_GP_CHANNELS memory channels, then KeccakSponge (136 rate bytes), PoseidonSponge (32), ShaExtendSponge, ShaCompressSponge,
    ShaCompress (4); ctl_logic (:340-477): CPU, KeccakSponge, ShaExtend, ShaCompress."""
    from zkm_amd import tables as T
    AR, CPU, PO, PS, KK, KS, SE, SES, SC, SCS, LO, ME = range(12)
    _, ctls = T.all_cross_table_lookups()
    assert len(ctls) == 15

    def runs(sides):
        out = []
        for t, _ in sides:
            if out and out[-1][0] == t:
                out[-1][1] += 1
            else:
                out.append([t, 1])
        return [tuple(r) for r in out]
    mem_looking, mem_looked = ctls[14]
    assert mem_looked[0] == ME
    r = runs(mem_looking)
    assert [t for t, _ in r] == [CPU, KS, PS, SES, SCS, SC], r
    assert dict(r)[KS] == 136 and dict(r)[PS] == 32 and dict(r)[SC] == 4
    logic_looking, logic_looked = ctls[13]
    assert logic_looked[0] == LO and [t for t, _ in runs(logic_looking)] == [CPU, KS, SE, SC]
    # the looked tables of the fifteen lookups, in all_cross_table_lookups() order (all_stark.rs:137-155)
    assert [looked[0] for _, looked in ctls] == [AR, PS, PO, PO, KS, KK, KK, SES, SE, SE, SCS, SC, SC, LO, ME]


def test_prove_segment_sizing_and_descriptors(zkm, oracle):
    """zkm_prove_segment sizes a whole AllStark segment without a GPU, from the description inside the library; it agrees with the
    oracle's sizing of the same tables + lookups built through zkm_amd/tables.py, and the exported lookups are the fifteen of
    all_cross_table_lookups() (all_stark.rs:136-155)."""
    from zkm_amd import tables as T
    lib = zkm.load()
    seg = np.load(os.path.join(ROOT, "tests", "golden", "segment12.npz"))
    log_n = [int(x) for x in seg["log_n"]]
    ctl_tables, ctls = T.all_cross_table_lookups()
    tables = [(T.TABLE_ENUM_ORDER[i], seg["t%d" % i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
    want_total, want_offs = oracle.all_proof_words(tables, ctls)
    cfg = zkm.StarkConfig()
    lib.zkm_standard_config(C.byref(cfg))
    ptrs = (C.c_void_p * 12)(*[t[1].ctypes.data for t in tables])
    lg = (C.c_uint * 12)(*log_n)
    offs = (C.c_size_t * 13)()
    err = C.c_char_p()
    assert lib.zkm_prove_segment(None, C.byref(cfg), ptrs, lg, None, 0, None, offs, None, C.byref(err)) == 0
    assert list(offs) == list(want_offs) and offs[12] == want_total
    n, ns = C.c_size_t(), C.c_size_t()
    assert lib.zkm_all_stark_ctls(None, C.byref(n), None, C.byref(ns)) == 0
    assert n.value == 15 and ns.value == sum(len(looking) for looking, _ in ctls)
    assert [lib.zkm_table_enum_index(t) for t in T.TABLE_ENUM_ORDER] == list(range(12)) and lib.zkm_table_enum_index(99) == -1
    assert lib.zkm_all_stark_ctl_table(99) is None and lib.zkm_all_stark_ctl_table(T.TABLE_CPU) is not None
    lg[1] = 99  # a table height the library cannot size
    assert lib.zkm_prove_segment(None, C.byref(cfg), ptrs, lg, None, 0, None, offs, None, C.byref(err)) != 0


def test_rust_sys_block_names_every_export(zkm):
    """integration/rust/zkm_hip_sys.rs (the extern "C" block a maintainer adds to the plonky2 fork) declares exactly the exported
    functions of include/zkm_hip.h."""
    text = open(os.path.join(ROOT, "integration", "rust", "zkm_hip_sys.rs")).read()
    rust = set(re.findall(r"pub fn (zkm_[a-z0-9_]+)\s*\(", text))
    assert rust == set(header_functions()), rust ^ set(header_functions())
