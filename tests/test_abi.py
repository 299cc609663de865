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


# ------------------------------------------------------------------ struct layouts: header == ctypes / numpy mirrors == Rust mirror
def c_layouts(tmp_path):
    """sizeof / offsetof of every struct of include/zkm_hip.h, from the C compiler (tools/abi_layout.c built with gcc as C99)."""
    import json
    import subprocess
    exe = str(tmp_path / "abi_layout")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tools", "abi_layout.c")])
    return json.loads(subprocess.check_output([exe]))


def header_structs():
    """{struct name: [field names in declaration order]} parsed from the header text (comments stripped)."""
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "zkm_hip.h")).read(), flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(zkm_[a-z0-9_]+)\s*;", text, flags=re.S):
        fields = []
        for decl in filter(None, (d.strip() for d in body.split(";"))):
            # "const uint64_t* const* columns" / "uint32_t a, b" / "size_t x[3], y": the names are the identifiers before , [ or the end
            decl = re.sub(r"\[[^\]]*\]", "", decl)
            first, *rest = decl.split(",")
            fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1])
            fields += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", r)[-1] for r in rest]
        out[name] = fields
    return out


def test_abi_layout_tool_covers_every_header_struct(tmp_path):
    """tools/abi_layout.c must list every struct of the header with every field, in order: a field added to the header and forgotten
    there (or in a mirror, below) is an error here, not a silent hole in the lock."""
    lay = c_layouts(tmp_path)
    hdr = header_structs()
    assert set(lay) == set(hdr), set(lay) ^ set(hdr)
    for name, fields in hdr.items():
        assert [f[0] for f in lay[name]["fields"]] == fields, name
        # fields tile the struct without overlap, in order
        end = 0
        for _, off, size in lay[name]["fields"]:
            assert off >= end, name
            end = off + size
        assert end <= lay[name]["size"], name


def test_python_mirrors_match_the_c_layout(zkm, tmp_path):
    """The ctypes Structures and numpy dtypes that cross the C ABI (zkm_amd/__init__.py, zkm_amd/ctl.py) against the compiler's layout:
    total size, and offset + size of every field in declaration order."""
    import numpy as np
    lay = c_layouts(tmp_path)
    mirrors = zkm.abi_mirrors()
    assert set(mirrors) == set(lay), set(mirrors) ^ set(lay)
    for name, m in mirrors.items():
        want = [(off, size) for _, off, size in lay[name]["fields"]]
        if isinstance(m, np.dtype):
            got = [(m.fields[f][1], m.fields[f][0].itemsize) for f in m.names]
            size = m.itemsize
            if name == "zkm_cross_table_lookup":      # the nested zkm_ctl_side `looked` is flattened into two u32 in the dtype
                got = got[:2] + [(got[2][0], got[2][1] + got[3][1])]
                assert m.names[2:] == ("looked_table", "looked_colset") and got[2][1] == lay["zkm_ctl_side"]["size"]
            names = None
        else:
            got = [(getattr(m, f).offset, getattr(m, f).size) for f, _ in m._fields_]
            size = C.sizeof(m)
            names = [f for f, _ in m._fields_]
        assert size == lay[name]["size"], (name, size, lay[name]["size"])
        assert got == want, (name, got, want)
        if names is not None:
            assert names == [f[0] for f in lay[name]["fields"]], name


RUST_PRIM = {"u64": (8, 8), "usize": (8, 8), "u32": (4, 4), "c_uint": (4, 4), "c_int": (4, 4), "i32": (4, 4), "u8": (1, 1)}


def rust_structs():
    """{name: [(field, type text)]} of the #[repr(C)] structs in integration/rust/zkm_hip_sys.rs (uncompiled here: parsed)."""
    text = open(os.path.join(ROOT, "integration", "rust", "zkm_hip_sys.rs")).read()
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^{;]*?pub struct (zkm_[a-z0-9_]+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in filter(None, (x.strip() for x in re.split(r",(?![^\[]*\])", m.group(2)))):
            fm = re.match(r"(?:pub\s+)?([A-Za-z_][A-Za-z0-9_]*)\s*:\s*(.+)$", f, flags=re.S)
            assert fm, (m.group(1), f)
            fields.append((fm.group(1), " ".join(fm.group(2).split())))
        out[m.group(1)] = fields
    return out


def rust_type_layout(ty, structs, memo):
    """(size, align) of a Rust type under repr(C) on x86-64 / LP64."""
    if ty.startswith("*const") or ty.startswith("*mut"):
        return 8, 8
    arr = re.match(r"\[(.+);\s*(\d+)\]$", ty)
    if arr:
        s, a = rust_type_layout(arr.group(1).strip(), structs, memo)
        return s * int(arr.group(2)), a
    if ty in RUST_PRIM:
        return RUST_PRIM[ty]
    assert ty in structs, "unknown Rust type %r" % ty
    return rust_struct_layout(ty, structs, memo)[:2]


def rust_struct_layout(name, structs, memo):
    if name not in memo:
        off, align, fields = 0, 1, []
        for f, ty in structs[name]:
            s, a = rust_type_layout(ty, structs, memo)
            off = (off + a - 1) // a * a
            fields.append((f, off, s))
            off += s
            align = max(align, a)
        memo[name] = ((off + align - 1) // align * align, align, fields)
    return memo[name]


def test_rust_mirror_matches_the_c_layout(tmp_path):
    """The #[repr(C)] structs of integration/rust/zkm_hip_sys.rs, laid out by the repr(C) rules (fields in order, each aligned to its
    type, size rounded to the struct's alignment), against the compiler's layout of the header -- names, offsets, sizes, total size."""
    lay = c_layouts(tmp_path)
    structs = rust_structs()
    opaque = {k for k, v in structs.items() if [f for f, _ in v] == ["_p"]}
    assert opaque == {"zkm_ctx", "zkm_batch", "zkm_pool", "zkm_staged"}
    assert set(structs) - opaque == set(lay), (set(structs) - opaque) ^ set(lay)
    memo = {}
    for name in lay:
        size, align, fields = rust_struct_layout(name, structs, memo)
        assert [list(f) for f in fields] == lay[name]["fields"], (name, fields, lay[name]["fields"])
        assert (size, align) == (lay[name]["size"], lay[name]["align"]), name


def build_c_smoke(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_smoke")
    libdir = os.path.join(ROOT, "zkm_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "c_smoke.c"), "-L" + libdir, "-lzkmhip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    return exe


def test_pure_c_caller_builds_and_reports_errors(zkm, oracle, tmp_path):
    """tools/c_smoke.c: the header used from pedantic C99 and linked against libzkmhip.so with no Python in the process.  Without a GPU
    the run must end in the library's error channel (nonzero status + malloc'd message), not in a crash."""
    import subprocess
    import numpy as np
    from zkm_amd import tables as T
    exe = build_c_smoke(tmp_path)
    seg = np.load(os.path.join(ROOT, "tests", "golden", "segment12.npz"))
    log_n = [int(x) for x in seg["log_n"]]
    ctl_tables, ctls = T.all_cross_table_lookups()
    tables = [(T.TABLE_ENUM_ORDER[i], seg["t%d" % i], T.WIDTH[T.TABLE_ENUM_ORDER[i]], log_n[i], ctl_tables[i]) for i in range(12)]
    img = zkm.segment_image(tables, ctls, public_values=[1, 2, 3])
    path = tmp_path / "segment.zkmtrace"
    img.tofile(path)
    r = subprocess.run([exe, str(path), str(tmp_path / "proofs.bin")], capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr
    else:
        assert r.returncode == 1 and "c_smoke: zkm_ctx_create:" in r.stderr, (r.returncode, r.stderr)
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"NOTATRACE" + b"\0" * 119)
    r = subprocess.run([exe, str(bad), str(tmp_path / "p.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "magic" in r.stderr


def test_every_tuning_key_is_documented_in_the_header():
    """zkm_ctx_set_tuning's keys live in core.hip; include/zkm_hip.h is what an integrator reads: the two lists must be the same, and
    INTEGRATION.md must not name a key that does not exist."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    core = open(os.path.join(root, "zkm_amd", "csrc", "core.hip")).read()
    body = core[core.index("int zkm_ctx_set_tuning("):]
    body = body[:body.index("ZKM_API_END")]
    code_keys = set(re.findall(r'k == "([a-z_0-9]+)"', body))
    header = open(os.path.join(root, "include", "zkm_hip.h")).read()
    doc_keys = set(re.findall(r'^ \*   "([a-z_0-9]+)"', header, flags=re.M))
    assert code_keys and code_keys == doc_keys, (sorted(code_keys - doc_keys), sorted(doc_keys - code_keys))
    integ = open(os.path.join(root, "INTEGRATION.md")).read()
    named = set(re.findall(r'zkm_ctx_set_tuning\([^)]*?"([a-z_0-9]+)"', integ)) | set(re.findall(r'set_tuning\("([a-z_0-9]+)"', integ))
    assert named <= code_keys, sorted(named - code_keys)


def test_every_tuning_key_is_documented_in_the_header():
    """zkm_ctx_set_tuning: the keys the library accepts (csrc/core.hip) are exactly the keys include/zkm_hip.h documents -- a knob added to
    one side only fails here."""
    core = open(os.path.join(ROOT, "zkm_amd", "csrc", "core.hip")).read()
    accepted = set(re.findall(r'k == "([a-z_0-9]+)"', core))
    header = open(os.path.join(ROOT, "include", "zkm_hip.h")).read()
    documented = set(re.findall(r'\*\s+"([a-z_0-9]+)"\s', header))
    assert accepted and accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))
