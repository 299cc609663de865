"""The Rust side of the boundary (integration/rust/*.rs, tools/ref_dump/ref_dump.rs) cannot be compiled in this image (no rustc /
cargo).  What CAN be checked without a compiler is checked here against the reference tree itself: every `crate::` item those files
import exists in /root/reference/prover/src with a visibility a file of the same crate may use, every struct literal names exactly
the fields the reference's struct has, every enum variant and every field reached through a typed binding exists, and every call of
an imported reference function passes as many arguments as its definition takes.  Skipped where the reference tree is absent (the
GPU box).  (plonky2:: paths cannot be resolved: the dependency is un-vendored, prover/Cargo.toml:17-20.)"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/prover/src"
# files that become part of the zkm-prover crate (`crate::` == prover/src); challenger_hip.rs / zkm_hip_sys.rs go into the plonky2 fork
CRATE_FILES = ["integration/rust/prove_hip.rs", "integration/rust/proof_blob.rs", "tools/ref_dump/ref_dump.rs"]
OWN_MODULES = {"proof_blob": "integration/rust/proof_blob.rs", "prove_hip": "integration/rust/prove_hip.rs"}

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (only checked where /root/reference exists)")


def strip_comments(src):
    """comments and the contents of string literals out of the way (file:line citations look like field accesses)"""
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', src)


def functions(src):
    """[(signature text, body text)] of every fn item of a source"""
    out = []
    for m in re.finditer(r"\bfn\s+[A-Za-z0-9_]+", src):
        i = src.find("{", m.end())
        semi = src.find(";", m.end())
        if i < 0 or (0 <= semi < i):
            continue
        depth, j = 1, i + 1
        while depth and j < len(src):
            depth += {"{": 1, "}": -1}.get(src[j], 0)
            j += 1
        out.append((src[m.start():i], src[i:j]))
    return out


def module_file(path):
    """crate::a::b -> prover/src/a/b.rs or a/b/mod.rs (or one of our own modules)"""
    if path and path[0] in OWN_MODULES and len(path) == 1:
        return os.path.join(ROOT, OWN_MODULES[path[0]])
    base = os.path.join(REF, *path)
    for cand in (base + ".rs", os.path.join(base, "mod.rs")):
        if os.path.isfile(cand):
            return cand
    return None


def item_visibility(src, name):
    """'pub' / 'pub(crate)' / 'private' / None for an item defined (or re-exported) in a module source"""
    m = re.search(r"^[ \t]*(pub(?:\([a-z: ]+\))?\s+)?(?:unsafe\s+|const\s+|async\s+)*(?:fn|struct|enum|trait|const|static|type|mod|union)\s+%s\b" % re.escape(name), src, flags=re.M)
    if m:
        v = (m.group(1) or "").strip()
        return "pub" if v == "pub" else ("pub(crate)" if v.startswith("pub(") else "private")
    if re.search(r"^[ \t]*pub(?:\([a-z: ]+\))?\s+use\s+[^;]*\b%s\b" % re.escape(name), src, flags=re.M):
        return "pub"
    return None


def crate_imports(src):
    """[(module path tuple, item)] of every `use crate::...;` (nested braces one level deep, `self` and `*` skipped)"""
    out = []
    for m in re.finditer(r"\buse\s+crate::([A-Za-z0-9_:]+)(?:::\{([^}]*)\})?\s*;", src):
        path = m.group(1).split("::")
        if m.group(2) is None:
            out.append((tuple(path[:-1]), path[-1]))
        else:
            for it in m.group(2).split(","):
                it = it.strip().split(" as ")[0].strip()
                if it and it not in ("self", "*"):
                    out.append((tuple(path), it))
    return out


def struct_fields(src, name):
    """{field: visibility} of `struct name {...}` in a module source, or None"""
    m = re.search(r"\bstruct\s+%s\b[^{;(]*\{" % re.escape(name), src)
    if not m:
        return None
    depth, i = 1, m.end()
    while depth and i < len(src):
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    body = src[m.end():i - 1]
    fields, depth, cur = {}, 0, ""
    for ch in body + ",":
        if ch in "<([{":
            depth += 1
        elif ch in ">)]}":
            depth -= 1
        if ch == "," and depth == 0:
            f = re.match(r"\s*(?:#\[[^\]]*\]\s*)*(pub(?:\([a-z: ]+\))?\s+)?([a-z_][A-Za-z0-9_]*)\s*:\s*(.*)$", cur.strip(), flags=re.S)
            if f:
                fields[f.group(2)] = ((f.group(1) or "private").strip(), f.group(3).strip())
            cur = ""
        else:
            cur += ch
    return fields


def enum_variants(src, name):
    m = re.search(r"\benum\s+%s\b[^{]*\{(.*?)\n\}" % re.escape(name), src, flags=re.S)
    return set(re.findall(r"^\s*([A-Z][A-Za-z0-9_]*)\b", m.group(1), flags=re.M)) if m else None


def fn_arity(src, name):
    m = re.search(r"\bfn\s+%s\b\s*(?:<[^{;]*?>)?\s*\(" % re.escape(name), src, flags=re.S)
    if not m:
        return None
    depth, i, args, cur = 1, m.end(), [], ""
    while depth and i < len(src):
        ch = src[i]
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>" and not (ch == ">" and src[i - 1] == "-"):
            depth -= 1
        if depth == 0:
            break
        if ch == "," and depth == 1:
            args.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        args.append(cur)
    return len([a for a in args if not re.match(r"\s*&?\s*(mut\s+)?self\b", a)])


def call_arities(src, name):
    """argument counts of every call `name(...)` / `name::<..>(...)` in a source (definitions excluded)"""
    out = []
    for m in re.finditer(r"(?<![A-Za-z0-9_])(?<!fn )%s\s*(?:::\s*<[^;{}]*?>)?\s*\(" % re.escape(name), src):
        if re.search(r"\bfn\s+$", src[max(0, m.start() - 4):m.start()]):
            continue
        depth, i, n, cur = 1, m.end(), 0, ""
        while depth and i < len(src):
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if depth == 0:
                break
            if ch == "," and depth == 1:
                n += 1
                cur = ""
            else:
                cur += ch
            i += 1
        out.append(n + (1 if cur.strip() else 0))
    return out


def sources():
    return {f: strip_comments(open(os.path.join(ROOT, f)).read()) for f in CRATE_FILES}


def resolve_all():
    """{item name: (module source, module file)} for every crate:: import of the integration files"""
    found, problems = {}, []
    for f, src in sources().items():
        for path, item in crate_imports(src):
            mf = module_file(list(path))
            if not mf:
                problems.append("%s: module crate::%s does not exist in the reference" % (f, "::".join(path)))
                continue
            msrc = strip_comments(open(mf).read())
            vis = item_visibility(msrc, item)
            if vis is None:
                problems.append("%s: crate::%s::%s is not defined in %s" % (f, "::".join(path), item, os.path.relpath(mf, "/root/reference")))
            elif vis == "private" and not mf.startswith(ROOT):
                problems.append("%s: crate::%s::%s is private in the reference (needs pub(crate))" % (f, "::".join(path), item))
            else:
                found[item] = (msrc, mf)
    return found, problems


def test_every_crate_import_resolves_in_the_reference():
    found, problems = resolve_all()
    assert not problems, "\n".join(problems)
    assert len(found) >= 25          # (the files import about thirty reference items: a parser that finds none proves nothing)


def test_struct_literals_name_the_reference_fields():
    found, _ = resolve_all()
    problems, checked = [], 0
    for f, src in sources().items():
        for name, (msrc, mf) in found.items():
            fields = struct_fields(msrc, name)
            if fields is None:
                continue
            # `Name { a: .., b, }` / `Name::<..> { .. }` in expression position (not `struct Name {`, not a `match` arm pattern with `..`)
            for m in re.finditer(r"(?<![A-Za-z0-9_])%s\s*(?:::\s*<[^{};]*>)?\s*\{" % re.escape(name), src):
                if re.search(r"\b(struct|impl|for|enum|trait)\s+$", src[max(0, m.start() - 8):m.start()]):
                    continue
                depth, i = 1, m.end()
                while depth and i < len(src):
                    depth += {"{": 1, "}": -1}.get(src[i], 0)
                    i += 1
                body = src[m.end():i - 1]
                used, d, cur = [], 0, ""
                for ch in body + ",":
                    if ch in "<([{":
                        d += 1
                    elif ch in ">)]}":
                        d -= 1
                    if ch == "," and d == 0:
                        g = re.match(r"\s*([a-z_][A-Za-z0-9_]*)\s*(?::|$)", cur.strip())
                        if g:
                            used.append(g.group(1))
                        cur = ""
                    else:
                        cur += ch
                partial = ".." in body
                checked += 1
                for u in used:
                    if u not in fields:
                        problems.append("%s: %s { %s } -- no such field in %s" % (f, name, u, os.path.relpath(mf, "/root/reference")))
                    elif fields[u][0] == "private":
                        problems.append("%s: %s.%s is a private field in the reference" % (f, name, u))
                if not partial and set(used) != set(fields) and not mf.startswith(ROOT):
                    problems.append("%s: %s literal sets %s, the reference struct has %s" % (f, name, sorted(used), sorted(fields)))
    assert not problems, "\n".join(problems)
    assert checked >= 4


def test_enum_variants_and_typed_field_accesses_exist():
    found, _ = resolve_all()
    problems, checked = [], 0
    for f, src in sources().items():
        for name, (msrc, mf) in found.items():
            variants = enum_variants(msrc, name)
            if variants:
                for v in set(re.findall(r"\b%s::([A-Z][A-Za-z0-9_]*)\b" % re.escape(name), src)):
                    checked += 1
                    if v not in variants and not re.search(r"\bfn\s+%s\b|\bconst\s+%s\b" % (v, v), msrc):
                        problems.append("%s: %s::%s is not a variant of the reference enum" % (f, name, v))
            fields = struct_fields(msrc, name)
            if not fields:
                continue
            # parameters typed with the struct (`x: &Name`, `x: Name<..>`, `x: &mut Name`), followed through the body of THEIR function
            for sig, body in functions(src):
              for b in set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*:\s*&?\s*(?:mut\s+)?%s\b" % re.escape(name), sig)):
                if re.search(r"\|[^|]*\b%s\b[^|]*\|" % re.escape(b), body):
                    continue      # (a closure parameter of the same name shadows it somewhere: not followed)
                for chain in set(re.findall(r"(?<![A-Za-z0-9_.])%s((?:\.[a-z_][A-Za-z0-9_]*)+)" % re.escape(b), body)):
                    cur_fields, cur_src = fields, msrc
                    for part in chain.strip(".").split("."):
                        if cur_fields is None:
                            break
                        if re.search(r"\bfn\s+%s\b" % re.escape(part), cur_src) or part not in cur_fields:
                            if part not in cur_fields and not re.search(r"\bfn\s+%s\b" % re.escape(part), cur_src) and part not in (
                                    "len", "iter", "clone", "as_ptr", "as_ref", "map", "to_vec", "first", "into_iter", "is_empty"):
                                problems.append("%s: %s%s -- `%s` is neither a field nor a method of %s in the reference" % (f, b, chain, part, name))
                            break
                        checked += 1
                        if cur_fields[part][0] == "private":
                            problems.append("%s: %s%s -- field `%s` is private in the reference" % (f, b, chain, part))
                        ty = re.match(r"&?\s*([A-Z][A-Za-z0-9_]*)", cur_fields[part][1])
                        cur_fields = struct_fields(cur_src, ty.group(1)) if ty else None
    assert not problems, "\n".join(problems)
    assert checked >= 15


def test_calls_of_reference_functions_pass_the_right_number_of_arguments():
    found, _ = resolve_all()
    problems, checked = [], 0
    for f, src in sources().items():
        for name, (msrc, mf) in found.items():
            want = fn_arity(msrc, name)
            if want is None or not re.search(r"\bfn\s+%s\b" % re.escape(name), msrc):
                continue
            for got in call_arities(src, name):
                checked += 1
                if got != want:
                    problems.append("%s: %s(...) called with %d arguments, the reference definition (%s) takes %d" % (
                        f, name, got, os.path.relpath(mf, "/root/reference") if not mf.startswith(ROOT) else os.path.relpath(mf, ROOT), want))
    assert not problems, "\n".join(problems)
    assert checked >= 5


# ------------------------------------------------------------------ the plonky2 side: paths and call-site arities
# plonky2 itself is un-vendored (prover/Cargo.toml:17-20), so its items cannot be looked up in source.  What the reference tree DOES hold
# is its own use of the pinned fork: every `use plonky2::a::b::Item` anywhere under /root/reference is proof that the item exists at that
# path in zkMIPS/plonky2@zkm_dev, and every call site of a PolynomialBatch method fixes that method's argument list.
PLONKY2_FILES = {   # file -> how it names the plonky2 crate ("crate" for files that go INTO the fork, "plonky2" for zkm-prover files)
    "integration/rust/oracle_hip.rs": "crate", "integration/rust/challenger_hip.rs": "crate",
    "integration/rust/prove_hip.rs": "plonky2", "integration/rust/proof_blob.rs": "plonky2", "tools/ref_dump/ref_dump.rs": "plonky2",
}
# items of plonky2 0.1.4 the reference never imports by name (it reaches them through type inference or not at all): the MODULE must still
# be one the reference imports from, unless listed here with the module
KNOWN_UNIMPORTED = {
    ("hash", "hash_types", "HashOut"), ("fri", "proof", "FriInitialTreeProof"), ("fri", "proof", "FriQueryRound"), ("fri", "proof", "FriQueryStep"),
    ("hash", "poseidon", "PoseidonHash"), ("hash", "merkle_proofs", "MerkleProof"), ("field", "fft", "FftRootTable"),
    ("fri", "structure", "FriOracleInfo"), ("fri", "structure", "FriPolynomialInfo"),
}
KNOWN_UNIMPORTED_MODULES = {("hash", "merkle_proofs"), ("field", "fft")}
OURS = {"hip"}          # plonky2::hip::* is the module this integration adds to the fork


def use_items(src, root):
    """[(module path tuple, item)] of every `use <root>::...;` -- nested braces one level deep, multi-line lists, `self` / `*` skipped"""
    out = []
    for m in re.finditer(r"\buse\s+%s::([A-Za-z0-9_:]+?)(?:::\{([^}]*)\})?\s*;" % re.escape(root), src, flags=re.S):
        path = m.group(1).split("::")
        if m.group(2) is None:
            out.append((tuple(path[:-1]), path[-1]))
        else:
            for it in m.group(2).replace("\n", " ").split(","):
                it = it.strip().split(" as ")[0].strip()
                if it and it not in ("self", "*"):
                    out.append((tuple(path), it))
    return out


def reference_plonky2_imports():
    seen = set()
    for base, _, files in os.walk("/root/reference"):
        if "/target" in base or "/.git" in base:
            continue
        for fn in files:
            if fn.endswith(".rs"):
                try:
                    src = strip_comments(open(os.path.join(base, fn), errors="replace").read())
                except OSError:
                    continue
                seen.update(use_items(src, "plonky2"))
    return seen


_REF_TEXT = []


def ref_text():
    if not _REF_TEXT:
        parts = []
        for base, _, files in os.walk(REF):
            parts += [strip_comments(open(os.path.join(base, fn)).read()) for fn in files if fn.endswith(".rs")]
        _REF_TEXT.append("\n".join(parts))
    return _REF_TEXT[0]


def test_every_plonky2_path_is_one_the_reference_itself_uses():
    ref = reference_plonky2_imports()
    assert len(ref) >= 60
    ref_modules = {mod for mod, _ in ref}
    problems, checked = [], 0
    for f, root in PLONKY2_FILES.items():
        src = strip_comments(open(os.path.join(ROOT, f)).read())
        for mod, item in use_items(src, root):
            if root == "crate" and (not mod or mod[0] in OURS):
                continue
            if root == "plonky2" and mod and mod[0] in OURS:
                continue
            if root == "crate" and f.startswith("integration/rust/") and PLONKY2_FILES[f] == "crate" and not mod:
                continue
            checked += 1
            if (mod, item) in ref:
                continue
            if mod + (item,) in KNOWN_UNIMPORTED and (mod in ref_modules or mod in KNOWN_UNIMPORTED_MODULES):
                continue
            if len(mod) >= 2 and (mod[:-1], mod[-1]) in ref and ("%s::%s" % (mod[-1], item)) in ref_text():
                continue            # an enum variant / associated item of an imported type, spelled that way in the reference itself
            where = sorted("::".join(m) for m, i in ref if i == item)
            problems.append("%s: plonky2::%s::%s -- the reference never imports that path%s" % (
                f, "::".join(mod), item, (" (it imports %s from plonky2::%s)" % (item, ", plonky2::".join(where))) if where else ""))
    assert not problems, "\n".join(problems)
    assert checked >= 40


def reference_call_arities(method):
    """argument counts of every `PolynomialBatch::[<..>::]method(` (associated functions) or `.method(` (methods) call under prover/src"""
    out = []
    for base, _, files in os.walk(REF):
        for fn in files:
            if not fn.endswith(".rs"):
                continue
            src = strip_comments(open(os.path.join(base, fn)).read())
            # mark the qualified / method calls with a unique name, then count the arguments of the marked calls only
            marked = re.sub(r"PolynomialBatch\s*(?:::\s*<[^;{}()]*?>)?\s*::\s*%s\b" % re.escape(method), "ZKMSITE_" + method, src)
            marked = re.sub(r"\.\s*%s\b(?=\s*\()" % re.escape(method), ".ZKMSITE_" + method, marked)
            out += call_arities(marked, "ZKMSITE_" + method)
    return out


def test_polynomial_batch_facade_takes_what_the_reference_call_sites_pass():
    """integration/rust/oracle_hip.rs: HipPolynomialBatch::{from_values, from_coeffs, get_lde_values_packed, prove_openings} have the
    argument lists of the reference's call sites (prover.rs:154-163, 514-521, 579-586, 621-627, 687, 723-748 and the six benchmark
    tests), and every zkm_* function the facade calls is declared in zkm_hip_sys.rs with that many parameters."""
    src = strip_comments(open(os.path.join(ROOT, "integration", "rust", "oracle_hip.rs")).read())
    for method, min_sites in (("from_values", 8), ("from_coeffs", 1), ("get_lde_values_packed", 5), ("prove_openings", 1)):
        sites = reference_call_arities(method)
        assert len(sites) >= min_sites and len(set(sites)) == 1, (method, sites)
        assert fn_arity(src, method) == sites[0], "%s: the facade takes %s arguments, the reference passes %d" % (method, fn_arity(src, method), sites[0])
    # the fields the reference reads become methods here: both must exist
    for name in ("cap", "polynomials", "hip_batch", "get_lde_values"):
        assert re.search(r"\bpub fn %s\b" % name, src), name
    assert re.search(r"\bDrop\s+for\s+HipPolynomialBatch", src) and "zkm_batch_free" in src
    sys_src = strip_comments(open(os.path.join(ROOT, "integration", "rust", "zkm_hip_sys.rs")).read())
    checked = 0
    for f in ("integration/rust/oracle_hip.rs", "integration/rust/prove_hip.rs"):
        body = strip_comments(open(os.path.join(ROOT, f)).read())
        for name in sorted(set(re.findall(r"\b(zkm_[a-z0-9_]+)\s*\(", body))):
            if re.search(r"\bfn\s+%s\b" % name, body):
                continue        # (a Rust helper of the file itself: zkm_config, zkm_table_id)
            want = fn_arity(sys_src, name)
            assert want is not None, "%s calls %s, which zkm_hip_sys.rs does not declare" % (f, name)
            for got in call_arities(body, name):
                checked += 1
                assert got == want, "%s: %s called with %d arguments, declared with %d" % (f, name, got, want)
    assert checked >= 15
