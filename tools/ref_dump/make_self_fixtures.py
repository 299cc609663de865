#!/usr/bin/env python3
"""Writes fixtures with the SCHEMA of tools/ref_dump/ref_dump.rs, but computed by the CPU oracle.

Purpose: exercise tests/test_reference_fixtures.py end to end without a Rust toolchain (the test generates these into a
temporary directory and checks the oracle / HIP path against them).  These are NOT reference data and are never committed
under tests/golden/ -- only files written by ref_dump.rs on a machine that can build the reference belong there.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
P = 0xFFFFFFFF00000001
SHIFT = 14293326489335486720
M64 = (1 << 64) - 1


def splitmix_at(seed, k):
    """oracle/prover.c: splitmix_at; k may be a numpy uint64 array."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.asarray(k, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def felts(seed, k0, count):
    x = splitmix_at(seed, np.arange(k0, k0 + count, dtype=np.uint64))
    return np.where(x >= np.uint64(P), x - np.uint64(P), x)


def seeded_columns(seed, ncols, n):
    return felts(seed, 1, ncols * n)  # column-major: element (r, c) = felt(seed, c n + r + 1)


def ints(a):
    return [int(x) for x in np.asarray(a).reshape(-1)]


def main(out_dir):
    from oracle.oracle_py import Oracle
    o = Oracle()
    os.makedirs(out_dir, exist_ok=True)
    ntt = []
    for log_n in (3, 5, 8):
        n, seed = 1 << log_n, 100 + log_n
        cols = seeded_columns(seed, 3, n)
        rows = lambda a: [ints(a[c * n:(c + 1) * n]) for c in range(3)]
        ntt.append({"log_n": log_n, "ncols": 3, "seed": seed, "coset_shift": SHIFT, "fft": rows(o.ntt(cols, log_n)),
                    "ifft": rows(o.ntt(cols, log_n, inverse=True)), "coset_fft": rows(o.ntt(cols, log_n, coset_shift=SHIFT)),
                    "coset_ifft": rows(o.ntt(cols, log_n, inverse=True, coset_shift=SHIFT))})
    hashes = []
    for ln in (0, 1, 4, 5, 8, 9, 16, 17, 262):
        x = felts(300, 1, ln)
        hashes.append({"len": ln, "seed": 300, "hash_no_pad": ints(o.hash_no_pad(x)), "hash_or_noop": ints(o.hash_or_noop(x))})
    lr = felts(301, 1, 8)
    commits = []
    for ncols, log_n in ((13, 5), (262, 5), (4, 6), (3, 4)):
        n, seed = 1 << log_n, 200 + ncols
        b = o.batch_from_values(seeded_columns(seed, ncols, n), ncols, log_n)
        nn = 4 * n
        li, pi = [0, 1, 3, nn // 2 + 3, nn - 1], [0, 5, nn - 1]
        co = b.coeffs()
        b2 = o.batch_from_coeffs(co, ncols, log_n)
        commits.append({"ncols": ncols, "log_n": log_n, "seed": seed, "rate_bits": 2, "cap_height": 4,
                        "coeffs": [ints(co[c * n:(c + 1) * n]) for c in range(ncols)], "cap": ints(b.cap()), "cap_from_coeffs": ints(b2.cap()),
                        "leaf_indices": li, "leaves": [ints(b.leaf(i)) for i in li], "lde_rows_natural_index": [ints(b.lde_row(i)) for i in li],
                        "path_indices": pi, "paths": [ints(b.merkle_path(i)) for i in pi]})
    ch = o.challenger()
    script, k = [], 1
    for nobs, nget in ((3, 2), (8, 1), (9, 9), (0, 3), (17, 1), (1, 8)):
        obs = felts(400, k + 1, nobs)
        k += nobs
        o.observe(ch, obs)
        script.append({"observe": ints(obs), "get": [int(o.challenge(ch)) for _ in range(nget)]})
    ext = [int(o.challenge(ch)), int(o.challenge(ch))]
    o.observe(ch, felts(400, 1000, 1))
    st = np.zeros(12, dtype=np.uint64)
    o.lib.zko_challenger_compact(C.byref(ch), st.ctypes.data_as(C.POINTER(C.c_uint64)))
    json.dump({"schema": 1, "source": "SELF (CPU oracle) -- not reference data", "ntt": ntt, "hash": hashes,
               "two_to_one": {"seed": 301, "out": ints(o.two_to_one(lr[:4], lr[4:]))}, "commit": commits,
               "challenger": {"seed": 400, "script": script, "extension_challenge": ext, "compact_state": ints(st),
                              "challenge_after_compact": int(o.challenge(ch))}},
              open(os.path.join(out_dir, "reference_primitives.json"), "w"))
    # PoseidonStark proof
    seed, num_perms, log_n = 7, 125, 7
    trace = o.poseidon_trace(seed, num_perms, log_n)
    n = 1 << log_n
    och = o.challenger()
    blob = o.prove(trace, log_n, np.zeros(4 * n, dtype=np.uint64), [1, 1], challenger=och)
    st = np.zeros(12, dtype=np.uint64)
    o.lib.zko_challenger_compact(C.byref(och), st.ctypes.data_as(C.POINTER(C.c_uint64)))
    json.dump({"schema": 1, "table": "PoseidonStark", "seed": seed, "num_perms": num_perms, "log_n": log_n, "ncols": 262,
               "aux": "zeros, 2 x (1 helper + Z)", "num_helpers": [1, 1], "trace_column_0": ints(trace[:n]), "trace_column_last": ints(trace[261 * n:]),
               "challenger_after": {"state": ints(st)}, "blob": ints(blob)}, open(os.path.join(out_dir, "reference_poseidon_proof.json"), "w"))
    # KeccakStark proof
    seed, num_perms, log_n = 11, 5, 7
    inputs = splitmix_at(seed, np.arange(1, 25 * num_perms + 1, dtype=np.uint64)).reshape(num_perms, 25)
    trace = o.keccak_trace(inputs, np.zeros(num_perms, dtype=np.uint64), log_n)
    n = 1 << log_n
    blob = o.prove(trace, log_n, np.zeros(4 * n, dtype=np.uint64), [1, 1], ncols=2431, table_id=3)
    w, a, q, z, cap, layers, fin, nq = (int(blob[i]) for i in (2, 3, 4, 5, 6, 7, 8, 9))
    cap = 1 << cap
    upto = 16 + 12 + 3 * cap * 4 + 4 * w + 4 * a + z + 2 * q + layers * cap * 4 + 2 * fin + 1
    rw = (blob.size - upto) // nq
    json.dump({"schema": 1, "table": "KeccakStark", "seed": seed, "num_perms": num_perms, "log_n": log_n, "ncols": 2431,
               "aux": "zeros, 2 x (1 helper + Z)", "num_helpers": [1, 1], "blob_words": int(blob.size), "blob_up_to_pow": ints(blob[:upto]),
               "first_query_round": ints(blob[upto:upto + rw])}, open(os.path.join(out_dir, "reference_keccak_proof.json"), "w"))


TABLE_NAMES = ["Arithmetic", "Cpu", "Poseidon", "PoseidonSponge", "Keccak", "KeccakSponge", "ShaExtend", "ShaExtendSponge", "ShaCompress",
               "ShaCompressSponge", "Logic", "Memory"]   # Table::all(), all_stark.rs:117-134


def write_traces_bin(path, traces, ncols, log_n):
    """ref_dump.rs::ref_dump_all_proof's trace file: magic, ntables, (ncols, log_n) per table, then the tables column-major."""
    with open(path, "wb") as f:
        f.write(b"ZKMTBLS1")
        f.write(np.array([len(traces)] + [v for t in range(len(traces)) for v in (ncols[t], log_n[t])], dtype="<u8").tobytes())
        for t in traces:
            f.write(np.ascontiguousarray(t, dtype="<u8").tobytes())


def read_traces(path):
    """-> (traces, ncols, log_n) from the dumper's .bin or from the .npz tools/ref_dump/pack_traces.py makes of it."""
    if path.endswith(".npz"):
        z = np.load(path)
        k = len(z["log_n"])
        return [z["t%d" % i] for i in range(k)], [int(x) for x in z["ncols"]], [int(x) for x in z["log_n"]]
    raw = np.fromfile(path, dtype="<u8")
    assert raw[0] == int.from_bytes(b"ZKMTBLS1", "little"), "not a ZKMTBLS1 trace file"
    k = int(raw[1])
    ncols, log_n = [int(x) for x in raw[2:2 + 2 * k:2]], [int(x) for x in raw[3:3 + 2 * k:2]]
    out, off = [], 2 + 2 * k
    for t in range(k):
        words = ncols[t] << log_n[t]
        out.append(raw[off:off + words].astype(np.uint64))
        off += words
    assert off == raw.size, "trailing words in the trace file"
    return out, ncols, log_n


def blob_up_to_pow(blob):
    w, a, q, z, cap, layers, fin = (int(blob[i]) for i in (2, 3, 4, 5, 6, 7, 8))
    cap = 1 << cap
    return 16 + 12 + 3 * cap * 4 + 4 * w + 4 * a + z + 2 * q + layers * cap * 4 + 2 * fin + 1


def main_all_proof(out_dir):
    """reference_all_proof.json + reference_all_proof_traces.bin with the schema of ref_dump.rs::ref_dump_all_proof, computed by the
    CPU oracle from the committed twelve-table test segment (tests/golden/segment12.npz).  SELF data: schema exercise only."""
    from oracle.oracle_py import Oracle
    from zkm_amd import ctl as zc
    from zkm_amd import tables as T
    o = Oracle()
    os.makedirs(out_dir, exist_ok=True)
    seg = np.load(os.path.join(ROOT, "tests", "golden", "segment12.npz"))
    log_n = [int(x) for x in seg["log_n"]]
    traces = [seg["t%d" % i] for i in range(12)]
    ncols = [T.WIDTH[T.TABLE_ENUM_ORDER[i]] for i in range(12)]
    write_traces_bin(os.path.join(out_dir, "reference_all_proof_traces.bin"), traces, ncols, log_n)
    ctl_tables, ctls = T.all_cross_table_lookups()
    tables = [(T.TABLE_ENUM_ORDER[i], traces[i], ncols[i], log_n[i], ctl_tables[i]) for i in range(12)]
    pub = list(range(1, 9)) + list(range(11, 19)) + [i & 0xFF for i in range(32)]   # 8 + 8 root limbs, 32 userdata bytes
    proofs, chal, offs = o.prove_with_traces(tables, ctls, public_values=pub)
    caps, plist, ctl_dump = [], [], []
    for t in range(12):
        blob = proofs[offs[t]:offs[t + 1]]
        c4 = (1 << int(blob[6])) * 4
        caps.append(ints(blob[28:28 + c4]))
        plist.append({"blob_words": int(blob.size), "blob_up_to_pow": ints(blob[:blob_up_to_pow(blob)])})
    for t, (zs, ids) in enumerate(zc.derive_zs(12, ctls, chal)):
        n = 1 << log_n[t]
        aux = o.ctl_data(ctl_tables[t], zs, ids, traces[t], ncols[t], log_n[t]).reshape(-1, n)
        nh = [int(x) for x in zs["num_helpers"]]
        hoff = np.concatenate([[0], np.cumsum(nh)]).astype(int)
        zcols = aux[sum(nh):]
        ctl_dump.append({"num_zs": len(zs), "num_helpers": nh, "num_colsets": [int(x) for x in zs["ncolsets"]],
                         "beta": ints(zs["beta"]), "gamma": ints(zs["gamma"]), "z_first": ints(zcols[:, 0]), "z_last": ints(zcols[:, n - 1]),
                         "helper_first": [ints(aux[hoff[k]:hoff[k + 1], 0]) for k in range(len(zs))]})
    json.dump({"schema": 1, "source": "SELF (CPU oracle) -- not reference data",
               "program": {"elf": "tests/golden/segment12.npz", "args": [], "seg_size": 0, "segment": 0, "segments": 1, "total_steps": 0},
               "config": {"rate_bits": 2, "cap_height": 4, "num_challenges": 2},
               "tables": [{"name": TABLE_NAMES[i], "enum_index": i, "ncols": ncols[i], "log_n": log_n[i]} for i in range(12)],
               "traces_file": "reference_all_proof_traces.bin", "public_values_words": pub, "trace_caps": caps, "ctl_challenges": ints(chal),
               "ctl_data": ctl_dump, "proofs": plist}, open(os.path.join(out_dir, "reference_all_proof.json"), "w"))


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else "ref_dump_self"
    main(d)
    main_all_proof(d)
