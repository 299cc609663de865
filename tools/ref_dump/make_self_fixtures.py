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


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "ref_dump_self")
