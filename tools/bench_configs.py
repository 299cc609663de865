#!/usr/bin/env python3
"""Timings of BASELINE.json configs 4 and 5 (the two configs that are defined as measurements, not as the headline).

  config 4  "Full FRI commit+fold on 2^22-row trace, 1 GPU, rocprof HBM GB/s vs roofline"
            = PolynomialBatch::prove_openings (prover.rs:618-628, instance stark.rs:91-148) on 13 trace + 4 auxiliary + 4 quotient
            polynomials of 2^22 coefficients (MemoryStark width): alpha-combination, division by (X - z), final-polynomial
            LDE to 2^24 F2 values, commit phase (five arity-16 folds, 4 final coefficients), proof of work, 37 query rounds.
  config 5  "Keccak-sponge STARK table (precompile path), 2^20 rows, Keccak-f[1600] HIP kernel + Merkle cap"
            = KeccakSpongeStark::generate_trace (keccak_sponge_stark.rs:222-444, keccakf_u32s at :410) on seeded random messages,
            then the trace commitment (from_values) of the 470 x 2^20 table; plus the plain Keccak-f batch kernel (K15).

  python tools/bench_configs.py fri|sponge|all       one JSON object per config (bench.py quotes them as `fri_2_22` / `keccak_sponge_2_20`)
Parity for both configs is in tests/ (test_prove_openings_bit_exact[22], test_config5_*): this file only times.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

P = 0xFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0


def fri_algorithmic_bytes(n, W, A, Q, Z, rate_bits=2, arity_bits=4, layers=5):
    """SURVEY.md 8(d), rows K11-K13, for one prove_openings call."""
    k11 = 8 * n * (2 * (W + A) + Q + Z) + 16 * n          # every coefficient polynomial of the three batches read, final F2 poly written
    k12 = 128 * n                                          # 2 base-field NTTs of length 4n: 2 x 16 x 4n
    k13, m = 0, n << rate_bits
    for _ in range(layers):
        folded = m >> arity_bits
        k13 += 16 * m + 16 * folded + 2 * folded * 32      # read the layer, write the folded values and the digests of its leaves
        k13 += 2 * 16 * folded                             # coset re-evaluation of the folded layer (the next layer's LDE)
        m = folded
    return {"K11_combine": k11, "K12_final_lde": k12, "K13_commit_fold": k13, "total": k11 + k12 + k13}


PMC_SYMBOLS = {"fri_combine": ["k_fri_combine"], "fri_fold": ["k_fri_fold"],
               "fri_divide_linear": ["k_seg_totals", "k_seg_scan", "k_seg_scan_final", "k_seg_combine"]}


def load_pmc(name):
    """profiles/<name> (tools/summarize_config_pmc.py) if it was collected on THIS code (bench.py's fingerprint), else (None, False)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        from bench import code_fingerprint
        return d, d.get("code_fingerprint") == code_fingerprint()
    except Exception:  # noqa: BLE001
        return None, False


def fri_2_22(ctx, log_n=22, reps=3):
    import zkm_amd
    W, A, Q, Z = 13, 4, 4, 2
    n = 1 << log_n
    rng = np.random.default_rng(4)
    tv, av, qc = (rng.integers(0, P, k * n, dtype=np.uint64) for k in (W, A, Q))
    ctx.synchronize()
    t0 = time.perf_counter()
    tb = zkm_amd.PolynomialBatch.from_values(ctx, ctx.alloc(tv.size).upload(tv), W, log_n)
    ab = zkm_amd.PolynomialBatch.from_values(ctx, ctx.alloc(av.size).upload(av), A, log_n)
    qb = zkm_amd.PolynomialBatch.from_coeffs(ctx, ctx.alloc(qc.size).upload(qc), Q, log_n)
    ctx.synchronize()
    commit_first_s = time.perf_counter() - t0              # includes the uploads and the first-use tables
    proof = ctx.prove_openings(tb, ab, qb, Z)               # warm-up
    ctx.profile(True)
    ctx.profile_reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        proof = ctx.prove_openings(tb, ab, qb, Z)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rec = ctx.profile_records()
    ctx.profile(False)
    # commit time of the three oracles alone (device-resident inputs, warm tables)
    dv = ctx.alloc(tv.size).upload(tv)
    ctx.synchronize()
    t0 = time.perf_counter()
    tb2 = zkm_amd.PolynomialBatch.from_values(ctx, dv, W, log_n)
    ctx.synchronize()
    commit_trace_s = time.perf_counter() - t0
    tb2.free()
    dv.free()
    alg = fri_algorithmic_bytes(n, W, A, Q, Z)
    kernel_ms = {k: round(v[1] / reps, 3) for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1]) if not k.startswith("stage/")}
    # Per kernel, against the HBM roofline: the bytes the kernel must move (each input word read once, each output word written once)
    # / its own HIP-event time.  (SURVEY 8(d)'s K11 figure counts the trace and auxiliary polynomials once per opening batch; the
    # combination kernel reads every coefficient ONCE and forms all batches in that pass, so its own figure is smaller than K11.)
    folds, m = 0, n
    for _ in range(5):
        folds += 16 * m + 16 * (m >> 4)
        m >>= 4
    kernel_alg = {"fri_combine": 8 * n * (W + A + Q) + 3 * 16 * n,          # W + A + Q polynomials in, three F2 composites out
                  "fri_divide_linear": 3 * 16 * n + 16 * n,                  # three composites in, the final F2 polynomial out
                  "fri_fold": folds}                                          # every layer's coefficients in, 1/16 of them out
    pmc, fresh = load_pmc("fri_2_22_pmc.json")
    per_kernel = {}
    for name, syms in PMC_SYMBOLS.items():
        if name not in rec or not rec[name][1]:
            continue
        ms = rec[name][1] / reps
        e = {"ms_per_call": round(ms, 4), "launches_per_call": rec[name][0] / reps, "algorithmic_bytes": kernel_alg[name],
             "achieved_GBps": kernel_alg[name] / ms / 1e6, "frac_of_hbm_peak": kernel_alg[name] / ms / 1e6 / HBM_PEAK_GBS}
        if fresh:
            calls = pmc.get("calls_profiled", reps + 1)
            tr = sum(pmc["kernels"].get(sy, {}).get("fetch_size_bytes_total", 0) + pmc["kernels"].get(sy, {}).get("write_size_bytes_total", 0) for sy in syms)
            e["traffic_bytes"] = tr / calls
            e["traffic_GBps"] = tr / calls / ms / 1e6
            e["traffic_frac_of_hbm_peak"] = tr / calls / ms / 1e6 / HBM_PEAK_GBS
        per_kernel[name] = e
    out = {"workload": "prove_openings on 13 + 4 + 4 polynomials of 2^%d coefficients (seed 4), LDE 2^%d F2 values, folds 4 x %d, PoW 16 bits, "
                       "37 queries; BASELINE.json configs[3]" % (log_n, log_n + 2, 5),
           "ms": dt * 1e3, "proof_words": int(proof.size), "kernel_ms": kernel_ms, "kernel_ms_sum": round(sum(kernel_ms.values()), 3),
           "launches": int(sum(v[0] for k, v in rec.items() if not k.startswith("stage/")) / reps),
           "per_kernel_hbm": per_kernel,
           "algorithmic_bytes_survey_8d": alg,
           "commit_trace_13_cols_ms": commit_trace_s * 1e3, "commit_three_oracles_first_call_ms": commit_first_s * 1e3,
           "note": "per_kernel_hbm: each FRI kernel's own bytes / its own HIP-event time against 8 TB/s (traffic_* = FETCH_SIZE x 2 + WRITE_SIZE of "
                   "separate rocprofv3 --pmc passes, %s); the call as a whole is not an HBM figure -- over half of its kernel time is "
                   "Poseidon hashing of the five layer trees (merkle_leaves_ext, merkle_compress), which is VALU-bound" %
                   ("profiles/fri_2_22_pmc.json" if fresh else "absent: no fresh counter file for this code, run tools/gpu_configs.sh")}
    for b in (tb, ab, qb):
        b.free()
    return out


def sponge_ops(seed, target_rows, max_len=1088):
    """Seeded random messages, lengths uniform in [1, 1088) bytes, as many as fit in target_rows rows."""
    nops = max(1, int(target_rows / 4.6))
    while True:
        rng = np.random.default_rng(seed)
        lens = rng.integers(1, max_len, nops)
        rows = int(np.sum(lens // 136 + 1))
        if rows <= target_rows:
            break
        nops = int(nops * 0.98)
    off = np.zeros(nops + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    data = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
    meta = np.zeros((nops, 4), dtype=np.uint64)
    meta[:, 1] = 3
    meta[:, 2] = rng.integers(0, 1 << 24, nops)
    meta[:, 3] = np.arange(nops) * 7 + 1
    return data, off, meta.reshape(-1), rows, nops


def keccak_sponge_2_20(ctx, log_n=20, reps=3):
    import zkm_amd
    W = 470
    n = 1 << log_n
    data, off, meta, rows, nops = sponge_ops(5, n)
    buf, used = ctx.keccak_sponge_trace(data, off, meta, log_n)        # warm-up
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.keccak_sponge_trace(data, off, meta, log_n, out=buf)
    ctx.synchronize()
    witness_host_s = (time.perf_counter() - t0) / reps                  # message bytes (124 MB) in pageable host memory: upload on the clock
    # the measured configuration: message bytes resident in HBM, like every other input of this file
    padded = np.zeros((data.size + 7) // 8 * 8, dtype=np.uint8)
    padded[:data.size] = data
    d_data = ctx.alloc(padded.size // 8).upload(padded.view(np.uint64))
    ctx.keccak_sponge_trace(d_data, off, meta, log_n, out=buf)
    ctx.profile(True)
    ctx.profile_reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.keccak_sponge_trace(d_data, off, meta, log_n, out=buf)
    ctx.synchronize()
    witness_s = (time.perf_counter() - t0) / reps
    rec_w = ctx.profile_records()
    d_data.free()
    b = zkm_amd.PolynomialBatch.from_values(ctx, buf, W, log_n)          # warm-up
    b.free()
    ctx.profile_reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b = zkm_amd.PolynomialBatch.from_values(ctx, buf, W, log_n)
        ctx.synchronize()
        b.free()
    commit_s = (time.perf_counter() - t0) / reps
    rec_c = ctx.profile_records()
    # K15: the plain Keccak-f[1600] batch kernel on device-resident states (400 B per permutation: 200 in + 200 out)
    k = 1 << 22
    st = ctx.alloc(25 * k).upload(np.random.default_rng(15).integers(0, 1 << 63, 25 * k, dtype=np.uint64))
    ctx.keccakf_batch(st)
    ctx.profile_reset()
    for _ in range(reps):
        ctx.keccakf_batch(st)
    rec_k = ctx.profile_records()
    ctx.profile(False)
    st.free()
    buf.free()
    kw_ms = rec_w["keccak_sponge_trace"][1] / rec_w["keccak_sponge_trace"][0]
    kf_ms = rec_k["keccakf"][1] / rec_k["keccakf"][0]
    return {"workload": "KeccakSpongeStark 470 cols x 2^%d rows: %d seeded random messages (seed 5, lengths uniform in [1, 1088) bytes), %d rows used "
                        "= %d Keccak-f permutations; witness kernel, then from_values; BASELINE.json configs[4]" % (log_n, nops, rows, rows),
            # (`witness_ms` keeps the meaning it had through round 3 -- host-resident message bytes, upload on the clock --
            # and the device-resident variant has its own key, so same-named keys compare like with like across rounds)
            "witness_ms": witness_host_s * 1e3, "witness_kernel_ms": kw_ms, "witness_ms_device_inputs": witness_s * 1e3,
            "witness_note": "witness_ms: the %.0f MB of message bytes start in pageable host memory and are uploaded inside the call (the key's "
                            "meaning in rounds 1-3; round 4 reported the device-resident figure under this name); witness_ms_device_inputs: message "
                            "bytes resident in HBM (offsets and per-operation words still come from the host: 3.6 MB)" % (data.size / 1e6),
            "keccakf_permutations": rows, "witness_permutations_per_s": rows / (kw_ms / 1e3),
            "witness_bytes_written": 8 * W * n, "witness_call_GBps": 8 * W * n / witness_host_s / 1e9,
            "witness_call_GBps_device_inputs": 8 * W * n / witness_s / 1e9,
            "commit_ms": commit_s * 1e3, "commit_kernel_ms": {k2: round(v[1] / reps, 3) for k2, v in sorted(rec_c.items(), key=lambda kv: -kv[1][1]) if not k2.startswith("stage/")},
            "keccakf_batch": {"states": k, "ms": kf_ms, "permutations_per_s": k / (kf_ms / 1e3), "GBps": 400.0 * k / (kf_ms / 1e3) / 1e9,
                              "frac_of_hbm_peak": 400.0 * k / (kf_ms / 1e3) / 1e9 / HBM_PEAK_GBS}}


if __name__ == "__main__":
    import zkm_amd
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    c = zkm_amd.Context(0)
    out = {}
    if which in ("fri", "all"):
        out["fri_2_22"] = fri_2_22(c)
    if which in ("sponge", "all"):
        out["keccak_sponge_2_20"] = keccak_sponge_2_20(c)
    print(json.dumps(out, indent=1))
    c.close()
