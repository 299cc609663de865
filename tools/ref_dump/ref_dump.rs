//! ref_dump.rs -- reference-side fixture dumper: pins the CPU oracle (and through it the HIP path) to plonky2 BYTES.
//!
//! The build image has no Rust toolchain, so `oracle/` is pinned to the reference only at the primitive level
//! (Keccak-f KAT, Poseidon constants / test vectors, SHA table vectors); the pipeline conventions of the un-vendored plonky2
//! fork (FFT order, bit-reversed leaves, overwrite-mode sponge, Challenger pop order, FRI fold order, PoW
//! position) are recalled.  This file closes that gap the moment anybody can run `cargo test` on the reference:
//!
//!   1. cp tools/ref_dump/ref_dump.rs        $REF/prover/src/ref_dump.rs
//!      cp integration/rust/proof_blob.rs    $REF/prover/src/proof_blob.rs
//!      add to $REF/prover/src/lib.rs:       pub mod proof_blob;   #[cfg(test)] mod ref_dump;
//!   2. ZKM_REF_DUMP_DIR=/tmp/ref cargo test --release -p zkm-prover ref_dump -- --nocapture
//!   3. cp /tmp/ref/reference_*.json /tmp/ref/reference_all_proof_traces.bin <this repo>/tests/golden/
//!      (the trace file of the whole-segment dump is ~10 MB at the default 1024-cycle segment; `python tools/ref_dump/pack_traces.py`
//!      turns it into a compressed .npz a fraction of that size -- the tests read either)
//!   4. python -m pytest tests/test_reference_fixtures.py            (CPU: oracle vs fixtures)
//!      python -m pytest tests/test_reference_fixtures.py -m gpu     (GPU: HIP path vs fixtures)
//!
//! Everything is seeded with SplitMix64 exactly like `oracle/prover.c: splitmix_at` / the `zkm_poseidon_trace` kernel, so the
//! fixtures hold outputs only; the Python side regenerates the inputs.  Reference entry points used:
//!   PolynomialValues::{ifft, coset_ifft}, PolynomialCoeffs::{fft, coset_fft, lde}   (plonky2 field/polynomial, field/fft.rs)
//!   PolynomialBatch::from_values / from_coeffs, merkle_tree.{cap, leaves, prove}     (plonky2 fri/oracle.rs, hash/merkle_tree.rs)
//!   PoseidonHash::{hash_no_pad, hash_or_noop, two_to_one}, Challenger                  (plonky2 hash/poseidon.rs, iop/challenger.rs)
//!   PoseidonStark / KeccakStark generate_trace + prove_single_table                    (poseidon_stark.rs:751-816, keccak_stark.rs:689-754)
//! NOT COMPILED in the build image.
use std::fs;
use std::path::PathBuf;

use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::polynomial::{PolynomialCoeffs, PolynomialValues};
use plonky2::field::types::{Field, PrimeField64};
use plonky2::fri::oracle::PolynomialBatch;
use plonky2::hash::hash_types::HashOut;
use plonky2::hash::poseidon::PoseidonHash;
use plonky2::iop::challenger::Challenger;
use plonky2::plonk::config::{GenericConfig, Hasher, PoseidonGoldilocksConfig};
use plonky2::util::timing::TimingTree;
use serde_json::{json, Value};

use crate::config::StarkConfig;
use crate::cross_table_lookup::{Column, CtlData, CtlZData, Filter, GrandProductChallenge, GrandProductChallengeSet};
use crate::keccak::keccak_stark::{KeccakStark, NUM_INPUTS};
use crate::poseidon::constants::SPONGE_WIDTH;
use crate::poseidon::poseidon_stark::PoseidonStark;
use crate::proof_blob::stark_proof_to_blob;
use crate::prover::prove_single_table;

const D: usize = 2;
type C = PoseidonGoldilocksConfig;
type F = <C as GenericConfig<D>>::F;
const P: u64 = 0xFFFF_FFFF_0000_0001;

/// oracle/prover.c: splitmix_at(seed, k)
fn splitmix_at(seed: u64, k: u64) -> u64 {
    let mut z = seed.wrapping_add(k.wrapping_mul(0x9E37_79B9_7F4A_7C15));
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    z ^ (z >> 31)
}
/// canonical field element number k of stream `seed` (k counts from 1, as the trace generator does)
fn felt(seed: u64, k: u64) -> F {
    let x = splitmix_at(seed, k);
    F::from_canonical_u64(if x >= P { x - P } else { x })
}
fn words(v: &[F]) -> Vec<u64> {
    v.iter().map(|x| x.to_canonical_u64()).collect()
}
fn digest_words(h: &[HashOut<F>]) -> Vec<u64> {
    h.iter().flat_map(|d| d.elements.iter().map(|x| x.to_canonical_u64())).collect()
}
fn out_dir() -> PathBuf {
    let d = PathBuf::from(std::env::var("ZKM_REF_DUMP_DIR").unwrap_or_else(|_| "ref_dump_out".into()));
    fs::create_dir_all(&d).unwrap();
    d
}
fn write(name: &str, v: &Value) {
    fs::write(out_dir().join(name), serde_json::to_string(v).unwrap()).unwrap();
}
/// column c of a seeded ncols x n matrix: element (row r, column c) = felt(seed, c * n + r + 1)
fn seeded_columns(seed: u64, ncols: usize, n: usize) -> Vec<PolynomialValues<F>> {
    (0..ncols).map(|c| PolynomialValues::new((0..n).map(|r| felt(seed, (c * n + r + 1) as u64)).collect())).collect()
}

#[test]
fn ref_dump_primitives() {
    let shift = F::coset_shift(); // MULTIPLICATIVE_GROUP_GENERATOR (App. A.1)
    // ---- A.3 FFT conventions: natural order in and out
    let mut ntt = vec![];
    for &log_n in &[3usize, 5, 8] {
        let n = 1usize << log_n;
        let seed = 100 + log_n as u64;
        let cols = seeded_columns(seed, 3, n);
        let fwd: Vec<Vec<u64>> = cols.iter().map(|c| words(&PolynomialCoeffs::new(c.values.clone()).fft().values)).collect();
        let inv: Vec<Vec<u64>> = cols.iter().map(|c| words(&c.clone().ifft().coeffs)).collect();
        let cfwd: Vec<Vec<u64>> = cols.iter().map(|c| words(&PolynomialCoeffs::new(c.values.clone()).coset_fft(shift).values)).collect();
        let cinv: Vec<Vec<u64>> = cols.iter().map(|c| words(&c.clone().coset_ifft(shift).coeffs)).collect();
        ntt.push(json!({"log_n": log_n, "ncols": 3, "seed": seed, "coset_shift": shift.to_canonical_u64(),
                        "fft": fwd, "ifft": inv, "coset_fft": cfwd, "coset_ifft": cinv}));
    }
    // ---- A.4 hashing modes
    let mut hashes = vec![];
    for &len in &[0usize, 1, 4, 5, 8, 9, 16, 17, 262] {
        let x: Vec<F> = (0..len).map(|i| felt(300, i as u64 + 1)).collect();
        hashes.push(json!({"len": len, "seed": 300,
                           "hash_no_pad": words(&PoseidonHash::hash_no_pad(&x).elements),
                           "hash_or_noop": words(&PoseidonHash::hash_or_noop(&x).elements)}));
    }
    let l = HashOut { elements: [felt(301, 1), felt(301, 2), felt(301, 3), felt(301, 4)] };
    let r = HashOut { elements: [felt(301, 5), felt(301, 6), felt(301, 7), felt(301, 8)] };
    let two_to_one = words(&<PoseidonHash as Hasher<F>>::two_to_one(l, r).elements);
    // ---- A.5 / A.6 PolynomialBatch: coefficients, leaves (bit-reversed LDE rows), cap, Merkle paths, digests
    let config = StarkConfig::standard_fast_config();
    let (rate_bits, cap_height) = (config.fri_config.rate_bits, config.fri_config.cap_height);
    let mut commits = vec![];
    for &(ncols, log_n) in &[(13usize, 5usize), (262, 5), (4, 6), (3, 4)] {
        let n = 1usize << log_n;
        let seed = 200 + ncols as u64;
        let mut timing = TimingTree::default();
        let b = PolynomialBatch::<F, C, D>::from_values(seeded_columns(seed, ncols, n), rate_bits, false, cap_height, &mut timing, None);
        let coeffs: Vec<Vec<u64>> = b.polynomials.iter().map(|p| words(&p.coeffs)).collect();
        let nn = n << rate_bits;
        let leaves: Vec<Vec<u64>> = [0usize, 1, 3, nn / 2 + 3, nn - 1].iter().map(|&i| words(&b.merkle_tree.leaves[i])).collect();
        let lde_rows: Vec<Vec<u64>> = [0usize, 1, 3, nn / 2 + 3, nn - 1].iter().map(|&i| words(b.get_lde_values(i, 1))).collect();
        let paths: Vec<Vec<u64>> = [0usize, 5, nn - 1].iter().map(|&i| digest_words(&b.merkle_tree.prove(i).siblings)).collect();
        // from_coeffs on the recovered coefficients must give the same cap (prover.rs:576-587 path)
        let b2 = PolynomialBatch::<F, C, D>::from_coeffs(b.polynomials.clone(), rate_bits, false, cap_height, &mut timing, None);
        commits.push(json!({"ncols": ncols, "log_n": log_n, "seed": seed, "rate_bits": rate_bits, "cap_height": cap_height,
                            "coeffs": coeffs, "cap": digest_words(&b.merkle_tree.cap.0), "cap_from_coeffs": digest_words(&b2.merkle_tree.cap.0),
                            "leaf_indices": [0, 1, 3, nn / 2 + 3, nn - 1], "leaves": leaves, "lde_rows_natural_index": lde_rows,
                            "path_indices": [0, 5, nn - 1], "paths": paths}));
    }
    // ---- A.7 Challenger: observe / get interleavings, extension challenge, compact
    let mut ch = Challenger::<F, PoseidonHash>::new();
    let mut script = vec![];
    let mut k = 1u64;
    for &(nobs, nget) in &[(3usize, 2usize), (8, 1), (9, 9), (0, 3), (17, 1), (1, 8)] {
        let obs: Vec<F> = (0..nobs).map(|_| { k += 1; felt(400, k) }).collect();
        ch.observe_elements(&obs);
        let got: Vec<u64> = (0..nget).map(|_| ch.get_challenge().to_canonical_u64()).collect();
        script.push(json!({"observe": words(&obs), "get": got}));
    }
    let ext = ch.get_extension_challenge::<D>();
    use plonky2::field::extension::FieldExtension;
    let ext_w: [F; D] = ext.to_basefield_array();
    ch.observe_element(felt(400, 1000));
    let compact: Vec<u64> = words(ch.compact().as_ref());
    let after_compact = ch.get_challenge().to_canonical_u64();
    write("reference_primitives.json", &json!({
        "schema": 1, "source": "zkMIPS/zkm prover + zkMIPS/plonky2@zkm_dev via tools/ref_dump/ref_dump.rs",
        "ntt": ntt, "hash": hashes, "two_to_one": {"seed": 301, "out": two_to_one}, "commit": commits,
        "challenger": {"seed": 400, "script": script, "extension_challenge": words(&ext_w), "compact_state": compact,
                       "challenge_after_compact": after_compact}}));
}

fn fake_ctl<FF: Field>(degree: usize, num_challenges: usize) -> (CtlData<FF>, GrandProductChallengeSet<FF>) {
    // the benchmark's fake CTL data, poseidon_stark.rs:786-799 / keccak_stark.rs:724-737
    let z = CtlZData {
        helper_columns: vec![PolynomialValues::zero(degree)],
        z: PolynomialValues::zero(degree),
        challenge: GrandProductChallenge { beta: FF::ZERO, gamma: FF::ZERO },
        columns: vec![],
        filter: vec![Some(Filter::new_simple(Column::constant(FF::ZERO)))],
    };
    (CtlData { zs_columns: vec![z.clone(); num_challenges] }, GrandProductChallengeSet { challenges: vec![z.challenge; num_challenges] })
}

#[test]
fn ref_dump_poseidon_proof() {
    // poseidon_benchmark (poseidon_stark.rs:751-816) with seeded inputs: input r, word i = felt(seed, 12 r + i + 1)  (== zkm_poseidon_trace)
    let (seed, num_perms) = (7u64, 125usize);
    let stark = PoseidonStark::<F, D>::default();
    let config = StarkConfig::standard_fast_config();
    let input: Vec<([F; SPONGE_WIDTH], usize)> =
        (0..num_perms).map(|r| (core::array::from_fn(|i| felt(seed, (12 * r + i + 1) as u64)), 0)).collect();
    let mut timing = TimingTree::default();
    let trace = stark.generate_trace(&input, 8);
    let commit = PolynomialBatch::<F, C, D>::from_values(trace.clone(), config.fri_config.rate_bits, false, config.fri_config.cap_height,
                                                         &mut timing, None);
    let degree = 1 << commit.degree_log;
    let (ctl_data, ctl_ch) = fake_ctl::<F>(degree, config.num_challenges);
    let mut ch = Challenger::<F, PoseidonHash>::new();
    let proof = prove_single_table(&stark, &config, &trace, &commit, &ctl_data, &ctl_ch, &mut ch, &mut timing).unwrap();
    let blob = stark_proof_to_blob::<F, C, D>(&proof, &config);
    let trace_col0: Vec<u64> = words(&trace[0].values);
    let trace_col_last: Vec<u64> = words(&trace[trace.len() - 1].values);
    write("reference_poseidon_proof.json", &json!({
        "schema": 1, "table": "PoseidonStark", "seed": seed, "num_perms": num_perms, "log_n": commit.degree_log, "ncols": trace.len(),
        "aux": "zeros, 2 x (1 helper + Z)", "num_helpers": [1, 1],
        "trace_column_0": trace_col0, "trace_column_last": trace_col_last,
        "challenger_after": {"state": words(ch.compact().as_ref())},
        "pow_note": "pow_witness and everything after it may differ between runs of the reference itself (rayon find_any, App. A.9)",
        "blob": blob}));
}

#[test]
fn ref_dump_keccak_proof() {
    // keccak_benchmark (keccak_stark.rs:689-754) with seeded inputs: permutation p, lane i = splitmix_at(seed, 25 p + i + 1) (raw u64)
    let (seed, num_perms) = (11u64, 5usize);
    let stark = KeccakStark::<F, D>::default();
    let config = StarkConfig::standard_fast_config();
    let input: Vec<([u64; NUM_INPUTS], usize)> =
        (0..num_perms).map(|p| (core::array::from_fn(|i| splitmix_at(seed, (25 * p + i + 1) as u64)), 0)).collect();
    let mut timing = TimingTree::default();
    let trace = stark.generate_trace(input, 8);
    let commit = PolynomialBatch::<F, C, D>::from_values(trace.clone(), config.fri_config.rate_bits, false, config.fri_config.cap_height,
                                                         &mut timing, None);
    let degree = 1 << commit.degree_log;
    let (ctl_data, ctl_ch) = fake_ctl::<F>(degree, config.num_challenges);
    let mut ch = Challenger::<F, PoseidonHash>::new();
    let proof = prove_single_table(&stark, &config, &trace, &commit, &ctl_data, &ctl_ch, &mut ch, &mut timing).unwrap();
    let blob = stark_proof_to_blob::<F, C, D>(&proof, &config);
    // the blob is large (2431 columns x 37 queries): keep everything up to the PoW witness, and the first query round
    let header_to_pow = {
        let (w, a, q, z) = (blob[2] as usize, blob[3] as usize, blob[4] as usize, blob[5] as usize);
        let (cap, layers, fin) = (1usize << blob[6], blob[7] as usize, blob[8] as usize);
        16 + 12 + 3 * cap * 4 + 4 * w + 4 * a + z + 2 * q + layers * cap * 4 + 2 * fin + 1
    };
    let round_words = (blob.len() - header_to_pow) / blob[9] as usize;
    write("reference_keccak_proof.json", &json!({
        "schema": 1, "table": "KeccakStark", "seed": seed, "num_perms": num_perms, "log_n": commit.degree_log, "ncols": trace.len(),
        "aux": "zeros, 2 x (1 helper + Z)", "num_helpers": [1, 1], "blob_words": blob.len(),
        "blob_up_to_pow": &blob[..header_to_pow], "first_query_round": &blob[header_to_pow..header_to_pow + round_words]}));
}

/// The whole-segment dump: everything `prove_with_traces` (prover.rs:130-232) adds on top of the single-table proofs above -- the
/// twelve traces of a REAL segment, the twelve-cap + public-values transcript seed (:182-190, get_challenges.rs:91-105), the CTL
/// challenge draw (cross_table_lookup.rs:560-576), `cross_table_lookup_data`'s per-table Z / helper order (:634-703, group_by :807) and
/// the twelve proofs on the shared challenger (:234-438).
///
/// Input: the reference's own test program `emulator/test-vectors/hello` with the arguments of emulator/src/tests.rs:54, split into
/// segments of ZKM_REF_DUMP_SEG_SIZE cycles (default 1024: tables of 2^6 .. 2^12 rows, a ~10 MB trace file), segment 0.
/// Output: `reference_all_proof.json` + `reference_all_proof_traces.bin` (the witness generator's `[Vec<PolynomialValues<F>>; 12]`,
/// generation/mod.rs:25-76, flattened column-major as util.rs:37-46 lays a table out):
///     u64 magic "ZKMTBLS1", u64 ntables, ntables x (u64 ncols, u64 log_n), then per table ncols x 2^log_n canonical u64, little-endian.
/// tests/test_reference_fixtures.py feeds the traces to `zkm_prove_segment` (HIP) and to the oracle's prove_with_traces and compares
/// every dumped quantity; each assertion names the SURVEY App. C item it depends on.
#[test]
fn ref_dump_all_proof() {
    use std::io::{BufReader, Write};

    use plonky2::iop::challenger::Challenger;
    use zkm_emulator::utils::{load_elf_with_patch, split_prog_into_segs};

    use crate::all_stark::{AllStark, Table, NUM_TABLES};
    use crate::cpu::kernel::assembler::segment_kernel;
    use crate::cross_table_lookup::{cross_table_lookup_data, get_grand_product_challenge_set};
    use crate::generation::generate_traces;
    use crate::get_challenges::observe_public_values;
    use crate::prover::prove_with_traces;
    use crate::stark::Stark;

    let seg_size: usize = std::env::var("ZKM_REF_DUMP_SEG_SIZE").ok().and_then(|s| s.parse().ok()).unwrap_or(1024);
    let elf = std::env::var("ZKM_REF_DUMP_ELF").unwrap_or_else(|_| "../emulator/test-vectors/hello".into());
    let seg_dir = out_dir().join("segments");
    let state = load_elf_with_patch(&elf, vec!["aab", "ccd"]);
    let (total_steps, seg_num, _state) = split_prog_into_segs(state, seg_dir.to_str().unwrap(), "", seg_size);
    let kernel = segment_kernel("", "", "", BufReader::new(fs::File::open(seg_dir.join("0")).unwrap()));

    let all_stark = AllStark::<F, D>::default();
    let config = StarkConfig::standard_fast_config();
    let (rate_bits, cap_height) = (config.fri_config.rate_bits, config.fri_config.cap_height);
    let mut timing = TimingTree::default();
    let (traces, public_values, _outputs) = generate_traces::<F, C, D>(&all_stark, &kernel, &config, &mut timing).unwrap();

    // ---- the traces, as the prover receives them
    let mut bin: Vec<u8> = vec![];
    bin.extend_from_slice(b"ZKMTBLS1");
    bin.extend_from_slice(&(NUM_TABLES as u64).to_le_bytes());
    let mut shapes = vec![];
    for (t, table) in traces.iter().zip(Table::all()) {
        let log_n = t[0].len().trailing_zeros() as u64;
        bin.extend_from_slice(&(t.len() as u64).to_le_bytes());
        bin.extend_from_slice(&log_n.to_le_bytes());
        shapes.push(json!({"name": format!("{:?}", table), "enum_index": table as usize, "ncols": t.len(), "log_n": log_n}));
    }
    for t in traces.iter() {
        for col in t.iter() {
            for x in col.values.iter() {
                bin.extend_from_slice(&x.to_canonical_u64().to_le_bytes());
            }
        }
    }
    fs::File::create(out_dir().join("reference_all_proof_traces.bin")).unwrap().write_all(&bin).unwrap();

    // ---- prover.rs:144-190 by hand: commitments, transcript seed, CTL challenges
    let commitments: Vec<PolynomialBatch<F, C, D>> = traces
        .iter()
        .map(|t| PolynomialBatch::<F, C, D>::from_values(t.clone(), rate_bits, false, cap_height, &mut timing, None))
        .collect();
    let trace_caps: Vec<Vec<u64>> = commitments.iter().map(|c| digest_words(&c.merkle_tree.cap.0)).collect();
    let mut challenger = Challenger::<F, PoseidonHash>::new();
    for c in &commitments {
        challenger.observe_cap(&c.merkle_tree.cap);
    }
    observe_public_values::<F, C, D>(&mut challenger, &public_values).unwrap();
    // the words observe_public_values absorbs, in order (get_challenges.rs:14-21, 91-105): 8 + 8 root limbs, then one element per userdata byte
    let mut public_words: Vec<u64> = vec![];
    public_words.extend(public_values.roots_before.root.iter().map(|&x| x as u64));
    public_words.extend(public_values.roots_after.root.iter().map(|&x| x as u64));
    public_words.extend(public_values.userdata.iter().map(|&x| x as u64));
    let ctl_challenges = get_grand_product_challenge_set(&mut challenger, config.num_challenges);
    let challenges_flat: Vec<u64> =
        ctl_challenges.challenges.iter().flat_map(|c| [c.beta.to_canonical_u64(), c.gamma.to_canonical_u64()]).collect();
    let challenger_state_after_ctl_challenges = words(challenger.compact().as_ref());

    // ---- prover.rs:191-200: CtlData per table -- the Z order and helper counts are what the auxiliary commitment is built from
    let ctl_data = cross_table_lookup_data::<F, D>(
        &traces,
        &all_stark.cross_table_lookups,
        &ctl_challenges,
        all_stark.arithmetic_stark.constraint_degree(),
    );
    let ctl_dump: Vec<Value> = ctl_data
        .iter()
        .map(|d| {
            let zs = &d.zs_columns;
            json!({
                "num_zs": zs.len(),
                "num_helpers": zs.iter().map(|z| z.helper_columns.len()).collect::<Vec<_>>(),
                "num_colsets": zs.iter().map(|z| z.columns.len()).collect::<Vec<_>>(),
                "beta": zs.iter().map(|z| z.challenge.beta.to_canonical_u64()).collect::<Vec<_>>(),
                "gamma": zs.iter().map(|z| z.challenge.gamma.to_canonical_u64()).collect::<Vec<_>>(),
                "z_first": zs.iter().map(|z| z.z.values[0].to_canonical_u64()).collect::<Vec<_>>(),
                "z_last": zs.iter().map(|z| z.z.values[z.z.values.len() - 1].to_canonical_u64()).collect::<Vec<_>>(),
                "helper_first": zs.iter().map(|z| z.helper_columns.iter().map(|h| h.values[0].to_canonical_u64()).collect::<Vec<_>>()).collect::<Vec<_>>(),
            })
        })
        .collect();
    drop(ctl_data);

    // ---- the proof itself, through the reference's own entry point (same inputs: same transcript)
    let proof = prove_with_traces::<F, C, D>(&all_stark, &config, traces, public_values, &mut timing).unwrap();
    let proofs: Vec<Value> = proof
        .stark_proofs
        .iter()
        .map(|p| {
            let blob = stark_proof_to_blob::<F, C, D>(p, &config);
            let (w, a, q, z) = (blob[2] as usize, blob[3] as usize, blob[4] as usize, blob[5] as usize);
            let (cap, layers, fin) = (1usize << blob[6], blob[7] as usize, blob[8] as usize);
            let upto = 16 + 12 + 3 * cap * 4 + 4 * w + 4 * a + z + 2 * q + layers * cap * 4 + 2 * fin + 1;
            json!({"blob_words": blob.len(), "blob_up_to_pow": &blob[..upto]})
        })
        .collect();
    let proof_challenges: Vec<u64> =
        proof.ctl_challenges.challenges.iter().flat_map(|c| [c.beta.to_canonical_u64(), c.gamma.to_canonical_u64()]).collect();
    assert_eq!(proof_challenges, challenges_flat, "the hand-built transcript seed must be the one prove_with_traces used");

    write("reference_all_proof.json", &json!({
        "schema": 1, "source": "zkMIPS/zkm prover + zkMIPS/plonky2@zkm_dev via tools/ref_dump/ref_dump.rs::ref_dump_all_proof",
        "program": {"elf": elf, "args": ["aab", "ccd"], "seg_size": seg_size, "segment": 0, "segments": seg_num, "total_steps": total_steps},
        "config": {"rate_bits": rate_bits, "cap_height": cap_height, "num_challenges": config.num_challenges},
        "tables": shapes, "traces_file": "reference_all_proof_traces.bin",
        "public_values_words": public_words,
        "trace_caps": trace_caps,
        "ctl_challenges": challenges_flat,
        "challenger_state_after_ctl_challenges": challenger_state_after_ctl_challenges,
        "ctl_data": ctl_dump,
        "proofs": proofs,
        "pow_note": "each blob stops after its pow_witness; the witness itself may differ between runs of the reference (rayon find_any, App. A.9), and with it the transcript of every LATER table -- compare table k only while the witnesses of tables < k coincide",
    }));
}
