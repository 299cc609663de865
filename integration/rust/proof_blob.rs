//! proof_blob.rs -- `StarkProofWithMetadata` <-> the proof blob of `include/zkm_hip.h`, both directions.
//!
//! Lives in the zkm-prover crate (`prover/src/proof_blob.rs`, `pub mod proof_blob;` in `prover/src/lib.rs`): it needs the
//! `pub(crate)` field `StarkProofWithMetadata::init_challenger_state` (prover/src/proof.rs:191-201).
//!
//! * `stark_proof_from_blob`  rebuilds what `prove_single_table` returns (prover/src/prover.rs:633-640) from the words
//!   `zkm_prove_single_table[_ctl]` / `zkm_prove_with_traces` wrote: the consumers are the native verifier
//!   (prover/src/verifier.rs:60-79) and the recursion circuits' witness setters
//!   (prover/src/recursive_verifier.rs:536-560 `set_stark_proof_target`, which read every field rebuilt here).
//! * `stark_proof_to_blob`    is the inverse; `tools/ref_dump/ref_dump.rs` uses it to write reference-made proofs in the
//!   blob layout so that parity with the HIP path is a plain word-for-word comparison.
//!
//! Layout (include/zkm_hip.h, "Proof blob"): 16 header words, init_challenger_state[12], three caps, the StarkOpeningSet
//! fields in struct order (proof.rs:281-297), FRI caps, final poly, PoW witness, then per query round the three
//! FriInitialTreeProof (evals, siblings) pairs and one FriQueryStep (evals, siblings) per layer.  F2 = [c0, c1].
//!
//! NOT COMPILED in the build image (no cargo / rustc there); plonky2 items are named as in
//! zkMIPS/plonky2@zkm_dev (plonky2 0.1.4): fri/proof.rs, hash/merkle_tree.rs, hash/merkle_proofs.rs, hash/hash_types.rs.

use plonky2::field::extension::{Extendable, FieldExtension};
use plonky2::field::polynomial::PolynomialCoeffs;
use plonky2::field::types::{Field, PrimeField64};
use plonky2::fri::proof::{FriInitialTreeProof, FriProof, FriQueryRound, FriQueryStep};
use plonky2::hash::hash_types::{HashOut, RichField};
use plonky2::hash::merkle_proofs::MerkleProof;
use plonky2::hash::merkle_tree::MerkleCap;
use plonky2::plonk::config::{GenericConfig, Hasher};
use plonky2::hash::hashing::PlonkyPermutation;

use crate::config::StarkConfig;
use crate::proof::{StarkOpeningSet, StarkProof, StarkProofWithMetadata};

pub const ZKM_PROOF_MAGIC: u64 = 0x5a4b_4d50_524f_4f46;
const HEADER_WORDS: usize = 16;

/// Cursor over the blob; every getter advances by what it read.
struct Reader<'a> {
    w: &'a [u64],
    at: usize,
}

impl<'a> Reader<'a> {
    fn words(&mut self, n: usize) -> &'a [u64] {
        let s = &self.w[self.at..self.at + n];
        self.at += n;
        s
    }
    fn base<F: Field>(&mut self, n: usize) -> Vec<F> {
        self.words(n).iter().map(|&x| F::from_canonical_u64(x)).collect()
    }
    fn ext<F: RichField + Extendable<D>, const D: usize>(&mut self, n: usize) -> Vec<F::Extension> {
        (0..n)
            .map(|_| {
                let c: Vec<F> = self.base::<F>(D);
                <F::Extension as FieldExtension<D>>::from_basefield_array(c.try_into().unwrap())
            })
            .collect()
    }
    fn digests<F: RichField, H: Hasher<F, Hash = HashOut<F>>>(&mut self, n: usize) -> Vec<H::Hash> {
        (0..n)
            .map(|_| HashOut { elements: self.base::<F>(4).try_into().unwrap() })
            .collect()
    }
}

/// The inverse of `stark_proof_to_blob`.  Panics on a malformed blob (the blob comes from libzkmhip.so in the same process).
pub fn stark_proof_from_blob<F, C, const D: usize>(blob: &[u64]) -> StarkProofWithMetadata<F, C, D>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    assert_eq!(blob[0], ZKM_PROOF_MAGIC, "not a zkm proof blob");
    let (degree_bits, w, a, q, z) = (blob[1] as usize, blob[2] as usize, blob[3] as usize, blob[4] as usize, blob[5] as usize);
    let (cap_height, layers, final_len, num_queries) = (blob[6] as usize, blob[7] as usize, blob[8] as usize, blob[9] as usize);
    let (rate_bits, arity_bits) = (blob[10] as usize, blob[11] as usize);
    let lde_bits = degree_bits + rate_bits;
    let cap_len = 1usize << cap_height;
    let mut r = Reader { w: blob, at: HEADER_WORDS };

    // init_challenger_state: <C::Hasher as Hasher<F>>::Permutation (PoseidonPermutation: 12 elements)
    let init_state: Vec<F> = r.base::<F>(12);
    let init_challenger_state = <C::Hasher as Hasher<F>>::Permutation::new(init_state.into_iter());

    let trace_cap = MerkleCap(r.digests::<F, C::Hasher>(cap_len));
    let auxiliary_polys_cap = MerkleCap(r.digests::<F, C::Hasher>(cap_len));
    let quotient_polys_cap = MerkleCap(r.digests::<F, C::Hasher>(cap_len));

    let openings = StarkOpeningSet {
        local_values: r.ext::<F, D>(w),
        next_values: r.ext::<F, D>(w),
        auxiliary_polys: r.ext::<F, D>(a),
        auxiliary_polys_next: r.ext::<F, D>(a),
        ctl_zs_first: r.base::<F>(z),
        quotient_polys: r.ext::<F, D>(q),
    };

    let commit_phase_merkle_caps: Vec<MerkleCap<F, C::Hasher>> =
        (0..layers).map(|_| MerkleCap(r.digests::<F, C::Hasher>(cap_len))).collect();
    let final_poly = PolynomialCoeffs::new(r.ext::<F, D>(final_len));
    let pow_witness = F::from_canonical_u64(r.words(1)[0]);

    let oracle_cols = [w, a, q];
    let query_round_proofs = (0..num_queries)
        .map(|_| {
            // FriInitialTreeProof: one (leaf, MerkleProof) per oracle, oracle order = stark.rs:17-19 (trace, auxiliary, quotient)
            let evals_proofs = oracle_cols
                .iter()
                .map(|&n| {
                    let evals = r.base::<F>(n);
                    let siblings = r.digests::<F, C::Hasher>(lde_bits - cap_height);
                    (evals, MerkleProof { siblings })
                })
                .collect();
            let steps = (0..layers)
                .map(|i| {
                    let evals = r.ext::<F, D>(1 << arity_bits);
                    let siblings = r.digests::<F, C::Hasher>(lde_bits - arity_bits * (i + 1) - cap_height);
                    FriQueryStep { evals, merkle_proof: MerkleProof { siblings } }
                })
                .collect();
            FriQueryRound { initial_trees_proof: FriInitialTreeProof { evals_proofs }, steps }
        })
        .collect();
    assert_eq!(r.at, blob.len(), "proof blob has trailing words");

    StarkProofWithMetadata {
        init_challenger_state,
        proof: StarkProof {
            trace_cap,
            auxiliary_polys_cap,
            quotient_polys_cap,
            openings,
            opening_proof: FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness },
        },
    }
}

fn push_base<F: PrimeField64>(out: &mut Vec<u64>, v: &[F]) {
    out.extend(v.iter().map(|x| x.to_canonical_u64()));
}
fn push_ext<F: RichField + Extendable<D>, const D: usize>(out: &mut Vec<u64>, v: &[F::Extension]) {
    for e in v {
        let c: [F; D] = e.to_basefield_array();
        push_base(out, &c);
    }
}
fn push_digests<F: RichField>(out: &mut Vec<u64>, v: &[HashOut<F>]) {
    for d in v {
        push_base(out, &d.elements);
    }
}

/// A reference-made proof in the blob layout (used by the fixture dumper; also handy for differential debugging).
pub fn stark_proof_to_blob<F, C, const D: usize>(p: &StarkProofWithMetadata<F, C, D>, config: &StarkConfig) -> Vec<u64>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    let proof = &p.proof;
    let fri = &config.fri_config;
    let degree_bits = proof.recover_degree_bits(config);
    let arity_bits = if proof.opening_proof.query_round_proofs[0].steps.is_empty() {
        4 // ConstantArityBits(4, 5) of standard_fast_config (config.rs:25); unused when there is no layer
    } else {
        proof.opening_proof.query_round_proofs[0].steps[0].evals.len().trailing_zeros() as usize
    };
    let mut out = vec![
        ZKM_PROOF_MAGIC,
        degree_bits as u64,
        proof.openings.local_values.len() as u64,
        proof.openings.auxiliary_polys.len() as u64,
        proof.openings.quotient_polys.len() as u64,
        proof.openings.ctl_zs_first.len() as u64,
        fri.cap_height as u64,
        proof.opening_proof.commit_phase_merkle_caps.len() as u64,
        proof.opening_proof.final_poly.coeffs.len() as u64,
        proof.opening_proof.query_round_proofs.len() as u64,
        fri.rate_bits as u64,
        arity_bits as u64,
        0,
        0,
        0,
        0,
    ];
    push_base(&mut out, p.init_challenger_state.as_ref());
    push_digests(&mut out, &proof.trace_cap.0);
    push_digests(&mut out, &proof.auxiliary_polys_cap.0);
    push_digests(&mut out, &proof.quotient_polys_cap.0);
    push_ext::<F, D>(&mut out, &proof.openings.local_values);
    push_ext::<F, D>(&mut out, &proof.openings.next_values);
    push_ext::<F, D>(&mut out, &proof.openings.auxiliary_polys);
    push_ext::<F, D>(&mut out, &proof.openings.auxiliary_polys_next);
    push_base(&mut out, &proof.openings.ctl_zs_first);
    push_ext::<F, D>(&mut out, &proof.openings.quotient_polys);
    for cap in &proof.opening_proof.commit_phase_merkle_caps {
        push_digests(&mut out, &cap.0);
    }
    push_ext::<F, D>(&mut out, &proof.opening_proof.final_poly.coeffs);
    out.push(proof.opening_proof.pow_witness.to_canonical_u64());
    for round in &proof.opening_proof.query_round_proofs {
        for (evals, path) in &round.initial_trees_proof.evals_proofs {
            push_base(&mut out, evals);
            push_digests(&mut out, &path.siblings);
        }
        for step in &round.steps {
            push_ext::<F, D>(&mut out, &step.evals);
            push_digests(&mut out, &step.merkle_proof.siblings);
        }
    }
    out
}
