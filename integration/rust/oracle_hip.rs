//! oracle_hip.rs -- `PolynomialBatch` on the GPU: the plug-in point SURVEY 8(b) names, as a file.
//!
//! Goes into the plonky2 fork as `plonky2/src/fri/oracle_hip.rs` (next to `fri/oracle.rs`; `pub mod oracle_hip;` in `fri/mod.rs`).  The
//! zkm-prover crate keeps calling the names it calls today -- every call site below is quoted with the number of arguments it passes,
//! and `tests/test_rust_names.py` holds this file's signatures against those call sites in /root/reference/prover/src:
//!
//!   `PolynomialBatch::<F, C, D>::from_values(trace.clone(), rate_bits, false, cap_height, timing, None)`   prover.rs:154-163, :514-521
//!   `PolynomialBatch::from_coeffs(all_quotient_chunks, rate_bits, false, cap_height, timing, None)`          prover.rs:579-586
//!   `c.merkle_tree.cap.clone()`                                                                             prover.rs:180, :524, :588, :631
//!   `trace_commitment.get_lde_values_packed(i_start, step)`                                                  prover.rs:687, :723-748
//!   `c.polynomials` (coefficients, for StarkOpeningSet::new and check_constraints)                          proof.rs:310-320, prover.rs:824-830
//!   `PolynomialBatch::prove_openings(&instance, &initial_merkle_trees, challenger, &fri_params, timing)`    prover.rs:621-627
//!
//! `HipPolynomialBatch<F, C, D>` has the same constructors and accessors with the same argument lists; what is a field in plonky2
//! (`merkle_tree.cap`, `polynomials`) is a method here (`cap()`, `polynomials()`), because the data lives in HBM and comes to the host
//! only when asked for.  A maintainer either swaps the type at the call sites (`use plonky2::fri::oracle_hip::HipPolynomialBatch as
//! PolynomialBatch` plus `.merkle_tree.cap` -> `.cap()`), or -- the shorter patch -- replaces the bodies of `prove_with_traces` /
//! `prove_single_table` by the one-call entry points of `prove_hip.rs`, which use `hip_batch()` of this type for the trace commitment.
//!
//! NOT COMPILED in the build image (no cargo / rustc there); plonky2 items are named as in zkMIPS/plonky2@zkm_dev (plonky2 0.1.4).
use std::marker::PhantomData;

use crate::field::extension::{Extendable, FieldExtension};
use crate::field::packed::PackedField;
use crate::field::polynomial::{PolynomialCoeffs, PolynomialValues};
use crate::field::types::{Field, PrimeField64};
use crate::fri::proof::{FriInitialTreeProof, FriProof, FriQueryRound, FriQueryStep};
use crate::fri::reduction_strategies::FriReductionStrategy;
use crate::fri::structure::{FriBatchInfo, FriInstanceInfo};
use crate::fri::FriParams;
use crate::hash::hash_types::{HashOut, RichField};
use crate::hash::merkle_proofs::MerkleProof;
use crate::hash::merkle_tree::MerkleCap;
use crate::hash::poseidon::PoseidonHash;
use crate::hip::sys::*;
use crate::iop::challenger::Challenger;
use crate::plonk::config::{GenericConfig, Hasher};
use crate::util::timing::TimingTree;

/// The process-wide context the infallible plonky2 signatures (`from_values` returns `Self`, not `Result`) draw on: created on first
/// use on device `ZKM_HIP_DEVICE` (default 0).  A context is single-owner (include/zkm_hip.h): callers that prove from several
/// threads hold one `HipPolynomialBatch` family per thread through `with_ctx`, or use the pool of `prove_hip.rs`.
pub fn default_ctx() -> *mut zkm_ctx {
    use std::sync::OnceLock;
    struct Shared(*mut zkm_ctx);
    unsafe impl Send for Shared {}
    unsafe impl Sync for Shared {}
    static CTX: OnceLock<Shared> = OnceLock::new();
    CTX.get_or_init(|| {
        let device: i32 = std::env::var("ZKM_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut ctx = std::ptr::null_mut();
        let mut err = std::ptr::null_mut();
        check(unsafe { zkm_ctx_create(device, &mut ctx, &mut err) }, err).expect("libzkmhip: zkm_ctx_create");
        Shared(ctx)
    })
    .0
}

/// `PolynomialBatch<F, C, D>` (plonky2 fri/oracle.rs) with the polynomials, their LDE and the Merkle tree resident in HBM.
pub struct HipPolynomialBatch<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize> {
    batch: *mut zkm_batch,
    ctx: *mut zkm_ctx,
    pub degree_log: usize,
    pub rate_bits: usize,
    pub blinding: bool,
    num_polys: usize,
    cap_height: usize,
    _phantom: PhantomData<C>,
}

impl<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize> Drop for HipPolynomialBatch<F, C, D> {
    fn drop(&mut self) {
        unsafe { zkm_batch_free(self.batch) };
    }
}

impl<F, C, const D: usize> HipPolynomialBatch<F, C, D>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
{
    /// `PolynomialBatch::from_values` (call sites prover.rs:154-163, :514-521): iNTT, coset LDE at rate 2^rate_bits, Poseidon Merkle tree
    /// with a 2^cap_height cap -- one `zkm_batch_commit_columns` call reading the columns where they lie (every `PolynomialValues` is
    /// its own allocation; `GoldilocksField` is `#[repr(transparent)]` over u64).  `blinding` must be false (STARK commitments never
    /// blind: stark.rs:100-101); `timing` and `fft_root_table` are accepted for signature parity and unused (the library keeps its own
    /// twiddle tables and event-based profile).
    pub fn from_values(
        values: Vec<PolynomialValues<F>>,
        rate_bits: usize,
        blinding: bool,
        cap_height: usize,
        timing: &mut TimingTree,
        fft_root_table: Option<&crate::field::fft::FftRootTable<F>>,
    ) -> Self {
        let _ = (timing, fft_root_table);
        Self::commit_with_ctx(default_ctx(), values.iter().map(|p| p.values.as_slice()).collect(), true, rate_bits, blinding, cap_height)
    }

    /// `PolynomialBatch::from_coeffs` (call site prover.rs:579-586, the quotient chunks): coset LDE + tree from coefficients.
    pub fn from_coeffs(
        polynomials: Vec<PolynomialCoeffs<F>>,
        rate_bits: usize,
        blinding: bool,
        cap_height: usize,
        timing: &mut TimingTree,
        fft_root_table: Option<&crate::field::fft::FftRootTable<F>>,
    ) -> Self {
        let _ = (timing, fft_root_table);
        Self::commit_with_ctx(default_ctx(), polynomials.iter().map(|p| p.coeffs.as_slice()).collect(), false, rate_bits, blinding, cap_height)
    }

    /// The same on an explicit context (one per proving thread).
    pub fn commit_with_ctx(ctx: *mut zkm_ctx, columns: Vec<&[F]>, columns_are_values: bool, rate_bits: usize, blinding: bool, cap_height: usize) -> Self {
        assert!(!blinding, "libzkmhip commits without blinding (the STARK oracles of stark.rs:91-148 never blind)");
        const { assert!(core::mem::size_of::<F>() == 8) };
        let n = columns.first().map_or(0, |c| c.len());
        assert!(n.is_power_of_two() && columns.iter().all(|c| c.len() == n), "columns of one power-of-two length");
        let ptrs: Vec<*const u64> = columns.iter().map(|c| c.as_ptr() as *const u64).collect();
        let mut batch = std::ptr::null_mut();
        let mut err = std::ptr::null_mut();
        check(
            unsafe {
                zkm_batch_commit_columns(ctx, ptrs.as_ptr(), ptrs.len(), n.trailing_zeros(), columns_are_values as i32, rate_bits as u32,
                                         cap_height as u32, &mut batch, &mut err)
            },
            err,
        )
        .expect("libzkmhip: zkm_batch_commit_columns");
        Self { batch, ctx, degree_log: n.trailing_zeros() as usize, rate_bits, blinding, num_polys: ptrs.len(), cap_height, _phantom: PhantomData }
    }

    /// The commitment as the library's handle: `trace_batch` of zkm_prove_single_table, an oracle of zkm_fri_prove.
    pub fn hip_batch(&self) -> *const zkm_batch {
        self.batch
    }

    pub fn num_polys(&self) -> usize {
        self.num_polys
    }

    /// `.merkle_tree.cap` (prover.rs:180, :524, :588, :631): 2^cap_height digests.
    pub fn cap(&self) -> MerkleCap<F, C::Hasher> {
        let mut w = vec![0u64; 4 << self.cap_height];
        assert_eq!(unsafe { zkm_batch_cap(self.batch, w.as_mut_ptr()) }, 0, "zkm_batch_cap");
        MerkleCap(w.chunks_exact(4).map(|d| HashOut { elements: core::array::from_fn(|i| F::from_canonical_u64(d[i])) }).collect())
    }

    /// `.polynomials` (proof.rs:310-320, prover.rs:824-830): the coefficient vectors, downloaded.
    pub fn polynomials(&self) -> Vec<PolynomialCoeffs<F>> {
        let n = 1usize << self.degree_log;
        let mut w = vec![0u64; self.num_polys * n];
        assert_eq!(unsafe { zkm_batch_coeffs(self.batch, w.as_mut_ptr()) }, 0, "zkm_batch_coeffs");
        w.chunks_exact(n).map(|c| PolynomialCoeffs::new(c.iter().map(|&x| F::from_canonical_u64(x)).collect())).collect()
    }

    /// `get_lde_values(index, step)` of plonky2: the values of every polynomial at LDE point `index * step` (natural order).
    pub fn get_lde_values(&self, index: usize, step: usize) -> Vec<F> {
        let mut w = vec![0u64; self.num_polys];
        assert_eq!(unsafe { zkm_batch_lde_rows(self.batch, index, step, 1, w.as_mut_ptr()) }, 0, "zkm_batch_lde_rows");
        w.into_iter().map(F::from_canonical_u64).collect()
    }

    /// `get_lde_values_packed(index_start, step)` (prover.rs:687, :723-748): P::WIDTH consecutive rows (index_start + k) * step,
    /// k < P::WIDTH, packed lane by lane -- one `zkm_batch_lde_rows` call of P::WIDTH rows.
    pub fn get_lde_values_packed<P>(&self, index_start: usize, step: usize) -> Vec<P>
    where
        P: PackedField<Scalar = F>,
    {
        let mut w = vec![0u64; P::WIDTH * self.num_polys];
        assert_eq!(unsafe { zkm_batch_lde_rows(self.batch, index_start, step, P::WIDTH, w.as_mut_ptr()) }, 0, "zkm_batch_lde_rows");
        // rows are row-major: row k at w[k * num_polys ..]; leaf_vals[j] = column j of every row
        (0..self.num_polys)
            .map(|j| {
                let lanes: Vec<F> = (0..P::WIDTH).map(|k| F::from_canonical_u64(w[k * self.num_polys + j])).collect();
                *P::from_slice(&lanes)
            })
            .collect()
    }

    /// `PolynomialBatch::prove_openings` (call site prover.rs:621-627; plonky2 fri/oracle.rs): alpha from the transcript, one
    /// alpha-reduced quotient per batch chained with `shift_poly`, commit phase, final polynomial, proof of work, query rounds --
    /// one `zkm_fri_prove` call on the oracles' handles; the transcript goes over and comes back (`challenger_hip.rs`).
    pub fn prove_openings(
        instance: &FriInstanceInfo<F, D>,
        oracles: &[&Self],
        challenger: &mut Challenger<F, C::Hasher>,
        fri_params: &FriParams,
        timing: &mut TimingTree,
    ) -> FriProof<F, C::Hasher, D> {
        let _ = timing;
        assert!(D == 2, "libzkmhip implements the quadratic extension of Goldilocks");
        assert_eq!(instance.oracles.len(), oracles.len());
        let cfg = fri_config_words(fri_params);
        let polys: Vec<Vec<zkm_fri_poly>> = instance
            .batches
            .iter()
            .map(|b: &FriBatchInfo<F, D>| {
                b.polynomials.iter().map(|p| zkm_fri_poly { oracle: p.oracle_index as u32, poly: p.polynomial_index as u32 }).collect()
            })
            .collect();
        let batches: Vec<zkm_fri_batch> = instance
            .batches
            .iter()
            .zip(&polys)
            .map(|(b, ps)| {
                let c: [F; D] = b.point.to_basefield_array();
                zkm_fri_batch { point: [c[0].to_canonical_u64(), c[1].to_canonical_u64()], polys: ps.as_ptr(), npolys: ps.len() }
            })
            .collect();
        let handles: Vec<*const zkm_batch> = oracles.iter().map(|o| o.hip_batch()).collect();
        let cols: Vec<usize> = oracles.iter().map(|o| o.num_polys).collect();
        let log_n = oracles[0].degree_log as u32;
        let words = unsafe { zkm_fri_proof_words(&cfg, log_n, cols.as_ptr(), cols.len()) };
        assert!(words != 0, "libzkmhip: unsupported FriParams");
        let mut blob = vec![0u64; words];
        let mut ch = challenger.to_zkm();
        let mut err = std::ptr::null_mut();
        check(
            unsafe {
                zkm_fri_prove(oracles[0].ctx, &cfg, handles.as_ptr(), handles.len(), batches.as_ptr(), batches.len(), &mut ch, blob.as_mut_ptr(),
                              &mut err)
            },
            err,
        )
        .expect("libzkmhip: zkm_fri_prove");
        challenger.set_from_zkm(&ch);
        fri_proof_from_blob::<F, C, D>(&blob)
    }
}

/// `FriParams` -> the config words zkm_fri_prove reads (rate, cap height, proof-of-work bits, queries, constant arity, final
/// polynomial length); `num_challenges` is not used by the FRI entry point.
pub fn fri_config_words(p: &FriParams) -> zkm_stark_config {
    let (arity_bits, final_poly_bits) = match p.config.reduction_strategy {
        FriReductionStrategy::ConstantArityBits(a, f) => (a as u32, f as u32),
        _ => panic!("libzkmhip supports FriReductionStrategy::ConstantArityBits (prover/src/config.rs:25)"),
    };
    zkm_stark_config {
        rate_bits: p.config.rate_bits as u32,
        cap_height: p.config.cap_height as u32,
        pow_bits: p.config.proof_of_work_bits,
        num_challenges: 2,
        num_queries: p.config.num_query_rounds as u32,
        arity_bits,
        final_poly_bits,
    }
}

/// The FRI proof blob of zkm_fri_prove (include/zkm_hip.h "FRI proof blob") -> `FriProof`.
pub fn fri_proof_from_blob<F, C, const D: usize>(blob: &[u64]) -> FriProof<F, C::Hasher, D>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    assert_eq!(blob[0], 0x4650_4952_464d_4b5a, "not a zkm FRI proof blob");
    let (degree_bits, noracles, cap_height, layers) = (blob[1] as usize, blob[2] as usize, blob[3] as usize, blob[4] as usize);
    let (final_len, num_queries, rate_bits, arity_bits) = (blob[5] as usize, blob[6] as usize, blob[7] as usize, blob[8] as usize);
    let oracle_cols: Vec<usize> = (0..noracles).map(|o| blob[16 + o] as usize).collect();
    let lde_bits = degree_bits + rate_bits;
    let mut at = 24usize;
    let mut words = |n: usize| {
        let s = &blob[at..at + n];
        at += n;
        s
    };
    let base = |w: &[u64]| -> Vec<F> { w.iter().map(|&x| F::from_canonical_u64(x)).collect() };
    let ext = |w: &[u64]| -> Vec<F::Extension> {
        w.chunks_exact(D).map(|c| <F::Extension as FieldExtension<D>>::from_basefield_array(core::array::from_fn(|i| F::from_canonical_u64(c[i])))).collect()
    };
    let digests = |w: &[u64]| -> Vec<HashOut<F>> { w.chunks_exact(4).map(|d| HashOut { elements: core::array::from_fn(|i| F::from_canonical_u64(d[i])) }).collect() };
    let commit_phase_merkle_caps: Vec<MerkleCap<F, C::Hasher>> = (0..layers).map(|_| MerkleCap(digests(words(4 << cap_height)))).collect();
    let final_poly = PolynomialCoeffs::new(ext(words(D * final_len)));
    let pow_witness = F::from_canonical_u64(words(1)[0]);
    let query_round_proofs = (0..num_queries)
        .map(|_| {
            let evals_proofs = oracle_cols
                .iter()
                .map(|&n| {
                    let evals = base(words(n));
                    let siblings = digests(words(4 * (lde_bits - cap_height)));
                    (evals, MerkleProof { siblings })
                })
                .collect();
            let steps = (0..layers)
                .map(|i| {
                    let evals = ext(words(D << arity_bits));
                    let siblings = digests(words(4 * (lde_bits - arity_bits * (i + 1) - cap_height)));
                    FriQueryStep { evals, merkle_proof: MerkleProof { siblings } }
                })
                .collect();
            FriQueryRound { initial_trees_proof: FriInitialTreeProof { evals_proofs }, steps }
        })
        .collect();
    assert_eq!(at, blob.len(), "FRI proof blob has trailing words");
    FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness }
}
