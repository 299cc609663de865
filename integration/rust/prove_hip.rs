//! prove_hip.rs -- the two zkm-prover functions of the hot path routed through libzkmhip.so.
//!
//! Goes into the zkm-prover crate as `prover/src/prove_hip.rs`; `prove_with_traces` (prover/src/prover.rs:130-232) and
//! `prove_single_table` (:441-641) call these when the `hip` feature is on.  Everything else of the crate is unchanged:
//! witness generation (`generate_traces`, generation/mod.rs:25-76) feeds it, `verify_proof` (verifier.rs:27-176) and the
//! recursion layer (`fixed_recursive_verifier.rs:769-777`) consume what it returns.
//! NOT COMPILED in the build image (no cargo / rustc there).
use anyhow::Result;
use plonky2::field::extension::Extendable;
use plonky2::field::polynomial::PolynomialValues;
use plonky2::field::types::{Field, PrimeField64};
use plonky2::hash::hash_types::{HashOut, RichField};
use plonky2::hip::sys::*;
use plonky2::iop::challenger::Challenger;
use plonky2::plonk::config::{GenericConfig, Hasher};

use crate::all_stark::{Table, NUM_TABLES};
use crate::config::StarkConfig;
use crate::cross_table_lookup::{GrandProductChallenge, GrandProductChallengeSet};
use crate::proof::{AllProof, PublicValues, StarkProofWithMetadata};
use crate::proof_blob::stark_proof_from_blob;

/// `Table` (all_stark.rs:96-110) -> ZKM_TABLE_* id; inverse of zkm_table_enum_index
pub fn zkm_table_id(t: Table) -> i32 {
    match t {
        Table::Arithmetic => ZKM_TABLE_ARITHMETIC,
        Table::Cpu => ZKM_TABLE_CPU,
        Table::Poseidon => ZKM_TABLE_POSEIDON,
        Table::PoseidonSponge => ZKM_TABLE_POSEIDON_SPONGE,
        Table::Keccak => ZKM_TABLE_KECCAK,
        Table::KeccakSponge => ZKM_TABLE_KECCAK_SPONGE,
        Table::ShaExtend => ZKM_TABLE_SHA_EXTEND,
        Table::ShaExtendSponge => ZKM_TABLE_SHA_EXTEND_SPONGE,
        Table::ShaCompress => ZKM_TABLE_SHA_COMPRESS,
        Table::ShaCompressSponge => ZKM_TABLE_SHA_COMPRESS_SPONGE,
        Table::Logic => ZKM_TABLE_LOGIC,
        Table::Memory => ZKM_TABLE_MEMORY,
    }
}

pub fn zkm_config(c: &StarkConfig) -> zkm_stark_config {
    use plonky2::fri::reduction_strategies::FriReductionStrategy::ConstantArityBits;
    let (arity_bits, final_poly_bits) = match c.fri_config.reduction_strategy {
        ConstantArityBits(a, f) => (a as u32, f as u32),
        _ => panic!("libzkmhip supports FriReductionStrategy::ConstantArityBits (config.rs:25)"),
    };
    zkm_stark_config {
        rate_bits: c.fri_config.rate_bits as u32,
        cap_height: c.fri_config.cap_height as u32,
        pow_bits: c.fri_config.proof_of_work_bits,
        num_challenges: c.num_challenges as u32,
        num_queries: c.fri_config.num_query_rounds as u32,
        arity_bits,
        final_poly_bits,
    }
}

/// Vec<PolynomialValues<F>> -> column-major canonical words (the layout prover/src/util.rs:37-46 produces).
/// Traces are BORROWED by the library: the `trace_poly_values.clone()` of prover.rs:155-157 disappears.
pub fn flatten<F: PrimeField64>(cols: &[PolynomialValues<F>]) -> Vec<u64> {
    let n = cols.first().map_or(0, |c| c.len());
    let mut out = Vec::with_capacity(cols.len() * n);
    for c in cols {
        debug_assert_eq!(c.len(), n);
        out.extend(c.values.iter().map(|x| x.to_canonical_u64()));
    }
    out
}

/// Body of `prove_with_traces` (prover.rs:130-232 + prove_with_commitments :234-438): one library call for the whole segment.
/// The AllStark cross-table-lookup description (all_stark.rs:136-542) ships inside the library (zkm_prove_segment), so
/// nothing but the twelve traces, the public values and the config crosses the boundary.
pub fn prove_with_traces_hip<F, C, const D: usize>(
    ctx: *mut zkm_ctx,
    config: &StarkConfig,
    trace_poly_values: &[Vec<PolynomialValues<F>>; NUM_TABLES],
    public_values: PublicValues,
    public_value_words: &[u64], // what observe_public_values (get_challenges.rs:13-60) feeds the transcript, in order
) -> Result<AllProof<F, C, D>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    let cfg = zkm_config(config);
    let flat: Vec<Vec<u64>> = trace_poly_values.iter().map(|t| flatten(t)).collect();
    let ptrs: Vec<*const u64> = flat.iter().map(|v| v.as_ptr()).collect();
    let log_n: Vec<u32> = trace_poly_values.iter().map(|t| t[0].len().trailing_zeros()).collect();
    let mut offs = vec![0usize; NUM_TABLES + 1];
    let mut err = std::ptr::null_mut();
    // sizing pass (proofs_out = NULL), then the proving pass
    check(unsafe { zkm_prove_segment(std::ptr::null_mut(), &cfg, ptrs.as_ptr(), log_n.as_ptr(), public_value_words.as_ptr(),
                                     public_value_words.len(), std::ptr::null_mut(), offs.as_mut_ptr(), std::ptr::null_mut(), &mut err) }, err)?;
    let mut blob = vec![0u64; offs[NUM_TABLES]];
    let mut chal = vec![0u64; 2 * config.num_challenges];
    check(unsafe { zkm_prove_segment(ctx, &cfg, ptrs.as_ptr(), log_n.as_ptr(), public_value_words.as_ptr(), public_value_words.len(),
                                     blob.as_mut_ptr(), offs.as_mut_ptr(), chal.as_mut_ptr(), &mut err) }, err)?;
    let stark_proofs: [StarkProofWithMetadata<F, C, D>; NUM_TABLES] =
        core::array::from_fn(|t| stark_proof_from_blob::<F, C, D>(&blob[offs[t]..offs[t + 1]]));
    let ctl_challenges = GrandProductChallengeSet {
        challenges: (0..config.num_challenges)
            .map(|k| GrandProductChallenge { beta: F::from_canonical_u64(chal[2 * k]), gamma: F::from_canonical_u64(chal[2 * k + 1]) })
            .collect(),
    };
    Ok(AllProof { stark_proofs, ctl_challenges, public_values })
}

/// Body of `prove_single_table` (prover.rs:441-641) for the benchmark shape the reference's own tests use
/// (poseidon_stark.rs:751-816, keccak_stark.rs:689-754): existing trace values, CtlData given as auxiliary columns.
pub fn prove_single_table_hip<F, C, const D: usize>(
    ctx: *mut zkm_ctx,
    table: Table,
    config: &StarkConfig,
    trace_poly_values: &[PolynomialValues<F>],
    aux_columns: &[PolynomialValues<F>], // ctl helper columns ++ ctl z columns (prover.rs:497-508)
    num_ctl_helper_polys: &[usize],      // CtlData::num_ctl_helper_polys (cross_table_lookup.rs:474-481)
    challenger: &mut Challenger<F, C::Hasher>,
) -> Result<StarkProofWithMetadata<F, C, D>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F, Hasher = plonky2::hash::poseidon::PoseidonHash>,
{
    let cfg = zkm_config(config);
    let (trace, aux) = (flatten(trace_poly_values), flatten(aux_columns));
    let log_n = trace_poly_values[0].len().trailing_zeros();
    let helpers: Vec<u32> = num_ctl_helper_polys.iter().map(|&x| x as u32).collect();
    let words = unsafe { zkm_proof_words(&cfg, log_n, trace_poly_values.len(), aux_columns.len(), helpers.len()) };
    anyhow::ensure!(words != 0, "libzkmhip: unsupported StarkConfig");
    let mut blob = vec![0u64; words];
    let mut ch = challenger.to_zkm();
    let mut err = std::ptr::null_mut();
    check(unsafe { zkm_prove_single_table(ctx, zkm_table_id(table), &cfg, trace.as_ptr(), trace_poly_values.len(), log_n, std::ptr::null(),
                                          aux.as_ptr(), aux_columns.len(), helpers.as_ptr(), helpers.len(), &mut ch, blob.as_mut_ptr(),
                                          &mut err) }, err)?;
    challenger.set_from_zkm(&ch);
    Ok(stark_proof_from_blob::<F, C, D>(&blob))
}
