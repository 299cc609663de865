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

/// One pointer per column of a `Vec<PolynomialValues<F>>` -- each column is its own heap allocation (prover.rs:154-163 hands the
/// vector to `PolynomialBatch::from_values`), and `GoldilocksField` is `#[repr(transparent)]` over a u64, so the library reads the
/// columns where they lie: no 2 GiB `flatten` copy on the host, and the `trace_poly_values.clone()` of prover.rs:155-157 disappears
/// (inputs are BORROWED for the call).  A `GoldilocksField` may hold any u64 representing its residue (products are reduced below
/// 2^64, not below p -- e.g. the s-box values of poseidon_stark.rs:133-160); the library's transforms accept any representative and
/// it canonicalises the device copy it keeps for the CTL / lookup columns, so no `to_canonical_u64()` pass is needed here.
pub fn column_ptrs<F: PrimeField64>(cols: &[PolynomialValues<F>]) -> Vec<*const u64> {
    const { assert!(core::mem::size_of::<F>() == 8) };
    let n = cols.first().map_or(0, |c| c.len());
    cols.iter()
        .map(|c| {
            debug_assert_eq!(c.len(), n);
            c.values.as_ptr() as *const u64
        })
        .collect()
}

/// What `observe_public_values` (get_challenges.rs:91-105) feeds the transcript, in its order: the eight u32 limbs of
/// `roots_before.root`, the eight of `roots_after.root` (observe_root :13-20, `from_canonical_u32`), then every byte of `userdata`
/// as one field element (`from_canonical_u8`, :101-103).  This is the `public_values` argument of zkm_prove_segment /
/// zkm_prove_with_traces (observed right after the trace caps, prover.rs:182-187).
pub fn public_values_words(pv: &PublicValues) -> Vec<u64> {
    let mut w = Vec::with_capacity(16 + pv.userdata.len());
    w.extend(pv.roots_before.root.iter().map(|&limb| limb as u64));
    w.extend(pv.roots_after.root.iter().map(|&limb| limb as u64));
    w.extend(pv.userdata.iter().map(|&b| b as u64));
    w
}

/// Body of `prove_with_traces` (prover.rs:130-232 + prove_with_commitments :234-438): one library call for the whole segment.
/// The AllStark cross-table-lookup description (all_stark.rs:136-542) ships inside the library (zkm_prove_segment_columns), so
/// nothing but the column pointers of the twelve traces, the public values and the config crosses the boundary.
pub fn prove_with_traces_hip<F, C, const D: usize>(
    ctx: *mut zkm_ctx,
    config: &StarkConfig,
    trace_poly_values: &[Vec<PolynomialValues<F>>; NUM_TABLES],
    public_values: PublicValues,
) -> Result<AllProof<F, C, D>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    let cfg = zkm_config(config);
    let public_value_words = public_values_words(&public_values);
    let cols: Vec<Vec<*const u64>> = trace_poly_values.iter().map(|t| column_ptrs(t)).collect();
    let ptrs: Vec<*const *const u64> = cols.iter().map(|v| v.as_ptr()).collect();
    let log_n: Vec<u32> = trace_poly_values.iter().map(|t| t[0].len().trailing_zeros()).collect();
    let mut offs = vec![0usize; NUM_TABLES + 1];
    let mut err = std::ptr::null_mut();
    // sizing pass (proofs_out = NULL), then the proving pass
    check(unsafe { zkm_prove_segment_columns(std::ptr::null_mut(), &cfg, ptrs.as_ptr(), log_n.as_ptr(), public_value_words.as_ptr(),
                                     public_value_words.len(), std::ptr::null_mut(), offs.as_mut_ptr(), std::ptr::null_mut(), &mut err) }, err)?;
    let mut blob = vec![0u64; offs[NUM_TABLES]];
    let mut chal = vec![0u64; 2 * config.num_challenges];
    check(unsafe { zkm_prove_segment_columns(ctx, &cfg, ptrs.as_ptr(), log_n.as_ptr(), public_value_words.as_ptr(), public_value_words.len(),
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

/// K independent segments in LOCK-STEP (zkm_prove_segments_columns, include/zkm_hip.h): the reference's driver proves the segments of
/// a program one after the other (prover/examples/utils/src/utils.rs:57-68, 105-133), each a `prove_with_traces` on its own
/// transcript; with the traces of K segments at hand (K calls of `generate_traces`, generation/mod.rs:25-76) this proves them in one
/// library call -- one launch per stage for all segments whose table has the same height -- and returns K `AllProof`s, each word
/// for word what `prove_with_traces_hip` returns for that segment alone.
pub fn prove_segments_hip<F, C, const D: usize>(
    ctx: *mut zkm_ctx,
    config: &StarkConfig,
    segments: &[([Vec<PolynomialValues<F>>; NUM_TABLES], PublicValues)],
) -> Result<Vec<AllProof<F, C, D>>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    let cfg = zkm_config(config);
    let k = segments.len();
    let pubs: Vec<Vec<u64>> = segments.iter().map(|(_, pv)| public_values_words(pv)).collect();
    let cols: Vec<Vec<Vec<*const u64>>> = segments.iter().map(|(tr, _)| tr.iter().map(|t| column_ptrs(t)).collect()).collect();
    let tabs: Vec<Vec<*const *const u64>> = cols.iter().map(|seg| seg.iter().map(|v| v.as_ptr()).collect()).collect();
    let log_n: Vec<Vec<u32>> = segments.iter().map(|(tr, _)| tr.iter().map(|t| t[0].len().trailing_zeros()).collect()).collect();
    let mut err = std::ptr::null_mut();
    // sizing pass per segment (proofs_out = NULL), then ONE proving call for all of them
    let mut offs = vec![vec![0usize; NUM_TABLES + 1]; k];
    for s in 0..k {
        check(unsafe { zkm_prove_segment_columns(std::ptr::null_mut(), &cfg, tabs[s].as_ptr(), log_n[s].as_ptr(), pubs[s].as_ptr(), pubs[s].len(),
                                         std::ptr::null_mut(), offs[s].as_mut_ptr(), std::ptr::null_mut(), &mut err) }, err)?;
    }
    let mut blobs: Vec<Vec<u64>> = offs.iter().map(|o| vec![0u64; o[NUM_TABLES]]).collect();
    let mut chals: Vec<Vec<u64>> = (0..k).map(|_| vec![0u64; 2 * config.num_challenges]).collect();
    let seg_ptrs: Vec<*const *const *const u64> = tabs.iter().map(|v| v.as_ptr()).collect();
    let log_ptrs: Vec<*const u32> = log_n.iter().map(|v| v.as_ptr()).collect();
    let pub_ptrs: Vec<*const u64> = pubs.iter().map(|v| v.as_ptr()).collect();
    let pub_lens: Vec<usize> = pubs.iter().map(|v| v.len()).collect();
    let blob_ptrs: Vec<*mut u64> = blobs.iter_mut().map(|v| v.as_mut_ptr()).collect();
    let chal_ptrs: Vec<*mut u64> = chals.iter_mut().map(|v| v.as_mut_ptr()).collect();
    check(unsafe { zkm_prove_segments_columns(ctx, &cfg, k, seg_ptrs.as_ptr(), log_ptrs.as_ptr(), pub_ptrs.as_ptr(), pub_lens.as_ptr(),
                                      blob_ptrs.as_ptr(), chal_ptrs.as_ptr(), &mut err) }, err)?;
    Ok(segments.iter().enumerate().map(|(s, (_, pv))| {
        let stark_proofs: [StarkProofWithMetadata<F, C, D>; NUM_TABLES] =
            core::array::from_fn(|t| stark_proof_from_blob::<F, C, D>(&blobs[s][offs[s][t]..offs[s][t + 1]]));
        let ctl_challenges = GrandProductChallengeSet {
            challenges: (0..config.num_challenges)
                .map(|c| GrandProductChallenge { beta: F::from_canonical_u64(chals[s][2 * c]), gamma: F::from_canonical_u64(chals[s][2 * c + 1]) })
                .collect(),
        };
        AllProof { stark_proofs, ctl_challenges, public_values: pv.clone() }
    }).collect())
}

/// A host trace on its way into HBM BEHIND the proof the context is working on (include/zkm_hip.h "staged traces"): the driver that
/// has segment i + 1's traces while segment i is being proven stages them, then hands `ptr()` to the next prove call of the SAME
/// context -- that proof runs the device-resident path and the 2.2 GB of a 262 x 2^20 trace cross PCIe behind its predecessor
/// (17.05 proofs/s from host memory against 17.15 device-resident, profiles/r06_host_resident.txt).  The columns must stay alive
/// and unchanged until `ready(true)` or drop; pinned memory (zkm_host_alloc, or the Vecs registered once with zkm_host_register)
/// makes `stage` return at once, pageable memory makes it take the time of the upload.
pub struct StagedTrace(*mut zkm_staged);

impl StagedTrace {
    pub fn stage<F: PrimeField64>(ctx: *mut zkm_ctx, cols: &[PolynomialValues<F>], canonical: bool) -> Result<Self> {
        let ptrs = column_ptrs(cols);
        let log_n = cols[0].len().trailing_zeros();
        let mut h = std::ptr::null_mut();
        let mut err = std::ptr::null_mut();
        check(unsafe { zkm_trace_stage_columns(ctx, ptrs.as_ptr(), ptrs.len(), log_n, canonical as i32, &mut h, &mut err) }, err)?;
        Ok(Self(h))
    }
    /// All twelve tables of one segment in ONE call (zkm_segment_stage_columns: one device block, one pair of events); `tables()` gives
    /// the twelve device matrices to pass as that segment's `traces` to zkm_prove_segments.
    pub fn stage_segment<F: PrimeField64>(ctx: *mut zkm_ctx, traces: &[Vec<PolynomialValues<F>>; NUM_TABLES], canonical: bool) -> Result<Self> {
        let cols: Vec<Vec<*const u64>> = traces.iter().map(|t| column_ptrs(t)).collect();
        let tabs: Vec<*const *const u64> = cols.iter().map(|v| v.as_ptr()).collect();
        let log_n: Vec<u32> = traces.iter().map(|t| t[0].len().trailing_zeros()).collect();
        let mut h = std::ptr::null_mut();
        let mut err = std::ptr::null_mut();
        check(unsafe { zkm_segment_stage_columns(ctx, tabs.as_ptr(), log_n.as_ptr(), canonical as i32, &mut h, &mut err) }, err)?;
        Ok(Self(h))
    }
    pub fn tables(&mut self) -> Result<[*const u64; NUM_TABLES]> {
        let mut out = [std::ptr::null(); NUM_TABLES];
        anyhow::ensure!(unsafe { zkm_staged_segment_ptrs(self.0, out.as_mut_ptr()) } == 0, "libzkmhip: not a staged segment");
        Ok(out)
    }
    /// the device matrix (column-major, ncols x 2^log_n), ordered behind the upload on the context's compute stream
    pub fn ptr(&mut self) -> *const u64 {
        unsafe { zkm_staged_ptr(self.0) }
    }
    pub fn ready(&mut self, wait: bool) -> bool {
        unsafe { zkm_staged_ready(self.0, wait as i32) == 1 }
    }
}

impl Drop for StagedTrace {
    fn drop(&mut self) {
        unsafe { zkm_staged_free(self.0) };
    }
}

/// The driver loop the staged path is made for: single-table proofs of a sequence of host traces (the benchmark shape of
/// poseidon_stark.rs:751-816, one transcript per proof), the NEXT trace crossing PCIe behind the CURRENT proof.  Every proof is word for
/// word what `prove_single_table_hip` returns for that trace (tests/test_gpu_large_parity.py holds the staged 2^20-row proof against the
/// oracle's).  The traces' columns should be pinned (zkm_host_alloc) or registered once (zkm_host_register): see INTEGRATION.md section 4.
pub fn prove_single_tables_pipelined_hip<F, C, const D: usize>(
    ctx: *mut zkm_ctx,
    table: Table,
    config: &StarkConfig,
    traces: &[Vec<PolynomialValues<F>>],
    aux_columns: &[PolynomialValues<F>],
    num_ctl_helper_polys: &[usize],
) -> Result<Vec<StarkProofWithMetadata<F, C, D>>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F, Hasher = plonky2::hash::poseidon::PoseidonHash>,
{
    let cfg = zkm_config(config);
    let helpers: Vec<u32> = num_ctl_helper_polys.iter().map(|&x| x as u32).collect();
    let aux: Vec<u64> = aux_columns.iter().flat_map(|c| c.values.iter().map(|x| x.to_canonical_u64())).collect();
    let mut proofs = Vec::with_capacity(traces.len());
    let mut staged = match traces.first() {
        Some(t) => Some(StagedTrace::stage(ctx, t, false)?),
        None => None,
    };
    for (i, trace) in traces.iter().enumerate() {
        let mut cur = staged.take().expect("staged above");
        if let Some(next) = traces.get(i + 1) {
            staged = Some(StagedTrace::stage(ctx, next, false)?);          // returns at once: the copy streams work behind the proof below
        }
        let log_n = trace[0].len().trailing_zeros();
        let words = unsafe { zkm_proof_words(&cfg, log_n, trace.len(), aux_columns.len(), helpers.len()) };
        anyhow::ensure!(words != 0, "libzkmhip: unsupported StarkConfig");
        let mut blob = vec![0u64; words];
        let mut ch = Challenger::<F, C::Hasher>::new().to_zkm();
        let mut err = std::ptr::null_mut();
        check(unsafe { zkm_prove_single_table(ctx, zkm_table_id(table), &cfg, cur.ptr(), trace.len(), log_n, std::ptr::null(), aux.as_ptr(),
                                              aux_columns.len(), helpers.as_ptr(), helpers.len(), &mut ch, blob.as_mut_ptr(), &mut err) }, err)?;
        proofs.push(stark_proof_from_blob::<F, C, D>(&blob));
        drop(cur);                                                           // zkm_staged_free: the block goes back to the context's allocator
    }
    Ok(proofs)
}

/// A pool of contexts over the GPUs of one node, owned by the ONE process that drives all segments of a program -- the shape of the
/// reference's driver (prover/examples/utils/src/utils.rs:57-68 `prove_single_seg_common`, :105-133 `prove_multi_seg_common`: a loop
/// of `prove_with_traces` calls).  `contexts_per_device` worker threads per device live inside the library (include/zkm_hip.h
/// zkm_pool_*); nothing here is `Send` across the FFI: the pool is used from the thread that holds it, one call at a time.
pub struct HipPool(*mut zkm_pool);

impl HipPool {
    pub fn new(devices: &[i32], contexts_per_device: usize) -> Result<Self> {
        let mut pool = std::ptr::null_mut();
        let mut err = std::ptr::null_mut();
        check(unsafe { zkm_pool_create(devices.as_ptr(), devices.len(), contexts_per_device, &mut pool, &mut err) }, err)?;
        Ok(Self(pool))
    }
    pub fn workers(&self) -> usize {
        unsafe { zkm_pool_workers(self.0) }
    }
    pub fn set_tuning(&mut self, key: &std::ffi::CStr, value: u64) -> Result<()> {
        let mut err = std::ptr::null_mut();
        check(unsafe { zkm_pool_set_tuning(self.0, key.as_ptr(), value, &mut err) }, err)
    }
}

impl Drop for HipPool {
    fn drop(&mut self) {
        unsafe { zkm_pool_destroy(self.0) };
    }
}

/// All segments of a program over all GPUs of the pool: the segments are cut into lock-step groups of at most `max_stack` (0 = 8), the
/// pool's workers pull the groups from a queue, and every `AllProof` is word for word what `prove_with_traces_hip` returns for that
/// segment alone, whatever the device count.  The traces are read where `generate_traces` left them (host `Vec`s, one pointer per
/// column): every device of the pool reads host memory.
pub fn prove_segments_multi_hip<F, C, const D: usize>(
    pool: &mut HipPool,
    config: &StarkConfig,
    max_stack: usize,
    segments: &[([Vec<PolynomialValues<F>>; NUM_TABLES], PublicValues)],
) -> Result<Vec<AllProof<F, C, D>>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    let cfg = zkm_config(config);
    let k = segments.len();
    let pubs: Vec<Vec<u64>> = segments.iter().map(|(_, pv)| public_values_words(pv)).collect();
    let cols: Vec<Vec<Vec<*const u64>>> = segments.iter().map(|(tr, _)| tr.iter().map(|t| column_ptrs(t)).collect()).collect();
    let tabs: Vec<Vec<*const *const u64>> = cols.iter().map(|seg| seg.iter().map(|v| v.as_ptr()).collect()).collect();
    let log_n: Vec<Vec<u32>> = segments.iter().map(|(tr, _)| tr.iter().map(|t| t[0].len().trailing_zeros()).collect()).collect();
    let mut err = std::ptr::null_mut();
    let mut offs = vec![vec![0usize; NUM_TABLES + 1]; k];
    for s in 0..k {     // sizing pass per segment (no context needed: proofs_out = NULL)
        check(unsafe { zkm_prove_segment_columns(std::ptr::null_mut(), &cfg, tabs[s].as_ptr(), log_n[s].as_ptr(), pubs[s].as_ptr(), pubs[s].len(),
                                         std::ptr::null_mut(), offs[s].as_mut_ptr(), std::ptr::null_mut(), &mut err) }, err)?;
    }
    let mut blobs: Vec<Vec<u64>> = offs.iter().map(|o| vec![0u64; o[NUM_TABLES]]).collect();
    let mut chals: Vec<Vec<u64>> = (0..k).map(|_| vec![0u64; 2 * config.num_challenges]).collect();
    let seg_ptrs: Vec<*const *const *const u64> = tabs.iter().map(|v| v.as_ptr()).collect();
    let log_ptrs: Vec<*const u32> = log_n.iter().map(|v| v.as_ptr()).collect();
    let pub_ptrs: Vec<*const u64> = pubs.iter().map(|v| v.as_ptr()).collect();
    let pub_lens: Vec<usize> = pubs.iter().map(|v| v.len()).collect();
    let blob_ptrs: Vec<*mut u64> = blobs.iter_mut().map(|v| v.as_mut_ptr()).collect();
    let chal_ptrs: Vec<*mut u64> = chals.iter_mut().map(|v| v.as_mut_ptr()).collect();
    check(unsafe { zkm_pool_prove_segments_columns(pool.0, &cfg, k, max_stack, seg_ptrs.as_ptr(), log_ptrs.as_ptr(), pub_ptrs.as_ptr(),
                                           pub_lens.as_ptr(), blob_ptrs.as_ptr(), chal_ptrs.as_ptr(), &mut err) }, err)?;
    Ok(segments.iter().enumerate().map(|(s, (_, pv))| {
        let stark_proofs: [StarkProofWithMetadata<F, C, D>; NUM_TABLES] =
            core::array::from_fn(|t| stark_proof_from_blob::<F, C, D>(&blobs[s][offs[s][t]..offs[s][t + 1]]));
        let ctl_challenges = GrandProductChallengeSet {
            challenges: (0..config.num_challenges)
                .map(|c| GrandProductChallenge { beta: F::from_canonical_u64(chals[s][2 * c]), gamma: F::from_canonical_u64(chals[s][2 * c + 1]) })
                .collect(),
        };
        AllProof { stark_proofs, ctl_challenges, public_values: pv.clone() }
    }).collect())
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
    let log_n = trace_poly_values[0].len().trailing_zeros();
    let helpers: Vec<u32> = num_ctl_helper_polys.iter().map(|&x| x as u32).collect();
    let words = unsafe { zkm_proof_words(&cfg, log_n, trace_poly_values.len(), aux_columns.len(), helpers.len()) };
    anyhow::ensure!(words != 0, "libzkmhip: unsupported StarkConfig");
    let mut err = std::ptr::null_mut();
    // the trace commitment of prover.rs:154-163 from the column pointers (no flatten); the few auxiliary columns go over as one block
    let tcols = column_ptrs(trace_poly_values);
    let mut trace_batch: *mut zkm_batch = std::ptr::null_mut();
    check(unsafe { zkm_batch_commit_columns(ctx, tcols.as_ptr(), tcols.len(), log_n, 1, cfg.rate_bits, cfg.cap_height, &mut trace_batch,
                                            &mut err) }, err)?;
    let aux: Vec<u64> = aux_columns.iter().flat_map(|c| c.values.iter().map(|x| x.to_canonical_u64())).collect();
    let mut blob = vec![0u64; words];
    let mut ch = challenger.to_zkm();
    let rc = unsafe { zkm_prove_single_table(ctx, zkm_table_id(table), &cfg, std::ptr::null(), trace_poly_values.len(), log_n, trace_batch,
                                             aux.as_ptr(), aux_columns.len(), helpers.as_ptr(), helpers.len(), &mut ch, blob.as_mut_ptr(),
                                             &mut err) };
    unsafe { zkm_batch_free(trace_batch) };
    check(rc, err)?;
    challenger.set_from_zkm(&ch);   // (the library hands the transcript back only when the proof exists)
    Ok(stark_proof_from_blob::<F, C, D>(&blob))
}
