//! challenger_hip.rs -- plonky2 `Challenger<F, PoseidonHash>` <-> `zkm_challenger` (include/zkm_hip.h).
//!
//! Lives in the plonky2 fork next to `iop/challenger.rs` (the three fields are `pub(crate)` there):
//!     sponge_state: H::Permutation, input_buffer: Vec<F>, output_buffer: Vec<F>
//! The prover threads ONE transcript through all tables (prover/src/prover.rs:182-190, 466, 524-527, 588-591, 610),
//! so the conversion must be lossless in both directions, including pending inputs and unread outputs:
//!   * `in_buf[..n_in]`   = input_buffer in push order;
//!   * `out_buf[..n_out]` = output_buffer in Vec order -- `get_challenge` pops from the BACK, i.e. out_buf[n_out - 1] first.
//! NOT COMPILED in the build image (no cargo / rustc there).
use crate::field::types::{Field, PrimeField64};
use crate::hash::hash_types::RichField;
use crate::hash::poseidon::PoseidonHash;
use crate::hip::sys::zkm_challenger;
use crate::iop::challenger::Challenger;
use crate::hash::hashing::PlonkyPermutation;

impl<F: RichField> Challenger<F, PoseidonHash> {
    pub fn to_zkm(&self) -> zkm_challenger {
        let mut c = zkm_challenger::default();
        for (d, s) in c.state.iter_mut().zip(self.sponge_state.as_ref()) {
            *d = s.to_canonical_u64();
        }
        assert!(self.input_buffer.len() <= 8 && self.output_buffer.len() <= 8);
        for (d, s) in c.in_buf.iter_mut().zip(&self.input_buffer) {
            *d = s.to_canonical_u64();
        }
        for (d, s) in c.out_buf.iter_mut().zip(&self.output_buffer) {
            *d = s.to_canonical_u64();
        }
        c.n_in = self.input_buffer.len() as u32;
        c.n_out = self.output_buffer.len() as u32;
        c
    }

    pub fn set_from_zkm(&mut self, c: &zkm_challenger) {
        let state: Vec<F> = c.state.iter().map(|&x| F::from_canonical_u64(x)).collect();
        self.sponge_state.set_from_slice(&state, 0);
        self.input_buffer = c.in_buf[..c.n_in as usize].iter().map(|&x| F::from_canonical_u64(x)).collect();
        self.output_buffer = c.out_buf[..c.n_out as usize].iter().map(|&x| F::from_canonical_u64(x)).collect();
    }
}
